/*
 * vit_amd.h -- C ABI of libvit_amd.so: MI355X (gfx950) ViT encoder forward path.
 *
 * Boundary note. The reference repository has NO plugin / operator / FFI interface for a
 * vision encoder: its only boundary to a vision model is a network RPC that ships an
 * encoded image file and gets text back --
 *   ollama.Client.generate(model, prompt, images=[b64])      Vlm_coder.py:205, :381
 *   GenerativeModel.generate_content([prompt, Image])        vlm_cad.py:171
 * (SURVEY.md section 8b). The entry points below are therefore what a local encoder in
 * front of those call sites binds: tensors in, embeddings out. Each kernel entry point
 * cites the definition it computes (transformers 5.15.0 `models/vit/modeling_vit.py` =
 * "MV", `models/clip/modeling_clip.py` = "MC"), which is the specification SURVEY.md
 * section 8a pins this build to.
 *
 * Conventions
 *  - Plain C: pointers and sizes only. `stream` is a hipStream_t passed as void*
 *    (NULL = the null stream). Every pointer named *_dev / documented "device" is device
 *    memory owned by the CALLER, valid until the stream has executed the call.
 *  - All calls return 0 on success or a negative VIT_E_* code; vit_last_error() returns a
 *    thread-local human-readable message for the last failure. No exceptions cross the ABI.
 *  - Launches are asynchronous on `stream`. Nothing on the hot path (vit_forward and the
 *    per-kernel entry points) allocates, frees or synchronises.
 *  - A vit_ctx is bound to the device that was current at vit_create and is not
 *    thread-safe: one context per device / stream.
 *  - dtype codes: VIT_F32 = 0, VIT_BF16 = 1. In VIT_BF16 mode matrices and activations are
 *    bf16, all accumulation / LayerNorm statistics / softmax are fp32, and vectors (biases,
 *    LayerNorm affine, position / class embeddings) stay fp32.
 */
#ifndef VIT_AMD_H
#define VIT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIT_F32 0
#define VIT_BF16 1

#define VIT_VARIANT_VIT 0  /* HF ViTModel semantics: final LayerNorm over all tokens (MV:385) */
#define VIT_VARIANT_CLIP 1 /* HF CLIPVisionModel: pre_layrnorm, un-normed hidden, post_layernorm(CLS) (MC:641-656) */

#define VIT_ACT_GELU 0       /* exact erf GELU (configuration_vit.py:54) */
#define VIT_ACT_QUICK_GELU 1 /* x*sigmoid(1.702x) (configuration_clip.py:105) */

/* epilogue codes of vit_gemm_bias_act */
#define VIT_EPI_BIAS 0       /* C = A W^T + bias                       (q/k/v: MV:216-218) */
#define VIT_EPI_GELU 1       /* C = gelu(A W^T + bias)                 (fc1:  MV:250-251) */
#define VIT_EPI_QUICK_GELU 2 /* C = quick_gelu(A W^T + bias)           (CLIP fc1) */
#define VIT_EPI_RESIDUAL 3   /* C = A W^T + bias + residual            (o_proj+res MV:236,277; fc2+res MV:252,284) */
#define VIT_EPI_DGELU 4      /* C = (A W^T) * gelu'(aux)               (backward of MV:251; aux = fc1 pre-activation) */
#define VIT_EPI_DQUICK_GELU 5 /* C = (A W^T) * quick_gelu'(aux)        (CLIP) */

#define VIT_OK 0
#define VIT_E_INVALID (-1)   /* bad argument / unsupported shape */
#define VIT_E_STATE (-2)     /* weights missing, wrong device, ... */
#define VIT_E_HIP (-3)       /* a HIP runtime call failed */
#define VIT_E_NOMEM (-4)     /* workspace too small / allocation failed */

typedef struct vit_ctx vit_ctx;

/* Field names follow ViTConfig (configuration_vit.py:50-65). */
typedef struct vit_config {
    int32_t hidden_size;         /* D */
    int32_t num_hidden_layers;   /* L */
    int32_t num_attention_heads; /* H */
    int32_t intermediate_size;   /* F */
    int32_t image_size;          /* square input, pixels */
    int32_t patch_size;          /* P */
    int32_t num_channels;        /* C */
    int32_t hidden_act;          /* VIT_ACT_* */
    int32_t variant;             /* VIT_VARIANT_* */
    int32_t qkv_bias;            /* 0/1 */
    int32_t dtype;               /* compute dtype VIT_F32 | VIT_BF16 */
    float layer_norm_eps;
} vit_config;

/* ---- library ---- */
const char* vit_last_error(void);
/* ABI version: bumped when an entry point is added or a signature changes. History: 1 = inference path and
 * per-kernel entry points; 2 = + training step, folded LayerNorm, heads, preprocessing, position interpolation,
 * vit_forward_static; 3 = vit_backward_stage accepts any stash that a vit_forward_train of this context filled
 * (several in flight), few-tile GEMM route. The library returns the VIT_ABI_VERSION it was built with. */
#define VIT_ABI_VERSION 3
int vit_abi_version(void);

/* ---- context: weights live in one arena allocated here (NOT on the hot path) ---- */
int vit_create(const vit_config* cfg, vit_ctx** out);
int vit_destroy(vit_ctx* ctx);

/* Copy/convert one tensor of an HF state dict into the arena. `name` is the HF key
 * (SURVEY.md 8a row a11), e.g. "layers.3.attention.q_proj.weight" (ViT) or
 * "encoder.layers.3.self_attn.q_proj.weight" (CLIP). q/k/v are packed into one [3D, D]
 * matrix so the three projections (MV:216-218) run as one GEMM. `src_dev` is device
 * memory of `src_dtype` (VIT_F32 or VIT_BF16) holding `numel` contiguous elements. */
int vit_load_weight(vit_ctx* ctx, const char* name, const void* src_dev, int src_dtype, int64_t numel,
                    void* stream);
/* Number of expected tensors not loaded yet (0 = ready). If `first_missing` is non-NULL
 * it receives a pointer to the name of one missing tensor (owned by the context). */
int vit_missing_weights(vit_ctx* ctx, const char** first_missing);

/* Bytes of caller-provided scratch vit_forward needs for `batch` images. */
size_t vit_workspace_bytes(const vit_ctx* ctx, int batch);

/* Whole encoder forward: ViTModel.forward (MV:356-388) / CLIPVisionModel.forward
 * (MC:641-656).
 *   pixels_dev   [batch, C, image, image], dtype pixel_dtype (VIT_F32 or VIT_BF16)
 *   hidden_out   [batch, N, D] in the context dtype  (last_hidden_state)
 *   pooled_out   [batch, D]    in the context dtype  (ViT: CLS row of hidden_out;
 *                CLIP: post_layernorm of the CLS row). May be NULL.
 *   workspace    >= vit_workspace_bytes(ctx, batch) bytes, 256-byte aligned.
 * VIT_BF16 contexts run batches of >= 2 x 6304 token rows (2 x 32 images at 197 tokens) as two contiguous chunks (5 : 3 of the batch: DESIGN.md 3.5), one on `stream` and one on
 * a stream the context owns (forked from and joined back into `stream` with events, so the call
 * keeps its stream semantics and stays graph-capturable): one half's kernels occupy the CUs the
 * other half's last round of GEMM tiles leaves idle (+5 % images/s at ViT-B/16 batch 256).
 * Environment VIT_AMD_STREAMS=1..4 at vit_create overrides the number of streams.
 * VIT_BF16 contexts also fold LN1 / LN2 into the q/k/v and fc1 GEMMs (vit_gemm_ln_folded below): the context
 * holds gamma-scaled copies of those weights and rebuilds them on `stream` at the next vit_forward after a
 * vit_load_weight or vit_sgd_step. Every batch size gives the same bits for the same image. Environment
 * VIT_AMD_FUSE_LN=0 at vit_create keeps LayerNorm a separate kernel. */
int vit_forward(vit_ctx* ctx, const void* pixels_dev, int pixel_dtype, int batch, void* hidden_out,
                void* pooled_out, void* workspace, size_t workspace_bytes, void* stream);

/* vit_forward for callers that cannot allocate or hold a workspace (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc). It has NO counterpart in the reference (which has no
 * encoder at all, see the boundary note above). The workspace is a file-scope device array inside the
 * library, VIT_STATIC_SCRATCH_BYTES per GPU, held until the process ends; the call and everything under it
 * allocates nothing. Same arguments, results (bit for bit) and errors as vit_forward, minus the workspace;
 * batches whose vit_workspace_bytes(ctx, batch) exceeds VIT_STATIC_SCRATCH_BYTES are rejected with
 * VIT_E_NOMEM (ViT-B/16 bf16: up to 10 images per call). All calls on one GPU share that GPU's array:
 * calls that overlap in time on one GPU (other streams, threads or contexts) race -- serialise them. */
#define VIT_STATIC_SCRATCH_BYTES (32u << 20)
int vit_forward_static(vit_ctx* ctx, const void* pixels_dev, int pixel_dtype, int batch, void* hidden_out,
                       void* pooled_out, void* stream);

/* ---- per-kernel entry points (each unit-testable against oracle/vit_oracle.py) ---- */

/* nn.LayerNorm over the last dim, biased variance, fp32 statistics (MV:261-262,274,281).
 * x,y: [rows, dim] of `dtype` with the given row strides in elements; gamma,beta: fp32. */
int vit_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, int rows, int dim,
                      int64_t x_row_stride, int64_t y_row_stride, float eps, int dtype, void* stream);

/* nn.Linear with fused epilogue: C[M,N] = epi(A[M,K] W[N,K]^T + bias[N] (+ residual[M,N])).
 * A, W, residual, C of `dtype`, row-major contiguous; bias fp32 or NULL. K % 8 == 0 and
 * N % 8 == 0 (bf16) / K % 4 == 0 and N % 4 == 0 (fp32). (MV:216-218, 236, 250-252) */
int vit_gemm_bias_act(const void* A, const void* W, const float* bias, const void* residual, void* C, int M,
                      int N, int K, int epilogue, int dtype, void* stream);

/* Fused multi-head self-attention, softmax(Q K^T * head_dim^-0.5) V with fp32 online
 * softmax and no materialised score matrix (eager_attention_forward, MV:164-189; no mask,
 * not causal). qkv: [batch*seq, 3*heads*head_dim] rows laid out q|k|v (the packed QKV GEMM
 * output); out: [batch*seq, heads*head_dim] (= attn_output after transpose(1,2).reshape,
 * MV:187,235). head_dim % 16 == 0, head_dim <= 128. */
int vit_attention_fwd(const void* qkv, void* out, int batch, int seq, int heads, int head_dim, int dtype,
                      void* stream);

/* Patch embedding as an im2col-free strided GEMM + CLS prepend + position add
 * (ViTPatchEmbeddings.forward MV:62-69, ViTEmbeddings.forward MV:129-161; CLIP MC:202-218).
 *   pixels [batch, C, image, image] of `dtype`; weight [D, Kpad] of `dtype` where row d holds
 *   the Conv2d kernel flattened (c,i,j) and zero-padded to Kpad = vit_patch_kpad(C,P,dtype);
 *   bias fp32 [D] or NULL; cls fp32 [D]; pos fp32 [N, D]; x_out [batch, N, D] of `dtype`. */
int vit_patch_embed_fwd(const void* pixels, const void* weight, const float* bias, const float* cls,
                        const float* pos, void* x_out, int batch, int channels, int image, int patch,
                        int dim, int dtype, void* stream);
int vit_patch_kpad(int channels, int patch, int dtype);

/* ---- LayerNorm folded into the Linear that follows it (bf16). vit_forward computes LN1 -> q/k/v projection
 * (MV:274 feeding MV:216-218) and LN2 -> fc1 (MV:281 feeding MV:250) this way: with xh = (x - mean) * rstd,
 *   (xh * gamma + beta) W^T + b  =  rstd * (x Wg^T) - (mean * rstd) * c + d,
 *   Wg[n,k] = W[n,k] * gamma[k],  c[n] = sum_k Wg[n,k],  d[n] = b[n] + sum_k W[n,k] * beta[k],
 * so the normalised activations are never written: a statistics pass over x replaces the LayerNorm kernel and
 * the GEMM's epilogue applies the two row terms. The three steps, exposed for parity tests and timing:
 *  vit_layernorm_stats: stats[row] = (mean * rstd, rstd), biased variance, fp32 (same statistics as
 *    vit_layernorm_fwd); x is [rows, dim] bf16 with the given row stride (elements).
 *  vit_layernorm_fold: weight [out_dim, in_dim] bf16 -> weight_folded (same shape, bf16), colsum[out_dim] (sum of
 *    the ROUNDED folded row, so the mean term cancels exactly what the MFMA multiplies), bias_folded[out_dim];
 *    bias may be NULL.
 *  vit_gemm_ln_folded: C[M,N] = act(rstd * (x Wg^T) - mean * rstd * c + d), epilogue VIT_EPI_BIAS / _GELU /
 *    _QUICK_GELU. Same result as vit_layernorm_fwd followed by vit_gemm_bias_act up to bf16 rounding points
 *    (the normalised row is not rounded to bf16 on the way). VIT_AMD_FUSE_LN=0 in the environment at vit_create
 *    makes vit_forward use the two separate kernels instead. */
int vit_layernorm_stats(const void* x, float* stats, int rows, int dim, int64_t x_row_stride, float eps, int dtype,
                        void* stream);
int vit_layernorm_fold(const void* weight, const float* gamma, const float* beta, const float* bias, void* weight_folded,
                       float* colsum, float* bias_folded, int out_dim, int in_dim, void* stream);
int vit_gemm_ln_folded(const void* x, const void* weight_folded, const float* bias_folded, const float* colsum,
                       const float* stats, void* C, int M, int N, int K, int epilogue, void* stream);

/* ---- The statistics of the NEXT folded LayerNorm from the residual GEMM that writes its input (bf16). In vit_forward
 * the o_proj and fc2 GEMMs (MV:236 + :277, MV:252 + :284) write the rows that LN2 / the next layer's LN1 (MV:281, :274)
 * normalise; their epilogue also leaves, per 64-column group g and row r, the pair (sum, sum of squares) of the bf16 values
 * it stored: row_part[(g * M + r) * 2 .. + 1], N / 64 slabs of M pairs. vit_row_stats_finalize adds the slabs left to right
 * and writes stats[r] = (mean * rstd, rstd) -- bit for bit what vit_layernorm_stats computes from the stored rows (both
 * run one instruction sequence, csrc/vit_common.h row16_stats), so whether a forward took this route (big batches) or the
 * stand-alone pass (small ones, VIT_AMD_ROW_STATS=0) cannot be seen in its output. Exposed for parity tests and timing.
 *  vit_gemm_residual_row_stats: C = A W^T + bias + residual as vit_gemm_bias_act(VIT_EPI_RESIDUAL), plus row_part; only
 *    shapes the persistent kernel takes (N % 64 == 0, K % 128 == 0, K >= 256, M * N >= 4 Mi), VIT_E_INVALID otherwise. */
int vit_gemm_residual_row_stats(const void* A, const void* W, const float* bias, const void* residual, void* C,
                                float* row_part, int M, int N, int K, void* stream);
int vit_row_stats_finalize(const float* row_part, float* stats, int rows, int dim, float eps, void* stream);

/* ---- training-step kernels (VIT_BF16 activations, fp32 statistics and parameter gradients).
 * The reference has no training code at all; these are the backward
 * counterparts of the entry points above, needed by BASELINE.json configs[4]. ---- */

/* vit_gemm_bias_act with the training outputs. `residual_or_aux`: residual for
 * VIT_EPI_RESIDUAL, the fc1 pre-activation for VIT_EPI_DGELU / VIT_EPI_DQUICK_GELU (backward of
 * MV:251: C = (A W^T) * act'(aux)). `C_pre` (may be NULL; activation epilogues only) also
 * receives the pre-activation A W^T + bias. `out_f32` != 0 (VIT_EPI_BIAS only): C is fp32 --
 * used for weight gradients dW = dY^T X with A = dY^T, W = X^T. bf16 mode only. */
int vit_gemm_ex(const void* A, const void* W, const float* bias, const void* residual_or_aux, void* C, void* C_pre,
                int M, int N, int K, int epilogue, int out_f32, void* stream);

/* vit_attention_fwd that also writes lse [batch, heads, seq] fp32: log2 of the softmax
 * denominator of the scaled scores (P = exp2(S * dh^-0.5 * log2 e - lse)). */
int vit_attention_fwd_train(const void* qkv, void* out, float* lse, int batch, int seq, int heads, int head_dim,
                            void* stream);
/* Backward of eager_attention_forward (MV:164-189): dqkv [batch*seq, 3*heads*head_dim] from qkv,
 * the forward output `out`, its gradient `d_out` and `lse`; `delta_scratch` is fp32
 * [batch, heads, seq] (unused by the one-pass kernel that takes head_dim 64 at 97..224 tokens). Bitwise
 * reproducible (no atomics). */
int vit_attention_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, float* delta_scratch,
                      void* dqkv, int batch, int seq, int heads, int head_dim, void* stream);

/* Backward of nn.LayerNorm (MV:261-262): dx = LN'(x)^T dy (+ d_residual if non-NULL, the
 * gradient of the skip branch, MV:277,284); dgamma / dbeta are ACCUMULATED (+=, fp32 atomics). */
int vit_layernorm_bwd(const void* x, const void* dy, const float* gamma, const void* d_residual, void* dx,
                      float* dgamma, float* dbeta, int rows, int dim, float eps, void* stream);

/* dst[C, Rpad] = src[R, C]^T (bf16), columns R..Rpad zero; colsum (may be NULL) += column sums of
 * src in fp32 (bias gradients). group > 0: src is [B, group+1, C] and the first row of every
 * group is skipped (token rows without CLS). C % 8 == 0, Rpad % 8 == 0. */
int vit_transpose_pad(const void* src, void* dst, float* colsum, int R, int C, int Rpad, int group, void* stream);

/* Weight / bias gradient of nn.Linear (backward of MV:216-218, 236, 250-252): dW[out, in] (fp32) =
 * dY^T X over `tokens` rows, db[out] (fp32, may be NULL) += column sums of dY. dY [tokens, out] and
 * X [tokens, in] are bf16 row-major exactly as the backward / the stash hold them: no transposed
 * copies. splits > 1 cuts the token range into that many parts, one fp32 slab each in `slabs`
 * (splits * out * in floats), summed in a fixed order. out % 8 == 0, in % 8 == 0. */
int vit_linear_wgrad(const void* dY, const void* X, float* dW, float* db, float* slabs, int tokens, int out_features,
                     int in_features, int splits, void* stream);

/* ---- whole training step (VIT_BF16 contexts). Buffers are caller-owned device memory:
 *   stash      vit_train_stash_bytes(ctx, batch): activations the backward re-reads
 *   workspace  vit_train_workspace_bytes(ctx, batch): scratch of forward_train and backward
 *   grads      vit_grad_bytes(ctx): fp32 gradients of every parameter; vit_grad_slot gives the
 *              byte offset / shape / row pitch of a tensor by its HF key name, and the tensors of
 *              transformer layer l are one contiguous range (vit_grad_layer_range) = one
 *              all-reduce bucket. ---- */
size_t vit_train_stash_bytes(const vit_ctx* ctx, int batch);
size_t vit_train_workspace_bytes(const vit_ctx* ctx, int batch);
size_t vit_grad_bytes(const vit_ctx* ctx);
int vit_grad_slot(vit_ctx* ctx, const char* name, size_t* offset_bytes, int64_t* rows, int64_t* cols, int64_t* pitch);
int vit_grad_layer_range(vit_ctx* ctx, int layer, size_t* offset_bytes, size_t* bytes);
/* fp32 master weights (from the loaded weights) + transposed weights for the data-gradient GEMMs.
 * Allocates; call once after loading, not on the hot path. */
int vit_train_init(vit_ctx* ctx, void* stream);
int vit_train_refresh_transposes(vit_ctx* ctx, void* stream);
/* vit_forward that also fills `stash` (ViTModel.forward MV:356-388 with every intermediate kept).
 * Like vit_forward it runs big enough batches as two chunks on two streams. */
int vit_forward_train(vit_ctx* ctx, const void* pixels_dev, int pixel_dtype, int batch, void* hidden_out,
                      void* pooled_out, void* stash, size_t stash_bytes, void* workspace, size_t workspace_bytes,
                      void* stream);
/* Backward in stages, in order on one stream: stage 0 = head (needs d_hidden [batch, N, D] bf16, the
 * gradient of last_hidden_state; CLIP contexts also take d_pooled [batch, D] or NULL; ViT contexts
 * fold d_pooled into the CLS rows of d_hidden themselves), stages 1..L = layers L-1..0,
 * stage L+1 = embeddings (needs the pixels). Stage 0 zeroes `grads`. After stage s the gradients
 * written by that stage are final, so the caller may all-reduce them while later stages run.
 * `stash` must be one that a successful vit_forward_train of THIS context filled, with the same `batch` (and, for
 * the embedding stage, the same pixel dtype); several stashes may be in flight (micro-batches): VIT_E_INVALID otherwise. */
int vit_backward_stage(vit_ctx* ctx, int stage, const void* pixels_dev, int pixel_dtype, int batch,
                       const void* d_hidden, const void* d_pooled, const void* stash, void* grads, void* workspace,
                       size_t workspace_bytes, void* stream);
/* SGD: master -= lr * grads for every parameter, then refresh the compute copies. */
int vit_sgd_step(vit_ctx* ctx, const void* grads, float lr, void* stream);

/* ---- the steps directly before and after the encoder ---- */

/* Heads on the pooled row. out[b, :] = act(W x_b + bias), x_b = x + b * x_row_stride (elements), so
 * the CLS row of last_hidden_state [batch, N, D] is read in place with x_row_stride = N * D.
 *   VIT_HEAD_TANH   + bias: ViTPooler.forward, pooled = tanh(dense(hidden[:, 0]))       (MV:289-301)
 *   VIT_HEAD_LINEAR no bias: CLIPVisionModelWithProjection.visual_projection(pooled)    (MC:907, :950)
 * x, weight [out_dim, in_dim], out [batch, out_dim] of `dtype`; bias fp32 or NULL; in_dim and
 * x_row_stride multiples of 8 (bf16) / 4 (fp32). */
#define VIT_HEAD_LINEAR 0
#define VIT_HEAD_TANH 1
int vit_head_fwd(const void* x, int64_t x_row_stride, const void* weight, const float* bias, void* out, int batch,
                 int in_dim, int out_dim, int activation, int dtype, void* stream);

/* Image preprocessing: decoded uint8 images [batch, in_height, in_width, channels] (HWC, all the same
 * size) -> pixel_values [batch, channels, H, W] fp32 or bf16. This is what the remote model servers
 * do with the file the reference sends them (Vlm_coder.py:24-26,:205,:381; vlm_cad.py:40,:171);
 * the arithmetic follows transformers' PIL pipeline bit for bit: resize (Pillow ImagingResample,
 * 8-bit fixed point, horizontal then vertical) -> center_crop (zero padding when the image is
 * smaller) -> rescale -> normalize (image_processing_backends.py:619-653, image_transforms.py:89-126,
 * :313-381, :384-442, :445-526). `resample` takes PIL's filter codes. */
#define VIT_RESAMPLE_LANCZOS 1
#define VIT_RESAMPLE_BILINEAR 2
#define VIT_RESAMPLE_BICUBIC 3
#define VIT_RESAMPLE_BOX 4
#define VIT_RESAMPLE_HAMMING 5
typedef struct vit_preproc vit_preproc;
typedef struct vit_preproc_config {
    int32_t in_height, in_width, channels; /* channels: 1 or 3 */
    int32_t resize_height, resize_width;   /* 0, 0 = do_resize False */
    int32_t crop_height, crop_width;       /* 0, 0 = do_center_crop False */
    int32_t resample;                      /* VIT_RESAMPLE_* */
    int32_t do_rescale, do_normalize;
    double rescale_factor;                 /* 1/255 */
    float image_mean[4], image_std[4];
} vit_preproc_config;
/* Builds the tap tables for one input geometry and uploads them (allocates; not on the hot path). */
int vit_preproc_create(const vit_preproc_config* cfg, vit_preproc** out);
int vit_preproc_destroy(vit_preproc* plan);
int vit_preproc_output_size(const vit_preproc* plan, int* height, int* width);
size_t vit_preproc_workspace_bytes(const vit_preproc* plan, int batch);
/* Two launches on `stream`; `workspace` (16-byte aligned) holds the uint8 intermediate of the
 * horizontal pass. Nothing is allocated. */
int vit_preproc_run(vit_preproc* plan, const void* images_dev, int batch, void* pixel_values_out, int out_dtype,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Position-embedding interpolation for another input resolution:
 * ViTEmbeddings.interpolate_pos_encoding (MV:89-127; CLIP MC:161-200): CLS row kept, the
 * [src_grid, src_grid, dim] patch rows resized to [dst_grid_h, dst_grid_w, dim] with
 * nn.functional.interpolate(mode="bicubic", align_corners=False). src [1 + src_grid^2, dim] and
 * dst [1 + dst_grid_h * dst_grid_w, dim] are fp32 device tables. */
int vit_pos_embed_interpolate(const float* src, float* dst, int dim, int src_grid, int dst_grid_h, int dst_grid_w,
                              void* stream);
/* A second context for inputs of image_size x image_size (a multiple of the patch size): all weights
 * are device-to-device copies of `src`'s, the position table is interpolated -- what
 * model(pixel_values, interpolate_pos_encoding=True) computes on every call (MV:147-148), done once.
 * Allocates like vit_create (not on the hot path). Square resolutions only. */
int vit_clone_resized(vit_ctx* src, int image_size, vit_ctx** out, void* stream);

/* Elementwise dtype conversion (plumbing for weight / pixel ingestion). */
int vit_convert(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t numel, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIT_AMD_H */
