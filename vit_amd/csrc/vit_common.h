// Shared device/host helpers for the gfx950 ViT kernels. CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vit_amd.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// error plumbing (vit_capi.hip)
int vit_set_error(int code, const char* fmt, ...);
#define VIT_CHECK_HIP(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) return vit_set_error(VIT_E_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
#define VIT_REQUIRE(cond, ...)                                      \
    do {                                                            \
        if (!(cond)) return vit_set_error(VIT_E_INVALID, __VA_ARGS__); \
    } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- activations (fp32 math) ----
__device__ __forceinline__ float act_gelu_erf(float x) {
    // exact GELU: 0.5 x (1 + erf(x / sqrt 2))   (configuration_vit.py:54, ACT2FN["gelu"])
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float act_quick_gelu(float x) {
    // x * sigmoid(1.702 x)   (configuration_clip.py:105, ACT2FN["quick_gelu"])
    return x / (1.0f + __expf(-1.702f * x));
}

// GELU of the bf16 kernels, on two elements at once: in the persistent GEMM's epilogue the activation is the
// exposed cost (128 elements per lane and tile): no transcendental, and the multiplies / FMAs
// are written on float2 so they become v_pk_mul_f32 / v_pk_fma_f32 (2 elements per issue).
//   erf(z) = z h(z^2) with h a degree-8 minimax-style fit on z^2 in [0, 9] (Chebyshev-node least
//   squares), z^2 clamped to 9 and the product clamped to [-1, 1], so both tails are exact.
// |gelu error| <= 7.7e-5 over all x (largest near |x| = 4, where one bf16 ulp of the result is
// 1.6e-2; <= 2e-5 for |x| <= 2), i.e. below the bf16 rounding of the stored activation.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, float c) { return __builtin_elementwise_fma(a, b, f32x2{c, c}); }
__device__ __forceinline__ f32x2 act_gelu_pk(f32x2 x) {
    // every multiply-add is an explicit fma: the same bits from every kernel that inlines this,
    // whatever the compiler would have contracted
    const f32x2 z = x * 0.70710678118654752440f;
    f32x2 u = z * z;
    u.x = fminf(u.x, 9.0f);
    u.y = fminf(u.y, 9.0f);
    f32x2 p = pk_fma(u, f32x2{4.9182759198629356e-08f, 4.9182759198629356e-08f}, -2.2677306787954876e-06f);
    p = pk_fma(p, u, 4.6147291868692264e-05f);
    p = pk_fma(p, u, -0.0005535572418011725f);
    p = pk_fma(p, u, 0.004437862429767847f);
    p = pk_fma(p, u, -0.02564961276948452f);
    p = pk_fma(p, u, 0.11186250299215317f);
    p = pk_fma(p, u, -0.3758186101913452f);
    p = pk_fma(p, u, 1.1283628940582275f);
    f32x2 s = z * p;
    s.x = __builtin_amdgcn_fmed3f(s.x, -1.0f, 1.0f);
    s.y = __builtin_amdgcn_fmed3f(s.y, -1.0f, 1.0f);
    const f32x2 t = x * 0.5f;
    return __builtin_elementwise_fma(t, s, t);
}
// the same function on one element (generic GEMM kernel): identical bits to act_gelu_pk
__device__ __forceinline__ float act_gelu_poly(float x) { return act_gelu_pk(f32x2{x, x}).x; }

// gelu'(x) = Phi(x) + x phi(x), Phi(-|x|) = erfc(|x|/sqrt2)/2 from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7;
// backward of ViTMLP's activation)
__device__ __forceinline__ float act_dgelu_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752440f, 1.0f));
    const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.4426950408889634f));
    float p = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    p = fmaf(p, t, 0.5f * 1.421413741f);
    p = fmaf(p, t, 0.5f * -0.284496736f);
    p = fmaf(p, t, 0.5f * 0.254829592f);
    p = p * t * e;                                  // 0.5 erfc(|x|/sqrt2) = Phi(-|x|)
    const float cdf = x > 0.f ? 1.0f - p : p;
    return fmaf(x * e, 0.3989422804014327f, cdf);
}
__device__ __forceinline__ float act_dquick_gelu(float x) {
    const float sg = 1.0f / (1.0f + __expf(-1.702f * x));
    return sg * (1.0f + 1.702f * x * (1.0f - sg));
}
// multiplier of the backward epilogues: d act(aux) / d aux
template <int EPI>
__device__ __forceinline__ float act_grad(float aux) {
    if constexpr (EPI == VIT_EPI_DGELU) return act_dgelu_as(aux);
    else return act_dquick_gelu(aux);
}

// activations of the bf16 kernels (fp32 parity mode keeps erff)
template <int EPI>
__device__ __forceinline__ float apply_act_bf16(float v) {
    if constexpr (EPI == VIT_EPI_GELU) return act_gelu_poly(v);
    if constexpr (EPI == VIT_EPI_QUICK_GELU) return act_quick_gelu(v);
    return v;
}

template <int EPI>
__device__ __forceinline__ float apply_act(float v) {
    if constexpr (EPI == VIT_EPI_GELU) return act_gelu_erf(v);
    if constexpr (EPI == VIT_EPI_QUICK_GELU) return act_quick_gelu(v);
    return v;
}

// max over the two lane halves (lane l and lane l ^ 32) without the LDS round trip of a shuffle
__device__ __forceinline__ float max_across_halves(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
}

// wave-wide sum over 64 lanes
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// internal launchers (one per .hip file), all async on `stream`
int launch_layernorm(const void* x, const float* g, const float* b, void* y, int rows, int dim, int64_t xs,
                     int64_t ys, float eps, int dtype, hipStream_t stream);

struct PatchGeom {  // im2col-free A operand of the patch-embedding GEMM
    int channels, image, patch, grid;  // grid = image / patch
    int kreal;                         // C*P*P (k >= kreal reads as zero)
};

// optional GEMM outputs used by the training path (bf16 kernels only)
struct GemmExt {
    void* c_pre;   // activation epilogues: also store the pre-activation A W^T + bias here (bf16 [M,N])
    int out_f32;   // plain epilogue: C is fp32 [M,N] (weight gradients)
    int split_k;   // > 1 (fp32 output, persistent kernel only): K is cut into split_k ranges, each
                   // written to its own fp32 slab, then summed in a fixed order (reproducible)
    float* slabs;  // split_k * M * N floats
    // LayerNorm folded into this GEMM (persistent kernel, bias / activation epilogues): A is the
    // UN-normalised row, W carries gamma, and the epilogue computes rstd * acc - (mean * rstd) * c + bias
    // with ln_stats[row] = (mean * rstd, rstd) and c = ln_colsum (see ln_fold_kernel, layernorm.hip)
    const float* ln_stats;   // [M][2]
    const float* ln_colsum;  // [N]
};

// C = epi(A W^T + bias (+res)); `patch` non-null => A is the pixel tensor, rows are patches,
// output rows are remapped past the CLS slot and `pos` (fp32 [N,D]) is added. For the
// VIT_EPI_D* epilogues `res` is the pre-activation the gradient is multiplied with.
int launch_gemm(const void* A, const void* W, const float* bias, const void* res, void* C, int M, int N, int K,
                int epi, int dtype, const PatchGeom* patch, const float* pos, hipStream_t stream,
                const GemmExt* ext = nullptr);
int launch_attention(const void* qkv, void* out, int batch, int seq, int heads, int head_dim, int dtype,
                     hipStream_t stream, float* lse = nullptr);
// training-step kernels (train_ops.hip, attention_bwd.hip); bf16 activations
int launch_attention_bwd(const void* qkv, const void* O, const void* dO, const float* lse, float* delta, void* dqkv,
                         int batch, int seq, int heads, int head_dim, hipStream_t stream);
int launch_transpose_pad(const void* src, void* dst, float* colsum, int R, int C, int Rpad, int group,
                         hipStream_t stream);
int launch_layernorm_bwd(const void* x, const void* dy, const float* gamma, const void* dres, void* dx, float* dgamma,
                         float* dbeta, int rows, int dim, int64_t xs, int64_t dys, int64_t dxs, float eps,
                         hipStream_t stream, float* partials = nullptr);
// layernorm_bwd `partials`: optional scratch of kLnBwdMaxBlocks * 2 * dim floats for a two-stage dgamma / dbeta sum
constexpr int kLnBwdMaxBlocks = 2048;
int launch_embed_grad(const void* dx0, float* dpos, float* dcls, int B, int N, int D, hipStream_t stream);
int launch_sgd(float* master, const float* grad, void* copy_bf16, float lr, int64_t n, hipStream_t stream);
// one slot of launch_multi_convert (train_ops.hip): n contiguous elements of the fp32 master arena at element
// offset src go to byte offset dst of the weight arena as dtype
struct ConvSlot {
    long long src, dst, n;
    int dtype, vec4;  // vec4: offsets and length allow 4-element accesses
};
// one entry of launch_multi_transpose (train_ops.hip): src [R, C] bf16 -> dst [C, R]; blocks
// [first_block, first_block + ceil(R/64) * ceil(C/64)) of the launch belong to it
struct TransposeSlot {
    const void* src;
    void* dst;
    int R, C, first_block, pad;
};
int launch_multi_transpose(const TransposeSlot* table_dev, int n, int total_blocks, hipStream_t stream);
int launch_multi_convert(const float* master, void* arena, const ConvSlot* table_dev, int nslots, hipStream_t stream);
// patch rows gathered into a dense bf16 matrix [rows, kp] (layernorm.hip) and the patch-embedding GEMM on it
// (gemm_bf16_8p.hip): the route for patch sizes the pixel loader of the persistent kernel does not take
int launch_im2col_rows(const void* pixels, void* A, long rows, int kp, const PatchGeom* g, hipStream_t stream);
bool gemm_bf16_8p_patch_eligible(int M, int N, const PatchGeom* g);
bool gemm_bf16_8p_patch_dense_eligible(int M, int N, int K, const PatchGeom* g);
int launch_gemm_bf16_8p_patch_dense(const void* A, const void* W, const float* bias, const float* pos, void* C, int M, int N,
                                    int K, const PatchGeom* g, hipStream_t stream);
// LayerNorm folded into the next GEMM (layernorm.hip): per-row (mean * rstd, rstd), and the weight fold
int launch_ln_stats(const void* x, float* stats, int rows, int dim, int64_t xs, float eps, int dtype, hipStream_t stream);
int launch_ln_fold(const void* W, int64_t pitch, const float* gamma, const float* beta, const float* bias, void* Wg, float* c,
                   float* d, int N, int K, hipStream_t stream);
int launch_convert(const void* src, int sdt, void* dst, int ddt, int64_t n, hipStream_t stream);
int launch_im2col_t(const void* pixels, int pixel_dtype, void* dst, int R, int Rpad, const PatchGeom* g, int kpad,
                    hipStream_t stream);
int launch_cls_rows(const float* cls, const float* pos, void* x, int batch, int seq, int dim, int dtype,
                    hipStream_t stream);
