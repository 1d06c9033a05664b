// C ABI of libvit_amd.so (declared in include/vit_amd.h): context / weight arena, the whole
// encoder forward (ViTModel.forward modeling_vit.py:356-388, CLIPVisionModel.forward
// modeling_clip.py:641-656) as a fixed sequence of kernel launches on the caller's stream,
// and thin argument-checking wrappers around the individual kernels.
#include "vit_ctx.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

int launch_gemm_bf16(const void* A, const void* W, const float* bias, const void* res, void* C, int M, int N, int K,
                     int epi, const PatchGeom* patch, const float* pos, hipStream_t stream, const GemmExt* ext);
int launch_gemm_f32(const void* A, const void* W, const float* bias, const void* res, void* C, int M, int N, int K,
                    int epi, const PatchGeom* patch, const float* pos, hipStream_t stream);
int launch_convert_2d(const void* src, int sdt, void* dst, int ddt, int64_t rows, int64_t cols, int64_t dst_pitch,
                      hipStream_t stream);

void vit_train_free(vit_ctx* ctx);  // vit_train.hip

// ---------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

int vit_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int launch_gemm(const void* A, const void* W, const float* bias, const void* res, void* C, int M, int N, int K, int epi,
                int dtype, const PatchGeom* patch, const float* pos, hipStream_t stream, const GemmExt* ext) {
    if (dtype == VIT_BF16) return launch_gemm_bf16(A, W, bias, res, C, M, N, K, epi, patch, pos, stream, ext);
    if (dtype == VIT_F32) {
        if (ext && (ext->c_pre || ext->out_f32) ) return vit_set_error(VIT_E_INVALID, "gemm: training outputs are bf16-mode only");
        if (epi > VIT_EPI_RESIDUAL) return vit_set_error(VIT_E_INVALID, "gemm: backward epilogues are bf16-mode only");
        return launch_gemm_f32(A, W, bias, res, C, M, N, K, epi, patch, pos, stream);
    }
    return vit_set_error(VIT_E_INVALID, "gemm: bad dtype %d", dtype);
}

namespace {

struct ArenaBuilder {
    size_t cur = 0;
    size_t gcur = 0;  // fp32 gradient / master arena, in floats (rows * pitch per tensor, 64-float aligned)
    std::map<std::string, Slot>* slots;
    size_t add(const std::string& name, int64_t rows, int64_t cols, int64_t pitch, int dtype, size_t* goff = nullptr) {
        const size_t off = cur;
        (*slots)[name] = Slot{off, rows, cols, pitch, dtype, false, gcur};
        if (goff) *goff = gcur;
        cur = align256(cur + (size_t)rows * pitch * esize(dtype));
        gcur = (gcur + (size_t)rows * pitch + 63) & ~(size_t)63;
        return off;
    }
    size_t raw(size_t bytes) {  // unnamed zero-initialised region
        const size_t off = cur;
        cur = align256(cur + bytes);
        return off;
    }
};

}  // namespace

// ---------------------------------------------------------------- static scratch (vit_forward_static)
// Workspace of the no-allocation entry point: one file-scope device array per GPU, held until the process ends.
// In host code `scratch` is only a stand-in for the symbol: its device address is asked for once per device in
// vit_create (hipGetSymbolAddress with that device current) and kept in g_scratch_ptr; every use goes through it.
static __device__ __align__(256) unsigned char scratch[VIT_STATIC_SCRATCH_BYTES];
static const int kMaxDevices = 64;
static unsigned char* g_scratch_ptr[kMaxDevices] = {nullptr};
static std::mutex g_scratch_mu;  // vit_create may be called from several threads (one context per device / stream)

static int resolve_static_scratch(int device) {
    if (device < 0 || device >= kMaxDevices) return VIT_OK;  // vit_forward_static reports it when asked for
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[device]) return VIT_OK;
    void* p = nullptr;
    VIT_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    g_scratch_ptr[device] = (unsigned char*)p;
    return VIT_OK;
}
static unsigned char* static_scratch_of(int device) {
    if (device < 0 || device >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    return g_scratch_ptr[device];
}

extern "C" const char* vit_last_error(void) { return g_err; }
extern "C" int vit_abi_version(void) { return VIT_ABI_VERSION; }

extern "C" int vit_patch_kpad(int channels, int patch, int dtype) {
    const int k = channels * patch * patch;
    return dtype == VIT_BF16 ? (k + 7) / 8 * 8 : k;
}

extern "C" int vit_create(const vit_config* cfg, vit_ctx** out) {
    VIT_REQUIRE(cfg != nullptr && out != nullptr, "vit_create: null argument");
    *out = nullptr;
    const vit_config& c = *cfg;
    VIT_REQUIRE(c.dtype == VIT_F32 || c.dtype == VIT_BF16, "vit_create: dtype must be VIT_F32 or VIT_BF16");
    VIT_REQUIRE(c.variant == VIT_VARIANT_VIT || c.variant == VIT_VARIANT_CLIP, "vit_create: bad variant %d", c.variant);
    VIT_REQUIRE(c.hidden_act == VIT_ACT_GELU || c.hidden_act == VIT_ACT_QUICK_GELU, "vit_create: bad hidden_act");
    VIT_REQUIRE(c.hidden_size > 0 && c.num_hidden_layers > 0 && c.num_attention_heads > 0 && c.intermediate_size > 0,
                "vit_create: non-positive size");
    VIT_REQUIRE(c.hidden_size % c.num_attention_heads == 0, "vit_create: hidden_size %% num_attention_heads != 0");
    VIT_REQUIRE(c.patch_size > 0 && c.image_size % c.patch_size == 0, "vit_create: image_size %% patch_size != 0");
    VIT_REQUIRE(c.num_channels > 0, "vit_create: num_channels must be positive");
    const int dh = c.hidden_size / c.num_attention_heads;
    VIT_REQUIRE(dh % 16 == 0 && dh <= 128, "vit_create: head_dim %d unsupported (multiple of 16, <= 128)", dh);
    VIT_REQUIRE(c.hidden_size % 8 == 0 && c.intermediate_size % 8 == 0, "vit_create: D and F must be multiples of 8");
    VIT_REQUIRE(c.hidden_size <= 2048, "vit_create: hidden_size > 2048 unsupported by the LayerNorm kernel");

    vit_ctx* ctx = new vit_ctx();
    ctx->cfg = c;
    if (hipError_t e0 = hipGetDevice(&ctx->device); e0 != hipSuccess) {
        delete ctx;
        return vit_set_error(VIT_E_HIP, "vit_create: hipGetDevice: %s", hipGetErrorString(e0));
    }
    if (int rc = resolve_static_scratch(ctx->device); rc != VIT_OK) {
        delete ctx;
        return rc;
    }
    const int D = c.hidden_size, F = c.intermediate_size, L = c.num_hidden_layers;
    const int g = c.image_size / c.patch_size;
    ctx->N = g * g + 1;
    ctx->dh = dh;
    ctx->kreal = c.num_channels * c.patch_size * c.patch_size;
    ctx->kpad = vit_patch_kpad(c.num_channels, c.patch_size, c.dtype);
    const int T = c.dtype;
    const bool clip = c.variant == VIT_VARIANT_CLIP;

    ArenaBuilder ab;
    ab.slots = &ctx->slots;
    struct LayerOff { size_t ln1g, ln1b, ln2g, ln2b, bqkv, bo, b1, b2, wqkv, wo, w1, w2; };
    std::vector<LayerOff> lo(L);
    ctx->glayers.resize(L);
    ctx->g_pre_g = ctx->g_pre_b = ctx->g_patch_b = 0;
    size_t o_patch_w, o_patch_b, o_cls, o_pos, o_pre_g = 0, o_pre_b = 0, o_fin_g, o_fin_b;

    if (!clip) {
        o_cls = ab.add("embeddings.cls_token", 1, D, D, VIT_F32, &ctx->g_cls);
        o_pos = ab.add("embeddings.position_embeddings", 1, (int64_t)ctx->N * D, (int64_t)ctx->N * D, VIT_F32, &ctx->g_pos);
        o_patch_w = ab.add("embeddings.patch_embeddings.projection.weight", D, ctx->kreal, ctx->kpad, T, &ctx->g_patch_w);
        o_patch_b = ab.add("embeddings.patch_embeddings.projection.bias", 1, D, D, VIT_F32, &ctx->g_patch_b);
    } else {
        o_cls = ab.add("embeddings.class_embedding", 1, D, D, VIT_F32, &ctx->g_cls);
        o_pos = ab.add("embeddings.position_embedding.weight", 1, (int64_t)ctx->N * D, (int64_t)ctx->N * D, VIT_F32, &ctx->g_pos);
        o_patch_w = ab.add("embeddings.patch_embedding.weight", D, ctx->kreal, ctx->kpad, T, &ctx->g_patch_w);
        o_patch_b = ab.raw((size_t)D * 4);  // Conv2d(bias=False), modeling_clip.py:148-154
        o_pre_g = ab.add("pre_layrnorm.weight", 1, D, D, VIT_F32, &ctx->g_pre_g);
        o_pre_b = ab.add("pre_layrnorm.bias", 1, D, D, VIT_F32, &ctx->g_pre_b);
    }
    for (int l = 0; l < L; ++l) {
        char p[96];
        const char *ln1, *ln2, *q, *k, *v, *o;
        if (!clip) {
            snprintf(p, sizeof(p), "layers.%d.", l);
            ln1 = "layernorm_before"; ln2 = "layernorm_after";
            q = "attention.q_proj"; k = "attention.k_proj"; v = "attention.v_proj"; o = "attention.o_proj";
        } else {
            snprintf(p, sizeof(p), "encoder.layers.%d.", l);
            ln1 = "layer_norm1"; ln2 = "layer_norm2";
            q = "self_attn.q_proj"; k = "self_attn.k_proj"; v = "self_attn.v_proj"; o = "self_attn.out_proj";
        }
        const std::string pre(p);
        LayerG& G = ctx->glayers[l];
        G.begin = ab.gcur;
        lo[l].ln1g = ab.add(pre + ln1 + ".weight", 1, D, D, VIT_F32, &G.ln1_g);
        lo[l].ln1b = ab.add(pre + ln1 + ".bias", 1, D, D, VIT_F32, &G.ln1_b);
        lo[l].ln2g = ab.add(pre + ln2 + ".weight", 1, D, D, VIT_F32, &G.ln2_g);
        lo[l].ln2b = ab.add(pre + ln2 + ".bias", 1, D, D, VIT_F32, &G.ln2_b);
        // packed [3D, D] QKV: three slots into one matrix / one bias vector
        lo[l].wqkv = ab.cur;
        const bool has_qkv_bias = clip || c.qkv_bias;
        const char* names[3] = {q, k, v};
        for (int j = 0; j < 3; ++j) {
            (*ab.slots)[pre + names[j] + ".weight"] =
                Slot{lo[l].wqkv + (size_t)j * D * D * esize(T), D, D, D, T, false, ab.gcur + (size_t)j * D * D};
        }
        ab.cur = align256(ab.cur + (size_t)3 * D * D * esize(T));
        G.wqkv = ab.gcur;
        ab.gcur = (ab.gcur + (size_t)3 * D * D + 63) & ~(size_t)63;
        G.bqkv = ab.gcur;
        lo[l].bqkv = ab.cur;
        for (int j = 0; j < 3; ++j)
            if (has_qkv_bias)
                (*ab.slots)[pre + names[j] + ".bias"] = Slot{lo[l].bqkv + (size_t)j * D * 4, 1, D, D, VIT_F32, false, ab.gcur + (size_t)j * D};
        ab.cur = align256(ab.cur + (size_t)3 * D * 4);
        ab.gcur = (ab.gcur + (size_t)3 * D + 63) & ~(size_t)63;
        lo[l].wo = ab.add(pre + o + ".weight", D, D, D, T, &G.wo);
        lo[l].bo = ab.add(pre + o + ".bias", 1, D, D, VIT_F32, &G.bo);
        lo[l].w1 = ab.add(pre + "mlp.fc1.weight", F, D, D, T, &G.w1);
        lo[l].b1 = ab.add(pre + "mlp.fc1.bias", 1, F, F, VIT_F32, &G.b1);
        lo[l].w2 = ab.add(pre + "mlp.fc2.weight", D, F, F, T, &G.w2);
        lo[l].b2 = ab.add(pre + "mlp.fc2.bias", 1, D, D, VIT_F32, &G.b2);
        G.end = ab.gcur;
    }
    if (!clip) {
        o_fin_g = ab.add("layernorm.weight", 1, D, D, VIT_F32, &ctx->g_fin_g);
        o_fin_b = ab.add("layernorm.bias", 1, D, D, VIT_F32, &ctx->g_fin_b);
    } else {
        o_fin_g = ab.add("post_layernorm.weight", 1, D, D, VIT_F32, &ctx->g_fin_g);
        o_fin_b = ab.add("post_layernorm.bias", 1, D, D, VIT_F32, &ctx->g_fin_b);
    }
    ctx->arena_bytes = ab.cur;
    ctx->g_total = ab.gcur;
    ctx->train = nullptr;
    hipError_t e = hipMalloc((void**)&ctx->arena, ctx->arena_bytes);
    if (e != hipSuccess) {
        delete ctx;
        return vit_set_error(VIT_E_NOMEM, "vit_create: hipMalloc(%zu) failed: %s", ab.cur, hipGetErrorString(e));
    }
    e = hipMemset(ctx->arena, 0, ctx->arena_bytes);  // K padding and absent biases read as zero
    if (e == hipSuccess) e = hipDeviceSynchronize();  // loads may come on any stream
    if (e != hipSuccess) {
        (void)hipFree(ctx->arena);
        delete ctx;
        return vit_set_error(VIT_E_HIP, "vit_create: hipMemset failed: %s", hipGetErrorString(e));
    }
    unsigned char* A = ctx->arena;
    ctx->patch_w = A + o_patch_w;
    ctx->patch_b = (float*)(A + o_patch_b);
    ctx->cls = (float*)(A + o_cls);
    ctx->pos = (float*)(A + o_pos);
    ctx->pre_g = (float*)(A + o_pre_g);
    ctx->pre_b = (float*)(A + o_pre_b);
    ctx->fin_g = (float*)(A + o_fin_g);
    ctx->fin_b = (float*)(A + o_fin_b);
    ctx->layers.resize(L);
    for (int l = 0; l < L; ++l) {
        LayerW& w = ctx->layers[l];
        w.ln1_g = (float*)(A + lo[l].ln1g); w.ln1_b = (float*)(A + lo[l].ln1b);
        w.ln2_g = (float*)(A + lo[l].ln2g); w.ln2_b = (float*)(A + lo[l].ln2b);
        w.bqkv = (float*)(A + lo[l].bqkv); w.bo = (float*)(A + lo[l].bo);
        w.b1 = (float*)(A + lo[l].b1); w.b2 = (float*)(A + lo[l].b2);
        w.wqkv = A + lo[l].wqkv; w.wo = A + lo[l].wo; w.w1 = A + lo[l].w1; w.w2 = A + lo[l].w2;
    }
    ctx->n_missing = (int)ctx->slots.size();
    // LayerNorm folded into the QKV / fc1 GEMMs (bf16 contexts), and a 128-padded copy of the patch weight for
    // patch sizes the persistent kernel's pixel loader does not take. The copies live in their own arena,
    // allocated here so that no forward ever allocates.
    ctx->fold_arena = nullptr;
    ctx->fold_dirty = true;
    ctx->patch_kp = 0;
    ctx->patch_w_kp = nullptr;
    const char* fl = getenv("VIT_AMD_FUSE_LN");
    ctx->ln_fuse = (T == VIT_BF16 && !(fl && atoi(fl) == 0)) ? 1 : 0;
    if (T == VIT_BF16) {
        PatchGeom pg0{c.num_channels, c.image_size, c.patch_size, g, ctx->kreal};
        const int np0 = g * g;
        if (!gemm_bf16_8p_patch_eligible(np0 * 64, D, &pg0) && np0 >= 64 && D % 8 == 0) ctx->patch_kp = (ctx->kreal + 127) / 128 * 128;
        if (ctx->patch_kp < 256) ctx->patch_kp = 0;
    }
    if (ctx->ln_fuse || ctx->patch_kp) {
        const size_t per = ctx->ln_fuse ? align256((size_t)3 * D * D * 2) + align256((size_t)F * D * 2) + 2 * align256((size_t)3 * D * 4) +
                                              2 * align256((size_t)F * 4)
                                        : 0;
        const size_t pw = ctx->patch_kp ? align256((size_t)D * ctx->patch_kp * 2) : 0;
        e = hipMalloc((void**)&ctx->fold_arena, per * L + pw);
        if (e == hipSuccess && pw) e = hipMemset(ctx->fold_arena, 0, pw);  // the K padding stays zero
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            (void)hipFree(ctx->arena);
            if (ctx->fold_arena) (void)hipFree(ctx->fold_arena);
            delete ctx;
            return vit_set_error(VIT_E_NOMEM, "vit_create: hipMalloc(%zu) for the folded weights failed: %s", per * L + pw, hipGetErrorString(e));
        }
        unsigned char* p = ctx->fold_arena;
        auto take = [&](size_t bytes) { unsigned char* r = p; p += align256(bytes); return r; };
        if (pw) ctx->patch_w_kp = take((size_t)D * ctx->patch_kp * 2);
        if (ctx->ln_fuse) {
            ctx->folds.resize(L);
            for (int l = 0; l < L; ++l) {
                LnFold& f = ctx->folds[l];
                f.wqkv = take((size_t)3 * D * D * 2); f.w1 = take((size_t)F * D * 2);
                f.cqkv = (float*)take((size_t)3 * D * 4); f.dqkv = (float*)take((size_t)3 * D * 4);
                f.c1 = (float*)take((size_t)F * 4); f.d1 = (float*)take((size_t)F * 4);
            }
        }
    }
    // two-stream forward: on by default for bf16 contexts, VIT_AMD_DUAL_STREAM=0/1 overrides
    ctx->ev_fork = nullptr;
    for (int i = 0; i < vit_ctx::kMaxStreams - 1; ++i) { ctx->aux_stream[i] = nullptr; ctx->ev_join[i] = nullptr; }
    const char* ds = getenv("VIT_AMD_STREAMS");
    ctx->nstreams = ds ? atoi(ds) : (c.dtype == VIT_BF16 ? 2 : 1);
    if (ctx->nstreams < 1) ctx->nstreams = 1;
    if (ctx->nstreams > vit_ctx::kMaxStreams) ctx->nstreams = vit_ctx::kMaxStreams;
    if (ctx->nstreams > 1) {
        e = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming);
        for (int i = 0; i < ctx->nstreams - 1 && e == hipSuccess; ++i)
            e = hipStreamCreateWithFlags(&ctx->aux_stream[i], hipStreamNonBlocking);
        // all join events exist whenever there is a side stream (the backward uses two of them on aux_stream[0])
        for (int i = 0; i < vit_ctx::kMaxStreams - 1 && e == hipSuccess; ++i)
            e = hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            vit_destroy(ctx);
            return vit_set_error(VIT_E_HIP, "vit_create: stream / event creation failed: %s", hipGetErrorString(e));
        }
    }
    *out = ctx;
    return VIT_OK;
}

extern "C" int vit_destroy(vit_ctx* ctx) {
    if (!ctx) return VIT_OK;
    vit_train_free(ctx);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (int i = 0; i < vit_ctx::kMaxStreams - 1; ++i) {
        if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
        if (ctx->aux_stream[i]) (void)hipStreamDestroy(ctx->aux_stream[i]);
    }
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->fold_arena) (void)hipFree(ctx->fold_arena);
    delete ctx;
    return VIT_OK;
}

extern "C" int vit_load_weight(vit_ctx* ctx, const char* name, const void* src_dev, int src_dtype, int64_t numel,
                               void* stream) {
    VIT_REQUIRE(ctx && name && src_dev, "vit_load_weight: null argument");
    VIT_REQUIRE(src_dtype == VIT_F32 || src_dtype == VIT_BF16, "vit_load_weight: bad source dtype %d", src_dtype);
    auto it = ctx->slots.find(name);
    if (it == ctx->slots.end()) return vit_set_error(VIT_E_INVALID, "vit_load_weight: unexpected key '%s'", name);
    Slot& s = it->second;
    if (numel != s.rows * s.cols)
        return vit_set_error(VIT_E_INVALID, "vit_load_weight: '%s' has %lld elements, expected %lld", name,
                             (long long)numel, (long long)(s.rows * s.cols));
    int rc = launch_convert_2d(src_dev, src_dtype, ctx->arena + s.offset, s.dtype, s.rows, s.cols, s.pitch,
                               (hipStream_t)stream);
    if (rc == VIT_OK && !s.loaded) {
        s.loaded = true;
        --ctx->n_missing;
    }
    ctx->fold_dirty = true;
    return rc;
}

extern "C" int vit_missing_weights(vit_ctx* ctx, const char** first_missing) {
    if (!ctx) return vit_set_error(VIT_E_INVALID, "vit_missing_weights: null context");
    int n = 0;
    for (auto& kv : ctx->slots)
        if (!kv.second.loaded) {
            if (n == 0) ctx->missing_name = kv.first;
            ++n;
        }
    if (first_missing) *first_missing = n ? ctx->missing_name.c_str() : nullptr;
    return n;
}

// ---------------------------------------------------------------- forward
namespace {
struct Workspace {
    size_t x, ln, qkv, ctxv, h, pix, stats, rows, total;
};
Workspace plan_workspace(const vit_ctx* ctx, int batch) {
    const vit_config& c = ctx->cfg;
    const size_t M = (size_t)batch * ctx->N, es = esize(c.dtype);
    Workspace w{};
    size_t cur = 0;
    auto take = [&](size_t bytes) { size_t o = cur; cur = align256(cur + bytes); return o; };
    w.x = take(M * c.hidden_size * es);
    w.ln = take(M * c.hidden_size * es);
    w.qkv = take(M * 3 * c.hidden_size * es);
    w.ctxv = take(M * c.hidden_size * es);
    w.h = take(M * c.intermediate_size * es);
    w.pix = take((size_t)batch * c.num_channels * c.image_size * c.image_size * es);
    w.stats = take(M * 8);  // folded LayerNorm: (mean * rstd, rstd) per row
    w.rows = take(ctx->patch_kp ? (size_t)batch * (ctx->N - 1) * ctx->patch_kp * 2 : 0);  // gathered patch rows
    w.total = cur;
    return w;
}
}  // namespace

// A batch is cut into up to ctx->nstreams contiguous chunks of at least kMinChunkRows token rows (32 images of
// 197 tokens: below that a chunk's GEMMs are too few tiles to be worth a stream), one stream each
int vit_chunk_begin(int batch, int k, int i);
int vit_num_chunks(const vit_ctx* ctx, int batch) {
    static const long kMinChunkRows = 32 * 197;
    int k = ctx->nstreams;
    // the SMALLEST chunk decides (two chunks are cut 5 : 3, vit_chunk_begin): below the minimum its N = D GEMMs fall
    // under the persistent kernel's size threshold
    auto smallest = [&](int kk) {
        long m = batch;
        for (int i = 0; i < kk; ++i) {
            const long nb = vit_chunk_begin(batch, kk, i + 1) - vit_chunk_begin(batch, kk, i);
            if (nb < m) m = nb;
        }
        return m;
    };
    while (k > 1 && smallest(k) * ctx->N < kMinChunkRows) --k;
    return k;
}
static int num_chunks(const vit_ctx* ctx, int batch) { return vit_num_chunks(ctx, batch); }
int vit_chunk_begin(int batch, int k, int i);
// First image of chunk i of k. Two chunks are cut 5 : 3, not in halves: equal chunks run the same kernels for the same
// time on both streams, so their last tile rounds (the idle CUs the second stream is there to fill) coincide; unequal
// chunks drift out of phase. Measured (tools/split_sweep.sh, one box): ViT-B/16 batch 256 9.00 ms at 50 %, 8.74 at 58 %,
// 8.67 at 62 %, 8.70 at 64 %, 8.75 at 70 %; CLIP-L/14-336 batch 128 and ViT-H/14-448 batch 32: -0.3 % .. -0.8 %.
// Lab switch VIT_AMD_SPLIT_PCT: share of the batch in the first chunk, in percent.
int vit_chunk_begin(int batch, int k, int i) {
    static const int pct = getenv("VIT_AMD_SPLIT_PCT") ? atoi(getenv("VIT_AMD_SPLIT_PCT")) : 62;
    if (k == 2 && i == 1 && pct > 0 && pct < 100) {
        const int b = (int)(((long)batch * pct + 50) / 100);
        return b < 1 ? 1 : (b > batch - 1 ? batch - 1 : b);
    }
    return (int)((long)batch * i / k);
}
static int chunk_begin(int batch, int k, int i) { return vit_chunk_begin(batch, k, i); }

extern "C" size_t vit_workspace_bytes(const vit_ctx* ctx, int batch) {
    if (!ctx || batch <= 0) return 0;
    const int k = num_chunks(ctx, batch);
    size_t total = 0;
    for (int i = 0; i < k; ++i) total += plan_workspace(ctx, chunk_begin(batch, k, i + 1) - chunk_begin(batch, k, i)).total;
    return total;
}

#define VIT_TRY(expr)                 \
    do {                              \
        int _rc = (expr);             \
        if (_rc != VIT_OK) return _rc; \
    } while (0)

static int forward_range(vit_ctx* ctx, const void* pixels_dev, int pixel_dtype, int batch, void* hidden_out,
                         void* pooled_out, void* workspace, hipStream_t stream, bool concurrent);

// Folded LayerNorm (modeling_vit.py:274,281 feeding :216-218 / :250): LN(x) W^T + b is computed as
// rstd * (x Wg^T) - (mean * rstd) * c + d inside the GEMM's epilogue, so the normalised activations are never
// written or read: a statistics pass over x (8 bytes out per row) replaces the LayerNorm kernel. The folded
// copies are rebuilt on the forward's stream after any weight change (load, SGD step).
static int refresh_folds(vit_ctx* ctx, hipStream_t stream) {
    if (!ctx->fold_dirty || (!ctx->ln_fuse && !ctx->patch_kp)) return VIT_OK;
    const int D = ctx->cfg.hidden_size, F = ctx->cfg.intermediate_size;
    if (ctx->patch_kp)  // [D, kreal] out of the [D, kpad] arena copy into the 128-padded one
        VIT_CHECK_HIP(hipMemcpy2DAsync(ctx->patch_w_kp, (size_t)ctx->patch_kp * 2, ctx->patch_w, (size_t)ctx->kpad * 2,
                                       (size_t)ctx->kreal * 2, D, hipMemcpyDeviceToDevice, stream));
    for (int l = 0; ctx->ln_fuse && l < ctx->cfg.num_hidden_layers; ++l) {
        const LayerW& lw = ctx->layers[l];
        const LnFold& f = ctx->folds[l];
        VIT_TRY(launch_ln_fold(lw.wqkv, D, lw.ln1_g, lw.ln1_b, lw.bqkv, f.wqkv, f.cqkv, f.dqkv, 3 * D, D, stream));
        VIT_TRY(launch_ln_fold(lw.w1, D, lw.ln2_g, lw.ln2_b, lw.b1, f.w1, f.c1, f.d1, F, D, stream));
    }
    ctx->fold_dirty = false;
    return VIT_OK;
}

extern "C" int vit_forward(vit_ctx* ctx, const void* pixels_dev, int pixel_dtype, int batch, void* hidden_out,
                           void* pooled_out, void* workspace, size_t workspace_bytes, void* stream_) {
    VIT_REQUIRE(ctx && pixels_dev && hidden_out && workspace, "vit_forward: null argument");
    VIT_REQUIRE(batch > 0, "vit_forward: batch must be positive (got %d)", batch);
    VIT_REQUIRE(pixel_dtype == VIT_F32 || pixel_dtype == VIT_BF16, "vit_forward: bad pixel dtype %d", pixel_dtype);
    VIT_REQUIRE(((uintptr_t)workspace & 255) == 0, "vit_forward: workspace must be 256-byte aligned");
    const char* missing = nullptr;
    const int nmiss = ctx->n_missing ? vit_missing_weights(ctx, &missing) : 0;
    if (nmiss) return vit_set_error(VIT_E_STATE, "vit_forward: %d weights not loaded (e.g. '%s')", nmiss, missing);
    int dev;
    VIT_CHECK_HIP(hipGetDevice(&dev));
    if (dev != ctx->device)
        return vit_set_error(VIT_E_STATE, "vit_forward: context lives on device %d, current device is %d", ctx->device, dev);
    const size_t need = vit_workspace_bytes(ctx, batch);
    if (workspace_bytes < need)
        return vit_set_error(VIT_E_NOMEM, "vit_forward: workspace %zu < required %zu bytes", workspace_bytes, need);
    hipStream_t stream = (hipStream_t)stream_;
    VIT_TRY(refresh_folds(ctx, stream));
    const int k = num_chunks(ctx, batch);
    if (k == 1) return forward_range(ctx, pixels_dev, pixel_dtype, batch, hidden_out, pooled_out, workspace, stream, false);

    // chunks on separate streams: fork after whatever the caller queued, join before returning control.
    // One chunk's kernels fill the CUs that another chunk's last round of tiles leaves idle.
    const vit_config& c = ctx->cfg;
    const size_t es = esize(c.dtype), pes = esize(pixel_dtype);
    const size_t img = (size_t)c.num_channels * c.image_size * c.image_size;
    VIT_CHECK_HIP(hipEventRecord(ctx->ev_fork, stream));
    size_t ws_off = 0;
    // After the fork every side stream is joined back into `stream` whatever happens in between: an error in
    // one chunk must not leave a side stream running against the workspace when the call returns.
    int rc = VIT_OK;
    for (int i = 0; i < k; ++i) {
        const int b0 = chunk_begin(batch, k, i), nb = chunk_begin(batch, k, i + 1) - b0;
        hipStream_t st = i == 0 ? stream : ctx->aux_stream[i - 1];
        bool forked = false;
        if (i > 0 && rc == VIT_OK) {
            const hipError_t e = hipStreamWaitEvent(st, ctx->ev_fork, 0);
            if (e != hipSuccess) rc = vit_set_error(VIT_E_HIP, "vit_forward: hipStreamWaitEvent: %s", hipGetErrorString(e));
            else forked = true;
        }
        if (rc == VIT_OK)
            rc = forward_range(ctx, (const char*)pixels_dev + (size_t)b0 * img * pes, pixel_dtype, nb,
                               (char*)hidden_out + (size_t)b0 * ctx->N * c.hidden_size * es,
                               pooled_out ? (char*)pooled_out + (size_t)b0 * c.hidden_size * es : nullptr,
                               (char*)workspace + ws_off, st, true);
        ws_off += plan_workspace(ctx, nb).total;
        if (forked) {
            hipError_t e = hipEventRecord(ctx->ev_join[i - 1], st);
            if (e == hipSuccess) e = hipStreamWaitEvent(stream, ctx->ev_join[i - 1], 0);
            if (e != hipSuccess && rc == VIT_OK) rc = vit_set_error(VIT_E_HIP, "vit_forward: stream join: %s", hipGetErrorString(e));
        }
    }
    return rc;
}

// vit_forward on the library's static scratch array instead of a caller workspace: no hipMalloc / hipFree or any
// allocating routine on this path (vit_forward itself allocates nothing). Every call on a GPU shares that GPU's
// array, so calls that overlap on one GPU (two streams, two threads, two contexts) race; nothing in the workspace
// has to start at zero (every buffer is written before it is read), so the last call's contents are harmless.
extern "C" int vit_forward_static(vit_ctx* ctx, const void* pixels_dev, int pixel_dtype, int batch, void* hidden_out,
                                  void* pooled_out, void* stream) {
    VIT_REQUIRE(ctx != nullptr, "vit_forward_static: null context");
    VIT_REQUIRE(batch > 0, "vit_forward_static: batch must be positive (got %d)", batch);
    const size_t need = vit_workspace_bytes(ctx, batch);
    if (need > (size_t)VIT_STATIC_SCRATCH_BYTES)
        return vit_set_error(VIT_E_NOMEM, "vit_forward_static: batch %d needs %zu workspace bytes, the static scratch holds %zu "
                                          "(use vit_forward with a caller workspace)", batch, need, (size_t)VIT_STATIC_SCRATCH_BYTES);
    unsigned char* const ptr = static_scratch_of(ctx->device);
    if (ptr == nullptr)
        return vit_set_error(VIT_E_STATE, "vit_forward_static: no static scratch resolved for device %d", ctx->device);
    return vit_forward(ctx, pixels_dev, pixel_dtype, batch, hidden_out, pooled_out, ptr, (size_t)VIT_STATIC_SCRATCH_BYTES, stream);
}

// the whole forward of `batch` images on one stream
static int forward_range(vit_ctx* ctx, const void* pixels_dev, int pixel_dtype, int batch, void* hidden_out,
                         void* pooled_out, void* workspace, hipStream_t stream, bool concurrent) {
    const Workspace w = plan_workspace(ctx, batch);
    const vit_config& c = ctx->cfg;
    const int T = c.dtype, D = c.hidden_size, F = c.intermediate_size, N = ctx->N, H = c.num_attention_heads;
    const int M = batch * N;
    const size_t es = esize(T);
    unsigned char* ws = (unsigned char*)workspace;
    void* x = ws + w.x;
    void* ln = ws + w.ln;
    void* qkv = ws + w.qkv;
    void* cx = ws + w.ctxv;
    void* h = ws + w.h;
    const bool clip = c.variant == VIT_VARIANT_CLIP;
    const int act_epi = c.hidden_act == VIT_ACT_GELU ? VIT_EPI_GELU : VIT_EPI_QUICK_GELU;
    float* stats = (float*)(ws + w.stats);
    // folded LayerNorm at every batch size (both GEMM kernels carry the epilogue and give the same bits, so a
    // sub-batch reproduces the rows of the full batch)
    const bool fuse = ctx->ln_fuse && !ctx->fold_dirty;
    // Row sums from the residual epilogues: the o_proj / fc2 GEMM that writes x also leaves, per 64-column group, the
    // (sum, sum of squares) of every row it stored, and a tiny launch turns the D / 64 slabs into the statistics the next
    // folded LayerNorm reads -- instead of a pass over x (13 us at ViT-B/16 batch 256, 24 of them per forward). Same numbers bit for bit as launch_ln_stats (one instruction sequence,
    // vit_common.h), so which route ran is not visible in the output. The slabs live in the workspace's LayerNorm-output buffer, which a folded forward never
    // uses. Lab switch VIT_AMD_ROW_STATS=0: always the stand-alone pass.
    static const bool row_stats_on = !(getenv("VIT_AMD_ROW_STATS") && atoi(getenv("VIT_AMD_ROW_STATS")) == 0);
    const bool rst = fuse && row_stats_on && T == VIT_BF16 && ln_stats_one_pass(D) && gemm_writes_row_stats(M, D, D) && gemm_writes_row_stats(M, D, F) &&
                     (size_t)(D / 64) * M * 8 <= (size_t)M * D * es;
    float* part = (float*)ln;
    bool have_part = false;  // `part` holds the row sums of x (left by the GEMM that wrote it)
    // a folded LayerNorm's statistics: the slabs through the finalising launch, or a pass over x
    auto ln_source = [&](GemmExt& e) -> int {
        if (have_part) VIT_TRY(launch_row_stats_finalize(part, stats, M, D, c.layer_norm_eps, stream));
        else VIT_TRY(launch_ln_stats(x, stats, M, D, D, c.layer_norm_eps, T, stream));
        e.ln_stats = stats;
        have_part = false;
        return VIT_OK;
    };

    // embeddings (modeling_vit.py:129-161 / modeling_clip.py:202-218)
    const void* pix = pixels_dev;
    if (pixel_dtype != T) {
        const int64_t n = (int64_t)batch * c.num_channels * c.image_size * c.image_size;
        VIT_TRY(launch_convert(pixels_dev, pixel_dtype, ws + w.pix, T, n, stream));
        pix = ws + w.pix;
    }
    VIT_TRY(launch_cls_rows(ctx->cls, ctx->pos, x, batch, N, D, T, stream));
    PatchGeom pg{c.num_channels, c.image_size, c.patch_size, c.image_size / c.patch_size, ctx->kreal};
    const int Mp = batch * (N - 1);
    if (ctx->patch_kp && !ctx->fold_dirty && gemm_bf16_8p_patch_dense_eligible(Mp, D, ctx->patch_kp, &pg)) {
        // P = 14 and the like: gather the patch rows once (16-byte stores), then the persistent GEMM with the
        // CLS-slot / position epilogue; the generic kernel's element gather is 6x slower at CLIP-L/14-336
        VIT_TRY(launch_im2col_rows(pix, ws + w.rows, Mp, ctx->patch_kp, &pg, stream));
        VIT_TRY(launch_gemm_bf16_8p_patch_dense(ws + w.rows, ctx->patch_w_kp, ctx->patch_b, ctx->pos, x, Mp, D, ctx->patch_kp, &pg, stream));
    } else {
        VIT_TRY(launch_gemm(pix, ctx->patch_w, ctx->patch_b, nullptr, x, Mp, D, ctx->kpad, VIT_EPI_BIAS, T, &pg, ctx->pos, stream));
    }
    if (clip) VIT_TRY(launch_layernorm(x, ctx->pre_g, ctx->pre_b, x, M, D, D, D, c.layer_norm_eps, T, stream));

    for (int l = 0; l < c.num_hidden_layers; ++l) {
        const LayerW& lw = ctx->layers[l];
        const bool last = (l == c.num_hidden_layers - 1);
        // x + Attn(LN1(x))   (modeling_vit.py:273-277)
        if (fuse) {
            GemmExt e{};
            e.concurrent = concurrent;
            e.ln_colsum = ctx->folds[l].cqkv;
            VIT_TRY(ln_source(e));
            VIT_TRY(launch_gemm(x, ctx->folds[l].wqkv, ctx->folds[l].dqkv, nullptr, qkv, M, 3 * D, D, VIT_EPI_BIAS, T, nullptr,
                                nullptr, stream, &e));
        } else {
            VIT_TRY(launch_layernorm(x, lw.ln1_g, lw.ln1_b, ln, M, D, D, D, c.layer_norm_eps, T, stream));
            VIT_TRY(launch_gemm(ln, lw.wqkv, lw.bqkv, nullptr, qkv, M, 3 * D, D, VIT_EPI_BIAS, T, nullptr, nullptr, stream));
        }
        VIT_TRY(launch_attention(qkv, cx, batch, N, H, ctx->dh, T, stream));
        GemmExt er{};  // o_proj / fc2 + residual: only the chunk hint
        er.split_k = 1;
        er.concurrent = concurrent;
        if (rst) er.row_part = part;
        VIT_TRY(launch_gemm(cx, lw.wo, lw.bo, x, x, M, D, D, VIT_EPI_RESIDUAL, T, nullptr, nullptr, stream, &er));
        have_part = rst;
        // x + MLP(LN2(x))    (modeling_vit.py:280-284)
        if (fuse) {
            GemmExt e{};
            e.concurrent = concurrent;
            e.ln_colsum = ctx->folds[l].c1;
            VIT_TRY(ln_source(e));
            VIT_TRY(launch_gemm(x, ctx->folds[l].w1, ctx->folds[l].d1, nullptr, h, M, F, D, act_epi, T, nullptr, nullptr, stream, &e));
        } else {
            VIT_TRY(launch_layernorm(x, lw.ln2_g, lw.ln2_b, ln, M, D, D, D, c.layer_norm_eps, T, stream));
            VIT_TRY(launch_gemm(ln, lw.w1, lw.b1, nullptr, h, M, F, D, act_epi, T, nullptr, nullptr, stream));
        }
        // CLIP returns the un-normed stream: let the last fc2 write it straight to hidden_out
        void* dst = (clip && last) ? hidden_out : x;
        er.row_part = (rst && !last) ? part : nullptr;  // the last layer's output meets no folded LayerNorm
        VIT_TRY(launch_gemm(h, lw.w2, lw.b2, x, dst, M, D, F, VIT_EPI_RESIDUAL, T, nullptr, nullptr, stream, &er));
        have_part = rst && !last;
    }
    if (!clip) {
        // final LayerNorm over all tokens; pooled = its CLS row (modeling_vit.py:385-388, no pooler)
        VIT_TRY(launch_layernorm(x, ctx->fin_g, ctx->fin_b, hidden_out, M, D, D, D, c.layer_norm_eps, T, stream));
        if (pooled_out)  // the CLS rows of hidden_out, strided copy (same bits as a second LayerNorm launch on them)
            VIT_CHECK_HIP(hipMemcpy2DAsync(pooled_out, (size_t)D * es, hidden_out, (size_t)N * D * es, (size_t)D * es, batch,
                                           hipMemcpyDeviceToDevice, stream));
    } else if (pooled_out) {
        // post_layernorm on the CLS row only (modeling_clip.py:649-651)
        VIT_TRY(launch_layernorm(hidden_out, ctx->fin_g, ctx->fin_b, pooled_out, batch, D, (int64_t)N * D, D,
                                 c.layer_norm_eps, T, stream));
    }
    return VIT_OK;
}

// ---------------------------------------------------------------- per-kernel entry points
extern "C" int vit_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, int rows, int dim,
                                 int64_t x_row_stride, int64_t y_row_stride, float eps, int dtype, void* stream) {
    VIT_REQUIRE(x && gamma && beta && y, "vit_layernorm_fwd: null argument");
    return launch_layernorm(x, gamma, beta, y, rows, dim, x_row_stride, y_row_stride, eps, dtype, (hipStream_t)stream);
}

// ---- LayerNorm folded into the Linear that follows it (what vit_forward does for LN1 -> QKV, LN2 -> fc1)
extern "C" int vit_layernorm_stats(const void* x, float* stats, int rows, int dim, int64_t x_row_stride, float eps, int dtype,
                                   void* stream) {
    VIT_REQUIRE(x && stats, "vit_layernorm_stats: null argument");
    return launch_ln_stats(x, stats, rows, dim, x_row_stride, eps, dtype, (hipStream_t)stream);
}

extern "C" int vit_layernorm_fold(const void* weight, const float* gamma, const float* beta, const float* bias,
                                  void* weight_folded, float* colsum, float* bias_folded, int out_dim, int in_dim,
                                  void* stream) {
    VIT_REQUIRE(weight && gamma && beta && weight_folded && colsum && bias_folded, "vit_layernorm_fold: null argument");
    return launch_ln_fold(weight, in_dim, gamma, beta, bias, weight_folded, colsum, bias_folded, out_dim, in_dim, (hipStream_t)stream);
}

extern "C" int vit_gemm_ln_folded(const void* x, const void* weight_folded, const float* bias_folded, const float* colsum,
                                  const float* stats, void* C, int M, int N, int K, int epilogue, void* stream) {
    VIT_REQUIRE(x && weight_folded && bias_folded && colsum && stats && C, "vit_gemm_ln_folded: null argument");
    VIT_REQUIRE(epilogue == VIT_EPI_BIAS || epilogue == VIT_EPI_GELU || epilogue == VIT_EPI_QUICK_GELU,
                "vit_gemm_ln_folded: epilogue %d (bias, GELU or quick-GELU only)", epilogue);
    GemmExt e{};
    e.ln_stats = stats; e.ln_colsum = colsum;
    return launch_gemm(x, weight_folded, bias_folded, nullptr, C, M, N, K, epilogue, VIT_BF16, nullptr, nullptr, (hipStream_t)stream, &e);
}

extern "C" int vit_gemm_residual_row_stats(const void* A, const void* W, const float* bias, const void* residual, void* C,
                                           float* row_part, int M, int N, int K, void* stream) {
    VIT_REQUIRE(A && W && residual && C && row_part, "vit_gemm_residual_row_stats: null argument");
    VIT_REQUIRE(gemm_writes_row_stats(M, N, K), "vit_gemm_residual_row_stats: shape M=%d N=%d K=%d is not one the persistent kernel takes (N %% 64 == 0, K %% 128 == 0, M * N >= 4 Mi)", M, N, K);
    GemmExt e{};
    e.row_part = row_part;
    return launch_gemm(A, W, bias, residual, C, M, N, K, VIT_EPI_RESIDUAL, VIT_BF16, nullptr, nullptr, (hipStream_t)stream, &e);
}

extern "C" int vit_row_stats_finalize(const float* row_part, float* stats, int rows, int dim, float eps, void* stream) {
    VIT_REQUIRE(row_part && stats, "vit_row_stats_finalize: null argument");
    return launch_row_stats_finalize(row_part, stats, rows, dim, eps, (hipStream_t)stream);
}

extern "C" int vit_gemm_bias_act(const void* A, const void* W, const float* bias, const void* residual, void* C, int M,
                                 int N, int K, int epilogue, int dtype, void* stream) {
    VIT_REQUIRE(A && W && C, "vit_gemm_bias_act: null argument");
    VIT_REQUIRE(epilogue >= VIT_EPI_BIAS && epilogue <= VIT_EPI_RESIDUAL, "vit_gemm_bias_act: bad epilogue %d", epilogue);
    VIT_REQUIRE(epilogue != VIT_EPI_RESIDUAL || residual != nullptr, "vit_gemm_bias_act: residual epilogue without residual");
    return launch_gemm(A, W, bias, residual, C, M, N, K, epilogue, dtype, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int vit_attention_fwd(const void* qkv, void* out, int batch, int seq, int heads, int head_dim, int dtype,
                                 void* stream) {
    VIT_REQUIRE(qkv && out, "vit_attention_fwd: null argument");
    return launch_attention(qkv, out, batch, seq, heads, head_dim, dtype, (hipStream_t)stream);
}

extern "C" int vit_patch_embed_fwd(const void* pixels, const void* weight, const float* bias, const float* cls,
                                   const float* pos, void* x_out, int batch, int channels, int image, int patch, int dim,
                                   int dtype, void* stream) {
    VIT_REQUIRE(pixels && weight && cls && pos && x_out, "vit_patch_embed_fwd: null argument");
    VIT_REQUIRE(batch > 0 && channels > 0 && patch > 0 && image > 0 && image % patch == 0 && dim > 0,
                "vit_patch_embed_fwd: bad geometry");
    const int g = image / patch, N = g * g + 1;
    PatchGeom pg{channels, image, patch, g, channels * patch * patch};
    VIT_TRY(launch_cls_rows(cls, pos, x_out, batch, N, dim, dtype, (hipStream_t)stream));
    return launch_gemm(pixels, weight, bias, nullptr, x_out, batch * (N - 1), dim, vit_patch_kpad(channels, patch, dtype),
                       VIT_EPI_BIAS, dtype, &pg, pos, (hipStream_t)stream);
}

int launch_head(const void* x, int64_t xs, const void* W, const float* bias, void* out, int batch, int K, int N, int act,
                int dtype, hipStream_t stream);  // heads.hip

extern "C" int vit_head_fwd(const void* x, int64_t x_row_stride, const void* weight, const float* bias, void* out,
                            int batch, int in_dim, int out_dim, int activation, int dtype, void* stream) {
    VIT_REQUIRE(x && weight && out, "vit_head_fwd: null argument");
    return launch_head(x, x_row_stride, weight, bias, out, batch, in_dim, out_dim, activation, dtype, (hipStream_t)stream);
}

extern "C" int vit_convert(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t numel, void* stream) {
    VIT_REQUIRE(src && dst, "vit_convert: null argument");
    return launch_convert(src, src_dtype, dst, dst_dtype, numel, (hipStream_t)stream);
}

// ---------------------------------------------------------------- training-step entry points
extern "C" int vit_gemm_ex(const void* A, const void* W, const float* bias, const void* residual_or_aux, void* C,
                           void* C_pre, int M, int N, int K, int epilogue, int out_f32, void* stream) {
    VIT_REQUIRE(A && W && C, "vit_gemm_ex: null argument");
    VIT_REQUIRE(epilogue >= VIT_EPI_BIAS && epilogue <= VIT_EPI_DQUICK_GELU, "vit_gemm_ex: bad epilogue %d", epilogue);
    GemmExt ext{C_pre, out_f32, 1, nullptr};
    return launch_gemm(A, W, bias, residual_or_aux, C, M, N, K, epilogue, VIT_BF16, nullptr, nullptr, (hipStream_t)stream,
                       &ext);
}

extern "C" int vit_attention_fwd_train(const void* qkv, void* out, float* lse, int batch, int seq, int heads,
                                       int head_dim, void* stream) {
    VIT_REQUIRE(qkv && out && lse, "vit_attention_fwd_train: null argument");
    return launch_attention(qkv, out, batch, seq, heads, head_dim, VIT_BF16, (hipStream_t)stream, lse);
}

extern "C" int vit_attention_bwd(const void* qkv, const void* out, const void* d_out, const float* lse,
                                 float* delta_scratch, void* dqkv, int batch, int seq, int heads, int head_dim,
                                 void* stream) {
    VIT_REQUIRE(qkv && out && d_out && lse && delta_scratch && dqkv, "vit_attention_bwd: null argument");
    return launch_attention_bwd(qkv, out, d_out, lse, delta_scratch, dqkv, batch, seq, heads, head_dim,
                                (hipStream_t)stream);
}

extern "C" int vit_layernorm_bwd(const void* x, const void* dy, const float* gamma, const void* d_residual, void* dx,
                                 float* dgamma, float* dbeta, int rows, int dim, float eps, void* stream) {
    VIT_REQUIRE(x && dy && gamma && dx && dgamma && dbeta, "vit_layernorm_bwd: null argument");
    return launch_layernorm_bwd(x, dy, gamma, d_residual, dx, dgamma, dbeta, rows, dim, dim, dim, dim, eps,
                                (hipStream_t)stream);
}

int launch_gemm_tn_bf16(const void* dY, const void* X, float* dW, float* db, float* slabs, int T, int Mo, int No,
                        int splits, hipStream_t stream);  // gemm_tn_bf16.hip

extern "C" int vit_linear_wgrad(const void* dY, const void* X, float* dW, float* db, float* slabs, int tokens,
                                int out_features, int in_features, int splits, void* stream) {
    VIT_REQUIRE(dY && X && dW, "vit_linear_wgrad: null argument");
    return launch_gemm_tn_bf16(dY, X, dW, db, slabs, tokens, out_features, in_features, splits, (hipStream_t)stream);
}

extern "C" int vit_transpose_pad(const void* src, void* dst, float* colsum, int R, int C, int Rpad, int group,
                                 void* stream) {
    VIT_REQUIRE(src && dst, "vit_transpose_pad: null argument");
    return launch_transpose_pad(src, dst, colsum, R, C, Rpad, group, (hipStream_t)stream);
}
