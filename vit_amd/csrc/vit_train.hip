// Training step of the encoder through the C ABI (BASELINE.json configs[4]): a forward that
// stashes what the backward needs, the backward in stages (head, one stage per layer, embeddings)
// so that the caller can start the gradient all-reduce of a layer while the next one is being
// differentiated, and an SGD step on fp32 master weights. bf16 activations and matrix weights,
// fp32 statistics, accumulation and parameter gradients.
//
// The reference has no training code; the math differentiated here is
// ViTLayer.forward (modeling_vit.py:266-286), ViTEmbeddings.forward (:129-161), the final
// LayerNorm (:385) and the CLIP variant (modeling_clip.py:362-383, 641-656).
//
// Weight gradients dW = dY^T X run on the NT GEMM with both operands transposed to
// [features, tokens] (zero padded along tokens) and, because their output is small and their
// reduction dimension is the token count, split over K into fp32 slabs that are summed in a
// fixed order (reproducible; no atomics on matrices).
#include "vit_ctx.h"
#include <stdlib.h>

#include <stdio.h>
#include <string.h>

int launch_convert_2d(const void* src, int sdt, void* dst, int ddt, int64_t rows, int64_t cols, int64_t dst_pitch,
                      hipStream_t stream);

extern "C" int vit_train_refresh_transposes(vit_ctx* ctx, void* stream_);
int vit_num_chunks(const vit_ctx* ctx, int batch);  // vit_capi.hip: the stream split rule of vit_forward
int vit_chunk_begin(int batch, int k, int i);        // vit_capi.hip: first image of chunk i of k (two chunks: 5 : 3)

struct TrainState {
    float* master = nullptr;        // fp32 master copy of every parameter, gradient-arena layout
    unsigned char* wt = nullptr;    // transposed bf16 weights for the data-gradient GEMMs
    size_t wt_bytes = 0;
    struct LayerT { void *wqkvT, *woT, *w1T, *w2T; };
    std::vector<LayerT> layers;
    ConvSlot* conv_table = nullptr;  // device: every parameter tensor, master arena -> weight arena (vit_sgd_step)
    int nconv = 0;
    TransposeSlot* tr_table = nullptr;  // device: the 4 transposed weights of every layer (vit_train_refresh_transposes)
    int ntr = 0, tr_blocks = 0;
    // what a successful vit_forward_train wrote into a caller's stash, by stash address: every backward stage on that
    // stash must be called with the same batch (the stash layout is derived from it) and, for the embedding stage, the
    // same pixel dtype. Several stashes may be in flight (micro-batches, gradient accumulation); the table is bounded.
    struct Filled { int batch, pixel_dtype; };
    std::map<const void*, Filled> filled;
    static constexpr size_t kMaxFilled = 256;
};

int launch_gemm_tn_bf16(const void* dY, const void* X, float* dW, float* db, float* slabs, int T, int Mo, int No,
                        int splits, hipStream_t stream);  // gemm_tn_bf16.hip

namespace {

#define VIT_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != VIT_OK) return _rc; \
    } while (0)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// split-K factor of a weight-gradient GEMM with `tiles` output tiles: fill ~256 CUs
inline int wgrad_splits(int out_rows, int out_cols) {
    const int tiles = ceil_div(out_rows, 256) * ceil_div(out_cols, 256);
    int s = 256 / tiles;
    if (s < 1) s = 1;
    if (s > 32) s = 32;
    return s;
}

// geometry shared by the stash / workspace planners and the stages
struct TGeo {
    int B, N, D, F, H, L, M, Mp, Mpp, Kp;  // Mp: padded token count of the transposes; Mpp: same without CLS rows
    int big;                               // 1: problem large enough for split-K persistent weight-gradient GEMMs
};
TGeo tgeo(const vit_ctx* ctx, int batch) {
    const vit_config& c = ctx->cfg;
    TGeo g{};
    g.B = batch; g.N = ctx->N; g.D = c.hidden_size; g.F = c.intermediate_size; g.H = c.num_attention_heads;
    g.L = c.num_hidden_layers; g.M = batch * ctx->N; g.Kp = ctx->kpad;
    g.big = g.M >= 8192;
    // padded so that every split count used below divides it into whole pairs of 64-wide K-tiles
    // (lcm of the split counts would be wasteful: each GEMM pads its own operand instead)
    g.Mp = round_up(g.M, 128);
    g.Mpp = round_up(batch * (ctx->N - 1), 128);
    return g;
}

struct Stash {  // byte offsets
    size_t xe;                 // CLIP: embeddings before pre_layrnorm [M, D]
    std::vector<size_t> x;     // L + 1 residual streams [M, D]
    struct Lay { size_t xmid, ln1, qkv, ctx, lse, ln2, u, h; };
    std::vector<Lay> lay;
    size_t total;
};
Stash plan_stash(const vit_ctx* ctx, int batch) {
    const TGeo g = tgeo(ctx, batch);
    Stash s;
    size_t cur = 0;
    auto take = [&](size_t bytes) { size_t o = cur; cur = align256(cur + bytes); return o; };
    const size_t md = (size_t)g.M * g.D * 2, mf = (size_t)g.M * g.F * 2;
    s.xe = ctx->cfg.variant == VIT_VARIANT_CLIP ? take(md) : 0;
    s.x.resize(g.L + 1);
    s.lay.resize(g.L);
    for (int l = 0; l <= g.L; ++l) s.x[l] = take(md);
    for (int l = 0; l < g.L; ++l) {
        Stash::Lay& y = s.lay[l];
        y.xmid = take(md); y.ln1 = take(md); y.qkv = take(3 * md); y.ctx = take(md);
        y.lse = take((size_t)g.B * g.H * g.N * 4); y.ln2 = take(md); y.u = take(mf); y.h = take(mf);
    }
    s.total = cur;
    return s;
}

struct TWork {  // byte offsets
    size_t dx, dx2, dln, du, dqkv, t1, t2, delta, slabs, pix, lnred, total;
    int mp_max;  // largest padded token count any transpose uses
};
TWork plan_twork(const vit_ctx* ctx, int batch) {
    const TGeo g = tgeo(ctx, batch);
    TWork w{};
    size_t cur = 0;
    auto take = [&](size_t bytes) { size_t o = cur; cur = align256(cur + bytes); return o; };
    const size_t md = (size_t)g.M * g.D * 2;
    // worst-case padding: a split count s pads the token count to a multiple of 128 * s (s <= 32)
    w.mp_max = g.big ? round_up(g.M, 128 * 32) : g.Mp;
    const int frows = g.F > 3 * g.D ? g.F : 3 * g.D;
    int srows = g.F > g.D ? g.F : g.D;
    if (g.Kp > srows) srows = g.Kp;
    w.dx = take(md); w.dx2 = take(md); w.dln = take(md);
    w.du = take((size_t)g.M * g.F * 2);
    w.dqkv = take(3 * md);
    w.t1 = take((size_t)frows * w.mp_max * 2);
    w.t2 = take((size_t)srows * w.mp_max * 2);
    w.delta = take((size_t)g.B * g.H * g.N * 4);
    // slabs: splits * rows * cols floats, largest over the weight-gradient GEMMs
    size_t slab = 0;
    if (g.big) {
        const int shapes[5][2] = {{g.D, g.F}, {g.F, g.D}, {g.D, g.D}, {3 * g.D, g.D}, {g.D, g.Kp}};
        for (auto& sh : shapes) {
            const size_t need = (size_t)wgrad_splits(sh[0], sh[1]) * sh[0] * sh[1] * 4;
            if (need > slab) slab = need;
        }
    }
    w.slabs = take(slab);
    w.pix = take((size_t)batch * ctx->cfg.num_channels * ctx->cfg.image_size * ctx->cfg.image_size * 2);
    w.lnred = take((size_t)kLnBwdMaxBlocks * 2 * g.D * 4);
    w.total = cur;
    return w;
}

// dW[rows_out, cols_out] (fp32, at `dw`) = dY^T X where dY is [Mtok, rows_out] and X is
// [Mtok, cols_out] (both bf16, row-major, X possibly the CLS-free rows when group > 0), and
// db[rows_out] (may be null) += column sums of dY.
int wgrad(const vit_ctx* ctx, const TGeo& g, unsigned char* ws, const TWork& w, const void* dY, int rows_out,
          const void* X, int cols_out, int Mtok, int group_dy, float* dw, float* db, hipStream_t stream,
          const void* x_pre_transposed = nullptr) {
    int splits = g.big ? wgrad_splits(rows_out, cols_out) : 1;
    if (group_dy == 0 && x_pre_transposed == nullptr) {
        // token-major operands as they are: transposing MFMA kernel (gemm_tn_bf16.hip), bias gradient fused
        if (Mtok / splits < 256) splits = 1;
        return launch_gemm_tn_bf16(dY, X, dw, db, (float*)(ws + w.slabs), Mtok, rows_out, cols_out, splits, stream);
    }
    if ((long)rows_out * cols_out * splits < 64L * 256 * 256) splits = 1;  // too small for the persistent kernel
    const int mp = round_up(Mtok, 128 * splits);
    if (mp / splits < 256) splits = 1;
    void* t1 = ws + w.t1;
    void* t2 = ws + w.t2;
    VIT_TRY(launch_transpose_pad(dY, t1, db, Mtok, rows_out, mp, group_dy, stream));
    if (x_pre_transposed == nullptr) VIT_TRY(launch_transpose_pad(X, t2, nullptr, Mtok, cols_out, mp, 0, stream));
    GemmExt ext{nullptr, 1, splits, (float*)(ws + w.slabs)};
    return launch_gemm(t1, x_pre_transposed ? x_pre_transposed : t2, nullptr, nullptr, dw, rows_out, cols_out, mp,
                       VIT_EPI_BIAS, VIT_BF16, nullptr, nullptr, stream, &ext);
}

int check_train_ctx(const vit_ctx* ctx, const char* who) {
    if (!ctx) return vit_set_error(VIT_E_INVALID, "%s: null context", who);
    if (ctx->cfg.dtype != VIT_BF16) return vit_set_error(VIT_E_STATE, "%s: the training step needs a VIT_BF16 context", who);
    if (ctx->n_missing) return vit_set_error(VIT_E_STATE, "%s: %d weights not loaded", who, ctx->n_missing);
    int dev;
    VIT_CHECK_HIP(hipGetDevice(&dev));
    if (dev != ctx->device) return vit_set_error(VIT_E_STATE, "%s: context lives on device %d, current device is %d", who, ctx->device, dev);
    return VIT_OK;
}

}  // namespace

void vit_train_free(vit_ctx* ctx) {
    if (!ctx || !ctx->train) return;
    if (ctx->train->master) (void)hipFree(ctx->train->master);
    if (ctx->train->wt) (void)hipFree(ctx->train->wt);
    if (ctx->train->conv_table) (void)hipFree(ctx->train->conv_table);
    if (ctx->train->tr_table) (void)hipFree(ctx->train->tr_table);
    delete ctx->train;
    ctx->train = nullptr;
}

extern "C" size_t vit_train_stash_bytes(const vit_ctx* ctx, int batch) {
    if (!ctx || batch <= 0) return 0;
    return plan_stash(ctx, batch).total;
}
extern "C" size_t vit_train_workspace_bytes(const vit_ctx* ctx, int batch) {
    if (!ctx || batch <= 0) return 0;
    return plan_twork(ctx, batch).total;
}
extern "C" size_t vit_grad_bytes(const vit_ctx* ctx) { return ctx ? ctx->g_total * 4 : 0; }

extern "C" int vit_grad_slot(vit_ctx* ctx, const char* name, size_t* offset_bytes, int64_t* rows, int64_t* cols,
                             int64_t* pitch) {
    VIT_REQUIRE(ctx && name, "vit_grad_slot: null argument");
    auto it = ctx->slots.find(name);
    if (it == ctx->slots.end()) return vit_set_error(VIT_E_INVALID, "vit_grad_slot: unknown key '%s'", name);
    if (offset_bytes) *offset_bytes = it->second.goff * 4;
    if (rows) *rows = it->second.rows;
    if (cols) *cols = it->second.cols;
    if (pitch) *pitch = it->second.pitch;
    return VIT_OK;
}

// stage -1 (or num_layers): everything that is not a transformer layer (embeddings, norms);
// layers' ranges are contiguous so one all-reduce per stage covers them.
extern "C" int vit_grad_layer_range(vit_ctx* ctx, int layer, size_t* offset_bytes, size_t* bytes) {
    VIT_REQUIRE(ctx && offset_bytes && bytes, "vit_grad_layer_range: null argument");
    VIT_REQUIRE(layer >= 0 && layer < ctx->cfg.num_hidden_layers, "vit_grad_layer_range: bad layer %d", layer);
    *offset_bytes = ctx->glayers[layer].begin * 4;
    *bytes = (ctx->glayers[layer].end - ctx->glayers[layer].begin) * 4;
    return VIT_OK;
}

// Allocate the fp32 master weights (initialised from the loaded weights) and the transposed
// bf16 copies the data-gradient GEMMs read. Not on the hot path.
extern "C" int vit_train_init(vit_ctx* ctx, void* stream_) {
    VIT_TRY(check_train_ctx(ctx, "vit_train_init"));
    hipStream_t stream = (hipStream_t)stream_;
    if (!ctx->train) {
        TrainState* t = new TrainState();
        const vit_config& c = ctx->cfg;
        const size_t D = c.hidden_size, F = c.intermediate_size;
        hipError_t e = hipMalloc((void**)&t->master, ctx->g_total * 4);
        if (e != hipSuccess) { delete t; return vit_set_error(VIT_E_NOMEM, "vit_train_init: hipMalloc master failed"); }
        const size_t per_layer = align256(3 * D * D * 2) + align256(D * D * 2) + 2 * align256(D * F * 2);
        t->wt_bytes = per_layer * c.num_hidden_layers;
        e = hipMalloc((void**)&t->wt, t->wt_bytes);
        if (e != hipSuccess) { (void)hipFree(t->master); delete t; return vit_set_error(VIT_E_NOMEM, "vit_train_init: hipMalloc transposes failed"); }
        t->layers.resize(c.num_hidden_layers);
        size_t cur = 0;
        for (int l = 0; l < c.num_hidden_layers; ++l) {
            t->layers[l].wqkvT = t->wt + cur; cur += align256(3 * D * D * 2);
            t->layers[l].woT = t->wt + cur;   cur += align256(D * D * 2);
            t->layers[l].w1T = t->wt + cur;   cur += align256(D * F * 2);
            t->layers[l].w2T = t->wt + cur;   cur += align256(D * F * 2);
        }
        std::vector<ConvSlot> tab;
        for (auto& kv : ctx->slots) {
            const Slot& s = kv.second;
            const long long n = (long long)s.rows * s.pitch;
            const size_t es = esize(s.dtype);
            tab.push_back(ConvSlot{(long long)s.goff, (long long)s.offset, n, s.dtype,
                                   (s.goff % 4 == 0 && s.offset % (4 * es) == 0 && n % 4 == 0) ? 1 : 0});
        }
        t->nconv = (int)tab.size();
        e = hipMalloc((void**)&t->conv_table, tab.size() * sizeof(ConvSlot));
        if (e == hipSuccess) e = hipMemcpy(t->conv_table, tab.data(), tab.size() * sizeof(ConvSlot), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(t->master); (void)hipFree(t->wt);
            if (t->conv_table) (void)hipFree(t->conv_table);
            delete t;
            return vit_set_error(VIT_E_NOMEM, "vit_train_init: slot table upload failed: %s", hipGetErrorString(e));
        }
        std::vector<TransposeSlot> tt;
        int fb = 0;
        auto add_tr = [&](const void* src, void* dst, int R, int C) {
            tt.push_back(TransposeSlot{src, dst, R, C, fb, 0});
            fb += ceil_div(R, 64) * ceil_div(C, 64);
        };
        for (int l = 0; l < c.num_hidden_layers; ++l) {
            const LayerW& w = ctx->layers[l];
            add_tr(w.wqkv, t->layers[l].wqkvT, 3 * (int)D, (int)D);  // [3D,D] -> [D,3D]
            add_tr(w.wo, t->layers[l].woT, (int)D, (int)D);
            add_tr(w.w1, t->layers[l].w1T, (int)F, (int)D);          // [F,D] -> [D,F]
            add_tr(w.w2, t->layers[l].w2T, (int)D, (int)F);          // [D,F] -> [F,D]
        }
        t->ntr = (int)tt.size();
        t->tr_blocks = fb;
        e = hipMalloc((void**)&t->tr_table, tt.size() * sizeof(TransposeSlot));
        if (e == hipSuccess) e = hipMemcpy(t->tr_table, tt.data(), tt.size() * sizeof(TransposeSlot), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(t->master); (void)hipFree(t->wt); (void)hipFree(t->conv_table);
            if (t->tr_table) (void)hipFree(t->tr_table);
            delete t;
            return vit_set_error(VIT_E_NOMEM, "vit_train_init: transpose table upload failed: %s", hipGetErrorString(e));
        }
        ctx->train = t;
    }
    VIT_CHECK_HIP(hipMemsetAsync(ctx->train->master, 0, ctx->g_total * 4, stream));
    for (auto& kv : ctx->slots) {
        const Slot& s = kv.second;
        VIT_TRY(launch_convert_2d(ctx->arena + s.offset, s.dtype, ctx->train->master + s.goff, VIT_F32, s.rows, s.pitch,
                                  s.pitch, stream));
    }
    return vit_train_refresh_transposes(ctx, stream_);
}

// (re)build W^T for the data-gradient GEMMs from the current bf16 weights
extern "C" int vit_train_refresh_transposes(vit_ctx* ctx, void* stream_) {
    VIT_REQUIRE(ctx && ctx->train, "vit_train_refresh_transposes: call vit_train_init first");
    // D and F are multiples of 8 (vit_create), so the transposes need no row padding: one launch for all of them
    return launch_multi_transpose(ctx->train->tr_table, ctx->train->ntr, ctx->train->tr_blocks, (hipStream_t)stream_);
}

// ---------------------------------------------------------------- forward that stashes
extern "C" int vit_forward_train(vit_ctx* ctx, const void* pixels_dev, int pixel_dtype, int batch, void* hidden_out,
                                 void* pooled_out, void* stash_, size_t stash_bytes, void* workspace,
                                 size_t workspace_bytes, void* stream_) {
    VIT_TRY(check_train_ctx(ctx, "vit_forward_train"));
    VIT_REQUIRE(pixels_dev && hidden_out && stash_ && workspace, "vit_forward_train: null argument");
    VIT_REQUIRE(batch > 0, "vit_forward_train: batch must be positive");
    VIT_REQUIRE(pixel_dtype == VIT_F32 || pixel_dtype == VIT_BF16, "vit_forward_train: bad pixel dtype %d", pixel_dtype);
    const Stash st = plan_stash(ctx, batch);
    const TWork tw = plan_twork(ctx, batch);
    if (stash_bytes < st.total) return vit_set_error(VIT_E_NOMEM, "vit_forward_train: stash %zu < required %zu bytes", stash_bytes, st.total);
    if (workspace_bytes < tw.total) return vit_set_error(VIT_E_NOMEM, "vit_forward_train: workspace %zu < required %zu bytes", workspace_bytes, tw.total);
    hipStream_t stream = (hipStream_t)stream_;
    const vit_config& c = ctx->cfg;
    const TGeo g = tgeo(ctx, batch);
    unsigned char* S = (unsigned char*)stash_;
    unsigned char* ws = (unsigned char*)workspace;
    const bool clip = c.variant == VIT_VARIANT_CLIP;
    const int act_epi = c.hidden_act == VIT_ACT_GELU ? VIT_EPI_GELU : VIT_EPI_QUICK_GELU;
    const int T = VIT_BF16;

    // One contiguous range of images [b0, b0 + nb) on one stream: every stash tensor is [M, width], so a range
    // is a row offset into each of them.
    auto run = [&](int b0, int nb, hipStream_t stream) -> int {
        const size_t ro = (size_t)b0 * g.N;  // first token row of the range
        const int Mr = nb * g.N;
        auto R = [&](size_t off, int width) { return (void*)(S + off + ro * (size_t)width * 2); };
        const size_t img = (size_t)c.num_channels * c.image_size * c.image_size;
        const void* pix = (const char*)pixels_dev + (size_t)b0 * img * esize(pixel_dtype);
        if (pixel_dtype != T) {
            void* dst = ws + tw.pix + (size_t)b0 * img * 2;
            VIT_TRY(launch_convert(pix, pixel_dtype, dst, T, (int64_t)nb * img, stream));
            pix = dst;
        }
        void* x0 = R(clip ? st.xe : st.x[0], g.D);
        VIT_TRY(launch_cls_rows(ctx->cls, ctx->pos, x0, nb, g.N, g.D, T, stream));
        PatchGeom pg{c.num_channels, c.image_size, c.patch_size, c.image_size / c.patch_size, ctx->kreal};
        VIT_TRY(launch_gemm(pix, ctx->patch_w, ctx->patch_b, nullptr, x0, nb * (g.N - 1), g.D, ctx->kpad, VIT_EPI_BIAS, T, &pg,
                            ctx->pos, stream));
        if (clip)
            VIT_TRY(launch_layernorm(x0, ctx->pre_g, ctx->pre_b, R(st.x[0], g.D), Mr, g.D, g.D, g.D, c.layer_norm_eps, T, stream));
        for (int l = 0; l < g.L; ++l) {
            const LayerW& lw = ctx->layers[l];
            const Stash::Lay& y = st.lay[l];
            void* x = R(st.x[l], g.D);
            VIT_TRY(launch_layernorm(x, lw.ln1_g, lw.ln1_b, R(y.ln1, g.D), Mr, g.D, g.D, g.D, c.layer_norm_eps, T, stream));
            VIT_TRY(launch_gemm(R(y.ln1, g.D), lw.wqkv, lw.bqkv, nullptr, R(y.qkv, 3 * g.D), Mr, 3 * g.D, g.D, VIT_EPI_BIAS, T,
                                nullptr, nullptr, stream));
            float* lse = (float*)(S + y.lse) + (size_t)b0 * g.H * g.N;
            VIT_TRY(launch_attention(R(y.qkv, 3 * g.D), R(y.ctx, g.D), nb, g.N, g.H, ctx->dh, T, stream, lse));
            VIT_TRY(launch_gemm(R(y.ctx, g.D), lw.wo, lw.bo, x, R(y.xmid, g.D), Mr, g.D, g.D, VIT_EPI_RESIDUAL, T, nullptr,
                                nullptr, stream));
            VIT_TRY(launch_layernorm(R(y.xmid, g.D), lw.ln2_g, lw.ln2_b, R(y.ln2, g.D), Mr, g.D, g.D, g.D, c.layer_norm_eps, T,
                                     stream));
            GemmExt ext{R(y.u, g.F), 0, 1, nullptr};
            VIT_TRY(launch_gemm(R(y.ln2, g.D), lw.w1, lw.b1, nullptr, R(y.h, g.F), Mr, g.F, g.D, act_epi, T, nullptr, nullptr,
                                stream, &ext));
            VIT_TRY(launch_gemm(R(y.h, g.F), lw.w2, lw.b2, R(y.xmid, g.D), R(st.x[l + 1], g.D), Mr, g.D, g.F, VIT_EPI_RESIDUAL, T,
                                nullptr, nullptr, stream));
        }
        void* xl = R(st.x[g.L], g.D);
        void* hid = (char*)hidden_out + ro * g.D * 2;
        void* pool = pooled_out ? (char*)pooled_out + (size_t)b0 * g.D * 2 : nullptr;
        if (!clip) {
            VIT_TRY(launch_layernorm(xl, ctx->fin_g, ctx->fin_b, hid, Mr, g.D, g.D, g.D, c.layer_norm_eps, T, stream));
        } else {
            VIT_CHECK_HIP(hipMemcpyAsync(hid, xl, (size_t)Mr * g.D * 2, hipMemcpyDeviceToDevice, stream));
        }
        if (pool)
            VIT_TRY(launch_layernorm(xl, ctx->fin_g, ctx->fin_b, pool, nb, g.D, (int64_t)g.N * g.D, g.D, c.layer_norm_eps, T,
                                     stream));
        return VIT_OK;
    };

    // the stash counts as filled only once every launch of this call has been enqueued without an error
    ctx->train->filled.erase(stash_);
    auto mark_filled = [&](int rc) {
        if (rc == VIT_OK) {
            if (ctx->train->filled.size() >= TrainState::kMaxFilled) ctx->train->filled.clear();
            ctx->train->filled[stash_] = TrainState::Filled{batch, pixel_dtype};
        }
        return rc;
    };
    // as vit_forward: two chunks on two streams, so one chunk's kernels fill the CUs the other's last
    // round of GEMM tiles leaves idle (the stash keeps its single-batch layout)
    const int k = vit_num_chunks(ctx, batch) >= 2 ? 2 : 1;
    if (k == 1) return mark_filled(run(0, batch, stream));
    const int h0 = vit_chunk_begin(batch, 2, 1);
    VIT_CHECK_HIP(hipEventRecord(ctx->ev_fork, stream));
    VIT_CHECK_HIP(hipStreamWaitEvent(ctx->aux_stream[0], ctx->ev_fork, 0));
    // from here on the side stream is joined back whatever happens (an error must not leave it running)
    int rc = run(0, h0, stream);
    if (rc == VIT_OK) rc = run(h0, batch - h0, ctx->aux_stream[0]);
    hipError_t e = hipEventRecord(ctx->ev_join[0], ctx->aux_stream[0]);
    if (e == hipSuccess) e = hipStreamWaitEvent(stream, ctx->ev_join[0], 0);
    if (e != hipSuccess && rc == VIT_OK) rc = vit_set_error(VIT_E_HIP, "vit_forward_train: stream join: %s", hipGetErrorString(e));
    return mark_filled(rc);
}

// ---------------------------------------------------------------- backward, in stages
// stage 0: head (final norm), stages 1..L: layers L-1..0, stage L+1: embeddings. Stages must run
// in order on one stream; the running gradient of the residual stream lives in the workspace.
extern "C" int vit_backward_stage(vit_ctx* ctx, int stage, const void* pixels_dev, int pixel_dtype, int batch,
                                  const void* d_hidden, const void* d_pooled, const void* stash_, void* grads_,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
    VIT_TRY(check_train_ctx(ctx, "vit_backward_stage"));
    VIT_REQUIRE(ctx->train, "vit_backward_stage: call vit_train_init first");
    VIT_REQUIRE(stash_ && grads_ && workspace, "vit_backward_stage: null argument");
    VIT_REQUIRE(batch > 0, "vit_backward_stage: batch must be positive (got %d)", batch);
    const auto filled = ctx->train->filled.find(stash_);
    VIT_REQUIRE(filled != ctx->train->filled.end(), "vit_backward_stage: stash %p was not filled by a vit_forward_train of this context",
                stash_);
    VIT_REQUIRE(filled->second.batch == batch, "vit_backward_stage: batch %d does not match the vit_forward_train that filled stash %p (batch %d)",
                batch, stash_, filled->second.batch);
    const TGeo g = tgeo(ctx, batch);
    VIT_REQUIRE(stage >= 0 && stage <= g.L + 1, "vit_backward_stage: stage %d outside [0, %d]", stage, g.L + 1);
    const Stash st = plan_stash(ctx, batch);
    const TWork tw = plan_twork(ctx, batch);
    if (workspace_bytes < tw.total) return vit_set_error(VIT_E_NOMEM, "vit_backward_stage: workspace %zu < required %zu bytes", workspace_bytes, tw.total);
    hipStream_t stream = (hipStream_t)stream_;
    const vit_config& c = ctx->cfg;
    const unsigned char* S = (const unsigned char*)stash_;
    unsigned char* ws = (unsigned char*)workspace;
    float* G = (float*)grads_;
    const bool clip = c.variant == VIT_VARIANT_CLIP;
    const int dact = c.hidden_act == VIT_ACT_GELU ? VIT_EPI_DGELU : VIT_EPI_DQUICK_GELU;
    const int T = VIT_BF16;
    void* dx = ws + tw.dx;
    void* dx2 = ws + tw.dx2;
    void* dln = ws + tw.dln;
    const size_t md = (size_t)g.M * g.D * 2;

    if (stage == 0) {
        VIT_REQUIRE(d_hidden != nullptr, "vit_backward_stage: stage 0 needs d_hidden");
        VIT_CHECK_HIP(hipMemsetAsync(G, 0, ctx->g_total * 4, stream));  // vector gradients accumulate
        const void* xl = S + st.x[g.L];
        if (!clip) {
            // last_hidden_state = LayerNorm(x_L) (modeling_vit.py:385); pooled = its CLS row: the caller
            // folds d_pooled into d_hidden's CLS rows
            VIT_REQUIRE(d_pooled == nullptr, "vit_backward_stage: ViT variant: add d_pooled to the CLS rows of d_hidden");
            VIT_TRY(launch_layernorm_bwd(xl, d_hidden, ctx->fin_g, nullptr, dx, G + ctx->g_fin_g, G + ctx->g_fin_b, g.M, g.D,
                                         g.D, g.D, g.D, c.layer_norm_eps, stream, (float*)(ws + tw.lnred)));
        } else {
            // last_hidden_state = x_L; pooled = post_layernorm(x_L[:, 0]) (modeling_clip.py:649-656)
            VIT_CHECK_HIP(hipMemcpyAsync(dx, d_hidden, md, hipMemcpyDeviceToDevice, stream));
            if (d_pooled)
                VIT_TRY(launch_layernorm_bwd(xl, d_pooled, ctx->fin_g, dx, dx, G + ctx->g_fin_g, G + ctx->g_fin_b, batch, g.D,
                                             (int64_t)g.N * g.D, g.D, (int64_t)g.N * g.D, c.layer_norm_eps, stream));
        }
        return VIT_OK;
    }

    if (stage <= g.L) {
        const int l = g.L - stage;
        const LayerW& lw = ctx->layers[l];
        const LayerG& lg = ctx->glayers[l];
        const TrainState::LayerT& lt = ctx->train->layers[l];
        const Stash::Lay& y = st.lay[l];
        void* du = ws + tw.du;
        void* dqkv = ws + tw.dqkv;
        // Weight gradients run on a side stream: a data-gradient GEMM with N = 768 has 2.31 rounds of tiles on 256 CUs
        // and leaves most of the chip idle for its last round; the weight gradient of the same dY (one round of ~250
        // workgroups) starts on the CUs it frees. `to_side` hands a freshly written dY to the side stream.
        static const bool side_on = !(getenv("VIT_AMD_BWD_SIDE_STREAM") && atoi(getenv("VIT_AMD_BWD_SIDE_STREAM")) == 0);
        hipStream_t sb = (side_on && g.big && ctx->nstreams > 1 && ctx->aux_stream[0]) ? ctx->aux_stream[0] : stream;
        const bool two = sb != stream;
        auto to_side = [&]() -> int {
            if (two) {
                VIT_CHECK_HIP(hipEventRecord(ctx->ev_fork, stream));
                VIT_CHECK_HIP(hipStreamWaitEvent(sb, ctx->ev_fork, 0));
            }
            return VIT_OK;
        };
        auto layer_stage = [&]() -> int {
            // ---- MLP branch: x_out = x_mid + fc2(act(fc1(LN2(x_mid))))   (modeling_vit.py:280-284)
            VIT_TRY(to_side());
            VIT_TRY(wgrad(ctx, g, ws, tw, dx, g.D, S + y.h, g.F, g.M, 0, G + lg.w2, G + lg.b2, sb));
            if (two) VIT_CHECK_HIP(hipEventRecord(ctx->ev_join[0], sb));  // dx has been read
            VIT_TRY(launch_gemm(dx, lt.w2T, nullptr, S + y.u, du, g.M, g.F, g.D, dact, T, nullptr, nullptr, stream));
            VIT_TRY(launch_gemm(du, lt.w1T, nullptr, nullptr, dln, g.M, g.D, g.F, VIT_EPI_BIAS, T, nullptr, nullptr, stream));
            VIT_TRY(launch_layernorm_bwd(S + y.xmid, dln, lw.ln2_g, dx, dx2, G + lg.ln2_g, G + lg.ln2_b, g.M, g.D, g.D, g.D, g.D,
                                         c.layer_norm_eps, stream, (float*)(ws + tw.lnred)));
            // ---- attention branch: x_mid = x + o_proj(attn(qkv(LN1(x))))   (modeling_vit.py:273-277)
            // the fc1 weight gradient is handed over only here (du stays valid until the stage's join): beside the
            // LayerNorm backward it only starved that kernel, beside the attention backward (byte-bound) it is free: -0.7 %
            VIT_TRY(to_side());
            VIT_TRY(wgrad(ctx, g, ws, tw, du, g.F, S + y.ln2, g.D, g.M, 0, G + lg.w1, G + lg.b1, sb));
            VIT_TRY(wgrad(ctx, g, ws, tw, dx2, g.D, S + y.ctx, g.D, g.M, 0, G + lg.wo, G + lg.bo, sb));
            VIT_TRY(launch_gemm(dx2, lt.woT, nullptr, nullptr, dln, g.M, g.D, g.D, VIT_EPI_BIAS, T, nullptr, nullptr, stream));
            VIT_TRY(launch_attention_bwd(S + y.qkv, S + y.ctx, dln, (const float*)(S + y.lse), (float*)(ws + tw.delta), dqkv,
                                         batch, g.N, g.H, ctx->dh, stream));
            const bool has_qkv_bias = clip || c.qkv_bias;
            VIT_TRY(to_side());
            VIT_TRY(wgrad(ctx, g, ws, tw, dqkv, 3 * g.D, S + y.ln1, g.D, g.M, 0, G + lg.wqkv, has_qkv_bias ? G + lg.bqkv : nullptr,
                          sb));
            if (two) VIT_CHECK_HIP(hipEventRecord(ctx->ev_join[1], sb));  // the side stream's last kernel of this stage
            VIT_TRY(launch_gemm(dqkv, lt.wqkvT, nullptr, nullptr, dln, g.M, g.D, 3 * g.D, VIT_EPI_BIAS, T, nullptr, nullptr, stream));
            if (two) VIT_CHECK_HIP(hipStreamWaitEvent(stream, ctx->ev_join[0], 0));  // dx is overwritten next
            VIT_TRY(launch_layernorm_bwd(S + st.x[l], dln, lw.ln1_g, dx2, dx, G + lg.ln1_g, G + lg.ln1_b, g.M, g.D, g.D, g.D, g.D,
                                         c.layer_norm_eps, stream, (float*)(ws + tw.lnred)));
            // join: du, dx2, dqkv and the slabs are free again when the next stage (or the caller) goes on
            if (two) VIT_CHECK_HIP(hipStreamWaitEvent(stream, ctx->ev_join[1], 0));
            return VIT_OK;
        };
        const int rc = layer_stage();
        if (rc != VIT_OK && two) {
            // error path: whatever the side stream was handed must still be joined before the caller reuses the buffers
            if (hipEventRecord(ctx->ev_join[1], sb) == hipSuccess) (void)hipStreamWaitEvent(stream, ctx->ev_join[1], 0);
        }
        return rc;
    }

    // ---- embeddings (modeling_vit.py:129-161, 62-69; modeling_clip.py:202-218, 642)
    VIT_REQUIRE(pixels_dev != nullptr, "vit_backward_stage: the embedding stage needs the pixels");
    VIT_REQUIRE(pixel_dtype == filled->second.pixel_dtype, "vit_backward_stage: pixel dtype %d differs from vit_forward_train's %d",
                pixel_dtype, filled->second.pixel_dtype);
    const void* dxe = dx;
    if (clip) {
        VIT_TRY(launch_layernorm_bwd(S + st.xe, dx, ctx->pre_g, nullptr, dx2, G + ctx->g_pre_g, G + ctx->g_pre_b, g.M, g.D,
                                     g.D, g.D, g.D, c.layer_norm_eps, stream, (float*)(ws + tw.lnred)));
        dxe = dx2;
    }
    VIT_TRY(launch_embed_grad(dxe, G + ctx->g_pos, G + ctx->g_cls, batch, g.N, g.D, stream));
    {
        const int Mtok = batch * (g.N - 1);
        int splits = g.big ? wgrad_splits(g.D, g.Kp) : 1;
        if ((long)g.D * g.Kp * splits < 64L * 256 * 256) splits = 1;
        const int mp = round_up(Mtok, 128 * splits);
        if (mp / splits < 256) splits = 1;
        const void* pix = pixels_dev;
        PatchGeom pg{c.num_channels, c.image_size, c.patch_size, c.image_size / c.patch_size, ctx->kreal};
        VIT_TRY(launch_im2col_t(pix, pixel_dtype, ws + tw.t2, Mtok, mp, &pg, ctx->kpad, stream));
        VIT_TRY(launch_transpose_pad(dxe, ws + tw.t1, clip ? nullptr : G + ctx->g_patch_b, Mtok, g.D, mp, g.N - 1, stream));
        GemmExt ext{nullptr, 1, splits, (float*)(ws + tw.slabs)};
        VIT_TRY(launch_gemm(ws + tw.t1, ws + tw.t2, nullptr, nullptr, G + ctx->g_patch_w, g.D, ctx->kpad, mp, VIT_EPI_BIAS, T,
                            nullptr, nullptr, stream, &ext));
    }
    return VIT_OK;
}

// ---------------------------------------------------------------- optimizer
// master -= lr * grad for every parameter, then refresh the bf16 / fp32 compute copies and
// the transposed weights. `grads_` has the gradient-arena layout (after the all-reduce).
extern "C" int vit_sgd_step(vit_ctx* ctx, const void* grads_, float lr, void* stream_) {
    VIT_TRY(check_train_ctx(ctx, "vit_sgd_step"));
    VIT_REQUIRE(ctx->train && grads_, "vit_sgd_step: call vit_train_init first");
    hipStream_t stream = (hipStream_t)stream_;
    VIT_TRY(launch_sgd(ctx->train->master, (const float*)grads_, nullptr, lr, (int64_t)ctx->g_total, stream));
    ctx->fold_dirty = true;  // the inference forward's folded LayerNorm copies are stale now
    VIT_TRY(launch_multi_convert(ctx->train->master, ctx->arena, ctx->train->conv_table, ctx->train->nconv, stream));
    return vit_train_refresh_transposes(ctx, stream_);
}
