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

// One-time launch set-up is PER DEVICE (a function attribute is set on the current device's copy of the code object;
// CU counts and occupancy are device properties): a `static PerDeviceInt` in a launcher holds one int per device,
// 0 = nothing done yet on that device. Concurrent first launches from two threads write the same value (relaxed
// atomics; hipFuncSetAttribute is idempotent).
#include <atomic>
struct PerDeviceInt {
    static constexpr int kMax = 64;
    std::atomic<int> v[kMax];
    // slot of the current device, or nullptr (device id out of range / no current device: do the set-up every time)
    std::atomic<int>* slot() {
        int d = -1;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMax) return nullptr;
        return &v[d];
    }
};
// CU count of the current device (cached per device)
static inline int vit_cu_count(int* ncu) {
    static PerDeviceInt cache;
    std::atomic<int>* s = cache.slot();
    int n = s ? s->load(std::memory_order_relaxed) : 0;
    if (n == 0) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return vit_set_error(VIT_E_HIP, "CU count: %s", hipGetErrorString(e));
        if (s) s->store(n, std::memory_order_relaxed);
    }
    *ncu = n;
    return VIT_OK;
}
#define VIT_TRY_RC(expr)              \
    do {                              \
        int _rc = (expr);             \
        if (_rc != VIT_OK) return _rc; \
    } while (0)

// ---- LayerNorm row statistics in one pass (the folded LayerNorm of vit_forward) ----
// Two kernels produce them and must agree bit for bit (a small batch takes the stand-alone pass, a big one gets the sums
// from the residual epilogue of the GEMM that wrote the rows): both cut a row into groups of 64 columns, give the 16 lanes
// of a DPP row 4 consecutive stored bf16 values each (two packed pairs d0, d1), and run THIS instruction sequence:
// v_dot2c_f32_bf16 (exact products, fp32 accumulation: sm = ((0 + x0 + x1) + x2 + x3) in the instruction's own order), then
// the sum over the 16 lanes by four rotate-and-add steps. Only lane 0's result is used: it is
// (((a0 + a8) + (a12 + a4)) + ((a14 + a6) + (a10 + a2))) + (the same tree over the odd lanes). Hand-written: through the
// DPP builtin the compiler emits a zeroed "old" register and a separate move per step (30 instructions per row instead of
// 18). Inline asm gets no hazard handling from the compiler: a DPP operand needs two wait states behind the VALU write of its
// register (the two interleaved chains + s_nop 0), and a VALU read of a v_dot2c result three (s_nop 1 + the other chain's
// instruction); a v_dot2c accumulating onto its own kind's result needs none.
__device__ __forceinline__ void row16_stats(unsigned d0, unsigned d1, float& sm, float& sq) {
    sm = 0.f;
    sq = 0.f;
    asm("v_dot2c_f32_bf16 %0, %2, %4\n\t"
        "v_dot2c_f32_bf16 %1, %2, %2\n\t"
        "v_dot2c_f32_bf16 %0, %3, %4\n\t"
        "v_dot2c_f32_bf16 %1, %3, %3\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf"
        : "+v"(sm), "+v"(sq)
        : "v"(d0), "v"(d1), "v"(0x3F803F80u));
}
// (sum, sum of squares) of a whole row -> (mean * rstd, rstd). One pass: var = E[x^2] - mean^2 in fp32, clamped at 0; its
// relative error is about 2^-23 (1 + mean^2 / var) sqrt(dim), i.e. below bf16's own rounding of the inputs until
// |mean| reaches a few hundred standard deviations. Every operation pinned (no contraction): same bits in both kernels.
__device__ __forceinline__ float2 ln_from_sums(float sm, float sq, float inv_d, float eps) {
    const float mean = __fmul_rn(sm, inv_d);
    const float var = fmaxf(__fmaf_rn(sq, inv_d, -__fmul_rn(mean, mean)), 0.f);
    const float rstd = 1.0f / sqrtf(__fadd_rn(var, eps));
    return make_float2(__fmul_rn(mean, rstd), rstd);
}

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
// exposed cost (128 elements per lane and tile): no transcendental, every operation packed (2 elements per issue).
//   gelu(x) = x Phi(x),  Phi(x) = 1/2 + x q(x^2),  x q(x^2) = erf(x / sqrt2) / 2
// with q a degree-8 minimax-style fit (Chebyshev-node least squares of erf(z) = z h(z^2) on z^2 in [0, 9], rescaled
// to x^2). Phi is clamped to [0, 1] by the CLAMP bit of the packed fma itself: beyond the fitted interval the
// polynomial keeps |x q| >= 1/2 + 7e-6 (checked to |x| = 56; its leading coefficient is positive), so both tails are
// exact without clamping x^2. 11 packed instructions per pair (the form with z = x / sqrt2, min(z^2, 9), med3 and
// x/2 (1 + erf) took 13 packed + 4 single ones).
// |gelu error| <= 8.5e-5 over all x (largest near |x| = 4.2, where one bf16 ulp of the result is
// 1.6e-2; <= 2e-5 for |x| <= 2), i.e. below the bf16 rounding of the stored activation.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, float c) { return __builtin_elementwise_fma(a, b, f32x2{c, c}); }
// Phi(x) = clamp(1/2 + x q(x^2), 0, 1) on two elements
__device__ __forceinline__ f32x2 act_phi_pk(f32x2 x, f32x2 u) {
    // every multiply-add is an explicit fma: the same bits from every kernel that inlines this,
    // whatever the compiler would have contracted
    f32x2 p = pk_fma(u, f32x2{6.79247283419393e-11f, 6.79247283419393e-11f}, -6.263780161219756e-09f);
    p = pk_fma(p, u, 2.549301711951557e-07f);
    p = pk_fma(p, u, -6.116001259215409e-06f);
    p = pk_fma(p, u, 9.806382877286524e-05f);
    p = pk_fma(p, u, -0.0011335634626448154f);
    p = pk_fma(p, u, 0.009887341409921646f);
    p = pk_fma(p, u, -0.06643597036600113f);
    p = pk_fma(p, u, 0.39893653988838196f);
    f32x2 phi;  // clamp(x p + 1/2, 0, 1): the compiler spends two v_max ... clamp on this, the instruction has the bit
    asm("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(phi) : "v"(x), "v"(p));
    return phi;
}
__device__ __forceinline__ f32x2 act_gelu_pk(f32x2 x) { return x * act_phi_pk(x, x * x); }
// two pairs at once, the two Horner chains interleaved: a packed fma that reads the previous one's result needs a wait
// state (the compiler fills it with s_nop when it has only one chain to schedule). Same bits as act_gelu_pk.
__device__ __forceinline__ void act_gelu_pk2(f32x2& xa, f32x2& xb) {
    const f32x2 ua = xa * xa, ub = xb * xb;
    f32x2 pa = pk_fma(ua, f32x2{6.79247283419393e-11f, 6.79247283419393e-11f}, -6.263780161219756e-09f);
    f32x2 pb = pk_fma(ub, f32x2{6.79247283419393e-11f, 6.79247283419393e-11f}, -6.263780161219756e-09f);
#define VIT_GELU_STEP(C) pa = pk_fma(pa, ua, C); pb = pk_fma(pb, ub, C)
    VIT_GELU_STEP(2.549301711951557e-07f);
    VIT_GELU_STEP(-6.116001259215409e-06f);
    VIT_GELU_STEP(9.806382877286524e-05f);
    VIT_GELU_STEP(-0.0011335634626448154f);
    VIT_GELU_STEP(0.009887341409921646f);
    VIT_GELU_STEP(-0.06643597036600113f);
    VIT_GELU_STEP(0.39893653988838196f);
#undef VIT_GELU_STEP
    f32x2 fa, fb;
    asm("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(fa) : "v"(xa), "v"(pa));
    asm("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(fb) : "v"(xb), "v"(pb));
    xa = xa * fa;
    xb = xb * fb;
}
// the same function on one element (generic GEMM kernel): identical bits to act_gelu_pk
__device__ __forceinline__ float act_gelu_poly(float x) { return act_gelu_pk(f32x2{x, x}).x; }

// gelu'(x) = Phi(x) + x phi(x) (backward of ViTMLP's activation): Phi from the packed polynomial above (|error| <= 2e-5),
// phi(x) = exp2(-x^2 log2(e) / 2) / sqrt(2 pi). 13 packed instructions + 2 exponentials per pair; the form it replaces
// (Abramowitz-Stegun 7.1.26 erfc with a reciprocal and an exponential per element, ~25 issue slots each) made the
// fc2 data-gradient GEMM's epilogue a quarter of its launch.
__device__ __forceinline__ f32x2 act_dgelu_pk(f32x2 x) {
    const f32x2 u = x * x;
    const f32x2 phi = act_phi_pk(x, u);
    const f32x2 a = u * (-0.5f * 1.4426950408889634f);
    const f32x2 xe = x * f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    return __builtin_elementwise_fma(xe, f32x2{0.3989422804014327f, 0.3989422804014327f}, phi);
}
__device__ __forceinline__ float act_dgelu_as(float x) { return act_dgelu_pk(f32x2{x, x}).x; }
__device__ __forceinline__ float act_dquick_gelu(float x) {
    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * (-1.702f * 1.4426950408889634f)));
    return sg * (1.0f + 1.702f * x * (1.0f - sg));
}
// quick_gelu of the bf16 kernels: x * rcp(1 + exp2(-1.702 log2(e) x)). The IEEE division of act_quick_gelu is a dozen
// instructions per element (v_div_scale x2, v_rcp, five fmas, v_div_fmas, v_div_fixup); v_rcp_f32 alone is good to
// 1 ulp of fp32, far below the bf16 rounding of the stored activation. The fp32 kernels keep the division.
__device__ __forceinline__ f32x2 act_quick_gelu_pk(f32x2 x) {
    const f32x2 a = x * (-1.702f * 1.4426950408889634f);
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)} + 1.0f;
    return x * f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
__device__ __forceinline__ float act_quick_gelu_fast(float x) { return act_quick_gelu_pk(f32x2{x, x}).x; }
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
    if constexpr (EPI == VIT_EPI_QUICK_GELU) return act_quick_gelu_fast(v);
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
    // the caller runs another chunk of the same forward on a second stream at the same time: keep 256-row tiles
    // (gemm_bf16_8p.hip, few-tile planner)
    int concurrent;
    // residual epilogue of the persistent kernel (N % 64 == 0): also write, per 64-column group g and row r, the pair
    // (sum, sum of squares) of the stored bf16 values to row_part[(g * M + r) * 2] -- N / 64 slabs of M float pairs, turned
    // into the next folded LayerNorm's statistics by launch_row_stats_finalize (layernorm.hip)
    float* row_part;
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
// stats[row] from the N / 64 slabs of (sum, sum of squares) a residual GEMM wrote (GemmExt::row_part): the same numbers, bit
// for bit, as launch_ln_stats on the rows that GEMM stored (dim % 64 == 0)
bool gemm_writes_row_stats(int M, int N, int K);  // gemm_bf16.hip
bool ln_stats_one_pass(int dim);                  // layernorm.hip: launch_ln_stats takes the one-pass kernel for this width
int launch_row_stats_finalize(const float* row_part, float* stats, int rows, int dim, float eps, hipStream_t stream);
int launch_ln_fold(const void* W, int64_t pitch, const float* gamma, const float* beta, const float* bias, void* Wg, float* c,
                   float* d, int N, int K, hipStream_t stream);
int launch_convert(const void* src, int sdt, void* dst, int ddt, int64_t n, hipStream_t stream);
int launch_im2col_t(const void* pixels, int pixel_dtype, void* dst, int R, int Rpad, const PatchGeom* g, int kpad,
                    hipStream_t stream);
int launch_cls_rows(const float* cls, const float* pos, void* x, int batch, int seq, int dim, int dtype,
                    hipStream_t stream);
