// bf16 MFMA GEMM with fused epilogues:  C[M,N] = epi(A[M,K] * W[N,K]^T + bias (+ residual))
//
// Computes nn.Linear (+activation / +residual) of the encoder: q/k/v packed projection
// (modeling_vit.py:216-218), o_proj + residual (:236,:277), fc1 + GELU (:250-251), fc2 +
// residual (:252,:284), and -- with the patch A-loader -- Conv2d(kernel=stride=P) of
// ViTPatchEmbeddings.forward (:62-69) without an im2col buffer, with the CLS-slot row remap
// and position add of ViTEmbeddings.forward (:129-161) fused in the epilogue.
//
// gfx950 design (see DESIGN.md "GEMM"):
//  * 256 x BN x 64 block tile (BN = 256 or 128), 8 waves (512 threads), 1 block / CU.
//  * Operands go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4), two LDS buffers, one
//    barrier per K-tile; the next tile's DMA flies under the current tile's MFMAs.
//  * LDS rows are 128 B (64 bf16); 16-B chunk c of row r is stored at chunk position
//    c ^ ((r>>1)&7). The DMA destination is lane-linear, so the swizzle is applied to the
//    per-lane SOURCE address and to the ds_read_b128 address (conflict-free reads).
//  * v_mfma_f32_16x16x32_bf16 with SWAPPED operands (A-operand = weight rows, B-operand =
//    activation rows) so every lane ends up with 4 consecutive output columns per fragment;
//    the weight rows are staged in a permuted order so that the 4 fragments of a wave give a
//    lane 16 CONSECUTIVE columns -> two 16-B global stores per row, full 128-B lines per
//    16-lane group.
//  * blockIdx -> tile map is XCD-aware (bijective remap): the tiles an XCD runs back to back
//    share their activation panel in that XCD's L2.
#include "vit_common.h"

namespace {

constexpr int BM = 256;
constexpr int BK = 64;
constexpr int NTHREADS = 512;

enum AMode { A_DENSE = 0, A_PATCH_ALIGNED = 1, A_PATCH_GATHER = 2 };
constexpr int EPI_PATCH = 100;  // internal: bias + pos add + row remap past the CLS slot

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0, 0, 0, 0};

struct GemmParams {
    const bf16_t* A;
    const bf16_t* W;
    const float* bias;
    const bf16_t* res;
    bf16_t* C;      // bf16 output, or fp32 output when out_f32
    bf16_t* Cpre;   // optional second output: pre-activation (activation epilogues)
    int out_f32;
    const float* pos;
    int M, N, K;
    int NT;  // tiles along N
    PatchGeom pg;
    const float* ln_stats;  // LNF: [M][2] = (mean * rstd, rstd) of the un-normalised A rows
    const float* ln_c;      // LNF: [N] column sums of the gamma-folded weight
};

template <int BN>
struct Geo {
    static constexpr int WAVES_M = (BN == 256) ? 2 : 4;
    static constexpr int WAVES_N = 8 / WAVES_M;
    static constexpr int WM = BM / WAVES_M;
    static constexpr int WN = BN / WAVES_N;
    static constexpr int MI = WM / 16;
    static constexpr int NI = WN / 16;
    static_assert(WN == 64 && NI == 4, "epilogue assumes 16 consecutive columns per lane");
    static constexpr int A_ELEMS = BM * BK;
    static constexpr int W_ELEMS = BN * BK;
    static constexpr int LDS_BYTES = 2 * (A_ELEMS + W_ELEMS) * 2;
};

// LDS row r' of the weight tile holds this tile-local output column (see header comment).
__device__ __forceinline__ int w_row_perm(int rp) {
    const int wn = rp >> 6, rr = rp & 63, ni = rr >> 4, i = rr & 15;
    return (wn << 6) + ((i >> 2) << 4) + (ni << 2) + (i & 3);
}

__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_wave_base), 16, 0, 0);
}

template <int BN, int AMODE>
struct Stager {
    // per-thread constant parts
    const bf16_t* a_src[4];   // row base pointers (dense) / patch base (patch modes)
    const bf16_t* w_src[BN / 64];
    int a_sw[4];              // swizzle term (r>>1)&7 of each staged row
    int w_sw[BN / 64];
    int p;                    // chunk position inside the 128-B LDS row
    int wave;

    __device__ __forceinline__ void init(const GemmParams& P, int m0, int n0, int tid) {
        p = tid & 7;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = t * 64 + (tid >> 3);
            a_sw[t] = (r >> 1) & 7;
            int gm = m0 + r;
            if (gm > P.M - 1) gm = P.M - 1;
            if constexpr (AMODE == A_DENSE) {
                a_src[t] = P.A + (size_t)gm * P.K;
            } else {
                const int np = P.pg.grid * P.pg.grid;
                const int b = gm / np, rem = gm - b * np;
                const int py = rem / P.pg.grid, px = rem - py * P.pg.grid;
                a_src[t] = P.A + ((size_t)b * P.pg.channels * P.pg.image + (size_t)py * P.pg.patch) * P.pg.image +
                           (size_t)px * P.pg.patch;
            }
        }
#pragma unroll
        for (int t = 0; t < BN / 64; ++t) {
            const int rp = t * 64 + (tid >> 3);
            w_sw[t] = (rp >> 1) & 7;
            int gn = n0 + w_row_perm(rp);
            if (gn > P.N - 1) gn = P.N - 1;
            w_src[t] = P.W + (size_t)gn * P.K;
        }
    }

    // element offset of patch-matrix column kk inside one image patch (relative to a_src)
    __device__ __forceinline__ size_t patch_off(const GemmParams& P, int kk) const {
        const int pp = P.pg.patch * P.pg.patch;
        const int ch = kk / pp, r2 = kk - ch * pp;
        const int i = r2 / P.pg.patch, j = r2 - i * P.pg.patch;
        return ((size_t)ch * P.pg.image + i) * P.pg.image + j;
    }

    __device__ __forceinline__ void stage(const GemmParams& P, bf16_t* As, bf16_t* Ws, int k0) const {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kk = k0 + ((p ^ a_sw[t]) << 3);
            bf16_t* dst = As + (t * 64 + wave * 8) * BK;  // wave-uniform; DMA adds lane*16 B
            if constexpr (AMODE == A_DENSE) {
                const void* src = (kk < P.K) ? (const void*)(a_src[t] + kk) : (const void*)g_zero16;
                glds16(src, dst);
            } else if constexpr (AMODE == A_PATCH_ALIGNED) {
                const void* src = (kk < P.pg.kreal) ? (const void*)(a_src[t] + patch_off(P, kk)) : (const void*)g_zero16;
                glds16(src, dst);
            } else {
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k1 = kk + e;
                    v[e] = (k1 < P.pg.kreal) ? a_src[t][patch_off(P, k1)] : (bf16_t)0.0f;
                }
                const int lane = threadIdx.x & 63;
                *(bf16x8*)(dst + lane * 8) = v;
            }
        }
#pragma unroll
        for (int t = 0; t < BN / 64; ++t) {
            const int kk = k0 + ((p ^ w_sw[t]) << 3);
            const void* src = (kk < P.K) ? (const void*)(w_src[t] + kk) : (const void*)g_zero16;
            glds16(src, Ws + (t * 64 + wave * 8) * BK);
        }
    }
};

// LNF: LayerNorm folded in (GemmExt::ln_stats): value = rstd * acc - (mean * rstd) * c + bias, the same
// fmaf chain as the persistent kernel's epilogue so that both kernels give the same bits.
template <int BN, int AMODE, int EPI, bool LNF = false>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_kernel(const GemmParams P) {
    using G = Geo<BN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* As = (bf16_t*)smem;                          // [2][BM*BK]
    bf16_t* Ws = (bf16_t*)smem + 2 * G::A_ELEMS;         // [2][BN*BK]

    // XCD-aware bijective remap of the linear block id (blocks b and b+8 share an XCD)
    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    const int mt = tile / P.NT, nt = tile - mt * P.NT;
    const int m0 = mt * BM, n0 = nt * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / G::WAVES_N, wn = wave - wm * G::WAVES_N;
    const int i = lane & 15, q = lane >> 4;

    Stager<BN, AMODE> st;
    st.init(P, m0, n0, tid);

    f32x4 acc[G::MI][G::NI];
#pragma unroll
    for (int a = 0; a < G::MI; ++a)
#pragma unroll
        for (int b = 0; b < G::NI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (P.K + BK - 1) / BK;
    st.stage(P, As, Ws, 0);

    const int a_off = (wm * G::WM + i) * BK;
    const int w_off = (wn * G::WN + i) * BK;
    const int sw = i >> 1;

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // tile kt landed for every wave; buffer buf^1 is free again
        if (kt + 1 < nk) st.stage(P, As + (buf ^ 1) * G::A_ELEMS, Ws + (buf ^ 1) * G::W_ELEMS, (kt + 1) * BK);
        const bf16_t* ab = As + buf * G::A_ELEMS + a_off;
        const bf16_t* wb = Ws + buf * G::W_ELEMS + w_off;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int pos = ((4 * s + q) ^ sw) << 3;
            bf16x8 wf[G::NI], xf[G::MI];
#pragma unroll
            for (int b = 0; b < G::NI; ++b) wf[b] = *(const bf16x8*)(wb + b * 16 * BK + pos);
#pragma unroll
            for (int a = 0; a < G::MI; ++a) xf[a] = *(const bf16x8*)(ab + a * 16 * BK + pos);
#pragma unroll
            for (int a = 0; a < G::MI; ++a)
#pragma unroll
                for (int b = 0; b < G::NI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
        }
    }

    // ---- epilogue: lane (i, q) owns, for every mi, row m0+wm*WM+mi*16+i and the 16
    // consecutive columns n0 + wn*64 + q*16 + [0,16): acc[mi][ni][j] is column ni*4+j.
    const int nb = n0 + wn * G::WN + q * 16;
    float bias[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bias[e] = (P.bias != nullptr && nb + e < P.N) ? P.bias[nb + e] : 0.f;

    float lnc[16];
    if constexpr (LNF) {
#pragma unroll
        for (int e = 0; e < 16; ++e) lnc[e] = (nb + e < P.N) ? P.ln_c[nb + e] : 0.f;
    }
    const int np = (EPI == EPI_PATCH) ? P.pg.grid * P.pg.grid : 1;
#pragma unroll
    for (int a = 0; a < G::MI; ++a) {
        const int m = m0 + wm * G::WM + a * 16 + i;
        if (m >= P.M) continue;
        float v[16];
        if constexpr (LNF) {
            const float2 st = *(const float2*)(P.ln_stats + (size_t)m * 2);
#pragma unroll
            for (int b = 0; b < G::NI; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[b * 4 + j] = fmaf(st.y, acc[a][b][j], fmaf(-st.x, lnc[b * 4 + j], bias[b * 4 + j]));
        } else {
#pragma unroll
            for (int b = 0; b < G::NI; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[b * 4 + j] = acc[a][b][j] + bias[b * 4 + j];
        }
        size_t orow = (size_t)m;
        if constexpr (EPI == EPI_PATCH) {
            const int bimg = m / np, pr = m - bimg * np + 1;  // token index inside the image (CLS = 0)
            orow = (size_t)m + bimg + 1;
            const float* pp = P.pos + (size_t)pr * P.N + nb;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (nb + e < P.N) v[e] += pp[e];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (nb + h * 8 >= P.N) continue;
            const size_t o = orow * P.N + nb + h * 8;
            if constexpr (EPI == VIT_EPI_GELU || EPI == VIT_EPI_QUICK_GELU) {
                if (P.Cpre != nullptr) {
                    bf16x8 pv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pv[e] = (bf16_t)v[h * 8 + e];
                    *(bf16x8*)(P.Cpre + o) = pv;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[h * 8 + e] = apply_act_bf16<EPI>(v[h * 8 + e]);
            }
            if constexpr (EPI == VIT_EPI_RESIDUAL || EPI == VIT_EPI_DGELU || EPI == VIT_EPI_DQUICK_GELU) {
                const bf16x8 r = *(const bf16x8*)(P.res + o);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if constexpr (EPI == VIT_EPI_RESIDUAL) v[h * 8 + e] += (float)r[e];
                    else v[h * 8 + e] *= act_grad<EPI>((float)r[e]);
                }
            }
            if (P.out_f32) {
                float* cf = (float*)P.C + o;
                *(f32x4*)cf = f32x4{v[h * 8 + 0], v[h * 8 + 1], v[h * 8 + 2], v[h * 8 + 3]};
                *(f32x4*)(cf + 4) = f32x4{v[h * 8 + 4], v[h * 8 + 5], v[h * 8 + 6], v[h * 8 + 7]};
            } else {
                bf16x8 ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) ov[e] = (bf16_t)v[h * 8 + e];
                *(bf16x8*)(P.C + o) = ov;
            }
        }
    }
}

template <int BN, int AMODE, int EPI, bool LNF = false>
int launch_one(const GemmParams& P, hipStream_t stream) {
    using G = Geo<BN>;
    auto kern = gemm_bf16_kernel<BN, AMODE, EPI, LNF>;
    static PerDeviceInt attr_set;  // per instantiation and device; the attribute is idempotent
    std::atomic<int>* as = attr_set.slot();
    if (!as || !as->load(std::memory_order_relaxed)) {
        VIT_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
        if (as) as->store(1, std::memory_order_relaxed);
    }
    const int MT = ceil_div(P.M, BM);
    dim3 grid(MT * P.NT), block(NTHREADS);
    hipLaunchKernelGGL(kern, grid, block, G::LDS_BYTES, stream, P);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

template <int BN, int AMODE>
int dispatch_epi(const GemmParams& P, int epi, hipStream_t stream) {
    if constexpr (AMODE != A_DENSE) {
        return launch_one<BN, AMODE, EPI_PATCH>(P, stream);
    } else {
        if (P.ln_stats) {
            switch (epi) {
                case VIT_EPI_BIAS: return launch_one<BN, AMODE, VIT_EPI_BIAS, true>(P, stream);
                case VIT_EPI_GELU: return launch_one<BN, AMODE, VIT_EPI_GELU, true>(P, stream);
                case VIT_EPI_QUICK_GELU: return launch_one<BN, AMODE, VIT_EPI_QUICK_GELU, true>(P, stream);
            }
            return vit_set_error(VIT_E_INVALID, "gemm: folded LayerNorm with epilogue %d", epi);
        }
        switch (epi) {
            case VIT_EPI_BIAS: return launch_one<BN, AMODE, VIT_EPI_BIAS>(P, stream);
            case VIT_EPI_GELU: return launch_one<BN, AMODE, VIT_EPI_GELU>(P, stream);
            case VIT_EPI_QUICK_GELU: return launch_one<BN, AMODE, VIT_EPI_QUICK_GELU>(P, stream);
            case VIT_EPI_RESIDUAL: return launch_one<BN, AMODE, VIT_EPI_RESIDUAL>(P, stream);
            case VIT_EPI_DGELU: return launch_one<BN, AMODE, VIT_EPI_DGELU>(P, stream);
            case VIT_EPI_DQUICK_GELU: return launch_one<BN, AMODE, VIT_EPI_DQUICK_GELU>(P, stream);
        }
        return vit_set_error(VIT_E_INVALID, "gemm: unknown epilogue %d", epi);
    }
}

// Tile-count heuristic: prefer the 256x256 tile unless the 256x128 tile wastes markedly
// less of the last round of 256 CUs (one block per CU).
int pick_bn(int M, int N) {
    if (N <= 128) return 128;
    auto eff = [&](int bn) {
        const long tiles = (long)ceil_div(M, BM) * ceil_div(N, bn);
        const long rounds = (tiles + 255) / 256;
        return (double)((long)M * N) / ((double)rounds * 256.0 * BM * bn);
    };
    return (eff(128) > eff(256) * 1.06) ? 128 : 256;
}

}  // namespace

bool gemm_bf16_8p_eligible(int M, int N, int K);
int launch_gemm_bf16_8p(const void* A, const void* W, const float* bias, const void* res, void* C, int M, int N, int K,
                        int epi, const GemmExt* ext, hipStream_t stream);
bool gemm_bf16_8p_patch_eligible(int M, int N, const PatchGeom* g);
int launch_gemm_bf16_8p_patch(const void* pixels, const void* W, const float* bias, const float* pos, void* C, int M, int N,
                              const PatchGeom* g, hipStream_t stream);

// generic tile kernel of this file (any K % 8 == 0, patch loaders); also the lab's baseline
int launch_gemm_bf16_generic(const void* A, const void* W, const float* bias, const void* res, void* C, int M, int N,
                             int K, int epi, const PatchGeom* patch, const float* pos, hipStream_t stream,
                             const GemmExt* ext = nullptr);

int launch_gemm_bf16(const void* A, const void* W, const float* bias, const void* res, void* C, int M, int N, int K,
                     int epi, const PatchGeom* patch, const float* pos, hipStream_t stream, const GemmExt* ext) {
    VIT_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
    VIT_REQUIRE(K % 8 == 0 && N % 8 == 0, "gemm(bf16): K and N must be multiples of 8 (K=%d N=%d)", K, N);
    VIT_REQUIRE((epi != VIT_EPI_RESIDUAL && epi != VIT_EPI_DGELU && epi != VIT_EPI_DQUICK_GELU) || res != nullptr,
                "gemm: epilogue %d needs the residual / pre-activation operand", epi);
    // big Linear layers go to the persistent deep-pipelined kernel (gemm_bf16_8p.hip); tiny
    // problems (less than one 256x256 tile of work per 4 CUs) and K tails stay here
    const int splits = (ext && ext->split_k > 1) ? ext->split_k : 1;
    if (patch == nullptr && gemm_bf16_8p_eligible(M, N, K) && (long)M * N * splits >= 64L * 256 * 256)
        return launch_gemm_bf16_8p(A, W, bias, res, C, M, N, K, epi, ext, stream);
    VIT_REQUIRE(!(ext && ext->row_part), "gemm: row statistics requested for a shape the persistent kernel does not take (ask gemm_writes_row_stats first)");
    // patch embedding with whole-pixel-row K-tiles (P = 16, 32): persistent kernel with the pixel loader
    if (patch != nullptr && ext == nullptr && epi == VIT_EPI_BIAS && pos != nullptr && K == patch->kreal &&
        gemm_bf16_8p_patch_eligible(M, N, patch) && (long)M * N >= 64L * 256 * 256)
        return launch_gemm_bf16_8p_patch(A, W, bias, pos, C, M, N, patch, stream);
    VIT_REQUIRE(splits == 1, "gemm: split-K requested for a shape the persistent kernel does not take");
    return launch_gemm_bf16_generic(A, W, bias, res, C, M, N, K, epi, patch, pos, stream, ext);
}

// true when a bf16 residual GEMM of this shape goes to the persistent kernel and can leave the row sums of its output
// (GemmExt::row_part) -- the same test as the dispatch above
bool gemm_writes_row_stats(int M, int N, int K) {
    return N % 64 == 0 && gemm_bf16_8p_eligible(M, N, K) && (long)M * N >= 64L * 256 * 256;
}

int launch_gemm_bf16_generic(const void* A, const void* W, const float* bias, const void* res, void* C, int M, int N,
                             int K, int epi, const PatchGeom* patch, const float* pos, hipStream_t stream,
                             const GemmExt* ext) {
    GemmParams P{};
    P.A = (const bf16_t*)A; P.W = (const bf16_t*)W; P.bias = bias; P.res = (const bf16_t*)res; P.C = (bf16_t*)C;
    P.pos = pos; P.M = M; P.N = N; P.K = K;
    P.Cpre = ext ? (bf16_t*)ext->c_pre : nullptr;
    P.out_f32 = ext ? ext->out_f32 : 0;
    VIT_REQUIRE(!P.out_f32 || (epi == VIT_EPI_BIAS && !patch), "gemm: fp32 output only with the plain epilogue");
    if (ext && ext->ln_stats) {
        VIT_REQUIRE(ext->ln_colsum != nullptr && !patch && !P.out_f32 && !P.Cpre, "gemm: folded LayerNorm takes the plain bf16 epilogues only");
        P.ln_stats = ext->ln_stats; P.ln_c = ext->ln_colsum;
    }
    const int bn = pick_bn(M, N);
    P.NT = ceil_div(N, bn);
    if (patch) {
        P.pg = *patch;
        VIT_REQUIRE(pos != nullptr, "patch gemm needs the position table");
        const bool aligned = (patch->patch % 8 == 0) && (patch->image % 8 == 0);
        if (bn == 256)
            return aligned ? dispatch_epi<256, A_PATCH_ALIGNED>(P, epi, stream)
                           : dispatch_epi<256, A_PATCH_GATHER>(P, epi, stream);
        return aligned ? dispatch_epi<128, A_PATCH_ALIGNED>(P, epi, stream)
                       : dispatch_epi<128, A_PATCH_GATHER>(P, epi, stream);
    }
    if (bn == 256) return dispatch_epi<256, A_DENSE>(P, epi, stream);
    return dispatch_epi<128, A_DENSE>(P, epi, stream);
}
