// fp32 GEMM (parity mode): C[M,N] = epi(A[M,K] * W[N,K]^T + bias (+ residual)), exact fp32
// products and accumulation on v_mfma_f32_32x32x2_f32 (bit-identical to a k-ordered fmaf
// chain, so the 1e-4 relative bar of BASELINE.json's fp32 mode is met with margin).
// Same math and epilogues as gemm_bf16.hip (modeling_vit.py:216-218, 236, 250-252, 62-69);
// not a throughput path: 128x128x16 tiles, register staging, single LDS buffer.
#include "vit_common.h"

namespace {

constexpr int TM = 128, TN = 128, TK = 16, LDT = TM + 4;  // LDS tiles are k-major: [TK][LDT]
constexpr int EPI_PATCH = 100;

struct GemmParamsF32 {
    const float* A;
    const float* W;
    const float* bias;
    const float* res;
    float* C;
    const float* pos;
    int M, N, K, NT;
    int is_patch;
    PatchGeom pg;
};

__device__ __forceinline__ size_t patch_off_f32(const PatchGeom& g, int kk) {
    const int pp = g.patch * g.patch;
    const int ch = kk / pp, r2 = kk - ch * pp;
    const int i = r2 / g.patch, j = r2 - i * g.patch;
    return ((size_t)ch * g.image + i) * g.image + j;
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmParamsF32 P) {
    __shared__ float As[TK * LDT];
    __shared__ float Ws[TK * LDT];
    const int mt = blockIdx.x / P.NT, nt = blockIdx.x - mt * P.NT;
    const int m0 = mt * TM, n0 = nt * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hi = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // staging map: 2 passes x (64 rows x 4 k-quads)
    const int srow = tid >> 2, kq = (tid & 3) * 4;
    const float* a_row[2];
    const float* w_row[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int gm = m0 + t * 64 + srow;
        if (gm > P.M - 1) gm = P.M - 1;
        if (P.is_patch) {
            const int np = P.pg.grid * P.pg.grid;
            const int b = gm / np, rem = gm - b * np, py = rem / P.pg.grid, px = rem - py * P.pg.grid;
            a_row[t] = P.A + ((size_t)b * P.pg.channels * P.pg.image + (size_t)py * P.pg.patch) * P.pg.image +
                       (size_t)px * P.pg.patch;
        } else {
            a_row[t] = P.A + (size_t)gm * P.K;
        }
        int gn = n0 + t * 64 + srow;
        if (gn > P.N - 1) gn = P.N - 1;
        w_row[t] = P.W + (size_t)gn * P.K;
    }
    const int kreal = P.is_patch ? P.pg.kreal : P.K;

    for (int k0 = 0; k0 < P.K; k0 += TK) {
        float av[2][4], wv[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = k0 + kq + e;
                if (P.is_patch)
                    av[t][e] = (kk < kreal) ? a_row[t][patch_off_f32(P.pg, kk)] : 0.f;
                else
                    av[t][e] = (kk < P.K) ? a_row[t][kk] : 0.f;
                wv[t][e] = (kk < P.K) ? w_row[t][kk] : 0.f;
            }
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                As[(kq + e) * LDT + t * 64 + srow] = av[t][e];
                Ws[(kq + e) * LDT + t * 64 + srow] = wv[t][e];
            }
        __syncthreads();
#pragma unroll
        for (int kp = 0; kp < TK / 2; ++kp) {
            float xf[2], wf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) xf[a] = As[(2 * kp + hi) * LDT + wm * 64 + a * 32 + r32];
#pragma unroll
            for (int b = 0; b < 2; ++b) wf[b] = Ws[(2 * kp + hi) * LDT + wn * 64 + b * 32 + r32];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[b], xf[a], acc[a][b], 0, 0, 0);
        }
    }

    // swapped operands: acc[a][b] holds C^T: column (lane&31) = row m, register r = column
    // n = (r&3) + 8*(r>>2) + 4*hi  -> 4 consecutive n per register quad.
    const int np = P.is_patch ? P.pg.grid * P.pg.grid : 1;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int m = m0 + wm * 64 + a * 32 + r32;
        if (m >= P.M) continue;
        size_t orow = (size_t)m;
        int pr = 0;
        if constexpr (EPI == EPI_PATCH) {
            const int bimg = m / np;
            pr = m - bimg * np + 1;
            orow = (size_t)m + bimg + 1;
        }
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + b * 32 + 8 * g + 4 * hi;
                if (n >= P.N) continue;  // N % 4 == 0
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = acc[a][b][g * 4 + j] + (P.bias ? P.bias[n + j] : 0.f);
                    v[j] = apply_act<EPI>(t);
                    if constexpr (EPI == EPI_PATCH) v[j] += P.pos[(size_t)pr * P.N + n + j];
                    if constexpr (EPI == VIT_EPI_RESIDUAL) v[j] += P.res[orow * P.N + n + j];
                }
                *(float4*)(P.C + orow * P.N + n) = make_float4(v[0], v[1], v[2], v[3]);
            }
    }
}

template <int EPI>
int launch_f32(const GemmParamsF32& P, hipStream_t stream) {
    dim3 grid(ceil_div(P.M, TM) * P.NT), block(256);
    hipLaunchKernelGGL(gemm_f32_kernel<EPI>, grid, block, 0, stream, P);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

}  // namespace

int launch_gemm_f32(const void* A, const void* W, const float* bias, const void* res, void* C, int M, int N, int K,
                    int epi, const PatchGeom* patch, const float* pos, hipStream_t stream) {
    VIT_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
    VIT_REQUIRE(N % 4 == 0, "gemm(fp32): N must be a multiple of 4 (N=%d)", N);
    VIT_REQUIRE(epi != VIT_EPI_RESIDUAL || res != nullptr, "gemm: residual epilogue without residual");
    GemmParamsF32 P{};
    P.A = (const float*)A; P.W = (const float*)W; P.bias = bias; P.res = (const float*)res; P.C = (float*)C;
    P.pos = pos; P.M = M; P.N = N; P.K = K; P.NT = ceil_div(N, TN);
    if (patch) {
        VIT_REQUIRE(pos != nullptr, "patch gemm needs the position table");
        P.is_patch = 1; P.pg = *patch;
        return launch_f32<EPI_PATCH>(P, stream);
    }
    switch (epi) {
        case VIT_EPI_BIAS: return launch_f32<VIT_EPI_BIAS>(P, stream);
        case VIT_EPI_GELU: return launch_f32<VIT_EPI_GELU>(P, stream);
        case VIT_EPI_QUICK_GELU: return launch_f32<VIT_EPI_QUICK_GELU>(P, stream);
        case VIT_EPI_RESIDUAL: return launch_f32<VIT_EPI_RESIDUAL>(P, stream);
    }
    return vit_set_error(VIT_E_INVALID, "gemm: unknown epilogue %d", epi);
}
