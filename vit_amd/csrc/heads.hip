// Heads applied to the encoder's pooled row:
//   ViTPooler.forward (modeling_vit.py:289-301):  pooled = tanh(dense(hidden[:, 0]))
//   CLIPVisionModelWithProjection (modeling_clip.py:907, :950): image_embeds = visual_projection(pooled),
//   a bias-free Linear.
// out[b, n] = act(sum_k x[b, k] W[n, k] + bias[n]) for a [batch, K] input whose rows may be strided
// (the CLS row of [batch, N, D] is read in place, no gather copy).
//
// This is a skinny problem (batch x D x D, 0.1 % of the encoder's FLOPs), bound by reading W once:
// a wave owns FB output features x RB input rows, lanes split K in 16-byte vectors, fp32 FMA,
// one wave-wide reduction per output. No LDS, no MFMA.
#include "vit_common.h"
#include <math.h>

namespace {

constexpr int FB = 4;  // output features per wave
constexpr int RB = 8;  // input rows per wave

template <typename T>
struct Vec;
template <>
struct Vec<float> {
    static constexpr int W = 4;
    typedef f32x4 type;
};
template <>
struct Vec<bf16_t> {
    static constexpr int W = 8;
    typedef bf16x8 type;
};

template <typename T, int ACT>
__global__ __launch_bounds__(256) void head_linear_kernel(const T* __restrict__ x, int64_t xs, const T* __restrict__ W,
                                                          const float* __restrict__ bias, T* __restrict__ out, int batch,
                                                          int K, int N) {
    constexpr int VW = Vec<T>::W;
    using V = typename Vec<T>::type;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * FB;
    const int b0 = blockIdx.y * RB;
    if (n0 >= N) return;

    const T* wrow[FB];
    const T* xrow[RB];
#pragma unroll
    for (int f = 0; f < FB; ++f) wrow[f] = W + (size_t)(n0 + f < N ? n0 + f : N - 1) * K;
#pragma unroll
    for (int r = 0; r < RB; ++r) xrow[r] = x + (size_t)(b0 + r < batch ? b0 + r : batch - 1) * xs;

    float acc[FB][RB];
#pragma unroll
    for (int f = 0; f < FB; ++f)
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[f][r] = 0.f;

    for (int k = lane * VW; k < K; k += 64 * VW) {
        float wv[FB][VW];
#pragma unroll
        for (int f = 0; f < FB; ++f) {
            const V v = *(const V*)(wrow[f] + k);
#pragma unroll
            for (int e = 0; e < VW; ++e) wv[f][e] = (float)v[e];
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const V v = *(const V*)(xrow[r] + k);
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                const float xe = (float)v[e];
#pragma unroll
                for (int f = 0; f < FB; ++f) acc[f][r] = fmaf(xe, wv[f][e], acc[f][r]);
            }
        }
    }
#pragma unroll
    for (int f = 0; f < FB; ++f)
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[f][r] = wave_sum(acc[f][r]);

    // lane f*RB + r writes output (b0 + r, n0 + f)
    if (lane < FB * RB) {
        const int f = lane / RB, r = lane - f * RB;
        float v = 0.f;
#pragma unroll
        for (int ff = 0; ff < FB; ++ff)
#pragma unroll
            for (int rr = 0; rr < RB; ++rr)
                if (ff == f && rr == r) v = acc[ff][rr];
        if (n0 + f < N && b0 + r < batch) {
            if (bias != nullptr) v += bias[n0 + f];
            if constexpr (ACT == VIT_HEAD_TANH) v = tanhf(v);
            out[(size_t)(b0 + r) * N + n0 + f] = (T)v;
        }
    }
}

template <typename T>
int launch_head_t(const void* x, int64_t xs, const void* W, const float* bias, void* out, int batch, int K, int N,
                  int act, hipStream_t stream) {
    dim3 grid(ceil_div(N, 4 * FB), ceil_div(batch, RB)), block(256);
    if (act == VIT_HEAD_TANH)
        hipLaunchKernelGGL((head_linear_kernel<T, VIT_HEAD_TANH>), grid, block, 0, stream, (const T*)x, xs, (const T*)W,
                           bias, (T*)out, batch, K, N);
    else
        hipLaunchKernelGGL((head_linear_kernel<T, VIT_HEAD_LINEAR>), grid, block, 0, stream, (const T*)x, xs,
                           (const T*)W, bias, (T*)out, batch, K, N);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

}  // namespace

int launch_head(const void* x, int64_t xs, const void* W, const float* bias, void* out, int batch, int K, int N, int act,
                int dtype, hipStream_t stream) {
    VIT_REQUIRE(batch > 0 && K > 0 && N > 0, "head: empty problem batch=%d in=%d out=%d", batch, K, N);
    VIT_REQUIRE(act == VIT_HEAD_LINEAR || act == VIT_HEAD_TANH, "head: bad activation %d", act);
    VIT_REQUIRE(dtype == VIT_BF16 || dtype == VIT_F32, "head: bad dtype %d", dtype);
    const int vw = dtype == VIT_BF16 ? 8 : 4;
    VIT_REQUIRE(K % vw == 0 && xs % vw == 0 && xs >= K, "head: in_dim and the row stride must be multiples of %d (in=%d stride=%lld)",
                vw, K, (long long)xs);
    VIT_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)W & 15) == 0, "head: x and weight must be 16-byte aligned");
    if (dtype == VIT_BF16) return launch_head_t<bf16_t>(x, xs, W, bias, out, batch, K, N, act, stream);
    return launch_head_t<float>(x, xs, W, bias, out, batch, K, N, act, stream);
}
