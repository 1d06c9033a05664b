// Fused multi-head self-attention forward: out = softmax(Q K^T * dh^-0.5) V
// (eager_attention_forward, modeling_vit.py:164-189: fp32 softmax, no mask, not causal;
// called from ViTAttention.forward :207-238 / CLIPAttention). No score matrix is ever
// written: K/V chunks are staged in LDS and an online (running max / running sum) softmax
// is kept in registers.
//
// Layout. Input is the packed QKV GEMM output [B*N, 3*H*dh] (row = token: q | k | v), so
// the [B,N,H,dh] -> [B,H,N,dh] transposes of modeling_vit.py:216-218 are pure address
// arithmetic; output is [B*N, H*dh] (= transpose(1,2).reshape of :187,:235).
//
// gfx950 design (see DESIGN.md "Attention"):
//  * one workgroup = up to 8 waves, each wave owns 32 query rows of one (batch, head);
//    N=197 -> 7 waves cover the whole sequence, K/V of the head are staged ONCE per block.
//  * every MFMA product keeps the QUERY on the lane: S^T = K Q^T (v_mfma_f32_32x32x16_bf16,
//    A = K rows from LDS, B = Q rows from registers) leaves each lane with 16 keys of one
//    query, so the row max / row sum are in-register reductions plus one lane^32 exchange;
//    the exponentiated accumulator IS the B operand of O^T = V^T P^T (no LDS round trip,
//    no lane movement), and the softmax rescale / final 1/l are lane-local scalars.
//  * V^T fragments come from the row-major V tile by ds_read_b64_tr_b16 (hardware transpose).
//  * LDS rows are padded (K: dh*2+16 B, V: odd multiple of 64 B) -> conflict-free reads.
#include "vit_common.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr float LOG2E = 1.4426950408889634f;

template <int DH>
struct AttnGeo {
    static constexpr int KSTEPS = DH / 16;        // QK^T k-steps
    static constexpr int DB = (DH + 31) / 32;     // 32-row blocks of O^T
    static constexpr int KS = DH * 2 + 16;        // K row stride, bytes
    static constexpr int VS = (DB | 1) * 64;      // V row stride, bytes (odd multiple of 64)
    static constexpr int CHUNKS = DH / 8;         // 16-B chunks per row
};

__device__ __forceinline__ bf16x8 pack8(const f32x16& p, int s) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)p[8 * s + e];
    return o;
}

template <int DH>
__global__ __launch_bounds__(512) void attention_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                            float* __restrict__ lse, int N, int H, int KC,
                                                            float scale_log2e) {
    using G = AttnGeo<DH>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Kl = smem;                       // [KC][KS]
    unsigned char* Vl = smem + (size_t)KC * G::KS;  // [KC][VS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int D = H * DH;
    const size_t rs = (size_t)3 * D;  // qkv row stride, elements
    const bf16_t* base = qkv + (size_t)b * N * rs + (size_t)h * DH;

    const int q0 = (blockIdx.y * nwaves + wave) * 32;
    const int qrow = (q0 + r < N) ? q0 + r : N - 1;
    bf16x8 qf[G::KSTEPS];
#pragma unroll
    for (int ks = 0; ks < G::KSTEPS; ++ks) qf[ks] = *(const bf16x8*)(base + (size_t)qrow * rs + ks * 16 + hi * 8);

    f32x16 o[G::DB];
#pragma unroll
    for (int d = 0; d < G::DB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // transposed-read lane geometry (ds_read_b64_tr_b16): 16-lane group g, lane i in group
    const int g = lane >> 4, i16 = lane & 15;
    const int tr_key = 4 * (g >> 1) + (i16 >> 2);              // + t*32 + 16*s + 8*half
    const int tr_col = ((g & 1) * 16 + 4 * (i16 & 3)) * 2;     // bytes, + db*64

    for (int kc0 = 0; kc0 < N; kc0 += KC) {
        int nkeys = N - kc0;
        nkeys = (nkeys > KC) ? KC : ((nkeys + 31) & ~31);
        __syncthreads();  // previous chunk fully consumed
        // stage K/V: every thread issues a batch of 2 x SB 16-byte loads before the first LDS
        // write, so a whole chunk's loads are in flight together (one HBM latency per chunk).
        constexpr int SB = 4;
        const int total = nkeys * G::CHUNKS;
        for (int idx0 = tid; idx0 < total; idx0 += SB * blockDim.x) {
            bf16x8 kv[SB], vv[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = idx0 + u * blockDim.x;
                const int key = idx / G::CHUNKS, c = idx - key * G::CHUNKS;
                int gk = kc0 + key;
                if (gk > N - 1) gk = N - 1;  // pad keys re-read the last row (their scores are masked)
                const bf16_t* p = base + (size_t)gk * rs + c * 8;
                if (idx < total) {
                    kv[u] = *(const bf16x8*)(p + D);
                    vv[u] = *(const bf16x8*)(p + 2 * D);
                }
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = idx0 + u * blockDim.x;
                const int key = idx / G::CHUNKS, c = idx - key * G::CHUNKS;
                if (idx < total) {
                    *(bf16x8*)(Kl + (size_t)key * G::KS + c * 16) = kv[u];
                    *(bf16x8*)(Vl + (size_t)key * G::VS + c * 16) = vv[u];
                }
            }
        }
        __syncthreads();

        const int ntiles = nkeys >> 5;
        for (int t = 0; t < ntiles; ++t) {
            // ---- S^T tile: 32 keys x 32 queries ----
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
            const unsigned char* kp = Kl + (size_t)(t * 32 + r) * G::KS + hi * 16;
#pragma unroll
            for (int ks = 0; ks < G::KSTEPS; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kp + ks * 32);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
            }
            const int key0 = kc0 + t * 32;
            if (key0 + 32 > N) {  // wave-uniform: only the last tile of the sequence
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                    if (key >= N) s[e] = -INFINITY;
                }
            }
            float mx = s[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = fmaxf(mx, s[e]);
            mx = max_across_halves(mx);
            const float m_new = fmaxf(m_run, mx * scale_log2e);
            // the running max rarely moves after the first tiles: rescale O / l only when some
            // query of the wave saw a larger score (wave-uniform branch; exact, no threshold)
            if (__any(m_new > m_run)) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < G::DB; ++d)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
            }
            float psum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[e] = __builtin_amdgcn_exp2f(s[e] * scale_log2e - m_run);
                psum += s[e];
            }
            l_run += psum;

            // ---- O^T += V^T P^T ----
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                const bf16x8 pb = pack8(s, sk);
                const unsigned char* vp = Vl + (size_t)(t * 32 + 16 * sk + tr_key) * G::VS + tr_col;
#pragma unroll
                for (int d = 0; d < G::DB; ++d) {
                    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) short4v*)(vp + d * 64));
                    const short4v hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) short4v*)(vp + 8 * G::VS + d * 64));
                    const short8v both = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, both), pb, o[d], 0, 0, 0);
                }
            }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    // training forward: base-2 log-sum-exp of the scaled scores, P = exp2(S c - lse) in the backward
    if (lse != nullptr && hi == 0 && q0 + r < N) lse[((size_t)b * H + h) * N + q0 + r] = m_run + log2f(l_tot);
    if (q0 + r < N) {
        bf16_t* op = out + ((size_t)b * N + q0 + r) * D + (size_t)h * DH;
#pragma unroll
        for (int d = 0; d < G::DB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = d * 32 + 8 * g4 + 4 * hi;
                if (d0 < DH) {
                    bf16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(o[d][g4 * 4 + e] * inv);
                    *(bf16x4*)(op + d0) = v;
                }
            }
    }
}

// ---- fp32 parity-mode kernel: same structure on v_mfma_f32_32x32x2_f32 (exact fp32) ----
template <int DH>
__global__ __launch_bounds__(512) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                           int N, int H, int KC, float scale_log2e) {
    constexpr int DB = (DH + 31) / 32;
    constexpr int KSF = DH + 1;      // K row stride, floats (odd -> conflict-free column reads)
    constexpr int VSF = DB * 32;     // V row stride, floats
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Kl = (float*)smem;              // [KC][KSF]
    float* Vl = Kl + (size_t)KC * KSF;     // [KC][VSF]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int D = H * DH;
    const size_t rs = (size_t)3 * D;
    const float* base = qkv + (size_t)b * N * rs + (size_t)h * DH;

    const int q0 = (blockIdx.y * nwaves + wave) * 32;
    const int qrow = (q0 + r < N) ? q0 + r : N - 1;
    float qf[DH / 2];
#pragma unroll
    for (int t = 0; t < DH / 2; ++t) qf[t] = base[(size_t)qrow * rs + 2 * t + hi];

    f32x16 o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    for (int kc0 = 0; kc0 < N; kc0 += KC) {
        int nkeys = N - kc0;
        nkeys = (nkeys > KC) ? KC : ((nkeys + 31) & ~31);
        __syncthreads();
        for (int idx = tid; idx < nkeys * VSF; idx += blockDim.x) {
            const int key = idx / VSF, c = idx - key * VSF;
            const int gk = kc0 + key;
            float kv = 0.f, vv = 0.f;
            if (gk < N && c < DH) {
                const float* p = base + (size_t)gk * rs + c;
                kv = p[D];
                vv = p[2 * D];
            }
            if (c < DH) Kl[(size_t)key * KSF + c] = kv;
            Vl[(size_t)key * VSF + c] = vv;
        }
        __syncthreads();

        const int ntiles = nkeys >> 5;
        for (int t = 0; t < ntiles; ++t) {
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
            const float* kp = Kl + (size_t)(t * 32 + r) * KSF + hi;
#pragma unroll
            for (int kk = 0; kk < DH / 2; ++kk)
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * kk], qf[kk], s, 0, 0, 0);
            const int key0 = kc0 + t * 32;
            if (key0 + 32 > N) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                    if (key >= N) s[e] = -INFINITY;
                }
            }
            float mx = s[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = fmaxf(mx, s[e]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx * scale_log2e);
            const float alpha = exp2f(m_run - m_new);
            m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[e] = exp2f(s[e] * scale_log2e - m_new);
                psum += s[e];
            }
            l_run = l_run * alpha + psum;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
            // O^T += V^T P^T: k-step e pairs key (e&3)+8(e>>2) [lanes 0-31] with the same +4
            // [lanes 32-63] -- exactly what accumulator register e of each half holds.
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float* vp = Vl + (size_t)(t * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi) * VSF + r;
#pragma unroll
                for (int d = 0; d < DB; ++d)
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[d * 32], s[e], o[d], 0, 0, 0);
            }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q0 + r < N) {
        float* op = out + ((size_t)b * N + q0 + r) * D + (size_t)h * DH;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = d * 32 + 8 * g4 + 4 * hi;
                if (d0 < DH)
                    *(float4*)(op + d0) = make_float4(o[d][g4 * 4 + 0] * inv, o[d][g4 * 4 + 1] * inv,
                                                      o[d][g4 * 4 + 2] * inv, o[d][g4 * 4 + 3] * inv);
            }
    }
}

// waves per block: cover ceil(N/32) query tiles with the least padding, prefer more waves
void pick_waves(int N, int* waves, int* qblocks) {
    const int n32 = ceil_div(N, 32);
    int best_w = 1, best_cost = 1 << 30;
    for (int w = 8; w >= 1; --w) {
        if (w < 4 && n32 >= 4) break;
        const int cost = ceil_div(n32, w) * w;
        if (cost < best_cost) { best_cost = cost; best_w = w; }
    }
    *waves = best_w;
    *qblocks = ceil_div(n32, best_w);
}

template <int DH>
int launch_attn_bf16(const void* qkv, void* out, float* lse, int batch, int N, int H, hipStream_t stream) {
    using G = AttnGeo<DH>;
    int waves, qblocks;
    pick_waves(N, &waves, &qblocks);
    int kcmax = (80 * 1024 / (G::KS + G::VS)) / 32 * 32;
    int KC = ((N + 31) / 32) * 32;
    if (KC > kcmax) KC = kcmax;
    const size_t lds = (size_t)KC * (G::KS + G::VS);
    auto kern = attention_bf16_kernel<DH>;
    static PerDeviceInt attr_set;  // per instantiation and device
    std::atomic<int>* as = attr_set.slot();
    if (!as || !as->load(std::memory_order_relaxed)) {
        VIT_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        if (as) as->store(1, std::memory_order_relaxed);
    }
    const float scale_log2e = LOG2E / sqrtf((float)DH);
    hipLaunchKernelGGL(kern, dim3(batch * H, qblocks), dim3(waves * 64), lds, stream, (const bf16_t*)qkv, (bf16_t*)out,
                       lse, N, H, KC, scale_log2e);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

template <int DH>
int launch_attn_f32(const void* qkv, void* out, int batch, int N, int H, hipStream_t stream) {
    constexpr int DB = (DH + 31) / 32;
    constexpr int per_key = (DH + 1 + DB * 32) * 4;
    int waves, qblocks;
    pick_waves(N, &waves, &qblocks);
    int kcmax = (80 * 1024 / per_key) / 32 * 32;
    int KC = ((N + 31) / 32) * 32;
    if (KC > kcmax) KC = kcmax;
    const size_t lds = (size_t)KC * per_key;
    auto kern = attention_f32_kernel<DH>;
    static PerDeviceInt attr_set;  // per instantiation and device
    std::atomic<int>* as = attr_set.slot();
    if (!as || !as->load(std::memory_order_relaxed)) {
        VIT_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        if (as) as->store(1, std::memory_order_relaxed);
    }
    const float scale_log2e = LOG2E / sqrtf((float)DH);
    hipLaunchKernelGGL(kern, dim3(batch * H, qblocks), dim3(waves * 64), lds, stream, (const float*)qkv, (float*)out, N,
                       H, KC, scale_log2e);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

}  // namespace

// persistent LDS-DMA kernel for head_dim 64 (attention_dh64.hip)
bool attention_dh64_eligible(int N);
int launch_attention_dh64(const void* qkv, void* out, float* lse, int batch, int N, int H, hipStream_t stream);

// persistent LDS-DMA ring for head dims whose rows are not 128 bytes (attention_ring.hip): head_dim 80 (ViT-H/14)
bool attention_ring_eligible(int N, int head_dim);
int launch_attention_ring(const void* qkv, void* out, float* lse, int batch, int N, int H, int head_dim, hipStream_t stream);

int launch_attention(const void* qkv, void* out, int batch, int seq, int heads, int head_dim, int dtype,
                     hipStream_t stream, float* lse) {
    VIT_REQUIRE(batch > 0 && seq > 0 && heads > 0, "attention: empty problem");
    VIT_REQUIRE(head_dim % 16 == 0 && head_dim >= 16 && head_dim <= 128,
                "attention: head_dim must be a multiple of 16 in [16,128], got %d", head_dim);
    VIT_REQUIRE(dtype == VIT_BF16 || dtype == VIT_F32, "attention: bad dtype %d", dtype);
    VIT_REQUIRE(lse == nullptr || dtype == VIT_BF16, "attention: the log-sum-exp output is bf16-mode only");
    if (dtype == VIT_BF16 && head_dim == 64 && attention_dh64_eligible(seq))
        return launch_attention_dh64(qkv, out, lse, batch, seq, heads, stream);
    static const bool ring_on = !(getenv("VIT_AMD_ATTN_RING") && atoi(getenv("VIT_AMD_ATTN_RING")) == 0);
    if (dtype == VIT_BF16 && ring_on && attention_ring_eligible(seq, head_dim))
        return launch_attention_ring(qkv, out, lse, batch, seq, heads, head_dim, stream);
#define ATTN_CASE(DHV)                                                                          \
    case DHV:                                                                                   \
        return dtype == VIT_BF16 ? launch_attn_bf16<DHV>(qkv, out, lse, batch, seq, heads, stream) \
                                 : launch_attn_f32<DHV>(qkv, out, batch, seq, heads, stream);
    switch (head_dim) {
        ATTN_CASE(16) ATTN_CASE(32) ATTN_CASE(48) ATTN_CASE(64) ATTN_CASE(80) ATTN_CASE(96) ATTN_CASE(112) ATTN_CASE(128)
    }
#undef ATTN_CASE
    return vit_set_error(VIT_E_INVALID, "attention: unsupported head_dim %d", head_dim);
}
