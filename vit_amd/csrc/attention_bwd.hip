// Fused multi-head self-attention BACKWARD (bf16 I/O, fp32 math): gradients of
// out = softmax(Q K^T * dh^-0.5) V (eager_attention_forward, modeling_vit.py:164-189) with
// respect to the packed projection output qkv. No N x N matrix is ever written: P is recomputed
// from Q, K and the forward's per-row log-sum-exp (base 2, already scaled).
//
//   P  = exp2(S c - L2),  S = Q K^T,  c = dh^-0.5 log2(e)
//   dP = dO V^T,  delta = rowsum(dO * O),  dS = P * (dP - delta)
//   dQ = dS K * dh^-0.5,   dK = dS^T Q * dh^-0.5,   dV = P^T dO
//
// Two launches of one kernel template, each with the OUTER index on the MFMA lane so that every
// sum runs over accumulator rows (an accumulator tile is then directly the B operand of the next
// product, no transposes, no atomics, bitwise reproducible):
//   MODE_DQ : outer = 32 queries per wave (Q, dO fragments in registers), inner = keys streamed
//             through LDS:  S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T.
//   MODE_DKV: outer = 32 keys per wave (K, V fragments in registers), inner = queries streamed
//             through LDS:  S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS.
// Transposed operands (K^T, dO^T, Q^T) come from row-major LDS images by ds_read_b64_tr_b16.
// S and dP are computed twice (7 instead of 5 products) -- the price of having no cross-wave
// reduction; at N = 197 the backward is HBM/latency bound anyway (DESIGN.md "Backward").
//
// Round 2: the kernel is PERSISTENT (one workgroup per CU walks (batch, head, outer block) items) and the global loads
// of the next staging unit (the next chunk of inner rows, or the next item's first chunk and outer fragments) are
// issued into registers before the current chunk's products, so the HBM latency that the one-item-per-workgroup form
// exposed at every item (1 workgroup per CU: nothing else to run) hides behind them. Same operations in the same
// order per item: bit-identical results.
#include "vit_common.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr float LOG2E = 1.4426950408889634f;
enum { MODE_DQ = 0, MODE_DKV = 1 };

template <int DH>
struct BGeo {
    static constexpr int KSTEPS = DH / 16;
    static constexpr int DB = (DH + 31) / 32;
    static constexpr int RS = DH * 2 + 16;      // row-read image stride, bytes (conflict-free ds_read_b128)
    static constexpr int TS = (DB | 1) * 64;    // tr-read image stride, bytes (odd multiple of 64)
    static constexpr int CHUNKS = DH / 8;
    static constexpr int PIT = DH <= 64 ? 4 : 2;  // staging loads (16 bytes of each operand) a thread holds in flight per chunk
};

__device__ __forceinline__ bf16x8 pack8b(const f32x16& p, int s) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)p[8 * s + e];
    return o;
}

// A^T-operand fragment (32 rows of d-block `d`, k = 16 inner rows starting at row0 + 16*sk) read
// transposed from a row-major [inner][d] image with stride TS.
template <int TS>
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* img, int row0, int sk, int d, int tr_key, int tr_col) {
    const unsigned char* vp = img + (size_t)(row0 + 16 * sk + tr_key) * TS + tr_col + d * 64;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(vp));
    const short4v hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(vp + 8 * TS));
    const short8v both = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

template <int DH, int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu((MODE == MODE_DQ && DH <= 64) ? 4 : 2))) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                       const float* __restrict__ lse, const float* __restrict__ delta,
                                                       bf16_t* __restrict__ dqkv, int N, int H, int IC, int blocks, int nitems,
                                                       float scale_log2e, float scale) {
    using G = BGeo<DH>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // inner-operand images: I1 = K (DQ) / Q (DKV), I2 = V (DQ) / dO (DKV)
    unsigned char* R1 = smem;                          // [IC][RS] row image of I1
    unsigned char* R2 = R1 + (size_t)IC * G::RS;       // [IC][RS] row image of I2
    unsigned char* T1 = R2 + (size_t)IC * G::RS;       // [IC][TS] tr image of I1
    unsigned char* T2 = T1 + (size_t)IC * G::TS;       // [IC][TS] tr image of I2 (DKV only)
    float* rowL = (float*)(T2 + (MODE == MODE_DKV ? (size_t)IC * G::TS : 0));  // [IC] L2 of inner queries (DKV)
    float* rowD = rowL + IC;                                                   // [IC] delta of inner queries (DKV)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = blockDim.x, nwaves = blockDim.x >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int D = H * DH;
    const size_t rs = (size_t)3 * D;
    const int nchunks = (N + IC - 1) / IC;
    const int g = lane >> 4, i16 = lane & 15;
    const int tr_key = 4 * (g >> 1) + (i16 >> 2);
    const int tr_col = ((g & 1) * 16 + 4 * (i16 & 3)) * 2;

    // ---- staging registers of the NEXT unit. Only the dK/dV launch prefetches: its 157 VGPRs allow one workgroup per
    // CU anyway, so nothing else hides its loads. The dQ launch fits 124 VGPRs = two workgroups per CU, which overlap
    // each other's staging; the prefetch registers would cost it the second workgroup (122 -> 138 us, measured).
    constexpr bool PF = (MODE == MODE_DKV);
    constexpr int PIT = PF ? G::PIT : 1;
    bf16x8 pv1[PIT], pv2[PIT];
    float pL = 0.f, pD = 0.f;                    // DKV: L2 / delta of inner row tid of the chunk
    auto item_coords = [&](int item, int* b, int* h, int* o0) {
        const int bh = item / blocks, by = item - bh * blocks;
        *b = bh / H;
        *h = bh - *b * H;
        *o0 = (by * nwaves + wave) * 32;
    };
    auto chunk_rows = [&](int chunk) {
        int nin = N - chunk * IC;
        return (nin > IC) ? IC : ((nin + 31) & ~31);
    };
    auto prefetch = [&](int item, int chunk) {
        if constexpr (!PF) return;
        int b, h, o0;
        item_coords(item, &b, &h, &o0);
        const bf16_t* qbase = qkv + (size_t)b * N * rs + (size_t)h * DH;          // q | k (+D) | v (+2D)
        const bf16_t* dobase = dO + (size_t)b * N * D + (size_t)h * DH;
        const int ic0 = chunk * IC, total = chunk_rows(chunk) * G::CHUNKS;
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int idx = tid + it * nthreads;
            if (idx < total) {
                const int row = idx / G::CHUNKS, c = idx - row * G::CHUNKS;
                int gr = ic0 + row;
                if (gr > N - 1) gr = N - 1;  // pad rows re-read the last row; they are masked below
                if constexpr (MODE == MODE_DQ) {
                    const bf16_t* p = qbase + (size_t)gr * rs + c * 8;
                    pv1[it] = *(const bf16x8*)(p + D);       // K
                    pv2[it] = *(const bf16x8*)(p + 2 * D);   // V
                } else {
                    pv1[it] = *(const bf16x8*)(qbase + (size_t)gr * rs + c * 8);      // Q
                    pv2[it] = *(const bf16x8*)(dobase + (size_t)gr * D + c * 8);     // dO
                }
            }
        }
        if constexpr (MODE == MODE_DKV) {
            if (tid < chunk_rows(chunk)) {
                const int gr = ic0 + tid;
                const size_t base = ((size_t)b * H + h) * N;
                pL = gr < N ? lse[base + gr] : INFINITY;  // P = exp2(.. - inf) = 0 for pad queries
                pD = gr < N ? delta[base + gr] : 0.f;
            }
        }
    };

    bf16x8 f1[G::KSTEPS], f2[G::KSTEPS];
    float oL = 0.f, oD = 0.f;  // DQ: per-lane query constants
    // outer fragments (B operands) of an item: lane&31 = outer row, 8 consecutive d of half hi per k-step. Loaded for the
    // first item here, for every later one right behind the last products of its predecessor (under that item's epilogue)
    auto load_outer = [&](int item) {
        int b, h, o0;
        item_coords(item, &b, &h, &o0);
        const bf16_t* qbase = qkv + (size_t)b * N * rs + (size_t)h * DH;
        const bf16_t* dobase = dO + (size_t)b * N * D + (size_t)h * DH;
        const int orow = (o0 + r < N) ? o0 + r : N - 1;
#pragma unroll
        for (int ks = 0; ks < G::KSTEPS; ++ks) {
            if constexpr (MODE == MODE_DQ) {
                f1[ks] = *(const bf16x8*)(qbase + (size_t)orow * rs + ks * 16 + hi * 8);             // Q
                f2[ks] = *(const bf16x8*)(dobase + (size_t)orow * D + ks * 16 + hi * 8);             // dO
            } else {
                f1[ks] = *(const bf16x8*)(qbase + (size_t)orow * rs + D + ks * 16 + hi * 8);         // K
                f2[ks] = *(const bf16x8*)(qbase + (size_t)orow * rs + 2 * D + ks * 16 + hi * 8);     // V
            }
        }
        if constexpr (MODE == MODE_DQ) {
            const size_t base = ((size_t)b * H + h) * N;
            oL = lse[base + orow];
            oD = delta[base + orow];
        }
    };
    f32x16 acc1[G::DB], acc2[G::DB];  // DQ: acc1 = dQ^T.  DKV: acc1 = dK^T, acc2 = dV^T

    int item = blockIdx.x, chunk = 0;
    if (item < nitems) {
        prefetch(item, chunk);
        load_outer(item);
    }
    while (item < nitems) {
        const int ic0 = chunk * IC, nin = chunk_rows(chunk);
        __syncthreads();  // the previous unit's products (and an epilogue's reads of its LDS block) are done
        // ---- commit the staged unit: registers -> the LDS images (dQ launch: global -> LDS here)
        {
            const int total = nin * G::CHUNKS;
            if constexpr (PF) {
#pragma unroll
                for (int it = 0; it < PIT; ++it) {
                    const int idx = tid + it * nthreads;
                    if (idx < total) {
                        const int row = idx / G::CHUNKS, c = idx - row * G::CHUNKS;
                        *(bf16x8*)(R1 + (size_t)row * G::RS + c * 16) = pv1[it];
                        *(bf16x8*)(R2 + (size_t)row * G::RS + c * 16) = pv2[it];
                        *(bf16x8*)(T1 + (size_t)row * G::TS + c * 16) = pv1[it];
                        if constexpr (MODE == MODE_DKV) *(bf16x8*)(T2 + (size_t)row * G::TS + c * 16) = pv2[it];
                    }
                }
            } else {
                int b, h, o0;
                item_coords(item, &b, &h, &o0);
                const bf16_t* qbase = qkv + (size_t)b * N * rs + (size_t)h * DH;
                for (int idx = tid; idx < total; idx += nthreads) {
                    const int row = idx / G::CHUNKS, c = idx - row * G::CHUNKS;
                    int gr = ic0 + row;
                    if (gr > N - 1) gr = N - 1;  // pad rows re-read the last row; they are masked below
                    const bf16_t* p = qbase + (size_t)gr * rs + c * 8;
                    const bf16x8 v1 = *(const bf16x8*)(p + D);       // K
                    const bf16x8 v2 = *(const bf16x8*)(p + 2 * D);   // V
                    *(bf16x8*)(R1 + (size_t)row * G::RS + c * 16) = v1;
                    *(bf16x8*)(R2 + (size_t)row * G::RS + c * 16) = v2;
                    *(bf16x8*)(T1 + (size_t)row * G::TS + c * 16) = v1;
                }
            }
            if constexpr (MODE == MODE_DKV) {
                if (tid < nin) {
                    rowL[tid] = pL;
                    rowD[tid] = pD;
                }
            }
            if (chunk == 0) {
#pragma unroll
                for (int d = 0; d < G::DB; ++d)
#pragma unroll
                    for (int e = 0; e < 16; ++e) { acc1[d][e] = 0.f; acc2[d][e] = 0.f; }
            }
        }
        __syncthreads();
        // The outer fragments (global loads issued under the previous item's epilogue, or before the loop) are forced
        // complete HERE, before the prefetch goes out. Left to the compiler the wait sits at their first use inside the
        // tile loop, and because its bookkeeping must also cover the path on which no prefetch was issued (last unit,
        // lanes past the chunk) it is a vmcnt(7) .. vmcnt(0) ladder there: vmcnt counts in order, so that ladder also
        // waited for the YOUNGER prefetch loads during the first tile of every chunk.
        if constexpr (PF) {
#pragma unroll
            for (int ks = 0; ks < G::KSTEPS; ++ks) asm volatile("" : "+v"(f1[ks]), "+v"(f2[ks]));
        }
        // ---- the next unit's loads go out now and land during this unit's products
        int nitem = item, nchunk = chunk + 1;
        if (nchunk == nchunks) {
            nchunk = 0;
            nitem = item + gridDim.x;
        }
        if (nitem < nitems) prefetch(nitem, nchunk);

        const int ntiles = nin >> 5;
        for (int t = 0; t < ntiles; ++t) {
            // ---- S tile and dP tile: rows = 32 inner rows (registers), col = outer row (lane)
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
            const unsigned char* p1 = R1 + (size_t)(t * 32 + r) * G::RS + hi * 16;
            const unsigned char* p2 = R2 + (size_t)(t * 32 + r) * G::RS + hi * 16;
            // (Reading all row fragments in one batch and issuing the transposed fragments of the accumulation before the
            // P / dS arithmetic was measured for the dK/dV launch: within noise with one k-step of fragments ahead, slower
            // with both (10 spills at 256 VGPRs). With the products removed the launch still takes 58 % of its time: it is
            // bound by the staging round trips and the store epilogue, not by the LDS latency inside a tile.)
#pragma unroll
            for (int ks = 0; ks < G::KSTEPS; ++ks) {
                const bf16x8 a1 = *(const bf16x8*)(p1 + ks * 32);
                const bf16x8 a2 = *(const bf16x8*)(p2 + ks * 32);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, f1[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, f2[ks], dp, 0, 0, 0);
            }
            // ---- P and dS in place (register e is inner row (e&3) + 8(e>>2) + 4hi of the tile)
            if constexpr (MODE == MODE_DQ) {
                const int key0 = ic0 + t * 32;
                if (key0 + 32 > N) {  // wave-uniform: only the sequence's last tile has keys to mask (P = exp2(-inf) = 0)
                    int lim = N - key0 - 4 * hi;
                    asm volatile("" : "+v"(lim));  // (opaque: the 16 lane masks would be hoisted into 32 SGPRs)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if ((e & 3) + 8 * (e >> 2) >= lim) s[e] = -INFINITY;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float p = __builtin_amdgcn_exp2f(s[e] * scale_log2e - oL);
                    s[e] = p * (dp[e] - oD);  // dS^T
                }
            } else {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 L = *(const f32x4*)(rowL + t * 32 + 8 * g4 + 4 * hi);
                    const f32x4 Dl = *(const f32x4*)(rowD + t * 32 + 8 * g4 + 4 * hi);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = g4 * 4 + j;
                        const float p = __builtin_amdgcn_exp2f(s[e] * scale_log2e - L[j]);
                        dp[e] = p * (dp[e] - Dl[j]);  // dS
                        s[e] = p;                     // P
                    }
                }
            }
            // ---- accumulate: sums over the 32 inner rows (two k-steps of 16)
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                if constexpr (MODE == MODE_DQ) {
                    const bf16x8 dsb = pack8b(s, sk);
#pragma unroll
                    for (int d = 0; d < G::DB; ++d)
                        acc1[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<G::TS>(T1, t * 32, sk, d, tr_key, tr_col), dsb,
                                                                          acc1[d], 0, 0, 0);  // dQ^T += K^T dS^T
                } else {
                    const bf16x8 pb = pack8b(s, sk), dsb = pack8b(dp, sk);
#pragma unroll
                    for (int d = 0; d < G::DB; ++d) {
                        acc2[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<G::TS>(T2, t * 32, sk, d, tr_key, tr_col), pb,
                                                                          acc2[d], 0, 0, 0);  // dV^T += dO^T P
                        acc1[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<G::TS>(T1, t * 32, sk, d, tr_key, tr_col), dsb,
                                                                          acc1[d], 0, 0, 0);  // dK^T += Q^T dS
                    }
                }
            }
        }

        if (chunk == nchunks - 1) {
            // ---- store: the accumulators hold lane = outer row, registers = d (4 consecutive per group). Written from
            // there, one head's 128-byte row segment went out as 16 separate 8-byte pieces; the tile is turned through a
            // wave-private LDS block instead (the operand images are dead after the barrier) and leaves as 16-byte
            // chunks, 8 lanes per row.
            int b, h, o0;
            item_coords(item, &b, &h, &o0);
            if (nitem < nitems) load_outer(nitem);
            __syncthreads();
            unsigned char* ob = smem + (size_t)wave * 32 * G::RS;
#pragma unroll
            for (int which = 0; which < (MODE == MODE_DKV ? 2 : 1); ++which) {
#pragma unroll
                for (int d = 0; d < G::DB; ++d)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int d0 = d * 32 + 8 * g4 + 4 * hi;
                        if (d0 < DH) {
                            bf16x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                v[e] = which == 0 ? (bf16_t)(acc1[d][g4 * 4 + e] * scale) : (bf16_t)acc2[d][g4 * 4 + e];
                            *(bf16x4*)(ob + (size_t)r * G::RS + d0 * 2) = v;
                        }
                    }
                __syncthreads();  // (the block is wave-private; the barriers only pin the order of its writes and reads)
                bf16_t* obase = dqkv + (size_t)b * N * rs + (size_t)h * DH + (MODE == MODE_DKV ? D : 0) + (which ? D : 0);
                for (int idx = lane; idx < 32 * G::CHUNKS; idx += 64) {
                    const int row = idx / G::CHUNKS, c = idx - row * G::CHUNKS;
                    const bf16x8 v = *(const bf16x8*)(ob + (size_t)row * G::RS + c * 16);
                    if (o0 + row < N) *(bf16x8*)(obase + (size_t)(o0 + row) * rs + c * 8) = v;  // dQ / dK (which 0), dV (which 1)
                }
                if (MODE == MODE_DKV && which == 0) __syncthreads();
            }
        }
        item = nitem;
        chunk = nchunk;
    }
}


// ---------------------------------------------------------------------------------------------------------------
// One-pass form for sequences whose operands fit the LDS whole (head_dim 64, 97..224 tokens: ViT-B/16's 197): one
// persistent workgroup per CU takes a (batch, head) and runs BOTH halves on it back to back --
//   phase A (= MODE_DQ):  K, V staged, each wave's 32 queries outer:  dQ
//   phase B (= MODE_DKV): Q, dO staged, each wave's 32 keys outer:    dK, dV
// so Q, K, V, dO come from HBM once (phase B's staging hits the L2 that phase A just filled; its outer K / V fragments
// are read from phase A's LDS images before they are overwritten), delta = rowsum(dO * O) is computed by the waves of
// phase A from the O rows of their own queries (no delta launch, no delta round trip), and the three launches become
// one: 620 MB instead of 1.0 GB per call at ViT-B/16 batch 256. Same products in the same order as the two-launch
// form; the staging of the NEXT phase is in flight in registers during the current phase's products (as PF above).
template <int DH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void attn_bwd_fused_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ O, const bf16_t* __restrict__ dO, const float* __restrict__ lse,
    bf16_t* __restrict__ dqkv, int N, int H, int NP, int nitems, float scale_log2e, float scale) {
    using G = BGeo<DH>;
    constexpr int PIT = DH / 16;  // NP * CHUNKS 16-byte pieces per operand over 2 NP threads
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // phase A images: K rows, V rows, K transposed-read image; phase B: Q rows, dO rows, Q and dO transposed-read images,
    // then the lse / delta of every query. The two sets overlap (rowL / rowD lie beyond both).
    unsigned char* const RA1 = smem;
    unsigned char* const RA2 = RA1 + (size_t)NP * G::RS;
    unsigned char* const TA1 = RA2 + (size_t)NP * G::RS;
    unsigned char* const RB1 = smem;
    unsigned char* const RB2 = RB1 + (size_t)NP * G::RS;
    unsigned char* const TB1 = RB2 + (size_t)NP * G::RS;
    unsigned char* const TB2 = TB1 + (size_t)NP * G::TS;
    float* const rowL = (float*)(TB2 + (size_t)NP * G::TS);
    float* const rowD = rowL + NP;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = blockDim.x;  // = 2 NP: every wave has 32 outer rows
    const int r = lane & 31, hi = lane >> 5;
    const int D = H * DH;
    const size_t rs = (size_t)3 * D;
    const int g = lane >> 4, i16 = lane & 15;
    const int tr_key = 4 * (g >> 1) + (i16 >> 2);
    const int tr_col = ((g & 1) * 16 + 4 * (i16 & 3)) * 2;
    const int o0 = wave * 32;
    const int ntiles = NP >> 5;

    bf16x8 pv1[PIT], pv2[PIT];
    float pL = 0.f;
    // staging loads of a phase: A = (K, V), B = (Q, dO) + the queries' lse
    auto prefetch = [&](int item, bool phaseB) {
        const int b = item / H, h = item - b * H;
        const bf16_t* qbase = qkv + (size_t)b * N * rs + (size_t)h * DH;
        const bf16_t* dobase = dO + (size_t)b * N * D + (size_t)h * DH;
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int idx = tid + it * nthreads;
            const int row = idx / G::CHUNKS, c = idx - row * G::CHUNKS;
            const int gr = row < N ? row : N - 1;  // pad rows re-read the last row; they are masked below
            if (!phaseB) {
                const bf16_t* p = qbase + (size_t)gr * rs + c * 8;
                pv1[it] = *(const bf16x8*)(p + D);       // K
                pv2[it] = *(const bf16x8*)(p + 2 * D);   // V
            } else {
                pv1[it] = *(const bf16x8*)(qbase + (size_t)gr * rs + c * 8);   // Q
                pv2[it] = *(const bf16x8*)(dobase + (size_t)gr * D + c * 8);   // dO
            }
        }
        if (phaseB && tid < NP) pL = tid < N ? lse[(size_t)item * N + tid] : INFINITY;  // P = exp2(.. - inf) = 0 for pad queries
    };
    auto commit = [&](bool phaseB) {
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int idx = tid + it * nthreads;
            const int row = idx / G::CHUNKS, c = idx - row * G::CHUNKS;
            *(bf16x8*)(RA1 + (size_t)row * G::RS + c * 16) = pv1[it];
            *(bf16x8*)(RA2 + (size_t)row * G::RS + c * 16) = pv2[it];
            *(bf16x8*)(TA1 + (size_t)row * G::TS + c * 16) = pv1[it];
            if (phaseB) *(bf16x8*)(TB2 + (size_t)row * G::TS + c * 16) = pv2[it];
        }
        if (phaseB && tid < NP) rowL[tid] = pL;
    };

    bf16x8 f1[G::KSTEPS], f2[G::KSTEPS];
    float oL = 0.f, oD = 0.f;
    // phase A's outer fragments: the wave's 32 queries (Q, dO), their lse, and delta from their O rows
    auto load_outer_a = [&](int item) {
        const int b = item / H, h = item - b * H;
        const int orow = (o0 + r < N) ? o0 + r : N - 1;
        const bf16_t* qp = qkv + (size_t)b * N * rs + (size_t)h * DH + (size_t)orow * rs + hi * 8;
        const bf16_t* dp = dO + (size_t)b * N * D + (size_t)h * DH + (size_t)orow * D + hi * 8;
        const bf16_t* op = O + (size_t)b * N * D + (size_t)h * DH + (size_t)orow * D + hi * 8;
        bf16x8 fo[G::KSTEPS];
#pragma unroll
        for (int ks = 0; ks < G::KSTEPS; ++ks) {
            f1[ks] = *(const bf16x8*)(qp + ks * 16);
            f2[ks] = *(const bf16x8*)(dp + ks * 16);
            fo[ks] = *(const bf16x8*)(op + ks * 16);
        }
        oL = lse[(size_t)item * N + orow];
        float d = 0.f;
#pragma unroll
        for (int ks = 0; ks < G::KSTEPS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) d = fmaf((float)f2[ks][e], (float)fo[ks][e], d);
        oD = d + __shfl_xor(d, 32, 64);  // the other half of the row's head dims
    };
    f32x16 acc1[G::DB], acc2[G::DB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int d = 0; d < G::DB; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc1[d][e] = 0.f; acc2[d][e] = 0.f; }
    };
    // one accumulator set -> bf16 rows of dqkv at column offset `col` (through the wave's LDS block: 16-byte stores)
    auto store_acc = [&](const f32x16 (&acc)[G::DB], float mul, int item, int col) {
        const int b = item / H, h = item - b * H;
        unsigned char* ob = smem + (size_t)wave * 32 * G::RS;
#pragma unroll
        for (int d = 0; d < G::DB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = d * 32 + 8 * g4 + 4 * hi;
                if (d0 < DH) {
                    bf16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(acc[d][g4 * 4 + e] * mul);
                    *(bf16x4*)(ob + (size_t)r * G::RS + d0 * 2) = v;
                }
            }
        // the block is wave-private and a wave's LDS operations execute in order: no barrier, only the compiler is held
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bf16_t* obase = dqkv + (size_t)b * N * rs + (size_t)h * DH + col;
        for (int idx = lane; idx < 32 * G::CHUNKS; idx += 64) {
            const int row = idx / G::CHUNKS, c = idx - row * G::CHUNKS;
            const bf16x8 v = *(const bf16x8*)(ob + (size_t)row * G::RS + c * 16);
            if (o0 + row < N) *(bf16x8*)(obase + (size_t)(o0 + row) * rs + c * 8) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads are in before the wave writes its block again
    };

    int item = blockIdx.x;
    if (item < nitems) {
        prefetch(item, false);
        load_outer_a(item);
    }
    while (item < nitems) {
        const int nitem = item + gridDim.x;
        // ================= phase A: dQ
        __syncthreads();  // the previous item's last LDS reads are done
        commit(false);
        zero_acc();
        if (hi == 0) rowD[o0 + r] = (o0 + r < N) ? oD : 0.f;  // for phase B (outside both image sets)
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < G::KSTEPS; ++ks) asm volatile("" : "+v"(f1[ks]), "+v"(f2[ks]));  // outer loads complete before the prefetch goes out
        prefetch(item, true);
        for (int t = 0; t < ntiles; ++t) {
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
            const unsigned char* p1 = RA1 + (size_t)(t * 32 + r) * G::RS + hi * 16;
            const unsigned char* p2 = RA2 + (size_t)(t * 32 + r) * G::RS + hi * 16;
#pragma unroll
            for (int ks = 0; ks < G::KSTEPS; ++ks) {
                const bf16x8 a1 = *(const bf16x8*)(p1 + ks * 32);
                const bf16x8 a2 = *(const bf16x8*)(p2 + ks * 32);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, f1[ks], s, 0, 0, 0);    // S^T = K Q^T
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, f2[ks], dp, 0, 0, 0);  // dP^T = V dO^T
            }
            const int key0 = t * 32;
            if (key0 + 32 > N) {  // wave-uniform: only the sequence's last tile has keys to mask
                int lim = N - key0 - 4 * hi;
                asm volatile("" : "+v"(lim));
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if ((e & 3) + 8 * (e >> 2) >= lim) s[e] = -INFINITY;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __builtin_amdgcn_exp2f(s[e] * scale_log2e - oL);
                s[e] = p * (dp[e] - oD);  // dS^T
            }
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                const bf16x8 dsb = pack8b(s, sk);
#pragma unroll
                for (int d = 0; d < G::DB; ++d)
                    acc1[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<G::TS>(TA1, t * 32, sk, d, tr_key, tr_col), dsb, acc1[d], 0, 0, 0);  // dQ^T += K^T dS^T
            }
        }
        // phase B's outer fragments (the wave's 32 keys: K, V rows) from phase A's images, before they are overwritten
        {
            const unsigned char* p1 = RA1 + (size_t)(o0 + r) * G::RS + hi * 16;
            const unsigned char* p2 = RA2 + (size_t)(o0 + r) * G::RS + hi * 16;
#pragma unroll
            for (int ks = 0; ks < G::KSTEPS; ++ks) {
                f1[ks] = *(const bf16x8*)(p1 + ks * 32);
                f2[ks] = *(const bf16x8*)(p2 + ks * 32);
            }
        }
        __syncthreads();  // every wave is done with the phase A images
        store_acc(acc1, scale, item, 0);  // dQ
        // ================= phase B: dK, dV
        __syncthreads();
        commit(true);
        zero_acc();
        __syncthreads();
        if (nitem < nitems) prefetch(nitem, false);
        for (int t = 0; t < ntiles; ++t) {
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
            const unsigned char* p1 = RB1 + (size_t)(t * 32 + r) * G::RS + hi * 16;
            const unsigned char* p2 = RB2 + (size_t)(t * 32 + r) * G::RS + hi * 16;
#pragma unroll
            for (int ks = 0; ks < G::KSTEPS; ++ks) {
                const bf16x8 a1 = *(const bf16x8*)(p1 + ks * 32);
                const bf16x8 a2 = *(const bf16x8*)(p2 + ks * 32);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, f1[ks], s, 0, 0, 0);    // S = Q K^T
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, f2[ks], dp, 0, 0, 0);  // dP = dO V^T
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 L = *(const f32x4*)(rowL + t * 32 + 8 * g4 + 4 * hi);
                const f32x4 Dl = *(const f32x4*)(rowD + t * 32 + 8 * g4 + 4 * hi);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = g4 * 4 + j;
                    const float p = __builtin_amdgcn_exp2f(s[e] * scale_log2e - L[j]);
                    dp[e] = p * (dp[e] - Dl[j]);  // dS
                    s[e] = p;                     // P
                }
            }
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                const bf16x8 pb = pack8b(s, sk), dsb = pack8b(dp, sk);
#pragma unroll
                for (int d = 0; d < G::DB; ++d) {
                    acc2[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<G::TS>(TB2, t * 32, sk, d, tr_key, tr_col), pb, acc2[d], 0, 0, 0);   // dV^T += dO^T P
                    acc1[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<G::TS>(TB1, t * 32, sk, d, tr_key, tr_col), dsb, acc1[d], 0, 0, 0);  // dK^T += Q^T dS
                }
            }
        }
        if (nitem < nitems) load_outer_a(nitem);
        __syncthreads();
        store_acc(acc1, scale, item, D);      // dK
        store_acc(acc2, 1.0f, item, 2 * D);   // dV
        item = nitem;
    }
}

// the one-pass kernel takes this problem: head_dim 64, every wave of 4..7 has 32 outer rows, everything fits the LDS
template <int DH>
constexpr int fused_row_bytes() { return 2 * BGeo<DH>::RS + 2 * BGeo<DH>::TS + 8; }
inline bool bwd_fused_eligible(int N, int head_dim) {
    const int np = (N + 31) / 32 * 32;
    return head_dim == 64 && np >= 128 && np <= 256 && (size_t)np * fused_row_bytes<64>() <= 156 * 1024;
}

int launch_bwd_fused64(const void* qkv, const void* O, const void* dO, const float* lse, void* dqkv, int batch, int N, int H,
                       hipStream_t stream) {
    constexpr int DH = 64;
    const int NP = (N + 31) / 32 * 32;
    const size_t lds = (size_t)NP * fused_row_bytes<DH>();
    auto kern = attn_bwd_fused_kernel<DH>;
    static PerDeviceInt attr_set;
    std::atomic<int>* as = attr_set.slot();
    if (!as || !as->load(std::memory_order_relaxed)) {
        VIT_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        if (as) as->store(1, std::memory_order_relaxed);
    }
    int ncu = 0;
    VIT_TRY_RC(vit_cu_count(&ncu));
    const long nitems = (long)batch * H;
    VIT_REQUIRE(nitems < (1L << 31), "attention_bwd: too many (batch, head) items");
    const int grid = nitems < ncu ? (int)nitems : ncu;
    const float scale = 1.0f / sqrtf((float)DH);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NP * 2), lds, stream, (const bf16_t*)qkv, (const bf16_t*)O, (const bf16_t*)dO, lse,
                       (bf16_t*)dqkv, N, H, NP, (int)nitems, scale * LOG2E, scale);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

// delta[b, h, q] = sum_d dO[b, q, h, d] * O[b, q, h, d]
template <int DH>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ O, const bf16_t* __restrict__ dO,
                                                         float* __restrict__ delta, int B, int N, int H) {
    constexpr int CPH = DH / 8;  // 16-byte chunks per head
    if constexpr ((CPH & (CPH - 1)) == 0 && CPH <= 16) {
        // one lane per 16-byte chunk: a wave instruction reads 1 KiB of consecutive memory; the CPH lanes of a
        // (row, head) sum their products by a butterfly (CPH is a power of two and divides 64)
        const long f = (long)blockIdx.x * 256 + threadIdx.x;  // chunk index over B * N * H * CPH
        const long total = (long)B * N * H * CPH;
        float s = 0.f;
        if (f < total) {
            const bf16x8 a = *(const bf16x8*)(O + f * 8), bb = *(const bf16x8*)(dO + f * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)bb[e];
        }
#pragma unroll
        for (int o = CPH >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (f < total && (threadIdx.x & (CPH - 1)) == 0) {
            const long idx = f / CPH;                      // (row, head), head fastest
            const int h = (int)(idx % H);
            const long row = idx / H;
            const int b = (int)(row / N), q = (int)(row - (long)b * N);
            delta[((size_t)b * H + h) * N + q] = s;
        }
        return;
    }
    const int idx = blockIdx.x * 256 + threadIdx.x;  // over B * N * H, head fastest
    if (idx >= B * N * H) return;
    const int h = idx % H, row = idx / H;            // row = b * N + q
    const bf16_t* o = O + (size_t)row * H * DH + (size_t)h * DH;
    const bf16_t* g = dO + (size_t)row * H * DH + (size_t)h * DH;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DH / 8; ++c) {
        const bf16x8 a = *(const bf16x8*)(o + c * 8), bb = *(const bf16x8*)(g + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)bb[e];
    }
    const int b = row / N, q = row - b * N;
    delta[((size_t)b * H + h) * N + q] = s;
}

void pick_waves_b(int N, int* waves, int* blocks) {
    const int n32 = ceil_div(N, 32);
    int best_w = 1, best_cost = 1 << 30;
    for (int w = 8; w >= 1; --w) {
        if (w < 4 && n32 >= 4) break;
        const int cost = ceil_div(n32, w) * w;
        if (cost < best_cost) { best_cost = cost; best_w = w; }
    }
    *waves = best_w;
    *blocks = ceil_div(n32, best_w);
}

template <int DH, int MODE>
int launch_bwd_part(const void* qkv, const void* dO, const float* lse, const float* delta, void* dqkv, int batch, int N,
                    int H, hipStream_t stream) {
    using G = BGeo<DH>;
    constexpr int per_row = 2 * G::RS + (MODE == MODE_DKV ? 2 : 1) * G::TS + (MODE == MODE_DKV ? 8 : 0);
    // dK/dV: one workgroup per CU (it prefetches): room for long chunks. dQ: two per CU.
    constexpr int LDS_MAX = (MODE == MODE_DKV ? 96 : 64) * 1024;
    int waves, blocks;
    pick_waves_b(N, &waves, &blocks);
    if (waves < 4) waves = 4;  // staging wants threads (waves without outer rows only load)
    const int threads = waves * 64;
    int icmax = (LDS_MAX / per_row) / 32 * 32;
    const int ic_regs = (G::PIT * threads / G::CHUNKS) / 32 * 32;  // what PIT staging loads per thread cover
    if (MODE == MODE_DKV && icmax > ic_regs) icmax = ic_regs;
    if (MODE == MODE_DKV && icmax > threads / 32 * 32) icmax = threads / 32 * 32;  // one (L2, delta) pair per thread
    VIT_REQUIRE(icmax >= 32, "attention_bwd: no room for a 32-row chunk");
    // chunks of equal length (197 rows: 112 + 112 rather than 192 + 32: every chunk costs two barriers and a staging pass)
    const int npad = ((N + 31) / 32) * 32;
    const int nchunks = ceil_div(npad, icmax);
    const int IC = ((ceil_div(npad, nchunks) + 31) / 32) * 32;
    size_t lds = (size_t)IC * per_row;
    if (lds < (size_t)waves * 32 * G::RS) lds = (size_t)waves * 32 * G::RS;  // the store epilogue's per-wave blocks
    auto kern = attn_bwd_kernel<DH, MODE>;
    static PerDeviceInt attr_set;  // per instantiation and device
    std::atomic<int>* as = attr_set.slot();
    if (!as || !as->load(std::memory_order_relaxed)) {
        VIT_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX + 8 * 1024));
        if (as) as->store(1, std::memory_order_relaxed);
    }
    int ncu = 0;
    VIT_TRY_RC(vit_cu_count(&ncu));
    int per_cu = 1;
    VIT_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, threads, lds));
    if (per_cu < 1) per_cu = 1;
    const long nitems = (long)batch * H * blocks;
    VIT_REQUIRE(nitems < (1L << 31), "attention_bwd: too many (batch, head, block) items");
    const long resident = (long)ncu * per_cu;
    const int grid = nitems < resident ? (int)nitems : (int)resident;
    const float scale = 1.0f / sqrtf((float)DH);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, (const bf16_t*)qkv, (const bf16_t*)dO, lse, delta,
                       (bf16_t*)dqkv, N, H, IC, blocks, (int)nitems, scale * LOG2E, scale);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

template <int DH>
int launch_bwd(const void* qkv, const void* O, const void* dO, const float* lse, float* delta, void* dqkv, int batch,
               int N, int H, hipStream_t stream) {
    // VIT_AMD_ATTN_BWD_FUSED=0: the three-launch form everywhere (A/B switch; tests/test_gpu_switches.py)
    static const bool fused_on = !(getenv("VIT_AMD_ATTN_BWD_FUSED") && atoi(getenv("VIT_AMD_ATTN_BWD_FUSED")) == 0);
    if (fused_on && bwd_fused_eligible(N, DH)) return launch_bwd_fused64(qkv, O, dO, lse, dqkv, batch, N, H, stream);
    constexpr int CPH = DH / 8;
    const long total = (long)batch * N * H * (((CPH & (CPH - 1)) == 0 && CPH <= 16) ? CPH : 1);
    hipLaunchKernelGGL(attn_delta_kernel<DH>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)O,
                       (const bf16_t*)dO, delta, batch, N, H);
    VIT_CHECK_HIP(hipGetLastError());
    int rc = launch_bwd_part<DH, MODE_DQ>(qkv, dO, lse, delta, dqkv, batch, N, H, stream);
    if (rc != VIT_OK) return rc;
    return launch_bwd_part<DH, MODE_DKV>(qkv, dO, lse, delta, dqkv, batch, N, H, stream);
}

}  // namespace

// qkv [B*N, 3D] packed projection output, O [B*N, D] forward attention output, dO its gradient,
// lse [B, H, N] fp32 (base-2 log-sum-exp of the scaled scores, from the training forward),
// delta [B, H, N] fp32 scratch, dqkv [B*N, 3D] output.
int launch_attention_bwd(const void* qkv, const void* O, const void* dO, const float* lse, float* delta, void* dqkv,
                         int batch, int seq, int heads, int head_dim, hipStream_t stream) {
    VIT_REQUIRE(batch > 0 && seq > 0 && heads > 0, "attention_bwd: empty problem");
    VIT_REQUIRE(head_dim % 16 == 0 && head_dim >= 16 && head_dim <= 128,
                "attention_bwd: head_dim must be a multiple of 16 in [16,128], got %d", head_dim);
#define ATTNB_CASE(DHV) \
    case DHV: return launch_bwd<DHV>(qkv, O, dO, lse, delta, dqkv, batch, seq, heads, stream);
    switch (head_dim) {
        ATTNB_CASE(16) ATTNB_CASE(32) ATTNB_CASE(48) ATTNB_CASE(64) ATTNB_CASE(80) ATTNB_CASE(96) ATTNB_CASE(112) ATTNB_CASE(128)
    }
#undef ATTNB_CASE
    return vit_set_error(VIT_E_INVALID, "attention_bwd: unsupported head_dim %d", head_dim);
}
