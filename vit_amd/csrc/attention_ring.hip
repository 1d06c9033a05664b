// Fused multi-head self-attention forward for head dims that are not 64 (ViT-H/14: head_dim 80), bf16 -- the
// persistent LDS-DMA ring of attention_dh64.hip carried over to rows that are not 128 bytes (same math:
// eager_attention_forward, modeling_vit.py:164-189; fp32 online softmax, no mask, no score matrix; same MFMA dataflow
// with the query on the lane as attention.hip / attention_dh64.hip).
//
// What changes with 160-byte rows:
//  * LDS rows are PADDED (K and Q: 176 B = 11 chunks of 16 B, V: 192 B = 12 chunks) -- the strides attention.hip uses,
//    conflict-free for ds_read_b128 (K, Q) and ds_read_b64_tr_b16 (V) without a swizzle. An LDS-DMA writes lane-linear,
//    so a padded image needs the instruction's active lanes to cover whole padded rows: a wave's piece is FOUR rows,
//    issued with 44 (K, Q) or 48 (V) lanes active; the lanes that land on pad chunks re-read the row's last chunk.
//    Eight waves x 4 rows = one 32-key tile; every wave issues one K and one V instruction per tile.
//  * Q K^T takes 5 k-steps, O^T = V^T P^T three 32-row blocks of head_dim (the third half empty), 11 MFMA per tile.
//  * counted waits as in attention_dh64.hip with two DMA instructions per tile and eight Q pieces per item.
#include "vit_common.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float kDeferLog2 = 6.0f;  // as attention_dh64.hip
constexpr int NDMA = 8;  // waves of a workgroup that issue the K / V DMA (4 rows each = one 32-key tile); a workgroup has 8..12 waves
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DH>
struct RingGeo {
    static constexpr int CH = DH / 8;              // 16-byte chunks of a row
    static constexpr int KCH = CH + 1;             // padded K / Q row, chunks (odd -> conflict-free ds_read_b128)
    static constexpr int KROW = KCH * 16;
    static constexpr int DB = (DH + 31) / 32;      // 32-row blocks of O^T
    static constexpr int VCH = ((DB | 1) * 64) / 16;  // padded V row, chunks (odd multiple of 64 bytes)
    static constexpr int VROW = VCH * 16;
    static constexpr int KS = DH / 16;             // k-steps of Q K^T
    static constexpr int KTILE = 32 * KROW, VTILE = 32 * VROW;
    static constexpr int SLOT = KTILE + VTILE;     // one ring slot: K rows then V rows of 32 keys
    static constexpr int QSLOT = 32 * KROW;        // one wave's 32 query rows
    static constexpr int NQ = 8;                   // Q DMA pieces per wave and item (4 rows each)
    static constexpr int ST = DH / 16;             // output store instructions of an active wave per item (16 bytes per lane each)
    static_assert(DH % 16 == 0 && 4 * KCH <= 64 && 4 * VCH <= 64, "a four-row piece must fit one wave instruction");
};

struct RingSrc {
    const unsigned char* q;
    const unsigned char* k;
    const unsigned char* v;
    long b_stride, h_stride;  // bytes
    int rs2;                  // bytes between consecutive tokens of one head
};

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ bf16x8 pack8r(const f32x16& p, int s) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)p[8 * s + e];
    return o;
}

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_wave_base, unsigned lane_off, int uni_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_wave_base), 16, (int)lane_off, uni_off, 0, 0);
}

// NWV waves per workgroup = 32-query tiles per item. The sequence's tiles are dealt out over ceil(tiles / NWV) items per
// (batch, head), and an item costs about the same whatever the number of its waves that have queries (every wave walks
// all key tiles in lock step), so NWV is chosen per sequence length to minimise the item count: 1025 tokens = 33 tiles
// are 3 items of 11 waves instead of 5 items of 8 (the fifth with a single query row).
template <int DH, int R, bool LSE, int NWV>
__global__ __launch_bounds__(NWV * 64) void attention_ring_kernel(const RingSrc src, bf16_t* __restrict__ out,
                                                                  float* __restrict__ lse, int N, int H, int nitems,
                                                                  int qblocks, float scale_log2e) {
    using G = RingGeo<DH>;
    constexpr int ST = G::ST + (LSE ? 1 : 0);
    constexpr int NQ = G::NQ;
    constexpr int C = 2;  // DMA instructions per wave and tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // R slots, then NWV Q slots

    const unsigned smem_lds = (unsigned)(uintptr_t)LDS_PTR(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hi = lane >> 5;
    const int D = H * DH;
    const int rs2 = src.rs2;
    const int nt = (N + 31) >> 5;
    const bool dma_wave = wave < NDMA;  // wave-uniform

    const int nmy = (nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (nmy <= 0) return;  // uniform

    auto item_bh = [&](int j, int* qb) {
        const int item = blockIdx.x + j * gridDim.x;
        const int bh = item / qblocks;
        *qb = item - bh * qblocks;
        return bh;
    };
    auto item_rsrc = [&](const unsigned char* opbase, int bh) {
        const int b = bh / H, h = bh - b * H;
        const unsigned char* base = opbase + (size_t)b * src.b_stride + (size_t)h * src.h_stride;
        const unsigned span = (unsigned)(N - 1) * (unsigned)rs2 + (unsigned)(DH * 2);
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, span, 0x00020000);
    };

    // ---- this wave's pieces of a tile: K rows 4w..4w+3 (lanes 0..4*KCH-1) and V rows 4w..4w+3 (lanes 0..4*VCH-1)
    const int k_prow = 4 * wave + lane / G::KCH, k_pc = lane % G::KCH;
    const int v_prow = 4 * wave + lane / G::VCH, v_pc = lane % G::VCH;
    const unsigned k_pconst = (unsigned)((k_pc < G::CH ? k_pc : G::CH - 1) * 16);
    const unsigned v_pconst = (unsigned)((v_pc < G::CH ? v_pc : G::CH - 1) * 16);
    const bool k_lane = lane < 4 * G::KCH, v_lane = lane < 4 * G::VCH;
    unsigned char* const k_pdst = smem + 4 * wave * G::KROW;
    unsigned char* const v_pdst = smem + G::KTILE + 4 * wave * G::VROW;
    auto dma_tile = [&](__amdgpu_buffer_rsrc_t krs, __amdgpu_buffer_rsrc_t vrs, int t, int slot) {
        const int lim = N - 1 - t * 32;  // rows past the sequence re-read its last row (their scores are masked)
        const int kr = k_prow < lim ? k_prow : lim, vr = v_prow < lim ? v_prow : lim;
        if (k_lane) dma16(krs, k_pdst + slot * G::SLOT, (unsigned)kr * (unsigned)rs2 + k_pconst, t * 32 * rs2);
        if (v_lane) dma16(vrs, v_pdst + slot * G::SLOT, (unsigned)vr * (unsigned)rs2 + v_pconst, t * 32 * rs2);
    };

    // ---- read-side lane geometry
    unsigned kro[G::KS];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) kro[ks] = (unsigned)(r * G::KROW + (ks * 2 + hi) * 16);
    const int g = lane >> 4, i16 = lane & 15;
    const int tr_key = 4 * (g >> 1) + (i16 >> 2);
    const int tr_col = ((g & 1) * 16 + 4 * (i16 & 3)) * 2;
    const unsigned vro = (unsigned)(G::KTILE + tr_key * G::VROW + tr_col);  // + 16*sk*VROW (+ 8*VROW) + d*64

    // ---- Q: the wave's 32 query rows go to its own LDS slot by DMA (8 pieces of 4 rows), one item ahead
    unsigned char* const qslot = smem + R * G::SLOT + wave * G::QSLOT;
    auto dma_q = [&](__amdgpu_buffer_rsrc_t rsrc, int q0) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            int qr = q0 + 4 * i + lane / G::KCH;
            if (qr > N - 1) qr = N - 1;
            if (k_lane) dma16(rsrc, qslot + i * 4 * G::KROW, (unsigned)qr * (unsigned)rs2 + k_pconst, 0);
        }
    };
    bf16x8 qf[G::KS];

    // ---- prologue
    int qb;
    int bh = item_bh(0, &qb);
    int q0 = (qb * NWV + wave) * 32;
    __amdgpu_buffer_rsrc_t pkr = item_rsrc(src.k, bh), pvr = item_rsrc(src.v, bh);  // item the DMA cursor is in
    dma_q(item_rsrc(src.q, bh), q0);
    int pj = 0, pt = 0;
    auto dma_next = [&](int slot) {
        dma_tile(pkr, pvr, pt, slot);
        if (pt + 1 < nt) {
            ++pt;
        } else if (pj + 1 < nmy) {
            ++pj;
            pt = 0;
            int dummy;
            const int nb = item_bh(pj, &dummy);
            pkr = item_rsrc(src.k, nb);
            pvr = item_rsrc(src.v, nb);
        }
    };
    if (dma_wave) {
#pragma unroll
        for (int s = 0; s < R - 1; ++s) dma_next(s);
    }

    f32x16 o[G::DB];  // written by the first tile of every item
    float m_run = 0.f, l_run = 0.f;

    int slot = 0, pslot = R - 1;
    bool prev_active = false;
    // One tile step. FIRST (p = 0 of an item): the running max is the tile's own, O and l start from this tile (the PV
    // products take a zero accumulator, nothing is zeroed or rescaled) -- as in attention_dh64.hip.
    auto tile_step = [&](auto FIRSTC, int j, int p, bool active) {
        constexpr bool FIRST = decltype(FIRSTC)::value;
        {
            // counted wait: this wave's pieces of the current tile have landed. Younger operations that may stay in
            // flight: (R-2)*C tile DMAs; while p <= R-2 also the previous item's ST stores (if the wave had queries in
            // it) and, from p = 1, the NQ loads of the Q prefetch.
            {
                const bool inwin = p <= R - 2;
                const bool stin = inwin && prev_active;
                const bool qin = inwin && p >= 1;
                if (dma_wave) {
                    if (stin) {
                        if (qin) wait_vm<(R - 2) * C + NQ + ST>(); else wait_vm<(R - 2) * C + ST>();
                    } else {
                        if (qin) wait_vm<(R - 2) * C + NQ>(); else wait_vm<(R - 2) * C>();
                    }
                } else if (p == 0) {
                    // a wave without tile DMAs only has its Q prefetch to wait for (issued at p = 0 of the previous item,
                    // older than that item's stores): before its Q fragments are read, everything but the stores is in
                    if (prev_active) wait_vm<ST>(); else wait_vm<0>();
                }
            }
            asm volatile("s_barrier" ::: "memory");
            if constexpr (FIRST) {
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) qf[ks] = *(const bf16x8*)(qslot + kro[ks]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) asm volatile("" : "+v"(qf[ks]));
                const int jn = j + 1 < nmy ? j + 1 : j;
                int nqb;
                const int nbh = item_bh(jn, &nqb);
                dma_q(item_rsrc(src.q, nbh), (nqb * NWV + wave) * 32);
            }
            if (dma_wave) dma_next(pslot);
            pslot = pslot + 1 == R ? 0 : pslot + 1;

            if (active) {
                const unsigned char* Kb = smem + slot * G::SLOT;
                const unsigned kb_lds = smem_lds + (unsigned)(slot * G::SLOT);
                f32x16 s;
#pragma unroll
                for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    const bf16x8 kf = *(const bf16x8*)(Kb + kro[ks]);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
                }
                // V^T fragments by ds_read_b64_tr_b16 (inline asm: through the builtin the compiler orders every such
                // read behind vmcnt(0)), issued now and waited for before the PV products
                short4v vl[G::DB][2][2];
#pragma unroll
                for (int sk = 0; sk < 2; ++sk)
#pragma unroll
                    for (int d = 0; d < G::DB; ++d) {
                        const unsigned va = kb_lds + vro + (unsigned)(16 * sk * G::VROW + d * 64);
                        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
                                     : "=&v"(vl[d][sk][0]), "=&v"(vl[d][sk][1])
                                     : "v"(va), "n"(8 * G::VROW)
                                     : "memory");
                    }
                const int key0 = p * 32;
                if (key0 + 32 > N) {  // wave-uniform: only the last tile of the sequence
                    // (the limit is made opaque here: hoisted out of the loops, the 16 lane masks live in 32 SGPRs and
                    // push the loop's own scalars into spill lanes)
                    int lim = N - key0 - 4 * hi;
                    asm volatile("" : "+v"(lim));
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if ((e & 3) + 8 * (e >> 2) >= lim) s[e] = -INFINITY;
                }
                float mx = s[0];
#pragma unroll
                for (int e = 1; e < 16; ++e) mx = fmaxf(mx, s[e]);
                mx = max_across_halves(mx);
                if constexpr (FIRST) {
                    m_run = mx * scale_log2e;
                } else {
                    // the running max is left alone until a score outgrows it by more than kDeferLog2 (base 2): P stays
                    // <= 2^kDeferLog2, and the O / l rescale, which random scores trigger in most tiles (one of a wave's 32
                    // queries meets a new maximum: 63 % of the tiles even at tile 32 of 1025 tokens), runs only when it must
                    const float m_new = fmaxf(m_run, mx * scale_log2e);
                    if (__any(m_new > m_run + kDeferLog2)) {
                        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                        m_run = m_new;
                        l_run *= alpha;
#pragma unroll
                        for (int d = 0; d < G::DB; ++d)
#pragma unroll
                            for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
                    }
                }
                float psum = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    s[e] = __builtin_amdgcn_exp2f(s[e] * scale_log2e - m_run);
                    psum += s[e];
                }
                if constexpr (FIRST) l_run = psum; else l_run += psum;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int sk = 0; sk < 2; ++sk)
#pragma unroll
                    for (int d = 0; d < G::DB; ++d) {
                        asm volatile("" : "+v"(vl[d][sk][0]), "+v"(vl[d][sk][1]));
                    }
#pragma unroll
                for (int sk = 0; sk < 2; ++sk) {
                    const bf16x8 pb = pack8r(s, sk);
#pragma unroll
                    for (int d = 0; d < G::DB; ++d) {
                        const short8v both = __builtin_shufflevector(vl[d][sk][0], vl[d][sk][1], 0, 1, 2, 3, 4, 5, 6, 7);
                        if (FIRST && sk == 0) {
                            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, both), pb, z, 0, 0, 0);
                        } else {
                            o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, both), pb, o[d], 0, 0, 0);
                        }
                    }
                }
            }
            slot = slot + 1 == R ? 0 : slot + 1;
        }
    };

    for (int j = 0; j < nmy; ++j) {
        const bool active = q0 < N;  // wave-uniform
        tile_step(std::true_type{}, j, 0, active);
        for (int p = 1; p < nt; ++p) tile_step(std::false_type{}, j, p, active);

        // ---- item finished: ST stores per active wave, never waited for explicitly
        if (active) {
            const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
            const float inv = __builtin_amdgcn_rcpf(l_tot);  // 1 ulp; the result is rounded to bf16
            const int b = bh / H, h = bh - b * H;
            if constexpr (LSE) {
                if (hi == 0 && q0 + r < N) lse[(size_t)bh * N + q0 + r] = m_run + log2f(l_tot);
            }
            {
                // The lane halves trade 8-byte pieces (v_permlane32_swap) so that every lane holds 8 consecutive head
                // dims: 16-byte stores (DH % 16 == 0: a pair of pieces is in or out as a whole)
                // through a buffer resource over the image's rows of this head: 32-bit lane offsets, rows >= N out of range
                const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(out + (size_t)b * N * D + (size_t)h * DH), 0, (unsigned)(N - 1) * (unsigned)(D * 2) + (unsigned)(DH * 2), 0x00020000);
                const unsigned ooff = (q0 + r < N) ? (unsigned)(q0 + r) * (unsigned)(D * 2) + (unsigned)(16 * hi) : 0x80000000u;
#pragma unroll
                for (int d = 0; d < G::DB; ++d) {
                    unsigned w[4][2];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        bf16x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(o[d][g4 * 4 + e] * inv);
                        const u32x2 u = __builtin_bit_cast(u32x2, v);
                        w[g4][0] = u[0];
                        w[g4][1] = u[1];
                    }
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        if (d * 32 + 16 * pr < DH) {
                            u32x4 pk;
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const auto sw = __builtin_amdgcn_permlane32_swap(w[2 * pr][k], w[2 * pr + 1][k], false, false);
                                pk[k] = (unsigned)sw[0];
                                pk[2 + k] = (unsigned)sw[1];
                            }
                            __builtin_amdgcn_raw_buffer_store_b128(pk, orsrc, (int)(ooff + (unsigned)(d * 64 + 32 * pr)), 0, 0);
                        }
                    }
                }
            }
        }
        prev_active = active;
        if (j + 1 < nmy) {
            bh = item_bh(j + 1, &qb);
            q0 = (qb * NWV + wave) * 32;
        }
    }
    wait_vm<0>();  // the dummy DMAs of the tail must land before the LDS is released
}

template <int DH, int R, int NWV>
int launch_ring(const void* qkv, void* out, float* lse, int batch, int N, int H, hipStream_t stream) {
    using G = RingGeo<DH>;
    constexpr int LDS = R * G::SLOT + NWV * G::QSLOT;
    const int qblocks = ceil_div(ceil_div(N, 32), NWV);
    const long nitems = (long)batch * H * qblocks;
    VIT_REQUIRE(nitems < (1L << 31), "attention: too many (batch, head, query block) items");
    VIT_REQUIRE((long)N * 3 * H * DH * 2 < (1L << 31), "attention: one image's qkv rows exceed 2 GiB");
    static PerDeviceInt resident_of;  // resident workgroups per device; 0 = the attributes are not set on it yet
    std::atomic<int>* rs = resident_of.slot();
    int resident = rs ? rs->load(std::memory_order_relaxed) : 0;
    if (resident == 0) {
        int cus, per_cu = 0;
        VIT_TRY_RC(vit_cu_count(&cus));
        VIT_CHECK_HIP(hipFuncSetAttribute((const void*)attention_ring_kernel<DH, R, false, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        VIT_CHECK_HIP(hipFuncSetAttribute((const void*)attention_ring_kernel<DH, R, true, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        VIT_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)attention_ring_kernel<DH, R, false, NWV>, NWV * 64, LDS));
        resident = cus * (per_cu > 0 ? per_cu : 1);
        if (rs) rs->store(resident, std::memory_order_relaxed);
    }
    const int grid = nitems < resident ? (int)nitems : resident;
    const float scale_log2e = LOG2E / sqrtf((float)DH);
    const int D = H * DH;
    const unsigned char* p = (const unsigned char*)qkv;
    RingSrc src{p, p + (size_t)D * 2, p + (size_t)2 * D * 2, (long)N * 3 * D * 2, (long)DH * 2, 3 * D * 2};
    if (lse)
        hipLaunchKernelGGL((attention_ring_kernel<DH, R, true, NWV>), dim3(grid), dim3(NWV * 64), LDS, stream, src, (bf16_t*)out, lse, N,
                           H, (int)nitems, qblocks, scale_log2e);
    else
        hipLaunchKernelGGL((attention_ring_kernel<DH, R, false, NWV>), dim3(grid), dim3(NWV * 64), LDS, stream, src, (bf16_t*)out, lse, N,
                           H, (int)nitems, qblocks, scale_log2e);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

// Ring depth for head_dim 80: 3 slots + 8 Q slots = 78.5 KiB, two workgroups per CU, the DMA two tiles ahead. Measured
// at batch 32 x 1025 tokens x 16 heads: depth 3 / 5 / 8 = 360 / 365 / 362 us (depth does not matter once the DMA is one
// tile ahead; two workgroups per CU shorten the tail at small batch: 111 vs 115 us at batch 8).
constexpr int kRing80 = 3;
constexpr int kRingMinTokens = 193;  // shorter sequences keep attention.hip's staged kernel (few tiles per item: the ring never fills)

}  // namespace

// head_dim 80, bf16, at least R-1 key tiles per item (the counted waits assume the stores and Q loads of one item
// boundary have left the window before the next boundary)
bool attention_ring_eligible(int N, int head_dim) {
    static_assert(kRingMinTokens > 32 * (kRing80 - 1), "an item must span at least R-1 key tiles");
    return head_dim == 80 && N >= kRingMinTokens;
}

int launch_attention_ring(const void* qkv, void* out, float* lse, int batch, int N, int H, int head_dim, hipStream_t stream) {
    VIT_REQUIRE(attention_ring_eligible(N, head_dim), "attention ring: head_dim %d / N %d not eligible", head_dim, N);
    // Waves per workgroup: items per (batch, head) x the relative cost of an item, measured at 1025 tokens, batch 32
    // (8 waves = two per SIMD: 69; 9..12 = three on at least one SIMD: 87..100). 33 tiles -> 12 waves (3 items, 299 us
    // against 345 us with 5 items of 8), 9 tiles -> 9, 16 tiles -> 8. VIT_AMD_RING_WAVES forces a count (lab).
    static const int forced = getenv("VIT_AMD_RING_WAVES") ? atoi(getenv("VIT_AMD_RING_WAVES")) : 0;
    static const int item_cost[13] = {0, 0, 0, 0, 0, 0, 0, 0, 69, 87, 92, 100, 100};
    const int tiles = ceil_div(N, 32);
    int nwv = 8;
    for (int w = 9; w <= 12; ++w)
        if (ceil_div(tiles, w) * item_cost[w] <= ceil_div(tiles, nwv) * item_cost[nwv]) nwv = w;
    if (forced >= 8 && forced <= 12) nwv = forced;
    switch (nwv) {
        case 9: return launch_ring<80, kRing80, 9>(qkv, out, lse, batch, N, H, stream);
        case 10: return launch_ring<80, kRing80, 10>(qkv, out, lse, batch, N, H, stream);
        case 11: return launch_ring<80, kRing80, 11>(qkv, out, lse, batch, N, H, stream);
        case 12: return launch_ring<80, kRing80, 12>(qkv, out, lse, batch, N, H, stream);
    }
    return launch_ring<80, kRing80, 8>(qkv, out, lse, batch, N, H, stream);
}
