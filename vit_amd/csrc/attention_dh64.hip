// Fused multi-head self-attention forward, head_dim 64, bf16 -- the persistent LDS-DMA form of
// attention.hip (same math: eager_attention_forward, modeling_vit.py:164-189; fp32 online softmax,
// no mask, no score matrix; same MFMA dataflow with the query on the lane).
//
// Why a second kernel. Per-workgroup stamps of the one-item-per-workgroup kernel (ViT-B/16, batch
// 256: 12.2 us per item) showed 3.5 us of exposed load latency (Q + first K/V chunk), 7.2 us of
// products and 1.5 us of output-store drain, with the two resident workgroups of a CU in lock
// step. A register prefetch one or two tiles ahead is too short (HBM latency under load is
// 2-3 us, a tile is 0.6-1 us) and more registers cost the second workgroup. So:
//  * PERSISTENT: grid = resident workgroups (2 per CU), each walks items
//    (batch, head, query block) blockIdx.x, +gridDim.x, ...
//  * K/V go HBM -> LDS by LDS-DMA through a buffer resource, no registers in flight, into a ring
//    of R = 5 tiles of 32 keys; the DMA runs R-1 tiles (3-4 us) ahead and crosses item
//    boundaries, so after the prologue no load latency is exposed. 8 waves: wave w issues piece
//    w of every tile (K rows 8w.. / V rows 8(w-4)..), one 1-KiB instruction per wave and tile.
//  * LDS tiles are unpadded 128-byte rows (the DMA writes lane-linear); bank conflicts are
//    removed by swizzling on the SOURCE side: K chunk c of row r sits at chunk c ^ ((r>>1)&7)
//    (conflict-free for the ds_read_b128 lane groups of a 32-row A operand), V's two 64-byte
//    halves swap on rows with (r>>1)&1 (conflict-free for ds_read_b64_tr_b16).
//  * the next item's Q rows arrive by DMA in a per-wave LDS slot a whole item ahead; an item's
//    output stores are never waited for: vmcnt is counted (in-order): before tile k's barrier a wave
//    allows exactly the younger operations to stay outstanding -- R-2 DMAs, plus the previous
//    item's stores and the Q prefetch while they are younger than DMA(k).
//  * one raw s_barrier per tile; waves without queries (N = 197: the 8th) only load.
#include "vit_common.h"
#include <math.h>
#include <type_traits>

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int DH = 64;
constexpr int R = 5;           // ring depth, tiles
constexpr int SLOT = 8192;     // one tile: K 32 x 128 B, then V 32 x 128 B
constexpr int NW = 8;          // waves per workgroup
constexpr int NQ = 4;          // Q DMA pieces per wave and item
constexpr int QSLOT = 4096;    // one wave's 32 query rows x 128 B
constexpr float kDeferLog2 = 6.0f;  // the running max is left alone until a score outgrows it by more than this (base 2)

// Where Q, K, V live: operand bases and byte strides (batch, head, token). The packed QKV GEMM
// output [B*N, 3*H*64] is what the library uses; a head-major [3][B][H][N][64] source was timed with
// the same kernel and is no faster (93.8 vs 94.5 us at ViT-B/16 batch 256), so the 128-byte row
// segments of the packed layout are not what limits this kernel.
struct AttnSrc {
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

__device__ __forceinline__ bf16x8 pack8(const f32x16& p, int s) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)p[8 * s + e];
    return o;
}

typedef int int4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool LSE>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void attention_dh64_kernel(
    const AttnSrc src, bf16_t* __restrict__ out, float* __restrict__ lse, int N, int H, int nitems, int qblocks,
    float scale_log2e) {
    constexpr int ST = 4 + (LSE ? 1 : 0);  // vector-memory stores an active wave issues per item
    __shared__ __attribute__((aligned(16))) unsigned char smem[R * SLOT + NW * QSLOT];

    const unsigned smem_lds = (unsigned)(uintptr_t)LDS_PTR(smem);  // LDS byte address of the ring
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hi = lane >> 5;
    const int D = H * DH;
    const int rs2 = src.rs2;  // token stride, bytes
    const int nt = (N + 31) >> 5;  // key tiles per item

    const int nmy = (nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;  // items of this workgroup
    if (nmy <= 0) return;  // uniform

    // ---- item geometry ----
    auto item_bh = [&](int j, int* qb) {
        const int item = blockIdx.x + j * gridDim.x;
        const int bh = item / qblocks;
        *qb = item - bh * qblocks;
        return bh;
    };
    // buffer resource over one operand (Q, K or V) of one (batch, head): N rows of 128 bytes, rs2 apart
    auto item_rsrc = [&](const unsigned char* opbase, int bh) {
        const int b = bh / H, h = bh - b * H;
        const unsigned char* base = opbase + (size_t)b * src.b_stride + (size_t)h * src.h_stride;
        const unsigned span = (unsigned)(N - 1) * (unsigned)rs2 + (unsigned)(DH * 2);
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, span, 0x00020000);
    };
    const unsigned char* const kvbase = wave < 4 ? src.k : src.v;  // the operand this wave's DMA piece belongs to

    // ---- DMA piece of this wave: rows 8*(wave&3).. of K (wave < 4) or V (wave >= 4) ----
    const int prow = 8 * (wave & 3) + (lane >> 3);
    const int ppos = lane & 7;
    const int pchunk = (wave < 4) ? (ppos ^ ((prow >> 1) & 7)) : (ppos ^ (((prow >> 1) & 1) << 2));
    const unsigned pconst = (unsigned)(pchunk * 16);
    unsigned char* const pdst = smem + (wave < 4 ? 0 : 4096) + (wave & 3) * 1024;
    auto dma_tile = [&](__amdgpu_buffer_rsrc_t rsrc, int t, int slot) {
        int lim = N - 1 - t * 32;  // rows past the sequence re-read its last row (their scores are masked)
        const int row = prow < lim ? prow : lim;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(pdst + slot * SLOT), 16, (int)((unsigned)row * (unsigned)rs2 + pconst),
                                                 t * 32 * rs2, 0, 0);
    };

    // ---- read-side lane geometry ----
    const int ksw = (r >> 1) & 7;
    unsigned kro[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kro[ks] = (unsigned)(r * 128 + (((ks * 2 + hi) ^ ksw) << 4));
    const int g = lane >> 4, i16 = lane & 15;
    const int tr_key = 4 * (g >> 1) + (i16 >> 2);
    const int tr_col = ((g & 1) * 16 + 4 * (i16 & 3)) * 2;
    const int vsw = (tr_key >> 1) & 1;
    unsigned vro[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) vro[d] = (unsigned)(4096 + tr_key * 128 + ((d ^ vsw) << 6) + tr_col);

    // ---- Q: the wave's 32 query rows go to its own LDS slot by DMA (4 pieces of 8 rows), one item
    // ahead; same row-major swizzled image as a K tile, so the fragment reads use kro[] too.
    // (Plain register loads were tried: the compiler's own vmcnt bookkeeping does not count the
    // DMA instructions, so its wait for the loaded registers drains the whole DMA pipeline.)
    unsigned char* const qslot = smem + R * SLOT + wave * QSLOT;
    auto dma_q = [&](__amdgpu_buffer_rsrc_t rsrc, int q0) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int row = 8 * i + (lane >> 3);
            const int c = ppos ^ ((row >> 1) & 7);
            int qr = q0 + row;
            if (qr > N - 1) qr = N - 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(qslot + i * 1024), 16, qr * rs2 + c * 16, 0, 0, 0);
        }
    };
    bf16x8 qf[4];

    // ---- prologue: Q of item 0, DMA of the first R-1 tiles of this workgroup's tile stream ----
    int qb;
    int bh = item_bh(0, &qb);
    int q0 = (qb * NW + wave) * 32;
    __amdgpu_buffer_rsrc_t prs = item_rsrc(kvbase, bh);  // resource of the item the DMA cursor is in
    dma_q(item_rsrc(src.q, bh), q0);
    int pj = 0, pt = 0;  // DMA cursor: item index in this workgroup's list, tile
    auto dma_next = [&](int slot) {
        dma_tile(prs, pt, slot);
        // advance; past the end the cursor stays on the last tile (dummy re-fetch keeps the counts uniform)
        if (pt + 1 < nt) {
            ++pt;
        } else if (pj + 1 < nmy) {
            ++pj;
            pt = 0;
            int dummy;
            prs = item_rsrc(kvbase, item_bh(pj, &dummy));
        }
    };
#pragma unroll
    for (int s = 0; s < R - 1; ++s) dma_next(s);

    f32x16 o[2];                      // written by the first tile of every item
    float m_run = 0.f, l_run = 0.f;

    int slot = 0;          // ring slot of the current tile
    int pslot = R - 1;     // ring slot the next DMA goes to
    bool prev_active = false;  // did this wave store outputs for the previous item
    // One tile step. FIRST (p = 0 of an item): the running max is the tile's own, O and l start from this tile (the PV
    // products take a zero accumulator, nothing is zeroed or rescaled).
    auto tile_step = [&](auto FIRSTC, int j, int p, bool active) {
        constexpr bool FIRST = decltype(FIRSTC)::value;
        // ---- counted wait: everything up to this wave's piece of the current tile has landed.
        // Younger operations that may stay in flight: R-2 DMAs; while p <= R-2 also the previous
        // item's ST stores (if this wave had queries in it) and, from p = 1, the NQ loads of the Q prefetch.
        {
            const bool inwin = p <= R - 2;
            const bool stin = inwin && prev_active;
            const bool qin = inwin && p >= 1;
            if (stin) {
                if (qin) wait_vm<R - 2 + NQ + ST>(); else wait_vm<R - 2 + ST>();
            } else {
                if (qin) wait_vm<R - 2 + NQ>(); else wait_vm<R - 2>();
            }
        }
        asm volatile("s_barrier" ::: "memory");  // every wave's piece landed; the previous tile's slot is free
        if constexpr (FIRST) {
            // this item's Q fragments (DMA issued one item ago, older than the tile just waited
            // for) -> registers; then the slot is free for the next item's Q (clamped to the last
            // item so every item issues the same number of DMAs)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qslot + kro[ks]);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3])::"memory");
            const int jn = j + 1 < nmy ? j + 1 : j;
            int nqb;
            const int nbh = item_bh(jn, &nqb);
            dma_q(item_rsrc(src.q, nbh), (nqb * NW + wave) * 32);
        }
        dma_next(pslot);
        pslot = pslot + 1 == R ? 0 : pslot + 1;

        if (active) {
            const unsigned char* Kb = smem + slot * SLOT;
            const unsigned kb_lds = smem_lds + (unsigned)(slot * SLOT);
            // ---- S^T tile: 32 keys x 32 queries ----
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(Kb + kro[ks]);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
            }
            // V^T fragments of both key halves by ds_read_b64_tr_b16, issued now (the tile landed before
            // the barrier) and waited for just before the PV products, so their latency hides behind
            // the softmax. Inline asm: through the builtin the compiler orders every such read behind
            // vmcnt(0) (it may alias a DMA target), which would drain the ring it reads from.
            short4v vl[8];
            {
                const unsigned va0 = kb_lds + vro[0], va1 = kb_lds + vro[1];
                asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:1024\n\t"
                             "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:1024\n\t"
                             "ds_read_b64_tr_b16 %4, %8 offset:2048\n\tds_read_b64_tr_b16 %5, %8 offset:3072\n\t"
                             "ds_read_b64_tr_b16 %6, %9 offset:2048\n\tds_read_b64_tr_b16 %7, %9 offset:3072"
                             : "=&v"(vl[0]), "=&v"(vl[1]), "=&v"(vl[2]), "=&v"(vl[3]), "=&v"(vl[4]), "=&v"(vl[5]),
                               "=&v"(vl[6]), "=&v"(vl[7])
                             : "v"(va0), "v"(va1)
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
                // The running max follows a query's scores only when one of them outgrows it by more than kDeferLog2
                // (base-2 units): until then P = exp2(S c - m_run) <= 2^kDeferLog2, which bf16 P and the fp32 sums hold
                // just as well, and the O / l rescale -- 18 instructions that random scores trigger in nearly every
                // tile (one of a wave's 32 queries meets a new maximum) -- runs only when it has to. Exact either way:
                // when the branch fires everything at the old max is scaled once (cdna_hip_programming.md T13).
                const float m_new = fmaxf(m_run, mx * scale_log2e);
                if (__any(m_new > m_run + kDeferLog2)) {
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                    m_run = m_new;
                    l_run *= alpha;
#pragma unroll
                    for (int d = 0; d < 2; ++d)
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
            // ---- O^T += V^T P^T ----
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(vl[0]), "+v"(vl[1]), "+v"(vl[2]), "+v"(vl[3]), "+v"(vl[4]), "+v"(vl[5]), "+v"(vl[6]), "+v"(vl[7])::"memory");
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                const bf16x8 pb = pack8(s, sk);
                const short8v b0 = __builtin_shufflevector(vl[4 * sk + 0], vl[4 * sk + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                const short8v b1 = __builtin_shufflevector(vl[4 * sk + 2], vl[4 * sk + 3], 0, 1, 2, 3, 4, 5, 6, 7);
                if (FIRST && sk == 0) {
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b0), pb, z, 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b1), pb, z, 0, 0, 0);
                } else {
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b0), pb, o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b1), pb, o[1], 0, 0, 0);
                }
            }
        }
        slot = slot + 1 == R ? 0 : slot + 1;
    };

    for (int j = 0; j < nmy; ++j) {
        const bool active = q0 < N;  // wave-uniform
        tile_step(std::true_type{}, j, 0, active);
        for (int p = 1; p < nt; ++p) tile_step(std::false_type{}, j, p, active);

        // ---- item finished: ST stores per active wave, never waited for explicitly ----
        if (active) {
            const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
            const float inv = __builtin_amdgcn_rcpf(l_tot);  // 1 ulp; the result is rounded to bf16
            const int b = bh / H, h = bh - b * H;
            if constexpr (LSE) {
                if (hi == 0 && q0 + r < N) lse[(size_t)bh * N + q0 + r] = m_run + log2f(l_tot);
            }
            {
                // The lane halves trade 8-byte pieces (v_permlane32_swap) so that every lane holds 8 consecutive head
                // dims: 16-byte stores, four per item instead of eight, each row's 128 bytes touched by four instructions.
                // Through a buffer resource over the image's rows of this head: a 32-bit lane offset, rows >= N out of range.
                const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(out + (size_t)b * N * D + (size_t)h * DH), 0, (unsigned)(N - 1) * (unsigned)(D * 2) + (unsigned)(DH * 2), 0x00020000);
                const unsigned ooff = (q0 + r < N) ? (unsigned)(q0 + r) * (unsigned)(D * 2) + (unsigned)(16 * hi) : 0x80000000u;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
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
                        // lanes < 32 keep piece 2pr and receive their partner's; lanes >= 32 keep piece 2pr+1
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
        prev_active = active;
        if (j + 1 < nmy) {
            bh = item_bh(j + 1, &qb);
            q0 = (qb * NW + wave) * 32;
        }
    }
    wait_vm<0>();  // the dummy DMAs of the tail must land before the LDS is released
}

}  // namespace

// head_dim 64, bf16. Needs at least R-1 key tiles per item (the counted waits assume the stores and
// Q loads of one item boundary are out of the window before the next boundary).
bool attention_dh64_eligible(int N) { return N > 32 * (R - 2); }

int launch_attention_dh64(const void* qkv, void* out, float* lse, int batch, int N, int H, hipStream_t stream) {
    const int qblocks = ceil_div(ceil_div(N, 32), NW);
    const long nitems = (long)batch * H * qblocks;
    VIT_REQUIRE(nitems < (1L << 31), "attention: too many (batch, head, query block) items");
    VIT_REQUIRE((long)N * 3 * H * DH * 2 < (1L << 31), "attention: one image's qkv rows exceed 2 GiB");
    static PerDeviceInt resident_of;  // resident workgroups, per device
    std::atomic<int>* rs = resident_of.slot();
    int resident = rs ? rs->load(std::memory_order_relaxed) : 0;
    if (resident == 0) {
        int cus, per_cu = 0;
        VIT_TRY_RC(vit_cu_count(&cus));
        VIT_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)attention_dh64_kernel<false>, NW * 64, 0));
        resident = cus * (per_cu > 0 ? per_cu : 1);
        if (rs) rs->store(resident, std::memory_order_relaxed);
    }
    const int grid = nitems < resident ? (int)nitems : resident;
    const float scale_log2e = LOG2E / sqrtf((float)DH);
    const int D = H * DH;
    const unsigned char* p = (const unsigned char*)qkv;
    AttnSrc src{p, p + (size_t)D * 2, p + (size_t)2 * D * 2, (long)N * 3 * D * 2, (long)DH * 2, 3 * D * 2};
    if (lse)
        hipLaunchKernelGGL(attention_dh64_kernel<true>, dim3(grid), dim3(NW * 64), 0, stream, src, (bf16_t*)out, lse, N, H,
                           (int)nitems, qblocks, scale_log2e);
    else
        hipLaunchKernelGGL(attention_dh64_kernel<false>, dim3(grid), dim3(NW * 64), 0, stream, src, (bf16_t*)out, lse, N, H,
                           (int)nitems, qblocks, scale_log2e);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}
