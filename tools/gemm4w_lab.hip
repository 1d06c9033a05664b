// LAB ONLY (not part of the product library; included by tools/gemm_lab.hip): the 4-wave retile of the persistent GEMM
// proposed in round 1's review, built, checked bit-equal to the shipped kernel and measured SLOWER.
//
// bf16 MFMA GEMM, persistent kernel with ONE wave per SIMD (4 waves x 128x128 per 256x256 tile):
//     C[M,N] = epi(A[M,K] * W[N,K]^T + bias),   K % 128 == 0, K >= 256
// (same contract as gemm_bf16_8p.hip; modeling_vit.py:216-218,:236,:250-252).
//
// gfx950 design:
//  * 256 threads = 4 waves (wr, wc) in 2 x 2, each owns a 128 x 128 quarter of the 256 x 256 tile: 64 accumulators of
//    16x16 (256 registers, in AGPRs), 512 registers per lane available at one wave per SIMD.
//  * Operands arrive as UNITS of 16 KiB = 128 LDS rows x 128 B (64 bf16 of k): A unit u holds tile rows
//    wr'*128 + u*64 + [0,64) for wr' = 0,1 (LDS rows wr'*64 + ...), B unit u the W rows of tile columns
//    wc'*128 + 8c + 4u + ni (LDS row wc'*64 + ni*16 + c): with that interleave a lane's 8 fragments of one output
//    row are 8 CONSECUTIVE columns (one 16-byte store, 16 lanes cover 256 B of a row).
//  * A K-tile (BK = 64) is 4 phases of 32 v_mfma_f32_16x16x32_bf16 (a 64x64 quadrant x K = 64 per wave):
//        ph1 (A0,B0)  ph2 (A0,B1)  ph3 (A1,B1)  ph4 (A1,B0)
//    Every unit is read from LDS ONCE per wave (8 ds_read_b128) into registers one phase before its first use, and
//    the reads, the MFMAs and the LDS-DMA of a later unit are interleaved in one instruction stream (no partner wave):
//    unit stream  s = 4t+0: A0(t), 4t+1: B0(t), 4t+2: B1(t), 4t+3: A1(t); global phase g = 4t + ph - 1 reads unit g+2,
//    multiplies what phase g-1 (and earlier) read, and stages unit g+9 into ring slot (g+1) & 7 (8 slots = 128 KiB):
//    the DMA runs 7 units = 1.75 K-tiles ahead with counted waits only (vmcnt(24): 6 units may stay in flight).
//  * ONE s_barrier per phase: behind it every wave's pieces of unit g+2 have landed (each wave waited for its own at
//    the end of phase g-1) and every wave's reads of unit g+1 have returned (lgkmcnt(0) at the end of phase g-1), so
//    slot (g+1) & 7 may be overwritten.
//  * The unit stream is continuous across output tiles (the DMA cursor switches to the next tile 2.25 K-tiles before
//    the MFMAs do), LDS image and swizzle as in gemm_bf16_8p.hip.
#include "../vit_amd/csrc/vit_common.h"
#include <type_traits>

namespace g4 {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 256, BN = 256, BK = 64, NTHREADS = 256;
constexpr int UNIT = 128 * BK * 2;   // 16 KiB
constexpr int NSLOT = 8;
constexpr int LDS_BYTES = NSLOT * UNIT;  // 128 KiB

#define G4_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define G4_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define G4_SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void bufdma16(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, int uni_off, unsigned char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_wave_base), 16, (int)lane_off, uni_off, 0, 0);
}
// ds_read_b128 through inline asm: a C++ LDS load after an LDS-DMA draws a vmcnt(0) from the compiler
template <int OFF>
__device__ __forceinline__ void lds_read16(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

struct Params {
    const bf16_t* A;
    const bf16_t* W;
    const float* bias;
    bf16_t* C;
    int M, N, K, lda;
    int ldw;     // row pitch of W (elements); lab: may differ from lda (timing-only variants read a W allocated with slack)
    int NT, T;   // tiles along N, tiles in all
    int dbg;     // lab: bit0 = no output stores
};

struct Frag {
    bf16x8 f[4][2];  // [16-row block][k-step of 32]
};

// VAR (lab only, timing-only builds): bit0 no DMA in the phases, bit1 no barrier, bit2 no LDS reads, bit3 no MFMA,
// bit4 (A) / bit5 (W) operand panels addressed as if K-tile-packed ([K-tile][256 rows][128 B] inside each 256-row panel: same bytes, same
// panel, contiguous 32 KiB per K-tile instead of 256 pieces of 128 B at the row stride)
template <int EPI, int VAR = 0>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm4w_kernel(const Params P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // ---- tile list (as gemm_bf16_8p.hip: blocks b and b+8 share an XCD and walk one contiguous chunk together)
    const int G = gridDim.x, bid = blockIdx.x;
    const int XL = G < 8 ? G : 8;
    const int x = bid % XL, l = bid / XL;
    const int gx = (G - x + XL - 1) / XL;
    const int tq = P.T / XL, tr = P.T % XL;
    const int cnt = tq + (x < tr ? 1 : 0);
    const int start = x * tq + (x < tr ? x : tr);
    if (l >= cnt) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int i = lane & 15, q = lane >> 4;

    // ---- DMA side: this wave's piece of a unit is LDS rows wave*32 + j*8 + (lane>>3), j = 0..3, chunk position lane&7
    unsigned a_of[2][4], w_of[2][4];
    __amdgpu_buffer_rsrc_t Ar, Wr;
    auto set_tile = [&](int tile_idx) __attribute__((always_inline)) {
        const int mt = tile_idx / P.NT, nt = tile_idx - mt * P.NT;
        const int m0 = mt * BM, n0 = nt * BN;
        int lane_v = lane;
        const unsigned lda2 = (unsigned)P.lda * 2u, ldw2 = (unsigned)P.ldw * 2u;
        asm volatile("" : "+v"(lane_v));  // recomputed here, once per tile: hoisted out of the loop these values only get spilled
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wave * 32 + j * 8 + (lane_v >> 3);   // LDS row of the unit
                const unsigned ch = (unsigned)(((lane_v & 7) ^ ((r >> 1) & 7)) << 4);  // source chunk (bytes) for this position
                const int rr = r & 63;
                int ar = (r >> 6) * 128 + u * 64 + rr;
                if (m0 + ar > P.M - 1) ar = P.M - 1 - m0;
                unsigned ta = (unsigned)ar * ((VAR & 16) ? 128u : lda2);   // kept apart from the add: fused into a 64-bit mad the offset lives in a register PAIR
                asm volatile("" : "+v"(ta));
                a_of[u][j] = ta + ch;
                int wrow = (r >> 6) * 128 + (rr & 15) * 8 + u * 4 + (rr >> 4);
                if (n0 + wrow > P.N - 1) wrow = P.N - 1 - n0;
                unsigned tw = (unsigned)wrow * ((VAR & 32) ? 128u : ldw2);
                asm volatile("" : "+v"(tw));
                w_of[u][j] = tw + ch;
            }
        const unsigned span = (unsigned)BM * (unsigned)P.lda * 2u;
        Ar = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A + (size_t)m0 * P.lda), 0, span, 0x00020000);
        Wr = __builtin_amdgcn_make_buffer_rsrc((void*)(P.W + (size_t)n0 * P.ldw), 0, (unsigned)BN * ldw2, 0x00020000);
    };
    unsigned char* const dma_base = smem + wave * 4096;
    // TYPE 0: A0, 1: B0, 2: B1, 3: A1
    auto stage = [&](auto TYPEC, int slot, int kt) __attribute__((always_inline)) {
        constexpr int TYPE = decltype(TYPEC)::value;
        constexpr int u = (TYPE == 0 || TYPE == 1) ? 0 : 1;
        unsigned char* d = dma_base + slot * UNIT;
        const int koa = kt * ((VAR & 16) ? 256 * 128 : BK * 2), kow = kt * ((VAR & 32) ? 256 * 128 : BK * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (TYPE == 0 || TYPE == 3) bufdma16(Ar, a_of[u][j], koa, d + j * 1024);
            else bufdma16(Wr, w_of[u][j], kow, d + j * 1024);
        }
    };
    auto stage1 = [&](auto TYPEC, int slot, int kt, auto JC) __attribute__((always_inline)) {
        constexpr int TYPE = decltype(TYPEC)::value, j = decltype(JC)::value;
        constexpr int u = (TYPE == 0 || TYPE == 1) ? 0 : 1;
        unsigned char* d = dma_base + slot * UNIT;
        const int koa = kt * ((VAR & 16) ? 256 * 128 : BK * 2), kow = kt * ((VAR & 32) ? 256 * 128 : BK * 2);
        if constexpr (TYPE == 0 || TYPE == 3) bufdma16(Ar, a_of[u][j], koa, d + j * 1024);
        else bufdma16(Wr, w_of[u][j], kow, d + j * 1024);
    };

    // ---- read side
    const unsigned smem_lds = (unsigned)(uintptr_t)LDS_PTR(smem);
    const int sw = i >> 1;
    const unsigned ck0 = (unsigned)((q ^ sw) << 4), ck1 = (unsigned)(((4 + q) ^ sw) << 4);
    // [k-step][slot half]: slots 4..7 lie beyond the 16-bit offset field
    unsigned ra[2][2], rb[2][2];
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
        ra[0][hs] = smem_lds + (unsigned)(hs * 4 * UNIT + (wr * 64 + i) * 128) + ck0;
        ra[1][hs] = smem_lds + (unsigned)(hs * 4 * UNIT + (wr * 64 + i) * 128) + ck1;
        rb[0][hs] = smem_lds + (unsigned)(hs * 4 * UNIT + (wc * 64 + i) * 128) + ck0;
        rb[1][hs] = smem_lds + (unsigned)(hs * 4 * UNIT + (wc * 64 + i) * 128) + ck1;
    }

    Frag AX, AY, BX, BY;
    f32x4 acc[2][4][2][4];  // [ha][mi][hb][ni]

    // one read of a unit: fragment (block b, k-step ks) for n = 0..7 -> b = n >> 1, ks = n & 1
    auto read1 = [&](Frag& F, auto SLOTC, bool isA, auto NC) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(SLOTC)::value, n = decltype(NC)::value;
        constexpr int b = n >> 1, ks = n & 1;
        const unsigned addr = isA ? ra[ks][SLOT >> 2] : rb[ks][SLOT >> 2];
        lds_read16<(SLOT & 3) * UNIT + b * 2048>(F.f[b][ks], addr);
    };

    const int nk = P.K / BK;  // even, >= 4

    // ---- one phase. G8 = g & 7 (compile time): reads unit g+2 from slot (G8+2)&7 into RD, multiplies MA x MB into
    // acc[HA][.][HB][.], stages unit g+9 (type (G8+1)&3, K-tile ckt) into slot (G8+1)&7. VMN = the counted wait.
    int ckt = 0;          // K-tile of the DMA cursor
    int cursor_next = 0;  // tile the cursor moves to when it passes the end of its K range
    auto phase = [&](auto G8C, Frag& MA, Frag& MB, auto HAC, auto HBC, Frag& RD, auto ZEROC, auto VMNC) __attribute__((always_inline)) {
        constexpr int G8 = decltype(G8C)::value, HA = decltype(HAC)::value, HB = decltype(HBC)::value;
        constexpr bool ZERO = decltype(ZEROC)::value;
        constexpr int VMN = decltype(VMNC)::value;
        constexpr int RSLOT = (G8 + 2) & 7, DSLOT = (G8 + 1) & 7, DTYPE = (G8 + 1) & 3;
        constexpr bool RD_IS_A = ((G8 + 2) & 3) == 0 || ((G8 + 2) & 3) == 3;
        using DT = std::integral_constant<int, DTYPE>;
        using RS = std::integral_constant<int, RSLOT>;
        if constexpr (!(VAR & 2)) __builtin_amdgcn_s_barrier();
        G4_SB();
        if constexpr (DTYPE == 0) {  // the cursor moves to the next K-tile (and, past the end, to the next tile)
            ++ckt;
            if (ckt == nk) {
                ckt = 0;
                set_tile(cursor_next);
            }
        }
        // group: (first 4) two LDS reads, (last 4) one DMA, 4 MFMA
        auto group = [&](auto GC) __attribute__((always_inline)) {
            constexpr int grp = decltype(GC)::value;
            if constexpr (!(VAR & 4) && grp < 4) {
                read1(RD, RS{}, RD_IS_A, std::integral_constant<int, 2 * grp>{});
                read1(RD, RS{}, RD_IS_A, std::integral_constant<int, 2 * grp + 1>{});
            }
            if constexpr (!(VAR & 1) && grp >= 4) stage1(DT{}, DSLOT, ckt, std::integral_constant<int, (grp >= 4 ? grp - 4 : 0)>{});
            G4_SB();
            constexpr int ks = grp >> 2, mi = grp & 3;
            if constexpr (!(VAR & 8)) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const f32x4 c = (ZERO && ks == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[HA][mi][HB][ni];
                    acc[HA][mi][HB][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(MA.f[mi][ks], MB.f[ni][ks], c, 0, 0, 0);
                }
            }
            G4_SB();
        };
        group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{});
        group(std::integral_constant<int, 2>{}); group(std::integral_constant<int, 3>{});
        group(std::integral_constant<int, 4>{}); group(std::integral_constant<int, 5>{});
        group(std::integral_constant<int, 6>{}); group(std::integral_constant<int, 7>{});
        G4_WAIT_LGKM0();
        G4_WAIT_VM(VMN);
#pragma unroll
        for (int b = 0; b < 4; ++b) asm volatile("" : "+v"(RD.f[b][0]), "+v"(RD.f[b][1]));
        G4_SB();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>;
    using I7 = std::integral_constant<int, 7>;
    using BT = std::true_type;
    using BF = std::false_type;
    // two K-tiles (8 phases). ZERO: the first of them starts the accumulators; ST: stores of the preceding epilogue
    // that may still be in flight (they are younger than the units waited for in the first 6 phases)
    auto ktile_pair = [&](auto ZEROC, auto STC) __attribute__((always_inline)) {
        constexpr int ST = decltype(STC)::value;
        using V0 = std::integral_constant<int, (24 + ST > 63 ? 63 : 24 + ST)>;
        using V1 = std::integral_constant<int, 24>;
        phase(I0{}, AX, BX, I0{}, I0{}, BY, ZEROC, V0{});
        phase(I1{}, AX, BY, I0{}, I1{}, AY, ZEROC, V0{});
        phase(I2{}, AY, BY, I1{}, I1{}, AX, ZEROC, V0{});
        phase(I3{}, AY, BX, I1{}, I0{}, BY, ZEROC, V0{});
        phase(I4{}, AX, BY, I0{}, I0{}, BX, BF{}, V0{});
        phase(I5{}, AX, BX, I0{}, I1{}, AY, BF{}, V0{});
        phase(I6{}, AY, BX, I1{}, I1{}, AX, BF{}, V1{});
        phase(I7{}, AY, BY, I1{}, I0{}, BX, BF{}, V1{});
    };

    // ---- prologue: units 0..7 (K-tiles 0 and 1), read units 0 and 1, stage unit 8 = A0(2)
    int lt = l;
    int cur_tile = start + lt;
    set_tile(cur_tile);
    stage(I0{}, 0, 0);
    stage(I1{}, 1, 0);
    stage(I2{}, 2, 0);
    stage(I3{}, 3, 0);
    stage(I0{}, 4, 1);
    stage(I1{}, 5, 1);
    stage(I2{}, 6, 1);
    stage(I3{}, 7, 1);
    G4_WAIT_VM(24);
    __builtin_amdgcn_s_barrier();
    G4_SB();
    read1(AX, I0{}, true, I0{}); read1(AX, I0{}, true, I1{}); read1(AX, I0{}, true, I2{}); read1(AX, I0{}, true, I3{});
    read1(AX, I0{}, true, I4{}); read1(AX, I0{}, true, I5{}); read1(AX, I0{}, true, I6{}); read1(AX, I0{}, true, I7{});
    read1(BX, I1{}, false, I0{}); read1(BX, I1{}, false, I1{}); read1(BX, I1{}, false, I2{}); read1(BX, I1{}, false, I3{});
    read1(BX, I1{}, false, I4{}); read1(BX, I1{}, false, I5{}); read1(BX, I1{}, false, I6{}); read1(BX, I1{}, false, I7{});
    G4_WAIT_LGKM0();
#pragma unroll
    for (int b = 0; b < 4; ++b) asm volatile("" : "+v"(AX.f[b][0]), "+v"(AX.f[b][1]), "+v"(BX.f[b][0]), "+v"(BX.f[b][1]));
    // unit 0's slot is free once every wave has read it -- but unit 8 goes to slot 0 only behind a barrier
    __builtin_amdgcn_s_barrier();
    G4_SB();
    ckt = 2;
    stage(I0{}, 0, 2);
    // Everything staged so far lands before the loop: the first six phases of a tile count the preceding epilogue's
    // stores into their waits (none are in flight before the first tile, so the units they wait for must be in already)
    G4_WAIT_VM(0);

    for (;;) {
        const int nlt = lt + gx;
        const bool has_next = nlt < cnt;
        cursor_next = has_next ? start + nlt : cur_tile;  // no next tile: harmless re-prefetch of this one
        ktile_pair(BT{}, std::integral_constant<int, 32>{});
        for (int t = 2; t < nk; t += 2) ktile_pair(BF{}, I0{});

        // ---- epilogue: lane (i, q) holds for row ha*64 + mi*16 + 4q + j of the wave's quarter the 8 consecutive
        // columns 8i + (4 hb + ni)
        {
            const int mt = cur_tile / P.NT, nt = cur_tile - mt * P.NT;
            const int m0 = mt * BM, n0 = nt * BN;
            const int nb = n0 + wc * 128 + 8 * i;
            const bool col_ok = nb < P.N;
            const int rows_here = (P.M - m0 < BM) ? P.M - m0 : BM;
            const __amdgpu_buffer_rsrc_t Cr = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(P.C + (size_t)m0 * P.N), 0, (P.dbg & 1) ? 0u : (unsigned)rows_here * (unsigned)P.N * 2u, 0x00020000);
            const unsigned voff0 = col_ok ? (unsigned)((wr * 128 + 4 * q) * P.N + nb) * 2u : 0x80000000u;
            float bias[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (P.bias != nullptr && col_ok) {
                const float4 b0 = *(const float4*)(P.bias + nb), b1 = *(const float4*)(P.bias + nb + 4);
                bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w;
                bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
            }
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    f32x4 v[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        v[c] = acc[ha][mi][c >> 2][c & 3] + bias[c];
                        if constexpr (EPI == VIT_EPI_GELU) {  // the packed polynomial on row pairs (adjacent registers)
                            f32x2 a{v[c][0], v[c][1]}, b{v[c][2], v[c][3]};
                            act_gelu_pk2(a, b);
                            v[c] = f32x4{a.x, a.y, b.x, b.y};
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned voff = voff0 + (unsigned)((ha * 64 + mi * 16 + j) * P.N) * 2u;
                        bf16x8 ov;
#pragma unroll
                        for (int c = 0; c < 8; ++c) ov[c] = (bf16_t)v[c][j];
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ov), Cr, (int)voff, 0, 0);
                        G4_SB();
                    }
                }
        }
        if (!has_next) break;
        lt = nlt;
        cur_tile = cursor_next;
    }
    G4_WAIT_VM(0);  // the dummy prefetch of the last tile must land before the LDS is released
}

template <int EPI, int VAR = 0>
int launch(const Params& P, int ncu, hipStream_t stream) {
    auto kern = gemm4w_kernel<EPI, VAR>;
    static bool attr_set = false;
    if (!attr_set) {
        VIT_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    const int G = P.T < ncu ? P.T : ncu;
    hipLaunchKernelGGL(kern, dim3(G), dim3(NTHREADS), LDS_BYTES, stream, P);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

}  // namespace g4

#ifdef G8_LAB
int g4_dbg = 0;
#endif
int launch_gemm_bf16_4w(const void* A, const void* W, const float* bias, const void* res, void* C, int M, int N, int K, int epi,
                        const GemmExt* ext, hipStream_t stream) {
    VIT_REQUIRE(K % 128 == 0 && K >= 256 && N % 8 == 0 && M >= 1 && (size_t)256 * K * 2 < ((size_t)1 << 31), "gemm4w: shape M=%d N=%d K=%d not eligible", M, N, K);
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        VIT_CHECK_HIP(hipGetDevice(&dev));
        VIT_CHECK_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    g4::Params P{};
    P.A = (const bf16_t*)A; P.W = (const bf16_t*)W; P.bias = bias; P.C = (bf16_t*)C;
    P.M = M; P.N = N; P.K = K; P.lda = K; P.ldw = K;
#ifdef G8_LAB
    if (g4_dbg & (1 << 20)) P.ldw = K + 64;
    if (g4_dbg & (1 << 21)) P.ldw = K + 8;
    if (g4_dbg & (1 << 22)) P.ldw = K + 32;
#endif
    P.NT = ceil_div(N, g4::BN);
    P.T = ceil_div(M, g4::BM) * P.NT;
#ifdef G8_LAB
    P.dbg = g4_dbg;
#endif
#ifdef G8_LAB
    switch ((g4_dbg >> 1) & 0xFF) {
        case 1: return g4::launch<VIT_EPI_BIAS, 1>(P, ncu, stream);
        case 2: return g4::launch<VIT_EPI_BIAS, 2>(P, ncu, stream);
        case 3: return g4::launch<VIT_EPI_BIAS, 3>(P, ncu, stream);
        case 4: return g4::launch<VIT_EPI_BIAS, 4>(P, ncu, stream);
        case 8: return g4::launch<VIT_EPI_BIAS, 8>(P, ncu, stream);
        case 6: return g4::launch<VIT_EPI_BIAS, 6>(P, ncu, stream);
        case 7: return g4::launch<VIT_EPI_BIAS, 7>(P, ncu, stream);
        case 5: return g4::launch<VIT_EPI_BIAS, 5>(P, ncu, stream);
        case 16: return g4::launch<VIT_EPI_BIAS, 16>(P, ncu, stream);
        case 32: return epi == VIT_EPI_GELU ? g4::launch<VIT_EPI_GELU, 32>(P, ncu, stream) : g4::launch<VIT_EPI_BIAS, 32>(P, ncu, stream);
        case 48: return g4::launch<VIT_EPI_BIAS, 48>(P, ncu, stream);
        case 56: return g4::launch<VIT_EPI_BIAS, 56>(P, ncu, stream);
    }
#endif
    switch (epi) {
        case VIT_EPI_BIAS: return g4::launch<VIT_EPI_BIAS>(P, ncu, stream);
        case VIT_EPI_GELU: return g4::launch<VIT_EPI_GELU>(P, ncu, stream);
    }
    return vit_set_error(VIT_E_INVALID, "gemm4w: epilogue %d not built", epi);
}
