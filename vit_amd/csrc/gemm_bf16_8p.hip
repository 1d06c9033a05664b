// bf16 MFMA GEMM, persistent deep-pipelined kernel for the encoder's big Linear layers:
//     C[M,N] = epi(A[M,K] * W[N,K]^T + bias (+ residual)),   K % 128 == 0, K >= 256
// (q/k/v packed projection modeling_vit.py:216-218, o_proj + residual :236,:277, fc1 + GELU
// :250-251, fc2 + residual :252,:284), and -- EPI_PATCH -- the patch embedding (Conv2d with
// kernel = stride = P, modeling_vit.py:62-69) for patch sizes whose K-tiles are whole pixel rows
// of one channel (P = 16, 32): the A operand is DMA'd straight from the pixel tensor (per-lane
// offset of the patch row + a wave-uniform offset per K-tile), the epilogue moves rows past each
// image's CLS slot and adds the position table (:129-161). Shapes it does not take (K tails, other
// patch sizes) stay on gemm_bf16.hip.
//
// gfx950 design (DESIGN.md "GEMM"):
//  * Persistent: one 512-thread workgroup per CU walks a list of 256 x 256 output tiles. Blocks
//    b and b+8 share an XCD; each XCD label owns one contiguous chunk of the (mt-major) tile
//    space and its blocks walk it together, so neighbouring tiles (same A panel) meet in one L2.
//  * 256 x 256 x 64 block tile, 8 waves = 2 teams (wr) x 4 (wc), 128 KiB LDS = 2 K-tile buffers
//    x {A_h0, A_h1, B_h0, B_h1} half-tiles (128 rows x 64 bf16 each). Wave (wr, wc) owns output
//    rows {h*128 + wr*64 + [0,64)}, h = 0,1, and the 64 columns wc*64 + [0,64) (32 staged in each
//    B half): one 64x32 quadrant per (A half, B half) pair.
//  * A K-tile is 4 phases of 16 v_mfma_f32_16x16x32_bf16 (one quadrant x K=64):
//        ph1 (A0,B0)  ph2 (A0,B1)  ph3 (A1,B1)  ph4 (A1,B0)
//    Each phase = {ds_read_b128 this phase's new fragments, issue ONE half-tile of LDS-DMA
//    (2 x buffer_load_dwordx4 ... lds per thread), s_barrier, lgkmcnt(0), 16 MFMA, s_barrier}.
//    The two teams run staggered by one barrier, so on every SIMD one wave's MFMA section
//    overlaps its partner's LDS-read / DMA-issue section.
//  * Half-tiles retire at different phases (A_h0, B_h0 after ph1; B_h1 after ph2; A_h1 after
//    ph3), so the DMA stream runs 3-4 half-tiles ahead of its consumer:
//        ph1: stage B_h1(t+1)   ph2: stage A_h1(t+1)   ph3: stage A_h0(t+2)   ph4: stage B_h0(t+2)
//    with counted waits only -- vmcnt(6) at ph1 (retires B_h1/A_h1 of the current K-tile, read
//    from ph2/ph3 on) and vmcnt(8) at ph4 (retires A_h0/B_h0 of the next K-tile). Every buffer
//    is restaged >= 2 phases after its last ds_read and read >= 1 phase after the wait that
//    retires it (the rule for staggered wave groups).
//  * The K-tile stream is continuous across output tiles: the last two K-tiles of a tile
//    already stage the first two of the next one, so the DMA pipeline never drains.
//  * LDS rows are 128 B; 16-B chunk c of row r sits at chunk position c ^ ((r>>1)&7). The DMA
//    destination is lane-linear, so the swizzle is on the per-lane SOURCE offset and on the
//    ds_read_b128 address: conflict-free fragment reads. DMA addresses are buffer-resource
//    base (SGPR) + 32-bit lane offset + SGPR k offset: no 64-bit per-lane arithmetic.
//  * Epilogue: weight rows are staged interleaved (LDS row wc*32 + ni*16 + c of B half h holds
//    column wc*64 + 4c + 2h + ni), so the four 16-column fragments of a wave give every lane 4
//    CONSECUTIVE columns and 16 lanes cover one full 128-byte line of an output row. Stores and
//    residual loads go through buffer resources (hardware range check masks the ragged M/N
//    edges, no branches); the residual tile is fetched up front into the dead fragment
//    registers. Measured on MI355X (tools/store_lab.hip): this store shape moves a 128 KiB tile
//    in 2.3 us per CU, the natural 16-rows-x-64-B MFMA shape needs 4.5 us.
//  * Residual epilogue with row sums (template flag RST; vit_forward's o_proj / fc2): a wave's 64 columns of a tile are one
//    64-column group of the output row, so it also leaves (sum, sum of squares) of the bf16 values it stores, per row and
//    group -- the next folded LayerNorm's statistics without a pass over the rows (row16_stats in vit_common.h: the same
//    instruction sequence as the stand-alone statistics kernel, hence the same bits).
#include "vit_common.h"
#include <type_traits>
#include <stdlib.h>

namespace g8 {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef G8_STORE_AUX
#define G8_STORE_AUX 2  // cache policy of the output stores: nt (streamed through L2, not retained)
#endif
constexpr int BM = 256, BN = 256, BK = 64, NTHREADS = 512;
constexpr int HALF_BYTES = 128 * BK * 2;        // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES;       // 64 KiB: A_h0, A_h1, B_h0, B_h1
constexpr int LDS_BYTES = 2 * BUF_BYTES;        // 128 KiB
constexpr int OFF_A0 = 0, OFF_A1 = HALF_BYTES, OFF_B0 = 2 * HALF_BYTES, OFF_B1 = 3 * HALF_BYTES;

#define G8_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define G8_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// LDS-DMA through a buffer resource: address = resource base + per-lane byte offset (VGPR) +
// wave-uniform byte offset (SGPR). No 64-bit per-lane address arithmetic.
__device__ __forceinline__ void bufdma16(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, int uni_off,
                                         unsigned char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_wave_base), 16, (int)lane_off, uni_off, 0, 0);
}

// Folded LayerNorm: read n of the epilogue brings two rows' (mean * rstd, rstd) of the lane's rows
// (ha = n / (2 MI), mi = (n >> 1) % MI, row pair n & 1; an A half has 32 MI rows) into slot n % 3 of a three-deep window.
template <int N, int MI>
__device__ __forceinline__ void st_read(f32x4 (&spw)[3], unsigned st_lds) {
    static_assert(N < 4 * MI, "statistics read past the tile");
    constexpr int off = ((N / (2 * MI)) * 32 * MI + ((N >> 1) % MI) * 16 + (N & 1) * 2) * 8;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(spw[N % 3]) : "v"(st_lds), "n"(off) : "memory");
}
template <int MI>
__device__ __forceinline__ void st_read_sw(int n, f32x4 (&spw)[3], unsigned st_lds) {  // n is a constant after unrolling
#define G8_ST_CASE(K) case K: if constexpr (K < 4 * MI) st_read<K, MI>(spw, st_lds); break
    switch (n) {
        G8_ST_CASE(2); G8_ST_CASE(3); G8_ST_CASE(4); G8_ST_CASE(5); G8_ST_CASE(6); G8_ST_CASE(7); G8_ST_CASE(8);
        G8_ST_CASE(9); G8_ST_CASE(10); G8_ST_CASE(11); G8_ST_CASE(12); G8_ST_CASE(13); G8_ST_CASE(14); G8_ST_CASE(15);
    }
#undef G8_ST_CASE
}

// ---------------------------------------------------------------------------------------
// Persistent form: one workgroup per CU walks a list of output tiles. The K-tile stream is
// continuous across tile boundaries (the last two K-tiles of a tile already stage the first
// two K-tiles of the next tile), so the DMA pipeline never drains; the epilogue of both
// teams runs in the same barrier interval.
struct PParams {
    const bf16_t* A;
    const bf16_t* W;
    const float* bias;
    const bf16_t* res;
    bf16_t* C;       // bf16 output (OUTM 0, 2) or, reinterpreted, the fp32 output (OUTM 1)
    bf16_t* Cpre;    // OUTM 2: second output, the pre-activation A W^T + bias (training forward)
    int M, N, K;     // K = reduction length of ONE split
    int lda;         // row pitch (elements) of A and W: K * splits
    int NT;      // tiles along N
    int MT, GC;  // tiles along M (of 64 MI rows each); column-group width of the tile order (see tile_coords)
    int TMN;     // output tiles per split = MT * NT
    int T;       // total work items = TMN * splits; split s writes the fp32 slab C + s*M*N (OUTM 1)
    // LNF (template flag): LayerNorm folded in -- epilogue value = rstd * acc - (mean * rstd) * ln_c + bias
    const float* ln_stats;   // [M][2] = (mean * rstd, rstd)
    const float* ln_c;       // [N]
    // RST (template flag) on the residual epilogue (N % 64 == 0): row sums of the stored bf16 output, one slab per 64-column group:
    // row_part[(col / 64) * part_pitch + row] = (sum x, sum x^2) over those 64 columns (see row_stats_finalize_kernel)
    float* row_part;
    long part_pitch;         // rows of one slab (the whole launch's M, also for a row segment of the few-tile planner)
    // EPI_PATCH only: A is the pixel tensor [B, C, image, image]; GEMM row g = b * np + py * grid + px
    const float* pos;        // position table fp32 [np + 1, N]
    int np, grid, patch, image;
    unsigned img_bytes;      // C * image * image * 2
    int a_kshift, a_kmask;   // K-tile kt starts at byte (kt >> a_kshift) * a_chs + (kt & a_kmask) * a_kts of a patch
    unsigned a_chs, a_kts;
};

constexpr int EPI_PATCH = 100;        // internal: bias + position add + row remap past the CLS slot, A from pixels
constexpr int EPI_PATCH_DENSE = 101;  // the same epilogue on a dense A (rows = patches already gathered, any patch size)

// EPI: VIT_EPI_* or EPI_PATCH; for VIT_EPI_DGELU / VIT_EPI_DQUICK_GELU `res` is the pre-activation the
// gradient is multiplied with. OUTM: 0 = bf16 C, 1 = fp32 C (weight gradients), 2 = bf16 C plus
// the pre-activation in Cpre.
// MI: 16-row MFMA fragments per wave and A half -- the tile is 64 MI rows x 256 columns (MI = 4: 256 x 256, the shape
// everything above describes). MI = 3 / 2 are the FEW-TILE forms for shapes whose 256-row tiles fill the chip badly
// (CLIP-L/14 at 16 images: 148 tiles on 256 CUs; pick_mi below): an A half holds 32 MI rows in its first LDS rows (the
// DMA lanes of the other rows are out of range and move nothing), a wave owns 16 MI of them, and a phase has 4 MI MFMAs.
// Everything else -- schedule, waits, LDS image, k order -- is the same, so an output element gets the SAME BITS whatever
// MI computed it: the tile height is a pure speed choice, like the tile order.
template <int EPI, int OUTM, bool LNF = false, int MI = 4, bool RST = false>
__global__ __launch_bounds__(NTHREADS, 2) void gemm8p_persist_kernel(const PParams P) {
    static_assert(!RST || (EPI == VIT_EPI_RESIDUAL && OUTM == 0 && !LNF), "row sums ride on the bf16 residual epilogue");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert(MI >= 2 && MI <= 4, "tile height 128 / 192 / 256");
    static_assert(MI == 4 || (EPI != EPI_PATCH && EPI != EPI_PATCH_DENSE), "the patch epilogues assume 256-row tiles");
    constexpr int BMT = 64 * MI;  // tile rows
    constexpr int HR = 32 * MI;   // rows of an A half
    constexpr int WRR = 16 * MI;  // rows of a wave inside an A half

    // ---- tile list of this workgroup. Blocks b and b+8 share an XCD (label x = b % XL); label
    // x owns one contiguous chunk of the (mt-major) tile space and its blocks walk the chunk
    // together, so an XCD's L2 sees neighbouring tiles (shared A panels / W panels).
    const int G = gridDim.x, bid = blockIdx.x;
    const int XL = G < 8 ? G : 8;
    const int x = bid % XL, l = bid / XL;
    const int gx = (G - x + XL - 1) / XL;          // blocks with this label
    const int tq = P.T / XL, tr = P.T % XL;
    const int cnt = tq + (x < tr ? 1 : 0);         // tiles in this label's chunk
    const int start = x * tq + (x < tr ? x : tr);
    if (l >= cnt) return;                          // uniform per block, before any barrier

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int i = lane & 15, q = lane >> 4;
    const int p = tid & 7;

    // Tile order inside a split: column groups of GC tile columns, and inside a group mt-major. The 32 workgroups of
    // an XCD walk consecutive tiles, i.e. ~32 / GC tile rows x GC tile columns at a time: with GC chosen so that the
    // group's W panels (GC x 256 x K) stay in the XCD's 4 MiB L2 beside the A panels passing through, W is fetched
    // from beyond L2 once per group instead of once per round (GC >= NT: plain mt-major order).
    auto tile_coords = [&](int t2, int* mt, int* nt) {
        const int per_group = P.MT * P.GC;
        const int g = t2 / per_group, rem = t2 - g * per_group;
        const int left = P.NT - g * P.GC;
        const int gc = left < P.GC ? left : P.GC;
        const int m = rem / gc;
        *mt = m;
        *nt = g * P.GC + rem - m * gc;
    };
    unsigned a_of[2][2], w_of[2][2];          // per-lane byte offsets from the tile's first A / W row
    __amdgpu_buffer_rsrc_t Ar, Wr;            // wave-uniform buffer resources based at those rows
    auto set_tile = [&](int tile_idx) {   // staging offsets / bases for a tile
        const int sp = tile_idx / P.TMN, t2 = tile_idx - sp * P.TMN;
        int mt, nt;
        tile_coords(t2, &mt, &nt);
        const int m0 = mt * BMT, n0 = nt * BN;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = j * 64 + (tid >> 3);
                const int sw = (r >> 1) & 7;
                int ar = h * HR + r;
                if (m0 + ar > P.M - 1) ar = P.M - 1 - m0;
                if constexpr (EPI == EPI_PATCH) {
                    // patch row: offset of its first pixel from the tile's first image, plus this lane's
                    // 16-byte chunk inside the K-tile (chunk c = 8 pixels: row (8c / P), column (8c % P))
                    const int g = m0 + ar, b = g / P.np, rem = g - b * P.np;
                    const int py = rem / P.grid, px = rem - py * P.grid;
                    const int c8 = (p ^ sw) << 3;
                    const unsigned pix = (unsigned)((py * P.patch + c8 / P.patch) * P.image + px * P.patch + c8 % P.patch);
                    a_of[h][j] = (unsigned)(b - m0 / P.np) * P.img_bytes + pix * 2u;
                } else {
                    a_of[h][j] = ((unsigned)ar * (unsigned)P.lda + (unsigned)((p ^ sw) << 3)) * 2u;
                    if (MI < 4 && r >= HR) a_of[h][j] = 0x80000000u;  // LDS rows the tile does not use: out of range, nothing moves
                }
                const int rr = r & 31;
                // B half h, LDS row (wc*32 + ni*16 + c) holds tile column wc*64 + 4c + (2h + ni): the four
                // 16-column MFMA fragments of a wave interleave to 4 consecutive columns per lane
                int wrow = (r >> 5) * 64 + (rr & 15) * 4 + h * 2 + (rr >> 4);
                if (n0 + wrow > P.N - 1) wrow = P.N - 1 - n0;
                w_of[h][j] = ((unsigned)wrow * (unsigned)P.lda + (unsigned)((p ^ sw) << 3)) * 2u;
            }
        // num_records spans the (clamped) rows of the tile; offsets never exceed it
        const unsigned span = (unsigned)BN * (unsigned)P.lda * 2u;
        const unsigned span_a = (unsigned)BMT * (unsigned)P.lda * 2u;
        const size_t koff = (size_t)sp * P.K;  // this split's first k
        if constexpr (EPI == EPI_PATCH) {
            // based at the tile's first image; the (clamped) rows of a tile lie in at most 256 / np + 2 images
            const int b0 = m0 / P.np, bl = (P.M - 1) / P.np;
            const int nimg = (bl - b0 + 1 < 256 / P.np + 2) ? bl - b0 + 1 : 256 / P.np + 2;
            Ar = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)P.A + (size_t)b0 * P.img_bytes), 0,
                                                   (unsigned)nimg * P.img_bytes, 0x00020000);
        } else {
            Ar = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A + (size_t)m0 * P.lda + koff), 0, span_a, 0x00020000);
        }
        Wr = __builtin_amdgcn_make_buffer_rsrc((void*)(P.W + (size_t)n0 * P.lda + koff), 0, span, 0x00020000);
    };
    unsigned char* const dma_base = smem + wave * 1024;
    auto stage_a = [&](int h, int buf, int kt) {
        unsigned char* d = dma_base + buf * BUF_BYTES + (h ? OFF_A1 : OFF_A0);
        int ko = kt * (BK * 2);
        if constexpr (EPI == EPI_PATCH) ko = (int)((unsigned)(kt >> P.a_kshift) * P.a_chs + (unsigned)(kt & P.a_kmask) * P.a_kts);
        bufdma16(Ar, a_of[h][0], ko, d);
        bufdma16(Ar, a_of[h][1], ko, d + 8192);
    };
    auto stage_b = [&](int h, int buf, int kt) {
        unsigned char* d = dma_base + buf * BUF_BYTES + (h ? OFF_B1 : OFF_B0);
        bufdma16(Wr, w_of[h][0], kt * (BK * 2), d);
        bufdma16(Wr, w_of[h][1], kt * (BK * 2), d + 8192);
    };

    const int sw = i >> 1;
    const unsigned rd_a = (unsigned)((wr * WRR + i) * 128);
    const unsigned rd_b = (unsigned)((wc * 32 + i) * 128);
    const unsigned ck0 = (unsigned)((q ^ sw) << 4), ck1 = (unsigned)(((4 + q) ^ sw) << 4);

    bf16x8 af[MI][2];
    bf16x8 bfr[2][2][2];
    f32x4 acc[2][MI][2][2];

    auto read_a = [&](int buf, int h) {
        const unsigned char* b0 = smem + buf * BUF_BYTES + (h ? OFF_A1 : OFF_A0) + rd_a;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            af[mi][0] = *(const bf16x8*)(b0 + mi * 2048 + ck0);
            af[mi][1] = *(const bf16x8*)(b0 + mi * 2048 + ck1);
        }
    };
    auto read_b = [&](int buf, int h) {
        const unsigned char* b0 = smem + buf * BUF_BYTES + (h ? OFF_B1 : OFF_B0) + rd_b;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            bfr[h][ni][0] = *(const bf16x8*)(b0 + ni * 2048 + ck0);
            bfr[h][ni][1] = *(const bf16x8*)(b0 + ni * 2048 + ck1);
        }
    };
    // ZERO: first K-tile of an output tile -- the ks = 0 MFMA starts from a zero accumulator
    auto mma = [&](int ha, int hb, auto ZEROC) {
        constexpr bool ZERO = decltype(ZEROC)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const f32x4 c = (ZERO && ks == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[ha][mi][hb][ni];
                    acc[ha][mi][hb][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi][ks], bfr[hb][ni][ks], c, 0, 0, 0);
                }
        __builtin_amdgcn_s_setprio(0);
    };
#define G8P_SECTION(HA, HB, Z)             \
    __builtin_amdgcn_s_barrier();          \
    G8_WAIT_LGKM0();                       \
    __builtin_amdgcn_sched_barrier(0);     \
    mma(HA, HB, Z);                        \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier()

    const int nk = P.K / BK;  // even, >= 4 (checked on the host)

    // MODE 0: steady; 1: K-tile nk-2 (after ph2 the DMA stream switches to the next tile);
    // 2: K-tile nk-1 (all stages belong to the next tile). FIRST: K-tile 0 of a tile (its
    // B_h1/A_h1 were retired before the loop / at the start of the previous epilogue).
    int next_tile = 0;
    __amdgpu_buffer_rsrc_t Sr = Ar;  // LNF: this tile's rows of ln_stats (rows >= M read as 0)
    __amdgpu_buffer_rsrc_t BLr = Ar; // this tile's 256 bias values (waves 0-3) / ln_c values (waves 4-7); columns >= N read as 0
    constexpr bool SIDE = (EPI != EPI_PATCH && EPI != EPI_PATCH_DENSE);  // bias / ln_c travel through LDS
    auto ktile = [&](auto BUFC, auto MODEC, auto FIRSTC, int t) {
        constexpr int BUF = decltype(BUFC)::value, MODE = decltype(MODEC)::value;
        constexpr bool FIRST = decltype(FIRSTC)::value;
        using Z = std::integral_constant<bool, FIRST>;
        // LNF, last K-tile of a tile: this tile's 256 x (mean * rstd, rstd) go to LDS behind the ring, one 4-byte
        // DMA per lane (wave w brings dwords 64w..64w+63). Issued BEFORE the eight half-tile stage instructions of
        // this K-tile, so the vmcnt(8) of ph4 retires it in every wave, and ph4's two barriers separate that wait
        // from the epilogue, where the waves read each other's pieces.
        if constexpr (LNF && MODE == 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Sr, LDS_PTR(smem + LDS_BYTES + wave * 256), 4, lane * 4, wave * 256, 0, 0);
        // The tile's bias (and ln_c) values take the same route, 1 KiB each behind the statistics: a global load in the
        // epilogue is the youngest vector-memory operation there, so the wait for it (vmcnt(0)) also waits for the
        // next tile's staged K-tiles, with every store of the epilogue queued behind that.
        if constexpr (SIDE && MODE == 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(BLr, LDS_PTR(smem + LDS_BYTES + 2048 + wave * 256), 4, lane * 4, (wave & 3) * 256, 0, 0);
        // ph1: (A0,B0)
        read_b(BUF, 0);
        read_a(BUF, 0);
        stage_b(1, BUF ^ 1, MODE == 2 ? 0 : t + 1);
        if constexpr (!FIRST) G8_WAIT_VM(6);
        G8P_SECTION(0, 0, Z{});
        // ph2: (A0,B1)
        read_b(BUF, 1);
        stage_a(1, BUF ^ 1, MODE == 2 ? 0 : t + 1);
        G8P_SECTION(0, 1, Z{});
        if constexpr (MODE == 1) set_tile(next_tile);
        // ph3: (A1,B1)
        read_a(BUF, 1);
        stage_a(0, BUF, MODE == 0 ? t + 2 : (MODE == 1 ? 0 : 1));
        G8P_SECTION(1, 1, Z{});
        // ph4: (A1,B0)
        stage_b(0, BUF, MODE == 0 ? t + 2 : (MODE == 1 ? 0 : 1));
        G8_WAIT_VM(8);
        G8P_SECTION(1, 0, Z{});
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using BT = std::true_type;
    using BF = std::false_type;

    // ---- prologue for the first tile: stage K-tile 0 whole and A_h0/B_h0 of K-tile 1
    int lt = l;
    int cur_tile = start + lt;
    set_tile(cur_tile);
    stage_a(0, 0, 0);
    stage_b(0, 0, 0);
    stage_b(1, 0, 0);
    stage_a(1, 0, 0);
    stage_a(0, 1, 1);
    stage_b(0, 1, 1);
    G8_WAIT_VM(4);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // team 1 runs one barrier behind team 0

    for (;;) {
        const int nlt = lt + gx;
        const bool has_next = nlt < cnt;
        next_tile = has_next ? start + nlt : cur_tile;  // no next tile: harmless re-prefetch of this one
        if constexpr (LNF || SIDE) {
            int mt, nt;
            tile_coords(cur_tile % P.TMN, &mt, &nt);
            const int m0 = mt * BMT, n0 = nt * BN;
            if constexpr (LNF) {
                const int rows_here = (P.M - m0 < BMT) ? P.M - m0 : BMT;
                Sr = __builtin_amdgcn_make_buffer_rsrc((void*)(P.ln_stats + (size_t)m0 * 2), 0, (unsigned)rows_here * 8u, 0x00020000);
            }
            if constexpr (SIDE) {
                const float* side = wave < 4 ? P.bias : (LNF ? P.ln_c : nullptr);
                const int cols_here = (P.N - n0 < BN) ? P.N - n0 : BN;
                // no bias: zero records, every lane out of range, the DMA writes zeros
                BLr = __builtin_amdgcn_make_buffer_rsrc((void*)(side ? side + n0 : P.ln_stats), 0, side ? (unsigned)cols_here * 4u : 0u, 0x00020000);
            }
        }
        ktile(I0{}, I0{}, BT{}, 0);
        ktile(I1{}, I0{}, BF{}, 1);
        for (int t = 2; t < nk - 2; t += 2) {
            ktile(I0{}, I0{}, BF{}, t);
            ktile(I1{}, I0{}, BF{}, t + 1);
        }
        ktile(I0{}, I1{}, BF{}, nk - 2);
        ktile(I1{}, I2{}, BF{}, nk - 1);

        // Both teams run the epilogue in the same barrier interval: team 0 waits one barrier
        // before it, team 1 one barrier after it (which also restores the stagger).
        if (wr == 0) __builtin_amdgcn_s_barrier();
        G8_WAIT_VM(4);  // retire B_h1/A_h1 of the next tile's K-tile 0 before any epilogue traffic
        {
            // lane (c = lane&15, g = lane>>4): acc[ha][mi][hb][ni][j] is output row
            // ha*128 + wr*64 + mi*16 + 4g + j, column wc*64 + 4c + (2hb + ni): 8-byte stores, 16 lanes
            // cover one full 128-byte line of a row. Stores (and residual loads) go through buffer
            // resources based at the tile's first row: rows >= M fall outside num_records and are
            // dropped by the hardware range check; lanes whose columns are >= N get an offset that
            // is always out of range. No branches.
            const int sp = cur_tile / P.TMN, t2 = cur_tile - sp * P.TMN;
            int mt, nt;
            tile_coords(t2, &mt, &nt);
            const int m0 = mt * BMT, n0 = nt * BN;
            const int nb = n0 + wc * 64 + 4 * i;
            const bool col_ok = nb < P.N;  // N % 8 == 0: 4 columns all in or all out
            const int rows_here = (P.M - m0 < BMT) ? P.M - m0 : BMT;
            if constexpr (EPI == EPI_PATCH || EPI == EPI_PATCH_DENSE) {
                // GEMM row g = b * np + patch goes to output row g + b + 1 (one CLS slot per image) and gets
                // position row 1 + patch. The tile's rows lie in at most 5 images (np >= 64, host check):
                // b - b0 = number of image starts e1, e1 + np, ... at or below the local row.
                const int b0 = m0 / P.np;
                const int e1 = (b0 + 1) * P.np - m0;
                const unsigned out_rows = (unsigned)(rows_here + (m0 + rows_here - 1) / P.np - b0);
                const __amdgpu_buffer_rsrc_t Cr = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(P.C + (size_t)(m0 + b0 + 1) * P.N), 0, out_rows * (unsigned)P.N * 2u, 0x00020000);
                const __amdgpu_buffer_rsrc_t Qr = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)P.pos, 0, (unsigned)(P.np + 1) * (unsigned)P.N * 4u, 0x00020000);
                float bias[4] = {0.f, 0.f, 0.f, 0.f};
                if (P.bias != nullptr && col_ok) {
                    const float4 b4 = *(const float4*)(P.bias + nb);
                    bias[0] = b4.x; bias[1] = b4.y; bias[2] = b4.z; bias[3] = b4.w;
                }
#pragma unroll
                for (int hm = 0; hm < 4; ++hm) {  // 64 of the lane's rows in 4 batches of 16 (ha = hm >> 1, mi = 2 (hm & 1) + ...)
                    const int ha = hm >> 1;
                    u32x4 pp[8];        // position rows of this batch, fetched up front into the dead fragment registers
                    unsigned vo[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const int lr = ha * 128 + wr * 64 + ((hm & 1) * 2 + (s >> 2)) * 16 + 4 * q + (s & 3);
                        const int extra = (lr >= e1) + (lr >= e1 + P.np) + (lr >= e1 + 2 * P.np) + (lr >= e1 + 3 * P.np);
                        const int pr = m0 + lr - (b0 + extra) * P.np + 1;
                        vo[s] = col_ok ? (unsigned)((lr + extra) * P.N + nb) * 2u : 0x80000000u;
                        pp[s] = __builtin_amdgcn_raw_buffer_load_b128(Qr, col_ok ? (int)((unsigned)(pr * P.N + nb) * 4u) : (int)0x80000000u, 0, 0);
                    }
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const int mi = (hm & 1) * 2 + (s >> 2), j = s & 3;
                        // (bit_cast of the whole vector: __builtin_bit_cast on a vector ELEMENT reads element 0)
                        const f32x4 pf = __builtin_bit_cast(f32x4, pp[s]);
                        bf16x4 ov;
#pragma unroll
                        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                            for (int ni = 0; ni < 2; ++ni)
                                ov[hb * 2 + ni] = (bf16_t)(acc[ha][mi][hb][ni][j] + bias[hb * 2 + ni] + pf[hb * 2 + ni]);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ov), Cr, (int)vo[s], 0, 0);
                    }
                }
            } else {
            constexpr bool USES_RES = (EPI == VIT_EPI_RESIDUAL || EPI == VIT_EPI_DGELU || EPI == VIT_EPI_DQUICK_GELU);
            constexpr unsigned OES = (OUTM == 1) ? 4u : 2u;  // output element size
            const unsigned span = (unsigned)rows_here * (unsigned)P.N * 2u;
            const __amdgpu_buffer_rsrc_t Cr = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((char*)P.C + ((size_t)sp * P.M + m0) * P.N * OES), 0, (unsigned)rows_here * (unsigned)P.N * OES,
                0x00020000);
            __amdgpu_buffer_rsrc_t Rr = Cr, Pr = Cr;
            if constexpr (USES_RES)
                Rr = __builtin_amdgcn_make_buffer_rsrc((void*)(P.res + (size_t)m0 * P.N), 0, span, 0x00020000);
            if constexpr (OUTM == 2)
                Pr = __builtin_amdgcn_make_buffer_rsrc((void*)(P.Cpre + (size_t)m0 * P.N), 0, span, 0x00020000);
            const unsigned voff0 = col_ok ? (unsigned)((wr * WRR + 4 * q) * P.N + nb) * 2u : 0x80000000u;
            // RST: this wave's 64-column group of the tile has its own slab of (sum, sum of squares) pairs; lane i == 0 of every
            // 16-lane row stores, the other lanes (and rows >= M) are out of range
            __amdgpu_buffer_rsrc_t Tr = Cr;
            unsigned toff0 = 0x80000000u;
            if constexpr (RST) {
                Tr = __builtin_amdgcn_make_buffer_rsrc((void*)(P.row_part + ((size_t)(nt * 4 + wc) * P.part_pitch + m0) * 2), 0,
                                                       (unsigned)rows_here * 8u, 0x00020000);
                if (i == 0 && col_ok) toff0 = (unsigned)((wr * WRR + 4 * q) * 8);
            }
            // bias, ln_c (column sums of the folded weight) and the tile's (mean * rstd, rstd) are in LDS behind the ring
            // (DMA of the last K-tile), read by inline asm (a C++ LDS load behind an LDS-DMA draws a vmcnt(0)). The
            // statistics are read two rows at a time, two reads ahead of their use (all 32 rows' pairs up front held 64
            // more VGPRs through the epilogue and cost 7-8 us per launch; one read ahead left the LDS latency exposed
            // sixteen times per tile, 17 us per launch of fc1 + GELU).
            const unsigned side_lds = (unsigned)(uintptr_t)LDS_PTR(smem) + (unsigned)(LDS_BYTES + 2048 + (wc * 64 + 4 * i) * 4);
            const unsigned st_lds = (unsigned)(uintptr_t)LDS_PTR(smem) + (unsigned)(LDS_BYTES + (wr * WRR + 4 * q) * 8);
            f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f}, lnc4 = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 spw[3];
            if constexpr (!SIDE) {
                if (P.bias != nullptr && col_ok) bias4 = *(const f32x4*)(P.bias + nb);
            } else {
                asm volatile("ds_read_b128 %0, %1" : "=v"(bias4) : "v"(side_lds) : "memory");
            }
            if constexpr (!SIDE) {
            } else if constexpr (LNF) {
                asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(lnc4) : "v"(side_lds) : "memory");
                st_read<0, MI>(spw, st_lds);
                st_read<1, MI>(spw, st_lds);
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bias4), "+v"(lnc4)::"memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bias4)::"memory");
            }
            const float bias[4] = {bias4[0], bias4[1], bias4[2], bias4[3]};
            const float lnc[4] = {lnc4[0], lnc4[1], lnc4[2], lnc4[3]};
            // residual / aux tile: the first RW of the lane's 8 MI row loads are issued up front into the (now dead) fragment
            // registers, load s + RW when row s has been consumed. (All 32 up front: 64 registers, and the spills they
            // forced were reloaded behind a vmcnt(0) at the top of every epilogue.)
            constexpr int NROW = 8 * MI;  // rows of a lane: (ha, mi, j)
            constexpr int RW = ((EPI == VIT_EPI_RESIDUAL) ? 6 : 4) * MI;
            u32x2 rres[NROW];
            auto res_load = [&](int s) {
                const unsigned voff = voff0 + (unsigned)(((s / (4 * MI)) * HR + ((s >> 2) % MI) * 16 + (s & 3)) * P.N) * 2u;
                return __builtin_amdgcn_raw_buffer_load_b64(Rr, (int)voff, 0, 0);  // (an nt hint here costs 0.7 % per forward: the rows were just written by the previous kernel)
            };
            if constexpr (USES_RES) {
#pragma unroll
                for (int s = 0; s < RW; ++s) rres[s] = res_load(s);
            }
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned voff = voff0 + (unsigned)((ha * HR + mi * 16 + j) * P.N) * 2u;
                        float v[4];
                        if constexpr (LNF) {
                            // two rows' (mean * rstd, rstd): read n, issued two reads ago
                            const int n = (ha * MI + mi) * 2 + (j >> 1);
                            if ((j & 1) == 0) {
                                if (n + 2 < 4 * MI) {
                                    // (the index must be a constant expression: spelled out per value of n)
                                    st_read_sw<MI>(n + 2, spw, st_lds);
                                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(spw[n % 3])::"memory");
                                } else if (n + 2 == 4 * MI) {
                                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(spw[n % 3])::"memory");
                                } else {
                                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(spw[n % 3])::"memory");
                                }
                            }
                            const f32x4 sp = spw[n % 3];
                            const float mr = (j & 1) ? sp[2] : sp[0], rs = (j & 1) ? sp[3] : sp[1];
                            // two packed fmas per column pair (same bits as the fmaf chain of the generic kernel)
#pragma unroll
                            for (int hb = 0; hb < 2; ++hb) {
                                const f32x2 t = __builtin_elementwise_fma(f32x2{-mr, -mr}, f32x2{lnc[hb * 2], lnc[hb * 2 + 1]},
                                                                          f32x2{bias[hb * 2], bias[hb * 2 + 1]});
                                const f32x2 r2 = __builtin_elementwise_fma(f32x2{rs, rs}, f32x2{acc[ha][mi][hb][0][j], acc[ha][mi][hb][1][j]}, t);
                                v[hb * 2] = r2.x;
                                v[hb * 2 + 1] = r2.y;
                            }
                        } else {
#pragma unroll
                            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                                for (int ni = 0; ni < 2; ++ni) v[hb * 2 + ni] = acc[ha][mi][hb][ni][j] + bias[hb * 2 + ni];
                        }
                        if constexpr (OUTM == 2) {
                            bf16x4 pv;
#pragma unroll
                            for (int e = 0; e < 4; ++e) pv[e] = (bf16_t)v[e];
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pv), Pr, (int)voff, 0, G8_STORE_AUX);
                        }
                        if constexpr (EPI == VIT_EPI_GELU) {
                            f32x2 g0 = f32x2{v[0], v[1]}, g1 = f32x2{v[2], v[3]};
                            act_gelu_pk2(g0, g1);
                            v[0] = g0.x; v[1] = g0.y; v[2] = g1.x; v[3] = g1.y;
                        }
                        if constexpr (EPI == VIT_EPI_QUICK_GELU) {
                            const f32x2 g0 = act_quick_gelu_pk(f32x2{v[0], v[1]}), g1 = act_quick_gelu_pk(f32x2{v[2], v[3]});
                            v[0] = g0.x; v[1] = g0.y; v[2] = g1.x; v[3] = g1.y;
                        }
                        if constexpr (USES_RES) {
                            const bf16x4 r = __builtin_bit_cast(bf16x4, rres[(ha * MI + mi) * 4 + j]);
                            if constexpr (EPI == VIT_EPI_DGELU) {
                                const f32x2 g0 = act_dgelu_pk(f32x2{(float)r[0], (float)r[1]}), g1 = act_dgelu_pk(f32x2{(float)r[2], (float)r[3]});
                                v[0] *= g0.x; v[1] *= g0.y; v[2] *= g1.x; v[3] *= g1.y;
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if constexpr (EPI == VIT_EPI_RESIDUAL) v[e] += (float)r[e];
                                    else v[e] *= act_grad<EPI>((float)r[e]);
                                }
                            }
                        }
                        if constexpr (OUTM == 1) {
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                                                                         __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])},
                                                                   Cr, (int)(voff * 2u), 0, G8_STORE_AUX);
                        } else {
                            bf16x4 ov;
#pragma unroll
                            for (int e = 0; e < 4; ++e) ov[e] = (bf16_t)v[e];
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ov), Cr, (int)voff, 0, G8_STORE_AUX);
                            if constexpr (RST) {
                                // sums of the STORED values (what the next LayerNorm reads)
                                const u32x2 du = __builtin_bit_cast(u32x2, ov);
                                float sm, sq;
                                row16_stats(du.x, du.y, sm, sq);
                                __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, sm), __builtin_bit_cast(unsigned, sq)}, Tr,
                                                                      (int)(toff0 + (unsigned)((ha * HR + mi * 16 + j) * 8)), 0, 0);
                            }
                        }
                        if constexpr (USES_RES) {
                            const int sn = (ha * MI + mi) * 4 + j + RW;
                            if (sn < NROW) {
                                __builtin_amdgcn_sched_barrier(0);  // not hoisted: the window is what keeps the registers down
                                rres[sn] = res_load(sn);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
            }  // !EPI_PATCH
        }
        if (wr == 1) __builtin_amdgcn_s_barrier();
        if (!has_next) break;
        lt = nlt;
        cur_tile = next_tile;
    }
    G8_WAIT_VM(0);  // the dummy prefetch of the last tile must land before the LDS is released
    if (wr == 0) __builtin_amdgcn_s_barrier();  // balance team 1's extra barrier
}

// Column-group width of the tile order (PParams::GC). Lab switch VIT_AMD_GEMM_GC: 0 = plain order, n = groups of n.
inline int group_cols(int NT, int K) {
    static const int forced = getenv("VIT_AMD_GEMM_GC") ? atoi(getenv("VIT_AMD_GEMM_GC")) : -1;
    if (forced == 0) return NT;
    if (forced > 0) return forced < NT ? forced : NT;
    if (NT <= 6) return NT;
    const int groups = (NT + 5) / 6;  // about six columns each: 32 workgroups of an XCD cover ~5.3 x 6 tiles, the squarest cut
    return (NT + groups - 1) / groups;
}

template <int EPI, int OUTM, bool LNF = false, int MI = 4, bool RST = false>
int launch_persist(const PParams& P, int ncu, hipStream_t stream) {
    auto kern = gemm8p_persist_kernel<EPI, OUTM, LNF, MI, RST>;
    static PerDeviceInt attr_set;  // per instantiation and device
    std::atomic<int>* as = attr_set.slot();
    if (!as || !as->load(std::memory_order_relaxed)) {
        VIT_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 4096));
        if (as) as->store(1, std::memory_order_relaxed);
    }
    const int G = P.T < ncu ? P.T : ncu;
    hipLaunchKernelGGL(kern, dim3(G), dim3(NTHREADS), LDS_BYTES + 4096, stream, P);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

// ---- few-tile planner. A launch of T tiles on ncu CUs takes ceil(T / ncu) tile ROUNDS however few tiles the last
// round has; a small batch (the per-GPU shard of a strong-scaled batch: CLIP-L/14-336 at 16 images is 148 / 444 / 592
// tiles of 256 x 256 on 256 CUs) loses up to 40 % of a GEMM that way. The kernel exists in three tile heights (64 MI
// rows, same bits), and a row range can be cut in two launches (rows are independent), so the plan is: either one launch
// at the best height, or a head of whole rounds of 256-row tiles plus a tail at its own best height. Cost model in
// microseconds, fitted to tools/time_fewtile.py on MI355X (round 3): a K-tile takes 0.36 + 0.22 MI (the load section of a
// phase does not shrink with the tile), the epilogue 0.8 MI, a launch 4. A pure function of (M, N, K, CUs): a given
// shape always takes the same route; the bits do not depend on the route anyway (tests/test_gpu_switches.py).
struct RowPlan {
    int nseg;
    int rows[2], mi[2];
};
inline double tile_us(int mi, int nk) { return nk * (0.36 + 0.22 * mi) + 0.8 * mi; }
inline double single_us(int M, int NT, int nk, int ncu, int mi) {
    const long tiles = (long)ceil_div(M, 64 * mi) * NT;
    return 4.0 + (double)((tiles + ncu - 1) / ncu) * tile_us(mi, nk);
}
// Inside a two-chunk vit_forward (GemmExt::concurrent) every launch keeps 256-row tiles: there the other stream's kernels
// fill the CUs a partial round leaves idle -- shorter tiles (less work per operand byte) then cost more than they save (same
// box, planner on / off inside the chunks: ViT-B/16 batch 256 8.70 / 8.48 ms, CLIP-L/14 batch 128 43.2 / 42.6, ViT-H/14 batch 64
// 82.4 / 80.9). A launch that has the chip to itself gains at every size: single-stream sizes CLIP-L/14 batch 16 6.43 / 6.81 ms,
// ViT-B/16 batch 32 1.87 / 2.14, ViT-H/14 batch 8 12.06 / 12.27; full-batch launches (M = 50,432: a tail of 128- or 192-row
// tiles behind whole rounds) o_proj + residual 71 / 74 us, fc2 + residual 208 / 216, VIT_AMD_STREAMS=1 forward 8.86 / 8.92 ms.
inline RowPlan plan_rows(int M, int N, int K, int ncu, bool allow) {
    // lab switch VIT_AMD_GEMM_MI: 4 = 256-row tiles only (the kernel of rounds 1-2), 3 / 2 = that height everywhere
    static const int forced = getenv("VIT_AMD_GEMM_MI") ? atoi(getenv("VIT_AMD_GEMM_MI")) : 0;
    RowPlan best{1, {M, 0}, {4, 0}};
    if (forced >= 2 && forced <= 4) {
        best.mi[0] = forced;
        return best;
    }
    if (!allow) return best;
    const int NT = ceil_div(N, BN), nk = K / BK;
    double best_us = single_us(M, NT, nk, ncu, 4);
    auto best_single = [&](int rows, int* mi_out) {
        double b = single_us(rows, NT, nk, ncu, 4);
        *mi_out = 4;
        for (int mi = 3; mi >= 2; --mi) {
            const double c = single_us(rows, NT, nk, ncu, mi);
            if (c < b * 0.97) { b = c; *mi_out = mi; }
        }
        return b;
    };
    int mi1;
    const double one = best_single(M, &mi1);
    if (one < best_us) { best_us = one; best.mi[0] = mi1; }
    // head: r whole rounds of 256-row tiles (as many tile rows as fit r * ncu tiles), tail: the rest
    for (int r = 1; r <= 8; ++r) {
        const int head_rows = (int)((long)r * ncu / NT) * BM;
        if (head_rows <= 0) continue;
        if (head_rows >= M) break;
        int mit;
        const double c = single_us(head_rows, NT, nk, ncu, 4) + best_single(M - head_rows, &mit);
        if (c < best_us * 0.97) {
            best_us = c;
            best = RowPlan{2, {head_rows, M - head_rows}, {4, mit}};
        }
    }
    return best;
}

// inference-forward epilogues exist in all three tile heights; everything else is 256-row only.
// Runs the row plan: segment s covers rows [row0, row0 + rows) of A, C, the residual and the LayerNorm statistics.
template <int EPI, bool LNF, bool RST = false>
int launch_persist_planned(const PParams& P0, int ncu, bool allow, hipStream_t stream) {
    const RowPlan plan = plan_rows(P0.M, P0.N, P0.K, ncu, allow);
    int row0 = 0;
    for (int sgm = 0; sgm < plan.nseg; ++sgm) {
        PParams P = P0;
        const int mi = plan.mi[sgm];
        P.M = plan.rows[sgm];
        P.A = P0.A + (size_t)row0 * P0.lda;
        P.C = P0.C + (size_t)row0 * P0.N;
        if (P0.res) P.res = P0.res + (size_t)row0 * P0.N;
        if (P0.ln_stats) P.ln_stats = P0.ln_stats + (size_t)row0 * 2;
        if (P0.row_part) P.row_part = P0.row_part + (size_t)row0 * 2;
        P.MT = ceil_div(P.M, 64 * mi);
        P.TMN = P.MT * P.NT;
        P.T = P.TMN;
        int rc;
        if (mi == 3) rc = launch_persist<EPI, 0, LNF, 3, RST>(P, ncu, stream);
        else if (mi == 2) rc = launch_persist<EPI, 0, LNF, 2, RST>(P, ncu, stream);
        else rc = launch_persist<EPI, 0, LNF, 4, RST>(P, ncu, stream);
        if (rc != VIT_OK) return rc;
        row0 += plan.rows[sgm];
    }
    return VIT_OK;
}

}  // namespace g8

int launch_reduce_slabs(const float* slabs, float* out, int splits, int64_t n, hipStream_t stream);

// true when the 8-phase kernel takes this problem
bool gemm_bf16_8p_eligible(int M, int N, int K) {
    return K % 128 == 0 && K >= 256 && N % 8 == 0 && M >= 1 && (size_t)256 * K * 2 < ((size_t)1 << 31);  // K here is the full row pitch
}

// Patch sizes the persistent kernel's A loader takes: a 64-element K-tile must be whole pixel rows of one
// channel (P divides 64, P*P a multiple of 64 -> P = 8, 16, 32, 64), and K = C*P*P a legal K.
bool gemm_bf16_8p_patch_eligible(int M, int N, const PatchGeom* g) {
    const int P = g->patch, K = g->channels * P * P;
    return P >= 8 && 64 % P == 0 && (P * P) % 64 == 0 && g->image % 8 == 0 && g->image == g->grid * P && g->kreal == K &&
           g->grid * g->grid >= 64 && M % (g->grid * g->grid) == 0 && gemm_bf16_8p_eligible(M, N, K) &&
           (size_t)g->channels * g->image * g->image * 2 * 7 < ((size_t)1 << 32) && (size_t)(g->grid * g->grid + 1) * N * 4 < ((size_t)1 << 32);
}

int launch_gemm_bf16_8p(const void* A, const void* W, const float* bias, const void* res, void* C, int M, int N, int K,
                        int epi, const GemmExt* ext, hipStream_t stream) {
    VIT_REQUIRE(gemm_bf16_8p_eligible(M, N, K), "gemm8p: shape M=%d N=%d K=%d not eligible", M, N, K);
    int ncu = 0;
    VIT_TRY_RC(vit_cu_count(&ncu));
    g8::PParams P{};
    P.A = (const bf16_t*)A; P.W = (const bf16_t*)W; P.bias = bias; P.res = (const bf16_t*)res; P.C = (bf16_t*)C;
    P.Cpre = ext ? (bf16_t*)ext->c_pre : nullptr;
    const int splits = (ext && ext->split_k > 1) ? ext->split_k : 1;
    VIT_REQUIRE(K % (128 * splits) == 0 && K / splits >= 256, "gemm8p: K=%d not divisible into %d splits of whole K-tile pairs", K, splits);
    VIT_REQUIRE(splits == 1 || (ext->out_f32 && ext->slabs != nullptr), "gemm8p: split-K needs fp32 output and a slab workspace");
    P.M = M; P.N = N; P.K = K / splits; P.lda = K;
    P.NT = ceil_div(N, g8::BN);
    P.MT = ceil_div(M, g8::BM);
    P.GC = g8::group_cols(P.NT, K / splits);
    P.TMN = P.MT * P.NT;
    P.T = P.TMN * splits;
    if (splits > 1) P.C = (bf16_t*)ext->slabs;
    const bool f32out = ext && ext->out_f32;
    const bool pre = P.Cpre != nullptr;
    const bool few_ok = !(ext && ext->concurrent);
    VIT_REQUIRE(!(f32out && (epi != VIT_EPI_BIAS || pre)), "gemm8p: fp32 output only with the plain epilogue");
    VIT_REQUIRE(!pre || epi == VIT_EPI_GELU || epi == VIT_EPI_QUICK_GELU, "gemm8p: pre-activation output needs an activation epilogue");
    VIT_REQUIRE((size_t)256 * N * (f32out ? 4 : 2) < ((size_t)1 << 31), "gemm8p: N too large");
    if (ext && ext->ln_stats) {
        VIT_REQUIRE(ext->ln_colsum != nullptr && !f32out && !pre && splits == 1, "gemm8p: folded LayerNorm takes the plain bf16 epilogues only");
        P.ln_stats = ext->ln_stats; P.ln_c = ext->ln_colsum;
        switch (epi) {
            case VIT_EPI_BIAS: return g8::launch_persist_planned<VIT_EPI_BIAS, true>(P, ncu, few_ok, stream);
            case VIT_EPI_GELU: return g8::launch_persist_planned<VIT_EPI_GELU, true>(P, ncu, few_ok, stream);
            case VIT_EPI_QUICK_GELU: return g8::launch_persist_planned<VIT_EPI_QUICK_GELU, true>(P, ncu, few_ok, stream);
        }
        return vit_set_error(VIT_E_INVALID, "gemm8p: folded LayerNorm with epilogue %d", epi);
    }
    if (f32out) {
        int rc = g8::launch_persist<VIT_EPI_BIAS, 1>(P, ncu, stream);
        if (rc != VIT_OK || splits == 1) return rc;
        return launch_reduce_slabs((const float*)ext->slabs, (float*)C, splits, (int64_t)M * N, stream);
    }
    switch (epi) {
        case VIT_EPI_BIAS: return g8::launch_persist<VIT_EPI_BIAS, 0>(P, ncu, stream);
        case VIT_EPI_GELU: return pre ? g8::launch_persist<VIT_EPI_GELU, 2>(P, ncu, stream) : g8::launch_persist<VIT_EPI_GELU, 0>(P, ncu, stream);
        case VIT_EPI_QUICK_GELU: return pre ? g8::launch_persist<VIT_EPI_QUICK_GELU, 2>(P, ncu, stream) : g8::launch_persist<VIT_EPI_QUICK_GELU, 0>(P, ncu, stream);
        case VIT_EPI_RESIDUAL:
            if (ext && ext->row_part) {
                VIT_REQUIRE(N % 64 == 0, "gemm8p: row statistics need N %% 64 == 0 (N=%d)", N);
                P.row_part = ext->row_part; P.part_pitch = M;
                return g8::launch_persist_planned<VIT_EPI_RESIDUAL, false, true>(P, ncu, few_ok, stream);
            }
            return g8::launch_persist_planned<VIT_EPI_RESIDUAL, false>(P, ncu, few_ok, stream);
        case VIT_EPI_DGELU: return g8::launch_persist<VIT_EPI_DGELU, 0>(P, ncu, stream);
        case VIT_EPI_DQUICK_GELU: return g8::launch_persist<VIT_EPI_DQUICK_GELU, 0>(P, ncu, stream);
    }
    return vit_set_error(VIT_E_INVALID, "gemm8p: unknown epilogue %d", epi);
}

// Patch embedding on the persistent kernel: pixels bf16 [B, C, image, image] -> C rows b*(np+1) + 1 + patch
// (the CLS rows are written by the caller), + bias + position rows 1..np.
int launch_gemm_bf16_8p_patch(const void* pixels, const void* W, const float* bias, const float* pos, void* C, int M, int N,
                              const PatchGeom* g, hipStream_t stream) {
    VIT_REQUIRE(gemm_bf16_8p_patch_eligible(M, N, g), "gemm8p: patch geometry not eligible");
    int ncu = 0;
    VIT_TRY_RC(vit_cu_count(&ncu));
    const int P = g->patch, K = g->channels * P * P;
    g8::PParams q{};
    q.A = (const bf16_t*)pixels; q.W = (const bf16_t*)W; q.bias = bias; q.C = (bf16_t*)C; q.pos = pos;
    q.M = M; q.N = N; q.K = K; q.lda = K;
    q.NT = ceil_div(N, g8::BN);
    q.MT = ceil_div(M, g8::BM);
    q.GC = q.NT;
    q.TMN = q.MT * q.NT;
    q.T = q.TMN;
    q.np = g->grid * g->grid; q.grid = g->grid; q.patch = P; q.image = g->image;
    q.img_bytes = (unsigned)g->channels * g->image * g->image * 2u;
    const int kpc = P * P / 64;  // K-tiles per channel: a power of two for P = 8..64
    int sh = 0;
    while ((1 << sh) < kpc) ++sh;
    q.a_kshift = sh; q.a_kmask = kpc - 1;
    q.a_chs = (unsigned)g->image * g->image * 2u;          // next channel
    q.a_kts = (unsigned)(64 / P) * g->image * 2u;          // next 64 / P pixel rows
    return g8::launch_persist<g8::EPI_PATCH, 0>(q, ncu, stream);
}

// The same epilogue on an already gathered patch matrix A [M, K] (bf16, K a legal K of this kernel, zero
// padded): the route for patch sizes whose pixel rows are not 16-byte pieces (P = 14).
bool gemm_bf16_8p_patch_dense_eligible(int M, int N, int K, const PatchGeom* g) {
    return gemm_bf16_8p_eligible(M, N, K) && g->grid * g->grid >= 64 && M % (g->grid * g->grid) == 0 &&
           (size_t)(g->grid * g->grid + 1) * N * 4 < ((size_t)1 << 32) && (long)M * N >= 64L * 256 * 256;
}
int launch_gemm_bf16_8p_patch_dense(const void* A, const void* W, const float* bias, const float* pos, void* C, int M, int N,
                                    int K, const PatchGeom* g, hipStream_t stream) {
    VIT_REQUIRE(gemm_bf16_8p_patch_dense_eligible(M, N, K, g), "gemm8p: dense patch problem not eligible");
    int ncu = 0;
    VIT_TRY_RC(vit_cu_count(&ncu));
    g8::PParams q{};
    q.A = (const bf16_t*)A; q.W = (const bf16_t*)W; q.bias = bias; q.C = (bf16_t*)C; q.pos = pos;
    q.M = M; q.N = N; q.K = K; q.lda = K;
    q.NT = ceil_div(N, g8::BN);
    q.MT = ceil_div(M, g8::BM);
    q.GC = q.NT;
    q.TMN = q.MT * q.NT;
    q.T = q.TMN;
    q.np = g->grid * g->grid; q.grid = g->grid; q.patch = g->patch; q.image = g->image;
    return g8::launch_persist<g8::EPI_PATCH_DENSE, 0>(q, ncu, stream);
}
