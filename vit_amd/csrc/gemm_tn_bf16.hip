// Weight-gradient GEMM without transposes:  dW[m, n] = sum_t dY[t, m] * X[t, n]   (fp32 out)
//
// Backward of nn.Linear (modeling_vit.py:216-218, 236, 250-252) with respect to the weight; the
// reduction runs over TOKENS, and both operands are token-major ([T, features], exactly as the
// backward produces dY and the forward stashed X). The persistent GEMM of gemm_bf16_8p.hip wants
// both operands K-contiguous, which cost two transposes per weight gradient (15 % of the training
// step); this kernel reads the token-major tiles as they are and transposes inside the LDS read:
//
//  * tile 256 (m) x 256 (n), 8 waves (2 x 4), wave tile 128 x 64, one workgroup per CU, an LDS ring of
//    four slots of [32 tokens][256 features] per operand, fed by buffer-resource LDS-DMA three slots
//    ahead (rows past T are outside the resource and read as zero: no padded copies), one counted
//    vmcnt wait per slot; a slot is a read section and an MFMA section with a barrier after each, the two
//    wave rows one barrier apart (one wave's products cover the other's LDS reads on every SIMD).
//  * MFMA fragments (16 features x 32 tokens, 8 consecutive tokens per lane) come out of the
//    token-major tile through ds_read_b64_tr_b16: a 16-lane group loads 4 token rows x 16 features
//    and each lane receives one feature's 4 tokens. 8 token rows x 32 bytes are read per LDS cycle,
//    so the 32-byte feature block index is XOR-swizzled with f(t) = (t & 3) | ((t >> 3) & 1) << 2 --
//    applied on the DMA's SOURCE address -- which makes those 8 segments cover all 64 banks.
//  * split-K over tokens (few output tiles: 9 .. 144), one fp32 slab per split, summed in a fixed
//    order by reduce_slabs (reproducible); the bias gradient (column sums of dY) rides along as one
//    extra MFMA against a fragment of ones, dealt out over n-tiles and wave columns, and is added to
//    db with one fp32 atomic per feature and split (as the transpose kernel did before).
#include "vit_common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int SLOT_BYTES = 32768;            // 32 tokens x 256 features of A (16 KB) then of B (16 KB)
constexpr int NSLOT = 4;                     // ring depth (the DMA runs NSLOT - 1 slots ahead; 5 slots = all 160 KiB measured the same)
constexpr int LDS_BYTES = NSLOT * SLOT_BYTES;

struct TnParams {
    const bf16_t* A;   // dY [T, Mo]
    const bf16_t* B;   // X  [T, No]
    float* slabs;      // [splits][Mo][No]
    float* dbias;      // [Mo] or null (accumulated)
    int T, Mo, No;
    int tiles_m, tiles_n;
    int tok_per_split;  // multiple of 64
};

__device__ __forceinline__ int swz(int t) { return (t & 3) | (((t >> 3) & 1) << 2); }

// two transposed reads (token rows +0..3 and +4..7) of one fragment; no wait
#define TR2(lo, hi, addr, off)                                                                        \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"         \
                 : "=&v"(lo), "=&v"(hi)                                                               \
                 : "v"(addr), "n"(off), "n"((off) + 2048)                                             \
                 : "memory")

__global__ __launch_bounds__(512) void gemm_tn_bf16_kernel(const TnParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned smem_lds = (unsigned)(uintptr_t)LDS_PTR(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int q = lane >> 4, i16 = lane & 15;

    // XCD-aware bijective remap (blocks b and b+8 share an XCD): an XCD runs a contiguous range of work
    // ids = the tiles of one split, which share their token rows of dY and X in that XCD's L2
    int wid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
        wid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    const int tiles = P.tiles_m * P.tiles_n;
    const int split = wid / tiles, tile = wid - split * tiles;
    const int tm = tile / P.tiles_n, tn = tile - tm * P.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int t_begin = split * P.tok_per_split;

    // ---- DMA staging (16-byte chunk lane & 31 of a token row)
    const __amdgpu_buffer_rsrc_t Ar =
        __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (unsigned)P.T * (unsigned)P.Mo * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t Br =
        __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, (unsigned)P.T * (unsigned)P.No * 2u, 0x00020000);
    // four slots of 32 tokens ([A 16 KB][B 16 KB]); wave w, piece j covers token rows 2*(2w + j) + (lane >> 5)
    unsigned ra_src[2], rb_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 2 * (2 * wave + j) + (lane >> 5);
        const int p = lane & 31;
        const int c = ((((p >> 1) ^ swz(r)) << 1) | (p & 1)) * 8;
        ra_src[j] = (m0 + c < P.Mo) ? (unsigned)(r * P.Mo + m0 + c) * 2u : 0x80000000u;
        rb_src[j] = (n0 + c < P.No) ? (unsigned)(r * P.No + n0 + c) * 2u : 0x80000000u;
    }
    const int ns = P.tok_per_split / 32;
    auto stage_slot = [&](int s) {  // slots past the split's end are fetched out of range (zeros, no traffic): vmcnt stays uniform
        unsigned char* d = smem + (s & (NSLOT - 1)) * SLOT_BYTES + wave * 2048;
        const int tok = t_begin + s * 32;
        const unsigned kill = s < ns ? 0u : 0x80000000u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Ar, LDS_PTR(d + j * 1024), 16, (int)(ra_src[j] | kill), tok * P.Mo * 2, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(Br, LDS_PTR(d + SLOT_BYTES / 2 + j * 1024), 16, (int)(rb_src[j] | kill),
                                                     tok * P.No * 2, 0, 0);
        }
    };

    // ---- fragment read addresses (LDS byte addresses, buffer 0, k-step 0, first half) ----
    const int f = (i16 >> 2) | ((q & 1) << 2);  // = swz(token row) of every row this lane reads
    const unsigned rowpart = (unsigned)((8 * q + (i16 >> 2)) * 512 + 8 * (i16 & 3));
    unsigned a_off[8], b_off[4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) a_off[mi] = smem_lds + rowpart + (unsigned)(((wm * 8 + mi) ^ f) * 32);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
        b_off[ni] = smem_lds + SLOT_BYTES / 2 + rowpart + (unsigned)(((wn * 4 + ni) ^ f) * 32);

    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Bias gradient: fragment row mi of this wave row belongs to ONE (n-tile, wave column) pair -- every n-tile sees the
    // same dY rows, so the eight extra MFMAs per k-step are dealt out over n-tiles and wave columns (all eight in the
    // wn = 0 waves of the first n-tile made those waves, and with them the whole launch, 25 % longer).
    const int tn_mod = P.tiles_n < 8 ? P.tiles_n : 8;
    unsigned bmask = 0;  // wave-uniform
    if (P.dbias != nullptr) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
            if (tn == mi % tn_mod && wn == ((mi / tn_mod) & 3)) bmask |= 1u << mi;
    }
    f32x4 bacc[8];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) bacc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

    // One slot = a READ section (the 24 transposed fragment reads of the slot, the DMA of the slot three ahead, the
    // counted wait that retires this wave's pieces of the NEXT slot, lgkmcnt(0)), a barrier, an MFMA section (32 + bias
    // products), a barrier. The two wave rows (teams wm = 0 / 1, one wave of each per SIMD) run one barrier apart, so on
    // every SIMD one wave's products cover the other's LDS latency. Hazards, in barrier intervals (team 0 reads slot s in
    // interval 2s, team 1 in 2s + 1): slot s + 1 is read from interval 2s + 2 on and every wave has waited for its pieces
    // by the end of interval 2s + 1; slot s + 3 overwrites the buffer of slot s - 1 from interval 2s on, whose last reads
    // (team 1, interval 2s - 1) were waited for (lgkmcnt(0)) before that interval's closing barrier.
    short4v bl[4], bh[4], al[8], ah[8];
    auto read_slot = [&](const unsigned bo) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            TR2(bl[ni], bh[ni], b_off[ni] + bo, 0);
        }
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            TR2(al[mi], ah[mi], a_off[mi] + bo, 0);
        }
    };
    auto mma_slot = [&]() {
        bf16x8 bfr[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
            bfr[ni] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(bl[ni], bh[ni], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const bf16x8 afr = __builtin_bit_cast(bf16x8, __builtin_shufflevector(al[mi], ah[mi], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], afr, acc[mi][ni], 0, 0, 0);
            if (bmask & (1u << mi)) bacc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, afr, bacc[mi], 0, 0, 0);
        }
    };
#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s) stage_slot(s);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (NSLOT - 2)) : "memory");  // slot 0 landed everywhere
    if (wm == 1) asm volatile("s_barrier" ::: "memory");            // team 1 runs one barrier behind team 0
    for (int s = 0; s < ns; ++s) {
        read_slot((unsigned)((s & (NSLOT - 1)) * SLOT_BYTES));
        stage_slot(s + NSLOT - 1);
        asm volatile("s_waitcnt vmcnt(%8)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier"
                     : "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]), "+v"(bl[2]), "+v"(bh[2]), "+v"(bl[3]), "+v"(bh[3])
                     : "n"(4 * (NSLOT - 2)) : "memory");
        asm volatile("" : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(al[2]), "+v"(ah[2]), "+v"(al[3]), "+v"(ah[3]));
        asm volatile("" : "+v"(al[4]), "+v"(ah[4]), "+v"(al[5]), "+v"(ah[5]), "+v"(al[6]), "+v"(ah[6]), "+v"(al[7]), "+v"(ah[7]));
        __builtin_amdgcn_sched_barrier(0);
        mma_slot();  // (s_setprio around the products, the DMA issued before the reads: both within noise)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
    }
    if (wm == 0) asm volatile("s_barrier" ::: "memory");  // balance team 1's extra barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the out-of-range tail DMAs still target this LDS

    // ---- epilogue: D = (X^T-fragment rows n) x (dY^T-fragment cols m): lane holds m = lane & 15 and the
    // four consecutive n = 4 * (lane >> 4) + e  ->  one 16-byte store per fragment
    float* slab = P.slabs + (size_t)split * P.Mo * P.No;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int m = m0 + wm * 128 + mi * 16 + i16;
        if (m < P.Mo) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = n0 + wn * 64 + ni * 16 + 4 * q;
                if (n < P.No) *(f32x4*)(slab + (size_t)m * P.No + n) = acc[mi][ni];
            }
        }
    }
    if (bmask != 0 && q == 0) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = m0 + wm * 128 + mi * 16 + i16;
            if ((bmask & (1u << mi)) && m < P.Mo) atomicAdd(P.dbias + m, bacc[mi][0]);
        }
    }
}

}  // namespace

int launch_reduce_slabs(const float* slabs, float* out, int splits, int64_t n, hipStream_t stream);

// dW (fp32 [Mo, No]) = dY^T X over T tokens; db (fp32 [Mo], may be null) += column sums of dY.
// slabs: splits * Mo * No floats of scratch when splits > 1.
int launch_gemm_tn_bf16(const void* dY, const void* X, float* dW, float* db, float* slabs, int T, int Mo, int No,
                        int splits, hipStream_t stream) {
    VIT_REQUIRE(T > 0 && Mo > 0 && No > 0 && splits >= 1, "gemm_tn: bad shape T=%d Mo=%d No=%d splits=%d", T, Mo, No, splits);
    VIT_REQUIRE(Mo % 8 == 0 && No % 8 == 0, "gemm_tn: feature counts must be multiples of 8 (Mo=%d No=%d)", Mo, No);
    VIT_REQUIRE((long)T * (Mo > No ? Mo : No) * 2 < (1L << 31), "gemm_tn: an operand exceeds 2 GiB");
    VIT_REQUIRE(splits == 1 || slabs != nullptr, "gemm_tn: split-K needs the slab scratch");
    static PerDeviceInt attr_set;  // per device
    std::atomic<int>* as = attr_set.slot();
    if (!as || !as->load(std::memory_order_relaxed)) {
        VIT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        if (as) as->store(1, std::memory_order_relaxed);
    }
    TnParams P{};
    P.A = (const bf16_t*)dY; P.B = (const bf16_t*)X; P.dbias = db;
    P.slabs = splits > 1 ? slabs : dW;
    P.T = T; P.Mo = Mo; P.No = No;
    P.tiles_m = ceil_div(Mo, BM); P.tiles_n = ceil_div(No, BN);
    P.tok_per_split = ceil_div(ceil_div(T, splits), BK) * BK;
    hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3(P.tiles_m * P.tiles_n * splits), dim3(512), LDS_BYTES, stream, P);
    VIT_CHECK_HIP(hipGetLastError());
    if (splits > 1) return launch_reduce_slabs(slabs, dW, splits, (int64_t)Mo * No, stream);
    return VIT_OK;
}
