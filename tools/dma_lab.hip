// DMA-delivery lab (development only): how fast can a CU pull GEMM operand K-tiles from L2 into LDS with
// buffer_load ... lds, as a function of how much is kept in flight? No consumers: every K-tile lands on the same
// 64 KiB of LDS (WAW only, harmless for timing). Same addresses as the persistent GEMM: a workgroup walks 256 x 256
// output tiles (XCD-chunked list), per K-tile (64 elements) it fetches 256 A rows x 128 B and 256 W rows x 128 B.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/dma_lab.hip -o tools/dma_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// (the LDS-DMA builtin must sit in a __device__ function: used directly in a __global__ template it silently
// suppresses the host-side instantiation of the kernel)
__device__ __forceinline__ void bufdma16(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, int uni_off, unsigned char* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds), 16, (int)lane_off, uni_off, 0, 0);
}

struct DmaParams {
    const unsigned short* A;
    const unsigned short* W;
    int M, N, K, NT, T;
    unsigned long long* stamps;
};

// INFL = 1-KiB DMA instructions a wave keeps in flight (a counted wait after every issue), so INFL * NW KiB per CU;
// NW waves per workgroup. RB: bytes of a row fetched per K-step (128 = BK 64, 64 = BK 32).
// NRD: ds_read_b128 per wave and K-tile issued beside the DMA (conflict-free swizzled addresses, results discarded);
// NBAR: workgroup barriers per K-tile
template <int NW, int INFL, int RB, int NRD = 0, int NBAR = 0>
__global__ __launch_bounds__(NW * 64, 1) void dma_kernel(DmaParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int G = gridDim.x, bid = blockIdx.x;
    const int XL = G < 8 ? G : 8;
    const int x = bid % XL, l = bid / XL;
    const int gx = (G - x + XL - 1) / XL;
    const int tq = p.T / XL, tr = p.T % XL;
    const int cnt = tq + (x < tr ? 1 : 0);
    const int start = x * tq + (x < tr ? x : tr);
    if (l >= cnt) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int IPW = 64 / NW;            // 1-KiB instructions per wave and 64-wide K-tile (A and W together)
    constexpr int LPR = RB / 16;            // lanes per row
    constexpr int RPI = 64 / LPR;           // rows per instruction
    constexpr int KSTEPS = 128 / RB;        // K-steps per 64-wide K-tile
    constexpr int IPS = IPW / KSTEPS;       // instructions per wave and K-step
    unsigned char* const dst = smem + wave * (IPW * 1024);
    const int nk = p.K * 2 / RB;
    unsigned long long t0 = 0, t1 = 0;
    int done = 0;
    for (int lt = l; lt < cnt; lt += gx) {
        const int tile = start + lt;
        const int mt = tile / p.NT, nt = tile - mt * p.NT;
        // instruction j of this wave (per K-step): rows (wave * IPS + j) * RPI + lane / LPR of the 512-row (A then W) panel pair
        unsigned off[IPS];
        bool isws[IPS];
        const __amdgpu_buffer_rsrc_t Ar = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (size_t)mt * 256 * p.K), 0, 256u * (unsigned)p.K * 2u, 0x00020000);
        const __amdgpu_buffer_rsrc_t Wr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (size_t)nt * 256 * p.K), 0, 256u * (unsigned)p.K * 2u, 0x00020000);
#pragma unroll
        for (int j = 0; j < IPS; ++j) {
            const int r = (wave * IPS + j) * RPI + lane / LPR;   // 0..511
            const bool isw = ((wave * IPS + j) * RPI) >= 256;
            const int rr = isw ? r - 256 : r;
            int row = (isw ? nt * 256 : mt * 256) + rr;
            const int lim = isw ? p.N : p.M;
            if (row > lim - 1) row = lim - 1;
            const int c = (lane % LPR) ^ ((rr >> 1) & (LPR - 1));
            off[j] = ((unsigned)(row - (isw ? nt * 256 : mt * 256)) * (unsigned)p.K + (unsigned)(c << 3)) * 2u;
            isws[j] = isw;
        }
        if (done == 1) t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned rd = (unsigned)((wave * 16 + (lane & 15)) * 128 + (((lane >> 4) ^ ((lane & 15) >> 1)) << 4));
        for (int ks = 0; ks < nk; ++ks) {
#pragma unroll
            for (int j = 0; j < IPS; ++j) {
                bufdma16(isws[j] ? Wr : Ar, off[j], ks * RB, dst + ((ks % KSTEPS) * IPS + j) * 1024);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFL - 1) : "memory");
                if constexpr (NRD > 0) {
#pragma unroll
                    for (int r = 0; r < NRD / IPS; ++r) {
                        typedef float f4 __attribute__((ext_vector_type(4)));
                        f4 v = *(const f4*)(smem + ((rd + (unsigned)((j * (NRD / IPS) + r) * 2048)) & 65535u));
                        asm volatile("" ::"v"(v));
                    }
                }
                if constexpr (NBAR > 0) { if (j % (IPS / NBAR) == 0) __builtin_amdgcn_s_barrier(); }
            }
        }
        if (done == 1) t1 = __builtin_amdgcn_s_memrealtime();
        ++done;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.stamps && tid == 0) { p.stamps[bid * 2] = t0; p.stamps[bid * 2 + 1] = t1; }
}

template <int NW, int INFL, int RB, int NRD = 0, int NBAR = 0>
void run(const char* name, DmaParams p, int grid, int rounds) {
    auto k = dma_kernel<NW, INFL, RB, NRD, NBAR>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ms;
    for (int r = 0; r < rounds; ++r) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), 65536, 0, p);
        hipEventRecord(e0, 0);
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), 65536, 0, p);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1); ms.push_back(t / 5);
    }
    std::sort(ms.begin(), ms.end());
    hipDeviceSynchronize();
    std::vector<unsigned long long> hs(grid * 2);
    hipMemcpy(hs.data(), p.stamps, grid * 16, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int b = 0; b < grid; ++b) if (hs[b * 2] && hs[b * 2 + 1]) { sum += (hs[b * 2 + 1] - hs[b * 2]) / 100.0; ++n; }
    const int ktiles = p.K / 64;
    const double total_ktiles = (double)p.T * ktiles;
    const double per_cu = ms[rounds / 2] * 1e3 / (total_ktiles / grid);
    printf("%-28s grid %3d  launch %8.1f us  = %.3f us per 64-K-tile per CU (%.1f GB/s per CU, %.2f TB/s chip); stamped 2nd tile: %.3f us per K-tile\n",
           name, grid, ms[rounds / 2] * 1e3, per_cu, 65536.0 / per_cu * 1e-3, 65536.0 / per_cu * 1e-6 * grid,
           n ? sum / n / ktiles : 0.0);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int M = 50432, N = argc > 1 ? atoi(argv[1]) : 2304, K = argc > 2 ? atoi(argv[2]) : 768;
    unsigned short *A, *W;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2);
    hipMemset(A, 0x3c, (size_t)M * K * 2); hipMemset(W, 0x3c, (size_t)N * K * 2);
    DmaParams p{}; p.A = A; p.W = W; p.M = M; p.N = N; p.K = K; p.NT = (N + 255) / 256; p.T = ((M + 255) / 256) * p.NT;
    hipMalloc(&p.stamps, 256 * 16); hipMemset(p.stamps, 0, 256 * 16);
    printf("shape M %d N %d K %d: %d tiles\n", M, N, K, p.T);
    const int R = 5;
    for (int grid : {256, 80}) {
        run<8, 8, 128>("8w 64KiB", p, grid, R);
        run<8, 8, 128, 24>("8w 64KiB + 24 ds_read/wave", p, grid, R);
        run<8, 8, 128, 16>("8w 64KiB + 16 ds_read/wave", p, grid, R);
        run<8, 8, 128, 8>("8w 64KiB + 8 ds_read/wave", p, grid, R);
        run<8, 8, 128, 0, 8>("8w 64KiB + 8 barriers", p, grid, R);
        run<8, 8, 128, 0, 4>("8w 64KiB + 4 barriers", p, grid, R);
        run<8, 8, 128, 24, 8>("8w 64KiB + 24 rd + 8 bar", p, grid, R);
        run<8, 8, 128, 24, 4>("8w 64KiB + 24 rd + 4 bar", p, grid, R);
        run<8, 16, 128, 24, 8>("8w 128KiB + 24 rd + 8 bar", p, grid, R);
        run<8, 16, 128, 24, 0>("8w 128KiB + 24 rd", p, grid, R);
        run<4, 16, 128, 32, 0>("4w 64KiB + 32 rd", p, grid, R);
        run<4, 16, 128, 32, 4>("4w 64KiB + 32 rd + 4 bar", p, grid, R);
    }
    return 0;
}
