// Lab: per-CU store throughput for two epilogue store shapes (development only).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
// pattern 0: lane (i=lane&15,q=lane>>4): row = base+i, 16B at col byte (q*16) + seg*64  -> 16 rows x 64B per instr
// pattern 1: full rows: lane l: row = base + (l>>5)... 2 rows x 512B per instr
// pattern 2: 4 rows x 256 B per instr
template <int PAT>
__global__ __launch_bounds__(512) void k(char* C, size_t rowbytes, int tiles_per_wg, int nrows_total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f4 v = {1.f * tid, 2.f, 3.f, 4.f};
    for (int t = 0; t < tiles_per_wg; ++t) {
        // tile = 256 rows x 512 B, tile index
        const size_t tile = (size_t)blockIdx.x + (size_t)t * gridDim.x;
        const size_t ntc = rowbytes / 512;
        const size_t tr = tile / ntc, tc = tile % ntc;
        char* base = C + (tr * 256) * rowbytes + tc * 512;
        if (PAT == 0) {
            const int i = lane & 15, q = lane >> 4;
            const int wr = wave >> 2, wc = wave & 3;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        const int row = ha * 128 + wr * 64 + mi * 16 + i;
                        const int colb = hb * 256 + wc * 64 + q * 16;
                        *(f4*)(base + (size_t)row * rowbytes + colb) = v;
                    }
        } else if (PAT == 1) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int row = s * 16 + wave * 2 + (lane >> 5);
                *(f4*)(base + (size_t)row * rowbytes + (lane & 31) * 16) = v;
            }
        } else if (PAT == 3) {  // 16 rows x 64 B per instr, 4 consecutive lanes per row (half lines)
            const int wr = wave >> 2, wc = wave & 3;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        const int row = ha * 128 + wr * 64 + mi * 16 + (lane >> 2);
                        const int colb = hb * 256 + wc * 64 + (lane & 3) * 16;
                        *(f4*)(base + (size_t)row * rowbytes + colb) = v;
                    }
        } else if (PAT == 5) {  // dwordx2: 4 rows x 128 B per instr (16 lanes x 8 B per row), 32 instrs
            const int wr = wave >> 2, wc = wave & 3;
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 v2 = {v[0], v[1]};
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const int ha = s >> 4, mi = (s >> 2) & 3, j = s & 3;
                const int row = ha * 128 + wr * 64 + mi * 16 + (lane >> 4) * 4 + j;
                const int colb = wc * 128 + (lane & 15) * 8;
                *(f2*)(base + (size_t)row * rowbytes + colb) = v2;
            }
        } else if (PAT == 4) {  // 8 rows x 128 B per instr (full lines, 8 lanes per row)
            const int wr = wave >> 2, wc = wave & 3;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int row = (s >> 3) * 128 + wr * 64 + (s & 7) * 8 + (lane >> 3);
                const int colb = wc * 128 + (lane & 7) * 16;
                *(f4*)(base + (size_t)row * rowbytes + colb) = v;
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) {  // 4 rows x 256B per instr: quadrant-like (half width)
                const int half = s & 1;
                const int row = (s >> 1) * 32 + wave * 4 + (lane >> 4);
                *(f4*)(base + (size_t)row * rowbytes + half * 256 + (lane & 15) * 16) = v;
            }
        }
    }
}
int main() {
    const int M = 50432, N = 2304;  // bf16 -> rowbytes 4608
    const size_t rowbytes = (size_t)N * 2;
    char* C; hipMalloc(&C, (size_t)M * rowbytes + (1 << 20));
    const int tiles = (M / 256) * (N * 2 / 512);
    for (int G : {16, 64, 128, 256}) {
    const int tpw = 6;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 6; ++pat) {
        float ms;
        auto run = [&]() {
            if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(512), 0, 0, C, rowbytes, tpw, M);
            if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(G), dim3(512), 0, 0, C, rowbytes, tpw, M);
            if (pat == 3) hipLaunchKernelGGL(k<3>, dim3(G), dim3(512), 0, 0, C, rowbytes, tpw, M);
            if (pat == 4) hipLaunchKernelGGL(k<4>, dim3(G), dim3(512), 0, 0, C, rowbytes, tpw, M);
            if (pat == 5) hipLaunchKernelGGL(k<5>, dim3(G), dim3(512), 0, 0, C, rowbytes, tpw, M);
            if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(G), dim3(512), 0, 0, C, rowbytes, tpw, M);
        };
        run();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        const double bytes = (double)G * tpw * 256 * 512;
        printf("G %d pat %d: %.1f us  %.2f TB/s  per-tile %.2f us\n", G, pat, ms * 1e3, bytes / (ms * 1e-3) / 1e12, ms * 1e3 / tpw);
    }
    }
    return 0;
}
