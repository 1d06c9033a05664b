// Kernel lab (development only, not part of the product library): A/B-times GEMM variants
// against the shipped kernel on the ViT-B/16 batch-256 shapes, random bf16 operands,
// interleaved rounds in one process (cdna_hip_programming.md rule 24).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gemm_lab.hip -o tools/gemm_lab
// (the in-kernel stamps and the no-store timing build of rounds 1-2 -- G8_LAB in gemm_bf16_8p.hip -- were removed from the
// shipped kernel in round 3; they are in the history at commit f9e38f8)
#include <type_traits>
#include "../vit_amd/csrc/gemm_bf16.hip"
#include "../vit_amd/csrc/gemm_bf16_8p.hip"
#define G8_LAB 1  // the lab hooks of the lab-only 4-wave kernel (nothing in vit_amd/csrc reads it)
#include "gemm4w_lab.hip"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <functional>
#include <string.h>

static char g_err[512];
int vit_set_error(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    fprintf(stderr, "error %d: %s\n", code, g_err);
    return code;
}

int launch_reduce_slabs(const float*, float*, int, int64_t, hipStream_t) { return 0; }  // lab stub (split-K unused here)

namespace lab {

template <int WM_, int WN_, int NSLOTS, int EPI>
__global__ __launch_bounds__(64 * WM_ * WN_, 2) void gemm_ring(const GemmParams P) {
    constexpr int T = 64 * WM_ * WN_, RBM = 128 * WM_, RBN = 64 * WN_, RBK = 32;
    constexpr int A_BYTES = RBM * 64, W_BYTES = RBN * 64, SLOT = A_BYTES + W_BYTES;
    constexpr int NA = RBM * 4 / T, NW = RBN * 4 / T, NPT = NA + NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    const int mt = tile / P.NT, nt = tile - mt * P.NT;
    const int m0 = mt * RBM, n0 = nt * RBN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN_, wn = wave - wm * WN_;
    const int i = lane & 15, q = lane >> 4;

    unsigned a_of[NA], w_of[NW];   // 32-bit element offsets: uniform base + per-lane offset (saddr form)
    const int p = tid & 3;
#pragma unroll
    for (int t = 0; t < NA; ++t) {
        const int r = t * (T / 4) + (tid >> 2);
        int gm = m0 + r; if (gm > P.M - 1) gm = P.M - 1;
        a_of[t] = (unsigned)gm * (unsigned)P.K + (unsigned)((p ^ ((r >> 1) & 3)) << 3);
    }
#pragma unroll
    for (int t = 0; t < NW; ++t) {
        const int rp = t * (T / 4) + (tid >> 2);
        int gn = n0 + w_row_perm(rp); if (gn > P.N - 1) gn = P.N - 1;
        w_of[t] = (unsigned)gn * (unsigned)P.K + (unsigned)((p ^ ((rp >> 1) & 3)) << 3);
    }
    auto stage = [&](int slot, int k0) {
        if (k0 > P.K - RBK) k0 = P.K - RBK;  // tail stages re-read the last K-tile (never consumed)
        unsigned char* sb = smem + slot * SLOT;
        const bf16_t* ab = P.A + k0;
        const bf16_t* wb = P.W + k0;
#pragma unroll
        for (int t = 0; t < NA; ++t) glds16(ab + a_of[t], sb + (t * (T / 4) + wave * 16) * 64);
#pragma unroll
        for (int t = 0; t < NW; ++t) glds16(wb + w_of[t], sb + A_BYTES + (t * (T / 4) + wave * 16) * 64);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (P.K + RBK - 1) / RBK;
#pragma unroll
    for (int t = 0; t < NSLOTS; ++t) stage(t, t * RBK);

    const int pos = (q ^ ((i >> 1) & 3)) << 4;
    const int a_off = (wm * 128 + i) * 64 + pos;
    const int w_off = A_BYTES + (wn * 64 + i) * 64 + pos;

    bf16x8 wf0[4], wf1[4], xa[4], xb[4];
    auto ldw = [&](bf16x8 (&wf)[4], int slot) {
#pragma unroll
        for (int b = 0; b < 4; ++b) wf[b] = *(const bf16x8*)(smem + slot * SLOT + w_off + b * 1024);
    };
    auto ldx = [&](bf16x8 (&xf)[4], int slot, int mh) {
#pragma unroll
        for (int a = 0; a < 4; ++a) xf[a] = *(const bf16x8*)(smem + slot * SLOT + a_off + (mh * 4 + a) * 1024);
    };
    auto mma = [&](bf16x8 (&wf)[4], bf16x8 (&xf)[4], int mh) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[mh * 4 + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[mh * 4 + a][b], 0, 0, 0);
    };
#define WAIT_BARRIER(N) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory")

    WAIT_BARRIER((NSLOTS - 1) * NPT);
    ldw(wf0, 0);
    ldx(xa, 0, 0);
    int slot = 0;
#pragma unroll 2
    for (int kt = 0; kt < nk; ++kt) {
        int nslot = slot + 1; if (nslot == NSLOTS) nslot = 0;
        ldx(xb, slot, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(wf0, xa, 0);
        __builtin_amdgcn_sched_barrier(0);
        WAIT_BARRIER((NSLOTS - 2) * NPT);
        stage(slot, (kt + NSLOTS) * RBK);
        ldw(wf1, nslot);
        ldx(xa, nslot, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(wf0, xb, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 4; ++b) wf0[b] = wf1[b];
        slot = nslot;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain dummy stages before exit

    const int nb = n0 + wn * 64 + q * 16;
    float bias[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bias[e] = (P.bias != nullptr && nb + e < P.N) ? P.bias[nb + e] : 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int m = m0 + wm * 128 + a * 16 + i;
        if (m >= P.M) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (nb + h * 8 >= P.N) continue;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)apply_act<EPI>(acc[a][h * 2 + e / 4][e % 4] + bias[h * 8 + e]);
            *(bf16x8*)(P.C + (size_t)m * P.N + nb + h * 8) = o;
        }
    }
}

template <int WM_, int WN_, int NSLOTS, int EPI>
void launch_ring(GemmParams P, hipStream_t s) {
    constexpr int RBM = 128 * WM_, RBN = 64 * WN_;
    constexpr int LDS = NSLOTS * (RBM + RBN) * 64;
    auto k = gemm_ring<WM_, WN_, NSLOTS, EPI>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    P.NT = (P.N + RBN - 1) / RBN;
    const int MT = (P.M + RBM - 1) / RBM;
    hipLaunchKernelGGL(k, dim3(MT * P.NT), dim3(64 * WM_ * WN_), LDS, s, P);
}

}  // namespace lab

static void fill(std::vector<uint16_t>& v, float scale, unsigned seed) {
    unsigned s = seed;
    for (auto& x : v) {
        s = s * 1664525u + 1013904223u;
        float f = ((int)(s >> 8) % 20001 - 10000) * 1e-4f * scale;
        unsigned u; memcpy(&u, &f, 4);
        x = (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
    }
}

int main(int argc, char** argv) {
    const char* shape_f = argc > 1 ? argv[1] : "";
    const char* var_f = argc > 2 ? argv[2] : "";
    const int rounds = argc > 3 ? atoi(argv[3]) : 5;
    struct Shape { const char* name; int M, N, K; int epi; };
    Shape shapes[] = {{"qkv", 50432, 2304, 768, VIT_EPI_BIAS}, {"oproj", 50432, 768, 768, VIT_EPI_BIAS},
                      {"fc1", 50432, 3072, 768, VIT_EPI_BIAS}, {"fc1_gelu", 50432, 3072, 768, VIT_EPI_GELU},
                      {"fc2", 50432, 768, 3072, VIT_EPI_BIAS}, {"oproj_res", 50432, 768, 768, VIT_EPI_RESIDUAL},
                      {"fc2_res", 50432, 768, 3072, VIT_EPI_RESIDUAL}, {"ragged", 1000, 1000, 640, VIT_EPI_RESIDUAL}, {"ragged_g", 300, 520, 256, VIT_EPI_GELU}};
    hipStream_t st; hipStreamCreate(&st);
    for (auto& sh : shapes) {
        if (*shape_f && strcmp(shape_f, "all") && strcmp(shape_f, sh.name)) continue;
        const size_t na = (size_t)sh.M * sh.K, nw = (size_t)sh.N * sh.K, nc = (size_t)sh.M * sh.N;
        std::vector<uint16_t> ha(na), hw(nw);
        fill(ha, 1.0f, 1); fill(hw, 0.05f, 2);
        std::vector<float> hb(sh.N);
        for (int j = 0; j < sh.N; ++j) hb[j] = 0.01f * (j % 37 - 18);
        bf16_t *A, *W, *C0, *C1, *R; float* B;
        hipMalloc(&A, na * 2); hipMalloc(&W, nw * 2 + (size_t)sh.N * 256); hipMalloc(&C0, nc * 2); hipMalloc(&C1, nc * 2); hipMalloc(&B, sh.N * 4); hipMalloc(&R, nc * 2);
        { std::vector<uint16_t> hr(nc); fill(hr, 1.0f, 3); hipMemcpy(R, hr.data(), nc * 2, hipMemcpyHostToDevice); }
        hipMemcpy(A, ha.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), nw * 2, hipMemcpyHostToDevice);
        hipMemcpy(B, hb.data(), sh.N * 4, hipMemcpyHostToDevice);
        GemmParams P{}; P.A = A; P.W = W; P.bias = B; P.M = sh.M; P.N = sh.N; P.K = sh.K;
        struct Var { const char* name; std::function<void(bf16_t*)> run; };
        const bool gelu = sh.epi == VIT_EPI_GELU;
        std::vector<Var> vars;
        vars.push_back({"shipped", [&](bf16_t* C) { launch_gemm_bf16_generic(A, W, B, R, C, sh.M, sh.N, sh.K, sh.epi, nullptr, nullptr, st); }});
        vars.push_back({"ring<2,4,4>", [&](bf16_t* C) { GemmParams Q = P; Q.C = C; gelu ? lab::launch_ring<2, 4, 4, VIT_EPI_GELU>(Q, st) : lab::launch_ring<2, 4, 4, VIT_EPI_BIAS>(Q, st); }});
        vars.push_back({"g8p", [&](bf16_t* C) { launch_gemm_bf16_8p(A, W, B, R, C, sh.M, sh.N, sh.K, sh.epi, nullptr, st); }});
        if (sh.epi == VIT_EPI_BIAS || sh.epi == VIT_EPI_GELU) {
            vars.push_back({"g4w", [&](bf16_t* C) { launch_gemm_bf16_4w(A, W, B, R, C, sh.M, sh.N, sh.K, sh.epi, nullptr, st); }});
            vars.push_back({"ns4w", [&](bf16_t* C) { g4_dbg = 1; launch_gemm_bf16_4w(A, W, B, R, C, sh.M, sh.N, sh.K, sh.epi, nullptr, st); g4_dbg = 0; }});
            if (sh.epi == VIT_EPI_BIAS) {  // timing only: W read at a padded row pitch (the allocation has the slack)
                vars.push_back({"4w_ldw64", [&](bf16_t* C) { g4_dbg = (1 << 20); launch_gemm_bf16_4w(A, W, B, R, C, sh.M, sh.N, sh.K, sh.epi, nullptr, st); g4_dbg = 0; }});
                vars.push_back({"4w_ldw8", [&](bf16_t* C) { g4_dbg = (1 << 21); launch_gemm_bf16_4w(A, W, B, R, C, sh.M, sh.N, sh.K, sh.epi, nullptr, st); g4_dbg = 0; }});
                vars.push_back({"4w_ldw32", [&](bf16_t* C) { g4_dbg = (1 << 22); launch_gemm_bf16_4w(A, W, B, R, C, sh.M, sh.N, sh.K, sh.epi, nullptr, st); g4_dbg = 0; }});
            }
            if (sh.epi == VIT_EPI_GELU)
                vars.push_back({"4w_pkW", [&](bf16_t* C) { g4_dbg = 32 << 1; launch_gemm_bf16_4w(A, W, B, R, C, sh.M, sh.N, sh.K, sh.epi, nullptr, st); g4_dbg = 0; }});
            if (sh.epi == VIT_EPI_BIAS)
                for (int v : {1, 2, 3, 4, 5, 6, 7, 8, 16, 32, 48, 56}) {
                    static const char* nm[57] = {"", "4w_nodma", "4w_nobar", "4w_nodmabar", "4w_nords", "4w_nodma_rds", "4w_nords_bar", "4w_mfmaonly", "4w_nomfma"};
                    nm[16] = "4w_pkA"; nm[32] = "4w_pkW"; nm[48] = "4w_pkAW"; nm[56] = "4w_pkAW_nomfma";
                    vars.push_back({nm[v], [&, v](bf16_t* C) { g4_dbg = (v >= 16 && v != 56 ? 0 : 1) | (v << 1); launch_gemm_bf16_4w(A, W, B, R, C, sh.M, sh.N, sh.K, sh.epi, nullptr, st); g4_dbg = 0; }});
                }
        }
        // residual epilogue that also writes the row sums for the next LayerNorm (GemmExt::row_part); same C bits
        static float* rpart = nullptr;
        if (!rpart) hipMalloc(&rpart, (size_t)60000 * 8 * 64);
        if (sh.epi == VIT_EPI_RESIDUAL && sh.N % 64 == 0)
            vars.push_back({"8p_rst", [&](bf16_t* C) {
                GemmExt e{}; e.row_part = rpart;
                launch_gemm_bf16_8p(A, W, B, R, C, sh.M, sh.N, sh.K, sh.epi, &e, st); }});
        // the shipped kernel once more, behind the variant above: the order of the variants inside a round is fixed, and a variant
        // that follows ten launches of a heavier one starts on a chip that has already lowered its clock
        vars.push_back({"g8p_again", [&](bf16_t* C) { launch_gemm_bf16_8p(A, W, B, R, C, sh.M, sh.N, sh.K, sh.epi, nullptr, st); }});  // (kept as the check of the harness itself)
        // folded-LayerNorm epilogue (GemmExt::ln_stats): bias / activation shapes only; the statistics are arbitrary
        // numbers (timing), so this variant is reported as a mismatch against "shipped"
        static float *lnst = nullptr, *lnc = nullptr;
        if (!lnst) {
            hipMalloc(&lnst, (size_t)60000 * 8); hipMemset(lnst, 0, (size_t)60000 * 8);
            hipMalloc(&lnc, (size_t)8192 * 4); hipMemset(lnc, 0, (size_t)8192 * 4);
        }
        if (sh.epi == VIT_EPI_BIAS || sh.epi == VIT_EPI_GELU)
            vars.push_back({"lnf8p", [&](bf16_t* C) {
                GemmExt e{}; e.ln_stats = lnst; e.ln_colsum = lnc;
                launch_gemm_bf16_8p(A, W, B, R, C, sh.M, sh.N, sh.K, sh.epi, &e, st); }});
        vars.push_back({"ring<2,2,3>", [&](bf16_t* C) { GemmParams Q = P; Q.C = C; gelu ? lab::launch_ring<2, 2, 3, VIT_EPI_GELU>(Q, st) : lab::launch_ring<2, 2, 3, VIT_EPI_BIAS>(Q, st); }});
        vars.push_back({"ring<1,4,3>", [&](bf16_t* C) { GemmParams Q = P; Q.C = C; gelu ? lab::launch_ring<1, 4, 3, VIT_EPI_GELU>(Q, st) : lab::launch_ring<1, 4, 3, VIT_EPI_BIAS>(Q, st); }});
        if (*var_f) { std::vector<Var> keep; for (auto& v : vars) if (!strcmp(v.name, "shipped") || strstr(var_f, v.name)) keep.push_back(v); vars = keep; }
        // correctness vs shipped
        vars[0].run(C0); hipStreamSynchronize(st);
        std::vector<uint16_t> h0(nc), h1(nc);
        hipMemcpy(h0.data(), C0, nc * 2, hipMemcpyDeviceToHost);
        for (size_t v = 1; v < vars.size(); ++v) {
            hipMemset(C1, 0xFF, nc * 2);
            vars[v].run(C1); hipStreamSynchronize(st);
            hipError_t e = hipGetLastError();
            hipMemcpy(h1.data(), C1, nc * 2, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (size_t j = 0; j < nc; ++j) bad += (h0[j] != h1[j]);
            printf("%-9s %-12s mismatches vs shipped: %zu / %zu (%s)\n", sh.name, vars[v].name, bad, nc, hipGetErrorString(e));
        }
        // timing: interleaved rounds
        const int iters = 10;
        std::vector<std::vector<float>> ms(vars.size());
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        // The chip lowers its clock within a millisecond or two of sustained matrix load, so a variant that follows ten
        // launches of a lighter kernel (the generic "shipped" one) is timed on a faster chip than the variant behind it: the
        // same kernel measured 195.5 us in second place and 203.5 us in fourth (fc2 + residual, round 3). Every block
        // therefore starts with ten untimed launches of its own variant, and the order rotates from round to round.
        for (int r = 0; r < rounds; ++r)
            for (size_t vv = 0; vv < vars.size(); ++vv) {
                const size_t v = (vv + (size_t)r) % vars.size();
                for (int it = 0; it < iters; ++it) vars[v].run(C1);
                hipEventRecord(e0, st);
                for (int it = 0; it < iters; ++it) vars[v].run(C1);
                hipEventRecord(e1, st); hipEventSynchronize(e1);
                float t; hipEventElapsedTime(&t, e0, e1); ms[v].push_back(t / iters);
            }
        const double flop = 2.0 * sh.M * sh.N * sh.K;
        for (size_t v = 0; v < vars.size(); ++v) {
            std::sort(ms[v].begin(), ms[v].end());
            printf("%-9s %-12s median %8.1f us  %7.1f TF   (min %8.1f us)\n", sh.name, vars[v].name, ms[v][rounds / 2] * 1e3,
                   flop / (ms[v][rounds / 2] * 1e-3) / 1e12, ms[v][0] * 1e3);
        }
        hipFree(A); hipFree(W); hipFree(C0); hipFree(C1); hipFree(B); hipFree(R);
        fflush(stdout);
    }
    return 0;
}
