// Kernel lab (development only, not part of the product library): times attention_dh64 and its
// timing-only variants (ATTN_VAR, see attention_dh64.hip) on packed random QKV.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DATTN_VAR=n] tools/attn_lab.hip -o tools/attn_lab
//   ./tools/attn_lab <batch> <tokens> <heads> [reps]
#define ATTN_VAR 0  // the timing-only variants (softmax / products / waits removed) lived in attention_dh64.hip until round 3 (history: f9e38f8)
#include "../vit_amd/csrc/attention_dh64.hip"
#include <stdarg.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

static char g_err[512];
int vit_set_error(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    fprintf(stderr, "error %d: %s\n", code, g_err);
    return code;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, N = argc > 2 ? atoi(argv[2]) : 197, H = argc > 3 ? atoi(argv[3]) : 12;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    if (!attention_dh64_eligible(N)) { fprintf(stderr, "N too small\n"); return 1; }
    const size_t D = (size_t)H * 64, nq = (size_t)B * N * 3 * D, no = (size_t)B * N * D;
    std::vector<unsigned short> h(nq);
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < nq; ++i) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        const float f = ((int)((st >> 40) & 0xFFFF) - 32768) / 32768.0f;  // uniform [-1, 1)
        unsigned u; memcpy(&u, &f, 4);
        h[i] = (unsigned short)(u >> 16);
    }
    void *qkv, *out;
    CK(hipMalloc(&qkv, nq * 2));
    CK(hipMalloc(&out, no * 2));
    CK(hipMemcpy(qkv, h.data(), nq * 2, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        if (launch_attention_dh64(qkv, out, nullptr, B, N, H, s)) return 1;
    CK(hipStreamSynchronize(s));
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, s));
        if (launch_attention_dh64(qkv, out, nullptr, B, N, H, s)) return 1;
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms * 1e3f);
    }
    std::vector<unsigned short> ho(no);
    CK(hipMemcpy(ho.data(), out, no * 2, hipMemcpyDeviceToHost));
    unsigned long long hsh = 1469598103934665603ull;
    double sum = 0;
    for (size_t i = 0; i < no; ++i) {
        hsh = (hsh ^ ho[i]) * 1099511628211ull;
        unsigned u = (unsigned)ho[i] << 16; float f; memcpy(&f, &u, 4); sum += f;
    }
    std::sort(ts.begin(), ts.end());
    const double fl = 4.0 * B * H * (double)N * N * 64;
    printf("var %d  B=%d N=%d H=%d: median %.1f us (min %.1f)  %.0f TF  out hash %016llx sum %.6g\n", ATTN_VAR, B, N, H, ts[ts.size() / 2], ts[0],
           fl / ts[ts.size() / 2] * 1e-6, hsh, sum);
    return 0;
}
