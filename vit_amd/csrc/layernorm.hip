// LayerNorm forward: y = (x - mean) / sqrt(var + eps) * gamma + beta, biased variance,
// fp32 statistics (nn.LayerNorm as used at modeling_vit.py:261-262,274,281,385 and
// modeling_clip.py:605,607). HBM-bound: one wavefront per row, the row lives in registers
// between the load and the store (read once, write once), 8/16-byte coalesced accesses,
// mean and centred variance by two in-register passes + a 64-lane butterfly reduction.
// Also the small elementwise kernels of the path (dtype convert, CLS rows).
#include "vit_common.h"
#include <stdlib.h>

namespace {

template <typename T>
struct Vec4;  // 4 elements of T as one coalesced access
template <>
struct Vec4<float> {
    __device__ static f32x4 load(const float* p) { return *(const f32x4*)p; }
    __device__ static void store(float* p, f32x4 v) { *(f32x4*)p = v; }
};
template <>
struct Vec4<bf16_t> {
    __device__ static f32x4 load(const bf16_t* p) {
        const bf16x4 t = *(const bf16x4*)p;
        return f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
    }
    __device__ static void store(bf16_t* p, f32x4 v) {
        bf16x4 t;
        t[0] = (bf16_t)v[0]; t[1] = (bf16_t)v[1]; t[2] = (bf16_t)v[2]; t[3] = (bf16_t)v[3];
        *(bf16x4*)p = t;
    }
};

// NIT = ceil(dim / 256): register-resident row, 4 elements per lane per iteration.
template <typename T, int NIT>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ y, int rows,
                                                        int dim, int64_t xs, int64_t ys, float eps) {
    const int lane = threadIdx.x & 63;
    const int wave_in_grid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const float inv_d = 1.0f / (float)dim;
    for (int row = wave_in_grid; row < rows; row += nwaves) {
        const T* xr = x + (size_t)row * xs;
        f32x4 v[NIT];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < dim) {
                v[it] = Vec4<T>::load(xr + c);
                s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
            } else {
                v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        const float mean = wave_sum(s) * inv_d;
        float ss = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < dim) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[it][e] - mean;
                    ss += d * d;
                }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(ss) * inv_d + eps);
        T* yr = y + (size_t)row * ys;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < dim) {
                const f32x4 g = *(const f32x4*)(gamma + c);
                const f32x4 b = *(const f32x4*)(beta + c);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[it][e] - mean) * rstd * g[e] + b[e];
                Vec4<T>::store(yr + c, o);
            }
        }
    }
}

// Row statistics only: stats[row] = (mean * rstd, rstd). Used when the normalisation itself is folded
// into the GEMM that consumes the row (vit_capi.hip "LayerNorm folded into the next GEMM"): the row is
// read once and nothing but 8 bytes per row is written.
// Half a wave per row, 16-byte loads (8 bf16 per lane and step, NIT = ceil(dim / 256) steps), R row pairs per
// wave in flight: a read-only pass wants wide loads and many bytes in flight.
// R row pairs per wave: 4 for big inputs (bytes in flight), 1 below ~20k rows, where 4 leaves one block per CU and the
// launch is latency-bound (CLIP-L/14 at 16 images: 7.1 -> see DESIGN); a row's arithmetic does not depend on R.
template <int NIT, int R>
__global__ __launch_bounds__(256) void ln_stats_kernel(const bf16_t* __restrict__ x, float2* __restrict__ stats, int rows,
                                                       int dim, int64_t xs, float eps) {
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    const int wave_in_grid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const float inv_d = 1.0f / (float)dim;
    const int row0 = wave_in_grid * (2 * R);
    if (row0 >= rows) return;  // whole wave
    bf16x8 v[R][NIT];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int row = row0 + 2 * r + half;
        if (row > rows - 1) row = rows - 1;  // the tail re-reads the last row
        const bf16_t* xr = x + (size_t)row * xs;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 32 + hl) * 8;
            if (c < dim) {
                v[r][it] = *(const bf16x8*)(xr + c);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[r][it][e] = (bf16_t)0.0f;
            }
        }
    }
    // sum over the 32 lanes of a half wave (xor 16 .. 1 stays inside the half)
    auto half_sum = [](float a) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        return a;
    };
    // explicit fmaf and one code shape for every r: a row's statistics must not depend on which slot of the
    // wave it landed in (a permuted batch gives the permuted output bit for bit)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const bf16x8 t = v[r][it];
            s += (((float)t[0] + (float)t[1]) + ((float)t[2] + (float)t[3])) + (((float)t[4] + (float)t[5]) + ((float)t[6] + (float)t[7]));
        }
        const float mean = half_sum(s) * inv_d;
        float ss = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 32 + hl) * 8;
            if (c < dim) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (float)v[r][it][e] - mean;
                    ss = fmaf(d, d, ss);
                }
            }
        }
        const float rstd = 1.0f / sqrtf(fmaf(half_sum(ss), inv_d, eps));
        const int row = row0 + 2 * r + half;
        if (hl == 0 && row < rows) stats[row] = make_float2(mean * rstd, rstd);
    }
}

// The same statistics in ONE pass, for dim % 64 == 0 (every ViT width): the arithmetic of the residual epilogue of the
// persistent GEMM (row16_stats, vit_common.h), so that a row gets the same (mean * rstd, rstd) whichever of the two produced
// them. A 16-lane DPP row takes one row at a time: per 64-column group 4 stored values per lane, the group's (sum, sum of
// squares) from row16_stats, the groups added left to right, ln_from_sums at the end; lane 0 of the row holds the result.
// A wave handles 4 * RW rows, all of a row's loads issued before the first use (8 bytes per lane and group).
template <int NG, int RW>
__global__ __launch_bounds__(256) void row_stats_kernel(const bf16_t* __restrict__ x, float2* __restrict__ stats, int rows,
                                                        int64_t xs, float inv_d, float eps) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const int wave_in_grid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int row0 = wave_in_grid * (4 * RW);
    if (row0 >= rows) return;  // whole wave
    u32x2 v[RW][NG];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        int row = row0 + r * 4 + q;
        if (row > rows - 1) row = rows - 1;  // the tail re-reads the last row
        const bf16_t* xr = x + (size_t)row * xs + 4 * i;
#pragma unroll
        for (int g = 0; g < NG; ++g) v[r][g] = *(const u32x2*)(xr + g * 64);
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float a, b;
            row16_stats(v[r][g].x, v[r][g].y, a, b);
            sm = __fadd_rn(sm, a);
            sq = __fadd_rn(sq, b);
        }
        const int row = row0 + r * 4 + q;
        if (i == 0 && row < rows) stats[row] = ln_from_sums(sm, sq, inv_d, eps);
    }
}

// ... and from the slabs a residual GEMM left: slab g holds (sum, sum of squares) over columns [64 g, 64 g + 64) of every row.
// One thread per row, the slabs added left to right like row_stats_kernel does; NG > 0: all loads issued before the first add
// (the launch is one memory round trip long), NG = 0: any number of slabs, four at a time.
template <int NG>
__global__ __launch_bounds__(256) void row_stats_finalize_kernel(const float2* __restrict__ part, float2* __restrict__ stats,
                                                                 int rows, int ng, float inv_d, float eps) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float sm = 0.f, sq = 0.f;
    if constexpr (NG > 0) {
        float2 p[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) p[g] = part[(size_t)g * rows + row];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            sm = __fadd_rn(sm, p[g].x);
            sq = __fadd_rn(sq, p[g].y);
        }
    } else {
        for (int g0 = 0; g0 < ng; g0 += 4) {
            float2 p[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k] = (g0 + k < ng) ? part[(size_t)(g0 + k) * rows + row] : make_float2(0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (g0 + k < ng) {
                    sm = __fadd_rn(sm, p[k].x);
                    sq = __fadd_rn(sq, p[k].y);
                }
        }
    }
    stats[row] = ln_from_sums(sm, sq, inv_d, eps);
}

// Fold a LayerNorm's affine part into the Linear that follows it. With xh = (x - mean) * rstd:
//   (xh * gamma + beta) W^T + bias = rstd * (x Wg^T) - (mean * rstd) * c + d,
//   Wg[n,k] = W[n,k] * gamma[k] (rounded to bf16 like any weight), c[n] = sum_k Wg[n,k] (of the ROUNDED
//   values, so the mean term cancels exactly what the MFMA multiplies), d[n] = bias[n] + sum_k W[n,k] beta[k].
// One wave per output row n.
__global__ __launch_bounds__(256) void ln_fold_kernel(const bf16_t* __restrict__ W, int64_t pitch, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ bias,
                                                      bf16_t* __restrict__ Wg, float* __restrict__ c, float* __restrict__ d, int N,
                                                      int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;  // whole wave
    float sc = 0.f, sd = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = (float)W[(size_t)n * pitch + k];
        const bf16_t wg = (bf16_t)(w * gamma[k]);
        Wg[(size_t)n * pitch + k] = wg;
        sc += (float)wg;
        sd += w * beta[k];
    }
    sc = wave_sum(sc);
    sd = wave_sum(sd);
    if (lane == 0) {
        c[n] = sc;
        d[n] = sd + (bias ? bias[n] : 0.f);
    }
}

template <typename T, int NIT>
int launch_ln_t(const void* x, const float* g, const float* b, void* y, int rows, int dim, int64_t xs, int64_t ys,
                float eps, hipStream_t stream) {
    int blocks = ceil_div(rows, 4);
    if (blocks > 256 * 8) blocks = 256 * 8;  // grid-stride beyond 8 blocks per CU
    hipLaunchKernelGGL((layernorm_kernel<T, NIT>), dim3(blocks), dim3(256), 0, stream, (const T*)x, g, b, (T*)y, rows,
                       dim, xs, ys, eps);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

template <typename T>
int launch_ln_nit(const void* x, const float* g, const float* b, void* y, int rows, int dim, int64_t xs, int64_t ys,
                  float eps, hipStream_t stream) {
    switch (ceil_div(dim, 256)) {
        case 1: return launch_ln_t<T, 1>(x, g, b, y, rows, dim, xs, ys, eps, stream);
        case 2: return launch_ln_t<T, 2>(x, g, b, y, rows, dim, xs, ys, eps, stream);
        case 3: return launch_ln_t<T, 3>(x, g, b, y, rows, dim, xs, ys, eps, stream);
        case 4: return launch_ln_t<T, 4>(x, g, b, y, rows, dim, xs, ys, eps, stream);
        case 5: return launch_ln_t<T, 5>(x, g, b, y, rows, dim, xs, ys, eps, stream);
        case 6: return launch_ln_t<T, 6>(x, g, b, y, rows, dim, xs, ys, eps, stream);
        case 7: return launch_ln_t<T, 7>(x, g, b, y, rows, dim, xs, ys, eps, stream);
        case 8: return launch_ln_t<T, 8>(x, g, b, y, rows, dim, xs, ys, eps, stream);
    }
    return vit_set_error(VIT_E_INVALID, "layernorm: dim %d > 2048 unsupported", dim);
}

// ---- elementwise: dtype convert ----
template <typename S, typename D>
__global__ __launch_bounds__(256) void convert_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            Vec4<D>::store(dst + i, Vec4<S>::load(src + i));
        } else {
            for (int64_t j = i; j < n; ++j) dst[j] = (D)(float)src[j];
        }
    }
}

template <typename S, typename D>
int launch_convert_t(const void* src, void* dst, int64_t n, hipStream_t stream) {
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL((convert_kernel<S, D>), dim3((int)blocks), dim3(256), 0, stream, (const S*)src, (D*)dst, n);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

// ---- pitched convert (weight ingestion: rows of `cols` elements into rows of `pitch`) ----
template <typename S, typename D>
__global__ __launch_bounds__(256) void convert_2d_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t rows,
                                                         int64_t cols, int64_t pitch) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols, c = i - r * cols;
        dst[r * pitch + c] = (D)(float)src[i];
    }
}

template <typename S, typename D>
int launch_convert_2d_t(const void* src, void* dst, int64_t rows, int64_t cols, int64_t pitch, hipStream_t stream) {
    int64_t blocks = (rows * cols + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL((convert_2d_kernel<S, D>), dim3((int)blocks), dim3(256), 0, stream, (const S*)src, (D*)dst, rows,
                       cols, pitch);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

// ---- CLS rows: x[b, 0, :] = cls + pos[0]  (ViTEmbeddings.forward, modeling_vit.py:147-158) ----
template <typename T>
__global__ __launch_bounds__(256) void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                                       T* __restrict__ x, int batch, int seq, int dim) {
    const int total = batch * dim;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / dim, d = i - b * dim;
        x[(size_t)b * seq * dim + d] = (T)(cls[d] + pos[d]);
    }
}

// ---- patch rows: A[b * np + py * grid + px][c * P * P + i * P + j] = pixels[b, c, py * P + i, px * P + j], columns
// >= C * P * P zero (the Conv2d-as-GEMM operand, modeling_vit.py:62-69, written out for patch sizes whose pixel rows
// are not 16-byte pieces). One thread per 8 output columns: 16-byte stores along k.
__global__ __launch_bounds__(256) void im2col_rows_kernel(const bf16_t* __restrict__ pix, bf16_t* __restrict__ A, long rows,
                                                          int kp, PatchGeom g) {
    const int kchunks = kp >> 3;
    const long total = rows * kchunks;
    const int np = g.grid * g.grid, pp = g.patch * g.patch;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx / kchunks;
        const int k0 = (int)(idx - row * kchunks) << 3;
        const int b = (int)(row / np), rem = (int)(row - (long)b * np);
        const int py = rem / g.grid, px = rem - py * g.grid;
        const bf16_t* src = pix + ((size_t)b * g.channels * g.image + (size_t)py * g.patch) * g.image + (size_t)px * g.patch;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            if (k < g.kreal) {
                const int c = k / pp, r2 = k - c * pp;
                const int i = r2 / g.patch, j = r2 - i * g.patch;
                v[e] = src[((size_t)c * g.image + i) * g.image + j];
            } else {
                v[e] = (bf16_t)0.0f;
            }
        }
        *(bf16x8*)(A + row * kp + k0) = v;
    }
}

}  // namespace

int launch_im2col_rows(const void* pixels, void* A, long rows, int kp, const PatchGeom* g, hipStream_t stream) {
    VIT_REQUIRE(rows > 0 && kp % 8 == 0 && kp >= g->kreal, "im2col_rows: bad shape");
    long blocks = (rows * (kp >> 3) + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(im2col_rows_kernel, dim3((int)blocks), dim3(256), 0, stream, (const bf16_t*)pixels, (bf16_t*)A, rows, kp, *g);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

int launch_layernorm(const void* x, const float* g, const float* b, void* y, int rows, int dim, int64_t xs, int64_t ys,
                     float eps, int dtype, hipStream_t stream) {
    VIT_REQUIRE(rows > 0 && dim > 0, "layernorm: empty input rows=%d dim=%d", rows, dim);
    VIT_REQUIRE(dim % 4 == 0 && xs % 4 == 0 && ys % 4 == 0, "layernorm: dim and row strides must be multiples of 4");
    if (dtype == VIT_BF16) return launch_ln_nit<bf16_t>(x, g, b, y, rows, dim, xs, ys, eps, stream);
    if (dtype == VIT_F32) return launch_ln_nit<float>(x, g, b, y, rows, dim, xs, ys, eps, stream);
    return vit_set_error(VIT_E_INVALID, "layernorm: bad dtype %d", dtype);
}

// Which statistics kernel a width takes: the one-pass form (dim % 64 == 0: its bits are what the residual GEMM's row sums
// reproduce) or the two-pass form (any dim % 8 == 0; no cancellation in the variance). VIT_AMD_STATS_ONE_PASS=0 in the
// environment keeps the two-pass kernel for every width -- and with it the row sums of the residual epilogues off
// (vit_capi.hip), since they could not match its bits.
bool ln_stats_one_pass(int dim) {
    static const bool on = !(getenv("VIT_AMD_STATS_ONE_PASS") && atoi(getenv("VIT_AMD_STATS_ONE_PASS")) == 0);
    return on && dim % 64 == 0;
}

int launch_ln_stats(const void* x, float* stats, int rows, int dim, int64_t xs, float eps, int dtype, hipStream_t stream) {
    VIT_REQUIRE(rows > 0 && dim > 0 && dim % 8 == 0 && xs % 8 == 0 && dim <= 2048, "ln_stats: bad shape rows=%d dim=%d (multiples of 8)", rows, dim);
    VIT_REQUIRE(dtype == VIT_BF16, "ln_stats: bf16 only");
    if (ln_stats_one_pass(dim) && xs % 4 == 0) {
        // the one-pass form whose bits the residual GEMM's row sums reproduce (row_stats_kernel); other widths keep the
        // two-pass kernel below. Which of the two runs depends on dim alone, never on the batch.
        const int ng = dim / 64;
        const bool small_r = rows < 20000;  // as below: few rows want more waves, many rows more bytes in flight per wave
        const int rw = small_r ? 1 : 2;
        const int blocks_r = ceil_div(rows, 16 * rw);
        const float inv_d = 1.0f / (float)dim;
#define VIT_ROW_STATS_CASE(NG)                                                                                                        \
    case NG:                                                                                                                         \
        if (small_r) hipLaunchKernelGGL((row_stats_kernel<NG, 1>), dim3(blocks_r), dim3(256), 0, stream, (const bf16_t*)x, (float2*)stats, rows, xs, inv_d, eps); \
        else hipLaunchKernelGGL((row_stats_kernel<NG, 2>), dim3(blocks_r), dim3(256), 0, stream, (const bf16_t*)x, (float2*)stats, rows, xs, inv_d, eps);        \
        break;
        switch (ng) {
            VIT_ROW_STATS_CASE(1) VIT_ROW_STATS_CASE(2) VIT_ROW_STATS_CASE(3) VIT_ROW_STATS_CASE(4) VIT_ROW_STATS_CASE(5) VIT_ROW_STATS_CASE(6)
            VIT_ROW_STATS_CASE(7) VIT_ROW_STATS_CASE(8) VIT_ROW_STATS_CASE(9) VIT_ROW_STATS_CASE(10) VIT_ROW_STATS_CASE(11) VIT_ROW_STATS_CASE(12)
            VIT_ROW_STATS_CASE(13) VIT_ROW_STATS_CASE(14) VIT_ROW_STATS_CASE(15) VIT_ROW_STATS_CASE(16) VIT_ROW_STATS_CASE(17) VIT_ROW_STATS_CASE(18)
            VIT_ROW_STATS_CASE(19) VIT_ROW_STATS_CASE(20) VIT_ROW_STATS_CASE(21) VIT_ROW_STATS_CASE(22) VIT_ROW_STATS_CASE(23) VIT_ROW_STATS_CASE(24)
            VIT_ROW_STATS_CASE(25) VIT_ROW_STATS_CASE(26) VIT_ROW_STATS_CASE(27) VIT_ROW_STATS_CASE(28) VIT_ROW_STATS_CASE(29) VIT_ROW_STATS_CASE(30)
            VIT_ROW_STATS_CASE(31) VIT_ROW_STATS_CASE(32)
        }
#undef VIT_ROW_STATS_CASE
        VIT_CHECK_HIP(hipGetLastError());
        return VIT_OK;
    }
    const bool small = rows < 20000;
    const int blocks = ceil_div(rows, small ? 8 : 32);  // 4 waves x R row pairs, one step per wave
    const bf16_t* xp = (const bf16_t*)x;
    float2* sp = (float2*)stats;
#define VIT_LN_STATS_CASE(NIT)                                                                                                   \
    case NIT:                                                                                                                    \
        if (small) hipLaunchKernelGGL((ln_stats_kernel<NIT, 1>), dim3(blocks), dim3(256), 0, stream, xp, sp, rows, dim, xs, eps); \
        else hipLaunchKernelGGL((ln_stats_kernel<NIT, 4>), dim3(blocks), dim3(256), 0, stream, xp, sp, rows, dim, xs, eps);       \
        break;
    switch (ceil_div(dim, 256)) {
        VIT_LN_STATS_CASE(1) VIT_LN_STATS_CASE(2) VIT_LN_STATS_CASE(3) VIT_LN_STATS_CASE(4)
        VIT_LN_STATS_CASE(5) VIT_LN_STATS_CASE(6) VIT_LN_STATS_CASE(7) VIT_LN_STATS_CASE(8)
    }
#undef VIT_LN_STATS_CASE
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

int launch_row_stats_finalize(const float* row_part, float* stats, int rows, int dim, float eps, hipStream_t stream) {
    VIT_REQUIRE(rows > 0 && dim > 0 && dim % 64 == 0, "row_stats_finalize: bad shape rows=%d dim=%d", rows, dim);
    const int ng = dim / 64;
    const dim3 grid(ceil_div(rows, 256));
    const float inv_d = 1.0f / (float)dim;
    const float2* pp = (const float2*)row_part;
    float2* sp = (float2*)stats;
    switch (ng) {  // ViT-B (768), ViT-L (1024), ViT-H (1280): fully unrolled; anything else takes the loop
        case 12: hipLaunchKernelGGL(row_stats_finalize_kernel<12>, grid, dim3(256), 0, stream, pp, sp, rows, ng, inv_d, eps); break;
        case 16: hipLaunchKernelGGL(row_stats_finalize_kernel<16>, grid, dim3(256), 0, stream, pp, sp, rows, ng, inv_d, eps); break;
        case 20: hipLaunchKernelGGL(row_stats_finalize_kernel<20>, grid, dim3(256), 0, stream, pp, sp, rows, ng, inv_d, eps); break;
        default: hipLaunchKernelGGL(row_stats_finalize_kernel<0>, grid, dim3(256), 0, stream, pp, sp, rows, ng, inv_d, eps); break;
    }
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

int launch_ln_fold(const void* W, int64_t pitch, const float* gamma, const float* beta, const float* bias, void* Wg, float* c,
                   float* d, int N, int K, hipStream_t stream) {
    VIT_REQUIRE(N > 0 && K > 0 && pitch >= K, "ln_fold: bad shape");
    hipLaunchKernelGGL(ln_fold_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, stream, (const bf16_t*)W, pitch, gamma, beta, bias,
                       (bf16_t*)Wg, c, d, N, K);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

int launch_convert(const void* src, int sdt, void* dst, int ddt, int64_t n, hipStream_t stream) {
    VIT_REQUIRE(n >= 0, "convert: negative size");
    if (n == 0) return VIT_OK;
    if (sdt == VIT_F32 && ddt == VIT_BF16) return launch_convert_t<float, bf16_t>(src, dst, n, stream);
    if (sdt == VIT_BF16 && ddt == VIT_F32) return launch_convert_t<bf16_t, float>(src, dst, n, stream);
    if (sdt == ddt && (sdt == VIT_F32 || sdt == VIT_BF16)) {
        VIT_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)n * (sdt == VIT_F32 ? 4 : 2), hipMemcpyDeviceToDevice, stream));
        return VIT_OK;
    }
    return vit_set_error(VIT_E_INVALID, "convert: bad dtypes %d -> %d", sdt, ddt);
}

int launch_cls_rows(const float* cls, const float* pos, void* x, int batch, int seq, int dim, int dtype,
                    hipStream_t stream) {
    const int blocks = ceil_div(batch * dim, 256);
    if (dtype == VIT_BF16)
        hipLaunchKernelGGL(cls_rows_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, cls, pos, (bf16_t*)x, batch, seq, dim);
    else
        hipLaunchKernelGGL(cls_rows_kernel<float>, dim3(blocks), dim3(256), 0, stream, cls, pos, (float*)x, batch, seq, dim);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

int launch_convert_2d(const void* src, int sdt, void* dst, int ddt, int64_t rows, int64_t cols, int64_t dst_pitch,
                      hipStream_t stream) {
    VIT_REQUIRE(rows > 0 && cols > 0 && dst_pitch >= cols, "convert_2d: bad shape");
    if (sdt == VIT_F32 && ddt == VIT_BF16) return launch_convert_2d_t<float, bf16_t>(src, dst, rows, cols, dst_pitch, stream);
    if (sdt == VIT_BF16 && ddt == VIT_F32) return launch_convert_2d_t<bf16_t, float>(src, dst, rows, cols, dst_pitch, stream);
    if (sdt == VIT_F32 && ddt == VIT_F32) return launch_convert_2d_t<float, float>(src, dst, rows, cols, dst_pitch, stream);
    if (sdt == VIT_BF16 && ddt == VIT_BF16) return launch_convert_2d_t<bf16_t, bf16_t>(src, dst, rows, cols, dst_pitch, stream);
    return vit_set_error(VIT_E_INVALID, "convert_2d: bad dtypes %d -> %d", sdt, ddt);
}
