// HBM-bound kernels of the training step (bf16 activations, fp32 statistics / gradients):
//   * transpose with zero padding (+ fused column sums = bias gradients): the weight-gradient
//     GEMMs dW = dY^T X run on the NT kernel with both operands transposed to [features, tokens]
//   * LayerNorm backward (dx with the residual-branch gradient fused in, dgamma / dbeta)
//     for nn.LayerNorm as used at modeling_vit.py:261-262,274,281,385
//   * embedding gradients (cls / position, modeling_vit.py:129-161), transposed im2col of the
//     pixels for the patch-projection weight gradient (modeling_vit.py:62-69)
//   * SGD step on fp32 master weights with bf16 re-quantisation.
#include "vit_common.h"

namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ transpose (+ column sums)
// src: [R, C] bf16 (row r lives at memory row r, or r + r/group + 1 when group > 0: the token
// rows of [B, N, D] without the CLS rows). dst: [C, Rpad] bf16, columns R..Rpad zero.
// colsum (optional): fp32 [C], += sum over r of src[r, c] (pre-zeroed by the caller).
constexpr int TT = 64;
__device__ __forceinline__ void transpose_pad_tile(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                   float* __restrict__ colsum, int R, int C, int Rpad, int group, int bx,
                                                   int by) {
    __shared__ unsigned short tile[TT][TT + 2];
    const int r0 = bx * TT, c0 = by * TT;
    const int tid = threadIdx.x;
    // load: 64 rows x 64 cols, 8 bf16 per thread per pass, 2 passes
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int lr = p * 32 + (tid >> 3), lc = (tid & 7) * 8;
        const int r = r0 + lr, c = c0 + lc;
        short8v v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r < R && c < C) {
            const size_t mr = group > 0 ? (size_t)r + r / group + 1 : (size_t)r;
            v = *(const short8v*)(src + mr * C + c);  // C % 8 == 0
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[lr][lc + e] = (unsigned short)v[e];
    }
    __syncthreads();
    // store: thread owns column lc = tid>>2 (one dst row) and 16 consecutive r (2 x 16-byte stores)
    const int lc = tid >> 2, lr0 = (tid & 3) * 16;
    const int c = c0 + lc;
    float part = 0.f;
    short8v o[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned short u = tile[lr0 + h * 8 + e][lc];
            o[h][e] = (short)u;
            part += __uint_as_float((unsigned)u << 16);
        }
    if (c < C) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = r0 + lr0 + h * 8;
            if (r < Rpad) *(short8v*)(dst + (size_t)c * Rpad + r) = o[h];  // Rpad % 8 == 0; rows >= R are zero
        }
    }
    if (colsum != nullptr) {
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        if ((tid & 3) == 0 && c < C) atomicAdd(colsum + c, part);
    }
}

__global__ __launch_bounds__(256) void transpose_pad_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                            float* __restrict__ colsum, int R, int C, int Rpad,
                                                            int group) {
    transpose_pad_tile(src, dst, colsum, R, C, Rpad, group, blockIdx.x, blockIdx.y);
}

// Several plain transposes ([R, C] -> [C, R], R % 8 == 0) in one launch: block b belongs to the table entry whose
// [first_block, first_block + tiles) range holds it (a training step refreshes 4 transposed weights per layer).
__global__ __launch_bounds__(256) void multi_transpose_kernel(const TransposeSlot* __restrict__ tab, int n) {
    int i = 0, hi = n - 1;  // uniform binary search (scalar loads): last entry with first_block <= blockIdx.x
    while (i < hi) {
        const int mid = (i + hi + 1) >> 1;
        if ((int)blockIdx.x >= tab[mid].first_block) i = mid; else hi = mid - 1;
    }
    const TransposeSlot t = tab[i];
    const int local = blockIdx.x - t.first_block, tx = (t.R + TT - 1) / TT;
    transpose_pad_tile((const bf16_t*)t.src, (bf16_t*)t.dst, nullptr, t.R, t.C, t.R, 0, local % tx, local / tx);
}

// ------------------------------------------------------------------ LayerNorm backward
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) (+ dres),  g = dy * gamma,
// dgamma += sum_rows dy * xhat,  dbeta += sum_rows dy   (fp32 atomics, one per column per block)
template <int NIT>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                            const float* __restrict__ gamma,
                                                            const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int rows, int dim, int64_t xs, int64_t dys, int64_t dxs,
                                                            float eps, float* __restrict__ partials) {
    __shared__ float red[2][4][NIT * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_in_grid = blockIdx.x * 4 + wave;
    const int nwaves = gridDim.x * 4;
    const float inv_d = 1.0f / (float)dim;
    f32x4 gam[NIT], ag[NIT], ab[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * 4;
        gam[it] = c < dim ? *(const f32x4*)(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        ag[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the next row's x / dy / residual gradient are loaded (raw) before this row's four wave reductions run
    u32x2 nx[NIT], nd[NIT], nr[NIT];
    auto fetch = [&](int row) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 4;
            nx[it] = nd[it] = nr[it] = u32x2{0u, 0u};
            if (c < dim && row < rows) {
                nx[it] = *(const u32x2*)(x + (size_t)row * xs + c);
                nd[it] = *(const u32x2*)(dy + (size_t)row * dys + c);
                if (dres != nullptr) nr[it] = *(const u32x2*)(dres + (size_t)row * dxs + c);
            }
        }
    };
    fetch(wave_in_grid);
    for (int row = wave_in_grid; row < rows; row += nwaves) {
        f32x4 xv[NIT], gv[NIT];
        u32x2 rv[NIT];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const bf16x4 a = __builtin_bit_cast(bf16x4, nx[it]), b = __builtin_bit_cast(bf16x4, nd[it]);
            rv[it] = nr[it];
#pragma unroll
            for (int e = 0; e < 4; ++e) { xv[it][e] = (float)a[e]; gv[it][e] = (float)b[e]; }
            s += (xv[it][0] + xv[it][1]) + (xv[it][2] + xv[it][3]);
        }
        fetch(row + nwaves);
        const float mean = wave_sum(s) * inv_d;
        float ss = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < dim) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = xv[it][e] - mean; ss += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(ss) * inv_d + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < dim) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (xv[it][e] - mean) * rstd;
                    const float dyv = gv[it][e];
                    ag[it][e] += dyv * xh;
                    ab[it][e] += dyv;
                    const float g = dyv * gam[it][e];
                    xv[it][e] = xh;
                    gv[it][e] = g;
                    s1 += g;
                    s2 += g * xh;
                }
            }
        }
        s1 = wave_sum(s1) * inv_d;
        s2 = wave_sum(s2) * inv_d;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < dim) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rstd * (gv[it][e] - s1 - xv[it][e] * s2);
                if (dres != nullptr) {
                    const bf16x4 r = __builtin_bit_cast(bf16x4, rv[it]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += (float)r[e];
                }
                bf16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (bf16_t)o[e];
                *(bf16x4*)(dx + (size_t)row * dxs + c) = ov;
            }
        }
    }
    // block reduction of the per-wave column partials, then one atomic per column per block
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[0][wave][(it * 64 + lane) * 4 + e] = ag[it][e];
            red[1][wave][(it * 64 + lane) * 4 + e] = ab[it][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < dim; c += 256) {
        const float g = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        const float b = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
        if (partials != nullptr) {  // [block][2][dim], summed by ln_bwd_reduce_kernel (32 atomics per column, not 2048)
            partials[((size_t)blockIdx.x * 2 + 0) * dim + c] = g;
            partials[((size_t)blockIdx.x * 2 + 1) * dim + c] = b;
        } else {
            atomicAdd(dgamma + c, g);
            atomicAdd(dbeta + c, b);
        }
    }
}

// second stage of the dgamma / dbeta reduction: grid (ceil(2*dim / 256), chunks); a block sums its chunk of
// the per-block partials for 256 columns and adds the result with one atomic per column
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ partials, int nblocks, int dim,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int col = blockIdx.x * 256 + threadIdx.x;  // 0 .. 2*dim: gamma columns then beta columns
    if (col >= 2 * dim) return;
    const int which = col >= dim ? 1 : 0, c = col - which * dim;
    const int per = (nblocks + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
    // 8 independent loads in flight per thread (a dependent chain of 64 L2 round trips took 17 us)
    const float* p = partials + (size_t)which * dim + c;
    const size_t rs = (size_t)2 * dim;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(b + u) * rs];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += v[u];
    }
    for (; b < b1; ++b) a[0] += p[(size_t)b * rs];
    const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    if (b1 > b0) atomicAdd((which ? dbeta : dgamma) + c, s);
}

// ------------------------------------------------------------------ embedding gradients
// dx0: [B, N, D] bf16. dpos[n, d] = sum_b dx0[b, n, d]; dcls[d] = dpos[0, d].
__global__ __launch_bounds__(256) void embed_grad_kernel(const bf16_t* __restrict__ dx0, float* __restrict__ dpos,
                                                         float* __restrict__ dcls, int B, int N, int D) {
    const int idx = blockIdx.x * 256 + threadIdx.x;  // over N * D / 4
    const int per_row = D / 4;
    if (idx >= N * per_row) return;
    const int n = idx / per_row, c = (idx - n * per_row) * 4;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) {
        const bf16x4 v = *(const bf16x4*)(dx0 + ((size_t)b * N + n) * D + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += (float)v[e];
    }
    *(f32x4*)(dpos + (size_t)n * D + c) = f32x4{a[0], a[1], a[2], a[3]};
    if (n == 0) *(f32x4*)(dcls + c) = f32x4{a[0], a[1], a[2], a[3]};
}

// patches^T: dst[k, r] = pixel (b, ch, py*P + i, px*P + j), k = (ch, i, j), r = (b, py, px); dst is
// [Kpad, Rpad] bf16, zero outside [kreal, R]. Pixels may be fp32 or bf16.
template <typename TP>
__global__ __launch_bounds__(256) void im2col_t_kernel(const TP* __restrict__ px, bf16_t* __restrict__ dst, int R,
                                                       int Rpad, PatchGeom g, int kpad) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (r >= Rpad) return;
    float v = 0.f;
    if (r < R && k < g.kreal) {
        const int np = g.grid * g.grid;
        const int b = r / np, rem = r - b * np, py = rem / g.grid, pxi = rem - py * g.grid;
        const int pp = g.patch * g.patch;
        const int ch = k / pp, r2 = k - ch * pp, i = r2 / g.patch, j = r2 - i * g.patch;
        v = (float)px[(((size_t)b * g.channels + ch) * g.image + (size_t)py * g.patch + i) * g.image + (size_t)pxi * g.patch + j];
    }
    dst[(size_t)k * Rpad + r] = (bf16_t)v;
}

// ------------------------------------------------------------------ SGD
// master -= lr * grad (fp32); optional bf16 copy of the updated value (row pitch handled by caller
// passing contiguous ranges).
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ master, const float* __restrict__ grad,
                                                  bf16_t* __restrict__ copy, float lr, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        f32x4 w = *(const f32x4*)(master + i);
        const f32x4 g = *(const f32x4*)(grad + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] -= lr * g[e];
        *(f32x4*)(master + i) = w;
        if (copy) {
            bf16x4 c;
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = (bf16_t)w[e];
            *(bf16x4*)(copy + i) = c;
        }
    } else {
        for (int64_t j = i; j < n; ++j) {
            const float w = master[j] - lr * grad[j];
            master[j] = w;
            if (copy) copy[j] = (bf16_t)w;
        }
    }
}

// out[i] = sum_s slabs[s][i] in a fixed order (split-K weight gradients)
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                           int splits, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;  // n % 4 == 0
    f32x4 a = *(const f32x4*)(slabs + i);
    for (int s = 1; s < splits; ++s) {
        const f32x4 b = *(const f32x4*)(slabs + (size_t)s * n + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += b[e];
    }
    *(f32x4*)(out + i) = a;
}

}  // namespace

int launch_reduce_slabs(const float* slabs, float* out, int splits, int64_t n, hipStream_t stream) {
    VIT_REQUIRE(n % 4 == 0, "reduce_slabs: n %% 4 != 0");
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, slabs, out, splits, n);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

int launch_transpose_pad(const void* src, void* dst, float* colsum, int R, int C, int Rpad, int group,
                         hipStream_t stream) {
    VIT_REQUIRE(R > 0 && C > 0 && Rpad >= R, "transpose: bad shape R=%d C=%d Rpad=%d", R, C, Rpad);
    VIT_REQUIRE(C % 8 == 0 && Rpad % 8 == 0, "transpose: C and Rpad must be multiples of 8 (C=%d Rpad=%d)", C, Rpad);
    dim3 grid(ceil_div(Rpad, TT), ceil_div(C, TT));
    hipLaunchKernelGGL(transpose_pad_kernel, grid, dim3(256), 0, stream, (const bf16_t*)src, (bf16_t*)dst, colsum, R, C,
                       Rpad, group);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

int launch_multi_transpose(const TransposeSlot* table_dev, int n, int total_blocks, hipStream_t stream) {
    if (n <= 0) return VIT_OK;
    hipLaunchKernelGGL(multi_transpose_kernel, dim3(total_blocks), dim3(256), 0, stream, table_dev, n);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

int launch_layernorm_bwd(const void* x, const void* dy, const float* gamma, const void* dres, void* dx, float* dgamma,
                         float* dbeta, int rows, int dim, int64_t xs, int64_t dys, int64_t dxs, float eps,
                         hipStream_t stream, float* partials) {
    VIT_REQUIRE(rows > 0 && dim > 0 && dim % 4 == 0 && dim <= 2048, "layernorm_bwd: bad shape rows=%d dim=%d", rows, dim);
    const int nit = ceil_div(dim, 256);
    int blocks = ceil_div(rows, 4);
    if (blocks > kLnBwdMaxBlocks) blocks = kLnBwdMaxBlocks;
#define LNB_CASE(NI)                                                                                               \
    case NI:                                                                                                       \
        hipLaunchKernelGGL(layernorm_bwd_kernel<NI>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x,         \
                           (const bf16_t*)dy, gamma, (const bf16_t*)dres, (bf16_t*)dx, dgamma, dbeta, rows, dim, xs, \
                           dys, dxs, eps, partials);                                                               \
        break;
    switch (nit) {
        LNB_CASE(1) LNB_CASE(2) LNB_CASE(3) LNB_CASE(4) LNB_CASE(5) LNB_CASE(6) LNB_CASE(7) LNB_CASE(8)
    }
#undef LNB_CASE
    VIT_CHECK_HIP(hipGetLastError());
    if (partials != nullptr) {
        hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(ceil_div(2 * dim, 256), 64), dim3(256), 0, stream, partials, blocks, dim,
                           dgamma, dbeta);
        VIT_CHECK_HIP(hipGetLastError());
    }
    return VIT_OK;
}

int launch_embed_grad(const void* dx0, float* dpos, float* dcls, int B, int N, int D, hipStream_t stream) {
    VIT_REQUIRE(D % 4 == 0, "embed_grad: D %% 4 != 0");
    hipLaunchKernelGGL(embed_grad_kernel, dim3(ceil_div(N * (D / 4), 256)), dim3(256), 0, stream, (const bf16_t*)dx0,
                       dpos, dcls, B, N, D);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

int launch_im2col_t(const void* pixels, int pixel_dtype, void* dst, int R, int Rpad, const PatchGeom* g, int kpad,
                    hipStream_t stream) {
    dim3 grid(ceil_div(Rpad, 256), kpad);
    if (pixel_dtype == VIT_F32)
        hipLaunchKernelGGL(im2col_t_kernel<float>, grid, dim3(256), 0, stream, (const float*)pixels, (bf16_t*)dst, R, Rpad,
                           *g, kpad);
    else
        hipLaunchKernelGGL(im2col_t_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)pixels, (bf16_t*)dst, R,
                           Rpad, *g, kpad);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

// fp32 master arena -> the bf16 / fp32 weight arena, every tensor in ONE launch: block (s, j) takes a stride of
// slot s (a training step used to issue one small convert per tensor, ~150 launches of a few microseconds)
__global__ __launch_bounds__(256) void multi_convert_kernel(const float* __restrict__ master, unsigned char* __restrict__ arena,
                                                            const ConvSlot* __restrict__ tab) {
    const ConvSlot s = tab[blockIdx.x];
    const float* src = master + s.src;
    const long long stride = (long long)gridDim.y * 256;
    if (s.vec4) {
        for (long long i = ((long long)blockIdx.y * 256 + threadIdx.x) * 4; i < s.n; i += stride * 4) {
            const f32x4 v = *(const f32x4*)(src + i);
            if (s.dtype == VIT_BF16) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
                *(bf16x4*)(arena + s.dst + i * 2) = o;
            } else {
                *(f32x4*)(arena + s.dst + i * 4) = v;
            }
        }
    } else {
        for (long long i = (long long)blockIdx.y * 256 + threadIdx.x; i < s.n; i += stride) {
            if (s.dtype == VIT_BF16) ((bf16_t*)(arena + s.dst))[i] = (bf16_t)src[i];
            else ((float*)(arena + s.dst))[i] = src[i];
        }
    }
}

int launch_multi_convert(const float* master, void* arena, const ConvSlot* table_dev, int nslots, hipStream_t stream) {
    if (nslots <= 0) return VIT_OK;
    hipLaunchKernelGGL(multi_convert_kernel, dim3(nslots, 32), dim3(256), 0, stream, master, (unsigned char*)arena, table_dev);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

int launch_sgd(float* master, const float* grad, void* copy_bf16, float lr, int64_t n, hipStream_t stream) {
    if (n <= 0) return VIT_OK;
    const int64_t blocks = (n / 4 + 256) / 256;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, master, grad, (bf16_t*)copy_bf16, lr, n);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}
