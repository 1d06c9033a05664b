// Image preprocessing on the GPU: decoded uint8 HWC image -> pixel_values.
//
// This is the step the remote VLM servers run after the reference hands them an encoded image
// (Vlm_coder.py:24-26, :205, :381; vlm_cad.py:40, :171). The arithmetic specification is the
// in-container `transformers` PIL pipeline (image_processing_backends.py:619-653: resize ->
// center_crop -> rescale -> normalize; image_transforms.py:89-126, :313-381, :384-442, :445-526),
// whose resize is Pillow's ImagingResample (src/libImaging/Resample.c, Pillow 12.2.0): a separable
// convolution with antialiasing support, 8-bit fixed point (22 fractional bits), horizontal pass
// first, the intermediate rounded to uint8. Everything here is integer / table work and is
// BIT-EXACT against that pipeline (tests/test_gpu_preprocess.py); it is HBM-bound byte traffic and
// deliberately not shaped into a GEMM.
//
//  * vit_preproc_create (not on the hot path) computes the integer tap tables of both passes on
//    the host in double precision exactly as Resample.c does, plus a 256-entry table per channel
//    of ((v * rescale_factor) - mean) / std evaluated with numpy's operation order, and uploads them.
//  * pass H: one workgroup stages ROWS_H input rows in LDS with dword loads and produces the
//    resized-width rows (uint8, row pitch padded to 16 B) -- only the rows / columns the crop needs.
//  * pass V: one workgroup per ROWS_V output rows; a lane owns 4 adjacent pixels x C channels (dword
//    loads of the intermediate), applies the vertical taps, clips, looks the value up in the LDS
//    copy of the table and writes planar [B, C, H, W] fp32 / bf16 with coalesced stores.
#include "vit_common.h"

#include <math.h>
#include <string.h>

#include <vector>

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;  // Resample.c
constexpr int ROWS_H = 4;

struct Taps {
    int ksize = 0;
    std::vector<int> bounds;  // [out][2]: first input index, tap count
    std::vector<int> coef;    // [out][ksize], fixed point
};

double filt_box(double x) { return (x > -0.5 && x <= 0.5) ? 1.0 : 0.0; }
double filt_bilinear(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}
double filt_hamming(double x) {
    if (x < 0.0) x = -x;
    if (x == 0.0) return 1.0;
    if (x >= 1.0) return 0.0;
    x = x * M_PI;
    return sin(x) / x * (0.54 + 0.46 * cos(x));
}
double filt_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
double sinc(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}
double filt_lanczos(double x) {
    if (-3.0 <= x && x < 3.0) return sinc(x) * sinc(x / 3);
    return 0.0;
}

// precompute_coeffs + normalize_coeffs_8bpc of Resample.c for the whole-image box (in0 = 0, in1 = in_size)
bool make_taps(int in_size, int out_size, int resample, Taps* t) {
    double (*filter)(double);
    double fsupport;
    switch (resample) {
        case VIT_RESAMPLE_LANCZOS: filter = filt_lanczos; fsupport = 3.0; break;
        case VIT_RESAMPLE_BILINEAR: filter = filt_bilinear; fsupport = 1.0; break;
        case VIT_RESAMPLE_BICUBIC: filter = filt_bicubic; fsupport = 2.0; break;
        case VIT_RESAMPLE_BOX: filter = filt_box; fsupport = 0.5; break;
        case VIT_RESAMPLE_HAMMING: filter = filt_hamming; fsupport = 1.0; break;
        default: return false;
    }
    double scale, filterscale;
    filterscale = scale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = fsupport * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    t->ksize = ksize;
    t->bounds.assign((size_t)out_size * 2, 0);
    t->coef.assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (int x = 0; x < xmax; ++x) {
            const double v = k[x] * (1 << PRECISION_BITS);
            t->coef[(size_t)xx * ksize + x] = (k[x] < 0) ? (int)(-0.5 + v) : (int)(0.5 + v);
        }
        t->bounds[xx * 2 + 0] = xmin;
        t->bounds[xx * 2 + 1] = xmax;
    }
    return true;
}

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

struct HParams {
    const unsigned char* in;  // [B, in_h, in_w, C]
    unsigned char* tmp;       // [B, nrows, tpitch]; row r = input row y0 + r, byte (xo - x0) * C + c
    const int* bounds;
    const int* coef;
    int ksize;
    int in_h, in_w, C;
    int y0, nrows;  // input rows the vertical pass reads
    int x0, ncols;  // resized columns the crop keeps
    int tpitch;
    size_t in_bytes;  // total bytes of `in` (dword loads never cross it)
};

template <int C>
__global__ __launch_bounds__(256) void resample_h_kernel(const HParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * ROWS_H;
    const int rows = min(ROWS_H, P.nrows - r0);
    const int rowbytes = P.in_w * C;
    const int lpitch = (rowbytes + 3 + 3) & ~3;  // staged row: up to 3 lead bytes + row, dword multiple
    unsigned int* sm32 = (unsigned int*)smem;

    for (int r = 0; r < rows; ++r) {
        const size_t g0 = ((size_t)b * P.in_h + P.y0 + r0 + r) * rowbytes;
        const size_t a0 = g0 & ~(size_t)3;
        const int nd = (int)((g0 + rowbytes - a0 + 3) >> 2);
        for (int d = tid; d < nd; d += 256) {
            const size_t ga = a0 + (size_t)d * 4;
            unsigned int v;
            if (ga + 4 <= P.in_bytes) {
                v = *(const unsigned int*)(P.in + ga);
            } else {
                v = 0;
                for (int e = 0; e < 4; ++e)
                    if (ga + e < P.in_bytes) v |= (unsigned int)P.in[ga + e] << (8 * e);
            }
            sm32[(r * lpitch >> 2) + d] = v;
        }
    }
    __syncthreads();

    // work item = 4 adjacent output pixels of one row: the taps of a pixel are loaded once for its C
    // channels, and the 4*C result bytes leave as C dword stores (tmp rows are 16-B pitched and hold
    // whole 4-pixel groups, so the stores are aligned and inside the pitch)
    const int groups = (P.ncols + 3) >> 2;
    for (int it = tid; it < rows * groups; it += 256) {
        const int r = it / groups, g = it - r * groups;
        const size_t g0 = ((size_t)b * P.in_h + P.y0 + r0 + r) * rowbytes;
        const unsigned char* row = smem + r * lpitch + (int)(g0 & 3);
        unsigned int val[4 * C];  // result bytes of the group, in tmp order (pixel-major, channel-minor)
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const int xo = g * 4 + px;
            const int xx = P.x0 + (xo < P.ncols ? xo : P.ncols - 1);  // tail pixels recompute the last one
            const int xmin = P.bounds[2 * xx], n = P.bounds[2 * xx + 1];
            const int* k = P.coef + (size_t)xx * P.ksize;
            const unsigned char* src = row + xmin * C;
            int ss[C];
#pragma unroll
            for (int c = 0; c < C; ++c) ss[c] = 1 << (PRECISION_BITS - 1);
            for (int x = 0; x < n; ++x) {
                const int kx = k[x];
#pragma unroll
                for (int c = 0; c < C; ++c) ss[c] += (int)src[x * C + c] * kx;
            }
#pragma unroll
            for (int c = 0; c < C; ++c) val[px * C + c] = (unsigned int)clip8(ss[c]);
        }
        unsigned int packed[C];
#pragma unroll
        for (int d = 0; d < C; ++d)
            packed[d] = val[4 * d] | (val[4 * d + 1] << 8) | (val[4 * d + 2] << 16) | (val[4 * d + 3] << 24);
        unsigned int* dst = (unsigned int*)(P.tmp + ((size_t)b * P.nrows + r0 + r) * P.tpitch + (size_t)g * 4 * C);
#pragma unroll
        for (int d = 0; d < C; ++d) dst[d] = packed[d];
    }
}

struct VParams {
    const unsigned char* tmp;
    void* out;  // [B, C, out_h, out_w]
    const int* bounds;
    const int* coef;
    const float* lut;  // [C][256]
    int ksize;
    int C;
    int y0, nrows, x0, ncols, tpitch;
    int rs_h, rs_w;     // resized image
    int top, left;      // crop origin inside the resized image (may be negative: zero padding)
    int out_h, out_w;
};

constexpr int ROWS_V = 4;  // output rows per workgroup

template <typename T, int C>
__global__ __launch_bounds__(256) void resample_v_kernel(const VParams P) {
    __shared__ float lut[C * 256];
    const int tid = threadIdx.x;
    for (int i = tid; i < C * 256; i += 256) lut[i] = P.lut[i];
    __syncthreads();

    const int y0 = blockIdx.x * ROWS_V, b = blockIdx.y;
    const int rows = min(ROWS_V, P.out_h - y0);
    const size_t plane = (size_t)P.out_h * P.out_w;
    T* outb = (T*)P.out + (size_t)b * C * plane;

    // work item = 4 adjacent pixels x C channels (C dwords of the intermediate) of one output row
    const int groups = (P.ncols + 3) >> 2;
    for (int it = tid; it < rows * groups; it += 256) {
        const int r = it / groups, g = it - r * groups;
        const int y = y0 + r, sy = y + P.top;
        if (sy < 0 || sy >= P.rs_h) continue;  // padding row, filled below
        const int ymin = P.bounds[2 * sy], n = P.bounds[2 * sy + 1];
        const int* k = P.coef + (size_t)sy * P.ksize;
        const unsigned char* p = P.tmp + ((size_t)b * P.nrows + (ymin - P.y0)) * P.tpitch + (size_t)g * 4 * C;
        int acc[4 * C];
#pragma unroll
        for (int e = 0; e < 4 * C; ++e) acc[e] = 1 << (PRECISION_BITS - 1);
        for (int t = 0; t < n; ++t) {
            const int kt = k[t];
            const unsigned int* pw = (const unsigned int*)(p + (size_t)t * P.tpitch);
#pragma unroll
            for (int d = 0; d < C; ++d) {
                const unsigned int w = pw[d];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * d + e] += (int)((w >> (8 * e)) & 255u) * kt;
            }
        }
        T* out = outb + (size_t)y * P.out_w;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const int xo = g * 4 + px;            // column inside the kept range
            const int x = P.x0 + xo - P.left;     // output column
            if (xo < P.ncols && x >= 0 && x < P.out_w) {
#pragma unroll
                for (int c = 0; c < C; ++c) out[c * plane + x] = (T)lut[c * 256 + clip8(acc[px * C + c])];
            }
        }
    }
    // padding (image smaller than the crop, image_transforms.py:505-524): whole rows above / below the
    // resized image, columns left / right of it
    const int xl = -P.left;            // first output column that has data (if > 0)
    const int xr = P.rs_w - P.left;    // one past the last output column with data
    const bool any_row_pad = (y0 + P.top < 0) || (y0 + rows - 1 + P.top >= P.rs_h);
    if (any_row_pad || xl > 0 || xr < P.out_w) {
        for (int i = tid; i < rows * C * P.out_w; i += 256) {
            const int r = i / (C * P.out_w), rem = i - r * (C * P.out_w);
            const int c = rem / P.out_w, x = rem - c * P.out_w;
            const int sy = y0 + r + P.top;
            if (sy < 0 || sy >= P.rs_h || x < xl || x >= xr) outb[c * plane + (size_t)(y0 + r) * P.out_w + x] = (T)lut[c * 256];
        }
    }
}

int floordiv2(int a) { return (a >= 0) ? a / 2 : -((-a + 1) / 2); }

}  // namespace

struct vit_preproc {
    vit_preproc_config cfg;
    int device;
    int rs_h, rs_w, out_h, out_w, top, left;
    int y0, nrows, x0, ncols, tpitch;
    int kh, kv;
    unsigned char* dev = nullptr;  // one allocation: bounds_h | coef_h | bounds_v | coef_v | lut
    int *bounds_h, *coef_h, *bounds_v, *coef_v;
    float* lut;
};

extern "C" int vit_preproc_create(const vit_preproc_config* cfg, vit_preproc** out) {
    VIT_REQUIRE(cfg && out, "vit_preproc_create: null argument");
    *out = nullptr;
    const vit_preproc_config& c = *cfg;
    VIT_REQUIRE(c.in_height > 0 && c.in_width > 0, "vit_preproc_create: empty input image");
    // (4-channel input is not plain resampling in the specification: Image.resize premultiplies alpha)
    VIT_REQUIRE(c.channels == 1 || c.channels == 3, "vit_preproc_create: channels must be 1 or 3 (got %d)", c.channels);
    VIT_REQUIRE(c.resize_height >= 0 && c.resize_width >= 0 && (c.resize_height > 0) == (c.resize_width > 0),
                "vit_preproc_create: resize_height / resize_width must both be set or both be 0");
    VIT_REQUIRE(c.crop_height >= 0 && c.crop_width >= 0 && (c.crop_height > 0) == (c.crop_width > 0),
                "vit_preproc_create: crop_height / crop_width must both be set or both be 0");
    VIT_REQUIRE((size_t)ROWS_H * (((size_t)c.in_width * c.channels + 6) & ~(size_t)3) <= 65536,
                "vit_preproc_create: input rows wider than %d bytes are unsupported", 65536 / ROWS_H - 8);
    auto P = new vit_preproc();
    P->cfg = c;
    P->rs_h = c.resize_height > 0 ? c.resize_height : c.in_height;
    P->rs_w = c.resize_width > 0 ? c.resize_width : c.in_width;
    P->out_h = c.crop_height > 0 ? c.crop_height : P->rs_h;
    P->out_w = c.crop_width > 0 ? c.crop_width : P->rs_w;
    // center_crop (image_transforms.py:492-497): top = (h - crop_h) // 2 with Python floor division;
    // a negative origin is the zero-padding branch (:505-524), same mapping.
    P->top = floordiv2(P->rs_h - P->out_h);
    P->left = floordiv2(P->rs_w - P->out_w);

    Taps th, tv;
    if (!make_taps(c.in_width, P->rs_w, c.resample, &th) || !make_taps(c.in_height, P->rs_h, c.resample, &tv)) {
        delete P;
        return vit_set_error(VIT_E_INVALID, "vit_preproc_create: unsupported resample filter %d (PIL codes 1..5)",
                             c.resample);
    }
    P->kh = th.ksize;
    P->kv = tv.ksize;
    // resized rows / columns the crop keeps, and the input rows their vertical taps touch
    const int sy0 = P->top > 0 ? P->top : 0, sy1 = (P->top + P->out_h < P->rs_h) ? P->top + P->out_h : P->rs_h;
    const int sx0 = P->left > 0 ? P->left : 0, sx1 = (P->left + P->out_w < P->rs_w) ? P->left + P->out_w : P->rs_w;
    int ylo = c.in_height, yhi = 0;
    for (int sy = sy0; sy < sy1; ++sy) {
        const int a = tv.bounds[2 * sy], n = tv.bounds[2 * sy + 1];
        if (a < ylo) ylo = a;
        if (a + n > yhi) yhi = a + n;
    }
    if (yhi <= ylo) { ylo = 0; yhi = 1; }
    P->y0 = ylo;
    P->nrows = yhi - ylo;
    P->x0 = sx0;
    P->ncols = sx1 > sx0 ? sx1 - sx0 : 1;
    P->tpitch = (((P->ncols + 3) & ~3) * c.channels + 15) & ~15;  // whole 4-pixel groups, 16-B rows

    // ((v * rescale_factor) - mean) / std with numpy's operation order (image_transforms.py:120-124,
    // :419-439): float64 multiply, cast to float32, float32 subtract, float32 divide.
    std::vector<float> lut((size_t)c.channels * 256);
    for (int ch = 0; ch < c.channels; ++ch)
        for (int v = 0; v < 256; ++v) {
            float f = c.do_rescale ? (float)((double)v * c.rescale_factor) : (float)v;
            if (c.do_normalize) {
                const float d = f - c.image_mean[ch];
                f = d / c.image_std[ch];
            }
            lut[(size_t)ch * 256 + v] = f;
        }

    const size_t b_bh = th.bounds.size() * 4, b_ch = th.coef.size() * 4, b_bv = tv.bounds.size() * 4,
                 b_cv = tv.coef.size() * 4, b_lut = lut.size() * 4;
    size_t off[5], cur = 0;
    const size_t sz[5] = {b_bh, b_ch, b_bv, b_cv, b_lut};
    for (int i = 0; i < 5; ++i) { off[i] = cur; cur = (cur + sz[i] + 255) & ~(size_t)255; }
    hipError_t e = hipGetDevice(&P->device);
    if (e == hipSuccess) e = hipMalloc((void**)&P->dev, cur);
    if (e != hipSuccess) {
        delete P;
        return vit_set_error(VIT_E_NOMEM, "vit_preproc_create: hipMalloc(%zu) failed: %s", cur, hipGetErrorString(e));
    }
    const void* src[5] = {th.bounds.data(), th.coef.data(), tv.bounds.data(), tv.coef.data(), lut.data()};
    for (int i = 0; i < 5 && e == hipSuccess; ++i) e = hipMemcpy(P->dev + off[i], src[i], sz[i], hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(P->dev);
        delete P;
        return vit_set_error(VIT_E_HIP, "vit_preproc_create: table upload failed: %s", hipGetErrorString(e));
    }
    P->bounds_h = (int*)(P->dev + off[0]);
    P->coef_h = (int*)(P->dev + off[1]);
    P->bounds_v = (int*)(P->dev + off[2]);
    P->coef_v = (int*)(P->dev + off[3]);
    P->lut = (float*)(P->dev + off[4]);
    *out = P;
    return VIT_OK;
}

extern "C" int vit_preproc_destroy(vit_preproc* p) {
    if (!p) return VIT_OK;
    if (p->dev) (void)hipFree(p->dev);
    delete p;
    return VIT_OK;
}

extern "C" int vit_preproc_output_size(const vit_preproc* p, int* height, int* width) {
    VIT_REQUIRE(p && height && width, "vit_preproc_output_size: null argument");
    *height = p->out_h;
    *width = p->out_w;
    return VIT_OK;
}

extern "C" size_t vit_preproc_workspace_bytes(const vit_preproc* p, int batch) {
    if (!p || batch <= 0) return 0;
    return ((size_t)batch * p->nrows * p->tpitch + 255) & ~(size_t)255;
}

extern "C" int vit_preproc_run(vit_preproc* p, const void* images_dev, int batch, void* pixel_values_out, int out_dtype,
                               void* workspace, size_t workspace_bytes, void* stream_) {
    VIT_REQUIRE(p && images_dev && pixel_values_out && workspace, "vit_preproc_run: null argument");
    VIT_REQUIRE(batch > 0 && batch <= 65535, "vit_preproc_run: batch must be in [1, 65535] (got %d)", batch);
    VIT_REQUIRE(out_dtype == VIT_F32 || out_dtype == VIT_BF16, "vit_preproc_run: bad output dtype %d", out_dtype);
    VIT_REQUIRE(((uintptr_t)images_dev & 3) == 0, "vit_preproc_run: images must be 4-byte aligned");
    VIT_REQUIRE(((uintptr_t)workspace & 15) == 0, "vit_preproc_run: workspace must be 16-byte aligned");
    const size_t need = vit_preproc_workspace_bytes(p, batch);
    if (workspace_bytes < need)
        return vit_set_error(VIT_E_NOMEM, "vit_preproc_run: workspace %zu < required %zu bytes", workspace_bytes, need);
    int dev;
    VIT_CHECK_HIP(hipGetDevice(&dev));
    if (dev != p->device)
        return vit_set_error(VIT_E_STATE, "vit_preproc_run: plan lives on device %d, current device is %d", p->device, dev);
    hipStream_t stream = (hipStream_t)stream_;
    const vit_preproc_config& c = p->cfg;

    HParams H{};
    H.in = (const unsigned char*)images_dev;
    H.tmp = (unsigned char*)workspace;
    H.bounds = p->bounds_h; H.coef = p->coef_h; H.ksize = p->kh;
    H.in_h = c.in_height; H.in_w = c.in_width; H.C = c.channels;
    H.y0 = p->y0; H.nrows = p->nrows; H.x0 = p->x0; H.ncols = p->ncols; H.tpitch = p->tpitch;
    H.in_bytes = (size_t)batch * c.in_height * c.in_width * c.channels;
    const int lpitch = (c.in_width * c.channels + 3 + 3) & ~3;
    if (c.channels == 1)
        hipLaunchKernelGGL(resample_h_kernel<1>, dim3(ceil_div(p->nrows, ROWS_H), batch), dim3(256),
                           (size_t)ROWS_H * lpitch, stream, H);
    else
        hipLaunchKernelGGL(resample_h_kernel<3>, dim3(ceil_div(p->nrows, ROWS_H), batch), dim3(256),
                           (size_t)ROWS_H * lpitch, stream, H);
    VIT_CHECK_HIP(hipGetLastError());

    VParams V{};
    V.tmp = (const unsigned char*)workspace;
    V.out = pixel_values_out;
    V.bounds = p->bounds_v; V.coef = p->coef_v; V.lut = p->lut; V.ksize = p->kv; V.C = c.channels;
    V.y0 = p->y0; V.nrows = p->nrows; V.x0 = p->x0; V.ncols = p->ncols; V.tpitch = p->tpitch;
    V.rs_h = p->rs_h; V.rs_w = p->rs_w; V.top = p->top; V.left = p->left; V.out_h = p->out_h; V.out_w = p->out_w;
    dim3 grid(ceil_div(p->out_h, ROWS_V), batch), block(256);
#define PRE_V(T, CH) hipLaunchKernelGGL((resample_v_kernel<T, CH>), grid, block, 0, stream, V)
    if (out_dtype == VIT_F32) {
        if (c.channels == 1) PRE_V(float, 1); else PRE_V(float, 3);
    } else {
        if (c.channels == 1) PRE_V(bf16_t, 1); else PRE_V(bf16_t, 3);
    }
#undef PRE_V
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}
