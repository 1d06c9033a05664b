// Position-embedding interpolation for inputs of another resolution:
// ViTEmbeddings.interpolate_pos_encoding (modeling_vit.py:89-127) /
// CLIPVisionEmbeddings.interpolate_pos_encoding (modeling_clip.py:161-200): the CLS row is kept,
// the [g, g, D] patch rows are resized with nn.functional.interpolate(mode="bicubic",
// align_corners=False): source coordinate (dst + 0.5) * in/out - 0.5, four taps per axis with
// clamped indices, cubic-convolution weights with A = -0.75, no antialiasing.
// Runs once per new resolution (vit_clone_resized), not per forward: one lane per (position, 4
// channels), fp32 in / fp32 out, coalesced along D.
#include "vit_ctx.h"

namespace {

__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
    const float A = -0.75f;
    const float x1 = t, x2 = 1.0f - t;
    c[0] = ((A * (x1 + 1.0f) - 5.0f * A) * (x1 + 1.0f) + 8.0f * A) * (x1 + 1.0f) - 4.0f * A;
    c[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
    c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    c[3] = ((A * (x2 + 1.0f) - 5.0f * A) * (x2 + 1.0f) + 8.0f * A) * (x2 + 1.0f) - 4.0f * A;
}

__global__ __launch_bounds__(256) void pos_interp_kernel(const float* __restrict__ src, float* __restrict__ dst, int D,
                                                         int gs, int gh, int gw, float sy, float sx) {
    const int pos = blockIdx.x;  // 0 = CLS, 1 + oy * gw + ox
    if (pos == 0) {
        for (int d = threadIdx.x; d < D; d += 256) dst[d] = src[d];
        return;
    }
    const int oy = (pos - 1) / gw, ox = (pos - 1) - oy * gw;
    const float ry = sy * (oy + 0.5f) - 0.5f, rx = sx * (ox + 0.5f) - 0.5f;
    const float fy = floorf(ry), fx = floorf(rx);
    float cy[4], cx[4];
    cubic_coeffs(fminf(fmaxf(ry - fy, 0.f), 1.f), cy);
    cubic_coeffs(fminf(fmaxf(rx - fx, 0.f), 1.f), cx);
    const int iy = (int)fy, ix = (int)fx;
    int yi[4], xi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        yi[k] = max(min(iy - 1 + k, gs - 1), 0);
        xi[k] = max(min(ix - 1 + k, gs - 1), 0);
    }
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float row = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) row += src[(size_t)(1 + yi[i] * gs + xi[j]) * D + d] * cx[j];
            acc += row * cy[i];
        }
        dst[(size_t)pos * D + d] = acc;
    }
}

}  // namespace

int launch_pos_interp(const float* src, float* dst, int D, int gs, int gh, int gw, hipStream_t stream) {
    VIT_REQUIRE(D > 0 && gs > 0 && gh > 0 && gw > 0, "pos_embed_interpolate: non-positive size");
    // area_pixel_compute_scale: (float)input / output when only `size=` is given
    const float sy = (float)gs / (float)gh, sx = (float)gs / (float)gw;
    hipLaunchKernelGGL(pos_interp_kernel, dim3(1 + gh * gw), dim3(256), 0, stream, src, dst, D, gs, gh, gw, sy, sx);
    VIT_CHECK_HIP(hipGetLastError());
    return VIT_OK;
}

extern "C" int vit_pos_embed_interpolate(const float* src, float* dst, int dim, int src_grid, int dst_grid_h,
                                         int dst_grid_w, void* stream) {
    VIT_REQUIRE(src && dst, "vit_pos_embed_interpolate: null argument");
    return launch_pos_interp(src, dst, dim, src_grid, dst_grid_h, dst_grid_w, (hipStream_t)stream);
}

extern "C" int vit_clone_resized(vit_ctx* src, int image_size, vit_ctx** out, void* stream_) {
    VIT_REQUIRE(src && out, "vit_clone_resized: null argument");
    *out = nullptr;
    const char* missing = nullptr;
    const int nmiss = vit_missing_weights(src, &missing);
    if (nmiss) return vit_set_error(VIT_E_STATE, "vit_clone_resized: %d weights not loaded (e.g. '%s')", nmiss, missing);
    VIT_REQUIRE(image_size > 0 && image_size % src->cfg.patch_size == 0,
                "vit_clone_resized: image_size %d is not a multiple of the patch size %d", image_size, src->cfg.patch_size);
    vit_config c = src->cfg;
    c.image_size = image_size;
    vit_ctx* dst = nullptr;
    int rc = vit_create(&c, &dst);
    if (rc != VIT_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    const int gs = src->cfg.image_size / src->cfg.patch_size, gd = image_size / c.patch_size;
    for (auto& kv : dst->slots) {
        Slot& d = kv.second;
        const Slot& s = src->slots.at(kv.first);
        const bool is_pos = kv.first == "embeddings.position_embeddings" || kv.first == "embeddings.position_embedding.weight";
        if (is_pos) {
            rc = launch_pos_interp((const float*)(src->arena + s.offset), (float*)(dst->arena + d.offset), c.hidden_size, gs,
                                   gd, gd, stream);
            if (rc != VIT_OK) break;
        } else {
            const size_t bytes = (size_t)s.rows * s.pitch * esize(s.dtype);
            hipError_t e = hipMemcpyAsync(dst->arena + d.offset, src->arena + s.offset, bytes, hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) {
                rc = vit_set_error(VIT_E_HIP, "vit_clone_resized: copy of '%s' failed: %s", kv.first.c_str(), hipGetErrorString(e));
                break;
            }
        }
        d.loaded = true;
    }
    if (rc != VIT_OK) {
        vit_destroy(dst);
        return rc;
    }
    dst->n_missing = 0;
    *out = dst;
    return VIT_OK;
}
