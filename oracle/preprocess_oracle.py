"""CPU oracle for image preprocessing and the pooled-row heads -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file; ``vit_amd`` never does.

PARITY UNPINNED BY THE REFERENCE: the reference sends an encoded image file to a remote model
server (Vlm_coder.py:24-26, :205, :381; vlm_cad.py:40, :171) and holds no preprocessing code,
tests or vectors (SURVEY.md sections 0, 8c, 8f). The algorithm is restated from the third-party
pipeline the survey names as the specification:

* ``transformers`` 5.15.0 PIL backend: ``image_processing_backends.py:619-653`` (resize ->
  center_crop -> rescale -> normalize), ``image_transforms.py:89-126`` (rescale), ``:246-310``
  (shortest-edge output size), ``:313-381`` (resize through PIL), ``:384-442`` (normalize),
  ``:445-526`` (center_crop incl. zero padding);
* Pillow 12.2.0 ``src/libImaging/Resample.c`` (``precompute_coeffs``, ``normalize_coeffs_8bpc``,
  ``ImagingResampleHorizontal_8bpc`` / ``Vertical_8bpc``): separable convolution whose support
  grows with the down-scale factor, coefficients normalised in double then rounded to 22-bit
  fixed point, accumulators start at 2^21, results clipped to uint8 after EACH pass, horizontal
  pass first.

``oracle/gen_golden_preprocess.py`` pins this restatement against those two packages (bit-exact)
in the build container and writes ``tests/golden/preprocess_*.npz``.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2

LANCZOS, BILINEAR, BICUBIC, BOX, HAMMING = 1, 2, 3, 4, 5  # PIL.Image.Resampling codes


def _box(x):
    return 1.0 if -0.5 < x <= 0.5 else 0.0


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _hamming(x):
    x = abs(x)
    if x == 0.0:
        return 1.0
    if x >= 1.0:
        return 0.0
    x = x * math.pi
    return math.sin(x) / x * (0.54 + 0.46 * math.cos(x))


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    return _sinc(x) * _sinc(x / 3) if -3.0 <= x < 3.0 else 0.0


_FILTERS = {LANCZOS: (_lanczos, 3.0), BILINEAR: (_bilinear, 1.0), BICUBIC: (_bicubic, 2.0), BOX: (_box, 0.5),
            HAMMING: (_hamming, 1.0)}


def _trunc(v):  # C's (int) cast
    return int(v)


def precompute_coeffs(in_size, out_size, resample):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the whole-image box.
    Returns (bounds [out,2] int, coef [out,ksize] int)."""
    filt, fsupport = _FILTERS[resample]
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    coef = np.zeros((out_size, ksize), dtype=np.int64)
    for xx in range(out_size):
        center = 0 + (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(_trunc(center - support + 0.5), 0)
        xmax = min(_trunc(center + support + 0.5), in_size) - xmin
        k = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        for x, w in enumerate(k):
            v = w * (1 << PRECISION_BITS)
            coef[xx, x] = _trunc(-0.5 + v) if w < 0 else _trunc(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, coef


def _pass(img, bounds, coef, axis):
    """One 8-bit pass along `axis` of an HWC uint8 array."""
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + img.shape[1:], dtype=np.uint8)
    for xx in range(bounds.shape[0]):
        lo, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        k = coef[xx, :n].reshape((n,) + (1,) * (img.ndim - 1))
        ss = (1 << (PRECISION_BITS - 1)) + (img[lo:lo + n] * k).sum(axis=0)
        out[xx] = np.clip(ss >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize(img, height, width, resample):
    """Image.resize((width, height), resample) for an HWC uint8 array (modes L / RGB)."""
    h, w = img.shape[:2]
    out = img
    if width != w:
        out = _pass(out, *precompute_coeffs(w, width, resample), axis=1)
    if height != h:
        out = _pass(out, *precompute_coeffs(h, height, resample), axis=0)
    return out


def shortest_edge_size(height, width, size):
    """get_resize_output_image_size(..., default_to_square=False) (image_transforms.py:295-310)."""
    short, long = (width, height) if width <= height else (height, width)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if width <= height else (new_short, new_long)


def center_crop(img, crop_h, crop_w):
    """image_transforms.py:445-526 on an HWC array: crop, zero-padding when the image is smaller."""
    h, w = img.shape[:2]
    top, left = (h - crop_h) // 2, (w - crop_w) // 2
    out = np.zeros((crop_h, crop_w) + img.shape[2:], dtype=img.dtype)
    for y in range(crop_h):
        sy = y + top
        if 0 <= sy < h:
            x0, x1 = max(0, -left), min(crop_w, w - left)
            if x1 > x0:
                out[y, x0:x1] = img[sy, x0 + left:x1 + left]
    return out


def preprocess(img, *, do_resize=True, size=None, resample=BILINEAR, do_center_crop=False, crop_size=None,
               do_rescale=True, rescale_factor=1 / 255, do_normalize=True, image_mean=(0.5, 0.5, 0.5),
               image_std=(0.5, 0.5, 0.5)):
    """One HWC uint8 image -> CHW float32 pixel_values (image_processing_backends.py:619-653).
    `size` is {"height","width"} or {"shortest_edge"}; `crop_size` is {"height","width"}."""
    assert img.dtype == np.uint8 and img.ndim == 3
    if do_resize:
        if "shortest_edge" in size:
            nh, nw = shortest_edge_size(img.shape[0], img.shape[1], size["shortest_edge"])
        else:
            nh, nw = size["height"], size["width"]
        img = pil_resize(img, nh, nw, resample)
    if do_center_crop:
        img = center_crop(img, crop_size["height"], crop_size["width"])
    x = img
    if do_rescale:
        x = (x.astype(np.float64) * rescale_factor).astype(np.float32)
    if do_normalize:
        x = x.astype(np.float32)
        c = x.shape[2]
        x = (x - np.array(image_mean[:c], dtype=np.float32)) / np.array(image_std[:c], dtype=np.float32)
    return np.ascontiguousarray(np.moveaxis(x, 2, 0))


# ---- heads on the pooled row ----
def vit_pooler(hidden, weight, bias):
    """ViTPooler.forward (modeling_vit.py:289-301): tanh(dense(hidden[:, 0]))."""
    import torch
    return torch.tanh(hidden[:, 0].float() @ weight.float().t() + bias.float())


def visual_projection(pooled, weight):
    """CLIPVisionModelWithProjection (modeling_clip.py:907, :950): bias-free Linear on pooler_output."""
    return pooled.float() @ weight.float().t()
