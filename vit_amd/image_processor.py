"""GPU image preprocessing with the call surface of transformers' image processors
(``ViTImageProcessor`` / ``CLIPImageProcessor``: same keyword names, ``pixel_values`` out).

This is the step the remote model servers perform on the file the reference's scripts send them
(Vlm_coder.py:24-26, :205, :381; vlm_cad.py:40, :171); with a local encoder it runs here, on the
GPU, directly in front of ``ViTEncoder``. The arithmetic (Pillow's fixed-point resampling, centre
crop, rescale, normalise) is done by ``vit_preproc_run`` in libvit_amd.so and is bit-exact against
the transformers PIL pipeline (image_processing_backends.py:619-653). Decoding a PNG/JPEG file is
not part of it: images arrive as decoded uint8 ``[H, W, C]`` arrays.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .ops import dtype_code

# PIL.Image.Resampling codes (NEAREST = 0 is a different algorithm in Pillow and is not offered)
LANCZOS, BILINEAR, BICUBIC, BOX, HAMMING = 1, 2, 3, 4, 5

IMAGENET_STANDARD_MEAN = (0.5, 0.5, 0.5)
IMAGENET_STANDARD_STD = (0.5, 0.5, 0.5)
OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class _Plan:
    """One vit_preproc handle (tap tables for one input geometry) + its workspace."""

    def __init__(self, cfg: "_lib.vit_preproc_config", device: torch.device):
        self.handle = ctypes.c_void_p()
        self.device = device
        with torch.cuda.device(device):
            _lib.check(_lib.load().vit_preproc_create(ctypes.byref(cfg), ctypes.byref(self.handle)))
        h, w = ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.load().vit_preproc_output_size(self.handle, ctypes.byref(h), ctypes.byref(w)))
        self.out_hw = (h.value, w.value)
        self.ws: Optional[torch.Tensor] = None

    def workspace(self, batch: int) -> torch.Tensor:
        need = int(_lib.load().vit_preproc_workspace_bytes(self.handle, batch))
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self.ws

    def close(self):
        if self.handle:
            _lib.load().vit_preproc_destroy(self.handle)
            self.handle = None


class ImageProcessor:
    """``ImageProcessor(**kwargs)(images)`` -> ``{"pixel_values": Tensor[B, C, H, W]}`` on the GPU.

    Keyword names and defaults are those of ``ViTImageProcessor`` (models/vit/image_processing_pil_vit.py):
    resize to 224x224 bilinear, rescale 1/255, normalise with mean = std = 0.5. ``clip()`` gives the
    ``CLIPImageProcessor`` defaults (shortest edge 224 bicubic, centre crop 224, CLIP mean / std).
    """

    def __init__(self, do_resize: bool = True, size: Optional[Dict[str, int]] = None, resample: int = BILINEAR,
                 do_center_crop: bool = False, crop_size: Optional[Dict[str, int]] = None, do_rescale: bool = True,
                 rescale_factor: float = 1 / 255, do_normalize: bool = True,
                 image_mean: Sequence[float] = IMAGENET_STANDARD_MEAN, image_std: Sequence[float] = IMAGENET_STANDARD_STD,
                 device="cuda", dtype: torch.dtype = torch.float32):
        self.do_resize = do_resize
        self.size = dict(size) if size is not None else {"height": 224, "width": 224}
        if not ("shortest_edge" in self.size or ("height" in self.size and "width" in self.size)):
            # same condition as image_processing_backends.py:557-561
            raise ValueError("Size must contain 'height' and 'width' keys, or 'shortest_edge' key. Got "
                             f"{self.size}.")
        self.resample = int(resample)
        if self.resample not in (LANCZOS, BILINEAR, BICUBIC, BOX, HAMMING):
            raise ValueError(f"resample must be a PIL filter code in 1..5 (got {resample})")
        self.do_center_crop = do_center_crop
        self.crop_size = dict(crop_size) if crop_size is not None else {"height": 224, "width": 224}
        if do_center_crop and not ("height" in self.crop_size and "width" in self.crop_size):
            raise ValueError(f"The size dictionary must have keys 'height' and 'width'. Got {self.crop_size.keys()}")
        self.do_rescale = do_rescale
        self.rescale_factor = float(rescale_factor)
        self.do_normalize = do_normalize
        self.image_mean = tuple(float(m) for m in image_mean)
        self.image_std = tuple(float(s) for s in image_std)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("ImageProcessor runs on the GPU only (no CPU fallback); pass device='cuda[:i]'")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.dtype = dtype
        self._plans: Dict[Tuple[int, int, int], _Plan] = {}  # insertion-ordered: least recently used first
        self.max_plans = 64  # each plan holds small device tables + a workspace; oldest geometry is dropped

    @classmethod
    def vit(cls, image_size: int = 224, **kw) -> "ImageProcessor":
        return cls(size={"height": image_size, "width": image_size}, **kw)

    @classmethod
    def clip(cls, image_size: int = 224, **kw) -> "ImageProcessor":
        return cls(size={"shortest_edge": image_size}, resample=BICUBIC, do_center_crop=True,
                   crop_size={"height": image_size, "width": image_size}, image_mean=OPENAI_CLIP_MEAN,
                   image_std=OPENAI_CLIP_STD, **kw)

    # ---- geometry (host logic, CPU-testable) ----
    def resized_hw(self, height: int, width: int) -> Tuple[int, int]:
        """Output size of the resize step (image_processing_backends.py:545-556,
        image_transforms.py:295-310 for the shortest-edge rule)."""
        if not self.do_resize:
            return height, width
        if "shortest_edge" in self.size:
            s = self.size["shortest_edge"]
            short, long = (width, height) if width <= height else (height, width)
            new_short, new_long = s, int(s * long / short)
            return (new_long, new_short) if width <= height else (new_short, new_long)
        return self.size["height"], self.size["width"]

    def output_hw(self, height: int, width: int) -> Tuple[int, int]:
        if self.do_center_crop:
            return self.crop_size["height"], self.crop_size["width"]
        return self.resized_hw(height, width)

    def plan_config(self, height: int, width: int, channels: int) -> "_lib.vit_preproc_config":
        if channels not in (1, 3):
            raise ValueError(f"images must have 1 or 3 channels (got {channels})")
        if self.do_normalize and (len(self.image_mean) < channels or len(self.image_std) < channels):
            # same condition as image_transforms.py:422-432
            raise ValueError(f"mean must have {channels} elements if it is an iterable, got {len(self.image_mean)}")
        rh, rw = self.resized_hw(height, width) if self.do_resize else (0, 0)
        ch, cw = (self.crop_size["height"], self.crop_size["width"]) if self.do_center_crop else (0, 0)
        mean = (ctypes.c_float * 4)(*(list(self.image_mean[:channels]) + [0.0] * (4 - channels)))
        std = (ctypes.c_float * 4)(*(list(self.image_std[:channels]) + [1.0] * (4 - channels)))
        return _lib.vit_preproc_config(in_height=height, in_width=width, channels=channels, resize_height=rh,
                                       resize_width=rw, crop_height=ch, crop_width=cw, resample=self.resample,
                                       do_rescale=int(self.do_rescale), do_normalize=int(self.do_normalize),
                                       rescale_factor=self.rescale_factor, image_mean=mean, image_std=std)

    # ---- run ----
    def _as_batches(self, images):
        """Normalise the input to a list of uint8 tensors [b, H, W, C]."""
        if isinstance(images, (np.ndarray, torch.Tensor)) or hasattr(images, "__array_interface__"):
            images = [images]
        out = []
        for im in images:
            if not torch.is_tensor(im):
                im = torch.from_numpy(np.ascontiguousarray(np.asarray(im)))
            if im.dtype != torch.uint8:
                raise TypeError(f"images must be uint8 (decoded pixels), got {im.dtype}")
            if im.dim() == 2:
                im = im[..., None]
            if im.dim() == 3:
                im = im[None]
            if im.dim() != 4:
                raise ValueError("each image must be [H, W], [H, W, C] or [B, H, W, C]")
            out.append(im)
        return out

    def preprocess(self, images, return_tensors: str = "pt") -> Dict[str, torch.Tensor]:
        batches = self._as_batches(images)
        lib = _lib.load()
        results = []
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            for im in batches:
                b, h, w, c = im.shape
                key = (h, w, c)
                plan = self._plans.pop(key, None)
                if plan is None:
                    plan = _Plan(self.plan_config(h, w, c), self.device)
                    while len(self._plans) >= self.max_plans:
                        # the evicted plan may still be in use by queued launches: drain before freeing
                        torch.cuda.current_stream().synchronize()
                        self._plans.pop(next(iter(self._plans))).close()
                self._plans[key] = plan  # most recently used last
                src = im.to(self.device, non_blocking=True).contiguous()
                oh, ow = plan.out_hw
                dst = torch.empty(b, c, oh, ow, dtype=self.dtype, device=self.device)
                ws = plan.workspace(b)
                _lib.check(lib.vit_preproc_run(plan.handle, src.data_ptr(), b, dst.data_ptr(), dtype_code(self.dtype),
                                               ws.data_ptr(), ws.numel(), stream))
                results.append(dst)
        shapes = {tuple(r.shape[1:]) for r in results}
        if len(shapes) > 1:
            raise ValueError(f"images preprocess to different shapes {sorted(shapes)}; enable do_center_crop or "
                             "a fixed height/width size to batch them")
        return {"pixel_values": results[0] if len(results) == 1 else torch.cat(results, dim=0)}

    __call__ = preprocess

    def close(self) -> None:
        for p in self._plans.values():
            p.close()
        self._plans.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
