"""Local vision tower in front of the reference's VLM call sites.

The reference never runs a vision encoder itself: ``Vlm_coder.py`` base64-encodes an image file
(:24-26, :199-203) and hands it to ``client.generate(model=..., prompt=..., images=[image_b64],
options=...)`` (:205, :381); ``vlm_cad.py`` loads the file (:40) and calls
``vlm_model.generate_content([prompt, img], ...)`` (:171). This adapter takes exactly what those
call sites hold -- a file path, the file's bytes, or the base64 string -- and produces the
embeddings on the local MI355X: decode (Pillow; decoding is host work and not part of the GPU
path) -> ``ImageProcessor`` (GPU) -> ``ViTEncoder`` (GPU).

``wrap_generate`` returns a callable with the signature of ``ollama.Client.generate`` as the
reference uses it: it encodes every entry of ``images=`` locally, keeps the result in
``.last_embeddings``, and then forwards the call unchanged (or, with no inner client, returns the
embeddings in the response dict) -- so either script can adopt the local tower by replacing one
name and nothing else.
"""
from __future__ import annotations

import base64
import io
from typing import Callable, List, Mapping, Optional, Sequence, Union

import numpy as np
import torch

from .config import ViTConfig
from .encoder import EncoderOutput, ViTEncoder
from .image_processor import ImageProcessor

ImageLike = Union[str, bytes, np.ndarray, torch.Tensor]


def decode_image(data: bytes) -> np.ndarray:
    """Encoded file bytes (PNG/JPEG/...) -> uint8 [H, W, 3]. Host-side, via Pillow."""
    try:
        from PIL import Image
    except ImportError as e:  # pragma: no cover - Pillow is in the image
        raise RuntimeError("decoding image files needs Pillow; pass decoded uint8 arrays instead") from e
    with Image.open(io.BytesIO(data)) as im:
        return np.array(im.convert("RGB"))  # a writable copy; RGB as CLIPImageProcessor's do_convert_rgb


class LocalVisionTower:
    """image (path / bytes / base64 / decoded array) -> EncoderOutput on the local GPU."""

    def __init__(self, config: ViTConfig, state_dict: Mapping[str, torch.Tensor], processor: Optional[ImageProcessor] = None,
                 dtype: torch.dtype = torch.bfloat16, device="cuda", **encoder_kwargs):
        self.encoder = ViTEncoder(config, dtype=dtype, device=device, **encoder_kwargs)
        self.encoder.load_state_dict(state_dict)
        if processor is None:
            make = ImageProcessor.clip if config.variant == "clip" else ImageProcessor.vit
            processor = make(config.image_size, device=device, dtype=dtype)
        self.processor = processor

    def _decoded(self, image: ImageLike) -> Union[np.ndarray, torch.Tensor]:
        if isinstance(image, (np.ndarray, torch.Tensor)):
            return image
        if isinstance(image, bytes):
            return decode_image(image)
        if isinstance(image, str):
            # what Vlm_coder.py:24-26 does with image_path -- or the base64 text it produces from it
            try:
                with open(image, "rb") as f:
                    return decode_image(f.read())
            except (FileNotFoundError, OSError, ValueError):
                return decode_image(base64.b64decode(image, validate=True))
        raise TypeError(f"unsupported image type {type(image)}")

    def encode(self, images: Union[ImageLike, Sequence[ImageLike]]) -> EncoderOutput:
        if isinstance(images, (str, bytes, np.ndarray, torch.Tensor)):
            images = [images]
        decoded = [self._decoded(im) for im in images]
        return self.encoder(self.processor(decoded)["pixel_values"])


def wrap_generate(tower: LocalVisionTower, inner: Optional[Callable] = None) -> Callable:
    """A drop-in for ``client.generate`` (Vlm_coder.py:205, :381)."""

    def generate(model: str = "", prompt: str = "", images: Optional[List[ImageLike]] = None, options=None, **kw):
        generate.last_embeddings = tower.encode(images) if images else None
        if inner is not None:
            return inner(model=model, prompt=prompt, images=images, options=options, **kw)
        out = generate.last_embeddings
        return {"response": "", "model": model,
                "image_embeds": None if out is None else (out.image_embeds if out.image_embeds is not None
                                                          else out.pooler_output),
                "patch_embeds": None if out is None else out.last_hidden_state}

    generate.last_embeddings = None
    return generate
