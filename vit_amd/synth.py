"""Counter-based synthetic tensors (weights and images).

``torch.randn`` streams are not a cross-version contract, and full-size weights are too
large to commit, so synthetic weights and images come from a pure
integer hash of (seed, tensor name, element index). The only floating-point step is one
int -> float64 -> float32 scale, so every platform produces identical bits.

Distribution: sum of four 16-bit uniforms (Irwin-Hall, n=4), centred and scaled to
unit variance: close enough to N(0,1) for random-init weights and synthetic images.
"""
from __future__ import annotations

import zlib

import torch

from .config import ViTConfig

_M32 = 0xFFFFFFFF
_IH4_STD = 37837.22  # sqrt(4 * (65536**2 - 1) / 12)


def _hash32(x: torch.Tensor) -> torch.Tensor:
    """lowbias32 integer hash evaluated in int64 with explicit 32-bit masks."""
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def _stream_key(seed: int, name: str) -> int:
    return (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & _M32


def normal(shape, seed: int, name: str, std: float = 1.0, mean: float = 0.0,
           device="cpu", chunk: int = 1 << 24) -> torch.Tensor:
    """Deterministic fp32 tensor of ``shape``; element i depends only on (seed, name, i)."""
    n = 1
    for s in shape:
        n *= int(s)
    key = _stream_key(seed, name)
    out = torch.empty(n, dtype=torch.float32, device=device)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        idx = torch.arange(lo, hi, dtype=torch.int64, device=device)
        a = _hash32(idx * 2 + key)
        b = _hash32((idx * 2 + 1) ^ ((key * 0x85EBCA6B) & _M32))
        s = (a & 0xFFFF) + (a >> 16) + (b & 0xFFFF) + (b >> 16)
        z = (s - 131070).to(torch.float64) / _IH4_STD
        out[lo:hi] = (z * std + mean).to(torch.float32)
    return out.reshape(*shape)


def state_dict_shapes(cfg: ViTConfig) -> dict:
    """HF key names and shapes."""
    d, f, p, c, n = cfg.hidden_size, cfg.intermediate_size, cfg.patch_size, cfg.num_channels, cfg.seq_len
    sd = {}
    if cfg.variant == "vit":
        sd["embeddings.cls_token"] = (1, 1, d)
        sd["embeddings.position_embeddings"] = (1, n, d)
        sd["embeddings.patch_embeddings.projection.weight"] = (d, c, p, p)
        sd["embeddings.patch_embeddings.projection.bias"] = (d,)
        for i in range(cfg.num_hidden_layers):
            pre = f"layers.{i}."
            for proj in ("q_proj", "k_proj", "v_proj", "o_proj"):
                sd[pre + f"attention.{proj}.weight"] = (d, d)
                if proj == "o_proj" or cfg.qkv_bias:
                    sd[pre + f"attention.{proj}.bias"] = (d,)
            for ln in ("layernorm_before", "layernorm_after"):
                sd[pre + ln + ".weight"] = (d,)
                sd[pre + ln + ".bias"] = (d,)
            sd[pre + "mlp.fc1.weight"] = (f, d)
            sd[pre + "mlp.fc1.bias"] = (f,)
            sd[pre + "mlp.fc2.weight"] = (d, f)
            sd[pre + "mlp.fc2.bias"] = (d,)
        sd["layernorm.weight"] = (d,)
        sd["layernorm.bias"] = (d,)
    else:
        sd["embeddings.class_embedding"] = (d,)
        sd["embeddings.patch_embedding.weight"] = (d, c, p, p)
        sd["embeddings.position_embedding.weight"] = (n, d)
        sd["pre_layrnorm.weight"] = (d,)
        sd["pre_layrnorm.bias"] = (d,)
        for i in range(cfg.num_hidden_layers):
            pre = f"encoder.layers.{i}."
            for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
                sd[pre + f"self_attn.{proj}.weight"] = (d, d)
                sd[pre + f"self_attn.{proj}.bias"] = (d,)
            for ln in ("layer_norm1", "layer_norm2"):
                sd[pre + ln + ".weight"] = (d,)
                sd[pre + ln + ".bias"] = (d,)
            sd[pre + "mlp.fc1.weight"] = (f, d)
            sd[pre + "mlp.fc1.bias"] = (f,)
            sd[pre + "mlp.fc2.weight"] = (d, f)
            sd[pre + "mlp.fc2.bias"] = (d,)
        sd["post_layernorm.weight"] = (d,)
        sd["post_layernorm.bias"] = (d,)
    return sd


def _is_norm_key(k: str) -> bool:
    return "norm" in k


def synthetic_state_dict(cfg: ViTConfig, seed: int = 0, device="cpu") -> dict:
    """Random-init weights with the HF init scale (modeling_vit.py:323-332: N(0, 0.02))
    but NON-trivial biases and LayerNorm affine, so bias/affine bugs cannot hide
    (SURVEY.md section 7 "Random init hides bugs")."""
    sd = {}
    for k, shape in state_dict_shapes(cfg).items():
        if _is_norm_key(k):
            if k.endswith(".weight"):
                sd[k] = normal(shape, seed, k, std=0.1, mean=1.0, device=device)
            else:
                sd[k] = normal(shape, seed, k, std=0.05, device=device)
        elif k.endswith(".bias"):
            sd[k] = normal(shape, seed, k, std=0.05, device=device)
        else:
            sd[k] = normal(shape, seed, k, std=0.02, device=device)
    return sd


def synthetic_images(cfg: ViTConfig, batch: int, seed: int = 1, device="cpu") -> torch.Tensor:
    """pixel_values ~ N(0,1), fp32, [B, C, H, W]."""
    return normal((batch, cfg.num_channels, cfg.image_size, cfg.image_size), seed, "pixel_values", device=device)


def synthetic_uint8_image(height: int, width: int, channels: int = 3, seed: int = 2, name: str = "image",
                          device="cpu") -> torch.Tensor:
    """Decoded-image stand-in, uint8 [H, W, C]: 8x8 blocks that are black, white or a flat random
    colour (hard edges make bicubic / Lanczos overshoot, so the clip-to-uint8 paths are exercised)
    with per-pixel noise on a quarter of the pixels. Pure integer hash of (seed, name, index)."""
    key = _stream_key(seed, name)
    y = torch.arange(height, dtype=torch.int64, device=device).view(-1, 1, 1)
    x = torch.arange(width, dtype=torch.int64, device=device).view(1, -1, 1)
    c = torch.arange(channels, dtype=torch.int64, device=device).view(1, 1, -1)
    blk = _hash32(((y // 8) * 4099 + (x // 8)) * 7 + key)            # per block
    blk_c = _hash32(blk + c * 0x9E37 + 1)                              # per block and channel
    kind = blk % 4                                                     # 0 black, 1 white, 2/3 flat colour
    base = torch.where(kind == 0, torch.zeros_like(blk_c), torch.where(kind == 1, torch.full_like(blk_c, 255),
                                                                       blk_c & 255))
    pix = _hash32(((y * width + x) * channels + c) ^ ((key * 0x85EBCA6B) & _M32))
    noisy = (pix >> 8) % 4 == 0
    return torch.where(noisy, pix & 255, base).to(torch.uint8)
