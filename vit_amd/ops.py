"""Torch-tensor wrappers over the per-kernel C-ABI entry points.

Each function mirrors the torch module/function it replaces on the encoder path
(names and argument meaning as in ``transformers`` ``modeling_vit.py``): tensors are CUDA
(ROCm) tensors; results are new tensors. All work is done by libvit_amd.so on the current
torch stream -- there is no eager fallback.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import VIT_BF16, VIT_F32, EPI_BIAS, EPI_GELU, EPI_QUICK_GELU, EPI_RESIDUAL  # noqa: F401

_DT = {torch.float32: VIT_F32, torch.bfloat16: VIT_BF16}
_EPI = {None: EPI_BIAS, "none": EPI_BIAS, "gelu": EPI_GELU, "quick_gelu": EPI_QUICK_GELU}


def dtype_code(t: torch.dtype) -> int:
    try:
        return _DT[t]
    except KeyError:
        raise TypeError(f"vit_amd supports float32 and bfloat16 tensors, got {t}") from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (vit_amd has no CPU path)")
    return t.contiguous()


def _f32(t, name):
    return None if t is None else _dev(t, name).float()


def _ptr(t):
    return None if t is None else t.data_ptr()


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float) -> torch.Tensor:
    """``nn.LayerNorm(D, eps)`` over the last dim (modeling_vit.py:261-262)."""
    x = _dev(x, "x")
    w, b = _f32(weight, "weight"), _f32(bias, "bias")
    d = x.shape[-1]
    rows = x.numel() // d if d else 0
    y = torch.empty_like(x)
    _lib.check(_lib.load().vit_layernorm_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), rows, d, d, d,
                                             float(eps), dtype_code(x.dtype), _stream()))
    return y


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None, activation: str | None = None,
           residual: torch.Tensor | None = None) -> torch.Tensor:
    """``F.linear(x, weight, bias)`` followed by an activation or a residual add
    (modeling_vit.py:216-218, 236+277, 250-252+284). weight is [out, in]."""
    x = _dev(x, "x")
    w = _dev(weight, "weight").to(x.dtype)
    b = _f32(bias, "bias")
    k = x.shape[-1]
    n = w.shape[0]
    if w.shape[1] != k:
        raise ValueError(f"linear: x has {k} features, weight expects {w.shape[1]}")
    m = x.numel() // k if k else 0
    out = torch.empty(*x.shape[:-1], n, dtype=x.dtype, device=x.device)
    if residual is not None:
        if activation not in (None, "none"):
            raise ValueError("linear: activation and residual are mutually exclusive epilogues")
        residual = _dev(residual, "residual").to(x.dtype)
        if residual.shape != out.shape:
            raise ValueError("linear: residual shape mismatch")
        epi = EPI_RESIDUAL
    else:
        if activation not in _EPI:
            raise ValueError(f"linear: unknown activation {activation!r}")
        epi = _EPI[activation]
    _lib.check(_lib.load().vit_gemm_bias_act(x.data_ptr(), w.data_ptr(), _ptr(b), _ptr(residual), out.data_ptr(), m, n,
                                             k, epi, dtype_code(x.dtype), _stream()))
    return out


def attention(qkv: torch.Tensor, num_heads: int) -> torch.Tensor:
    """Fused softmax(QK^T/sqrt(dh))V on the packed projection output.

    qkv: [B, N, 3*D] (q | k | v along the last dim) -> [B, N, D], equal to
    eager_attention_forward(...)'s attn_output after ``.reshape(B, N, D)``
    (modeling_vit.py:164-189, 235)."""
    qkv = _dev(qkv, "qkv")
    b, n, d3 = qkv.shape
    if d3 % (3 * num_heads):
        raise ValueError("attention: last dim must be 3 * num_heads * head_dim")
    d = d3 // 3
    out = torch.empty(b, n, d, dtype=qkv.dtype, device=qkv.device)
    _lib.check(_lib.load().vit_attention_fwd(qkv.data_ptr(), out.data_ptr(), b, n, num_heads, d // num_heads,
                                             dtype_code(qkv.dtype), _stream()))
    return out


def patch_embed(pixel_values: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None, cls_token: torch.Tensor,
                position_embeddings: torch.Tensor, patch_size: int) -> torch.Tensor:
    """ViTEmbeddings.forward (modeling_vit.py:129-161): Conv2d(k=s=P) patch projection as an
    im2col-free GEMM, CLS prepend and position add. weight is the Conv2d weight [D, C, P, P]."""
    px = _dev(pixel_values, "pixel_values")
    b, c, h, w_ = px.shape
    if h != w_ or h % patch_size:
        raise ValueError("patch_embed: square images divisible by the patch size only")
    dt = dtype_code(px.dtype)
    d = weight.shape[0]
    kreal = c * patch_size * patch_size
    kpad = _lib.load().vit_patch_kpad(c, patch_size, dt)
    wmat = _dev(weight, "weight").reshape(d, kreal).to(px.dtype)
    if kpad != kreal:  # zero-pad K to the kernel's 16-byte chunk granularity
        wmat = torch.nn.functional.pad(wmat, (0, kpad - kreal))
    n = (h // patch_size) ** 2 + 1
    cls = _f32(cls_token, "cls_token").reshape(d)
    pos = _f32(position_embeddings, "position_embeddings").reshape(n, d)
    bias = _f32(bias, "bias")
    out = torch.empty(b, n, d, dtype=px.dtype, device=px.device)
    _lib.check(_lib.load().vit_patch_embed_fwd(px.data_ptr(), wmat.data_ptr(), _ptr(bias), cls.data_ptr(),
                                               pos.data_ptr(), out.data_ptr(), b, c, h, patch_size, d, dt, _stream()))
    return out
