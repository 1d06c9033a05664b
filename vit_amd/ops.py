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


def layer_norm_stats(x: torch.Tensor, eps: float) -> torch.Tensor:
    """Per-row ``(mean * rstd, rstd)`` of nn.LayerNorm's statistics, fp32 ``[..., 2]`` (bf16 input)."""
    x = _bf16(x, "x")
    d = x.shape[-1]
    rows = x.numel() // d
    stats = torch.empty(*x.shape[:-1], 2, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().vit_layernorm_stats(x.data_ptr(), stats.data_ptr(), rows, d, d, float(eps), _lib.VIT_BF16, _stream()))
    return stats


def layer_norm_fold(weight: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, bias: torch.Tensor | None = None):
    """Fold a LayerNorm's affine part into the Linear after it: returns ``(weight * gamma [bf16], colsum, bias')``
    for ``linear_ln_folded`` (include/vit_amd.h, vit_layernorm_fold)."""
    w = _bf16(weight, "weight")
    g, b0 = _f32(gamma, "gamma"), _f32(beta, "beta")
    b = _f32(bias, "bias")
    n, k = w.shape
    wg = torch.empty_like(w)
    c = torch.empty(n, dtype=torch.float32, device=w.device)
    d = torch.empty(n, dtype=torch.float32, device=w.device)
    _lib.check(_lib.load().vit_layernorm_fold(w.data_ptr(), g.data_ptr(), b0.data_ptr(), _ptr(b), wg.data_ptr(), c.data_ptr(),
                                              d.data_ptr(), n, k, _stream()))
    return wg, c, d


def linear_ln_folded(x: torch.Tensor, folded, stats: torch.Tensor, activation: str | None = None) -> torch.Tensor:
    """``act(F.linear(layer_norm(x), weight, bias))`` from the UN-normalised ``x`` [bf16], the output of
    ``layer_norm_fold`` and ``layer_norm_stats(x)`` (modeling_vit.py:274 + :216-218, :281 + :250-251)."""
    x = _bf16(x, "x")
    wg, c, d = folded
    k, n = x.shape[-1], wg.shape[0]
    if wg.shape[1] != k:
        raise ValueError(f"linear_ln_folded: x has {k} features, weight expects {wg.shape[1]}")
    m = x.numel() // k
    out = torch.empty(*x.shape[:-1], n, dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().vit_gemm_ln_folded(x.data_ptr(), wg.data_ptr(), d.data_ptr(), c.data_ptr(), stats.data_ptr(),
                                              out.data_ptr(), m, n, k, _EPI[activation], _stream()))
    return out


def row_stats_finalize(row_part: torch.Tensor, eps: float, stats: torch.Tensor | None = None) -> torch.Tensor:
    """(mean * rstd, rstd) per row from the ``[dim / 64, rows, 2]`` fp32 row sums a residual GEMM left."""
    ng, m, _ = row_part.shape
    if stats is None:
        stats = torch.empty(m, 2, dtype=torch.float32, device=row_part.device)
    _lib.check(_lib.load().vit_row_stats_finalize(row_part.data_ptr(), stats.data_ptr(), m, ng * 64, float(eps), _stream()))
    return stats


def linear_residual_row_stats(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None, residual: torch.Tensor,
                              eps: float, out=None, part=None, finalize: bool = True):
    """``F.linear(x, weight, bias) + residual`` [bf16] together with the folded-LayerNorm statistics of the rows it stored
    (what ``layer_norm_stats(out, eps)`` computes, bit for bit): the residual epilogue leaves per-64-column row sums and a
    small launch finalises them (modeling_vit.py:236 + :277 -> :281, :252 + :284 -> :274). Returns (out, stats, row_part)."""
    x = _bf16(x, "x")
    w = _bf16(weight, "weight")
    r = _bf16(residual, "residual")
    b = _f32(bias, "bias")
    k, n = x.shape[-1], w.shape[0]
    if w.shape[1] != k:
        raise ValueError(f"linear_residual_row_stats: x has {k} features, weight expects {w.shape[1]}")
    m = x.numel() // k
    if out is None:
        out = torch.empty(*x.shape[:-1], n, dtype=x.dtype, device=x.device)
    if part is None:
        part = torch.empty(max(n // 64, 1), m, 2, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().vit_gemm_residual_row_stats(x.data_ptr(), w.data_ptr(), _ptr(b), r.data_ptr(), out.data_ptr(),
                                                       part.data_ptr(), m, n, k, _stream()))
    stats = row_stats_finalize(part, eps) if finalize else None
    return out, stats, part


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


# --------------------------------------------------------------------------- training-step ops
_EPI_GRAD = {"gelu": _lib.EPI_DGELU, "quick_gelu": _lib.EPI_DQUICK_GELU}


def _bf16(t: torch.Tensor, name: str) -> torch.Tensor:
    t = _dev(t, name)
    if t.dtype != torch.bfloat16:
        raise TypeError(f"{name}: the training kernels take bfloat16 activations, got {t.dtype}")
    return t


def linear_train(x, weight, bias, activation: str):
    """fc1 of ViTMLP in a training forward: returns (act(x W^T + b), x W^T + b) -- the
    pre-activation is what the backward multiplies with (modeling_vit.py:250-251)."""
    x, w = _bf16(x, "x"), _bf16(weight, "weight")
    b = _f32(bias, "bias")
    m, k, n = x.numel() // x.shape[-1], x.shape[-1], w.shape[0]
    out = torch.empty(*x.shape[:-1], n, dtype=x.dtype, device=x.device)
    pre = torch.empty_like(out)
    _lib.check(_lib.load().vit_gemm_ex(x.data_ptr(), w.data_ptr(), _ptr(b), None, out.data_ptr(), pre.data_ptr(), m, n, k,
                                       _EPI[activation], 0, _stream()))
    return out, pre


def linear_dgrad(dy, weight_t, pre_activation=None, activation: str | None = None):
    """dX = dY W as an NT GEMM on the transposed weight ``weight_t`` = W^T ([in, out]); with
    ``pre_activation`` the result is multiplied by act'(pre_activation) in the epilogue."""
    dy, wt = _bf16(dy, "dy"), _bf16(weight_t, "weight_t")
    m, k, n = dy.numel() // dy.shape[-1], dy.shape[-1], wt.shape[0]
    out = torch.empty(*dy.shape[:-1], n, dtype=dy.dtype, device=dy.device)
    epi, aux = _lib.EPI_BIAS, None
    if pre_activation is not None:
        epi, aux = _EPI_GRAD[activation], _bf16(pre_activation, "pre_activation").data_ptr()
    _lib.check(_lib.load().vit_gemm_ex(dy.data_ptr(), wt.data_ptr(), None, aux, out.data_ptr(), None, m, n, k, epi, 0,
                                       _stream()))
    return out


def transpose_pad(x2d, pad_to: int = 128, colsum: bool = False, skip_first_of: int = 0):
    """[R, C] bf16 -> [C, Rpad] (zero padded), optionally with the fp32 column sums of the input."""
    x2d = _bf16(x2d, "x")
    rows_mem, c = x2d.shape
    r = rows_mem if skip_first_of == 0 else rows_mem // (skip_first_of + 1) * skip_first_of
    rpad = (r + pad_to - 1) // pad_to * pad_to
    out = torch.empty(c, rpad, dtype=x2d.dtype, device=x2d.device)
    cs = torch.zeros(c, dtype=torch.float32, device=x2d.device) if colsum else None
    _lib.check(_lib.load().vit_transpose_pad(x2d.data_ptr(), out.data_ptr(), _ptr(cs), r, c, rpad, skip_first_of,
                                             _stream()))
    return (out, cs) if colsum else out


def linear_wgrad(dy_t, x_t):
    """dW [out, in] (fp32) = dY^T X from the transposed, zero-padded operands
    dy_t = dY^T [out, Mpad], x_t = X^T [in, Mpad]."""
    dy_t, x_t = _bf16(dy_t, "dy_t"), _bf16(x_t, "x_t")
    if dy_t.shape[1] != x_t.shape[1]:
        raise ValueError("linear_wgrad: operands disagree on the padded token count")
    m, k, n = dy_t.shape[0], dy_t.shape[1], x_t.shape[0]
    out = torch.empty(m, n, dtype=torch.float32, device=dy_t.device)
    _lib.check(_lib.load().vit_gemm_ex(dy_t.data_ptr(), x_t.data_ptr(), None, None, out.data_ptr(), None, m, n, k,
                                       _lib.EPI_BIAS, 1, _stream()))
    return out


def attention_train(qkv, num_heads: int):
    """attention() that also returns the base-2 log-sum-exp [B, H, N] the backward needs."""
    qkv = _bf16(qkv, "qkv")
    b, n, d3 = qkv.shape
    d = d3 // 3
    out = torch.empty(b, n, d, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(b, num_heads, n, dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.load().vit_attention_fwd_train(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), b, n, num_heads,
                                                   d // num_heads, _stream()))
    return out, lse


def attention_backward(qkv, out, d_out, lse, num_heads: int):
    """Gradient of attention() w.r.t. the packed qkv (eager_attention_forward, modeling_vit.py:164-189)."""
    qkv, out, d_out = _bf16(qkv, "qkv"), _bf16(out, "out"), _bf16(d_out, "d_out")
    b, n, d3 = qkv.shape
    d = d3 // 3
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(b, num_heads, n, dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.load().vit_attention_bwd(qkv.data_ptr(), out.data_ptr(), d_out.data_ptr(), lse.data_ptr(),
                                             delta.data_ptr(), dqkv.data_ptr(), b, n, num_heads, d // num_heads,
                                             _stream()))
    return dqkv


def layer_norm_backward(x, dy, weight, eps: float, d_residual=None):
    """Returns (dx [+ d_residual], dgamma, dbeta) of nn.LayerNorm (modeling_vit.py:261-262)."""
    x, dy = _bf16(x, "x"), _bf16(dy, "dy")
    w = _f32(weight, "weight")
    d = x.shape[-1]
    rows = x.numel() // d
    dx = torch.empty_like(x)
    dg = torch.zeros(d, dtype=torch.float32, device=x.device)
    db = torch.zeros(d, dtype=torch.float32, device=x.device)
    dres = None if d_residual is None else _bf16(d_residual, "d_residual").data_ptr()
    _lib.check(_lib.load().vit_layernorm_bwd(x.data_ptr(), dy.data_ptr(), w.data_ptr(), dres, dx.data_ptr(),
                                             dg.data_ptr(), db.data_ptr(), rows, d, float(eps), _stream()))
    return dx, dg, db
