"""CPU oracle for the ViT encoder forward path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file; the product path (``vit_amd``) never does and fails loudly when its HIP
library is missing.

PARITY UNPINNED BY THE REFERENCE: ``/root/reference`` contains no ViT implementation, no
tests and no golden vectors for this path (SURVEY.md sections 0 and 8c). The algorithm is
therefore restated from the third-party definition the survey pins the build to --
``transformers`` 5.15.0 (on torch 2.10.0): ``models/vit/modeling_vit.py`` and
``models/clip/modeling_clip.py`` -- and this restatement is pinned against that package
by ``oracle/gen_golden.py`` (run in the build container only), whose outputs are the
fixtures under ``tests/golden/``.

Everything here is plain fp32 torch on whatever device the inputs live on (CPU in
tests); no ``transformers`` import, no custom kernels.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def patch_embed(pixel_values, weight, bias, patch):
    """Conv2d(kernel=stride=P) as an explicit GEMM over flattened patches.

    Follows ViTPatchEmbeddings.forward (modeling_vit.py:62-69) and the CLIP
    ``patch_embedding`` call (modeling_clip.py:209-210; ``bias=False`` there).
    Patch order is row-major over the grid; K index order is (c, i, j)."""
    b, c, h, w = pixel_values.shape
    gh, gw = h // patch, w // patch
    x = pixel_values.reshape(b, c, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5)
    x = x.reshape(b, gh * gw, c * patch * patch)
    out = x @ weight.reshape(weight.shape[0], -1).t()
    if bias is not None:
        out = out + bias
    return out


def embeddings(pixel_values, sd, cfg):
    """CLS prepend + position add: ViTEmbeddings.forward (modeling_vit.py:129-161),
    CLIPVisionEmbeddings.forward (modeling_clip.py:202-218)."""
    b = pixel_values.shape[0]
    if cfg.variant == "vit":
        pe = patch_embed(pixel_values, sd["embeddings.patch_embeddings.projection.weight"],
                         sd["embeddings.patch_embeddings.projection.bias"], cfg.patch_size)
        cls = sd["embeddings.cls_token"].reshape(1, 1, -1).expand(b, 1, -1)
        pos = sd["embeddings.position_embeddings"].reshape(1, cfg.seq_len, -1)
    else:
        pe = patch_embed(pixel_values, sd["embeddings.patch_embedding.weight"], None, cfg.patch_size)
        cls = sd["embeddings.class_embedding"].reshape(1, 1, -1).expand(b, 1, -1)
        pos = sd["embeddings.position_embedding.weight"].reshape(1, cfg.seq_len, -1)
    return torch.cat([cls, pe], dim=1) + pos


def layer_norm(x, weight, bias, eps):
    """nn.LayerNorm over the last dim with biased variance (modeling_vit.py:261-262)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * weight + bias


def linear(x, weight, bias=None):
    """nn.Linear, weight [out, in]: the q/k/v, o_proj, fc1 and fc2 projections (modeling_vit.py:216-218, :236,
    :250-252; CLIP modeling_clip.py:296-333, :349-353)."""
    y = x @ weight.t()
    return y if bias is None else y + bias


def activation(x, name):
    if name == "gelu":  # exact erf GELU, configuration_vit.py:54
        return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))
    if name == "quick_gelu":  # configuration_clip.py:105
        return x * torch.sigmoid(1.702 * x)
    raise ValueError(name)


def attention(x, wq, bq, wk, bk, wv, bv, heads):
    """q/k/v projections + softmax(QK^T * dh^-0.5) V: ViTAttention.forward
    (modeling_vit.py:207-238) with eager_attention_forward (:164-189).
    Returns the [B, N, D] context BEFORE the output projection."""
    b, n, d = x.shape
    dh = d // heads

    def proj(w, bias):
        y = x @ w.t()
        if bias is not None:
            y = y + bias
        return y.reshape(b, n, heads, dh).transpose(1, 2)

    q, k, v = proj(wq, bq), proj(wk, bk), proj(wv, bv)
    s = (q @ k.transpose(2, 3)) * dh ** -0.5
    p = torch.softmax(s.float(), dim=-1).to(q.dtype)
    return (p @ v).transpose(1, 2).reshape(b, n, d)


def _layer_keys(cfg, i):
    if cfg.variant == "vit":
        pre = f"layers.{i}."
        return dict(ln1=pre + "layernorm_before", ln2=pre + "layernorm_after",
                    q=pre + "attention.q_proj", k=pre + "attention.k_proj", v=pre + "attention.v_proj",
                    o=pre + "attention.o_proj", fc1=pre + "mlp.fc1", fc2=pre + "mlp.fc2")
    pre = f"encoder.layers.{i}."
    return dict(ln1=pre + "layer_norm1", ln2=pre + "layer_norm2",
                q=pre + "self_attn.q_proj", k=pre + "self_attn.k_proj", v=pre + "self_attn.v_proj",
                o=pre + "self_attn.out_proj", fc1=pre + "mlp.fc1", fc2=pre + "mlp.fc2")


def encoder_layer(x, sd, cfg, i, trace=None):
    """Pre-LN block: ViTLayer.forward (modeling_vit.py:266-286) /
    CLIPEncoderLayer.forward (modeling_clip.py:362-383)."""
    k = _layer_keys(cfg, i)
    g = lambda name: sd.get(name)  # noqa: E731
    h = layer_norm(x, sd[k["ln1"] + ".weight"], sd[k["ln1"] + ".bias"], cfg.layer_norm_eps)
    if trace is not None:
        trace[f"l{i}.ln1"] = h
    ctx = attention(h, sd[k["q"] + ".weight"], g(k["q"] + ".bias"), sd[k["k"] + ".weight"], g(k["k"] + ".bias"),
                    sd[k["v"] + ".weight"], g(k["v"] + ".bias"), cfg.num_attention_heads)
    if trace is not None:
        trace[f"l{i}.ctx"] = ctx
    x = x + (ctx @ sd[k["o"] + ".weight"].t() + sd[k["o"] + ".bias"])
    if trace is not None:
        trace[f"l{i}.attn_res"] = x
    h = layer_norm(x, sd[k["ln2"] + ".weight"], sd[k["ln2"] + ".bias"], cfg.layer_norm_eps)
    h = activation(h @ sd[k["fc1"] + ".weight"].t() + sd[k["fc1"] + ".bias"], cfg.hidden_act)
    if trace is not None:
        trace[f"l{i}.act"] = h
    x = x + (h @ sd[k["fc2"] + ".weight"].t() + sd[k["fc2"] + ".bias"])
    if trace is not None:
        trace[f"l{i}.out"] = x
    return x


@torch.no_grad()
def forward(pixel_values, sd, cfg, trace=None):
    """Whole encoder. Returns (last_hidden_state [B,N,D], pooled [B,D]).

    ViT (ViTModel.forward, modeling_vit.py:356-388, add_pooling_layer=False):
    last_hidden_state = final LayerNorm of all tokens; pooled = its CLS row.
    CLIP (CLIPVisionModel.forward, modeling_clip.py:641-656): pre_layrnorm before layer
    0; last_hidden_state is UN-normed; pooled = post_layernorm(CLS row)."""
    sd = {k: v.float() for k, v in sd.items()}
    x = embeddings(pixel_values.float(), sd, cfg)
    if trace is not None:
        trace["embed"] = x
    if cfg.variant == "clip":
        x = layer_norm(x, sd["pre_layrnorm.weight"], sd["pre_layrnorm.bias"], cfg.layer_norm_eps)
    for i in range(cfg.num_hidden_layers):
        x = encoder_layer(x, sd, cfg, i, trace)
    if cfg.variant == "vit":
        x = layer_norm(x, sd["layernorm.weight"], sd["layernorm.bias"], cfg.layer_norm_eps)
        return x, x[:, 0]
    pooled = layer_norm(x[:, 0], sd["post_layernorm.weight"], sd["post_layernorm.bias"], cfg.layer_norm_eps)
    return x, pooled


def interpolate_pos_encoding(pos, new_grid_h, new_grid_w):
    """ViTEmbeddings.interpolate_pos_encoding (modeling_vit.py:89-127; CLIP modeling_clip.py:161-200)
    on a [1 + g*g, D] table: CLS row kept, patch rows bicubic-resized (align_corners=False)."""
    pos = pos.reshape(-1, pos.shape[-1]).float()
    g = int(round((pos.shape[0] - 1) ** 0.5))
    d = pos.shape[1]
    patch = pos[1:].reshape(1, g, g, d).permute(0, 3, 1, 2)
    patch = F.interpolate(patch, size=(new_grid_h, new_grid_w), mode="bicubic", align_corners=False)
    return torch.cat([pos[:1], patch.permute(0, 2, 3, 1).reshape(-1, d)], dim=0)


def resized_state_dict(sd, cfg, image_size):
    """State dict + config of the same model at another (square) resolution."""
    import dataclasses
    key = "embeddings.position_embeddings" if cfg.variant == "vit" else "embeddings.position_embedding.weight"
    g = image_size // cfg.patch_size
    new = dict(sd)
    p = interpolate_pos_encoding(sd[key], g, g)
    new[key] = p.reshape(1, -1, p.shape[-1]) if cfg.variant == "vit" else p
    return new, dataclasses.replace(cfg, image_size=image_size)


def quantize_state_dict_bf16(sd):
    """Round weights to bf16 and back: what the bf16 GPU path actually holds. Lets a test
    separate 'weights were rounded' from 'arithmetic is wrong'."""
    return {k: v.to(torch.bfloat16).float() for k, v in sd.items()}


def forward_autograd(pixel_values, sd, cfg):
    """Same function as forward() with autograd enabled: torch.autograd of this definition is the
    oracle of the training-step kernels (the reference has no backward of its own)."""
    return forward.__wrapped__(pixel_values, sd, cfg)
