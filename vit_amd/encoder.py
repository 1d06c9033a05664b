"""Host-side mirror of the encoder interface (``ViTModel`` / ``CLIPVisionModel`` call
surface: ``pixel_values`` in, ``last_hidden_state`` / ``pooler_output`` out) over the C ABI.

The reference repository's scripts hand an encoded image to a remote VLM
(Vlm_coder.py:205,381; vlm_cad.py:171) -- a local encoder slots in right before those
calls; see INTEGRATION.md. The whole forward runs inside ``vit_forward`` (libvit_amd.so);
torch only owns the tensors, the workspace and the stream.
"""
from __future__ import annotations

import ctypes
import dataclasses
from dataclasses import dataclass
from typing import Mapping, Optional

import torch

from . import _lib
from .config import ViTConfig
from .ops import dtype_code


@dataclass
class EncoderOutput:
    """Same field names as transformers' BaseModelOutputWithPooling."""
    last_hidden_state: torch.Tensor
    pooler_output: torch.Tensor
    image_embeds: Optional[torch.Tensor] = None  # CLIPVisionModelWithProjection output (projection_dim set)

    def __iter__(self):
        yield self.last_hidden_state
        yield self.pooler_output


class ViTEncoder:
    """MI355X ViT / CLIP-vision encoder.

    >>> enc = ViTEncoder(ViTConfig(), dtype=torch.bfloat16, device="cuda:0")
    >>> enc.load_state_dict(hf_state_dict)      # HF key names
    >>> out = enc(pixel_values)                 # out.last_hidden_state [B,N,D], out.pooler_output [B,D]

    Heads. ``add_pooling_layer=True`` (ViT) makes ``pooler_output`` =
    tanh(dense(CLS)) as ``ViTModel(config, add_pooling_layer=True)`` does (modeling_vit.py:289-301;
    keys ``pooler.dense.{weight,bias}``); ``projection_dim=P`` (CLIP) adds ``image_embeds`` =
    visual_projection(pooler_output) as ``CLIPVisionModelWithProjection`` does (modeling_clip.py:907,
    :950; key ``visual_projection.weight``). Both run ``vit_head_fwd`` on the pooled row.
    """

    def __init__(self, config: ViTConfig, dtype: torch.dtype = torch.bfloat16, device="cuda",
                 add_pooling_layer: bool = False, projection_dim: Optional[int] = None):
        if add_pooling_layer and config.variant != "vit":
            raise ValueError("add_pooling_layer is the ViTModel pooler; CLIP contexts use projection_dim")
        if projection_dim is not None and config.variant != "clip":
            raise ValueError("projection_dim is the CLIP visual_projection; ViT contexts use add_pooling_layer")
        self.add_pooling_layer = add_pooling_layer
        self.projection_dim = projection_dim
        self._head: dict = {}
        self.config = config
        self.dtype = dtype
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("ViTEncoder runs on the GPU only (no CPU fallback); pass device='cuda[:i]'")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        lib = _lib.load()
        cfg = _lib.vit_config(
            hidden_size=config.hidden_size, num_hidden_layers=config.num_hidden_layers,
            num_attention_heads=config.num_attention_heads, intermediate_size=config.intermediate_size,
            image_size=config.image_size, patch_size=config.patch_size, num_channels=config.num_channels,
            hidden_act=config.act_id, variant=config.variant_id, qkv_bias=int(config.qkv_bias),
            dtype=dtype_code(dtype), layer_norm_eps=config.layer_norm_eps)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.vit_create(ctypes.byref(cfg), ctypes.byref(handle)))
        self._ctx = handle
        self._ws: Optional[torch.Tensor] = None

    # ---- weights ----
    def load_state_dict(self, state_dict: Mapping[str, torch.Tensor], strict: bool = True) -> None:
        """Feed an HF-style state dict (``ViTModel`` or ``CLIPVisionModel`` key names; an
        optional ``vit.`` / ``vision_model.`` prefix is stripped). q/k/v are packed inside the
        library. With strict=True unknown keys and missing tensors raise."""
        lib = _lib.load()
        self.weights_changed()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            for key, t in state_dict.items():
                name = key
                for prefix in ("vit.", "vision_model."):
                    if name.startswith(prefix):
                        name = name[len(prefix):]
                if not torch.is_tensor(t) or not t.is_floating_point():
                    continue
                if name.startswith("pooler.dense.") and self.config.variant == "vit" and not self.add_pooling_layer:
                    # a stock ViTModel checkpoint (HF default add_pooling_layer=True) carries the pooler; this encoder
                    # was built without it (our default), so the two tensors are not needed
                    continue
                if name in self._head_keys():
                    want = self._head_keys()[name]
                    if tuple(t.shape) != want:
                        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {want}")
                    # matrices in the compute dtype, bias fp32 (same rule as the arena)
                    self._head[name] = t.to(self.device, torch.float32 if name.endswith(".bias") else self.dtype).contiguous()
                    continue
                if t.dtype not in (torch.float32, torch.bfloat16):
                    t = t.float()
                src = t.to(self.device).contiguous()
                rc = lib.vit_load_weight(self._ctx, name.encode(), src.data_ptr(), dtype_code(src.dtype), src.numel(),
                                         stream)
                if rc != 0:
                    msg = lib.vit_last_error().decode()
                    if not strict and "unexpected key" in msg:
                        continue
                    raise _lib.VitError(rc, msg)
            torch.cuda.current_stream().synchronize()  # sources may be temporaries
        if strict:
            first = ctypes.c_char_p()
            n = lib.vit_missing_weights(self._ctx, ctypes.byref(first))
            if n:
                raise KeyError(f"{n} tensors missing from state_dict, e.g. {first.value.decode()!r}")
            for k in self._head_keys():
                if k not in self._head:
                    raise KeyError(f"1 tensors missing from state_dict, e.g. {k!r}")

    def weights_changed(self) -> None:
        """Mark the per-resolution clones made by ``with_image_size`` stale (each holds a device copy of the weights
        as they were when it was made). A stale clone stays a valid object: its next ``forward`` re-clones from this
        encoder (weights copied device to device, position table re-interpolated). Called by ``load_state_dict`` and
        by ``ViTTrainer.sgd_step``; costs nothing until a clone is used again."""
        for e in self.__dict__.get("_resized", {}).values():
            e._stale = True

    def _refresh_clone(self) -> None:
        parent = self._parent
        if parent._ctx is None:
            raise RuntimeError("this resized encoder's parent has been closed")
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            if self._ctx is not None and self._ctx:
                torch.cuda.synchronize(self.device)
                _lib.load().vit_destroy(self._ctx)
                self._ctx = None
            _lib.check(_lib.load().vit_clone_resized(parent._ctx, self.config.image_size, ctypes.byref(handle),
                                                     torch.cuda.current_stream().cuda_stream))
        self._ctx = handle
        self._head = parent._head
        self._stale = False

    def _head_keys(self) -> dict:
        d = self.config.hidden_size
        if self.add_pooling_layer:
            return {"pooler.dense.weight": (d, d), "pooler.dense.bias": (d,)}
        if self.projection_dim is not None:
            return {"visual_projection.weight": (self.projection_dim, d)}
        return {}

    def _head_fwd(self, x: torch.Tensor, weight: torch.Tensor, bias, activation: int) -> torch.Tensor:
        out = torch.empty(x.shape[0], weight.shape[0], dtype=self.dtype, device=self.device)
        _lib.check(_lib.load().vit_head_fwd(x.data_ptr(), x.stride(0), weight.data_ptr(),
                                            bias.data_ptr() if bias is not None else None, out.data_ptr(), x.shape[0],
                                            weight.shape[1], weight.shape[0], activation, dtype_code(self.dtype),
                                            torch.cuda.current_stream().cuda_stream))
        return out

    # ---- forward ----
    def workspace_bytes(self, batch: int) -> int:
        return int(_lib.load().vit_workspace_bytes(self._ctx, batch))

    def _workspace(self, batch: int) -> torch.Tensor:
        need = self.workspace_bytes(batch)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def with_image_size(self, image_size: int) -> "ViTEncoder":
        """A second encoder for ``image_size`` x ``image_size`` inputs: same weights, position table
        interpolated as ``interpolate_pos_encoding`` does (modeling_vit.py:89-127,
        modeling_clip.py:161-200). Cached per size until the weights change (``weights_changed``)."""
        if image_size == self.config.image_size:
            return self
        cache = self.__dict__.setdefault("_resized", {})
        if image_size not in cache:
            if image_size <= 0 or image_size % self.config.patch_size:
                raise ValueError(f"image size {image_size} is not a multiple of the patch size {self.config.patch_size}")
            new = object.__new__(ViTEncoder)
            new.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("_ctx", "_ws", "_resized", "config")})
            new.config = dataclasses.replace(self.config, image_size=image_size)
            new._ws = None
            new._ctx = None
            new._parent = self
            new._refresh_clone()
            cache[image_size] = new
        return cache[image_size]

    def forward(self, pixel_values: torch.Tensor, interpolate_pos_encoding: bool = False) -> EncoderOutput:
        cfg = self.config
        if getattr(self, "_stale", False):
            self._refresh_clone()
        if self._ctx is None:
            raise RuntimeError("this ViTEncoder has been closed")
        if pixel_values.dim() != 4:
            raise ValueError("pixel_values must be [batch, channels, height, width]")
        b, c, h, w = pixel_values.shape
        if interpolate_pos_encoding and (h != cfg.image_size or w != cfg.image_size) and c == cfg.num_channels:
            if h != w:
                raise ValueError(f"interpolate_pos_encoding supports square inputs only (got {h}*{w})")
            return self.with_image_size(h).forward(pixel_values)
        if c != cfg.num_channels:
            # same check as ViTPatchEmbeddings.forward (modeling_vit.py:63-68)
            raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the "
                             f"configuration. Expected {cfg.num_channels} but got {c}.")
        if h != cfg.image_size or w != cfg.image_size:
            # same check as ViTEmbeddings.forward (modeling_vit.py:151-155)
            raise ValueError(f"Input image size ({h}*{w}) doesn't match model ({cfg.image_size}*{cfg.image_size}).")
        if b == 0:
            return EncoderOutput(torch.empty(0, cfg.seq_len, cfg.hidden_size, dtype=self.dtype, device=self.device),
                                 torch.empty(0, cfg.hidden_size, dtype=self.dtype, device=self.device),
                                 None if self.projection_dim is None else
                                 torch.empty(0, self.projection_dim, dtype=self.dtype, device=self.device))
        if pixel_values.dtype not in (torch.float32, torch.bfloat16):
            pixel_values = pixel_values.float()
        px = pixel_values.to(self.device).contiguous()
        hidden = torch.empty(b, cfg.seq_len, cfg.hidden_size, dtype=self.dtype, device=self.device)
        pooled = torch.empty(b, cfg.hidden_size, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            ws = self._workspace(b)
            _lib.check(_lib.load().vit_forward(self._ctx, px.data_ptr(), dtype_code(px.dtype), b, hidden.data_ptr(),
                                               pooled.data_ptr(), ws.data_ptr(), ws.numel(),
                                               torch.cuda.current_stream().cuda_stream))
            embeds = None
            if self.add_pooling_layer:
                # ViTPooler reads the CLS row of the final-LayerNorm output in place (row stride N*D)
                pooled = self._head_fwd(hidden[:, 0], self._head["pooler.dense.weight"],
                                        self._head["pooler.dense.bias"], _lib.HEAD_TANH)
            elif self.projection_dim is not None:
                embeds = self._head_fwd(pooled, self._head["visual_projection.weight"], None, _lib.HEAD_LINEAR)
        return EncoderOutput(hidden, pooled, embeds)

    __call__ = forward

    def close(self) -> None:
        for e in self.__dict__.pop("_resized", {}).values():
            e.close()
        if getattr(self, "_ctx", None) is not None and self._ctx:
            torch.cuda.synchronize(self.device)
            _lib.load().vit_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
