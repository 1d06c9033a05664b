"""Host side of the training step (BASELINE.json configs[4]): forward that stashes, staged
backward, gradient all-reduce per stage, SGD on fp32 master weights -- all compute inside
libvit_amd.so; torch owns the buffers, the streams and the process group.

The reference has no training code. The surface mirrors what a
``transformers`` user does around ``ViTModel``: ``out = model(pixel_values)``, a loss on
``out.last_hidden_state`` / ``out.pooler_output``, ``backward``, optimizer step; parameter
gradients are addressed by the HF state-dict key names.
"""
from __future__ import annotations

import ctypes
from typing import Callable, Dict, Optional

import torch

from . import _lib
from .encoder import EncoderOutput, ViTEncoder
from .ops import dtype_code


class ViTTrainer:
    def __init__(self, encoder: ViTEncoder):
        if encoder.dtype != torch.bfloat16:
            raise ValueError("the training step runs in bf16 (fp32 master weights and gradients)")
        if encoder.add_pooling_layer or encoder.projection_dim is not None:
            # the heads (vit_head_fwd) have no backward: training through them would silently leave the head weights
            # without gradients and apply d_pooled to the wrong tensor
            raise ValueError("ViTTrainer trains the encoder without a head: build the ViTEncoder with "
                             "add_pooling_layer=False and projection_dim=None")
        self.enc = encoder
        self.cfg = encoder.config
        self.device = encoder.device
        self._lib = _lib.load()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vit_train_init(encoder._ctx, torch.cuda.current_stream().cuda_stream))
        nbytes = int(self._lib.vit_grad_bytes(encoder._ctx))
        self.grads = torch.zeros(nbytes // 4, dtype=torch.float32, device=self.device)
        self._stash: Optional[torch.Tensor] = None
        self._ws: Optional[torch.Tensor] = None
        self._batch = 0
        self._px: Optional[torch.Tensor] = None
        self.num_stages = self.cfg.num_hidden_layers + 2

    # ---- buffers
    def _buffers(self, batch: int):
        need_s = int(self._lib.vit_train_stash_bytes(self.enc._ctx, batch))
        need_w = int(self._lib.vit_train_workspace_bytes(self.enc._ctx, batch))
        if self._stash is None or self._stash.numel() < need_s:
            self._stash = None
            self._stash = torch.empty(need_s, dtype=torch.uint8, device=self.device)
        if self._ws is None or self._ws.numel() < need_w:
            self._ws = None
            self._ws = torch.empty(need_w, dtype=torch.uint8, device=self.device)

    # ---- forward / backward
    def forward(self, pixel_values: torch.Tensor) -> EncoderOutput:
        cfg = self.cfg
        b = pixel_values.shape[0]
        if pixel_values.dtype not in (torch.float32, torch.bfloat16):
            pixel_values = pixel_values.float()
        px = pixel_values.to(self.device).contiguous()
        hidden = torch.empty(b, cfg.seq_len, cfg.hidden_size, dtype=torch.bfloat16, device=self.device)
        pooled = torch.empty(b, cfg.hidden_size, dtype=torch.bfloat16, device=self.device)
        with torch.cuda.device(self.device):
            self._buffers(b)
            _lib.check(self._lib.vit_forward_train(self.enc._ctx, px.data_ptr(), dtype_code(px.dtype), b, hidden.data_ptr(),
                                                   pooled.data_ptr(), self._stash.data_ptr(), self._stash.numel(),
                                                   self._ws.data_ptr(), self._ws.numel(),
                                                   torch.cuda.current_stream().cuda_stream))
        self._batch, self._px = b, px
        return EncoderOutput(hidden, pooled)

    def backward(self, d_hidden: Optional[torch.Tensor] = None, d_pooled: Optional[torch.Tensor] = None,
                 on_stage: Optional[Callable[[int], None]] = None) -> None:
        """Back-propagate dL/d(last_hidden_state) and/or dL/d(pooler_output) into ``self.grads``.
        ``on_stage(s)`` is called after stage s has been enqueued (s = 0 head, 1..L layers L-1..0,
        L+1 embeddings) -- the hook the data-parallel wrapper uses to start that stage's all-reduce."""
        cfg, b = self.cfg, self._batch
        if self._px is None:
            raise RuntimeError("backward() before forward()")
        if d_hidden is None:
            d_hidden = torch.zeros(b, cfg.seq_len, cfg.hidden_size, dtype=torch.bfloat16, device=self.device)
        d_hidden = d_hidden.to(self.device, torch.bfloat16).contiguous()
        dp_ptr = None
        if d_pooled is not None:
            d_pooled = d_pooled.to(self.device, torch.bfloat16).contiguous()
            if cfg.variant == "vit":   # pooled = CLS row of last_hidden_state (modeling_vit.py:385-388, no pooler)
                d_hidden = d_hidden.clone()
                d_hidden[:, 0, :] += d_pooled
            else:
                dp_ptr = d_pooled.data_ptr()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            for s in range(self.num_stages):
                _lib.check(self._lib.vit_backward_stage(self.enc._ctx, s, self._px.data_ptr(), dtype_code(self._px.dtype), b,
                                                        d_hidden.data_ptr(), dp_ptr, self._stash.data_ptr(),
                                                        self.grads.data_ptr(), self._ws.data_ptr(), self._ws.numel(), stream))
                if on_stage is not None:
                    on_stage(s)

    # ---- gradients by HF key name
    def grad(self, name: str) -> torch.Tensor:
        off, rows, cols, pitch = ctypes.c_size_t(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self._lib.vit_grad_slot(self.enc._ctx, name.encode(), ctypes.byref(off), ctypes.byref(rows),
                                           ctypes.byref(cols), ctypes.byref(pitch)))
        o = off.value // 4
        return self.grads[o:o + rows.value * pitch.value].view(rows.value, pitch.value)[:, :cols.value]

    def layer_grad_range(self, layer: int) -> torch.Tensor:
        """Contiguous fp32 view of every gradient of transformer layer ``layer`` (one bucket)."""
        off, nb = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(self._lib.vit_grad_layer_range(self.enc._ctx, layer, ctypes.byref(off), ctypes.byref(nb)))
        return self.grads[off.value // 4:(off.value + nb.value) // 4]

    def named_grads(self, shapes: Dict[str, tuple]) -> Dict[str, torch.Tensor]:
        return {k: self.grad(k).reshape(s) for k, s in shapes.items()}

    def sgd_step(self, lr: float) -> None:
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vit_sgd_step(self.enc._ctx, self.grads.data_ptr(), float(lr),
                                              torch.cuda.current_stream().cuda_stream))
        self.enc.weights_changed()  # resized clones (interpolate_pos_encoding) hold copies of the old weights


class DataParallelTrainer:
    """Batch-sharded DP training: every rank runs the local step; the gradients of a backward
    stage are all-reduced (averaged) on a side stream as soon as that stage
    is enqueued, overlapping RCCL over xGMI with the next layer's backward. One bucket per
    transformer layer (contiguous by construction) plus one for embeddings / norms."""

    def __init__(self, trainer: ViTTrainer, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.t = trainer
        self.group = group
        self.world = dist.get_world_size(group)
        self.comm = torch.cuda.Stream(device=trainer.device) if trainer.device.type == "cuda" else None
        # RCCL averages inside the collective; gloo (the CPU tests) has no AVG and scales in a second pass
        self._avg = dist.get_backend(group) == "nccl"

    def _reduce(self, view: torch.Tensor):
        if self._avg:
            self.dist.all_reduce(view, op=self.dist.ReduceOp.AVG, group=self.group)
        else:
            self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, group=self.group)
            view.mul_(1.0 / self.world)

    def backward(self, d_hidden=None, d_pooled=None):
        t = self.t
        L = t.cfg.num_hidden_layers
        on_gpu = self.comm is not None
        main = torch.cuda.current_stream(t.device) if on_gpu else None

        def side(fn):
            """Run fn on the communication stream after everything enqueued so far on the main stream."""
            if not on_gpu:
                fn()
                return
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)
                fn()

        def on_stage(s: int):
            if 1 <= s <= L:
                side(lambda: self._reduce(t.layer_grad_range(L - s)))

        t.backward(d_hidden, d_pooled, on_stage)

        # the non-layer gradients (head, embeddings) are final only after the last stage
        def rest():
            for view in self.non_layer_ranges():
                self._reduce(view)

        side(rest)
        if on_gpu:
            main.wait_stream(self.comm)

    def non_layer_ranges(self):
        """fp32 views of the gradient arena outside the (contiguous, ordered) layer ranges."""
        t = self.t
        L = t.cfg.num_hidden_layers
        esz = t.grads.element_size()
        first = (t.layer_grad_range(0).data_ptr() - t.grads.data_ptr()) // esz
        hi = t.layer_grad_range(L - 1)
        last = (hi.data_ptr() - t.grads.data_ptr()) // esz + hi.numel()
        out = []
        if first > 0:
            out.append(t.grads[:first])
        if last < t.grads.numel():
            out.append(t.grads[last:])
        return out
