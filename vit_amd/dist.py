"""Batch-sharded data parallelism for inference: one process per GPU, RCCL over xGMI.

Images are independent and weights are replicated, so the ONLY
exchange step of the forward path is one all-gather of the embeddings: each rank encodes a
contiguous slice of the global batch and the CLS/pooled embeddings (optionally the full
hidden states) are gathered on every rank. xGMI is point-to-point (7 links per GPU), and one
all-gather lets every peer's slice travel its own link concurrently; the CLS gather is
latency-bound (B_local*D*2 bytes), so it is issued once per batch, never per layer.

Backend-agnostic on purpose: ``nccl`` (= RCCL on ROCm) on GPUs, ``gloo`` in CPU tests.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of the global batch owned by ``rank``. The first
    ``global_batch % world`` ranks get one extra image; slices are disjoint and ordered."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    if global_batch < 0:
        raise ValueError("negative batch")
    base, extra = divmod(global_batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local: torch.Tensor, global_rows: int, group=None) -> torch.Tensor:
    """Gather row-sharded tensors ([rows_r, ...] on rank r, rows per ``shard_bounds``) into
    [global_rows, ...] on every rank. Equal shards use one all_gather_into_tensor; ragged
    shards are padded to the largest shard and trimmed after the gather."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(global_rows, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} rows, expected {hi - lo}")
    local = local.contiguous()
    base, extra = divmod(global_rows, world)
    if extra == 0:
        out = torch.empty((global_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    width = base + 1
    padded = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    buf = torch.empty((world * width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    parts = []
    for r in range(world):
        rlo, rhi = shard_bounds(global_rows, r, world)
        parts.append(buf[r * width: r * width + (rhi - rlo)])
    return torch.cat(parts, dim=0)


class DataParallelEncoder:
    """Wraps a local encoder callable (``pixel_values -> (hidden, pooled)``).

    ``encode_global`` takes the GLOBAL batch (same tensor on every rank, or only this rank's
    slice with ``presharded=True``), runs the local slice and returns gathered embeddings.
    """

    def __init__(self, encoder: Callable, group=None):
        self.encoder = encoder
        self.group = group

    def encode_local(self, pixel_values: torch.Tensor):
        return self.encoder(pixel_values)

    def encode_global(self, pixel_values: torch.Tensor, global_batch: Optional[int] = None,
                      presharded: bool = False, gather_hidden: bool = False):
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        if presharded:
            if global_batch is None:
                raise ValueError("presharded input needs global_batch")
            local_px = pixel_values
        else:
            global_batch = pixel_values.shape[0]
            lo, hi = shard_bounds(global_batch, rank, world)
            local_px = pixel_values[lo:hi]
        hidden, pooled = self.encoder(local_px)
        pooled_all = all_gather_rows(pooled, global_batch, self.group)
        hidden_all = all_gather_rows(hidden, global_batch, self.group) if gather_hidden else None
        return hidden_all, pooled_all
