"""CPU, world_size 2 over gloo: the batch-shard + embedding all-gather path of vit_amd.dist.
The local encoder is the CPU oracle (tests may use it); the product code under test is the
sharding and gather logic that bench.py / RCCL use unchanged on GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vit_amd.dist import shard_bounds


def test_shard_bounds_cover_batch_exactly():
    for gb in (0, 1, 7, 8, 128, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, global_batch, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import vit_oracle
        from vit_amd import synth
        from vit_amd.config import ViTConfig
        from vit_amd.dist import DataParallelEncoder
        torch.set_num_threads(1)
        cfg = ViTConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                        image_size=32, patch_size=8)
        sd = synth.synthetic_state_dict(cfg, seed=0)
        px = synth.synthetic_images(cfg, global_batch, seed=1)
        dp = DataParallelEncoder(lambda p: vit_oracle.forward(p, sd, cfg))
        hidden, pooled = dp.encode_global(px, gather_hidden=True)
        ref_h, ref_p = vit_oracle.forward(px, sd, cfg)
        ok = torch.allclose(pooled, ref_p, atol=1e-5) and torch.allclose(hidden, ref_h, atol=1e-5)
        ok = ok and tuple(pooled.shape) == (global_batch, 64)
        # presharded entry: each rank passes only its slice
        lo, hi = shard_bounds(global_batch, rank, world)
        _, pooled2 = dp.encode_global(px[lo:hi], global_batch=global_batch, presharded=True)
        ok = ok and torch.equal(pooled2, pooled)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [6, 5, 1])
def test_two_rank_gather_matches_single_process(global_batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, global_batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = dict(q.get(timeout=5) for _ in range(2))
    assert results == {0: True, 1: True}


# ---------------------------------------------------------------- DP training: bucketed all-reduce
class _FakeCfg:
    num_hidden_layers = 3


class _FakeTrainer:
    """Stands in for ViTTrainer on CPU: a gradient arena with 3 contiguous layer ranges between a
    head and a tail region; backward() fills it with rank-dependent values stage by stage."""

    def __init__(self, rank):
        self.cfg = _FakeCfg()
        self.device = torch.device("cpu")
        self.grads = torch.zeros(10 + 3 * 7 + 5)
        self.rank = rank
        self.calls = []

    def layer_grad_range(self, layer):
        return self.grads[10 + 7 * layer: 10 + 7 * (layer + 1)]

    def backward(self, d_hidden, d_pooled, on_stage):
        L = self.cfg.num_hidden_layers
        for s in range(L + 2):
            if 1 <= s <= L:
                self.layer_grad_range(L - s).fill_(float((self.rank + 1) * (L - s + 1)))
            elif s == 0:
                self.grads[31:].fill_(float(self.rank + 1) * 100)
            else:
                self.grads[:10].fill_(float(self.rank + 1) * 1000)
            self.calls.append(s)
            on_stage(s)


def _dp_train_worker(rank, world, port):
    import torch.distributed as dist
    from vit_amd.trainer import DataParallelTrainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = _FakeTrainer(rank)
        dp = DataParallelTrainer(t)
        dp.backward()
        mean = sum(r + 1 for r in range(world)) / world
        assert t.calls == [0, 1, 2, 3, 4]
        for layer in range(3):
            assert torch.allclose(t.layer_grad_range(layer), torch.full((7,), mean * (layer + 1)))
        assert torch.allclose(t.grads[31:], torch.full((5,), mean * 100))
        assert torch.allclose(t.grads[:10], torch.full((10,), mean * 1000))
        views = dp.non_layer_ranges()
        assert [v.numel() for v in views] == [10, 5]
    finally:
        dist.destroy_process_group()


def test_dp_trainer_reduces_every_gradient_once_world2():
    import torch.multiprocessing as mp
    mp.spawn(_dp_train_worker, args=(2, 29617), nprocs=2, join=True)
