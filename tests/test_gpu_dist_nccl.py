"""The data-parallel wrappers on a real RCCL ("nccl") process group of world size 1 on the one GPU of the test
box: DataParallelTrainer's communication-stream / event logic around vit_backward_stage (which itself forks a
side stream) and DataParallelEncoder's gather, checked against the plain single-GPU objects. With one rank the
all-reduce is the identity and the 1/world scale is 1.0, so the weight-matrix gradients (fixed-order sums, no
atomics) must come out bit for bit; vector gradients (fp32 atomics) to rounding."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from vit_amd import synth
from vit_amd.config import ViTConfig
from vit_amd.dist import DataParallelEncoder
from vit_amd.encoder import ViTEncoder
from vit_amd.trainer import DataParallelTrainer, ViTTrainer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def nccl_group():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    yield
    dist.destroy_process_group()


def test_data_parallel_trainer_world1_matches_plain_trainer(nccl_group):
    # 8320 token rows: split-K weight gradients on the library's side stream + the wrapper's comm stream
    cfg = ViTConfig(hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=1024,
                    image_size=64, patch_size=8)
    batch = 128
    sd = synth.synthetic_state_dict(cfg, seed=0)
    px = synth.synthetic_images(cfg, batch, seed=1, device=DEV).to(torch.bfloat16)
    w_p = synth.normal((batch, cfg.hidden_size), 5, "wp", 0.5, device=DEV).to(torch.bfloat16)
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV)
    enc.load_state_dict(sd)
    tr = ViTTrainer(enc)
    tr.forward(px)
    tr.backward(None, w_p)
    torch.cuda.synchronize()
    plain = tr.grads.clone()
    dp = DataParallelTrainer(tr)
    for _ in range(3):  # repeated: a missing event between the streams shows up as a changing sum
        tr.forward(px)
        dp.backward(None, w_p)
        torch.cuda.synchronize()
        for k, v in sd.items():
            if v.dim() >= 2 and "embeddings" not in k:
                g = tr.grad(k)
                lo = g.storage_offset()
                ref = plain[lo:lo + g.shape[0] * g.stride(0)].view(g.shape[0], g.stride(0))[:, :g.shape[1]]
                assert torch.equal(g, ref), k
        assert torch.allclose(tr.grads, plain, rtol=1e-3, atol=1e-5)
    # every gradient range went through exactly one all-reduce: ranges tile the arena
    L = cfg.num_hidden_layers
    covered = sum(tr.layer_grad_range(l).numel() for l in range(L)) + sum(v.numel() for v in dp.non_layer_ranges())
    assert covered == tr.grads.numel()
    enc.close()


def test_data_parallel_encoder_world1_matches_encoder(nccl_group):
    cfg = ViTConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                    image_size=64, patch_size=16)
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV)
    enc.load_state_dict(synth.synthetic_state_dict(cfg, seed=0))
    px = synth.synthetic_images(cfg, 7, seed=2, device=DEV).to(torch.bfloat16)
    ref = enc(px)
    hidden, pooled = DataParallelEncoder(enc).encode_global(px, gather_hidden=True)
    assert torch.equal(pooled, ref.pooler_output)
    assert torch.equal(hidden, ref.last_hidden_state)
    enc.close()
