"""BASELINE.json configs[4] at size: the ViT-B/16 224^2 training step (D 768, 12 heads, F 3072, 12 layers).

(a) 8 images against torch.autograd of the CPU oracle on the same bf16-rounded weights: every parameter
    gradient cosine >= 0.99 (fp32 reference, bf16 activations on the GPU), worst one printed.
(b) the full batch of 256 images (50,432 token rows -- the bench shapes: persistent GEMMs, split-K weight
    gradients on the side stream), checked through size-independent properties: finite; weight matrices
    bit-reproducible from step to step; the loss is a sum over images, so the gradient of the batch equals
    the sum of the gradients of its two halves and does not depend on the order of the images.
"""
import pytest
import torch

from oracle import vit_oracle
from tests.helpers import cosine, min_row_cosine
from vit_amd import synth
from vit_amd.config import ViTConfig
from vit_amd.encoder import ViTEncoder
from vit_amd.trainer import ViTTrainer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_vit_b16_gradients_match_autograd_batch8():
    cfg = ViTConfig.vit_b16_224()
    batch = 8
    sd = vit_oracle.quantize_state_dict_bf16(synth.synthetic_state_dict(cfg, seed=0))
    px = synth.synthetic_images(cfg, batch, seed=1).to(torch.bfloat16).float()
    w_p = synth.normal((batch, cfg.hidden_size), 5, "wp", 0.5).to(torch.bfloat16).float()
    w_h = synth.normal((batch, cfg.seq_len, cfg.hidden_size), 5, "wh", 0.02).to(torch.bfloat16).float()
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    hidden, pooled = vit_oracle.forward_autograd(px, sdg, cfg)
    ((hidden * w_h).sum() + (pooled * w_p).sum()).backward()
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV)
    enc.load_state_dict(sd)
    tr = ViTTrainer(enc)
    out = tr.forward(px)
    assert min_row_cosine(out.pooler_output, pooled.detach()) >= 0.999
    tr.backward(w_h, w_p)
    torch.cuda.synchronize()
    worst = (1.0, "")
    for k, v in sdg.items():
        got = tr.grad(k).reshape(v.shape).cpu()
        if k.endswith("k_proj.bias"):  # exactly 0 in exact arithmetic (softmax shift invariance)
            vb = sdg[k.replace("k_proj", "v_proj")].grad
            assert got.abs().max() <= 0.05 * vb.abs().max(), k
            continue
        c = cosine(got, v.grad)
        worst = min(worst, (c, k))
        assert c >= 0.99, (k, c)
    print(f"ViT-B/16 batch 8: worst gradient cosine {worst[0]:.5f} ({worst[1]})")
    enc.close()


def test_vit_b16_full_batch_training_step_properties():
    cfg = ViTConfig.vit_b16_224()
    batch = 256
    sd = synth.synthetic_state_dict(cfg, seed=0, device=DEV)
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV)
    enc.load_state_dict(sd)
    tr = ViTTrainer(enc)
    px = synth.synthetic_images(cfg, batch, seed=2, device=DEV).to(torch.bfloat16)
    d_p = synth.normal((batch, cfg.hidden_size), 5, "dp", 0.5, device=DEV).to(torch.bfloat16)
    mats = [k for k, v in sd.items() if v.dim() >= 2 and "embeddings" not in k]
    assert len(mats) == 12 * 6

    def grads_of(pixels, d_pooled):
        out = tr.forward(pixels)
        tr.backward(None, d_pooled)
        torch.cuda.synchronize()
        return out, tr.grads.clone()

    out, g = grads_of(px, d_p)
    assert torch.isfinite(out.last_hidden_state.float()).all() and torch.isfinite(g).all()
    assert g.abs().max() > 0
    # (1) bit-reproducible weight matrices (fixed-order split-K sums, no atomics)
    _, g2 = grads_of(px, d_p)
    for k in mats:
        v = tr.grad(k)
        lo = v.storage_offset()
        sl = slice(lo, lo + v.shape[0] * v.stride(0))
        assert torch.equal(g[sl], g2[sl]), k
    # (2) order of the images does not matter (fp32 sums in another order: rounding only)
    perm = torch.randperm(batch, generator=torch.Generator().manual_seed(0)).to(DEV)
    _, gp = grads_of(px[perm].contiguous(), d_p[perm].contiguous())
    rel = ((gp - g).norm() / g.norm()).item()
    assert rel < 2e-3, rel
    # (3) additivity over images: grad(batch) = grad(first half) + grad(second half)
    _, ga = grads_of(px[:128].contiguous(), d_p[:128].contiguous())
    _, gb = grads_of(px[128:].contiguous(), d_p[128:].contiguous())
    rel = ((ga + gb - g).norm() / g.norm()).item()
    assert rel < 2e-3, rel
    print(f"ViT-B/16 batch 256 training step: |grad| = {g.norm().item():.4f}, half-sum residual {rel:.2e}")
    # (4) one SGD step lowers the (linear) objective <pooled, d_p> along -gradient
    before = (out.pooler_output.float() * d_p.float()).sum().item()
    tr.forward(px)
    tr.backward(None, d_p)
    tr.sgd_step(1e-4)
    after = (tr.forward(px).pooler_output.float() * d_p.float()).sum().item()
    assert after < before, (before, after)
    enc.close()
