"""GPU parity of the whole training step (vit_forward_train / vit_backward_stage / vit_sgd_step
through the C ABI) against torch.autograd of the CPU oracle on the same bf16-rounded weights.

Bars: every parameter gradient cosine >= 0.99 against the fp32 oracle gradient (bf16 activations,
fp32 accumulation) with the worst one printed; a step of SGD moves the loss the same way.
"""
import pytest
import torch

from oracle import vit_oracle
from tests.helpers import cosine, load_golden, min_row_cosine, rel_err
from vit_amd import synth
from vit_amd.encoder import ViTEncoder
from vit_amd.trainer import ViTTrainer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def oracle_grads(cfg, sd, px, w_h, w_p):
    sdg = {k: v.clone().float().requires_grad_(True) for k, v in sd.items()}
    hidden, pooled = vit_oracle.forward_autograd(px, sdg, cfg)
    loss = (hidden * w_h).sum() + (pooled * w_p).sum()
    loss.backward()
    return loss.item(), {k: v.grad for k, v in sdg.items()}, hidden.detach(), pooled.detach()


@pytest.mark.parametrize("name", ["tiny_vit_p8", "tiny_vit_p14_dh80", "tiny_clip_p14", "tiny_vit_p16_n197"])
def test_gradients_match_autograd(name):
    cfg, batch, _ = load_golden(name)
    sd = vit_oracle.quantize_state_dict_bf16(synth.synthetic_state_dict(cfg, seed=0))
    px = synth.synthetic_images(cfg, batch, seed=1).to(torch.bfloat16).float()
    w_h = synth.normal((batch, cfg.seq_len, cfg.hidden_size), 5, "wh", 0.05).to(torch.bfloat16).float()
    w_p = synth.normal((batch, cfg.hidden_size), 5, "wp", 0.5).to(torch.bfloat16).float()
    _, ref, ref_h, ref_p = oracle_grads(cfg, sd, px, w_h, w_p)

    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV)
    enc.load_state_dict(sd)
    tr = ViTTrainer(enc)
    out = tr.forward(px)
    assert min_row_cosine(out.pooler_output, ref_p) >= 0.999
    assert min_row_cosine(out.last_hidden_state, ref_h) >= 0.998
    # the stashing forward is the same function as the inference forward; it materialises the LayerNorm outputs
    # (the weight gradients need them) where the inference forward folds LayerNorm into the next GEMM, so the
    # two differ by bf16 rounding points only
    inf = enc(px)
    assert rel_err(inf.last_hidden_state, out.last_hidden_state) < 2e-2
    assert min_row_cosine(inf.last_hidden_state, out.last_hidden_state.float().cpu()) >= 0.9995
    tr.backward(w_h, w_p)
    torch.cuda.synchronize()
    worst = (1.0, "")
    for k, g in ref.items():
        got = tr.grad(k).reshape(g.shape).cpu()
        if k.endswith("k_proj.bias"):
            # softmax is invariant to a shift of all scores of a query, so this gradient is exactly 0
            # in exact arithmetic: only check that what bf16 leaves is small
            vb = ref[k.replace("k_proj", "v_proj")]
            assert got.abs().max() <= 0.05 * vb.abs().max(), (k, got.abs().max().item(), vb.abs().max().item())
            continue
        c = cosine(got, g)
        worst = min(worst, (c, k))
        rel = ((got - g).norm() / g.norm().clamp_min(1e-20)).item()
        assert c >= 0.99, (k, c, rel)
    print(f"{name}: worst gradient cosine {worst[0]:.5f} ({worst[1]})")
    # reproducible
    g1 = tr.grads.clone()
    tr.backward(w_h, w_p)
    torch.cuda.synchronize()
    for k, g in ref.items():  # matrix gradients take no atomics: bitwise equal run to run
        if g.dim() >= 2 and "position" not in k and "cls" not in k and "class" not in k:
            o = tr.grad(k)
            lo = o.storage_offset()
            assert torch.equal(o, g1[lo:lo + o.shape[0] * o.stride(0)].view(o.shape[0], o.stride(0))[:, :o.shape[1]]), k
    assert torch.allclose(tr.grads, g1, rtol=1e-3, atol=1e-5)


def test_sgd_step_decreases_loss_like_oracle():
    cfg, batch, _ = load_golden("tiny_vit_p8")
    sd = vit_oracle.quantize_state_dict_bf16(synth.synthetic_state_dict(cfg, seed=0))
    px = synth.synthetic_images(cfg, batch, seed=1).to(torch.bfloat16).float()
    target = synth.normal((batch, cfg.hidden_size), 6, "tgt", 1.0)

    def loss_and_grad(pooled):
        diff = pooled.float().cpu() - target
        return 0.5 * (diff ** 2).sum().item(), diff

    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV)
    enc.load_state_dict(sd)
    tr = ViTTrainer(enc)
    losses = []
    for _ in range(4):
        out = tr.forward(px)
        l, d = loss_and_grad(out.pooler_output)
        losses.append(l)
        tr.backward(None, d.to(torch.bfloat16))
        tr.sgd_step(2e-3)
    assert losses[-1] < losses[0] * 0.9, losses
    # oracle SGD in fp32 from the same start
    sdo = {k: v.clone().float().requires_grad_(True) for k, v in sd.items()}
    lo = []
    for _ in range(4):
        _, pooled = vit_oracle.forward_autograd(px, sdo, cfg)
        loss = 0.5 * ((pooled - target) ** 2).sum()
        lo.append(loss.item())
        grads = torch.autograd.grad(loss, list(sdo.values()))
        with torch.no_grad():
            for p, g in zip(sdo.values(), grads):
                p -= 2e-3 * g
    print("losses gpu", [round(x, 3) for x in losses], "oracle", [round(x, 3) for x in lo])
    for a, b in zip(losses, lo):
        assert abs(a - b) <= 0.05 * abs(b) + 1e-3


def test_split_k_weight_gradients_match_autograd():
    """Token count >= 8192: the weight-gradient GEMMs run split-K on the persistent kernel and the
    data-gradient GEMMs on its activation-gradient epilogue (the full-size code path)."""
    from vit_amd.config import ViTConfig
    cfg = ViTConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=1024,
                    image_size=64, patch_size=8)
    batch = 128  # 128 * 65 = 8320 tokens
    sd = vit_oracle.quantize_state_dict_bf16(synth.synthetic_state_dict(cfg, seed=0))
    px = synth.synthetic_images(cfg, batch, seed=1).to(torch.bfloat16).float()
    w_p = synth.normal((batch, cfg.hidden_size), 5, "wp", 0.5).to(torch.bfloat16).float()
    w_h = torch.zeros(batch, cfg.seq_len, cfg.hidden_size)
    _, ref, _, ref_p = oracle_grads(cfg, sd, px, w_h, w_p)
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV)
    enc.load_state_dict(sd)
    tr = ViTTrainer(enc)
    out = tr.forward(px)
    assert min_row_cosine(out.pooler_output, ref_p) >= 0.999
    tr.backward(None, w_p)
    torch.cuda.synchronize()
    worst = (1.0, "")
    for k, g in ref.items():
        if k.endswith("k_proj.bias"):
            continue
        c = cosine(tr.grad(k).reshape(g.shape).cpu(), g)
        worst = min(worst, (c, k))
        assert c >= 0.99, (k, c)
    print(f"split-K path: worst gradient cosine {worst[0]:.5f} ({worst[1]})")
    # this size also puts the weight gradients on the side stream (vit_train.hip, layer stage): their matrices are
    # summed in a fixed order, so a second step must reproduce them bit for bit -- a missing event between the two
    # streams would show up here as a difference
    first = {k: tr.grad(k).clone() for k, g in ref.items() if g.dim() >= 2 and "embeddings" not in k}
    assert len(first) >= 8
    for _ in range(2):
        tr.forward(px)
        tr.backward(None, w_p)
        torch.cuda.synchronize()
        for k, g in first.items():
            assert torch.equal(tr.grad(k), g), k


def test_two_stashes_in_flight_and_unknown_stash_rejected():
    """Micro-batching: two vit_forward_train calls into two stashes, then their backwards in the other order. Each
    backward must give the gradients of a plain forward + backward of the same micro-batch bit for bit (matrix
    gradients take no atomics), and a stash that no vit_forward_train of this context filled is refused."""
    from vit_amd import _lib
    cfg, batch, _ = load_golden("tiny_vit_p16_n197")
    sd = vit_oracle.quantize_state_dict_bf16(synth.synthetic_state_dict(cfg, seed=0))
    px = [synth.synthetic_images(cfg, batch, seed=s).to(torch.bfloat16) for s in (1, 2)]
    w_p = synth.normal((batch, cfg.hidden_size), 5, "wp", 0.5).to(torch.bfloat16)
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV)
    enc.load_state_dict(sd)
    tr = ViTTrainer(enc)
    key = ("layers.0.mlp.fc1.weight", "layers.0.attention.q_proj.weight", "layers.0.attention.o_proj.weight")
    want = []
    for p in px:
        tr.forward(p)
        tr.backward(None, w_p)
        torch.cuda.synchronize()
        want.append({k: tr.grad(k).clone() for k in key})
    held = []
    for p in px:
        tr._stash = None  # a fresh stash per micro-batch, both alive
        tr.forward(p)
        held.append((tr._stash, tr._px))
    for i in (1, 0):
        tr._stash, tr._px = held[i]
        tr.backward(None, w_p)
        torch.cuda.synchronize()
        for k in key:
            assert torch.equal(tr.grad(k), want[i][k]), (i, k)
    tr._stash = torch.empty_like(held[0][0])
    with pytest.raises(_lib.VitError, match="was not filled"):
        tr.backward(None, w_p)
