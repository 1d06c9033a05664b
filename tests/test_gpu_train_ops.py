"""GPU parity tests of the training-step kernels against torch autograd (fp32, CPU) on the
same bf16-rounded inputs. The reference has no training code: the oracle for
each op is the derivative of the oracle's forward definition, taken by torch.autograd.

Bars: bf16 outputs -- relative error (max-abs / max-abs-ref) <= 2e-2 and cosine >= 0.9995;
fp32 outputs (parameter gradients) -- relative error <= 5e-3 (inputs are bf16, sums are fp32).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import cosine, rel_err
from vit_amd import ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def rnd(shape, name, std=1.0, mean=0.0):
    return synth.normal(shape, 21, name, std=std, mean=mean).to(BF)


def close_bf16(got, ref, tol=2e-2):
    assert rel_err(got.float(), ref) <= tol, rel_err(got.float(), ref)
    assert cosine(got.float(), ref) >= 0.9995


@pytest.mark.parametrize("m,n,k", [(300, 256, 128), (1000, 520, 640), (4096, 768, 768)])
@pytest.mark.parametrize("act", ["gelu", "quick_gelu"])
def test_linear_train_and_dgrad_act(m, n, k, act):
    x, w, b = rnd((m, k), "x"), rnd((n, k), "w", 0.05), synth.normal((n,), 21, "b", 0.1)
    out, pre = ops.linear_train(x.to(DEV), w.to(DEV), b.to(DEV), act)
    ref_pre = x.float() @ w.float().t() + b
    ref = F.gelu(ref_pre) if act == "gelu" else ref_pre * torch.sigmoid(1.702 * ref_pre)
    close_bf16(pre.cpu(), ref_pre)
    close_bf16(out.cpu(), ref)
    # backward of fc2 with the activation derivative fused: d_pre = (dy W2) * act'(pre)
    dy, w2 = rnd((m, k), "dy"), rnd((k, n), "w2", 0.05)          # fc2: [k out, n in]
    pre_b = pre.cpu().float().requires_grad_(True)
    h = F.gelu(pre_b) if act == "gelu" else pre_b * torch.sigmoid(1.702 * pre_b)
    (h @ w2.float().t()).backward(dy.float())
    got = ops.linear_dgrad(dy.to(DEV), w2.t().contiguous().to(DEV), pre, act)
    close_bf16(got.cpu(), pre_b.grad)


@pytest.mark.parametrize("m,n,k", [(34, 64, 128), (394, 768, 256), (5000, 3072, 768)])
def test_wgrad_via_transposes(m, n, k):
    """dW = dY^T X and db = colsum(dY) from the transposed, zero-padded operands."""
    dy, x = rnd((m, n), "dy"), rnd((m, k), "x")
    dy_t, db = ops.transpose_pad(dy.to(DEV), colsum=True)
    x_t = ops.transpose_pad(x.to(DEV))
    assert dy_t.shape == (n, (m + 127) // 128 * 128)
    assert torch.equal(dy_t[:, :m].cpu(), dy.t())
    assert not dy_t[:, m:].any()
    dw = ops.linear_wgrad(dy_t, x_t)
    ref = dy.float().t() @ x.float()
    assert rel_err(dw, ref) <= 5e-3
    assert rel_err(db, dy.float().sum(0)) <= 5e-3


def test_transpose_skips_cls_rows():
    b, n, d = 3, 17, 64
    x = rnd((b, n, d), "x3")
    t = ops.transpose_pad(x.reshape(b * n, d).to(DEV), skip_first_of=n - 1)
    ref = x[:, 1:, :].reshape(b * (n - 1), d).t()
    assert torch.equal(t[:, : b * (n - 1)].cpu(), ref)


@pytest.mark.parametrize("rows,d", [(50, 64), (999, 768), (300, 1280)])
@pytest.mark.parametrize("with_res", [False, True])
def test_layernorm_backward(rows, d, with_res):
    x, dy = rnd((rows, d), "x", 1.5, 0.3), rnd((rows, d), "dy")
    g, bt = synth.normal((d,), 21, "g", 0.2, 1.0), synth.normal((d,), 21, "bt", 0.1)
    dres = rnd((rows, d), "dres") if with_res else None
    xr, gr, br = x.float().requires_grad_(True), g.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    F.layer_norm(xr, (d,), gr, br, 1e-5).backward(dy.float())
    ref_dx = xr.grad + (dres.float() if with_res else 0)
    dx, dg, db = ops.layer_norm_backward(x.to(DEV), dy.to(DEV), g.to(DEV), 1e-5,
                                         None if dres is None else dres.to(DEV))
    close_bf16(dx.cpu(), ref_dx)
    assert rel_err(dg, gr.grad) <= 5e-3
    assert rel_err(db, br.grad) <= 5e-3


@pytest.mark.parametrize("b,n,h,dh", [(2, 17, 4, 16), (2, 197, 3, 64), (1, 577, 2, 64), (1, 300, 2, 80), (2, 100, 2, 128),
                                      # one-pass kernel (head_dim 64, 97..224 tokens): smallest, largest, exact tile multiple,
                                      # the first length past it, and more (batch, head) items than CUs
                                      (2, 97, 2, 64), (1, 224, 2, 64), (2, 192, 1, 64), (1, 225, 2, 64), (23, 197, 12, 64)])
def test_attention_backward(b, n, h, dh):
    d = h * dh
    qkv = rnd((b, n, 3 * d), "qkv", 1.0)
    d_out = rnd((b, n, d), "do", 1.0)
    out, lse = ops.attention_train(qkv.to(DEV), h)
    ref_qkv = qkv.float().requires_grad_(True)
    q, k, v = ref_qkv.reshape(b, n, 3, h, dh).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * dh ** -0.5
    ref_out = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(b, n, d)
    ref_out.backward(d_out.float())
    close_bf16(out.cpu(), ref_out.detach())
    ref_lse = torch.logsumexp(s.detach(), -1) / math.log(2.0)
    assert (lse.cpu() - ref_lse).abs().max() <= 2e-2
    dqkv = ops.attention_backward(qkv.to(DEV), out, d_out.to(DEV), lse, h)
    ref = ref_qkv.grad
    for j, name in enumerate("qkv"):
        got_j, ref_j = dqkv.cpu().float()[..., j * d:(j + 1) * d], ref[..., j * d:(j + 1) * d]
        assert rel_err(got_j, ref_j) <= 3e-2, (name, rel_err(got_j, ref_j))
        assert cosine(got_j, ref_j) >= 0.999, (name, cosine(got_j, ref_j))
    # bitwise reproducible
    assert torch.equal(dqkv, ops.attention_backward(qkv.to(DEV), out, d_out.to(DEV), lse, h))


@pytest.mark.parametrize("tokens,out_f,in_f,splits", [
    (51, 64, 128, 1),        # fewer tokens than one 64-token tile: rows past T must read as zero
    (64, 8, 8, 1),           # smallest legal features
    (50, 264, 520, 1), (64, 256, 256, 1), (130, 256, 512, 2),  # two or three 32-token slots per split: the ring's prologue covers the whole loop
    (197 * 3, 768, 768, 1),  # ragged tokens, exact 256-tiles
    (1000, 264, 520, 3),     # ragged tiles in both feature dims, splits that do not divide the token count
    (4096, 2304, 768, 4), (8320, 768, 3072, 7), (8320, 3072, 768, 5),
])
def test_linear_wgrad_token_major(tokens, out_f, in_f, splits):
    """vit_linear_wgrad (transposing MFMA kernel) against fp32 dY^T X / column sums on the same bf16 operands."""
    import torch
    from vit_amd import _lib, synth
    lib = _lib.load()
    dev = "cuda:0"
    dy = synth.normal((tokens, out_f), 5, "dy").to(dev, torch.bfloat16)
    x = synth.normal((tokens, in_f), 5, "x").to(dev, torch.bfloat16)
    dw = torch.full((out_f, in_f), float("nan"), device=dev)
    db = torch.ones(out_f, device=dev)  # accumulated into
    slabs = torch.full((splits * out_f * in_f,), float("nan"), device=dev)
    _lib.check(lib.vit_linear_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), slabs.data_ptr(), tokens,
                                    out_f, in_f, splits, None))
    torch.cuda.synchronize()
    want = dy.float().t() @ x.float()
    wantb = 1.0 + dy.float().sum(0)
    scale = want.abs().max().item()
    assert (dw - want).abs().max().item() <= 2e-4 * scale + 1e-3  # fp32 accumulation of exact bf16 products
    assert (db - wantb).abs().max().item() <= 2e-4 * wantb.abs().max().item() + 1e-3
    # reproducible: the slab sum runs in a fixed order (bias uses atomics and is excluded)
    dw2 = torch.empty_like(dw)
    _lib.check(lib.vit_linear_wgrad(dy.data_ptr(), x.data_ptr(), dw2.data_ptr(), None, slabs.data_ptr(), tokens,
                                    out_f, in_f, splits, None))
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2)
