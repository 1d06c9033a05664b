"""GEMM parity AT THE BENCH SHAPES (BASELINE.json configs[1] / [4]: M = 256 * 197 = 50432 token rows, and a
ragged M = 50431), every epilogue of the big-layer kernel that vit_forward / the training step launch there --
bias, GELU, quick-GELU, residual, the folded-LayerNorm variants, the training second output (pre-activation),
the activation-gradient epilogues and the fp32 output -- through the C ABI.

References: (1) oracle/vit_oracle.py on the CPU for a sample of rows (first tile, last ragged tile, a stride
through the middle; rows of a Linear are independent), (2) a plain fp32 torch matmul of the same bf16-rounded
operands for the whole matrix. Bar: 2 bf16 ulps of the output scale (max-abs error / max-abs reference
<= 2^-7), inputs rounded to bf16 first so that only the kernel's arithmetic is measured.
"""
import math

import pytest
import torch

from oracle import vit_oracle
from tests.helpers import rel_err
from vit_amd import ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16_TOL = 1.0 / 128
SHAPES = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]  # (N, K): q/k/v, o_proj, fc1, fc2 of ViT-B/16
MS = [50432, 50431]
# configs[2] / [3] at the batch they are benched at: CLIP-L/14-336 batch 128 (M = 128 * 577) and ViT-H/14-448 batch 64
# (M = 64 * 1025), each with a ragged M as well; (M, N, K) for q/k/v, o_proj, fc1, fc2
L14_M, H14_M = 128 * 577, 64 * 1025
BIG = [(m, n, k) for m in (L14_M, L14_M - 1) for (n, k) in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096))] + \
      [(m, n, k) for m in (H14_M, H14_M - 1) for (n, k) in ((3840, 1280), (1280, 1280), (5120, 1280), (1280, 5120))]
ALL_MNK = [(m, n, k) for (n, k) in SHAPES for m in MS] + BIG


def gen(shape, name, std=1.0):
    return synth.normal(shape, 23, name, std=std, device=DEV).to(torch.bfloat16)


def sample_rows(m):
    idx = list(range(0, 300)) + list(range(m - 300, m)) + list(range(300, m - 300, 997))
    return torch.tensor(idx)


def full_ref(x, w, bias):
    y = x.float() @ w.float().t()
    return y if bias is None else y + bias


def check(out, ref_full, x, w, bias, post, rows):
    """out vs the whole-matrix fp32 torch reference and vs the CPU oracle on `rows`."""
    assert torch.isfinite(out.float()).all()
    assert rel_err(out, post(ref_full)) < BF16_TOL
    xr, wr = x[rows.to(DEV)].float().cpu(), w.float().cpu()
    o = vit_oracle.linear(xr, wr, None if bias is None else bias.cpu())
    o = post(o, rows)
    assert rel_err(out[rows.to(DEV)], o) < BF16_TOL


@pytest.mark.parametrize("m,n,k", ALL_MNK)
def test_forward_epilogues_at_bench_shapes(m, n, k):
    x, w = gen((m, k), "x"), gen((n, k), "w", 1 / math.sqrt(k))
    bias = synth.normal((n,), 23, "b", 0.5, device=DEV)
    res = gen((m, n), "r")
    rows = sample_rows(m)
    ref = full_ref(x, w, bias)

    def dev_or_cpu(t, like):
        return t.to(like.device)

    check(ops.linear(x, w, bias), ref, x, w, bias, lambda y, r=None: y, rows)
    check(ops.linear(x, w, None), full_ref(x, w, None), x, w, None, lambda y, r=None: y, rows)
    for act in ("gelu", "quick_gelu"):
        check(ops.linear(x, w, bias, activation=act), ref, x, w, bias, lambda y, r=None, a=act: vit_oracle.activation(y, a), rows)
    check(ops.linear(x, w, bias, residual=res), ref, x, w, bias,
          lambda y, r=None: y + (res.float() if r is None else res[r.to(DEV)].float().cpu()), rows)
    # training forward: activation output plus the pre-activation as a second output
    out, pre = ops.linear_train(x, w, bias, "gelu")
    check(pre, ref, x, w, bias, lambda y, r=None: y, rows)
    check(out, ref, x, w, bias, lambda y, r=None: vit_oracle.activation(y, "gelu"), rows)


@pytest.mark.parametrize("m,n,k", [(m, n, k) for (n, k) in ((2304, 768), (3072, 768)) for m in MS] +
                         [(m, n, k) for m in (L14_M, L14_M - 1) for (n, k) in ((3072, 1024), (4096, 1024))] +
                         [(m, n, k) for m in (H14_M, H14_M - 1) for (n, k) in ((3840, 1280), (5120, 1280))])
def test_folded_layernorm_gemm_at_bench_shapes(m, n, k):
    """LN1 -> q/k/v and LN2 -> fc1 as vit_forward runs them: statistics pass + GEMM on the un-normalised rows."""
    x = (gen((m, k), "x").float() * 1.5 + 0.3).to(torch.bfloat16)
    w, bias = gen((n, k), "w", 1 / math.sqrt(k)), synth.normal((n,), 23, "b", 0.5, device=DEV)
    g, b = synth.normal((k,), 23, "g", 0.1, 1.0, device=DEV), synth.normal((k,), 23, "beta", 0.1, device=DEV)
    eps = 1e-5 if k == 1024 else 1e-12  # CLIP / ViT (configuration_clip.py:106, configuration_vit.py:58)
    stats = ops.layer_norm_stats(x, eps)
    folded = ops.layer_norm_fold(w, g, b, bias)
    rows = sample_rows(m)
    xr = x[rows.to(DEV)].float().cpu()
    ln = vit_oracle.layer_norm(xr, g.cpu(), b.cpu(), eps)
    lin = vit_oracle.linear(ln, w.float().cpu(), bias.cpu())
    ln_full = torch.nn.functional.layer_norm(x.float(), (k,), g, b, eps)
    lin_full = ln_full @ w.float().t() + bias
    for act in (None, "gelu", "quick_gelu"):
        out = ops.linear_ln_folded(x, folded, stats, act)
        post = (lambda y: y) if act is None else (lambda y, a=act: vit_oracle.activation(y, a))
        assert torch.isfinite(out.float()).all()
        # the folded path rounds Wg = W * gamma to bf16 where the reference multiplies in fp32: one more bf16
        # rounding on the weights, so 3 ulps of the output scale
        assert rel_err(out[rows.to(DEV)], post(lin)) < 1.5 * BF16_TOL
        assert rel_err(out, post(lin_full)) < 1.5 * BF16_TOL


@pytest.mark.parametrize("n,k,act", [(768, 3072, "gelu"), (768, 3072, "quick_gelu"), (768, 768, None), (768, 2304, None)])
@pytest.mark.parametrize("m", MS)
def test_backward_epilogues_at_bench_shapes(m, n, k, act):
    """Data gradients dX = dY W (fc2 with act'(pre-activation) fused; o_proj; q/k/v) on the transposed weights."""
    dy, wt = gen((m, k), "dy"), gen((n, k), "wt", 1 / math.sqrt(k))
    rows = sample_rows(m)
    ref = full_ref(dy, wt, None)
    if act is None:
        check(ops.linear_dgrad(dy, wt), ref, dy, wt, None, lambda y, r=None: y, rows)
        return
    pre = gen((m, n), "pre", 1.5)
    z = pre.float().requires_grad_(True)
    vit_oracle.activation(z, act).sum().backward()
    dact = z.grad
    out = ops.linear_dgrad(dy, wt, pre, act)
    check(out, ref, dy, wt, None, lambda y, r=None: y * (dact if r is None else dact[r.to(DEV)].cpu()), rows)


def test_fp32_output_at_bench_shapes():
    """Weight-gradient-shaped product on the persistent kernel's fp32 output: [768, tokens] x [3072, tokens]^T."""
    tokens = 50432
    a, b = gen((768, tokens), "a"), gen((3072, tokens), "b", 1 / math.sqrt(tokens))
    out = ops.linear_wgrad(a, b)
    ref = a.float() @ b.float().t()
    assert out.dtype == torch.float32
    assert rel_err(out, ref) < 1e-4
    rows = torch.arange(0, 768, 37)
    o = vit_oracle.linear(a[rows.to(DEV)].float().cpu(), b.float().cpu(), None)
    assert rel_err(out[rows.to(DEV)], o) < 1e-4
