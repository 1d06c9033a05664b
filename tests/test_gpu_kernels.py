"""GPU parity tests, per kernel, through the C ABI, against oracle/vit_oracle.py.

Tolerances. fp32 mode: max-abs error / max-abs reference <= 1e-4 (BASELINE.json's fp32 bar).
bf16 mode: inputs are rounded to bf16 first and the oracle runs on the ROUNDED inputs, so what
is measured is the kernel's arithmetic (fp32 accumulate, one bf16 rounding of the output):
bound 2^-8 relative to the output scale plus accumulated-input noise, stated per test.
"""
import math

import pytest
import torch

from oracle import vit_oracle
from tests.helpers import rel_err, min_row_cosine
from vit_amd import ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16_TOL = 1.0 / 128  # 2 bf16 ulps of the output scale


def rnd(shape, name, std=1.0, seed=7):
    return synth.normal(shape, seed, name, std=std)


def q(t):  # what a bf16 kernel actually sees
    return t.to(torch.bfloat16).float()


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("rows,dim", [(1, 64), (5, 160), (197, 768), (1000, 1024), (77, 1280), (3, 2048), (9, 4)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("eps", [1e-12, 1e-5])
def test_layernorm(rows, dim, dtype, eps):
    x = rnd((rows, dim), "ln.x") * 3 + 0.5
    w, b = rnd((dim,), "ln.w", 0.1) + 1, rnd((dim,), "ln.b", 0.1)
    if dtype == torch.bfloat16:
        x = q(x)
    ref = vit_oracle.layer_norm(x, w, b, eps)
    out = ops.layer_norm(x.to(DEV, dtype), w.to(DEV), b.to(DEV), eps)
    assert out.dtype == dtype and out.shape == ref.shape
    assert rel_err(out, ref) < (1e-5 if dtype == torch.float32 else BF16_TOL)


def test_layernorm_constant_rows_and_3d_input():
    # zero variance with eps=1e-12 (ViT's eps): output must be exactly beta, not NaN
    x = torch.full((2, 3, 768), 2.5)
    w, b = rnd((768,), "w") + 1, rnd((768,), "b")
    out = ops.layer_norm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-12)
    assert torch.isfinite(out).all()
    assert rel_err(out, b.expand(2, 3, 768)) < 1e-6


# ------------------------------------------------------------------ GEMM + epilogues
GEMM_SHAPES = [
    (1, 8, 8), (17, 64, 64), (51, 192, 64), (20, 320, 160), (20, 160, 320),  # tiny / ragged K (160 % 64 != 0)
    (394, 768, 768), (394, 2304, 768), (300, 768, 3072), (513, 3072, 768),   # ViT-B layer shapes, ragged M
    (256, 128, 64), (257, 136, 72),                                        # BN=128 path, N not multiple of tile
]


@pytest.mark.parametrize("m,n,k", GEMM_SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("epi", ["none", "gelu", "quick_gelu", "residual", "nobias"])
def test_linear(m, n, k, dtype, epi):
    x, w = rnd((m, k), "g.x"), rnd((n, k), "g.w", 1 / math.sqrt(k))
    bias = None if epi == "nobias" else rnd((n,), "g.b", 0.5)
    res = rnd((m, n), "g.r") if epi == "residual" else None
    if dtype == torch.bfloat16:
        x, w = q(x), q(w)
        res = None if res is None else q(res)
    ref = x @ w.t()
    if bias is not None:
        ref = ref + bias
    if epi in ("gelu", "quick_gelu"):
        ref = vit_oracle.activation(ref, epi)
    if res is not None:
        ref = ref + res
    out = ops.linear(x.to(DEV, dtype), w.to(DEV, dtype), None if bias is None else bias.to(DEV),
                     activation=epi if epi in ("gelu", "quick_gelu") else None,
                     residual=None if res is None else res.to(DEV, dtype))
    assert out.dtype == dtype and tuple(out.shape) == (m, n)
    assert rel_err(out, ref) < (1e-5 if dtype == torch.float32 else BF16_TOL)


@pytest.mark.parametrize("m,n,k", [(5001, 1000, 256), (4200, 1096, 384)])
@pytest.mark.parametrize("epi", ["none", "gelu", "quick_gelu", "residual", "nobias"])
def test_linear_persistent_kernel_ragged_columns(m, n, k, epi):
    # M * N >= 64 tiles and K % 128 == 0: the persistent kernel, with a ragged last tile ROW and a ragged last tile
    # COLUMN (n % 256 = 232 / 72): its bias / folded terms travel through LDS by a DMA whose out-of-range lanes must
    # read as zero, and the stores of columns >= n must be dropped
    x, w = q(rnd((m, k), "rg.x")), q(rnd((n, k), "rg.w", 1 / math.sqrt(k)))
    bias = None if epi == "nobias" else rnd((n,), "rg.b", 0.5)
    res = q(rnd((m, n), "rg.r")) if epi == "residual" else None
    ref = x @ w.t()
    if bias is not None:
        ref = ref + bias
    if epi in ("gelu", "quick_gelu"):
        ref = vit_oracle.activation(ref, epi)
    if res is not None:
        ref = ref + res
    out = ops.linear(x.to(DEV, torch.bfloat16), w.to(DEV, torch.bfloat16), None if bias is None else bias.to(DEV),
                     activation=epi if epi in ("gelu", "quick_gelu") else None,
                     residual=None if res is None else res.to(DEV, torch.bfloat16))
    assert tuple(out.shape) == (m, n)
    assert rel_err(out, ref) < BF16_TOL
    # the last rows / columns on their own (a wrong edge hides in a whole-matrix norm)
    assert rel_err(out[-40:, -80:], ref[-40:, -80:]) < 2 * BF16_TOL


def test_linear_identity_asymmetric():
    # A = I with an asymmetric W catches transposed / permuted output maps exactly
    n = 256
    w = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251) - 125.0   # exact in bf16
    eye = torch.eye(n)
    for dtype in (torch.float32, torch.bfloat16):
        out = ops.linear(eye.to(DEV, dtype), w.to(DEV, dtype))
        assert torch.equal(out.float().cpu(), w.t())


def test_linear_residual_in_place_semantics():
    # the encoder calls the residual epilogue with C aliasing the residual
    from vit_amd import _lib
    m, n, k = 300, 768, 768
    x, w, r = q(rnd((m, k), "x")), q(rnd((n, k), "w", 0.03)), q(rnd((m, n), "r"))
    xd, wd, rd = x.to(DEV, torch.bfloat16), w.to(DEV, torch.bfloat16), r.to(DEV, torch.bfloat16)
    _lib.check(_lib.load().vit_gemm_bias_act(xd.data_ptr(), wd.data_ptr(), None, rd.data_ptr(), rd.data_ptr(), m, n, k,
                                             _lib.EPI_RESIDUAL, _lib.VIT_BF16, torch.cuda.current_stream().cuda_stream))
    assert rel_err(rd, x @ w.t() + r) < BF16_TOL


def test_linear_rejects_bad_shapes():
    from vit_amd import _lib
    x = torch.zeros(4, 12, device=DEV, dtype=torch.bfloat16)
    w = torch.zeros(8, 12, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(_lib.VitError):
        ops.linear(x, w)  # K % 8 != 0 in bf16
    with pytest.raises(ValueError):
        ops.linear(x, torch.zeros(8, 16, device=DEV, dtype=torch.bfloat16))


# ------------------------------------------------------------------ attention
ATTN_CASES = [  # batch, seq, heads, head_dim
    (2, 10, 2, 80), (3, 17, 4, 16), (2, 197, 12, 64), (1, 577, 4, 64), (1, 1025, 2, 80), (2, 33, 2, 128),
    (1, 1, 1, 32), (2, 32, 3, 64), (1, 225, 2, 64), (1, 300, 1, 48),
    # head_dim 64 persistent LDS-DMA kernel: the eligibility edge (N = 96 falls back, 97 is the first
    # taken), exact tile multiples, one and several query blocks with idle waves, more items than
    # resident workgroups (37*16 = 592 items > 512) and fewer items than workgroups
    (2, 96, 2, 64), (2, 97, 2, 64), (3, 128, 2, 64), (2, 129, 3, 64), (2, 256, 2, 64), (2, 257, 2, 64),
    (37, 197, 16, 64), (1, 1025, 3, 64), (5, 161, 7, 64),
    # head_dim 80 ring kernel, one case per workgroup size its policy picks (tiles = ceil(n / 32)):
    # 7 tiles -> 8 waves, 9 -> 9, 10 -> 10, 11 -> 11, 16 -> 8 (two full items), 21 -> 11 (two items), 33 -> 12
    (3, 193, 2, 80), (2, 257, 3, 80), (2, 320, 2, 80), (2, 340, 2, 80), (1, 512, 2, 80), (1, 660, 2, 80), (2, 1025, 2, 80),
]


def ref_attention(qkv, heads):
    b, n, d3 = qkv.shape
    d = d3 // 3
    dh = d // heads
    qh, kh, vh = [t.reshape(b, n, heads, dh).transpose(1, 2) for t in qkv.split(d, dim=-1)]
    p = torch.softmax((qh @ kh.transpose(2, 3)) * dh ** -0.5, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(b, n, d)


@pytest.mark.parametrize("b,n,h,dh", ATTN_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention(b, n, h, dh, dtype):
    qkv = rnd((b, n, 3 * h * dh), "a.qkv", 1.5)
    if dtype == torch.bfloat16:
        qkv = q(qkv)
    ref = ref_attention(qkv, h)
    out = ops.attention(qkv.to(DEV, dtype), h)
    assert out.dtype == dtype and out.shape == ref.shape
    # bf16: P is rounded to bf16 before PV (as torch's .to(query.dtype) does, modeling_vit.py:183)
    assert rel_err(out, ref) < (2e-5 if dtype == torch.float32 else 2 * BF16_TOL)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_online_softmax_rescale_branch(dtype):
    # Force the running max to jump at a LATER key tile/chunk (a spiked K row far into the
    # sequence): exercises the rescale of O and l, which bounded random data rarely does.
    b, n, h, dh = 1, 577, 2, 64
    qkv = rnd((b, n, 3 * h * dh), "spike", 0.5)
    qkv[0, 500, h * dh: 2 * h * dh] *= 25.0   # key 500 of both heads (3rd LDS chunk)
    qkv[0, 70, h * dh: 2 * h * dh] *= 12.0    # and a smaller one in tile 2
    if dtype == torch.bfloat16:
        qkv = q(qkv)
    ref = ref_attention(qkv, h)
    out = ops.attention(qkv.to(DEV, dtype), h)
    assert torch.isfinite(out).all()
    assert rel_err(out, ref) < (2e-5 if dtype == torch.float32 else 2 * BF16_TOL)


# head_dim 80 ring kernel with MORE items than resident workgroups (nmy > 1: the item-boundary code -- the counted
# waits with the previous item's stores and the Q prefetch in flight, the DMA cursor crossing into the next item's
# K / V, the o / m / l reset): 8 waves x 640 items (two workgroups per CU: 512 resident), 12 waves x 576 items of
# 1025 tokens (the ViT-H/14 shape; 256 resident; the third query block has 9 of 12 waves with queries), 11 waves x 288
# items whose second query block has one wave without queries, 8 waves x 2 query blocks x 640 items.
RING_MULTI_ITEM = [(40, 193, 16, 80), (12, 1025, 16, 80), (9, 660, 16, 80), (20, 512, 16, 80)]


@pytest.mark.parametrize("b,n,h,dh", RING_MULTI_ITEM)
@pytest.mark.parametrize("train", [False, True])
def test_attention_ring_kernel_many_items_per_workgroup(b, n, h, dh, train):
    qkv = q(rnd((b, n, 3 * h * dh), "ring.qkv", 1.5))
    d = h * dh
    qh, kh, vh = [t.reshape(b, n, h, dh).transpose(1, 2) for t in qkv.split(d, dim=-1)]
    s = (qh @ kh.transpose(2, 3)) * dh ** -0.5
    ref = (torch.softmax(s, dim=-1) @ vh).transpose(1, 2).reshape(b, n, d)
    x = qkv.to(DEV, torch.bfloat16)
    if train:
        out, lse = ops.attention_train(x, h)
        ref_lse = torch.logsumexp(s, -1) / math.log(2.0)
        assert (lse.cpu() - ref_lse).abs().max() <= 2e-2
        out2, lse2 = ops.attention_train(x, h)
        assert torch.equal(lse, lse2)
    else:
        out = ops.attention(x, h)
        out2 = ops.attention(x, h)
    assert torch.equal(out, out2)  # repeatable: a miscounted wait shows as run-to-run differences first
    assert rel_err(out, ref) < 2 * BF16_TOL
    # per (batch, head) block, so that one bad item cannot hide in the whole-tensor norm
    err = (out.float().cpu() - ref).reshape(b, n, h, dh).abs().amax(dim=(1, 3))
    scale = ref.reshape(b, n, h, dh).abs().amax(dim=(1, 3))
    assert bool((err <= 2 * BF16_TOL * scale + 1e-3).all()), float((err / scale).max())


def test_attention_uniform_scores_average_values():
    # q = 0 -> uniform softmax -> every output row is the mean of V (exact property)
    b, n, h, dh = 2, 197, 3, 64
    qkv = rnd((b, n, 3 * h * dh), "u")
    qkv[..., : h * dh] = 0
    out = ops.attention(qkv.to(DEV), h)
    mean_v = qkv[..., 2 * h * dh:].mean(dim=1, keepdim=True).expand(b, n, h * dh)
    assert rel_err(out, mean_v) < 1e-5


# ------------------------------------------------------------------ patch embedding
@pytest.mark.parametrize("b,c,img,p,d", [(3, 3, 32, 8, 64), (2, 3, 42, 14, 160), (2, 3, 224, 16, 128),
                                          (1, 3, 336, 14, 64), (2, 1, 30, 5, 32), (5, 3, 224, 16, 768),
                                          # big enough for the persistent kernel's pixel loader (bf16): ragged last
                                          # row tile / image seams inside tiles, ragged last column tile, P = 32
                                          (30, 3, 224, 16, 768), (28, 3, 224, 16, 776), (86, 3, 256, 32, 768)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_bias", [True, False])
def test_patch_embed(b, c, img, p, d, dtype, with_bias):
    if dtype == torch.bfloat16 and d % 8:
        pytest.skip("bf16 needs D % 8 == 0")
    n = (img // p) ** 2 + 1
    px, w = rnd((b, c, img, img), "p.x"), rnd((d, c, p, p), "p.w", 0.05)
    bias = rnd((d,), "p.b", 0.3) if with_bias else None
    cls, pos = rnd((1, 1, d), "p.cls"), rnd((1, n, d), "p.pos")
    if dtype == torch.bfloat16:
        px, w = q(px), q(w)
    pe = vit_oracle.patch_embed(px, w, bias, p)
    ref = torch.cat([cls.expand(b, 1, d), pe], dim=1) + pos
    out = ops.patch_embed(px.to(DEV, dtype), w.to(DEV, dtype), None if bias is None else bias.to(DEV), cls.to(DEV),
                          pos.to(DEV), p)
    assert tuple(out.shape) == (b, n, d)
    assert rel_err(out, ref) < (1e-5 if dtype == torch.float32 else BF16_TOL)
    assert min_row_cosine(out, ref) > 0.9999


# ------------------------------------------------------------------ LayerNorm folded into the next Linear
def test_layer_norm_stats_matches_torch():
    x = q(rnd((1000, 768), "st.x") * 3 + 0.7)
    st = ops.layer_norm_stats(x.to(DEV, torch.bfloat16), 1e-6).cpu()
    mean = x.mean(-1)
    rstd = 1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-6)
    assert rel_err(st[:, 1], rstd) < 1e-5
    assert rel_err(st[:, 0], mean * rstd) < 1e-5


@pytest.mark.parametrize("m,n,k", [(300, 192, 128), (5600, 768, 256), (5555, 776, 384), (1576, 3072, 768)])
@pytest.mark.parametrize("act", [None, "gelu", "quick_gelu"])
def test_linear_ln_folded_matches_layernorm_then_linear(m, n, k, act):
    """vit_layernorm_stats + vit_layernorm_fold + vit_gemm_ln_folded == act(linear(layer_norm(x))) (fp32 torch),
    on the generic tile kernel (small) and the persistent kernel (big), ragged edges included; and the two
    kernels give the same bits for the same rows."""
    eps = 1e-6
    x = q(rnd((m, k), "f.x") * 2 + 0.5)
    w, b = q(rnd((n, k), "f.w", 0.05)), rnd((n,), "f.b", 0.3)
    g, be = 1 + 0.3 * rnd((k,), "f.g"), 0.2 * rnd((k,), "f.be")
    ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(x, (k,), g, be, eps), w, b)
    if act == "gelu":
        ref = torch.nn.functional.gelu(ref)
    elif act == "quick_gelu":
        ref = ref * torch.sigmoid(1.702 * ref)
    xd = x.to(DEV, torch.bfloat16)
    folded = ops.layer_norm_fold(w.to(DEV, torch.bfloat16), g.to(DEV), be.to(DEV), b.to(DEV))
    stats = ops.layer_norm_stats(xd, eps)
    out = ops.linear_ln_folded(xd, folded, stats, act)
    assert rel_err(out, ref) < BF16_TOL
    assert min_row_cosine(out, ref) > 0.9999
    # unfolded two-kernel path for scale: the folded one is at least as close to fp32
    two = ops.linear(ops.layer_norm(xd, g.to(DEV), be.to(DEV), eps), w.to(DEV), b.to(DEV), activation=act)
    assert rel_err(out, ref) <= 1.2 * rel_err(two, ref) + 1e-4
    # a small slice of rows runs on the generic kernel: same bits
    sub = ops.linear_ln_folded(xd[:40].contiguous(), folded, stats[:40].contiguous(), act)
    assert torch.equal(sub, out[:40])


def test_layer_norm_stats_and_folded_gemm_are_row_position_independent():
    """A row's statistics and its folded-GEMM output must not depend on where the row sits (which of a wave's
    row slots, which tile): permuted input rows give the permuted output bit for bit, and repeats are identical."""
    m, k, n = 25216, 768, 2304
    x = (rnd((m, k), "pp.x") * 2 + 0.5).to(DEV, torch.bfloat16)
    perm = torch.randperm(m, generator=torch.Generator().manual_seed(3)).to(DEV)
    st = ops.layer_norm_stats(x, 1e-12)
    assert torch.equal(ops.layer_norm_stats(x[perm].contiguous(), 1e-12), st[perm])
    assert torch.equal(ops.layer_norm_stats(x, 1e-12), st)
    folded = ops.layer_norm_fold(rnd((n, k), "pp.w", 0.02).to(DEV, torch.bfloat16), (1 + 0.1 * rnd((k,), "pp.g")).to(DEV),
                                 (0.1 * rnd((k,), "pp.b")).to(DEV), (0.1 * rnd((n,), "pp.c")).to(DEV))
    out = ops.linear_ln_folded(x, folded, st)
    assert torch.equal(ops.linear_ln_folded(x, folded, st), out)
    assert torch.equal(ops.linear_ln_folded(x[perm].contiguous(), folded, st[perm].contiguous()), out[perm])


@pytest.mark.parametrize("m,n,k", [(5600, 768, 768), (5555, 768, 3072), (9232, 1024, 1024), (4616, 1024, 4096),
                                   (6000, 1280, 256), (25216, 768, 768), (50432, 768, 3072)])
def test_residual_gemm_row_sums_give_the_statistics_pass_bit_for_bit(m, n, k):
    """vit_gemm_residual_row_stats + vit_row_stats_finalize: the residual epilogue of the persistent GEMM leaves per-64-column
    (sum, sum of squares) of the rows it stores (every tile height of the few-tile planner, head / tail launches, ragged M),
    and the finalised (mean * rstd, rstd) are the SAME BITS vit_layernorm_stats computes from the stored rows -- so a forward
    may take either route. The output itself is the plain residual GEMM's, and the sums agree with torch on the stored values."""
    eps = 1e-6
    x = (rnd((m, k), "rs.x") * 1.5 + 0.2).to(DEV, torch.bfloat16)
    w = rnd((n, k), "rs.w", 0.03).to(DEV, torch.bfloat16)
    b = rnd((n,), "rs.b", 0.2).to(DEV)
    r = (rnd((m, n), "rs.r") * 2 + 0.4).to(DEV, torch.bfloat16)
    out, stats, part = ops.linear_residual_row_stats(x, w, b, r, eps)
    assert torch.equal(out, ops.linear(x, w, b, residual=r))
    assert torch.equal(stats, ops.layer_norm_stats(out, eps))
    o = out.double().view(m, n // 64, 64)
    assert rel_err(part[:, :, 0].t(), o.sum(-1)) < 1e-6
    assert rel_err(part[:, :, 1].t(), (o * o).sum(-1)) < 1e-6
    o = out.double()
    rstd = 1.0 / torch.sqrt(o.var(-1, unbiased=False) + eps)
    assert rel_err(stats[:, 1], rstd) < 1e-5
    assert rel_err(stats[:, 0], o.mean(-1) * rstd) < 1e-5
    out2, stats2, _ = ops.linear_residual_row_stats(x, w, b, r, eps)
    assert torch.equal(out2, out) and torch.equal(stats2, stats)  # repeatable


def test_one_pass_statistics_error_bound_with_a_large_mean():
    """The one-pass form (var = E[x^2] - mean^2 in fp32) loses accuracy as |mean| / std grows: about 2^-23 (1 + mean^2 / var)
    sqrt(dim) relative in var. Pinned here at |mean| = 20 std (far beyond what a residual stream shows): rstd within 1e-3,
    i.e. still below the bf16 rounding of the normalised values; at |mean| <= std it is 1e-6."""
    for shift, tol in ((0.5, 2e-6), (20.0, 1e-3)):
        x = (rnd((2048, 768), "lm.x") + shift).to(DEV, torch.bfloat16)
        st = ops.layer_norm_stats(x, 1e-6).double()
        xd = x.double()
        rstd = 1.0 / torch.sqrt(xd.var(-1, unbiased=False) + 1e-6)
        assert rel_err(st[:, 1], rstd) < tol, shift
        assert rel_err(st[:, 0], xd.mean(-1) * rstd) < tol, shift


def test_residual_row_stats_rejects_shapes_the_persistent_kernel_does_not_take():
    x = rnd((512, 256), "rj.x").to(DEV, torch.bfloat16)
    w = rnd((768, 256), "rj.w", 0.03).to(DEV, torch.bfloat16)
    r = rnd((512, 768), "rj.r").to(DEV, torch.bfloat16)
    with pytest.raises(RuntimeError, match="persistent kernel"):
        ops.linear_residual_row_stats(x, w, None, r, 1e-6)


def _all_bf16_values(lo, hi, count):
    """every finite bf16 value in [lo, hi], tiled to `count` entries (the GEMM below passes them through unchanged)"""
    bits = torch.arange(0, 1 << 16, dtype=torch.int32).to(torch.int16)
    v = bits.view(torch.bfloat16)
    v = v[torch.isfinite(v.float()) & (v.float() >= lo) & (v.float() <= hi)]
    reps = (count + v.numel() - 1) // v.numel()
    return v.repeat(reps)[:count]


@pytest.mark.parametrize("act", ["gelu", "quick_gelu"])
def test_activation_epilogue_error_bound_on_every_bf16_input(act):
    """The persistent GEMM's activation epilogues against the exact functions (fp64) on EVERY bf16 value in [-12, 12]:
    A holds the values, W is the identity, so the pre-activation is the value itself. gelu is the packed polynomial of
    vit_common.h (|error| <= 8.5e-5 stated there) and quick_gelu the v_rcp form; on top of that the output is rounded
    to bf16 once (half an ulp of the result)."""
    m, n = 16384, 256  # m * n = 64 * 256 * 256: the smallest problem the persistent kernel takes
    a = _all_bf16_values(-12.0, 12.0, m * n).view(m, n).to(DEV)
    w = torch.eye(n, dtype=torch.bfloat16, device=DEV)
    y = ops.linear(a, w, None, activation=act).double().cpu()
    x = a.double().cpu()
    if act == "gelu":
        want = 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))
        fn_err = 8.5e-5
    else:
        want = x * torch.sigmoid(1.702 * x)
        fn_err = 1e-6 * want.abs() + 1e-30
    half_ulp = want.abs() * 2.0 ** -8  # bf16: 8 significant bits, so half an ulp <= 2^-9 * 2^ceil(log2) <= 2^-8 |y|
    err = (y - want).abs()
    assert bool((err <= fn_err + half_ulp + 1e-38).all()), float((err - fn_err - half_ulp).max())
    assert float(y[x >= 6].sub(x[x >= 6]).abs().max()) == 0.0  # upper tail: act(x) rounds to x
    if act == "gelu":
        assert float(y[x <= -9].abs().max()) == 0.0            # lower tail: Phi is clamped to exactly 0


def test_gelu_backward_multiplier_error_bound_on_every_bf16_input():
    """gelu'(aux) of the data-gradient epilogue (VIT_EPI_DGELU) on every bf16 aux in [-12, 12]: dY W = 1 everywhere, so the
    output is the multiplier itself, Phi(x) + x phi(x)."""
    m, n, k = 16384, 256, 256
    aux = _all_bf16_values(-12.0, 12.0, m * n).view(m, n).to(DEV)
    dy = torch.zeros(m, k, dtype=torch.bfloat16, device=DEV)
    dy[:, 0] = 1.0
    wt = torch.zeros(n, k, dtype=torch.bfloat16, device=DEV)
    wt[:, 0] = 1.0
    g = ops.linear_dgrad(dy, wt, pre_activation=aux, activation="gelu").double().cpu()
    x = aux.double().cpu()
    want = 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)
    err = (g - want).abs()
    assert float((err - want.abs() * 2.0 ** -8).max()) <= 6e-5  # Phi from the polynomial: <= 4.3e-5; exp2 / rounding on top
