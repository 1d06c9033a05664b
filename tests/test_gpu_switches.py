"""GPU tests of the library's environment switches: each switch is read once per process, so every case runs a
small script in a child process (the child is started BEFORE it touches the GPU; the parent of this test never execs).

* VIT_AMD_GEMM_GC: the column-grouped tile order of the persistent GEMM only changes WHICH workgroup computes a tile
  and when -- the output must be bit-identical to the plain mt-major order, for every epilogue that kernel has.
* VIT_AMD_GEMM_MI: tile height of the persistent GEMM (few-tile planner of gemm_bf16_8p.hip): same bits at every height.
* VIT_AMD_ATTN_BWD_FUSED: the one-pass attention backward against the three-launch form it replaces.
* VIT_AMD_ATTN_RING: head_dim-80 attention on the LDS-DMA ring kernel against the staged kernel it replaces (both
  fp32 online softmax over 32-key tiles; checked to bf16 tolerance, and both against the oracle in test_gpu_kernels).
* VIT_AMD_RING_WAVES: waves per workgroup of the head_dim-80 ring kernel; VIT_AMD_BWD_SIDE_STREAM: weight gradients beside the
  data gradients or behind them -- neither may change a bit.
* VIT_AMD_ROW_STATS: the folded LayerNorm's statistics from the residual GEMMs' row sums (big chunks) or from the stand-alone
  pass (small chunks, or everywhere with the switch at 0): one instruction sequence, so the encoder output is bit-identical.
* VIT_AMD_STATS_ONE_PASS: 0 = the two-pass statistics kernel for every width (and no row sums): same embeddings to bf16 accuracy.
* VIT_AMD_SPLIT_PCT / VIT_AMD_STREAMS: where vit_forward cuts the batch for its two streams changes which launch a row
  belongs to, never its arithmetic: the encoder output must be bit-identical for every cut and for a single stream.
"""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GEMM_SCRIPT = r"""
import hashlib, json, sys, torch
from vit_amd import ops, synth
dev = "cuda:0"
out = {}
# N = 2304 (9 tile columns -> groups of 5 + 4) and N = 3072 (12 -> 6 + 6), ragged M, K % 128 == 0 and
# M * N >= 64 tiles: the persistent kernel takes both
for (m, n, k) in [(2000, 2304, 256), (1400, 3072, 384)]:
    x = synth.normal((m, k), 3, "x").to(dev, torch.bfloat16)
    w = synth.normal((n, k), 4, "w", std=0.05).to(dev, torch.bfloat16)
    b = synth.normal((n,), 5, "b", std=0.1).to(dev)
    r = synth.normal((m, n), 6, "r").to(dev, torch.bfloat16)
    for epi in (None, "gelu", "quick_gelu"):
        y = ops.linear(x, w, b, activation=epi)
        out[f"{m}x{n}x{k}:{epi}"] = hashlib.sha256(y.cpu().view(torch.int16).numpy().tobytes()).hexdigest()
    y = ops.linear(x, w, b, residual=r)
    out[f"{m}x{n}x{k}:res"] = hashlib.sha256(y.cpu().view(torch.int16).numpy().tobytes()).hexdigest()
print(json.dumps(out))
"""

ATTN_SCRIPT = r"""
import json, sys, torch
from vit_amd import ops, synth
dev = "cuda:0"
outs = {}
# one item per workgroup (2 x 4 heads), and more items than resident workgroups (nmy > 1: 8 waves x 640 items,
# 12 waves x 576 items of the ViT-H/14 sequence length)
for (b, n, h) in [(2, 257, 4), (40, 193, 16), (12, 1025, 16)]:
    qkv = synth.normal((b, n, 3 * h * 80), 11, "qkv").to(dev, torch.bfloat16)
    outs[f"{b}x{n}x{h}"] = ops.attention(qkv, h).cpu()
torch.save(outs, sys.argv[1])
print(json.dumps({"ok": all(bool(torch.isfinite(o.float()).all()) for o in outs.values())}))
"""


FEWTILE_SCRIPT = r"""
import hashlib, json, sys, torch
from vit_amd import ops, synth
dev = "cuda:0"
out = {}
h = lambda t: hashlib.sha256(t.cpu().view(torch.int16).numpy().tobytes()).hexdigest()
# few-tile shapes: the planner picks 192-row tiles (M = 9232, N = 1024: 196 tiles in one round), a
# head of 256-row tiles plus a shorter tail, or 128-row tiles, by shape; ragged M and N
for (m, n, k) in [(9232, 1024, 1024), (9232, 3072, 1024), (6304, 768, 2048), (4616, 4096, 1024), (5001, 1000, 256)]:
    x = (synth.normal((m, k), 3, "x") * 1.5 + 0.3).to(dev, torch.bfloat16)
    w = synth.normal((n, k), 4, "w", std=0.05).to(dev, torch.bfloat16)
    b = synth.normal((n,), 5, "b", std=0.1).to(dev)
    r = synth.normal((m, n), 6, "r").to(dev, torch.bfloat16)
    g, be = synth.normal((k,), 7, "g", 0.1, 1.0).to(dev), synth.normal((k,), 7, "be", 0.1).to(dev)
    st = ops.layer_norm_stats(x, 1e-5)
    fold = ops.layer_norm_fold(w, g, be, b)
    for epi in (None, "gelu", "quick_gelu"):
        out[f"{m}x{n}x{k}:lnf:{epi}"] = h(ops.linear_ln_folded(x, fold, st, epi))
    out[f"{m}x{n}x{k}:res"] = h(ops.linear(x, w, b, residual=r))
    inplace = r.clone()   # the encoder's call: C aliases the residual
    from vit_amd import _lib
    _lib.check(_lib.load().vit_gemm_bias_act(x.data_ptr(), w.data_ptr(), b.data_ptr(), inplace.data_ptr(), inplace.data_ptr(), m, n, k,
                                             _lib.EPI_RESIDUAL, _lib.VIT_BF16, torch.cuda.current_stream().cuda_stream))
    out[f"{m}x{n}x{k}:res_inplace"] = h(inplace)
print(json.dumps(out))
"""

ATTN_BWD_SCRIPT = r"""
import json, sys, torch
from vit_amd import ops, synth
dev = "cuda:0"
outs = {}
# 197 tokens x head_dim 64: the one-pass backward kernel (more items than CUs: 24 * 12 = 288 > 256), and 224 / 128 tokens
# (its largest / smallest sequence)
for (b, n, h) in [(24, 197, 12), (3, 224, 2), (2, 128, 3)]:
    qkv = synth.normal((b, n, 3 * h * 64), 13, "qkv").to(dev, torch.bfloat16)
    do = synth.normal((b, n, h * 64), 13, "do").to(dev, torch.bfloat16)
    out, lse = ops.attention_train(qkv, h)
    g = ops.attention_backward(qkv, out, do, lse, h)
    assert torch.equal(g, ops.attention_backward(qkv, out, do, lse, h))  # reproducible
    outs[f"{b}x{n}x{h}"] = g.cpu()
torch.save(outs, sys.argv[1])
print(json.dumps({"ok": all(bool(torch.isfinite(o.float()).all()) for o in outs.values())}))
"""

RING_WAVES_SCRIPT = r"""
import json, sys, torch
from vit_amd import ops, synth
qkv = synth.normal((3, 1025, 3 * 2 * 80), 17, "qkv", std=1.5).to("cuda:0", torch.bfloat16)
o = ops.attention(qkv, 2)
torch.save(o.cpu(), sys.argv[1])
print(json.dumps({"ok": bool(torch.isfinite(o.float()).all())}))
"""

SIDE_STREAM_SCRIPT = r"""
import hashlib, json, torch
from vit_amd import synth
from vit_amd.config import ViTConfig
from vit_amd.encoder import ViTEncoder
from vit_amd.trainer import ViTTrainer
dev = torch.device("cuda:0")
cfg = ViTConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=1024, image_size=64, patch_size=8)
enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=dev)
enc.load_state_dict(synth.synthetic_state_dict(cfg, seed=0, device=dev))
tr = ViTTrainer(enc)
px = synth.synthetic_images(cfg, 128, seed=1, device=dev).to(torch.bfloat16)   # 8320 tokens: the side stream is used
w = synth.normal((128, cfg.hidden_size), 5, "wp", 0.5, device=dev).to(torch.bfloat16)
out = tr.forward(px)
tr.backward(None, w)
torch.cuda.synchronize()
h = lambda t: hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()
keys = ["layers.0.mlp.fc1.weight", "layers.1.mlp.fc2.weight", "layers.0.attention.q_proj.weight", "layers.1.attention.o_proj.weight"]
print(json.dumps({k: h(tr.grad(k)) for k in keys}))
"""

SPLIT_SCRIPT = r"""
import hashlib, json, torch
from vit_amd import synth
from vit_amd.config import ViTConfig
from vit_amd.encoder import ViTEncoder
dev = torch.device("cuda:0")
cfg = ViTConfig.vit_b16_224()
enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=dev)
enc.load_state_dict(synth.synthetic_state_dict(cfg, seed=0, device=dev))
px = synth.synthetic_images(cfg, 72, seed=3, device=dev).to(torch.bfloat16)   # 72 images: two chunks of >= 32 images
out = enc(px)
torch.cuda.synchronize()
h = lambda t: hashlib.sha256(t.cpu().view(torch.int16).numpy().tobytes()).hexdigest()
print(json.dumps({"hidden": h(out.last_hidden_state), "pooled": h(out.pooler_output)}))
"""


def run_child(script, env_extra, *args):
    env = dict(os.environ)
    env.update(env_extra)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.run([sys.executable, "-c", script, *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_grouped_tile_order_gives_the_same_bits_as_plain_order():
    plain = run_child(GEMM_SCRIPT, {"VIT_AMD_GEMM_GC": "0"})
    grouped = run_child(GEMM_SCRIPT, {"VIT_AMD_GEMM_GC": "-1"})   # the default policy
    narrow = run_child(GEMM_SCRIPT, {"VIT_AMD_GEMM_GC": "2"})
    assert plain.keys() == grouped.keys() and len(plain) == 8
    assert plain == grouped
    assert plain == narrow


def test_few_tile_planner_gives_the_same_bits_as_256_row_tiles():
    """VIT_AMD_GEMM_MI: 4 = 256-row tiles only, 3 / 2 = 192- / 128-row tiles everywhere, unset = the planner (tile height
    and head / tail cut by shape). The k order of an output element is the same in all of them: bit-identical outputs."""
    ref = run_child(FEWTILE_SCRIPT, {"VIT_AMD_GEMM_MI": "4"})
    assert len(ref) == 25
    for mi in ("3", "2"):
        assert run_child(FEWTILE_SCRIPT, {"VIT_AMD_GEMM_MI": mi}) == ref, mi
    assert run_child(FEWTILE_SCRIPT, {}) == ref
    for k in ref:
        if k.endswith(":res"):
            assert ref[k] == ref[k + "_inplace"], k


def test_ring_attention_matches_staged_kernel(tmp_path):
    import torch
    a, b = str(tmp_path / "ring.pt"), str(tmp_path / "staged.pt")
    assert run_child(ATTN_SCRIPT, {"VIT_AMD_ATTN_RING": "1"}, a)["ok"]
    assert run_child(ATTN_SCRIPT, {"VIT_AMD_ATTN_RING": "0"}, b)["ok"]
    ring, staged = torch.load(a, weights_only=True), torch.load(b, weights_only=True)
    assert ring.keys() == staged.keys() and len(ring) == 3
    assert ring["2x257x4"].shape == (2, 257, 320)
    for key in ring:
        r, s = ring[key].float(), staged[key].float()
        scale = s.abs().max().item()
        assert (r - s).abs().max().item() <= scale / 128, key  # 2 bf16 ulps of the output scale


def test_one_pass_attention_backward_matches_three_launch_form(tmp_path):
    """VIT_AMD_ATTN_BWD_FUSED=0 keeps the delta + dQ + dK/dV launches for the shapes the one-pass kernel takes (head_dim 64,
    97..224 tokens). Same products in the same order; delta is summed in another order (from the query's own O row in
    registers instead of the delta kernel's butterfly), so the gradients agree to bf16 rounding, not bit for bit."""
    import torch
    a, b = str(tmp_path / "fused.pt"), str(tmp_path / "three.pt")
    assert run_child(ATTN_BWD_SCRIPT, {}, a)["ok"]
    assert run_child(ATTN_BWD_SCRIPT, {"VIT_AMD_ATTN_BWD_FUSED": "0"}, b)["ok"]
    fused, three = torch.load(a, weights_only=True), torch.load(b, weights_only=True)
    assert fused.keys() == three.keys() and len(fused) == 3
    for key in fused:
        f, t = fused[key].float(), three[key].float()
        d = t.shape[-1] // 3
        for j in range(3):  # dq, dk, dv have different scales
            fj, tj = f[..., j * d:(j + 1) * d], t[..., j * d:(j + 1) * d]
            assert (fj - tj).abs().max().item() <= tj.abs().max().item() / 64, (key, j)
            # both are 0.24 % rms from the fp64 gradient (tools/cmp_attn_bwd.py: bf16 P / dS operands and outputs)
            assert ((fj - tj).double().norm() / tj.double().norm()).item() <= 5e-3, (key, j)


def test_ring_attention_wave_counts_agree(tmp_path):
    """VIT_AMD_RING_WAVES forces the waves per workgroup of the head_dim-80 ring kernel (8..12; the default policy picks
    12 at 1025 tokens). The wave count only changes which wave owns which 32 queries: same bits for every count."""
    import torch
    outs = []
    for w in ("8", "9", "10", "11", "12", ""):
        f = str(tmp_path / f"w{w}.pt")
        assert run_child(RING_WAVES_SCRIPT, {"VIT_AMD_RING_WAVES": w} if w else {}, f)["ok"]
        outs.append(torch.load(f, weights_only=True))
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_backward_side_stream_does_not_change_the_weight_gradients():
    """VIT_AMD_BWD_SIDE_STREAM=0 runs the weight gradients on the main stream instead of beside the data gradients: the
    matrices are summed in a fixed order either way, so they must be bit-identical."""
    on = run_child(SIDE_STREAM_SCRIPT, {})
    off = run_child(SIDE_STREAM_SCRIPT, {"VIT_AMD_BWD_SIDE_STREAM": "0"})
    assert len(on) == 4 and on == off


def test_row_stats_from_the_residual_epilogue_do_not_change_the_bits():
    """72 images of ViT-B/16 run as chunks of 45 and 27: the first chunk's o_proj / fc2 GEMMs are big enough to write the
    row sums, the second chunk's are not and take the statistics pass -- and with VIT_AMD_ROW_STATS=0 every chunk does."""
    on = run_child(SPLIT_SCRIPT, {})
    off = run_child(SPLIT_SCRIPT, {"VIT_AMD_ROW_STATS": "0"})
    assert on == off
    assert run_child(SPLIT_SCRIPT, {"VIT_AMD_STREAMS": "1"}) == run_child(SPLIT_SCRIPT, {"VIT_AMD_STREAMS": "1", "VIT_AMD_ROW_STATS": "0"})


STATS_SCRIPT = r"""
import json, sys, torch
from vit_amd import synth
from vit_amd.config import ViTConfig
from vit_amd.encoder import ViTEncoder
dev = torch.device("cuda:0")
cfg = ViTConfig.vit_b16_224()
enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=dev)
enc.load_state_dict(synth.synthetic_state_dict(cfg, seed=0, device=dev))
px = synth.synthetic_images(cfg, 40, seed=3, device=dev).to(torch.bfloat16)
out = enc(px)
torch.cuda.synchronize()
torch.save({"hidden": out.last_hidden_state.cpu(), "pooled": out.pooler_output.cpu()}, sys.argv[1])
print(json.dumps({"ok": True}))
"""


def test_two_pass_statistics_switch_agrees_to_bf16_accuracy(tmp_path):
    """VIT_AMD_STATS_ONE_PASS=0: the two-pass statistics kernel for every width (and no row sums from the residual epilogues):
    other roundings of the statistics, so not the same bits -- the same embeddings to bf16 accuracy."""
    import torch
    a, b = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    run_child(STATS_SCRIPT, {}, a)
    run_child(STATS_SCRIPT, {"VIT_AMD_STATS_ONE_PASS": "0"}, b)
    one, two = torch.load(a), torch.load(b)
    for k in ("hidden", "pooled"):
        x, y = one[k].double(), two[k].double()
        # two bf16 forwards whose statistics differ in the last bits part ways at the first flipped rounding and stay as far
        # apart as either is from the fp32 oracle (12 layers of random weights): 0.8 % rms measured, 1 % against the goldens
        assert ((x - y).pow(2).mean().sqrt() / y.pow(2).mean().sqrt()).item() < 1.5e-2, k
        cos = torch.nn.functional.cosine_similarity(x.flatten(0, -2), y.flatten(0, -2), dim=-1)
        assert cos.min().item() > 0.9995, k


def test_stream_split_point_does_not_change_the_bits():
    ref = run_child(SPLIT_SCRIPT, {"VIT_AMD_STREAMS": "1"})
    for env in ({}, {"VIT_AMD_SPLIT_PCT": "50"}, {"VIT_AMD_SPLIT_PCT": "45"}):
        assert run_child(SPLIT_SCRIPT, env) == ref, env
