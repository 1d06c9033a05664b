#!/usr/bin/env python3
"""Few-tile GEMM shapes (small batches: the per-GPU shard of a strong-scaled batch) on the persistent kernel at the tile
height the policy picks (default) or a forced one: VIT_AMD_GEMM_MI=4|3|2 in the environment (read once per process, so
run this script once per value on one box). Prints one JSON line: microseconds per launch per shape, and a hash of the
outputs (the tile height must not change a bit).
  for mi in 4 3 2 0; do VIT_AMD_GEMM_MI=$mi python tools/time_fewtile.py; done"""
import hashlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops, synth

dev = "cuda:0"
SHAPES = {  # name: (tokens, [(N, K, kind)])
    "l14_b16": (16 * 577, 1024, 4096), "l14_b32": (32 * 577, 1024, 4096), "l14_b8": (8 * 577, 1024, 4096),
    "b16_b32": (32 * 197, 768, 3072), "b16_b48": (48 * 197, 768, 3072), "h14_b8": (8 * 1025, 1280, 5120),
    "h14_b4": (4 * 1025, 1280, 5120),
}


def t_us(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


out = {"mi": os.environ.get("VIT_AMD_GEMM_MI", "policy")}
h = hashlib.sha256()
for name, (m, d, f) in SHAPES.items():
    bf = torch.bfloat16
    x = synth.normal((m, d), 5, "x").to(dev, bf)
    hid = synth.normal((m, f), 5, "h").to(dev, bf)
    wqkv, wo, w1, w2 = [synth.normal(s, 5, n, std=0.02).to(dev, bf) for s, n in
                        (((3 * d, d), "wqkv"), ((d, d), "wo"), ((f, d), "w1"), ((d, f), "w2"))]
    g, b = synth.normal((d,), 5, "g", 0.1, 1.0, device=dev), synth.normal((d,), 5, "b", 0.05, device=dev)
    bq, bo, b1 = [synth.normal((k,), 5, n, 0.05, device=dev) for k, n in ((3 * d, "bq"), (d, "bo"), (f, "b1"))]
    st = ops.layer_norm_stats(x, 1e-5)
    fq, f1 = ops.layer_norm_fold(wqkv, g, b, bq), ops.layer_norm_fold(w1, g, b, b1)
    fns = {"qkv": lambda: ops.linear_ln_folded(x, fq, st), "o_proj_res": lambda: ops.linear(x, wo, bo, residual=x),
           "fc1_act": lambda: ops.linear_ln_folded(x, f1, st, "quick_gelu" if d == 1024 else "gelu"),
           "fc2_res": lambda: ops.linear(hid, w2, bo, residual=x)}
    rec = {}
    for k, fn in fns.items():
        rec[k] = round(t_us(fn), 1)
        h.update(fn().cpu().view(torch.int16).numpy().tobytes())
    rec["sum"] = round(sum(rec.values()), 1)
    out[name] = rec
out["hash"] = h.hexdigest()[:16]
print(json.dumps(out))
