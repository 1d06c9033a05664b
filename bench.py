#!/usr/bin/env python3
"""Headline benchmark: images/sec, ViT-B/16 224^2 bf16 forward (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W      (N > 1: this process starts the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
  --config {b16,l14,h14} --scaling {weak,strong} --batch B select BASELINE.json configs[2] / [3]
  (default: configs[1], ViT-B/16, weak scaling, 256 images per GPU).

A step = one forward of the hot path (vit_forward through the C ABI) over one batch of 256
synthetic images per GPU, already resident in HBM; for N > 1 the batch is sharded (weak
scaling, 256 images per rank) and the step ends with the RCCL all-gather of the CLS
embeddings. W untimed steps, then exactly K timed steps bracketed by barrier +
torch.cuda.synchronize(); the time is the MAX over ranks; rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      dominant kernel (the bf16 MFMA GEMM instance that takes the most time per
                forward): algorithmic FLOPs per launch / average launch duration measured live
                with events on the launch stream, against the 2.5 PFLOP/s dense bf16 MFMA peak.
  kernels       the same measurement for every kernel of the path (GEMMs/attention vs MFMA
                peak, LayerNorm vs 8 TB/s HBM).
  cpu_baseline  the CPU oracle (oracle/vit_oracle.py, fp32, batch 8) timed on this box's host
                cores (rank 0, N = 1 only) -- a reported baseline, not a target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from vit_amd import ops, synth  # noqa: E402
from vit_amd.config import ViTConfig  # noqa: E402
from vit_amd.encoder import ViTEncoder  # noqa: E402

MFMA_BF16_PEAK = 2.5e15  # dense bf16 FLOP/s, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
HBM_PEAK = 8.0e12        # bytes/s spec, same guide
LOCAL_BATCH = 256


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def time_kernel(fn, iters=20, warm=3):
    """Average launch duration in seconds, events recorded on the launch (current) stream."""
    for _ in range(warm):
        fn()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) * 1e-3 / iters


def kernel_rooflines(cfg: ViTConfig, batch: int, dev):
    """Per-kernel live measurement at the bench shapes (synthetic N(0,1)-scale operands)."""
    m, d, f, n, h = batch * cfg.seq_len, cfg.hidden_size, cfg.intermediate_size, cfg.seq_len, cfg.num_attention_heads
    bf = torch.bfloat16

    def t(shape, name, std=1.0):
        return synth.normal(shape, 11, name, std=std, device=dev).to(bf)

    x, hid = t((m, d), "x"), t((m, f), "h")
    wqkv, wo, w1, w2 = t((3 * d, d), "wqkv", 0.02), t((d, d), "wo", 0.02), t((f, d), "w1", 0.02), t((d, f), "w2", 0.02)
    bqkv, bo, b1 = [synth.normal((k,), 11, nm, 0.05, device=dev) for k, nm in ((3 * d, "bq"), (d, "bo"), (f, "b1"))]
    qkv = t((batch, n, 3 * d), "qkv")
    g, b = synth.normal((d,), 11, "g", 0.1, 1.0, device=dev), synth.normal((d,), 11, "b", 0.05, device=dev)
    px = synth.synthetic_images(cfg, batch, seed=3, device=dev).to(bf)
    pw = t((d, cfg.num_channels, cfg.patch_size, cfg.patch_size), "pw", 0.02)
    cls, pos = synth.normal((d,), 11, "cls", device=dev), synth.normal((n, d), 11, "pos", device=dev)
    L = cfg.num_hidden_layers
    # vit_forward folds LN1 / LN2 into the QKV / fc1 GEMMs: a statistics pass over x, then the GEMM on the
    # un-normalised rows with the row terms in its epilogue (include/vit_amd.h, vit_gemm_ln_folded)
    stats = ops.layer_norm_stats(x, cfg.layer_norm_eps)
    fqkv, f1 = ops.layer_norm_fold(wqkv, g, b, bqkv), ops.layer_norm_fold(w1, g, b, b1)
    # the o_proj / fc2 GEMMs of vit_forward also leave the row sums of what they store (the next folded LayerNorm's statistics,
    # DESIGN 11.10), finalised by a small launch; the statistics pass itself is left for layer 0
    rs_out, rs_part = torch.empty_like(x), torch.empty(max(d // 64, 1), m, 2, dtype=torch.float32, device=dev)
    eps = cfg.layer_norm_eps
    # (small batches: vit_forward takes the plain residual GEMM and the statistics pass, and so does this list)
    row_sums = d % 64 == 0 and d % 128 == 0 and f % 128 == 0 and m * d >= 64 * 256 * 256

    def res_gemm(a, w):
        if row_sums:
            return ops.linear_residual_row_stats(a, w, bo, x, eps, rs_out, rs_part, finalize=False)
        return ops.linear(a, w, bo, residual=x)
    specs = [  # name, launches per forward, callable, flops per launch, bytes per launch (HBM-bound kernels)
        ("gemm_qkv", L, lambda: ops.linear_ln_folded(x, fqkv, stats), 2 * m * 3 * d * d, None),
        ("gemm_o_proj_res", L, lambda: res_gemm(x, wo), 2 * m * d * d, None),
        ("gemm_fc1_gelu", L, lambda: ops.linear_ln_folded(x, f1, stats, cfg.hidden_act), 2 * m * f * d, None),
        ("gemm_fc2_res", L, lambda: res_gemm(hid, w2), 2 * m * d * f, None),
        ("row_stats_finalize", (2 * L - 2) if row_sums else 0, lambda: ops.row_stats_finalize(rs_part, eps, stats), None, m * max(d // 64, 1) * 8 + m * 8),
        # attention at N = 197, dh = 64 is 99 FLOP per byte (Q, K, V read once, output written once)
        # against a machine balance of ~310: its roofline is HBM; both figures are reported
        ("attention", L, lambda: ops.attention(qkv, h), 4 * batch * h * n * n * cfg.head_dim, 4 * m * d * 2),
        ("layernorm_stats", 2 if row_sums else 2 * L, lambda: ops.layer_norm_stats(x, cfg.layer_norm_eps), None, m * d * 2 + m * 8),
        ("layernorm", 1, lambda: ops.layer_norm(x, g, b, cfg.layer_norm_eps), None, 2 * m * d * 2),
        ("patch_embed", 1, lambda: ops.patch_embed(px, pw, bo, cls, pos, cfg.patch_size),
         2 * batch * cfg.num_patches * cfg.patch_dim * d, None),
    ]
    out = {}
    for name, per_fwd, fn, flops, nbytes in specs:
        dt = time_kernel(fn)
        if flops is not None and nbytes is not None:
            out[name] = {"bound": "hbm", "achieved": nbytes / dt / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": nbytes / dt / HBM_PEAK, "mfma_tflops": flops / dt / 1e12,
                         "mfma_frac": flops / dt / MFMA_BF16_PEAK, "us_per_launch": dt * 1e6,
                         "launches_per_forward": per_fwd}
        elif flops is not None:
            ach = flops / dt
            out[name] = {"bound": "mfma", "achieved": ach / 1e12, "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                         "frac": ach / MFMA_BF16_PEAK, "us_per_launch": dt * 1e6, "launches_per_forward": per_fwd}
        else:
            ach = nbytes / dt
            out[name] = {"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": ach / HBM_PEAK, "us_per_launch": dt * 1e6, "launches_per_forward": per_fwd}
    if 64 % cfg.patch_size:
        # the stand-alone operator has no workspace and gathers per element; vit_forward first gathers the patch rows
        # into its workspace (im2col_rows_kernel) and runs the persistent GEMM on them (profiles/: ~0.2 ms in all)
        out["patch_embed"]["note"] = "stand-alone vit_patch_embed_fwd (gather loader); vit_forward uses im2col_rows + the persistent GEMM"
    return out


def pmc_traffic_bytes(kernel_key: str):
    """Bytes beyond L2 per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (the newest profiles/r*_traffic.json: separate FETCH_SIZE / WRITE_SIZE runs of this same script, with
    the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts half of a wide coalesced
    read). Counters cannot be read from inside the process, so this is the recorded figure for
    the same workload, or None when the file or the kernel is missing."""
    try:
        import glob
        newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]  # named per round: r03a > r02d
        with open(newest) as f:
            rec = json.load(f)["per_launch_kb"].get(kernel_key)
        return None if rec is None else (2.0 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024.0
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(cfg: ViTConfig, budget_s: float = 12.0):
    """The CPU oracle on this host: fp32, batch 8 (BASELINE.json configs[0])."""
    from oracle import vit_oracle
    sd = synth.synthetic_state_dict(cfg, seed=0)
    px = synth.synthetic_images(cfg, 8, seed=1)
    vit_oracle.forward(px, sd, cfg)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        vit_oracle.forward(px, sd, cfg)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 50:
            break
    return {"value": 8 * n / el, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} forwards of batch 8, ViT-B/16 224^2 fp32, oracle/vit_oracle.py (plain torch CPU)"}


def train_main(args, cfg, enc, px, rank, world, dev, dist):
    """BASELINE.json configs[4]: forward + backward + SGD on 256 images per GPU; for N > 1 the
    gradients are all-reduced over RCCL, one bucket per transformer layer, overlapped with the
    backward of the next layer. Synthetic regression target on the CLS embedding."""
    from vit_amd.trainer import DataParallelTrainer, ViTTrainer
    tr = ViTTrainer(enc)
    dp = DataParallelTrainer(tr) if world > 1 else None
    target = synth.normal((LOCAL_BATCH, cfg.hidden_size), 7 + rank, "target", device=dev)

    def step():
        out = tr.forward(px)
        d = (out.pooler_output.float() - target).mul_(1.0 / LOCAL_BATCH).to(torch.bfloat16)
        if dp is not None:
            dp.backward(None, d)
        else:
            tr.backward(None, d)
        tr.sgd_step(1e-3)
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    assert torch.isfinite(out.pooler_output).all(), "non-finite embeddings"
    if rank == 0:
        value = world * LOCAL_BATCH * args.steps / elapsed
        flops = value * 3 * cfg.flops_per_image()  # backward = 2x forward (dgrad + wgrad)
        print(json.dumps({
            "metric": "images/sec ViT-B/16 224^2 bf16 training step (forward + backward + SGD)", "value": value,
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "ViT-B/16 224^2 bf16 training step, batch 256 per MI355X (BASELINE.json configs[4])",
                       "global_batch": world * LOCAL_BATCH, "seq_len": cfg.seq_len,
                       "parallelism": f"dp{world}" + (" + RCCL gradient all-reduce per layer" if world > 1 else "")},
            "roofline": {"bound": "mfma", "achieved": flops / 1e12, "peak": world * MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                         "frac": flops / (world * MFMA_BF16_PEAK), "traffic": None,
                         "kernel": "whole step, 3 x forward FLOPs"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def other_configs(dev, steps: int = 5, warm: int = 2):
    """The BASELINE.json configurations the headline line does not time, measured on this GPU after the timed region
    (one GPU, inputs resident, `steps` timed steps after `warm` untimed ones, a fresh encoder each, freed afterwards):
    configs[2] at its global batch of 128 and at the 16 images a rank holds when that batch is sharded over 8 GPUs,
    configs[3] at its per-GPU batch, and the configs[4] training step. Same timing rule as the headline loop."""
    import gc
    out = {}

    def timed(fn):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, r

    for key, cname, batch in (("l14_b128", "l14", 128), ("l14_b16", "l14", 16), ("h14", "h14", CONFIGS["h14"][4])):
        cfg = CONFIGS[cname][0]()
        enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=dev)
        enc.load_state_dict(synth.synthetic_state_dict(cfg, seed=0, device=dev))
        px = synth.synthetic_images(cfg, batch, seed=1, device=dev).to(torch.bfloat16)
        dt, o = timed(lambda: enc(px))
        assert torch.isfinite(o.pooler_output).all(), key
        out[key] = {"workload": f"{CONFIGS[cname][1]} bf16 forward, batch {batch} on one MI355X ({CONFIGS[cname][2]})",
                    "ms_per_step": 1e3 * dt, "images_per_sec": batch / dt,
                    "end_to_end_frac": batch / dt * cfg.flops_per_image() / MFMA_BF16_PEAK, "steps": steps}
        enc.close()
        del enc, px, o
        gc.collect()
        torch.cuda.empty_cache()

    from vit_amd.trainer import ViTTrainer
    cfg = ViTConfig.vit_b16_224()
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=dev)
    enc.load_state_dict(synth.synthetic_state_dict(cfg, seed=0, device=dev))
    tr = ViTTrainer(enc)
    px = synth.synthetic_images(cfg, LOCAL_BATCH, seed=1, device=dev).to(torch.bfloat16)
    target = synth.normal((LOCAL_BATCH, cfg.hidden_size), 7, "target", device=dev)

    def train_step():
        o = tr.forward(px)
        d = (o.pooler_output.float() - target).mul_(1.0 / LOCAL_BATCH).to(torch.bfloat16)
        tr.backward(None, d)
        tr.sgd_step(1e-3)
        return o

    dt, o = timed(train_step)
    assert torch.isfinite(o.pooler_output).all(), "train_b16"
    out["train_b16"] = {"workload": "ViT-B/16 224^2 bf16 training step (forward + backward + SGD), batch 256 on one MI355X (configs[4])",
                        "ms_per_step": 1e3 * dt, "images_per_sec": LOCAL_BATCH / dt,
                        "end_to_end_frac": LOCAL_BATCH / dt * 3 * cfg.flops_per_image() / MFMA_BF16_PEAK, "steps": steps}
    enc.close()
    del tr, enc, px, o
    gc.collect()
    torch.cuda.empty_cache()
    return out


CONFIGS = {
    # key: (ViTConfig factory, short name, BASELINE.json config, default scaling, default batch
    #       [weak: images per GPU; strong: global images])
    "b16": (ViTConfig.vit_b16_224, "ViT-B/16 224^2", "configs[1]", "weak", 256),
    "l14": (ViTConfig.clip_l14_336, "ViT-L/14 336^2 (CLIP vision tower)", "configs[2]", "strong", 128),
    # h14: batch per GPU "chosen to fill" (SURVEY.md 8d): swept 32 / 48 / 64 / 96 on one MI355X (round 3, one box):
    # 771.9 / 777.3 / 783.2 / 783.0 images/s -- 64 images = 65,600 token rows is where the per-image rate levels off
    "h14": (ViTConfig.vit_h14_448, "ViT-H/14 448^2", "configs[3]", "weak", 64),
}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` as typed: this parent has touched no GPU (nothing above initialises HIP) and starts
    N fresh rank processes through torch.distributed.run as a CHILD process (never an exec), then exits with its
    code. The ranks re-enter this file with RANK / LOCAL_RANK / WORLD_SIZE set."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernels", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short runs of configs[2..4] that the default headline line appends as other_configs")
    ap.add_argument("--workload", choices=("forward", "train"), default="forward",
                    help="forward = the headline metric (BASELINE.json configs[1]); train = fwd+bwd+SGD step with the "
                         "gradient all-reduce (configs[4]), reported as images/sec of the training step")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="b16",
                    help="b16 = ViT-B/16 224 (configs[1], the headline), l14 = CLIP ViT-L/14 336 (configs[2]), "
                         "h14 = ViT-H/14 448 (configs[3])")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="weak: --batch images per GPU; strong: --batch images in all, sharded over the GPUs "
                         "(default: weak for b16 / h14, strong for l14 as configs[2] asks)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (weak) or global images (strong)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    factory, cname, cref, dscaling, dbatch = CONFIGS[args.config]
    scaling = args.scaling or dscaling
    batch = args.batch or dbatch
    cfg = factory()
    if scaling == "weak":
        local_batch, global_batch = batch, batch * world
        lo = rank * batch
    else:
        from vit_amd.dist import shard_bounds
        global_batch = batch
        lo, hi = shard_bounds(global_batch, rank, world)
        local_batch = hi - lo
        if local_batch == 0:
            raise SystemExit(f"strong scaling: global batch {global_batch} leaves rank {rank} of {world} empty")
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=dev)
    enc.load_state_dict(synth.synthetic_state_dict(cfg, seed=0, device=dev))
    px = synth.synthetic_images(cfg, local_batch, seed=1 + rank, device=dev).to(torch.bfloat16)
    if args.workload == "train":
        if args.config != "b16" or scaling != "weak" or batch != LOCAL_BATCH:
            raise SystemExit("--workload train is BASELINE.json configs[4]: ViT-B/16, weak scaling, 256 images per GPU")
        return train_main(args, cfg, enc, px, rank, world, dev, dist)
    from vit_amd.dist import all_gather_rows

    def step():
        out = enc(px)
        if world > 1:  # the path's one exchange step: all-gather of the CLS embeddings (RCCL)
            all_gather_rows(out.pooler_output, global_batch)
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    assert torch.isfinite(out.pooler_output).all(), "non-finite embeddings"

    if rank == 0:
        images = global_batch * args.steps
        value = images / elapsed
        flops = value * cfg.flops_per_image()
        per = f"batch {batch} per MI355X" if scaling == "weak" else f"global batch {batch} sharded over {world} MI355X"
        line = {
            "metric": f"images/sec {cname} bf16 forward" if args.config != "b16" else "images/sec ViT-B/16 224^2 bf16 forward",
            "value": value, "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{cname} bf16 forward, {per} (BASELINE.json {cref})",
                       "global_batch": global_batch, "seq_len": cfg.seq_len,
                       "parallelism": f"dp{world}" + (" + RCCL all-gather of CLS embeddings" if world > 1 else "")},
            "end_to_end": {"achieved": flops / 1e12, "peak": world * MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                           "frac": flops / (world * MFMA_BF16_PEAK)},
        }
        if not args.no_kernels:
            ks = kernel_rooflines(cfg, local_batch, dev)
            dom = max((k for k in ks if k.startswith("gemm")),
                      key=lambda k: ks[k]["us_per_launch"] * ks[k]["launches_per_forward"])
            r = ks[dom]
            line["roofline"] = {"bound": r["bound"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"],
                                "frac": r["frac"],
                                "traffic": pmc_traffic_bytes(dom) if (args.config == "b16" and local_batch == LOCAL_BATCH) else None,
                                "kernel": "bf16 MFMA GEMM (" + dom + ")", "us_per_launch": r["us_per_launch"],
                                # one operator call of the full batch: the few-tile planner runs it as a head of whole
                                # 256-row-tile rounds plus a tail of shorter tiles (template instances <..., 4> and
                                # <..., 3|2> of gemm8p_persist_kernel; their rocprof averages add up to this figure)
                                "launch": "one GEMM call = head + tail launch of gemm8p_persist_kernel (DESIGN 11.1)"}
            line["kernels"] = ks
        if world == 1 and args.config == "b16" and not args.no_other_configs:
            enc.close()
            del enc, px, out
            torch.cuda.empty_cache()
            line["other_configs"] = other_configs(dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ViTConfig.vit_b16_224())
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
