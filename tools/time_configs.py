"""Forward throughput of the other BASELINE.json configurations on one GPU (CLIP-L/14-336 batch 128/8 = 16
and 128, ViT-H/14-448 batch 32), synthetic weights. Usage: python tools/time_configs.py [name[:batch] ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import synth
from vit_amd.config import ViTConfig
from vit_amd.encoder import ViTEncoder

PEAK = 2.5e15
WANT = sys.argv[1:]
for name, cfg, batch in (("vit_b16_224", ViTConfig.vit_b16_224(), 256), ("clip_l14_336", ViTConfig.clip_l14_336(), 128),
                         ("clip_l14_336", ViTConfig.clip_l14_336(), 16), ("vit_h14_448", ViTConfig.vit_h14_448(), 32)):
    if WANT and name not in WANT and f"{name}:{batch}" not in WANT:
        continue
    sd = synth.synthetic_state_dict(cfg, seed=0, device="cuda")
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device="cuda:0")
    enc.load_state_dict(sd)
    del sd
    px = synth.synthetic_images(cfg, batch, seed=1, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        enc(px)
    torch.cuda.synchronize()
    n = 10
    t = time.perf_counter()
    for _ in range(n):
        enc(px)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    fl = cfg.flops_per_image() * batch
    print(f"{name} batch {batch}: {dt * 1e3:.2f} ms/step, {batch / dt:.0f} img/s, {fl / dt / 1e12:.0f} TFLOP/s = {fl / dt / PEAK * 100:.1f}% of bf16 peak")
    enc.close()
    torch.cuda.empty_cache()
