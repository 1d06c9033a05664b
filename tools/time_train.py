#!/usr/bin/env python3
"""Lab: time the ViT-B/16 bf16 training step (forward_train + staged backward + SGD) on one GPU."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import synth
from vit_amd.config import ViTConfig
from vit_amd.encoder import ViTEncoder
from vit_amd.trainer import ViTTrainer

def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    cfg = ViTConfig.vit_b16_224()
    dev = torch.device("cuda", 0)
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=dev)
    enc.load_state_dict(synth.synthetic_state_dict(cfg, seed=0, device=dev))
    tr = ViTTrainer(enc)
    px = synth.synthetic_images(cfg, batch, seed=1, device=dev).to(torch.bfloat16)
    tgt = synth.normal((batch, cfg.hidden_size), 7, "t", device=dev)
    def step():
        out = tr.forward(px)
        d = (out.pooler_output.float() - tgt).to(torch.bfloat16)
        tr.backward(None, d)
        tr.sgd_step(1e-4)
        return out
    for _ in range(2): step()
    torch.cuda.synchronize()
    for name, fn in (("forward_train", lambda: tr.forward(px)), ("step", step)):
        t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        fl = cfg.flops_per_image() * batch * (1 if name == "forward_train" else 3)
        print(f"{name}: {dt*1e3:.2f} ms  {batch/dt:.0f} img/s  {fl/dt/1e12:.0f} TFLOP/s (3x fwd FLOPs for the step)")
    print("loss finite:", torch.isfinite(step().pooler_output).all().item(), "mem GB:", torch.cuda.max_memory_allocated()/2**30)

if __name__ == "__main__":
    main()
