#!/usr/bin/env python3
"""Lab: time individual ops at the bench shapes (ViT-B/16 batch 256 by default)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops, synth
from vit_amd.config import ViTConfig
import bench

def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "attention"
    cfgname = sys.argv[2] if len(sys.argv) > 2 else "b16"
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    cfg = {"b16": ViTConfig.vit_b16_224, "l14": ViTConfig.clip_l14_336, "h14": ViTConfig.vit_h14_448}[cfgname]()
    dev = torch.device("cuda", 0)
    n, d, h = cfg.seq_len, cfg.hidden_size, cfg.num_attention_heads
    if which == "attention":
        qkv = synth.normal((batch, n, 3 * d), 11, "qkv", device=dev).to(torch.bfloat16)
        ref = None
        for rep in range(3):
            dt = bench.time_kernel(lambda: ops.attention(qkv, h))
            fl = 4 * batch * h * n * n * cfg.head_dim
            print(f"attention {cfgname} B={batch}: {dt*1e6:.1f} us  {fl/dt/1e12:.1f} TF  ({(qkv.numel()*2 + qkv.numel()*2//3)/dt/1e12:.2f} TB/s)")
        # correctness vs torch sdpa in fp32 on a slice
        q, k, v = qkv[:4].float().reshape(4, n, 3, h, cfg.head_dim).permute(2, 0, 3, 1, 4)
        o = torch.nn.functional.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(4, n, d)
        got = ops.attention(qkv[:4].contiguous(), h).float()
        print("max abs err vs fp32 sdpa:", (got - o).abs().max().item(), "ref max", o.abs().max().item())
    elif which == "layernorm":
        x = synth.normal((batch * n, d), 11, "x", device=dev).to(torch.bfloat16)
        g, b = synth.normal((d,), 11, "g", 0.1, 1.0, device=dev), synth.normal((d,), 11, "b", 0.05, device=dev)
        for rep in range(3):
            dt = bench.time_kernel(lambda: ops.layer_norm(x, g, b, 1e-12))
            print(f"layernorm: {dt*1e6:.1f} us  {2*x.numel()*2/dt/1e12:.2f} TB/s")

if __name__ == "__main__":
    main()
