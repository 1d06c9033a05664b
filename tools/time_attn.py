"""Times vit_attention_fwd at the encoder shapes (bf16) and prints TFLOP/s; VIT_AMD_ATTN_RING=0 selects the staged kernel."""
import sys, torch
sys.path.insert(0, ".")
from vit_amd import ops, synth
cases = [(32, 1025, 16, 80), (64, 577, 16, 64), (256, 197, 12, 64), (8, 1025, 16, 80)]
if len(sys.argv) > 1: cases = [cases[int(i)] for i in sys.argv[1:]]   # e.g. "2" = the ViT-B/16 case only (counter runs)
for b, n, h, dh in cases:
    qkv = synth.normal((b, n, 3 * h * dh), 3, "qkv", device="cuda:0").to(torch.bfloat16)
    for _ in range(3): ops.attention(qkv, h)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.attention(qkv, h)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"attention b={b} n={n} h={h} dh={dh}: {us:.1f} us  {4.0*b*h*n*n*dh/us*1e-6:.0f} TFLOP/s", flush=True)
