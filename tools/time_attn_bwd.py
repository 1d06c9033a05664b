"""Times vit_attention_bwd (delta + dQ + dK/dV launches) at the headline shape. Usage: python3 tools/time_attn_bwd.py [batch] [seq] [heads] [dh]"""
import sys
import torch
sys.path.insert(0, ".")
from vit_amd import ops

b, n, h, dh = (int(x) for x in (sys.argv[1:5] + ["256", "197", "12", "64"][len(sys.argv) - 1:]))
torch.manual_seed(0)
qkv = (torch.randn(b, n, 3 * h * dh, device="cuda") * 0.5).to(torch.bfloat16)
out, lse = ops.attention_train(qkv, h)
do = torch.randn_like(out)
for _ in range(3):
    ops.attention_backward(qkv, out, do, lse, h)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    ops.attention_backward(qkv, out, do, lse, h)
e1.record()
torch.cuda.synchronize()
print(f"attention_bwd b={b} n={n} h={h} dh={dh}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per call")
