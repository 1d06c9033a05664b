"""Lab: LayerNorm / statistics-pass timings at the three configs' row shapes (full batch and half batch)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
import bench

for name, rows, d in (("b16 b256", 50432, 768), ("b16 half", 25216, 768), ("clip-l b128", 73856, 1024), ("clip-l half", 36928, 1024),
                      ("vit-h b32", 32800, 1280), ("vit-h half", 16400, 1280)):
    x = torch.randn(rows, d, device="cuda").to(torch.bfloat16)
    g, b = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
    t1 = bench.time_kernel(lambda: ops.layer_norm_stats(x, 1e-5))
    t2 = bench.time_kernel(lambda: ops.layer_norm(x, g, b, 1e-5))
    mb = rows * d * 2 / 1e6
    print(f"{name:12s} stats {t1*1e6:7.1f} us ({mb/t1/1e6:5.2f} TB/s)   layernorm {t2*1e6:7.1f} us ({2*mb/t2/1e6:5.2f} TB/s)")
