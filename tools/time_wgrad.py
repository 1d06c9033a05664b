"""lab: event timing of vit_linear_wgrad (token-major weight gradient) at the ViT-B/16 batch-256 shapes."""
import sys, torch
sys.path.insert(0, ".")
from vit_amd import _lib, synth
lib = _lib.load()
dev = "cuda:0"
T = 50432
for mo, no, splits in [(2304, 768, 9), (768, 768, 28), (3072, 768, 7), (768, 3072, 7)]:
    dy = synth.normal((T, mo), 3, "dy", device=dev).to(torch.bfloat16)
    x = synth.normal((T, no), 4, "x", device=dev).to(torch.bfloat16)
    dw = torch.empty(mo, no, dtype=torch.float32, device=dev)
    db = torch.zeros(mo, dtype=torch.float32, device=dev)
    slabs = torch.empty(splits, mo, no, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        _lib.check(lib.vit_linear_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), slabs.data_ptr(), T, mo, no, splits, st))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"wgrad T={T} out={mo} in={no} splits={splits}: {us:.1f} us (with slab reduction)  {2.0*T*mo*no/us*1e-6:.0f} TFLOP/s", flush=True)
