"""Time the preprocessing kernels on the GPU: a batch of decoded 768x1024 RGB
renders -> CLIP 336 pixel_values, events on the launch stream; prints achieved GB/s against the
algorithmic bytes (uint8 input rows the crop needs + output tensor). Also times the numpy oracle on
one image as the CPU figure. Usage: python tools/time_preprocess.py [--batch 64]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vit_amd import synth  # noqa: E402
from vit_amd.image_processor import ImageProcessor  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--height", type=int, default=768)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    img = synth.synthetic_uint8_image(a.height, a.width, 3, seed=2, name="bench", device="cuda")
    batch = img[None].expand(a.batch, -1, -1, -1).contiguous()
    for name, proc in (("clip336_bf16", ImageProcessor.clip(336, device="cuda:0", dtype=torch.bfloat16)),
                       ("vit224_bf16", ImageProcessor.vit(224, device="cuda:0", dtype=torch.bfloat16))):
        for _ in range(3):
            out = proc(batch)["pixel_values"]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            out = proc(batch)["pixel_values"]
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        alg = batch.numel() + out.numel() * out.element_size()
        print(f"{name}: batch {a.batch} {a.height}x{a.width} -> {tuple(out.shape[1:])}: {ms * 1e3:.1f} us/batch, "
              f"{a.batch / ms * 1e3:.0f} img/s, {alg / ms / 1e6:.1f} GB/s algorithmic (in {batch.numel() / 1e6:.1f} MB + "
              f"out {out.numel() * out.element_size() / 1e6:.1f} MB)")
    if a.cpu:
        from oracle import preprocess_oracle as po
        x = img.cpu().numpy()
        t = time.perf_counter()
        po.preprocess(x, size={"shortest_edge": 336}, resample=po.BICUBIC, do_center_crop=True,
                      crop_size={"height": 336, "width": 336})
        print(f"numpy oracle, 1 image, 1 core: {time.perf_counter() - t:.3f} s")


if __name__ == "__main__":
    main()
