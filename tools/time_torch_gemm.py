"""Lab only: what torch.nn.functional.linear (hipBLASLt) reaches on the encoder's GEMM shapes.
Gives the library figure beside tools/gemm_lab's numbers for DESIGN.md; nothing in the product uses it."""
import torch

SHAPES = {"qkv": (50432, 2304, 768), "oproj": (50432, 768, 768), "fc1": (50432, 3072, 768), "fc2": (50432, 768, 3072)}


def main():
    dev = torch.device("cuda:0")
    for name, (M, N, K) in SHAPES.items():
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        b = torch.randn(N, device=dev, dtype=torch.bfloat16)
        for _ in range(5):
            torch.nn.functional.linear(a, w, b)
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.nn.functional.linear(a, w, b)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        med = ts[len(ts) // 2]
        print(f"{name:6s} torch linear  median {med:8.1f} us  {2.0 * M * N * K / med * 1e-6:8.1f} TF  (min {ts[0]:.1f} us)", flush=True)


if __name__ == "__main__":
    main()
