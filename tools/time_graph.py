"""lab: the forward captured in a HIP graph (torch.cuda.CUDAGraph) against plain launches.
  python tools/time_graph.py [b16|l14|h14] [batch]     (default: ViT-B/16 batch 256)"""
import sys, time, torch
sys.path.insert(0, ".")
from vit_amd import synth
from vit_amd.config import ViTConfig
from vit_amd.encoder import ViTEncoder
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "b16"
cfg = {"b16": ViTConfig.vit_b16_224, "l14": ViTConfig.clip_l14_336, "h14": ViTConfig.vit_h14_448}[name]()
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=dev)
enc.load_state_dict(synth.synthetic_state_dict(cfg, seed=0, device=dev))
px = synth.synthetic_images(cfg, batch, seed=1, device=dev).to(torch.bfloat16)
def run(fn, n=30):
    for _ in range(8): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def plain(): return enc(px)
print("plain  %.3f ms" % run(plain), flush=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): enc(px)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    out = enc(px)
torch.cuda.synchronize()
print("graph  %.3f ms" % run(g.replay), flush=True)
print("plain  %.3f ms" % run(plain), flush=True)
print("graph  %.3f ms" % run(g.replay), flush=True)
ref = enc(px)
g.replay(); torch.cuda.synchronize()
print("same bits:", torch.equal(ref.last_hidden_state, out.last_hidden_state), torch.equal(ref.pooler_output, out.pooler_output))
