import sys, os, torch, math
sys.path.insert(0, ".")
from vit_amd import ops, synth
b, n, h, dh = 24, 197, 12, 64
d = h * dh
qkv = synth.normal((b, n, 3 * d), 13, "qkv").to(torch.bfloat16)
do = synth.normal((b, n, d), 13, "do").to(torch.bfloat16)
x = qkv.cuda(); g = do.cuda()
out, lse = ops.attention_train(x, h)
dq = ops.attention_backward(x, out, g, lse, h).float().cpu()
ref_qkv = qkv.double().requires_grad_(True)
q, k, v = ref_qkv.reshape(b, n, 3, h, dh).permute(2, 0, 3, 1, 4)
s = (q @ k.transpose(-1, -2)) * dh ** -0.5
ro = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(b, n, d)
ro.backward(do.double())
ref = ref_qkv.grad
for j, name in enumerate("qkv"):
    a, r = dq[..., j*d:(j+1)*d].double(), ref[..., j*d:(j+1)*d]
    per_item = ((a - r).reshape(b, n, h, dh).pow(2).sum((1, 3)) / r.reshape(b, n, h, dh).pow(2).sum((1, 3))).sqrt()
    print(os.environ.get("VIT_AMD_ATTN_BWD_FUSED", "1"), name, "rel rms err %.5f" % ((a - r).norm() / r.norm()).item(),
          "worst item %.5f" % per_item.max().item(), "median item %.5f" % per_item.median().item())
