"""GPU parity tests of the whole encoder through vit_forward (C ABI), against the golden
fixtures (HF transformers outputs) and the oracle.

Bars (BASELINE.json north_star): fp32 mode max-abs / max-abs-ref <= 1e-4; bf16 mode CLS-embedding
cosine >= 0.999 with the max-abs error printed.
"""
import numpy as np
import pytest
import torch

from oracle import vit_oracle
from tests.helpers import cosine, load_golden, min_row_cosine, rel_err
from vit_amd import _lib, synth
from vit_amd.config import ViTConfig
from vit_amd.encoder import ViTEncoder

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TINY = ["tiny_vit_p8", "tiny_vit_p14_dh80", "tiny_clip_p14", "tiny_vit_p16_n197"]


def build(cfg, dtype):
    enc = ViTEncoder(cfg, dtype=dtype, device=DEV)
    enc.load_state_dict(synth.synthetic_state_dict(cfg, seed=0))
    return enc


@pytest.mark.parametrize("name", TINY)
def test_fp32_matches_golden_tiny(name):
    cfg, batch, fx = load_golden(name)
    out = build(cfg, torch.float32)(synth.synthetic_images(cfg, batch, seed=1))
    assert rel_err(out.last_hidden_state, torch.from_numpy(fx["hidden"])) < 1e-4
    assert rel_err(out.pooler_output, torch.from_numpy(fx["pooled"])) < 1e-4


@pytest.mark.parametrize("name", TINY)
def test_bf16_matches_golden_tiny(name):
    cfg, batch, fx = load_golden(name)
    out = build(cfg, torch.bfloat16)(synth.synthetic_images(cfg, batch, seed=1))
    ref_h, ref_p = torch.from_numpy(fx["hidden"]), torch.from_numpy(fx["pooled"])
    print(f"{name}: bf16 pooled max-abs {(out.pooler_output.float().cpu() - ref_p).abs().max():.4f}")
    assert min_row_cosine(out.pooler_output, ref_p) >= 0.999
    assert min_row_cosine(out.last_hidden_state, ref_h) >= 0.998


def test_fp32_matches_golden_vit_b16():
    cfg, batch, fx = load_golden("vit_b16_224")
    out = build(cfg, torch.float32)(synth.synthetic_images(cfg, batch, seed=1))
    assert rel_err(out.pooler_output, torch.from_numpy(fx["pooled"])) < 1e-4
    rows = out.last_hidden_state[:, :: max(1, cfg.seq_len // 8)]
    assert rel_err(rows, torch.from_numpy(fx["hidden_rows"])) < 1e-4
    assert abs(out.last_hidden_state.double().sum().item() - float(fx["hidden_sum"])) < 1e-4 * float(fx["hidden_abs_sum"])


def test_bf16_matches_golden_vit_b16():
    cfg, batch, fx = load_golden("vit_b16_224")
    out = build(cfg, torch.bfloat16)(synth.synthetic_images(cfg, batch, seed=1).to(torch.bfloat16))
    ref = torch.from_numpy(fx["pooled"])
    err = (out.pooler_output.float().cpu() - ref).abs().max().item()
    cos = min_row_cosine(out.pooler_output, ref)
    print(f"vit_b16_224 bf16: CLS min cosine {cos:.6f}, max-abs {err:.4f} (ref max {ref.abs().max():.3f})")
    assert cos >= 0.999
    assert min_row_cosine(out.last_hidden_state[:, :: max(1, cfg.seq_len // 8)], torch.from_numpy(fx["hidden_rows"])) >= 0.998


@pytest.mark.parametrize("name,dtype", [("clip_l14_336", torch.bfloat16), ("clip_l14_336", torch.float32),
                                        ("vit_h14_448", torch.bfloat16), ("vit_h14_448", torch.float32)])
def test_large_configs_match_golden(name, dtype):
    cfg, batch, fx = load_golden(name)
    out = build(cfg, dtype)(synth.synthetic_images(cfg, batch, seed=1))
    ref = torch.from_numpy(fx["pooled"])
    if dtype == torch.float32:
        assert rel_err(out.pooler_output, ref) < 1e-4
        assert rel_err(out.last_hidden_state[:, :: max(1, cfg.seq_len // 8)], torch.from_numpy(fx["hidden_rows"])) < 1e-4
    else:
        err = (out.pooler_output.float().cpu() - ref).abs().max().item()
        print(f"{name} bf16: CLS min cosine {min_row_cosine(out.pooler_output, ref):.6f}, max-abs {err:.4f}")
        assert min_row_cosine(out.pooler_output, ref) >= 0.999


def test_full_batch_properties_vit_b16_bf16():
    """BASELINE.json configs[1] size (batch 256): size-independent properties.
    Images are independent, so (a) a permuted batch gives the permuted output bit-exactly,
    (b) a sub-batch gives the same rows bit-exactly, (c) every embedding is finite and a
    sample of rows matches the CPU oracle."""
    cfg = ViTConfig.vit_b16_224()
    enc = build(cfg, torch.bfloat16)
    px = synth.synthetic_images(cfg, 256, seed=2, device=DEV).to(torch.bfloat16)
    out = enc(px)
    assert torch.isfinite(out.last_hidden_state).all()
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(0)).to(DEV)
    out_p = enc(px[perm])
    assert torch.equal(out_p.pooler_output, out.pooler_output[perm])
    assert torch.equal(out_p.last_hidden_state, out.last_hidden_state[perm])
    out_s = enc(px[100:108])
    assert torch.equal(out_s.last_hidden_state, out.last_hidden_state[100:108])
    # 4 images: every GEMM of the layer is on the generic tile kernel now (the persistent kernel takes the big
    # ones) -- both kernels give the same bits, folded LayerNorm and GELU epilogues included
    out_t = enc(px[200:204])
    assert torch.equal(out_t.last_hidden_state, out.last_hidden_state[200:204])
    sd = vit_oracle.quantize_state_dict_bf16(synth.synthetic_state_dict(cfg, seed=0))
    idx = [0, 101, 255]
    _, ref = vit_oracle.forward(px[idx].float().cpu(), sd, cfg)
    assert min_row_cosine(out.pooler_output[idx], ref) >= 0.999


@pytest.mark.parametrize("name,batch,subs,oracle_rows", [
    ("clip_l14_336", 128, ((40, 48), (101, 103)), (0, 77, 127)),   # BASELINE.json configs[2], the batch it is benched at
    ("vit_h14_448", 64, ((8, 16), (61, 63)), (0, 45)),             # configs[3]: batch per GPU of the bench line
])
def test_full_batch_properties_large_configs_bf16(name, batch, subs, oracle_rows):
    """CLIP-L/14-336 at batch 128 and ViT-H/14-448 at batch 64 -- what bench.py times for configs[2] / [3]: the
    persistent GEMM at K = 1024 / 4096 / 1280 / 5120 with quick_gelu / gelu and folded LayerNorm, the 5 : 3 two-stream
    cut at 577 / 1025 tokens, the dense patch-row route (P = 14) and the dh-64 / dh-80 ring attention kernels with many
    items per workgroup. Size-independent properties: finite; a permuted batch gives the permuted output bit for bit;
    an 8-image and a 2-image sub-batch are bit-equal to the same rows of the full batch (ties the full-batch path to
    the golden-checked small-batch path: 2 images run the generic kernels on one stream); and a few CLS rows against
    the CPU oracle on bf16-rounded weights, cosine >= 0.999 with the max-abs error printed (modeling_clip.py:362-383,
    641-656; modeling_vit.py:266-286, 385)."""
    cfg, _, _ = load_golden(name)
    enc = build(cfg, torch.bfloat16)
    px = synth.synthetic_images(cfg, batch, seed=2, device=DEV).to(torch.bfloat16)
    out = enc(px)
    hidden, pooled = out.last_hidden_state.clone(), out.pooler_output.clone()
    assert torch.isfinite(hidden).all() and torch.isfinite(pooled).all()
    perm = torch.randperm(batch, generator=torch.Generator().manual_seed(0)).to(DEV)
    out_p = enc(px[perm].contiguous())
    assert torch.equal(out_p.pooler_output, pooled[perm])
    assert torch.equal(out_p.last_hidden_state, hidden[perm])
    del out_p
    for lo, hi in subs:
        out_s = enc(px[lo:hi].contiguous())
        assert torch.equal(out_s.last_hidden_state, hidden[lo:hi]), (lo, hi)
        assert torch.equal(out_s.pooler_output, pooled[lo:hi]), (lo, hi)
    sd = vit_oracle.quantize_state_dict_bf16(synth.synthetic_state_dict(cfg, seed=0))
    idx = list(oracle_rows)
    _, ref = vit_oracle.forward(px[idx].float().cpu(), sd, cfg)
    got = pooled[idx].float().cpu()
    cos = min_row_cosine(got, ref)
    print(f"{name} bf16 batch {batch}: CLS min cosine {cos:.6f}, max-abs {(got - ref).abs().max():.4f} "
          f"(reference max-abs {ref.abs().max():.3f})")
    assert cos >= 0.999
    enc.close()


def test_hf_prefixes_and_bf16_sources_accepted():
    cfg, batch, fx = load_golden("tiny_vit_p8")
    sd = {"vit." + k: v.to(torch.bfloat16) for k, v in synth.synthetic_state_dict(cfg, seed=0).items()}
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV)
    enc.load_state_dict(sd)
    out = enc(synth.synthetic_images(cfg, batch, seed=1))
    assert min_row_cosine(out.pooler_output, torch.from_numpy(fx["pooled"])) >= 0.999


def test_error_behaviour():
    cfg, batch, _ = load_golden("tiny_vit_p8")
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV)
    px = synth.synthetic_images(cfg, batch, seed=1)
    with pytest.raises(_lib.VitError, match="not loaded"):
        enc(px)                                                   # weights missing
    sd = synth.synthetic_state_dict(cfg, seed=0)
    with pytest.raises(_lib.VitError, match="unexpected key"):
        enc.load_state_dict({**sd, "bogus.weight": torch.zeros(3)})
    with pytest.raises(_lib.VitError, match="elements"):
        enc.load_state_dict({"layernorm.weight": torch.zeros(3)})
    missing = dict(sd)
    missing.pop("layernorm.bias")
    with pytest.raises(KeyError):
        ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV).load_state_dict(missing)
    enc.load_state_dict(sd)
    with pytest.raises(ValueError, match="channel dimension"):
        enc(torch.zeros(1, 1, 32, 32))
    with pytest.raises(ValueError, match="doesn't match model"):
        enc(torch.zeros(1, 3, 40, 40))
    empty = enc(torch.zeros(0, 3, 32, 32))
    assert tuple(empty.last_hidden_state.shape) == (0, cfg.seq_len, cfg.hidden_size)
    # too-small workspace is refused by the library itself
    lib = _lib.load()
    ws = torch.empty(256, dtype=torch.uint8, device=DEV)
    h = torch.empty(batch, cfg.seq_len, cfg.hidden_size, dtype=torch.bfloat16, device=DEV)
    rc = lib.vit_forward(enc._ctx, px.to(DEV).data_ptr(), _lib.VIT_F32, batch, h.data_ptr(), None, ws.data_ptr(), 256,
                         None)
    assert rc == -4 and b"workspace" in lib.vit_last_error()


def test_determinism_repeat_runs():
    cfg, batch, _ = load_golden("tiny_vit_p16_n197")
    enc = build(cfg, torch.bfloat16)
    px = synth.synthetic_images(cfg, batch, seed=1)
    a = enc(px).last_hidden_state.clone()
    for _ in range(3):
        assert torch.equal(enc(px).last_hidden_state, a)


@pytest.mark.gpu
def test_two_stream_forward_equals_single_stream(monkeypatch):
    """vit_forward cuts batches whose smaller chunk keeps >= 6304 token rows into two chunks (5 : 3) on two streams
    (VIT_AMD_STREAMS = 1..4, read at vit_create).
    Images are independent and the kernels deterministic: the outputs must be bit-identical to the
    single-stream run, for an odd batch (uneven halves) and back-to-back calls on the same workspace."""
    import torch
    from vit_amd import synth
    from vit_amd.config import ViTConfig
    from vit_amd.encoder import ViTEncoder
    cfg = ViTConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256)
    sd = synth.synthetic_state_dict(cfg, seed=0)
    # 141 images x 197 tokens: two chunks 87 | 54, three of 47, four of 35 -- every chunk above the 6304-row minimum
    px = synth.synthetic_images(cfg, 141, seed=1).cuda().to(torch.bfloat16)
    outs = []
    for k in ("1", "2", "3", "4"):
        monkeypatch.setenv("VIT_AMD_STREAMS", k)
        enc = ViTEncoder(cfg, dtype=torch.bfloat16, device="cuda:0")
        enc.load_state_dict(sd)
        a = enc(px)
        b = enc(px)  # second call reuses the workspace while nothing else synchronises in between
        torch.cuda.synchronize()
        assert torch.equal(a.last_hidden_state, b.last_hidden_state)
        outs.append((a.last_hidden_state.clone(), a.pooler_output.clone()))
        enc.close()
    for h, p in outs[1:]:
        assert torch.equal(h, outs[0][0]) and torch.equal(p, outs[0][1])


@pytest.mark.gpu
def test_forward_under_graph_capture():
    """vit_forward allocates nothing and forks / joins its second stream with events, so a warmed-up call
    can be captured into a HIP graph (torch.cuda.graph) and replayed on new pixel values."""
    import torch
    from vit_amd import synth
    from vit_amd.config import ViTConfig
    from vit_amd.encoder import ViTEncoder
    cfg = ViTConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256)
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device="cuda:0")
    enc.load_state_dict(synth.synthetic_state_dict(cfg, seed=0))
    px = synth.synthetic_images(cfg, 70, seed=1).cuda().to(torch.bfloat16)
    px2 = synth.synthetic_images(cfg, 70, seed=2).cuda().to(torch.bfloat16)
    static_in = px.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm-up outside capture: one-time attribute / occupancy queries, workspace
        enc(static_in)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = enc(static_in)
    static_in.copy_(px2)
    g.replay()
    torch.cuda.synchronize()
    want = enc(px2)
    torch.cuda.synchronize()
    assert torch.equal(out.last_hidden_state, want.last_hidden_state)
    assert torch.equal(out.pooler_output, want.pooler_output)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["vit", "clip"])
def test_folded_layernorm_matches_unfolded_and_oracle(monkeypatch, variant):
    """Big enough batches fold LayerNorm into the QKV / fc1 GEMMs (vit_capi.hip refresh_folds: rstd * (x Wg^T) -
    mean * rstd * c + d in the epilogue, VIT_AMD_FUSE_LN=0 turns it off). Same function, different rounding
    points: both paths must sit equally close to the fp32 oracle (non-trivial gamma / beta), and the folded
    copies must follow a weight reload."""
    kw = dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512)
    cfg = ViTConfig(**kw) if variant == "vit" else ViTConfig(variant="clip", hidden_act="quick_gelu", layer_norm_eps=1e-5, **kw)
    batch = 44  # 44 * 197 rows: QKV and fc1 both go to the persistent kernel

    def weights(seed):
        sd = synth.synthetic_state_dict(cfg, seed=seed)
        g = torch.Generator().manual_seed(100 + seed)
        for k in sd:
            if "norm" in k and k.endswith("weight"):
                sd[k] = 1.0 + 0.3 * torch.randn(sd[k].shape, generator=g)
            elif "norm" in k and k.endswith("bias"):
                sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
        return vit_oracle.quantize_state_dict_bf16(sd)

    px = synth.synthetic_images(cfg, batch, seed=1).to(torch.bfloat16)
    outs = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("VIT_AMD_FUSE_LN", fuse)
        enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV)
        for seed in (0, 1):  # the second load must refresh the folded copies
            sd = weights(seed)
            enc.load_state_dict(sd)
            outs[(fuse, seed)] = enc(px).last_hidden_state.float().cpu()
        enc.close()
    for seed in (0, 1):
        ref = vit_oracle.forward(px.float(), weights(seed), cfg)[0]
        e_f, e_u = rel_err(outs[("1", seed)], ref), rel_err(outs[("0", seed)], ref)
        print(f"{variant} seed {seed}: folded rel err {e_f:.5f}, unfolded {e_u:.5f}")
        assert e_f < max(1.3 * e_u, 2e-3) and e_f < 2e-2
        assert min_row_cosine(outs[("1", seed)], ref) >= 0.999
        assert not torch.equal(outs[("1", seed)], outs[("0", seed)])  # the folded path really ran
    assert rel_err(outs[("1", 0)], outs[("1", 1)]) > 0.1  # different weights, different outputs


@pytest.mark.gpu
@pytest.mark.parametrize("image,batch", [(112, 256), (126, 208)])  # 64 patches (tile-aligned seams) / 81 (ragged)
def test_patch14_dense_route_matches_oracle_and_small_batches(image, batch):
    """Patch sizes whose pixel rows are not 16-byte pieces (P = 14): big batches gather the patch rows into a dense
    matrix and run the persistent GEMM on it (vit_capi.hip forward_range), small ones use the generic kernel's
    gather loader. Same bits for the same image, and both match the oracle."""
    cfg = ViTConfig(variant="clip", hidden_act="quick_gelu", layer_norm_eps=1e-5, hidden_size=256, num_hidden_layers=1,
                    num_attention_heads=4, intermediate_size=512, image_size=image, patch_size=14)
    sd = vit_oracle.quantize_state_dict_bf16(synth.synthetic_state_dict(cfg, seed=0))
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV)
    enc.load_state_dict(sd)
    px = synth.synthetic_images(cfg, batch, seed=4).to(torch.bfloat16)  # >= 16384 patch rows x 256: persistent kernel
    out = enc(px)
    sub = enc(px[17:23])                                                # generic kernel
    assert torch.equal(sub.last_hidden_state, out.last_hidden_state[17:23])
    ref_h, ref_p = vit_oracle.forward(px[:16].float(), sd, cfg)
    assert rel_err(out.last_hidden_state[:16], ref_h) < 2e-2
    assert min_row_cosine(out.pooler_output[:16], ref_p) >= 0.999
