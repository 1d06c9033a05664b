"""GPU parity of the preprocessing and head rows through the C ABI.

Preprocessing is integer / table work: the HIP path must be BIT-EXACT against the golden vectors
(transformers' PIL pipeline) and against the oracle on seeded inputs; bf16 output must equal the
fp32 result rounded to bf16. Heads: fp32 within 1e-4 relative, bf16 within bf16 round-off."""
import ast
import glob
import os

import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as po
from tests.helpers import GOLDEN, rel_err
from vit_amd import synth

pytestmark = pytest.mark.gpu

CASES = sorted(os.path.basename(p)[len("preprocess_"):-4] for p in glob.glob(os.path.join(GOLDEN, "preprocess_*.npz")))


def make_processor(kw, dtype=torch.float32):
    from vit_amd.image_processor import ImageProcessor
    return ImageProcessor(device="cuda:0", dtype=dtype, **kw)


@pytest.mark.parametrize("name", CASES)
def test_golden_bit_exact(name):
    z = np.load(os.path.join(GOLDEN, f"preprocess_{name}.npz"), allow_pickle=False)
    h, w, c = (int(v) for v in z["in_shape"])
    img = synth.synthetic_uint8_image(h, w, c, seed=2, name=name)
    kw = ast.literal_eval(str(z["kwargs_json"]))
    proc = make_processor(kw)
    out = proc(img)["pixel_values"]
    torch.cuda.synchronize()
    assert out.shape[0] == 1 and out.dtype == torch.float32
    assert np.array_equal(out[0].cpu().numpy(), z["pixel_values"])
    # bf16 output = the same values rounded once
    out16 = make_processor(kw, torch.bfloat16)(img)["pixel_values"]
    assert torch.equal(out16[0].cpu(), torch.from_numpy(z["pixel_values"]).to(torch.bfloat16))


@pytest.mark.parametrize("resample", [po.LANCZOS, po.BILINEAR, po.BICUBIC, po.BOX, po.HAMMING])
@pytest.mark.parametrize("shape", [((61, 47, 3), (32, 40)), ((33, 130, 1), (64, 64)), ((16, 16, 3), (16, 16)),
                                   ((7, 9, 3), (1, 1)), ((200, 3, 3), (3, 200))])
def test_oracle_bit_exact_filters(resample, shape):
    (h, w, c), (oh, ow) = shape
    img = synth.synthetic_uint8_image(h, w, c, seed=7, name=f"f{h}x{w}")
    kw = dict(size={"height": oh, "width": ow}, resample=resample, image_mean=(0.3, 0.5, 0.7)[:c],
              image_std=(0.2, 0.25, 0.3)[:c])
    want = po.preprocess(img.numpy(), do_resize=True, do_center_crop=False, crop_size=None, do_rescale=True,
                         rescale_factor=1 / 255, do_normalize=True, **kw)
    got = make_processor(kw)(img)["pixel_values"][0].cpu().numpy()
    assert np.array_equal(got, want)


def test_batch_and_ragged_inputs():
    """A [B,H,W,C] batch equals per-image calls; a list of different sizes is grouped per geometry."""
    from vit_amd.image_processor import ImageProcessor
    proc = ImageProcessor.clip(64, device="cuda:0")
    a = torch.stack([synth.synthetic_uint8_image(90, 70, 3, seed=11, name=f"b{i}") for i in range(5)])
    out = proc(a)["pixel_values"]
    for i in range(5):
        assert torch.equal(out[i], proc(a[i])["pixel_values"][0])
    b = synth.synthetic_uint8_image(75, 120, 3, seed=11, name="other")
    both = proc([a, b.numpy()])["pixel_values"]
    assert both.shape == (6, 3, 64, 64)
    want = po.preprocess(b.numpy(), do_resize=True, size={"shortest_edge": 64}, resample=po.BICUBIC,
                         do_center_crop=True, crop_size={"height": 64, "width": 64},
                         image_mean=proc.image_mean, image_std=proc.image_std)
    assert np.array_equal(both[5].cpu().numpy(), want)


def test_large_image_properties():
    """Full-size input (a 1536x2048 render, as the reference's scripts would send): a constant image stays
    constant, and the GPU result equals the oracle on it (size-independent property + direct check)."""
    from vit_amd.image_processor import ImageProcessor
    proc = ImageProcessor.clip(336, device="cuda:0")
    const = torch.full((1536, 2048, 3), 201, dtype=torch.uint8)
    out = proc(const)["pixel_values"][0].cpu()
    for ch in range(3):
        v = np.float32(np.float32(np.float64(201) * (1 / 255)) - np.float32(proc.image_mean[ch])) / np.float32(proc.image_std[ch])
        assert torch.all(out[ch] == float(v))
    img = synth.synthetic_uint8_image(768, 1024, 3, seed=13, name="big")
    want = po.preprocess(img.numpy(), do_resize=True, size={"shortest_edge": 336}, resample=po.BICUBIC,
                         do_center_crop=True, crop_size={"height": 336, "width": 336},
                         image_mean=proc.image_mean, image_std=proc.image_std)
    assert np.array_equal(proc(img)["pixel_values"][0].cpu().numpy(), want)


def test_errors():
    from vit_amd import _lib
    from vit_amd.image_processor import ImageProcessor
    proc = ImageProcessor(device="cuda:0")
    with pytest.raises(TypeError, match="uint8"):
        proc(torch.zeros(8, 8, 3))
    with pytest.raises(ValueError, match="1 or 3 channels"):
        proc(torch.zeros(8, 8, 4, dtype=torch.uint8))
    with pytest.raises(_lib.VitError):  # rows wider than the LDS staging limit
        proc(torch.zeros(2, 6000, 3, dtype=torch.uint8))


# ---- heads ----
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_heads_golden_and_oracle(dtype):
    from vit_amd import _lib
    lib = _lib.load()
    from vit_amd.ops import dtype_code
    z = np.load(os.path.join(GOLDEN, "heads_tiny.npz"), allow_pickle=False)
    dev = "cuda:0"
    w = synth.normal((64, 64), 0, "pooler.dense.weight", std=0.2).to(dev, dtype)
    b = synth.normal((64,), 0, "pooler.dense.bias", std=0.1).to(dev)
    hidden = synth.normal((5, 17, 64), 1, "hidden").to(dev, dtype)
    out = torch.empty(5, 64, dtype=dtype, device=dev)
    _lib.check(lib.vit_head_fwd(hidden.data_ptr(), 17 * 64, w.data_ptr(), b.data_ptr(), out.data_ptr(), 5, 64, 64,
                                _lib.HEAD_TANH, dtype_code(dtype), None))
    wp = synth.normal((40, 64), 0, "visual_projection.weight", std=0.2).to(dev, dtype)
    pooled = synth.normal((5, 64), 1, "pooled").to(dev, dtype)
    outp = torch.empty(5, 40, dtype=dtype, device=dev)
    _lib.check(lib.vit_head_fwd(pooled.data_ptr(), 64, wp.data_ptr(), None, outp.data_ptr(), 5, 64, 40,
                                _lib.HEAD_LINEAR, dtype_code(dtype), None))
    torch.cuda.synchronize()
    if dtype == torch.float32:
        # BASELINE.json fp32 bar: 1e-4 relative
        assert rel_err(out, torch.from_numpy(z["pooler"])) < 1e-4
        assert rel_err(outp, torch.from_numpy(z["projection"])) < 1e-4
    else:
        # operands rounded to bf16: compare with the oracle on the same rounded operands; one bf16 ulp of output
        want = po.vit_pooler(hidden.cpu(), w.cpu(), b.cpu())
        wantp = po.visual_projection(pooled.cpu(), wp.cpu())
        assert (out.float().cpu() - want).abs().max() <= 2 ** -8 * want.abs().max()
        assert (outp.float().cpu() - wantp).abs().max() <= 2 ** -8 * wantp.abs().max()


@pytest.mark.parametrize("batch,d,p", [(1, 768, 768), (256, 768, 512), (37, 1024, 768), (9, 1280, 1024)])
def test_heads_full_sizes(batch, d, p):
    from vit_amd import _lib
    from vit_amd.ops import dtype_code
    lib = _lib.load()
    dev = "cuda:0"
    n = 5
    hidden = synth.normal((batch, n, d), 3, "h").to(dev, torch.bfloat16)
    w = synth.normal((p, d), 3, "w", std=0.03).to(dev, torch.bfloat16)
    b = synth.normal((p,), 3, "b", std=0.1).to(dev)
    out = torch.empty(batch, p, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.vit_head_fwd(hidden.data_ptr(), n * d, w.data_ptr(), b.data_ptr(), out.data_ptr(), batch, d, p,
                                _lib.HEAD_TANH, dtype_code(torch.bfloat16), None))
    torch.cuda.synchronize()
    want = torch.tanh(hidden[:, 0].float().cpu() @ w.float().cpu().t() + b.cpu())
    assert (out.float().cpu() - want).abs().max() <= 2 ** -8


def test_encoder_with_heads():
    """ViTEncoder(add_pooling_layer=True) / (projection_dim=P) route the head keys and match the oracle."""
    from oracle import vit_oracle
    from vit_amd.config import ViTConfig
    from vit_amd.encoder import ViTEncoder
    cfg = ViTConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, image_size=32,
                    patch_size=8)
    sd = synth.synthetic_state_dict(cfg, seed=0)
    sd["pooler.dense.weight"] = synth.normal((64, 64), 0, "pooler.dense.weight", std=0.2)
    sd["pooler.dense.bias"] = synth.normal((64,), 0, "pooler.dense.bias", std=0.1)
    px = synth.synthetic_images(cfg, 3, seed=1)
    enc = ViTEncoder(cfg, dtype=torch.float32, device="cuda:0", add_pooling_layer=True)
    enc.load_state_dict(sd)
    out = enc(px.cuda())
    core = {k: v for k, v in sd.items() if not k.startswith("pooler.")}
    hid, _ = vit_oracle.forward(px, core, cfg)
    want = po.vit_pooler(hid, sd["pooler.dense.weight"], sd["pooler.dense.bias"])
    assert rel_err(out.pooler_output, want) < 1e-4
    with pytest.raises(KeyError):
        e2 = ViTEncoder(cfg, dtype=torch.float32, device="cuda:0", add_pooling_layer=True)
        e2.load_state_dict(core)

    ccfg = ViTConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                     hidden_act="quick_gelu", layer_norm_eps=1e-5, image_size=42, patch_size=14, variant="clip")
    csd = synth.synthetic_state_dict(ccfg, seed=0)
    wp = synth.normal((40, 128), 0, "visual_projection.weight", std=0.2)
    cpx = synth.synthetic_images(ccfg, 2, seed=1)
    cenc = ViTEncoder(ccfg, dtype=torch.float32, device="cuda:0", projection_dim=40)
    cenc.load_state_dict({**csd, "visual_projection.weight": wp})
    cout = cenc(cpx.cuda())
    _, pooled = vit_oracle.forward(cpx, csd, ccfg)
    assert cout.image_embeds.shape == (2, 40)
    assert rel_err(cout.image_embeds, po.visual_projection(pooled, wp)) < 1e-4


def test_image_to_embedding_pipeline():
    """uint8 image -> ImageProcessor -> ViTEncoder equals oracle preprocessing -> oracle encoder (fp32)."""
    from oracle import vit_oracle
    from vit_amd.config import ViTConfig
    from vit_amd.encoder import ViTEncoder
    from vit_amd.image_processor import ImageProcessor
    cfg = ViTConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, image_size=32,
                    patch_size=8)
    sd = synth.synthetic_state_dict(cfg, seed=0)
    img = synth.synthetic_uint8_image(57, 83, 3, seed=4, name="pipe")
    proc = ImageProcessor.vit(32, device="cuda:0")
    enc = ViTEncoder(cfg, dtype=torch.float32, device="cuda:0")
    enc.load_state_dict(sd)
    out = enc(proc(img)["pixel_values"])
    px = torch.from_numpy(po.preprocess(img.numpy(), size={"height": 32, "width": 32}))[None]
    hid, pooled = vit_oracle.forward(px, sd, cfg)
    assert rel_err(out.last_hidden_state, hid) < 1e-4


def test_plan_cache_is_bounded():
    from vit_amd.image_processor import ImageProcessor
    proc = ImageProcessor.vit(16, device="cuda:0")
    proc.max_plans = 3
    outs = {}
    for w in (20, 21, 22, 23, 24, 20):
        img = synth.synthetic_uint8_image(18, w, 3, seed=9, name=f"c{w}")
        out = proc(img)["pixel_values"][0].cpu().numpy()
        want = po.preprocess(img.numpy(), size={"height": 16, "width": 16})
        assert np.array_equal(out, want)
        assert len(proc._plans) <= 3
