"""CPU tests of the preprocessing / heads rows: the oracle against the
committed golden vectors (generated from transformers' PIL pipeline by oracle/gen_golden_preprocess.py),
and the host-side geometry logic of vit_amd.image_processor. No GPU, no compute calls into the library."""
import ast
import glob
import os

import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as po
from tests.helpers import GOLDEN
from vit_amd import synth

CASES = sorted(os.path.basename(p)[len("preprocess_"):-4] for p in glob.glob(os.path.join(GOLDEN, "preprocess_*.npz")))


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f"preprocess_{name}.npz"), allow_pickle=False)
    h, w, c = (int(v) for v in z["in_shape"])
    return synth.synthetic_uint8_image(h, w, c, seed=2, name=name), ast.literal_eval(str(z["kwargs_json"])), z["pixel_values"]


def test_cases_present():
    assert len(CASES) >= 10


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_golden_bit_exact(name):
    img, kw, want = load_case(name)
    got = po.preprocess(img.numpy(), **kw)
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got, want)


def test_identity_resize_is_a_copy():
    img = synth.synthetic_uint8_image(19, 23, 3, seed=5, name="id").numpy()
    for f in (po.LANCZOS, po.BILINEAR, po.BICUBIC, po.BOX, po.HAMMING):
        # same-size taps collapse to one coefficient of 2^22: the pass is the identity
        b, k = po.precompute_coeffs(23, 23, f)
        assert np.array_equal(po._pass(img, b, k, axis=1), img)


def test_coefficients_sum_to_one():
    for f in (po.LANCZOS, po.BILINEAR, po.BICUBIC, po.BOX, po.HAMMING):
        for (i, o) in [(451, 224), (37, 224), (300, 7), (5, 5)]:
            b, k = po.precompute_coeffs(i, o, f)
            assert np.all(np.abs(k.sum(axis=1) - (1 << po.PRECISION_BITS)) <= k.shape[1])
            assert np.all(b[:, 0] >= 0) and np.all(b[:, 0] + b[:, 1] <= i) and np.all(b[:, 1] <= k.shape[1])


def test_constant_image_stays_constant():
    img = np.full((40, 61, 3), 137, dtype=np.uint8)
    out = po.pil_resize(img, 17, 29, po.BICUBIC)
    assert np.all(out == 137)


def test_center_crop_padding_branch():
    img = np.arange(3 * 5 * 1, dtype=np.uint8).reshape(3, 5, 1) + 1
    out = po.center_crop(img, 6, 4)  # taller than the image (zero rows), narrower (columns cut)
    assert out.shape == (6, 4, 1)
    top = (3 - 6) // 2  # -2 -> rows 2..4 hold data
    assert np.all(out[:2] == 0) and np.all(out[5:] == 0)
    assert np.array_equal(out[2:5, :, 0], img[:, 0:4, 0])
    assert top == -2


def test_heads_golden():
    z = np.load(os.path.join(GOLDEN, "heads_tiny.npz"), allow_pickle=False)
    w = synth.normal((64, 64), 0, "pooler.dense.weight", std=0.2)
    b = synth.normal((64,), 0, "pooler.dense.bias", std=0.1)
    hidden = synth.normal((5, 17, 64), 1, "hidden")
    assert np.abs(po.vit_pooler(hidden, w, b).numpy() - z["pooler"]).max() < 1e-6
    wp = synth.normal((40, 64), 0, "visual_projection.weight", std=0.2)
    pooled = synth.normal((5, 64), 1, "pooled")
    assert np.abs(po.visual_projection(pooled, wp).numpy() - z["projection"]).max() < 1e-6


# ---- host logic of the product class (no GPU needed: constructed without touching the device) ----
def _host_processor(**kw):
    from vit_amd.image_processor import ImageProcessor
    p = ImageProcessor.__new__(ImageProcessor)
    base = dict(do_resize=True, size={"height": 224, "width": 224}, resample=2, do_center_crop=False,
                crop_size={"height": 224, "width": 224}, do_rescale=True, rescale_factor=1 / 255, do_normalize=True,
                image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5))
    base.update(kw)
    for k, v in base.items():
        setattr(p, k, v)
    p._plans = {}
    return p


def test_shortest_edge_rule_matches_oracle():
    p = _host_processor(size={"shortest_edge": 224})
    for (h, w) in [(300, 451), (451, 300), (224, 224), (1000, 37), (480, 640)]:
        assert p.resized_hw(h, w) == po.shortest_edge_size(h, w, 224)


def test_plan_config_fields():
    p = _host_processor(size={"shortest_edge": 32}, do_center_crop=True, crop_size={"height": 48, "width": 40}, resample=3)
    c = p.plan_config(40, 90, 3)
    assert (c.in_height, c.in_width, c.channels) == (40, 90, 3)
    assert (c.resize_height, c.resize_width) == (32, 72)
    assert (c.crop_height, c.crop_width) == (48, 40)
    assert c.resample == 3 and c.do_rescale == 1 and c.do_normalize == 1
    with pytest.raises(ValueError):
        p.plan_config(40, 90, 4)


def test_gpu_only_constructor():
    from vit_amd.image_processor import ImageProcessor
    with pytest.raises(ValueError, match="GPU only"):
        ImageProcessor(device="cpu")
    with pytest.raises(ValueError, match="resample"):
        ImageProcessor(resample=0, device="cpu")


def test_pos_interp_oracle_matches_golden():
    from oracle import vit_oracle
    z = np.load(os.path.join(GOLDEN, "pos_interp_tiny.npz"), allow_pickle=False)
    pos = synth.normal((1, 17, 64), 0, "pos", std=0.5)
    for key in ("g6x6", "g2x2", "g5x9", "g1x1"):
        gh, gw = (int(v) for v in key[1:].split("x"))
        got = vit_oracle.interpolate_pos_encoding(pos, gh, gw).numpy()
        assert np.abs(got - z[key]).max() <= 1e-6
    p2 = synth.normal((5, 64), 0, "pos2", std=0.5)
    assert np.abs(vit_oracle.interpolate_pos_encoding(p2, 3, 3).numpy() - z["clip_g3x3"]).max() <= 1e-6
    # same grid: bicubic with align_corners=False at scale 1 is the identity
    assert torch.allclose(vit_oracle.interpolate_pos_encoding(pos, 4, 4), pos[0], atol=1e-6)
