"""Position-embedding interpolation through the C ABI: the kernel against the golden
vectors from transformers, and a resized context against the oracle run at the new resolution."""
import ctypes
import dataclasses
import os

import numpy as np
import pytest
import torch

from oracle import vit_oracle
from tests.helpers import GOLDEN, cosine, rel_err
from vit_amd import _lib, synth
from vit_amd.config import ViTConfig
from vit_amd.encoder import ViTEncoder

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _interp(pos, d, gs, gh, gw):
    src = pos.reshape(-1, d).contiguous().cuda()
    dst = torch.empty(1 + gh * gw, d, device="cuda")
    _lib.check(_lib.load().vit_pos_embed_interpolate(src.data_ptr(), dst.data_ptr(), d, gs, gh, gw, None))
    torch.cuda.synchronize()
    return dst.cpu()


def test_kernel_matches_golden():
    z = np.load(os.path.join(GOLDEN, "pos_interp_tiny.npz"), allow_pickle=False)
    pos = synth.normal((1, 17, 64), 0, "pos", std=0.5)
    for key in ("g6x6", "g2x2", "g5x9", "g1x1"):
        gh, gw = (int(v) for v in key[1:].split("x"))
        # fp32 bar of BASELINE.json: 1e-4 relative (observed ~1e-7: only the summation order differs)
        assert rel_err(_interp(pos, 64, 4, gh, gw), torch.from_numpy(z[key])) < 1e-5
    p2 = synth.normal((5, 64), 0, "pos2", std=0.5)
    assert rel_err(_interp(p2, 64, 2, 3, 3), torch.from_numpy(z["clip_g3x3"])) < 1e-5


@pytest.mark.parametrize("gs,gd,d", [(14, 16, 768), (24, 32, 1024), (16, 14, 1280), (14, 14, 768)])
def test_kernel_full_sizes(gs, gd, d):
    pos = synth.normal((1 + gs * gs, d), 5, "p", std=0.02)
    assert rel_err(_interp(pos, d, gs, gd, gd), vit_oracle.interpolate_pos_encoding(pos, gd, gd)) < 1e-5


@pytest.mark.parametrize("variant", ["vit", "clip"])
def test_resized_encoder(variant):
    if variant == "vit":
        cfg = ViTConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                        image_size=32, patch_size=8)
    else:
        cfg = ViTConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                        hidden_act="quick_gelu", layer_norm_eps=1e-5, image_size=42, patch_size=14, variant="clip")
    sd = synth.synthetic_state_dict(cfg, seed=0)
    enc = ViTEncoder(cfg, dtype=torch.float32, device="cuda:0")
    enc.load_state_dict(sd)
    new_size = cfg.image_size + 2 * cfg.patch_size
    sd2, cfg2 = vit_oracle.resized_state_dict(sd, cfg, new_size)
    px = synth.synthetic_images(cfg2, 2, seed=1)
    with pytest.raises(ValueError, match="doesn't match model"):
        enc(px.cuda())
    out = enc(px.cuda(), interpolate_pos_encoding=True)
    hid, pooled = vit_oracle.forward(px, sd2, cfg2)
    assert out.last_hidden_state.shape == hid.shape
    assert rel_err(out.last_hidden_state, hid) < 1e-4 and rel_err(out.pooler_output, pooled) < 1e-4
    # the original context is untouched and still serves its own resolution
    px0 = synth.synthetic_images(cfg, 2, seed=1)
    hid0, _ = vit_oracle.forward(px0, sd, cfg)
    assert rel_err(enc(px0.cuda()).last_hidden_state, hid0) < 1e-4
    assert enc.with_image_size(new_size) is enc.with_image_size(new_size)
    with pytest.raises(ValueError, match="square"):
        enc(torch.zeros(1, 3, new_size, cfg.image_size, device="cuda"), interpolate_pos_encoding=True)
    with pytest.raises(ValueError, match="multiple of the patch size"):
        enc.with_image_size(cfg.image_size + 1)


def test_resized_bf16_b16_384():
    """ViT-B/16 weights at 384x384 (577 tokens), bf16: CLS cosine vs the oracle >= 0.999 (BASELINE.json bar)."""
    cfg = ViTConfig(num_hidden_layers=2)
    sd = synth.synthetic_state_dict(cfg, seed=0)
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device="cuda:0")
    enc.load_state_dict(sd)
    sd2, cfg2 = vit_oracle.resized_state_dict(sd, cfg, 384)
    px = synth.synthetic_images(cfg2, 2, seed=1)
    out = enc(px.cuda(), interpolate_pos_encoding=True)
    hid, pooled = vit_oracle.forward(px, sd2, cfg2)
    assert out.last_hidden_state.shape == (2, 577, 768)
    assert cosine(out.pooler_output.float(), pooled) >= 0.999
    assert cosine(out.last_hidden_state.float(), hid) >= 0.999


def test_clone_requires_loaded_weights():
    cfg = ViTConfig(hidden_size=64, num_hidden_layers=1, num_attention_heads=4, intermediate_size=128, image_size=32,
                    patch_size=8)
    enc = ViTEncoder(cfg, dtype=torch.float32, device="cuda:0")
    with pytest.raises(_lib.VitError, match="not loaded"):
        enc.with_image_size(48)


def test_resized_clone_follows_weight_changes():
    """A clone made by with_image_size() holds a copy of the weights: after load_state_dict() or an SGD step the
    cached clone must not be served again."""
    from vit_amd.trainer import ViTTrainer
    cfg = ViTConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, image_size=32, patch_size=8)
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV)
    enc.load_state_dict(synth.synthetic_state_dict(cfg, seed=0))
    px = synth.synthetic_images(dataclasses.replace(cfg, image_size=48), 3, seed=1, device=DEV).to(torch.bfloat16)
    a = enc(px, interpolate_pos_encoding=True).pooler_output.clone()
    # (1) new weights through load_state_dict
    enc.load_state_dict(synth.synthetic_state_dict(cfg, seed=5))
    b = enc(px, interpolate_pos_encoding=True).pooler_output.clone()
    fresh = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV)
    fresh.load_state_dict(synth.synthetic_state_dict(cfg, seed=5))
    assert torch.equal(b, fresh(px, interpolate_pos_encoding=True).pooler_output)
    assert not torch.equal(a, b)
    # (2) an SGD step
    tr = ViTTrainer(enc)
    p32 = synth.synthetic_images(cfg, 3, seed=2, device=DEV).to(torch.bfloat16)
    out = tr.forward(p32)
    tr.backward(None, torch.ones_like(out.pooler_output))
    tr.sgd_step(0.5)
    c = enc(px, interpolate_pos_encoding=True).pooler_output
    assert not torch.equal(b, c)
    # the clone of the updated encoder equals a clone made from scratch now
    enc.weights_changed()
    assert torch.equal(c, enc(px, interpolate_pos_encoding=True).pooler_output)
    # (3) a clone the caller HOLDS stays a valid object across weight changes: its next forward refreshes it
    held = enc.with_image_size(48)
    enc.load_state_dict(synth.synthetic_state_dict(cfg, seed=5))
    assert torch.equal(held(px).pooler_output, b)
    assert enc.with_image_size(48) is held
    enc.close()
    with pytest.raises(RuntimeError, match="closed"):
        enc(p32)


def test_trainer_rejects_heads_and_pooler_keys_are_skipped():
    from vit_amd.trainer import ViTTrainer
    cfg = ViTConfig(hidden_size=64, num_hidden_layers=1, num_attention_heads=4, intermediate_size=128, image_size=32, patch_size=8)
    sd = synth.synthetic_state_dict(cfg, seed=0)
    sd_hf = dict(sd)  # a stock ViTModel checkpoint carries the pooler
    sd_hf["pooler.dense.weight"] = torch.zeros(64, 64)
    sd_hf["pooler.dense.bias"] = torch.zeros(64)
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV)
    enc.load_state_dict(sd_hf)  # strict, no 'unexpected key'
    with_head = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV, add_pooling_layer=True)
    with_head.load_state_dict(sd_hf)
    with pytest.raises(ValueError, match="without a head"):
        ViTTrainer(with_head)
