"""The local-vision-tower adapter fed the way the reference's call sites feed
their remote VLM: a file path (Vlm_coder.py:24-26), its base64 text (:205, :381 `images=[image_b64]`)."""
import base64
import io

import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as po
from oracle import vit_oracle
from tests.helpers import rel_err
from vit_amd import synth
from vit_amd.config import ViTConfig

pytestmark = pytest.mark.gpu
PIL_Image = pytest.importorskip("PIL.Image")


def _png(img: np.ndarray) -> bytes:
    buf = io.BytesIO()
    PIL_Image.fromarray(img).save(buf, format="PNG")  # lossless: decoded pixels == img
    return buf.getvalue()


def test_tower_matches_oracle_chain(tmp_path):
    from vit_amd.image_processor import OPENAI_CLIP_MEAN, OPENAI_CLIP_STD
    from vit_amd.vlm_shim import LocalVisionTower, wrap_generate
    cfg = ViTConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                    hidden_act="quick_gelu", layer_norm_eps=1e-5, image_size=42, patch_size=14, variant="clip")
    sd = synth.synthetic_state_dict(cfg, seed=0)
    wp = synth.normal((40, 128), 0, "visual_projection.weight", std=0.2)
    tower = LocalVisionTower(cfg, {**sd, "visual_projection.weight": wp}, dtype=torch.float32, device="cuda:0",
                             projection_dim=40)
    imgs = [synth.synthetic_uint8_image(61, 95, 3, seed=21, name="a").numpy(),
            synth.synthetic_uint8_image(120, 77, 3, seed=21, name="b").numpy()]
    path = tmp_path / "drawing.png"
    path.write_bytes(_png(imgs[0]))
    b64 = base64.b64encode(_png(imgs[1])).decode("utf-8")

    px = torch.stack([torch.from_numpy(po.preprocess(
        im, do_resize=True, size={"shortest_edge": 42}, resample=po.BICUBIC, do_center_crop=True,
        crop_size={"height": 42, "width": 42}, image_mean=OPENAI_CLIP_MEAN, image_std=OPENAI_CLIP_STD)) for im in imgs])
    hid, pooled = vit_oracle.forward(px, sd, cfg)
    want = po.visual_projection(pooled, wp)

    out = tower.encode([str(path), b64])
    assert rel_err(out.last_hidden_state, hid) < 1e-4
    assert rel_err(out.image_embeds, want) < 1e-4

    calls = []

    def fake_client_generate(model, prompt, images, options):
        calls.append((model, prompt, len(images), options))
        return {"response": "ok"}

    gen = wrap_generate(tower, fake_client_generate)
    r = gen(model="qwen2.5vl:72b", prompt="describe", images=[b64], options={"temperature": 0.2})
    assert r == {"response": "ok"} and calls == [("qwen2.5vl:72b", "describe", 1, {"temperature": 0.2})]
    assert rel_err(gen.last_embeddings.image_embeds, want[1:]) < 1e-4
    local = wrap_generate(tower)(model="m", prompt="p", images=[imgs[0]])
    assert rel_err(local["image_embeds"], want[:1]) < 1e-4 and local["patch_embeds"].shape == (1, 10, 128)
