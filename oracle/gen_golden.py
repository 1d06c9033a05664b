"""Pin the CPU oracle against ``transformers`` and write the golden fixtures.

Runs ONLY in the build container (it imports ``transformers``, which never travels to
the GPU box). For every configuration it:
  1. builds the HF model from a local config object (no network, HF_HUB_OFFLINE=1),
  2. loads the counter-based synthetic weights of ``vit_amd.synth`` (non-trivial biases
     and LayerNorm affine) under the HF key names, strict,
  3. runs HF (eager attention, fp32) and ``oracle.vit_oracle.forward`` on the same
     synthetic images and asserts they agree to fp32 round-off,
  4. stores the HF outputs (the expected values) in ``tests/golden/<name>.npz``.
Inputs are not stored: tests regenerate them bit-exactly from (seed, name, index).

Usage:  python oracle/gen_golden.py [--full]
"""
from __future__ import annotations

import argparse
import os
import sys

os.environ.setdefault("HF_HUB_OFFLINE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import vit_oracle  # noqa: E402
from vit_amd import synth  # noqa: E402
from vit_amd.config import ViTConfig  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

# name -> (config, batch, store_full_hidden)
TINY = {
    "tiny_vit_p8": (ViTConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                              image_size=32, patch_size=8), 3, True),
    "tiny_vit_p14_dh80": (ViTConfig(hidden_size=160, num_hidden_layers=2, num_attention_heads=2,
                                    intermediate_size=320, image_size=42, patch_size=14), 2, True),
    "tiny_clip_p14": (ViTConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                hidden_act="quick_gelu", layer_norm_eps=1e-5, image_size=42, patch_size=14,
                                variant="clip"), 2, True),
    "tiny_vit_p16_n197": (ViTConfig(hidden_size=128, num_hidden_layers=1, num_attention_heads=2,
                                    intermediate_size=256, image_size=224, patch_size=16), 2, True),
}
FULL = {
    "vit_b16_224": (ViTConfig.vit_b16_224(), 8, False),
    "clip_l14_336": (ViTConfig.clip_l14_336(), 2, False),
    "vit_h14_448": (ViTConfig.vit_h14_448(), 1, False),
}


def build_hf(cfg: ViTConfig):
    import transformers
    if cfg.variant == "vit":
        hc = transformers.ViTConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                                    num_attention_heads=cfg.num_attention_heads,
                                    intermediate_size=cfg.intermediate_size, hidden_act=cfg.hidden_act,
                                    layer_norm_eps=cfg.layer_norm_eps, image_size=cfg.image_size,
                                    patch_size=cfg.patch_size, num_channels=cfg.num_channels,
                                    qkv_bias=cfg.qkv_bias, attn_implementation="eager")
        return transformers.ViTModel(hc, add_pooling_layer=False)
    hc = transformers.CLIPVisionConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                                       num_attention_heads=cfg.num_attention_heads,
                                       intermediate_size=cfg.intermediate_size, hidden_act=cfg.hidden_act,
                                       layer_norm_eps=cfg.layer_norm_eps, image_size=cfg.image_size,
                                       patch_size=cfg.patch_size, num_channels=cfg.num_channels,
                                       attn_implementation="eager")
    return transformers.CLIPVisionModel(hc)


def run(name, cfg, batch, store_full):
    sd = synth.synthetic_state_dict(cfg, seed=0)
    px = synth.synthetic_images(cfg, batch, seed=1)
    model = build_hf(cfg).eval()
    hf_sd = sd if cfg.variant == "vit" else {"vision_model." + k if not k.startswith("vision_model.") else k: v
                                             for k, v in sd.items()}
    try:
        model.load_state_dict(hf_sd, strict=True)
    except RuntimeError:
        # CLIPVisionModel in this transformers version has no "vision_model." prefix
        model.load_state_dict(sd, strict=True)
    with torch.no_grad():
        out = model(pixel_values=px)
    hf_hidden = out.last_hidden_state.float()
    hf_pooled = out.pooler_output.float() if out.pooler_output is not None else hf_hidden[:, 0]

    trace = {}
    o_hidden, o_pooled = vit_oracle.forward(px, sd, cfg, trace=trace)
    scale = hf_hidden.abs().max().item()
    err_h = (o_hidden - hf_hidden).abs().max().item()
    err_p = (o_pooled - hf_pooled).abs().max().item()
    print(f"{name}: hidden{tuple(hf_hidden.shape)} oracle-vs-HF max-abs {err_h:.3e} (pooled {err_p:.3e}), "
          f"|hidden|max {scale:.3f}")
    assert err_h <= 2e-5 * max(1.0, scale), (name, err_h)
    assert err_p <= 2e-5 * max(1.0, hf_pooled.abs().max().item()), (name, err_p)

    fx = {"config_json": np.array(repr(cfg.to_dict())), "batch": np.array(batch),
          "pooled": hf_pooled.numpy(), "cls": hf_hidden[:, 0].numpy(),
          "layer_cls_sum": np.array([trace[f"l{i}.out"][:, 0].double().sum().item()
                                     for i in range(cfg.num_hidden_layers)]),
          "hidden_sum": np.array(hf_hidden.double().sum().item()),
          "hidden_abs_sum": np.array(hf_hidden.double().abs().sum().item())}
    if store_full:
        fx["hidden"] = hf_hidden.numpy()
        fx["embed"] = trace["embed"].numpy()
        fx["l0_ln1"] = trace["l0.ln1"].numpy()
        fx["l0_ctx"] = trace["l0.ctx"].numpy()
        fx["l0_act"] = trace["l0.act"].numpy()
        fx["l0_out"] = trace["l0.out"].numpy()
    else:
        # a strided sample of patch-token rows so non-CLS paths are pinned too
        fx["hidden_rows"] = hf_hidden[:, :: max(1, cfg.seq_len // 8)].numpy()
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **fx)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also the full-size BASELINE.json configs (slow, big)")
    args = ap.parse_args()
    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(0)
    for name, (cfg, b, full) in TINY.items():
        run(name, cfg, b, full)
    if args.full:
        for name, (cfg, b, full) in FULL.items():
            run(name, cfg, b, full)


if __name__ == "__main__":
    main()
