"""CPU: the oracle restatement against the golden fixtures (HF transformers outputs)."""
import numpy as np
import pytest
import torch

from oracle import vit_oracle
from tests.helpers import load_golden, rel_err
from vit_amd import synth

TINY = ["tiny_vit_p8", "tiny_vit_p14_dh80", "tiny_clip_p14", "tiny_vit_p16_n197"]


@pytest.mark.parametrize("name", TINY)
def test_oracle_matches_golden_tiny(name):
    cfg, batch, fx = load_golden(name)
    sd = synth.synthetic_state_dict(cfg, seed=0)
    px = synth.synthetic_images(cfg, batch, seed=1)
    trace = {}
    hidden, pooled = vit_oracle.forward(px, sd, cfg, trace=trace)
    assert rel_err(hidden, torch.from_numpy(fx["hidden"])) < 1e-5
    assert rel_err(pooled, torch.from_numpy(fx["pooled"])) < 1e-5
    assert rel_err(trace["embed"], torch.from_numpy(fx["embed"])) < 1e-6
    assert rel_err(trace["l0.ctx"], torch.from_numpy(fx["l0_ctx"])) < 1e-5


def _full(name):
    cfg, batch, fx = load_golden(name)
    sd = synth.synthetic_state_dict(cfg, seed=0)
    px = synth.synthetic_images(cfg, batch, seed=1)
    trace = {}
    hidden, pooled = vit_oracle.forward(px, sd, cfg, trace=trace)
    assert rel_err(hidden[:, 0], torch.from_numpy(fx["cls"])) < 2e-5
    assert rel_err(pooled, torch.from_numpy(fx["pooled"])) < 2e-5
    rows = hidden[:, :: max(1, cfg.seq_len // 8)]
    assert rel_err(rows, torch.from_numpy(fx["hidden_rows"])) < 2e-5
    sums = np.array([trace[f"l{i}.out"][:, 0].double().sum().item() for i in range(cfg.num_hidden_layers)])
    np.testing.assert_allclose(sums, fx["layer_cls_sum"], rtol=0, atol=2e-3)


def test_oracle_matches_golden_vit_b16():
    _full("vit_b16_224")


@pytest.mark.slow
@pytest.mark.parametrize("name", ["clip_l14_336", "vit_h14_448"])
def test_oracle_matches_golden_large(name):
    _full(name)


def test_quick_gelu_and_gelu_definitions():
    x = torch.linspace(-4, 4, 101)
    assert torch.allclose(vit_oracle.activation(x, "gelu"), torch.nn.functional.gelu(x), atol=1e-6)
    assert torch.allclose(vit_oracle.activation(x, "quick_gelu"), x * torch.sigmoid(1.702 * x), atol=1e-7)


def test_layer_norm_matches_torch():
    x = torch.randn(5, 7, 48)
    w, b = torch.randn(48), torch.randn(48)
    ref = torch.nn.functional.layer_norm(x, (48,), w, b, 1e-12)
    assert torch.allclose(vit_oracle.layer_norm(x, w, b, 1e-12), ref, atol=1e-5)
