"""vit_forward_static: the no-allocation entry point (include/vit_amd.h). It has no counterpart in the
reference; it is checked bit for bit against vit_forward (the existing entry point with a caller
workspace) through the C ABI, on two calls in a row, the second with a smaller input, so that whatever
the first call left in the library's static scratch array must not matter."""
import ctypes

import pytest
import torch

from vit_amd import _lib, synth
from vit_amd.config import ViTConfig
from vit_amd.encoder import ViTEncoder
from vit_amd.ops import dtype_code

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SCRATCH_BYTES = 32 << 20  # VIT_STATIC_SCRATCH_BYTES


def _forward_static(enc, px):
    cfg = enc.config
    b = px.shape[0]
    hidden = torch.empty(b, cfg.seq_len, cfg.hidden_size, dtype=enc.dtype, device=enc.device)
    pooled = torch.empty(b, cfg.hidden_size, dtype=enc.dtype, device=enc.device)
    with torch.cuda.device(enc.device):
        rc = _lib.load().vit_forward_static(enc._ctx, px.data_ptr(), dtype_code(px.dtype), b, hidden.data_ptr(),
                                            pooled.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc)
    return hidden, pooled


@pytest.mark.parametrize("variant,dtype", [("vit", torch.bfloat16), ("vit", torch.float32), ("clip", torch.bfloat16)])
def test_forward_static_matches_forward_twice(variant, dtype):
    if variant == "vit":
        cfg = ViTConfig(hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=512,
                        image_size=96, patch_size=16)
    else:
        cfg = ViTConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                        image_size=70, patch_size=14, variant="clip", hidden_act="quick_gelu", layer_norm_eps=1e-5)
    enc = ViTEncoder(cfg, dtype=dtype, device=DEV)
    enc.load_state_dict(synth.synthetic_state_dict(cfg, seed=0))
    for batch, seed in ((9, 1), (4, 2)):  # second call: smaller input over the first call's leftovers
        assert enc.workspace_bytes(batch) <= SCRATCH_BYTES
        px = synth.synthetic_images(cfg, batch, seed=seed, device=DEV).to(dtype).contiguous()
        ref = enc(px)
        hidden, pooled = _forward_static(enc, px)
        torch.cuda.synchronize()
        assert torch.isfinite(hidden.float()).all()
        assert torch.equal(hidden, ref.last_hidden_state)
        assert torch.equal(pooled, ref.pooler_output)
    enc.close()


def test_forward_static_vit_b16_limit_and_rejection():
    """ViT-B/16 bf16: the largest batch the static scratch holds runs and matches vit_forward; one image
    more is rejected with VIT_E_NOMEM and a message that names both sizes."""
    cfg = ViTConfig.vit_b16_224()
    enc = ViTEncoder(cfg, dtype=torch.bfloat16, device=DEV)
    enc.load_state_dict(synth.synthetic_state_dict(cfg, seed=0))
    fit = max(b for b in range(1, 64) if enc.workspace_bytes(b) <= SCRATCH_BYTES)
    assert fit == 10
    for batch in (fit, 3):
        px = synth.synthetic_images(cfg, batch, seed=3, device=DEV).to(torch.bfloat16).contiguous()
        ref = enc(px)
        hidden, pooled = _forward_static(enc, px)
        assert torch.equal(hidden, ref.last_hidden_state)
        assert torch.equal(pooled, ref.pooler_output)
    px = synth.synthetic_images(cfg, fit + 1, seed=3, device=DEV).to(torch.bfloat16).contiguous()
    with pytest.raises(_lib.VitError, match="static scratch holds") as ei:
        _forward_static(enc, px)
    assert ei.value.code == -4
    with pytest.raises(_lib.VitError, match="batch must be positive"):
        _lib.check(_lib.load().vit_forward_static(enc._ctx, px.data_ptr(), 1, 0, px.data_ptr(), None, None))
    enc.close()
