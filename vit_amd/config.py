"""Encoder configuration.

Field names follow ``ViTConfig`` (transformers/models/vit/configuration_vit.py:50-65)
and ``CLIPVisionConfig`` (transformers/models/clip/configuration_clip.py:95-110) so a
user of either class can pass the same keyword arguments. The reference repository
itself defines no encoder config; these field names are the
in-container specification the survey pins the build to (section 8a).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict

VARIANT_VIT = 0   # HF ViTModel semantics (final LayerNorm over all tokens)
VARIANT_CLIP = 1  # HF CLIPVisionModel semantics (pre-LN, un-normed hidden, post-LN on CLS)

ACT_GELU = 0        # exact erf GELU (ViT default "gelu")
ACT_QUICK_GELU = 1  # x * sigmoid(1.702 x) (CLIP default "quick_gelu")

_ACT_IDS = {"gelu": ACT_GELU, "quick_gelu": ACT_QUICK_GELU}


@dataclass
class ViTConfig:
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    hidden_act: str = "gelu"
    layer_norm_eps: float = 1e-12
    image_size: int = 224
    patch_size: int = 16
    num_channels: int = 3
    qkv_bias: bool = True
    variant: str = "vit"  # "vit" | "clip"

    def __post_init__(self):
        if self.variant not in ("vit", "clip"):
            raise ValueError(f"variant must be 'vit' or 'clip', got {self.variant!r}")
        if self.hidden_act not in _ACT_IDS:
            raise ValueError(f"hidden_act must be one of {sorted(_ACT_IDS)}, got {self.hidden_act!r}")
        if self.hidden_size % self.num_attention_heads:
            raise ValueError("hidden_size must be divisible by num_attention_heads")
        if self.image_size % self.patch_size:
            raise ValueError("image_size must be divisible by patch_size")

    # ---- derived sizes (SURVEY.md section 8 notation) ----
    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid

    @property
    def seq_len(self) -> int:  # N = patches + CLS
        return self.num_patches + 1

    @property
    def patch_dim(self) -> int:  # K_p = C * P * P
        return self.num_channels * self.patch_size * self.patch_size

    @property
    def variant_id(self) -> int:
        return VARIANT_CLIP if self.variant == "clip" else VARIANT_VIT

    @property
    def act_id(self) -> int:
        return _ACT_IDS[self.hidden_act]

    def to_dict(self) -> dict:
        return asdict(self)

    # ---- algorithmic work ----
    def flops_per_image(self) -> int:
        """Forward FLOPs per image counted as 2*M*N*K for every contraction."""
        n, d, f, h, dh = self.seq_len, self.hidden_size, self.intermediate_size, self.num_attention_heads, self.head_dim
        patch = 2 * self.num_patches * self.patch_dim * d
        gemms = 2 * n * d * (3 * d) + 2 * n * d * d + 2 * 2 * n * d * f
        attn = 2 * 2 * h * n * n * dh
        return patch + self.num_hidden_layers * (gemms + attn)

    # ---- named configurations of BASELINE.json ----
    @staticmethod
    def vit_b16_224() -> "ViTConfig":
        return ViTConfig()

    @staticmethod
    def clip_l14_336() -> "ViTConfig":
        return ViTConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                         hidden_act="quick_gelu", layer_norm_eps=1e-5, image_size=336, patch_size=14,
                         variant="clip")

    @staticmethod
    def vit_h14_448() -> "ViTConfig":
        return ViTConfig(hidden_size=1280, num_hidden_layers=32, num_attention_heads=16, intermediate_size=5120,
                         image_size=448, patch_size=14)
