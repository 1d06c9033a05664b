"""MI355X-native ViT encoder."""
from .config import ViTConfig  # noqa: F401
