"""`from layers.backbones import LSSFPN` (reference: layers/backbones/__init__.py:1-3) -> the HIP-backed camera branch."""
from mm_training_amd.layers.backbones import LSSFPN as _impl

LSSFPN = _impl

__all__ = ("LSSFPN",)
