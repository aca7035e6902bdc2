"""`from layers.heads import BEVDepthHead` (reference: layers/heads/__init__.py:1-3)."""
from mm_training_amd.layers.heads import BEVDepthHead as _impl

BEVDepthHead = _impl

__all__ = ("BEVDepthHead",)
