"""Top-level alias of mm_training_amd.layers under the reference's package name (layers/__init__.py:1-3 exports BEVDepthHead)."""
from mm_training_amd.layers.heads import BEVDepthHead as _head

BEVDepthHead = _head

__all__ = ("BEVDepthHead",)
