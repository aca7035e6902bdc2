"""Detection head namespace (mirror of the reference's ``layers.heads``)."""
from . import bev_depth_head as _head

BEVDepthHead = _head.BEVDepthHead

__all__ = ("BEVDepthHead",)
