"""Camera branch namespace; exposes the LSSFPN mirror under the name the reference's
``layers.backbones`` package uses, so ``models/bev_depth.py`` imports resolve unchanged."""
from . import lss_fpn as _lss_fpn

LSSFPN = _lss_fpn.LSSFPN

__all__ = ("LSSFPN",)
