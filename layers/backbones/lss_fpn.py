"""`from layers.backbones.lss_fpn import LSSFPN` (how models/bev_depth.py:5 and test/test_layers/test_backbone.py import it)."""
import mm_training_amd.layers.backbones.lss_fpn as _impl

LSSFPN = _impl.LSSFPN
DepthNet = _impl.DepthNet
ASPP = _impl.ASPP
