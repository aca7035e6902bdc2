"""`from layers.heads.bev_depth_head import BEVDepthHead` (how models/bev_depth.py:6 imports it)."""
import mm_training_amd.layers.heads.bev_depth_head as _impl

BEVDepthHead = _impl.BEVDepthHead
