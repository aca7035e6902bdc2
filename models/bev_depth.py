"""`from models.bev_depth import BEVDepthLiDAR` (how exps/mm_training_aim.py:31 imports the network)."""
import mm_training_amd.models.bev_depth as _impl

BEVDepth = _impl.BEVDepth
BEVFuseLayer = _impl.BEVFuseLayer
BEVDepthLiDAR = _impl.BEVDepthLiDAR
