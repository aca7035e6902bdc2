"""Model-level mirrors of the reference's ``models`` package: the camera-only BEVDepth detector,
the camera|LiDAR fusion layer and the multi-modal BEVDepthLiDAR network whose three LiDAR calls
(``voxelize`` / ``pts_voxel_encoder`` / ``pts_middle_encoder``) and camera branch run on the HIP
kernels of this repository."""
from . import bev_depth as _bev_depth

BEVDepth = _bev_depth.BEVDepth
BEVFuseLayer = _bev_depth.BEVFuseLayer
BEVDepthLiDAR = _bev_depth.BEVDepthLiDAR

__all__ = ("BEVDepth", "BEVFuseLayer", "BEVDepthLiDAR")
