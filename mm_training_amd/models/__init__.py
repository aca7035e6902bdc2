from .bev_depth import BEVDepth, BEVDepthLiDAR, BEVFuseLayer

__all__ = ['BEVDepth', 'BEVDepthLiDAR', 'BEVFuseLayer']
