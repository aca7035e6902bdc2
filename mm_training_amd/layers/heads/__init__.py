from .bev_depth_head import BEVDepthHead

__all__ = ['BEVDepthHead']
