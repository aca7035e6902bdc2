from .encoder import LidarEncoder, hard_voxelize_batch, hard_voxelize_mean_batch, simple_vfe, pillar_scatter

__all__ = ["LidarEncoder", "hard_voxelize_batch", "hard_voxelize_mean_batch", "simple_vfe", "pillar_scatter"]
