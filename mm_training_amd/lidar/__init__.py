from .encoder import (LidarEncoder, hard_voxelize_batch, hard_voxelize_mean_batch, pillar_scatter,
                      pillar_scatter_from_table, pillar_scatter_strided, simple_vfe)

__all__ = ["LidarEncoder", "hard_voxelize_batch", "hard_voxelize_mean_batch", "simple_vfe", "pillar_scatter",
           "pillar_scatter_from_table", "pillar_scatter_strided"]
