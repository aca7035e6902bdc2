"""HIP-backed replacement for the reference package ``ops.voxel_pooling``: the public name
is the autograd op ``voxel_pooling(geom_xyz, input_features, voxel_num)``; the extension
module lives next to it as ``voxel_pooling_ext`` (a ctypes binding of libmmt_hip.so)."""
from . import voxel_pooling as _op
from . import voxel_pooling_ext  # noqa: F401  (importable like the reference's pybind module)

from .plan import VoxelPoolingPlan, voxel_pooling_planned  # noqa: F401  (cached-sort variant)

voxel_pooling = _op.voxel_pooling
voxel_pooling_bf16 = _op.voxel_pooling_bf16      # bf16 feature storage, fp32 accumulate

__all__ = ("voxel_pooling", "voxel_pooling_bf16", "voxel_pooling_ext", "VoxelPoolingPlan", "voxel_pooling_planned")
