"""`from ops.voxel_pooling import voxel_pooling` (reference: ops/voxel_pooling/__init__.py:1-3) -> the HIP-backed autograd op."""
from mm_training_amd.ops.voxel_pooling import voxel_pooling as _impl

from . import voxel_pooling_ext  # noqa: F401  (same submodule layout as the reference package)

voxel_pooling = _impl

__all__ = ("voxel_pooling",)
