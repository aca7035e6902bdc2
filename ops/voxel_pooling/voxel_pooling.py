"""Alias of the reference module path ops/voxel_pooling/voxel_pooling.py (VoxelPooling autograd Function and
`voxel_pooling = VoxelPooling.apply`, :8-72): both names are the objects of mm_training_amd.ops.voxel_pooling.voxel_pooling."""
import importlib

# (the package attribute `voxel_pooling` is the op itself, as in the reference's __init__, so fetch the MODULE by name)
_impl = importlib.import_module("mm_training_amd.ops.voxel_pooling.voxel_pooling")

VoxelPooling = _impl.VoxelPooling
voxel_pooling = _impl.voxel_pooling
