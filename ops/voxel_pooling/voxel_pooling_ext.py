"""Alias of the reference's extension module ops.voxel_pooling.voxel_pooling_ext (built by setup.py:60-67 from
src/voxel_pooling_forward.cpp / _cuda.cu): `voxel_pooling_forward_wrapper` with the reference's 10 arguments
(voxel_pooling_forward.cpp:24-25), here the ctypes binding of libmmt_hip.so, plus the new backward wrapper."""
import mm_training_amd.ops.voxel_pooling.voxel_pooling_ext as _impl

voxel_pooling_forward_wrapper = _impl.voxel_pooling_forward_wrapper
voxel_pooling_backward_wrapper = _impl.voxel_pooling_backward_wrapper
