"""Cached-plan voxel pooling ("geometry + quantise with cached sort").

The reference re-derives the point -> BEV-cell assignment inside every ``voxel_pooling`` call
(ops/voxel_pooling/src/voxel_pooling_forward_cuda.cu:19-29) although ``geom_xyz`` only depends
on the camera calibration (layers/backbones/lss_fpn.py:328-361,461-462; BDA is disabled there,
:355-360).  ``VoxelPoolingPlan(geom_xyz, voxel_num)`` does the sort by cell once;
``voxel_pooling_planned(plan, input_features)`` then returns the same ``[B, C, ny, nx]``
channels-last view as ``voxel_pooling(geom_xyz, input_features, voxel_num)`` from a pure
segmented gather (no atomics: the result is bit-reproducible) and back-propagates through
the same HIP backward kernel, fed by the plan's cached ``pos_memo``.

An ADDITIONAL API beside the drop-in op; HIP only, like everything in this package.
"""
import ctypes

import torch
from torch.autograd import Function

from . import voxel_pooling_ext
from .voxel_pooling import _voxel_num_to_ints
from ... import _lib


class VoxelPoolingPlan:
    """Device-resident sort of the points of ``geom_xyz`` by BEV cell.

    Building launches asynchronously and then reads four counters back (one stream
    synchronisation per plan, none per step)."""

    def __init__(self, geom_xyz: torch.Tensor, voxel_num):
        voxel_pooling_ext._check_input(geom_xyz, "geom_xyz_tensor", torch.int32)
        if geom_xyz.shape[-1] != 3:
            raise RuntimeError("geom_xyz must be [B, ..., 3]")
        B = int(geom_xyz.shape[0])
        P = geom_xyz.numel() // (3 * B)
        nx, ny, nz = _voxel_num_to_ints(voxel_num)
        dev = geom_xyz.device
        self.batch_size, self.num_points, self.grid = B, P, (nx, ny, nz)
        n_plan = _lib.lib().mmt_voxel_pooling_plan_elems(B, P, nx, ny)
        n_ws = _lib.lib().mmt_voxel_pooling_plan_workspace_bytes(B, P, nx, ny)
        if n_plan < 0 or n_ws < 0:
            raise _lib.MmtError(_lib.lib().mmt_last_error().decode())
        self.plan = torch.empty(n_plan, dtype=torch.int32, device=dev)
        self.pos_memo = torch.empty((B, P, 3), dtype=torch.int32, device=dev)
        workspace = torch.empty(n_ws, dtype=torch.uint8, device=dev)
        info = (ctypes.c_int32 * 4)()
        with torch.cuda.device(dev):
            stream = voxel_pooling_ext._stream()
            _lib.call("mmt_voxel_pooling_plan_build", B, P, nx, ny, nz, geom_xyz.data_ptr(),
                      self.pos_memo.data_ptr(), self.plan.data_ptr(), n_plan,
                      workspace.data_ptr(), n_ws, stream)
            _lib.call("mmt_voxel_pooling_plan_info", self.plan.data_ptr(),
                      ctypes.cast(info, ctypes.c_void_p), stream)
        self.num_items, self.num_kept, self.num_multi, self.num_partial = (int(v) for v in info)
        self._partial = {}

    def partial_buffer(self, num_channels):
        """Scratch for cells that hold more than one item (reused across steps)."""
        buf = self._partial.get(num_channels)
        if buf is None:
            buf = torch.empty(max(1, self.num_partial) * num_channels, dtype=torch.float32,
                              device=self.plan.device)
            self._partial[num_channels] = buf
        return buf


def planned_forward_into(plan: VoxelPoolingPlan, input_features: torch.Tensor,
                         out: torch.Tensor, row_stride: int):
    """Launch the planned forward writing the pooled rows at ``out.data_ptr() + cell * row_stride``
    (floats).  ``out`` may be a wider channels-last buffer (camera|LiDAR concat)."""
    voxel_pooling_ext._check_input(input_features, "input_features_tensor", torch.float32)
    B, P = plan.batch_size, plan.num_points
    C = int(input_features.shape[-1])
    nx, ny, _ = plan.grid
    if input_features.numel() != B * P * C:
        raise RuntimeError("input_features do not match the plan (batch_size, num_points)")
    if out.dtype != torch.float32 or not out.is_cuda:
        raise RuntimeError("output must be a float32 CUDA tensor")
    partial = plan.partial_buffer(C)
    with torch.cuda.device(input_features.device):
        voxel_pooling_ext._timed_call(
            "forward", "mmt_voxel_pooling_forward_planned", B, P, C, nx, ny, plan.plan.data_ptr(),
            plan.num_items, plan.num_multi, plan.num_partial, input_features.data_ptr(),
            out.data_ptr(), int(row_stride), partial.data_ptr(), partial.numel(),
            voxel_pooling_ext._stream())


class VoxelPoolingPlanned(Function):
    @staticmethod
    def forward(ctx, plan: VoxelPoolingPlan, input_features: torch.Tensor) -> torch.Tensor:
        assert input_features.is_contiguous()
        B = plan.batch_size
        C = int(input_features.shape[-1])
        nx, ny, _ = plan.grid
        # every BEV row is written by the kernel: no zero-fill (voxel_pooling.py:37-38)
        output_features = torch.empty((B, ny, nx, C), dtype=torch.float32, device=input_features.device)
        planned_forward_into(plan, input_features, output_features, C)
        ctx.plan = plan
        ctx.feat_shape = input_features.shape
        return output_features.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_output_features):
        plan = ctx.plan
        nx, ny, _ = plan.grid
        B, P = plan.batch_size, plan.num_points
        C = ctx.feat_shape[-1]
        grad_input_features = torch.empty(ctx.feat_shape, dtype=torch.float32, device=plan.plan.device)
        workspace = torch.empty(voxel_pooling_ext.backward_workspace_elems(B, P, C, nx, ny),
                                dtype=torch.float32, device=plan.plan.device)
        voxel_pooling_ext.voxel_pooling_backward_wrapper(
            B, P, C, nx, ny, plan.pos_memo, grad_output_features, grad_input_features, workspace)
        return None, grad_input_features


def voxel_pooling_planned(plan: VoxelPoolingPlan, input_features: torch.Tensor) -> torch.Tensor:
    return VoxelPoolingPlanned.apply(plan, input_features)
