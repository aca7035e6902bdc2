"""``voxel_pooling(geom_xyz, input_features, voxel_num)`` -- drop-in for
ops/voxel_pooling/voxel_pooling.py:8-72 of the reference.

Same call signature, same result (a ``[B, C, ny, nx]`` permuted VIEW of the
channels-last ``[B, ny, nx, C]`` buffer -- i.e. a torch.channels_last tensor), same
gradient contract ``(None, grad_input_features, None)``.  Differences, all host-side:
  * the backward is one HIP gather kernel instead of three boolean-mask selects and
    an advanced-index gather (voxel_pooling.py:60-66);
  * no ``zeros_like(input_features)`` is allocated in forward (voxel_pooling.py:29);
    backward writes every row of a fresh buffer;
  * ``pos_memo`` is not pre-filled with -1 on the host side (voxel_pooling.py:40):
    the kernel writes -1 rows for dropped points itself;
  * ``voxel_num`` may be a tensor (CPU or CUDA; a CUDA tensor costs ONE device->host
    copy instead of the reference's five) or a plain sequence of three ints.
"""
import torch
from torch.autograd import Function

from . import voxel_pooling_ext
from ... import _lib


def _voxel_num_to_ints(voxel_num):
    if isinstance(voxel_num, torch.Tensor):
        vals = voxel_num.detach().to("cpu").tolist()
    else:
        vals = list(voxel_num)
    if len(vals) != 3:
        raise ValueError("voxel_num must hold 3 values (x, y, z)")
    return int(vals[0]), int(vals[1]), int(vals[2])


class VoxelPooling(Function):
    @staticmethod
    def forward(ctx, geom_xyz: torch.Tensor, input_features: torch.Tensor,
                voxel_num) -> torch.Tensor:
        """geom_xyz int32 [B, ..., 3]; input_features fp32 [B, ..., C]; voxel_num (x, y, z).

        Returns the (B, C, H, W) bev feature map (voxel_pooling.py:12-24)."""
        assert geom_xyz.is_contiguous()
        assert input_features.is_contiguous()
        ctx.mark_non_differentiable(geom_xyz)
        feat_shape = input_features.shape
        geom_xyz = geom_xyz.reshape(geom_xyz.shape[0], -1, geom_xyz.shape[-1])
        input_features = input_features.reshape(geom_xyz.shape[0], -1, input_features.shape[-1])
        assert geom_xyz.shape[1] == input_features.shape[1]
        batch_size, num_points, num_channels = input_features.shape
        nx, ny, nz = _voxel_num_to_ints(voxel_num)
        output_features = input_features.new_zeros(batch_size, ny, nx, num_channels)
        pos_memo = torch.empty((batch_size, num_points, 3), dtype=torch.int32,
                               device=input_features.device)
        voxel_pooling_ext.voxel_pooling_forward_wrapper(
            batch_size, num_points, num_channels, nx, ny, nz, geom_xyz, input_features,
            output_features, pos_memo, flags=_lib.VP_ALGO_AUTO | _lib.VP_WRITE_DROPPED)
        ctx.save_for_backward(pos_memo)
        ctx.feat_shape = feat_shape
        ctx.grid = (nx, ny)
        return output_features.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_output_features):
        (pos_memo,) = ctx.saved_tensors
        nx, ny = ctx.grid
        batch_size, num_points, _ = pos_memo.shape
        num_channels = ctx.feat_shape[-1]
        grad_input_features = torch.empty(ctx.feat_shape, dtype=torch.float32,
                                          device=pos_memo.device)
        # scratch for (a) transposing an NCHW-contiguous gradient (what `.contiguous()` at
        # lss_fpn.py:467 produces) to channels-last and (b) the per-point row offsets of the
        # prepare pass; comes from torch's caching allocator, no device malloc per step
        workspace = torch.empty(
            voxel_pooling_ext.backward_workspace_elems(batch_size, num_points, num_channels, nx, ny),
            dtype=torch.float32, device=pos_memo.device)
        voxel_pooling_ext.voxel_pooling_backward_wrapper(
            batch_size, num_points, num_channels, nx, ny, pos_memo, grad_output_features,
            grad_input_features, workspace)
        return None, grad_input_features, None


voxel_pooling = VoxelPooling.apply


class VoxelPoolingBF16(Function):
    """Same op with bf16 feature storage (SURVEY section 8 row g1 / BASELINE configs[4]): ``input_features`` bf16,
    fp32 accumulation, fp32 ``[B, C, ny, nx]`` result; the gradient comes back as bf16 (the gathered fp32 BEV-gradient
    row rounded to nearest even).  Not part of the reference's surface -- its extension rejects non-float32 tensors."""

    @staticmethod
    def forward(ctx, geom_xyz, input_features, voxel_num):
        assert geom_xyz.is_contiguous()
        assert input_features.is_contiguous()
        ctx.mark_non_differentiable(geom_xyz)
        feat_shape = input_features.shape
        geom_xyz = geom_xyz.reshape(geom_xyz.shape[0], -1, geom_xyz.shape[-1])
        input_features = input_features.reshape(geom_xyz.shape[0], -1, input_features.shape[-1])
        assert geom_xyz.shape[1] == input_features.shape[1]
        batch_size, num_points, num_channels = input_features.shape
        nx, ny, nz = _voxel_num_to_ints(voxel_num)
        output_features = torch.zeros((batch_size, ny, nx, num_channels), dtype=torch.float32, device=input_features.device)
        pos_memo = torch.empty((batch_size, num_points, 3), dtype=torch.int32, device=input_features.device)
        voxel_pooling_ext.voxel_pooling_forward_wrapper_bf16(
            batch_size, num_points, num_channels, nx, ny, nz, geom_xyz, input_features, output_features, pos_memo,
            flags=_lib.VP_WRITE_DROPPED)
        ctx.save_for_backward(pos_memo)
        ctx.feat_shape = feat_shape
        ctx.grid = (nx, ny)
        return output_features.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_output_features):
        (pos_memo,) = ctx.saved_tensors
        nx, ny = ctx.grid
        batch_size, num_points, _ = pos_memo.shape
        num_channels = ctx.feat_shape[-1]
        grad_input_features = torch.empty(ctx.feat_shape, dtype=torch.bfloat16, device=pos_memo.device)
        workspace = torch.empty(
            voxel_pooling_ext.backward_workspace_elems(batch_size, num_points, num_channels, nx, ny),
            dtype=torch.float32, device=pos_memo.device)
        voxel_pooling_ext.voxel_pooling_backward_wrapper_bf16(
            batch_size, num_points, num_channels, nx, ny, pos_memo, grad_output_features.float(),
            grad_input_features, workspace)
        return None, grad_input_features, None


voxel_pooling_bf16 = VoxelPoolingBF16.apply
