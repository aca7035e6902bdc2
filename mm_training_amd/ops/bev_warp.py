"""BEV-augmentation warp of the camera BEV map -- HIP only.

``bev_warp_affine(x, bda_mat)`` is BEVDepth.bev_augment_image (models/bev_depth.py:69-84:
two kornia get_affine_matrix2d products around ``bda_mat[:3, :3]`` + ``kornia.warp_affine``)
as one launch of ``mmt_bev_warp_affine`` on the channels-last map voxel pooling produces;
``bev_warp_concat(x, bda_mat, other)`` additionally lands the warped map inside the
camera|LiDAR concat buffer (models/bev_depth.py:187-192: ``torch.cat([img_bev, lidar_bev], 1)``)
instead of materialising it and copying it again.
"""
import torch
from torch.autograd import Function

from .. import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _channels_last(x, name):
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDAtensor ")
    if x.dtype != torch.float32 or x.dim() != 4:
        raise RuntimeError(f"{name} must be a float32 [B, C, H, W] tensor")
    return x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)


class BevWarpConcat(Function):
    """out[:, :C] = warp(x), out[:, C:] = other (other may be None); out is channels-last."""

    @staticmethod
    def forward(ctx, x, bda_mat, other):
        x = _channels_last(x, "x")
        B, C, H, W = x.shape
        bda = bda_mat.detach().float().contiguous()
        c_other = 0 if other is None else int(other.shape[1])
        out = torch.empty((B, C + c_other, H, W), dtype=torch.float32, device=x.device,
                          memory_format=torch.channels_last)
        with torch.cuda.device(x.device):
            _lib.timed_call("bev_warp", "mmt_bev_warp_affine", B, H, W, C, bda.data_ptr(), x.data_ptr(), C, out.data_ptr(), C + c_other, _stream())
        if other is not None:
            out[:, C:] = other
        ctx.save_for_backward(bda)
        ctx.dims = (B, C, H, W, c_other)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (bda,) = ctx.saved_tensors
        B, C, H, W, c_other = ctx.dims
        grad_out = _channels_last(grad_out, "grad_out")
        grad_x = torch.empty((B, H, W, C), dtype=torch.float32, device=grad_out.device).permute(0, 3, 1, 2)       # (assigned: every row written once)
        with torch.cuda.device(grad_out.device):
            _lib.timed_call("bev_warp_backward", "mmt_bev_warp_affine_backward_assign", B, H, W, C, bda.data_ptr(), grad_out.data_ptr(), C + c_other,
                      grad_x.data_ptr(), C, _stream())
        grad_other = grad_out[:, C:] if c_other else None
        return grad_x, None, grad_other


class BevWarpConcatPillars(Function):
    """The whole input of the fusion layer in two launches (models/bev_depth.py:176 + :181-183 + :188-192):
        out[:, :C]  = warp(x)                                  (mmt_bev_warp_affine, as BevWarpConcat)
        out[:, C:]  = pillar canvas sampled at (i * sy, j * sx)  (mmt_pillar_scatter_nhwc[_table]_strided)
    written straight into one channels-last [B, C + Cl, H, W] buffer: the full-resolution canvas, its nearest resize, the
    slice copy into the concat buffer and -- in backward -- the canvas-sized gradient of the resize never exist.  The
    backward gathers the pillar rows' gradients out of the concat buffer's gradient at its channel offset.
    feats [M, Cl] fp32 rows + coors [M, 4] (+ the voxelizer's table, or None for the map form): LidarEncoder.forward_rows."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, bda_mat, feats, coors, table, ny, nx, max_voxels):
        x = _channels_last(x, "x")
        B, C, H, W = x.shape
        if feats.dtype != torch.float32 or not feats.is_cuda:
            raise RuntimeError(f"voxel_features must be a float32 CUDAtensor (found {feats.dtype})")
        if ny % H or nx % W:
            raise RuntimeError(f"bev_warp_concat_pillars: the pillar grid ({ny}, {nx}) is not an integer multiple of the map ({H}, {W})")
        feats = feats.contiguous()
        M, Cl = feats.shape
        sy, sx = ny // H, nx // W
        bda = bda_mat.detach().float().contiguous()
        out = torch.empty((B, C + Cl, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        lidar_ptr = out.data_ptr() + 4 * C
        cell_map = None
        with torch.cuda.device(x.device):
            _lib.timed_call("bev_warp", "mmt_bev_warp_affine", B, H, W, C, bda.data_ptr(), x.data_ptr(), C, out.data_ptr(), C + Cl, _stream())
            if table is not None:
                _lib.timed_call("scatter", "mmt_pillar_scatter_nhwc_table_strided", Cl, B, ny, nx, max_voxels, sy, sx,
                                feats.data_ptr(), table.data_ptr(), lidar_ptr, C + Cl, _stream())
            else:
                cell_map = torch.empty((B * H * W,), dtype=torch.int32, device=x.device)
                _lib.timed_call("scatter", "mmt_pillar_scatter_nhwc_strided", M, Cl, B, ny, nx, sy, sx, feats.data_ptr() if M else 0,
                                coors.data_ptr() if M else 0, lidar_ptr, C + Cl, cell_map.data_ptr(), _stream())
        ctx.save_for_backward(bda, coors, *([cell_map] if cell_map is not None else []))
        ctx.dims = (B, C, H, W, M, Cl, ny, nx, sy, sx)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        bda, coors, *rest = ctx.saved_tensors
        B, C, H, W, M, Cl, ny, nx, sy, sx = ctx.dims
        grad_out = _channels_last(grad_out.float(), "grad_out")
        grad_x = grad_feats = None
        with torch.cuda.device(grad_out.device):
            if ctx.needs_input_grad[0]:
                grad_x = torch.empty((B, H, W, C), dtype=torch.float32, device=grad_out.device).permute(0, 3, 1, 2)       # (assigned: every row written once)
                _lib.timed_call("bev_warp_backward", "mmt_bev_warp_affine_backward_assign", B, H, W, C, bda.data_ptr(), grad_out.data_ptr(), C + Cl,
                          grad_x.data_ptr(), C, _stream())
            if ctx.needs_input_grad[2]:
                grad_feats = torch.empty((M, Cl), dtype=torch.float32, device=grad_out.device)
                _lib.timed_call("scatter_backward", "mmt_pillar_scatter_nhwc_strided_backward", M, Cl, B, ny, nx, sy, sx,
                                grad_out.data_ptr() + 4 * C, C + Cl, coors.data_ptr(), rest[0].data_ptr() if rest else 0,
                                grad_feats.data_ptr(), _stream())
        return grad_x, None, grad_feats, None, None, None, None, None


def bev_warp_concat_pillars(x, bda_mat, feats, coors, table, ny, nx, max_voxels):
    """Camera map x [B, C, H, W] + pillar rows -> the fusion layer's input [B, C + Cl, H, W] (channels_last); (ny, nx) = the
    pillar grid, an integer multiple of (H, W)."""
    coors = coors.contiguous()
    if coors.dtype != torch.int32:
        coors = coors.int()
    return BevWarpConcatPillars.apply(x, bda_mat, feats, coors, table, int(ny), int(nx), int(max_voxels))


def bev_warp_affine(x, bda_mat):
    return BevWarpConcat.apply(x, bda_mat, None)


def bev_warp_concat(x, bda_mat, other):
    return BevWarpConcat.apply(x, bda_mat, other)
