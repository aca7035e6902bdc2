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
            _lib.call("mmt_bev_warp_affine", B, H, W, C, bda.data_ptr(), x.data_ptr(), C, out.data_ptr(), C + c_other, _stream())
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
        grad_x = torch.zeros((B, H, W, C), dtype=torch.float32, device=grad_out.device).permute(0, 3, 1, 2)
        with torch.cuda.device(grad_out.device):
            _lib.call("mmt_bev_warp_affine_backward", B, H, W, C, bda.data_ptr(), grad_out.data_ptr(), C + c_other,
                      grad_x.data_ptr(), C, _stream())
        grad_other = grad_out[:, C:] if c_other else None
        return grad_x, None, grad_other


def bev_warp_affine(x, bda_mat):
    return BevWarpConcat.apply(x, bda_mat, None)


def bev_warp_concat(x, bda_mat, other):
    return BevWarpConcat.apply(x, bda_mat, other)
