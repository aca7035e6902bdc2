"""Producers of voxel_pooling's operands as HIP ops (libmmt_hip.so, include/mmt_hip.h):

  quantize_geometry(xyz, voxel_coord, voxel_size)      lss_fpn.py:461-462
  frustum_geometry(frustum, combine, voxel_coord, ...) lss_fpn.py:328-361 + :461-462 fused
  lift_features(depth, context)                        lss_fpn.py:441-463 (autograd)

All take CUDA tensors, launch on the current stream and never fall back to torch ops.
"""
import torch
from torch.autograd import Function

from .. import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(t, name, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDAtensor ")
    if t.dtype != dtype:
        raise RuntimeError(f"expected scalar type {dtype} but found {t.dtype} for {name}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous ")


def quantize_geometry(xyz, voxel_coord, voxel_size):
    """((xyz - (voxel_coord - voxel_size / 2)) / voxel_size).int() -> int32, same shape.

    voxel_coord / voxel_size: 3 host floats (or CPU tensors / lists)."""
    _need_cuda(xyz, "xyz")
    if xyz.shape[-1] != 3:
        raise RuntimeError("xyz must have a trailing dimension of 3")
    out = torch.empty(xyz.shape, dtype=torch.int32, device=xyz.device)
    vc = _lib.float3(torch.as_tensor(voxel_coord).tolist())
    vs = _lib.float3(torch.as_tensor(voxel_size).tolist())
    with torch.cuda.device(xyz.device):
        _lib.call("mmt_quantize_geometry", xyz.numel() // 3, xyz.data_ptr(), vc, vs, out.data_ptr(), _stream())
    return out


def frustum_geometry(frustum, combine, voxel_coord, voxel_size, return_xyz=False):
    """frustum [D,fH,fW,4] fp32, combine [B,N,4,4] fp32 (= sensor2ego @ inverse(intrin))
    -> geom int32 [B,N,D,fH,fW,3] (and the fp32 ego points if return_xyz)."""
    _need_cuda(frustum, "frustum")
    _need_cuda(combine, "combine")
    if frustum.dim() != 4 or frustum.shape[-1] != 4 or combine.shape[-2:] != (4, 4):
        raise RuntimeError("frustum must be [D,fH,fW,4] and combine [...,4,4]")
    D, fH, fW, _ = frustum.shape
    lead = tuple(combine.shape[:-2])
    BN = 1
    for v in lead:
        BN *= v
    geom = torch.empty(lead + (D, fH, fW, 3), dtype=torch.int32, device=frustum.device)
    xyz = torch.empty(lead + (D, fH, fW, 3), dtype=torch.float32, device=frustum.device) if return_xyz else None
    vc = _lib.float3(torch.as_tensor(voxel_coord).tolist())
    vs = _lib.float3(torch.as_tensor(voxel_size).tolist())
    with torch.cuda.device(frustum.device):
        _lib.call("mmt_frustum_geometry", BN, D * fH * fW, frustum.data_ptr(), combine.data_ptr(), vc, vs,
                  geom.data_ptr(), xyz.data_ptr() if return_xyz else 0, _stream())
    return (geom, xyz) if return_xyz else geom


class LiftFeatures(Function):
    """feats[bn,d,h,w,c] = depth[bn,d,h,w] * context[bn,c,h,w], written channels-last."""

    @staticmethod
    def forward(ctx, depth, context):
        _need_cuda(depth, "depth")
        _need_cuda(context, "context")
        BN, D, fH, fW = depth.shape
        C = context.shape[1]
        if context.shape != (BN, C, fH, fW):
            raise RuntimeError("context must be [BN, C, fH, fW] matching depth [BN, D, fH, fW]")
        feats = torch.empty((BN, D, fH, fW, C), dtype=torch.float32, device=depth.device)
        with torch.cuda.device(depth.device):
            _lib.call("mmt_lift_features", BN, D, fH * fW, C, depth.data_ptr(), context.data_ptr(),
                      feats.data_ptr(), _stream())
        ctx.save_for_backward(depth, context)
        return feats

    @staticmethod
    def backward(ctx, grad_feats):
        depth, context = ctx.saved_tensors
        BN, D, fH, fW = depth.shape
        C = context.shape[1]
        grad_feats = grad_feats.contiguous()
        grad_depth = torch.empty_like(depth)
        grad_context = torch.empty_like(context)
        with torch.cuda.device(depth.device):
            _lib.call("mmt_lift_features_backward", BN, D, fH * fW, C, depth.data_ptr(),
                      context.data_ptr(), grad_feats.data_ptr(), grad_depth.data_ptr(),
                      grad_context.data_ptr(), _stream())
        return grad_depth, grad_context


def lift_features(depth, context):
    return LiftFeatures.apply(depth.contiguous(), context.contiguous())
