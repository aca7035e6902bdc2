"""Producers of voxel_pooling's operands as HIP ops (libmmt_hip.so, include/mmt_hip.h):

  quantize_geometry(xyz, voxel_coord, voxel_size)      lss_fpn.py:461-462
  frustum_geometry(frustum, combine, voxel_coord, ...) lss_fpn.py:328-361 + :461-462 fused
  lift_features(depth, context)                        lss_fpn.py:441-463 (autograd)

All take CUDA tensors, launch on the current stream and never fall back to torch ops.
"""
import os

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


class LiftFeaturesBF16(Function):
    """The same lift with the big tensor stored in bf16: depth / context fp32 in,
    feats bf16 [BN, D, fH, fW, C] = bf16(depth * context) out; backward reads a bf16 gradient, sums in fp32."""

    @staticmethod
    def forward(ctx, depth, context):
        _need_cuda(depth, "depth")
        _need_cuda(context, "context")
        BN, D, fH, fW = depth.shape
        C = context.shape[1]
        if context.shape != (BN, C, fH, fW):
            raise RuntimeError("context must be [BN, C, fH, fW] matching depth [BN, D, fH, fW]")
        feats = torch.empty((BN, D, fH, fW, C), dtype=torch.bfloat16, device=depth.device)
        with torch.cuda.device(depth.device):
            _lib.call("mmt_lift_features_bf16", BN, D, fH * fW, C, depth.data_ptr(), context.data_ptr(), feats.data_ptr(), _stream())
        ctx.save_for_backward(depth, context)
        return feats

    @staticmethod
    def backward(ctx, grad_feats):
        depth, context = ctx.saved_tensors
        BN, D, fH, fW = depth.shape
        C = context.shape[1]
        grad_feats = grad_feats.to(torch.bfloat16).contiguous()
        grad_depth = torch.empty_like(depth)
        grad_context = torch.empty_like(context)
        with torch.cuda.device(depth.device):
            _lib.call("mmt_lift_features_backward_bf16", BN, D, fH * fW, C, depth.data_ptr(), context.data_ptr(),
                      grad_feats.data_ptr(), grad_depth.data_ptr(), grad_context.data_ptr(), _stream())
        return grad_depth, grad_context


def lift_features(depth, context, storage_dtype=torch.float32):
    """depth [BN,D,fH,fW] x context [BN,C,fH,fW] (fp32) -> feats [BN,D,fH,fW,C] in `storage_dtype` (fp32 or bf16)."""
    if storage_dtype == torch.bfloat16:
        return LiftFeaturesBF16.apply(depth.contiguous(), context.contiguous())
    return LiftFeatures.apply(depth.contiguous(), context.contiguous())


def _pixel_rows(t, D):
    """(tensor, row stride in elements) with the D channel values of every pixel of t [BN, D, fH, fW] contiguous in memory and
    the pixels' rows equally spaced in (bn, h, w) order -- true for a channels_last tensor and for a channel slice of one;
    anything else is brought to channels_last first (one copy)."""
    BN, _, fH, fW = t.shape
    R = t.stride(3)
    if not (t.stride(1) == 1 and R >= D and t.stride(2) == fW * R and t.stride(0) == fH * fW * R):
        t = t.contiguous(memory_format=torch.channels_last)
        if t.stride(1) != 1:                  # (a size-1 dimension can leave contiguous() with arbitrary strides)
            t = t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        R = D
    return t, R


class DepthSoftmax(Function):
    """depth = logits.softmax(1) (lss_fpn.py:423) and the oracle-depth overwrite (:427-438) as one launch in the layout the fused
    lift-splat reads (mmt_depth_softmax_forward / _backward).  Returns (probs, depth_used): probs = the plain softmax, a
    channels_last [BN, D, fH, fW] fp32 tensor (what is_return_depth hands to the depth loss); depth_used = the tensor the
    lift multiplies the context with -- None when it would equal probs (no oracle, fp32)."""

    @staticmethod
    def forward(ctx, logits, oracle, used_bf16, plan_lookup=None):
        if not logits.is_cuda:
            raise RuntimeError("depth logits must be a CUDAtensor ")
        if logits.dtype not in (torch.float32, torch.bfloat16) or logits.dim() != 4:
            raise RuntimeError(f"depth logits must be a float32 / bfloat16 [B*N, D, fH, fW] tensor (found {logits.dtype})")
        BN, D, fH, fW = logits.shape
        logits, R = _pixel_rows(logits, D)
        o_ptr, o_R = 0, 0
        if oracle is not None:
            if tuple(oracle.shape) != (BN, D, fH, fW):
                raise RuntimeError("depth_softmax: the oracle must have the shape of the depth logits")
            oracle, o_R = _pixel_rows(oracle.detach().float(), D)
            o_ptr = oracle.data_ptr()
        probs = torch.empty((BN, fH, fW, D), dtype=torch.float32, device=logits.device)
        used = None
        if oracle is not None or used_bf16:
            used = torch.empty((BN, fH, fW, D), dtype=torch.bfloat16 if used_bf16 else torch.float32, device=logits.device)
        lt = _lib.DTYPE_BF16 if logits.dtype == torch.bfloat16 else _lib.DTYPE_F32
        ut = _lib.DTYPE_BF16 if used_bf16 else _lib.DTYPE_F32
        u_ptr = used.data_ptr() if used is not None else 0
        with torch.cuda.device(logits.device):
            if plan_lookup is not None:
                # the plan form's calibration lookup rides in this launch (mmt_depth_softmax_forward_plan_prepare)
                combine, (fu, fv, fd), voxel_num, vc, vs, plan_cache = plan_lookup
                B, N = combine.shape[:2]
                if (B * N, D, fH, fW) != (BN, fd.numel(), fv.numel(), fu.numel()):
                    raise RuntimeError("depth_softmax: the lookup that rides along is for another batch shape")
                if not softmax_rides_lookup(logits, R, oracle, o_R, D, used_bf16):
                    raise RuntimeError("depth_softmax: these rows cannot carry the lookup (softmax_rides_lookup)")
                ptr, nbytes = _plan_ptr(plan_cache)
                nx, ny, nz = [int(v) for v in voxel_num]
                _lib.timed_call("softmax", "mmt_depth_softmax_forward_plan_prepare", BN * fH * fW, D, logits.data_ptr(), R, lt, probs.data_ptr(),
                                o_ptr, o_R, u_ptr, ut, B, N, fH, fW, nx, ny, nz, combine.data_ptr(), fu.data_ptr(), fv.data_ptr(), fd.data_ptr(),
                                _lib.float3([float(v) for v in vc]), _lib.float3([float(v) for v in vs]), ptr, nbytes, _stream())
            else:
                _lib.timed_call("softmax", "mmt_depth_softmax_forward", BN * fH * fW, D, logits.data_ptr(), R, lt, probs.data_ptr(),
                                o_ptr, o_R, u_ptr, ut, _stream())
        ctx.save_for_backward(probs, *([oracle] if oracle is not None else []))
        ctx.meta = (BN, D, fH, fW, lt, ut, o_R, logits.dtype)
        return probs.permute(0, 3, 1, 2), (used.permute(0, 3, 1, 2) if used is not None else None)

    @staticmethod
    def backward(ctx, grad_probs, grad_used):
        probs, *rest = ctx.saved_tensors
        BN, D, fH, fW, lt, ut, o_R, ldtype = ctx.meta

        def rows(g, dtype):
            if g is None:
                return None
            if g.dtype != dtype:
                g = g.to(dtype)
            return g if g.is_contiguous(memory_format=torch.channels_last) and g.stride(1) == 1 else _pixel_rows(g, D)[0]

        gp = rows(grad_probs, torch.float32)
        gu = rows(grad_used, torch.bfloat16 if ut == _lib.DTYPE_BF16 else torch.float32)
        if gp is not None and gp.stride(3) != D:
            gp = gp.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        if gu is not None and gu.stride(3) != D:
            gu = gu.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        grad_logits = torch.empty((BN, fH, fW, D), dtype=ldtype, device=probs.device)
        with torch.cuda.device(probs.device):
            _lib.timed_call("softmax_backward", "mmt_depth_softmax_backward", BN * fH * fW, D, probs.data_ptr(),
                            gp.data_ptr() if gp is not None else 0, gu.data_ptr() if gu is not None else 0, ut,
                            rest[0].data_ptr() if rest else 0, o_R, grad_logits.data_ptr(), lt, _stream())
        return grad_logits.permute(0, 3, 1, 2), None, None, None


def softmax_rides_lookup(logits_rows, row_stride, oracle_rows, oracle_stride, D, used_bf16):
    """Whether mmt_depth_softmax_forward_plan_prepare takes these rows: 16-byte pieces (D % 4 == 0, every row on a 16-byte (fp32) /
    8-byte (bf16) boundary)."""
    eb = 2 if logits_rows.dtype == torch.bfloat16 else 4
    ok = D % 4 == 0 and D <= 512 and row_stride % 4 == 0 and logits_rows.data_ptr() % (4 * eb) == 0
    if oracle_rows is not None:
        ok = ok and oracle_stride % 4 == 0 and oracle_rows.data_ptr() % 16 == 0
    return ok


def depth_softmax(logits, oracle=None, used_dtype=torch.float32, plan_lookup=None):
    """logits [B*N, D, fH, fW] (fp32 or bf16; free when channels_last or a channel slice of a channels_last tensor)
    -> (probs, depth_used), both channels_last [B*N, D, fH, fW]: probs = softmax(logits, 1) in fp32; depth_used = probs with the
    rows of foreground pixels (torch.max(oracle, 1).values > 0) replaced by the oracle's, in `used_dtype` (fp32 or bf16).
    Without an oracle and with used_dtype fp32, depth_used IS probs.
    plan_lookup = (combine, axes, voxel_num, voxel_coord, voxel_size, plan_cache): plan_prepare's operands -- the lookup of the batch's
    calibrations then rides in the softmax's launch (no launch of its own in the steady state); when the rows cannot carry it
    (softmax_rides_lookup) it is made as a call of its own first."""
    if plan_lookup is not None:
        BN, D, fH, fW = logits.shape
        combine, axes = plan_lookup[0], plan_lookup[1]
        shape_ok = (combine.shape[0] * combine.shape[1], axes[2].numel(), axes[1].numel(), axes[0].numel()) == (BN, D, fH, fW)

        def pieces_ok(t):       # (what _pixel_rows has to copy comes out contiguous and aligned)
            R = t.stride(3)
            natural = t.stride(1) == 1 and R >= D and t.stride(2) == fW * R and t.stride(0) == fH * fW * R
            return (not natural) or (R % 4 == 0 and t.data_ptr() % (4 * t.element_size()) == 0)

        if not (shape_ok and D % 4 == 0 and D <= 512 and logits.dim() == 4 and pieces_ok(logits)
                and (oracle is None or oracle.dtype != torch.float32 or pieces_ok(oracle))):
            plan_prepare(*plan_lookup)
            plan_lookup = None
    probs, used = DepthSoftmax.apply(logits, oracle, used_dtype == torch.bfloat16, plan_lookup)
    return probs, (probs if used is None else used)


def _lss_flags(pixel_major, column_backward=False):
    return ((_lib.LSS_PIXEL_MAJOR if pixel_major else 0) | (_lib.LSS_COLUMN_BACKWARD if column_backward else 0)
            | (_lib.LSS_TILE_KERNELS if os.environ.get("MMT_LIFT_SPLAT_TILES", "0") == "1" else 0))


def column_mismatch_fraction(geom_xyz, voxel_num, pixel_major=False):
    """Share of the kept frustum points whose BEV cell differs from "their column's" cell (the smallest kept cell among the 16
    image rows of their block at the same depth bin) -- what the column backward kernel (MMT_LSS_COLUMN_BACKWARD) has to handle
    point by point.  0 for a level rig, ~0.04 for the reference's nuScenes calibration.  Returns a 0-dim tensor on the
    device of geom (no host sync here).  geom int32 [B,N,D,fH,fW,3], or [B,N,fH,fW,D,3] with pixel_major."""
    nx, ny, nz = [int(v) for v in (voxel_num.tolist() if isinstance(voxel_num, torch.Tensor) else voxel_num)]
    g = geom_xyz if pixel_major else geom_xyz.permute(0, 1, 3, 4, 2, 5)          # [B,N,fH,fW,D,3]
    x, y, z = g[..., 0].long(), g[..., 1].long(), g[..., 2].long()
    kept = (x >= 0) & (x < nx) & (y >= 0) & (y < ny) & (z >= 0) & (z < nz)
    big = torch.where(kept, y * nx + x, torch.full_like(x, 1 << 40))
    B, N, fH, fW, D = big.shape
    pad = (-fH) % 16
    if pad:
        big = torch.cat([big, big.new_full((B, N, pad, fW, D), 1 << 40)], 2)
        kept = torch.cat([kept, kept.new_zeros((B, N, pad, fW, D))], 2)
    blocks = big.view(B, N, -1, 16, fW, D)
    ref = blocks.amin(3, keepdim=True)
    mism = (kept.view(B, N, -1, 16, fW, D) & (blocks != ref)).sum()
    return mism.float() / kept.sum().clamp(min=1).float()


class LiftSplat(Function):
    """Fused lift + voxel_pooling (lss_fpn.py:441-464 in one pass): the
    [B, N, D, fH, fW, C] feature tensor is never materialised.  Additional entry point beside
    the drop-in ``voxel_pooling``; same result up to fp32 summation order.

    mmt_lss_splat_forward / _backward: ray walks (forward: a workgroup owns an image column and sums depth * context in
    registers while the BEV cell stays the same; backward: a lane group owns a pixel, no atomics).  MMT_LIFT_SPLAT_TILES=1
    selects the second-generation frustum-tile kernels, MMT_LIFT_SPLAT_V1=1 the first-generation pair (chunks of
    consecutive points; pixel-major backward on pos_memo) -- both kept for A/B runs, the latter also for fH > 512.
    column_backward=True: the backward of a (nearly) level rig on the matrix cores (MMT_LSS_COLUMN_BACKWARD, lift_splat_col.hip)."""

    @staticmethod
    def forward(ctx, geom_xyz, depth, context, voxel_num, pixel_major=False, column_backward=False):
        _need_cuda(geom_xyz, "geom_xyz", torch.int32)
        if pixel_major:
            B, N, fH, fW, D = geom_xyz.shape[:5]
        else:
            B, N, D, fH, fW = geom_xyz.shape[:5]
        BN, HW, C = B * N, fH * fW, context.shape[1]
        if tuple(depth.shape) != (BN, D, fH, fW) or tuple(context.shape) != (BN, C, fH, fW):
            raise RuntimeError("lift_splat: depth must be [B*N, D, fH, fW] and context [B*N, C, fH, fW]")
        if not (depth.is_cuda and context.is_cuda):
            raise RuntimeError("depth / context must be a CUDAtensor ")
        bf16 = depth.dtype == torch.bfloat16 and context.dtype == torch.bfloat16   # bf16 storage of both operands (row g1)
        sd = torch.bfloat16 if bf16 else torch.float32
        tiled = fH <= 512 and C <= 256 and C % 16 == 0 and os.environ.get("MMT_LIFT_SPLAT_V1", "0") != "1"
        if pixel_major and not tiled:
            raise RuntimeError("lift_splat: the pixel-major layout needs the frustum-tile kernels (C % 16 == 0, C <= 256, fH <= 512)")
        # point order of depth: [BN, D, HW] (reference) or [BN, HW, D] (pixel-major: free for a channels_last depth tensor)
        depth_c = (depth.to(sd).permute(0, 2, 3, 1) if pixel_major else depth.to(sd)).contiguous()
        ctx_nhwc = context.to(sd).permute(0, 2, 3, 1).contiguous()         # free for channels_last nets
        nx, ny, nz = [int(v) for v in (voxel_num.tolist() if isinstance(voxel_num, torch.Tensor) else voxel_num)]
        sfx = "_bf16" if bf16 else ""
        with torch.cuda.device(depth.device):
            if tiled:   # the backward redoes the kept test from geom: no pos_memo is written or kept
                out = torch.empty((B, ny, nx, C), dtype=torch.float32, device=depth.device)     # MMT_LSS_ZERO_OUTPUT fills it
                _lib.timed_call("lift_splat_forward", "mmt_lss_splat_forward" + sfx, B, N, D, fH, fW, C, nx, ny, nz,
                                geom_xyz.data_ptr(), depth_c.data_ptr(), ctx_nhwc.data_ptr(), out.data_ptr(), 0,
                                _lss_flags(pixel_major) | _lib.LSS_ZERO_OUTPUT, _stream())
                ctx.save_for_backward(geom_xyz, depth_c, ctx_nhwc)
            else:
                out = torch.zeros((B, ny, nx, C), dtype=torch.float32, device=depth.device)
                pos_memo = torch.empty((B, N * D * HW, 3), dtype=torch.int32, device=depth.device)
                _lib.timed_call("lift_splat_forward", "mmt_lift_splat_forward" + sfx, B, N, D, HW, C, nx, ny, nz,
                                geom_xyz.data_ptr(), depth_c.data_ptr(), ctx_nhwc.data_ptr(), out.data_ptr(), pos_memo.data_ptr(),
                                _lib.VP_WRITE_DROPPED, _stream())
                ctx.save_for_backward(pos_memo, depth_c, ctx_nhwc)
        ctx.dims = (B, N, D, fH, fW, C, nx, ny, nz)
        ctx.bf16, ctx.tiled, ctx.pixel_major, ctx.column_backward = bf16, tiled, bool(pixel_major), bool(column_backward)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_out):
        index, depth_c, ctx_nhwc = ctx.saved_tensors          # index: geom (tiled) or pos_memo (first generation)
        B, N, D, fH, fW, C, nx, ny, nz = ctx.dims
        if grad_out.stride(1) != 1 or grad_out.dtype != torch.float32:
            grad_out = grad_out.float().contiguous(memory_format=torch.channels_last)
        sb, sc, sy, sx = grad_out.stride()
        grad_depth = torch.empty_like(depth_c)
        sfx = "_bf16" if ctx.bf16 else ""
        with torch.cuda.device(depth_c.device):
            if ctx.tiled:
                grad_ctx = torch.empty(ctx_nhwc.shape, dtype=torch.float32, device=depth_c.device)    # every element is written
                _lib.timed_call("lift_splat_backward", "mmt_lss_splat_backward" + sfx, B, N, D, fH, fW, C, nx, ny, nz,
                                index.data_ptr(), depth_c.data_ptr(), ctx_nhwc.data_ptr(), grad_out.data_ptr(), sb, sc, sy, sx,
                                grad_depth.data_ptr(), grad_ctx.data_ptr(), _lss_flags(ctx.pixel_major, ctx.column_backward), _stream())
                if ctx.bf16:
                    grad_ctx = grad_ctx.to(torch.bfloat16)
            else:
                grad_ctx = torch.empty_like(ctx_nhwc)
                _lib.timed_call("lift_splat_backward", "mmt_lift_splat_backward" + sfx, B, N, D, fH * fW, C, nx, ny,
                                index.data_ptr(), depth_c.data_ptr(), ctx_nhwc.data_ptr(), grad_out.data_ptr(), sb, sc, sy, sx,
                                grad_depth.data_ptr(), grad_ctx.data_ptr(), _stream())
        if ctx.pixel_major:      # [BN, fH, fW, D] storage = a channels_last [BN, D, fH, fW] gradient
            grad_depth = grad_depth.permute(0, 3, 1, 2)
        return None, grad_depth, grad_ctx.permute(0, 3, 1, 2), None, None, None


def lift_splat(geom_xyz, depth, context, voxel_num, pixel_major=False, column_backward=False):
    """geom int32 [B,N,D,fH,fW,3], depth [B*N,D,fH,fW], context [B*N,C,fH,fW] -> BEV fp32 [B,C,ny,nx].
    depth AND context in bf16 select the bf16-storage kernels (fp32 products and sums, bf16 gradients back).
    pixel_major=True: geom is [B,N,fH,fW,D,3] (frustum_geometry of the frustum permuted to [fH,fW,D,4]) and the kernels read
    depth in [B*N,fH,fW,D] memory order -- what a channels_last depth tensor already is -- and return its gradient in that
    order: whole 64- / 192-byte runs per pixel and tile instead of 8- / 24-byte pieces."""
    return LiftSplat.apply(geom_xyz.contiguous(), depth, context, voxel_num, bool(pixel_major), bool(column_backward))


def frustum_axes(frustum):
    """The three axes of a create_frustum-style frustum [D, fH, fW, 4] (lss_fpn.py:308-326): (u [fW], v [fH], d [D]) with
    frustum[k, h, w] == (u[w], v[h], d[k], 1) for every point -- what the camera form of the fused lift-splat takes instead
    of a geom tensor.  Returns None when the frustum is not that outer product (then only the geom form applies)."""
    u, v, d = frustum[0, 0, :, 0], frustum[0, :, 0, 1], frustum[:, 0, 0, 2]
    D, fH, fW, _ = frustum.shape
    same = (torch.equal(frustum[..., 0], u.view(1, 1, fW).expand(D, fH, fW)) and
            torch.equal(frustum[..., 1], v.view(1, fH, 1).expand(D, fH, fW)) and
            torch.equal(frustum[..., 2], d.view(D, 1, 1).expand(D, fH, fW)) and
            bool((frustum[..., 3] == 1).all()))
    return (u.contiguous(), v.contiguous(), d.contiguous()) if same else None


def camera_form_supported(B, N, D, fH, fW, C):
    """True when the camera-form kernels take this shape in both directions (mmt_lss_camera_form_supported)."""
    return bool(_lib.lib().mmt_lss_camera_form_supported(int(B), int(N), int(D), int(fH), int(fW), int(C)))


def exclusive_cache_used(B, N, D, fH, fW, C):
    """True when the camera-form forward of this shape takes an exclusive-cell cache (mmt_lss_exclusive_cache_used)."""
    return bool(_lib.lib().mmt_lss_exclusive_cache_used(int(B), int(N), int(D), int(fH), int(fW), int(C)))


def last_kernel_family(backward=False, detail=False):
    """Kernel family the process's last fused lift-splat forward / backward call launched:
    "ray" | "tile" | "column" | "none", + "+camera" for the camera form (mmt_lss_last_kernel_family).
    detail=True appends "+register" / "+block" (the forward's register / block walk) and "+exclusive" (an exclusive-cell
    cache was used)."""
    v = _lib.lib().mmt_lss_last_kernel_family(1 if backward else 0)
    name = _lib.LSS_FAMILY.get(v & 0xF, "?") + ("+camera" if v & 0x10 else "")
    if detail:
        name += (("+register" if v & _lib.LSS_FAMILY_REGISTER else "") + ("+block" if v & _lib.LSS_FAMILY_BLOCK else "") +
                 ("+exclusive" if v & _lib.LSS_FAMILY_EXCLUSIVE else ""))
    return name


class LiftSplatCamera(Function):
    """Fused get_geometry + quantise + lift + voxel_pooling (lss_fpn.py:328-361, :461-462, :441-464): the camera form
    of the fused op (mmt_lss_splat_forward_cam / _backward_cam).  The kernels compute every point's voxel index from
    `combine` [B, N, 4, 4] (= sensor2ego @ inverse(intrin)) and the frustum axes with the arithmetic of
    mmt_frustum_geometry, so no geom tensor exists in either direction.  depth / grad_depth in pixel-major
    ([B*N, fH, fW, D] = channels_last) order; the BEV map is zero-filled by the forward call itself."""

    @staticmethod
    def forward(ctx, combine, axes, grid, depth, context, voxel_num, column_backward, column_stats, summary, summary_cached,
                exclusive_cache=None):
        fu, fv, fd = axes
        vc, vs = grid
        B, N = combine.shape[:2]
        D, fH, fW = fd.numel(), fv.numel(), fu.numel()
        BN, C = B * N, context.shape[1]
        _need_cuda(combine, "combine")
        for t, name in ((fu, "frustum_u"), (fv, "frustum_v"), (fd, "frustum_d")):
            _need_cuda(t, name)
        if tuple(combine.shape[2:]) != (4, 4):
            raise RuntimeError("lift_splat_camera: combine must be [B, N, 4, 4]")
        if tuple(depth.shape) != (BN, D, fH, fW) or tuple(context.shape) != (BN, C, fH, fW):
            raise RuntimeError("lift_splat_camera: depth must be [B*N, D, fH, fW] and context [B*N, C, fH, fW]")
        if not (depth.is_cuda and context.is_cuda):
            raise RuntimeError("depth / context must be a CUDAtensor ")
        bf16 = depth.dtype == torch.bfloat16 and context.dtype == torch.bfloat16
        sd = torch.bfloat16 if bf16 else torch.float32
        depth_c = depth.to(sd).permute(0, 2, 3, 1).contiguous()            # free for channels_last nets
        ctx_nhwc = context.to(sd).permute(0, 2, 3, 1).contiguous()
        nx, ny, nz = [int(v) for v in (voxel_num.tolist() if isinstance(voxel_num, torch.Tensor) else voxel_num)]
        out = torch.empty((B, ny, nx, C), dtype=torch.float32, device=depth.device)
        vc_c, vs_c = _lib.float3(vc), _lib.float3(vs)
        if summary is None:        # written by this forward, read by its backward
            summary, summary_cached = new_column_summary(B, N, D, fH, fW, depth.device), False
        elif tuple(summary.shape) != (BN, (fH + 15) // 16, fW, D, 2) or summary.dtype != torch.int32 or not summary.is_contiguous():
            raise RuntimeError("lift_splat_camera: column summary must be a contiguous int32 [B*N, ceil(fH/16), fW, D, 2] tensor")
        if exclusive_cache is not None and (exclusive_cache.dtype != torch.int32 or not exclusive_cache.is_contiguous() or
                                            exclusive_cache.device != depth.device):
            raise RuntimeError("lift_splat_camera: the exclusive-cell cache must be a contiguous int32 tensor on the inputs' device")
        with torch.cuda.device(depth.device):
            _lib.timed_call("lift_splat_forward", "mmt_lss_splat_forward_cam" + ("_bf16" if bf16 else ""), B, N, D, fH, fW, C,
                            nx, ny, nz, combine.data_ptr(), fu.data_ptr(), fv.data_ptr(), fd.data_ptr(), vc_c, vs_c,
                            depth_c.data_ptr(), ctx_nhwc.data_ptr(), out.data_ptr(), 0, summary.data_ptr(),
                            exclusive_cache.data_ptr() if exclusive_cache is not None else 0,
                            exclusive_cache.numel() * 4 if exclusive_cache is not None else 0,
                            _lib.LSS_PIXEL_MAJOR | _lib.LSS_ZERO_OUTPUT | (_lib.LSS_SUMMARY_CACHED if summary_cached else 0), _stream())
        ctx.save_for_backward(combine, fu, fv, fd, depth_c, ctx_nhwc, summary)
        ctx.dims = (B, N, D, fH, fW, C, nx, ny, nz)
        ctx.grid = (tuple(float(v) for v in vc), tuple(float(v) for v in vs))
        ctx.bf16, ctx.column_backward, ctx.column_stats = bf16, bool(column_backward), column_stats
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_out):
        combine, fu, fv, fd, depth_c, ctx_nhwc, summary = ctx.saved_tensors
        B, N, D, fH, fW, C, nx, ny, nz = ctx.dims
        if grad_out.stride(1) != 1 or grad_out.dtype != torch.float32:
            grad_out = grad_out.float().contiguous(memory_format=torch.channels_last)
        sb, sc, sy, sx = grad_out.stride()
        grad_depth = torch.empty_like(depth_c)
        grad_ctx = torch.empty(ctx_nhwc.shape, dtype=torch.float32, device=depth_c.device)    # every element is written
        stats = ctx.column_stats
        flags = _lib.LSS_PIXEL_MAJOR | (_lib.LSS_COLUMN_BACKWARD if ctx.column_backward else 0)
        with torch.cuda.device(depth_c.device):
            _lib.timed_call("lift_splat_backward", "mmt_lss_splat_backward_cam" + ("_bf16" if ctx.bf16 else ""), B, N, D, fH, fW, C,
                            nx, ny, nz, combine.data_ptr(), fu.data_ptr(), fv.data_ptr(), fd.data_ptr(),
                            _lib.float3(ctx.grid[0]), _lib.float3(ctx.grid[1]), depth_c.data_ptr(), ctx_nhwc.data_ptr(),
                            grad_out.data_ptr(), sb, sc, sy, sx, grad_depth.data_ptr(), grad_ctx.data_ptr(), summary.data_ptr(),
                            stats.data_ptr() if stats is not None else 0, flags, _stream())
        if ctx.bf16:
            grad_ctx = grad_ctx.to(torch.bfloat16)
        return None, None, None, grad_depth.permute(0, 3, 1, 2), grad_ctx.permute(0, 3, 1, 2), None, None, None, None, None, None


def plan_form_supported(B, N, D, fH, fW, C, voxel_num):
    """True when the plan form of the fused lift-splat forward takes this shape (mmt_lss_plan_supported)."""
    nx, ny, nz = [int(v) for v in (voxel_num.tolist() if isinstance(voxel_num, torch.Tensor) else voxel_num)]
    return bool(_lib.lib().mmt_lss_plan_supported(int(B), int(N), int(D), int(fH), int(fW), int(C), nx, ny, nz))


def new_plan_cache(num_cams, D, fH, fW, voxel_num, device, slots=16):
    """Plan cache of the plan-form forward (include/mmt_hip.h `plan_cache`): `slots` calibrations (a sample's camera matrices)
    with their learnt plans; needs slots >= the batch size of the calls that use it.  A uint8 CUDA tensor owned by the
    caller for good; the library owns its contents.  ~3.9 MB per slot at BASELINE configs[3], ~13 MB at configs[4]."""
    nx, ny = [int(v) for v in (voxel_num.tolist() if isinstance(voxel_num, torch.Tensor) else voxel_num)][:2]
    nbytes = int(_lib.lib().mmt_lss_plan_cache_bytes(int(num_cams), int(D), int(fH), int(fW), nx, ny, int(slots)))
    if nbytes <= 0:
        raise RuntimeError(f"new_plan_cache: bad arguments (N={num_cams}, D={D}, fH={fH}, fW={fW}, nx={nx}, ny={ny}, slots={slots})")
    # (uninitialised is fine: the probe recognises a table that is not its own -- but zeros make the counters readable at once)
    return torch.zeros(nbytes + 256, dtype=torch.uint8, device=device)


def _plan_ptr(cache):
    """(256-byte aligned pointer into the cache tensor, bytes from there)"""
    p = cache.data_ptr()
    a = (p + 255) & ~255
    return a, cache.numel() - (a - p)


def plan_cache_counters(cache):
    """dict(hit, learnt, brute, resets, calls, slots, stale) of a plan cache (synchronises the current stream).  stale: forwards told
    MMT_LSS_PLAN_PREPARED that found a verdict no lookup had left (they wrote nothing for that sample)."""
    import ctypes
    out = (ctypes.c_int64 * 8)()
    ptr, nbytes = _plan_ptr(cache)
    with torch.cuda.device(cache.device):
        _lib.call("mmt_lss_plan_cache_counters", ptr, nbytes, out, _stream())
    return dict(hit=int(out[0]), learnt=int(out[1]), brute=int(out[2]), resets=int(out[3]), calls=int(out[4]), slots=int(out[5]), stale=int(out[6]))


def plan_prepare(combine, axes, voxel_num, voxel_coord, voxel_size, plan_cache):
    """mmt_lss_plan_prepare: look the batch's calibrations up in `plan_cache` and learn the unknown ones.  Depends on the
    matrices only -- call it as early in the step as they exist; the forward is then told MMT_LSS_PLAN_PREPARED."""
    fu, fv, fd = axes
    B, N = combine.shape[:2]
    D, fH, fW = fd.numel(), fv.numel(), fu.numel()
    _need_cuda(combine, "combine")
    nx, ny, nz = [int(v) for v in (voxel_num.tolist() if isinstance(voxel_num, torch.Tensor) else voxel_num)]
    vc = [float(v) for v in (voxel_coord.tolist() if isinstance(voxel_coord, torch.Tensor) else voxel_coord)]
    vs = [float(v) for v in (voxel_size.tolist() if isinstance(voxel_size, torch.Tensor) else voxel_size)]
    ptr, nbytes = _plan_ptr(plan_cache)
    with torch.cuda.device(combine.device):
        _lib.timed_call("lift_splat_plan_prepare", "mmt_lss_plan_prepare", B, N, D, fH, fW, nx, ny, nz, combine.data_ptr(), fu.data_ptr(), fv.data_ptr(),
                        fd.data_ptr(), _lib.float3(vc), _lib.float3(vs), ptr, nbytes, _stream())


class LiftSplatPlan(Function):
    """The plan form of the fused op (mmt_lss_splat_forward_plan; lss_fpn.py:328-361, :461-462, :441-464): the
    output-stationary forward -- no zero fill, no atomics, bit-identical from call to call -- on the plan the library has
    learnt for the batch's calibrations in `plan_cache`.  Backward: the camera form's kernels on the column summary this
    forward hands them (mmt_lss_splat_backward_cam)."""

    @staticmethod
    def forward(ctx, combine, axes, grid, depth, context, voxel_num, column_backward, column_stats, plan_cache, prepared, brute):
        fu, fv, fd = axes
        vc, vs = grid
        B, N = combine.shape[:2]
        D, fH, fW = fd.numel(), fv.numel(), fu.numel()
        BN, C = B * N, context.shape[1]
        _need_cuda(combine, "combine")
        for t, name in ((fu, "frustum_u"), (fv, "frustum_v"), (fd, "frustum_d")):
            _need_cuda(t, name)
        if tuple(combine.shape[2:]) != (4, 4):
            raise RuntimeError("lift_splat_plan: combine must be [B, N, 4, 4]")
        if tuple(depth.shape) != (BN, D, fH, fW) or tuple(context.shape) != (BN, C, fH, fW):
            raise RuntimeError("lift_splat_plan: depth must be [B*N, D, fH, fW] and context [B*N, C, fH, fW]")
        if not (depth.is_cuda and context.is_cuda):
            raise RuntimeError("depth / context must be a CUDAtensor ")
        if plan_cache is None or plan_cache.dtype != torch.uint8 or plan_cache.device != depth.device:
            raise RuntimeError("lift_splat_plan: plan_cache must be the uint8 tensor of new_plan_cache on the inputs' device")
        bf16 = depth.dtype == torch.bfloat16 and context.dtype == torch.bfloat16
        sd = torch.bfloat16 if bf16 else torch.float32
        depth_c = depth.to(sd).permute(0, 2, 3, 1).contiguous()            # free for channels_last nets
        ctx_nhwc = context.to(sd).permute(0, 2, 3, 1).contiguous()
        nx, ny, nz = [int(v) for v in (voxel_num.tolist() if isinstance(voxel_num, torch.Tensor) else voxel_num)]
        out = torch.empty((B, ny, nx, C), dtype=torch.float32, device=depth.device)     # every element is written
        need_bwd = any(ctx.needs_input_grad[3:5])
        summary = new_column_summary(B, N, D, fH, fW, depth.device) if need_bwd else None
        ptr, nbytes = _plan_ptr(plan_cache)
        flags = _lib.LSS_PIXEL_MAJOR | (_lib.LSS_PLAN_PREPARED if prepared else 0) | (_lib.LSS_PLAN_BRUTE if brute else 0)
        with torch.cuda.device(depth.device):
            _lib.timed_call("lift_splat_forward", "mmt_lss_splat_forward_plan" + ("_bf16" if bf16 else ""), B, N, D, fH, fW, C, nx, ny, nz,
                            combine.data_ptr(), fu.data_ptr(), fv.data_ptr(), fd.data_ptr(), _lib.float3(vc), _lib.float3(vs),
                            depth_c.data_ptr(), ctx_nhwc.data_ptr(), out.data_ptr(), summary.data_ptr() if summary is not None else 0,
                            ptr, nbytes, flags, _stream())
        if need_bwd:
            ctx.save_for_backward(combine, fu, fv, fd, depth_c, ctx_nhwc, summary)
        ctx.dims = (B, N, D, fH, fW, C, nx, ny, nz)
        ctx.grid = (tuple(float(v) for v in vc), tuple(float(v) for v in vs))
        ctx.bf16, ctx.column_backward, ctx.column_stats = bf16, bool(column_backward), column_stats
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_out):
        grads = LiftSplatCamera.backward(ctx, grad_out)          # same saved tensors, same attributes
        return grads[0], grads[1], grads[2], grads[3], grads[4], None, None, None, None, None, None


def lift_splat_plan(combine, axes, depth, context, voxel_num, voxel_coord, voxel_size, plan_cache, prepared=False, column_backward=False,
                    column_stats=None, brute=False):
    """lift_splat_camera's operands + the plan cache of new_plan_cache -> BEV fp32 [B,C,ny,nx] (channels_last memory), through the
    output-stationary plan-form forward.  prepared=True: plan_prepare already ran for this batch on this stream."""
    vc = [float(v) for v in (voxel_coord.tolist() if isinstance(voxel_coord, torch.Tensor) else voxel_coord)]
    vs = [float(v) for v in (voxel_size.tolist() if isinstance(voxel_size, torch.Tensor) else voxel_size)]
    return LiftSplatPlan.apply(combine.contiguous(), tuple(axes), (vc, vs), depth, context, voxel_num, bool(column_backward), column_stats,
                               plan_cache, bool(prepared), bool(brute))


def new_column_summary(B, N, D, fH, fW, device):
    """Uninitialised column summary of the camera form (include/mmt_hip.h `column_summary`): int32 [B*N, ceil(fH/16), fW, D, 2],
    8 bytes per (16-row block of a column, depth bin) -- written by a forward, read by its backward and by later forwards
    of the same calibration."""
    return torch.empty((B * N, (fH + 15) // 16, fW, D, 2), dtype=torch.int32, device=device)


def new_exclusive_cache(num_cams, voxel_num, device, slots=1024):
    """Zero-initialised exclusive-cell cache of the camera-form forward (include/mmt_hip.h `exclusive_cache`): per
    calibration (a sample's `num_cams` matrices) the BEV cells that a single run of the forward reaches; the library
    learns them on the device and stores such runs instead of adding them atomically.  ~4 * nx * ny bytes per slot
    (64 KiB on a 128 x 128 map).  One cache per module and stream; hand it to every lift_splat_camera call."""
    nx, ny = [int(v) for v in (voxel_num.tolist() if isinstance(voxel_num, torch.Tensor) else voxel_num)][:2]
    nbytes = int(_lib.lib().mmt_lss_exclusive_cache_bytes(int(num_cams), nx, ny, int(slots)))
    if nbytes <= 0:
        raise RuntimeError(f"new_exclusive_cache: bad arguments (N={num_cams}, nx={nx}, ny={ny}, slots={slots})")
    return torch.zeros(nbytes // 4, dtype=torch.int32, device=device)


def lift_splat_camera(combine, axes, depth, context, voxel_num, voxel_coord, voxel_size, column_backward=False, column_stats=None,
                      summary=None, summary_cached=False, exclusive_cache=None):
    """combine fp32 [B,N,4,4], axes = frustum_axes(frustum), depth [B*N,D,fH,fW], context [B*N,C,fH,fW] (fp32, or both bf16)
    -> BEV fp32 [B,C,ny,nx] (channels_last memory).  voxel_coord / voxel_size: 3 host floats each (the module buffers
    lss_fpn.py:278-285).  column_stats: optional int64 [2 * _lib.LSS_STATS_SLOTS] CUDA tensor the column backward
    accumulates (mismatching, kept) point-count pairs into (sum over the pairs = the totals).
    summary: None (a fresh column summary is written by the forward and read by the backward), or a tensor from
    new_column_summary kept by the caller across steps: summary_cached=False writes it, summary_cached=True declares that an
    earlier call wrote it for the SAME combine / axes / grid (unchanged calibration) and the forward reads it instead of
    computing the geometry.  exclusive_cache: None or the tensor of new_exclusive_cache (kept by the caller for good)."""
    vc = [float(v) for v in (voxel_coord.tolist() if isinstance(voxel_coord, torch.Tensor) else voxel_coord)]
    vs = [float(v) for v in (voxel_size.tolist() if isinstance(voxel_size, torch.Tensor) else voxel_size)]
    return LiftSplatCamera.apply(combine.contiguous(), tuple(axes), (vc, vs), depth, context, voxel_num, bool(column_backward),
                                 column_stats, summary, bool(summary_cached), exclusive_cache)
