"""Deformable 3x3 convolution (mmcv 'DCN' / DeformConv2dPack, lss_fpn.py:189-197).

Default: the implicit-GEMM kernels of csrc/deform_conv_mfma.hip (mmt_dcn_forward / mmt_dcn_backward: bilinear taps sampled
into LDS tiles, exact-fp32 MFMA, no column buffer).  Shapes they do not take (C/groups not a multiple of 64, O/groups not
64 or 128) go through the HIP im2col / col2im kernels + torch.mm for the grouped GEMMs (`_DeformConv3x3Columns`)."""
import os

import torch
from torch.autograd import Function

from .. import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


_WS = {}


def _workspace(nbytes, device):
    """One workspace per device, grown on demand (nothing is retained in it between calls; calls on one stream are ordered)."""
    key = (device.type, device.index)
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _WS[key] = t
    return t


class _DeformConv3x3Mfma(Function):
    """x [B,C,H,W], offset [B,18,H,W], weight [O, C/groups, 3, 3] -> [B,O,H,W] (channels_last view)."""

    @staticmethod
    def forward(ctx, x, offset, weight, groups):
        x_nhwc = x.float().permute(0, 2, 3, 1).contiguous()        # free for channels_last inputs
        off_nhwc = offset.float().permute(0, 2, 3, 1).contiguous()
        w = weight.float().contiguous()
        B, H, W, C = x_nhwc.shape
        O = w.shape[0]
        if off_nhwc.shape != (B, H, W, 18) or tuple(w.shape) != (O, C // groups, 3, 3):
            raise RuntimeError("deform_conv3x3: expected offset [B,18,H,W] and weight [O, C/groups, 3, 3]")
        out = torch.empty((B, H, W, O), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            ws = _workspace(_lib.lib().mmt_dcn_mfma_workspace_bytes(B, H, W, C, O, groups), x.device)
            _lib.timed_call("dcn_forward", "mmt_dcn_forward", B, H, W, C, O, groups, x_nhwc.data_ptr(), off_nhwc.data_ptr(), w.data_ptr(), out.data_ptr(),
                      ws.data_ptr(), ws.numel(), int(os.environ.get("MMT_DCN_FWD_CONFIG", "0")), _stream())
        ctx.save_for_backward(x_nhwc, off_nhwc, w)
        ctx.dims = (B, H, W, C, O, groups)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_out):
        x_nhwc, off_nhwc, w = ctx.saved_tensors
        B, H, W, C, O, groups = ctx.dims
        if _lib.lib().mmt_dcn_backward_form(B, H, W, C, O, groups) != 2 and not os.environ.get("MMT_DCN_BACKWARD_GENERAL"):
            # images past the gather form's 768 pixels: the library's general data-gradient kernel sums its LDS windows with float
            # atomics (3-4x slower than the column form).  Rebuild the columns here -- they were never stored by the forward --
            # and take the column backward.
            Cg, Og, N = C // groups, O // groups, B * H * W
            col = torch.empty((groups, N, 9 * Cg), dtype=torch.float32, device=x_nhwc.device)
            with torch.cuda.device(x_nhwc.device):
                _lib.call("mmt_dcn_im2col", B, H, W, C, groups, x_nhwc.data_ptr(), off_nhwc.data_ptr(), col.data_ptr(), _stream())
            wmat = w.reshape(groups, Og, Cg, 9).permute(0, 3, 2, 1).reshape(groups, 9 * Cg, Og)
            grad_x, grad_off, grad_wmat = _columns_backward(x_nhwc, off_nhwc, col, wmat, grad_out, ctx.dims)
            grad_w = grad_wmat.reshape(groups, 9, Cg, Og).permute(0, 3, 2, 1).reshape(O, Cg, 3, 3)
            return grad_x.permute(0, 3, 1, 2), grad_off.permute(0, 3, 1, 2), grad_w, None
        go = grad_out.float().permute(0, 2, 3, 1).contiguous()      # free for a channels_last gradient
        grad_x = torch.empty_like(x_nhwc)
        grad_off = torch.empty_like(off_nhwc)
        grad_w = torch.empty_like(w)
        with torch.cuda.device(x_nhwc.device):
            ws = _workspace(_lib.lib().mmt_dcn_mfma_workspace_bytes(B, H, W, C, O, groups), x_nhwc.device)
            _lib.timed_call("dcn_backward", "mmt_dcn_backward", B, H, W, C, O, groups, x_nhwc.data_ptr(), off_nhwc.data_ptr(), w.data_ptr(), go.data_ptr(),
                      grad_x.data_ptr(), grad_off.data_ptr(), grad_w.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
        return grad_x.permute(0, 3, 1, 2), grad_off.permute(0, 3, 1, 2), grad_w, None


class _DeformConv3x3Columns(Function):
    @staticmethod
    def forward(ctx, x, offset, weight, groups):
        if not x.is_cuda:
            raise RuntimeError("x must be a CUDAtensor ")
        x_nhwc = x.float().permute(0, 2, 3, 1).contiguous()        # free for channels_last inputs
        off_nhwc = offset.float().permute(0, 2, 3, 1).contiguous()
        B, H, W, C = x_nhwc.shape
        O = weight.shape[0]
        Cg, Og, N = C // groups, O // groups, B * H * W
        if off_nhwc.shape != (B, H, W, 18) or tuple(weight.shape) != (O, Cg, 3, 3):
            raise RuntimeError("deform_conv3x3: expected offset [B,18,H,W] and weight [O, C/groups, 3, 3]")
        col = torch.empty((groups, N, 9 * Cg), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.call("mmt_dcn_im2col", B, H, W, C, groups, x_nhwc.data_ptr(), off_nhwc.data_ptr(),
                      col.data_ptr(), _stream())
        wmat = weight.float().reshape(groups, Og, Cg, 9).permute(0, 3, 2, 1).reshape(groups, 9 * Cg, Og)
        # one GEMM per weight group, each writing its Og columns of the channels-last result in place (leading dimension O):
        # a batched GEMM leaves [g, N, Og] and the transposition to [N, g * Og] was a 35 MB copy per call (and one more for the
        # gradient in backward)
        out2d = torch.empty((N, O), dtype=torch.float32, device=x.device)
        for g in range(groups):
            torch.mm(col[g], wmat[g], out=out2d[:, g * Og:(g + 1) * Og])
        out_nhwc = out2d.view(B, H, W, O)
        ctx.save_for_backward(x_nhwc, off_nhwc, col, wmat)
        ctx.dims = (B, H, W, C, O, groups)
        return out_nhwc.permute(0, 3, 1, 2)                         # channels_last view

    @staticmethod
    def backward(ctx, grad_out):
        x_nhwc, off_nhwc, col, wmat = ctx.saved_tensors
        grad_x, grad_off, grad_wmat = _columns_backward(x_nhwc, off_nhwc, col, wmat, grad_out, ctx.dims)
        B, H, W, C, O, groups = ctx.dims
        Cg, Og = C // groups, O // groups
        grad_weight = grad_wmat.reshape(groups, 9, Cg, Og).permute(0, 3, 2, 1).reshape(O, Cg, 3, 3)
        return grad_x.permute(0, 3, 1, 2), grad_off.permute(0, 3, 1, 2), grad_weight, None


def _columns_backward(x_nhwc, off_nhwc, col, wmat, grad_out, dims):
    """The three gradients from the column buffer: two vendor GEMMs + the sorted (or atomic) col2im.  Returns channels-last
    grad_x, grad_offset and the weight gradient as [groups, 9 * Cg, Og]."""
    B, H, W, C, O, groups = dims
    Cg, Og, N = C // groups, O // groups, B * H * W
    go2d = grad_out.float().permute(0, 2, 3, 1).reshape(N, O)    # free for a channels_last gradient
    grad_wmat = torch.empty((groups, 9 * Cg, Og), dtype=torch.float32, device=go2d.device)
    grad_col = torch.empty((groups, N, 9 * Cg), dtype=torch.float32, device=go2d.device)
    # The weight gradient col[g]^T @ go_g is a reduction over all N pixels into a 9*Cg x Og matrix (1152 x 128 at BASELINE
    # configs[3]: nine output tiles, 36 TFLOP/s as one GEMM).  Split into S batched pieces over the pixels + one sum it fills
    # the chip: 547 -> 210 us for the four groups (tools/scratch/dcn_wgrad_gemm.py: S = 4 / 8 / 16 / 32 -> 277 / 210 / 262 / 395).
    S = 8 if N % 8 == 0 and N >= 4096 else 1
    parts = torch.empty((groups, S, 9 * Cg, Og), dtype=torch.float32, device=go2d.device) if S > 1 else None
    for g in range(groups):                                     # the group's Og gradient columns are read in place (leading dimension O)
        go_g = go2d[:, g * Og:(g + 1) * Og]
        if S > 1:
            torch.bmm(col[g].view(S, N // S, 9 * Cg).transpose(1, 2), go2d.view(S, N // S, O)[:, :, g * Og:(g + 1) * Og], out=parts[g])
        else:
            torch.mm(col[g].t(), go_g, out=grad_wmat[g])
    # the column gradient of all groups as ONE strided-batch GEMM (the groups' gradient columns read in place: batch stride Og,
    # row stride O): 276 -> 230 us against four GEMMs, bit-identical (tools/scratch/dcn_gradcol_gemm.py)
    torch.bmm(go2d.view(N, groups, Og).permute(1, 0, 2), wmat.transpose(1, 2), out=grad_col)
    if S > 1:
        torch.sum(parts, 1, out=grad_wmat)
    grad_off = torch.empty_like(off_nhwc)
    lpg = Cg // 4
    sorted_ok = H * W <= 4096 and Cg % 4 == 0 and lpg <= 64 and (lpg & (lpg - 1)) == 0
    with torch.cuda.device(x_nhwc.device):
        if sorted_ok:
            # contributions sorted by destination pixel, then a pure gather: no global atomics, no zero-fill
            grad_x = torch.empty_like(x_nhwc)
            ws = torch.empty((_lib.lib().mmt_dcn_col2im_workspace_elems(B, H, W),), dtype=torch.int32, device=x_nhwc.device)
            _lib.call("mmt_dcn_col2im_sorted", B, H, W, C, groups, x_nhwc.data_ptr(), off_nhwc.data_ptr(),
                      grad_col.data_ptr(), grad_x.data_ptr(), grad_off.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
        else:
            grad_x = torch.zeros_like(x_nhwc)
            _lib.call("mmt_dcn_col2im", B, H, W, C, groups, x_nhwc.data_ptr(), off_nhwc.data_ptr(),
                      grad_col.data_ptr(), grad_x.data_ptr(), grad_off.data_ptr(), _stream())
    return grad_x, grad_off, grad_wmat


def deform_conv3x3(x, offset, weight, groups=1, columns=False):
    """x [B,C,H,W], offset [B,18,H,W], weight [O, C/groups, 3, 3] -> [B,O,H,W] (channels_last).
    columns=True forces the im2col / col2im + GEMM form (the fallback; A/B tools and tests)."""
    if not x.is_cuda:
        raise RuntimeError("x must be a CUDAtensor ")
    groups = int(groups)
    B, C, H, W = x.shape
    if not columns and _lib.lib().mmt_dcn_mfma_supported(B, H, W, C, weight.shape[0], groups):
        return _DeformConv3x3Mfma.apply(x, offset, weight, groups)
    return _DeformConv3x3Columns.apply(x, offset, weight, groups)
