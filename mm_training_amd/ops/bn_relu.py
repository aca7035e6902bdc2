"""Fused training-mode BatchNorm2d (+ residual add) (+ ReLU) for the dense nets -- HIP
(``mmt_bn_relu_forward/backward``, csrc/bn_relu.hip): 3 + 5 streaming passes over the
activation instead of 5 + 8 with MIOpen BatchNorm + ATen add / relu.

``bn_act(bn, x, residual=None, relu=True)`` takes a plain ``nn.BatchNorm2d`` (its parameters,
running statistics, momentum and eps are used and updated exactly like ``bn(x)`` would) and
returns ``relu(bn(x) + residual)``.  The fused path needs training mode, CUDA channels-last
activations (fp32, or bf16 inside an autocast region: bf16 in and out, fp32 statistics and
arithmetic) and a supported channel count; everything else (eval mode, CPU tensors of the gloo
unit tests, odd C) runs the ordinary torch modules -- these layers are PyTorch plumbing around
the hot path, not part of it.
"""
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib

ENABLED = os.environ.get("MMT_FUSED_BN", "1") != "0"


_stream = _lib.raw_stream


def _supported(bn, x):
    """fp32 activations anywhere; bf16 activations inside an autocast region (the arithmetic is fp32 either way)."""
    c = x.shape[1]
    return (ENABLED and bn.training and x.is_cuda and x.dim() == 4
            and (x.dtype == torch.float32 or (x.dtype == torch.bfloat16 and torch.is_autocast_enabled()))
            and bn.track_running_stats and bn.momentum is not None and bn.affine
            and c % 4 == 0 and (c <= 1024 or (c % 256 == 0 and c <= 2048))
            and x.is_contiguous(memory_format=torch.channels_last))


def _frozen_inference(bn, x, residual):
    c = x.shape[1] if x.dim() == 4 else 0
    return (ENABLED and not bn.training and x.is_cuda and x.dim() == 4 and bn.affine and bn.track_running_stats
            and not x.requires_grad and not bn.weight.requires_grad and not bn.bias.requires_grad
            and (residual is None or (not residual.requires_grad and residual.dtype == x.dtype
                                      and residual.is_contiguous(memory_format=torch.channels_last)))
            and (x.dtype == torch.float32 or (x.dtype == torch.bfloat16 and torch.is_autocast_enabled()))
            and c % 4 == 0 and (c <= 1024 or (c % 256 == 0 and c <= 2048))
            and x.is_contiguous(memory_format=torch.channels_last))


class _BnAct(Function):
    """x, residual, y and their gradients share one dtype (fp32 or bf16); parameters, statistics and arithmetic are fp32."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, workspace, momentum, eps, relu, fork=False):
        B, C, H, W = x.shape
        R = B * H * W
        act = _lib.DTYPE_BF16 if x.dtype == torch.bfloat16 else _lib.DTYPE_F32
        y = torch.empty_like(x)                                   # preserves channels_last
        save = torch.empty(4 * C, dtype=torch.float32, device=x.device)
        res_ptr = 0
        if residual is not None:
            if not residual.is_contiguous(memory_format=torch.channels_last):
                residual = residual.contiguous(memory_format=torch.channels_last)
            res_ptr = residual.data_ptr()
        with _lib.on_device(x.device):
            _lib.call("mmt_bn_relu_forward_ex", R, C, x.data_ptr(), res_ptr, weight.data_ptr(), bias.data_ptr(),
                      running_mean.data_ptr(), running_var.data_ptr(), float(momentum), float(eps), int(relu),
                      workspace.data_ptr(), save.data_ptr(), y.data_ptr(), act, _stream(x.device))
        ctx.mark_non_differentiable(running_mean, running_var)
        need_y = relu and residual is not None
        ctx.save_for_backward(x, y if need_y else None, save, workspace)
        ctx.cfg = (R, C, bool(relu), residual is not None, act)
        if fork:
            # the output two or three times, as aliases of one buffer: the backward then receives the gradient of every use on its
            # own and adds them while loading (no accumulation pass by autograd in between)
            ctx.set_materialize_grads(False)
            return (y,) + tuple(y.view_as(y) for _ in range(int(fork) - 1))
        return y

    @staticmethod
    def backward(ctx, *grads):
        x, y, save, workspace = ctx.saved_tensors
        R, C, relu, has_res, act = ctx.cfg
        grads = [g for g in grads if g is not None]
        if not grads:                            # (no alias was used in what was differentiated)
            grads = [torch.zeros_like(x)]
        grad_y = grads[0]
        grad_y2 = grads[1] if len(grads) > 1 else None
        grad_y3 = grads[2] if len(grads) > 2 else None

        def _prep(g):
            if g.dtype != x.dtype:
                g = g.to(x.dtype)
            return g if g.is_contiguous(memory_format=torch.channels_last) else g.contiguous(memory_format=torch.channels_last)
        # a channel slice of a wider channels-last tensor (the gradient of one input of a torch.cat: ASPP's five branches) is read
        # in place through its row stride -- .contiguous() was a 35 MB copy per branch and step at BASELINE configs[3]
        pitch = 0
        Bn, Cn, Hn, Wn = grad_y.shape
        sb, sc, sh, sw = grad_y.stride()
        if (grad_y.dtype == x.dtype and not grad_y.is_contiguous(memory_format=torch.channels_last) and sc == 1 and sw > Cn and sw % 4 == 0
                and sh == Wn * sw and sb == Hn * Wn * sw and grad_y.data_ptr() % (16 if grad_y.element_size() == 4 else 8) == 0):
            pitch = sw
        else:
            grad_y = _prep(grad_y)
        grad_y2 = _prep(grad_y2) if grad_y2 is not None else None
        grad_y3 = _prep(grad_y3) if grad_y3 is not None else None
        grad_x = torch.empty_like(x)
        grad_res = torch.empty_like(x) if has_res else None
        grad_w = torch.empty(C, dtype=torch.float32, device=x.device)
        grad_b = torch.empty(C, dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            _lib.call("mmt_bn_relu_backward_ex2", R, C, x.data_ptr(), y.data_ptr() if y is not None else 0, grad_y.data_ptr(),
                      grad_y2.data_ptr() if grad_y2 is not None else 0, grad_y3.data_ptr() if grad_y3 is not None else 0, pitch,
                      save.data_ptr(), int(relu), int(has_res), workspace.data_ptr(), grad_x.data_ptr(),
                      grad_res.data_ptr() if has_res else 0, grad_w.data_ptr(), grad_b.data_ptr(), act, _stream(x.device))
        return grad_x, grad_res, grad_w, grad_b, None, None, None, None, None, None, None


_SCRATCH = {}
_NEED = {}


def _workspace(bn, device, width=None):
    """Scratch for the per-workgroup partial sums, one per (device, STREAM): consumed inside each call on the calling
    stream, so one buffer sized for the widest layer serves every BatchNorm the stream runs -- and layers that run on
    different streams at the same time (the task heads, layers/heads/bev_depth_head.py) never share one.  The backward
    of a layer runs on its forward's stream (autograd) and receives the same buffer."""
    width = max(2048, width or bn.num_features)
    need = _NEED.get(width)
    if need is None:
        need = _NEED[width] = _lib.lib().mmt_bn_workspace_elems(width)
    key = (device, _lib.raw_stream(device) if device.type == "cuda" else 0)
    ws = _SCRATCH.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=device)
        _SCRATCH[key] = ws
    return ws


FORK = os.environ.get("MMT_BN_FORK", "1") != "0"


def bn_act(bn, x, residual=None, relu=True, fork=False):
    """relu?(bn(x) [+ residual]) with ``bn`` an ``nn.BatchNorm2d`` (see module docstring).  ``fork=True`` (or 2; 3 for three uses)
    returns the result as a tuple of aliases ``(y, y', ..)`` of one buffer for a caller that uses it twice (a residual block's output: the next block's first
    convolution and its identity): the two gradients then meet inside the fused backward (added while loading,
    ``mmt_bn_relu_backward_ex2``) instead of in an accumulation pass of autograd's -- three streams over the activation less per
    residual join.  On the unfused path the pair is the same tensor twice."""
    if _supported(bn, x):
        if residual is not None and residual.dtype != x.dtype:
            residual = residual.to(x.dtype)              # (autograd casts the gradient back)
        # inside an autocast region: the statistics and the normalisation run in fp32 on the fused kernels -- what autocast does
        # for batch_norm anyway -- with bf16 activations in and out, instead of MIOpen's NHWC batch norm (an out-of-bounds
        # access inside it was met twice: DESIGN section 4, fuzz_dense; tools/scratch/soak_streams.py)
        # (nothing inside the Function is an autocast-wrapped operator: no `autocast(enabled=False)` region around it)
        return _lib.apply_function(_BnAct, x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                   _workspace(bn, x.device), bn.momentum, bn.eps, relu, (2 if fork is True else int(fork)) if (fork and FORK) else 0)
    if _frozen_inference(bn, x, residual):
        # eval-mode BatchNorm of a FROZEN layer on an input that carries no gradient (the image backbone's stem under the
        # reference's frozen_stages=0): one pass, no autograd node
        act = _lib.DTYPE_BF16 if x.dtype == torch.bfloat16 else _lib.DTYPE_F32
        y = torch.empty_like(x)
        C = x.shape[1]
        with _lib.on_device(x.device):
            _lib.call("mmt_bn_relu_inference", x.shape[0] * x.shape[2] * x.shape[3], C, x.data_ptr(),
                      residual.data_ptr() if residual is not None else 0, bn.weight.data_ptr(), bn.bias.data_ptr(),
                      bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.eps), int(relu),
                      _workspace(bn, x.device).data_ptr(), y.data_ptr(), act, _lib.raw_stream(x.device))
        return (y,) * (2 if fork is True else int(fork)) if fork else y
    out = bn(x)
    if residual is not None:
        out = out + residual
    out = F.relu(out, inplace=True) if relu else out
    return (out,) * (2 if fork is True else int(fork)) if fork else out


class ConvBNAct(torch.nn.Sequential):
    """``nn.Sequential(conv, BatchNorm2d[, ReLU])`` with the same child indices (so state_dict keys
    are those of the plain Sequential the reference's mmcv / mmdet modules produce) whose forward
    runs the normalisation (+ ReLU) through ``bn_act``."""

    fork = False               # 2 | 3: the output as that many aliases (bn_act), for a caller that reads it that often

    def forward(self, x):
        relu = len(self) > 2 and isinstance(self[2], torch.nn.ReLU)
        out = bn_act(self[1], self[0](x), relu=relu, fork=self.fork)
        for extra in list(self)[3 if relu else 2:]:
            out = extra(out)
        return out
