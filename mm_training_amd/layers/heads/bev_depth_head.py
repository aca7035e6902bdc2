"""BEVDepthHead -- mirror of layers/heads/bev_depth_head.py (a CenterPoint head,
mmdet3d CenterHead subclass in the reference) in plain PyTorch.

Same constructor arguments, ``forward(x)`` output structure (tuple over tasks of
``[dict(reg, height, dim, rot, vel, heatmap)]``), ``get_targets`` and ``loss``
semantics (bev_depth_head.py:85-111, :113-254, :256-312).  Host-side differences:
  * target drawing is vectorised on the device (no per-box Python loop, no per-box
    tensor creation: bev_depth_head.py:186-248);
  * the loss normalisers of all tasks are reduced across ranks with ONE all-reduce and
    stay on the device -- no ``.item()`` round trips (:273-276, :300-301).
"""
import os

import torch
import torch.distributed as dist
from torch import nn

from ..nets import ResNet, SECONDFPN
from ...ops.bn_relu import ConvBNAct
from ...ops.train_targets import centerpoint_targets

__all__ = ['BEVDepthHead']

bev_backbone_conf = dict(type='ResNet', in_channels=80, depth=18, num_stages=3, strides=(1, 2, 2),
                         dilations=(1, 1, 1), out_indices=[0, 1, 2], norm_eval=False, base_channels=160)
bev_neck_conf = dict(type='SECONDFPN', in_channels=[160, 320, 640], upsample_strides=[2, 4, 8],
                     out_channels=[64, 64, 128])


def clip_sigmoid(x, eps=1e-4):
    return torch.clamp(x.sigmoid(), min=eps, max=1 - eps)


def gaussian_radius(height, width, min_overlap=0.5):
    """mmdet3d.core.gaussian_radius, vectorised over tensors."""
    a1 = 1
    b1 = height + width
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + torch.sqrt(b1 ** 2 - 4 * a1 * c1)) / 2
    a2 = 4
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 + torch.sqrt(b2 ** 2 - 4 * a2 * c2)) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + torch.sqrt(b3 ** 2 - 4 * a3 * c3)) / 2
    return torch.minimum(torch.minimum(r1, r2), r3)


def _conv_module(cin, cout, k):
    return ConvBNAct(nn.Conv2d(cin, cout, k, 1, k // 2, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


_TASK_STREAMS = {}


def streams_in_use():
    """Every stream the task heads have been dealt to in this process."""
    return [s for group in _TASK_STREAMS.values() for s in group]


class _Fork(torch.autograd.Function):
    """x -> n aliases of x, one per stream; backward: the sum of the n gradients, formed on the stream the fork ran on.  Every
    alias is consumed on ONE stream, so each gradient slot of this node has a single producer stream and autograd's plain
    cross-stream hand-over applies; the branches' gradients are never accumulated across streams into one buffer."""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        grads = [g for g in grads if g is not None]
        if not grads:
            return None, None
        g0 = grads[0]
        if (len(grads) > 2 and g0.is_cuda and all(g.dtype == torch.float32 and g.shape == g0.shape and g.stride() == g0.stride() for g in grads)
                and (g0.is_contiguous() or g0.is_contiguous(memory_format=torch.channels_last))):
            # one pass over the n gradients (n reads + one write) instead of n - 1 accumulation kernels
            import ctypes
            from ... import _lib
            out = torch.empty_like(g0)
            for lo in range(0, len(grads), 31):
                part = ([out] if lo else []) + grads[lo:lo + 31]
                arr = (ctypes.c_void_p * len(part))(*[t.data_ptr() for t in part])
                with torch.cuda.device(g0.device):
                    _lib.call("mmt_add_n", len(part), arr, g0.numel(), out.data_ptr(), torch.cuda.current_stream(g0.device).cuda_stream)
            return out, None
        total = None
        for g in grads:
            total = g if total is None else total + g
        return total, None


class _SplitBlocks(torch.autograd.Function):
    """A channels-last map [B, n * w, H, W] -> n dense channels-last maps [B, w, H, W] (views of one buffer), one pass
    (`mmt_channel_blocks_split`); backward: the n gradients side by side again, one pass (`mmt_channel_blocks_gather`)."""

    @staticmethod
    def forward(ctx, wide, n):
        import ctypes
        from ... import _lib
        B, C, H, W = wide.shape
        w = C // n
        buf = torch.empty((n, B, H, W, w), dtype=wide.dtype, device=wide.device)
        parts = tuple(buf[k].permute(0, 3, 1, 2) for k in range(n))
        arr = (ctypes.c_void_p * n)(*[t.data_ptr() for t in parts])
        with _lib.on_device(wide.device):
            _lib.call("mmt_channel_blocks_split", B * H * W, n, w * wide.element_size(), wide.data_ptr(), arr, _lib.raw_stream(wide.device))
        ctx.dims = (B, C, H, W, n, w)
        ctx.set_materialize_grads(False)
        return parts

    @staticmethod
    def backward(ctx, *grads):
        import ctypes
        from ... import _lib
        B, C, H, W, n, w = ctx.dims
        ref = next((g for g in grads if g is not None), None)
        if ref is None:
            return None, None
        out = torch.empty((B, C, H, W), dtype=ref.dtype, device=ref.device, memory_format=torch.channels_last)
        zero = None
        ptrs = []
        keep = []
        for g in grads:
            if g is None:
                if zero is None:
                    zero = torch.zeros((B, w, H, W), dtype=ref.dtype, device=ref.device, memory_format=torch.channels_last)
                g = zero
            elif g.dtype != ref.dtype or not g.is_contiguous(memory_format=torch.channels_last):
                g = g.to(ref.dtype).contiguous(memory_format=torch.channels_last)
            keep.append(g)
            ptrs.append(g.data_ptr())
        arr = (ctypes.c_void_p * n)(*ptrs)
        with _lib.on_device(ref.device):
            _lib.call("mmt_channel_blocks_gather", B * H * W, n, w * ref.element_size(), arr, out.data_ptr(), _lib.raw_stream(ref.device))
        return out, None


class _FinalConvs(torch.autograd.Function):
    """The branches' final 3x3 convolutions (64 -> 1..4 channels each, bias) on the wide map [B, n * 64, H, W], one launch per
    direction (`mmt_heads_final_forward / _backward`, csrc/thin_conv.hip).  weight [KT, 64, 3, 3] channels-last = the branches' weights
    one after the other, bias [KT]; returns the n outputs as channel slices of ONE narrow map [B, KT, H, W]."""

    @staticmethod
    def forward(ctx, wide, weight, bias, ks):
        import ctypes
        from ... import _lib
        B, C, H, W = wide.shape
        n, kt = len(ks), sum(ks)
        out = torch.empty((B, kt, H, W), dtype=wide.dtype, device=wide.device, memory_format=torch.channels_last)
        karr = (ctypes.c_ubyte * n)(*ks)
        act = _lib.DTYPE_BF16 if wide.dtype == torch.bfloat16 else _lib.DTYPE_F32
        with _lib.on_device(wide.device):
            _lib.call("mmt_heads_final_forward", B, H, W, n, karr, wide.data_ptr(), weight.data_ptr(), bias.data_ptr(), out.data_ptr(),
                      act, _lib.raw_stream(wide.device))
        ctx.save_for_backward(wide, weight)
        ctx.ks = tuple(ks)
        ctx.set_materialize_grads(False)
        offs = [sum(ks[:j]) for j in range(n)]
        # the branches' outputs, and the whole narrow map once more (what the fused loss reads, _HeadLoss)
        return tuple(out[:, o:o + k] for o, k in zip(offs, ks)) + (out,)

    @staticmethod
    def backward(ctx, *grads):
        import ctypes
        from ... import _lib
        wide, weight = ctx.saved_tensors
        ks = ctx.ks
        B, C, H, W = wide.shape
        n, kt = len(ks), sum(ks)
        if all(g is None for g in grads):
            return None, None, None, None
        whole, grads = grads[-1], grads[:-1]
        if whole is not None and all(g is None for g in grads):
            # the gradient of the whole map (the fused loss): used as it comes
            gout = whole if whole.dtype == wide.dtype else whole.to(wide.dtype)
            if not gout.is_contiguous(memory_format=torch.channels_last):
                gout = gout.contiguous(memory_format=torch.channels_last)
        else:
            gout = torch.empty((B, kt, H, W), dtype=wide.dtype, device=wide.device, memory_format=torch.channels_last)
            if whole is not None:
                gout.copy_(whole)
            elif any(g is None for g in grads):
                gout.zero_()
            o = 0
            for g, k in zip(grads, ks):
                if g is not None:
                    if whole is not None:
                        gout[:, o:o + k].add_(g)
                    else:
                        gout[:, o:o + k].copy_(g)
                o += k
        need_z, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        gz = torch.empty_like(wide) if need_z else None
        gw = gb = ws = None
        if need_w:
            gw = torch.empty_like(weight)                                  # channels-last [KT, 64, 3, 3] = [KT][9][64]
            gb = torch.empty(kt, dtype=torch.float32, device=wide.device)
            ws = torch.empty(_lib.lib().mmt_heads_final_workspace_elems(B, H, n), dtype=torch.float32, device=wide.device)
        karr = (ctypes.c_ubyte * n)(*ks)
        act = _lib.DTYPE_BF16 if wide.dtype == torch.bfloat16 else _lib.DTYPE_F32
        with _lib.on_device(wide.device):
            _lib.call("mmt_heads_final_backward", B, H, W, n, karr, wide.data_ptr(), weight.data_ptr(), gout.data_ptr(),
                      gz.data_ptr() if need_z else 0, gw.data_ptr() if need_w else 0, gb.data_ptr() if need_w else 0,
                      ws.data_ptr() if need_w else 0, act, _lib.raw_stream(wide.device))
        return gz, gw, gb, None


class _Preds(dict):
    """A task's prediction dict (reg, height, dim, rot, vel, heatmap -- the reference's keys) that also remembers the ONE map the
    six tensors are channel slices of, and the task's first channel in it (for the fused loss; invisible to dict consumers)."""
    __slots__ = ("fused_map", "first_channel")


class _HeadLoss(torch.autograd.Function):
    """The head's loss on the fused heads' one output map and its gradient in two launches (`mmt_head_loss_forward_backward`)."""

    @staticmethod
    def forward(ctx, fmap, heatmaps, annos, inds, masks, norm, code_weights, box_weight):
        import ctypes
        from ... import _lib
        B, KT, H, W = fmap.shape
        T, M = len(heatmaps), annos[0].shape[1]
        grad = torch.empty((B, H, W, KT), dtype=torch.float32, device=fmap.device)
        partials = torch.empty(_lib.lib().mmt_head_loss_partials(B, H, W, T, M), dtype=torch.float32, device=fmap.device)
        arr = lambda ts: (ctypes.c_void_p * T)(*[t.data_ptr() for t in ts])
        act = _lib.DTYPE_BF16 if fmap.dtype == torch.bfloat16 else _lib.DTYPE_F32
        keep = (heatmaps, annos, inds, masks)                   # (contiguous copies, if any, live until the launch is queued)
        with _lib.on_device(fmap.device):
            _lib.call("mmt_head_loss_forward_backward", B, H, W, T, M, fmap.data_ptr(), arr(heatmaps), arr(annos), arr(inds), arr(masks),
                      norm.data_ptr(), code_weights.data_ptr(), float(box_weight), grad.data_ptr(), partials.data_ptr(), act,
                      _lib.raw_stream(fmap.device))
        del keep
        ctx.save_for_backward(grad)
        ctx.dims = (B, KT, H, W)
        return partials.sum()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        B, KT, H, W = ctx.dims
        return (grad * g).view(B, H, W, KT).permute(0, 3, 1, 2), None, None, None, None, None, None, None


class SeparateHead(nn.Module):
    def __init__(self, in_channels, heads, head_conv=64, final_kernel=3, init_bias=-2.19):
        super().__init__()
        self.heads = heads
        for head, (classes, num_conv) in heads.items():
            layers = []
            c = in_channels
            for _ in range(num_conv - 1):
                layers.append(_conv_module(c, head_conv, final_kernel))
                c = head_conv
            layers.append(nn.Conv2d(c, classes, final_kernel, 1, final_kernel // 2, bias=True))
            setattr(self, head, nn.Sequential(*layers))
        self.heatmap[-1].bias.data.fill_(init_bias)

    def forward(self, x):
        """x: the shared map, or one alias of it per head (BEVDepthHead forks it: the heads' gradients then come back separately and
        are added in one pass, _Fork)."""
        if isinstance(x, (tuple, list)):
            return {head: getattr(self, head)(xk) for head, xk in zip(self.heads, x)}
        return {head: getattr(self, head)(x) for head in self.heads}


class BEVDepthHead(nn.Module):
    def __init__(self, in_channels=256, tasks=None, bbox_coder=None, common_heads=dict(),
                 loss_cls=dict(type='GaussianFocalLoss', reduction='mean'),
                 loss_bbox=dict(type='L1Loss', reduction='mean', loss_weight=0.25),
                 gaussian_overlap=0.1, min_radius=2, train_cfg=None, test_cfg=None,
                 bev_backbone_conf=bev_backbone_conf, bev_neck_conf=bev_neck_conf,
                 separate_head=dict(type='SeparateHead', init_bias=-2.19, final_kernel=3),
                 share_conv_channel=64):
        super().__init__()
        self.class_names = [t['class_names'] for t in tasks]
        self.num_classes = [len(t['class_names']) for t in tasks]
        self.norm_bbox = True
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.gaussian_overlap = gaussian_overlap
        self.min_radius = min_radius
        self.loss_bbox_weight = float(loss_bbox.get('loss_weight', 1.0))
        # on-device once: building it inside loss() is a blocking pageable host->device copy every step
        self.register_buffer('code_weights', torch.tensor((train_cfg or {}).get('code_weights', [1.0] * 10), dtype=torch.float32),
                             persistent=False)
        bb = {k: v for k, v in dict(bev_backbone_conf).items() if k not in ('type', 'norm_eval')}
        self.trunk = ResNet(**bb)
        self.neck = SECONDFPN(**{k: v for k, v in dict(bev_neck_conf).items() if k != 'type'})
        self.shared_conv = _conv_module(in_channels, share_conv_channel, 3)
        # HIP streams the task heads are dealt to in training (see _forward_tasks_on_streams); 0 / 1 = the caller's stream only
        self.task_streams = int(os.environ.get("MMT_HEAD_STREAMS", "2"))
        # the branches' first ConvModules as one wide convolution + one BatchNorm in training (_forward_tasks_fused); 0: per branch
        # "auto" (default): see forward(); True / False (MMT_HEAD_FUSED=1 / 0): always / never
        self.fuse_branch_stems = {"0": False, "1": True}.get(os.environ.get("MMT_HEAD_FUSED", "auto"), "auto")
        # with the fused first layer: the branches' final convolutions as one hand-written kernel per direction (csrc/thin_conv.hip)
        self.fuse_final_convs = os.environ.get("MMT_HEAD_FINALS", "1") != "0"
        # with both: the loss and its gradient on the heads' one output map in two launches (csrc/head_loss.hip)
        self.fuse_loss = os.environ.get("MMT_HEAD_LOSS", "1") != "0"
        self.task_heads = nn.ModuleList()
        for n in self.num_classes:
            heads = dict(common_heads)
            heads['heatmap'] = (n, 2)
            self.task_heads.append(SeparateHead(share_conv_channel, heads, head_conv=64,
                                                final_kernel=separate_head.get('final_kernel', 3),
                                                init_bias=separate_head.get('init_bias', -2.19)))

    def forward(self, x):
        fpn_output = self.neck(self.trunk(x))
        x = self.shared_conv(fpn_output[0])
        fuse = self.fuse_branch_stems
        if fuse == "auto":
            # with the final convolutions in the hand-written kernels (no copies around them): always (configs[2], bs 8: 47.3 ms per
            # branch, 44.5 fused); without them: when the branches would share one stream, or on maps of at most 64 K pixels
            fuse = self.fuse_final_convs or self.task_streams <= 1 or x.shape[0] * x.shape[2] * x.shape[3] <= 65536
        if fuse and x.is_cuda and torch.is_grad_enabled() and self.training:
            stems = self._branch_stems(x)
            if stems is not None:
                return self._forward_tasks_fused(x, stems)
        if self.task_streams > 1 and x.is_cuda and torch.is_grad_enabled():
            return self._forward_tasks_on_streams(x, self.task_streams)
        if x.is_cuda and torch.is_grad_enabled() and x.requires_grad:
            per = [len(task.heads) for task in self.task_heads]
            aliases = _Fork.apply(x, sum(per))
            outs, first = [], 0
            for task, n in zip(self.task_heads, per):
                outs.append([task(aliases[first:first + n])])
                first += n
            return tuple(outs)
        return tuple([task(x)] for task in self.task_heads)

    def _branch_stems(self, x):
        """[(first convolution, its BatchNorm, final convolution)] over all branches of all tasks when they can run as one wide
        layer: every branch = ConvModule(3x3, no bias, BatchNorm, ReLU) + a final convolution, one shape for all, at most 32 of
        them, a channel total the fused BatchNorm takes.  None otherwise (the per-branch path runs)."""
        from ...ops.bn_relu import ConvBNAct
        stems = []
        for task in self.task_heads:
            for name in task.heads:
                seq = getattr(task, name)
                if len(seq) != 2 or not isinstance(seq[0], ConvBNAct) or len(seq[0]) != 3 or not isinstance(seq[0][2], nn.ReLU):
                    return None
                conv, bn = seq[0][0], seq[0][1]
                if not (isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d) and isinstance(seq[1], nn.Conv2d)):
                    return None
                stems.append((conv, bn, seq[1]))
        c0, b0, _ = stems[0]
        total = c0.out_channels * len(stems)
        same = all(c.weight.shape == c0.weight.shape and c.bias is None and c.stride == c0.stride == (1, 1) and c.padding == c0.padding
                   and c.dilation == c0.dilation == (1, 1) and c.groups == 1 and c.padding_mode == "zeros"
                   and b.training and b.affine and b.track_running_stats and b.momentum == b0.momentum and b.momentum is not None
                   and b.eps == b0.eps and c.weight.stride() == c0.weight.stride() for c, b, _ in stems)
        if not same or len(stems) > 32 or (c0.out_channels * x.element_size()) % 16 or not (total <= 1024 or (total % 256 == 0 and total <= 2048)):
            return None
        return stems

    def _stem_statistics(self, bns):
        """The branches' running statistics as ONE buffer each, the modules' own buffers views of it (the fused BatchNorm updates them
        in place; state_dict keys and values stay per branch).  Re-established when something (`.to()`, a deep copy) has given the
        modules separate buffers again."""
        w = bns[0].num_features
        cur = getattr(self, "_stem_stats", None)
        first, last = bns[0].running_mean, bns[-1].running_mean
        if (cur is not None and first.data_ptr() == cur[0].data_ptr() and last.data_ptr() == cur[0].data_ptr() + 4 * w * (len(bns) - 1)
                and bns[0].running_var.data_ptr() == cur[1].data_ptr() and bns[-1].running_var.data_ptr() == cur[1].data_ptr() + 4 * w * (len(bns) - 1)):
            return cur
        with torch.no_grad():
            mean = torch.cat([b.running_mean for b in bns])
            var = torch.cat([b.running_var for b in bns])
            for k, b in enumerate(bns):
                b.running_mean = mean[k * w:(k + 1) * w]
                b.running_var = var[k * w:(k + 1) * w]
        object.__setattr__(self, "_stem_stats", (mean, var))
        return mean, var

    def _forward_tasks_fused(self, x, stems):
        """The 24 branches' first ConvModules (64 -> 64, 3x3, BatchNorm, ReLU -- all on the one shared map) as ONE 64 -> 24 x 64
        convolution and ONE BatchNorm over the 1536 channels: the same sums per output channel and the same per-channel statistics,
        in 3 big kernels per direction instead of 24 x 3 small ones (MIOpen, fp32 [4, 64, 128, 128]: 0.90 / 0.92 / 0.93 ms forward /
        data gradient / weight gradient against 24 x 53 / 56 / 58 us; the BatchNorm passes run at streaming speed instead of 35 us
        of latency per 17 MB map; no 24-way gradient sum).  The parameters stay the modules' own (same `state_dict`): the wide
        weight is their torch.cat, whose backward hands each of them a view of the one weight gradient."""
        from ...ops import conv_overlap
        from ...ops import bn_relu
        from ... import _lib
        convs, bns, finals = zip(*stems)
        n = len(stems)
        weight = torch.cat([c.weight for c in convs], 0)
        mode = getattr(convs[0], "_mmt_overlap_mode", "inline")
        y = conv_overlap.conv2d(x, weight, None, convs[0].stride, convs[0].padding, convs[0].dilation, 1, mode, leaves=[c.weight for c in convs])
        mean, var = self._stem_statistics(bns)
        gamma = torch.cat([b.weight for b in bns])
        beta = torch.cat([b.bias for b in bns])
        if not y.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous(memory_format=torch.channels_last)
        z = _lib.apply_function(bn_relu._BnAct, y, None, gamma, beta, mean, var, bn_relu._workspace(bns[0], y.device, width=y.shape[1]),
                                bns[0].momentum, bns[0].eps, True, 0)
        if self.fuse_final_convs and self._finals_fit(finals, z):
            # the 24 final convolutions (64 -> 1..3 channels) in one launch per direction, straight on the wide map: no split, no
            # 24 gradients to put side by side again (_FinalConvs)
            weight = torch.cat([f.weight for f in finals], 0)
            if not weight.is_contiguous(memory_format=torch.channels_last):
                weight = weight.contiguous(memory_format=torch.channels_last)
            bias = torch.cat([f.bias for f in finals])
            ks = tuple(f.out_channels for f in finals)
            parts = _FinalConvs.apply(z, weight.float(), bias.float(), ks)
            outs, k = [], 0
            for task in self.task_heads:
                preds = _Preds((name, parts[k + i]) for i, name in enumerate(task.heads))
                preds.fused_map, preds.first_channel = parts[-1], sum(ks[:k])
                outs.append([preds])
                k += len(task.heads)
            return tuple(outs)
        parts = _SplitBlocks.apply(z, n)
        outs, k = [], 0
        for task in self.task_heads:
            out = {}
            for name in task.heads:
                out[name] = finals[k](parts[k])
                k += 1
            outs.append([out])
        return tuple(outs)

    @staticmethod
    def _finals_fit(finals, z):
        c = z.shape[1] // len(finals)
        return (c == 64 and len(finals) <= 32 and z.is_contiguous(memory_format=torch.channels_last)
                and all(f.kernel_size == (3, 3) and f.stride == (1, 1) and f.padding == (1, 1) and f.dilation == (1, 1) and f.groups == 1
                        and f.in_channels == 64 and 1 <= f.out_channels <= 4 and f.bias is not None and f.padding_mode == "zeros"
                        and f.weight.dtype == torch.float32 for f in finals))

    def _forward_tasks_on_streams(self, x, nstreams):
        """The task heads (24 independent conv-bn-relu-conv branches on one 4 x 64 x 128 x 128 map, ~55 us per kernel) dealt to
        `nstreams` HIP streams: their kernels fill each other's launch gaps and tails.  autograd runs every backward node on its
        forward stream (and orders the streams itself, including the gradient accumulation of the parameters), so the backward
        overlaps the same way.  Measured at BASELINE configs[3]: 2 streams 67.3 -> 66.1 ms per step; 3, 4, 8 streams no gain --
        a process has four hardware queues, and the main stream, the weight-gradient stream (ops/conv_overlap.py) or RCCL's
        stream, and these two fill them (GPU_MAX_HW_QUEUES=8 made every variant slower)."""
        main = torch.cuda.current_stream(x.device)
        key = (x.device.index, nstreams)
        if key not in _TASK_STREAMS:                 # per process and device, not per module (a module stays deepcopy- / pickle-able)
            _TASK_STREAMS[key] = [torch.cuda.Stream(device=x.device) for _ in range(nstreams)]
        streams = _TASK_STREAMS[key]
        outs = []
        # one alias per BRANCH (every alias is consumed on one stream): the 24 gradients meet in _Fork.backward, in one pass
        per = [len(task.heads) for task in self.task_heads]
        aliases = _Fork.apply(x, sum(per)) if x.requires_grad else (x,) * sum(per)
        first = 0
        for i, task in enumerate(self.task_heads):
            s = streams[i % nstreams]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                out = task(aliases[first:first + per[i]])
            first += per[i]
            x.record_stream(s)
            for v in out.values():
                v.record_stream(main)
            outs.append([out])
        for s in streams:
            main.wait_stream(s)
        return tuple(outs)

    # ------------------------------------------------------------------ targets
    @torch.no_grad()
    def get_targets(self, gt_bboxes_3d, gt_labels_3d):
        """list over samples of boxes [K,9] (x,y,z,w,l,h,yaw,vx,vy) and labels [K] ->
        (heatmaps, anno_boxes, inds, masks): lists over tasks of batched tensors.
        One HIP op (ops/train_targets.py); CUDA tensors only."""
        cfg = self.train_cfg
        osf = cfg['out_size_factor']
        return centerpoint_targets(
            gt_bboxes_3d, gt_labels_3d, self.num_classes, cfg['max_objs'] * cfg['dense_reg'],
            (int(cfg['grid_size'][0]) // osf, int(cfg['grid_size'][1]) // osf), cfg['point_cloud_range'],
            cfg['voxel_size'], osf, cfg['gaussian_overlap'], cfg['min_radius'], self.norm_bbox)

    @torch.no_grad()
    def get_targets_torch(self, gt_bboxes_3d, gt_labels_3d):
        """The same targets from vectorised torch ops: the cross-check of
        tests/test_train_targets_gpu.py and what the CPU/gloo test of the data-parallel loss
        normalisers (tests/test_dp_gloo.py) feeds the head with.  Not called by the training step."""
        per_sample = [self.get_targets_single(b, l) for b, l in zip(gt_bboxes_3d, gt_labels_3d)]
        out = []
        for field in range(4):
            out.append([torch.stack([s[field][t] for s in per_sample]) for t in range(len(self.task_heads))])
        return tuple(out)

    @torch.no_grad()
    def get_targets_single(self, boxes, labels):
        """Vectorised and free of host synchronisation: every task looks at ALL K boxes of the sample, ranks its own in the
        reference's order (class after class, input order inside a class: bev_depth_head.py:142-163) and keeps the first
        max_objs of them at their rank (:171,186-248) -- the reference's slots (tests/golden/centerpoint_targets.npz)."""
        cfg = self.train_cfg
        dev = boxes.device
        max_objs = cfg['max_objs'] * cfg['dense_reg']
        osf = cfg['out_size_factor']
        fx = int(cfg['grid_size'][0]) // osf
        fy = int(cfg['grid_size'][1]) // osf
        pc = cfg['point_cloud_range']
        vs = cfg['voxel_size']
        labels = labels.long()
        K = boxes.shape[0]
        heatmaps, anno_boxes, inds, masks = [], [], [], []
        if K > 0:
            ys = torch.arange(fy, device=dev, dtype=torch.float32).view(1, fy, 1)
            xs = torch.arange(fx, device=dev, dtype=torch.float32).view(1, 1, fx)
            width = boxes[:, 3] / vs[0] / osf
            length = boxes[:, 4] / vs[1] / osf
            radius = gaussian_radius(length, width, min_overlap=cfg['gaussian_overlap'])
            radius = torch.clamp(radius.nan_to_num(0).floor(), min=float(cfg['min_radius']))
            cx = (boxes[:, 0] - pc[0]) / vs[0] / osf
            cy = (boxes[:, 1] - pc[1]) / vs[1] / osf
            cxi, cyi = cx.to(torch.int32), cy.to(torch.int32)
            in_map = (width > 0) & (length > 0) & (cxi >= 0) & (cxi < fx) & (cyi >= 0) & (cyi < fy)
            # draw_heatmap_gaussian: sigma = (2r+1)/6, window |d| <= r, max-combine
            sigma = ((2 * radius + 1) / 6).view(K, 1, 1)
            dx = xs - cxi.view(K, 1, 1).float()
            dy = ys - cyi.view(K, 1, 1).float()
            r = radius.view(K, 1, 1)
            gauss = torch.exp(-(dx * dx + dy * dy) / (2 * sigma * sigma)) * ((dx.abs() <= r) & (dy.abs() <= r))
            dims = boxes[:, 3:6].log() if self.norm_bbox else boxes[:, 3:6]
            anno_all = torch.cat([(cx - cxi.float()).unsqueeze(1), (cy - cyi.float()).unsqueeze(1), boxes[:, 2:3], dims,
                                  torch.sin(boxes[:, 6:7]), torch.cos(boxes[:, 6:7]), boxes[:, 7:9]], 1)
            ind_all = (cyi * fx + cxi).long().clamp(0, fx * fy - 1)
        flag = 0
        for t, names in enumerate(self.class_names):
            n_cls = len(names)
            heatmap = torch.zeros((n_cls, fy, fx), device=dev)
            anno = torch.zeros((max_objs, 10), device=dev)
            ind = torch.zeros((max_objs,), dtype=torch.int64, device=dev)
            mask = torch.zeros((max_objs,), dtype=torch.uint8, device=dev)
            if K > 0:
                # the reference's slots (:141-163, :171): a task's boxes densely, class after class, each class in input order; its first
                # max_objs boxes only.  A kept box that is not drawn (outside the map, zero size) leaves its slot empty.
                in_task = (labels >= flag) & (labels < flag + n_cls)
                cls = (labels - flag).clamp(0, n_cls - 1)
                key = torch.where(in_task, cls * K + torch.arange(K, device=dev), torch.full_like(cls, n_cls * K))
                rank = torch.empty_like(key)
                rank[key.argsort()] = torch.arange(K, device=dev)
                valid = in_map & in_task & (rank < max_objs)
                slot = torch.where(valid, rank, torch.full_like(rank, max_objs))          # (invalid boxes write to a scratch row)
                heatmap.index_reduce_(0, cls, gauss * valid.view(K, 1, 1), 'amax', include_self=True)
                ind_x = torch.zeros((max_objs + 1,), dtype=torch.int64, device=dev)
                mask_x = torch.zeros((max_objs + 1,), dtype=torch.uint8, device=dev)
                anno_x = torch.zeros((max_objs + 1, 10), device=dev)
                ind_x[slot] = ind_all
                mask_x[slot] = 1
                anno_x[slot] = anno_all
                ind, mask, anno = ind_x[:max_objs], mask_x[:max_objs], anno_x[:max_objs]
            flag += n_cls
            heatmaps.append(heatmap)
            anno_boxes.append(anno)
            inds.append(ind)
            masks.append(mask)
        return heatmaps, anno_boxes, inds, masks

    # --------------------------------------------------------------------- loss
    def loss_normalisers(self, targets):
        """[2 * n_task] tensor: per task the number of positive heatmap cells (bev_depth_head.py:273-276) and of masked
        box slots (:300-301) -- the two quantities the reference mean-reduces across ranks."""
        heatmaps, _, _, masks = targets
        return torch.stack([h.eq(1).float().sum() for h in heatmaps] + [m.float().sum() for m in masks])

    def _fused_loss_map(self, preds_dicts, heatmaps, anno_boxes, inds, masks):
        """The one map all predictions are slices of, when the fused loss applies: the fused heads' outputs (_Preds) in the standard
        layout (per task reg 2, height 1, dim 3, rot 2, vel 2, heatmap 1 = 11 channels, tasks one after the other), single-class
        tasks, at most 8 of them, targets of the matching shapes on the map's device."""
        if not self.fuse_loss:
            return None
        fmap = None
        for t, preds in enumerate(preds_dicts):
            p = preds[0]
            m = getattr(p, "fused_map", None)
            if m is None or (fmap is not None and m is not fmap) or p.first_channel != 11 * t:
                return None
            if [(k, v.shape[1]) for k, v in p.items()] != [("reg", 2), ("height", 1), ("dim", 3), ("rot", 2), ("vel", 2), ("heatmap", 1)]:
                return None
            fmap = m
        T = len(preds_dicts)
        if fmap is None or not fmap.is_cuda or T > 8 or fmap.shape[1] != 11 * T or not fmap.is_contiguous(memory_format=torch.channels_last):
            return None
        B, _, H, W = fmap.shape
        for t in range(T):
            if (tuple(heatmaps[t].shape) != (B, 1, H, W) or anno_boxes[t].dim() != 3 or anno_boxes[t].shape[0] != B or anno_boxes[t].shape[2] != 10
                    or tuple(inds[t].shape) != tuple(anno_boxes[t].shape[:2]) or tuple(masks[t].shape) != tuple(inds[t].shape)
                    or anno_boxes[t].shape[1] != anno_boxes[0].shape[1] or heatmaps[t].device != fmap.device):
                return None
        return fmap

    def loss(self, targets, preds_dicts, normalisers=None, **kwargs):
        """`normalisers` (optional): use these instead of the (cross-rank mean of the) batch's own -- a single process
        that accumulates the gradients of N micro-batches reproduces N data-parallel ranks by passing the mean of the
        micro-batches' `loss_normalisers` (tests/test_dp_gpu.py)."""
        heatmaps, anno_boxes, inds, masks = targets
        n_task = len(preds_dicts)
        if normalisers is not None:
            norm = normalisers
        else:
            # all normalisers in one tensor -> one all-reduce, no host sync
            norm = self.loss_normalisers(targets)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dist.all_reduce(norm)
                norm = norm / dist.get_world_size()
        fmap = self._fused_loss_map(preds_dicts, heatmaps, anno_boxes, inds, masks)
        if fmap is not None:
            # every prediction is a channel slice of ONE map (the fused heads): loss + gradient in two launches (csrc/head_loss.hip)
            # instead of ~150 small ATen kernels on 64 K-element tensors, forward and backward
            cont = lambda ts: [t if t.is_contiguous() else t.contiguous() for t in ts]
            return _HeadLoss.apply(fmap, cont([h.float() for h in heatmaps]), cont([a.float() for a in anno_boxes]),
                                   cont([i.long() for i in inds]), cont([m.to(torch.uint8) for m in masks]), norm.float().contiguous(),
                                   self.code_weights, self.loss_bbox_weight)
        cls_norm = norm[:n_task].clamp(min=1)
        box_norm = norm[n_task:].clamp(min=1e-4)
        code_weights = self.code_weights
        total = 0
        for t, preds in enumerate(preds_dicts):
            p = preds[0]
            hm = clip_sigmoid(p['heatmap'].float())
            total = total + gaussian_focal_loss(hm, heatmaps[t]).sum() / cls_norm[t]
            pred_box = torch.cat((p['reg'], p['height'], p['dim'], p['rot'], p['vel']), 1).float()
            pred_box = pred_box.permute(0, 2, 3, 1).reshape(pred_box.size(0), -1, pred_box.size(1))
            pred_box = pred_box.gather(1, inds[t].unsqueeze(2).expand(-1, -1, pred_box.size(2)))
            target = anno_boxes[t]
            m = masks[t].unsqueeze(2).float() * (~torch.isnan(target)).float()
            w = m * code_weights
            total = total + self.loss_bbox_weight * ((pred_box - target.nan_to_num(0)).abs() * w).sum() / box_norm[t]
        return total


def gaussian_focal_loss(pred, target, alpha=2.0, gamma=4.0, eps=1e-12):
    """mmdet GaussianFocalLoss (elementwise)."""
    pos_weights = target.eq(1)
    neg_weights = (1 - target).pow(gamma)
    pos_loss = -(pred + eps).log() * (1 - pred).pow(alpha) * pos_weights
    neg_loss = -(1 - pred + eps).log() * pred.pow(alpha) * neg_weights
    return pos_loss + neg_loss
