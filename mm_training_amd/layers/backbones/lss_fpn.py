"""LSSFPN -- host-side mirror of layers/backbones/lss_fpn.py of the reference: same
constructor and ``forward(sweep_imgs, mats_dict, depth_oracle, timestamps,
is_return_depth)`` signature and outputs (lss_fpn.py:252-254, :469-529), so
models/bev_depth.py is a drop-in.

The hot path is HIP (libmmt_hip.so): frustum geometry + quantise
(``frustum_geometry``, replacing :328-361 + :461-462), lift + channels-last layout
(``lift_features``, replacing :441-463) and ``voxel_pooling`` (:463-464).  The conv
nets (ResNet / SECONDFPN / DepthNet) are plain PyTorch modules (MIOpen).
Reference quirks kept: voxel_num truncation (:286-289), depth softmax taken BEFORE
the per-camera un-flip of depth_feature (:423-425), BDA not applied in
get_geometry (:355-360), context_se constructed but never called (:183, :240-248).
"""
import os

import torch
from torch import nn

from ...ops.bev_geometry import (camera_form_supported, depth_softmax, exclusive_cache_used, frustum_axes, frustum_geometry, lift_features, lift_splat,
                                 lift_splat_camera, lift_splat_plan, new_column_summary, new_exclusive_cache, new_plan_cache, plan_form_supported, plan_prepare)
from ...ops.bn_relu import ConvBNAct, bn_act
from ...ops.voxel_pooling import VoxelPoolingPlan, voxel_pooling, voxel_pooling_bf16, voxel_pooling_planned
from ..nets import BasicBlock, DeformConv2dPack, ResNet, SECONDFPN

__all__ = ['LSSFPN']


class _ASPPModule(nn.Module):
    def __init__(self, inplanes, planes, kernel_size, padding, dilation):
        super().__init__()
        self.atrous_conv = nn.Conv2d(inplanes, planes, kernel_size, 1, padding, dilation, bias=False)
        self.bn = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU()
        nn.init.kaiming_normal_(self.atrous_conv.weight)

    def forward(self, x):
        return bn_act(self.bn, self.atrous_conv(x))


class ASPP(nn.Module):
    """lss_fpn.py:47-117."""

    def __init__(self, inplanes, mid_channels=256):
        super().__init__()
        self.aspp1 = _ASPPModule(inplanes, mid_channels, 1, 0, 1)
        self.aspp2 = _ASPPModule(inplanes, mid_channels, 3, 6, 6)
        self.aspp3 = _ASPPModule(inplanes, mid_channels, 3, 12, 12)
        self.aspp4 = _ASPPModule(inplanes, mid_channels, 3, 18, 18)
        self.global_avg_pool = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)),
                                             nn.Conv2d(inplanes, mid_channels, 1, bias=False),
                                             nn.BatchNorm2d(mid_channels), nn.ReLU())
        self.conv1 = nn.Conv2d(mid_channels * 5, mid_channels, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid_channels)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(0.5)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)

    def forward(self, x):
        # bilinear upsampling of a 1x1 map with align_corners=True (lss_fpn.py:96-99) is a broadcast
        x5 = self.global_avg_pool(x).expand(-1, -1, x.shape[2], x.shape[3])
        branches = (self.aspp1(x), self.aspp2(x), self.aspp3(x), self.aspp4(x), x5)
        if branches[0].is_contiguous(memory_format=torch.channels_last) and not branches[0].is_contiguous():
            # the broadcast 1x1 branch has no memory format of its own and would make cat() pick NCHW,
            # i.e. a 2560-channel layout conversion in front of the next convolution
            branches = tuple(b.contiguous(memory_format=torch.channels_last) for b in branches)
        x = torch.cat(branches, 1)
        return self.dropout(bn_act(self.bn1, self.conv1(x)))


class SELayer(nn.Module):
    """lss_fpn.py:144-157."""

    def __init__(self, channels):
        super().__init__()
        self.conv_reduce = nn.Conv2d(channels, channels, 1)
        self.act1 = nn.ReLU()
        self.conv_expand = nn.Conv2d(channels, channels, 1)
        self.gate = nn.Sigmoid()

    def forward(self, x, x_se):
        return x * self.gate(self.conv_expand(self.act1(self.conv_reduce(x_se))))


class DepthNet(nn.Module):
    """lss_fpn.py:160-248."""

    def __init__(self, in_channels, mid_channels, context_channels, depth_channels):
        super().__init__()
        self.reduce_conv = ConvBNAct(nn.Conv2d(in_channels, mid_channels, 3, 1, 1),
                                         nn.BatchNorm2d(mid_channels), nn.ReLU(inplace=True))
        self.context_conv = nn.Conv2d(mid_channels, context_channels, 1)
        self.depth_se = nn.Identity()
        self.context_se = SELayer(mid_channels)      # has parameters, never called (reference quirk)
        self.depth_conv = nn.Sequential(
            BasicBlock(mid_channels, mid_channels), BasicBlock(mid_channels, mid_channels),
            BasicBlock(mid_channels, mid_channels), ASPP(mid_channels, mid_channels),
            DeformConv2dPack(mid_channels, mid_channels, kernel_size=3, padding=1, groups=4),
            nn.Conv2d(mid_channels, depth_channels, 1))
        # the outputs that are read more than once leave as aliases, their gradients meet inside the fused BatchNorm backward
        # (ops/bn_relu.py::bn_act, fork): reduce_conv -> context_conv / the first block's convolution / its identity
        self.reduce_conv.fork = 3
        self.depth_conv[0].fork_output = self.depth_conv[1].fork_output = 2

    def forward_parts(self, x, mats_dict=None):
        """(depth logits [BN, D, fH, fW], context [BN, C, fH, fW]): the two halves of forward()'s concatenation.  LSSFPN slices
        the concatenation apart again two lines later (lss_fpn.py:423, :441-443), so the mirror asks for the halves and the
        cat (two copies, and two slice-gradient copies in backward) never runs in the step."""
        x = self.reduce_conv(x)
        xc, xd = (x[0], x[1:]) if isinstance(x, tuple) else (x, x)
        context = self.context_conv(xc)
        depth = self.depth_conv(self.depth_se(xd))
        return depth, context

    def forward(self, x, mats_dict=None):
        return torch.cat(self.forward_parts(x, mats_dict), 1)


class LSSFPN(nn.Module):
    def __init__(self, x_bound, y_bound, z_bound, d_bound, final_dim, downsample_factor,
                 output_channels, img_backbone_conf, img_neck_conf, depth_net_conf):
        super().__init__()
        self.downsample_factor = downsample_factor
        self.d_bound = d_bound
        self.final_dim = final_dim
        self.output_channels = output_channels
        # True (default whenever the kernels support the channel count): lift + voxel_pooling run as ONE fused
        # kernel pair: the [B,N,D,fH,fW,C] tensor of lss_fpn.py:441-463 -- 606 MB
        # written and read twice per step at cfg2 -- is never materialised.  Same result up to fp32 summation
        # order.  False: the reference's op sequence (lift -> voxel_pooling) on the drop-in ops.
        self.fused_lift_splat = output_channels % 16 == 0 and output_channels <= 256
        # Storage type of the hot-path operands: "bf16" keeps depth / context (fused path) or the
        # lifted feature matrix and its gradient (unfused path) in bf16; products and sums stay fp32, the BEV map is fp32.
        self.hot_path_dtype = "f32"
        # True (default): the fused kernels compute every point's voxel index themselves from the camera matrices and the
        # frustum axes (SURVEY section 8 rows f1 + f3: lss_fpn.py:328-361 and :461-462 folded into :441-464) -- no geom tensor
        # is written or read and no geometry kernel runs in the step.  False (or a shape / frustum the camera-form kernels do
        # not take): mmt_frustum_geometry writes geom and the geom form of the same kernels reads it.  Same cells bit for bit.
        self.camera_form = os.environ.get("MMT_LSS_GEOM_FORM", "0") != "1"
        # Backward kernel of the fused path: "ray" (per-pixel walk, any geometry at the same speed), "column" (matrix cores:
        # two small GEMMs per image column, the faster one while the pixels of a column share their BEV cell) or "auto":
        # column while fewer than 1 % of the kept points leave their column's cell.  With a mats_dict['calibration_id'] that
        # share is measured ONCE per calibration from the geometry (one scalar read back, never inside a graph capture, never
        # in the steady state); without one the column kernel itself counts the points it had to handle one by one
        # (`column_stats` of mmt_lss_splat_backward_cam) and the module reads the counters back LAZILY -- an asynchronous copy
        # polled on later steps, no synchronisation -- falls back to the ray walk while the share is above 1 % and probes the
        # column kernel again every `column_probe_period` steps, so a loader that switches rigs is followed.
        self.lift_splat_backward = "auto"
        self.column_probe_period = 256
        self._column_backward_choice = {}
        self._column_adaptive = None   # state of the lazy read-back (see _adaptive_column_choice)
        self._plan_cache = {}     # calibration_id -> VoxelPoolingPlan (see _forward_single_sweep)
        self._combine_cache = {}  # calibration_id -> camera matrices sensor2ego @ inverse(intrin)
        self._summary_cache = {}  # calibration_id -> column summary of the camera form (8 bytes per 16 points; ops/bev_geometry.py)
        # exclusive-cell caches of the camera-form forward, (device, cameras, stream) -> int32 tensor: which BEV cells a single
        # run reaches, learnt on the device per calibration (no id needed; include/mmt_hip.h `exclusive_cache`).
        # MMT_LSS_EXCL_SLOTS calibrations (default 1024: 64 MiB on a 128 x 128 map), 0 switches it off.
        self._excl_caches = {}
        self.exclusive_slots = int(os.environ.get("MMT_LSS_EXCL_SLOTS", "1024"))
        # True (default): the camera-form FORWARD runs in its plan form (ops/bev_geometry.py::lift_splat_plan, csrc/lift_splat_plan.hip):
        # output-stationary on a per-calibration plan the library learns on the device -- no zero fill, no atomics, every element
        # of the map written once, bit-identical from step to step.  The lookup of the batch's calibrations (mmt_lss_plan_prepare)
        # depends on the matrices only and is issued at the top of the sweep, in front of the image backbone.  Plan caches,
        # (device, cameras, stream) -> uint8 tensor, MMT_LSS_PLAN_SLOTS calibrations each (default 16, at least the batch size);
        # MMT_LSS_PLAN=0 keeps the ray walks.  A rig whose plans do not fit (served by the kernel's slow brute-force path) is
        # noticed through the cache's counters, read back lazily, and sent back to the ray walks.
        self.plan_form = os.environ.get("MMT_LSS_PLAN", "1") != "0"
        self.plan_slots = int(os.environ.get("MMT_LSS_PLAN_SLOTS", "16"))
        self.plan_named_caches = int(os.environ.get("MMT_LSS_PLAN_NAMED", "8"))    # caches kept for batches that name their calibrations
        self._plan_caches = {}
        self._plan_watch = None
        self._frustum_version = 0      # bumped by _refresh_frustum_axes: verdicts left in a plan cache belong to the axes they were looked up with
        rows = [x_bound, y_bound, z_bound]
        # lss_fpn.py:278-289, same expressions (Python doubles -> fp32 / truncating int64)
        self.register_buffer('voxel_size', torch.Tensor([row[2] for row in rows]))
        self.register_buffer('voxel_coord', torch.Tensor([row[0] + row[2] / 2.0 for row in rows]))
        self.register_buffer('voxel_num', torch.LongTensor([(row[1] - row[0]) / row[2] for row in rows]))
        self.register_buffer('frustum', self.create_frustum())
        # the same frustum in pixel-major order [fH, fW, D, 4]: the fused lift-splat kernels read geometry and depth per
        # (pixel, depth range) -- the order a channels_last depth tensor has -- see ops/bev_geometry.py::lift_splat
        self.register_buffer('frustum_pixel_major', self.frustum.permute(1, 2, 0, 3).contiguous(), persistent=False)
        # the three axes the frustum is the outer product of (create_frustum: u per column, v per row, d per bin, w = 1):
        # what the camera form of the fused kernels takes instead of a geom tensor
        axes = frustum_axes(self.frustum)
        self._has_frustum_axes = axes is not None
        for name, t in zip(('frustum_u', 'frustum_v', 'frustum_d'), axes if axes is not None else (torch.zeros(1),) * 3):
            self.register_buffer(name, t.clone(), persistent=False)
        self.depth_channels = self.frustum.shape[0]
        # host copies: no device->host sync per step (the reference indexes a CUDA tensor)
        self._voxel_num_host = [int(v) for v in self.voxel_num]
        self._voxel_size_host = [float(v) for v in self.voxel_size]
        self._voxel_coord_host = [float(v) for v in self.voxel_coord]

        bb = {k: v for k, v in dict(img_backbone_conf).items() if k not in ('type', 'init_cfg')}      # (no checkpoints here: random init)
        self.img_backbone = ResNet(**bb)
        nk = {k: v for k, v in dict(img_neck_conf).items() if k != 'type'}
        self.img_neck = SECONDFPN(**nk)
        self.depth_net = self._configure_depth_net(depth_net_conf)

    def _refresh_frustum_axes(self):
        """(Re-)derive what the camera form reads -- the frustum's three axes and its pixel-major copy -- from the `frustum`
        buffer.  `frustum` is persistent like the reference's (lss_fpn.py:291): a checkpoint may carry another one than the
        constructor built, and get_geometry uses the loaded values; so must the camera form."""
        fr = self.frustum
        self.frustum_pixel_major = fr.permute(1, 2, 0, 3).contiguous()
        axes = frustum_axes(fr)
        self._has_frustum_axes = axes is not None
        if axes is not None:
            self.frustum_u, self.frustum_v, self.frustum_d = (t.clone() for t in axes)
        self.depth_channels = fr.shape[0]
        # what was derived from the old frustum per calibration must not be served for the new one (the plan caches need no
        # clearing: the library signs them with the axes' contents and empties them itself)
        self._summary_cache.clear(); self._plan_cache.clear(); self._column_backward_choice.clear(); self._excl_caches.clear()
        self._column_adaptive = None
        self._frustum_version = getattr(self, "_frustum_version", 0) + 1      # (verdicts left in a plan cache are those of the old axes)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        if prefix + "frustum" in state_dict:
            self._refresh_frustum_axes()          # (a frustum that is no outer product of three axes: geom form from now on)

    def _configure_depth_net(self, depth_net_conf):
        return DepthNet(depth_net_conf['in_channels'], depth_net_conf['mid_channels'],
                        self.output_channels, self.depth_channels)

    def create_frustum(self):
        """lss_fpn.py:308-326 (same torch calls, evaluated on the host at init)."""
        ogfH, ogfW = self.final_dim
        fH, fW = ogfH // self.downsample_factor, ogfW // self.downsample_factor
        d_coords = torch.arange(*self.d_bound, dtype=torch.float).view(-1, 1, 1).expand(-1, fH, fW)
        D = d_coords.shape[0]
        x_coords = torch.linspace(0, ogfW - 1, fW, dtype=torch.float).view(1, 1, fW).expand(D, fH, fW)
        y_coords = torch.linspace(0, ogfH - 1, fH, dtype=torch.float).view(1, fH, 1).expand(D, fH, fW)
        return torch.stack((x_coords, y_coords, d_coords, torch.ones_like(d_coords)), -1).contiguous()

    def camera_matrices(self, sensor2ego_mat, intrin_mat, cache_key=None):
        """combine = sensor2ego @ inverse(intrin) [B, N, 4, 4] fp32 (lss_fpn.py:339-352), the only per-step geometry
        operand of the hot path; reused while `cache_key` (from mats_dict['calibration_id']) repeats."""
        if cache_key is not None:
            hit = self._combine_cache.get(cache_key)
            if hit is not None:
                return hit
        with torch.autocast("cuda", enabled=False):   # the integer index path is fp32 whatever the AMP mode
            combine = sensor2ego_mat.float().matmul(torch.linalg.inv_ex(intrin_mat.float())[0]).contiguous()
        if cache_key is not None and not torch.cuda.is_current_stream_capturing():
            if len(self._combine_cache) >= 64:
                self._combine_cache.pop(next(iter(self._combine_cache)))
            self._combine_cache[cache_key] = combine
        return combine

    def get_geometry_voxels(self, sensor2ego_mat, intrin_mat, bda_mat=None, pixel_major=False, combine=None):
        """get_geometry (lss_fpn.py:328-361) fused with the quantise (:461-462):
        returns int32 voxel coordinates [B,N,D,fH,fW,3] ([B,N,fH,fW,D,3] with pixel_major: same values, the
        order the fused kernels read).  bda_mat is ignored like in the reference (:355-360)."""
        if combine is None:
            combine = self.camera_matrices(sensor2ego_mat, intrin_mat)
        fr = self.frustum_pixel_major if pixel_major else self.frustum
        return frustum_geometry(fr.contiguous(), combine, self._voxel_coord_host, self._voxel_size_host)

    def _exclusive_cache_for(self, batch_size, num_cams, fH, fW, device):
        """The exclusive-cell cache of this (device, cameras, stream), or None where the library would ignore it: only the
        forward kernels that take a cache (mmt_lss_exclusive_cache_used: the register walk and, since round 4, the block
        walk) are handed one -- other shapes do not pay its memory."""
        if self.exclusive_slots <= 0 or not exclusive_cache_used(batch_size, num_cams, self.depth_channels, fH, fW, self.output_channels):
            return None
        key = (str(device), int(num_cams), torch.cuda.current_stream(device).cuda_stream)      # calls sharing a cache are stream-ordered
        cache = self._excl_caches.get(key)
        if cache is None:
            if torch.cuda.is_current_stream_capturing():
                return None                             # (allocate outside a capture: the cache must outlive the graph's pool)
            if len(self._excl_caches) >= 4:             # a handful of streams, not one cache per stream handle ever seen
                self._excl_caches.pop(next(iter(self._excl_caches)))
            # cap the table at ~256 MiB: fewer slots on a large BEV grid (a slot holds 4 bytes per cell)
            nx, ny = self._voxel_num_host[:2]
            slots = max(16, min(self.exclusive_slots, (256 << 20) // max(4 * nx * ny, 1)))
            cache = self._excl_caches[key] = new_exclusive_cache(num_cams, self._voxel_num_host, device, slots)
        return cache

    def _plan_cache_for(self, batch_size, num_cams, fH, fW, device, calib_key=None):
        """The plan cache of this (device, cameras, stream[, calibration id]), or None: plan form off, a shape it does not take, or
        a capture is running and the cache does not exist yet (it must outlive the graph's pool: warm up eagerly first).
        A batch that NAMES its calibrations (calib_key) gets a cache of its own, sized for just that batch: nothing else ever
        replaces its slots or its verdicts, so after the first lookup the forward can go by what is in the cache for as long
        as the module keeps it (the `plan_named_caches` most recently used ids; anonymous batches share one cache of
        `plan_slots` calibrations per (device, cameras, stream) and are looked up every call)."""
        if not self.plan_form or not plan_form_supported(batch_size, num_cams, self.depth_channels, fH, fW, self.output_channels, self._voxel_num_host):
            return None
        key = (str(device), int(num_cams), torch.cuda.current_stream(device).cuda_stream, calib_key)    # calls sharing a cache are stream-ordered
        cache = self._plan_caches.get(key)
        if cache is not None and getattr(cache, "_mmt_slots", 0) < batch_size:
            cache = None                                   # (a larger batch than the cache was sized for)
        if cache is None:
            if torch.cuda.is_current_stream_capturing():
                return None
            named = [k for k in self._plan_caches if k[3] is not None]
            if calib_key is not None and len(named) >= self.plan_named_caches:
                self._plan_caches.pop(named[0])
            elif calib_key is None and len(self._plan_caches) - len(named) >= 4:
                self._plan_caches.pop(next(k for k in self._plan_caches if k[3] is None))
            slots = max(int(batch_size), 2) if calib_key is not None else max(int(self.plan_slots), int(batch_size), 2)
            cache = self._plan_caches[key] = new_plan_cache(num_cams, self.depth_channels, fH, fW, self._voxel_num_host, device, slots)
            cache._mmt_slots = slots
        elif calib_key is not None:
            self._plan_caches[key] = self._plan_caches.pop(key)       # most recently used last
        return cache

    def plan_cache_counters(self):
        """dict(hit, learnt, brute, resets, calls, slots, stale) summed over this module's plan caches (synchronises: outside timed regions)."""
        from mm_training_amd.ops.bev_geometry import plan_cache_counters
        tot = dict(hit=0, learnt=0, brute=0, resets=0, calls=0, slots=0, stale=0)
        for cache in self._plan_caches.values():
            for k, v in plan_cache_counters(cache).items():
                tot[k] += v
        return tot

    def _watch_plan_cache(self, cache):
        """Lazy read-back of the plan cache's header (an asynchronous copy polled on later calls, no synchronisation): when samples
        keep being served by the brute-force path -- a rig whose plans overflow their slots -- the module goes back to the ray
        walks, which take any geometry at their own speed."""
        if torch.cuda.is_current_stream_capturing() or os.environ.get("MMT_LSS_PLAN_WATCH", "1") == "0":
            return
        w = self._plan_watch
        if w is None or w["dev"] != cache.device:
            w = self._plan_watch = dict(dev=cache.device, host=torch.zeros(64, dtype=torch.uint8).pin_memory(), event=torch.cuda.Event(),
                                        pending=False, calls=0, last=None, strikes=0)
        if w["pending"] and w["event"].query():
            words = w["host"].view(torch.int32)
            brute, calls = int(words[10]), int(words[12])           # CacheHeader: hits [8], built [9], brute [10], resets [11], calls [12], stale [13]
            if int(words[13]) > 0:
                raise RuntimeError("LSSFPN: a plan-form forward found a verdict that no lookup of its batch had left in the plan cache "
                                   f"({int(words[13])} sample-forwards wrote no output); the pooled maps of those steps are invalid")
            if w["last"] is not None and calls > w["last"][1]:
                w["strikes"] = w["strikes"] + 1 if brute > w["last"][0] else 0
                if w["strikes"] >= 2:
                    import warnings
                    warnings.warn("LSSFPN: the plan form keeps serving calibrations through its brute-force path (their plans overflow the "
                                  "cache's slots); falling back to the ray-walk forward")
                    self.plan_form = False
            w["last"], w["pending"] = (brute, calls), False
        w["calls"] += 1
        if not w["pending"] and w["calls"] % 16 == 0:
            off = (-cache.data_ptr()) % 256
            w["host"].copy_(cache[off:off + 64], non_blocking=True)
            w["event"].record()
            w["pending"] = True

    def exclusive_cache_counters(self):
        """(hits, learning calls, misses) summed over this module's exclusive-cell caches: per sample and forward call, whether
        the cache was used (the calibration is known: stage 3), still being learnt (stages 1 / 2) or not found (include/mmt_hip.h
        `exclusive_cache`, header words 32..34).  Reads the device: call it outside the timed region."""
        tot = [0, 0, 0]
        for cache in self._excl_caches.values():
            for i, v in enumerate(cache[32:35].tolist()):
                tot[i] += int(v)
        return dict(hit=tot[0], learning=tot[1], miss=tot[2])

    def _adaptive_column_choice(self, device):
        """"auto" without a calibration id: (use the column kernel?, the counters it accumulates into)."""
        st = self._column_adaptive
        if (st is None or st["stats"].device != device) and torch.cuda.is_current_stream_capturing():
            # no state yet and a capture is running: creating it here would allocate pinned host memory under capture (which
            # invalidates a capture in the default global mode) and put the counters into the graph's private pool.  The
            # captured step takes the ray walk (correct for any rig) without counters; warm up eagerly first to get the column kernel.
            return False, None
        if st is None or st["stats"].device != device:
            from mm_training_amd._lib import LSS_STATS_SLOTS
            st = self._column_adaptive = dict(stats=torch.zeros(2 * LSS_STATS_SLOTS, dtype=torch.int64, device=device),
                                              host=torch.zeros(2 * LSS_STATS_SLOTS, dtype=torch.int64).pin_memory(), event=torch.cuda.Event(),
                                              pending=False, last=(0, 0), column=True, ray_left=0, calls=0, share=None)
        if torch.cuda.is_current_stream_capturing():
            return st["column"], st["stats"]              # a captured step keeps the choice it was captured with
        if st["pending"] and st["event"].query():
            mism, kept = int(st["host"][0::2].sum()), int(st["host"][1::2].sum())
            dm, dk = mism - st["last"][0], kept - st["last"][1]
            st["last"], st["pending"] = (mism, kept), False
            if dk > 0:
                st["share"] = dm / dk
                if st["share"] >= 0.01:
                    st["column"], st["ray_left"] = False, int(self.column_probe_period)
        if not st["column"]:
            st["ray_left"] -= 1
            if st["ray_left"] <= 0:
                st["column"] = True                       # probe again: the column kernel reports on the next backward
        st["calls"] += 1
        if st["column"] and not st["pending"] and st["calls"] % 4 == 0:
            st["host"].copy_(st["stats"], non_blocking=True)     # sees every backward enqueued before this forward
            st["event"].record()
            st["pending"] = True
        return st["column"], st["stats"]

    def _use_column_backward(self, geom_pixel_major, calib_id, device=None):
        """(column kernel?, column_stats tensor or None).  geom_pixel_major: the geometry, or a callable producing it (only
        evaluated for the one measurement per calibration id)."""
        if self.lift_splat_backward != "auto":
            return self.lift_splat_backward == "column", None
        if calib_id is None:
            if not torch.is_grad_enabled():
                return False, None                             # nothing to decide for
            if device is None:
                device = self.frustum.device
            return self._adaptive_column_choice(device)
        choice = self._column_backward_choice.get(calib_id)
        if choice is None:
            if torch.cuda.is_current_stream_capturing() or not torch.is_grad_enabled():
                return False, None                             # nothing to decide for (or no way to read a scalar back)
            from mm_training_amd.ops.bev_geometry import column_mismatch_fraction
            geom = geom_pixel_major() if callable(geom_pixel_major) else geom_pixel_major
            choice = bool(column_mismatch_fraction(geom, self._voxel_num_host, pixel_major=True).item() < 0.01)
            if len(self._column_backward_choice) >= 64:
                self._column_backward_choice.pop(next(iter(self._column_backward_choice)))
            self._column_backward_choice[calib_id] = choice
        return choice, None

    def get_cam_feats(self, imgs):
        """[B, S, N, 3, H, W] images -> [B, S, N, C', fH, fW] neck features (lss_fpn.py:363-379)."""
        lead = imgs.shape[:3]
        feats = self.img_neck(self.img_backbone(imgs.reshape(-1, *imgs.shape[3:])))[0]
        return feats.view(*lead, *feats.shape[1:])

    def _forward_single_sweep(self, sweep_index, sweep_imgs, mats_dict, depth_oracle, is_return_depth=False):
        batch_size, num_sweeps, num_cams = sweep_imgs.shape[:3]
        # plan form of the fused forward: its lookup of the batch's calibrations depends on the matrices only -- issue it now, in
        # front of the image backbone, so that the forward kernel finds the verdicts ready
        plan_cache = plan_combine = pending_lookup = None
        if (self.fused_lift_splat and self.camera_form and self._has_frustum_axes and self.plan_form and isinstance(mats_dict, dict)
                and os.environ.get("MMT_LIFT_SPLAT_TILES", "0") != "1" and os.environ.get("MMT_LIFT_SPLAT_V1", "0") != "1"):
            fH_, fW_ = self.frustum_v.numel(), self.frustum_u.numel()
            calib0 = mats_dict.get('calibration_id', None)
            ckey0 = None if calib0 is None else (calib0, sweep_index, batch_size, num_cams, str(sweep_imgs.device))
            if camera_form_supported(batch_size, num_cams, self.depth_channels, fH_, fW_, self.output_channels):
                plan_cache = self._plan_cache_for(batch_size, num_cams, fH_, fW_, sweep_imgs.device, ckey0)
            if plan_cache is not None:
                plan_combine = self.camera_matrices(mats_dict['sensor2ego_mats'][:, sweep_index, ...], mats_dict['intrin_mats'][:, sweep_index, ...], ckey0)
                # The verdicts of the last prepared batch stay in the cache, and a batch that names its calibrations
                # (mats_dict['calibration_id']) has a cache of its own: when the same ids come again -- the steady state of a
                # loader with a fixed set of rigs -- no lookup is needed at all: neither lss_plan_probe nor lss_plan_build is
                # launched, the forward goes by the verdicts that are there (lss_fpn.py:328-361 has no per-step term for an
                # unchanged calibration either).  Anonymous batches are looked up every call.
                fresh = (ckey0 is not None and getattr(plan_cache, "_mmt_prepared_for", None) == ckey0
                         and getattr(plan_cache, "_mmt_frustum_version", None) == self._frustum_version
                         and not torch.cuda.is_current_stream_capturing())
                if not fresh:
                    lookup = (plan_combine, (self.frustum_u, self.frustum_v, self.frustum_d), self._voxel_num_host, self._voxel_coord_host,
                              self._voxel_size_host, plan_cache)
                    # The lookup rides in the depth softmax's launch (mmt_depth_softmax_forward_plan_prepare: its workgroups in front
                    # of the softmax's grid) -- as a launch of its own it costs 5-10 us of a step whose forward takes 28.
                    # MMT_PLAN_LOOKUP_RIDER=0: the launch of its own, here, in front of the image backbone (A/B).
                    if self.depth_channels <= 512 and os.environ.get("MMT_ATEN_SOFTMAX", "0") != "1" and os.environ.get("MMT_PLAN_LOOKUP_RIDER", "1") != "0":
                        pending_lookup = lookup
                    else:
                        plan_prepare(*lookup)
                    plan_cache._mmt_prepared_for = ckey0
                    plan_cache._mmt_frustum_version = self._frustum_version
                self._watch_plan_cache(plan_cache)
        img_feats = self.get_cam_feats(sweep_imgs)
        # the key frame (:389): every caller hands over ONE sweep here (forward slices sweep_imgs[:, k:k+1]); squeezing that axis is
        # a view in both directions, while `img_feats[:, 0]` costs the backward a zero-fill + a strided copy of the whole neck
        # output (select_backward: 16 + 47 us per step at BASELINE configs[3])
        source_features = img_feats.squeeze(1) if img_feats.shape[1] == 1 else img_feats[:, 0, ...]
        feats_in = source_features.reshape(batch_size * num_cams, *source_features.shape[2:])
        D, C = self.depth_channels, self.output_channels
        if hasattr(self.depth_net, "forward_parts"):
            depth_logits, context = self.depth_net.forward_parts(feats_in, mats_dict)
        else:       # any depth net with the reference's output: depth | context on the channel axis
            depth_feature = self.depth_net(feats_in, mats_dict)
            depth_logits, context = depth_feature[:, :D], depth_feature[:, D:D + C]
        # :423 + :427-438 as ONE launch, pixel-major (ops/bev_geometry.py::depth_softmax): `depth` is the plain softmax taken
        # BEFORE the per-camera un-flip (reference quirk, :423-425), `depth_used` what the lift multiplies with -- the oracle's
        # rows on foreground pixels (fg_mask = max over bins > 0), the softmax elsewhere.  bf16 storage (row g1): the fused
        # kernels' bf16 operand comes out of the same launch.
        used_bf16 = self.hot_path_dtype == "bf16" and self.fused_lift_splat
        if D <= 512 and os.environ.get("MMT_ATEN_SOFTMAX", "0") != "1":
            depth, depth_used = depth_softmax(depth_logits, depth_oracle, torch.bfloat16 if used_bf16 else torch.float32, plan_lookup=pending_lookup)
        else:       # (more bins than the kernel takes; or the A/B switch of bench.py --aten-softmax)
            if pending_lookup is not None:
                plan_prepare(*pending_lookup)
            depth = depth_logits.softmax(1)
            depth_used = depth
            if depth_oracle is not None:
                fg_mask = (torch.max(depth_oracle, dim=1, keepdim=True).values > 0.0)
                depth_used = torch.where(fg_mask, depth_oracle.to(depth.dtype), depth)
        flipped = mats_dict.get('flipped', None) if isinstance(mats_dict, dict) else None
        if flipped is not None:
            # hflip per camera (:425) -- after the softmax, so only the context half of depth_feature is still read
            fl = torch.as_tensor(flipped, device=context.device).view(-1, 1, 1, 1).bool()
            context = torch.where(fl, context.flip(-1), context)
        # SURVEY section 8 row f3 (cached sort): geometry, quantisation and the sort of the points by
        # BEV cell depend only on the calibration (BDA is not applied here, lss_fpn.py:355-360).  A data
        # pipeline that knows its calibration passes a hashable host-side mats_dict['calibration_id']
        # (e.g. the vehicle / log id of the batch); while it repeats, the plan is reused and neither
        # get_geometry nor the in-kernel sort run again.  Without the key nothing is cached.
        calib_id = mats_dict.get('calibration_id', None) if isinstance(mats_dict, dict) else None
        plan = None
        if calib_id is not None and not self.fused_lift_splat:
            key = (calib_id, sweep_index, batch_size, num_cams, str(context.device))
            plan = self._plan_cache.get(key)
            if plan is None:
                geom_xyz = self.get_geometry_voxels(mats_dict['sensor2ego_mats'][:, sweep_index, ...],
                                                    mats_dict['intrin_mats'][:, sweep_index, ...],
                                                    mats_dict.get('bda_mat', None))
                plan = VoxelPoolingPlan(geom_xyz.reshape(batch_size, -1, 3), self._voxel_num_host)
                if len(self._plan_cache) >= 8:          # a handful of rigs, not an unbounded map
                    self._plan_cache.pop(next(iter(self._plan_cache)))
                self._plan_cache[key] = plan
        fused_kind = None
        if plan is None and self.fused_lift_splat:
            fH, fW = context.shape[-2:]
            pixel_major_ok = fH <= 512 and os.environ.get("MMT_LIFT_SPLAT_V1", "0") != "1"
            ckey = None if calib_id is None else (calib_id, sweep_index, batch_size, num_cams, str(context.device))
            combine = plan_combine if plan_combine is not None else self.camera_matrices(
                mats_dict['sensor2ego_mats'][:, sweep_index, ...], mats_dict['intrin_mats'][:, sweep_index, ...], ckey)
            if (self.camera_form and self._has_frustum_axes and pixel_major_ok and os.environ.get("MMT_LIFT_SPLAT_TILES", "0") != "1"
                    and camera_form_supported(batch_size, num_cams, self.depth_channels, fH, fW, self.output_channels)):
                fused_kind = "camera"
            else:
                fused_kind = "geom_pm" if pixel_major_ok else "geom"     # fH > 512 / first-generation kernels: frustum order
                geom_xyz = self.get_geometry_voxels(None, None, pixel_major=pixel_major_ok, combine=combine)
        if plan is not None:
            feats = lift_features(depth_used.float(), context.float())
            feature_map = voxel_pooling_planned(plan, feats.view(batch_size, -1, feats.shape[-1]))
        elif fused_kind is not None:
            bf16 = self.hot_path_dtype == "bf16"
            dep_in, ctx_in = (depth_used.bfloat16(), context.bfloat16()) if bf16 else (depth_used.float(), context.float())
            if fused_kind == "camera":
                col_bwd, stats = self._use_column_backward(
                    lambda: self.get_geometry_voxels(None, None, pixel_major=True, combine=combine), calib_id, context.device)
                if plan_cache is not None and tuple(context.shape[-2:]) == (self.frustum_v.numel(), self.frustum_u.numel()):
                    feature_map = lift_splat_plan(combine, (self.frustum_u, self.frustum_v, self.frustum_d), dep_in, ctx_in, self._voxel_num_host,
                                                  self._voxel_coord_host, self._voxel_size_host, plan_cache, prepared=True,
                                                  column_backward=col_bwd, column_stats=stats)
                    return (feature_map, depth) if is_return_depth else feature_map
                # with a calibration id the geometry's column summary is kept too: later forwards read 0.5 byte per point
                # instead of computing the cells (without one every forward writes a fresh summary for its backward)
                summary, cached, keep = None, False, False
                if ckey is not None and not torch.cuda.is_current_stream_capturing():
                    summary = self._summary_cache.get(ckey)
                    cached = summary is not None
                    if summary is None:
                        summary, keep = new_column_summary(batch_size, num_cams, self.depth_channels, fH, fW, context.device), True
                feature_map = lift_splat_camera(combine, (self.frustum_u, self.frustum_v, self.frustum_d), dep_in, ctx_in,
                                                self._voxel_num_host, self._voxel_coord_host, self._voxel_size_host,
                                                column_backward=col_bwd, column_stats=stats, summary=summary, summary_cached=cached,
                                                exclusive_cache=self._exclusive_cache_for(batch_size, num_cams, fH, fW, context.device))
                if keep:
                    # only now: the forward that WRITES the summary has been enqueued.  Had it raised (bad shape, MMT_ERR_*,
                    # out of memory), an entry inserted beforehand would have been served as "cached" to the next step and
                    # its uninitialised words read as geometry.
                    if len(self._summary_cache) >= 64:
                        self._summary_cache.pop(next(iter(self._summary_cache)))
                    self._summary_cache[ckey] = summary
            elif fused_kind == "geom_pm":
                col_bwd, _ = self._use_column_backward(geom_xyz, calib_id if calib_id is not None else "_", context.device)
                feature_map = lift_splat(geom_xyz, dep_in, ctx_in, self._voxel_num_host, pixel_major=True, column_backward=col_bwd)
            else:
                feature_map = lift_splat(geom_xyz, dep_in, ctx_in, self._voxel_num_host, pixel_major=False)
        else:
            # lift straight into [B, N, D, fH, fW, C], then the drop-in voxel_pooling
            bf16 = self.hot_path_dtype == "bf16" and self.output_channels % 16 == 0
            feats = lift_features(depth_used.float(), context.float(), torch.bfloat16 if bf16 else torch.float32)
            feats = feats.view(batch_size, num_cams, *feats.shape[1:])
            # geometry AFTER the lift: its small kernels give the 606 MB of non-temporal lift stores time
            # to drain before the pooling kernel starts reading them (bench roofline.avg_ms: see DESIGN 3.1)
            geom_xyz = self.get_geometry_voxels(mats_dict['sensor2ego_mats'][:, sweep_index, ...],
                                                mats_dict['intrin_mats'][:, sweep_index, ...],
                                                mats_dict.get('bda_mat', None))
            feature_map = (voxel_pooling_bf16 if bf16 else voxel_pooling)(geom_xyz, feats, self._voxel_num_host)
        # the reference's `.contiguous()` (:467) would transpose to NCHW; the pooled map is
        # already a dense channels_last tensor, which the BEV convs consume directly
        if is_return_depth:
            return feature_map, depth
        return feature_map

    def forward(self, sweep_imgs, mats_dict, depth_oracle=None, timestamps=None, is_return_depth=False):
        """Key frame with gradients, older sweeps (if any; the aiMotive loader feeds one) without, BEV maps stacked
        on the channel axis (lss_fpn.py:469-529).  Returns the map, or (map, key-frame depth) if is_return_depth."""
        first = self._forward_single_sweep(0, sweep_imgs[:, :1], mats_dict, depth_oracle, is_return_depth=is_return_depth)
        sweeps = sweep_imgs.shape[1]
        if sweeps == 1:
            return first
        bev0, depth0 = first if is_return_depth else (first, None)
        with torch.no_grad():
            older = [self._forward_single_sweep(k, sweep_imgs[:, k:k + 1], mats_dict, depth_oracle) for k in range(1, sweeps)]
        stacked = torch.cat([bev0] + older, 1)
        return (stacked, depth0) if is_return_depth else stacked
