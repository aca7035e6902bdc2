"""Single-node data-parallel training step -- the hot loop of the reference's Lightning
module (exps/mm_training_aim.py:252-289 training_step, :114-215 depth labels / depth
loss, :510-512 normalise, :524-531 optimiser, :619-628 gradient clipping) without
Lightning: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI),
DistributedDataParallel's bucketed all-reduce overlapped with backward on RCCL's stream.

Host-side changes against the reference, none of which alter the arithmetic contract:
  * depth labels are produced by one vectorised scatter-min per batch instead of the
    B x N_cam Python loop + per-pixel "last write wins" map (:122-163); identical when no
    two points fall into the same pixel, deterministic otherwise;
  * no .item() / boolean-mask host syncs in the loss (see bev_depth_head.py mirror).
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import synthetic
from ..models.bev_depth import BEVDepthLiDAR
from ..ops.bn_relu import _supported as bn_supported
from ..ops.bn_relu import bn_act
from ..ops.train_targets import camera_flags_to_device, depth_labels, hflip, normalize_flip_images

IMG_MEAN = (0.485, 0.456, 0.406)
IMG_STD = (0.229, 0.224, 0.225)


def camera_rig(rig, B, N, W, H, seed=0):
    """(sensor2ego, intrin) [B, N, 4, 4] CPU tensors of the benchmark rigs:
      "analytic"        the level 6-camera fan of SURVEY section 8d (mm_training_amd.synthetic.camera_rig), yaw jitter per sample
      "pitched:<deg>"   the same with every camera pitched by <deg> about its own x axis (columns then straddle BEV cells)
      (s2e, K, (h, w))  a given rig (e.g. the reference fixture's real calibration, tests/golden/nusc_rig.npz: cameras that are not
                        level), its intrinsics rescaled from h x w to H x W, the same rig for every sample"""
    if isinstance(rig, (tuple, list)):          # (sensor2ego [N, 4, 4], intrin [N, 4, 4], (image height, image width) the intrinsics are for)
        s2e, K, (h0, w0) = rig
        if N != s2e.shape[0]:
            raise ValueError(f"the rig has {s2e.shape[0]} cameras, the configuration {N}")
        K = torch.as_tensor(K).clone().float()
        K[:, 0, :] *= W / float(w0)             # fx, skew, cx
        K[:, 1, :] *= H / float(h0)             # fy, cy
        return torch.as_tensor(s2e).float().unsqueeze(0).repeat(B, 1, 1, 1).contiguous(), K.unsqueeze(0).repeat(B, 1, 1, 1).contiguous()
    s2e, K = synthetic.camera_rig(B, N, W, H, jitter=0.02, seed=seed)
    if rig.startswith("pitched:"):
        a = math.radians(float(rig.split(":", 1)[1]))
        rx = torch.tensor([[1, 0, 0, 0], [0, math.cos(a), -math.sin(a), 0], [0, math.sin(a), math.cos(a), 0], [0, 0, 0, 1]], dtype=torch.float32)
        s2e = s2e.matmul(rx)
    elif rig != "analytic":
        raise ValueError(f"unknown rig {rig!r} (analytic | pitched:<deg> | a (sensor2ego, intrin, (h, w)) tuple)")
    return s2e, K


def synthetic_batch(cfg, device, seed=0, batch_size=None, rig="analytic"):
    """One batch shaped like collate_aim's output (dataset/src/aimotive_dataset.py:182-231):
    (sweep_imgs [B,1,N,3,H,W] in 0..255, mats dict, pointclouds list[B] of [Ni,F], gt_boxes, gt_labels)."""
    B = batch_size or cfg["batch_size"]
    N = cfg["num_cams"]
    H, W = cfg["final_dim"]
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randint(0, 256, (B, 1, N, 3, H, W), generator=g, dtype=torch.uint8).float()
    s2e, K = camera_rig(rig, B, N, W, H, seed=seed)
    mats = {
        "sensor2ego_mats": s2e.unsqueeze(1).to(device), "intrin_mats": K.unsqueeze(1).to(device),
        "extrinsics": torch.inverse(s2e).unsqueeze(1).to(device),
        "bda_mat": torch.eye(4).repeat(B, 1, 1).to(device),
        "flipped": torch.zeros(B * N, dtype=torch.bool, device=device),
    }
    pcs = [synthetic.lidar_frame(cfg["num_points"], cfg["point_features"], cfg["point_cloud_range"],
                                 num_radar=2000 if cfg["use_radar"] else 0, seed=seed * 1000 + b).to(device)
           for b in range(B)]
    pr = cfg["point_cloud_range"]
    boxes, labels = [], []
    for b in range(B):
        k = cfg["num_boxes"]
        xy = torch.rand(k, 2, generator=g) * torch.tensor([pr[3] - pr[0] - 8, pr[4] - pr[1] - 8]) + torch.tensor([pr[0] + 4, pr[1] + 4])
        z = torch.rand(k, 1, generator=g) * 2 - 2
        lab = torch.randint(0, 4, (k,), generator=g)
        prior = torch.tensor([[1.9, 4.6, 1.7], [2.5, 9.0, 3.2], [0.8, 2.1, 1.5], [0.7, 0.7, 1.75]])[lab]
        dims = prior * (0.9 + 0.2 * torch.rand(k, 3, generator=g))
        yaw = (torch.rand(k, 1, generator=g) * 2 - 1) * math.pi
        vel = torch.randn(k, 2, generator=g)
        boxes.append(torch.cat([xy, z, dims, yaw, vel], 1).to(device))
        labels.append(lab.to(device))
    return imgs.to(device), mats, pcs, boxes, labels


def _batched_bn_counters(model):
    """nn.BatchNorm2d bumps `num_batches_tracked` with one tiny kernel per layer and step (129 launches
    at cfg2).  The counter only matters for momentum=None; keep the state identical but advance all
    counters with ONE foreach launch per step: the modules' own increment is skipped by running
    F.batch_norm directly in training mode."""
    counters = []

    def bn_forward(self, x):
        if self.training and self.track_running_stats and self.momentum is not None:
            if x.is_cuda and torch.is_autocast_enabled() and bn_supported(self, x):
                return bn_act(self, x, relu=False)       # fused fp32 kernels, not MIOpen's NHWC batch norm (ops/bn_relu.py)
            return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, True, self.momentum, self.eps)
        return nn.BatchNorm2d.forward(self, x)

    import types
    for m in model.modules():
        if type(m) is nn.BatchNorm2d and m.track_running_stats and m.momentum is not None:
            m.forward = types.MethodType(bn_forward, m)
            if m.training:                                   # (a frozen layer -- the image backbone's norm1 -- counts nothing)
                counters.append(m.num_batches_tracked)
    return counters


@torch.no_grad()
def augment_images(images, depth_images, stage):
    """exps/mm_training_aim.py:89-112.  Per camera, with probability 1/2 (np.random.uniform(size = b*s*n) > 0.5: the
    reference's own draw from numpy's global generator), mirror the image [.., c, h, w] and its depth-label map
    [b*s*n, fH, fW, D] along w.  Returns (images, depth_images, flips), flips = the host boolean array the reference stores
    in mats['flipped'].  Two launches of mmt_hflip instead of two stacked Python lists of kornia.hflip results.  The
    training step itself runs the fused form (TrainStep.forward_loss); this is the reference's function as a function."""
    b, s, n, c, h, w = images.shape
    if stage != 'train':
        return images, depth_images, np.zeros((b * s * n), dtype=bool)
    flips = np.random.uniform(size=(b * s * n)) > 0.5
    fl = camera_flags_to_device(flips, images.device)
    images = hflip(images.reshape(b * s * n * c, h, w, 1), fl, group=c).view(b, s, n, c, h, w)
    depth_images = hflip(depth_images, fl)
    return images, depth_images, flips


class TrainStep(nn.Module):
    """Owns model + optimiser and runs one optimisation step per call."""

    def __init__(self, cfg, device, world_size=1, lr=None, bucket_cap_mb=64, amp=None):
        super().__init__()
        self.cfg = cfg
        self.device = device
        self.use_cam, self.use_lidar = cfg["use_cam"], cfg["use_lidar"]
        self.model = BEVDepthLiDAR(cfg["backbone_conf"], cfg["head_conf"], cfg["lidar_conf"], is_train_depth=True,
                                   use_cam=self.use_cam, use_lidar=self.use_lidar,
                                   fuse_layer_in_channels=cfg["fuse_layer_in_channels"]).to(device)
        # dense convs consume / produce channels_last: the pooled BEV map already is
        if os.environ.get("MMT_MEMORY_FORMAT", "channels_last") == "channels_last":
            for m in self.model.modules():
                if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                    m.to(memory_format=torch.channels_last)
        if self.use_cam:
            self.model.backbone.hot_path_dtype = cfg.get("hot_path_dtype", "f32")
        self.net = self.model
        # The convolutions' backward (ops/conv_overlap.py).  "deferred": weight gradients on a side HIP stream, joined once at the end
        # of the backward pass (68.6 -> 67.2 ms at configs[3]).  Under DDP the bucket hooks read each gradient inside the backward
        # pass, which would need a join per layer ("pair": measured slower than no overlap), so world size > 1 runs "inline" --
        # one stream, and the module's other duty only: narrow 16-bit convolutions in fp32 (a MIOpen bf16 kernel faults on them).
        self.conv_overlap = None
        overlap = os.environ.get("MMT_CONV_OVERLAP", "deferred")
        # Gradient exchange at world size > 1: "native" = dp/reducer.py (bucketed all-reduce on a communication stream, a few launches
        # per bucket, nothing read on the main stream inside the backward pass -- the deferred weight-gradient stream stays on);
        # "ddp" = torch's DistributedDataParallel (one small kernel per parameter on the main stream; convolutions then run inline)
        self.dp_reducer = os.environ.get("MMT_DP_REDUCER", "native") if world_size > 1 else None
        if overlap == "deferred" and world_size > 1 and self.dp_reducer != "native":
            overlap = "inline"
        if device.type == "cuda" and overlap != "off":
            from ..ops import conv_overlap
            conv_overlap.enable(self.model, overlap)
            self.conv_overlap = overlap
        if world_size > 1 and "MMT_HEAD_STREAMS" not in os.environ:
            # a process has four hardware queues: main + the weight-gradient / packing stream + RCCL's own leave one spare; the task
            # heads' two streams (0.2 ms of the step) would oversubscribe them next to a collective that must make progress
            self.model.head.task_streams = 0
        self.reducer = None
        if world_size > 1 and self.dp_reducer == "native":
            # DepthNet.context_se has parameters that never receive a gradient (lss_fpn.py:183): not registered with the reducer
            import torch.distributed as dist
            from .reducer import GradReducer
            if dist.is_initialized():                        # every rank starts from rank 0's parameters and buffers (what DDP does at construction)
                with torch.no_grad():
                    for t in list(self.model.parameters()) + list(self.model.buffers()):
                        dist.broadcast(t, 0)
            # Where the buckets are packed and the collectives issued: behind the weight gradients on THEIR stream (default), or on a
            # stream of the reducer's own at normal priority (MMT_REDUCER_STREAM=own).  The review of round 4 asked for the latter
            # (a collective queued behind a low-priority stream's backlog starts late); measured on ONE rank over RCCL
            # (profiles/r05_exchange_tax.txt) the extra stream costs the rank 4.2 ms of a 66.7 ms step against 0.9 ms on the
            # weight-gradient stream -- a process has four hardware queues, and main + weight gradients + a communication
            # stream + RCCL's own are a fifth stream's worth of multiplexing.  What bounds the exposed tail instead is the size of
            # the LAST bucket (dp/reducer.py: 6 MB, not 64).
            pack_on = None
            if self.conv_overlap in ("deferred", "pair") and os.environ.get("MMT_REDUCER_STREAM", "side") == "side":
                from ..ops import conv_overlap
                pack_on = conv_overlap.side_stream(device)   # behind the weight gradients, on their stream
            self.reducer = GradReducer(self.model.named_parameters(), world_size, bucket_mb=bucket_cap_mb, ignore=(".context_se.",),
                                       extra_streams=self._gradient_streams, stream=pack_on)
        elif world_size > 1:
            # DepthNet.context_se has parameters that never receive a gradient (lss_fpn.py:183; the
            # reference survives on Lightning's find_unused_parameters=True, which walks the autograd
            # graph on the host every step).  Tell the reducer to ignore exactly those instead: the
            # graph is then static and every bucket's all-reduce (RCCL, on its own stream) starts as
            # soon as its last gradient is produced.
            unused = [n for n, _ in self.model.named_parameters() if ".context_se." in n]
            nn.parallel.DistributedDataParallel._set_params_and_buffers_to_ignore_for_model(self.model, unused)
            self.net = nn.parallel.DistributedDataParallel(
                self.model, device_ids=[device.index] if device.type == "cuda" else None,
                bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True, find_unused_parameters=False,
                static_graph=True,
                # BatchNorm statistics are per rank (no SyncBN in the reference); re-broadcasting
                # rank 0's ~390 buffers before every forward would be one more collective per step
                broadcast_buffers=False)
        bs = cfg["batch_size"]
        self.grad_clip = 2.0
        # exps/mm_training_aim.py:575-608: gradient_clip_val 2 + AdamW.  On the GPU: clip + update as two launches over all
        # parameters (dp/optim.py; MMT_FUSED_OPT=0: torch's clip_grad_norm_ + fused AdamW, which is also what the CPU tests run)
        self.fused_optimizer = device.type == "cuda" and os.environ.get("MMT_FUSED_OPT", "1") != "0"
        if self.fused_optimizer:
            from .optim import ClipAdamW
            self.optimizer = ClipAdamW(self.model.parameters(), lr=lr or 1e-3 / 64 * bs, weight_decay=1e-7, max_norm=self.grad_clip)
        else:
            self.optimizer = torch.optim.AdamW(self.model.parameters(), lr=lr or 1e-3 / 64 * bs, weight_decay=1e-7,
                                               fused=(device.type == "cuda"))
        self.scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, [19, 23])
        db = cfg["backbone_conf"]["d_bound"]
        self.dbound = db
        self.downsample = cfg["backbone_conf"]["downsample_factor"]
        self.depth_channels = len(torch.arange(*db)) if self.use_cam else 0
        self.amp_dtype = torch.bfloat16 if (amp or cfg.get("dtype")) == "bf16" else None
        if self.fused_optimizer and self.amp_dtype is torch.bfloat16 and self.conv_overlap is not None \
                and os.environ.get("MMT_BF16_SHADOWS", "1") != "0":
            # the autocast convolutions' bf16 weights are written by the optimizer step itself (ClipAdamW.make_bf16_shadows), no cast
            # kernel per layer and step (253 launches: 1.05 ms of kernels and ~2 ms of host time per step at BASELINE configs[4],
            # which sits at the host/device crossover).  Three alternating 80-step runs on one box: 41.1-41.3 ms with,
            # 41.5-42.6 ms without
            from ..ops.conv_overlap import OverlapConv2d
            convs = [m.weight for m in self.model.modules() if isinstance(m, OverlapConv2d) and m.weight.requires_grad]
            self.optimizer.make_bf16_shadows(convs)
        # exps/mm_training_aim.py:258: training_step always runs augment_images(..., 'train'); :78 + :259: the depth labels are
        # handed to the model as the depth oracle when use_depth_loss is set (True in exps/conf_aim.py:23 and in the reference's
        # camera + LiDAR config exps/configs/lidar_cam.py:23)
        self.augment = bool(cfg.get("augment_images", True))
        self.pass_depth_labels = bool(cfg.get("use_depth_loss", True))
        self.register_buffer("mean", torch.tensor(IMG_MEAN).view(1, 1, 1, 3, 1, 1), persistent=False)
        self.register_buffer("std", torch.tensor(IMG_STD).view(1, 1, 1, 3, 1, 1), persistent=False)
        self.to(device)
        self._bn_counters = _batched_bn_counters(self.model) if device.type == "cuda" else []

    # ---- exps/mm_training_aim.py:510-512
    def normalize_images(self, sweep_imgs):
        return (sweep_imgs[:, :, :, :3] / 255.0 - self.mean) / self.std

    # ---- exps/mm_training_aim.py:114-163 + :180-215: one HIP op (ops/train_targets.py)
    @torch.no_grad()
    def get_depth_labels(self, images, mats, pointclouds):
        B, S, N, _, H, W = images.shape
        return depth_labels(pointclouds, mats["extrinsics"][:, 0], mats["intrin_mats"][:, 0], mats["bda_mat"],
                            (H, W), self.downsample, self.dbound, self.depth_channels)

    @torch.no_grad()
    def get_depth_labels_torch(self, images, mats, pointclouds):
        """The same labels from vectorised torch ops (scatter-amin) -- kept as the cross-check of
        tests/test_train_targets_gpu.py, not used by the training step."""
        B, S, N, _, H, W = images.shape
        ds = self.downsample
        fH, fW = H // ds, W // ds
        ext = mats["extrinsics"][:, 0]                     # [B,N,4,4] ego -> camera
        K = mats["intrin_mats"][:, 0]
        bda = mats["bda_mat"]
        gt = torch.full((B * N, fH * fW), 1e5, device=images.device)
        for b in range(B):                                 # point counts differ per sample
            xyz = pointclouds[b][:, :3] @ torch.linalg.inv_ex(bda[b, :3, :3])[0].T   # inv_ex: no host sync
            pts = torch.cat([xyz, torch.ones_like(xyz[:, :1])], 1)          # [P,4]
            cam = torch.einsum("nij,pj->npi", ext[b], pts)                   # [N,P,4]
            depth = cam[..., 2]
            proj = torch.einsum("nij,npj->npi", K[b], cam)
            u = proj[..., 0] / proj[..., 2]
            v = proj[..., 1] / proj[..., 2]
            ok = (depth > 1.0) & (u > 1) & (u < W - 1) & (v > 1) & (v < H - 1)
            cell = (v.long().clamp(0, H - 1) // ds) * fW + (u.long().clamp(0, W - 1) // ds)
            d = torch.where(ok, depth, torch.full_like(depth, 1e5))
            gt[b * N:(b + 1) * N].scatter_reduce_(1, cell, d, "amin", include_self=True)
        gt = (gt - (self.dbound[0] - self.dbound[2])) / self.dbound[2]
        gt = torch.where((gt < self.depth_channels) & (gt >= 0.0), gt, torch.zeros_like(gt))
        return F.one_hot(gt.long(), num_classes=self.depth_channels).float().view(-1, self.depth_channels)

    # ---- exps/mm_training_aim.py:165-178
    def get_depth_loss(self, depth_labels, depth_preds):
        depth_preds = depth_preds.permute(0, 2, 3, 1).reshape(-1, self.depth_channels).float()
        fg = (depth_labels.max(1).values > 0.0).float()
        bce = F.binary_cross_entropy(depth_preds.clamp(0, 1), depth_labels, reduction="none").sum(1)
        return 3.0 * (bce * fg).sum() / fg.sum().clamp(min=1.0)

    # ---- exps/mm_training_aim.py:89-112
    def augment_images(self, images, depth_images, stage):
        return augment_images(images, depth_images, stage)

    def forward_loss(self, batch):
        sweep_imgs, mats, pointclouds, gt_boxes, gt_labels = batch
        depth_labels_flat = input_depth = None
        if self.use_cam:
            # training_step :256-259 in two launch sequences: the flags are drawn like augment_images draws them (:98), then
            #   get_depth_labels + the label half of augment_images   -> mmt_depth_labels_flipped (mirrored label write)
            #   normalize_images + the image half of augment_images   -> mmt_normalize_flip_images (one pass, channels_last out)
            B, S, N, _, H, W = sweep_imgs.shape
            if S != 1:       # the labels, their flip flags and the oracle pass-through below are per key frame; the aiMotive loader feeds one sweep (aimotive_dataset.py:194)
                raise RuntimeError(f"TrainStep: {S} sweeps per sample -- the training step takes the key frame only (one sweep), like the reference's loader")
            flips = np.random.uniform(size=(B * S * N)) > 0.5 if self.augment else np.zeros((B * S * N), dtype=bool)
            fl = camera_flags_to_device(flips, sweep_imgs.device)
            depth_labels_flat = depth_labels(pointclouds, mats["extrinsics"][:, 0], mats["intrin_mats"][:, 0], mats["bda_mat"],
                                             (H, W), self.downsample, self.dbound, self.depth_channels, flipped=fl)
            sweep_imgs = normalize_flip_images(sweep_imgs, IMG_MEAN, IMG_STD, fl,
                                               channels_last=os.environ.get("MMT_MEMORY_FORMAT", "channels_last") == "channels_last")
            mats = dict(mats, flipped=fl)                    # :258 (the batch's own dict is left alone)
            if self.pass_depth_labels:                       # :259: depth_labels.permute(0, 3, 1, 2)
                input_depth = depth_labels_flat.view(B * S * N, H // self.downsample, W // self.downsample, -1).permute(0, 3, 1, 2)
        # targets do not depend on the network: build them first so nothing sits between the
        # forward and the backward kernels in the launch queue
        targets = self.model.get_targets(gt_boxes, gt_labels)
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            preds, depth_preds, _, _ = self.net((sweep_imgs, pointclouds), mats, input_depth)
        detection_loss = self.model.loss(targets, preds)
        depth_loss = self.get_depth_loss(depth_labels_flat, depth_preds) if self.use_cam else detection_loss.new_zeros(())
        return detection_loss + depth_loss, detection_loss, depth_loss

    def _gradient_streams(self):
        """The streams (besides the one a hook runs on) that may still be producing a gradient of this step."""
        from ..ops import conv_overlap
        from ..layers.heads import bev_depth_head
        streams = [torch.cuda.current_stream(self.device), torch.cuda.default_stream(self.device)]
        streams += conv_overlap.streams_in_use() + bev_depth_head.streams_in_use()
        return streams

    def finish_backward(self):
        """After `backward()` at world size > 1 with the native reducer: wait for the gradient all-reduces (the stream does, not the
        host) and point every .grad at the reduced values.  A no-op otherwise (DDP has done it inside backward())."""
        if self.reducer is not None:
            self.reducer.finish()

    def forward(self, batch):
        """One optimisation step; returns the (detached) loss tensors, no host sync."""
        self.optimizer.zero_grad(set_to_none=True)
        if self.reducer is not None:
            self.reducer.begin_step()                          # (a new backward pass, whatever became of the last one)
        loss, det, dep = self.forward_loss(batch)
        if self._bn_counters:
            torch._foreach_add_(self._bn_counters, 1)      # see _batched_bn_counters
        loss.backward()
        self.finish_backward()
        if not self.fused_optimizer:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_clip, foreach=True)
        self.optimizer.step()
        return loss.detach(), det.detach(), dep.detach()
