"""BEVDepth / BEVFuseLayer / BEVDepthLiDAR -- drop-in mirror of models/bev_depth.py.

Same constructor arguments and ``forward((img, lidar), mats_dict, lidar_oracle,
timestamps)`` -> ``(preds, depth_pred, lidar_bev, cam_bev)`` (models/bev_depth.py:163-200).
``lidar_conf`` builds a mm_training_amd.lidar.LidarEncoder (the three calls at :181-183
hit the HIP kernels) instead of an mmdet3d MVXFasterRCNN.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from ..layers.backbones.lss_fpn import LSSFPN
from ..layers.heads.bev_depth_head import BEVDepthHead
from ..lidar import LidarEncoder
from ..ops.bev_warp import bev_warp_affine, bev_warp_concat, bev_warp_concat_pillars

__all__ = ['BEVDepth', 'BEVFuseLayer', 'BEVDepthLiDAR']


def warp_affine_bev(x, mat23):
    """kornia.geometry.warp_affine(x, M, dsize=x.shape[2:]) (bilinear, zeros, align_corners=True):
    dst(p) = src(M^-1 p) in pixel coordinates."""
    b, _, h, w = x.shape
    M = torch.eye(3, device=x.device, dtype=x.dtype).repeat(b, 1, 1)
    M[:, :2, :] = mat23
    Minv = torch.inverse(M)
    ys, xs = torch.meshgrid(torch.arange(h, device=x.device, dtype=x.dtype),
                            torch.arange(w, device=x.device, dtype=x.dtype), indexing='ij')
    pts = torch.stack((xs, ys, torch.ones_like(xs)), -1).view(1, h * w, 3)
    src = pts @ Minv.transpose(1, 2)
    gx = 2 * src[..., 0] / max(w - 1, 1) - 1
    gy = 2 * src[..., 1] / max(h - 1, 1) - 1
    grid = torch.stack((gx, gy), -1).view(b, h, w, 2)
    return F.grid_sample(x, grid, mode='bilinear', padding_mode='zeros', align_corners=True)


class BEVDepth(nn.Module):
    """Camera-only detector: LSSFPN camera branch -> BEV augmentation warp -> BEVDepthHead.
    Constructor arguments and the attribute names other code reaches for (`backbone`, `head`, `is_train_depth`,
    `bev_augment_image`, `get_targets`, `loss`) are the reference's (models/bev_depth.py:13-130)."""

    def __init__(self, backbone_conf, head_conf, is_train_depth=False, use_cam=True):
        super().__init__()
        # registration order = models/bev_depth.py:26-29 (backbone, then head): `parameters()` order is what an optimizer
        # state dict of a reference checkpoint is keyed by (tests/test_host_logic.py::test_parameter_registration_order)
        if use_cam:
            self.backbone = LSSFPN(**backbone_conf)
        self.head = BEVDepthHead(**head_conf)
        self.is_train_depth = is_train_depth

    def bev_augment_image(self, x, bda_mat):
        """models/bev_depth.py:69-84: rotate/flip the camera BEV by the BEV-aug matrix about
        the map centre -- one HIP launch (ops/bev_warp.py)."""
        return bev_warp_affine(x, bda_mat)

    def bev_augment_image_torch(self, x, bda_mat):
        """The same warp from torch ops (matrix products, batched inverse, grid_sample): the
        cross-check of tests/test_bev_warp_gpu.py, not used by forward."""
        b, _, h, w = x.shape
        h, w = h - 1, w - 1
        t_pos = torch.eye(3, device=x.device, dtype=x.dtype)
        t_pos[0, 2], t_pos[1, 2] = w / 2, h / 2
        t_neg = torch.eye(3, device=x.device, dtype=x.dtype)
        t_neg[0, 2], t_neg[1, 2] = -w / 2, -h / 2
        mat = t_pos.unsqueeze(0) @ bda_mat[:, :3, :3].to(x.dtype) @ t_neg.unsqueeze(0)
        return warp_affine_bev(x, mat[:, :2, :3])

    def get_targets(self, gt_boxes, gt_labels):
        return self.head.get_targets(gt_boxes, gt_labels)

    def loss(self, targets, preds_dicts):
        return self.head.loss(targets, preds_dicts)

    def _camera_bev(self, images, mats_dict, depth_oracle, timestamps):
        """(un-augmented camera BEV map, depth distribution) of the key frame."""
        return self.backbone(images, mats_dict, depth_oracle, timestamps, is_return_depth=True)

    def forward(self, x, mats_dict, timestamps=None):
        cam_bev, depth = self._camera_bev(x[0], mats_dict, None, timestamps)
        return self.head(self.bev_augment_image(cam_bev, mats_dict['bda_mat'])), depth


class BEVFuseLayer(nn.Module):
    """models/bev_depth.py:133-145: 3x3 convolution of the camera|LiDAR stack, gated channel-wise by a squeeze
    (global average -> 1x1 convolution -> sigmoid) of its own output."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.conv_3 = nn.Conv2d(in_channels, in_channels, 3, padding=1)
        self.conv_1 = nn.Conv2d(in_channels, in_channels, 1)
        self.avg_pool = nn.AdaptiveAvgPool2d((1, 1))
        self.activation = nn.Sigmoid()

    def forward(self, x):
        mixed = self.conv_3(x)
        gate = self.activation(self.conv_1(self.avg_pool(mixed)))
        return mixed * gate


class BEVDepthLiDAR(BEVDepth):
    """Camera + LiDAR/radar detector (models/bev_depth.py:148-200).  `forward((img, lidar), mats_dict, lidar_oracle,
    timestamps)` -> `(preds, depth_pred, lidar_bev, cam_bev)`; a disabled modality contributes None."""

    def __init__(self, backbone_conf, head_conf, lidar_conf, is_train_depth=False, use_cam=True,
                 use_lidar=True, fuse_layer_in_channels=144, full_lidar_canvas=None):
        super().__init__(backbone_conf, head_conf, is_train_depth=False, use_cam=use_cam)
        self.use_cam, self.use_lidar = use_cam, use_lidar
        self.sync_free_lidar = os.environ.get("MMT_LIDAR_SYNC_FREE", "1") != "0"
        # The reference scatters the full-resolution pillar canvas and nearest-resizes it onto the camera grid
        # (models/bev_depth.py:183 + :188-190); the canvas itself is only handed back as `lidar_bev_ret`, which nothing in
        # the reference reads (exps/mm_training_aim.py:268 binds it and drops it).  False (default): with both modalities
        # and an integer grid ratio only the canvas cells the resize samples are scattered, straight into the camera|LiDAR
        # buffer, and the third return value is that LiDAR half [B, Cl, H, W].  True: the reference's op sequence and its
        # full-resolution third return value.  `full_lidar_canvas` (constructor argument beside the reference's; None: the
        # MMT_LIDAR_FULL_CANVAS environment variable, default off).  A caller that READS `lidar_bev_ret` -- evaluation with
        # reference parity, visualisation of the canvas -- constructs the module with full_lidar_canvas=True (INTEGRATION.md,
        # mapping table: the one return value whose shape differs from the reference's by default).
        self.full_lidar_canvas = (os.environ.get("MMT_LIDAR_FULL_CANVAS", "0") == "1") if full_lidar_canvas is None else bool(full_lidar_canvas)
        if use_lidar:
            self.lidar_encoder = LidarEncoder(**{k: v for k, v in dict(lidar_conf).items() if k != 'type'})
        if use_cam and use_lidar:
            self.bev_fuse = BEVFuseLayer(in_channels=fuse_layer_in_channels)

    def _lidar_bev(self, clouds):
        if self.sync_free_lidar:
            # voxelize + mean + scatter in the voxelizer's fixed-capacity layout: no device->host copy of the voxel
            # count in the middle of the step (LidarEncoder.forward_bev)
            return self.lidar_encoder.forward_bev(clouds)
        # the reference's three calls (models/bev_depth.py:181-183); voxelize() returns compacted [M, ...]
        # tensors, i.e. M travels to the host
        enc = self.lidar_encoder
        voxels, num_points, coors = enc.voxelize(clouds)
        return enc.pts_middle_encoder(enc.pts_voxel_encoder(voxels, num_points, coors), coors, len(clouds))

    def forward(self, x, mats_dict, lidar_oracle=None, timestamps=None):
        images, clouds = x
        cam_map, depth = self._camera_bev(images, mats_dict, lidar_oracle, timestamps) if self.use_cam else (None, None)
        if cam_map is None:
            lidar_map = self._lidar_bev(clouds)
            return self.head(lidar_map), depth, lidar_map, None
        if not self.use_lidar:
            cam_aug = self.bev_augment_image(cam_map, mats_dict['bda_mat'])
            return self.head(cam_aug), depth, None, cam_aug
        enc = self.lidar_encoder
        ny, nx = enc.output_shape
        H, W = cam_map.shape[-2:]
        if (self.sync_free_lidar and not self.full_lidar_canvas and enc.channels_last and enc.in_channels % 4 == 0
                and cam_map.shape[1] % 4 == 0 and ny % H == 0 and nx % W == 0):
            # both, integer ratio: BEV-aug warp of the camera map (:176) + the sampled pillar cells (:183 + :188-190), each
            # written once, straight into the camera|LiDAR buffer (:192)
            feats, coors, table = enc.forward_rows(clouds)
            stacked = bev_warp_concat_pillars(cam_map, mats_dict['bda_mat'], feats, coors, table, ny, nx, enc.max_voxels)
            c = cam_map.shape[1]
            return self.head(self.bev_fuse(stacked)), depth, stacked[:, c:], stacked[:, :c]
        # both: nearest-resize the pillar canvas onto the camera grid (:188-190), then BEV-aug warp (:176) and channel
        # concat (:189) in one pass -- the warped camera map is written straight into the camera|LiDAR buffer
        lidar_map = self._lidar_bev(clouds)
        lidar_small = lidar_map if lidar_map.shape[-2:] == cam_map.shape[-2:] else F.interpolate(lidar_map, size=cam_map.shape[-2:])
        stacked = bev_warp_concat(cam_map, mats_dict['bda_mat'], lidar_small)
        return self.head(self.bev_fuse(stacked)), depth, lidar_map, stacked[:, :cam_map.shape[1]]
