"""Seeded synthetic inputs of the BASELINE.json shapes (there is no dataset on the box).

Camera rig: analytic 6-camera fan -- yaw 0, +-55, 180, +-110 deg,
f = 0.8 * image width, principal point at the centre, cameras 1.5 m from the ego origin
and 1.5 m above ground, camera->ego axis swap [[0,0,1],[-1,0,0],[0,-1,0]].
LiDAR frames: column layout of dataset/src/data_loader.py:313-337 /
loaders/lidar_loader.py:86-91: [x, y, z, intensity/255, t_norm] (F=5), or with radar
[x, y, z, is_radar, speed, power, intensity, t] (F=8, data_loader.py:324-330).
"""
import math

import numpy as np
import torch

CAM_YAWS_DEG = (0.0, 55.0, -55.0, 180.0, 110.0, -110.0)


def camera_rig(batch_size, num_cams, img_w, img_h, jitter=0.0, seed=0):
    """Returns (sensor2ego [B,N,4,4], intrin [B,N,4,4]) fp32 CPU tensors."""
    rng = np.random.default_rng(seed)
    s2e = np.zeros((batch_size, num_cams, 4, 4), np.float32)
    K = np.zeros((batch_size, num_cams, 4, 4), np.float32)
    axis = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], np.float64)
    for b in range(batch_size):
        for n in range(num_cams):
            yaw = math.radians(CAM_YAWS_DEG[n % len(CAM_YAWS_DEG)]) + (rng.random() - 0.5) * jitter
            R = np.array([[math.cos(yaw), -math.sin(yaw), 0], [math.sin(yaw), math.cos(yaw), 0], [0, 0, 1]])
            s2e[b, n, :3, :3] = R @ axis
            s2e[b, n, :3, 3] = [1.5 * math.cos(yaw), 1.5 * math.sin(yaw), 1.5]
            s2e[b, n, 3, 3] = 1.0
            f = 0.8 * img_w
            K[b, n] = np.array([[f, 0, img_w / 2, 0], [0, f, img_h / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    return torch.from_numpy(s2e), torch.from_numpy(K)


def frustum_geometry_xyz(sensor2ego, intrin, final_dim, downsample, d_bound):
    """Plain-torch frustum points in ego coordinates [B,N,D,fH,fW,3] (test/bench input
    generator; the product kernels are in mm_training_amd.layers)."""
    H, W = final_dim
    fH, fW = H // downsample, W // downsample
    d = torch.arange(*d_bound, dtype=torch.float).view(-1, 1, 1).expand(-1, fH, fW)
    D = d.shape[0]
    xs = torch.linspace(0, W - 1, fW, dtype=torch.float).view(1, 1, fW).expand(D, fH, fW)
    ys = torch.linspace(0, H - 1, fH, dtype=torch.float).view(1, fH, 1).expand(D, fH, fW)
    p = torch.stack((xs * d, ys * d, d, torch.ones_like(d)), -1)          # [D,fH,fW,4]
    combine = sensor2ego.matmul(torch.inverse(intrin))                     # [B,N,4,4]
    xyz = torch.einsum("bnij,dhwj->bndhwi", combine, p)
    return xyz[..., :3].contiguous()


def quantize_cpu(xyz, x_bound, y_bound, z_bound):
    """lss_fpn.py:278-289,461-462 with torch CPU ops (input generator only)."""
    rows = [x_bound, y_bound, z_bound]
    voxel_size = torch.Tensor([r[2] for r in rows])
    voxel_coord = torch.Tensor([r[0] + r[2] / 2.0 for r in rows])
    voxel_num = torch.LongTensor([(r[1] - r[0]) / r[2] for r in rows])
    geom = ((xyz - (voxel_coord - voxel_size / 2.0)) / voxel_size).int()
    return geom, voxel_num


def rig_geometry(batch_size, num_cams=6, final_dim=(256, 704), downsample=16,
                 d_bound=(2.0, 58.0, 0.5), x_bound=(-51.2, 51.2, 0.8), y_bound=(-51.2, 51.2, 0.8),
                 z_bound=(-5.0, 3.0, 8.0), seed=0):
    """int32 geom [B,N,D,fH,fW,3] + voxel_num for the analytic rig (cfg2 by default)."""
    s2e, K = camera_rig(batch_size, num_cams, final_dim[1], final_dim[0], jitter=0.02, seed=seed)
    xyz = frustum_geometry_xyz(s2e, K, final_dim, downsample, d_bound)
    geom, voxel_num = quantize_cpu(xyz, x_bound, y_bound, z_bound)
    return geom.contiguous(), [int(v) for v in voxel_num]


def uniform_geometry(batch_size, num_points, nx, ny, seed=0):
    """Low-locality case: x,y ~ U{-n/4 .. 5n/4}, z = 0 (kept ~ 0.44)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(-nx // 4, nx + nx // 4, (batch_size, num_points), generator=g)
    y = torch.randint(-ny // 4, ny + ny // 4, (batch_size, num_points), generator=g)
    z = torch.zeros_like(x)
    return torch.stack([x, y, z], -1).int().contiguous()


def features(shape, seed=0):
    """U(-0.5, 0.5) fp32 like the reference test (test_voxel_pooling.py:20)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) - 0.5


def lidar_frame(num_points, num_features=5, pc_range=(-204.8, -25.6, -5.0, 204.8, 25.6, 3.0),
                num_radar=0, seed=0, margin=0.02):
    """One synthetic point cloud [N, F]; a `margin` fraction lies outside the range."""
    g = torch.Generator().manual_seed(seed)
    lo = torch.tensor(pc_range[:3])
    hi = torch.tensor(pc_range[3:])
    span = hi - lo
    xyz = lo - margin * span + torch.rand(num_points, 3, generator=g) * span * (1 + 2 * margin)
    pts = torch.zeros(num_points, num_features)
    pts[:, :3] = xyz
    if num_features == 5:
        pts[:, 3] = torch.rand(num_points, generator=g)          # intensity / 255
        pts[:, 4] = torch.rand(num_points, generator=g)          # normalised time
    else:
        is_radar = torch.zeros(num_points)
        is_radar[:num_radar] = 1.0
        pts[:, 3] = is_radar
        pts[:, 4] = torch.randn(num_points, generator=g) * 5 * is_radar    # speed
        pts[:, 5] = torch.rand(num_points, generator=g) * is_radar          # power
        pts[:, 6] = torch.rand(num_points, generator=g) * (1 - is_radar)    # intensity
        pts[:, 7] = torch.rand(num_points, generator=g)
    return pts
