"""Per-step label generation on the GPU -- HIP only.

``depth_labels`` replaces the reference's B x N_cam Python loop of point-cloud projections
(exps/mm_training_aim.py:114-163) and the block-minimum / bin / one-hot chain (:180-215) with
three launches of libmmt_hip.so (``mmt_depth_labels``): no dense H x W depth images, no
boolean-mask selects, no host synchronisation.
"""
import torch

from .. import _lib


def camera_flags_to_device(flags, device):
    """Per-camera boolean flags drawn on the host (numpy / list / CPU tensor) -> uint8 CUDA tensor, through pinned memory and
    an asynchronous copy: the step never waits for the device (exps/mm_training_aim.py:98 draws them with np.random)."""
    if torch.is_tensor(flags) and flags.is_cuda:
        return flags.to(torch.uint8).contiguous()
    host = torch.as_tensor(flags).to(torch.uint8).contiguous()
    return host.pin_memory().to(device, non_blocking=True)


def hflip(t, flipped, group=1):
    """out[i, r, w, :] = t[i, r, W-1-w, :] where flipped[i // group], else t[i, r, w, :] (mmt_hflip; kornia.hflip of the
    selected images).  t: contiguous fp32 CUDA [n, rows, W, E]; flipped: uint8 CUDA [n / group]."""
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4:
        raise RuntimeError("hflip: expected a float32 CUDAtensor [n, rows, W, E]")
    t = t.contiguous()
    n, rows, W, E = t.shape
    if flipped.numel() * group != n or flipped.dtype != torch.uint8 or not flipped.is_cuda:
        raise RuntimeError("hflip: flipped must be a uint8 CUDAtensor with n / group entries")
    out = torch.empty_like(t)
    with torch.cuda.device(t.device):
        _lib.call("mmt_hflip", n, int(group), rows, W, E, t.data_ptr(), flipped.data_ptr(), out.data_ptr(),
                  torch.cuda.current_stream().cuda_stream)
    return out


def normalize_flip_images(sweep_imgs, mean, std, flipped=None, channels_last=True, divisor=255.0):
    """normalize_images (exps/mm_training_aim.py:510-512) fused with the image half of augment_images (:100-104):
    sweep_imgs fp32 CUDA [B, S, N, C>=3, H, W] in 0..255 -> ((x[:, :, :, :3] / divisor) - mean) / std (the division evaluated like
    ATen evaluates a division by a Python scalar on the GPU: a multiplication by the fp32 reciprocal), camera i mirrored along w
    where flipped[i] (uint8 CUDA [B*S*N] or None), as a [B, S, N, 3, H, W] tensor.  channels_last: its memory is
    [B, S, N, H, W, 3], so `reshape(B*S*N, 3, H, W)` is a channels_last view -- what the image backbone's first convolution
    reads (no layout conversion in front of it)."""
    import ctypes
    import numpy as np
    if not sweep_imgs.is_cuda or sweep_imgs.dtype != torch.float32 or sweep_imgs.dim() != 6:
        raise RuntimeError("normalize_flip_images: expected a float32 CUDAtensor [B, S, N, C, H, W]")
    x = sweep_imgs.contiguous()
    B, S, N, C, H, W = x.shape
    n = B * S * N
    if flipped is not None and (flipped.numel() != n or flipped.dtype != torch.uint8 or not flipped.is_cuda):
        raise RuntimeError("normalize_flip_images: flipped must be a uint8 CUDAtensor [B*S*N]")
    out = torch.empty((B, S, N, H, W, 3) if channels_last else (B, S, N, 3, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call("mmt_normalize_flip_images", n, C, H, W, x.data_ptr(), ctypes.c_float(float(np.float32(1.0) / np.float32(divisor))),
                  _lib.float3(mean), _lib.float3(std), flipped.data_ptr() if flipped is not None else 0, out.data_ptr(),
                  1 if channels_last else 0, torch.cuda.current_stream().cuda_stream)
    return out.permute(0, 1, 2, 5, 3, 4) if channels_last else out


def depth_labels(pointclouds, extrinsics, intrinsics, bda_mat, img_hw, downsample, d_bound,
                 depth_channels, return_bins=False, flipped=None):
    """pointclouds: list of B CUDA tensors [Ni, F] (x, y, z first); extrinsics (ego -> camera) and
    intrinsics [B, N, 4, 4]; bda_mat [B, 4, 4]; img_hw = (H, W) of the network input.

    Returns the one-hot labels fp32 [B*N*fH*fW, depth_channels] (what get_downsampled_gt_depth
    returns, :213-214) and, with ``return_bins``, also the int32 bin index per cell.
    flipped: optional uint8 CUDA [B*N]; the label maps of those cameras are written mirrored along w -- the label half of
    augment_images (:105-110) folded into the label write (mmt_depth_labels_flipped)."""
    if not pointclouds or not pointclouds[0].is_cuda:
        raise RuntimeError("pointclouds must be a non-empty list of CUDAtensors ")
    dev = pointclouds[0].device
    B, N = int(extrinsics.shape[0]), int(extrinsics.shape[1])
    if len(pointclouds) != B:
        raise RuntimeError("one point cloud per sample expected")
    H, W = int(img_hw[0]), int(img_hw[1])
    F = int(pointclouds[0].shape[1])
    counts = [int(p.shape[0]) for p in pointclouds]            # shapes are host-side: no sync
    points = torch.cat([p.float() for p in pointclouds], 0).contiguous() if sum(counts) else \
        torch.zeros((1, F), device=dev)
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    offsets = _lib.device_ints(offs, dev)
    bda_inv = torch.linalg.inv_ex(bda_mat[:, :3, :3].float())[0].contiguous()       # inv_ex: no host sync
    n_ws = _lib.lib().mmt_depth_labels_workspace_elems(B, N, H, W, int(downsample))
    if n_ws < 0:
        raise RuntimeError("depth_labels: bad image size / downsample")
    fH, fW = H // downsample, W // downsample
    workspace = torch.empty(n_ws, dtype=torch.int32, device=dev)
    onehot = torch.empty((B * N * fH * fW, int(depth_channels)), dtype=torch.float32, device=dev)
    bins = torch.empty((B * N * fH * fW,), dtype=torch.int32, device=dev) if return_bins else None
    ext = extrinsics.float().contiguous()
    intr = intrinsics.float().contiguous()
    with torch.cuda.device(dev):
        if flipped is not None and (flipped.numel() != B * N or flipped.dtype != torch.uint8 or not flipped.is_cuda):
            raise RuntimeError("depth_labels: flipped must be a uint8 CUDAtensor [B*N]")
        _lib.call("mmt_depth_labels_flipped", B, N, F, max(counts), H, W, int(downsample), float(d_bound[0]), float(d_bound[2]),
                  int(depth_channels), points.data_ptr(), offsets.data_ptr(), ext.data_ptr(), intr.data_ptr(),
                  bda_inv.data_ptr(), workspace.data_ptr(), n_ws, bins.data_ptr() if return_bins else None,
                  onehot.data_ptr(), flipped.data_ptr() if flipped is not None else None, torch.cuda.current_stream().cuda_stream)
    return (onehot, bins) if return_bins else onehot


def centerpoint_targets(gt_boxes, gt_labels, class_counts, max_objs, feature_map_size, pc_range, voxel_size,
                        out_size_factor, gaussian_overlap, min_radius, norm_bbox=True):
    """BEVDepthHead.get_targets (layers/heads/bev_depth_head.py:86-254) in two launches.

    gt_boxes: list of B CUDA tensors [K_b, 9]; gt_labels: list of B tensors [K_b]; class_counts:
    number of classes per task (labels are task-major, as in the reference's class_names lists).
    Returns (heatmaps, anno_boxes, inds, masks): lists over tasks of batched tensors
    ([B, n_cls, fy, fx], [B, max_objs, 10], int64 [B, max_objs], uint8 [B, max_objs])."""
    import ctypes
    if not gt_boxes or not gt_boxes[0].is_cuda:
        raise RuntimeError("gt_boxes must be a non-empty list of CUDAtensors ")
    dev = gt_boxes[0].device
    B, T = len(gt_boxes), len(class_counts)
    fx, fy = int(feature_map_size[0]), int(feature_map_size[1])
    counts = [int(b.shape[0]) for b in gt_boxes]
    if sum(counts):
        boxes = torch.cat([b.float().reshape(-1, 9) for b in gt_boxes], 0).contiguous()
        labels = torch.cat([l.reshape(-1) for l in gt_labels], 0).to(torch.int32).contiguous()
    else:
        boxes, labels = torch.zeros((1, 9), device=dev), torch.zeros((1,), dtype=torch.int32, device=dev)
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    offsets = _lib.device_ints(offs, dev)
    heatmaps = [torch.empty((B, int(n), fy, fx), dtype=torch.float32, device=dev) for n in class_counts]
    annos = [torch.empty((B, max_objs, 10), dtype=torch.float32, device=dev) for _ in class_counts]
    inds = [torch.empty((B, max_objs), dtype=torch.int64, device=dev) for _ in class_counts]
    masks = [torch.empty((B, max_objs), dtype=torch.uint8, device=dev) for _ in class_counts]
    begins, run = [], 0
    for n in class_counts:
        begins.append(run)
        run += int(n)
    i32 = ctypes.c_int32 * T
    vp = ctypes.c_void_p * T
    ptrs = lambda ts: vp(*[t.data_ptr() for t in ts])
    with torch.cuda.device(dev):
        _lib.call("mmt_centerpoint_targets", B, T, ctypes.cast(i32(*begins), ctypes.c_void_p),
                  ctypes.cast(i32(*[int(n) for n in class_counts]), ctypes.c_void_p), int(max_objs), max(counts), fx, fy,
                  float(pc_range[0]), float(pc_range[1]), float(voxel_size[0]), float(voxel_size[1]), int(out_size_factor),
                  float(gaussian_overlap), int(min_radius), int(bool(norm_bbox)), boxes.data_ptr(), labels.data_ptr(),
                  offsets.data_ptr(), ctypes.cast(ptrs(heatmaps), ctypes.c_void_p), ctypes.cast(ptrs(annos), ctypes.c_void_p),
                  ctypes.cast(ptrs(inds), ctypes.c_void_p), ctypes.cast(ptrs(masks), ctypes.c_void_p),
                  torch.cuda.current_stream().cuda_stream)
    return heatmaps, annos, inds, masks
