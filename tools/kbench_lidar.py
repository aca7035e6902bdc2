#!/usr/bin/env python3
"""LiDAR-half kernel microbenchmark (run on the GPU box): voxelize / VFE / pillar scatter /
lift / geometry timings with HIP events + algorithmic GB/s (SURVEY.md section 8d formulas)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mm_training_amd import _lib, synthetic
from mm_training_amd.lidar import hard_voxelize_batch, hard_voxelize_mean_batch, simple_vfe, pillar_scatter
from mm_training_amd.ops.bev_geometry import frustum_geometry, lift_features, quantize_geometry


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def timeit_dispatch(fn, kinds, reps=20, warm=3):
    """Median kernel-side duration (ms) of the timed entry points `kinds` inside fn: HIP events attached to the
    dispatches (mmt_arm_kernel_timing), i.e. what rocprofv3 --kernel-trace reports, summed over the kinds."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    _lib.TIMING = {}
    try:
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        t = _lib.TIMING
    finally:
        _lib.TIMING = None
    per_rep = [sum(t[k][i][0].elapsed_time(t[k][i][1]) for k in kinds) for i in range(reps)]
    per_rep.sort()
    return per_rep[len(per_rep) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--points", type=int, default=40000)
    ap.add_argument("--features", type=int, default=5)
    ap.add_argument("--range", default="aim", choices=["aim", "nusc"])
    args = ap.parse_args()
    rng = [-204.8, -25.6, -5.0, 204.8, 25.6, 3.0] if args.range == "aim" else [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
    vs = [0.2, 0.2, 8.0]
    B, N, F = args.batch, args.points, args.features
    frames = [synthetic.lidar_frame(N, F, rng, num_radar=2000 if F == 8 else 0, seed=i).cuda() for i in range(B)]
    res = {"batch": B, "points": N, "F": F, "range": args.range}
    ms = timeit(lambda: hard_voxelize_batch(frames, vs, rng, 15, 25000, compact=False))
    voxels, num_points, coors, cnt = hard_voxelize_batch(frames, vs, rng, 15, 25000, compact=False)
    M = int(cnt.sum())
    vox_bytes = 4 * F * B * N + 16 * M + 4 * M + 4 * 15 * F * M
    res["voxelize_fixed_capacity"] = {"ms": ms, "voxels": M, "algorithmic_MB": vox_bytes / 1e6, "GBps": vox_bytes / ms / 1e6}
    # fused voxelize + mean on the persistent generation-stamped table (three kernels, no clearing pass): what
    # LidarEncoder.forward_bev runs.  Kernel-side durations (dispatch-attached events).
    for tag, mat in (("voxelize_mean_fused_materialized", True), ("voxelize_mean_fused", False)):
        kms = timeit_dispatch(lambda: hard_voxelize_mean_batch(frames, vs, rng, 15, 25000, 5, materialize_voxels=mat), ("voxelize",))
        wms = timeit(lambda: hard_voxelize_mean_batch(frames, vs, rng, 15, 25000, 5, materialize_voxels=mat))
        bts = 4 * F * B * N + 20 * M + 4 * 5 * M + (4 * 15 * F * M if mat else 0)
        res[tag] = {"kernel_ms": kms, "stream_ms_with_launch_overheads": wms, "algorithmic_MB": bts / 1e6, "GBps": bts / kms / 1e6,
                    "frac_of_peak": bts / kms / 1e6 / 8000.0}
    kms = timeit_dispatch(lambda: hard_voxelize_batch(frames, vs, rng, 15, 25000, compact=False), ("voxelize",))
    res["voxelize_fixed_capacity"]["kernel_ms_without_table_memset"] = kms
    ms = timeit(lambda: hard_voxelize_batch(frames, vs, rng, 15, 25000, compact=True))
    res["voxelize_compact_with_host_sync"] = {"ms": ms}
    v, n, c = hard_voxelize_batch(frames, vs, rng, 15, 25000, compact=True)
    ms = timeit(lambda: simple_vfe(v, n, 5))
    vb = 4 * 15 * F * M + 4 * M + 4 * 5 * M
    res["simple_vfe"] = {"ms": ms, "GBps": vb / ms / 1e6}
    C = 64
    feats = torch.randn(M, C, device="cuda")
    ny = int(round((rng[4] - rng[1]) / vs[1])); nx = int(round((rng[3] - rng[0]) / vs[0]))
    ms = timeit(lambda: pillar_scatter(feats, c, B, ny, nx))
    sb = 4 * C * M + 16 * M + 4 * C * B * ny * nx
    res["pillar_scatter"] = {"ms": ms, "canvas": [B, C, ny, nx], "algorithmic_MB": sb / 1e6, "GBps": sb / ms / 1e6}
    # channels-last canvas (what the channels_last BEV trunk consumes) and the two backward variants
    ms_cl = timeit(lambda: pillar_scatter(feats, c, B, ny, nx, channels_last=True))
    kms_cl = timeit_dispatch(lambda: pillar_scatter(feats, c, B, ny, nx, channels_last=True), ("scatter",))
    res["pillar_scatter_channels_last"] = {"ms": ms_cl, "kernel_ms": kms_cl, "GBps": sb / kms_cl / 1e6, "frac_of_peak": sb / kms_cl / 1e6 / 8000.0}
    for tag, cl in (("nchw", False), ("channels_last", True)):
        fr = feats.detach().clone().requires_grad_(True)
        cv = pillar_scatter(fr, c, B, ny, nx, channels_last=cl)
        go = torch.randn(B, C, ny, nx, device="cuda")
        if cl:
            go = go.contiguous(memory_format=torch.channels_last)
        ms_b = timeit(lambda: torch.autograd.grad(cv, fr, go, retain_graph=True))
        kms_b = timeit_dispatch(lambda: torch.autograd.grad(cv, fr, go, retain_graph=True), ("scatter_backward",))
        bb = 4 * C * feats.shape[0] * 2 + 16 * feats.shape[0]        # gradient rows read + written, coors
        res["pillar_scatter_backward_" + tag] = {"ms": ms_b, "kernel_ms": kms_b, "algorithmic_MB": bb / 1e6, "GBps": bb / kms_b / 1e6,
                                                 "frac_of_peak": bb / kms_b / 1e6 / 8000.0}
    # camera-side producers at cfg2
    s2e, K = synthetic.camera_rig(4, 6, 704, 256, jitter=0.02)
    combine = (s2e @ torch.inverse(K)).cuda()
    d = torch.arange(2.0, 58.0, 0.5).view(-1, 1, 1).expand(-1, 16, 44)
    xs = torch.linspace(0, 703, 44).view(1, 1, 44).expand(112, 16, 44)
    ys = torch.linspace(0, 255, 16).view(1, 16, 1).expand(112, 16, 44)
    frustum = torch.stack((xs, ys, d, torch.ones_like(d)), -1).contiguous().cuda()
    vc, vsz = [-50.8, -50.8, -1.0], [0.8, 0.8, 8.0]
    ms = timeit(lambda: frustum_geometry(frustum, combine, vc, vsz))
    BP = 4 * 6 * 112 * 16 * 44
    res["frustum_geometry"] = {"ms": ms, "GBps": (12 * BP + 16 * 112 * 16 * 44) / ms / 1e6}
    xyz_dev = torch.randn(BP, 3, device="cuda") * 40
    ms = timeit(lambda: quantize_geometry(xyz_dev, vc, vsz))
    res["quantize_geometry"] = {"ms": ms, "GBps": 24 * BP / ms / 1e6}
    depth = torch.rand(24, 112, 16, 44, device="cuda").softmax(1).requires_grad_(True)
    ctx = torch.randn(24, 80, 16, 44, device="cuda", requires_grad=True)
    ms = timeit(lambda: lift_features(depth, ctx))
    lb = 4 * 80 * BP + depth.numel() * 4 + ctx.numel() * 4
    res["lift_forward"] = {"ms": ms, "GBps": lb / ms / 1e6}
    out = lift_features(depth, ctx)
    go = torch.randn_like(out)
    ms = timeit(lambda: torch.autograd.grad(out, (depth, ctx), go, retain_graph=True))
    res["lift_backward"] = {"ms": ms, "GBps": lb / ms / 1e6}
    # fused lift-splat (row f1) vs the unfused pair at cfg2
    from mm_training_amd.ops.bev_geometry import lift_splat
    from mm_training_amd.ops.voxel_pooling import voxel_pooling
    geom6, vn = synthetic.rig_geometry(4)
    geom6 = geom6.cuda()
    ctx_cl = ctx.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    dep = depth.detach().requires_grad_(True)
    ms = timeit(lambda: lift_splat(geom6, dep, ctx_cl, vn))
    kms = timeit_dispatch(lambda: lift_splat(geom6, dep, ctx_cl, vn), ("lift_splat_forward",))
    res["fused_lift_splat_forward"] = {"ms": ms, "kernel_ms": kms, "kernel": "lss_splat_fwd_tile (frustum tiles, context tile in LDS)"}
    os.environ["MMT_LIFT_SPLAT_V1"] = "1"
    kms1 = timeit_dispatch(lambda: lift_splat(geom6, dep, ctx_cl, vn), ("lift_splat_forward",))
    res["fused_lift_splat_forward_v1_chunked"] = {"kernel_ms": kms1}
    os.environ["MMT_LIFT_SPLAT_V1"] = "0"
    o = lift_splat(geom6, dep, ctx_cl, vn)
    go2 = torch.randn(4, 128, 128, 80, device="cuda").permute(0, 3, 1, 2)
    ms = timeit(lambda: torch.autograd.grad(o, (dep, ctx_cl), go2, retain_graph=True))
    kms = timeit_dispatch(lambda: torch.autograd.grad(o, (dep, ctx_cl), go2, retain_graph=True), ("lift_splat_backward",))
    res["fused_lift_splat_backward"] = {"ms": ms, "kernel_ms": kms}
    def unfused():
        f = lift_features(dep, ctx_cl).view(4, 6, 112, 16, 44, 80)
        return voxel_pooling(geom6, f, vn)
    ms = timeit(unfused)
    res["unfused_lift_plus_pool_forward"] = {"ms": ms}
    o2 = unfused()
    ms = timeit(lambda: torch.autograd.grad(o2, (dep, ctx_cl), go2, retain_graph=True))
    res["unfused_lift_plus_pool_backward"] = {"ms": ms}
    # CPU baseline beside it: the oracle's sequential C restatement (1 core) on the same inputs
    from tests.soak.oracle_checks import lidar_and_producer_cpu_timings
    res["cpu_baseline_oracle_1core_ms"] = lidar_and_producer_cpu_timings(
        [f.cpu().numpy() for f in frames], vs, rng, feats.cpu().numpy(), B, ny, nx, xyz_dev.cpu().numpy(), vc, vsz,
        frustum.cpu().numpy(), combine.cpu().numpy(), depth.detach().cpu().numpy(), ctx.detach().cpu().numpy())
    # BEV-augmentation warp of the pooled camera map at the cfg-2 shape: forward gather and
    # atomics-free backward gather, each 2 x 21 MB of algorithmic traffic (+21 MB read-modify-write backward)
    from mm_training_amd import _lib
    Bw, Hw, Ww, Cw = 4, 128, 128, 80
    xw = torch.randn(Bw, Hw, Ww, Cw, device="cuda")
    yw, gw = torch.empty_like(xw), torch.zeros_like(xw)
    bda = torch.eye(4).repeat(Bw, 1, 1)
    bda[:, :2, :2] = torch.tensor([[0.96, -0.28], [0.28, 0.96]]) * 1.03
    bda = bda.cuda()
    st = torch.cuda.current_stream().cuda_stream
    wb = 2 * xw.numel() * 4
    ms = timeit(lambda: _lib.call("mmt_bev_warp_affine", Bw, Hw, Ww, Cw, bda.data_ptr(), xw.data_ptr(), Cw, yw.data_ptr(), Cw, st))
    res["bev_warp_forward"] = {"ms": ms, "GBps": wb / ms / 1e6}
    ms = timeit(lambda: _lib.call("mmt_bev_warp_affine_backward", Bw, Hw, Ww, Cw, bda.data_ptr(), yw.data_ptr(), Cw, gw.data_ptr(), Cw, st))
    res["bev_warp_backward"] = {"ms": ms, "GBps": (wb + xw.numel() * 4) / ms / 1e6}
    res["hbm_peak_GBps"] = 8000.0
    for k in ("voxelize_fixed_capacity", "simple_vfe", "pillar_scatter", "lift_forward", "lift_backward", "frustum_geometry", "quantize_geometry"):
        res[k]["frac_of_peak"] = res[k]["GBps"] / 8000.0
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
