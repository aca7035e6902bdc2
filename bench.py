#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X BEV-fusion hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One rank per GPU; pure data parallel over the batch axis ("weak" scaling: per-GPU
batch is fixed).  Default (--mode train): a step = one full training step of BASELINE.json
configs[1] (camera-only BEVDepth, bs=4/GPU) on synthetic frames resident in HBM; the
voxel_pooling kernels inside the step are timed with HIP events attached to their dispatches for the roofline lines.
--mode hotpath times only voxel_pooling forward+backward.  See DESIGN.md "Measurement".
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def use_shipped_miopen_db():
    """MIOpen's per-shape solver choice for the dense nets: the repo ships the user find/perf
    DB produced by one exhaustive search on an MI355X (mm_training_amd/miopen_db, ~100 KB of
    text).  Each process works on a private copy (MIOpen rewrites the files), so a fresh box gets
    the tuned solvers without the ~4 min search.  Must run before the first convolution."""
    src = os.path.join(ROOT, "mm_training_amd", "miopen_db")
    if "MIOPEN_USER_DB_PATH" in os.environ or not os.path.isdir(src):
        return False
    import shutil
    import tempfile
    dst = tempfile.mkdtemp(prefix="mmt_miopen_db_")
    for f in os.listdir(src):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    # HYBRID find mode: a find-DB hit returns the tuned solver without running anything, a miss
    # times the applicable solvers once (seconds) instead of trusting the immediate-mode heuristic.
    # The reference "naive" solvers (tens of ms per call, never chosen) are excluded from that
    # timing: they alone cost ~15 s of warm-up per process (profiles/r01_miopen_find_modes.txt).
    os.environ.setdefault("MIOPEN_FIND_MODE", "3")
    for d in ("FWD", "BWD", "WRW"):
        os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + d, "0")
    return True


SHIPPED_MIOPEN_DB = use_shipped_miopen_db()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)

# BASELINE.json configs[1]: camera-only BEVDepth, 6 cams 256x704, ds 16, D=112, C=80, BEV 128x128, bs=4
CFG2 = dict(batch=4, num_cams=6, final_dim=(256, 704), downsample=16, d_bound=(2.0, 58.0, 0.5),
            channels=80, x_bound=(-51.2, 51.2, 0.8), y_bound=(-51.2, 51.2, 0.8), z_bound=(-5.0, 3.0, 8.0))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="train", choices=["train", "hotpath"],
                    help="train: full data-parallel training step of the BASELINE config (default); "
                         "hotpath: only voxel_pooling forward+backward at the cfg-2 shape")
    ap.add_argument("--config", default="cfg2", help="cfg2 (BASELINE configs[1], default) | cfg3 | cfg4 | cfg5 | tiny")
    ap.add_argument("--miopen-tune", action="store_true", help="exhaustive MIOpen search (minutes of warm-up)")
    ap.add_argument("--cached-plan", action="store_true",
                    help="train mode: pass a calibration id so voxel_pooling reuses a cached sort (SURVEY 8/f3)")
    ap.add_argument("--fused-lift-splat", action="store_true",
                    help="camera branch uses the fused lift-splat kernels (row f1) instead of lift -> voxel_pooling; "
                         "the voxel_pooling roofline lines are then not produced")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--algo", type=int, default=0, help="voxel_pooling forward algorithm flag")
    return ap.parse_args()


def choose_backend(world, local_rank, ndev, requested="nccl"):
    """Backend and device index of a rank.  One GPU per rank -> RCCL ("nccl").  Fewer GPUs than ranks
    (functional rehearsal of the N>1 path on a 1-GPU box only): ranks share devices, which RCCL refuses, so
    EVERY rank falls back to gloo -- the decision depends on (world, ndev) alone, never on the rank, or
    rank 0 would pick RCCL while the others pick gloo and the rendezvous would hang."""
    if world > max(ndev, 1):
        return "gloo", local_rank % max(ndev, 1)
    return requested, local_rank


def init_dist(n_gpus):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ndev = torch.cuda.device_count()
    backend, local_rank = choose_backend(world, local_rank, ndev, os.environ.get("MMT_DIST_BACKEND", "nccl"))   # "nccl" is RCCL on ROCm
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    assert world == n_gpus, f"--gpus {n_gpus} but WORLD_SIZE={world}"
    return rank, local_rank, world


def barrier(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def algorithmic_bytes(BP, K, C, B, ny, nx):
    """BASELINE.md section 2 / SURVEY.md section 8d."""
    fwd = 12 * BP + 12 * BP + 4 * C * K + 4 * C * B * ny * nx
    bwd = 12 * BP + 4 * C * B * ny * nx + 4 * C * BP
    return fwd, bwd


def cpu_baseline(geom, feats, vn, grad_out_nhwc, budget_s=20.0):
    """The oracle's torch-CPU port of the reference semantics (BASELINE.md section 2:
    scatter_add_ forward + masked gather backward) timed on this host's cores.  The thread
    count is calibrated (all logical cores is pathologically slow for scatter_add_ on a
    many-core host); `cores` reports the count actually used."""
    import oracle
    B, P, C = feats.shape
    nx, ny, nz = vn

    def one():
        t0 = time.perf_counter()
        out, pos = oracle.torch_forward_scatter_add(geom, feats, nx, ny, nz)
        gi = oracle.torch_backward_gather(pos, grad_out_nhwc)
        return time.perf_counter() - t0, out, pos, gi

    ncpu = os.cpu_count() or 1
    t_start = time.perf_counter()
    best_threads, best_t = None, None
    for th in sorted({min(ncpu, 16), min(ncpu, 64), ncpu}):
        torch.set_num_threads(th)
        if best_threads is None:
            one()                                   # first-touch / allocator warm-up
        t = one()[0]
        if best_t is None or t < best_t:
            best_threads, best_t = th, t
        if time.perf_counter() - t_start > budget_s:
            break
    torch.set_num_threads(best_threads)
    times = []
    while len(times) < 3 or (time.perf_counter() - t_start < budget_s and len(times) < 30):
        t, out, pos, gi = one()
        times.append(t)
    times.sort()
    med = times[len(times) // 2]
    return {"value": B / med, "unit": "samples/s", "cores": best_threads, "kind": "port",
            "sample": f"{len(times)} reps of voxel_pooling fwd (torch scatter_add_) + bwd (masked gather) "
                      f"on CPU at the full cfg-2 shape B={B} P={P} C={C}, median {med * 1e3:.1f} ms/step, "
                      f"{best_threads} of {ncpu} logical cores (calibrated)"}, out, pos, gi


def pmc_traffic(kernels):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary (separate FETCH_SIZE /
    WRITE_SIZE passes with the gfx950 correction, profiles/r01_hotpath_cfg2_pmc.json).  PMC
    collection cannot run inside this process; the figure belongs to the same kernel, shape
    (cfg2) and geometry the roofline line is quoted on."""
    path = os.path.join(ROOT, "profiles", "r01_hotpath_cfg2_pmc.json")
    try:
        k = json.load(open(path))["kernels"]
        return float(sum(k[name]["traffic_bytes"] for name in kernels))
    except Exception:
        return None


def roofline_entry(kernel, nbytes, ms, pmc_kernels=()):
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "traffic": pmc_traffic(pmc_kernels) if pmc_kernels else None,
            "algorithmic_bytes": nbytes, "avg_ms": ms}


def train_main(args, rank, local_rank, world):
    """Full training step of a BASELINE config: synthetic frames -> depth labels -> BEVDepth
    (+LiDAR) forward -> detection + depth loss -> backward (DDP bucketed all-reduce over RCCL,
    overlapped) -> grad clip -> AdamW.  Nothing is skipped inside the timed region."""
    from mm_training_amd import _lib
    from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
    from mm_training_amd.ops.voxel_pooling import voxel_pooling_ext
    _lib.lib()
    # benchmark=True makes PyTorch ask MIOpen's find API, which is answered from the find DB
    # (only for the configuration the DB was produced on: an unknown shape would start a search)
    torch.backends.cudnn.benchmark = bool(args.miopen_tune) or (SHIPPED_MIOPEN_DB and args.config == "cfg2")
    dev = torch.device("cuda", local_rank)
    cfg = make_config(args.config)
    torch.manual_seed(0)
    ts = TrainStep(cfg, dev, world_size=world)
    if args.fused_lift_splat and cfg["use_cam"]:
        ts.model.backbone.fused_lift_splat = True
    B = cfg["batch_size"]
    # a small pool of distinct pre-generated batches resident in HBM (input is never the bottleneck)
    batches = [synthetic_batch(cfg, dev, seed=1000 * rank + i) for i in range(2)]
    if args.cached_plan:
        # each synthetic batch has its own (jittered) rig; its id lets LSSFPN reuse the
        # sort of the points by BEV cell instead of redoing geometry + sort every step
        for i, b in enumerate(batches):
            b[1]["calibration_id"] = ("synthetic", 1000 * rank + i)
    for i in range(args.warmup):
        ts(batches[i % len(batches)])
    voxel_pooling_ext.TIMING = {}
    barrier(world)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, det, dep = ts(batches[i % len(batches)])
    barrier(world)
    elapsed = time.perf_counter() - t0
    timing = voxel_pooling_ext.TIMING
    voxel_pooling_ext.TIMING = None
    t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank != 0:
        return
    res = {
        "metric": "training samples/sec at bs=%d/GPU; voxel_pooling HBM GB/s" % B,
        "value": world * B * args.steps / elapsed, "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if cfg["dtype"] == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": {
            "cfg2": "BASELINE configs[1]: camera-only BEVDepth (ResNet-50, 6 cams 256x704, D=112, C=80, BEV 128x128) "
                    "full training step (fwd + det/depth loss + bwd + clip + AdamW)",
            "cfg3": "BASELINE configs[2]: LiDAR-only pillar path, 40k pts, 0.2 m voxels",
            "cfg4": "BASELINE configs[3]: LiDAR+camera fusion (BEVDepth + pillar BEV concat)",
            "cfg5": "BASELINE configs[4]: LiDAR+radar+camera, 6 cams 512x1408, 80k pts, bf16",
            "tiny": "tiny smoke configuration"}[args.config],
            "global_batch": world * B, "parallelism": f"dp{world}", "mode": "train",
            "params_M": sum(p.numel() for p in ts.model.parameters()) / 1e6,
            "final_loss": float(loss), "miopen_exhaustive_search": bool(args.miopen_tune),
            "miopen_shipped_find_db": bool(SHIPPED_MIOPEN_DB and args.config == "cfg2"),
            "fused_lift_splat": bool(args.fused_lift_splat), "cached_plan": bool(args.cached_plan)},
    }
    if cfg["use_cam"] and timing.get("forward"):
        fwd_ms = sum(s.elapsed_time(e) for s, e in timing["forward"]) / len(timing["forward"])
        bwd_ms = sum(s.elapsed_time(e) for s, e in timing["backward"]) / len(timing["backward"])
        lss = ts.model.backbone
        with torch.no_grad():
            m = batches[0][1]
            geom = lss.get_geometry_voxels(m["sensor2ego_mats"][:, 0], m["intrin_mats"][:, 0])
        nx, ny, nz = lss._voxel_num_host
        g3 = geom.reshape(-1, 3)
        kept = ((g3[:, 0] >= 0) & (g3[:, 0] < nx) & (g3[:, 1] >= 0) & (g3[:, 1] < ny) & (g3[:, 2] >= 0) & (g3[:, 2] < nz))
        K, BP, C = int(kept.sum()), g3.shape[0], lss.output_channels
        fb, bb = algorithmic_bytes(BP, K, C, B, ny, nx)
        cfg2 = args.config == "cfg2"
        if args.cached_plan and lss._plan_cache:
            plan = next(iter(lss._plan_cache.values()))
            # cached sort: no geom read / pos_memo write per step; row ids + item descriptors instead
            fb = 4 * C * K + 4 * K + 16 * plan.num_items + 4 * C * B * ny * nx
            res["roofline"] = roofline_entry("vp_planned_items + vp_planned_fold (cached-plan voxel_pooling forward, "
                                             "inside the training step)", fb, fwd_ms, ())
        else:
            res["roofline"] = roofline_entry("vp_fwd_seg_gather (voxel_pooling forward, inside the training step)", fb, fwd_ms,
                                             ("vp_fwd_seg_gather",) if cfg2 else ())
        res["roofline_backward"] = roofline_entry("vp_bwd_prepare + vp_bwd_rows_vec4 (voxel_pooling backward, inside the training step)",
                                                  bb, bwd_ms, ("vp_bwd_prepare", "vp_bwd_rows_vec4") if cfg2 else ())
        res["config"]["kept_fraction"] = K / BP
        if world == 1 and not args.no_cpu_baseline:
            from mm_training_amd import synthetic
            P = BP // B
            feats_cpu = synthetic.features((B, P, C), seed=100)
            go = torch.randn(B, ny, nx, C, generator=torch.Generator().manual_seed(1))
            base, _, _, _ = cpu_baseline(geom.reshape(B, P, 3).cpu(), feats_cpu, (nx, ny, nz), go)
            res["cpu_baseline"] = base
    print(json.dumps(res), flush=True)


def main():
    args = parse()
    rank, local_rank, world = init_dist(args.gpus)
    if args.mode == "train":
        train_main(args, rank, local_rank, world)
        if world > 1:
            dist.destroy_process_group()
        return
    from mm_training_amd import _lib, synthetic
    from mm_training_amd.ops.voxel_pooling import voxel_pooling, voxel_pooling_ext
    _lib.lib()

    cfg = CFG2
    B, C = cfg["batch"], cfg["channels"]
    geom_cpu, vn = synthetic.rig_geometry(B, cfg["num_cams"], cfg["final_dim"], cfg["downsample"],
                                          cfg["d_bound"], cfg["x_bound"], cfg["y_bound"], cfg["z_bound"],
                                          seed=rank)
    feats_cpu = synthetic.features(tuple(geom_cpu.shape[:-1]) + (C,), seed=100 + rank)
    nx, ny, nz = vn
    P = feats_cpu[0].numel() // C
    geom = geom_cpu.cuda()
    feats = feats_cpu.cuda().requires_grad_(True)
    g = torch.Generator().manual_seed(1)
    grad_out = torch.randn(B, ny, nx, C, generator=g).cuda().permute(0, 3, 1, 2)  # channels-last grad

    def step():
        feats.grad = None
        out = voxel_pooling(geom, feats, vn)
        out.backward(grad_out)
        return out

    for _ in range(args.warmup):
        step()
    voxel_pooling_ext.TIMING = {}
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier(world)
    elapsed = time.perf_counter() - t0
    timing = voxel_pooling_ext.TIMING
    voxel_pooling_ext.TIMING = None

    t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    fwd_ms = sum(s.elapsed_time(e) for s, e in timing["forward"]) / len(timing["forward"])
    bwd_ms = sum(s.elapsed_time(e) for s, e in timing["backward"]) / len(timing["backward"])
    g3 = geom.reshape(-1, 3)
    kept = ((g3[:, 0] >= 0) & (g3[:, 0] < nx) & (g3[:, 1] >= 0) & (g3[:, 1] < ny)
            & (g3[:, 2] >= 0) & (g3[:, 2] < nz))
    K = int(kept.sum().item())
    fwd_bytes, bwd_bytes = algorithmic_bytes(B * P, K, C, B, ny, nx)

    if rank == 0:
        res = {
            "metric": "training samples/sec at bs=4/GPU (hot path: voxel_pooling fwd+bwd); voxel_pooling HBM GB/s",
            "value": world * B * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1] camera half: voxel_pooling forward+backward, "
                                   "6 cams 256x704 ds16 D=112 fH=16 fW=44 C=80, BEV 128x128x1, "
                                   "analytic 6-camera rig geometry, bs=4/GPU",
                       "global_batch": world * B, "points_per_sample": P, "kept_fraction": K / (B * P),
                       "parallelism": f"dp{world}", "mode": args.mode},
            "roofline": roofline_entry("vp_fwd_seg_gather (voxel_pooling forward)", fwd_bytes, fwd_ms, ("vp_fwd_seg_gather",)),
            "roofline_backward": roofline_entry("vp_bwd_prepare + vp_bwd_rows_vec4 (voxel_pooling backward)", bwd_bytes, bwd_ms,
                                                ("vp_bwd_prepare", "vp_bwd_rows_vec4")),
        }
        if world == 1 and not args.no_cpu_baseline:
            base, ref_out, ref_pos, ref_gi = cpu_baseline(geom_cpu.reshape(B, P, 3), feats_cpu.reshape(B, P, C),
                                                          vn, grad_out.permute(0, 2, 3, 1).contiguous().cpu())
            res["cpu_baseline"] = base
            # same-run parity of the measured path against the CPU port
            err = (out.detach().permute(0, 2, 3, 1).cpu() - ref_out).abs().max().item()
            gi_equal = bool(torch.equal(feats.grad.reshape(B, P, C).cpu(), ref_gi))
            res["parity"] = {"bev_max_abs_err": err, "grad_in_bit_exact": gi_equal}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
