#!/usr/bin/env python3
"""Kernel micro-benchmark for libmmt_hip (run on the GPU box):
   python tools/kbench.py [--shape cfg2|cfg1_full|cfg5] [--reps 20] [--geometry rig|uniform]
Times every voxel_pooling forward algorithm and the backward with HIP events and
prints achieved algorithmic GB/s (formulas: BASELINE.md section 2)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mm_training_amd import _lib, synthetic  # noqa: E402
from mm_training_amd.ops.voxel_pooling import voxel_pooling_ext  # noqa: E402

SHAPES = {
    "cfg2": dict(B=4, N=6, final_dim=(256, 704), ds=16, d_bound=(2.0, 58.0, 0.5), C=80),
    "cfg1_full": dict(B=1, N=6, final_dim=(256, 704), ds=8, d_bound=(1.0, 60.0, 0.5), C=64),
    "cfg5": dict(B=2, N=6, final_dim=(512, 1408), ds=16, d_bound=(2.0, 58.0, 0.5), C=80),
    # the reference's native aiMotive configuration (exps/conf_aim.py:1-3,16-18,42-52; 2 cameras)
    "aim": dict(B=4, N=2, final_dim=(704, 1280), ds=16, d_bound=(2.0, 206.4, 0.5), C=80,
                x_bound=(-204.8, 204.8, 0.8), y_bound=(-25.6, 25.6, 0.8)),
}


def timeit(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="cfg2")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--geometry", default="rig")
    ap.add_argument("--algos", default="3,35,2,1")
    args = ap.parse_args()
    sh = SHAPES[args.shape]
    B, C = sh["B"], sh["C"]
    if args.geometry == "rig":
        geom, vn = synthetic.rig_geometry(B, sh["N"], sh["final_dim"], sh["ds"], sh["d_bound"],
                                          sh.get("x_bound", (-51.2, 51.2, 0.8)), sh.get("y_bound", (-51.2, 51.2, 0.8)))
    else:
        fH, fW = sh["final_dim"][0] // sh["ds"], sh["final_dim"][1] // sh["ds"]
        D = int((sh["d_bound"][1] - sh["d_bound"][0]) / sh["d_bound"][2])
        geom = synthetic.uniform_geometry(B, sh["N"] * D * fH * fW, 128, 128).reshape(B, sh["N"], D, fH, fW, 3)
        vn = [128, 128, 1]
    nx, ny, nz = vn
    P = geom[0].numel() // 3
    feats = synthetic.features((B, P, C), seed=1).cuda()
    geom = geom.reshape(B, P, 3).cuda()
    g3 = geom.reshape(-1, 3)
    K = int((((g3[:, 0] >= 0) & (g3[:, 0] < nx) & (g3[:, 1] >= 0) & (g3[:, 1] < ny) & (g3[:, 2] >= 0) & (g3[:, 2] < nz)).sum()))
    BP = B * P
    fwd_bytes = 24 * BP + 4 * C * K + 4 * C * B * ny * nx
    bwd_bytes = 12 * BP + 4 * C * B * ny * nx + 4 * C * BP
    out = torch.zeros(B, ny, nx, C, device="cuda")
    pos = torch.empty(B, P, 3, dtype=torch.int32, device="cuda")
    res = {"shape": args.shape, "geometry": args.geometry, "BP": BP, "kept": K / BP, "fwd_MB": fwd_bytes / 1e6, "bwd_MB": bwd_bytes / 1e6}
    ref = None
    for algo in [int(a) for a in args.algos.split(",")]:
        flags = (algo & 0xF) | 0x10 | (algo & 0xFFE0)  # e.g. 3, 35 = 3|0x20 (chunk 1024), 67 = 3|0x40 (wave per slot), 3|(58<<8) = 232-point chunks

        def run():
            voxel_pooling_ext.voxel_pooling_forward_wrapper(B, P, C, nx, ny, nz, geom, feats, out, pos, flags=flags)
        out.zero_()
        run()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        err = (out - ref).abs().max().item()
        med, best = timeit(run, args.reps)
        res[f"fwd_algo{algo}"] = {"ms": med, "best_ms": best, "GBps": fwd_bytes / med / 1e6, "max_abs_diff_vs_first": err}
    # interleaved A/B of the listed algorithms (box-to-box and run-to-run noise is ~5 %: only numbers
    # taken in alternation inside one process are comparable)
    algos = [int(a) for a in args.algos.split(",")]
    if len(algos) > 1:
        per = {a_: [] for a_ in algos}
        for rnd in range(12):
            for a_ in algos:
                fl_ = (a_ & 0xF) | 0x10 | (a_ & 0xFFE0)
                med_, _ = timeit(lambda: voxel_pooling_ext.voxel_pooling_forward_wrapper(
                    B, P, C, nx, ny, nz, geom, feats, out, pos, flags=fl_), 8, warm=1)
                per[a_].append(med_)
        res["ab_interleaved_ms"] = {str(a_): sorted(v)[len(v) // 2] for a_, v in per.items()}
    memset_ms, _ = timeit(lambda: out.zero_(), args.reps)
    res["out_memset_ms"] = memset_ms
    go = torch.randn(B, ny, nx, C, device="cuda").permute(0, 3, 1, 2)
    gi = torch.empty(B, P, C, device="cuda")
    WS = None if os.environ.get("KBENCH_NO_WS") else torch.empty(voxel_pooling_ext.backward_workspace_elems(B, P, C, nx, ny), device="cuda")
    med, best = timeit(lambda: voxel_pooling_ext.voxel_pooling_backward_wrapper(B, P, C, nx, ny, pos, go, gi, WS), args.reps)
    res["bwd_nhwc"] = {"ms": med, "best_ms": best, "GBps": bwd_bytes / med / 1e6}
    go2 = go.contiguous()
    ws = torch.empty(voxel_pooling_ext.backward_workspace_elems(B, P, C, nx, ny), device="cuda")
    med, best = timeit(lambda: voxel_pooling_ext.voxel_pooling_backward_wrapper(B, P, C, nx, ny, pos, go2, gi, ws), args.reps)
    res["bwd_nchw_ws"] = {"ms": med, "best_ms": best, "GBps": bwd_bytes / med / 1e6}
    # alternating forward / backward (what a training step does): per-kernel times
    fl = 3 | 0x10
    fe, be = [], []
    for it in range(args.reps + 3):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        out.zero_()
        e[0].record()
        voxel_pooling_ext.voxel_pooling_forward_wrapper(B, P, C, nx, ny, nz, geom, feats, out, pos, flags=fl)
        e[1].record()
        e[2].record()
        voxel_pooling_ext.voxel_pooling_backward_wrapper(B, P, C, nx, ny, pos, go, gi, WS)
        e[3].record()
        if it >= 3:
            fe.append((e[0], e[1]))
            be.append((e[2], e[3]))
    torch.cuda.synchronize()
    medev = lambda evs: sorted(s_.elapsed_time(e_) for s_, e_ in evs)[len(evs) // 2]
    res["alternating"] = {"fwd_ms": medev(fe), "bwd_ms": medev(be), "fwd_GBps": fwd_bytes / medev(fe) / 1e6,
                          "bwd_GBps": bwd_bytes / medev(be) / 1e6}
    # cached-plan forward: plan built once, forward = segmented gather
    from mm_training_amd.ops.voxel_pooling import VoxelPoolingPlan
    from mm_training_amd.ops.voxel_pooling.plan import planned_forward_into
    plan = VoxelPoolingPlan(geom, vn)
    pout = torch.empty(B, ny, nx, C, device="cuda")
    planned_bytes = 4 * C * K + 4 * K + 16 * plan.num_items + 4 * C * B * ny * nx
    med, best = timeit(lambda: planned_forward_into(plan, feats, pout, C), args.reps)
    out.zero_()
    voxel_pooling_ext.voxel_pooling_forward_wrapper(B, P, C, nx, ny, nz, geom, feats, out, pos, flags=fl)
    res["fwd_planned"] = {"ms": med, "best_ms": best, "GBps_own_bytes": planned_bytes / med / 1e6,
                          "GBps_dropin_bytes": fwd_bytes / med / 1e6, "planned_MB": planned_bytes / 1e6,
                          "items": plan.num_items, "multi_cells": plan.num_multi, "partial_rows": plan.num_partial,
                          "max_abs_diff_vs_dropin": (pout - out).abs().max().item(),
                          "pos_memo_equal": bool(torch.equal(plan.pos_memo, pos))}
    fe = []
    for it in range(args.reps + 3):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        planned_forward_into(plan, feats, pout, C)
        e[1].record()
        voxel_pooling_ext.voxel_pooling_backward_wrapper(B, P, C, nx, ny, plan.pos_memo, go, gi, WS)
        if it >= 3:
            fe.append((e[0], e[1]))
    torch.cuda.synchronize()
    res["fwd_planned"]["alternating_ms"] = medev(fe)
    res["fwd_planned"]["alternating_GBps_own_bytes"] = planned_bytes / medev(fe) / 1e6
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); VoxelPoolingPlan(geom, vn); t1.record(); torch.cuda.synchronize()
    res["fwd_planned"]["plan_build_ms"] = t0.elapsed_time(t1)
    # isolated backward but with a cache-flushing 1 GiB read in between (cold MALL)
    flush = torch.empty(256 * 1024 * 1024, device="cuda")
    # full-size check against the oracle (C restatement, a few seconds on the host)
    if os.environ.get("KBENCH_VERIFY"):
        from tests.soak.oracle_checks import verify_pooling
        out.zero_()
        voxel_pooling_ext.voxel_pooling_forward_wrapper(B, P, C, nx, ny, nz, geom, feats, out, pos, flags=fl)
        voxel_pooling_ext.voxel_pooling_backward_wrapper(B, P, C, nx, ny, pos, go, gi, WS)
        torch.cuda.synchronize()
        res["verify"] = verify_pooling(geom, feats, out, pos, go, gi, nx, ny, nz)
    be = []
    for it in range(args.reps + 3):
        flush.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        voxel_pooling_ext.voxel_pooling_backward_wrapper(B, P, C, nx, ny, pos, go, gi, WS)
        e1.record()
        if it >= 3:
            be.append((e0, e1))
    torch.cuda.synchronize()
    res["bwd_after_flush_ms"] = medev(be)
    fe = []
    for it in range(args.reps + 3):
        flush.sum()
        out.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        voxel_pooling_ext.voxel_pooling_forward_wrapper(B, P, C, nx, ny, nz, geom, feats, out, pos, flags=fl)
        e1.record()
        if it >= 3:
            fe.append((e0, e1))
    torch.cuda.synchronize()
    res["fwd_after_flush_ms"] = medev(fe)
    # streaming ceilings on this box for reference
    a = torch.empty(BP * C, device="cuda")
    b = torch.empty_like(a)
    med, _ = timeit(lambda: b.copy_(a), args.reps)
    res["copy_GBps"] = 2 * a.numel() * 4 / med / 1e6
    med_, _ = timeit(lambda: b.zero_(), args.reps)
    res["memset_GBps"] = a.numel() * 4 / med_ / 1e6
    ze = []
    for it in range(args.reps):
        flush.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.zero_()
        e1.record()
        ze.append((e0, e1))
    torch.cuda.synchronize()
    res["memset_after_flush_GBps"] = a.numel() * 4 / medev(ze) / 1e6
    ce = []
    for it in range(args.reps):
        flush.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        ce.append((e0, e1))
    torch.cuda.synchronize()
    res["copy_after_flush_GBps"] = 2 * a.numel() * 4 / medev(ce) / 1e6
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
