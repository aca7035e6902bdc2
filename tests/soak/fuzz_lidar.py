#!/usr/bin/env python3
"""Soak / fuzz of the LiDAR kernels (voxelize -> VFE -> pillar scatter + backward) against the
oracle (run on the GPU box).  Exits non-zero on the first mismatch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import oracle
from mm_training_amd.lidar import (hard_voxelize_batch, hard_voxelize_mean_batch, pillar_scatter, pillar_scatter_from_table,
                                   pillar_scatter_strided, simple_vfe)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
it = 0
_next_note = time.time() + 60.0     # a line a minute: a silent GPU job is taken to be hung
while time.time() < t_end:
    if time.time() > _next_note:
        print("...", it, "configurations so far", flush=True)
        _next_note = time.time() + 60.0
    it += 1
    B = int(rng.integers(1, 5))
    F = int(rng.choice([3, 4, 5, 8]))
    vs = [float(rng.choice([0.1, 0.2, 0.5, 1.0])), float(rng.choice([0.1, 0.25, 0.5])), float(rng.choice([1.0, 4.0, 8.0]))]
    gx, gy, gz = int(rng.integers(1, 300)), int(rng.integers(1, 300)), int(rng.integers(1, 4))
    lo = [float(rng.uniform(-50, 0)), float(rng.uniform(-50, 0)), float(rng.uniform(-5, 0))]
    pcr = lo + [lo[0] + gx * vs[0], lo[1] + gy * vs[1], lo[2] + gz * vs[2]]
    T = int(rng.choice([1, 3, 15, 32]))
    V = int(rng.choice([1, 10, 500, 25000]))
    frames = []
    for b in range(B):
        n = int(rng.choice([0, 1, 5, 1000, 1024, 1025, 30000]))
        mode = rng.choice(["uniform", "cluster", "same"])
        span = np.array([gx * vs[0], gy * vs[1], gz * vs[2]])
        if mode == "uniform":
            xyz = np.array(lo) - 0.05 * span + rng.random((n, 3)) * span * 1.1
        elif mode == "cluster":
            c = np.array(lo) + rng.random((6, 3)) * span
            xyz = c[rng.integers(0, 6, n)] + rng.normal(0, 0.3, (n, 3))
        else:
            xyz = np.tile(np.array(lo) + 0.5 * span, (n, 1))
        pts = np.concatenate([xyz, rng.random((n, F - 3))], 1).astype(np.float32)
        frames.append(pts)
    cfg = dict(it=it, B=B, F=F, vs=vs, grid=(gx, gy, gz), T=T, V=V, sizes=[f.shape[0] for f in frames])
    if os.environ.get("FUZZ_VERBOSE"):
        print("cfg", cfg, flush=True)
    # the oracle's grid comes from round((max-min)/vs) in fp32, like the product
    rv, rn, rc = oracle.voxelize_batch(frames, vs, pcr, T, V)
    dev = [torch.from_numpy(f).cuda() for f in frames]
    v, n, c = hard_voxelize_batch(dev, vs, pcr, T, V)
    if not (np.array_equal(c.cpu().numpy(), rc) and np.array_equal(n.cpu().numpy(), rn)
            and np.array_equal(v.cpu().numpy().view(np.int32), rv.view(np.int32))):
        print("MISMATCH voxelize", cfg)
        sys.exit(1)
    M = rc.shape[0]
    nf = min(F, 5)
    m = simple_vfe(v, n, nf).cpu().numpy()
    if M and not np.allclose(m, oracle.simple_vfe(rv, rn, nf), rtol=1e-6, atol=1e-7, equal_nan=True):
        print("MISMATCH vfe", cfg)
        sys.exit(1)
    # fused voxelize + mean on the persistent generation-stamped table (never cleared between the random configurations)
    mat = bool(rng.integers(0, 2))
    v3, n3, c3, cnt3, m3 = hard_voxelize_mean_batch(dev, vs, pcr, T, V, nf, materialize_voxels=mat)
    live = (c3[:, 0] >= 0).cpu().numpy()
    rm = oracle.simple_vfe(rv, rn, nf) if M else np.zeros((0, nf), np.float32)
    if not (int(cnt3.sum()) == M == int(live.sum()) and np.array_equal(c3.cpu().numpy()[live], rc)
            and np.array_equal(n3.cpu().numpy()[live], rn) and np.allclose(m3.cpu().numpy()[live], rm, rtol=0, atol=0, equal_nan=True)
            and float(m3.cpu().numpy()[~live].__abs__().sum()) == 0.0 and (n3.cpu().numpy()[~live] == 0).all()):
        print("MISMATCH fused voxelize+mean", cfg)
        sys.exit(1)
    if mat and not np.array_equal(v3.cpu().numpy()[live].view(np.int32), rv.view(np.int32)):
        print("MISMATCH fused voxelize+mean (voxel rows)", cfg)
        sys.exit(1)
    g = oracle.grid_size(pcr, vs)
    nyy, nxx = int(g[1]), int(g[0])
    C = int(rng.choice([1, 5, 16, 64]))
    feats = rng.standard_normal((M, C)).astype(np.float32)
    ft = torch.from_numpy(feats).cuda().requires_grad_(True)
    cv = pillar_scatter(ft, c, B, nyy, nxx)
    if not np.array_equal(cv.detach().cpu().numpy(), oracle.pillar_scatter(feats, rc, B, nyy, nxx)):
        print("MISMATCH scatter", cfg)
        sys.exit(1)
    go = rng.standard_normal((B, C, nyy, nxx)).astype(np.float32)
    cv.backward(torch.from_numpy(go).cuda())
    if not np.array_equal(ft.grad.cpu().numpy(), oracle.pillar_scatter_backward(go, rc)):
        print("MISMATCH scatter backward", cfg)
        sys.exit(1)
    # round 4: the scatter at a sampled resolution (what a nearest resize by an integer ratio reads), map form on the compact
    # rows and, where the canvas is the voxel grid (one z layer), table form on the fixed-capacity rows -- against the oracle's
    # full canvas sampled [..., ::sy, ::sx] and its backward of the zero-stuffed gradient
    divs = lambda n: [d for d in (1, 2, 3, 4, 5, 8) if n % d == 0]
    sy, sx = int(rng.choice(divs(nyy))), int(rng.choice(divs(nxx)))
    C4 = int(rng.choice([4, 8, 64]))
    feats4 = rng.standard_normal((M, C4)).astype(np.float32)
    if gz == 1:      # (several z layers put several voxels into one canvas cell: the last-writer rule is the full scatter's test)
        f4 = torch.from_numpy(feats4).cuda().requires_grad_(True)
        small = pillar_scatter_strided(f4, c, B, nyy, nxx, sy, sx)
        ref_small = oracle.pillar_scatter(feats4, rc, B, nyy, nxx)[..., ::sy, ::sx]
        if not np.array_equal(small.detach().cpu().numpy(), ref_small):
            print("MISMATCH strided scatter (map form)", cfg, sy, sx)
            sys.exit(1)
        go4 = rng.standard_normal(ref_small.shape).astype(np.float32)
        small.backward(torch.from_numpy(go4).cuda())
        g_full = np.zeros((B, C4, nyy, nxx), np.float32)
        g_full[..., ::sy, ::sx] = go4
        if not np.array_equal(f4.grad.cpu().numpy(), oracle.pillar_scatter_backward(g_full, rc)):
            print("MISMATCH strided scatter backward (map form)", cfg, sy, sx)
            sys.exit(1)
        if sum(f.shape[0] for f in frames) < (1 << 23):
            _, n5, c5, cnt5, m5, table = hard_voxelize_mean_batch(dev, vs, pcr, T, V, nf, materialize_voxels=False, return_table=True)
            live5 = (c5[:, 0] >= 0).cpu().numpy()
            full_rows = rng.standard_normal((B * V, C4)).astype(np.float32)
            f5 = torch.from_numpy(full_rows).cuda().requires_grad_(True)
            small5 = pillar_scatter_strided(f5, c5, B, nyy, nxx, sy, sx, table=table, max_voxels=V)
            ref5 = oracle.pillar_scatter(full_rows[live5], rc, B, nyy, nxx)[..., ::sy, ::sx]
            if not np.array_equal(small5.detach().cpu().numpy(), ref5):
                print("MISMATCH strided scatter (table form)", cfg, sy, sx)
                sys.exit(1)
            small5.backward(torch.from_numpy(go4).cuda())
            gf5 = f5.grad.cpu().numpy()
            if not (np.array_equal(gf5[live5], oracle.pillar_scatter_backward(g_full, rc)) and float(np.abs(gf5[~live5]).sum()) == 0.0):
                print("MISMATCH strided scatter backward (table form)", cfg, sy, sx)
                sys.exit(1)
print("fuzz ok:", it, "random LiDAR configurations")
