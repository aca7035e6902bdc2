#!/usr/bin/env python3
"""Soak / fuzz of the CAMERA FORM of the fused lift-splat (run on the GPU box):  python tests/soak/fuzz_camera.py [seconds] [seed]
Random rigs (yaw jitter, pitch, roll), frustum shapes on both sides of every kernel choice (register walk: columns of up to
16 rows, D < 160, C <= 80; the LDS-record walk otherwise; column / ray backward), random batches drawn from a small pool of
calibrations -- repeated, reordered, duplicated inside a batch -- through ONE persistent exclusive-cell cache per shape with
few slots (so calibrations claim, mark, verify, use, collide and start over all the time), each call against the geom form on
mmt_frustum_geometry's cells: same cells (bit for bit), same map and gradients (1e-4 of the largest value).  Exits non-zero
on the first mismatch."""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from mm_training_amd import synthetic
from mm_training_amd.ops.bev_geometry import (camera_form_supported, frustum_axes, frustum_geometry, last_kernel_family, lift_splat,
                                              lift_splat_camera, lift_splat_plan, new_exclusive_cache, new_plan_cache, plan_cache_counters,
                                              plan_form_supported)
from tests.test_oracle_golden import _frustum_torch

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
t_end = time.time() + budget
_next_note = time.time() + 60.0


def fail(what, cfg):
    print("MISMATCH:", what, cfg, flush=True)
    sys.exit(1)


def rot(axis, deg):
    c, s = math.cos(math.radians(deg)), math.sin(math.radians(deg))
    m = torch.eye(4)
    i, j = {"x": (1, 2), "y": (0, 2), "z": (0, 1)}[axis]
    m[i, i], m[i, j], m[j, i], m[j, j] = c, -s, s, c
    return m


it = calls = used = plan_calls = plan_brute = 0
while time.time() < t_end:
    it += 1
    N = int(rng.integers(1, 7))
    fH = int(rng.choice([1, 3, 8, 16, 16, 17, 24]))
    fW = int(rng.integers(2, 30))
    D = int(rng.choice([5, 40, 112, 112, 150, 170]))
    C = int(rng.choice([64, 80, 80, 128]))
    B = int(rng.integers(1, 5))
    ds = 16
    H, W = fH * ds, fW * ds
    if not camera_form_supported(B, N, D, fH, fW, C):
        continue
    fr = _frustum_torch((H, W), ds, (2.0, 2.0 + 0.5 * D, 0.5))
    assert fr.shape[0] == D
    axes = [t.cuda() for t in frustum_axes(fr)]
    vs = [float(rng.choice([0.4, 0.8])), float(rng.choice([0.4, 0.8])), 8.0]
    nx, ny = int(rng.choice([64, 128, 200])), int(rng.choice([64, 128]))
    vc = [-nx * vs[0] / 2 + vs[0] / 2, -ny * vs[1] / 2 + vs[1] / 2, -1.0]
    vn = [nx, ny, 1]
    # a pool of calibrations: level rigs and tilted ones (columns that straddle cell borders)
    pool = []
    for k in range(int(rng.integers(2, 6))):
        s2e, K = synthetic.camera_rig(1, N, W, H, jitter=0.3, seed=int(rng.integers(1 << 30)))
        tilt = rot("x", float(rng.uniform(-6, 6))) if rng.random() < 0.4 else torch.eye(4)
        roll = rot("z", float(rng.uniform(-3, 3))) if rng.random() < 0.2 else torch.eye(4)
        pool.append(s2e.matmul(tilt).matmul(roll).matmul(torch.inverse(K)).contiguous())
    slots = int(rng.integers(1, 5))
    cache = new_exclusive_cache(N, vn, "cuda", slots)
    pcache = new_plan_cache(N, D, fH, fW, vn, "cuda", slots=B + int(rng.integers(0, 2))) if plan_form_supported(B, N, D, fH, fW, C, vn) else None
    column = bool(rng.random() < 0.5)
    cfg = dict(it=it, seed=seed, B=B, N=N, D=D, fH=fH, fW=fW, C=C, grid=vn, vs=vs, pool=len(pool), slots=slots, column=column)
    for call in range(int(rng.integers(3, 9))):
        if time.time() > _next_note:
            print("...", it, "shapes,", calls, "calls so far", flush=True)
            _next_note = time.time() + 60.0
        pick = [int(rng.integers(len(pool))) for _ in range(B)]
        cb = torch.cat([pool[p] for p in pick], 0).cuda()
        depth = torch.rand(B * N, D, fH, fW, device="cuda").softmax(1)
        ctx = torch.randn(B * N, C, fH, fW, device="cuda")
        d1, c1 = depth.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
        d2, c2 = depth.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
        o1 = lift_splat_camera(cb, axes, d1, c1, vn, vc, vs, column_backward=column, exclusive_cache=cache)
        fam = last_kernel_family(detail=True)
        geom = frustum_geometry(fr.cuda(), cb, vc, vs)                                  # [B, N, D, fH, fW, 3]
        o2 = lift_splat(geom.permute(0, 1, 3, 4, 2, 5).contiguous(), d2, c2, vn, pixel_major=True, column_backward=column)
        scale = max(1.0, float(o2.abs().max()))
        # (cells, not elements: a channel sum that cancels to exactly 0 in one order of additions need not in another)
        if float((o1 - o2).abs().max()) > 1e-4 * scale or not torch.equal((o1 != 0).any(1), (o2 != 0).any(1)):
            bad = ((o1 - o2).abs() > 1e-4 * scale).nonzero()
            fail("forward (family %s, call %d, picks %s, header %s): max diff %g of scale %g at %d elements (first %s), %d cells differ in being zero"
                 % (fam, call, pick, cache[:40].tolist(), float((o1 - o2).abs().max()), scale, len(bad), bad[:3].tolist(),
                    int(((o1 != 0).any(1) != (o2 != 0).any(1)).sum())), cfg)
        go = torch.randn_like(o2)
        o1.backward(go)
        o2.backward(go)
        if not torch.allclose(d1.grad, d2.grad, rtol=1e-4, atol=1e-4 * max(1.0, float(d2.grad.abs().max()))):
            fail("grad_depth (call %d)" % call, cfg)
        if not torch.allclose(c1.grad, c2.grad, rtol=1e-4, atol=1e-4 * max(1.0, float(c2.grad.abs().max()))):
            fail("grad_context (call %d)" % call, cfg)
        if pcache is not None:
            brute = bool(rng.random() < 0.1)
            d3, c3 = depth.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
            o3 = lift_splat_plan(cb, axes, d3, c3, vn, vc, vs, pcache, column_backward=column, brute=brute)
            if last_kernel_family() != "plan+camera":
                fail("plan form: family %s" % last_kernel_family(), cfg)
            if float((o3 - o2).abs().max()) > 1e-4 * scale or not torch.equal((o3 != 0).any(1), (o2 != 0).any(1)):
                fail("plan forward (call %d, picks %s, brute %s, counters %s): max diff %g of scale %g, %d cells differ in being zero"
                     % (call, pick, brute, plan_cache_counters(pcache), float((o3 - o2).abs().max()), scale, int(((o3 != 0).any(1) != (o2 != 0).any(1)).sum())), cfg)
            o3b = lift_splat_plan(cb, axes, depth, ctx, vn, vc, vs, pcache, brute=brute)
            if not torch.equal(o3b, o3.detach()):
                fail("plan forward not bit-identical on repeat (call %d, brute %s)" % (call, brute), cfg)
            o3.backward(go)
            if not (torch.equal(d3.grad, d1.grad) and torch.equal(c3.grad, c1.grad)):
                fail("plan form: gradients differ from the camera form's (call %d)" % call, cfg)
            plan_calls += 1
            plan_brute += int(brute)
        calls += 1
        if "exclusive" in fam:
            used += int(3 in cache[24:24 + min(B, 8)].tolist())
print("fuzz ok:", it, "shapes,", calls, "calls of the camera form against the geom form;", used, "of them stored single-run cells;",
      plan_calls, "calls of the plan form (", plan_brute, "through its brute-force path )")
