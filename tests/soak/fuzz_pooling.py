#!/usr/bin/env python3
"""Soak / fuzz of the voxel_pooling kernels against the oracle (run on the GPU box):
random shapes, clustered / uniform / degenerate geometry, every forward algorithm, both
backward paths and the fused lift-splat.  Exits non-zero on the first mismatch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import oracle
from mm_training_amd.ops.bev_geometry import lift_splat
from mm_training_amd.ops.voxel_pooling import VoxelPoolingPlan, voxel_pooling_ext
from mm_training_amd.ops.voxel_pooling.plan import planned_forward_into

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
    P = int(rng.choice([17, 500, 512, 1024, 3000, 20000, 70400, 150000]))
    C = int(rng.choice([4, 16, 64, 80, 128, 256, 36, 7, 320]))
    nx, ny, nz = int(rng.integers(1, 200)), int(rng.integers(1, 200)), int(rng.integers(1, 3))
    mode = rng.choice(["uniform", "cluster", "onecell", "runs", "alldrop"])
    if mode == "uniform":
        g = np.stack([rng.integers(-3, nx + 3, (B, P)), rng.integers(-3, ny + 3, (B, P)), rng.integers(-1, nz + 1, (B, P))], -1)
    elif mode == "cluster":
        cx, cy = rng.integers(0, nx, 8), rng.integers(0, ny, 8)
        k = rng.integers(0, 8, (B, P))
        g = np.stack([cx[k] + rng.integers(-2, 3, (B, P)), cy[k] + rng.integers(-2, 3, (B, P)), rng.integers(0, nz, (B, P))], -1)
    elif mode == "onecell":
        g = np.zeros((B, P, 3), np.int64); g[..., 0] = nx - 1; g[..., 1] = ny // 2
    elif mode == "runs":
        base = np.arange(P) // int(rng.integers(1, 40))
        g = np.stack([np.broadcast_to(base % nx, (B, P)), np.broadcast_to((base // nx) % ny, (B, P)), np.zeros((B, P), np.int64)], -1)
    else:
        g = np.full((B, P, 3), -5, np.int64)
    geom = np.ascontiguousarray(g.astype(np.int32))
    feats = (rng.random((B, P, C), dtype=np.float32) - 0.5)
    ref64 = oracle.voxel_pooling_forward_f64(geom, feats, nx, ny, nz)
    _, ref_pos = oracle.voxel_pooling_forward(geom, np.zeros((B, P, 1), np.float32), nx, ny, nz)
    gd, fd = torch.from_numpy(geom).cuda(), torch.from_numpy(feats).cuda()
    verbose = os.environ.get("FUZZ_VERBOSE")
    for algo in (0, 1, 2, 3, 4, 0x23):
        if verbose:
            print("cfg", dict(it=it, B=B, P=P, C=C, grid=(nx, ny, nz), mode=str(mode), algo=algo), flush=True)
        out = torch.zeros(B, ny, nx, C, device="cuda")
        pos = torch.empty(B, P, 3, dtype=torch.int32, device="cuda")
        voxel_pooling_ext.voxel_pooling_forward_wrapper(B, P, C, nx, ny, nz, gd, fd, out, pos, flags=algo | 0x10)
        if verbose:
            torch.cuda.synchronize()
        err = np.abs(out.cpu().numpy() - ref64).max()
        # fp32 accumulation of up to P addends per cell in arbitrary order: ~eps*sqrt(n)*|sum|
        tol = 1e-4 + 1e-6 * np.sqrt(P) * max(1.0, np.abs(ref64).max())
        if not np.array_equal(pos.cpu().numpy(), ref_pos) or err > tol:
            print("MISMATCH forward", dict(it=it, B=B, P=P, C=C, grid=(nx, ny, nz), mode=mode, algo=algo, err=float(err)))
            sys.exit(1)
    if C % 4 == 0 and C <= 256:   # cached-plan forward
        plan = VoxelPoolingPlan(gd, [nx, ny, nz])
        pout = torch.full((B, ny, nx, C), 3.0, device="cuda")
        planned_forward_into(plan, fd, pout, C)
        err = np.abs(pout.cpu().numpy() - ref64).max()
        if not np.array_equal(plan.pos_memo.cpu().numpy(), ref_pos) or err > tol:
            print("MISMATCH planned", dict(it=it, B=B, P=P, C=C, grid=(nx, ny, nz), mode=mode, err=float(err)))
            sys.exit(1)
    go = rng.standard_normal((B, ny, nx, C)).astype(np.float32)
    ref_gi = oracle.voxel_pooling_backward(ref_pos, go.transpose(0, 3, 1, 2))
    god = torch.from_numpy(go).cuda().permute(0, 3, 1, 2)
    posd = torch.from_numpy(ref_pos).cuda()
    for grad in (god, god.contiguous()):
        for ws in (None, torch.empty(voxel_pooling_ext.backward_workspace_elems(B, P, C, nx, ny), device="cuda")):
            gi = torch.empty(B, P, C, device="cuda")
            if verbose:
                print("bwd", dict(it=it, nchw=bool(grad.stride(1) != 1), ws=ws is not None), flush=True)
            voxel_pooling_ext.voxel_pooling_backward_wrapper(B, P, C, nx, ny, posd, grad, gi, ws)
            if verbose:
                torch.cuda.synchronize()
            if not np.array_equal(gi.cpu().numpy(), ref_gi):
                print("MISMATCH backward", dict(it=it, B=B, P=P, C=C, grid=(nx, ny, nz), mode=mode))
                sys.exit(1)
    # bf16 feature storage: fp32 accumulate vs the oracle on the rounded rows; grad_in = bf16(gather)
    if C % 8 == 0 and C <= 512:
        f16 = fd.bfloat16()
        r16 = oracle.voxel_pooling_forward_f64(geom, f16.float().cpu().numpy(), nx, ny, nz)
        out = torch.zeros(B, ny, nx, C, device="cuda")
        pos = torch.empty(B, P, 3, dtype=torch.int32, device="cuda")
        voxel_pooling_ext.voxel_pooling_forward_wrapper_bf16(B, P, C, nx, ny, nz, gd, f16, out, pos, flags=0x10)
        err = np.abs(out.cpu().numpy() - r16).max()
        if not np.array_equal(pos.cpu().numpy(), ref_pos) or err > tol:
            print("MISMATCH bf16 forward", dict(it=it, B=B, P=P, C=C, grid=(nx, ny, nz), mode=mode, err=float(err)))
            sys.exit(1)
        ws = torch.empty(voxel_pooling_ext.backward_workspace_elems(B, P, C, nx, ny), device="cuda")
        for grad in (god, god.contiguous()):
            gi = torch.empty(B, P, C, dtype=torch.bfloat16, device="cuda")
            voxel_pooling_ext.voxel_pooling_backward_wrapper_bf16(B, P, C, nx, ny, posd, grad, gi, ws)
            if not torch.equal(gi.cpu(), torch.from_numpy(ref_gi).bfloat16()):
                print("MISMATCH bf16 backward", dict(it=it, B=B, P=P, C=C, grid=(nx, ny, nz), mode=mode))
                sys.exit(1)
    # fused lift-splat on a factorised shape (frustum-tile forward; MMT_LIFT_SPLAT_V1=1 selects the chunked kernel)
    if C % 16 == 0 and C <= 256 and P % 6 == 0:
        N, HW = 2, 3
        D = P // (N * HW)
        if D * N * HW == P and D >= 1:
            depth = rng.random((B * N, D, 1, HW), dtype=np.float32)
            ctx = rng.standard_normal((B * N, C, 1, HW)).astype(np.float32)
            f2 = oracle.lift(depth, ctx).reshape(B, P, C)
            r2 = oracle.voxel_pooling_forward_f64(geom, f2, nx, ny, nz)
            if verbose:
                print("fused", dict(it=it, N=N, D=D, HW=HW), flush=True)
            o2 = lift_splat(gd.view(B, N, D, 1, HW, 3), torch.from_numpy(depth).cuda(), torch.from_numpy(ctx).cuda(), [nx, ny, nz])
            err = np.abs(o2.permute(0, 2, 3, 1).cpu().numpy() - r2).max()
            if err > 1e-4 + 1e-6 * np.sqrt(P) * max(1.0, np.abs(r2).max()):
                print("MISMATCH fused", dict(it=it, B=B, P=P, C=C, mode=mode, err=float(err)))
                sys.exit(1)
print("fuzz ok:", it, "random configurations")
