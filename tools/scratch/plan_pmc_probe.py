#!/usr/bin/env python3
"""Which launch of the plan form upsets rocprofv3 --pmc?  python tools/scratch/plan_pmc_probe.py prepare|forward|both"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_training_amd import synthetic
from mm_training_amd.ops.bev_geometry import lift_splat_plan, new_plan_cache, plan_prepare

what = sys.argv[1] if len(sys.argv) > 1 else "both"
B, N, D, fH, fW, C = 2, 2, 16, 4, 6, 64
H, W = fH * 16, fW * 16
s2e, K = synthetic.camera_rig(B, N, W, H, jitter=0.02, seed=0)
combine = s2e.matmul(torch.inverse(K)).contiguous().cuda()
axes = (torch.linspace(0, W - 1, fW).cuda(), torch.linspace(0, H - 1, fH).cuda(), torch.arange(2.0, 2.0 + 2.0 * D, 2.0).cuda())
vn, vc, vs = [64, 64, 1], [-25.6 + 0.4, -25.6 + 0.4, -1.0], [0.8, 0.8, 8.0]
cache = new_plan_cache(N, D, fH, fW, vn, "cuda", slots=4)
depth = torch.rand(B * N, D, fH, fW, device="cuda").softmax(1)
ctx = torch.randn(B * N, C, fH, fW, device="cuda")
if what in ("prepare", "both"):
    plan_prepare(combine, axes, vn, vc, vs, cache)
    torch.cuda.synchronize()
    print("prepare ok", flush=True)
if what in ("forward", "both"):
    out = lift_splat_plan(combine, axes, depth, ctx, vn, vc, vs, cache, prepared=(what == "both"))
    torch.cuda.synchronize()
    print("forward ok", float(out.abs().sum()), flush=True)
