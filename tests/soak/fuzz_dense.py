#!/usr/bin/env python3
"""Soak test of the round-1 additions against their references (run on the GPU box):
fused BatchNorm(+residual)(+ReLU) vs nn.BatchNorm2d + add + relu, the BEV warp vs the oracle,
depth labels vs the oracle, CenterPoint targets vs the oracle.  Random shapes, exits non-zero on
the first mismatch.   python tests/soak/fuzz_dense.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from torch import nn

import oracle
from mm_training_amd import synthetic
from mm_training_amd.ops import bn_relu
from mm_training_amd.ops.bev_warp import bev_warp_affine
from mm_training_amd.ops.train_targets import centerpoint_targets, depth_labels

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
it = 0


def fail(what, **kw):
    print("MISMATCH", what, kw)
    sys.exit(1)


_next_note = time.time() + 60.0     # a line a minute: a silent GPU job is taken to be hung
while time.time() < t_end:
    if time.time() > _next_note:
        print("...", it, "configurations so far", flush=True)
        _next_note = time.time() + 60.0
    it += 1
    VERB = os.environ.get("FUZZ_VERBOSE")
    # ---- fused BN
    B, H, W = int(rng.integers(1, 5)), int(rng.integers(1, 40)), int(rng.integers(1, 40))
    C = int(rng.choice([4, 8, 12, 64, 100, 160, 256, 320, 640, 1024, 2048]))
    use_res, relu = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    if B * H * W < 2:
        H = 2
    shape = (B, C, H, W)
    if VERB:
        print("bn", it, shape, use_res, relu, flush=True)
    g = torch.Generator(device="cuda").manual_seed(it)
    x0 = (torch.randn(shape, device="cuda", generator=g) * 1.3 + 0.2).contiguous(memory_format=torch.channels_last)
    r0 = torch.randn(shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last) if use_res else None
    go = torch.randn(shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    # reference: the BatchNorm formulas in float64 torch ops (autograd for the gradients).  NOT
    # nn.BatchNorm2d: MIOpen's NHWC batch-norm segfaults on some of these shapes (e.g. [1,320,31,15]).
    bn = nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, device="cuda", generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, device="cuda", generator=g) * 0.3)
    x = x0.clone().requires_grad_(True)
    r = r0.clone().requires_grad_(True) if use_res else None
    y = bn_relu.bn_act(bn, x, r, relu)
    if not isinstance(y.grad_fn, torch.autograd.function.BackwardCFunction):
        fail("bn fused path not taken", shape=shape)
    y.backward(go)
    xd = x0.double().requires_grad_(True)
    rd = r0.double().requires_grad_(True) if use_res else None
    wd, bd = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    mean = xd.mean((0, 2, 3), keepdim=True)
    var = xd.var((0, 2, 3), unbiased=False, keepdim=True)
    yd = (xd - mean) / torch.sqrt(var + bn.eps) * wd.view(1, -1, 1, 1) + bd.view(1, -1, 1, 1)
    if use_res:
        yd = yd + rd
    if relu:
        # the mask of the fused output: an activation within fp32 rounding of zero may land on either side,
        # which would change a whole channel's gradient statistics
        yd = yd * (y.detach() > 0).double()
    yd.backward(go.double())
    n = B * H * W
    rm = 0.1 * mean.flatten()
    rv = 0.9 + 0.1 * var.flatten() * (n / max(n - 1, 1))
    pairs = [("y", y.detach(), yd.detach()), ("gx", x.grad, xd.grad), ("gres", r.grad if use_res else None, rd.grad if use_res else None),
             ("gw", bn.weight.grad, wd.grad), ("gb", bn.bias.grad, bd.grad), ("rm", bn.running_mean, rm), ("rv", bn.running_var, rv)]
    for name, a, b in pairs:
        if a is None:
            continue
        tol = (2e-4 if name in ("gw", "gb") else 5e-5) * max(1.0, b.abs().max().item()) * (8.0 if n < 32 else 1.0)
        bad = ((a.double() - b).abs() > tol).float().mean().item()
        if bad > (2e-4 if name in ("y", "gx", "gres") else 0.0) and n >= 8:     # tiny batches: rstd amplifies rounding
            fail("bn", shape=shape, res=use_res, relu=relu, tensor=name, bad=bad, err=(a.double() - b).abs().max().item())
    # ---- the same layer with bf16 activations inside autocast (mmt_bn_relu_*_ex): fp32 statistics and arithmetic on the bf16 values
    if n >= 8:
        xb, rb, gb16 = x0.bfloat16(), (r0.bfloat16() if use_res else None), go.bfloat16()
        bn2 = nn.BatchNorm2d(C).cuda()
        bn2.load_state_dict({k: (v if "running" not in k and "num" not in k else bn2.state_dict()[k]) for k, v in bn.state_dict().items()})
        xq = xb.clone().requires_grad_(True)
        rq = rb.clone().requires_grad_(True) if use_res else None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yq = bn_relu.bn_act(bn2, xq, rq, relu)
        if yq.dtype != torch.bfloat16 or not isinstance(yq.grad_fn, torch.autograd.function.BackwardCFunction):
            fail("bn bf16 fused path not taken", shape=shape)
        yq.backward(gb16)
        xd = xb.double().requires_grad_(True)
        rd = rb.double().requires_grad_(True) if use_res else None
        wd, bd = bn2.weight.detach().double().requires_grad_(True), bn2.bias.detach().double().requires_grad_(True)
        mean = xd.mean((0, 2, 3), keepdim=True)
        var = xd.var((0, 2, 3), unbiased=False, keepdim=True)
        yd = (xd - mean) / torch.sqrt(var + bn2.eps) * wd.view(1, -1, 1, 1) + bd.view(1, -1, 1, 1)
        if use_res:
            yd = yd + rd
        if relu:
            yd = yd * (yq.detach() > 0).double()
        yd.backward(gb16.double())
        ulp = 2.0 ** -7
        pairs = [("y", yq.detach(), yd.detach(), ulp), ("gx", xq.grad, xd.grad, 2 * ulp), ("gres", rq.grad if use_res else None, rd.grad if use_res else None, ulp),
                 ("gw", bn2.weight.grad, wd.grad, 2e-4), ("gb", bn2.bias.grad, bd.grad, 2e-4),
                 ("rm", bn2.running_mean, 0.1 * mean.flatten(), 5e-5), ("rv", bn2.running_var, 0.9 + 0.1 * var.flatten() * (n / max(n - 1, 1)), 5e-5)]
        for name, a, b, rel in pairs:
            if a is None:
                continue
            tol = rel * max(1.0, b.abs().max().item()) * (8.0 if n < 32 else 1.0)
            bad = ((a.double() - b).abs() > tol).float().mean().item()
            if bad > (2e-4 if name in ("y", "gx", "gres") else 0.0):
                fail("bn bf16", shape=shape, res=use_res, relu=relu, tensor=name, bad=bad, err=(a.double() - b).abs().max().item())
    # ---- BEV warp
    B, H, W, C = int(rng.integers(1, 4)), int(rng.integers(2, 70)), int(rng.integers(2, 70)), int(rng.choice([4, 16, 80]))
    if VERB:
        torch.cuda.synchronize(); print("warp", (B, H, W, C), flush=True)
    xw = rng.standard_normal((B, H, W, C)).astype(np.float32)
    bda = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    for b in range(B):
        a_, s_ = rng.uniform(-3.2, 3.2), rng.uniform(0.7, 1.3)
        bda[b, :2, :2] = np.array([[np.cos(a_), -np.sin(a_)], [np.sin(a_), np.cos(a_)]]) * s_
        if rng.integers(0, 2):
            bda[b, :2, 1] *= -1
    yw = bev_warp_affine(torch.from_numpy(xw).cuda().permute(0, 3, 1, 2), torch.from_numpy(bda).cuda())
    ref = oracle.bev_warp_affine(xw, bda)
    if np.abs(yw.permute(0, 2, 3, 1).cpu().numpy() - ref).max() > 1e-5:
        fail("warp", shape=(B, H, W, C), err=float(np.abs(yw.permute(0, 2, 3, 1).cpu().numpy() - ref).max()))
    # backward (gather over the candidate output cells): adjoint identity <warp(x), g> == <x, warp^T(g)> in float64,
    # any rotation / flip / scale 0.7-1.3, and bit-reproducibility
    xg = torch.from_numpy(xw).cuda().permute(0, 3, 1, 2).requires_grad_(True)
    gw_ = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).cuda()
    yg = bev_warp_affine(xg, torch.from_numpy(bda).cuda())
    yg.backward(gw_)
    lhs = (yg.detach().double() * gw_.double()).sum().item()
    rhs = (xg.detach().double() * xg.grad.double()).sum().item()
    if abs(lhs - rhs) > 1e-6 * (yg.detach().abs().double() * gw_.abs().double()).sum().item() + 1e-6:
        fail("warp_backward_adjoint", shape=(B, H, W, C), lhs=lhs, rhs=rhs)
    g1 = xg.grad.clone()
    xg.grad = None
    bev_warp_affine(xg, torch.from_numpy(bda).cuda()).backward(gw_)
    if not torch.equal(g1, xg.grad):
        fail("warp_backward_not_reproducible", shape=(B, H, W, C))
    # ---- depth labels
    B, N = int(rng.integers(1, 4)), int(rng.integers(1, 7))
    H, W, ds = [(64, 96, 16), (256, 704, 16), (128, 352, 8), (32, 32, 4)][int(rng.integers(0, 4))]
    d_bound = [(2.0, 58.0, 0.5), (1.0, 60.0, 0.5)][int(rng.integers(0, 2))]
    if VERB:
        torch.cuda.synchronize(); print("labels", B, N, H, W, ds, flush=True)
    D = int((d_bound[1] - d_bound[0]) / d_bound[2])
    s2e, K = synthetic.camera_rig(B, N, W, H, jitter=0.05, seed=it)
    extr = torch.inverse(s2e)
    clouds = [np.concatenate([rng.uniform(-60, 60, (n, 2)), rng.uniform(-4, 4, (n, 1)), rng.uniform(0, 1, (n, 2))], 1).astype(np.float32)
              for n in rng.choice([0, 1, 300, 20000], B)]
    if sum(len(c) for c in clouds) == 0:
        clouds[0] = np.array([[10.0, 0.5, 0.0, 0.1, 0.2]], np.float32)
    eye = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    rb, ro = oracle.depth_labels(clouds, extr.numpy(), K.numpy(), eye, (H, W), ds, d_bound)
    oh, bins = depth_labels([torch.from_numpy(c).cuda() for c in clouds], extr.cuda(), K.cuda(), torch.from_numpy(eye).cuda(),
                            (H, W), ds, d_bound, D, return_bins=True)
    if not np.array_equal(bins.cpu().numpy(), rb) or not np.array_equal(oh.cpu().numpy(), ro):
        fail("depth_labels", B=B, N=N, H=H, W=W, bad=int((bins.cpu().numpy() != rb).sum()))
    # ---- CenterPoint targets
    B = int(rng.integers(1, 5))
    class_counts = [[1, 1, 1, 1], [2, 1, 3], [4], [1]][int(rng.integers(0, 4))]
    fx, fy, osf = [(128, 128, 4), (512, 64, 4), (40, 56, 2), (7, 5, 1)][int(rng.integers(0, 4))]
    vs = (0.2, 0.2, 8.0)
    pc = (-vs[0] * osf * fx / 2, -vs[1] * osf * fy / 2, -5.0, vs[0] * osf * fx / 2, vs[1] * osf * fy / 2, 3.0)
    max_objs = int(rng.choice([500, 12, 3]))
    if VERB:
        torch.cuda.synchronize(); print("targets", B, class_counts, fx, fy, max_objs, flush=True)
    bxs, lbs = [], []
    for b in range(B):
        k = int(rng.choice([0, 1, 7, 40]))
        xy = rng.uniform([pc[0] - 3, pc[1] - 3], [pc[3] + 3, pc[4] + 3], (k, 2))
        bxs.append(np.concatenate([xy, rng.uniform(-2, 1, (k, 1)), rng.uniform(0.0, 30.0, (k, 3)), rng.uniform(-3.2, 3.2, (k, 1)),
                                   rng.normal(size=(k, 2))], 1).astype(np.float32))
        lbs.append(rng.integers(0, sum(class_counts), k).astype(np.int64))
    if sum(len(b) for b in bxs) == 0:
        bxs[0] = np.array([[0.1, 0.1, 0, 2, 4, 1.5, 0.3, 1, 0]], np.float32)
        lbs[0] = np.array([0])
    hm, anno, ind, mask = centerpoint_targets([torch.from_numpy(b).cuda() for b in bxs], [torch.from_numpy(l).cuda() for l in lbs],
                                              class_counts, max_objs, (fx, fy), pc, vs, osf, 0.1, 2)
    begin = 0
    for t, n in enumerate(class_counts):
        for b in range(B):
            r_hm, r_anno, r_ind, r_mask = oracle.centerpoint_targets_task(bxs[b], lbs[b], begin, n, max_objs,      # (the cut at max_objs is per task, after the regrouping by class: bev_depth_head.py:186)
                                                                          fx, fy, pc, vs, osf, 0.1, 2)
            if np.abs(hm[t][b].cpu().numpy() - r_hm).max() > 1e-6 or int(mask[t][b].sum()) != int(r_mask.sum()):
                fail("targets", B=B, task=t, b=b, fx=fx, fy=fy, max_objs=max_objs)
        begin += n
print("fuzz ok:", it, "random rounds")
