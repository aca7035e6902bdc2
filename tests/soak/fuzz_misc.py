#!/usr/bin/env python3
"""Soak / fuzz of the remaining HIP ops (DCN, lift, fused lift-splat backward, quantise, frustum
geometry) against their references (run on the GPU box).  Exits non-zero on the first mismatch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import oracle
from mm_training_amd.layers.nets import DeformConv2dPack
from mm_training_amd.ops.bev_geometry import frustum_geometry, lift_features, lift_splat, quantize_geometry
from mm_training_amd.ops.deform_conv import deform_conv3x3
from mm_training_amd.ops.voxel_pooling import voxel_pooling

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
torch.manual_seed(seed)
t_end = time.time() + budget
it = 0
verbose = os.environ.get("FUZZ_VERBOSE")


def fail(what, cfg):
    print("MISMATCH", what, cfg)
    sys.exit(1)


_next_note = time.time() + 60.0     # a line a minute: a silent GPU job is taken to be hung
while time.time() < t_end:
    if time.time() > _next_note:
        print("...", it, "configurations so far", flush=True)
        _next_note = time.time() + 60.0
    it += 1
    # ---- DCN
    groups = int(rng.choice([1, 2, 4]))
    C = groups * 4 * int(rng.integers(1, 9))
    O = groups * int(rng.integers(1, 17))
    B, H, W = int(rng.integers(1, 4)), int(rng.integers(2, 20)), int(rng.integers(2, 30))   # H or W == 1: the grid_sample reference degenerates (align_corners)
    if it % 3 == 0:
        # every third configuration: a shape of the implicit-GEMM form (round 6, csrc/deform_conv_mfma.hip: C/groups a multiple of 64,
        # O/groups 64 or 128): frames on both sides of the gather form's 768 pixels, pixel counts that are multiples of nothing
        C = groups * int(rng.choice([64, 128]))
        O = groups * int(rng.choice([64, 128]))
        H, W = (int(rng.integers(2, 28)), int(rng.integers(2, 30))) if rng.random() < 0.8 else (int(rng.integers(20, 60)), int(rng.integers(30, 50)))
    cfg = dict(it=it, op="dcn", B=B, C=C, O=O, H=H, W=W, groups=groups)
    if verbose:
        print(cfg, flush=True)
    m = DeformConv2dPack(C, O, groups=groups).cuda()
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    off = torch.randn(B, 18, H, W, device="cuda") * float(rng.choice([0.3, 1.0, 3.0]))
    xa, oa = x.clone().requires_grad_(True), off.clone().requires_grad_(True)
    xb, ob = x.clone().requires_grad_(True), off.clone().requires_grad_(True)
    ya = deform_conv3x3(xa, oa, m.weight, groups)
    yb = m.forward_reference(xb, ob)
    sc = max(1.0, yb.abs().max().item())
    if (ya - yb).abs().max().item() > 1e-4 * sc:
        fail("dcn forward", cfg)
    go = torch.randn_like(yb)
    ga = torch.autograd.grad(ya, (xa, oa, m.weight), go)
    gb = torch.autograd.grad(yb, (xb, ob, m.weight), go)
    for a_, b_, nm in zip(ga, gb, ("dx", "doffset", "dweight")):
        bad = ((a_ - b_).abs() > 3e-4 * max(1.0, b_.abs().max().item())).sum().item()
        # the bilinear derivative jumps at integer coordinates; the reference reaches the sampling
        # position through grid normalisation (+-1 ulp), so a point within ~1e-6 of an integer may take
        # the other one-sided derivative there: tolerate a couple of isolated offset-gradient entries
        if bad > (2 if nm == "doffset" else 0):
            fail("dcn " + nm, cfg)
    # ---- lift + fused lift-splat backward vs the unfused chain
    Bc, N, D = int(rng.integers(1, 3)), int(rng.integers(1, 4)), int(rng.integers(1, 30))
    fH, fW = int(rng.integers(1, 6)), int(rng.integers(1, 20))
    Cc = int(rng.choice([16, 32, 64, 80, 96, 128]))
    cfg = dict(it=it, op="lift", B=Bc, N=N, D=D, fH=fH, fW=fW, C=Cc)
    if verbose:
        print(cfg, flush=True)
    depth = torch.rand(Bc * N, D, fH, fW, device="cuda").softmax(1)
    ctx = torch.randn(Bc * N, Cc, fH, fW, device="cuda")
    nx, ny = int(rng.integers(1, 30)), int(rng.integers(1, 30))
    geom = torch.stack([torch.randint(-1, nx + 1, (Bc, N, D, fH, fW)), torch.randint(-1, ny + 1, (Bc, N, D, fH, fW)),
                        torch.zeros(Bc, N, D, fH, fW, dtype=torch.long)], -1).int().cuda()
    if rng.random() < 0.5:      # frustum-like: the rows of a column share their cell, except a random 10 % of the points
        coherent = geom[:, :, :, :1].expand_as(geom)
        geom = torch.where(torch.rand(Bc, N, D, fH, fW, 1, device="cuda") < 0.1, geom, coherent).contiguous()
    d1, c1 = depth.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
    d2, c2 = depth.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
    f = lift_features(d2, c2)
    ref_f = (depth.unsqueeze(1) * ctx.unsqueeze(2)).permute(0, 2, 3, 4, 1)
    if not torch.equal(f, ref_f.contiguous()):
        fail("lift forward", cfg)
    o2 = voxel_pooling(geom, f.view(Bc, N, D, fH, fW, Cc), [nx, ny, 1])
    # kernel family (ray walks / frustum tiles) and point order (reference / pixel-major) at random
    os.environ["MMT_LIFT_SPLAT_TILES"] = "1" if rng.random() < 0.3 else "0"
    col_bwd = bool(rng.random() < 0.5)          # matrix-core column backward (random cells: every point is a mismatch)
    if rng.random() < 0.5:
        o1 = lift_splat(geom.permute(0, 1, 3, 4, 2, 5).contiguous(), d1, c1, [nx, ny, 1], pixel_major=True, column_backward=col_bwd)
    else:
        o1 = lift_splat(geom, d1, c1, [nx, ny, 1], column_backward=col_bwd)
    if (o1 - o2).abs().max().item() > 1e-4 * max(1.0, o2.abs().max().item()):
        fail("fused forward", cfg)
    go = torch.randn_like(o2)
    o1.backward(go)
    o2.backward(go)
    if not torch.allclose(d1.grad, d2.grad, rtol=1e-4, atol=1e-4 * max(1.0, d2.grad.abs().max().item())):
        fail("fused grad_depth", cfg)
    if not torch.allclose(c1.grad, c2.grad, rtol=1e-4, atol=1e-4 * max(1.0, c2.grad.abs().max().item())):
        fail("fused grad_context", cfg)
    # ---- quantise + geometry vs the oracle (bit-exact)
    n = int(rng.integers(1, 5000))
    xyz = (rng.standard_normal((n, 3)) * 60).astype(np.float32)
    vc = [float(rng.uniform(-60, -40)), float(rng.uniform(-60, -40)), -1.0]
    vs = [float(rng.choice([0.2, 0.4, 0.8])), float(rng.choice([0.2, 0.4, 0.8])), 8.0]
    q = quantize_geometry(torch.from_numpy(xyz).cuda(), vc, vs).cpu().numpy()
    if not np.array_equal(q, oracle.quantize(xyz, vc, vs)):
        fail("quantize", dict(it=it, n=n, vc=vc, vs=vs))
    Dg, gh, gw, BN = int(rng.integers(1, 20)), int(rng.integers(1, 6)), int(rng.integers(1, 12)), int(rng.integers(1, 8))
    fr = (rng.random((Dg, gh, gw, 4)) * 50).astype(np.float32)
    fr[..., 3] = 1
    cb = rng.standard_normal((BN, 4, 4)).astype(np.float32)
    gq, gx = frustum_geometry(torch.from_numpy(fr).cuda(), torch.from_numpy(cb).cuda(), vc, vs, return_xyz=True)
    rx = oracle.geometry(fr, cb[None])[0]
    if not (np.array_equal(gx.cpu().numpy(), rx) and np.array_equal(gq.cpu().numpy(), oracle.quantize(rx, vc, vs))):
        fail("frustum geometry", dict(it=it, D=Dg, gh=gh, gw=gw, BN=BN))
    # ---- round 4: depth softmax + oracle overwrite (lss_fpn.py:423-438) vs the torch expression in fp64: any bin count up to
    # 512 (16-byte and element-wise instantiations), fp32 / bf16 logits, fp32 / bf16 depth_used, channels_last / NCHW / slices
    # of a wider channels_last tensor at aligned and unaligned offsets, with and without the oracle, one or both consumers
    from mm_training_amd.ops.bev_geometry import depth_softmax
    BNs, Ds = int(rng.integers(1, 7)), int(rng.choice([1, 3, 10, 36, 59, 112, 113, 128, 260, 409, 512]))
    hs, ws = int(rng.integers(1, 9)), int(rng.integers(1, 12))
    pad, offc = int(rng.choice([0, 4, 7, 80])), 0
    wide = torch.randn(BNs, Ds + pad, hs, ws, device="cuda") * float(rng.choice([0.5, 3.0, 20.0]))
    if pad:
        offc = int(rng.integers(0, pad + 1))
    layout = rng.choice(["cl", "nchw", "slice"])
    src = wide.contiguous(memory_format=torch.channels_last) if layout != "nchw" else wide
    src = src[:, offc:offc + Ds] if layout == "slice" or pad else src
    lg_bf16, used_bf16 = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    if lg_bf16:
        src = src.bfloat16()
    x1 = src.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
    orc = None
    if rng.integers(0, 2):
        hot = torch.randint(0, Ds, (BNs, hs, ws), device="cuda")
        orc = (torch.nn.functional.one_hot(hot, Ds).float() * (torch.rand(BNs, hs, ws, 1, device="cuda") < 0.5)).permute(0, 3, 1, 2)
    pr, us = depth_softmax(x1, orc, torch.bfloat16 if used_bf16 else torch.float32)
    x64 = src.detach().double().requires_grad_(True)
    p64 = x64.softmax(1)
    u64 = p64 if orc is None else torch.where(orc.max(1, keepdim=True).values > 0, orc.double(), p64)
    scfg = dict(it=it, BN=BNs, D=Ds, h=hs, w=ws, layout=str(layout), offc=offc, pad=pad, lg_bf16=lg_bf16, used_bf16=used_bf16, oracle=orc is not None)
    if float((pr.detach().double() - p64.detach()).abs().max()) > 1e-6:
        fail("depth softmax probs", scfg)
    if float((us.detach().double() - u64.detach()).abs().max()) > (4e-3 if used_bf16 else 1e-6):
        fail("depth softmax depth_used", scfg)
    ga, gb = (torch.rand_like(p64) - 0.5), (torch.rand_like(p64) - 0.5)
    which = int(rng.integers(0, 3))
    loss = (pr.double() * ga).sum() * (which != 1) + (us.double() * gb).sum() * (which != 0)
    l64 = (p64 * ga).sum() * (which != 1) + (u64 * gb).sum() * (which != 0)
    loss.backward()
    l64.backward()
    tol = 1e-2 if (lg_bf16 or used_bf16) else 2e-6
    if x1.grad.dtype != src.dtype or float((x1.grad.double() - x64.grad).abs().max()) > tol:
        fail("depth softmax gradient", scfg)
print("fuzz ok:", it, "random configurations of DCN / lift / fused lift-splat / quantise / geometry / depth softmax")
