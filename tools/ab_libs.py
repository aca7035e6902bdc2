#!/usr/bin/env python3
"""Interleaved A/B of two builds of libmmt_hip.so in ONE process (run on the GPU box):
   python tools/ab_libs.py path/to/libA.so path/to/libB.so [--shape cfg2] [--rounds 12]
Process-to-process and box-to-box noise of these bandwidth-bound kernels is ~5 %, so only numbers
taken in alternation inside one process are comparable.  Times the drop-in forward, the backward
(alternating with the forward, as in a training step) and checks both builds agree."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mm_training_amd import _lib, synthetic  # noqa: E402
from tools.kbench import SHAPES  # noqa: E402


def load(path):
    h = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs=2)
    ap.add_argument("--shape", default="cfg2")
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--flags", type=lambda v: int(v, 0), default=3 | 0x10)
    ap.add_argument("--env-ab", default="", help="NAME=v1,v2,..: alternate this environment variable (a diagnostic switch read by "
                                                 "the SECOND library per call) and time forward / backward for every value")
    ap.add_argument("--flush", action="store_true", help="with --env-ab: a 1 GiB read before every backward (cold Infinity Cache)")
    args = ap.parse_args()
    sh = SHAPES[args.shape]
    B, C = sh["B"], sh["C"]
    geom, vn = synthetic.rig_geometry(B, sh["N"], sh["final_dim"], sh["ds"], sh["d_bound"],
                                      sh.get("x_bound", (-51.2, 51.2, 0.8)), sh.get("y_bound", (-51.2, 51.2, 0.8)))
    nx, ny, nz = vn
    P = geom[0].numel() // 3
    feats = synthetic.features((B, P, C), seed=1).cuda()
    geom = geom.reshape(B, P, 3).cuda()
    out = torch.zeros(B, ny, nx, C, device="cuda")
    pos = torch.empty(B, P, 3, dtype=torch.int32, device="cuda")
    go = torch.randn(B, ny, nx, C, device="cuda")          # channels-last gradient
    gi = torch.empty(B, P, C, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    libs = [load(p) for p in args.libs]
    ws_elems = libs[0].mmt_voxel_pooling_backward_workspace_elems(B, P, C, nx, ny)
    ws = torch.empty(ws_elems, device="cuda")
    sb, sy, sx, sc = ny * nx * C, nx * C, C, 1

    def fwd(h):
        rc = h.mmt_voxel_pooling_forward_ex(B, P, C, nx, ny, nz, geom.data_ptr(), feats.data_ptr(), out.data_ptr(),
                                            pos.data_ptr(), args.flags, st)
        assert rc == 0, h.mmt_last_error()

    def bwd(h):
        rc = h.mmt_voxel_pooling_backward(B, P, C, nx, ny, pos.data_ptr(), go.data_ptr(), sb, sc, sy, sx,
                                          gi.data_ptr(), ws.data_ptr(), ws_elems, st)
        assert rc == 0, h.mmt_last_error()

    # the lift (producer of the feature matrix, lss_fpn.py:423-463) and its backward, same builds
    N, D = sh["N"], int((sh["d_bound"][1] - sh["d_bound"][0]) / sh["d_bound"][2])
    HW = P // (N * D)
    depth = torch.rand(B * N, D, HW, device="cuda")
    context = torch.randn(B * N, C, HW, device="cuda")
    gd, gc = torch.empty_like(depth), torch.empty_like(context)

    def lift(h):
        rc = h.mmt_lift_features(B * N, D, HW, C, depth.data_ptr(), context.data_ptr(), gi.data_ptr(), st)
        assert rc == 0, h.mmt_last_error()

    def lift_bwd(h):
        rc = h.mmt_lift_features_backward(B * N, D, HW, C, depth.data_ptr(), context.data_ptr(), gi.data_ptr(),
                                          gd.data_ptr(), gc.data_ptr(), st)
        assert rc == 0, h.mmt_last_error()

    lift_res, lift_checks = [{"lift": [], "lift_bwd": []} for _ in libs], []
    for h in libs:
        lift(h); lift_bwd(h); torch.cuda.synchronize()
        lift_checks.append((gd.clone(), gc.clone()))
    for rnd in range(args.rounds + 2):
        for k, h in enumerate(libs):
            evs = []
            for it in range(6):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                e[0].record(); lift(h); e[1].record(); e[2].record(); lift_bwd(h); e[3].record()
                evs.append(e)
            torch.cuda.synchronize()
            if rnd >= 2:
                lift_res[k]["lift"].append(sorted(e[0].elapsed_time(e[1]) for e in evs)[3])
                lift_res[k]["lift_bwd"].append(sorted(e[2].elapsed_time(e[3]) for e in evs)[3])

    results, checks = [{"fwd": [], "bwd": []} for _ in libs], []
    for h in libs:
        out.zero_(); fwd(h); bwd(h); torch.cuda.synchronize()
        checks.append((out.clone(), pos.clone(), gi.clone()))
    for rnd in range(args.rounds + 2):
        for k, h in enumerate(libs):
            evs = []
            for it in range(6):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                out.zero_()
                e[0].record(); fwd(h); e[1].record(); e[2].record(); bwd(h); e[3].record()
                evs.append(e)
            torch.cuda.synchronize()
            if rnd >= 2:
                results[k]["fwd"].append(sorted(e[0].elapsed_time(e[1]) for e in evs)[3])
                results[k]["bwd"].append(sorted(e[2].elapsed_time(e[3]) for e in evs)[3])
    med = lambda v: sorted(v)[len(v) // 2]
    if args.env_ab:
        name, vals = args.env_ab.split("=")
        vals = vals.split(",")
        flush = torch.empty(256 * 1024 * 1024, device="cuda") if args.flush else None
        per = {m: {"fwd": [], "bwd": []} for m in vals}
        h, ref_gi = libs[1], checks[1][2]
        for rnd in range(args.rounds + 2):
            for m in vals:
                os.environ[name] = m
                evs = []
                for it in range(6):
                    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                    out.zero_()
                    e[0].record(); fwd(h); e[1].record()
                    if flush is not None:
                        flush.sum()
                    e[2].record(); bwd(h); e[3].record()
                    evs.append(e)
                torch.cuda.synchronize()
                assert torch.equal(gi, ref_gi), (name, m)
                if rnd >= 2:
                    per[m]["fwd"].append(sorted(e[0].elapsed_time(e[1]) for e in evs)[3])
                    per[m]["bwd"].append(sorted(e[2].elapsed_time(e[3]) for e in evs)[3])
        os.environ.pop(name, None)
        print(json.dumps({"shape": args.shape, "env": name, "flush": bool(args.flush),
                          "values": {m: {"fwd_us": round(med(r["fwd"]) * 1e3, 2), "bwd_us": round(med(r["bwd"]) * 1e3, 2)} for m, r in per.items()}}))
        return
    print(json.dumps({
        "shape": args.shape, "libs": args.libs,
        "fwd_us": [round(med(r["fwd"]) * 1e3, 2) for r in results],
        "bwd_us": [round(med(r["bwd"]) * 1e3, 2) for r in results],
        "lift_us": [round(med(r["lift"]) * 1e3, 2) for r in lift_res],
        "lift_bwd_us": [round(med(r["lift_bwd"]) * 1e3, 2) for r in lift_res],
        "lift_bwd_max_rel_diff": [float(((lift_checks[0][i] - lift_checks[1][i]).abs().max() / lift_checks[0][i].abs().max())) for i in (0, 1)],
        "pos_memo_equal": bool(torch.equal(checks[0][1], checks[1][1])),
        "grad_in_equal": bool(torch.equal(checks[0][2], checks[1][2])),
        "bev_max_abs_diff": float((checks[0][0] - checks[1][0]).abs().max())}))


if __name__ == "__main__":
    main()
