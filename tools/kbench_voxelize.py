#!/usr/bin/env python3
"""Voxelize + mean (mmt_hard_voxelize_mean: vox_cells / vox_own / vox_emit; builds before ABI 11: vox_link / vox_heads / vox_emit) at BASELINE shapes, per library given on the
command line (interleaved A/B of builds), dispatch-attached events = the three kernels' own time:
    python tools/kbench_voxelize.py [--points 40000 --batch 4 --cols 5] [lib.so ...]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mm_training_amd import _lib, synthetic
from tools.kbench_camera import load

RANGE, VSIZE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], [0.2, 0.2, 8.0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=40000)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--cols", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--native-range", action="store_true", help="the reference's own range [-204.8, -25.6, ..] = a 2048 x 256 grid (exps/conf_aim.py:16-18)")
    ap.add_argument("libs", nargs="*")
    args = ap.parse_args()
    _lib.lib()
    global RANGE
    if args.native_range:
        RANGE = [-204.8, -25.6, -5.0, 204.8, 25.6, 3.0]
    libs = args.libs or [_lib.LIB_PATH]
    hs = [load(p) for p in libs]
    B, F, V, T = args.batch, args.cols, 25000, 15
    pts = torch.cat([synthetic.lidar_frame(args.points, F, RANGE, num_radar=2000 if F == 8 else 0, seed=s) for s in range(B)], 0).cuda()
    offs = torch.tensor([i * args.points for i in range(B + 1)], dtype=torch.int32).cuda()
    grid = _lib.int3([int(round((RANGE[3 + k] - RANGE[k]) / VSIZE[k])) for k in range(3)])
    N = B * args.points
    out = {}
    res = {}
    for name, h in zip(libs, hs):
        if h.mmt_abi_version() >= 11:
            table = torch.zeros(int(h.mmt_voxelize_table_elems(B, grid, N)), dtype=torch.int32, device="cuda")
            scratch = torch.empty(int(h.mmt_voxelize_scratch_elems(B, grid, N, T)), dtype=torch.int32, device="cuda")
        else:                   # builds before the region-owner form: two-argument size functions
            import ctypes
            h.mmt_voxelize_table_elems.argtypes, h.mmt_voxelize_scratch_elems.argtypes = [ctypes.c_int, ctypes.c_void_p], [ctypes.c_int, ctypes.c_int64]
            table = torch.zeros(int(h.mmt_voxelize_table_elems(B, grid)), dtype=torch.int32, device="cuda")
            scratch = torch.empty(int(h.mmt_voxelize_scratch_elems(B, N)), dtype=torch.int32, device="cuda")
        coors = torch.empty((B * V, 4), dtype=torch.int32, device="cuda")
        nump = torch.empty((B * V,), dtype=torch.int32, device="cuda")
        cnt = torch.empty((B,), dtype=torch.int32, device="cuda")
        mean = torch.empty((B * V, 5), dtype=torch.float32, device="cuda")
        res[name] = (h, table, scratch, coors, nump, cnt, mean)
    st = torch.cuda.current_stream().cuda_stream

    def run(name):
        h, table, scratch, coors, nump, cnt, mean = res[name]
        rc = h.mmt_hard_voxelize_mean(B, N, F, pts.data_ptr(), offs.data_ptr(), _lib.float3(VSIZE), _lib.float3(RANGE[:3]), grid, T, V, 5, None,
                                      coors.data_ptr(), nump.data_ptr(), cnt.data_ptr(), mean.data_ptr(), table.data_ptr(), scratch.data_ptr(), st)
        assert rc == 0, h.mmt_last_error()

    for name in libs:
        for _ in range(3):
            run(name)
    torch.cuda.synchronize()
    ref = None
    for name in libs:               # the builds agree bit for bit
        run(name)
        torch.cuda.synchronize()
        cur = tuple(t.clone() for t in res[name][3:])
        if ref is not None:
            assert all(torch.equal(a, b) for a, b in zip(ref, cur)), "builds disagree"
        ref = cur
    times = {os.path.basename(n): [] for n in libs}
    for _ in range(args.rounds):
        for name in libs:
            h = res[name][0]
            ts = []
            for _ in range(20):
                s, e = _lib.KernelEvent(), _lib.KernelEvent()
                h.mmt_arm_kernel_timing(s.handle, e.handle)
                run(name)
                ts.append((s, e))
            torch.cuda.synchronize()
            v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts)
            times[os.path.basename(name)].append(round(v[len(v) // 2], 2))
    print(json.dumps(dict(points=args.points, batch=B, cols=F, voxels=int(ref[2].sum()), first_to_last_kernel_us=times)))


if __name__ == "__main__":
    main()
