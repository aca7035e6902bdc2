#!/usr/bin/env python3
"""Fused lift-splat micro-benchmark through the C ABI (run on the GPU box), interleaved over several builds of the library:
   python tools/kbench_fused.py [lib.so ...]      (default: the regular build)
Times mmt_lss_splat_forward (frustum tiles) with and without the pos_memo output, mmt_lift_splat_forward (chunked, first
generation) and both backward kernels at the cfg2 / cfg4 camera shape, kernel-side (dispatch-attached events)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mm_training_amd import _lib, synthetic


def load(path):
    h = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
    return h


def main():
    libs = sys.argv[1:] or [_lib.LIB_PATH]
    _lib.lib()
    hs = [load(p) for p in libs]
    B, N, D, fH, fW, C = 4, 6, 112, 16, 44, 80
    geom, vn = synthetic.rig_geometry(B)
    nx, ny, nz = vn
    geom = geom.cuda()
    g = torch.Generator().manual_seed(0)
    depth = torch.rand(B * N, D, fH, fW, generator=g).softmax(1).cuda()
    ctx = torch.randn(B * N, fH, fW, C, generator=g).cuda()
    out = torch.zeros(B, ny, nx, C, device="cuda")
    pos = torch.empty(B, N * D * fH * fW, 3, dtype=torch.int32, device="cuda")
    go = torch.randn(B, ny, nx, C, generator=g).cuda()
    gd, gc = torch.empty_like(depth), torch.empty_like(ctx)
    st = torch.cuda.current_stream().cuda_stream
    HW = fH * fW

    def timed(h, fn, reps=15, warm=3):
        evs = []
        for i in range(warm + reps):
            s, e = ctypes.c_void_p(), ctypes.c_void_p()
            h.mmt_timing_event_create(ctypes.byref(s)); h.mmt_timing_event_create(ctypes.byref(e))
            h.mmt_arm_kernel_timing(s, e)
            rc = fn(h)
            h.mmt_arm_kernel_timing(None, None)
            assert rc == 0, (rc, h.mmt_last_error())
            if i >= warm:
                evs.append((s, e))
        torch.cuda.synchronize()
        ts = []
        for s, e in evs:
            ms = ctypes.c_float()
            h.mmt_timing_elapsed_ms(s, e, ctypes.byref(ms))
            ts.append(ms.value)
        ts.sort()
        return ts[len(ts) // 2] * 1e3

    cases = {
        "tile_fwd_pos": lambda h: h.mmt_lss_splat_forward(B, N, D, fH, fW, C, nx, ny, nz, geom.data_ptr(), depth.data_ptr(), ctx.data_ptr(), out.data_ptr(), pos.data_ptr(), 0x10, st),
        "tile_fwd_nopos": lambda h: h.mmt_lss_splat_forward(B, N, D, fH, fW, C, nx, ny, nz, geom.data_ptr(), depth.data_ptr(), ctx.data_ptr(), out.data_ptr(), None, 0, st),
        "chunk_fwd": lambda h: h.mmt_lift_splat_forward(B, N, D, HW, C, nx, ny, nz, geom.data_ptr(), depth.data_ptr(), ctx.data_ptr(), out.data_ptr(), pos.data_ptr(), 0x10, st),
        "pixel_bwd": lambda h: h.mmt_lift_splat_backward(B, N, D, HW, C, nx, ny, pos.data_ptr(), depth.data_ptr(), ctx.data_ptr(), go.data_ptr(), ny * nx * C, 1, nx * C, C, gd.data_ptr(), gc.data_ptr(), st),
    }
    if all(hasattr(h, "mmt_lss_splat_backward") for h in hs):
        geom_pm = geom.permute(0, 1, 3, 4, 2, 5).contiguous()
        depth_pm = depth.permute(0, 2, 3, 1).contiguous()
        gd_pm = torch.empty_like(depth_pm)

        def tile_bwd(h, pm=0):
            gc.zero_()
            return h.mmt_lss_splat_backward(B, N, D, fH, fW, C, nx, ny, nz, (geom_pm if pm else geom).data_ptr(), (depth_pm if pm else depth).data_ptr(),
                                            ctx.data_ptr(), go.data_ptr(), ny * nx * C, 1, nx * C, C, (gd_pm if pm else gd).data_ptr(), gc.data_ptr(), pm, st)
        cases["tile_bwd"] = tile_bwd
        cases["tile_bwd_pixel_major"] = lambda h: tile_bwd(h, 0x100)
        cases["tile_fwd_pixel_major"] = lambda h: h.mmt_lss_splat_forward(B, N, D, fH, fW, C, nx, ny, nz, geom_pm.data_ptr(), depth_pm.data_ptr(), ctx.data_ptr(), out.data_ptr(), None, 0x100, st)
    for p_, h in zip(libs, hs):
        if "STAMPS" in p_:      # diagnostic build: pos_memo receives 8 s_memtime stamps per workgroup
            pos.zero_()
            h.mmt_lss_splat_forward(B, N, D, fH, fW, C, nx, ny, nz, geom.data_ptr(), depth.data_ptr(), ctx.data_ptr(), out.data_ptr(), pos.data_ptr(), 0x10, st)
            torch.cuda.synchronize()
            s64 = pos.view(-1)[:8 * 8192].view(torch.int64).view(-1, 8).cpu()
            s64 = s64[(s64[:, 0] != 0) & (s64[:, 5] != 0)]
            d = (s64[:, 1:6] - s64[:, 0:5]).float()
            print("STAMPS (s_memtime ticks, mean over %d workgroups): load+init %.0f | hash %.0f | count+scan %.0f | scatter %.0f | gather+flush %.0f | total %.0f"
                  % (len(s64), *d.mean(0).tolist(), d.sum(1).mean().item()))
    res = {}
    hs[0].mmt_lss_splat_forward(B, N, D, fH, fW, C, nx, ny, nz, geom.data_ptr(), depth.data_ptr(), ctx.data_ptr(), out.data_ptr(), pos.data_ptr(), 0x10, st)   # a valid pos_memo for the backward cases
    for rnd in range(3):
        for name, fn in cases.items():
            for p, h in zip(libs, hs):
                out.zero_()
                res.setdefault(name, {}).setdefault(os.path.basename(p), []).append(round(timed(h, fn), 1))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
