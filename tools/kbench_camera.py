#!/usr/bin/env python3
"""Camera form against geom form of the fused lift-splat, through the C ABI, interleaved in one process (run on the GPU box):
   python tools/kbench_camera.py [--shape cfg4|cfg5|aim] [--dtype f32|bf16] [--pitch DEG] [lib.so ...]
Times (dispatch-attached events = kernel durations): mmt_frustum_geometry, the forward ray walk, the column and the ray
backward in both forms -- warm, cold (1 GiB written between launches) and, for the forward, behind a caller-side zero-fill
against MMT_LSS_ZERO_OUTPUT (fill inside the timed sequence) -- after checking that both forms give the same cells
(pos_memo), map and gradients."""
import argparse, ctypes, json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mm_training_amd import _lib, synthetic

PM, COL, ZERO, WD, CACHED = 0x100, 0x400, 0x800, 0x10, 0x1000
SHAPES = {   # B, N, D, fH, fW, C, image, d_bound, x/y/z bounds
    "cfg4": (4, 6, 112, 16, 44, 80, (256, 704), (2.0, 58.0, 0.5), ((-51.2, 51.2, 0.8), (-51.2, 51.2, 0.8), (-5.0, 3.0, 8.0))),
    "cfg5": (2, 6, 112, 32, 88, 80, (512, 1408), (2.0, 58.0, 0.5), ((-51.2, 51.2, 0.8), (-51.2, 51.2, 0.8), (-5.0, 3.0, 8.0))),
    # the reference's native configuration (exps/conf_aim.py:1-3,16-18,42-52): 2 cameras, D = 409, 44 x 80, 512 x 64 grid
    "aim": (4, 2, 409, 44, 80, 80, (704, 1280), (1.0, 205.5, 0.5), ((-204.8, 204.8, 0.8), (-25.6, 25.6, 0.8), (-5.0, 3.0, 8.0))),
}


def load(path):
    h = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="cfg4")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--pitch", type=float, default=0.0)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--stamps", choices=["fwd", "col"], default=None, help="the library is a -DLSS_STAMPS build of lift_splat_tile.hip (fwd) "
                    "or lift_splat_col.hip (col) (tools/build_variant.py): print that kernel's in-kernel s_memtime phase breakdown instead of "
                    "timing.  The choice must match the build: a stamp build writes stamps where a plain one writes its results")
    ap.add_argument("--cases", default=None, help="regular expression: time only the cases whose name matches")
    ap.add_argument("libs", nargs="*")
    args = ap.parse_args()
    libs = args.libs or [_lib.LIB_PATH]
    _lib.lib()
    hs = [load(p) for p in libs]
    B, N, D, fH, fW, C, (H, W), d_bound, bounds = SHAPES[args.shape]
    bf16 = args.dtype == "bf16"
    sfx = "_bf16" if bf16 else ""
    sd = torch.bfloat16 if bf16 else torch.float32
    s2e, K = synthetic.camera_rig(B, N, W, H, jitter=0.02, seed=0)
    c_, s_ = math.cos(math.radians(args.pitch)), math.sin(math.radians(args.pitch))
    rx = torch.tensor([[1, 0, 0, 0], [0, c_, -s_, 0], [0, s_, c_, 0], [0, 0, 0, 1]], dtype=torch.float32)
    combine = s2e.matmul(rx).matmul(torch.inverse(K)).contiguous().cuda()
    d = torch.arange(*d_bound, dtype=torch.float)
    assert d.numel() == D, (d.numel(), D)
    fu = torch.linspace(0, W - 1, fW, dtype=torch.float).cuda()
    fv = torch.linspace(0, H - 1, fH, dtype=torch.float).cuda()
    fd = d.cuda()
    fr_pm = torch.stack((fu.view(1, fW, 1).expand(fH, fW, D), fv.view(fH, 1, 1).expand(fH, fW, D), fd.view(1, 1, D).expand(fH, fW, D),
                         torch.ones(fH, fW, D, device="cuda")), -1).contiguous()
    vs = [b[2] for b in bounds]
    vc = [b[0] + b[2] / 2.0 for b in bounds]
    nx, ny, nz = [int((b[1] - b[0]) / b[2]) for b in bounds]
    vc_c, vs_c = _lib.float3(vc), _lib.float3(vs)
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(0)
    depth = torch.rand(B * N, fH, fW, D, generator=g).softmax(-1).to(sd).cuda()        # pixel-major
    ctx = torch.randn(B * N, fH, fW, C, generator=g).to(sd).cuda()
    go = torch.randn(B, ny, nx, C, generator=g).cuda()
    out = torch.zeros(B, ny, nx, C, device="cuda")
    geom = torch.empty(B, N, fH, fW, D, 3, dtype=torch.int32, device="cuda")
    gd, gc = torch.empty_like(depth), torch.empty(ctx.shape, device="cuda")
    stats = torch.zeros(2 * _lib.LSS_STATS_SLOTS, dtype=torch.int64, device="cuda")
    P = N * D * fH * fW
    h0 = hs[0]

    def geometry(h):
        return h.mmt_frustum_geometry(B * N, fH * fW * D, fr_pm.data_ptr(), combine.data_ptr(), vc_c, vs_c, geom.data_ptr(), None, st)

    summary = torch.empty(B * N, (fH + 15) // 16, fW, D, 2, dtype=torch.int32, device="cuda")

    def fwd(h, cam, flags=PM, pos=None, sm=None, xc=None):
        if cam:
            return getattr(h, "mmt_lss_splat_forward_cam" + sfx)(B, N, D, fH, fW, C, nx, ny, nz, combine.data_ptr(), fu.data_ptr(), fv.data_ptr(),
                                                                fd.data_ptr(), vc_c, vs_c, depth.data_ptr(), ctx.data_ptr(), out.data_ptr(),
                                                                pos.data_ptr() if pos is not None else None,
                                                                sm.data_ptr() if sm is not None else None,
                                                                xc.data_ptr() if xc is not None else None, xc.numel() * 4 if xc is not None else 0, flags, st)
        return getattr(h, "mmt_lss_splat_forward" + sfx)(B, N, D, fH, fW, C, nx, ny, nz, geom.data_ptr(), depth.data_ptr(), ctx.data_ptr(),
                                                        out.data_ptr(), pos.data_ptr() if pos is not None else None, flags, st)

    def bwd(h, cam, flags, sm=None):
        if cam:
            return getattr(h, "mmt_lss_splat_backward_cam" + sfx)(B, N, D, fH, fW, C, nx, ny, nz, combine.data_ptr(), fu.data_ptr(), fv.data_ptr(),
                                                                 fd.data_ptr(), vc_c, vs_c, depth.data_ptr(), ctx.data_ptr(), go.data_ptr(),
                                                                 ny * nx * C, 1, nx * C, C, gd.data_ptr(), gc.data_ptr(),
                                                                 sm.data_ptr() if sm is not None else None, stats.data_ptr(), flags, st)
        return getattr(h, "mmt_lss_splat_backward" + sfx)(B, N, D, fH, fW, C, nx, ny, nz, geom.data_ptr(), depth.data_ptr(), ctx.data_ptr(),
                                                         go.data_ptr(), ny * nx * C, 1, nx * C, C, gd.data_ptr(), gc.data_ptr(), flags, st)

    if args.stamps:      # grad_context receives 4 s_memtime stamps per workgroup (100 MHz ticks) instead of its rows
        assert geometry(h0) == 0, h0.mmt_last_error()
        out.zero_(); assert fwd(h0, True, PM, None, summary) == 0, h0.mmt_last_error()
        flush = torch.zeros(256 * 1024 * 1024, device="cuda")
        # forward (lift_splat_tile.hip built with -DLSS_STAMPS): pos_memo receives 8 stamps per workgroup
        nwg_f = 8 * ((B * N + 7) // 8) * fW
        for label, cam, flags, sm in ((("geom form", False, PM, None), ("camera form", True, PM, None), ("camera form, summary cached", True, PM | CACHED, summary))
                                      if args.stamps == "fwd" else ()):
            for cold in (False, True):
                stamp_buf = torch.zeros(nwg_f * 8 * 2 + 16, dtype=torch.int64, device="cuda")
                for _ in range(3):
                    if cold:
                        flush.add_(1.0)
                    out.zero_()
                    stamp_buf.zero_()
                    assert fwd(h0, cam, flags, stamp_buf.view(torch.int32), sm) == 0, h0.mmt_last_error()
                torch.cuda.synchronize()
                s8 = stamp_buf[:nwg_f * 8].view(-1, 8).cpu()
                xcd = torch.arange(len(s8)) & 7                 # blockIdx & 7: the XCDs' s_memtime counters are not synchronised
                ok = (s8[:, 0] != 0) & (s8[:, 1] != 0) & (s8[:, 2] != 0)
                if int(ok.sum()) == 0:
                    print("forward", label, ": no stamps (not a -DLSS_STAMPS build of lift_splat_tile.hip)")
                    continue
                walk_end = s8[:, 2:6].max(1).values
                spans, starts = [], []
                for x in range(8):
                    m_ = ok & (xcd == x)
                    if int(m_.sum()):
                        t0 = s8[m_, 0].min()
                        spans.append(float(walk_end[m_].max() - t0))
                        starts.append(float((s8[m_, 0] - t0).float().mean()))
                print("forward %s %s: %d workgroups; mean start +%.0f | phase 1 (records + context -> LDS) %.0f | walk %.0f | first start to last walk end, per XCD: %.0f cycles"
                      % (label, "cold" if cold else "warm", int(ok.sum()), sum(starts) / len(starts), (s8[ok, 1] - s8[ok, 0]).float().mean(),
                         (walk_end[ok] - s8[ok, 1]).float().mean(), sum(spans) / len(spans)))
        for name, flags in ((("column", PM | COL),) if args.stamps == "col" else ()):      # (lift_splat_col.hip built with -DLSS_STAMPS)
            for cam in (0, 1, 2):
                for cold in (False, True):
                    for _ in range(3):
                        if cold:
                            flush.add_(1.0)
                        gc.zero_()
                        assert bwd(h0, cam == 1, flags, summary if cam == 2 else None) == 0, h0.mmt_last_error()
                    torch.cuda.synchronize()
                    nwg = 8 * ((B * N + 7) // 8) * fW * ((fH + 15) // 16) if name == "column" else 4096
                    s64 = gc.view(-1)[:16 * min(nwg, gc.numel() // 16)].view(torch.int64).view(-1, 8).cpu()      # (8 stamps per workgroup: tools/scratch/col_stamps8.py shows them all)
                    s64 = s64[(s64[:, 0] != 0) & (s64[:, 1] != 0) & (s64[:, 2] != 0)]
                    dd = (s64[:, 1:3] - s64[:, 0:2]).float()
                    span = (s64[:, 2].max() - s64[:, 0].min()).item()
                    print("%s backward %s %s: %d workgroups, phase A %.0f ticks | products / walk %.0f ticks ; first start to last end %.0f ticks (10 ns each)"
                          % (name, ("geom form", "camera form", "camera form + summary")[cam], "cold" if cold else "warm", len(s64), *dd.mean(0).tolist(), span))
        return
    # ---- agreement first
    assert geometry(h0) == 0, h0.mmt_last_error()
    pos_g = torch.full((B, P, 3), -7, dtype=torch.int32, device="cuda")
    pos_c = torch.full((B, P, 3), -9, dtype=torch.int32, device="cuda")
    out.zero_(); assert fwd(h0, False, PM | WD, pos_g) == 0, h0.mmt_last_error()
    out_g = out.clone()
    out.fill_(float("nan")); assert fwd(h0, True, PM | WD | ZERO, pos_c, summary) == 0, h0.mmt_last_error()
    pos_s = torch.full((B, P, 3), -11, dtype=torch.int32, device="cuda")
    out_c = out.clone()
    out.fill_(float("nan")); assert fwd(h0, True, PM | WD | ZERO | CACHED, pos_s, summary) == 0, h0.mmt_last_error()
    assert torch.equal(pos_s, pos_c) and float((out - out_c).abs().max()) <= 2e-5 * float(out_c.abs().max())
    kept = (pos_g[..., 0] >= 0).float().mean().item()
    info = dict(shape=args.shape, dtype=args.dtype, pitch=args.pitch, dims=[B, N, D, fH, fW, C], grid=[nx, ny, nz], kept_fraction=round(kept, 4),
                cells_identical=bool(torch.equal(pos_g, pos_c)), fwd_max_abs_diff=float((out - out_g).abs().max()), fwd_scale=float(out_g.abs().max()),
                family_fwd=hex(h0.mmt_lss_last_kernel_family(0)))
    for name, flags in (("ray", PM), ("column", PM | COL)):
        gd.fill_(1.5); gc.fill_(float("nan")); assert bwd(h0, False, flags) == 0, h0.mmt_last_error()
        fam_g = h0.mmt_lss_last_kernel_family(1)
        a = (gd.clone(), gc.clone())
        gd.fill_(2.5); gc.fill_(float("nan")); assert bwd(h0, True, flags) == 0, h0.mmt_last_error()
        info["bwd_%s_identical" % name] = bool(torch.equal(a[0], gd) and torch.equal(a[1], gc))
        gd.fill_(3.5); gc.fill_(float("nan")); assert bwd(h0, True, flags, summary) == 0, h0.mmt_last_error()
        info["bwd_%s_identical" % name] = info["bwd_%s_identical" % name] and bool(torch.equal(a[0], gd) and torch.equal(a[1], gc))
        info["family_bwd_%s" % name] = [hex(fam_g), hex(h0.mmt_lss_last_kernel_family(1))]
    torch.cuda.synchronize()
    info["column_stats"] = [int(stats[0::2].sum()), int(stats[1::2].sum())]
    print(json.dumps(info))
    assert info["cells_identical"] and info["bwd_ray_identical"] and info["bwd_column_identical"]

    flush = torch.zeros(256 * 1024 * 1024, device="cuda")

    def timed(h, fn, reps=15, warm=3, cold=False, zero_first=False):
        evs = []
        for i in range(warm + reps):
            if cold:
                flush.add_(1.0)
            if zero_first:
                out.zero_()
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
        return round(ts[len(ts) // 2] * 1e3, 1)

    # exclusive-cell cache (include/mmt_hip.h): four calls teach it this rig, then single-run cells are stored, not added
    XS = 256
    xbytes = h0.mmt_lss_exclusive_cache_bytes(N, nx, ny, XS)
    xcache = {p: torch.zeros(xbytes // 4, dtype=torch.int32, device="cuda") for p in libs}
    for p, h in zip(libs, hs):
        for _ in range(4):
            assert fwd(h, True, PM | ZERO, None, None, xcache[p]) == 0, h.mmt_last_error()
    torch.cuda.synchronize()
    xc0 = xcache[libs[0]]
    st0 = xc0[64 + 8 * 136 + XS * (4 + N * 16):].view(XS, -1)[int(xc0[8])]    # the states of sample 0's slot
    info["exclusive_cache"] = dict(modes=xc0[24:24 + B].tolist(), family=hex(h0.mmt_lss_last_kernel_family(0)),
                                   cells_hit=int((st0 != 0).sum()), cells_single_run=int((st0 > 0).sum()))
    # (diagnostic builds -- tools/build_variant.py -DLSS_EXP_NOFLUSH / _NOWALK -- never teach a cache: they time the first
    # library's, which stays in its steady state)
    by_handle = {id(h): xcache[libs[0] if os.environ.get("KBC_SHARE_CACHE") else p] for p, h in zip(libs, hs)}
    cases = {
        "frustum_geometry": (lambda h: geometry(h), {}),
        "fwd_cam_exclusive_cache(fill+kernel)": (lambda h: fwd(h, True, PM | ZERO, None, None, by_handle[id(h)]), {}),
        "fwd_cam_exclusive_cache_summary_cached(fill+kernel)": (lambda h: fwd(h, True, PM | ZERO | CACHED, None, summary, by_handle[id(h)]), {}),
        "fwd_cam_exclusive_cache_cold(fill+kernel)": (lambda h: fwd(h, True, PM | ZERO, None, None, by_handle[id(h)]), dict(cold=True)),
        "fwd_geom": (lambda h: fwd(h, False), {}),
        "fwd_cam": (lambda h: fwd(h, True), {}),
        "fwd_cam_writes_summary": (lambda h: fwd(h, True, PM, None, summary), {}),
        "fwd_cam_summary_cached": (lambda h: fwd(h, True, PM | CACHED, None, summary), {}),
        "fwd_cam_summary_cached_cold(fill+kernel)": (lambda h: fwd(h, True, PM | CACHED | ZERO, None, summary), dict(cold=True)),
        "fwd_geom_after_caller_zero_fill": (lambda h: fwd(h, False), dict(zero_first=True)),
        "fwd_cam_after_caller_zero_fill": (lambda h: fwd(h, True), dict(zero_first=True)),
        "fwd_geom_zero_output(fill+kernel)": (lambda h: fwd(h, False, PM | ZERO), {}),
        "fwd_cam_zero_output(fill+kernel)": (lambda h: fwd(h, True, PM | ZERO), {}),
        "fwd_geom_cold": (lambda h: fwd(h, False, PM | ZERO), dict(cold=True)),
        "fwd_cam_cold": (lambda h: fwd(h, True, PM | ZERO), dict(cold=True)),
        "col_bwd_geom": (lambda h: bwd(h, False, PM | COL), {}),
        "col_bwd_cam": (lambda h: bwd(h, True, PM | COL), {}),
        "col_bwd_cam_summary": (lambda h: bwd(h, True, PM | COL, summary), {}),
        "col_bwd_cam_summary_cold": (lambda h: bwd(h, True, PM | COL, summary), dict(cold=True)),
        "col_bwd_geom_cold": (lambda h: bwd(h, False, PM | COL), dict(cold=True)),
        "col_bwd_cam_cold": (lambda h: bwd(h, True, PM | COL), dict(cold=True)),
        "ray_bwd_geom": (lambda h: bwd(h, False, PM), {}),
        "ray_bwd_cam": (lambda h: bwd(h, True, PM), {}),
        "ray_bwd_cam_summary": (lambda h: bwd(h, True, PM, summary), {}),
        "ray_bwd_cam_summary_cold": (lambda h: bwd(h, True, PM, summary), dict(cold=True)),
        "ray_bwd_geom_cold": (lambda h: bwd(h, False, PM), dict(cold=True)),
        "ray_bwd_cam_cold": (lambda h: bwd(h, True, PM), dict(cold=True)),
    }
    # ---- plan form (lift_splat_plan.hip): the output-stationary forward on the learnt plan
    if hasattr(h0, "mmt_lss_splat_forward_plan"):
        PREP = 0x2000
        pbytes_of = {id(h): h.mmt_lss_plan_cache_bytes(N, D, fH, fW, nx, ny, max(B, 2)) for h in hs}      # (a variant build may size its slots differently)
        pcache = {p: torch.zeros(pbytes_of[id(h)] + 256, dtype=torch.uint8, device="cuda") for p, h in zip(libs, hs)}
        pptr = {id(h): (pcache[p].data_ptr() + 255) & ~255 for p, h in zip(libs, hs)}

        def fwd_plan(h, flags=PM, sm=None):
            return getattr(h, "mmt_lss_splat_forward_plan" + sfx)(B, N, D, fH, fW, C, nx, ny, nz, combine.data_ptr(), fu.data_ptr(), fv.data_ptr(),
                                                                 fd.data_ptr(), vc_c, vs_c, depth.data_ptr(), ctx.data_ptr(), out.data_ptr(),
                                                                 sm.data_ptr() if sm is not None else None, pptr[id(h)], pbytes_of[id(h)], flags, st)

        def prepare(h):
            return h.mmt_lss_plan_prepare(B, N, D, fH, fW, nx, ny, nz, combine.data_ptr(), fu.data_ptr(), fv.data_ptr(), fd.data_ptr(), vc_c, vs_c,
                                          pptr[id(h)], pbytes_of[id(h)], st)

        out.fill_(float("nan")); assert fwd_plan(h0) == 0, h0.mmt_last_error()
        out_p = out.clone()
        out.fill_(float("nan")); assert fwd_plan(h0, PM | PREP) == 0, h0.mmt_last_error()
        cnt = (ctypes.c_int64 * 8)()
        h0.mmt_lss_plan_cache_counters(pptr[id(h0)], pbytes_of[id(h0)], cnt, st)
        info["plan_form"] = dict(max_abs_diff_vs_camera=float((out_p - out_c).abs().max()), bit_identical_again=bool(torch.equal(out, out_p)),
                                 every_element_written=not bool(torch.isnan(out_p).any()), counters=list(cnt)[:6], family=hex(h0.mmt_lss_last_kernel_family(0)))
        assert info["plan_form"]["bit_identical_again"] and info["plan_form"]["every_element_written"]
        assert os.environ.get("KBC_ABLATION") or info["plan_form"]["max_abs_diff_vs_camera"] <= 2e-5 * float(out_c.abs().max())
        for p, h in zip(libs[1:], hs[1:]):
            assert fwd_plan(h) == 0, h.mmt_last_error()
        cases.update({
            "plan_prepare(probe+build, all known)": (lambda h: prepare(h), {}),
            "fwd_plan(prepare+kernel)": (lambda h: fwd_plan(h), {}),
            "fwd_plan_prepared(kernel)": (lambda h: fwd_plan(h, PM | PREP), {}),
            "fwd_plan_prepared_writes_summary(kernel)": (lambda h: fwd_plan(h, PM | PREP, summary), {}),
            "fwd_plan_prepared_cold(kernel)": (lambda h: fwd_plan(h, PM | PREP), dict(cold=True)),
            "fwd_plan_brute_force": (lambda h: fwd_plan(h, PM | PREP | 0x4000), dict(reps=3, warm=1)),
        })
    if args.cases:
        import re
        cases = {k: v for k, v in cases.items() if re.search(args.cases, k)}
    res = {}
    for rnd in range(args.rounds):
        for name, (fn, kw) in cases.items():
            for p, h in zip(libs, hs):
                out.zero_()
                res.setdefault(name, {}).setdefault(os.path.basename(p), []).append(timed(h, fn, **kw))
    print(json.dumps(dict(info, us=res), indent=1))


if __name__ == "__main__":
    main()
