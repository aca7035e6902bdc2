#!/usr/bin/env python3
"""Fused lift-splat micro-benchmark through the C ABI (run on the GPU box), interleaved over several builds of the library:
   python tools/kbench_fused.py [lib.so ...]      (default: the regular build)
Times mmt_lss_splat_forward / _backward -- ray walks (default kernels) and frustum tiles (MMT_LSS_TILE_KERNELS) -- in the
pixel-major and the frustum point order at the cfg2 / cfg4 camera shape, kernel-side (dispatch-attached events), for the level
analytic rig and for the same rig with the image rolled by 5 degrees (what the reference's image augmentation does to a
column of pixels); checks that both kernel families agree before timing."""
import ctypes, json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mm_training_amd import _lib, synthetic

PM, TILE, COL = 0x100, 0x200, 0x400


def load(path):
    h = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
    return h


def rolled_rig_geometry(B, roll_deg, N=6, final_dim=(256, 704), ds=16, d_bound=(2.0, 58.0, 0.5)):
    """rig geometry with the pixel grid rotated about the image centre (an image-rotation augmentation undone by ida^-1)"""
    s2e, K = synthetic.camera_rig(B, N, final_dim[1], final_dim[0], jitter=0.02, seed=0)
    H, W = final_dim
    fH, fW = H // ds, W // ds
    d = torch.arange(*d_bound, dtype=torch.float).view(-1, 1, 1).expand(-1, fH, fW)
    D = d.shape[0]
    xs = torch.linspace(0, W - 1, fW).view(1, 1, fW).expand(D, fH, fW) - W / 2
    ys = torch.linspace(0, H - 1, fH).view(1, fH, 1).expand(D, fH, fW) - H / 2
    c, s = math.cos(math.radians(roll_deg)), math.sin(math.radians(roll_deg))
    xr, yr = c * xs - s * ys + W / 2, s * xs + c * ys + H / 2
    p = torch.stack((xr * d, yr * d, d, torch.ones_like(d)), -1)
    xyz = torch.einsum("bnij,dhwj->bndhwi", s2e.matmul(torch.inverse(K)), p)[..., :3].contiguous()
    geom, vn = synthetic.quantize_cpu(xyz, (-51.2, 51.2, 0.8), (-51.2, 51.2, 0.8), (-5.0, 3.0, 8.0))
    return geom.contiguous(), [int(v) for v in vn]


def pitched_rig_geometry(B, pitch_deg, N=6, final_dim=(256, 704), ds=16, d_bound=(2.0, 58.0, 0.5)):
    """rig geometry with every camera pitched about its own x axis (a mounting tolerance): the pixels of a column no longer
    share their BEV cell exactly"""
    s2e, K = synthetic.camera_rig(B, N, final_dim[1], final_dim[0], jitter=0.02, seed=0)
    c, s = math.cos(math.radians(pitch_deg)), math.sin(math.radians(pitch_deg))
    Rx = torch.tensor([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=torch.float32)
    xyz = synthetic.frustum_geometry_xyz(s2e.matmul(Rx), K, final_dim, ds, d_bound)
    geom, vn = synthetic.quantize_cpu(xyz, (-51.2, 51.2, 0.8), (-51.2, 51.2, 0.8), (-5.0, 3.0, 8.0))
    return geom.contiguous(), [int(v) for v in vn]


def main():
    libs = sys.argv[1:] or [_lib.LIB_PATH]
    _lib.lib()
    hs = [load(p) for p in libs]
    B, N, D, fH, fW, C = 4, 6, 112, 16, 44, 80
    g = torch.Generator().manual_seed(0)
    depth = torch.rand(B * N, D, fH, fW, generator=g).softmax(1).cuda()
    depth_pm = depth.permute(0, 2, 3, 1).contiguous()
    ctx = torch.randn(B * N, fH, fW, C, generator=g).cuda()
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    rig = synthetic.rig_geometry(B)
    variants = {"rig": rig, "roll5": rolled_rig_geometry(B, 5.0)}
    if os.environ.get("KBF_EXTRA"):      # every point dropped / every point kept (same cells, z forced into range): what do dropped bins cost?
        variants["pitch1"] = pitched_rig_geometry(B, 1.0)
        variants["alldrop"] = (torch.full_like(rig[0], -1), rig[1])
        allk = rig[0].clone()
        allk[..., 0].clamp_(0, rig[1][0] - 1); allk[..., 1].clamp_(0, rig[1][1] - 1); allk[..., 2] = 0
        variants["allkept"] = (allk, rig[1])
    for gname, (geom, vn) in variants.items():
        nx, ny, nz = vn
        geom = geom.cuda()
        geom_pm = geom.permute(0, 1, 3, 4, 2, 5).contiguous()
        kept = ((geom[..., 0] >= 0) & (geom[..., 0] < nx) & (geom[..., 1] >= 0) & (geom[..., 1] < ny) & (geom[..., 2] >= 0) & (geom[..., 2] < nz))
        print(gname, "kept fraction %.3f" % kept.float().mean().item())
        out = torch.zeros(B, ny, nx, C, device="cuda")
        go = torch.randn(B, ny, nx, C, generator=g).cuda()
        gd, gd_pm, gc = torch.empty_like(depth), torch.empty_like(depth_pm), torch.empty_like(ctx)

        def fwd(h, flags):
            pm = flags & PM
            return h.mmt_lss_splat_forward(B, N, D, fH, fW, C, nx, ny, nz, (geom_pm if pm else geom).data_ptr(), (depth_pm if pm else depth).data_ptr(),
                                           ctx.data_ptr(), out.data_ptr(), None, flags & ~COL, st)

        def bwd(h, flags):
            pm = flags & PM
            return h.mmt_lss_splat_backward(B, N, D, fH, fW, C, nx, ny, nz, (geom_pm if pm else geom).data_ptr(), (depth_pm if pm else depth).data_ptr(),
                                            ctx.data_ptr(), go.data_ptr(), ny * nx * C, 1, nx * C, C, (gd_pm if pm else gd).data_ptr(), gc.data_ptr(), flags, st)

        def timed(h, fn, reps=15, warm=3, cold=False, zero_first=False):
            evs = []
            for i in range(warm + reps):
                if cold:        # 1 GiB of writes between launches: nothing of the operands is left in L2 / the Infinity Cache
                    flush.add_(1.0)
                if zero_first:  # as in the training step: the BEV map is zero-filled right before the forward's atomics
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
            return ts[len(ts) // 2] * 1e3

        # agreement of the two kernel families (and of both point orders)
        h = hs[0]
        ref = {}
        for name, flags in (("tile_pm", PM | TILE), ("ray_pm", PM), ("ray_frustum", 0), ("col_pm", PM | COL), ("col_frustum", COL)):
            out.zero_()
            assert fwd(h, flags) == 0, h.mmt_last_error()
            gc.fill_(float("nan")); gd.fill_(float("nan")); gd_pm.fill_(float("nan"))
            assert bwd(h, flags) == 0, h.mmt_last_error()
            cur = dict(out=out.clone(), gc=gc.clone(), gd=(gd_pm.permute(0, 3, 1, 2) if flags & PM else gd).clone())
            if not ref:
                ref = cur
            else:
                print(gname, name, "vs tile: max abs diff", {k: float((cur[k] - ref[k]).abs().max()) for k in cur},
                      "scale", {k: float(ref[k].abs().max()) for k in cur})
        for p_, hh in zip(libs, hs):
          for cold_ in (False, True):
            if "STAMPS" in p_ and gname == "rig":      # diagnostic build: grad_context receives 4 s_memtime stamps per workgroup (ray backward)
                gc.zero_()
                if cold_:
                    torch.zeros(256 * 1024 * 1024, device="cuda").add_(1.0)
                bwd(hh, PM | COL if os.environ.get("KBF_STAMP_COL") else PM)
                torch.cuda.synchronize()
                s64 = gc.view(-1)[:8 * 2048].view(torch.int64).view(-1, 4).cpu()
                s64 = s64[(s64[:, 0] != 0) & (s64[:, 2] != 0) & (s64[:, 1] != 0)]
                d = (s64[:, 1:3] - s64[:, 0:2]).float()
                span = (s64[:, 2].max() - s64[:, 0].min()).item()
                print("RAY BWD STAMPS %s (s_memtime ticks, %d workgroups): phase A %.0f | walk %.0f ; first start to last end %.0f" % ("cold" if cold_ else "warm", len(s64), *d.mean(0).tolist(), span))
        flush = torch.zeros(256 * 1024 * 1024, device="cuda")
        cases = {"ray_fwd_pm": lambda h: fwd(h, PM), "tile_fwd_pm": lambda h: fwd(h, PM | TILE), "ray_fwd_frustum": lambda h: fwd(h, 0),
                 "ray_bwd_pm": lambda h: bwd(h, PM), "tile_bwd_pm": lambda h: bwd(h, PM | TILE), "ray_bwd_frustum": lambda h: bwd(h, 0),
                 "col_bwd_pm": lambda h: bwd(h, PM | COL), "col_bwd_frustum": lambda h: bwd(h, COL)}
        for rnd in range(3):
            for name, fn in cases.items():
                for p, h in zip(libs, hs):
                    out.zero_()
                    res.setdefault(gname + ":" + name, {}).setdefault(os.path.basename(p), []).append(round(timed(h, fn), 1))
                    if name == "ray_fwd_pm" and gname == "rig":
                        res.setdefault(gname + ":" + name + ":after_zero_fill", {}).setdefault(os.path.basename(p), []).append(round(timed(h, fn, zero_first=True), 1))
                    if name in ("ray_fwd_pm", "ray_bwd_pm", "tile_fwd_pm", "tile_bwd_pm", "col_bwd_pm") and gname == "rig":
                        out.zero_()
                        res.setdefault(gname + ":" + name + ":cold", {}).setdefault(os.path.basename(p), []).append(round(timed(h, fn, cold=True), 1))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
