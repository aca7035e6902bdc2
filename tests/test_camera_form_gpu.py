"""GPU: the CAMERA FORM of the fused lift-splat (mmt_lss_splat_forward_cam / _backward_cam; SURVEY section 8 rows f1 + f3).
The kernels compute every frustum point's voxel index themselves (lss_fpn.py:328-361 + :461-462 folded into :441-464), so
the contract is: cells BIT-IDENTICAL to mmt_frustum_geometry (itself bit-exact against the oracle and pinned to the
reference's get_geometry / quantise on the golden calibrations), results equal to the geom form of the same kernels."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pos_from_geom(geom, vn):
    """pos_memo of voxel_pooling (..._cuda.cu:19-29) from int32 geom [B, P, 3]: (b, y, x) for kept points, -1 otherwise."""
    nx, ny, nz = vn
    g = geom.astype(np.int64)
    kept = (g[..., 0] >= 0) & (g[..., 0] < nx) & (g[..., 1] >= 0) & (g[..., 1] < ny) & (g[..., 2] >= 0) & (g[..., 2] < nz)
    pos = np.full(geom.shape, -1, np.int32)
    b = np.broadcast_to(np.arange(geom.shape[0]).reshape(-1, 1), kept.shape)
    pos[..., 0] = np.where(kept, b, -1)
    pos[..., 1] = np.where(kept, g[..., 1], -1)
    pos[..., 2] = np.where(kept, g[..., 0], -1)
    return pos, kept


def _frustum(final_dim, ds, d_bound):
    from tests.test_oracle_golden import _frustum_torch
    return _frustum_torch(final_dim, ds, d_bound)


def _run_forward_cam(combine, fr, vc, vs, vn, C=64, seed=0, pixel_major=True, bf16=False, summary=None, cached=False, excl=None):
    """-> (out [B,ny,nx,C] numpy, pos_memo [B, N*D*fH*fW, 3] numpy in the kernel's point order, depth, ctx (CPU tensors))"""
    from mm_training_amd import _lib
    from mm_training_amd.ops.bev_geometry import frustum_axes
    B, N = combine.shape[:2]
    D, fH, fW, _ = fr.shape
    axes = frustum_axes(fr)
    assert axes is not None
    fu, fv, fd = [a.cuda() for a in axes]
    g = torch.Generator().manual_seed(seed)
    depth = torch.rand(B * N, D, fH, fW, generator=g).softmax(1)
    ctx = torch.randn(B * N, C, fH, fW, generator=g)
    sd = torch.bfloat16 if bf16 else torch.float32
    dep_dev = (depth.permute(0, 2, 3, 1) if pixel_major else depth).to(sd).contiguous().cuda()
    ctx_dev = ctx.permute(0, 2, 3, 1).to(sd).contiguous().cuda()
    nx, ny, nz = vn
    out = torch.full((B, ny, nx, C), float("nan"), device="cuda")          # MMT_LSS_ZERO_OUTPUT must overwrite it
    pos = torch.full((B, N * D * fH * fW, 3), -7, dtype=torch.int32, device="cuda")
    flags = (_lib.LSS_PIXEL_MAJOR if pixel_major else 0) | _lib.LSS_ZERO_OUTPUT | _lib.VP_WRITE_DROPPED
    cb = combine.contiguous().cuda()
    _lib.call("mmt_lss_splat_forward_cam" + ("_bf16" if bf16 else ""), B, N, D, fH, fW, C, nx, ny, nz, cb.data_ptr(), fu.data_ptr(),
              fv.data_ptr(), fd.data_ptr(), _lib.float3(vc), _lib.float3(vs), dep_dev.data_ptr(), ctx_dev.data_ptr(), out.data_ptr(),
              pos.data_ptr(), summary.data_ptr() if summary is not None else 0, excl.data_ptr() if excl is not None else 0,
              excl.numel() * 4 if excl is not None else 0, flags | (_lib.LSS_SUMMARY_CACHED if cached else 0),
              torch.cuda.current_stream().cuda_stream)
    fam = _lib.lib().mmt_lss_last_kernel_family(0)
    assert fam & 0x1F == 0x11                                        # ray walk, camera form
    reg = fH <= 16 and C <= 80 and D < 160
    assert bool(fam & _lib.LSS_FAMILY_REGISTER) == reg and bool(fam & _lib.LSS_FAMILY_BLOCK) == (not reg)      # register walk on short columns, block walk otherwise
    return out.cpu().numpy(), pos.cpu().numpy(), depth, ctx


def _check_cells(combine, fr, vc, vs, vn, oracle_mod, C=64, check_map=True):
    """the camera form's cells == mmt_frustum_geometry's (bit for bit), in both point orders; BEV map vs the oracle."""
    from mm_training_amd.ops.bev_geometry import frustum_geometry
    B, N = combine.shape[:2]
    D, fH, fW, _ = fr.shape
    geom = frustum_geometry(fr.cuda(), combine.cuda(), vc, vs).cpu().numpy()              # [B,N,D,fH,fW,3]
    from mm_training_amd.ops.bev_geometry import new_column_summary
    summary = new_column_summary(B, N, D, fH, fW, "cuda").fill_(-12345)
    # no summary / summary written on the way / summary read instead of computing: the same cells every time
    for pm, sm, cached in ((True, None, False), (False, None, False), (True, summary, False), (True, summary, True), (False, summary, True)):
        out, pos, depth, ctx = _run_forward_cam(combine, fr, vc, vs, vn, C=C, pixel_major=pm, summary=sm, cached=cached)
        gk = geom.transpose(0, 1, 3, 4, 2, 5) if pm else geom
        ref_pos, kept = _pos_from_geom(np.ascontiguousarray(gk).reshape(B, -1, 3), vn)
        assert np.array_equal(pos, ref_pos)
        assert not np.isnan(out).any()
        if check_map and pm:
            feats = oracle_mod.lift(depth.numpy(), ctx.numpy()).reshape(B, -1, C)
            ref = oracle_mod.voxel_pooling_forward_f64(geom.reshape(B, -1, 3), feats, *vn)
            assert np.abs(out - ref).max() <= 1e-4
    return geom


def test_camera_form_cells_on_the_golden_rig(mmt_lib, oracle_mod, golden):
    """the analytic 6-camera rig of quant_geom.npz (the matrices the reference's get_geometry was run on)"""
    g = golden["quant_geom"]
    fr = torch.from_numpy(g["nusc_frustum"])
    cb = torch.from_numpy(g["rig_combine"])
    geom = _check_cells(cb, fr, g["nusc_voxel_coord"], g["nusc_voxel_size"], [128, 128, 1], oracle_mod, C=80)
    # and therefore with the oracle's own geometry + quantise
    ref_xyz = oracle_mod.geometry(g["nusc_frustum"], g["rig_combine"])
    assert np.array_equal(geom, oracle_mod.quantize(ref_xyz, g["nusc_voxel_coord"], g["nusc_voxel_size"]))


def test_camera_form_cells_on_the_reference_nuscenes_calibration(mmt_lib, oracle_mod, golden):
    """the real 6-camera calibration of the reference's fixture (test/data/nuscenes/infos.pkl), 900x1600: cameras that are
    not level (3.8 % of the kept points leave their column's cell)"""
    g = golden["quant_geom"]
    fr = _frustum((900, 1600), 16, (2.0, 58.0, 0.5))
    cb = torch.from_numpy(g["nusc_fixture_combine"])
    _check_cells(cb, fr, g["nusc_voxel_coord"], g["nusc_voxel_size"], [128, 128, 1], oracle_mod, C=64, check_map=False)


@pytest.mark.parametrize("case", ["exact_boundaries", "aim_grid", "pitched", "nonfinite", "unsorted_rows", "negative_voxel_size",
                                  "rolled_17_rows"])
def test_camera_form_cells_where_the_fast_quantise_must_fall_back(mmt_lib, oracle_mod, case):
    """mmt_quantize_fast (csrc/mmt_camera.h) decides most points with one multiplication and hands the rest to the exact
    division: quotients that ARE integers (every point on a cell boundary), huge / non-finite values, the aiMotive grid."""
    from mm_training_amd import synthetic
    if case == "exact_boundaries":
        # identity camera: xyz = (u*d, v*d, d) with u, v integers and d multiples of 0.5 -> on a 0.5 m grid every quotient
        # is an integer, on a 0.25 m grid too
        fr = _frustum((64, 96), 16, (1.0, 9.0, 0.5))
        cb = torch.eye(4).repeat(1, 2, 1, 1).contiguous()
        cb[0, 1, 0, 3] = -40.0                                            # second camera shifted: negative quotients too
        vc, vs, vn = [0.25 - 8.0, 0.25 - 8.0, 0.25], [0.5, 0.5, 0.5], [128, 128, 40]
    elif case == "aim_grid":
        s2e, K = synthetic.camera_rig(2, 2, 1280, 704, jitter=0.02, seed=1)
        fr = _frustum((704, 1280), 16, (1.0, 205.5, 0.5))                 # D = 409, 44 x 80 (exps/conf_aim.py:16-18,42-52)
        cb = s2e.matmul(torch.inverse(K))
        vc, vs, vn = [-204.8 + 0.4, -25.6 + 0.4, -5.0 + 4.0], [0.8, 0.8, 8.0], [512, 64, 1]
    elif case == "pitched":
        s2e, K = synthetic.camera_rig(2, 3, 320, 256, jitter=0.02, seed=2)
        c_, s_ = math.cos(math.radians(4.0)), math.sin(math.radians(4.0))
        rx = torch.tensor([[1, 0, 0, 0], [0, c_, -s_, 0], [0, s_, c_, 0], [0, 0, 0, 1]], dtype=torch.float32)
        fr = _frustum((256, 320), 16, (2.0, 58.0, 0.5))
        cb = s2e.matmul(rx).matmul(torch.inverse(K))
        vc, vs, vn = [-51.2 + 0.4, -51.2 + 0.4, -1.0], [0.8, 0.8, 8.0], [128, 128, 1]
    elif case in ("unsorted_rows", "negative_voxel_size", "rolled_17_rows"):
        # the end-row shortcut of mmt_cam_column_cells needs rows sorted by v and a monotone quantise: a frustum whose rows
        # are shuffled, a negative voxel size (no exact range thresholds) and a strongly rolled camera over two row blocks
        # must all take the row-by-row evaluation and still agree with mmt_frustum_geometry
        H = 272 if case == "rolled_17_rows" else 256
        s2e, K = synthetic.camera_rig(1, 3, 320, H, jitter=0.02, seed=4)
        fr = _frustum((H, 320), 16, (2.0, 58.0, 0.5))
        if case == "unsorted_rows":
            fr = fr[:, torch.tensor([3, 0, 15, 7, 1, 9, 2, 14, 4, 13, 5, 12, 6, 11, 8, 10])].contiguous()
        ang = math.radians(25.0 if case == "rolled_17_rows" else 1.0)
        rz = torch.tensor([[math.cos(ang), -math.sin(ang), 0, 0], [math.sin(ang), math.cos(ang), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32)
        cb = s2e.matmul(rz).matmul(torch.inverse(K))
        vs = [-0.8, 0.8, 8.0] if case == "negative_voxel_size" else [0.8, 0.8, 8.0]
        vc, vn = [(51.2 if vs[0] < 0 else -51.2) + vs[0] / 2, -51.2 + 0.4, -1.0], [128, 128, 1]
    else:
        fr = _frustum((32, 48), 16, (2.0, 10.0, 1.0))
        cb = torch.eye(4).repeat(1, 4, 1, 1).contiguous()
        cb[0, 0, 0, 0] = float("nan")
        cb[0, 1, 1, 3] = float("inf")
        cb[0, 2, 0, 2] = 3e37                                              # overflows to +-inf / saturates the conversion
        cb[0, 3, 2, 2] = -1e30
        vc, vs, vn = [0.4, 0.4, 0.4], [0.8, 0.8, 0.8], [128, 128, 16]
    _check_cells(cb, fr, vc, vs, vn, oracle_mod, C=64, check_map=case not in ("aim_grid", "nonfinite"))


@pytest.mark.parametrize("cfg", [(2, 3, 37, 16, 9, 80, 0.0), (1, 2, 112, 32, 10, 128, 0.0), (1, 2, 40, 20, 5, 64, 2.0),
                                 (4, 6, 112, 16, 44, 80, 0.0), (1, 1, 1, 1, 1, 64, 0.0), (3, 3, 17, 17, 2, 64, 5.0),
                                 (1, 2, 409, 44, 6, 80, 0.0),           # the aiMotive-native frustum (D = 409, fH = 44) at a reduced width
                                 (2, 6, 112, 32, 88, 80, 0.0)])         # BASELINE configs[4]'s camera shape in full (block walk, two row blocks)
@pytest.mark.parametrize("bf16", [False, True])
def test_camera_form_equals_geom_form(mmt_lib, cfg, bf16):
    """lift_splat_camera == lift_splat(frustum_geometry(...)): forward to fp32 summation order (atomics), both backward
    kernels BIT-identical (they are the same deterministic kernels fed the same cells); the column kernel's counters =
    the mismatch measure of the geometry."""
    from mm_training_amd import synthetic
    from mm_training_amd.ops.bev_geometry import (column_mismatch_fraction, frustum_axes, frustum_geometry, last_kernel_family,
                                                  lift_splat, lift_splat_camera)
    B, N, D, fH, fW, C, pitch = cfg
    H, W = fH * 16, fW * 16
    s2e, K = synthetic.camera_rig(B, N, W, H, jitter=0.02, seed=0)
    c_, s_ = math.cos(math.radians(pitch)), math.sin(math.radians(pitch))
    rx = torch.tensor([[1, 0, 0, 0], [0, c_, -s_, 0], [0, s_, c_, 0], [0, 0, 0, 1]], dtype=torch.float32)
    combine = s2e.matmul(rx).matmul(torch.inverse(K)).contiguous().cuda()
    fr = _frustum((H, W), 16, (2.0, 2.0 + 0.5 * D, 0.5))
    assert tuple(fr.shape) == (D, fH, fW, 4)
    axes = tuple(a.cuda() for a in frustum_axes(fr))
    vc, vs, vn = [-51.2 + 0.4, -51.2 + 0.4, -5.0 + 4.0], [0.8, 0.8, 8.0], [128, 128, 1]
    geom_pm = frustum_geometry(fr.permute(1, 2, 0, 3).contiguous().cuda(), combine, vc, vs)
    g = torch.Generator().manual_seed(7)
    sd = torch.bfloat16 if bf16 else torch.float32
    depth = torch.rand(B * N, D, fH, fW, generator=g).softmax(1).to(sd)
    ctx = torch.randn(B * N, C, fH, fW, generator=g).to(sd)
    go = torch.randn(B, C, vn[1], vn[0], generator=g).cuda().contiguous(memory_format=torch.channels_last)
    for column in (False, True):
        d1 = depth.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        c1 = ctx.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        d2 = depth.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        c2 = ctx.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        from mm_training_amd._lib import LSS_STATS_SLOTS
        stats = torch.zeros(2 * LSS_STATS_SLOTS, dtype=torch.int64, device="cuda")
        out_cam = lift_splat_camera(combine, axes, d1, c1, vn, vc, vs, column_backward=column, column_stats=stats)
        assert last_kernel_family() == "ray+camera"
        out_geo = lift_splat(geom_pm, d2, c2, vn, pixel_major=True, column_backward=column)
        assert last_kernel_family() == "ray"
        scale = max(1.0, out_geo.abs().max().item())
        assert (out_cam - out_geo).abs().max().item() <= 2e-5 * scale
        out_cam.backward(go)
        fam = last_kernel_family(backward=True)
        out_geo.backward(go)
        col_fits = D <= 128                                          # the column kernel's mismatch lists need D <= ~200: D = 409 walks
        assert fam == ("column+camera" if column and col_fits else "ray+camera") and last_kernel_family(backward=True) == fam.split("+")[0]
        assert torch.equal(d1.grad, d2.grad)
        assert torch.equal(c1.grad, c2.grad)
        if column and D <= 128:
            gq = geom_pm.long()
            kept = ((gq[..., 0] >= 0) & (gq[..., 0] < vn[0]) & (gq[..., 1] >= 0) & (gq[..., 1] < vn[1])
                    & (gq[..., 2] >= 0) & (gq[..., 2] < vn[2])).sum().item()
            frac = float(column_mismatch_fraction(geom_pm, vn, pixel_major=True))
            # a 1-in-8 pseudo-random sample of the workgroups (= (camera, column, 16-row block) units) reports
            # (the kernel's workgroup -> (camera, column, row block) map: cameras cut into `split` column segments so that the
            # (camera, segment) units are a multiple of 8 and every XCD carries the same number of workgroups)
            rb = (fH + 15) // 16
            split = 8 // math.gcd(B * N, 8)
            if split > fW:
                split = 1
            seg = (fW + split - 1) // split
            unit = torch.arange(8 * ((B * N * split + 7) // 8) * seg * rb, dtype=torch.int64)
            sampled = ((unit * 0x9E3779B1) & 0xFFFFFFFF) >> 29 == 0
            xcd, rest = unit & 7, unit >> 3
            cs_u = (rest // (seg * rb)) * 8 + xcd                    # (camera, segment)
            bn_u, col_u, rb_u = cs_u // split, (cs_u % split) * seg + (rest % (seg * rb)) // rb, rest % rb
            sampled = sampled & (cs_u < B * N * split) & (col_u < fW)
            kmask = ((gq[..., 0] >= 0) & (gq[..., 0] < vn[0]) & (gq[..., 1] >= 0) & (gq[..., 1] < vn[1])
                     & (gq[..., 2] >= 0) & (gq[..., 2] < vn[2])).view(B * N, fH, fW, D).cpu()
            big = torch.where(kmask, (gq[..., 1] * vn[0] + gq[..., 0]).view(B * N, fH, fW, D).cpu(), torch.full((1,), 1 << 40, dtype=torch.int64))
            want_kept = want_mis = 0
            for u_ in torch.nonzero(sampled & (bn_u < B * N)).flatten().tolist():
                blk_k = kmask[bn_u[u_], rb_u[u_] * 16:(rb_u[u_] + 1) * 16, col_u[u_]]
                blk_c = big[bn_u[u_], rb_u[u_] * 16:(rb_u[u_] + 1) * 16, col_u[u_]]
                want_kept += int(blk_k.sum())
                want_mis += int((blk_k & (blk_c != blk_c.amin(0, keepdim=True))).sum())
            assert int(stats[1::2].sum()) == want_kept and int(stats[0::2].sum()) == want_mis
            if kept > 20000:
                assert abs(int(stats[0::2].sum()) / max(int(stats[1::2].sum()), 1) - frac) < 0.01
            if pitch == 0.0 and fH <= 16:
                assert int(stats[0::2].sum()) == 0
        else:
            assert int(stats.sum()) == 0                                   # the ray walk does not report


def test_camera_form_argument_checks(mmt_lib):
    from mm_training_amd import _lib
    from mm_training_amd.ops.bev_geometry import camera_form_supported
    assert camera_form_supported(4, 6, 112, 16, 44, 80) and camera_form_supported(2, 6, 112, 32, 88, 80)
    assert camera_form_supported(4, 2, 409, 44, 80, 80)                   # the reference's native aiMotive shape
    assert not camera_form_supported(1, 1, 16, 4, 4, 48) and not camera_form_supported(1, 1, 16, 600, 4, 64)
    t = torch.zeros(4096, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    vc, vs = _lib.float3([0.4] * 3), _lib.float3([0.8] * 3)
    lib = _lib.lib()
    args = [t.data_ptr()] * 4 + [vc, vs] + [t.data_ptr()] * 3 + [0, 0, 0, 0]
    assert lib.mmt_lss_splat_forward_cam(1, 1, 4, 2, 2, 64, 8, 8, 1, *args, _lib.LSS_TILE_KERNELS, st) == -4
    assert lib.mmt_lss_splat_forward_cam(1, 1, 4, 2, 2, 48, 8, 8, 1, *args, 0, st) == -2
    assert lib.mmt_lss_splat_forward_cam(1, 1, 4, 2, 2, 64, 8, 8, 1, 0, *args[1:], 0, st) == -1
    assert lib.mmt_lss_splat_forward_cam(1, 1, 4, 2, 2, 64, 8, 8, 1, *args, 0x4000, st) == -4
    assert lib.mmt_lss_splat_forward_cam(1, 1, 4, 2, 2, 64, 8, 8, 1, *args, _lib.LSS_SUMMARY_CACHED, st) == -1      # cached, but no summary
    xargs = args[:11] + [t.data_ptr() + 4, 4096]                        # a misaligned / an undersized exclusive-cell cache
    assert lib.mmt_lss_splat_forward_cam(1, 1, 4, 2, 2, 64, 8, 8, 1, *xargs, 0, st) == -2
    xargs = args[:11] + [t.data_ptr(), 64 * 4]
    assert lib.mmt_lss_splat_forward_cam(1, 1, 4, 2, 2, 64, 8, 8, 1, *xargs, 0, st) == -2
    assert lib.mmt_lss_exclusive_cache_bytes(6, 128, 128, 4) == 4 * (64 + 8 * 136 + 4 * (4 + 96 + 128 * 128))
    assert lib.mmt_lss_exclusive_cache_bytes(0, 128, 128, 4) == 0 and lib.mmt_lss_exclusive_cache_bytes(6, 128, 128, 0) == 0
    bargs = [t.data_ptr()] * 4 + [vc, vs] + [t.data_ptr()] * 3 + [64 * 64, 1, 8 * 64, 64] + [t.data_ptr()] * 2 + [0, 0]
    assert lib.mmt_lss_splat_backward_cam(1, 1, 4, 2, 2, 64, 8, 8, 1, *bargs, _lib.LSS_TILE_KERNELS, st) == -4
    assert lib.mmt_lss_splat_backward_cam(1, 1, 4, 2, 2, 64, 8, 8, 1, *bargs[:9], 64 * 64, 2, 8 * 64, 64, *bargs[13:], 0, st) == -2
    assert lib.mmt_lss_splat_backward_cam(1, 1, 4, 2, 2, 64, 40000, 8, 1, *bargs[:15], t.data_ptr(), 0, 0, st) == -2     # summary: grid >= 32768


def _tiny_lssfpn(pitch_deg=0.0, seed=0):
    from mm_training_amd import synthetic
    from mm_training_amd.dp import make_config
    from mm_training_amd.layers.backbones import LSSFPN
    cfg = make_config("tiny")
    bc = dict(cfg["backbone_conf"], output_channels=64)                  # a width the ray / column kernels take
    torch.manual_seed(seed)
    m = LSSFPN(**bc).cuda().train()
    H, W = cfg["final_dim"]
    B, N = 2, cfg["num_cams"]
    s2e, K = synthetic.camera_rig(B, N, W, H, jitter=0.02, seed=0)
    c_, s_ = math.cos(math.radians(pitch_deg)), math.sin(math.radians(pitch_deg))
    rx = torch.tensor([[1, 0, 0, 0], [0, c_, -s_, 0], [0, s_, c_, 0], [0, 0, 0, 1]], dtype=torch.float32)
    mats = dict(sensor2ego_mats=s2e.matmul(rx).view(B, 1, N, 4, 4).cuda(), intrin_mats=K.view(B, 1, N, 4, 4).cuda(),
                bda_mat=torch.eye(4).repeat(B, 1, 1).cuda())
    imgs = torch.rand(B, 1, N, 3, H, W, generator=torch.Generator().manual_seed(1)).cuda()
    return m, imgs, mats


def test_lssfpn_camera_form_is_the_default_and_runs_no_geometry_kernel(mmt_lib, monkeypatch):
    """LSSFPN's fused branch: camera form by default -- same map and gradients as the geom form, and neither
    mmt_frustum_geometry nor a geom tensor in the step; with a calibration_id the camera matrices are cached too."""
    from mm_training_amd import _lib
    m, imgs, mats = _tiny_lssfpn()
    assert m.fused_lift_splat and m.camera_form and m._has_frustum_axes
    m.lift_splat_backward = "ray"
    calls = []
    real = _lib.call
    monkeypatch.setattr(_lib, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])

    def run(camera_form, mats_dict):
        m.camera_form = camera_form
        m.zero_grad(set_to_none=True)
        torch.manual_seed(3)
        calls.clear()
        bev = m(imgs, mats_dict)
        bev.square().mean().backward()
        return bev.detach().clone(), m.depth_net.context_conv.weight.grad.detach().clone(), list(calls)

    assert m.plan_form
    bev_p, g_p, calls_p = run(True, mats)                                  # the default: camera form, its forward in the plan form
    assert "mmt_lss_splat_forward_plan" in calls_p and "mmt_lss_splat_backward_cam" in calls_p and "mmt_frustum_geometry" not in calls_p
    # the lookup rides in the depth softmax's launch -- or, for rows that cannot take 16-byte pieces (this frustum's D), goes right in front of it
    rider, own = "mmt_depth_softmax_forward_plan_prepare" in calls_p, "mmt_lss_plan_prepare" in calls_p
    assert rider != own and (rider or calls_p.index("mmt_lss_plan_prepare") + 1 == calls_p.index("mmt_depth_softmax_forward"))
    assert (calls_p.index("mmt_depth_softmax_forward_plan_prepare") if rider else calls_p.index("mmt_lss_plan_prepare")) < calls_p.index("mmt_lss_splat_forward_plan")
    m.plan_form = False
    bev_c, g_c, calls_c = run(True, mats)
    # (two passes through the MIOpen backbone in front of the pooling: its split-K forward kernels leave the last bits open, and with the
    # stem frozen -- eval-mode norm1 on a randomly initialised conv1 -- the activations behind it are larger than batch-normalised ones)
    assert (bev_p - bev_c).abs().max().item() <= 5e-5 * max(1.0, bev_c.abs().max().item())
    assert (g_p - g_c).abs().max().item() <= 1e-3 * g_c.abs().max().item() + 1e-7
    bev_g, g_g, calls_g = run(False, mats)
    assert "mmt_lss_splat_forward_cam" in calls_c and "mmt_lss_splat_backward_cam" in calls_c
    assert "mmt_frustum_geometry" not in calls_c and "mmt_frustum_geometry" in calls_g
    assert (bev_c - bev_g).abs().max().item() <= 5e-5 * max(1.0, bev_g.abs().max().item())
    assert (g_c - g_g).abs().max().item() <= 1e-3 * g_g.abs().max().item() + 1e-7     # (two passes through MIOpen's split-K nets: against the tensor's own size)
    # calibration id: the matrices are computed once
    cached = dict(mats, calibration_id="rig-a")
    run(True, cached)
    assert len(m._combine_cache) == 1
    first = next(iter(m._combine_cache.values()))
    bev2, _, calls2 = run(True, cached)
    assert next(iter(m._combine_cache.values())) is first and "mmt_frustum_geometry" not in calls2
    assert len(m._summary_cache) == 1                                      # and so is the geometry's column summary
    assert torch.allclose(bev2, bev_c, rtol=0, atol=5e-5 * max(1.0, bev_c.abs().max().item()))


def test_lssfpn_follows_the_column_kernel_counters_without_a_calibration_id(mmt_lib):
    """"auto" without mats_dict['calibration_id']: the column kernel's own counters, read back lazily, move a pitched rig
    to the ray walk within a few steps and keep a level rig on the column kernel; no host synchronisation is needed for it."""
    from mm_training_amd.ops.bev_geometry import last_kernel_family
    for pitch, want in ((0.0, "column+camera"), (4.0, "ray+camera")):
        m, imgs, mats = _tiny_lssfpn(pitch)
        m.column_probe_period = 1000
        fams = []
        for _ in range(12):
            m.zero_grad(set_to_none=True)
            m(imgs, mats).square().mean().backward()
            fams.append(last_kernel_family(backward=True))
            torch.cuda.synchronize()                                       # (only so that the lazy copy has surely landed)
        assert fams[0] == "column+camera"                                  # the first steps probe with the column kernel
        assert fams[-1] == want, fams
        st = m._column_adaptive
        assert st["share"] is not None and ((st["share"] == 0.0) if pitch == 0.0 else (st["share"] > 0.01))
    # probing again after column_probe_period steps on the ray walk
    m.column_probe_period = 3
    m._column_adaptive["ray_left"] = 1
    m.zero_grad(set_to_none=True)
    m(imgs, mats).square().mean().backward()
    assert last_kernel_family(backward=True) == "column+camera"
