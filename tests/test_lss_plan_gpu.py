"""GPU: the PLAN FORM of the fused lift-splat forward (mmt_lss_plan_prepare / mmt_lss_splat_forward_plan; SURVEY section 8 rows
f1 + f3; lss_fpn.py:328-361 + :461-462 + :441-464).  Contract: the BEV map of the oracle's lift -> voxel_pooling on the cells of
mmt_frustum_geometry (<= 1e-4 of the map's scale against the fp64 sum), every element written, bit-identical from call to
call; the plan the device learns == the plan the host build of the same integer core makes from the device's column summary;
the column summary it hands the backward gives the camera form's gradients bit for bit."""
import ctypes
import math

import numpy as np
import pytest
import torch

from tests import plan_emul as P

pytestmark = pytest.mark.gpu


def _frustum(final_dim, ds, d_bound):
    from tests.test_oracle_golden import _frustum_torch
    return _frustum_torch(final_dim, ds, d_bound)


def _layout(N, D, fH, fW, nx, ny, cache):
    from mm_training_amd import _lib
    from mm_training_amd.ops.bev_geometry import _plan_ptr
    out = (ctypes.c_int64 * 12)()
    ptr, nbytes = _plan_ptr(cache)
    _lib.call("mmt_lss_plan_cache_layout", N, D, fH, fW, nx, ny, nbytes, out)
    names = ["slots", "slots_off", "slot_bytes", "summary_off", "records_off", "verdict_off", "jobs_cap", "runs_cap", "job_bytes", "max_b", "strips", "ntiles"]
    lay = dict(zip(names, [int(v) for v in out]))
    lay["base"] = ptr - cache.data_ptr()
    return lay


def _verdicts(cache, lay, B):
    o = lay["base"] + lay["verdict_off"]
    return cache[o:o + 64 * B].view(torch.int32).view(B, 16).cpu().numpy()      # slot, units, state, representative, group starts [9]


def _slot_arrays(cache, lay, slot, njobs):
    o = lay["base"] + lay["slots_off"] + slot * lay["slot_bytes"]
    summary = cache[o + lay["summary_off"]:o + lay["summary_off"] + 8 * lay["strips"] * lay["D"]].view(torch.int32).cpu().numpy()
    recs = cache[o + lay["records_off"]:o + lay["records_off"] + njobs * P.JOB_BYTES].cpu().numpy().reshape(njobs, P.JOB_BYTES)
    return summary, recs


def _forward(combine, axes, vc, vs, vn, depth, ctx, cache, bf16=False, prepared=False, brute=False, summary=None):
    """raw entry point: depth [BN, fH, fW, D], ctx [BN, fH, fW, C] CUDA tensors in the storage dtype -> out [B, ny, nx, C] (NaN-prefilled)"""
    from mm_training_amd import _lib
    from mm_training_amd.ops.bev_geometry import _plan_ptr
    fu, fv, fd = axes
    B, N = combine.shape[:2]
    D, fH, fW, C = fd.numel(), fv.numel(), fu.numel(), ctx.shape[-1]
    nx, ny, nz = vn
    out = torch.full((B, ny, nx, C), float("nan"), device="cuda")
    ptr, nbytes = _plan_ptr(cache)
    flags = _lib.LSS_PIXEL_MAJOR | (_lib.LSS_PLAN_PREPARED if prepared else 0) | (_lib.LSS_PLAN_BRUTE if brute else 0)
    _lib.call("mmt_lss_splat_forward_plan" + ("_bf16" if bf16 else ""), B, N, D, fH, fW, C, nx, ny, nz, combine.data_ptr(), fu.data_ptr(), fv.data_ptr(),
              fd.data_ptr(), _lib.float3(vc), _lib.float3(vs), depth.data_ptr(), ctx.data_ptr(), out.data_ptr(),
              summary.data_ptr() if summary is not None else 0, ptr, nbytes, flags, torch.cuda.current_stream().cuda_stream)
    assert _lib.lib().mmt_lss_last_kernel_family(0) == 0x14
    return out


def _case(combine, fr, vc, vs, vn, oracle_mod, C=64, bf16=False, seed=0, slots=None, check_host_plan=True, expect_state=1):
    """plan form on one geometry: map vs the oracle, determinism, brute-force path, the learnt plan vs the host build"""
    from mm_training_amd.ops.bev_geometry import frustum_axes, frustum_geometry, new_plan_cache, plan_cache_counters
    B, N = combine.shape[:2]
    D, fH, fW, _ = fr.shape
    nx, ny, nz = vn
    axes = tuple(a.cuda() for a in frustum_axes(fr))
    cb = combine.contiguous().cuda()
    geom = frustum_geometry(fr.cuda(), cb, vc, vs).cpu().numpy()                          # [B,N,D,fH,fW,3]: the cells the contract names
    g = torch.Generator().manual_seed(seed)
    sd = torch.bfloat16 if bf16 else torch.float32
    depth = torch.rand(B * N, fH, fW, D, generator=g).softmax(-1).to(sd)
    ctx = torch.randn(B * N, fH, fW, C, generator=g).to(sd)
    cache = new_plan_cache(N, D, fH, fW, vn, "cuda", slots=slots or max(B, 2))
    out = _forward(cb, axes, vc, vs, vn, depth.cuda(), ctx.cuda(), cache, bf16=bf16)
    o = out.cpu().numpy()
    assert not np.isnan(o).any()                                                           # every element written, no fill needed
    feats = oracle_mod.lift(depth.float().permute(0, 3, 1, 2).contiguous().numpy(), ctx.float().permute(0, 3, 1, 2).contiguous().numpy()).reshape(B, -1, C)
    ref = oracle_mod.voxel_pooling_forward_f64(geom.reshape(B, -1, 3), feats, nx, ny, nz)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(o - ref).max() <= 1e-4 * scale
    # bit-identical from call to call (prepared or not), and the brute-force path agrees and is deterministic too
    out2 = _forward(cb, axes, vc, vs, vn, depth.cuda(), ctx.cuda(), cache, bf16=bf16, prepared=True)
    assert torch.equal(out, out2)
    ob = _forward(cb, axes, vc, vs, vn, depth.cuda(), ctx.cuda(), cache, bf16=bf16, brute=True)
    assert np.abs(ob.cpu().numpy() - ref).max() <= 1e-4 * scale
    assert torch.equal(ob, _forward(cb, axes, vc, vs, vn, depth.cuda(), ctx.cuda(), cache, bf16=bf16, brute=True))
    lay = _layout(N, D, fH, fW, nx, ny, cache)
    lay["D"] = D
    vd = _verdicts(cache, lay, B)
    assert (vd[:, 2] == expect_state).all(), vd
    cnt = plan_cache_counters(cache)
    distinct = len({combine[b].numpy().tobytes() for b in range(B)})
    assert cnt["learnt"] == distinct and cnt["calls"] == 3 and cnt["hit"] == 2 * distinct
    if check_host_plan and expect_state == 1:
        for b in range(B):
            slot, njobs = int(vd[b, 0]), int(vd[b, 1])
            summary, recs = _slot_arrays(cache, lay, slot, njobs)
            # the device's summary has the semantics of the reference geometry (flags may be cleared where the host sets them)
            s_ref, rowcells = P.summary_from_geom(geom[b], nx, ny, nz)
            sd_ = summary.reshape(s_ref.shape)
            uni = (sd_[..., 1] & P.UNIFORM) != 0
            assert ((sd_[..., 1] & 0xFFFF) == (s_ref[..., 1] & 0xFFFF)).all()
            assert (~uni | ((s_ref[..., 1] & P.UNIFORM) != 0)).all()                     # a set bit is a true statement
            assert (sd_[..., 0] == s_ref[..., 0]).all()
            # same integer core, same summary -> the same records byte for byte
            n_host, recs_host, _ = P.build(N, D, fH, fW, nx, ny, summary, rowcells)
            assert n_host == njobs and np.array_equal(recs_host, recs)
    return out, cache


def test_plan_form_on_the_golden_rig(mmt_lib, oracle_mod, golden):
    g = golden["quant_geom"]
    fr = torch.from_numpy(g["nusc_frustum"])
    cb = torch.from_numpy(g["rig_combine"])
    _case(cb, fr, g["nusc_voxel_coord"], g["nusc_voxel_size"], [128, 128, 1], oracle_mod, C=80)


def test_plan_form_on_the_reference_nuscenes_calibration(mmt_lib, oracle_mod, golden):
    """the reference fixture's real 6-camera calibration at 900 x 1600: cameras that are not level -> mixed blocks in the plan"""
    g = golden["quant_geom"]
    fr = _frustum((900, 1600), 16, (2.0, 58.0, 0.5))
    cb = torch.from_numpy(g["nusc_fixture_combine"])
    _case(cb, fr, g["nusc_voxel_coord"], g["nusc_voxel_size"], [128, 128, 1], oracle_mod, C=64)


@pytest.mark.parametrize("case", ["exact_boundaries", "aim_grid", "pitched", "nonfinite", "unsorted_rows", "negative_voxel_size", "rolled_17_rows"])
def test_plan_form_on_the_hard_geometries(mmt_lib, oracle_mod, case):
    """the geometries tests/test_camera_form_gpu.py feeds the camera form (cells on boundaries, the aiMotive grid, pitch, roll over
    two row blocks, unsorted rows, a negative voxel size, non-finite matrices)"""
    from mm_training_amd import synthetic
    state = 1
    if case == "exact_boundaries":
        fr = _frustum((64, 96), 16, (1.0, 9.0, 0.5))
        cb = torch.eye(4).repeat(1, 2, 1, 1).contiguous()
        cb[0, 1, 0, 3] = -40.0
        vc, vs, vn = [0.25 - 8.0, 0.25 - 8.0, 0.25], [0.5, 0.5, 0.5], [128, 128, 40]
    elif case == "aim_grid":
        s2e, K = synthetic.camera_rig(2, 2, 1280, 704, jitter=0.02, seed=1)
        fr = _frustum((704, 1280), 16, (1.0, 205.5, 0.5))
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
        if case == "rolled_17_rows":
            state = None                      # a 25 degree roll: whichever of plan / brute force the capacity allows, the map must be right
    else:
        fr = _frustum((64, 48), 16, (2.0, 10.0, 1.0))
        cb = torch.eye(4).repeat(1, 4, 1, 1).contiguous()
        cb[0, 0, 0, 0] = float("nan")
        cb[0, 1, 1, 3] = float("inf")
        cb[0, 2, 0, 2] = 3e37
        cb[0, 3, 2, 2] = -1e30
        vc, vs, vn = [0.4, 0.4, 0.4], [0.8, 0.8, 0.8], [128, 128, 16]
    if state is None:
        from mm_training_amd.ops.bev_geometry import frustum_axes, frustum_geometry, new_plan_cache
        B, N = cb.shape[:2]
        D, fH, fW, _ = fr.shape
        axes = tuple(a.cuda() for a in frustum_axes(fr))
        geom = frustum_geometry(fr.cuda(), cb.cuda(), vc, vs).cpu().numpy()
        g = torch.Generator().manual_seed(0)
        depth = torch.rand(B * N, fH, fW, D, generator=g).softmax(-1)
        ctx = torch.randn(B * N, fH, fW, 64, generator=g)
        cache = new_plan_cache(N, D, fH, fW, vn, "cuda", slots=2)
        out = _forward(cb.cuda().contiguous(), axes, vc, vs, vn, depth.cuda(), ctx.cuda(), cache).cpu().numpy()
        feats = oracle_mod.lift(depth.permute(0, 3, 1, 2).contiguous().numpy(), ctx.permute(0, 3, 1, 2).contiguous().numpy()).reshape(B, -1, 64)
        ref = oracle_mod.voxel_pooling_forward_f64(geom.reshape(B, -1, 3), feats, *vn)
        assert np.abs(out - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max()))
        return
    # non-finite matrices: the oracle's map holds NaN where the kernel's does (NaN coordinates land in cell 0): compare where finite
    if case == "nonfinite":
        from mm_training_amd.ops.bev_geometry import frustum_axes, frustum_geometry, new_plan_cache
        B, N = cb.shape[:2]
        D, fH, fW, _ = fr.shape
        axes = tuple(a.cuda() for a in frustum_axes(fr))
        geom = frustum_geometry(fr.cuda(), cb.cuda(), vc, vs).cpu().numpy()
        g = torch.Generator().manual_seed(0)
        depth = torch.rand(B * N, fH, fW, D, generator=g).softmax(-1)
        ctx = torch.randn(B * N, fH, fW, 64, generator=g)
        cache = new_plan_cache(N, D, fH, fW, vn, "cuda", slots=2)
        out = _forward(cb.cuda().contiguous(), axes, vc, vs, vn, depth.cuda(), ctx.cuda(), cache).cpu().numpy()
        feats = oracle_mod.lift(depth.permute(0, 3, 1, 2).contiguous().numpy(), ctx.permute(0, 3, 1, 2).contiguous().numpy()).reshape(B, -1, 64)
        ref = oracle_mod.voxel_pooling_forward_f64(geom.reshape(B, -1, 3), feats, *vn)
        assert np.abs(out - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max()))
        return
    _case(cb, fr, vc, vs, vn, oracle_mod, C=64)


@pytest.mark.parametrize("cfg", [(2, 3, 37, 16, 9, 80, 0.0), (1, 2, 112, 32, 10, 128, 0.0), (1, 2, 40, 20, 5, 64, 2.0),
                                 (4, 6, 112, 16, 44, 80, 0.0), (3, 3, 17, 17, 2, 64, 5.0), (1, 1, 4, 1, 1, 64, 0.0),
                                 (2, 6, 112, 32, 88, 80, 0.0)])         # incl. BASELINE configs[3] / configs[4]'s camera shapes in full
@pytest.mark.parametrize("bf16", [False, True])
def test_plan_form_equals_camera_form_with_gradients(mmt_lib, cfg, bf16):
    """lift_splat_plan == lift_splat_camera: the map to fp32 summation order, both backward kernels BIT-identical (the plan
    form hands them the column summary of its slots)."""
    from mm_training_amd import synthetic
    from mm_training_amd.ops.bev_geometry import frustum_axes, last_kernel_family, lift_splat_camera, lift_splat_plan, new_plan_cache, plan_prepare
    B, N, D, fH, fW, C, pitch = cfg
    H, W = fH * 16, fW * 16
    s2e, K = synthetic.camera_rig(B, N, W, H, jitter=0.02, seed=0)
    c_, s_ = math.cos(math.radians(pitch)), math.sin(math.radians(pitch))
    rx = torch.tensor([[1, 0, 0, 0], [0, c_, -s_, 0], [0, s_, c_, 0], [0, 0, 0, 1]], dtype=torch.float32)
    combine = s2e.matmul(rx).matmul(torch.inverse(K)).contiguous().cuda()
    fr = _frustum((H, W), 16, (2.0, 2.0 + 0.5 * D, 0.5))
    axes = tuple(a.cuda() for a in frustum_axes(fr))
    vc, vs, vn = [-51.2 + 0.4, -51.2 + 0.4, -5.0 + 4.0], [0.8, 0.8, 8.0], [128, 128, 1]
    g = torch.Generator().manual_seed(7)
    sd = torch.bfloat16 if bf16 else torch.float32
    depth = torch.rand(B * N, D, fH, fW, generator=g).softmax(1).to(sd)
    ctx = torch.randn(B * N, C, fH, fW, generator=g).to(sd)
    go = torch.randn(B, C, vn[1], vn[0], generator=g).cuda().contiguous(memory_format=torch.channels_last)
    cache = new_plan_cache(N, D, fH, fW, vn, "cuda", slots=max(B, 2))
    for column in (False, True):
        d1 = depth.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        c1 = ctx.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        d2 = depth.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        c2 = ctx.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        plan_prepare(combine, axes, vn, vc, vs, cache)
        out_p = lift_splat_plan(combine, axes, d1, c1, vn, vc, vs, cache, prepared=True, column_backward=column)
        assert last_kernel_family() == "plan+camera"
        out_c = lift_splat_camera(combine, axes, d2, c2, vn, vc, vs, column_backward=column)
        scale = max(1.0, out_c.abs().max().item())
        assert (out_p - out_c).abs().max().item() <= 2e-5 * scale
        out_p.backward(go)
        fam = last_kernel_family(backward=True)
        out_c.backward(go)
        assert last_kernel_family(backward=True) == fam
        assert torch.equal(d1.grad, d2.grad)
        assert torch.equal(c1.grad, c2.grad)


@pytest.mark.parametrize("bf16", [False, True])
def test_plan_form_against_the_oracle_at_configs4_full_shape(mmt_lib, oracle_mod, bf16):
    """BASELINE configs[4]'s camera half in full -- (B, N, D, fH, fW, C) = (2, 6, 112, 32, 88, 80), fp32 and bf16 storage -- DIRECTLY
    against oracle.voxel_pooling_forward_f64(oracle.lift(...)) on the cells of mmt_frustum_geometry (not only against the camera
    form, as test_plan_form_equals_camera_form_with_gradients does): <= 1e-4 of the map's scale, every element written,
    bit-identical from call to call, the brute-force path as well."""
    from mm_training_amd import synthetic
    B, N, D, fH, fW, C = 2, 6, 112, 32, 88, 80
    H, W = fH * 16, fW * 16
    s2e, K = synthetic.camera_rig(B, N, W, H, jitter=0.02, seed=0)
    combine = s2e.matmul(torch.inverse(K)).contiguous()
    fr = _frustum((H, W), 16, (2.0, 2.0 + 0.5 * D, 0.5))
    vc, vs, vn = [-51.2 + 0.4, -51.2 + 0.4, -5.0 + 4.0], [0.8, 0.8, 8.0], [128, 128, 1]
    _case(combine, fr, vc, vs, vn, oracle_mod, C=C, bf16=bf16, seed=11, check_host_plan=False)


def test_plan_cache_duplicates_reordering_eviction_and_shape_change(mmt_lib, oracle_mod):
    """a batch whose samples share a calibration learns it once; reordered / mixed batches hit; more calibrations than slots
    evict the least recently used; a change of the frustum axes empties the table; every call's map is right"""
    from mm_training_amd import synthetic
    from mm_training_amd.ops.bev_geometry import frustum_axes, frustum_geometry, new_plan_cache, plan_cache_counters
    N, D, fH, fW, C = 3, 24, 16, 10, 64
    H, W = fH * 16, fW * 16
    fr = _frustum((H, W), 16, (2.0, 2.0 + 2.0 * D, 2.0))
    axes = tuple(a.cuda() for a in frustum_axes(fr))
    vc, vs, vn = [-51.2 + 0.4, -51.2 + 0.4, -1.0], [0.8, 0.8, 8.0], [128, 128, 1]
    rigs = []
    for seed in range(5):
        s2e, K = synthetic.camera_rig(1, N, W, H, jitter=0.3, seed=seed)
        rigs.append(s2e.matmul(torch.inverse(K))[0])
    g = torch.Generator().manual_seed(0)
    cache = new_plan_cache(N, D, fH, fW, vn, "cuda", slots=4)

    def run(ids, ax=axes, frus=fr):
        cb = torch.stack([rigs[i] for i in ids]).contiguous()
        B = len(ids)
        depth = torch.rand(B * N, fH, fW, D, generator=g).softmax(-1)
        ctx = torch.randn(B * N, fH, fW, C, generator=g)
        out = _forward(cb.cuda(), ax, vc, vs, vn, depth.cuda(), ctx.cuda(), cache).cpu().numpy()
        geom = frustum_geometry(frus.cuda(), cb.cuda(), vc, vs).cpu().numpy()
        feats = oracle_mod.lift(depth.permute(0, 3, 1, 2).contiguous().numpy(), ctx.permute(0, 3, 1, 2).contiguous().numpy()).reshape(B, -1, C)
        ref = oracle_mod.voxel_pooling_forward_f64(geom.reshape(B, -1, 3), feats, *vn)
        assert np.abs(out - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max()))
        return plan_cache_counters(cache)

    c = run([0, 0, 0, 0])
    assert (c["learnt"], c["hit"]) == (1, 0)
    c = run([0, 1, 0, 1])
    assert (c["learnt"], c["hit"]) == (2, 1)
    c = run([1, 0])
    assert (c["learnt"], c["hit"]) == (2, 3)
    c = run([2, 3, 2])                         # four calibrations in four slots
    assert (c["learnt"], c["hit"]) == (4, 3)
    c = run([4])                               # a fifth: evicts the least recently used (calibration 0 or 1; 1 and 0 were used in call 3, 2 and 3 in call 4)
    assert c["learnt"] == 5
    c = run([2, 3])                            # still there
    assert (c["learnt"], c["hit"]) == (5, 5)
    c = run([0, 1])                            # one of them was evicted, the other still known
    assert c["learnt"] == 6 and c["hit"] == 6
    # other frustum axes (same shape): the table is emptied, not served stale
    fr2 = _frustum((H, W), 16, (3.0, 3.0 + 2.0 * D, 2.0))
    axes2 = tuple(a.cuda() for a in frustum_axes(fr2))
    c = run([2], axes2, fr2)
    assert c["resets"] == 1 and (c["learnt"], c["hit"]) == (1, 0)


def test_plan_form_argument_checks(mmt_lib):
    from mm_training_amd import _lib
    from mm_training_amd.ops.bev_geometry import plan_form_supported
    assert plan_form_supported(4, 6, 112, 16, 44, 80, [128, 128, 1]) and plan_form_supported(2, 6, 112, 32, 88, 80, [128, 128, 1])
    assert plan_form_supported(4, 2, 409, 44, 80, 80, [512, 64, 1])
    assert not plan_form_supported(1, 1, 16, 4, 4, 48, [128, 128, 1]) and not plan_form_supported(1, 1, 3, 4, 4, 64, [128, 128, 1])     # C; D < 4
    assert not plan_form_supported(65, 1, 16, 4, 4, 64, [128, 128, 1]) and not plan_form_supported(1, 17, 16, 4, 4, 64, [128, 128, 1])
    lib = _lib.lib()
    need = lib.mmt_lss_plan_cache_bytes(1, 4, 2, 2, 8, 8, 2)
    assert need > 0 and lib.mmt_lss_plan_cache_bytes(0, 4, 2, 2, 8, 8, 2) == 0 and lib.mmt_lss_plan_cache_bytes(1, 4, 2, 2, 8, 8, 0) == 0
    t = torch.zeros(need + 4096, dtype=torch.uint8, device="cuda")
    base = (t.data_ptr() + 255) & ~255
    f = torch.zeros(4096, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    vc, vs = _lib.float3([0.4] * 3), _lib.float3([0.8] * 3)
    geo = [f.data_ptr()] * 4 + [vc, vs]
    ok = lambda *a: lib.mmt_lss_splat_forward_plan(*a)
    args = lambda B=1, C=64, flags=_lib.LSS_PIXEL_MAJOR, cache=base, nbytes=need, geo=geo: (B, 1, 4, 2, 2, C, 8, 8, 1, *geo, f.data_ptr(), f.data_ptr(), f.data_ptr(), 0, cache, nbytes, flags, st)
    assert ok(*args()) == 0
    assert ok(*args(flags=0)) == -4                                        # frustum point order: not in the plan form
    assert ok(*args(flags=_lib.LSS_PIXEL_MAJOR | 0x8000)) == -4
    assert ok(*args(C=48)) == -2
    assert ok(*args(cache=base + 16)) == -2                                # alignment
    assert ok(*args(B=3)) == -2                                            # two slots, three samples
    assert ok(*args(nbytes=4096)) == -2
    assert ok(*args(geo=[0] + geo[1:])) == -1
    torch.cuda.synchronize()


def test_prepared_flag_on_an_unprepared_cache_writes_nothing_and_is_counted(mmt_lib, golden):
    """ADVICE (round 5): mmt_lss_splat_forward_plan told MMT_LSS_PLAN_PREPARED goes by the verdicts in the cache.  On a cache no
    lookup has filled (verdict state Empty) it used to serve the samples from slot summaries that were never built -- a wrong map,
    no error.  Now such a sample's part of the map is left untouched and the header counts it (counters['stale']; LSSFPN's lazy
    read-back raises on it); the same call without the flag, and with it after a lookup, is correct."""
    from mm_training_amd.ops.bev_geometry import frustum_axes, new_plan_cache, plan_cache_counters, plan_prepare
    g = golden["quant_geom"]
    fr = torch.from_numpy(g["nusc_frustum"])
    cb = torch.from_numpy(g["rig_combine"]).contiguous().cuda()
    B, N = cb.shape[:2]
    D, fH, fW, _ = fr.shape
    vc, vs, vn = g["nusc_voxel_coord"], g["nusc_voxel_size"], [128, 128, 1]
    axes = tuple(a.cuda() for a in frustum_axes(fr))
    gen = torch.Generator().manual_seed(3)
    depth = torch.rand(B * N, fH, fW, D, generator=gen).softmax(-1).cuda()
    ctx = torch.randn(B * N, fH, fW, 64, generator=gen).cuda()
    cache = new_plan_cache(N, D, fH, fW, vn, "cuda", slots=max(B, 2))
    out = _forward(cb, axes, vc, vs, vn, depth, ctx, cache, prepared=True)
    assert bool(torch.isnan(out).all())                                    # nothing written
    # (the header is only signed by a lookup: before one, the counters read 0 -- the stale count shows after it)
    good = _forward(cb, axes, vc, vs, vn, depth, ctx, cache)
    assert not bool(torch.isnan(good).any())
    assert torch.equal(good, _forward(cb, axes, vc, vs, vn, depth, ctx, cache, prepared=True))
    assert plan_cache_counters(cache)["stale"] == 0
    # verdicts wiped behind the library's back (what a stale snapshot would look like): counted, nothing written
    lay = _layout(N, D, fH, fW, vn[0], vn[1], cache)
    o = lay["base"] + lay["verdict_off"]
    cache[o:o + 64 * B] = 0
    out = _forward(cb, axes, vc, vs, vn, depth, ctx, cache, prepared=True)
    assert bool(torch.isnan(out).all()) and plan_cache_counters(cache)["stale"] == B



@pytest.mark.parametrize("dtypes", [("f32", "f32"), ("f32", "bf16"), ("bf16", "bf16")])
@pytest.mark.parametrize("D", [40, 112, 200])
def test_lookup_riding_in_the_depth_softmax(mmt_lib, dtypes, D):
    """mmt_depth_softmax_forward_plan_prepare == mmt_depth_softmax_forward + mmt_lss_plan_prepare: the same probabilities bit for
    bit (the rows' arithmetic is shared), the same verdicts and counters as a lookup of its own on a twin cache -- through learning
    (first call), known calibrations (probe), batches seen before (snapshots) and a sample mix with a new calibration."""
    from mm_training_amd import synthetic
    from mm_training_amd.ops.bev_geometry import depth_softmax, frustum_axes, new_plan_cache, plan_cache_counters, plan_prepare
    lt, ut = [torch.bfloat16 if d == "bf16" else torch.float32 for d in dtypes]
    B, N, fH, fW = 3, 2, 16, 22
    H, W = fH * 16, fW * 16
    step = 56.0 / D
    fr = _frustum((H, W), 16, (2.0, 2.0 + D * step - 1e-3, step))
    axes = tuple(a.cuda() for a in frustum_axes(fr))
    assert axes[2].numel() == D
    vc, vs, vn = [-51.2 + 0.4, -51.2 + 0.4, -1.0], [0.8, 0.8, 8.0], [128, 128, 1]

    def batch(seed):
        s2e, K = synthetic.camera_rig(B, N, W, H, jitter=0.02, seed=seed)
        return s2e.matmul(torch.inverse(K)).contiguous().cuda()

    rider, twin = [new_plan_cache(N, D, fH, fW, vn, "cuda", slots=8) for _ in range(2)]
    lay = _layout(N, D, fH, fW, vn[0], vn[1], rider)
    g = torch.Generator().manual_seed(D)
    b0, b1 = batch(1), batch(2)
    mixed = torch.stack([b0[0], batch(3)[1], b1[2]]).contiguous()
    for it, comb in enumerate([b0, b0, b1, b0, b1, mixed, mixed, b0]):
        logits = (torch.randn(B * N, D, fH, fW, generator=g) * 3).cuda().to(lt).contiguous(memory_format=torch.channels_last)
        oracle = None
        if it % 2:
            oracle = torch.zeros(B * N, D, fH, fW)
            oracle[:, it % D, ::3, ::2] = 1.0
            oracle = oracle.cuda().contiguous(memory_format=torch.channels_last)
        p0, u0 = depth_softmax(logits, oracle, ut)
        plan_prepare(comb, axes, vn, vc, vs, twin)
        p1, u1 = depth_softmax(logits, oracle, ut, plan_lookup=(comb, axes, vn, vc, vs, rider))
        torch.cuda.synchronize()
        assert torch.equal(p0, p1) and torch.equal(u0, u1)
        assert np.array_equal(_verdicts(rider, lay, B), _verdicts(twin, _layout(N, D, fH, fW, vn[0], vn[1], twin), B)), it
        c0, c1 = plan_cache_counters(twin), plan_cache_counters(rider)
        assert c0 == c1 and c1["stale"] == 0, (it, c0, c1)
    assert c1["learnt"] == 2 * B + 1 and c1["calls"] == 8
    # the plans learnt inside the softmax's launch (256-thread workgroups) serve the forward like those of the lookup's own launch (1 024)
    C = 64
    depth = torch.rand(B * N, fH, fW, D, generator=g).cuda()
    ctx = torch.randn(B * N, fH, fW, C, generator=g).cuda()
    o_r = _forward(b0, axes, vc, vs, vn, depth, ctx, rider, prepared=True)
    o_t = _forward(b0, axes, vc, vs, vn, depth, ctx, twin, prepared=True)
    assert torch.equal(o_r, o_t) and bool(torch.isfinite(o_r).all())
    for slot in range(2 * B + 1):
        lr, lt_ = _layout(N, D, fH, fW, vn[0], vn[1], rider), _layout(N, D, fH, fW, vn[0], vn[1], twin)
        a_ = rider[lr["base"] + lr["slots_off"] + slot * lr["slot_bytes"]:lr["base"] + lr["slots_off"] + (slot + 1) * lr["slot_bytes"]]
        b_ = twin[lt_["base"] + lt_["slots_off"] + slot * lt_["slot_bytes"]:lt_["base"] + lt_["slots_off"] + (slot + 1) * lt_["slot_bytes"]]
        meta_a, meta_b = a_[:64].view(torch.int32).cpu().numpy(), b_[:64].view(torch.int32).cpu().numpy()
        assert np.array_equal(meta_a[:5], meta_b[:5]) and np.array_equal(meta_a[6:], meta_b[6:]), slot      # (word 5: the LRU stamp)
        njobs = int(meta_a[3])
        ra = a_[lr["records_off"]:lr["records_off"] + njobs * P.JOB_BYTES]
        rb = b_[lt_["records_off"]:lt_["records_off"] + njobs * P.JOB_BYTES]
        assert torch.equal(ra, rb), slot
    # rows that cannot carry the lookup (D % 4 != 0 here): depth_softmax makes the two launches itself
    # and the raw entry point refuses
    from mm_training_amd import _lib
    from mm_training_amd.ops.bev_geometry import _plan_ptr
    ptr, nbytes = _plan_ptr(rider)
    x = torch.randn(B * N * fH * fW, D + 2, device="cuda")
    out = torch.empty_like(x)
    with pytest.raises(RuntimeError, match="16-byte"):
        _lib.call("mmt_depth_softmax_forward_plan_prepare", B * N * fH * fW, D + 2, x.data_ptr(), D + 2, 0, out.data_ptr(), 0, 0, 0, 0, B, N, fH, fW,
                  vn[0], vn[1], vn[2], b0.data_ptr(), axes[0].data_ptr(), axes[1].data_ptr(), axes[2].data_ptr(), _lib.float3(vc), _lib.float3(vs), ptr, nbytes,
                  torch.cuda.current_stream().cuda_stream)
    with pytest.raises(RuntimeError, match="softmax rows for a batch"):
        _lib.call("mmt_depth_softmax_forward_plan_prepare", B * N * fH * fW - 1, D, x.data_ptr(), D + 2, 0, out.data_ptr(), 0, 0, 0, 0, B, N, fH, fW,
                  vn[0], vn[1], vn[2], b0.data_ptr(), axes[0].data_ptr(), axes[1].data_ptr(), axes[2].data_ptr(), _lib.float3(vc), _lib.float3(vs), ptr, nbytes,
                  torch.cuda.current_stream().cuda_stream)
