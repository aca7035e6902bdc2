"""GPU parity of the bf16 STORAGE path (SURVEY.md section 8 row g1 / BASELINE configs[4]: "bf16").

The reference has no bf16 behaviour -- its extension takes data_ptr<float>() only
(ops/voxel_pooling/src/voxel_pooling_forward.cpp:28-31; exps/conf_aim.py:30 "16 does not work yet") -- so SURVEY 5.6
defines it: bf16 storage of the big operands, fp32 accumulation, parity against the fp32 ORACLE evaluated on the
up-cast inputs.  Bars (stated here, checked below):
  * pos_memo bit-exact; pooled BEV features <= 1e-4 abs vs the fp64-accumulating oracle on the up-cast rows
    (accumulation is fp32, only the inputs were rounded, and the oracle sees the same rounded inputs);
  * grad_in bit-exact: the backward is a copy, the oracle's fp32 gather rounded to bf16 (nearest even) once;
  * bf16 lift: bit-exact with bf16(fp32(depth * context)) of the oracle's lift;
  * fused lift-splat: 1e-4 abs vs the oracle composition on the up-cast operands (fp32 products); its bf16 gradients
    within one bf16 ulp of the fp32 kernels' gradients on the same up-cast operands (the depth gradient is an LDS
    float-atomic sum, so its last fp32 bits depend on the order).
"""
import numpy as np
import pytest
import torch

from tests.golden.formula import hashed_f32

pytestmark = pytest.mark.gpu

ATOL = 1e-4
BF16_ULP = 2.0 ** -7      # relative spacing of bf16 values (8 significant bits)


def _close_bf16(a, b):
    a, b = a.float(), b.float()
    return bool(((a - b).abs() <= BF16_ULP * b.abs() + 1e-6).all())


def _geometry(kind, B, N, D, fH, fW):
    from mm_training_amd import synthetic
    if kind == "rig":
        geom, vn = synthetic.rig_geometry(B, N, (fH * 16, fW * 16), 16, (2.0, 2.0 + 0.5 * D, 0.5))
        assert tuple(geom.shape) == (B, N, D, fH, fW, 3)
        return geom, vn
    return synthetic.uniform_geometry(B, N * D * fH * fW, 128, 128).reshape(B, N, D, fH, fW, 3), [128, 128, 1]


@pytest.mark.parametrize("shape", [
    (2, 1, 10, 10, 60, 80, "uniform"),        # the reference test's size (P = 6000, C = 80)
    (1, 2, 7, 3, 5, 8, "uniform"),            # C = 8: one lane per row; P = 210 (not a multiple of anything)
    (3, 2, 14, 4, 11, 64, "rig"),
    (2, 6, 112, 32, 88, 80, "rig"),           # BASELINE configs[4] camera shape: 512x1408, P = 1 892 352 per sample
])
def test_voxel_pooling_bf16_against_fp32_oracle_on_upcast_inputs(mmt_lib, oracle_mod, shape):
    from mm_training_amd import synthetic
    from mm_training_amd.ops.voxel_pooling import voxel_pooling_bf16
    B, N, D, fH, fW, C, kind = shape
    geom, vn = _geometry(kind, B, N, D, fH, fW)
    P = N * D * fH * fW
    feats16 = synthetic.features((B, P, C), seed=2).bfloat16()
    up = feats16.float().numpy()                                            # what the kernel's fp32 accumulators see
    ref64 = oracle_mod.voxel_pooling_forward_f64(geom.reshape(B, P, 3).numpy(), up, *vn)
    _, ref_pos = oracle_mod.voxel_pooling_forward(geom.reshape(B, P, 3).numpy(), up, *vn)
    f = feats16.cuda().requires_grad_(True)
    out = voxel_pooling_bf16(geom.reshape(B, P, 3).cuda(), f, vn)
    assert out.dtype == torch.float32 and out.shape == (B, C, vn[1], vn[0])
    assert np.abs(out.detach().permute(0, 2, 3, 1).cpu().numpy() - ref64).max() <= ATOL
    grad_out = torch.from_numpy(hashed_f32((B, C, vn[1], vn[0]), salt=5)).cuda()
    for go in (grad_out.contiguous(memory_format=torch.channels_last), grad_out):       # fast path / layout pass
        f.grad = None
        out = voxel_pooling_bf16(geom.reshape(B, P, 3).cuda(), f, vn)
        out.backward(go)
        assert f.grad.dtype == torch.bfloat16
        ref_gi = torch.from_numpy(oracle_mod.voxel_pooling_backward(ref_pos, grad_out.cpu().numpy())).bfloat16()
        assert torch.equal(f.grad.cpu(), ref_gi), "grad_in = bf16(gathered fp32 row), exact"


@pytest.mark.parametrize("case", ["c64", "c80", "samecell"])
def test_voxel_pooling_bf16_edge_cases(mmt_lib, oracle_mod, golden, case):
    """The hand-made edge set of the fp32 op (coords -1 / 0 / nx-1 / nx, z filter, floats in (-1, 0), saturating values,
    500 points in one cell; the cases whose C is a multiple of 8) through the bf16 entry point: pos_memo identical to the
    reference-generated one (the index path does not depend on the storage type), BEV vs the oracle on the rounded rows,
    grad_in = the reference-generated fp32 gradient rounded to bf16, with and without workspaces."""
    from mm_training_amd.ops.voxel_pooling import voxel_pooling_ext
    g = golden["vp_edge"]
    nx, ny, nz = [int(v) for v in g["grid"]]
    geom, feats = g[case + "_geom"], g[case + "_feats"]
    B, P, C = feats.shape
    f16 = torch.from_numpy(feats).bfloat16()
    out = torch.zeros(B, ny, nx, C, device="cuda")
    pos = torch.full((B, P, 3), -1, dtype=torch.int32, device="cuda")
    voxel_pooling_ext.voxel_pooling_forward_wrapper_bf16(B, P, C, nx, ny, nz, torch.from_numpy(geom).cuda(), f16.cuda(), out, pos, flags=0)
    assert np.array_equal(pos.cpu().numpy(), g[case + "_pos_memo"])           # dropped rows left at the caller's -1
    ref64 = oracle_mod.voxel_pooling_forward_f64(geom, f16.float().numpy(), nx, ny, nz)
    assert np.abs(out.cpu().numpy() - ref64).max() <= ATOL
    go = torch.from_numpy(g[case + "_grad_out"]).cuda()
    full = voxel_pooling_ext.backward_workspace_elems(B, P, C, nx, ny)
    for ws in (None, torch.empty(B * ny * nx * C, device="cuda"), torch.empty(full, device="cuda")):
        gi = torch.empty(B, P, C, dtype=torch.bfloat16, device="cuda")
        voxel_pooling_ext.voxel_pooling_backward_wrapper_bf16(B, P, C, nx, ny, pos, go, gi, ws)
        assert torch.equal(gi.cpu(), torch.from_numpy(g[case + "_grad_in"]).bfloat16())


def test_bf16_wrappers_reject_what_they_cannot_take(mmt_lib):
    from mm_training_amd import _lib
    from mm_training_amd.ops.voxel_pooling import voxel_pooling_ext
    geom = torch.zeros(1, 8, 3, dtype=torch.int32, device="cuda")
    out = torch.zeros(1, 2, 2, 8, device="cuda")
    pos = torch.zeros(1, 8, 3, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="scalar type"):      # fp32 rows into the bf16 entry point
        voxel_pooling_ext.voxel_pooling_forward_wrapper_bf16(1, 8, 8, 2, 2, 1, geom, torch.zeros(1, 8, 8, device="cuda"), out, pos)
    with pytest.raises(RuntimeError, match="scalar type"):      # bf16 rows into the reference-signature fp32 entry point
        voxel_pooling_ext.voxel_pooling_forward_wrapper(1, 8, 8, 2, 2, 1, geom, torch.zeros(1, 8, 8, device="cuda").bfloat16(), out, pos)
    with pytest.raises(_lib.MmtError, match="C % 8"):
        voxel_pooling_ext.voxel_pooling_forward_wrapper_bf16(1, 8, 4, 2, 2, 1, geom, torch.zeros(1, 8, 4, device="cuda").bfloat16(),
                                                             torch.zeros(1, 2, 2, 4, device="cuda"), pos)
    with pytest.raises(_lib.MmtError, match="flag"):
        voxel_pooling_ext.voxel_pooling_forward_wrapper_bf16(1, 8, 8, 2, 2, 1, geom, torch.zeros(1, 8, 8, device="cuda").bfloat16(), out, pos, flags=1)


@pytest.mark.parametrize("cfg", [(3, 14, 4, 11, 16), (24, 112, 16, 44, 80)])
def test_lift_bf16_storage(mmt_lib, oracle_mod, cfg):
    from mm_training_amd.ops.bev_geometry import lift_features
    BN, D, fH, fW, C = cfg
    g = torch.Generator().manual_seed(3)
    depth = torch.rand(BN, D, fH, fW, generator=g).softmax(1)
    ctx = torch.randn(BN, C, fH, fW, generator=g)
    d1 = depth.cuda().requires_grad_(True)
    c1 = ctx.cuda().requires_grad_(True)
    feats = lift_features(d1, c1, torch.bfloat16)
    assert feats.dtype == torch.bfloat16 and feats.shape == (BN, D, fH, fW, C)
    ref = torch.from_numpy(oracle_mod.lift(depth.numpy(), ctx.numpy())).bfloat16()     # bf16(fp32(depth * context))
    assert torch.equal(feats.detach().cpu(), ref.view_as(feats))
    # backward: a bf16 gradient, fp32 sums == the fp32 lift backward fed the up-cast gradient
    go = torch.randn(feats.shape, generator=g).bfloat16().cuda()
    feats.backward(go)
    d2 = depth.cuda().requires_grad_(True)
    c2 = ctx.cuda().requires_grad_(True)
    lift_features(d2, c2).backward(go.float())
    assert torch.allclose(d1.grad, d2.grad, rtol=1e-5, atol=1e-6) and torch.allclose(c1.grad, c2.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("cfg", [(1, 2, 14, 4, 11, 16, "rig"), (2, 6, 112, 32, 88, 80, "rig"), (2, 6, 30, 16, 44, 64, "uniform")])
def test_fused_lift_splat_bf16(mmt_lib, oracle_mod, cfg):
    from mm_training_amd.ops.bev_geometry import lift_splat
    B, N, D, fH, fW, C, kind = cfg
    geom, vn = _geometry(kind, B, N, D, fH, fW)
    g = torch.Generator().manual_seed(0)
    depth16 = torch.rand(B * N, D, fH, fW, generator=g).softmax(1).bfloat16()
    ctx16 = torch.randn(B * N, C, fH, fW, generator=g).bfloat16()
    geom_d = geom.cuda()
    d1 = depth16.cuda().requires_grad_(True)
    c1 = ctx16.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = lift_splat(geom_d, d1, c1, vn)
    assert out.dtype == torch.float32
    feats = oracle_mod.lift(depth16.float().numpy(), ctx16.float().numpy()).reshape(B, -1, C)      # fp32 products of the up-cast operands
    ref = oracle_mod.voxel_pooling_forward_f64(geom.reshape(B, -1, 3).numpy(), feats, *vn)
    assert np.abs(out.detach().permute(0, 2, 3, 1).cpu().numpy() - ref).max() <= ATOL
    # gradients: the fp32 kernels on the same up-cast operands, rounded to bf16
    d2 = depth16.float().cuda().requires_grad_(True)
    c2 = ctx16.float().cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out2 = lift_splat(geom_d, d2, c2, vn)
    go = torch.randn(out.shape, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    out.backward(go)
    out2.backward(go)
    assert d1.grad.dtype == torch.bfloat16 and c1.grad.dtype == torch.bfloat16
    assert _close_bf16(d1.grad, d2.grad) and _close_bf16(c1.grad, c2.grad)
    # pixel-major layout, bf16 operands
    d3 = depth16.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    c3 = ctx16.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out3 = lift_splat(geom_d.permute(0, 1, 3, 4, 2, 5).contiguous(), d3, c3, vn, pixel_major=True)
    assert np.abs(out3.detach().permute(0, 2, 3, 1).cpu().numpy() - ref).max() <= ATOL
    out3.backward(go)
    assert _close_bf16(d3.grad, d2.grad) and _close_bf16(c3.grad, c2.grad)
    # the matrix-core column backward on bf16 operands (bf16 grad_depth leaves as 8-byte stores), both point orders; its fp32 sums
    # are taken in another order than the walk's (1e-5 on gradients of magnitude 50), so near-zero elements get an absolute
    # allowance scaled to the tensor instead of 1e-6
    def _close_col(a, b):
        a, b = a.float(), b.float()
        return bool(((a - b).abs() <= BF16_ULP * b.abs() + 1e-6 * max(1.0, b.abs().max().item())).all())
    d4 = depth16.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    c4 = ctx16.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    lift_splat(geom_d.permute(0, 1, 3, 4, 2, 5).contiguous(), d4, c4, vn, pixel_major=True, column_backward=True).backward(go)
    assert _close_col(d4.grad, d2.grad) and _close_col(c4.grad, c2.grad)
    d5, c5 = depth16.cuda().requires_grad_(True), ctx16.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    lift_splat(geom_d, d5, c5, vn, column_backward=True).backward(go)
    assert _close_col(d5.grad, d2.grad) and _close_col(c5.grad, c2.grad)


@pytest.mark.parametrize("fused", [True, False])
def test_lssfpn_bf16_hot_path_and_train_step(mmt_lib, fused):
    """LSSFPN.hot_path_dtype = "bf16" (what make_config("cfg5") selects): the BEV map stays within bf16 rounding of the
    fp32 path (inputs rounded to 8 significant bits, sums fp32) and a training step runs through the bf16 kernels."""
    from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
    cfg = make_config("tiny")
    cfg["hot_path_dtype"] = "bf16"
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    ts = TrainStep(cfg, dev, lr=2e-4)
    lss = ts.model.backbone
    lss.fused_lift_splat = fused
    assert lss.hot_path_dtype == "bf16"
    batch = synthetic_batch(cfg, dev, seed=3)
    x = ts.normalize_images(batch[0])
    ts.model.eval()
    with torch.no_grad():
        bev16 = lss(x, batch[1])
        lss.hot_path_dtype = "f32"
        bev32 = lss(x, batch[1])
        lss.hot_path_dtype = "bf16"
    assert bev16.dtype == torch.float32
    assert (bev16 - bev32).abs().max().item() <= 2.0 ** -6 * bev32.abs().max().item() + 1e-6
    ts.model.train()
    losses = [float(ts(batch)[0]) for _ in range(5)]
    assert all(l == l and abs(l) < 1e6 for l in losses) and losses[-1] < losses[0]
