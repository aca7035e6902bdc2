"""GPU parity tests of the cached-plan voxel pooling against the
oracle and the committed golden vectors.

Bar: the plan's pos_memo bit-exact; pooled BEV features within 1e-4 abs of the fp64 oracle;
bit-reproducible run to run; backward (through the cached pos_memo) bit-exact."""
import numpy as np
import pytest
import torch

from tests.golden.formula import hashed_f32

pytestmark = pytest.mark.gpu

ATOL = 1e-4


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _planned(geom, feats, vn):
    from mm_training_amd.ops.voxel_pooling import VoxelPoolingPlan, voxel_pooling_planned
    plan = VoxelPoolingPlan(geom, vn)
    out = voxel_pooling_planned(plan, feats)
    torch.cuda.synchronize()
    return plan, out


def test_reference_known_answer_test(mmt_lib, golden):
    """The reference's own test (test/test_ops/test_voxel_pooling.py) at its own shape."""
    g = golden["vp_ref_test"]
    geom, feats = _dev(g["geom"]), _dev(g["feats"]).requires_grad_(True)
    plan, out = _planned(geom, feats, [128, 128, 1])
    assert torch.equal(plan.pos_memo.cpu(), torch.from_numpy(g["pos_memo"]))
    assert out.shape == (2, 80, 128, 128) and out.is_contiguous(memory_format=torch.channels_last)
    gt = torch.from_numpy(g["out_nhwc"])
    assert (out.detach().permute(0, 2, 3, 1).cpu() - gt).abs().max().item() <= ATOL
    assert torch.allclose(gt.permute(0, 3, 1, 2).cuda(), out.detach(), rtol=1e-3, atol=1e-6)
    out.backward(_dev(hashed_f32((2, 80, 128, 128), salt=1)))
    assert torch.equal(feats.grad.cpu(), torch.from_numpy(g["grad_in"]))
    assert plan.num_kept == int((g["pos_memo"][..., 0] != -1).sum())


@pytest.mark.parametrize("case", ["c64", "c80", "alldrop", "samecell"])
def test_edge_cases(mmt_lib, golden, case):
    g = golden["vp_edge"]
    vn = [int(v) for v in g["grid"]]
    geom, feats = _dev(g[case + "_geom"]), _dev(g[case + "_feats"])
    if feats.shape[-1] % 4:
        pytest.skip("planned forward needs C % 4 == 0")
    plan, out = _planned(geom, feats, vn)
    assert torch.equal(plan.pos_memo.cpu(), torch.from_numpy(g[case + "_pos_memo"]))
    ref = torch.from_numpy(g[case + "_out_nchw"])
    assert (out.cpu() - ref).abs().max().item() <= ATOL


def test_rejects_unsupported_channels(mmt_lib):
    from mm_training_amd import _lib
    from mm_training_amd.ops.voxel_pooling import VoxelPoolingPlan, voxel_pooling_planned
    geom = torch.zeros(1, 8, 3, dtype=torch.int32, device="cuda")
    plan = VoxelPoolingPlan(geom, [2, 2, 1])
    with pytest.raises(_lib.MmtError, match="multiple of 4"):
        voxel_pooling_planned(plan, torch.zeros(1, 8, 6, device="cuda"))
    with pytest.raises(RuntimeError, match="do not match the plan"):
        voxel_pooling_planned(plan, torch.zeros(1, 9, 8, device="cuda"))
    with pytest.raises(RuntimeError, match="CUDA"):
        VoxelPoolingPlan(geom.cpu(), [2, 2, 1])


@pytest.mark.parametrize("seed", range(10))
def test_randomised_shapes(mmt_lib, oracle_mod, seed):
    """Random (B, P, C, grid) incl. nz > 1, hot cells longer than many items, single-point and
    empty cells; compared with the oracle and with the drop-in op."""
    from mm_training_amd.ops.voxel_pooling import voxel_pooling
    rng = np.random.default_rng(700 + seed)
    B = int(rng.integers(1, 4))
    P = int(rng.choice([1, 31, 32, 33, 63, 513, 1000, 2049, 5000, 20000]))
    C = int(rng.choice([4, 12, 20, 64, 80, 96, 256]))
    nx, ny, nz = int(rng.integers(1, 40)), int(rng.integers(1, 40)), int(rng.integers(1, 3))
    geom = np.stack([rng.integers(-2, nx + 2, (B, P)), rng.integers(-2, ny + 2, (B, P)),
                     rng.integers(-1, nz + 1, (B, P))], -1).astype(np.int32)
    if seed % 3 == 0:   # hot cell: hundreds of items folded by the second kernel
        geom[:, : P // 2] = [min(1, nx - 1), min(2, ny - 1), 0]
    feats = (rng.random((B, P, C), dtype=np.float32) - 0.5)
    ref, ref_pos = oracle_mod.voxel_pooling_forward(geom, feats, nx, ny, nz)
    ref64 = oracle_mod.voxel_pooling_forward_f64(geom, feats, nx, ny, nz)
    g, f = _dev(geom), _dev(feats).requires_grad_(True)
    plan, out = _planned(g, f, [nx, ny, nz])
    assert torch.equal(plan.pos_memo.cpu(), torch.from_numpy(ref_pos))
    got = out.detach().permute(0, 2, 3, 1).cpu().numpy()
    assert np.abs(got - ref64).max() <= ATOL
    assert plan.num_kept == int((ref_pos[..., 0] != -1).sum())
    assert plan.num_items >= B * ny * nx
    # against the drop-in op (different summation order, same tolerance)
    drop_in = voxel_pooling(g, f.detach(), [nx, ny, nz])
    assert (drop_in - out.detach()).abs().max().item() <= ATOL
    go = rng.standard_normal((B, C, ny, nx)).astype(np.float32)
    out.backward(_dev(go))
    assert np.array_equal(f.grad.cpu().numpy(), oracle_mod.voxel_pooling_backward(ref_pos, go))


def test_bit_reproducible_and_plan_reuse(mmt_lib):
    """The same plan serves any number of steps; two runs agree bit for bit (no atomics), and a
    second plan of the same geometry gives the same bits (stable sort)."""
    from mm_training_amd import synthetic
    from mm_training_amd.ops.voxel_pooling import VoxelPoolingPlan, voxel_pooling_planned
    geom, vn = synthetic.rig_geometry(2)
    geom = geom.cuda()
    shape = tuple(geom.shape[:-1]) + (80,)
    plan = VoxelPoolingPlan(geom, vn)
    f1, f2 = synthetic.features(shape, 1).cuda(), synthetic.features(shape, 2).cuda()
    a1 = voxel_pooling_planned(plan, f1).clone()
    a2 = voxel_pooling_planned(plan, f2).clone()
    b1 = voxel_pooling_planned(plan, f1)
    assert torch.equal(a1, b1)
    assert not torch.equal(a1, a2)
    c1 = voxel_pooling_planned(VoxelPoolingPlan(geom, vn), f1)
    assert torch.equal(a1, c1)
    # linearity
    a12 = voxel_pooling_planned(plan, f1 + 2 * f2)
    assert (a12 - (a1 + 2 * a2)).abs().max().item() <= 2 * ATOL


def test_sorted_order_is_stable(mmt_lib):
    """Inside the plan, the kept points are ordered by cell and, within a cell, by ascending
    point index (what makes the summation order a function of the geometry only)."""
    rng = np.random.default_rng(5)
    B, P, nx, ny = 2, 3000, 7, 5
    geom = np.stack([rng.integers(-1, nx + 1, (B, P)), rng.integers(-1, ny + 1, (B, P)),
                     np.zeros((B, P), np.int64)], -1).astype(np.int32)
    from mm_training_amd.ops.voxel_pooling import VoxelPoolingPlan
    plan = VoxelPoolingPlan(_dev(geom), [nx, ny, 1])
    order = plan.plan[16:16 + plan.num_kept].cpu().numpy().astype(np.int64)
    x, y = geom[..., 0].reshape(-1), geom[..., 1].reshape(-1)
    kept = (x >= 0) & (x < nx) & (y >= 0) & (y < ny)
    assert plan.num_kept == int(kept.sum())
    cell = (np.arange(B * P) // P * ny + y) * nx + x
    expect = np.lexsort((np.arange(B * P)[kept], cell[kept]))
    assert np.array_equal(order, np.arange(B * P)[kept][expect])


def test_writes_into_wider_channels_last_buffer(mmt_lib, oracle_mod):
    """out_row_stride > C: the pooled map lands in a channel slice of the camera|LiDAR concat buffer
    (models/bev_depth.py:187-192) and leaves the other channels alone."""
    from mm_training_amd.ops.voxel_pooling import VoxelPoolingPlan
    from mm_training_amd.ops.voxel_pooling.plan import planned_forward_into
    rng = np.random.default_rng(9)
    B, P, C, nx, ny, extra = 2, 4000, 80, 16, 12, 64
    geom = np.stack([rng.integers(-1, nx + 1, (B, P)), rng.integers(-1, ny + 1, (B, P)),
                     np.zeros((B, P), np.int64)], -1).astype(np.int32)
    feats = (rng.random((B, P, C), dtype=np.float32) - 0.5)
    ref64 = oracle_mod.voxel_pooling_forward_f64(geom, feats, nx, ny, 1)
    plan = VoxelPoolingPlan(_dev(geom), [nx, ny, 1])
    fused = torch.full((B, ny, nx, C + extra), 7.0, device="cuda")
    planned_forward_into(plan, _dev(feats), fused, C + extra)
    torch.cuda.synchronize()
    assert np.abs(fused[..., :C].cpu().numpy() - ref64).max() <= ATOL
    assert bool((fused[..., C:] == 7.0).all())


@pytest.mark.parametrize("name", ["cfg1_full", "cfg2", "cfg5"])
def test_full_size_against_oracle(mmt_lib, oracle_mod, name):
    from mm_training_amd import synthetic
    from tests.test_voxel_pooling_gpu import SHAPES
    B, N, D, fH, fW, C = SHAPES[name]
    ds = 16 if D == 112 else 8
    d_bound = (2.0, 58.0, 0.5) if D == 112 else (1.0, 60.0, 0.5)
    geom, vn = synthetic.rig_geometry(B, N, (fH * ds, fW * ds), ds, d_bound)
    feats = synthetic.features((B, N, D, fH, fW, C), seed=1)
    P = N * D * fH * fW
    ref64 = oracle_mod.voxel_pooling_forward_f64(geom.reshape(B, P, 3).numpy(), feats.reshape(B, P, C).numpy(), *vn)
    _, ref_pos = oracle_mod.voxel_pooling_forward(geom.reshape(B, P, 3).numpy(), feats.reshape(B, P, C).numpy(), *vn)
    f = feats.cuda().requires_grad_(True)
    plan, out = _planned(geom.cuda(), f, vn)
    assert np.array_equal(plan.pos_memo.cpu().numpy(), ref_pos)
    got = out.detach().permute(0, 2, 3, 1).cpu().numpy()
    assert np.abs(got - ref64).max() <= ATOL
    grad_out = torch.from_numpy(hashed_f32((B, C, vn[1], vn[0]), salt=3)).cuda()
    out.backward(grad_out.contiguous(memory_format=torch.channels_last))
    assert np.array_equal(f.grad.reshape(B, P, C).cpu().numpy(),
                          oracle_mod.voxel_pooling_backward(ref_pos, grad_out.cpu().numpy()))
