"""GPU parity tests of voxel_pooling (forward + backward) through the Python mirror
and the C ABI, against the oracle and the committed golden vectors.

Bar: pos_memo / kept mask bit-exact; pooled BEV features within 1e-4 abs of the
oracle (BASELINE.json north_star); backward bit-exact (pure copy)."""
import numpy as np
import pytest
import torch

from tests.golden.formula import hashed_f32

pytestmark = pytest.mark.gpu

ATOL = 1e-4  # north_star tolerance on the pooled BEV tensor


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _run_ext(mmt_lib, geom, feats, nx, ny, nz, flags):
    from mm_training_amd.ops.voxel_pooling import voxel_pooling_ext
    B, P, C = feats.shape
    out = torch.zeros(B, ny, nx, C, device="cuda")
    pos = torch.full((B, P, 3), -1, dtype=torch.int32, device="cuda")
    voxel_pooling_ext.voxel_pooling_forward_wrapper(B, P, C, nx, ny, nz, geom, feats, out, pos, flags=flags)
    torch.cuda.synchronize()
    return out, pos


@pytest.mark.parametrize("algo", [0, 1, 2, 3, 4, 0x24])
def test_reference_known_answer_test(mmt_lib, oracle_mod, golden, algo):
    """The reference's own test (test/test_ops/test_voxel_pooling.py) at its own shape."""
    g = golden["vp_ref_test"]
    geom, feats = _dev(g["geom"]), _dev(g["feats"])
    out, pos = _run_ext(mmt_lib, geom, feats, 128, 128, 1, algo)
    assert torch.equal(pos.cpu(), torch.from_numpy(g["pos_memo"]))
    gt = torch.from_numpy(g["out_nhwc"])
    assert (out.cpu() - gt).abs().max().item() <= ATOL
    # the reference's acceptance criterion, on the permuted view it returns
    assert torch.allclose(gt.permute(0, 3, 1, 2).cuda(), out.permute(0, 3, 1, 2), rtol=1e-3, atol=1e-6)


def test_python_op_forward_backward_golden(mmt_lib, golden):
    from mm_training_amd.ops.voxel_pooling import voxel_pooling
    g = golden["vp_ref_test"]
    geom = _dev(g["geom"]).reshape(2, 6, 10, 10, 10, 3).contiguous()
    feats = _dev(g["feats"]).reshape(2, 6, 10, 10, 10, 80).contiguous().requires_grad_(True)
    out = voxel_pooling(geom, feats, torch.tensor([128, 128, 1], dtype=torch.int, device="cuda"))
    assert out.shape == (2, 80, 128, 128)
    # same view layout as the reference: permute(0,3,1,2) of a [B,ny,nx,C] buffer
    assert out.stride() == (128 * 128 * 80, 1, 128 * 80, 80)
    assert out.is_contiguous(memory_format=torch.channels_last)
    assert (out.detach().permute(0, 2, 3, 1).cpu() - torch.from_numpy(g["out_nhwc"])).abs().max() <= ATOL
    grad_out = _dev(hashed_f32((2, 80, 128, 128), salt=1))       # NCHW-contiguous gradient
    out.backward(grad_out)
    assert feats.grad.shape == feats.shape
    assert torch.equal(feats.grad.reshape(2, -1, 80).cpu(), torch.from_numpy(g["grad_in"]))
    # channels-last gradient takes the no-transpose path (+ prepare pass) and must agree bit for bit
    feats.grad = None
    out2 = voxel_pooling(geom, feats, [128, 128, 1])
    out2.backward(grad_out.contiguous(memory_format=torch.channels_last))
    assert torch.equal(feats.grad.reshape(2, -1, 80).cpu(), torch.from_numpy(g["grad_in"]))


@pytest.mark.parametrize("case", ["c1", "c3", "c64", "c80", "c81", "alldrop", "samecell"])
@pytest.mark.parametrize("algo", [0, 1, 2, 3, 4, 0x23, 0x24])
def test_edge_cases(mmt_lib, golden, case, algo):
    from mm_training_amd.ops.voxel_pooling import voxel_pooling_ext
    g = golden["vp_edge"]
    nx, ny, nz = [int(v) for v in g["grid"]]
    geom, feats = _dev(g[case + "_geom"]), _dev(g[case + "_feats"])
    out, pos = _run_ext(mmt_lib, geom, feats, nx, ny, nz, algo)
    assert torch.equal(pos.cpu(), torch.from_numpy(g[case + "_pos_memo"]))
    ref = torch.from_numpy(g[case + "_out_nchw"]).permute(0, 2, 3, 1)
    assert (out.cpu() - ref).abs().max().item() <= ATOL
    # WRITE_DROPPED variant fills pos_memo itself
    B, P, C = feats.shape
    out2 = torch.zeros_like(out)
    pos2 = torch.empty_like(pos)
    voxel_pooling_ext.voxel_pooling_forward_wrapper(B, P, C, nx, ny, nz, geom, feats, out2, pos2, flags=algo | 0x10)
    assert torch.equal(pos2.cpu(), torch.from_numpy(g[case + "_pos_memo"]))
    # backward, NCHW gradient with and without workspace (strided slow path)
    go = _dev(g[case + "_grad_out"])
    full = voxel_pooling_ext.backward_workspace_elems(B, P, C, nx, ny)
    assert full == B * ny * nx * C + B * P
    for ws in (None, torch.empty(B * ny * nx * C, device="cuda"), torch.empty(full, device="cuda")):
        gi = torch.empty(B, P, C, device="cuda")
        voxel_pooling_ext.voxel_pooling_backward_wrapper(B, P, C, nx, ny, pos, go, gi, ws)
        assert torch.equal(gi.cpu(), torch.from_numpy(g[case + "_grad_in"]))


def test_accumulates_into_output_like_reference(mmt_lib):
    """out is += (atomicAdd into the caller's buffer, .cu:30-34), not overwritten."""
    torch.manual_seed(0)
    geom = torch.randint(0, 8, (1, 300, 3), dtype=torch.int32, device="cuda")
    geom[..., 2] = 0
    feats = torch.rand(1, 300, 16, device="cuda")
    from mm_training_amd.ops.voxel_pooling import voxel_pooling_ext
    out = torch.full((1, 8, 8, 16), 2.0, device="cuda")
    pos = torch.full((1, 300, 3), -1, dtype=torch.int32, device="cuda")
    voxel_pooling_ext.voxel_pooling_forward_wrapper(1, 300, 16, 8, 8, 1, geom, feats, out, pos)
    out0, _ = _run_ext(mmt_lib, geom, feats, 8, 8, 1, 0)
    assert (out - 2.0 - out0).abs().max().item() <= 1e-5


@pytest.mark.parametrize("widen", [1, 2, 7])
def test_backward_gradient_as_channel_slice_of_a_wider_buffer(mmt_lib, oracle_mod, widen):
    """grad_out handed over as the first C channels of a channels-last buffer `widen` times as wide (what
    autograd gives when the pooled map was written into the camera|LiDAR concat buffer): the prepare pass
    sweeps the whole span when it is at most 4x the dense size (widen 1, 2) and skips the sweep beyond (7)."""
    from mm_training_amd.ops.voxel_pooling import voxel_pooling_ext
    rng = np.random.default_rng(widen)
    B, P, C, nx, ny = 2, 3000, 80, 24, 20
    geom = np.stack([rng.integers(-2, nx + 2, (B, P)), rng.integers(-2, ny + 2, (B, P)), np.zeros((B, P), np.int64)], -1).astype(np.int32)
    feats = (rng.random((B, P, C), dtype=np.float32) - 0.5)
    _, ref_pos = oracle_mod.voxel_pooling_forward(geom, feats, nx, ny, 1)
    wide = torch.from_numpy(rng.standard_normal((B, ny, nx, C * widen)).astype(np.float32)).cuda()
    grad = wide.permute(0, 3, 1, 2)[:, :C]                       # [B, C, ny, nx] view, channel stride 1, row stride C*widen
    ref_gi = oracle_mod.voxel_pooling_backward(ref_pos, grad.contiguous().cpu().numpy())
    gi = torch.empty(B, P, C, device="cuda")
    ws = torch.empty(voxel_pooling_ext.backward_workspace_elems(B, P, C, nx, ny), device="cuda")
    voxel_pooling_ext.voxel_pooling_backward_wrapper(B, P, C, nx, ny, _dev(ref_pos), grad, gi, ws)
    assert np.array_equal(gi.cpu().numpy(), ref_gi)


def test_dispatch_attached_kernel_timing(mmt_lib):
    """bench.py's live roofline figure: with mm_training_amd._lib.TIMING set, forward and backward carry HIP events
    attached to their kernel dispatches (mmt_arm_kernel_timing); results are unchanged, every pair yields a
    positive duration, and nothing stays armed afterwards."""
    from mm_training_amd.ops.voxel_pooling import voxel_pooling, voxel_pooling_ext
    torch.manual_seed(1)
    B, P, C, nx, ny = 2, 40000, 80, 64, 48
    geom = torch.stack([torch.randint(-2, nx + 2, (B, P)), torch.randint(-2, ny + 2, (B, P)), torch.zeros(B, P, dtype=torch.long)], -1).int().cuda()
    feats = torch.rand(B, P, C, device="cuda")
    go = torch.randn(B, C, ny, nx, device="cuda")
    f0 = feats.clone().requires_grad_(True)
    ref = voxel_pooling(geom, f0, [nx, ny, 1])
    ref.backward(go)
    mmt_lib.TIMING = {}
    try:
        for grad in (go, go.contiguous(memory_format=torch.channels_last)):      # with and without the layout pass
            f1 = feats.clone().requires_grad_(True)
            out = voxel_pooling(geom, f1, [nx, ny, 1])
            out.backward(grad)
            assert (out - ref).abs().max().item() <= ATOL and torch.equal(f1.grad, f0.grad)
        # a non-default algorithm is timed by events recorded around the launch instead
        _run_ext(mmt_lib, geom, feats, nx, ny, 1, 1)
        torch.cuda.synchronize()
        timing = mmt_lib.TIMING
    finally:
        mmt_lib.TIMING = None
    assert len(timing["forward"]) == 3 and len(timing["backward"]) == 2
    for s_, e_ in timing["forward"] + timing["backward"]:
        ms = s_.elapsed_time(e_)
        assert 0.0 < ms < 50.0, ms
    # nothing left armed: an untimed call runs as usual
    out2, _ = _run_ext(mmt_lib, geom, feats, nx, ny, 1, 0)
    assert (out2.permute(0, 3, 1, 2) - ref).abs().max().item() <= ATOL


def test_error_behaviour(mmt_lib):
    from mm_training_amd.ops.voxel_pooling import voxel_pooling, voxel_pooling_ext
    geom = torch.zeros(1, 8, 3, dtype=torch.int32, device="cuda")
    feats = torch.zeros(1, 8, 4, device="cuda")
    out = torch.zeros(1, 2, 2, 4, device="cuda")
    pos = torch.zeros(1, 8, 3, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="contiguous"):
        voxel_pooling_ext.voxel_pooling_forward_wrapper(1, 8, 2, 2, 2, 1, geom, feats[..., ::2], out, pos)
    with pytest.raises(RuntimeError, match="scalar type"):
        voxel_pooling_ext.voxel_pooling_forward_wrapper(1, 8, 4, 2, 2, 1, geom.long(), feats, out, pos)
    with pytest.raises(RuntimeError, match="scalar type"):
        voxel_pooling_ext.voxel_pooling_forward_wrapper(1, 8, 4, 2, 2, 1, geom, feats.half(), out, pos)
    with pytest.raises(RuntimeError, match="CUDA"):
        voxel_pooling_ext.voxel_pooling_forward_wrapper(1, 8, 4, 2, 2, 1, geom.cpu(), feats, out, pos)
    with pytest.raises(AssertionError):
        voxel_pooling(geom, torch.zeros(1, 8, 8, device="cuda")[..., ::2], [2, 2, 1])


SHAPES = {
    # name: (B, N, D, fH, fW, C)  -- SURVEY.md section 8 shape table
    "cfg1_literal": (1, 6, 118, 32, 1, 64),
    "cfg1_full": (1, 6, 118, 32, 88, 64),
    "cfg2": (4, 6, 112, 16, 44, 80),
    "cfg5": (2, 6, 112, 32, 88, 80),
}


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("geometry", ["rig", "uniform"])
def test_full_size_against_oracle(mmt_lib, oracle_mod, name, geometry):
    from mm_training_amd import synthetic
    from mm_training_amd.ops.voxel_pooling import voxel_pooling
    B, N, D, fH, fW, C = SHAPES[name]
    if geometry == "rig":
        ds = 16 if D == 112 else 8
        final = (fH * ds, max(fW * ds, ds))
        d_bound = (2.0, 58.0, 0.5) if D == 112 else (1.0, 60.0, 0.5)
        geom, vn = synthetic.rig_geometry(B, N, final, ds, d_bound)
        assert tuple(geom.shape[1:5]) == (N, D, fH, fW)
    else:
        geom = synthetic.uniform_geometry(B, N * D * fH * fW, 128, 128).reshape(B, N, D, fH, fW, 3)
        vn = [128, 128, 1]
    feats = synthetic.features((B, N, D, fH, fW, C), seed=1)
    P = N * D * fH * fW
    ref, ref_pos = oracle_mod.voxel_pooling_forward(geom.reshape(B, P, 3).numpy(), feats.reshape(B, P, C).numpy(), *vn)
    ref64 = oracle_mod.voxel_pooling_forward_f64(geom.reshape(B, P, 3).numpy(), feats.reshape(B, P, C).numpy(), *vn)

    f = feats.cuda().requires_grad_(True)
    out = voxel_pooling(geom.cuda(), f, vn)
    got = out.detach().permute(0, 2, 3, 1).cpu().numpy()
    assert np.abs(got - ref64).max() <= ATOL, "pooled BEV features vs fp64 oracle"
    assert np.abs(got - ref).max() <= 2 * ATOL

    grad_out = torch.from_numpy(hashed_f32((B, C, vn[1], vn[0]), salt=3)).cuda()
    out.backward(grad_out.contiguous(memory_format=torch.channels_last))
    ref_gi = oracle_mod.voxel_pooling_backward(ref_pos, grad_out.cpu().numpy())
    assert np.array_equal(f.grad.reshape(B, P, C).cpu().numpy(), ref_gi)

    # size-independent properties
    kept = ref_pos[..., 0] != -1
    # checksum: everything kept lands somewhere, nothing else does
    tot = feats.reshape(B, P, C).numpy()[kept].astype(np.float64).sum()
    assert abs(got.astype(np.float64).sum() - tot) <= 1e-6 * max(1.0, kept.sum()) * 0.5 + 1e-2
    # adjoint identity <fwd(f), g> == <f, bwd(g)>
    lhs = (got.astype(np.float64) * grad_out.permute(0, 2, 3, 1).cpu().numpy()).sum()
    rhs = (feats.reshape(B, P, C).numpy().astype(np.float64) * ref_gi).sum()
    assert abs(lhs - rhs) <= 1e-3 * max(1.0, abs(rhs)) ** 0.5 + 5e-2


def test_linearity_and_determinism_of_index_path(mmt_lib):
    from mm_training_amd import synthetic
    from mm_training_amd.ops.voxel_pooling import voxel_pooling
    geom, vn = synthetic.rig_geometry(2)
    geom = geom.cuda()
    shape = tuple(geom.shape[:-1]) + (80,)
    f1, f2 = synthetic.features(shape, 1).cuda(), synthetic.features(shape, 2).cuda()
    o1 = voxel_pooling(geom, f1, vn)
    o2 = voxel_pooling(geom, f2, vn)
    o12 = voxel_pooling(geom, f1 + 2 * f2, vn)
    assert (o12 - (o1 + 2 * o2)).abs().max().item() <= 2 * ATOL


@pytest.mark.parametrize("seed", range(12))
def test_randomised_shapes_all_algorithms(mmt_lib, oracle_mod, seed):
    """Random (B, P, C, grid) incl. the fallback paths: C % 4 != 0, C > 256, nz > 1, P < chunk,
    P straddling chunk and batch boundaries; every forward algorithm + both backward paths."""
    from mm_training_amd.ops.voxel_pooling import voxel_pooling_ext
    rng = np.random.default_rng(100 + seed)
    B = int(rng.integers(1, 4))
    P = int(rng.choice([1, 63, 511, 512, 513, 1000, 2049, 5000]))
    C = int(rng.choice([1, 4, 12, 20, 64, 80, 96, 256, 260, 512]))
    nx, ny, nz = int(rng.integers(1, 40)), int(rng.integers(1, 40)), int(rng.integers(1, 3))
    geom = np.stack([rng.integers(-2, nx + 2, (B, P)), rng.integers(-2, ny + 2, (B, P)),
                     rng.integers(-1, nz + 1, (B, P))], -1).astype(np.int32)
    if seed % 3 == 0:   # hot cell
        geom[:, : P // 2] = [min(1, nx - 1), min(2, ny - 1), 0]
    feats = (rng.random((B, P, C), dtype=np.float32) - 0.5)
    ref, ref_pos = oracle_mod.voxel_pooling_forward(geom, feats, nx, ny, nz)
    ref64 = oracle_mod.voxel_pooling_forward_f64(geom, feats, nx, ny, nz)
    g, f = _dev(geom), _dev(feats)
    # 3 | (n/4 << 8): SEG_GATHER with n points per workgroup (MMT_VP_CHUNK_POINTS: 64, 232, 512)
    for algo in (0, 1, 2, 3, 4, 0x23, 3 | (16 << 8), 3 | (58 << 8), 3 | (128 << 8)):
        out, pos = _run_ext(mmt_lib, g, f, nx, ny, nz, algo)
        assert torch.equal(pos.cpu(), torch.from_numpy(ref_pos)), (algo, "pos_memo")
        assert np.abs(out.cpu().numpy() - ref64).max() <= ATOL, algo
    go = rng.standard_normal((B, C, ny, nx)).astype(np.float32)
    ref_gi = oracle_mod.voxel_pooling_backward(ref_pos, go)
    pos_d = _dev(ref_pos)
    full = voxel_pooling_ext.backward_workspace_elems(B, P, C, nx, ny)
    for grad in (_dev(go), _dev(go).contiguous(memory_format=torch.channels_last)):
        for ws in (None, torch.empty(full, device="cuda")):
            gi = torch.empty(B, P, C, device="cuda")
            voxel_pooling_ext.voxel_pooling_backward_wrapper(B, P, C, nx, ny, pos_d, grad, gi, ws)
            assert np.array_equal(gi.cpu().numpy(), ref_gi)


@pytest.mark.parametrize("algo", [0, 3, 4, 0x23, 3 | (16 << 8), 3 | (58 << 8)])
def test_full_chunks_of_distinct_cells(mmt_lib, oracle_mod, algo):
    """Regression (found by tests/soak/fuzz_pooling.py): a chunk whose points are ALL kept and ALL in
    different cells fills every slot of the chunk-local tables (ns == chunk size)."""
    B, P, C = 1, 4096, 256
    nx, ny, nz = 64, 64, 1
    idx = np.arange(P)
    geom = np.stack([idx % nx, idx // nx, np.zeros(P, np.int64)], -1).astype(np.int32)[None]
    feats = (np.random.default_rng(0).random((B, P, C), dtype=np.float32) - 0.5)
    ref, ref_pos = oracle_mod.voxel_pooling_forward(geom, feats, nx, ny, nz)
    out, pos = _run_ext(mmt_lib, _dev(geom), _dev(feats), nx, ny, nz, algo)
    assert torch.equal(pos.cpu(), torch.from_numpy(ref_pos))
    assert np.abs(out.cpu().numpy() - ref).max() <= ATOL
