"""GPU parity of the per-step label kernels against the oracle, the golden
vectors produced by the reference's own methods, and the vectorised torch cross-check."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _clouds(points, offsets):
    return [torch.from_numpy(points[offsets[b]:offsets[b + 1]]).cuda() for b in range(len(offsets) - 1)]


def test_depth_labels_reference_golden(mmt_lib, golden):
    """Expected bins come from exps/mm_training_aim.py get_depth_labels / get_depth_image /
    get_downsampled_gt_depth run on CPU (tests/golden/make_golden.py: make_depth_labels)."""
    from mm_training_amd.ops.train_targets import depth_labels
    g = golden["depth_labels"]
    H, W = [int(v) for v in g["img_hw"]]
    d_bound = [float(v) for v in g["d_bound"]]
    D = int((d_bound[1] - d_bound[0]) / d_bound[2])
    onehot, bins = depth_labels(_clouds(g["points"], g["offsets"]), torch.from_numpy(g["extrinsics"]).cuda(),
                                torch.from_numpy(g["intrinsics"]).cuda(), torch.from_numpy(g["bda"]).cuda(),
                                (H, W), int(g["downsample"]), d_bound, D, return_bins=True)
    assert np.array_equal(bins.cpu().numpy(), g["bins"])
    oh = onehot.cpu().numpy()
    assert oh.shape == (g["bins"].size, D)
    assert np.array_equal(oh.argmax(1), g["bins"]) and np.array_equal(oh.sum(1), np.ones(len(oh), np.float32))


@pytest.mark.parametrize("seed", range(6))
def test_depth_labels_random_against_oracle(mmt_lib, oracle_mod, seed):
    """Bit-exact against the C oracle (same fp32 operation order), incl. empty clouds, points behind
    the camera, NaN / inf coordinates, several points per pixel, a scaled + rotated BDA matrix."""
    from mm_training_amd import synthetic
    from mm_training_amd.ops.train_targets import depth_labels
    rng = np.random.default_rng(40 + seed)
    B, N = int(rng.integers(1, 4)), int(rng.integers(1, 7))
    H, W, ds = [(64, 96, 16), (256, 704, 16), (128, 352, 8)][seed % 3]
    d_bound = [(2.0, 58.0, 0.5), (1.0, 60.0, 0.5)][seed % 2]
    D = int((d_bound[1] - d_bound[0]) / d_bound[2])
    s2e, K = synthetic.camera_rig(B, N, W, H, jitter=0.05, seed=seed)
    extr = torch.inverse(s2e)
    bda = torch.eye(4).repeat(B, 1, 1)
    for b in range(B):
        a = float(rng.uniform(-0.4, 0.4))
        bda[b, :2, :2] = torch.tensor([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]], dtype=torch.float32) * float(rng.uniform(0.9, 1.1))
    clouds = []
    for b in range(B):
        n = int(rng.choice([0, 1, 500, 40000]))
        pc = np.concatenate([rng.uniform(-60, 60, (n, 2)), rng.uniform(-4, 4, (n, 1)), rng.uniform(0, 1, (n, 2))], 1).astype(np.float32)
        if n > 10:
            pc[0, 0] = np.nan
            pc[1, 1] = np.inf
            pc[2, :3] = 0.0
            pc[3:8, :3] = pc[8, :3]            # several points on one pixel
        clouds.append(pc)
    if sum(len(c) for c in clouds) == 0:
        clouds[0] = np.array([[10.0, 0.5, 0.0, 0.1, 0.2]], np.float32)
    ref_bins, ref_onehot = oracle_mod.depth_labels(clouds, extr.numpy(), K.numpy(), bda.numpy(), (H, W), ds, d_bound)
    onehot, bins = depth_labels([torch.from_numpy(c).cuda() for c in clouds], extr.cuda(), K.cuda(), bda.cuda(),
                                (H, W), ds, d_bound, D, return_bins=True)
    # the op inverts the BDA rotation with torch (fp32 LU), the oracle front-end in float64: feed the
    # oracle the matrix the op used when they differ in the last bit
    bda_inv = torch.linalg.inv_ex(bda[:, :3, :3].cuda())[0].cpu().numpy()
    if not np.array_equal(bda_inv, np.linalg.inv(bda[:, :3, :3].double().numpy()).astype(np.float32)):
        import ctypes
        fH, fW = H // ds, W // ds
        ref_bins = np.empty(B * N * fH * fW, np.int32)
        ref_onehot = np.empty((B * N * fH * fW, D), np.float32)
        pts = np.ascontiguousarray(np.concatenate(clouds, 0))
        offs = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).astype(np.int32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        e_, k_ = np.ascontiguousarray(extr.numpy()), np.ascontiguousarray(K.numpy())
        oracle_mod.lib().oracle_depth_labels(B, N, pts.shape[1], H, W, ds, ctypes.c_float(d_bound[0]), ctypes.c_float(d_bound[2]),
                                             D, p(pts), p(offs), p(e_), p(k_), p(np.ascontiguousarray(bda_inv)), 0,
                                             p(ref_bins), p(ref_onehot))
    assert np.array_equal(bins.cpu().numpy(), ref_bins)
    assert np.array_equal(onehot.cpu().numpy(), ref_onehot)


def test_depth_labels_match_torch_cross_check(mmt_lib):
    """The training step's HIP labels vs the vectorised torch restatement (different fp32 summation
    order inside einsum: a point on a pixel / bin boundary may flip, so a handful of cells may differ)."""
    from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
    cfg = make_config("tiny")
    dev = torch.device("cuda", 0)
    ts = TrainStep(cfg, dev)
    imgs, mats, pcs, _, _ = synthetic_batch(cfg, dev, seed=5)
    a = ts.get_depth_labels(imgs, mats, pcs)
    b = ts.get_depth_labels_torch(imgs, mats, pcs)
    assert a.shape == b.shape and a.dtype == b.dtype
    mismatch = (a.argmax(1) != b.argmax(1)).float().mean().item()
    assert mismatch <= 2e-3, mismatch
    assert (a.argmax(1) > 0).float().mean().item() > 0.05      # the synthetic cloud does label cells


def test_depth_labels_errors(mmt_lib):
    from mm_training_amd.ops.train_targets import depth_labels
    eye = torch.eye(4).repeat(1, 1, 1, 1).cuda()
    pc = [torch.zeros(4, 5).cuda()]
    with pytest.raises(RuntimeError, match="CUDA"):
        depth_labels([torch.zeros(4, 5)], eye, eye, eye[0], (64, 96), 16, (2.0, 58.0, 0.5), 112)
    with pytest.raises(RuntimeError):
        depth_labels(pc, eye, eye, eye[0], (60, 96), 16, (2.0, 58.0, 0.5), 112)     # H % downsample != 0
