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


def _sorted_rows(anno, ind, mask):
    """Slot-order-invariant view of one sample's (anno, ind) rows of valid slots."""
    m = mask.astype(bool)
    rows = np.concatenate([ind[m, None].astype(np.float64), anno[m].astype(np.float64)], 1)
    return rows[np.lexsort(rows.T[::-1])]


@pytest.mark.parametrize("seed", range(6))
def test_centerpoint_targets_against_oracle(mmt_lib, oracle_mod, seed):
    """Heat-maps (1e-6: fp32 expf vs the reference's float64 numpy window), centre indices, masks and
    regression rows against the sequential restatement of bev_depth_head.py:113-254 -- compared
    order-invariantly because the reference packs each task's boxes densely, class-major."""
    from mm_training_amd.ops.train_targets import centerpoint_targets
    rng = np.random.default_rng(900 + seed)
    B = int(rng.integers(1, 5))
    class_counts = [[1, 1, 1, 1], [2, 1, 3], [4]][seed % 3]
    n_cls = sum(class_counts)
    fx, fy, osf = [(128, 128, 4), (512, 64, 4), (40, 56, 2)][seed % 3]
    vs = (0.2, 0.2, 8.0)
    pc = (-vs[0] * osf * fx / 2, -vs[1] * osf * fy / 2, -5.0, vs[0] * osf * fx / 2, vs[1] * osf * fy / 2, 3.0)
    max_objs = [500, 12, 40][seed % 3]
    boxes, labels = [], []
    for b in range(B):
        k = int(rng.choice([0, 1, 7, 30]))
        xy = rng.uniform([pc[0] - 3, pc[1] - 3], [pc[3] + 3, pc[4] + 3], (k, 2))          # some centres outside the map
        dims = rng.uniform(0.3, 12.0, (k, 3))
        if k > 3:
            dims[0, 0] = 0.0                       # zero width: skipped (:183)
            xy[1] = [pc[0] + 0.01, pc[1] + 0.01]   # window clipped at the map corner
        bx = np.concatenate([xy, rng.uniform(-2, 1, (k, 1)), dims, rng.uniform(-3.2, 3.2, (k, 1)), rng.normal(size=(k, 2))], 1)
        boxes.append(bx.astype(np.float32))
        labels.append(rng.integers(0, n_cls, k).astype(np.int64))
    if sum(len(b) for b in boxes) == 0:
        boxes[0] = np.array([[0.5, 0.5, 0, 2, 4, 1.5, 0.3, 1, 0]], np.float32)
        labels[0] = np.array([0])
    hm, anno, ind, mask = centerpoint_targets([torch.from_numpy(b).cuda() for b in boxes], [torch.from_numpy(l).cuda() for l in labels],
                                              class_counts, max_objs, (fx, fy), pc, vs, osf, 0.1, 2)
    begin = 0
    for t, n in enumerate(class_counts):
        assert hm[t].shape == (B, n, fy, fx) and anno[t].shape == (B, max_objs, 10)
        for b in range(B):
            r_hm, r_anno, r_ind, r_mask = oracle_mod.centerpoint_targets_task(boxes[b], labels[b], begin, n,
                                                                              max_objs, fx, fy, pc, vs, osf, 0.1, 2)
            assert np.abs(hm[t][b].cpu().numpy() - r_hm).max() <= 1e-6
            assert np.array_equal(hm[t][b].cpu().numpy() == 1.0, r_hm == 1.0)            # the loss's positive set
            got = _sorted_rows(anno[t][b].cpu().numpy(), ind[t][b].cpu().numpy(), mask[t][b].cpu().numpy())
            ref = _sorted_rows(r_anno, r_ind, r_mask)
            assert got.shape == ref.shape
            assert np.allclose(got, ref, rtol=0, atol=2e-6)
            # invalid slots are zero
            m = mask[t][b].cpu().numpy().astype(bool)
            assert not anno[t][b].cpu().numpy()[~m].any() and not ind[t][b].cpu().numpy()[~m].any()
        begin += n


def test_centerpoint_targets_match_torch_cross_check(mmt_lib):
    """HIP targets == the vectorised torch restatement slot by slot (same slot convention)."""
    from mm_training_amd.dp import make_config, synthetic_batch
    from mm_training_amd.layers.heads.bev_depth_head import BEVDepthHead
    cfg = make_config("tiny")
    dev = torch.device("cuda", 0)
    head = BEVDepthHead(**cfg["head_conf"]).to(dev)
    _, _, _, boxes, labels = synthetic_batch(cfg, dev, seed=11)
    a = head.get_targets(boxes, labels)
    b = head.get_targets_torch(boxes, labels)
    for field in range(4):
        for t in range(len(a[field])):
            x, y = a[field][t], b[field][t]
            assert x.shape == y.shape and x.dtype == y.dtype, (field, t, x.shape, y.shape, x.dtype, y.dtype)
            # torch's GPU `tensor / python_scalar` multiplies by the rounded reciprocal, the kernel divides:
            # the centre offsets (values up to fx) may differ in the last bit
            assert torch.allclose(x.float(), y.float(), rtol=0, atol=2e-5 if field == 1 else 2e-6), (field, t)
