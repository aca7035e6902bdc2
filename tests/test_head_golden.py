"""The CenterPoint head's targets and loss, the depth loss and the fusion layer against fixtures produced by the REFERENCE'S OWN
code (tests/golden/make_golden.py execs layers/heads/bev_depth_head.py:113-254 and :256-312, exps/mm_training_aim.py:165-178 and
imports models/bev_depth.py:133-145; the un-vendored mmdet / mmdet3d callables they use enter by their published formulas):
SURVEY section 8 row f4 (targets) and the step's remaining pure-torch reference functions (row a13).

CPU: the oracle's restatement and the head's vectorised torch restatement.  GPU (-m gpu): the HIP target kernel, BEVDepthHead.loss,
TrainStep.get_depth_loss and BEVFuseLayer."""
import numpy as np
import pytest
import torch

from tests.golden.formula import hashed_f32

CASES = {"aim": [['car'], ['truck/bus'], ['motorcycle'], ['pedestrian']], "multi": [['a', 'b'], ['c'], ['d', 'e', 'f']], "small": [['a', 'b', 'c', 'd']]}
VS = (0.2, 0.2, 8.0)


def _sorted_rows(anno, ind, mask):
    """Slot-order-invariant view of one sample's (anno, ind) rows of valid slots (the reference packs a task's boxes densely,
    class-major; the implementations keep a box at its own index: the loss sums over masked slots, whatever their order)."""
    m = np.asarray(mask).astype(bool)
    rows = np.concatenate([np.asarray(ind)[m, None].astype(np.float64), np.asarray(anno)[m].astype(np.float64)], 1)
    return rows[np.lexsort(rows.T[::-1])]


def _case(g, name):
    fx, fy, osf, max_objs = [int(v) for v in g[name + "_cfg"]]
    return fx, fy, osf, max_objs, [float(v) for v in g[name + "_pc_range"]], [int(v) for v in g[name + "_class_counts"]], int(g[name + "_n_samples"])


def _check_task(g, name, b, t, hm, anno, ind, mask):
    r_hm, r_anno, r_ind, r_mask = (g[f"{name}_{f}_{b}_{t}"] for f in ("hm", "anno", "ind", "mask"))
    assert hm.shape == r_hm.shape
    assert np.abs(hm - r_hm).max() <= 1e-6                                     # (fp32 expf against the reference's float64 window)
    assert np.array_equal(hm == 1.0, r_hm == 1.0)                              # the loss's positive set
    got, ref = _sorted_rows(anno, ind, mask), _sorted_rows(r_anno, r_ind, r_mask)
    assert got.shape == ref.shape, (name, b, t, got.shape, ref.shape)
    assert np.array_equal(got[:, 0], ref[:, 0])                                # centre indices: exact
    assert np.allclose(got, ref, rtol=0, atol=2e-6)
    m = np.asarray(mask).astype(bool)
    assert int(m.sum()) == int(r_mask.sum())
    assert not np.asarray(anno)[~m].any() and not np.asarray(ind)[~m].any()    # invalid slots are zero


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_centerpoint_targets_match_the_reference(oracle_mod, golden, name):
    g = golden["centerpoint_targets"]
    fx, fy, osf, max_objs, pc, counts, B = _case(g, name)
    for b in range(B):
        boxes, labels = g[f"{name}_boxes_{b}"], g[f"{name}_labels_{b}"]
        begin = 0
        for t, n in enumerate(counts):
            hm, anno, ind, mask = oracle_mod.centerpoint_targets_task(boxes, labels, begin, n, max_objs, fx, fy, pc, VS, osf, 0.1, 2)
            _check_task(g, name, b, t, hm, anno, ind, mask)
            begin += n


def _head(name, g, device):
    from mm_training_amd.layers.heads.bev_depth_head import BEVDepthHead
    fx, fy, osf, max_objs, pc, counts, B = _case(g, name)
    train_cfg = dict(point_cloud_range=pc, grid_size=[fx * osf, fy * osf, 1], voxel_size=list(VS), out_size_factor=osf, dense_reg=1,
                     gaussian_overlap=0.1, max_objs=max_objs, min_radius=2, code_weights=[1.0] * 8 + [0.3, 0.3])
    tasks = [dict(num_class=len(c), class_names=c) for c in CASES[name]]
    head = BEVDepthHead(in_channels=16, tasks=tasks, common_heads=dict(reg=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
                        train_cfg=train_cfg, bev_backbone_conf=dict(type='ResNet', in_channels=16, depth=18, num_stages=3, strides=(1, 2, 2),
                                                                     dilations=(1, 1, 1), out_indices=[0, 1, 2], base_channels=16),
                        bev_neck_conf=dict(type='SECONDFPN', in_channels=[16, 32, 64], upsample_strides=[1, 2, 4], out_channels=[16, 16, 16]),
                        loss_cls=dict(type='GaussianFocalLoss', reduction='mean'), loss_bbox=dict(type='L1Loss', reduction='mean', loss_weight=0.25))
    return head.to(device), B, counts


@pytest.mark.parametrize("name", list(CASES))
def test_head_torch_targets_match_the_reference(golden, name):
    """BEVDepthHead.get_targets_single (the vectorised torch restatement; CPU tensors here)"""
    g = golden["centerpoint_targets"]
    head, B, counts = _head(name, g, "cpu")
    for b in range(B):
        hm, anno, ind, mask = head.get_targets_single(torch.from_numpy(g[f"{name}_boxes_{b}"]), torch.from_numpy(g[f"{name}_labels_{b}"]))
        for t in range(len(counts)):
            _check_task(g, name, b, t, hm[t].numpy(), anno[t].numpy(), ind[t].numpy(), mask[t].numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_hip_centerpoint_targets_match_the_reference(mmt_lib, golden, name):
    """cp_draw_kernel (ops/train_targets.py::centerpoint_targets) and BEVDepthHead.get_targets on top of it"""
    g = golden["centerpoint_targets"]
    head, B, counts = _head(name, g, "cuda")
    boxes = [torch.from_numpy(g[f"{name}_boxes_{b}"]).cuda() for b in range(B)]
    labels = [torch.from_numpy(g[f"{name}_labels_{b}"]).cuda() for b in range(B)]
    hm, anno, ind, mask = head.get_targets(boxes, labels)
    for t in range(len(counts)):
        assert hm[t].shape[0] == B
        for b in range(B):
            _check_task(g, name, b, t, hm[t][b].cpu().numpy(), anno[t][b].cpu().numpy(), ind[t][b].cpu().numpy(), mask[t][b].cpu().numpy())


def _loss_inputs(g, device):
    """the 'aim' targets stacked over their samples + the formula predictions of head_loss.npz"""
    t_ = g["centerpoint_targets"]
    fx, fy, osf, max_objs, pc, counts, B = _case(t_, "aim")
    stack = lambda f: [torch.from_numpy(np.stack([t_[f"aim_{f}_{b}_{t}"] for b in range(B)])).to(device) for t in range(len(counts))]
    targets = (stack("hm"), stack("anno"), stack("ind"), stack("mask"))
    salt = int(g["head_loss"]["first_salt"]) - 1
    preds = []
    for t, n in enumerate(counts):
        d = {}
        for k, c in [("heatmap", n), ("reg", 2), ("height", 1), ("dim", 3), ("rot", 2), ("vel", 2)]:
            salt += 1
            d[k] = (torch.from_numpy(4.0 * hashed_f32((B, c, fy, fx), salt=salt)) - (2.0 if k == "heatmap" else 0.0)).to(device)
        preds.append([d])
    return targets, preds


def test_head_loss_matches_the_reference_on_cpu(golden):
    head, _, _ = _head("aim", golden["centerpoint_targets"], "cpu")
    targets, preds = _loss_inputs(golden, "cpu")
    want = float(golden["head_loss"]["loss"])
    got = float(head.loss(targets, preds))
    assert abs(got - want) <= 1e-5 * abs(want), (got, want)


@pytest.mark.gpu
def test_head_loss_matches_the_reference(mmt_lib, golden):
    head, _, _ = _head("aim", golden["centerpoint_targets"], "cuda")
    targets, preds = _loss_inputs(golden, "cuda")
    want = float(golden["head_loss"]["loss"])
    got = float(head.loss(targets, preds))
    assert abs(got - want) <= 2e-5 * abs(want), (got, want)
    # and on the targets the HIP kernel makes from the same boxes (slot order differs from the reference's, the loss does not)
    t_ = golden["centerpoint_targets"]
    B = int(t_["aim_n_samples"])
    hip = head.get_targets([torch.from_numpy(t_[f"aim_boxes_{b}"]).cuda() for b in range(B)], [torch.from_numpy(t_[f"aim_labels_{b}"]).cuda() for b in range(B)])
    got2 = float(head.loss(hip, preds))
    assert abs(got2 - want) <= 2e-5 * abs(want), (got2, want)


def _depth_loss(device):
    from mm_training_amd.dp.trainer import TrainStep
    import types
    return types.MethodType(TrainStep.get_depth_loss, types.SimpleNamespace(depth_channels=20))


def test_depth_loss_matches_the_reference_on_cpu(golden):
    g = golden["depth_loss"]
    f = _depth_loss("cpu")
    got = float(f(torch.from_numpy(g["labels"]), torch.from_numpy(g["preds"])))
    assert abs(got - float(g["loss"])) <= 1e-5 * float(g["loss"])
    assert float(f(torch.zeros_like(torch.from_numpy(g["labels"])), torch.from_numpy(g["preds"]))) == float(g["loss_without_labels"]) == 0.0


@pytest.mark.gpu
def test_depth_loss_matches_the_reference(mmt_lib, golden):
    g = golden["depth_loss"]
    f = _depth_loss("cuda")
    got = float(f(torch.from_numpy(g["labels"]).cuda(), torch.from_numpy(g["preds"]).cuda()))
    assert abs(got - float(g["loss"])) <= 2e-5 * float(g["loss"])
    assert float(f(torch.zeros_like(torch.from_numpy(g["labels"])).cuda(), torch.from_numpy(g["preds"]).cuda())) == 0.0


def _fuse(g, device):
    from mm_training_amd.models.bev_depth import BEVFuseLayer
    m = BEVFuseLayer(int(g["x"].shape[1]))
    m.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w_")})      # the reference's parameter names
    return m.to(device)


def test_fuse_layer_matches_the_reference_on_cpu(golden):
    g = golden["fuse_layer"]
    m = _fuse(g, "cpu")
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = m(x)
    assert np.abs(y.detach().numpy() - g["y"]).max() <= 1e-5 * max(1.0, float(np.abs(g["y"]).max()))
    y.backward(torch.from_numpy(hashed_f32(tuple(y.shape), salt=int(g["grad_out_salt"]))))
    assert np.abs(x.grad.numpy() - g["grad_x"]).max() <= 1e-5 * max(1.0, float(np.abs(g["grad_x"]).max()))
    for k, p in m.named_parameters():
        assert np.abs(p.grad.numpy() - g["g_" + k]).max() <= 1e-4 * max(1.0, float(np.abs(g["g_" + k]).max())), k


@pytest.mark.gpu
def test_fuse_layer_matches_the_reference(mmt_lib, golden):
    g = golden["fuse_layer"]
    m = _fuse(g, "cuda")
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    y = m(x)
    assert np.abs(y.detach().cpu().numpy() - g["y"]).max() <= 2e-5 * max(1.0, float(np.abs(g["y"]).max()))
    y.backward(torch.from_numpy(hashed_f32(tuple(y.shape), salt=int(g["grad_out_salt"]))).cuda())
    assert np.abs(x.grad.cpu().numpy() - g["grad_x"]).max() <= 2e-5 * max(1.0, float(np.abs(g["grad_x"]).max()))
    for k, p in m.named_parameters():
        assert np.abs(p.grad.cpu().numpy() - g["g_" + k]).max() <= 2e-4 * max(1.0, float(np.abs(g["g_" + k]).max())), k
