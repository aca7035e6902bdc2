"""GPU: the caller's branches around the op against the reference's OWN LSSFPN._forward_single_sweep / forward
(layers/backbones/lss_fpn.py:381-529), run unmodified on CPU by tests/golden/make_golden.py::make_lss_forward and stored
as lss_forward.npz: depth softmax taken BEFORE the per-camera un-flip (:423-425), the oracle-depth overwrite (:427-438),
lift + geometry + quantise + voxel_pooling (:441-465), older sweeps under no_grad and the channel stacking (:516-529), and
the gradient the reference's autograd sends back.  The mirror (mm_training_amd/layers/backbones/lss_fpn.py) is driven
through the same stand-ins as the reference was (images = neck features, identity depth net) in each of its three camera
paths: camera form (default), geom form, and the unfused lift -> drop-in voxel_pooling sequence."""
import numpy as np
import pytest
import torch

from tests.golden.formula import hashed_f32

pytestmark = pytest.mark.gpu


class _Identity(torch.nn.Module):
    def forward(self, x, mats_dict=None):
        return x


def _mirror(g):
    from mm_training_amd.layers.backbones import LSSFPN
    b = g["bounds"]
    conf = dict(x_bound=b[0].tolist(), y_bound=b[1].tolist(), z_bound=b[2].tolist(), d_bound=g["d_bound"].tolist(),
                final_dim=tuple(int(v) for v in g["final_dim"]), downsample_factor=int(g["ds"]), output_channels=int(g["channels"]),
                img_backbone_conf=dict(type='ResNet', depth=18, base_channels=8, out_indices=[0, 1, 2, 3]),
                img_neck_conf=dict(type='SECONDFPN', in_channels=[8, 16, 32, 64], upsample_strides=[0.25, 0.5, 1, 2], out_channels=[8] * 4),
                depth_net_conf=dict(in_channels=32, mid_channels=32))
    m = LSSFPN(**conf).cuda().train()
    m.get_cam_feats = lambda imgs: imgs            # the fixture's "images" are the neck features (as for the reference run)
    m.depth_net = _Identity()
    return m


@pytest.mark.parametrize("path", ["plan", "camera", "geom", "unfused"])
@pytest.mark.parametrize("case", ["flip", "oracle", "single"])
def test_lssfpn_branches_match_the_reference_forward(mmt_lib, golden, case, path):
    from mm_training_amd import _lib
    g = golden["lss_forward"]
    m = _mirror(g)
    assert m.fused_lift_splat and m.camera_form and m.plan_form         # the default: the camera form's forward in its plan form
    if path == "camera":
        m.plan_form = False
    elif path == "geom":
        m.camera_form = False
    elif path == "unfused":
        m.fused_lift_splat = False
    imgs = torch.from_numpy(g[case + "_imgs"]).cuda().requires_grad_(True)
    sweeps = imgs.shape[1]
    B = imgs.shape[0]
    mats = dict(sensor2ego_mats=torch.from_numpy(g["sensor2ego"][:, :sweeps]).cuda(), intrin_mats=torch.from_numpy(g["intrin"][:, :sweeps]).cuda(),
                bda_mat=torch.eye(4).repeat(B, 1, 1).cuda(), flipped=torch.from_numpy(g[case + "_flipped"]))
    assert bool(g[case + "_flipped"].any()) and not bool(g[case + "_flipped"].all())
    oracle_depth = torch.from_numpy(g[case + "_depth_oracle"]).cuda() if (case + "_depth_oracle") in g.files else None
    calls = []
    real = _lib.call
    _lib.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
    try:
        bev, depth = m(imgs, mats, oracle_depth, None, is_return_depth=True)
    finally:
        _lib.call = real
    want = {"plan": "mmt_lss_splat_forward_plan", "camera": "mmt_lss_splat_forward_cam", "geom": "mmt_lss_splat_forward", "unfused": "mmt_voxel_pooling_forward_ex"}[path]
    assert want in calls, calls
    ref_bev, ref_depth = g[case + "_bev"], g[case + "_depth"]
    assert tuple(bev.shape) == ref_bev.shape and tuple(depth.shape) == ref_depth.shape
    assert np.abs(depth.detach().cpu().numpy() - ref_depth).max() <= 1e-6          # the key frame's softmax, BEFORE flip / oracle
    scale = max(1.0, float(np.abs(ref_bev).max()))
    assert np.abs(bev.detach().cpu().numpy() - ref_bev).max() <= 2e-5 * scale
    # the same cells are occupied (the integer path is exact)
    assert np.array_equal(bev.detach().abs().sum(1).cpu().numpy() > 0, np.abs(ref_bev).sum(1) > 0)
    go = torch.from_numpy(hashed_f32(ref_bev.shape, salt=int(g[case + "_grad_out_salt"]))).cuda()
    bev.backward(go)
    ref_g = g[case + "_grad_imgs"]
    gscale = max(1.0, float(np.abs(ref_g).max()))
    assert np.abs(imgs.grad.cpu().numpy() - ref_g).max() <= 5e-5 * gscale
    if sweeps > 1:
        assert float(imgs.grad[:, 1:].abs().max()) == 0.0                             # older sweeps ran under no_grad (:516-524)


def test_plan_lookup_is_skipped_while_the_calibration_ids_repeat(mmt_lib, golden):
    """SURVEY 8 row f3 (lss_fpn.py:328-361: no per-step term for an unchanged calibration): with mats_dict['calibration_id'] the
    module runs the lookup (mmt_lss_plan_prepare, or riding in the softmax: mmt_depth_softmax_forward_plan_prepare) once; while the ids repeat, NO lookup kernel is launched and
    the forward goes by the verdicts left in the id's own cache -- bit-identical output, also when two rigs alternate.  A new id,
    a batch without ids, an id whose cache was dropped (more than `plan_named_caches` ids) or a frustum change runs the lookup
    (and the result still matches the reference's forward)."""
    from mm_training_amd import _lib
    g = golden["lss_forward"]
    m = _mirror(g)
    case = "single"
    imgs = torch.from_numpy(g[case + "_imgs"]).cuda()
    sweeps, B = imgs.shape[1], imgs.shape[0]
    oracle_depth = torch.from_numpy(g[case + "_depth_oracle"]).cuda() if (case + "_depth_oracle") in g.files else None

    def mats(cid=None, shift=0.0):
        d = dict(sensor2ego_mats=torch.from_numpy(g["sensor2ego"][:, :sweeps]).cuda().clone(), intrin_mats=torch.from_numpy(g["intrin"][:, :sweeps]).cuda(),
                 bda_mat=torch.eye(4).repeat(B, 1, 1).cuda(), flipped=torch.from_numpy(g[case + "_flipped"]))
        d["sensor2ego_mats"][..., 0, 3] += shift
        if cid is not None:
            d["calibration_id"] = cid
        return d

    def run(md):
        calls = []
        real = _lib.call
        _lib.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
        try:
            with torch.no_grad():
                out = m(imgs, md, oracle_depth, None)
        finally:
            _lib.call = real
        assert "mmt_lss_splat_forward_plan" in calls, calls
        # (the lookup rides in the depth softmax's launch when there is one to make: mmt_depth_softmax_forward_plan_prepare)
        return out, calls.count("mmt_lss_plan_prepare") + calls.count("mmt_depth_softmax_forward_plan_prepare")

    ref = g[case + "_bev"]
    scale = max(1.0, float(np.abs(ref).max()))
    a, n = run(mats("rig-A"))
    assert n == 1 and np.abs(a.cpu().numpy() - ref).max() <= 2e-5 * scale
    for _ in range(3):
        b_, n = run(mats("rig-A"))
        assert n == 0 and torch.equal(a, b_)                     # steady state: zero lookup launches
    c, n = run(mats("rig-B", shift=0.37))                       # another rig: looked up (and learnt)
    assert n == 1 and not torch.equal(a, c)
    d, n = run(mats("rig-B", shift=0.37))
    assert n == 0 and torch.equal(c, d)
    for _ in range(2):                                          # two rigs alternating: each id has its own cache
        e, n = run(mats("rig-A"))
        assert n == 0 and torch.equal(a, e)
        e, n = run(mats("rig-B", shift=0.37))
        assert n == 0 and torch.equal(c, e)
    f, n = run(mats(None))                                      # no ids: always looked up (a cache of its own)
    assert n == 1 and torch.equal(a, f)
    f, n = run(mats(None))
    assert n == 1 and torch.equal(a, f)
    h, n = run(mats("rig-A"))                                   # (the anonymous batches did not touch the named caches)
    assert n == 0 and torch.equal(a, h)
    m.plan_named_caches = 2
    k, n = run(mats("rig-C", shift=-0.21))                      # a third id: the least recently used named cache (rig-B's) goes
    assert n == 1
    h, n = run(mats("rig-A"))
    assert n == 0 and torch.equal(a, h)
    e, n = run(mats("rig-B", shift=0.37))                       # rig-B again: a fresh cache, looked up and learnt again
    assert n == 1 and torch.equal(c, e)
    m._refresh_frustum_axes()                                    # a (re-)loaded frustum: what is in the cache belongs to the old axes
    i, n = run(mats("rig-A"))
    assert n == 1 and torch.equal(a, i)
