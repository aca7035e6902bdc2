#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE'S OWN PYTHON.

Runs only in the build container (needs /root/reference; never on the GPU box and
never from the test-suite).  It imports the reference's modules with the absent
third-party packages (mmcv, mmdet, mmdet3d, kornia) and the absent native extension
(ops.voxel_pooling.voxel_pooling_ext) stubbed in ``sys.modules``, executes the
reference's code on CPU tensors and stores inputs + expected outputs as .npz data.
No reference source text is stored -- only arrays.

What is executed from the reference:
  * test/test_ops/test_voxel_pooling.py:15-30  -- the known-answer test's input
    construction and its sequential ground-truth loop (source lines are read from
    the file at run time and exec'd; they are not copied into this repo).
  * ops/voxel_pooling/voxel_pooling.py VoxelPooling.forward/.backward -- run through
    torch.autograd with the extension call replaced by the C oracle (forward values
    are therefore cross-checked against the test loop above; backward is 100 %
    reference code: boolean-mask + advanced-index gather).
  * layers/backbones/lss_fpn.py LSSFPN.__init__ buffers (:278-289), create_frustum
    (:308-326), get_geometry (:328-361) and the quantise expression (:461-462,
    evaluated from the module's source text at run time).
  * layers/backbones/lss_fpn.py LSSFPN._forward_single_sweep / forward (:381-529) as a whole -- flipped cameras,
    oracle depth, two sweeps -- with the conv nets replaced by the identity (lss_forward.npz).
  * exps/mm_training_aim.py augment_images (:88-112) and normalize_images (:510-512), exec'd as plain functions with a
    seeded numpy generator (augment_images.npz).
  * layers/heads/bev_depth_head.py get_targets_single (:113-254) and loss (:256-312), exps/mm_training_aim.py get_depth_loss
    (:165-178), exec'd from the files' text on CPU tensors with the un-vendored mmdet / mmdet3d callables they use supplied by
    their published formulas; models/bev_depth.py BEVFuseLayer (:133-145) imported as it is (centerpoint_targets.npz,
    head_loss.npz, depth_loss.npz, fuse_layer.npz).

Usage:  python tests/golden/make_golden.py
"""
import importlib
import importlib.abc
import importlib.machinery
import inspect
import os
import re
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

import oracle  # noqa: E402  (the C restatement being pinned)
from tests.golden.formula import hashed_f32  # noqa: E402


# --------------------------------------------------------------- stub machinery
class _Dummy:
    """A class usable as base class / callable / attribute bag."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()

    def init_weights(self):
        pass


class _StubModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        obj = type(name, (_Dummy,), {})
        setattr(self, name, obj)
        return obj


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("mmcv", "mmdet", "mmdet3d", "kornia", "pytorch_lightning", "wandb", "cv2")

    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


def _install_stubs():
    sys.meta_path.insert(0, _StubFinder())
    ext = types.ModuleType("ops.voxel_pooling.voxel_pooling_ext")

    def voxel_pooling_forward_wrapper(B, P, C, nx, ny, nz, geom, feats, out, pos_memo):
        # stand-in for the missing CUDA extension: the C oracle, in place
        o, pm = oracle.voxel_pooling_forward(geom.numpy(), feats.numpy(), int(nx), int(ny),
                                             int(nz), out=out.numpy(), pos_memo=pos_memo.numpy())
        assert o.ctypes.data == out.numpy().ctypes.data
        return 1

    ext.voxel_pooling_forward_wrapper = voxel_pooling_forward_wrapper
    sys.modules["ops.voxel_pooling.voxel_pooling_ext"] = ext
    sys.path.insert(0, REF)
    # `ops`, `layers`, `models` must be the REFERENCE's packages: this repository's root-level alias packages of the same
    # names (regular packages) would otherwise win over the reference's `ops/` (a namespace package: no __init__.py)
    for name in ("ops", "layers", "models"):
        assert name not in sys.modules, name
        pkg = types.ModuleType(name)
        pkg.__path__ = [os.path.join(REF, name)]
        sys.modules[name] = pkg


def _ref_source_lines(relpath, lo, hi):
    with open(os.path.join(REF, relpath)) as f:
        lines = f.readlines()
    return textwrap.dedent("".join(lines[lo - 1:hi]))


def _bounds_expression():
    """The drop test of the reference's known-answer loop (test/test_ops/test_voxel_pooling.py:28, the `if` of the loop at
    :23-30), read from the file at run time and turned into an elementwise numpy predicate over (x, y, z, nx, ny, nz)."""
    cands = [l.strip() for l in _ref_source_lines("test/test_ops/test_voxel_pooling.py", 23, 30).splitlines() if l.strip().startswith("if ")]
    assert len(cands) == 1, cands
    mm = re.match(r"if (.+):$", cands[0])
    assert mm, cands[0]
    terms = [t.strip() for t in mm.group(1).split(" or ")]
    assert len(terms) == 6, terms
    expr = " | ".join("(" + t + ")" for t in terms)
    expr = re.sub(r"\b128\b", "N_X_OR_Y", expr)       # the test's grid is 128 x 128 x 1: generalise the literals per coordinate
    parts = expr.split(" | ")
    out = []
    for t in parts:
        coord = t[1]
        t = t.replace("N_X_OR_Y", {"x": "nx", "y": "ny"}.get(coord, "nz"))
        if coord == "z":
            t = re.sub(r">= 1\)", ">= nz)", t)
        out.append(t)
    return " | ".join(out)


def independent_pos_memo(geom, nx, ny, nz):
    """pos_memo WITHOUT the oracle: the reference test's own drop predicate on numpy arrays + the (b, y, x) row layout of
    voxel_pooling_forward_cuda.cu:27-29; dropped rows keep the caller's -1 (voxel_pooling.py:40)."""
    x, y, z = (geom[..., i].astype(np.int64) for i in range(3))
    drop = eval(_bounds_expression(), {"x": x, "y": y, "z": z, "nx": nx, "ny": ny, "nz": nz})
    pos = np.full(geom.shape, -1, np.int32)
    b = np.broadcast_to(np.arange(geom.shape[0]).reshape(-1, *([1] * (geom.ndim - 2))), x.shape)
    pos[..., 0] = np.where(drop, -1, b)
    pos[..., 1] = np.where(drop, -1, y)
    pos[..., 2] = np.where(drop, -1, x)
    return pos


def check_pos_memo_against_reference_backward(pos, grad_out_nchw, grad_in):
    """The reference's OWN backward (voxel_pooling.py:58-69, pure ATen) consumed the pos_memo its forward stub wrote; its
    grad_in must be exactly grad_out[b, :, y, x] on the rows the independent pos_memo keeps and 0 elsewhere."""
    B, P, C = grad_in.shape
    kept = pos[..., 0] != -1
    want = np.zeros_like(grad_in)
    bb, yy, xx = pos[..., 0][kept], pos[..., 1][kept], pos[..., 2][kept]
    want[kept] = grad_out_nchw[bb, :, yy, xx]
    assert np.array_equal(want, grad_in)


# -------------------------------------------------------------------- fixtures
def make_vp_ref_test():
    """The reference's known-answer test at its own shape + reference backward."""
    ns = {"torch": torch}
    exec("import numpy as np\n" + _ref_source_lines("test/test_ops/test_voxel_pooling.py", 15, 30), ns)
    geom_f = ns["geom_xyz"]              # [2, 6000, 3] float
    features = ns["features"]            # [2,6,10,10,10,80]
    gt = ns["gt_bev_featuremap"]         # [2,128,128,80] from the reference loop
    geom_i = geom_f.int()                # what the test feeds (geom_xyz.cuda().int())

    from ops.voxel_pooling import voxel_pooling  # reference autograd.Function, stub ext
    feats = features.clone().requires_grad_(True)
    out = voxel_pooling(geom_i.contiguous(), feats, torch.tensor([128, 128, 1], dtype=torch.int))
    assert out.shape == (2, 80, 128, 128)
    # forward cross-check: oracle-through-reference-wrapper == reference test loop
    diff = (out.detach().permute(0, 2, 3, 1) - gt).abs().max().item()
    assert diff == 0.0, diff
    grad_out = torch.from_numpy(hashed_f32(tuple(out.shape), salt=1))  # not stored
    out.backward(grad_out)
    grad_in = feats.grad.reshape(2, -1, 80)

    # pos_memo is not returned by the reference op.  It is derived here WITHOUT the oracle (the reference test's own drop
    # predicate, evaluated with numpy), validated against the reference's own backward output, and only then compared with
    # what the oracle writes -- so the stored array pins the oracle instead of being its output.
    pos_memo = independent_pos_memo(geom_i.numpy(), 128, 128, 1)
    check_pos_memo_against_reference_backward(pos_memo, grad_out.numpy(), grad_in.numpy())
    _, pos_oracle = oracle.voxel_pooling_forward(geom_i.numpy(), features.reshape(2, -1, 80).numpy(), 128, 128, 1)
    assert np.array_equal(pos_memo, pos_oracle)
    np.savez_compressed(
        os.path.join(HERE, "vp_ref_test.npz"),
        geom_float=geom_f.numpy(), geom=geom_i.numpy(),
        feats=features.reshape(2, -1, 80).numpy(),
        voxel_num=np.array([128, 128, 1], np.int32),
        out_nhwc=gt.numpy(), pos_memo=pos_memo,
        grad_in=grad_in.numpy())  # grad_out = formula.hashed_f32((2,80,128,128), salt=1)
    print("vp_ref_test: kept frac", float((pos_memo[..., 0] != -1).mean()))


def make_vp_edge():
    """Hand-made edge set pushed through the reference autograd.Function."""
    from ops.voxel_pooling import voxel_pooling
    rng = np.random.default_rng(7)
    cases = {}
    nx, ny, nz = 16, 12, 2
    INT_MAX, INT_MIN = 2**31 - 1, -2**31
    specials = [-1, 0, 1, nx - 1, nx, ny - 1, ny, nz - 1, nz, INT_MAX, INT_MIN, -7, 1000]
    for name, (B, P, C) in {"c1": (1, 67, 1), "c3": (2, 129, 3), "c64": (1, 300, 64),
                            "c80": (2, 257, 80), "c81": (1, 65, 81)}.items():
        geom = rng.choice(specials, size=(B, P, 3)).astype(np.int32)
        # make roughly half the points valid
        valid = rng.random((B, P)) < 0.5
        geom[valid, 0] = rng.integers(0, nx, valid.sum())
        geom[valid, 1] = rng.integers(0, ny, valid.sum())
        geom[valid, 2] = rng.integers(0, nz, valid.sum())
        feats = (rng.random((B, P, C), dtype=np.float32) - 0.5)
        cases[name] = (geom, feats)
    # all dropped, all same cell
    cases["alldrop"] = (np.full((2, 70, 3), -1, np.int32), rng.random((2, 70, 5), dtype=np.float32))
    same = np.zeros((1, 500, 3), np.int32)
    same[..., 0], same[..., 1] = 3, 5
    cases["samecell"] = (same, rng.random((1, 500, 80), dtype=np.float32) - 0.5)
    out = {"grid": np.array([nx, ny, nz], np.int32)}
    for name, (geom, feats) in cases.items():
        f = torch.from_numpy(feats.copy()).requires_grad_(True)
        o = voxel_pooling(torch.from_numpy(geom), f, torch.tensor([nx, ny, nz]))
        go = torch.from_numpy(rng.standard_normal(tuple(o.shape)).astype(np.float32))
        o.backward(go)
        pm = independent_pos_memo(geom, nx, ny, nz)                        # oracle-independent (see make_vp_ref_test)
        check_pos_memo_against_reference_backward(pm, go.numpy(), f.grad.numpy())
        assert np.array_equal(pm, oracle.voxel_pooling_forward(geom, feats, nx, ny, nz)[1])
        out[name + "_geom"] = geom
        out[name + "_feats"] = feats
        out[name + "_out_nchw"] = o.detach().numpy()
        out[name + "_pos_memo"] = pm
        out[name + "_grad_out"] = go.numpy()
        out[name + "_grad_in"] = f.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "vp_edge.npz"), **out)
    print("vp_edge:", sorted(cases))


def _make_lss(x_bound, y_bound, z_bound, d_bound, final_dim, downsample):
    import layers.backbones.lss_fpn as ref_lss
    ref_lss.LSSFPN._configure_depth_net = lambda self, conf: torch.nn.Identity()
    ref_lss.build_backbone = lambda conf: _Dummy()
    ref_lss.build_neck = lambda conf: _Dummy()
    m = ref_lss.LSSFPN(x_bound, y_bound, z_bound, d_bound, final_dim, downsample, 80, {}, {}, {})
    return ref_lss, m


def _quantize_with_reference(ref_lss, m, xyz):
    """Evaluate the reference's quantise expression (lss_fpn.py:461-462) verbatim
    from its source text, with `self` = the reference module instance."""
    src = inspect.getsource(ref_lss.LSSFPN._forward_single_sweep)
    mm = re.search(r"geom_xyz = (\(\(geom_xyz - .*?\.int\(\))", src, re.S)
    assert mm, "quantise expression not found in reference source"
    return eval(mm.group(1), {"self": m, "geom_xyz": xyz})


def _rig(B, N, W_img, H_img, seed=0):
    """Analytic 6-camera rig: yaw fan, f = 0.8*W, centre pp."""
    yaws = np.deg2rad([0, 55, -55, 180, 110, -110])[:N]
    rng = np.random.default_rng(seed)
    s2e = np.zeros((B, N, 4, 4), np.float32)
    K = np.zeros((B, N, 4, 4), np.float32)
    axis = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], np.float64)
    for b in range(B):
        for n, yaw in enumerate(yaws):
            yaw = yaw + (rng.random() - 0.5) * 0.02 * b
            R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
            s2e[b, n, :3, :3] = R @ axis
            s2e[b, n, :3, 3] = [1.5 * np.cos(yaw), 1.5 * np.sin(yaw), 1.5]
            s2e[b, n, 3, 3] = 1
            f = 0.8 * W_img
            K[b, n] = np.array([[f, 0, W_img / 2, 0], [0, f, H_img / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    return s2e, K


def make_quant_and_geom():
    out = {}
    grids = {
        "nusc": dict(x=[-51.2, 51.2, 0.8], y=[-51.2, 51.2, 0.8], z=[-5, 3, 8], d=[2.0, 58.0, 0.5],
                     dim=(256, 704), ds=16),
        "aim": dict(x=[-204.8, 204.8, 0.8], y=[-25.6, 25.6, 0.8], z=[-5, 3, 8], d=[2.0, 206.4, 0.5],
                    dim=(704, 1280), ds=16),
        "test": dict(x=[-10, 10, 0.5], y=[-10, 10, 0.5], z=[-5, 3, 8], d=[2.0, 22.0, 1.0],
                     dim=(64, 64), ds=4),
    }
    rng = np.random.default_rng(3)
    for name, gcf in grids.items():
        ref_lss, m = _make_lss(gcf["x"], gcf["y"], gcf["z"], gcf["d"], gcf["dim"], gcf["ds"])
        out[name + "_bounds"] = np.array([gcf["x"], gcf["y"], gcf["z"]], np.float64)
        out[name + "_d_bound"] = np.array(gcf["d"], np.float64)
        out[name + "_final_dim"] = np.array(gcf["dim"], np.int32)
        out[name + "_ds"] = np.array(gcf["ds"], np.int32)
        out[name + "_voxel_size"] = m.voxel_size.numpy()
        out[name + "_voxel_coord"] = m.voxel_coord.numpy()
        out[name + "_voxel_num"] = m.voxel_num.numpy()
        fr = m.frustum.numpy()
        if name != "aim":
            out[name + "_frustum"] = fr
        else:  # big: keep a strided sample + shape
            out[name + "_frustum_shape"] = np.array(fr.shape, np.int32)
            out[name + "_frustum_sample"] = fr[::37, ::5, ::7].copy()
        # quantise fixtures: xyz straddling cell boundaries + specials
        vs, vc, vn = m.voxel_size.numpy(), m.voxel_coord.numpy(), m.voxel_num.numpy()
        lo = vc - vs / 2
        n = 4096
        cell = rng.integers(-2, vn.max() + 2, size=(n, 3)).astype(np.float64)
        frac = rng.choice([0.0, 1e-7, -1e-7, 0.5, 0.999999, 1e-3, -1e-3], size=(n, 3))
        xyz = (lo[None] + (cell + frac) * vs[None]).astype(np.float32)
        # nudge by ulps
        ulp = rng.integers(-2, 3, size=(n, 3))
        xyz = np.where(ulp > 0, np.nextafter(xyz, np.float32(np.inf)), xyz)
        xyz = np.where(ulp < 0, np.nextafter(xyz, np.float32(-np.inf)), xyz)
        special = np.array([[0, 0, 0], [-0.3, -0.3, -4.9], [1e9, -1e9, 0], [3e9, -3e9, 1e20]], np.float32)
        special = special * 1.0 + lo[None] * np.array([[0], [1], [0], [0]], np.float32)
        xyz = np.concatenate([xyz, special.astype(np.float32)], 0)
        q = _quantize_with_reference(ref_lss, m, torch.from_numpy(xyz)).numpy()
        # torch-CPU .int() of out-of-int32-range values is x86 UB (INT_MIN); the
        # device semantics (saturate) differ only there -- store a validity mask.
        with np.errstate(all="ignore"):
            qf = (xyz - lo[None].astype(np.float32)) / vs[None]
        finite_ok = np.abs(qf) < 2147483648.0
        out[name + "_q_xyz"] = xyz
        out[name + "_q_expected"] = q
        out[name + "_q_inrange"] = finite_ok

    # geometry through the reference get_geometry on the analytic rig (nusc grid)
    gcf = grids["nusc"]
    ref_lss, m = _make_lss(gcf["x"], gcf["y"], gcf["z"], gcf["d"], gcf["dim"], gcf["ds"])
    s2e, K = _rig(2, 6, gcf["dim"][1], gcf["dim"][0])
    s2e_t, K_t = torch.from_numpy(s2e), torch.from_numpy(K)
    xyz = m.get_geometry(s2e_t, K_t, None)                      # [B,N,D,fH,fW,3]
    combine = s2e_t.matmul(torch.inverse(K_t))
    q = _quantize_with_reference(ref_lss, m, xyz)
    out["rig_sensor2ego"] = s2e
    out["rig_intrin"] = K
    out["rig_combine"] = combine.numpy()
    out["rig_xyz_sample"] = xyz.reshape(-1, 3)[::97].numpy().copy()
    out["rig_geom_sample"] = q.reshape(-1, 3)[::97].numpy().copy()
    out["rig_geom_sum"] = np.array([int(q.long().sum()), int((q.long() ** 2 % 1000003).sum())], np.int64)
    out["rig_shape"] = np.array(xyz.shape, np.int32)
    # the reference's own test fixture: a real 6-camera nuScenes calibration
    # (test/data/nuscenes/infos.pkl), full-resolution 900x1600 images, ds 16
    import pickle
    info = pickle.load(open(os.path.join(REF, "test/data/nuscenes/infos.pkl"), "rb"))[0]
    cams = ["CAM_FRONT_LEFT", "CAM_FRONT", "CAM_FRONT_RIGHT", "CAM_BACK_LEFT", "CAM_BACK", "CAM_BACK_RIGHT"]
    s2e_n = np.zeros((1, 6, 4, 4), np.float32)
    K_n = np.zeros((1, 6, 4, 4), np.float32)
    for i, cam in enumerate(cams):
        cs = info["cam_infos"][cam]["calibrated_sensor"]
        w, x, y, z = cs["rotation"]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        s2e_n[0, i, :3, :3] = R
        s2e_n[0, i, :3, 3] = cs["translation"]
        s2e_n[0, i, 3, 3] = 1
        K_n[0, i, :3, :3] = np.array(cs["camera_intrinsic"])
        K_n[0, i, 3, 3] = 1
    ref_lss, mn = _make_lss(gcf["x"], gcf["y"], gcf["z"], gcf["d"], (900, 1600), 16)
    s2e_t, K_t = torch.from_numpy(s2e_n), torch.from_numpy(K_n)
    xyz_n = mn.get_geometry(s2e_t, K_t, None)
    q_n = _quantize_with_reference(ref_lss, mn, xyz_n)
    out["nusc_fixture_combine"] = s2e_t.matmul(torch.inverse(K_t)).numpy()
    out["nusc_fixture_frustum_shape"] = np.array(mn.frustum.shape, np.int32)
    out["nusc_fixture_xyz_sample"] = xyz_n.reshape(-1, 3)[::211].numpy().copy()
    out["nusc_fixture_geom_sample"] = q_n.reshape(-1, 3)[::211].numpy().copy()
    kept = ((q_n[..., 0] >= 0) & (q_n[..., 0] < 128) & (q_n[..., 1] >= 0) & (q_n[..., 1] < 128) & (q_n[..., 2] >= 0) & (q_n[..., 2] < 1))
    out["nusc_fixture_kept_fraction"] = np.array(float(kept.float().mean()))
    print("nuScenes fixture rig: points", tuple(xyz_n.shape), "kept", float(kept.float().mean()))
    np.savez_compressed(os.path.join(HERE, "quant_geom.npz"), **out)
    print("quant_geom: grids", list(grids), "rig xyz", tuple(xyz.shape))
    return xyz.numpy(), q.numpy()


def make_lss_forward():
    """The caller's branches around the op, from the reference's own LSSFPN._forward_single_sweep / forward
    (layers/backbones/lss_fpn.py:381-468, :469-529) run UNMODIFIED on CPU tensors: the softmax taken BEFORE the per-camera
    un-flip of depth_feature (:423-425), the oracle-depth overwrite (:427-438), the lift (:441-443), get_geometry + quantise
    + voxel_pooling (:455-465, the extension call answered by the C oracle), the older sweeps under no_grad and the channel
    stacking (:516-529), plus the gradient the reference's autograd sends back to the network output.
    Stand-ins, all outside the lines under test: get_cam_feats returns the "images" as they are (the fixture's images ARE
    the neck features), the depth net is the identity (depth_feature = features: no arithmetic, so both sides see the same
    bits), kornia's hflip is torch.flip(-1) (its documented definition; kornia is not vendored), Tensor.cuda is a no-op."""
    import layers.backbones.lss_fpn as ref_lss
    D_BOUND, DIM, DS, C = [2.0, 26.0, 2.0], (64, 96), 16, 64
    bounds = ([-25.6, 25.6, 0.8], [-25.6, 25.6, 0.8], [-5, 3, 8])
    ref_lss.LSSFPN._configure_depth_net = lambda self, conf: torch.nn.Identity()
    ref_lss.build_backbone = lambda conf: _Dummy()
    ref_lss.build_neck = lambda conf: _Dummy()
    m = ref_lss.LSSFPN(*bounds, D_BOUND, DIM, DS, C, {}, {}, {})
    m.get_cam_feats = lambda imgs: imgs
    m._forward_depth_net = lambda feat, mats: feat
    ref_lss.kornia = types.SimpleNamespace(geometry=types.SimpleNamespace(transform=types.SimpleNamespace(hflip=lambda t: t.flip(-1))))
    D = m.depth_channels
    fH, fW = DIM[0] // DS, DIM[1] // DS
    B, S, N = 2, 2, 2
    rng = np.random.default_rng(11)
    s2e = np.zeros((B, S, N, 4, 4), np.float32)
    K = np.zeros((B, S, N, 4, 4), np.float32)
    for k in range(S):
        a, b_ = _rig(B, N, DIM[1], DIM[0], seed=20 + k)
        s2e[:, k], K[:, k] = a, b_
        s2e[:, k, :, 0, 3] += 0.7 * k                         # the older sweep saw the scene from elsewhere
    out = dict(d_bound=np.array(D_BOUND), final_dim=np.array(DIM, np.int32), ds=np.int32(DS), channels=np.int32(C),
               bounds=np.array(bounds, np.float64), sensor2ego=s2e, intrin=K)
    cuda_attr = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for case, (use_oracle, sweeps) in {"flip": (False, 2), "oracle": (True, 2), "single": (True, 1)}.items():
            imgs = torch.from_numpy(rng.standard_normal((B, sweeps, N, D + C, fH, fW)).astype(np.float32)).requires_grad_(True)
            flipped = np.array([True, False, False, True])[:B * N]
            oracle_t = None
            if use_oracle:
                o = np.zeros((B * N, D, fH, fW), np.float32)
                fg = rng.random((B * N, fH, fW)) < 0.5                               # foreground pixels carry a (soft) one-hot
                hot = rng.integers(0, D, (B * N, fH, fW))
                for idx in np.argwhere(fg):
                    o[idx[0], hot[tuple(idx)], idx[1], idx[2]] = 0.75
                    o[idx[0], (hot[tuple(idx)] + 1) % D, idx[1], idx[2]] = 0.25
                oracle_t = torch.from_numpy(o)
            mats = dict(sensor2ego_mats=torch.from_numpy(s2e[:, :sweeps]), intrin_mats=torch.from_numpy(K[:, :sweeps]),
                        bda_mat=torch.eye(4).repeat(B, 1, 1), flipped=flipped.tolist())
            bev, depth = m.forward(imgs, mats, oracle_t, None, is_return_depth=True)
            assert tuple(bev.shape) == (B, sweeps * C, int(m.voxel_num[1]), int(m.voxel_num[0])) and tuple(depth.shape) == (B * N, D, fH, fW)
            go = torch.from_numpy(hashed_f32(tuple(bev.shape), salt=31 + sweeps))
            bev.backward(go)
            assert imgs.grad[:, 1:].abs().max().item() == 0.0 if sweeps > 1 else True        # older sweeps: no_grad (:516-524)
            out[case + "_imgs"] = imgs.detach().numpy()
            out[case + "_flipped"] = flipped
            if use_oracle:
                out[case + "_depth_oracle"] = oracle_t.numpy()
            out[case + "_bev"] = bev.detach().numpy()
            out[case + "_depth"] = depth.detach().numpy()
            out[case + "_grad_imgs"] = imgs.grad.numpy()
            out[case + "_grad_out_salt"] = np.int32(31 + sweeps)
            print("lss_forward", case, "bev", tuple(bev.shape), "nonzero cells", int((bev.detach().abs().sum(1) > 0).sum()))
    finally:
        torch.Tensor.cuda = cuda_attr
    np.savez_compressed(os.path.join(HERE, "lss_forward.npz"), **out)


def make_depth_labels():
    """Depth supervision labels from the reference's own methods:
    exps/mm_training_aim.py get_depth_labels (:114-140), get_depth_image (:142-163) and
    get_downsampled_gt_depth (:180-215), read from the file at run time and exec'd as plain
    functions on CPU tensors (`self` = a namespace with the three attributes they use).
    Inputs are drawn so that no projected point sits within 0.02 px of a pixel boundary or of
    the image-border tests, no depth within 1e-3 of a bin boundary or of the depth > 1 test, and
    no two points share a pixel -- the expected labels then do not depend on the summation
    order of the reference's matmuls nor on which of two writes to a pixel lands last."""
    src = "\n".join(_ref_source_lines("exps/mm_training_aim.py", lo, hi) for lo, hi in ((114, 140), (142, 163), (180, 215)))
    ns = {"torch": torch, "F": torch.nn.functional}
    exec(src, ns)
    rng = np.random.default_rng(42)
    B, N, H, W, ds = 2, 3, 64, 96, 16
    d_bound = (2.0, 58.0, 0.5)
    D = int((d_bound[1] - d_bound[0]) / d_bound[2])
    s2e, K = (torch.from_numpy(m) for m in _rig(B, N, W, H, seed=7))   # sensor2ego, intrinsics [B,N,4,4]
    extr = torch.inverse(s2e.double()).float()
    ang = np.array([0.3, -0.2])
    bda = torch.eye(4).repeat(B, 1, 1)
    for b in range(B):
        c, s_ = np.cos(ang[b]), np.sin(ang[b])
        bda[b, :3, :3] = torch.tensor([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], dtype=torch.float32) * (1.0 if b == 0 else 1.05)
    clouds = []
    for b in range(B):
        cand = np.concatenate([rng.uniform(-60, 60, (6000, 2)), rng.uniform(-3, 3, (6000, 1)),
                               rng.uniform(0, 1, (6000, 2))], 1).astype(np.float32)
        Rinv = np.linalg.inv(bda[b, :3, :3].double().numpy())
        keep = np.ones(len(cand), bool)
        seen = [set() for _ in range(N)]
        for n in range(N):
            q = cand[:, :3].astype(np.float64) @ Rinv.T
            cam = (extr[b, n].double().numpy() @ np.concatenate([q, np.ones((len(q), 1))], 1).T)
            pr = K[b, n].double().numpy() @ cam
            depth, u, v = cam[2], pr[0] / pr[2], pr[1] / pr[2]
            fr = lambda a: np.abs(a - np.round(a))
            risky = (fr(u) < 0.02) | (fr(v) < 0.02) | (np.abs(depth - 1.0) < 1e-3) | (np.abs(pr[2]) < 1e-3)
            g = (depth - (d_bound[0] - d_bound[2])) / d_bound[2]
            risky |= fr(g) < 2e-3
            inside = (depth > 1) & (u > 1) & (u < W - 1) & (v > 1) & (v < H - 1)
            for i in np.nonzero(inside & keep)[0]:
                key = (int(v[i]), int(u[i]))
                if key in seen[n]:
                    keep[i] = False
                else:
                    seen[n].add(key)
            keep &= ~risky
        clouds.append(torch.from_numpy(cand[keep]))
    images = torch.zeros(B, 1, N, 3, H, W)
    mats = {"extrinsics": extr.unsqueeze(1), "intrin_mats": K.unsqueeze(1), "bda_mat": bda}
    self_ = types.SimpleNamespace(downsample_factor=ds, dbound=list(d_bound), depth_channels=D)
    self_.get_depth_image = lambda *a: ns["get_depth_image"](self_, *a)
    self_.get_downsampled_gt_depth = lambda *a: ns["get_downsampled_gt_depth"](self_, *a)
    labels = ns["get_depth_labels"](self_, images, mats, clouds)          # [B*N*fH*fW, D] one-hot float
    labels = labels.reshape(-1, D).numpy()
    bins = labels.argmax(1).astype(np.int32)
    assert labels.sum(1).min() == 1.0 and labels.sum(1).max() == 1.0
    for pixel_last in (False, True):
        ob, oh = oracle.depth_labels([c.numpy() for c in clouds], extr.numpy(), K.numpy(), bda.numpy(), (H, W), ds, d_bound,
                                     pixel_last=pixel_last)
        assert np.array_equal(ob, bins), (pixel_last, int((ob != bins).sum()))
        assert np.array_equal(oh, labels)
    np.savez_compressed(os.path.join(HERE, "depth_labels.npz"),
                        points=np.concatenate([c.numpy() for c in clouds], 0),
                        offsets=np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).astype(np.int32),
                        extrinsics=extr.numpy(), intrinsics=K.numpy(), bda=bda.numpy(),
                        img_hw=np.array([H, W], np.int32), downsample=np.int32(ds), d_bound=np.array(d_bound, np.float32),
                        bins=bins)
    print("depth_labels: points", [len(c) for c in clouds], "labelled cells", int((bins > 0).sum()), "of", bins.size)


def make_augment_images():
    """The per-step image augmentation from the reference's own lines: exps/mm_training_aim.py:88-112 (augment_images:
    flips = np.random.uniform(size = b*s*n) > 0.5, kornia hflip of the flagged images and label maps) and :510-512
    (normalize_images), read from the file at run time and exec'd as plain functions.  Stand-ins for the two absent
    third-party callables, by their documented definitions: kornia.geometry.transform.hflip(x) = x.flip(-1);
    torchvision.transforms.Normalize(mean, std)(x) = (x - mean[:, None, None]) / std[:, None, None].  numpy's global generator
    is seeded so that the test can draw the same flags.  Stored: the raw images (integer-valued, 4 channels so that the `:3`
    slice of normalize_images matters), the label maps, the normalised images (CPU arithmetic: a true division by 255), the
    flags and both augmented outputs, for the 'train' stage; the 'val' stage must hand everything back untouched."""
    src = _ref_source_lines("exps/mm_training_aim.py", 88, 112) + "\n" + _ref_source_lines("exps/mm_training_aim.py", 510, 512)
    hflip = lambda t: t.flip(-1)

    class _Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std

    from typing import Tuple
    ns = {"torch": torch, "np": np, "Tuple": Tuple,
          "kornia": types.SimpleNamespace(geometry=types.SimpleNamespace(transform=types.SimpleNamespace(hflip=hflip))),
          "torchvision": types.SimpleNamespace(transforms=types.SimpleNamespace(Normalize=_Normalize))}
    exec(src, ns)
    rng = np.random.default_rng(5)
    B, S, N, H, W, fH, fW, D = 2, 1, 3, 32, 48, 2, 3, 5
    raw = rng.integers(0, 256, (B, S, N, 4, H, W)).astype(np.float32)
    hot = rng.integers(0, D, (B * S * N, fH, fW))
    labels = np.eye(D, dtype=np.float32)[hot] * (rng.random((B * S * N, fH, fW, 1)) < 0.7)
    norm = ns["normalize_images"](None, torch.from_numpy(raw))
    assert tuple(norm.shape) == (B, S, N, 3, H, W)
    seed = 20240
    np.random.seed(seed)
    imgs_t, labels_t, flips = ns["augment_images"](None, norm.clone(), torch.from_numpy(labels.astype(np.float32)), 'train')
    assert flips.dtype == bool and flips.any() and not flips.all(), flips
    iv, lv, fv = ns["augment_images"](None, norm, torch.from_numpy(labels.astype(np.float32)), 'val')
    assert torch.equal(iv, norm) and torch.equal(lv, torch.from_numpy(labels.astype(np.float32))) and not fv.any()
    np.savez_compressed(os.path.join(HERE, "augment_images.npz"), raw=raw, labels=labels.astype(np.float32), normalized=norm.numpy(),
                        seed=np.int64(seed), flips=flips, aug_images=imgs_t.numpy(), aug_labels=labels_t.numpy())
    print("augment_images: flips", flips.astype(int).tolist(), "images", tuple(imgs_t.shape), "labels", tuple(labels_t.shape))



# ---- stand-ins for the third-party callables of the CenterPoint head (mmdet3d 1.0.0rc4 / mmdet 2.25.1, un-vendored: README.md:19-27),
# by their published definitions -- the reference's own lines call them, so the fixtures pin the reference's regrouping, packing,
# masking and normaliser arithmetic around them (like `hflip` / `Normalize` in make_augment_images)
def _gaussian_radius(det_size, min_overlap=0.5):
    """mmdet3d.core.utils.gaussian.gaussian_radius (CornerNet's three quadratic bounds on tensors)."""
    height, width = det_size
    a1, b1, c1 = 1, (height + width), width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + torch.sqrt(b1 ** 2 - 4 * a1 * c1)) / 2
    a2, b2, c2 = 4, 2 * (height + width), (1 - min_overlap) * width * height
    r2 = (b2 + torch.sqrt(b2 ** 2 - 4 * a2 * c2)) / 2
    a3, b3, c3 = 4 * min_overlap, -2 * min_overlap * (height + width), (min_overlap - 1) * width * height
    r3 = (b3 + torch.sqrt(b3 ** 2 - 4 * a3 * c3)) / 2
    return min(r1, r2, r3)


def _gaussian_2d(shape, sigma=1):
    m, n = [(ss - 1.) / 2. for ss in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def _draw_heatmap_gaussian(heatmap, center, radius, k=1):
    """mmdet3d.core.utils.gaussian.draw_heatmap_gaussian: a (2r+1)^2 float64 window, sigma = (2r+1)/6, max-combined in place."""
    diameter = 2 * radius + 1
    gaussian = _gaussian_2d((diameter, diameter), sigma=diameter / 6)
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape[0:2]
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    masked_heatmap = heatmap[y - top:y + bottom, x - left:x + right]
    masked_gaussian = torch.from_numpy(gaussian[radius - top:radius + bottom, radius - left:radius + right]).to(heatmap.device, torch.float32)
    if min(masked_gaussian.shape) > 0 and min(masked_heatmap.shape) > 0:
        torch.max(masked_heatmap, masked_gaussian * k, out=masked_heatmap)
    return heatmap


def _head_cases():
    """(name, class_names per task, feature map, out_size_factor, max_objs, boxes per sample) of the CenterPoint fixtures"""
    return [("aim", [['car'], ['truck/bus'], ['motorcycle'], ['pedestrian']], (128, 128), 4, 500, (40, 0, 7)),      # exps/conf_aim.py:125-130,143-160
            ("multi", [['a', 'b'], ['c'], ['d', 'e', 'f']], (512, 64), 4, 6, (30, 9)),                                 # more boxes than max_objs
            ("small", [['a', 'b', 'c', 'd']], (40, 56), 2, 16, (12,))]


def _head_inputs(case, rng):
    name, class_names, (fx, fy), osf, max_objs, counts = case
    vs = (0.2, 0.2, 8.0)
    pc = (-vs[0] * osf * fx / 2, -vs[1] * osf * fy / 2, -5.0, vs[0] * osf * fx / 2, vs[1] * osf * fy / 2, 3.0)
    n_cls = sum(len(c) for c in class_names)
    boxes, labels = [], []
    for k in counts:
        xy = rng.uniform([pc[0] - 3, pc[1] - 3], [pc[3] + 3, pc[4] + 3], (k, 2))          # some centres outside the map
        dims = rng.uniform(0.3, 12.0, (k, 3))
        if k > 3:
            dims[0, 0] = 0.0                       # zero width: skipped (:183)
            xy[1] = [pc[0] + 0.01, pc[1] + 0.01]   # window clipped at the map corner
            xy[2] = [pc[3] - 0.01, pc[4] - 0.01]
        bx = np.concatenate([xy, rng.uniform(-2, 1, (k, 1)), dims, rng.uniform(-3.2, 3.2, (k, 1)), rng.normal(size=(k, 2))], 1).astype(np.float32)
        # keep the centre coordinates away from cell boundaries (the int() of a float32 quotient must not hinge on a rounding)
        for a in (0, 1):
            q = (bx[:, a] - pc[a]) / vs[a] / osf
            bad = np.abs(q - np.round(q)) < 1e-3
            bx[bad, a] += 0.05
        boxes.append(bx)
        labels.append(rng.integers(0, n_cls, k).astype(np.int64))
    train_cfg = dict(point_cloud_range=list(pc), grid_size=[fx * osf, fy * osf, 1], voxel_size=list(vs), out_size_factor=osf, dense_reg=1,
                     gaussian_overlap=0.1, max_objs=max_objs, min_radius=2, code_weights=[1.0] * 8 + [0.3, 0.3])
    return boxes, labels, train_cfg


def make_centerpoint_targets():
    """CenterPoint training targets from the reference's own get_targets_single (layers/heads/bev_depth_head.py:113-254), read
    from the file at run time and exec'd as a plain function on CPU tensors (its four `device='cuda'` become 'cpu' in the
    text handed to exec), with mmdet3d's gaussian_radius / draw_heatmap_gaussian supplied by their published formulas.
    Cases: the aiMotive head (four single-class tasks, 128 x 128, max_objs 500; one sample without boxes), multi-class tasks
    with MORE boxes than max_objs, a small odd-sized map; boxes outside the map, of zero width, at the map corners."""
    src = _ref_source_lines("layers/heads/bev_depth_head.py", 113, 254).replace("device='cuda'", "device='cpu'")
    ns = {"torch": torch, "gaussian_radius": _gaussian_radius, "draw_heatmap_gaussian": _draw_heatmap_gaussian}
    exec(src, ns)
    rng = np.random.default_rng(77)
    out = {}
    for case in _head_cases():
        name, class_names, (fx, fy), osf, max_objs, counts = case
        boxes, labels, train_cfg = _head_inputs(case, rng)
        self_ = types.SimpleNamespace(train_cfg=train_cfg, class_names=class_names, task_heads=[None] * len(class_names), norm_bbox=True)
        out[name + "_n_samples"] = np.int32(len(counts))
        out[name + "_class_counts"] = np.array([len(c) for c in class_names], np.int32)
        out[name + "_cfg"] = np.array([fx, fy, osf, max_objs], np.int32)
        out[name + "_pc_range"] = np.array(train_cfg["point_cloud_range"], np.float32)
        for b, (bx, lb) in enumerate(zip(boxes, labels)):
            hm, anno, ind, mask = ns["get_targets_single"](self_, torch.from_numpy(bx), torch.from_numpy(lb))
            out[f"{name}_boxes_{b}"], out[f"{name}_labels_{b}"] = bx, lb
            for t in range(len(class_names)):
                out[f"{name}_hm_{b}_{t}"] = hm[t].numpy()
                out[f"{name}_anno_{b}_{t}"] = anno[t].numpy()
                out[f"{name}_ind_{b}_{t}"] = ind[t].numpy()
                out[f"{name}_mask_{b}_{t}"] = mask[t].numpy()
            print("centerpoint_targets:", name, "sample", b, "boxes", len(bx), "valid slots per task", [int(m.sum()) for m in mask])
    np.savez_compressed(os.path.join(HERE, "centerpoint_targets.npz"), **out)


def make_head_loss():
    """BEVDepthHead.loss (layers/heads/bev_depth_head.py:256-312) exec'd from the reference's text on CPU tensors: the masking,
    gather and normaliser arithmetic is the reference's; stand-ins by their published definitions for mmdet3d's clip_sigmoid
    (sigmoid clamped to [1e-4, 1 - 1e-4]), mmdet's reduce_mean (the identity in one process), GaussianFocalLoss (alpha 2,
    gamma 4, eps 1e-12, sum / avg_factor) and L1Loss (|pred - target| * weight, sum / avg_factor, loss_weight 0.25:
    exps/conf_aim.py:186-187), and CenterHead._gather_feat (gather of the rows at `ind`).  Targets: the 'aim' case of
    centerpoint_targets.npz stacked over its samples; predictions: seeded."""
    src = _ref_source_lines("layers/heads/bev_depth_head.py", 256, 312)
    clip_sigmoid = lambda x, eps=1e-4: torch.clamp(x.sigmoid_(), min=eps, max=1 - eps)

    def focal(pred, target, avg_factor, alpha=2.0, gamma=4.0, eps=1e-12):
        pos_w, neg_w = target.eq(1), (1 - target).pow(gamma)
        loss = -(pred + eps).log() * (1 - pred).pow(alpha) * pos_w - (1 - pred + eps).log() * pred.pow(alpha) * neg_w
        return loss.sum() / avg_factor

    def l1(pred, target, weight, avg_factor, loss_weight=0.25):
        return loss_weight * ((pred - target).abs() * weight).sum() / avg_factor

    def gather_feat(feat, ind):
        return feat.gather(1, ind.unsqueeze(2).expand(ind.size(0), ind.size(1), feat.size(2)))

    ns = {"torch": torch, "clip_sigmoid": clip_sigmoid, "reduce_mean": lambda t: t}
    exec(src, ns)
    g = np.load(os.path.join(HERE, "centerpoint_targets.npz"))
    B, T = int(g["aim_n_samples"]), len(g["aim_class_counts"])
    fx, fy = int(g["aim_cfg"][0]), int(g["aim_cfg"][1])
    stack = lambda f: [torch.from_numpy(np.stack([g[f"aim_{f}_{b}_{t}"] for b in range(B)])) for t in range(T)]
    targets = (stack("hm"), stack("anno"), stack("ind"), stack("mask"))
    heads = dict(reg=2, height=1, dim=3, rot=2, vel=2)
    preds, salt = [], 100
    for t in range(T):          # predictions by formula (tests/golden/formula.py): 4 x hashed_f32 in [-2, 2), heat-map logits shifted by -2
        d = {}
        for k, c in [("heatmap", int(g["aim_class_counts"][t]))] + list(heads.items()):
            salt += 1
            d[k] = torch.from_numpy(4.0 * hashed_f32((B, c, fy, fx), salt=salt)) - (2.0 if k == "heatmap" else 0.0)
        preds.append([d])
    code_weights = [1.0] * 8 + [0.3, 0.3]
    self_ = types.SimpleNamespace(loss_cls=focal, loss_bbox=l1, train_cfg=dict(code_weights=code_weights), _gather_feat=gather_feat)
    loss = ns["loss"](self_, targets, preds)
    np.savez_compressed(os.path.join(HERE, "head_loss.npz"), loss=np.float64(float(loss)), code_weights=np.array(code_weights, np.float32), first_salt=np.int64(101))
    print("head_loss:", float(loss))


def make_depth_loss():
    """get_depth_loss (exps/mm_training_aim.py:165-178) exec'd from the reference's text: BCE of the depth distribution on the
    labelled pixels (fg_mask = max over bins > 0), summed, / max(1, their count), x 3.  Labels one-hot on 60 % of the pixels."""
    src = _ref_source_lines("exps/mm_training_aim.py", 165, 178)
    import contextlib
    ns = {"torch": torch, "F": torch.nn.functional, "autocast": lambda enabled=False: contextlib.nullcontext()}
    exec(src, ns)
    rng = np.random.default_rng(8)
    BN, D, fH, fW = 6, 20, 4, 11
    hot = rng.integers(0, D, (BN * fH * fW,))
    labels = (np.eye(D, dtype=np.float32)[hot] * (rng.random((BN * fH * fW, 1)) < 0.6)).astype(np.float32)
    preds = torch.from_numpy(rng.standard_normal((BN, D, fH, fW)).astype(np.float32)).softmax(1)
    self_ = types.SimpleNamespace(depth_channels=D)
    loss = ns["get_depth_loss"](self_, torch.from_numpy(labels), preds)
    empty = ns["get_depth_loss"](self_, torch.zeros_like(torch.from_numpy(labels)), preds)          # no labelled pixel: 0 / max(1, 0)
    np.savez_compressed(os.path.join(HERE, "depth_loss.npz"), labels=labels, preds=preds.numpy(), loss=np.float64(float(loss)),
                        loss_without_labels=np.float64(float(empty)))
    print("depth_loss:", float(loss), float(empty))


def make_fuse_layer():
    """BEVFuseLayer (models/bev_depth.py:133-145): the reference's class itself, imported with the stubs in place, seeded
    weights, CPU forward and the input gradient of a seeded output gradient."""
    from models.bev_depth import BEVFuseLayer
    torch.manual_seed(12)
    C, B, H, W = 24, 2, 16, 12
    m = BEVFuseLayer(C)
    x = torch.randn(B, C, H, W, requires_grad=True)
    y = m(x)
    go = torch.from_numpy(hashed_f32(tuple(y.shape), salt=5))
    y.backward(go)
    np.savez_compressed(os.path.join(HERE, "fuse_layer.npz"), x=x.detach().numpy(), y=y.detach().numpy(), grad_x=x.grad.numpy(), grad_out_salt=np.int64(5),
                        **{"w_" + k: v.detach().numpy() for k, v in m.state_dict().items()},
                        **{"g_" + k: p.grad.numpy() for k, p in m.named_parameters()})
    print("fuse_layer:", tuple(y.shape), list(m.state_dict()))


def make_nusc_rig():
    """The separate matrices of the reference fixture's real 6-camera calibration (test/data/nuscenes/infos.pkl; quant_geom.npz
    holds only their product at 900 x 1600): sensor2ego and the 900 x 1600 intrinsics, for `bench.py --rig nuscenes`, which
    rescales the intrinsics to the benchmark's image size.  Data only (96 + 96 floats)."""
    import pickle
    info = pickle.load(open(os.path.join(REF, "test/data/nuscenes/infos.pkl"), "rb"))[0]
    cams = ["CAM_FRONT_LEFT", "CAM_FRONT", "CAM_FRONT_RIGHT", "CAM_BACK_LEFT", "CAM_BACK", "CAM_BACK_RIGHT"]
    s2e = np.zeros((6, 4, 4), np.float32)
    K = np.zeros((6, 4, 4), np.float32)
    for i, cam in enumerate(cams):
        cs = info["cam_infos"][cam]["calibrated_sensor"]
        w, x, y, z = cs["rotation"]
        s2e[i, :3, :3] = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        s2e[i, :3, 3] = cs["translation"]
        s2e[i, 3, 3] = 1
        K[i, :3, :3] = np.array(cs["camera_intrinsic"])
        K[i, 3, 3] = 1
    g = np.load(os.path.join(HERE, "quant_geom.npz"))
    assert np.allclose(torch.from_numpy(s2e).matmul(torch.inverse(torch.from_numpy(K))).numpy(), g["nusc_fixture_combine"][0], rtol=0, atol=1e-6)
    np.savez_compressed(os.path.join(HERE, "nusc_rig.npz"), sensor2ego=s2e, intrin=K, image_hw=np.array([900, 1600], np.int32))
    print("nusc_rig: 6 cameras, image 900 x 1600")


def main():
    _install_stubs()
    oracle.build()
    if "--only-depth-labels" in sys.argv:
        return make_depth_labels()
    if "--only-lss-forward" in sys.argv:
        return make_lss_forward()
    if "--only-augment-images" in sys.argv:
        return make_augment_images()
    if "--only-nusc-rig" in sys.argv:
        return make_nusc_rig()
    if "--only-head" in sys.argv:
        make_centerpoint_targets()
        make_head_loss()
        make_depth_loss()
        return make_fuse_layer()
    make_vp_ref_test()
    make_vp_edge()
    make_quant_and_geom()
    make_lss_forward()
    make_depth_labels()
    make_augment_images()
    make_centerpoint_targets()
    make_head_loss()
    make_depth_loss()
    make_fuse_layer()
    make_nusc_rig()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
