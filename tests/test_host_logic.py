"""CPU: host-side logic of the hot path that needs no GPU -- the measure LSSFPN uses to pick the backward kernel of the fused
lift-splat, and the forward / backward algorithmic-byte formulas of bench.py against their closed forms."""
import math

import pytest

import torch


def _rig(pitch_deg=0.0, B=1, N=2, H=64, W=96, D=20):
    from mm_training_amd import synthetic
    s2e, K = synthetic.camera_rig(B, N, W, H, jitter=0.02, seed=0)
    c, s = math.cos(math.radians(pitch_deg)), math.sin(math.radians(pitch_deg))
    rx = torch.tensor([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=torch.float32)
    xyz = synthetic.frustum_geometry_xyz(s2e.matmul(rx), K, (H, W), 16, (2.0, 2.0 + 0.5 * D, 0.5))
    geom, vn = synthetic.quantize_cpu(xyz, (-51.2, 51.2, 0.8), (-51.2, 51.2, 0.8), (-5.0, 3.0, 8.0))
    return geom.contiguous(), [int(v) for v in vn]


def test_column_mismatch_fraction_is_zero_for_a_level_rig_and_grows_with_pitch():
    """What LSSFPN.lift_splat_backward = "auto" looks at (ops/bev_geometry.py): the share of kept frustum points whose BEV cell
    differs from the smallest kept cell among the 16 image rows of their block at the same depth bin."""
    from mm_training_amd.ops.bev_geometry import column_mismatch_fraction
    level, vn = _rig(0.0)
    assert float(column_mismatch_fraction(level, vn)) == 0.0
    fracs = [float(column_mismatch_fraction(_rig(p)[0], vn)) for p in (1.0, 3.0, 8.0)]
    assert 0.0 < fracs[0] < fracs[1] < fracs[2] < 1.0
    # both point orders, and a brute-force count on a small case
    g, vn = _rig(3.0, H=80, W=48, D=9)        # fH = 5: a single, padded 16-row block
    pm = g.permute(0, 1, 3, 4, 2, 5).contiguous()
    assert float(column_mismatch_fraction(g, vn)) == float(column_mismatch_fraction(pm, vn, pixel_major=True))
    nx, ny, nz = vn
    B, N, D, fH, fW, _ = g.shape
    mism = kept = 0
    for b in range(B):
        for n in range(N):
            for d in range(D):
                for w in range(fW):
                    cells = []
                    for h in range(fH):
                        x, y, z = g[b, n, d, h, w].tolist()
                        if 0 <= x < nx and 0 <= y < ny and 0 <= z < nz:
                            cells.append(y * nx + x)
                    kept += len(cells)
                    mism += sum(1 for c in cells if c != min(cells))
    assert abs(float(column_mismatch_fraction(g, vn)) - mism / max(kept, 1)) < 1e-6


def test_bench_algorithmic_byte_formulas():
    """bench.py's roofline numerators are SURVEY 8(d)'s formulas; the cfg2 / cfg4 camera shape gives the figures DESIGN quotes."""
    import bench
    BP, C, B, ny, nx = 4 * 473088, 80, 4, 128, 128
    K = int(0.601 * BP)
    fwd, bwd = bench.algorithmic_bytes(BP, K, C, B, ny, nx)
    assert fwd == 24 * BP + 4 * C * K + 4 * C * B * ny * nx and bwd == 12 * BP + 4 * C * B * ny * nx + 4 * C * BP
    f, b_, l2f, l2b = bench.lift_splat_bytes(BP, K, C, B, 24 * 704, ny, nx)
    assert f == 56655872 and b_ == 69632000                       # 56.7 MB / 69.6 MB
    assert l2f == f + K * 4 * C and l2b == b_ + K * 4 * C
    f16, b16, _, _ = bench.lift_splat_bytes(BP, K, C, B, 24 * 704, ny, nx, feat_bytes=2)
    assert f16 < f and b16 < b_


def test_parameter_registration_order():
    """`parameters()` order is what an optimizer state dict of a reference checkpoint is keyed by.  Recorded from the
    reference's registration order (source order of the assignments): models/bev_depth.py:26-29 (backbone, head),
    :153-161 (lidar_encoder, bev_fuse); layers/backbones/lss_fpn.py:293-295 (img_backbone, img_neck, depth_net),
    :164-184 (reduce_conv, context_conv, depth_se, context_se, depth_conv), :53-90 (aspp1..4, global_avg_pool, conv1, bn1)."""
    from mm_training_amd.dp.configs import make_config
    from mm_training_amd.models.bev_depth import BEVDepth, BEVDepthLiDAR
    cfg = make_config("tiny")

    def first_seen(names, depth):
        out = []
        for n in names:
            p = ".".join(n.split(".")[:depth])
            if p not in out:
                out.append(p)
        return out

    cam = BEVDepth(cfg["backbone_conf"], cfg["head_conf"], is_train_depth=True)
    names = [n for n, _ in cam.named_parameters()]
    assert first_seen(names, 1) == ["backbone", "head"]
    assert first_seen([n for n in names if n.startswith("backbone.")], 2) == \
        ["backbone.img_backbone", "backbone.img_neck", "backbone.depth_net"]
    dn = [n[len("backbone.depth_net."):] for n in names if n.startswith("backbone.depth_net.")]
    assert first_seen(dn, 1) == ["reduce_conv", "context_conv", "context_se", "depth_conv"]
    aspp = [n[len("depth_conv.3."):] for n in dn if n.startswith("depth_conv.3.")]
    assert first_seen(aspp, 1) == ["aspp1", "aspp2", "aspp3", "aspp4", "global_avg_pool", "conv1", "bn1"]
    assert [n for n, _ in cam.named_buffers(recurse=True) if n.startswith("backbone.") and n.count(".") == 1][:4] == \
        ["backbone.voxel_size", "backbone.voxel_coord", "backbone.voxel_num", "backbone.frustum"]

    fusion = BEVDepthLiDAR(cfg["backbone_conf"], cfg["head_conf"], cfg["lidar_conf"],
                           fuse_layer_in_channels=cfg["fuse_layer_in_channels"])
    top = first_seen([n for n, _ in fusion.named_parameters()], 1)
    assert top[:2] == ["backbone", "head"] and top[-1] == "bev_fuse"        # the pillar encoder has no parameters
    assert [n for n, _ in fusion.named_children()] == ["backbone", "head", "lidar_encoder", "bev_fuse"]
    assert [n for n, _ in fusion.bev_fuse.named_parameters()] == ["conv_3.weight", "conv_3.bias", "conv_1.weight", "conv_1.bias"]


def test_exclusive_cache_sizing_and_the_module_switch(monkeypatch):
    """Host side of the exclusive-cell cache (include/mmt_hip.h `exclusive_cache`): its size, and that LSSFPN owns none
    when MMT_LSS_EXCL_SLOTS=0 and keeps no state of it in the module's state dict."""
    from mm_training_amd import _lib
    from mm_training_amd.dp.configs import make_config
    from mm_training_amd.layers.backbones.lss_fpn import LSSFPN
    lib = _lib.lib()
    header = 64 + 8 * (8 + 8 * 16)                                    # words: fixed part + the mailbox of 8 samples x 8 cameras
    per_slot = 4 + 6 * 16 + 128 * 128
    assert lib.mmt_lss_exclusive_cache_bytes(6, 128, 128, 1024) == 4 * (header + 1024 * per_slot)
    assert lib.mmt_lss_exclusive_cache_bytes(6, 128, 128, 1 << 20) == 4 * (header + (1 << 16) * per_slot)      # capped
    assert lib.mmt_lss_exclusive_cache_bytes(6, 40000, 40000, 4) == 0                                         # cell ids need 30 bits
    cfg = make_config("tiny")
    monkeypatch.setenv("MMT_LSS_EXCL_SLOTS", "0")
    off = LSSFPN(**cfg["backbone_conf"])
    assert off.exclusive_slots == 0 and off._exclusive_cache_for(4, 6, 16, 44, "cpu") is None
    monkeypatch.setenv("MMT_LSS_EXCL_SLOTS", "32")
    on = LSSFPN(**cfg["backbone_conf"])
    assert on.exclusive_slots == 32 and not any("excl" in k for k in on.state_dict())
    # the module asks the library which shapes take a cache at all (mmt_lss_exclusive_cache_used) and allocates for no other:
    # BASELINE configs[1] / [3] (16 rows, C = 80: register walk) do; the reference's native 409-bin frustum and more than
    # 8 cameras per sample do not
    assert lib.mmt_lss_exclusive_cache_used(4, 6, 112, 16, 44, 80) == 1
    assert lib.mmt_lss_exclusive_cache_used(4, 2, 409, 44, 80, 80) == 0
    assert lib.mmt_lss_exclusive_cache_used(1, 9, 112, 16, 44, 80) == 0 and lib.mmt_lss_exclusive_cache_used(1, 6, 112, 16, 44, 72) == 0


def test_loaded_frustum_reaches_the_camera_form():
    """`frustum` is a persistent buffer like the reference's (lss_fpn.py:291): after load_state_dict the camera form must read
    the LOADED frustum's axes (and fall back to the geom form for a frustum that is no outer product of three axes)."""
    import torch
    from mm_training_amd.dp.configs import make_config
    from mm_training_amd.layers.backbones.lss_fpn import LSSFPN
    cfg = make_config("tiny")
    m = LSSFPN(**cfg["backbone_conf"])
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["frustum"][..., 2] += 0.25                       # other depth bins than the constructor's d_bound gives
    m.load_state_dict(sd)
    assert torch.equal(m.frustum_d, sd["frustum"][:, 0, 0, 2]) and m._has_frustum_axes
    assert torch.equal(m.frustum_pixel_major, sd["frustum"].permute(1, 2, 0, 3))
    sd["frustum"][3, 1, 2, 0] += 1.0                    # one point off the grid of axes: no camera form for this frustum
    m.load_state_dict(sd)
    assert not m._has_frustum_axes and torch.equal(m.frustum, sd["frustum"])


def test_conv_overlap_switch_is_transparent_on_the_host():
    """ops/conv_overlap.py re-classes nn.Conv2d modules in place: same parameters and state_dict keys, deepcopy-safe, CPU /
    no-grad calls fall through to nn.Conv2d, and TrainStep picks the mode by world size (deferred needs no DDP hooks)."""
    import copy
    import torch
    from mm_training_amd.ops import conv_overlap
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.ConvTranspose2d(4, 2, 2, 2))
    keys = list(net.state_dict())
    x = torch.randn(2, 3, 8, 8)
    want = net(x)
    assert conv_overlap.enable(net, "deferred") == 1              # the transposed convolution is left alone
    assert isinstance(net[0], torch.nn.Conv2d) and type(net[0]).__name__ == "OverlapConv2d"
    assert list(net.state_dict()) == keys
    assert torch.equal(net(x), want)                              # CPU input: nn.Conv2d's own forward
    clone = copy.deepcopy(net)
    clone[0].weight.data.zero_()
    assert not torch.equal(clone(x), want) and torch.equal(net(x), want)      # the copy runs on ITS parameters
    assert conv_overlap.enable(net, "pair") == 1 and net[0]._mmt_overlap_mode == "pair"
    assert conv_overlap.enable(net, "inline") == 1 and net[0]._mmt_overlap_mode == "inline"
    with pytest.raises(ValueError):
        conv_overlap.enable(net, "both")
    conv_overlap.join()                                           # no side stream yet: nothing to wait for
