"""GPU: the drop-in model surface (models/bev_depth.py, layers/backbones) and one
data-parallel training step on the tiny configuration."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_lssfpn_signature_and_hot_path(mmt_lib, oracle_mod):
    """LSSFPN.forward(sweep_imgs, mats_dict, depth_oracle, timestamps, is_return_depth)
    -> ([B, C, ny, nx], depth [B*N, D, fH, fW]) like lss_fpn.py:469-529; the BEV map equals
    the oracle's pooling of the module's own lifted features and geometry."""
    import numpy as np
    from mm_training_amd.dp import make_config, synthetic_batch
    from mm_training_amd.layers.backbones import LSSFPN
    from mm_training_amd.ops.bev_geometry import lift_features
    cfg = make_config("tiny")
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = LSSFPN(**cfg["backbone_conf"]).to(dev).eval()
    assert m.fused_lift_splat            # the default path (fused lift-splat, row f1) is the one checked against the oracle
    imgs, mats, pcs, boxes, labels = synthetic_batch(cfg, dev)
    with torch.no_grad():
        bev, depth = m(imgs[:, :, :, :3] / 255.0, mats, None, None, is_return_depth=True)
        m.fused_lift_splat = False
        bev_unfused = m(imgs[:, :, :, :3] / 255.0, mats, None, None, is_return_depth=False)
        m.fused_lift_splat = True
    assert (bev - bev_unfused).abs().max().item() <= 1e-4
    B, N = imgs.shape[0], imgs.shape[2]
    D, fH, fW = m.frustum.shape[:3]
    C = cfg["backbone_conf"]["output_channels"]
    nx, ny, nz = [int(v) for v in m.voxel_num]
    assert bev.shape == (B, C, ny, nx) and depth.shape == (B * N, D, fH, fW)
    assert torch.allclose(depth.sum(1), torch.ones_like(depth.sum(1)), atol=1e-5)
    # recompute the op chain by hand against the oracle
    with torch.no_grad():
        feat = m.depth_net(m.get_cam_feats(imgs[:, :, :, :3] / 255.0)[:, 0].reshape(B * N, -1, fH, fW), mats)
        ctx = feat[:, D:D + C]
        lifted = lift_features(depth, ctx).view(B, N * D * fH * fW, C)
        geom = m.get_geometry_voxels(mats["sensor2ego_mats"][:, 0], mats["intrin_mats"][:, 0]).view(B, -1, 3)
    ref, _ = oracle_mod.voxel_pooling_forward(geom.cpu().numpy(), lifted.cpu().numpy(), nx, ny, nz)
    assert np.abs(bev.permute(0, 2, 3, 1).cpu().numpy() - ref).max() <= 1e-4


def test_bevdepth_lidar_forward_contract(mmt_lib):
    """BEVDepthLiDAR.forward((img, lidar), mats_dict, lidar_oracle) ->
    (preds, depth_pred, lidar_bev, cam_bev) (models/bev_depth.py:163-200)."""
    from mm_training_amd.dp import make_config, synthetic_batch
    from mm_training_amd.models import BEVDepthLiDAR
    cfg = make_config("tiny")
    dev = torch.device("cuda", 0)
    model = BEVDepthLiDAR(cfg["backbone_conf"], cfg["head_conf"], cfg["lidar_conf"], use_cam=True, use_lidar=True,
                          fuse_layer_in_channels=cfg["fuse_layer_in_channels"]).to(dev)
    imgs, mats, pcs, boxes, labels = synthetic_batch(cfg, dev)
    preds, depth_pred, lidar_bev, cam_bev = model((imgs[:, :, :, :3] / 255.0, pcs), mats, None)
    assert len(preds) == 4 and set(preds[0][0]) == {"reg", "height", "dim", "rot", "vel", "heatmap"}
    assert preds[0][0]["heatmap"].shape == (2, 1, 128, 128)
    # default: only the pillar cells the nearest resize samples are scattered (straight into the fusion buffer), and the third
    # return value -- bound and dropped by the reference, exps/mm_training_aim.py:268 -- is that LiDAR half
    assert cam_bev.shape == (2, 16, 128, 128) and lidar_bev.shape == (2, 8, 128, 128)
    assert depth_pred.shape[1] == model.backbone.depth_channels
    model.full_lidar_canvas = True          # the reference's op sequence and its full-resolution lidar_bev_ret
    _, _, lidar_full, cam_bev2 = model((imgs[:, :, :, :3] / 255.0, pcs), mats, None)
    assert lidar_full.shape == (2, 8, 512, 512) and cam_bev2.shape == cam_bev.shape


@pytest.mark.parametrize("fused", [False, True])
def test_training_step_decreases_loss(mmt_lib, fused):
    from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
    cfg = make_config("tiny")
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    ts = TrainStep(cfg, dev, lr=2e-4)
    ts.model.backbone.fused_lift_splat = fused
    batch = synthetic_batch(cfg, dev, seed=3)
    before = {n: p.detach().clone() for n, p in list(ts.model.named_parameters())[:5]}
    losses = [float(ts(batch)[0]) for _ in range(6)]
    assert all(l == l and abs(l) < 1e6 for l in losses)
    assert losses[-1] < losses[0]
    assert any(not torch.equal(before[n], p.detach()) for n, p in list(ts.model.named_parameters())[:5])
    # the unused context_se parameters (reference quirk) never get a gradient
    assert ts.model.backbone.depth_net.context_se.conv_reduce.weight.grad is None


def test_training_step_under_bf16_autocast(mmt_lib):
    """BASELINE configs[4] names bf16: the whole model under torch.autocast(bf16).  Every op of this repository that reads
    fp32 rows must survive what autocast hands it -- the learned pillar MLP (nn.Linear) returns bf16 features, which the
    pillar scatter used to read as fp32 rows: an out-of-bounds read that took the process down with "Memory access fault by
    GPU" (tools/repro_bf16_fault.py; round 2 blamed MIOpen for it)."""
    from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
    from mm_training_amd.lidar import pillar_scatter
    cfg = make_config("tiny")
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    ts = TrainStep(cfg, dev, lr=2e-4, amp="bf16")
    assert ts.amp_dtype == torch.bfloat16 and ts.model.lidar_encoder.pillar_mlp is not None
    batch = synthetic_batch(cfg, dev, seed=3)
    losses = [float(ts(batch)[0]) for _ in range(6)]
    assert all(l == l and abs(l) < 1e6 for l in losses) and losses[-1] < losses[0]
    # the op itself: bf16 rows under autocast are cast, bf16 rows without autocast are refused (never read as fp32)
    feats = torch.randn(50, 8, device=dev)
    coors = torch.stack([torch.zeros(50), torch.zeros(50), torch.arange(50) // 10, torch.arange(50) % 10], 1).int().to(dev)
    ref = pillar_scatter(feats, coors, 1, 8, 16, channels_last=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        got = pillar_scatter(feats.bfloat16(), coors, 1, 8, 16, channels_last=True)
    assert got.dtype == torch.float32 and torch.equal(got, pillar_scatter(feats.bfloat16().float(), coors, 1, 8, 16, channels_last=True))
    assert (got - ref).abs().max().item() < 0.05
    with pytest.raises(RuntimeError, match="Float"):
        pillar_scatter(feats.bfloat16(), coors, 1, 8, 16)


def test_lssfpn_picks_the_backward_kernel_from_the_geometry(mmt_lib):
    """LSSFPN.lift_splat_backward = "auto": the matrix-core column backward for a rig whose columns are level (no kept point
    leaves its column's cell), the ray walk otherwise; decided once per calibration id; "ray" / "column" override it."""
    import math
    from mm_training_amd import synthetic
    from mm_training_amd.dp import make_config
    from mm_training_amd.layers.backbones import LSSFPN
    cfg = make_config("tiny")
    m = LSSFPN(**cfg["backbone_conf"]).cuda()
    assert m.lift_splat_backward == "auto"
    H, W = cfg["final_dim"]
    s2e, K = synthetic.camera_rig(2, 2, W, H, jitter=0.02, seed=0)
    level = m.get_geometry_voxels(s2e.cuda(), K.cuda(), pixel_major=True)
    c_, s_ = math.cos(math.radians(3.0)), math.sin(math.radians(3.0))
    rx = torch.tensor([[1, 0, 0, 0], [0, c_, -s_, 0], [0, s_, c_, 0], [0, 0, 0, 1]], dtype=torch.float32)
    pitched = m.get_geometry_voxels(s2e.matmul(rx).cuda(), K.cuda(), pixel_major=True)
    with torch.no_grad():
        assert m._use_column_backward(level, None) == (False, None) and not m._column_backward_choice     # nothing to decide without a backward
    assert m._use_column_backward(level, "rig-a") == (True, None)
    assert m._use_column_backward(lambda: pitched, "rig-b") == (False, None)      # a callable is evaluated for the one measurement
    assert m._use_column_backward(pitched, "rig-a") == (True, None)        # remembered per calibration id, not re-measured
    assert m._column_backward_choice == {"rig-a": True, "rig-b": False}
    # without an id: the column kernel's own counters decide, lazily (tests/test_camera_form_gpu.py); the first answer is "column"
    col, stats = m._use_column_backward(None, None)
    assert col is True and stats.dtype == torch.int64 and stats.numel() == 2 * mmt_lib.LSS_STATS_SLOTS and stats.is_cuda
    m.lift_splat_backward = "ray"
    assert m._use_column_backward(level, "rig-a") == (False, None)
    m.lift_splat_backward = "column"
    assert m._use_column_backward(pitched, "rig-b") == (True, None)


def test_lssfpn_cached_plan_matches_uncached(mmt_lib):
    """mats_dict['calibration_id']: the plan is built on the first call, reused on
    the next ones, and BEV map + gradients agree with the uncached drop-in path."""
    from mm_training_amd.dp import make_config, synthetic_batch
    from mm_training_amd.layers.backbones import LSSFPN
    cfg = make_config("tiny")
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = LSSFPN(**cfg["backbone_conf"]).to(dev).train()
    assert m.fused_lift_splat            # default: the fused kernels; the cached plan belongs to the unfused op sequence
    m.fused_lift_splat = False
    imgs, mats, *_ = synthetic_batch(cfg, dev)
    x = imgs[:, :, :, :3] / 255.0

    def run(mats_dict):
        m.zero_grad(set_to_none=True)
        torch.manual_seed(1)
        bev = m(x, mats_dict)
        bev.square().mean().backward()
        g = m.depth_net.depth_conv[0].conv1.weight.grad if hasattr(m.depth_net.depth_conv[0], "conv1") \
            else next(p.grad for p in m.depth_net.parameters() if p.grad is not None)
        return bev.detach().clone(), g.detach().clone()

    ref_bev, ref_g = run(mats)
    cached = dict(mats, calibration_id=("rig", 0))
    assert len(m._plan_cache) == 0
    bev1, g1 = run(cached)
    assert len(m._plan_cache) == 1
    plan = next(iter(m._plan_cache.values()))
    bev2, g2 = run(cached)
    assert len(m._plan_cache) == 1 and next(iter(m._plan_cache.values())) is plan
    assert (bev1 - ref_bev).abs().max().item() <= 1e-4 * max(1.0, ref_bev.abs().max().item())
    # (bit-reproducibility of the op itself: test_voxel_pooling_plan_gpu; the nets in front of it run MIOpen's split-K kernels,
    # whose atomically accumulated sums differ in their last bits between two passes)
    assert (bev1 - bev2).abs().max().item() <= 1e-5 * max(1.0, bev1.abs().max().item())
    assert (g1 - ref_g).abs().max().item() <= 1e-3 * ref_g.abs().max().item() + 1e-7
    run(dict(mats, calibration_id=("rig", 1)))
    assert len(m._plan_cache) == 2
