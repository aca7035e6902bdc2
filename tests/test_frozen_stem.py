"""The image backbone's frozen stem (exps/conf_aim.py:57: ResNet built with frozen_stages=0, norm_eval=False -- mmdet's ResNet then keeps
conv1 + norm1 without gradients and norm1 in eval mode while the rest trains, and `train()` re-applies it)."""
import copy

import pytest
import torch


def test_frozen_stages_semantics_cpu():
    from mm_training_amd.layers.nets import ResNet
    torch.manual_seed(0)
    net = ResNet(depth=18, base_channels=8, frozen_stages=0, norm_eval=False)
    assert not net.bn1.training and net.stages[0][0].bn1.training
    net.eval()
    net.train()                                              # (mmdet's train() freezes again)
    assert net.training and not net.bn1.training and net.stages[0][0].bn1.training
    assert not net.conv1.weight.requires_grad and not net.bn1.weight.requires_grad and not net.bn1.bias.requires_grad
    assert all(p.requires_grad for p in net.stages.parameters())
    before = copy.deepcopy(net.bn1.state_dict())
    outs = net(torch.randn(2, 3, 64, 64))
    sum(o.square().mean() for o in outs).backward()
    assert net.conv1.weight.grad is None and net.bn1.weight.grad is None
    assert all(p.grad is not None for p in net.stages.parameters())
    for k, v in net.bn1.state_dict().items():
        assert torch.equal(v, before[k]), k                  # running statistics and the batch counter stand still
    # frozen_stages=1 also freezes the first stage; norm_eval keeps every BatchNorm in eval mode while training
    net1 = ResNet(depth=18, base_channels=8, frozen_stages=1, norm_eval=True).train()
    assert not any(p.requires_grad for p in net1.stages[0].parameters()) and all(p.requires_grad for p in net1.stages[1].parameters())
    assert not any(m.training for m in net1.modules() if isinstance(m, torch.nn.BatchNorm2d))
    # the default (the BEV trunk of the head: no frozen_stages in bev_backbone_conf) trains everything
    net2 = ResNet(depth=18, base_channels=8).train()
    assert net2.bn1.training and all(p.requires_grad for p in net2.parameters())


def test_torchvision_style_checkpoint_loads_and_a_random_frozen_stem_warns(tmp_path, monkeypatch):
    """ADVICE (round 5): the reference freezes a PRETRAINED stem (exps/conf_aim.py:57-60, torchvision://resnet50).  A checkpoint with
    torchvision / mmdet keys (layerN.*, fc.*) loads into this ResNet (stages.N-1.*), directly and through init_cfg with
    $MMT_PRETRAINED_DIR; a frozen stem that never received weights warns at its first training forward, a loaded or unfrozen one
    does not."""
    import warnings
    from mm_training_amd.layers.nets import ResNet
    torch.manual_seed(1)
    src = ResNet(depth=18, base_channels=8)
    with torch.no_grad():
        src.bn1.running_mean.normal_(); src.bn1.running_var.uniform_(0.5, 2.0)
    tv = {}
    for k, v in src.state_dict().items():
        parts = k.split(".")
        tv[".".join(["layer%d" % (int(parts[1]) + 1)] + parts[2:]) if parts[0] == "stages" else k] = v.clone()
    tv["fc.weight"], tv["fc.bias"] = torch.zeros(10, 64), torch.zeros(10)          # an ImageNet classifier rides along
    assert any(k.startswith("layer4.1.") for k in tv) and any(".downsample.0.weight" in k for k in tv)
    dst = ResNet(depth=18, base_channels=8, frozen_stages=0)
    missing, unexpected = dst.load_state_dict(dict(tv), strict=True)
    assert not missing and not unexpected and dst._weights_loaded
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        dst.train()(torch.randn(2, 3, 64, 64))                                      # loaded: silent
        ResNet(depth=18, base_channels=8).train()(torch.randn(2, 3, 64, 64))       # nothing frozen: silent
    with pytest.warns(UserWarning, match="RANDOM initialisation"):
        ResNet(depth=18, base_channels=8, frozen_stages=0).train()(torch.randn(2, 3, 64, 64))
    # the reference's init_cfg: resolved through MMT_PRETRAINED_DIR; a name that resolves to nothing says so
    torch.save({"state_dict": {"backbone." + k: v for k, v in tv.items()}}, tmp_path / "resnet18.pth")
    monkeypatch.setenv("MMT_PRETRAINED_DIR", str(tmp_path))
    via_cfg = ResNet(depth=18, base_channels=8, frozen_stages=0, init_cfg=dict(type="Pretrained", checkpoint="torchvision://resnet18"))
    assert via_cfg._weights_loaded and torch.equal(via_cfg.stages[3][1].conv2.weight, src.stages[3][1].conv2.weight)
    with pytest.warns(UserWarning, match="not found"):
        ResNet(depth=18, base_channels=8, init_cfg=dict(type="Pretrained", checkpoint="torchvision://resnet999"))


def test_bench_configurations_freeze_the_stem():
    from mm_training_amd.dp import make_config
    for name in ("cfg2", "cfg4", "cfg5", "tiny"):
        conf = make_config(name)["backbone_conf"]["img_backbone_conf"]
        assert conf["frozen_stages"] == 0 and conf["norm_eval"] is False


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_eval_mode_fused_batchnorm(mmt_lib, dtype):
    """mmt_bn_relu_inference against nn.BatchNorm2d in eval mode (+ residual) (+ ReLU)."""
    from mm_training_amd.ops.bn_relu import bn_act
    torch.manual_seed(0)
    for (B, C, H, W, use_res, relu) in ((2, 64, 16, 24, False, True), (3, 16, 5, 7, True, True), (1, 256, 8, 8, False, False), (2, 1536, 4, 4, True, False)):
        bn = torch.nn.BatchNorm2d(C).cuda()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
        bn.eval()
        for p in bn.parameters():
            p.requires_grad = False
        x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        res = torch.randn_like(x) if use_res else None
        ref = bn(x) + (res if use_res else 0)
        ref = torch.relu(ref) if relu else ref
        stats = (bn.running_mean.clone(), bn.running_var.clone())
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype is torch.bfloat16):
            y = bn_act(bn, x.to(dtype), res.to(dtype) if use_res else None, relu=relu)
        assert y.dtype == dtype and y.shape == x.shape and not y.requires_grad
        tol = 1e-5 if dtype is torch.float32 else 6e-2
        assert float((y.float() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max())), (C, dtype)
        assert torch.equal(bn.running_mean, stats[0]) and torch.equal(bn.running_var, stats[1])


@pytest.mark.gpu
def test_training_step_leaves_the_frozen_stem_alone(mmt_lib):
    import numpy as np
    from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
    torch.manual_seed(0)
    np.random.seed(0)
    cfg = make_config("tiny")
    ts = TrainStep(cfg, torch.device("cuda", 0))
    stem = ts.model.backbone.img_backbone
    assert not stem.bn1.training and not stem.conv1.weight.requires_grad
    before = {k: v.clone() for k, v in stem.state_dict().items() if k.startswith(("conv1.", "bn1."))}
    other = stem.stages[0][0].conv1.weight.detach().clone()
    losses = [float(ts(synthetic_batch(cfg, torch.device("cuda", 0), seed=i % 2))[0]) for i in range(4)]
    assert all(np.isfinite(losses))
    for k, v in stem.state_dict().items():
        if k in before:
            assert torch.equal(v, before[k]), k              # weights, running statistics and the batch counter
    assert not torch.equal(stem.stages[0][0].conv1.weight.detach(), other)
