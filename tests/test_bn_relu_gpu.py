"""GPU numerics of the fused BatchNorm (+ residual) (+ ReLU) kernels against plain PyTorch fp32
(nn.BatchNorm2d in training mode + add + relu, autograd for the gradients)."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu

SHAPES = [(4, 64, 32, 44), (2, 256, 16, 22), (2, 2048, 4, 6), (3, 160, 9, 7), (2, 12, 5, 7), (1, 1024, 3, 3), (24, 64, 64, 176)]


def _ref(bn, x, res, relu):
    y = bn(x)
    if res is not None:
        y = y + res
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("use_res", [False, True])
@pytest.mark.parametrize("relu", [True, False])
def test_forward_backward_running_stats(mmt_lib, shape, use_res, relu):
    from mm_training_amd.ops import bn_relu
    from mm_training_amd.ops.bn_relu import bn_act
    B, C, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(C * 7 + H)
    x0 = (torch.randn(shape, device="cuda", generator=g) * 1.7 + 0.4).contiguous(memory_format=torch.channels_last)
    r0 = torch.randn(shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last) if use_res else None
    go = torch.randn(shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    bn_a, bn_b = nn.BatchNorm2d(C).cuda(), nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn_a.weight.copy_(torch.rand(C, device="cuda", generator=g) + 0.5)
        bn_a.bias.copy_(torch.randn(C, device="cuda", generator=g) * 0.3)
        bn_b.load_state_dict(bn_a.state_dict())
    assert bn_relu._supported(bn_a, x0), "this shape must take the fused path"
    outs = []
    for bn, fused in ((bn_a, True), (bn_b, False)):
        x = x0.clone().requires_grad_(True)
        r = r0.clone().requires_grad_(True) if use_res else None
        for _ in range(2):                                   # two steps: running statistics accumulate
            y = bn_act(bn, x, r, relu) if fused else _ref(bn, x, r, relu)
        y.backward(go)
        outs.append((y.detach(), x.grad, r.grad if use_res else None, bn.weight.grad, bn.bias.grad,
                     bn.running_mean.clone(), bn.running_var.clone()))
    names = ("y", "grad_x", "grad_res", "grad_weight", "grad_bias", "running_mean", "running_var")
    for name, a, b in zip(names, outs[0], outs[1]):
        if a is None:
            continue
        scale = max(1.0, b.abs().max().item())
        # per-channel sums over up to 270 k fp32 terms (both sides): 2e-3 relative
        tol = (2e-3 if name.startswith("grad_w") or name.startswith("grad_b") else 2e-5) * scale
        # a ReLU mask decided on an activation within rounding of zero may differ: allow isolated outliers
        bad = ((a - b).abs() > tol).float().mean().item()
        assert bad <= (1e-5 if name in ("y", "grad_x", "grad_res") else 0.0), (name, bad, (a - b).abs().max().item())
    assert outs[0][0].is_contiguous(memory_format=torch.channels_last)


def test_fallback_paths(mmt_lib):
    """eval mode, NCHW-contiguous input and unsupported C run the ordinary torch modules."""
    from mm_training_amd.ops import bn_relu
    from mm_training_amd.ops.bn_relu import bn_act
    bn = nn.BatchNorm2d(6).cuda()
    x = torch.randn(2, 6, 5, 5, device="cuda")
    assert not bn_relu._supported(bn, x)
    assert torch.allclose(bn_act(bn, x), torch.relu(nn.functional.batch_norm(x, None, None, bn.weight, bn.bias, True)), atol=1e-6)
    bn8 = nn.BatchNorm2d(8).cuda().eval()
    x8 = torch.randn(2, 8, 5, 5, device="cuda").contiguous(memory_format=torch.channels_last)
    assert not bn_relu._supported(bn8, x8)
    assert torch.equal(bn_act(bn8, x8, relu=False), bn8(x8))


@pytest.mark.parametrize("shape", [(4, 64, 32, 44), (2, 256, 16, 22), (2, 2048, 4, 6), (3, 160, 9, 7), (2, 12, 5, 7)])
@pytest.mark.parametrize("use_res", [False, True])
@pytest.mark.parametrize("relu", [True, False])
def test_bf16_activations_inside_autocast(mmt_lib, shape, use_res, relu):
    """Inside torch.autocast(bf16) the activations arrive and leave as bf16 (mmt_bn_relu_*_ex, ABI 9); statistics, running
    statistics and arithmetic are fp32.  Reference: the fp32 torch modules on the up-cast inputs, output rounded to bf16 --
    forward within one bf16 ulp, running statistics to fp32 accuracy, gradients within bf16 rounding of grad_x."""
    from mm_training_amd.ops import bn_relu
    from mm_training_amd.ops.bn_relu import bn_act
    B, C, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(C * 5 + W)
    x0 = (torch.randn(shape, device="cuda", generator=g) * 1.7 + 0.4).bfloat16().contiguous(memory_format=torch.channels_last)
    r0 = torch.randn(shape, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last) if use_res else None
    go = torch.randn(shape, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    bn_a, bn_b = nn.BatchNorm2d(C).cuda(), nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn_a.weight.copy_(torch.rand(C, device="cuda", generator=g) + 0.5)
        bn_a.bias.copy_(torch.randn(C, device="cuda", generator=g) * 0.3)
        bn_b.load_state_dict(bn_a.state_dict())
    assert not bn_relu._supported(bn_a, x0)                   # bf16 outside an autocast region: the torch modules
    x = x0.clone().requires_grad_(True)
    r = r0.clone().requires_grad_(True) if use_res else None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert bn_relu._supported(bn_a, x0)
        y = bn_act(bn_a, x, r, relu)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(go)
    assert x.grad.dtype == torch.bfloat16 and bn_a.weight.grad.dtype == torch.float32
    xf = x0.float().requires_grad_(True)
    rf = r0.float().requires_grad_(True) if use_res else None
    yf = _ref(bn_b, xf, rf, relu)
    yf.backward(go.float())
    yf = yf.detach()
    ulp = 2.0 ** -7                                           # bf16: 8 bits of precision, round to nearest = half an ulp; one for the ReLU edge
    assert float((y.float() - yf).abs().max()) <= ulp * max(1.0, float(yf.abs().max()))
    assert torch.allclose(bn_a.running_mean, bn_b.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn_a.running_var, bn_b.running_var, rtol=1e-5, atol=1e-6)
    bad = ((x.grad.float() - xf.grad).abs() > ulp * max(1.0, float(xf.grad.abs().max()))).float().mean().item()
    assert bad <= 1e-5, bad                                   # (a ReLU mask decided within rounding of zero may differ)
    if use_res:
        assert float((r.grad.float() - rf.grad).abs().max()) <= ulp * max(1.0, float(rf.grad.abs().max()))
    for a, b in ((bn_a.weight.grad, bn_b.weight.grad), (bn_a.bias.grad, bn_b.bias.grad)):
        assert float((a - b).abs().max()) <= 2e-3 * max(1.0, float(b.abs().max()))
    # a residual of another dtype is cast on the way in and its gradient comes back in its own dtype
    if use_res:
        r32 = r0.float().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y2 = bn_act(bn_a, x0.clone().requires_grad_(True), r32, relu)
        y2.backward(go)
        assert r32.grad.dtype == torch.float32 and y2.dtype == torch.bfloat16


@pytest.mark.parametrize("shape", [(4, 64, 32, 44), (2, 256, 16, 22), (24, 64, 64, 176)])
@pytest.mark.parametrize("use_res", [True, False])
def test_forked_output_adds_its_two_gradients_inside_the_backward(mmt_lib, shape, use_res):
    """bn_act(..., fork=True): the output as a pair of aliases; the backward receives the gradient of either use and adds them while
    loading (mmt_bn_relu_backward_ex2).  Against plain autograd on the same graph (one output used twice), with either alias unused,
    and with both."""
    from mm_training_amd.ops.bn_relu import bn_act
    B, C, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(C + H)
    x0 = (torch.randn(shape, device="cuda", generator=g) * 1.3).contiguous(memory_format=torch.channels_last)
    r0 = torch.randn(shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last) if use_res else None
    w1 = torch.randn(shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    w2 = torch.randn(shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    bn_a, bn_b = nn.BatchNorm2d(C).cuda(), nn.BatchNorm2d(C).cuda()
    bn_b.load_state_dict(bn_a.state_dict())
    # three aliases: every one used
    bn_c = nn.BatchNorm2d(C).cuda()
    bn_c.load_state_dict(bn_a.state_dict())
    w3 = torch.randn(shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    grads3 = []
    for bn, fork in ((bn_a, 3), (bn_c, 0)):
        bn.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        r = r0.clone().requires_grad_(True) if use_res else None
        ys = bn_act(bn, x, r, True, fork=fork) if fork else (bn_act(bn, x, r, True),) * 3
        assert len(ys) == 3 and ys[0].data_ptr() == ys[2].data_ptr()
        ((ys[0] * w1).sum() + (ys[1] * w2).sum() + (ys[2] * w3).sum()).backward()
        grads3.append((x.grad, r.grad if use_res else None, bn.weight.grad, bn.bias.grad))
    for name, a, b in zip(("grad_x", "grad_res", "grad_weight", "grad_bias"), grads3[0], grads3[1]):
        if a is not None:
            tol = (2e-3 if name in ("grad_weight", "grad_bias") else 2e-5) * max(1.0, b.abs().max().item())
            assert ((a - b).abs() > tol).float().mean().item() <= 1e-5, ("three aliases", name, (a - b).abs().max().item())
    for use in ((True, True), (True, False), (False, True)):
        res = []
        for bn, fork in ((bn_a, True), (bn_b, False)):
            bn.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            r = r0.clone().requires_grad_(True) if use_res else None
            if fork:
                ya, yb = bn_act(bn, x, r, True, fork=True)
                assert ya.data_ptr() == yb.data_ptr()
            else:
                ya = yb = bn_act(bn, x, r, True)
            loss = (ya * w1).sum() * float(use[0]) + (yb * w2).sum() * float(use[1]) if all(use) else ((ya * w1).sum() if use[0] else (yb * w2).sum())
            loss.backward()
            res.append((x.grad, r.grad if use_res else None, bn.weight.grad, bn.bias.grad))
        for name, a, b in zip(("grad_x", "grad_res", "grad_weight", "grad_bias"), res[0], res[1]):
            if a is None:
                continue
            tol = (2e-3 if name in ("grad_weight", "grad_bias") else 2e-5) * max(1.0, b.abs().max().item())
            assert ((a - b).abs() > tol).float().mean().item() <= 1e-5, (use, name, (a - b).abs().max().item())


def test_resnet_stage_with_forked_blocks_equals_plain_blocks(mmt_lib):
    """layers/nets.py::ResNet hands block outputs on as pairs inside a stage: same outputs and gradients as with MMT_BN_FORK off."""
    from mm_training_amd.layers.nets import ResNet
    from mm_training_amd.ops import bn_relu
    torch.manual_seed(3)
    m = ResNet(depth=50, in_channels=3, base_channels=16, num_stages=3, strides=(1, 2, 2), out_indices=(0, 1, 2)).cuda().to(memory_format=torch.channels_last)
    x0 = torch.randn(2, 3, 64, 96, device="cuda").contiguous(memory_format=torch.channels_last)
    res = []
    try:
        for fork in (True, False):
            bn_relu.FORK = fork
            m.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            outs = m(x)
            sum((o * o).mean() for o in outs).backward()
            res.append(([o.detach() for o in outs], x.grad, [p.grad.clone() for p in m.parameters()]))
    finally:
        bn_relu.FORK = True
    for a, b in zip(res[0][0], res[1][0]):                     # (MIOpen's convolutions are not bit-stable from call to call)
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
    assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-4 * float(res[1][1].abs().max())
    for a, b in zip(res[0][2], res[1][2]):
        assert float((a - b).abs().max()) <= 2e-3 * max(1e-6, float(b.abs().max()))


@pytest.mark.parametrize("use_res", [False, True])
def test_gradient_that_is_a_channel_slice_is_read_in_place(mmt_lib, use_res):
    """The gradient of one input of a torch.cat arrives as a channel slice of the concatenation's gradient (rows a wider pitch
    apart): the backward reads it through its row stride (no .contiguous() copy) and gives the gradients of the dense case."""
    from mm_training_amd.ops.bn_relu import bn_act
    torch.manual_seed(5)
    B, C, H, W = 6, 64, 16, 44
    xs = [(torch.randn(B, C, H, W, device="cuda") * 1.5).contiguous(memory_format=torch.channels_last) for _ in range(3)]
    rs = [torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last) for _ in range(3)]
    wsum = torch.randn(B, 3 * C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    bns = [nn.BatchNorm2d(C).cuda() for _ in range(3)]
    res = []
    for dense in (False, True):
        for bn in bns:
            bn.zero_grad(set_to_none=True)
        xi = [x.clone().requires_grad_(True) for x in xs]
        ri = [r.clone().requires_grad_(True) for r in rs]
        ys = [bn_act(bn, x, r if use_res else None, True) for bn, x, r in zip(bns, xi, ri)]
        if dense:
            loss = sum((y * wsum[:, k * C:(k + 1) * C].contiguous(memory_format=torch.channels_last)).sum() for k, y in enumerate(ys))
        else:
            loss = (torch.cat(ys, 1) * wsum).sum()              # cat backward hands each branch a slice of wsum-shaped memory
        loss.backward()
        res.append([x.grad for x in xi] + ([r.grad for r in ri] if use_res else []) + [bn.weight.grad for bn in bns] + [bn.bias.grad for bn in bns])
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b), float((a - b).abs().max())
