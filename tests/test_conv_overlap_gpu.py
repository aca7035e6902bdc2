"""GPU: ops/conv_overlap.py -- the weight-gradient half of a convolution's backward on a side stream.  Same ATen / MIOpen calls as
nn.Conv2d's own backward, so the output is the same bits and the gradients agree to the last bits (MIOpen's split-K weight
gradients use fp32 atomics), in both modes, for the layer kinds the model
holds (1x1, 3x3 strided, dilated, grouped, with and without bias, channels_last), and a training step must produce the same
losses with the overlap on."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [
    # cin, cout, k, stride, padding, dilation, groups, bias
    (16, 32, 1, 1, 0, 1, 1, False),
    (16, 32, 3, 2, 1, 1, 1, False),
    (32, 32, 3, 1, 6, 6, 1, True),
    (32, 64, 3, 1, 1, 1, 4, True),
    (3, 16, 7, 2, 3, 1, 1, False),
]


@pytest.mark.parametrize("mode", ["inline", "pair", "deferred"])
def test_gradients_are_autograds(mmt_lib, mode):
    from mm_training_amd.ops import conv_overlap
    torch.manual_seed(0)
    for cin, cout, k, s, p, d, g, bias in CASES:
        ref = torch.nn.Conv2d(cin, cout, k, s, p, d, g, bias).cuda().to(memory_format=torch.channels_last)
        new = copy.deepcopy(ref)
        assert conv_overlap.enable(new, mode) == 1
        x = torch.randn(4, cin, 24, 40, device="cuda").contiguous(memory_format=torch.channels_last)
        outs = []
        for m in (ref, new):
            xi = x.clone().requires_grad_(True)
            y = m(xi)
            (y * torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)).sum().backward()
            # (deferred mode: the end-of-backward callback has made the main stream wait for the side stream)
            outs.append((y.detach(), xi.grad.clone(), m.weight.grad.clone(), m.bias.grad.clone() if bias else None))
        assert torch.equal(outs[0][0], outs[1][0])
        for a, b in zip(outs[0][1:], outs[1][1:]):
            assert (a is None) == (b is None)
            if a is not None:               # MIOpen's split-K weight gradients add with fp32 atomics: run-to-run last-bit differences
                assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()), (mode, cin, cout, k)
        # gradient accumulation: the weight already holds a gradient -> the layer joins on the spot, and the sum is right
        xi = x.clone().requires_grad_(True)
        new(xi).sum().backward()
        ref(xi).sum().backward()
        assert float((new.weight.grad - ref.weight.grad).abs().max()) <= 1e-5 * float(ref.weight.grad.abs().max())
    # an input that needs no gradient (the first layer of a net): only the weight half runs
    conv = torch.nn.Conv2d(3, 8, 3, 1, 1).cuda()
    conv_overlap.enable(conv, mode)
    conv(torch.randn(2, 3, 16, 16, device="cuda")).sum().backward()
    assert conv.weight.grad is not None and torch.isfinite(conv.weight.grad).all()
    # under autocast the casts happen inside the Function: bf16 arithmetic, gradients back in the parameters' / input's own dtype
    ref = torch.nn.Conv2d(16, 32, 3, 1, 1, bias=True).cuda().to(memory_format=torch.channels_last)
    new = copy.deepcopy(ref)
    conv_overlap.enable(new, mode)
    x = torch.randn(4, 16, 24, 40, device="cuda").contiguous(memory_format=torch.channels_last)
    got = []
    for m in (ref, new):
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(xi)
        assert y.dtype == torch.bfloat16
        y.float().square().sum().backward()
        assert xi.grad.dtype == torch.float32 and m.weight.grad.dtype == torch.float32 and m.bias.grad.dtype == torch.float32
        got.append((y.detach().float(), xi.grad.clone(), m.weight.grad.clone(), m.bias.grad.clone()))
    assert torch.equal(got[0][0], got[1][0])
    for a, b in zip(got[0][1:], got[1][1:]):
        assert float((a - b).abs().max()) <= 1e-2 * float(a.abs().max())
    # no-grad / eval calls fall through to nn.Conv2d
    with torch.no_grad():
        assert conv(torch.randn(2, 3, 16, 16, device="cuda")).shape == (2, 8, 16, 16)
    with pytest.raises(ValueError):
        conv_overlap.enable(conv, "sideways")


@pytest.mark.parametrize("mode", ["inline", "pair", "deferred"])
def test_training_step_losses_do_not_change(mmt_lib, mode, monkeypatch):
    from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
    cfg = make_config("tiny")
    dev = torch.device("cuda", 0)
    losses = {}
    for key in ("off", mode):
        monkeypatch.setenv("MMT_CONV_OVERLAP", key)
        torch.manual_seed(0)
        np.random.seed(0)
        ts = TrainStep(cfg, dev)
        assert ts.conv_overlap == (None if key == "off" else mode)
        ts.model.eval()                                   # no dropout; BatchNorm on running statistics: the steps compare exactly
        ts.model.backbone.fused_lift_splat = False        # deterministic pooling (no fp32 atomics)
        batch = synthetic_batch(cfg, dev, seed=3)
        imgs, mats, pcs, boxes, labels = batch
        batch = (imgs, dict(mats, calibration_id=("overlap", 0)), pcs, boxes, labels)
        losses[key] = [float(ts(batch)[0]) for _ in range(4)]
    assert all(np.isfinite(losses[mode]))
    # same kernels in another launch order on two streams: the DCN col2im atomics are the only source of a last-bit difference
    for a, b in zip(losses["off"], losses[mode]):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(a)), losses
    assert losses[mode][-1] < losses[mode][0]


@pytest.mark.parametrize("mode", ["pair", "deferred"])
def test_gradient_accumulation_with_layers_on_other_streams(mmt_lib, mode):
    """Layers whose forward ran on different streams (the task heads) run their backward there.  With gradient accumulation the
    weight gradient is consumed by an in-place add queued on the LAYER's stream, while the next layer -- on another stream --
    already launches its own weight gradient on the side stream: the block of the first must not be handed out again before
    that add has run (it was: test_dp_gpu's micro-batch reference then lost one contribution of the last layer of a stream)."""
    from mm_training_amd.ops import conv_overlap
    torch.manual_seed(0)
    ref = torch.nn.ModuleList([torch.nn.Conv2d(64, 64, 3, 1, 1, bias=False) for _ in range(6)]).cuda().to(memory_format=torch.channels_last)
    new = copy.deepcopy(ref)
    assert conv_overlap.enable(new, mode) == 6
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    main = torch.cuda.current_stream()
    for it in range(8):
        x = torch.randn(2, 64, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
        for nets, use_streams in ((ref, False), (new, True)):
            nets.zero_grad(set_to_none=True)
            for micro in range(3):                               # .grad is None only in the first
                xi = (x * (1 + micro)).requires_grad_(True)
                outs = []
                for i, conv in enumerate(nets):
                    if use_streams:
                        st = streams[i % 2]
                        st.wait_stream(main)
                        with torch.cuda.stream(st):
                            y = conv(xi)
                        y.record_stream(main)
                    else:
                        y = conv(xi)
                    outs.append(y)
                if use_streams:
                    for st in streams:
                        main.wait_stream(st)
                sum((o * (1 + i)).square().mean() for i, o in enumerate(outs)).backward()
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(ref, new)):
            err = float((a.weight.grad - b.weight.grad).abs().max()) / float(a.weight.grad.abs().max())
            assert err <= 1e-4, (mode, it, i, err)


def test_a_failed_backward_pass_does_not_cost_the_next_one_its_join(mmt_lib):
    """The end-of-backward join is queued once per backward pass.  A pass that dies with an exception never runs its callbacks;
    the next pass must queue its own (the bookkeeping is keyed by the pass, not a flag that would stay set)."""
    from mm_training_amd.ops import conv_overlap

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    torch.manual_seed(0)
    ref = torch.nn.Conv2d(32, 32, 3, 1, 1, bias=False).cuda().to(memory_format=torch.channels_last)
    new = copy.deepcopy(ref)
    conv_overlap.enable(new, "deferred")
    x = torch.randn(8, 32, 96, 96, device="cuda").contiguous(memory_format=torch.channels_last)
    with pytest.raises(RuntimeError, match="boom"):
        new(Boom.apply(x.clone().requires_grad_(True))).sum().backward()      # the convolution's backward ran, then the pass died
    for _ in range(5):
        new.zero_grad(set_to_none=True)
        ref.zero_grad(set_to_none=True)
        new(x).square().mean().backward()
        got = new.weight.grad.clone()                   # read on the caller's stream: only valid behind the join
        ref(x).square().mean().backward()
        assert float((got - ref.weight.grad).abs().max()) <= 1e-5 * float(ref.weight.grad.abs().max())


def test_narrow_16_bit_convolutions_compute_in_fp32(mmt_lib):
    """MIOpen's bf16 NHWC data-gradient kernel faults on `convbfp16 -n 4 -c 8 -H 16 -W 48 -k 8 -y 4 -x 4 -u 4 -v 4` (the tiny model's
    neck; ops/conv_overlap.py NARROW): inside an autocast region a convolution with fewer than 16 channels on both sides runs in
    fp32 on its fp32 operands (the result is rounded to bf16 once, and handed on as bf16 like autocast would); wider layers stay
    in bf16."""
    from mm_training_amd.ops import conv_overlap
    torch.manual_seed(0)
    for mode in ("inline", "deferred"):
        narrow = torch.nn.Conv2d(8, 8, 4, 4, 0, bias=False).cuda().to(memory_format=torch.channels_last)
        wide = torch.nn.Conv2d(8, 32, 3, 1, 1, bias=False).cuda().to(memory_format=torch.channels_last)
        conv_overlap.enable(narrow, mode)
        conv_overlap.enable(wide, mode)
        x = torch.randn(4, 8, 16, 48, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = narrow(x)
            z = wide(x)
        assert y.dtype == torch.bfloat16 and z.dtype == torch.bfloat16
        ref = torch.nn.functional.conv2d(x.detach(), narrow.weight.detach(), None, 4, 0)        # fp32 operands, not even rounded to bf16
        assert float((y.float() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max())       # one rounding of the result
        (y.float().square().sum() + z.float().square().sum()).backward()
        assert x.grad.dtype == torch.float32 and narrow.weight.grad.dtype == torch.float32 and torch.isfinite(x.grad).all()
        gy = (2 * y.detach().float())
        gw_ref = torch.nn.grad.conv2d_weight(x.detach(), narrow.weight.shape, gy, 4, 0)
        assert float((narrow.weight.grad - gw_ref).abs().max()) <= 1e-4 * float(gw_ref.abs().max())


@pytest.mark.parametrize("mode", ["inline", "deferred"])
def test_narrow_fp32_convolutions_bypass_miopen(mmt_lib, mode):
    """fp32 too: a convolution with fewer than 16 channels on both sides runs on ATen's own im2col + GEMM kernels (the fp32 sibling
    of MIOpen's faulting narrow kernel is what the tiny model would otherwise reach).  Reference: float64 on the CPU."""
    from mm_training_amd.ops import conv_overlap
    torch.manual_seed(1)
    for cin, cout, k, s, p, d, bias in ((8, 8, 4, 4, 0, 1, False), (8, 8, 3, 1, 1, 1, True), (12, 4, 3, 2, 1, 1, False), (8, 8, 3, 1, 2, 2, False)):
        conv = torch.nn.Conv2d(cin, cout, k, s, p, d, 1, bias).cuda().to(memory_format=torch.channels_last)
        ref = torch.nn.Conv2d(cin, cout, k, s, p, d, 1, bias).double()
        ref.load_state_dict({n: v.detach().cpu().double() for n, v in conv.state_dict().items()})
        conv_overlap.enable(conv, mode)
        x = torch.randn(4, cin, 16, 48, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        xr = x.detach().cpu().double().requires_grad_(True)
        y, yr = conv(x), ref(xr)
        assert y.dtype == torch.float32 and float((y.detach().cpu().double() - yr.detach()).abs().max()) <= 1e-5 * float(yr.abs().max())
        g = torch.randn_like(y)
        y.backward(g)
        yr.backward(g.cpu().double())
        for a, b in ((x.grad, xr.grad), (conv.weight.grad, ref.weight.grad)) + (((conv.bias.grad, ref.bias.grad),) if bias else ()):
            assert float((a.cpu().double() - b).abs().max()) <= 1e-5 * float(b.abs().max()), (cin, cout, k, s, d)


@pytest.mark.parametrize("mode", ["pair", "deferred"])
def test_deferred_mode_only_where_the_engine_does_nothing_with_the_gradient(mmt_lib, mode):
    """A weight used twice in one graph (the engine sums its two gradients in AccumulateGrad's input buffer, on the consumer's
    stream), a tensor hook on the weight, a gradient layout AccumulateGrad would not keep: each must make the layer join the side
    stream on the spot instead of deferring -- gradients equal to plain autograd's, many times over with the streams busy."""
    from mm_training_amd.ops import conv_overlap
    conv_overlap._uses.clear()                                    # (a test whose backward pass died may have left its forwards counted)
    torch.manual_seed(1)
    ref = torch.nn.Conv2d(32, 32, 3, 1, 1, bias=True).cuda().to(memory_format=torch.channels_last)
    new = copy.deepcopy(ref)
    conv_overlap.enable(new, mode)
    x = torch.randn(8, 32, 48, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    busy = torch.randn(4096, 4096, device="cuda")
    for it in range(6):
        grads = []
        for m in (ref, new):
            m.zero_grad(set_to_none=True)
            xi = x.clone().requires_grad_(True)
            (busy @ busy).sum()                                   # keep the main stream's queue long
            y = m(torch.relu(m(xi)))                              # the module applied twice: one weight, two gradients
            (y * y).mean().backward()
            grads.append((m.weight.grad.clone(), m.bias.grad.clone(), xi.grad.clone()))
        for a, b in zip(*grads):
            assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()), (mode, it)        # (two split-K weight gradients summed: fp32 atomics)
    assert not conv_overlap._uses                                 # the end-of-backward callback cleared the use counts
    # a hook on the weight: it sees (and here scales) the complete gradient
    new.zero_grad(set_to_none=True); ref.zero_grad(set_to_none=True)
    hooks = [m.weight.register_hook(lambda g: g * 2.0) for m in (ref, new)]
    for m in (ref, new):
        m(x).square().mean().backward()
    assert float((new.weight.grad - ref.weight.grad).abs().max()) <= 2e-5 * float(ref.weight.grad.abs().max())
    for h in hooks:
        h.remove()
    # the layout test on its own: a gradient whose strides differ from the weight's (where the size is not 1) is not deferred
    w = torch.nn.Parameter(torch.randn(8, 4, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last))
    assert conv_overlap._deferral_is_safe(w, torch.empty_like(w))
    assert not conv_overlap._deferral_is_safe(w, torch.empty(8, 4, 3, 3, device="cuda"))
    w1 = torch.nn.Parameter(torch.randn(8, 4, 1, 1, device="cuda"))                       # 1 x 1: every layout has the same strides where it matters
    assert conv_overlap._deferral_is_safe(w1, torch.empty(8, 4, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last))


def test_convolution_with_a_weight_that_is_the_cat_of_parameters(mmt_lib):
    """conv_overlap.conv2d(x, torch.cat(ws), leaves=ws) -- the task heads' first layer (layers/heads/bev_depth_head.py): every parameter
    receives a dim-0 view of the ONE weight gradient (its own layout), equal to the per-layer convolutions' gradients; the deferral
    applies per parameter (none while one of them holds a gradient already, i.e. gradient accumulation), whatever the mode."""
    import torch.nn.functional as F
    from mm_training_amd.ops import conv_overlap
    torch.manual_seed(0)
    ws = [torch.nn.Parameter(torch.randn(16, 32, 3, 3, device="cuda").mul_(0.1).contiguous(memory_format=torch.channels_last)) for _ in range(5)]
    x = torch.randn(2, 32, 24, 40, device="cuda").contiguous(memory_format=torch.channels_last)
    gy = torch.randn(2, 80, 24, 40, device="cuda").contiguous(memory_format=torch.channels_last)
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, torch.cat([w.detach() for w in ws], 0).requires_grad_(False), padding=1)          # (warm MIOpen)
    ref = [F.conv2d(xr, w, padding=1) for w in ws]
    torch.autograd.backward(ref, [gy[:, 16 * j:16 * j + 16] for j in range(5)])
    ref_gw, ref_gx = [w.grad.clone() for w in ws], xr.grad.clone()
    for mode in ("deferred", "pair", "inline"):
        for accumulate in (False, True):
            for w in ws:
                w.grad = None
            if accumulate:
                ws[2].grad = torch.ones_like(ws[2])                                             # one leaf holds a gradient already
            xi = x.clone().requires_grad_(True)
            weight = torch.cat(list(ws), 0)
            # the rule itself: safe only while every leaf would keep the view it is handed
            assert conv_overlap._deferral_is_safe(weight, torch.empty_like(weight), list(ws)) == (not accumulate)
            assert not conv_overlap._deferral_is_safe(weight, torch.empty(80, 32, 3, 3, device="cuda"), list(ws))   # (another layout)
            y = conv_overlap.conv2d(xi, weight, None, (1, 1), (1, 1), (1, 1), 1, mode, leaves=list(ws))
            y.backward(gy)
            torch.cuda.synchronize()
            assert float((xi.grad - ref_gx).abs().max()) <= 1e-4 * float(ref_gx.abs().max()), (mode, accumulate)
            for j, w in enumerate(ws):
                want = ref_gw[j] + (1.0 if (accumulate and j == 2) else 0.0)
                assert w.grad.shape == w.shape and w.grad.stride() == w.stride(), (mode, j)
                assert float((w.grad - want).abs().max()) <= 1e-4 * float(want.abs().max()), (mode, accumulate, j)
    assert not conv_overlap._uses
