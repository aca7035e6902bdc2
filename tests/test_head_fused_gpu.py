"""GPU: the task heads' 24 first ConvModules as ONE wide convolution + ONE BatchNorm (layers/heads/bev_depth_head.py::
_forward_tasks_fused) against the per-branch modules: the same outputs, input gradient, parameter gradients and running statistics to
the convolutions' rounding (another MIOpen kernel sums the same products in another order), the same `state_dict`, and the two copies
(`mmt_channel_blocks_split` / `_gather`) bit-exact against torch."""
import copy
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _head(cfg):
    from mm_training_amd.layers.heads.bev_depth_head import BEVDepthHead
    torch.manual_seed(0)
    head = BEVDepthHead(**cfg["head_conf"]).cuda()
    for m in head.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            m.to(memory_format=torch.channels_last)
    return head


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_channel_blocks_split_and_gather(mmt_lib, dtype):
    from mm_training_amd import _lib
    for (B, H, W, n, w) in ((2, 5, 7, 24, 64), (1, 3, 3, 3, 8), (4, 16, 16, 32, 16)):
        wide = torch.randn(B, n * w, H, W, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
        parts = [torch.empty(B, w, H, W, device="cuda", dtype=dtype).contiguous(memory_format=torch.channels_last) for _ in range(n)]
        arr = (ctypes.c_void_p * n)(*[t.data_ptr() for t in parts])
        _lib.call("mmt_channel_blocks_split", B * H * W, n, w * wide.element_size(), wide.data_ptr(), arr, 0)
        for k, t in enumerate(parts):
            assert torch.equal(t, wide[:, k * w:(k + 1) * w])
        back = torch.empty_like(wide)
        _lib.call("mmt_channel_blocks_gather", B * H * W, n, w * wide.element_size(), arr, back.data_ptr(), 0)
        assert torch.equal(back, wide)
    with pytest.raises(_lib.MmtError):
        _lib.call("mmt_channel_blocks_split", 4, 33, 64, wide.data_ptr(), arr, 0)
    with pytest.raises(_lib.MmtError):
        _lib.call("mmt_channel_blocks_split", 4, 2, 24, wide.data_ptr(), arr, 0)


def test_fused_branch_stems_equal_the_per_branch_modules(mmt_lib):
    from mm_training_amd.dp import make_config
    cfg = make_config("tiny")
    per_branch = _head(cfg)
    fused = copy.deepcopy(per_branch)
    per_branch.fuse_branch_stems, per_branch.task_streams = False, 0
    fused.fuse_branch_stems = True
    assert list(per_branch.state_dict()) == list(fused.state_dict())

    def run(head, x):
        head.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        if head.fuse_branch_stems:
            stems = head._branch_stems(xi)
            assert stems is not None and len(stems) == 24
            out = head._forward_tasks_fused(xi, stems)
        else:
            out = tuple([task(xi)] for task in head.task_heads)
        assert len(out) == 4 and set(out[0][0]) == {"reg", "height", "dim", "rot", "vel", "heatmap"}
        loss = sum((v.float() * (1 + i)).square().mean() for i, task in enumerate(out) for v in task[0].values())
        loss.backward()
        flat = {"out%d.%s" % (i, k): v.detach() for i, task in enumerate(out) for k, v in task[0].items()}
        flat["grad_x"] = xi.grad.clone()
        flat.update({"grad." + n: p.grad.clone() for n, p in head.named_parameters() if p.grad is not None})
        return flat

    worst = []
    for step in range(6):
        x = torch.randn(2, 64, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
        a, b = run(per_branch, x), run(fused, x)
        assert set(a) == set(b) and len(a) == 24 + 1 + 24 * 5
        rel = {k: float((a[k] - b[k]).abs().max()) / (float(a[k].abs().max()) + 1e-6) for k in a}
        for k in a:
            assert a[k].shape == b[k].shape
            if k.startswith("out"):
                assert rel[k] <= 2e-4, (step, k, rel[k])
        # gradients: the wide convolution sums the same products in another order, so of the 50 M pre-activations the few dozen that
        # are zero to the last bits open their ReLU on one path and not on the other -- single elements of a gradient move by their
        # whole value (the per-branch path against ITSELF shows the same, tests/test_head_streams_gpu.py), the tensors as a whole
        # do not: the bar is on the relative L2 distance, with a loose one on single elements
        l2 = {k: float((a[k].double() - b[k].double()).norm() / (a[k].double().norm() + 1e-12)) for k in a if not k.startswith("out")}
        assert max(l2.values()) <= 2e-2 and sorted(l2.values())[len(l2) // 2] <= 1e-4, (step, max(l2.items(), key=lambda kv: kv[1]))
        worst.append(max(rel.values()))
    assert max(worst) <= 0.25, worst
    # the running statistics moved identically, and stay the modules' own buffers under their own names
    sa, sb = per_branch.state_dict(), fused.state_dict()
    assert list(sa) == list(sb)
    for n in sa:
        if n.endswith("running_mean") or n.endswith("running_var"):
            assert torch.allclose(sa[n], sb[n], rtol=1e-4, atol=1e-6), n
    # every parameter's gradient is a tensor of the parameter's own layout (views of the one wide gradient)
    for n, p in fused.named_parameters():
        if p.grad is not None:
            assert p.grad.shape == p.shape and p.grad.stride() == p.stride(), n
    # the module's forward takes the fused path in training, the per-branch modules under no_grad / in eval mode; moving the module
    # (which gives every BatchNorm separate buffers again) re-establishes the shared buffer at the next step
    c = cfg["fuse_layer_in_channels"]
    full = torch.randn(2, c, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
    fused = fused.float().cuda()
    out = fused(full)
    sum(v.sum() for task in out for v in task[0].values()).backward()
    bns = [getattr(t, h)[0][1] for t in fused.task_heads for h in t.heads]
    assert bns[1].running_mean.data_ptr() == bns[0].running_mean.data_ptr() + 4 * bns[0].num_features
    fused.load_state_dict(sa)
    assert torch.equal(fused.state_dict()["task_heads.0.reg.0.1.running_mean"], sa["task_heads.0.reg.0.1.running_mean"])
    assert torch.equal(fused._stem_stats[0][:bns[0].num_features], bns[0].running_mean)
    fused.eval()
    with torch.no_grad():
        assert len(fused(full)) == 4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_final_convolutions_in_one_launch(mmt_lib, dtype):
    """_FinalConvs (mmt_heads_final_forward / _backward) against F.conv2d per branch: outputs, the gradient of the wide map, every
    branch's weight and bias gradient; widths that are not a multiple of the 32-pixel segment, one to four output channels."""
    import torch.nn.functional as F
    from mm_training_amd.layers.heads.bev_depth_head import _FinalConvs
    torch.manual_seed(0)
    for (B, H, W, ks) in ((2, 9, 37, (2, 1, 3, 2, 2, 1)), (1, 4, 5, (4,)), (2, 16, 64, (1, 2, 3, 4) * 6), (1, 3, 130, (3, 1) * 16)):
        n, kt = len(ks), sum(ks)
        wide = (torch.randn(B, n * 64, H, W, device="cuda") * 0.5).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ws = [torch.randn(k, 64, 3, 3, device="cuda").mul_(0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True) for k in ks]
        bs = [torch.randn(k, device="cuda").requires_grad_(True) for k in ks]
        gos = [torch.randn(B, k, H, W, device="cuda").to(dtype).float() for k in ks]          # (representable in the activation type)
        # reference: fp32 convolutions per branch on the (rounded) inputs
        ref_in = wide.detach().float().requires_grad_(True)
        refs = [F.conv2d(ref_in[:, j * 64:(j + 1) * 64], w, b, padding=1) for j, (w, b) in enumerate(zip(ws, bs))]
        torch.autograd.backward(refs, gos)
        ref_gw, ref_gb = [w.grad.clone() for w in ws], [b.grad.clone() for b in bs]
        for t in ws + bs:
            t.grad = None
        weight = torch.cat(ws, 0).contiguous(memory_format=torch.channels_last)
        assert weight.stride() == (576, 1, 192, 64)
        outs = _FinalConvs.apply(wide, weight, torch.cat(bs), tuple(ks))
        whole, outs = outs[-1], outs[:-1]                      # (the branches' outputs, and the whole narrow map they are slices of)
        assert whole.shape == (B, kt, H, W) and all(o.data_ptr() == whole[:, sum(ks[:j]):].data_ptr() for j, o in enumerate(outs))
        assert len(outs) == n and all(o.shape == r.shape and o.dtype == dtype for o, r in zip(outs, refs))
        fwd_tol = 2e-5 if dtype is torch.float32 else 2e-2
        for j, (o, r) in enumerate(zip(outs, refs)):
            assert float((o.detach().float() - r.detach()).abs().max()) <= fwd_tol * max(1.0, float(r.detach().abs().max())), (ks, j)
        torch.autograd.backward(outs, [g.to(dtype) for g in gos])
        bwd_tol = 1e-4 if dtype is torch.float32 else 3e-2
        assert float((wide.grad.float() - ref_in.grad).abs().max()) <= bwd_tol * max(1.0, float(ref_in.grad.abs().max()))
        for j in range(n):
            assert ws[j].grad.shape == ws[j].shape and ws[j].grad.stride() == ws[j].stride()
            assert float((ws[j].grad - ref_gw[j]).abs().max()) <= bwd_tol * max(1.0, float(ref_gw[j].abs().max())), (ks, j)
            assert float((bs[j].grad - ref_gb[j]).abs().max()) <= bwd_tol * max(1.0, float(ref_gb[j].abs().max())), (ks, j)
        # a branch that was not used hands back no gradient: zeros for it, the others unchanged
        wide.grad = None
        outs = _FinalConvs.apply(wide, weight.detach(), torch.cat(bs).detach(), tuple(ks))[:-1]
        outs[0].float().mul(gos[0]).sum().backward()
        only0 = torch.autograd.grad(F.conv2d(ref_in[:, :64], ws[0].detach(), bs[0].detach(), padding=1).mul(gos[0]).sum(), ref_in)[0]
        assert float((wide.grad.float() - only0).abs().max()) <= bwd_tol * max(1.0, float(only0.abs().max()))


@pytest.mark.parametrize("amp", [False, True])
def test_fused_head_loss_equals_the_torch_loss(mmt_lib, amp):
    """BEVDepthHead.loss on the fused heads' one output map (mmt_head_loss_forward_backward) against the same method's torch ops on
    the same predictions: the value, and the gradients of the shared map and of every parameter; with NaN targets, two boxes on one
    pixel, an empty task, and under autocast (bf16 map)."""
    import copy
    from mm_training_amd.dp import make_config
    cfg = make_config("tiny")
    head = _head(cfg)
    head.fuse_branch_stems = True
    torch.manual_seed(1)
    B = 2
    boxes, labels = [], []
    for b in range(B):
        k = 12
        xy = torch.rand(k, 2) * 80 - 40
        bx = torch.cat([xy, torch.rand(k, 1) * 2 - 2, torch.rand(k, 3) * 3 + 0.5, torch.rand(k, 1) * 6 - 3, torch.randn(k, 2)], 1)
        bx[1, :2] = bx[0, :2]                                            # two boxes of one class on one pixel
        lab = torch.randint(0, 3, (k,))                                   # (class 3 = the fourth task stays empty)
        lab[1] = lab[0]
        boxes.append(bx.cuda()); labels.append(lab.cuda())
    targets = head.get_targets(boxes, labels)
    targets[1][0][0, 0, 7:9] = float("nan")                               # NaN velocity targets carry no weight (bev_depth_head.py:296)
    x = torch.randn(B, 64, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
    results = []
    for fused in (True, False):
        h = copy.deepcopy(head)
        h.fuse_loss = fused
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            preds = h._forward_tasks_fused(xi, h._branch_stems(xi))
            assert (h._fused_loss_map(preds, *targets) is not None) == fused
        loss = h.loss(targets, preds)
        loss.backward()
        results.append((float(loss), xi.grad.clone(), {n: p.grad.clone() for n, p in h.named_parameters() if p.grad is not None}))
    (la, ga, pa), (lb, gb, pb) = results
    tol = 2e-2 if amp else 1e-4
    assert abs(la - lb) <= (2e-3 if amp else 1e-5) * abs(lb), (la, lb)
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-12))
    assert rel(ga, gb) <= tol, rel(ga, gb)
    assert set(pa) == set(pb) and len(pa) == 24 * 5
    worst = max(rel(pa[n], pb[n]) for n in pa)
    assert worst <= (5e-2 if amp else 2e-3), worst
