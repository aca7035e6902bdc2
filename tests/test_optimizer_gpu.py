"""ClipAdamW (dp/optim.py, mmt_clip_adamw_step) against torch: clip_grad_norm_ + torch.optim.AdamW on the same parameters and
gradients, several steps, assorted sizes and layouts (channels_last weights, tiny tensors, a tensor longer than a chunk that does
not end on one, a parameter without a gradient), the state dictionaries interchangeable."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 32, 3, 3), (256, 64, 1, 1), (7,), (1,), (100003,), (70000, 3), (16, 8, 3, 3), (5, 5)]
    ps = []
    for i, sh in enumerate(shapes):
        t = torch.randn(sh, generator=g).cuda()
        if len(sh) == 4 and i != 6:
            t = t.contiguous(memory_format=torch.channels_last)
        ps.append(torch.nn.Parameter(t))
    return ps


def _grads(ps, seed, scale):
    g = torch.Generator().manual_seed(seed)
    out = []
    for p in ps:
        t = (torch.randn(p.shape, generator=g) * scale).cuda()
        out.append(t.contiguous(memory_format=torch.channels_last) if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous() else t)
    return out


@pytest.mark.parametrize("max_norm", [2.0, 0.0, 1e6])
def test_clip_adamw_matches_torch(mmt_lib, max_norm):
    from mm_training_amd.dp.optim import ClipAdamW
    a, b = _params(0), [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in _params(0)]
    kw = dict(lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    ours, ref = ClipAdamW(a, max_norm=max_norm, **kw), torch.optim.AdamW(b, fused=True, **kw)
    for step in range(6):
        gs = _grads(a, 10 + step, 0.05 if step % 2 else 3.0)          # (small: below the clip norm; large: clipped)
        for k, (p, q, g) in enumerate(zip(a, b, gs)):
            p.grad = None if k == 7 else g.clone(memory_format=torch.preserve_format)
            q.grad = None if k == 7 else g.clone(memory_format=torch.preserve_format)
        if max_norm > 0:
            total = torch.nn.utils.clip_grad_norm_([q for q in b if q.grad is not None], max_norm, foreach=True)
        ours.step()
        ref.step()
        if max_norm > 0:
            assert abs(float(ours.last_norm[0]) - float(total)) <= 1e-5 * float(total)
            assert abs(float(ours.last_norm[1]) - min(1.0, max_norm / (float(total) + 1e-6))) <= 1e-5
        for k, (p, q) in enumerate(zip(a, b)):
            assert p.stride() == q.stride()
            err = float((p - q).abs().max())
            assert err <= 2e-6 * max(1.0, float(q.abs().max())), (step, k, err)
            if k != 7:
                assert torch.equal(p.grad, gs[k]), "the gradients themselves are left unscaled"
                for name in ("exp_avg", "exp_avg_sq"):
                    e = float((ours.state[p][name] - ref.state[q][name]).abs().max())
                    assert e <= 2e-6 * max(1e-3, float(ref.state[q][name].abs().max())), (step, k, name, e)
    assert torch.equal(a[7], _params(0)[7]) and 7 not in [i for i, p in enumerate(a) if p in ours.state and "exp_avg" in ours.state[p]]
    # the state dictionaries are interchangeable
    sd = ours.state_dict()
    ref2 = torch.optim.AdamW([torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in a], fused=True, **kw)
    sd2 = copy.deepcopy(sd)
    # (the parameter without a gradient has no state on either side)
    ref2.load_state_dict(sd2)
    ours2 = ClipAdamW([torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in a], max_norm=max_norm, **kw)
    ours2.load_state_dict(ref.state_dict())
    assert int(ours2.state[ours2.param_groups[0]["params"][0]]["step"]) == 6


def test_training_step_with_either_optimizer(mmt_lib, monkeypatch):
    """TrainStep with the two-launch optimizer against torch's clip + fused AdamW (MMT_FUSED_OPT=0): same losses over a few steps of
    the tiny configuration to the tolerance of the convolutions' own run-to-run differences."""
    import numpy as np
    from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
    losses = []
    for fused in ("1", "0"):
        monkeypatch.setenv("MMT_FUSED_OPT", fused)
        torch.manual_seed(0)
        np.random.seed(0)
        cfg = make_config("tiny")
        ts = TrainStep(cfg, torch.device("cuda", 0))
        assert ts.fused_optimizer == (fused == "1")
        batches = [synthetic_batch(cfg, torch.device("cuda", 0), seed=i) for i in range(2)]
        ls = []
        for i in range(6):
            np.random.seed(100 + i)
            ls.append(float(ts(batches[i % 2])[0]))
        losses.append(ls)
    assert all(np.isfinite(losses[0])) and losses[0][-1] < losses[0][0]
    assert np.allclose(losses[0], losses[1], rtol=2e-2), losses


def test_steps_queued_without_a_host_sync(mmt_lib):
    """The host runs ahead of the device: many steps are queued behind a long kernel before the first executes, every one with
    freshly allocated gradients.  Each step's kernels must see ITS gradients' addresses (they are staged through a ring of pinned
    buffers guarded by events, not through one buffer that the next step overwrites)."""
    from mm_training_amd.dp.optim import ClipAdamW
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    a = [torch.nn.Parameter(torch.zeros(50000, device="cuda")), torch.nn.Parameter(torch.zeros(64, 16, 3, 3, device="cuda"))]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    ours, ref = ClipAdamW(a, max_norm=0.0, **kw), torch.optim.AdamW(b, fused=True, **kw)
    busy = torch.randn(8192, 8192, device="cuda")
    torch.cuda.synchronize()
    for _ in range(6):
        busy = busy @ busy * 1e-4                     # (the device is busy for a while: everything below is queued behind it)
    keep = []
    for step in range(12):
        for p, q in zip(a, b):
            g = torch.full_like(p, float(step + 1))   # a fresh allocation per step, distinct contents per step
            p.grad, q.grad = g, g.clone()
            keep.append(g)
        ours.step()
        ref.step()
        for p in a:
            p.grad = None                             # (the allocator may hand the block to the next step's gradient)
        keep.clear()
    torch.cuda.synchronize()
    for p, q in zip(a, b):
        assert float((p - q).abs().max()) <= 1e-6 * max(1.0, float(q.abs().max()))


def test_bf16_shadows_follow_the_parameters(mmt_lib):
    """make_bf16_shadows: after every step the bf16 copy of a parameter equals `p.to(bfloat16)` bit for bit (channels_last weights,
    odd lengths, a tensor without a copy next to them), and the autocast convolution uses the copy -- until somebody else writes the
    weight, which refreshes it."""
    from mm_training_amd.dp.optim import ClipAdamW
    from mm_training_amd.ops import conv_overlap
    ps = _params(3)
    opt = ClipAdamW(ps, lr=3e-3, max_norm=2.0)
    with_copy = [ps[0], ps[1], ps[4], ps[6], ps[7]]
    opt.make_bf16_shadows(with_copy)
    for step in range(4):
        for p, g in zip(ps, _grads(ps, 20 + step, 1.0)):
            p.grad = g
        opt.step()
        for p in with_copy:
            sh = opt.shadows[p]
            assert sh.stride() == p.stride() and sh.dtype == torch.bfloat16
            assert torch.equal(sh, p.detach().to(torch.bfloat16)), step
    w = ps[0]
    version = w._version
    assert conv_overlap._cast_weight(w, torch.bfloat16) is opt.shadows[w] and w._version == version
    assert conv_overlap._cast_weight(ps[2], torch.bfloat16).dtype == torch.bfloat16          # (no copy registered: a cast)
    assert conv_overlap._cast_weight(w, torch.float16).dtype == torch.float16
    with torch.no_grad():
        w.mul_(2.0)                                                                       # another writer: the copy is stale
    got = conv_overlap._cast_weight(w, torch.bfloat16)
    assert got is opt.shadows[w] and torch.equal(got, w.detach().to(torch.bfloat16))
    for p, g in zip(ps, _grads(ps, 99, 1.0)):
        p.grad = g
    opt.step()
    assert torch.equal(conv_overlap._cast_weight(w, torch.bfloat16), w.detach().to(torch.bfloat16))
    # ADVICE (round 5): writers the version counter does not see.  A REPLACED storage (`p.data = ..`) is caught by its address -- by the
    # copy (refreshed) and by the optimizer (its pointer tables are rebuilt: the step lands in the new storage, the old one is left alone);
    # an in-place write through `.data` moves neither counter nor address: refresh_bf16_shadows() is the documented call for it
    old_storage = w.data
    keep = old_storage.clone()
    w.data = (old_storage * 0.5).contiguous(memory_format=torch.channels_last) if w.dim() == 4 else old_storage * 0.5
    assert w._version == version + 1 or True                                               # (whatever the counter did)
    got = conv_overlap._cast_weight(w, torch.bfloat16)
    assert torch.equal(got, w.detach().to(torch.bfloat16))
    for p, g in zip(ps, _grads(ps, 100, 1.0)):
        p.grad = g
    before = w.detach().clone()
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(old_storage, keep) and not torch.equal(w.detach(), before)          # written through the NEW address
    assert torch.equal(conv_overlap._cast_weight(w, torch.bfloat16), w.detach().to(torch.bfloat16))
    w.data.mul_(3.0)                                                                       # same storage, no counter: invisible ...
    conv_overlap.refresh_bf16_shadows()                                                    # ... until the caller says so
    assert torch.equal(conv_overlap._cast_weight(w, torch.bfloat16), w.detach().to(torch.bfloat16))


def test_step_counts_that_differ_inside_a_group_are_refused(mmt_lib):
    """torch.optim.AdamW de-biases every parameter with its own step count; ClipAdamW's one launch uses one count for the group.  A
    parameter that starts receiving gradients later would be de-biased with the wrong power of beta: refused when the tables are
    rebuilt for the new participating set, instead of diverging from torch silently."""
    from mm_training_amd.dp.optim import ClipAdamW
    ps = _params(5)
    opt = ClipAdamW(ps, lr=1e-3)
    for step in range(2):
        for p, g in zip(ps[:-1], _grads(ps[:-1], 40 + step, 1.0)):     # the last parameter gets no gradient yet
            p.grad = g
        opt.step()
    for p, g in zip(ps, _grads(ps, 50, 1.0)):
        p.grad = g
    with pytest.raises(RuntimeError, match="different step counts"):
        opt.step()


def test_bf16_training_step_with_and_without_shadows(mmt_lib, monkeypatch):
    """The bf16-autocast training step with the optimizer-maintained weight copies against the same step casting per layer
    (MMT_BF16_SHADOWS=0): the convolutions see the same bf16 weights either way."""
    import numpy as np
    from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
    losses = []
    for shadows in ("1", "0"):
        monkeypatch.setenv("MMT_BF16_SHADOWS", shadows)
        torch.manual_seed(0)
        np.random.seed(0)
        cfg = make_config("tiny")
        ts = TrainStep(cfg, torch.device("cuda", 0), amp="bf16")
        assert ts.amp_dtype is torch.bfloat16
        assert bool(ts.optimizer.shadows) == (shadows == "1")
        batches = [synthetic_batch(cfg, torch.device("cuda", 0), seed=i) for i in range(2)]
        ls = []
        for i in range(6):
            np.random.seed(100 + i)
            ls.append(float(ts(batches[i % 2])[0]))
        losses.append(ls)
    assert all(np.isfinite(losses[0])) and losses[0][-1] < losses[0][0]
    assert np.allclose(losses[0], losses[1], rtol=3e-2), losses
