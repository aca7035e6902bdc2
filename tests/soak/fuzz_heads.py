"""GPU soak: the task heads' hand-written kernels on random shapes -- `mmt_heads_final_forward / _backward` (csrc/thin_conv.hip) against
F.conv2d per branch, `mmt_channel_blocks_split / _gather` and `mmt_bn_relu_inference` against torch.  Guard pages are not available here;
what this catches is a wrong index at a segment / row / tensor end (values), and a fault.
usage: python tests/soak/fuzz_heads.py [seconds] [seed]"""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from mm_training_amd.layers.heads.bev_depth_head import _FinalConvs, _SplitBlocks          # noqa: E402
from mm_training_amd.ops.bn_relu import bn_act                                             # noqa: E402


def one_final(g, dtype):
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    B, H, W, n = r(1, 3), r(1, 20), r(1, 70), r(1, 32)
    ks = tuple(r(1, 4) for _ in range(n))
    wide = (torch.randn(B, n * 64, H, W, generator=g).cuda() * 0.5).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ws = [torch.randn(k, 64, 3, 3, generator=g).cuda().mul_(0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True) for k in ks]
    bs = [torch.randn(k, generator=g).cuda().requires_grad_(True) for k in ks]
    gos = [torch.randn(B, k, H, W, generator=g).cuda().to(dtype).float() for k in ks]
    ref_in = wide.detach().float().requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=False):           # (ATen's own convolution: MIOpen refuses some of these degenerate shapes)
        refs = [F.conv2d(ref_in[:, j * 64:(j + 1) * 64].contiguous(), w.contiguous(), b, padding=1) for j, (w, b) in enumerate(zip(ws, bs))]
    with torch.backends.cudnn.flags(enabled=False):
        torch.autograd.backward(refs, gos)
    ref_gw, ref_gb = [w.grad.clone() for w in ws], [b.grad.clone() for b in bs]
    for t in ws + bs:
        t.grad = None
    weight = torch.cat(ws, 0).contiguous(memory_format=torch.channels_last)
    outs = _FinalConvs.apply(wide, weight, torch.cat(bs), ks)[:-1]          # (the last output is the whole map)
    ft, bt = (3e-5, 2e-4) if dtype is torch.float32 else (2e-2, 3e-2)
    for o, rf in zip(outs, refs):
        assert float((o.detach().float() - rf.detach()).abs().max()) <= ft * max(1.0, float(rf.detach().abs().max())), ("fwd", B, H, W, ks)
    torch.autograd.backward(outs, [x.to(dtype) for x in gos])
    assert float((wide.grad.float() - ref_in.grad).abs().max()) <= bt * max(1.0, float(ref_in.grad.abs().max())), ("gz", B, H, W, ks)
    for j in range(n):
        assert float((ws[j].grad - ref_gw[j]).abs().max()) <= bt * max(1.0, float(ref_gw[j].abs().max())), ("gw", B, H, W, ks, j)
        assert float((bs[j].grad - ref_gb[j]).abs().max()) <= bt * max(1.0, float(ref_gb[j].abs().max())), ("gb", B, H, W, ks, j)


def one_split(g, dtype):
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    B, H, W, n, w = r(1, 3), r(1, 12), r(1, 12), r(1, 32), 8 * r(1, 8)
    wide = torch.randn(B, n * w, H, W, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    parts = _SplitBlocks.apply(wide, n)
    for k, t in enumerate(parts):
        assert torch.equal(t, wide.detach()[:, k * w:(k + 1) * w]), ("split", B, H, W, n, w)
    gs = [torch.randn_like(t) for t in parts]
    torch.autograd.backward(parts, gs)
    assert torch.equal(wide.grad, torch.cat(gs, 1)), ("gather", B, H, W, n, w)


def one_bn(g, dtype):
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    C = [4 * r(1, 64), 256 * r(1, 8)][r(0, 1)]
    B, H, W = r(1, 3), r(1, 9), r(1, 9)
    bn = torch.nn.BatchNorm2d(C).cuda().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    for p in bn.parameters():
        p.requires_grad = False
    x = torch.randn(B, C, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    ref = torch.relu(bn(x))
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype is torch.bfloat16):
        y = bn_act(bn, x.to(dtype), relu=True)
    tol = 1e-5 if dtype is torch.float32 else 6e-2
    assert float((y.float() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max())), ("bn", B, C, H, W)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0, n = time.time(), 0
    while time.time() - t0 < seconds:
        dtype = torch.float32 if n % 2 == 0 else torch.bfloat16
        one_final(g, dtype)
        one_split(g, dtype)
        one_bn(g, dtype)
        n += 1
    torch.cuda.synchronize()
    print("fuzz ok: %d random configurations each of the final convolutions / channel-block copies / eval-mode BatchNorm" % n)


if __name__ == "__main__":
    main()
