#!/usr/bin/env python3
"""Micro-benchmark of the fused BatchNorm(+residual)(+ReLU) kernels against nn.BatchNorm2d + add + relu
at the ResNet-50 activation shapes of the cfg2 step (run on the GPU box)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch import nn  # noqa: E402

from mm_training_amd.ops import bn_relu  # noqa: E402

SHAPES = [(24, 64, 64, 176), (24, 256, 64, 176), (24, 128, 32, 88), (24, 512, 32, 88), (24, 256, 16, 44),
          (24, 1024, 16, 44), (24, 512, 8, 22), (24, 2048, 8, 22), (4, 160, 64, 64), (4, 64, 128, 128)]


BF16 = "--bf16" in sys.argv          # bf16 activations inside autocast (mmt_bn_relu_*_ex) at the configs[4] shapes; fused kernels only
if BF16:
    SHAPES = [(12, 64, 128, 352), (12, 256, 128, 352), (12, 128, 64, 176), (12, 512, 64, 176), (12, 256, 32, 88), (12, 1024, 32, 88),
              (12, 512, 16, 44), (12, 2048, 16, 44), (2, 160, 64, 64), (2, 64, 128, 128)]


def run(fused, shape, use_res, reps=10):
    B, C, H, W = shape
    bn = nn.BatchNorm2d(C).cuda()
    dt = torch.bfloat16 if BF16 else torch.float32
    x = torch.randn(shape, device="cuda").to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn(shape, device="cuda").to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True) if use_res else None
    go = torch.randn(shape, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    bn_relu.ENABLED = fused
    flush = torch.empty(128 * 1024 * 1024, device="cuda")
    tf, tb = [], []
    for i in range(reps + 3):
        flush.zero_()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=BF16):
            y = bn_relu.bn_act(bn, x, r, True)
        e[1].record()
        flush.zero_()
        e[2].record()
        y.backward(go)
        e[3].record()
        x.grad = None
        bn.zero_grad(set_to_none=True)
        if r is not None:
            r.grad = None
        torch.cuda.synchronize()
        if i >= 3:
            tf.append(e[0].elapsed_time(e[1]))
            tb.append(e[2].elapsed_time(e[3]))
    med = lambda v: sorted(v)[len(v) // 2]
    return med(tf) * 1e3, med(tb) * 1e3


res = []
for shape in SHAPES:
    for use_res in (False, True):
        n = (2 if BF16 else 4) * shape[0] * shape[1] * shape[2] * shape[3] / 1e6          # MB per pass
        ff, fb = run(True, shape, use_res)
        uf, ub = (0.0, 0.0) if BF16 else run(False, shape, use_res)     # (MIOpen's NHWC batch norm under bf16: DESIGN 3.5b)
        fwd_passes, bwd_passes = (4 if use_res else 3), (8 if use_res else 5)    # reads + writes of the fused kernels
        res.append(dict(shape=shape, residual=use_res, MB_per_pass=round(n, 1), fused_fwd_us=round(ff, 1), fused_bwd_us=round(fb, 1),
                        torch_fwd_us=round(uf, 1), torch_bwd_us=round(ub, 1),
                        fused_fwd_TBps=round(fwd_passes * n / ff, 2), fused_bwd_TBps=round(bwd_passes * n / fb, 2)))
        print(json.dumps(res[-1]), flush=True)
