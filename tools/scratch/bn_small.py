"""Fused BatchNorm backward / forward on the small-R layers (ResNet-50 stage 4 at BASELINE configs[3]): per-kernel durations via events."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch import nn
from mm_training_amd.ops import bn_relu
for shape, res in (((24, 2048, 8, 22), False), ((24, 2048, 8, 22), True), ((24, 512, 8, 22), False), ((24, 512, 8, 22), True), ((24, 1024, 16, 44), False), ((24, 256, 16, 44), False)):
    out = {}
    for fused in (True, False):
        bn_relu.ENABLED = fused
        bn = nn.BatchNorm2d(shape[1]).cuda()
        x = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        r = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True) if res else None
        go = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last)
        tf, tb = [], []
        for i in range(60):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record(); y = bn_relu.bn_act(bn, x, r, True); e[1].record()
            e[2].record(); y.backward(go); e[3].record()
            torch.cuda.synchronize()
            if i >= 10:
                tf.append(e[0].elapsed_time(e[1]) * 1e3); tb.append(e[2].elapsed_time(e[3]) * 1e3)
            x.grad = None
            if r is not None: r.grad = None
        tf.sort(); tb.sort()
        out["fused" if fused else "torch"] = (round(tf[len(tf) // 2], 1), round(tb[len(tb) // 2], 1))
    print(shape, "res" if res else "plain", out, flush=True)
