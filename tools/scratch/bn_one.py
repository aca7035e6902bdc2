import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch import nn
from mm_training_amd.ops import bn_relu
shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]] or [(24, 2048, 8, 22)]
for shape in shapes:
    for res in (False, True):
        bn = nn.BatchNorm2d(shape[1]).cuda()
        x = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        r = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True) if res else None
        go = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last)
        for i in range(12):
            y = bn_relu.bn_act(bn, x, r, True); y.backward(go); x.grad = None
            if r is not None: r.grad = None
        torch.cuda.synchronize()
