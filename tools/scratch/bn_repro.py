import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch import nn
from mm_training_amd.ops import bn_relu
shape = tuple(int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (1, 320, 31, 15)
use_res, relu = True, False
for trial in range(20):
    x = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    go = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last)
    bn = nn.BatchNorm2d(shape[1]).cuda()
    print("supported", bn_relu._supported(bn, x), flush=True)
    y = bn_relu.bn_act(bn, x, r, relu); torch.cuda.synchronize(); print("fwd ok", flush=True)
    y.backward(go); torch.cuda.synchronize(); print("bwd ok", trial, flush=True)
