"""rocprofv3 target: the whole DCN operator at the DepthNet shape, implicit-GEMM form, a few forward + backward calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_training_amd.ops.deform_conv import deform_conv3x3
cfg5 = "--cfg5" in sys.argv
B, C, H, W, O, groups = (12, 512, 32, 88, 512, 4) if cfg5 else (24, 512, 16, 44, 512, 4)
torch.manual_seed(0)
x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
off = (torch.randn(B, 18, H, W, device="cuda") * 0.5).requires_grad_(True)
w = (torch.randn(O, C // groups, 3, 3, device="cuda") * 0.05).requires_grad_(True)
go = torch.randn(B, O, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
for _ in range(6):
    out = deform_conv3x3(x, off, w, groups, columns="--columns" in sys.argv)
    torch.autograd.grad(out, (x, off, w), go)
torch.cuda.synchronize()
