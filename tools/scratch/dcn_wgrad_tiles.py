import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_training_amd.ops.deform_conv import deform_conv3x3
B, C, H, W, O, groups = (24, 512, 16, 44, 512, 4)
torch.manual_seed(0)
x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
off = (torch.randn(B, 18, H, W, device="cuda") * 0.5).requires_grad_(True)
w = (torch.randn(O, C // groups, 3, 3, device="cuda") * 0.05).requires_grad_(True)
go = torch.randn(B, O, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
out = deform_conv3x3(x, off, w, groups)
for tile in (0, 1, 2, 3):
    for slots in (512, 768, 1024):
        os.environ["MMT_DCN_WGRAD_TILE"] = str(tile); os.environ["MMT_DCN_WGRAD_SLOTS"] = str(slots)
        for _ in range(2):
            torch.autograd.grad(out, (x, off, w), go, retain_graph=True)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            torch.autograd.grad(out, (x, off, w), go, retain_graph=True)
        e.record(); torch.cuda.synchronize()
        print("tile", tile, "slots", slots, "bwd ms", s.elapsed_time(e) / 5, flush=True)
