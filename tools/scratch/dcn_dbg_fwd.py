import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_training_amd.ops.deform_conv import deform_conv3x3
B, C, H, W, O, groups = (24, 512, 16, 44, 512, 4)
torch.manual_seed(0)
x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
off = (torch.randn(B, 18, H, W, device="cuda") * 0.5)
w = (torch.randn(O, C // groups, 3, 3, device="cuda") * 0.05)
for cfg in (2, 4):
    os.environ["MMT_DCN_FWD_CONFIG"] = str(cfg)
    for dbg in (0, 1, 2, 4, 8, 12, 14):
        os.environ["MMT_DCN_FWD_DEBUG"] = str(dbg)
        for _ in range(3):
            deform_conv3x3(x, off, w, groups)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            deform_conv3x3(x, off, w, groups)
        e.record(); torch.cuda.synchronize()
        print("cfg", cfg, "dbg", dbg, "fwd ms", s.elapsed_time(e) / 10, flush=True)
