#!/usr/bin/env python3
"""Kernel durations of the BEV warp (forward / backward) at BASELINE configs[3]'s map [4, 128, 128, 80], written into / read from the
160-channel camera | LiDAR buffer as the step does: python tools/scratch/warp_time.py [lib.so ...] (interleaved A/B)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_training_amd import _lib
from tools.kbench_camera import load
libs = sys.argv[1:] or [_lib.LIB_PATH]
_lib.lib()
hs = [load(p) for p in libs]
B, H, W, C, S = 4, 128, 128, 80, 160
g = torch.Generator().manual_seed(0)
ang = torch.tensor([0.3, -0.2, 0.1, 0.0])
bda = torch.eye(4).repeat(B, 1, 1)
bda[:, 0, 0] = torch.cos(ang) * 1.05; bda[:, 0, 1] = -torch.sin(ang) * 1.05; bda[:, 1, 0] = torch.sin(ang) * 1.05; bda[:, 1, 1] = torch.cos(ang) * 1.05
bda = bda.cuda().contiguous()
x = torch.randn(B, H, W, C, generator=g).cuda()
y = torch.zeros(B, H, W, S, device="cuda")
gx = torch.zeros(B, H, W, C, device="cuda")
st = torch.cuda.current_stream().cuda_stream
res = {}
for rnd in range(3):
    for p, h in zip(libs, hs):
        for name, fn in (("fwd", lambda: h.mmt_bev_warp_affine(B, H, W, C, bda.data_ptr(), x.data_ptr(), C, y.data_ptr(), S, st)),
                         ("bwd", lambda: h.mmt_bev_warp_affine_backward(B, H, W, C, bda.data_ptr(), y.data_ptr(), S, gx.data_ptr(), C, st)),
                         ("bwd_assign", lambda: h.mmt_bev_warp_affine_backward_assign(B, H, W, C, bda.data_ptr(), y.data_ptr(), S, gx.data_ptr(), C, st) if hasattr(h, "mmt_bev_warp_affine_backward_assign") else 0)):
            for _ in range(5): assert fn() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, {}).setdefault(os.path.basename(p), []).append(round(e0.elapsed_time(e1) * 50, 2))
print(res)
