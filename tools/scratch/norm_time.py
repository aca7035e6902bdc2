#!/usr/bin/env python3
"""Back-to-back launch time of mmt_normalize_flip_images at BASELINE configs[3]'s images [4*6, 3, 256, 704] (A/B: lib.so ...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctypes
from mm_training_amd import _lib
from tools.kbench_camera import load
libs = sys.argv[1:] or [_lib.LIB_PATH]
_lib.lib()
hs = [load(p) for p in libs]
n, H, W = 24, 256, 704
x = torch.rand(n, 3, H, W, device="cuda") * 255
fl = (torch.arange(n) % 2).to(torch.uint8).cuda()
mean = (ctypes.c_float * 3)(0.485, 0.456, 0.406); std = (ctypes.c_float * 3)(0.229, 0.224, 0.225)
st = torch.cuda.current_stream().cuda_stream
res = {}
for cl in (0, 1):
    out = torch.empty(n, 3, H, W, device="cuda")
    for rnd in range(3):
        for p, h in zip(libs, hs):
            fn = lambda: h.mmt_normalize_flip_images(n, 3, H, W, x.data_ptr(), 1.0 / 255.0, mean, std, fl.data_ptr(), out.data_ptr(), cl, st)
            for _ in range(5): assert fn() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            res.setdefault("nhwc" if cl else "nchw", {}).setdefault(os.path.basename(p), []).append(round(e0.elapsed_time(e1) * 50, 2))
print(res)
