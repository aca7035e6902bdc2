"""Feasibility of fusing the task heads' 24 first convolutions (64 -> 64, 3x3, one shared [4, 64, 128, 128] input) into ONE 64 -> 1536
convolution: MIOpen's time for forward / data gradient / weight gradient of both forms (find mode, channels_last).
usage: python tools/scratch/wide_head_conv.py [f32|bf16]"""
import sys
import torch
torch.backends.cudnn.benchmark = True
dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == "bf16" else torch.float32
dev = torch.device("cuda", 0)


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, H, W) in ((4, 128, 128),):
    x = torch.randn(B, 64, H, W, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
    for O, reps in ((64, 24), (1536, 1)):
        w = torch.randn(O, 64, 3, 3, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(B, O, H, W, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
        conv = lambda: torch.ops.aten.convolution(x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)
        bwd_d = lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])
        bwd_w = lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])
        t = [bench(f) for f in (conv, bwd_d, bwd_w)]
        print("%s O=%4d x%2d: fwd %7.1f us, data grad %7.1f us, weight grad %7.1f us  -> per step %7.1f / %7.1f / %7.1f us"
              % (str(dt)[6:], O, reps, t[0], t[1], t[2], t[0] * reps, t[1] * reps, t[2] * reps), flush=True)
