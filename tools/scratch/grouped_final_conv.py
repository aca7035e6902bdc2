"""Feasibility: the task heads' 24 FINAL convolutions (64 -> 1..3, 3x3, one per branch) as ONE grouped convolution (groups=24, 3 output
channels per group, unused ones zero) on the wide [B, 1536, 128, 128] map.  usage: python tools/scratch/grouped_final_conv.py [f32|bf16] [B]"""
import sys
import torch
torch.backends.cudnn.benchmark = True
dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == "bf16" else torch.float32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
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


for (C, O, G, reps) in ((64, 2, 1, 24), (1536, 72, 24, 1), (1536, 96, 24, 1)):
    x = torch.randn(B, C, 128, 128, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
    w = torch.randn(O, C // G, 3, 3, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, O, 128, 128, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
    conv = lambda: torch.ops.aten.convolution(x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], G)
    bwd_d = lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], G, [True, False, False])
    bwd_w = lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], G, [False, True, False])
    try:
        t = [bench(f) for f in (conv, bwd_d, bwd_w)]
        print("%s B=%d C=%4d O=%3d groups=%2d x%2d: fwd %7.1f us, data grad %7.1f us, weight grad %7.1f us -> per step %7.1f / %7.1f / %7.1f us"
              % (str(dt)[6:], B, C, O, G, reps, t[0], t[1], t[2], t[0] * reps, t[1] * reps, t[2] * reps), flush=True)
    except Exception as e:
        print("failed", C, O, G, repr(e)[:200], flush=True)
