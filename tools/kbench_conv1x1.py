"""1x1 stride-1 convolutions of the ResNet-50 image backbone at BASELINE configs[3]'s shape (24 images of 256x704, fp32,
channels_last): MIOpen's implicit-GEMM solver (what nn.Conv2d runs, with the shipped find DB) against the same contraction
as ONE plain GEMM on the [N*H*W, Cin] channels_last matrix (hipBLASLt / rocBLAS through torch.matmul).  Forward + backward
(data + weight gradients), HIP events around 20 repetitions."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

bench.use_shipped_miopen_db()
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.backends.cudnn.benchmark = True
SHAPES = [(64, 176, 64, 64), (64, 176, 64, 256), (64, 176, 256, 64), (32, 88, 256, 128), (32, 88, 128, 512), (32, 88, 512, 128),
          (16, 44, 512, 256), (16, 44, 256, 1024), (16, 44, 1024, 256), (8, 22, 1024, 512), (8, 22, 512, 2048), (8, 22, 2048, 512),
          (16, 44, 512, 512), (128, 128, 144, 144)]
N = 24


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


out = []
for H, W, ci, co in SHAPES:
    n = 4 if H == 128 else N
    x = torch.randn(n, ci, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(co, ci, 1, 1, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = torch.randn(n, co, H, W, device="cuda").contiguous(memory_format=torch.channels_last)

    def conv():
        x.grad = w.grad = None
        F.conv2d(x, w).backward(g)

    def gemm():
        x.grad = w.grad = None
        xm = x.permute(0, 2, 3, 1).reshape(-1, ci)
        y = (xm @ w.view(co, ci).t()).view(n, H, W, co).permute(0, 3, 1, 2)
        y.backward(g)

    def conv_f():
        with torch.no_grad():
            F.conv2d(x, w)

    def gemm_f():
        with torch.no_grad():
            (x.permute(0, 2, 3, 1).reshape(-1, ci) @ w.view(co, ci).t())

    flop = 2.0 * n * H * W * ci * co
    r = dict(shape=[n, H, W, ci, co], conv_fwd_ms=timeit(conv_f), gemm_fwd_ms=timeit(gemm_f), conv_fwdbwd_ms=timeit(conv), gemm_fwdbwd_ms=timeit(gemm))
    r["conv_tflops_fwdbwd"] = 3 * flop / r["conv_fwdbwd_ms"] / 1e9
    r["gemm_tflops_fwdbwd"] = 3 * flop / r["gemm_fwdbwd_ms"] / 1e9
    out.append(r)
    print(json.dumps(r), flush=True)
