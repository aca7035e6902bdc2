#!/usr/bin/env python3
"""MIOpen on the convolution shapes that dominate the cfg4 step (tools/conv_shapes.py), channels_last against NCHW, with MIOpen's
full find per problem: forward, data gradient and weight gradient timed separately (HIP events, 20 launches each).
    python tools/kbench_conv_layout.py > gpurun_out/kbench_conv_layout.jsonl"""
import json
import os
import sys

os.environ.setdefault("MIOPEN_FIND_MODE", "1")            # full find: every applicable solver is timed once
os.environ.setdefault("MIOPEN_FIND_ENFORCE", "1")
import torch  # noqa: E402

SHAPES = [
    # n, cin, h, w, cout, k, stride, pad, dilation
    (24, 512, 16, 44, 512, 3, 1, 1, 1),
    (24, 512, 16, 44, 512, 3, 1, 6, 6),
    (24, 256, 16, 44, 256, 3, 1, 1, 1),
    (4, 64, 128, 128, 64, 3, 1, 1, 1),
    (24, 64, 64, 176, 64, 3, 1, 1, 1),
    (24, 128, 32, 88, 128, 3, 1, 1, 1),
    (4, 144, 128, 128, 144, 3, 1, 1, 1),
    (4, 160, 32, 32, 160, 3, 1, 1, 1),
]


def timed(fn, n=20):
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


def main():
    torch.backends.cudnn.benchmark = True
    only = sys.argv[1:]
    for (n, cin, h, w, cout, k, s, p, d) in SHAPES:
        flops = 2.0 * n * (h // s) * (w // s) * cout * cin * k * k
        for fmt_name, fmt in (("channels_last", torch.channels_last), ("nchw", torch.contiguous_format)):
            if only and fmt_name not in only:
                continue
            x = torch.randn(n, cin, h, w, device="cuda").contiguous(memory_format=fmt)
            wt = torch.randn(cout, cin, k, k, device="cuda").contiguous(memory_format=fmt)
            y = torch.ops.aten.convolution(x, wt, None, [s, s], [p, p], [d, d], False, [0, 0], 1)
            gy = torch.randn_like(y)
            rec = {"shape": [n, cin, h, w], "cout": cout, "k": k, "stride": s, "dilation": d, "layout": fmt_name, "gflop": flops / 1e9}
            rec["fwd_us"] = timed(lambda: torch.ops.aten.convolution(x, wt, None, [s, s], [p, p], [d, d], False, [0, 0], 1))
            rec["bwd_data_us"] = timed(lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, [s, s], [p, p], [d, d], False, [0, 0], 1,
                                                                                   [True, False, False]))
            rec["bwd_weight_us"] = timed(lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, [s, s], [p, p], [d, d], False, [0, 0], 1,
                                                                                     [False, True, False]))
            for key in ("fwd", "bwd_data", "bwd_weight"):
                rec[key + "_tflops"] = round(flops / rec[key + "_us"] / 1e6, 1)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
