#!/usr/bin/env python3
"""Where the dense part of a training step goes, by operator and input shape (torch.profiler, device time):
    python tools/conv_shapes.py [config] [steps] > gpurun_out/conv_shapes_<config>.txt
The convolutions are MIOpen's; this table says which LAYERS cost what, i.e. which parts of the model the step's time belongs to."""
import os
import sys

# one stream: kernels that overlap would each be charged the time they share the card
os.environ.setdefault("MMT_CONV_OVERLAP", "off")
os.environ.setdefault("MMT_HEAD_STREAMS", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (MIOpen find DB / environment exactly as the benchmark sets them)

bench.use_shipped_miopen_db()           # before the first convolution (and before torch is imported, like bench.py)
import torch  # noqa: E402


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
    import numpy as np
    torch.backends.cudnn.benchmark = config in ("cfg2", "cfg3", "cfg4", "cfg5")
    dev = torch.device("cuda", 0)
    cfg = make_config(config)
    torch.manual_seed(0)
    np.random.seed(0)
    ts = TrainStep(cfg, dev)
    batches = [synthetic_batch(cfg, dev, seed=i) for i in range(2)]
    for i in range(6):
        ts(batches[i % 2])
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for i in range(steps):
            ts(batches[i % 2])
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        dt = getattr(e, "self_device_time_total", None)
        if dt is None:
            dt = getattr(e, "self_cuda_time_total", 0)
        if dt > 0:
            rows.append((dt / steps, e.count / steps, e.key, str(e.input_shapes)[:150]))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print("device time per step: %.2f ms over %d (operator, shape) rows" % (total / 1e3, len(rows)))
    for dt, n, key, shapes in rows[:120]:
        print("%9.1f us %6.1f calls  %-42s %s" % (dt, n, key[:42], shapes))
    # which kernels each convolution shape runs (an operator event lists the kernels it launched)
    per = {}
    for e in prof.events():
        if e.name in ("aten::miopen_convolution", "aten::convolution_backward", "aten::miopen_convolution_transpose") and e.kernels:
            d = per.setdefault((e.name, str(e.input_shapes)[:110]), {})
            for k in e.kernels:
                t = d.setdefault(k.name[:70], [0.0, 0])
                t[0] += k.duration / steps
                t[1] += 1.0 / steps
    print("\nkernels per convolution shape (us per step, launches per step):")
    for (name, shapes), d in sorted(per.items(), key=lambda kv: -sum(v[0] for v in kv[1].values()))[:60]:
        print("%9.1f us  %s %s" % (sum(v[0] for v in d.values()), name, shapes))
        for k, (t, n) in sorted(d.items(), key=lambda kv: -kv[1][0]):
            print("      %9.1f us %5.1f x  %s" % (t, n, k))


if __name__ == "__main__":
    main()
