"""Is the step host-bound?  N steps issued back to back: the host's clock when the last step has been ISSUED against the clock when
the device has finished.  (host_issue ~ total: the host is the bottleneck, or something inside the step waits for the device.)
With a third argument: cProfile over the issue loop, top of the cumulative list.
usage: python tools/scratch/host_vs_gpu.py cfg5 [steps] [profile]"""
import sys, time
sys.path.insert(0, ".")
from mm_training_amd.miopen_db import enable
enable()                               # (what bench.py does: MIOpen's measured solver choices, before the first convolution)
import numpy as np
import torch
torch.backends.cudnn.benchmark = True
from mm_training_amd.dp import TrainStep, make_config, synthetic_batch

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
cfg = make_config(name)
import os
if os.environ.get("PROBE_BATCH"):
    cfg["batch_size"] = int(os.environ["PROBE_BATCH"])     # same launches, other device time: tells host-bound from device-bound
ts = TrainStep(cfg, dev)
batches = [synthetic_batch(cfg, dev, seed=i) for i in range(2)]
for i in range(10):
    ts(batches[i % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
marks = []
for i in range(steps):
    ts(batches[i % 2])
    marks.append(time.perf_counter())
t_issue = time.perf_counter()
torch.cuda.synchronize()
t_done = time.perf_counter()
d = np.diff([t0] + marks) * 1e3
print("%s: host issue %.2f ms/step (median %.2f, first ten %.2f), device done %.2f ms/step, host finished %.1f ms before the device"
      % (name, (t_issue - t0) / steps * 1e3, np.median(d), d[:10].mean(), (t_done - t0) / steps * 1e3, (t_done - t_issue) * 1e3))

if len(sys.argv) > 3 and sys.argv[3] == "sync":
    torch.cuda.set_sync_debug_mode("warn")       # every host-blocking call of torch's own prints a warning with its stack
    import warnings
    warnings.simplefilter("always")
    ts(batches[0])
    torch.cuda.set_sync_debug_mode("default")
elif len(sys.argv) > 3:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(10):
        ts(batches[i % 2])
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(60)
