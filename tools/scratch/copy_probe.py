"""Which ops launch copy / layout-conversion kernels in one cfg2 training step (torch.profiler)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
cfg = make_config(sys.argv[1] if len(sys.argv) > 1 else "cfg2")
dev = torch.device("cuda", 0)
ts = TrainStep(cfg, dev)
batch = synthetic_batch(cfg, dev, seed=0)
for _ in range(3):
    ts(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    ts(batch)
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_input_shape=True)
rows = [e for e in ev if e.key in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::_to_copy", "aten::cat", "aten::add_", "aten::add", "aten::fill_", "aten::zero_", "aten::mul", "aten::where")]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:40]:
    print(f"{e.key:18s} n={e.count:4d} dev={e.device_time_total/1e3:8.3f} ms  shapes={str(e.input_shapes)[:110]}")
print("---- by stack for aten::copy_")
ev2 = prof.key_averages(group_by_stack_n=6)
rows = [e for e in ev2 if e.key == "aten::copy_"]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:12]:
    st = [s for s in e.stack if "mm_training_amd" in s or "bench" in s][:3]
    print(f"n={e.count:4d} dev={e.device_time_total/1e3:8.3f} ms  {st}")
