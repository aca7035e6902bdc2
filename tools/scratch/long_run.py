"""A long run of the default (multi-stream) training step: losses stay finite and decrease, reserved memory does not creep.
    python tools/scratch/long_run.py [config] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mm_training_amd import miopen_db  # noqa: E402

miopen_db.enable()
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mm_training_amd.dp import TrainStep, make_config, synthetic_batch  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
torch.backends.cudnn.benchmark = True
cfg = make_config(config)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
np.random.seed(0)
ts = TrainStep(cfg, dev)
batches = [synthetic_batch(cfg, dev, seed=i) for i in range(4)]
marks = {}
for i in range(steps):
    loss = ts(batches[i % 4])[0]
    if i in (20, steps // 2, steps - 1):
        torch.cuda.synchronize()
        marks[i] = (float(loss), torch.cuda.memory_reserved() / 2 ** 30, torch.cuda.max_memory_allocated() / 2 ** 30)
        print("step %d: loss %.4f, reserved %.2f GiB, peak allocated %.2f GiB" % ((i,) + marks[i]), flush=True)
first, mid, last = marks[20], marks[steps // 2], marks[steps - 1]
assert all(np.isfinite(v[0]) for v in marks.values()) and last[0] < first[0], marks
# (the pools of the side streams and MIOpen's workspaces settle within the first dozens of steps: the second half must be flat)
assert last[1] <= mid[1] * 1.02 + 0.1, "reserved memory keeps growing: %s" % (marks,)
print("long run ok:", config, steps, "steps; conv gradients", ts.conv_overlap, "head streams", ts.model.head.task_streams)
