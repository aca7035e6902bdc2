"""List host-synchronising torch calls inside one training step (torch.cuda.set_sync_debug_mode)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg = make_config(cfgname)
dev = torch.device("cuda", 0)
ts = TrainStep(cfg, dev)
batch = synthetic_batch(cfg, dev, seed=0)
for _ in range(2):
    ts(batch)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    ts(batch)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
import traceback
print(cfgname, "synchronising calls in one step:", len(w))
seen = {}
for x in w:
    key = (x.filename, x.lineno, str(x.message)[:80])
    seen[key] = seen.get(key, 0) + 1
for (f, l, m), n in seen.items():
    print(n, f.replace(os.getcwd() + "/", ""), l, m)
