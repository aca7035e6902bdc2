import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mm_training_amd.dp import make_config, TrainStep, synthetic_batch
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.backends.cudnn.benchmark = True
cfg = make_config(name)
ts = TrainStep(cfg, dev)
print(name, "params", sum(p.numel() for p in ts.model.parameters()) / 1e6, "M")
batch = synthetic_batch(cfg, dev, seed=0)
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, det, dep = ts(batch)
    torch.cuda.synchronize()
    print(i, f"{(time.perf_counter()-t0)*1e3:.1f} ms", float(loss), float(det), float(dep), flush=True)
print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)
