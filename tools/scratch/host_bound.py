import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # sets up the shipped MIOpen DB env before torch convs run
import torch
from mm_training_amd.dp import make_config, TrainStep, synthetic_batch
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
cfg = make_config("cfg2"); ts = TrainStep(cfg, dev); batch = synthetic_batch(cfg, dev)
for _ in range(5): ts(batch)
torch.cuda.synchronize()
hs, tt = [], []
for _ in range(10):
    t0 = time.perf_counter(); ts(batch); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    hs.append(t1 - t0); tt.append(t2 - t0)
print("host queue time ms", sorted(hs)[5] * 1e3, " total ms", sorted(tt)[5] * 1e3)
