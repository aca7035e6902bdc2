import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_training_amd.dp import make_config, TrainStep, synthetic_batch
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
cfg = make_config("cfg2")
ts = TrainStep(cfg, dev)
# capturable optimizer
ts.optimizer = torch.optim.AdamW(ts.model.parameters(), lr=1e-3 / 64 * 4, weight_decay=1e-7, fused=True, capturable=True)
batch = synthetic_batch(cfg, dev, seed=0)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(5):
        ts(batch)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10): ts(batch)
torch.cuda.synchronize(); print("eager ms/step", (time.perf_counter() - t0) * 100)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = ts(batch)
    torch.cuda.synchronize()
    for i in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20): g.replay()
    torch.cuda.synchronize(); print("graph ms/step", (time.perf_counter() - t0) * 50, "loss", float(out[0]))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:500])
