"""What the gradient exchange costs one rank before any communication: the cfg4 step in a ONE-rank RCCL group with torch's DDP wrap,
with the native reducer (dp/reducer.py) and without either
(`python3 tools/scratch/ddp_tax.py ddp|native|plain [steps]`; under rocprofv3 --kernel-trace --stats the kernel lists can be compared)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

bench.use_shipped_miopen_db()
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "ddp"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29711"), RANK="0", WORLD_SIZE="1")
if mode in ("ddp", "native"):
    os.environ["MMT_DP_REDUCER"] = mode
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from mm_training_amd.dp import TrainStep, make_config, synthetic_batch  # noqa: E402

torch.backends.cudnn.benchmark = True
cfg = make_config("cfg4")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
np.random.seed(0)
ts = TrainStep(cfg, dev, world_size=2 if mode in ("ddp", "native") else 1)      # (2 only selects the exchange; the group has one rank)
batches = [synthetic_batch(cfg, dev, seed=i) for i in range(2)]
for i in range(8):
    ts(batches[i % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    ts(batches[i % 2])
torch.cuda.synchronize()
print("RESULT %s conv=%s %.3f ms/step" % (mode, ts.conv_overlap, (time.perf_counter() - t0) / steps * 1e3), flush=True)
dist.destroy_process_group()
