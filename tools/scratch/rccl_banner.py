"""Does bench.init_dist keep RCCL's version banner out of stdout?  One rank on RCCL; prints OK / the offending lines to stderr."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29877", RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
import bench
bench._late_imports("cuda")
import torch, torch.distributed as dist
# a two-rank world cannot be formed on one card over RCCL: emulate init_dist's nccl branch with a one-rank group
os.environ["WORLD_SIZE"] = "1"
torch.cuda.set_device(0)
with bench._StdoutToStderr():
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dist.barrier()
t = torch.ones(4, device="cuda"); dist.all_reduce(t)
print('{"json": "line"}', flush=True)
dist.destroy_process_group()
''' % ROOT
p = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=300)
lines = [l for l in p.stdout.splitlines() if l.strip()]
print("stdout lines:", lines)
print("banner went to stderr:", "RCCL version" in p.stderr)
sys.exit(0 if lines == ['{"json": "line"}'] and p.returncode == 0 else 1)
