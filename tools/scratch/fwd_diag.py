import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from mm_training_amd import synthetic
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, sys.argv[1] if len(sys.argv) > 1 else "fwd_diag.so"))
lib.launch_diag.argtypes = [ctypes.c_int]*6 + [ctypes.c_void_p]*6
B, C = 4, 80
geom, vn = synthetic.rig_geometry(B)
P = geom[0].numel() // 3
geom = geom.reshape(B, P, 3).cuda(); feats = synthetic.features((B, P, C), 1).cuda()
out = torch.zeros(B, 128, 128, C, device="cuda"); pos = torch.empty(B, P, 3, dtype=torch.int32, device="cuda")
nchunks = (B * P + 511) // 512
dbg = torch.zeros(nchunks * 8, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    out.zero_()
    lib.launch_diag(B, P, C, 128, 128, 1, geom.data_ptr(), feats.data_ptr(), out.data_ptr(), pos.data_ptr(), dbg.data_ptr(), st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(10):
    lib.launch_diag(B, P, C, 128, 128, 1, geom.data_ptr(), feats.data_ptr(), out.data_ptr(), pos.data_ptr(), dbg.data_ptr(), st)
e1.record(); torch.cuda.synchronize()
print(sys.argv[1:], "kernel us:", e0.elapsed_time(e1) * 100)
d = dbg.cpu().numpy().reshape(nchunks, 8)
A1 = d[:, 1] - d[:, 0]; A234 = d[:, 2] - d[:, 1]; Bw0 = d[:, 3] - d[:, 2]; tail = d[:, 4] - d[:, 3]; tot = d[:, 4] - d[:, 0]
print("cycles (s_memtime = shader clk?) mean: A1(index+hash)=%.0f A2-4(count/scan/sort)=%.0f B(wave0)=%.0f tail(wait others)=%.0f total=%.0f" % (A1.mean(), A234.mean(), Bw0.mean(), tail.mean(), tot.mean()))
print("percentiles total:", np.percentile(tot, [5, 50, 95, 99]))
print("slots mean", d[:, 5].mean(), "max", d[:, 5].max())
rt = d[:, 6]
print("kernel span via memrealtime (100MHz ticks):", (rt.max() - rt.min()) / 100.0, "us")
# concurrency: sum of WG lifetimes / (span*CUs)
span_cycles = d[:, 4].max() - d[:, 0].min()
print("span cycles", span_cycles, "sum lifetimes/span = avg concurrent WGs:", tot.sum() / span_cycles, "per CU:", tot.sum() / span_cycles / 256)
kept_rows = d[:, 5]
