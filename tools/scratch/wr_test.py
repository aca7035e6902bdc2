import ctypes, os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "wr_test.so")
lib = ctypes.CDLL(so)
lib.launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
BP, C = 1892352, 80
n = BP * C // 4
dst = torch.empty(BP * C, device="cuda")
pos = torch.randint(0, 128, (BP, 3), dtype=torch.int32, device="cuda")
flush = torch.empty(256 * 1024 * 1024, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(which, grid, nt, cold, reps=10):
    evs = []
    for i in range(reps + 2):
        if cold: flush.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.launch(which, dst.data_ptr(), pos.data_ptr(), n, 20, grid, nt, st); e1.record()
        if i >= 2: evs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    return t
for which, name in [(0, "fill_gridstride"), (1, "fill_tiles"), (2, "pos_store")]:
    for grid in [2048, 4096, 16384, (n + 1023) // 1024]:
        for nt in [0, 1]:
            tw, tc = run(which, grid, nt, False), run(which, grid, nt, True)
            print(f"{name:16s} grid={grid:6d} nt={nt} warm {tw*1e3:7.1f} us ({605.6/tw/1e3:5.2f} TB/s)  cold {tc*1e3:7.1f} us ({605.6/tc/1e3:5.2f} TB/s)")
# torch memset for reference
for cold in [False, True]:
    evs = []
    for i in range(12):
        if cold: flush.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dst.zero_(); e1.record()
        if i >= 2: evs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    print("torch zero_ cold" if cold else "torch zero_ warm", f"{t*1e3:7.1f} us ({605.6/t/1e3:5.2f} TB/s)")
