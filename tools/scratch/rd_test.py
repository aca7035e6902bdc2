import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_training_amd import synthetic
from mm_training_amd.ops.voxel_pooling import VoxelPoolingPlan
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "rd_test.so"))
lib.launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
BP = 1892352
sink = torch.zeros(16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
flush = torch.empty(256 * 1024 * 1024, device="cuda")
def run(which, src, order, n, L, grid, cold=True, reps=10):
    evs = []
    for i in range(reps + 2):
        if cold: flush.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.launch(which, src.data_ptr(), order.data_ptr() if order is not None else 0, n, L, grid, sink.data_ptr(), st); e1.record()
        if i >= 2: evs.append((e0, e1))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
feats = torch.randn(BP * 80, device="cuda")
mb = BP * 80 * 4 / 1e6
for which in (0, 1):
    for grid in (2048, 4096, 8192, 16384):
        t = run(which, feats, None, BP * 20, 0, grid)
        print(f"stream_read U={4 if which==0 else 8} grid={grid:6d}: {t*1e3:7.1f} us  {mb/t/1e3:5.2f} TB/s")
geom, vn = synthetic.rig_geometry(4)
plan = VoxelPoolingPlan(geom.reshape(4, -1, 3).cuda(), vn)
K = plan.num_kept
cell_order = plan.plan[16:16 + K].clone()
orders = {
    "identity (all rows, sequential)": torch.arange(BP, dtype=torch.int32, device="cuda"),
    "kept rows, ascending (holes)": torch.sort(cell_order)[0].contiguous(),
    "kept rows, cell order (plan)": cell_order,
    "kept rows, random permutation": cell_order[torch.randperm(K, device="cuda")].contiguous(),
}
for name, order in orders.items():
    n = order.numel()
    for which, tag in ((2, "C4=20 U=4"), (3, "C4=20 U=8")):
        for L in (8, 16, 32, 64):
            for grid in (2048, 8192):
                t = run(which, feats, order, n, L, grid)
                print(f"{name:34s} {tag} L={L:3d} grid={grid:5d}: {t*1e3:7.1f} us  {n*320/1e6/t/1e3:5.2f} TB/s")
# 256-byte and 512-byte rows over the same buffer (identity / random)
for which, C4 in ((4, 16), (5, 32)):
    rows = BP * 80 // (C4 * 4)
    for name, order in (("identity", torch.arange(rows, dtype=torch.int32, device="cuda")),
                        ("random", torch.randperm(rows, device="cuda").to(torch.int32))):
        t = run(which, feats, order, rows, 32, 8192)
        print(f"rows of {C4*16} B {name:9s}: {t*1e3:7.1f} us  {rows*C4*16/1e6/t/1e3:5.2f} TB/s")
