import os, sys, ctypes
# needs a -DVOX_STAMPS build: python tools/build_variant.py stamps lidar_voxelize.hip -DVOX_STAMPS, then MMT_HIP_LIB=mm_training_amd/variants/libmmt_stamps.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mm_training_amd import _lib, synthetic
from mm_training_amd.lidar import hard_voxelize_mean_batch
RANGE, VSIZE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], [0.2, 0.2, 8.0]
pts = [synthetic.lidar_frame(40000, 5, RANGE, seed=s).cuda() for s in range(4)]
for _ in range(5):
    hard_voxelize_mean_batch(pts, VSIZE, RANGE, 15, 25000, 5, materialize_voxels=False)
torch.cuda.synchronize()
h = _lib.lib()
buf = np.zeros(1024 * 16, dtype=np.uint64)
h.mmt_vox_debug_stamps.argtypes = [ctypes.c_void_p]
assert h.mmt_vox_debug_stamps(buf.ctypes.data) == 0
st = buf.reshape(1024, 16).astype(np.int64)
def rep(name, a, b, rows):
    d = (st[rows, b] - st[rows, a]) * 10.0 / 1000.0   # 100 MHz -> us
    print(f"{name}: mean {d.mean():.2f} us  max {d.max():.2f}  min {d.min():.2f}")
o0 = int(os.environ.get("VOX_OWN_ROW0", "0"))      # fused cells + own launch: the owners follow the cells workgroups (ceil(N / 1024) + B of them)
own = slice(o0, o0 + 512)
t0 = st[own, 0].min()
print("own: first start..last end", (st[own, 4].max() - t0) / 100.0, "us; start spread", (st[own, 0].max() - t0) / 100.0)
rep("own init (offsets, LDS init)", 0, 1, own)
rep("own stream: loads + masks", 1, 6, own)
rep("own stream: count + append", 6, 7, own)
rep("own stream: barrier", 7, 2, own)
rep("own flush gather", 2, 5, own)
rep("own flush rounds", 5, 3, own)
rep("own epilogue", 3, 4, own)
em = slice(0, 628)
t0 = st[em, 8].min()
print("emit: first start..last end", (st[em, 12].max() - t0) / 100.0, "us; start spread", (st[em, 8].max() - t0) / 100.0)
rep("emit locate", 8, 9, em)
rep("emit prologue: loads + count", 9, 13, em)
rep("emit prologue: reduce", 13, 10, em)
rep("emit heads", 10, 11, em)
rep("emit stores", 11, 12, em)
