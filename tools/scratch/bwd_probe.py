"""What bounds vp_bwd_rows_vec4: same kernel, three pos_memo variants (real cells / every kept point in
cell 0 -> all gathers hit L1 / every point dropped -> buffer loads return 0 without fetching), each
timed alternating with a forward launch (cache state of a training step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_training_amd import synthetic
from mm_training_amd.ops.voxel_pooling import voxel_pooling_ext as ext
B, C = 4, 80
geom, vn = synthetic.rig_geometry(B)
nx, ny, nz = vn
P = geom[0].numel() // 3
geom = geom.reshape(B, P, 3).cuda(); feats = synthetic.features((B, P, C), 1).cuda()
out = torch.zeros(B, ny, nx, C, device="cuda"); pos = torch.empty(B, P, 3, dtype=torch.int32, device="cuda")
ext.voxel_pooling_forward_wrapper(B, P, C, nx, ny, nz, geom, feats, out, pos, flags=3 | 0x10)
pos_scratch = torch.empty_like(pos)
kept = pos[..., 0] != -1
pos_same = pos.clone(); pos_same[kept] = 0
pos_drop = torch.full_like(pos, -1)
go = torch.randn(B, ny, nx, C, device="cuda").permute(0, 3, 1, 2)
gi = torch.empty(B, P, C, device="cuda")
ws = torch.empty(ext.backward_workspace_elems(B, P, C, nx, ny), device="cuda")
def run(pm, with_ws=True, reps=15):
    evs = []
    for i in range(reps + 3):
        out.zero_()
        ext.voxel_pooling_forward_wrapper(B, P, C, nx, ny, nz, geom, feats, out, pos_scratch, flags=3 | 0x10)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ext.voxel_pooling_backward_wrapper(B, P, C, nx, ny, pm, go, gi, ws if with_ws else None); e1.record()
        if i >= 3: evs.append((e0, e1))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2] * 1e3
for rnd in range(2):
    for name, pm in (("real cells", pos), ("all kept -> cell 0", pos_same), ("all dropped", pos_drop)):
        print(f"{name:22s} prepare+main {run(pm):7.1f} us   main only (no workspace: reads pos_memo itself) {run(pm, False):7.1f} us")
evs = []
for i in range(18):
    out.zero_()
    ext.voxel_pooling_forward_wrapper(B, P, C, nx, ny, nz, geom, feats, out, pos_scratch, flags=3 | 0x10)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gi.zero_(); e1.record()
    if i >= 3: evs.append((e0, e1))
torch.cuda.synchronize()
print("torch zero_ of grad_in (605.6 MB), same alternation:", sorted(a.elapsed_time(b) for a, b in evs)[7] * 1e3, "us")
