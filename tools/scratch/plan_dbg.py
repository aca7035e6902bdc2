import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mm_training_amd import synthetic
from mm_training_amd.ops.bev_geometry import frustum_axes, new_plan_cache, plan_cache_counters
import tests.test_lss_plan_gpu as T
N, D, fH, fW, C = 2, 12, 6, 9, 64
H, W = fH * 16, fW * 16
fr = T._frustum((H, W), 16, (2.0, 2.0 + 2.0 * D, 2.0))
axes = tuple(a.cuda() for a in frustum_axes(fr))
vc, vs, vn = [-51.2 + 0.4, -51.2 + 0.4, -1.0], [0.8, 0.8, 8.0], [128, 128, 1]
rigs = []
for seed in range(5):
    s2e, K = synthetic.camera_rig(1, N, W, H, jitter=0.3, seed=seed)
    rigs.append(s2e.matmul(torch.inverse(K))[0])
cache = new_plan_cache(N, D, fH, fW, vn, "cuda", slots=4)
lay = T._layout(N, D, fH, fW, vn[0], vn[1], cache)
g = torch.Generator().manual_seed(0)
for ids in ([0, 0, 0, 0], [0, 1, 0, 1], [1, 0], [1, 0], [2, 3, 2]):
    cb = torch.stack([rigs[i] for i in ids]).contiguous().cuda()
    B = len(ids)
    depth = torch.rand(B * N, fH, fW, D, generator=g).softmax(-1).cuda()
    ctx = torch.randn(B * N, fH, fW, C, generator=g).cuda()
    out = T._forward(cb, axes, vc, vs, vn, depth, ctx, cache)
    torch.cuda.synchronize()
    vd = T._verdicts(cache, lay, B)
    hdr = cache[lay["base"]:lay["base"] + 64].view(torch.int32).cpu().numpy()
    print(ids, "verdict slot/units/state/rep:", vd[:, :4].tolist(), "hdr todo", hdr[7], "hits", hdr[8], "built", hdr[9], "calls", hdr[12], "stale", hdr[13], "snaps", hdr[14], hdr[15], "nan", bool(torch.isnan(out).any()), flush=True)
