"""the depth softmax's launch with / without the calibration lookup riding in it, and (MMT_RIDER_NOLOOKUP=1) the rider kernel's softmax alone"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_training_amd import _lib, synthetic
from mm_training_amd.ops.bev_geometry import depth_softmax, frustum_axes, new_plan_cache, plan_prepare
import tests.test_lss_plan_gpu as T
B, N, D, fH, fW = 4, 6, 112, 16, 44
H, W = fH * 16, fW * 16
fr = T._frustum((H, W), 16, (2.0, 58.0, 0.5))
axes = tuple(a.cuda() for a in frustum_axes(fr))
vc, vs, vn = [-51.2 + 0.4, -51.2 + 0.4, -1.0], [0.8, 0.8, 8.0], [128, 128, 1]
s2e, K = synthetic.camera_rig(B, N, W, H, jitter=0.02, seed=0)
comb = s2e.matmul(torch.inverse(K)).contiguous().cuda()
cache = new_plan_cache(N, D, fH, fW, vn, "cuda", slots=16)
for _ in range(3): plan_prepare(comb, axes, vn, vc, vs, cache)
lt = torch.bfloat16 if "bf16" in sys.argv else torch.float32
logits = torch.randn(B * N, D, fH, fW, device="cuda").to(lt).contiguous(memory_format=torch.channels_last)
oracle = torch.zeros(B * N, D, fH, fW, device="cuda").contiguous(memory_format=torch.channels_last); oracle[:, 3, ::2, ::2] = 1
def t(lk, reps=100):
    for _ in range(5): depth_softmax(logits, oracle, torch.float32, plan_lookup=lk)
    _lib.TIMING = {}
    for _ in range(reps): depth_softmax(logits, oracle, torch.float32, plan_lookup=lk)
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in _lib.TIMING["softmax"]); _lib.TIMING = None
    return "median %.2f us  mean %.2f us" % (ms[len(ms) // 2] * 1e3, sum(ms) / len(ms) * 1e3)
lk = (comb, axes, vn, vc, vs, cache)
print("plain      ", t(None)); print("with lookup", t(lk)); print("plain      ", t(None)); print("with lookup", t(lk))
