#!/usr/bin/env python3
"""Phase stamps of lss_col_bwd (a -DLSS_STAMPS build of lift_splat_col.hip writes 8 s_memtime stamps per workgroup into grad_context):
0 entry | 4 phase A's loads arrived | 3 mismatch pass done | 5 cells, flags, LDS stores done | 6 after the barrier | 1 context operand in registers | 2 products done
   python tools/build_variant.py colstamps lift_splat_col.hip -DLSS_STAMPS && python tools/scratch/col_stamps8.py [cfg4|cfg5] [f32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_training_amd import _lib, synthetic
from tools.kbench_camera import SHAPES, load
shape = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
bf16 = (sys.argv[2] if len(sys.argv) > 2 else "bf16") == "bf16"
_lib.lib()
h = load("mm_training_amd/variants/libmmt_colstamps.so")
B, N, D, fH, fW, C, (H, W), d_bound, bounds = SHAPES[shape]
sd = torch.bfloat16 if bf16 else torch.float32
s2e, K = synthetic.camera_rig(B, N, W, H, jitter=0.02, seed=0)
import math
pitch = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0        # degrees about the cameras' x axis (tools/kbench_camera.py --pitch)
c_, s_ = math.cos(math.radians(pitch)), math.sin(math.radians(pitch))
rx = torch.tensor([[1, 0, 0, 0], [0, c_, -s_, 0], [0, s_, c_, 0], [0, 0, 0, 1]], dtype=torch.float32)
combine = s2e.matmul(rx).matmul(torch.inverse(K)).contiguous().cuda()
fu = torch.linspace(0, W - 1, fW).cuda(); fv = torch.linspace(0, H - 1, fH).cuda(); fd = torch.arange(*d_bound, dtype=torch.float).cuda()
vs = [b[2] for b in bounds]; vc = [b[0] + b[2] / 2.0 for b in bounds]
nx, ny, nz = [int((b[1] - b[0]) / b[2]) for b in bounds]
vc_c, vs_c = _lib.float3(vc), _lib.float3(vs)
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
depth = torch.rand(B * N, fH, fW, D, generator=g).softmax(-1).to(sd).cuda()
ctx = torch.randn(B * N, fH, fW, C, generator=g).to(sd).cuda()
go = torch.randn(B, ny, nx, C, generator=g).cuda()
gd = torch.empty_like(depth)
nwg = 8 * ((B * N + 7) // 8) * fW * ((fH + 15) // 16)
gc = torch.zeros(max(ctx.numel(), nwg * 16 + 64), device="cuda")
stats = torch.zeros(2 * _lib.LSS_STATS_SLOTS, dtype=torch.int64, device="cuda")
sfx = "_bf16" if bf16 else ""
PM, COL = 0x100, 0x400
for _ in range(4):
    gc.zero_()
    rc = getattr(h, "mmt_lss_splat_backward_cam" + sfx)(B, N, D, fH, fW, C, nx, ny, nz, combine.data_ptr(), fu.data_ptr(), fv.data_ptr(), fd.data_ptr(), vc_c, vs_c,
                                                        depth.data_ptr(), ctx.data_ptr(), go.data_ptr(), ny * nx * C, 1, nx * C, C, gd.data_ptr(), gc.data_ptr(), None, stats.data_ptr(), PM | COL, st)
    assert rc == 0, h.mmt_last_error()
torch.cuda.synchronize()
s = gc.view(-1)[:nwg * 16].view(torch.int64).view(-1, 8).cpu()
s = s[(s[:, 0] != 0) & (s[:, 2] != 0) & (s[:, 3] != 0)]
names = [("start to arrival", 0, 4), ("cells+dep stores", 4, 7), ("reduce+park", 7, 5), ("barrier", 5, 6), ("ctx operand", 6, 1), ("products", 1, 2), ("sum + mismatch pass", 2, 3), ("life", 0, 3)]
for n, a, b in names:
    v = (s[:, b] - s[:, a]).float()
    print("%-12s mean %7.0f p50 %7.0f p90 %7.0f max %7.0f" % (n, v.mean(), v.median(), v.quantile(0.9), v.max()))
# timeline per XCD (workgroup i runs on XCD i % 8; the XCDs' counters are not synchronised)
full = gc.view(-1)[:nwg * 16].view(torch.int64).view(-1, 8).cpu()
for x in range(8):
    sx = full[x::8]
    sx = sx[(sx[:, 0] != 0) & (sx[:, 2] != 0)]
    t0 = sx[:, 0].min()
    st_, en_ = (sx[:, 0] - t0).float(), (sx[:, 2] - t0).float()
    q = torch.tensor([0.1, 0.5, 0.9, 1.0])
    print("xcd %d: %4d workgroups; starts p10/p50/p90/max %s; ends %s" % (x, len(sx), [int(v) for v in st_.quantile(q)], [int(v) for v in en_.quantile(q)]))
