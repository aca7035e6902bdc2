#!/usr/bin/env python3
"""In-kernel phase timing of lss_plan_fwd (run on the GPU box with a -DPLAN_STAMPS build of lift_splat_plan.hip):
   python tools/build_variant.py stamps lift_splat_plan.hip -DPLAN_STAMPS [-DPLAN_MAX_PAIR_RUNS=4 ...]
   python tools/kbench_plan_stamps.py [--shape cfg4|cfg5] [--dtype f32|bf16] mm_training_amd/variants/libmmt_stamps.so
The stamp build writes 8 words per workgroup where a plain build writes the column summary: s_memtime at kernel entry (0),
after the verdict (1), with the first record in LDS (2), after the first unit's pairs (3) and fold (4), at the workgroup's end
(5), the first record's (pairs << 32 | runs) (6) and the XCC id (7).  s_memtime counts shader cycles per XCD (the XCDs'
counters are not synchronised): spans are taken per XCD."""
import argparse, ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mm_training_amd import _lib, synthetic
from tools.kbench_camera import SHAPES, load


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="cfg4")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--cold", action="store_true")
    ap.add_argument("lib")
    args = ap.parse_args()
    _lib.lib()
    h = load(args.lib)
    B, N, D, fH, fW, C, (H, W), d_bound, bounds = SHAPES[args.shape]
    bf16 = args.dtype == "bf16"
    sd = torch.bfloat16 if bf16 else torch.float32
    s2e, K = synthetic.camera_rig(B, N, W, H, jitter=0.02, seed=0)
    combine = s2e.matmul(torch.inverse(K)).contiguous().cuda()
    fu = torch.linspace(0, W - 1, fW, dtype=torch.float).cuda()
    fv = torch.linspace(0, H - 1, fH, dtype=torch.float).cuda()
    fd = torch.arange(*d_bound, dtype=torch.float).cuda()
    vs = [b[2] for b in bounds]
    vc = [b[0] + b[2] / 2.0 for b in bounds]
    nx, ny, nz = [int((b[1] - b[0]) / b[2]) for b in bounds]
    vc_c, vs_c = _lib.float3(vc), _lib.float3(vs)
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(0)
    depth = torch.rand(B * N, fH, fW, D, generator=g).softmax(-1).to(sd).cuda()
    ctx = torch.randn(B * N, fH, fW, C, generator=g).to(sd).cuda()
    out = torch.zeros(B, ny, nx, C, device="cuda")
    pbytes = h.mmt_lss_plan_cache_bytes(N, D, fH, fW, nx, ny, max(B, 2))
    cache = torch.zeros(pbytes + 256, dtype=torch.uint8, device="cuda")
    ptr = (cache.data_ptr() + 255) & ~255
    nwg_max = B * 8192
    stamps = torch.zeros(nwg_max * 16, dtype=torch.int64, device="cuda")
    flush = torch.zeros(256 * 1024 * 1024, device="cuda")

    def fwd(flags):
        return getattr(h, "mmt_lss_splat_forward_plan" + ("_bf16" if bf16 else ""))(
            B, N, D, fH, fW, C, nx, ny, nz, combine.data_ptr(), fu.data_ptr(), fv.data_ptr(), fd.data_ptr(), vc_c, vs_c, depth.data_ptr(),
            ctx.data_ptr(), out.data_ptr(), stamps.data_ptr(), ptr, pbytes, flags, st)

    assert fwd(0x100) == 0, h.mmt_last_error()
    for _ in range(4):
        if args.cold:
            flush.add_(1.0)
        stamps.zero_()
        assert fwd(0x100 | 0x2000) == 0, h.mmt_last_error()
    torch.cuda.synchronize()
    s = stamps.view(-1, 16).cpu()
    live = s[:, 0] != 0
    s = s[live]
    worked = s[:, 2] != 0
    print("workgroups launched %d, with a unit %d" % (len(s), int(worked.sum())))
    clk = 2.1e3      # cycles per us, roughly (s_memtime ticks; the kernel's own clock is not known here)
    res = {}
    for x in sorted(set(s[:, 7].tolist())):
        m = s[:, 7] == x
        sx = s[m]
        t0 = sx[:, 0].min()
        end = torch.where(sx[:, 5] != 0, sx[:, 5], sx[:, 1]).max()
        w = sx[sx[:, 2] != 0]
        res[int(x)] = dict(wgs=int(m.sum()), span_cycles=int(end - t0), span_us=round(float(end - t0) / clk, 2),
                           last_start=int((sx[:, 0] - t0).max()), mean_start=int((sx[:, 0] - t0).float().mean()),
                           last_unit_start=int((w[:, 1] - t0).max()) if len(w) else 0)
    print(json.dumps(res, indent=1))
    w = s[worked]
    ph = dict(verdict=(w[:, 1] - w[:, 0]), record=(w[:, 2] - w[:, 1]), pairs=(w[:, 3] - w[:, 2]), fold=(w[:, 4] - w[:, 3]), life=(w[:, 5] - w[:, 0]))
    wp = w[w[:, 8] != 0]
    ph.update(p_issue=(wp[:, 8] - wp[:, 2]), p_arrive=(wp[:, 9] - wp[:, 8]), p_compute_rest=(wp[:, 10] - wp[:, 9]), p_barrier=(wp[:, 3] - wp[:, 10]))
    for k, v in ph.items():
        v = v.float()
        print("%-8s mean %7.0f  p50 %7.0f  p90 %7.0f  max %7.0f cycles" % (k, v.mean(), v.median(), v.quantile(0.9), v.max()))
    npairs, nruns = (w[:, 6] >> 32).float(), (w[:, 6] & 0xFFFFFFFF).float()
    rounds = torch.ceil(npairs / 16)
    for r in sorted(set(rounds.tolist())):
        m = rounds == r
        print("first unit with %d round(s) of pairs: %4d workgroups, pairs phase mean %6.0f cycles, runs mean %.0f" % (r, int(m.sum()), float(ph["pairs"][m].float().mean()), float(nruns[m].mean())))
    # when do units start / end relative to their XCD's first start
    rel_end = torch.zeros(len(s))
    for x in set(s[:, 7].tolist()):
        m = s[:, 7] == x
        t0 = s[m, 0].min()
        rel_end[m] = (torch.where(s[m, 5] != 0, s[m, 5], s[m, 1]) - t0).float()
    qs = torch.tensor([0.25, 0.5, 0.75, 0.9, 0.99, 1.0])
    print("workgroup end times (cycles after the XCD's first start), quantiles 25/50/75/90/99/100:", [int(v) for v in rel_end[worked].quantile(qs)])
    rel_start = torch.zeros(len(s))
    for x in set(s[:, 7].tolist()):
        m = s[:, 7] == x
        rel_start[m] = (s[m, 0] - s[m, 0].min()).float()
    print("workgroup start times, quantiles:", [int(v) for v in rel_start[worked].quantile(qs)])
    # per XCD, over the workgroups that had a unit: how well the units pack (sum of lives / slots against the span)
    for x in sorted(set(w[:, 7].tolist())):
        wx = w[w[:, 7] == x]
        med = wx[:, 0].float().median()
        wx = wx[(wx[:, 0].float() - med).abs() < 1e6]        # (a few workgroups carry a stamp of another clock domain: left out)
        t0 = wx[:, 0].min()
        st_rel = (wx[:, 0] - t0).float(); en_rel = (wx[:, 5] - t0).float()
        print("   xcd %d timeline: starts p10/p50/p90/max %6d %6d %6d %6d   ends p10/p50/p90/max %6d %6d %6d %6d   units whose life > 50k: %d" % (
            x, *[int(st_rel.quantile(q)) for q in (0.1, 0.5, 0.9, 1.0)], *[int(en_rel.quantile(q)) for q in (0.1, 0.5, 0.9, 1.0)], int(((wx[:, 5] - wx[:, 0]) > 50000).sum())))
        span = int((wx[:, 5] - t0).max())
        life = (wx[:, 5] - wx[:, 0]).float()
        order = torch.argsort(wx[:, 0])
        print("xcd %d: %4d units, span %6d cycles, sum of lives %8d (= %5.0f per slot at 128 slots), longest life %6d started at %6d; starts p50 %6d p90 %6d; pairs %d runs %d" % (
            x, len(wx), span, int(life.sum()), float(life.sum()) / 128, int(life.max()), int((wx[life.argmax(), 0] - t0)), int((wx[:, 0] - t0).float().median()), int((wx[:, 0] - t0).float().quantile(0.9)),
            int((wx[:, 6] >> 32).sum()), int((wx[:, 6] & 0xFFFFFFFF).sum())))


if __name__ == "__main__":
    main()
