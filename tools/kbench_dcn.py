#!/usr/bin/env python3
"""DCN backward micro-benchmark (run on the GPU box): mmt_dcn_col2im (fp32 atomics) against mmt_dcn_col2im_sorted
(LDS sort by destination pixel + gather) at the DepthNet shape [B*N = 24, C = 512, 16 x 44], groups 4."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mm_training_amd import _lib


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def main():
    res = {}
    for B, H, W, C, groups, scale in ((24, 16, 44, 512, 4, 0.0), (24, 16, 44, 512, 4, 1.0), (12, 32, 88, 512, 4, 1.0)):
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, H, W, C, generator=g).cuda()
        offset = (torch.randn(B, H, W, 18, generator=g) * scale).cuda()
        Cg, N = C // groups, B * H * W
        grad_col = torch.randn(groups, N, 9 * Cg, generator=g).cuda()
        gx, go = torch.zeros_like(x), torch.empty_like(offset)
        ws = torch.empty(_lib.lib().mmt_dcn_col2im_workspace_elems(B, H, W), dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream

        def atomic():
            gx.zero_()
            _lib.call("mmt_dcn_col2im", B, H, W, C, groups, x.data_ptr(), offset.data_ptr(), grad_col.data_ptr(), gx.data_ptr(), go.data_ptr(), st)

        def srt():
            _lib.call("mmt_dcn_col2im_sorted", B, H, W, C, groups, x.data_ptr(), offset.data_ptr(), grad_col.data_ptr(), gx.data_ptr(),
                      go.data_ptr(), ws.data_ptr(), ws.numel(), st)
        col = torch.empty(groups, N, 9 * Cg, device="cuda")

        def im2col():
            _lib.call("mmt_dcn_im2col", B, H, W, C, groups, x.data_ptr(), offset.data_ptr(), col.data_ptr(), st)
        nbytes = grad_col.numel() * 4 + x.numel() * 4 * 2 + offset.numel() * 4 * 2       # grad_col + x read, grad_x + grad_offset written
        ta, ts_, ti = timeit(atomic), timeit(srt), timeit(im2col)
        fbytes = col.numel() * 4 + x.numel() * 4 + offset.numel() * 4
        res[f"B{B}_{H}x{W}_C{C}_offsets{scale}"] = {"atomic_ms_incl_zero_fill": ta, "sorted_ms": ts_, "algorithmic_MB": nbytes / 1e6,
                                                   "im2col_ms": ti, "im2col_GBps": fbytes / ti / 1e6, "im2col_frac_of_peak": fbytes / ti / 1e6 / 8000.0,
                                                   "sorted_GBps": nbytes / ts_ / 1e6, "sorted_frac_of_peak": nbytes / ts_ / 1e6 / 8000.0}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
