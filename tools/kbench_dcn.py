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


def whole_op(res):
    """The whole operator, both forms, forward and backward (A/B in one process): the implicit-GEMM kernels
    (mmt_dcn_forward / mmt_dcn_backward) against im2col / col2im + the vendor GEMMs (ops/deform_conv.py, columns=True)."""
    from mm_training_amd.ops.deform_conv import deform_conv3x3
    for B, C, H, W, O, groups in ((24, 512, 16, 44, 512, 4), (12, 512, 32, 88, 512, 4)):
        torch.manual_seed(0)
        x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        off = (torch.randn(B, 18, H, W, device="cuda") * 0.5).requires_grad_(True)
        w = (torch.randn(O, C // groups, 3, 3, device="cuda") * 0.05).requires_grad_(True)
        go = torch.randn(B, O, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        flop = 2.0 * B * H * W * 9 * (C // groups) * O
        hbm_fwd = 4.0 * (x.numel() + off.numel() + w.numel() + B * H * W * O)
        hbm_bwd = 4.0 * (2 * x.numel() + 2 * off.numel() + 2 * w.numel() + B * H * W * O)
        entry = {"gflop_per_gemm": flop / 1e9, "algorithmic_MB_fwd": hbm_fwd / 1e6, "algorithmic_MB_bwd": hbm_bwd / 1e6}
        for name, columns in (("implicit_gemm", False), ("columns", True)):
            for waves in ((0, 1, 2, 3, 4) if not columns else (0,)):
                os.environ["MMT_DCN_FWD_CONFIG"] = str(waves)
                tf = timeit(lambda: deform_conv3x3(x, off, w, groups, columns=columns))
                key = name if waves == 0 else f"{name}_fwd_config{waves}"
                entry[key + "_fwd_ms"] = tf
                entry[key + "_fwd_TFLOPs"] = flop / tf / 1e9
            os.environ["MMT_DCN_FWD_CONFIG"] = "0"
            out = deform_conv3x3(x, off, w, groups, columns=columns)
            tb = timeit(lambda: torch.autograd.grad(out, (x, off, w), go, retain_graph=True))
            entry[name + "_bwd_ms"] = tb
            entry[name + "_bwd_TFLOPs"] = 2 * flop / tb / 1e9
        res[f"whole_op_B{B}_C{C}_{H}x{W}_g{groups}"] = entry


def main():
    res = {}
    whole_op(res)
    if "--whole-op-only" in sys.argv:
        print(json.dumps(res, indent=1))
        return
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
