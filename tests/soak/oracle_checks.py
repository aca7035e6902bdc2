"""Oracle-side helpers of the micro-benchmarks in tools/ (the only place they touch the oracle):
the full-size verification of `tools/kbench.py` (KBENCH_VERIFY=1) and the one-core CPU timings of the
oracle's C restatement that `tools/kbench_lidar.py` prints beside its GPU numbers."""
import time

import numpy as np

import oracle


def verify_pooling(geom, feats, out, pos, grad_out, grad_in, nx, ny, nz):
    """Forward / backward results of the kernels (torch CUDA tensors) against the oracle at full size."""
    r_out, r_pos = oracle.voxel_pooling_forward(geom.cpu().numpy(), feats.cpu().numpy(), nx, ny, nz)
    r_gi = oracle.voxel_pooling_backward(r_pos, grad_out.cpu().numpy())
    return {"pos_memo_equal": bool(np.array_equal(pos.cpu().numpy(), r_pos)),
            "bev_max_abs_err": float(np.abs(out.cpu().numpy() - r_out).max()),
            "grad_in_equal": bool(np.array_equal(grad_in.cpu().numpy(), r_gi))}


def _timed(fn):
    t0 = time.perf_counter()
    r = fn()
    return r, (time.perf_counter() - t0) * 1e3


def lidar_and_producer_cpu_timings(np_frames, vs, rng, feats, B, ny, nx, xyz, vc, vsz, frustum, combine, depth, ctx):
    """One-core milliseconds of the oracle's sequential C restatement on the inputs of kbench_lidar."""
    (rv, rn, rc), t_vox = _timed(lambda: oracle.voxelize_batch(np_frames, vs, rng, 15, 25000))
    _, t_vfe = _timed(lambda: oracle.simple_vfe(rv, rn, 5))
    _, t_sc = _timed(lambda: oracle.pillar_scatter(feats[:rc.shape[0]], rc, B, ny, nx))
    _, t_q = _timed(lambda: oracle.quantize(xyz, vc, vsz))
    _, t_g = _timed(lambda: oracle.geometry(frustum, combine))
    _, t_l = _timed(lambda: oracle.lift(depth, ctx))
    return {"voxelize": t_vox, "simple_vfe": t_vfe, "pillar_scatter": t_sc, "quantize_geometry": t_q,
            "frustum_geometry_no_quantize": t_g, "lift_forward": t_l}
