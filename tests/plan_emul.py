"""Test helpers for the output-stationary plan of the fused lift-splat forward (mm_training_amd/csrc/lss_plan_core.h):
the host build of the builder's integer core (tests/native/lss_plan_host.cpp, g++), a numpy decoder of the job records
and a numpy emulation of what the forward kernel does with them.  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "native", "lss_plan_host.cpp")
_CORE = os.path.join(_HERE, "..", "mm_training_amd", "csrc", "lss_plan_core.h")
_OUT = os.path.join(_HERE, "native", "_build", "libplanhost.so")

UNIFORM = 0x10000
MAX_RUNS, MAX_PAIR_RUNS, RUN_BINS, WINDOW_BINS, TILE = 96, 4, 4, 8, 8
HDR = np.dtype([("ncells", "<u2"), ("npairs", "<u2"), ("nruns", "<u2"), ("c0", "<u2"), ("tile", "<i4"), ("chain", "<u4")])
CHAIN_HEAD, CHAIN_LINK, CHAIN_MORE = 1, 2, 4
PAIR = np.dtype([("col", "<u2"), ("rb", "u1"), ("nruns", "u1"), ("run0", "<u2"), ("w0", "<u2")])
RUN = np.dtype([("d0", "<u2"), ("len", "u1"), ("pslot", "u1"), ("cell_local", "<u4"), ("wmask", "<u2", (8,))])
CELL_BEGIN_OFF, PAIRS_OFF, RUNS_OFF, JOB_BYTES = 16, 96, 96 + 8 * MAX_RUNS, 96 + 8 * MAX_RUNS + 24 * MAX_RUNS

_lib = None


def host_lib():
    global _lib
    if _lib is None:
        stale = (not os.path.exists(_OUT) or os.path.getmtime(_OUT) < max(os.path.getmtime(_SRC), os.path.getmtime(_CORE)))
        if stale:
            os.makedirs(os.path.dirname(_OUT), exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-o", _OUT, _SRC])
        _lib = ctypes.CDLL(_OUT)
        assert _lib.plan_host_job_bytes() == JOB_BYTES
    return _lib


def dims(N, D, fH, fW, nx, ny, runs_cap=0):
    out = np.zeros(14, np.int32)
    ok = host_lib().plan_host_dims(N, D, fH, fW, nx, ny, int(runs_cap), out.ctypes.data_as(ctypes.c_void_p))
    names = ["N", "D", "fH", "fW", "nb", "nx", "ny", "tiles_x", "tiles_y", "ntiles", "ncells_tm", "strips", "runs_cap", "jobs_cap"]
    d = dict(zip(names, [int(v) for v in out]))
    d["ok"] = bool(ok)
    return d


def summary_from_geom(geom, nx, ny, nz, clear_uniform=None):
    """geom int32 [N, D, fH, fW, 3] of ONE sample -> (summary int32 [N, nb, fW, D, 2], rowcells int32 [N, nb, fW, D, 16]) with the
    semantics of mmt_camera.h's column summary.  clear_uniform: optional bool [N, nb, fW, D]: blocks whose uniform bit is
    cleared although they are uniform (the device clears it wave-wide; a cleared bit only means 'look at the rows')."""
    N, D, fH, fW, _ = geom.shape
    nb = (fH + 15) // 16
    g = np.full((N, D, nb * 16, fW, 3), -(1 << 20), np.int64)
    g[:, :, :fH] = geom
    g = g.reshape(N, D, nb, 16, fW, 3).transpose(0, 2, 4, 1, 3, 5)          # [N, nb, fW, D, 16, 3]
    x, y, z = g[..., 0], g[..., 1], g[..., 2]
    rowvalid = (np.arange(nb * 16).reshape(nb, 16) < fH)[None, :, None, None, :]
    zin = (z >= 0) & (z < nz) & rowvalid
    xyin = (x >= 0) & (x < nx) & (y >= 0) & (y < ny)
    packed = np.where(xyin, (y << 16) | x, -1)
    rowcells = np.where(zin, packed, -1).astype(np.int32)                    # (only asked for rows with the z bit)
    zmask = (zin.astype(np.int64) << np.arange(16)).sum(-1)
    # uniform: every VALID row of the block has the first row's (x, y) -- in the grid or not -- or no row passes z
    same = ((x == x[..., :1]) & (y == y[..., :1])) | ~rowvalid
    uniform = same.all(-1) | (zmask == 0)
    if clear_uniform is not None:
        uniform = uniform & ~clear_uniform
    s0 = np.where(xyin[..., 0], packed[..., 0], -1)
    summary = np.stack([s0, zmask | np.where(uniform, UNIFORM, 0)], -1).astype(np.int32)
    return np.ascontiguousarray(summary), np.ascontiguousarray(rowcells)


def build(N, D, fH, fW, nx, ny, summary, rowcells, runs_cap=0):
    """-> (njobs or -1, records uint8 [njobs, JOB_BYTES], status)"""
    d = dims(N, D, fH, fW, nx, ny, runs_cap)
    records = np.zeros((d["jobs_cap"], JOB_BYTES), np.uint8)
    status = np.zeros(16, np.int32)
    summary = np.ascontiguousarray(summary, np.int32)
    rowcells = np.ascontiguousarray(rowcells, np.int32)
    n = host_lib().plan_host_build(N, D, fH, fW, nx, ny, int(runs_cap), summary.ctypes.data_as(ctypes.c_void_p),
                                   rowcells.ctypes.data_as(ctypes.c_void_p), records.ctypes.data_as(ctypes.c_void_p),
                                   status.ctypes.data_as(ctypes.c_void_p))
    return n, (records[:n] if n >= 0 else records[:0]), status


def decode(record):
    """one job record (uint8 [JOB_BYTES]) -> dict(header, cell_begin, pairs, runs)"""
    h = record[:16].view(HDR)[0]
    cb = record[CELL_BEGIN_OFF:CELL_BEGIN_OFF + 65].astype(np.int64)
    pairs = record[PAIRS_OFF:PAIRS_OFF + 8 * int(h["npairs"])].view(PAIR)
    runs = record[RUNS_OFF:RUNS_OFF + 24 * int(h["nruns"])].view(RUN)
    return dict(h=h, cell_begin=cb, pairs=pairs, runs=runs)


def job_cells(d, job):
    """(x, y) of the job's cells, in order"""
    tile, c0, n = int(job["h"]["tile"]), int(job["h"]["c0"]), int(job["h"]["ncells"])
    l = np.arange(c0, c0 + n)                 # places on the tile's Z curve: x bits at the even, y bits at the odd positions
    lx = (l & 1) | ((l >> 1) & 2) | ((l >> 2) & 4)
    ly = ((l >> 1) & 1) | ((l >> 2) & 2) | ((l >> 3) & 4)
    return (tile % d["tiles_x"]) * TILE + lx, (tile // d["tiles_x"]) * TILE + ly


def emulate_forward(d, records, depth, context, dtype=np.float64):
    """What lss_plan_fwd computes from the records of ONE sample: depth [N, fH, fW, D], context [N, fH, fW, C]
    -> BEV [ny, nx, C].  Cells not covered by any job stay NaN (a complete plan leaves none)."""
    N, fH, fW, D = depth.shape
    C = context.shape[-1]
    out = np.full((d["ny"], d["nx"], C), np.nan, dtype)
    dp = np.zeros((N, d["nb"] * 16, fW, D + WINDOW_BINS), dtype)
    dp[:, :fH, :, :D] = depth
    cx = np.zeros((N, d["nb"] * 16, fW, C), dtype)
    cx[:, :fH] = context
    chain_acc = None
    for rec in records:
        job = decode(rec)
        nruns = int(job["h"]["nruns"])
        partial = np.zeros((max(nruns, 1), C), dtype)
        written = np.zeros(max(nruns, 1), bool)
        for pr in job["pairs"]:
            n, w, rb = int(pr["col"]) // fW, int(pr["col"]) % fW, int(pr["rb"])
            assert 1 <= int(pr["nruns"]) <= MAX_PAIR_RUNS
            w0 = int(pr["w0"])                       # the pair's depth window: bins [w0, w0 + WINDOW_BINS) of every image row, w0 even
            assert w0 % 2 == 0
            for r in job["runs"][int(pr["run0"]):int(pr["run0"]) + int(pr["nruns"])]:
                assert w0 <= int(r["d0"]) and int(r["d0"]) + int(r["len"]) - w0 <= WINDOW_BINS
                assert all(int(m) == 0 for k, m in enumerate(r["wmask"]) if not int(r["d0"]) <= w0 + k < int(r["d0"]) + int(r["len"]))
                wgt = np.zeros(16, dtype)
                for k in range(WINDOW_BINS):             # what the kernel does: the window's bins under the run's row masks
                    bits = (int(r["wmask"][k]) >> np.arange(16)) & 1
                    wgt += bits * dp[n, rb * 16:(rb + 1) * 16, w, w0 + k]
                assert not written[int(r["pslot"])]
                partial[int(r["pslot"])] = wgt @ cx[n, rb * 16:(rb + 1) * 16, w]
                written[int(r["pslot"])] = True
        assert written[:nruns].all()
        xs, ys = job_cells(d, job)
        cb = job["cell_begin"]
        chain = int(job["h"]["chain"])
        if chain:                              # one cell over several records: partial rows summed record after record
            assert len(xs) == 1 and cb[0] == 0 and cb[1] == nruns
            if chain & CHAIN_HEAD:
                assert chain_acc is None
                chain_acc = np.zeros(C, dtype)
            else:
                assert chain & CHAIN_LINK and chain_acc is not None
            chain_acc = chain_acc + partial[:nruns].sum(0)
            if not chain & CHAIN_MORE:
                assert np.isnan(out[ys[0], xs[0], 0]), "cell written twice"
                out[ys[0], xs[0]] = chain_acc
                chain_acc = None
            continue
        assert chain_acc is None
        for i, (x, y) in enumerate(zip(xs, ys)):
            if x < d["nx"] and y < d["ny"]:
                assert np.isnan(out[y, x, 0]), "cell written twice"
                out[y, x] = partial[cb[i]:cb[i + 1]].sum(0) if cb[i + 1] > cb[i] else 0
            else:
                assert cb[i + 1] == cb[i]
    assert chain_acc is None
    return out


def reference_forward(geom, depth, context, nx, ny, nz, dtype=np.float64):
    """Direct scatter (voxel_pooling_forward_cuda.cu:16-34 on depth * context) for ONE sample: geom [N, D, fH, fW, 3],
    depth [N, fH, fW, D], context [N, fH, fW, C] -> [ny, nx, C]."""
    N, D, fH, fW, _ = geom.shape
    C = context.shape[-1]
    x, y, z = geom[..., 0].astype(np.int64), geom[..., 1].astype(np.int64), geom[..., 2].astype(np.int64)
    kept = (x >= 0) & (x < nx) & (y >= 0) & (y < ny) & (z >= 0) & (z < nz)
    out = np.zeros((ny * nx, C), dtype)
    n, dd, h, w = np.nonzero(kept)
    feats = depth[n, h, w, dd].astype(dtype)[:, None] * context[n, h, w].astype(dtype)
    np.add.at(out, y[kept] * nx + x[kept], feats)
    return out.reshape(ny, nx, C)
