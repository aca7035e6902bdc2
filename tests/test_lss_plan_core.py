"""CPU checks of the plan builder's integer core (mm_training_amd/csrc/lss_plan_core.h, compiled for the host by
tests/native/lss_plan_host.cpp): the job records it produces from a calibration's column summary cover every BEV cell once,
name every kept frustum point exactly once, respect the record limits, and -- run through a numpy emulation of the forward
kernel's arithmetic -- reproduce the direct scatter of lss_fpn.py:441-464 / voxel_pooling_forward_cuda.cu:16-34."""
import numpy as np
import pytest
import torch

from mm_training_amd import synthetic as S
from tests import plan_emul as P


def _rig(final_dim, ds, d_bound=(2.0, 58.0, 0.5), xb=(-51.2, 51.2, 0.8), yb=(-51.2, 51.2, 0.8), zb=(-5.0, 3.0, 8.0), pitch_deg=0.0, seed=0):
    s2e, K = S.camera_rig(1, 6, final_dim[1], final_dim[0], jitter=0.02, seed=seed)
    if pitch_deg:
        a = np.deg2rad(pitch_deg)
        R = torch.tensor([[1, 0, 0, 0], [0, np.cos(a), -np.sin(a), 0], [0, np.sin(a), np.cos(a), 0], [0, 0, 0, 1]], dtype=torch.float32)   # about the camera's x axis
        s2e = s2e.clone()
        s2e[0, ::2] = s2e[0, ::2] @ R          # every other camera pitched: columns leave their cell
    xyz = S.frustum_geometry_xyz(s2e, K, final_dim, ds, d_bound)
    geom, vn = S.quantize_cpu(xyz, xb, yb, zb)
    return geom[0].numpy(), [int(v) for v in vn]


def _check_plan(geom, vn, C=16, seed=0, clear=None, runs_cap=0):
    nx, ny, nz = vn
    N, D, fH, fW, _ = geom.shape
    d = P.dims(N, D, fH, fW, nx, ny, runs_cap)
    assert d["ok"]
    summary, rowcells = P.summary_from_geom(geom, nx, ny, nz, clear_uniform=clear)
    njobs, records, status = P.build(N, D, fH, fW, nx, ny, summary, rowcells, runs_cap=runs_cap)
    if njobs < 0:
        return njobs, None, status
    assert status[2] == 0 and status[1] == njobs and njobs <= d["jobs_cap"]
    # structure: tiles in order, cell ranges partition each tile, limits respected, partial rows = a permutation
    covered = np.zeros(d["ntiles"] * 64, np.int32)
    total_runs = 0
    for rec in records:
        job = P.decode(rec)
        h = job["h"]
        n = int(h["nruns"])
        assert n <= P.MAX_RUNS and 1 <= int(h["ncells"]) <= 64 and int(h["c0"]) + int(h["ncells"]) <= 64
        if not int(h["chain"]) & P.CHAIN_LINK:       # (the links of a chain cover their head's cell)
            covered[int(h["tile"]) * 64 + int(h["c0"]):int(h["tile"]) * 64 + int(h["c0"]) + int(h["ncells"])] += 1
        assert sorted(int(p) for p in job["runs"]["pslot"]) == list(range(n))
        cb = job["cell_begin"]
        assert cb[0] == 0 and (np.diff(cb[:int(h["ncells"]) + 1]) >= 0).all() and cb[int(h["ncells"])] == n
        assert int(job["pairs"]["nruns"].sum()) == n
        run0 = 0
        for pr in job["pairs"]:
            assert int(pr["run0"]) == run0
            run0 += int(pr["nruns"])
        for r in job["runs"]:                  # a run's partial row lies in its cell's range
            c = int(r["cell_local"])
            assert cb[c] <= int(r["pslot"]) < cb[c + 1] and 1 <= int(r["len"]) <= P.RUN_BINS
        total_runs += n
    assert (covered == 1).all()
    assert total_runs == status[0]
    # layout: eight groups of neighbouring tiles; inside a group the tiles by falling number of runs, a tile's records together
    gs = [int(v) for v in status[3:12]]
    assert gs[0] == 0 and gs[8] == njobs and all(a <= b for a, b in zip(gs, gs[1:]))
    tiles = [int(P.decode(r)["h"]["tile"]) for r in records]
    runs_of = {}
    for r in records:
        j = P.decode(r)
        runs_of[int(j["h"]["tile"])] = runs_of.get(int(j["h"]["tile"]), 0) + int(j["h"]["nruns"])
    for g in range(8):
        seq = tiles[gs[g]:gs[g + 1]]
        order = [t for i, t in enumerate(seq) if i == 0 or seq[i - 1] != t]
        assert len(order) == len(set(order)) and sorted(order) == list(range(g * d["ntiles"] // 8, (g + 1) * d["ntiles"] // 8))
        assert all(runs_of[a] >= runs_of[b] for a, b in zip(order, order[1:]))
    # arithmetic: emulated forward == direct scatter
    rng = np.random.default_rng(seed)
    depth = rng.random((N, fH, fW, D))
    context = rng.standard_normal((N, fH, fW, C))
    got = P.emulate_forward(d, records, depth, context)
    ref = P.reference_forward(geom, depth, context, nx, ny, nz)
    assert not np.isnan(got).any()
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-10)
    return njobs, records, status


def test_level_rig_small():
    geom, vn = _rig((64, 176), 16)             # fH 4, fW 11: rows past fH in the only block
    njobs, _, status = _check_plan(geom, vn)
    assert njobs >= 256


def test_level_rig_two_row_blocks_and_cleared_uniform_bits():
    geom, vn = _rig((320, 176), 16)            # fH 20: a second, partial block of rows
    N, D, fH, fW, _ = geom.shape
    _check_plan(geom, vn)
    clear = np.random.default_rng(1).random((N, 2, fW, D)) < 0.3        # uniform blocks walked row by row: same plan arithmetic
    _check_plan(geom, vn, clear=clear)


def test_pitched_rig_has_mixed_blocks():
    geom, vn = _rig((256, 176), 16, pitch_deg=4.0)
    nx, ny, nz = vn
    summary, _ = P.summary_from_geom(geom, nx, ny, nz)
    mixed = ((summary[..., 1] & P.UNIFORM) == 0).mean()
    assert mixed > 0.01
    _check_plan(geom, vn)


def test_non_multiple_of_eight_grid_and_fine_bins():
    geom, vn = _rig((64, 176), 16, d_bound=(2.0, 30.0, 0.125), xb=(-20.0, 20.8, 0.8), yb=(-10.0, 14.8, 0.8))   # 51 x 31 cells, runs longer than 4 bins
    assert vn[0] % 8 and vn[1] % 8
    _check_plan(geom, vn)


def test_random_geometry_every_block_mixed():
    rng = np.random.default_rng(3)
    N, D, fH, fW = 2, 6, 16, 5
    geom = np.stack([rng.integers(-2, 18, (N, D, fH, fW)), rng.integers(-2, 18, (N, D, fH, fW)), rng.integers(-1, 2, (N, D, fH, fW))], -1).astype(np.int32)
    njobs, _, status = _check_plan(geom, [16, 16, 1], runs_cap=4096)
    assert njobs > 0 and status[0] > 120         # more than the default capacity (2 * strips * D = 120)
    # the same geometry with the default capacity (twice a level rig's bound): reported as unplannable, nothing written
    njobs, _, status = _check_plan(geom, [16, 16, 1])
    assert njobs == -1 and status[2] == 1


def test_one_cell_takes_more_runs_than_a_job_holds():
    N, D, fH, fW = 1, 8, 16, 40                # 40 columns x 8 bins -> 2 runs of 4 bins per column, all into cell (3, 3)
    geom = np.zeros((N, D, fH, fW, 3), np.int32)
    geom[..., 0] = 3
    geom[..., 1] = 3
    njobs, recs, status = _check_plan(geom, [16, 16, 1])
    assert njobs == 4 and all(int(P.decode(r)["h"]["chain"]) == 0 for r in recs)         # 80 <= 96: an ordinary job
    geom2 = np.concatenate([geom, geom, geom], 3)     # 240 runs in one cell: a chain of three records for that cell
    njobs, recs, status = _check_plan(geom2, [16, 16, 1])
    chains = [int(P.decode(r)["h"]["chain"]) for r in recs]
    assert [c for c in chains if c] == [P.CHAIN_HEAD | P.CHAIN_MORE, P.CHAIN_LINK | P.CHAIN_MORE, P.CHAIN_LINK] and status[2] == 0


@pytest.mark.parametrize("final_dim", [(256, 704)])
def test_cfg4_shape_statistics(final_dim):
    geom, vn = _rig(final_dim, 16)
    njobs, records, status = _check_plan(geom, vn, C=8)
    nruns = int(status[0])
    # the numbers DESIGN 3.3f quotes for the cfg4 rig
    assert 15000 < nruns < 30000 and 256 <= njobs < 700, (nruns, njobs)
