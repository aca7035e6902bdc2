"""CPU: bench.py's bookkeeping -- the algorithmic-byte formulas of BASELINE.md section 2 /
SURVEY.md section 8d, the roofline object and the committed PMC traffic summary."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_algorithmic_bytes_match_baseline_md():
    b = _bench()
    BP, C, B, ny, nx = 1892352, 80, 4, 128, 128
    K = int(0.73 * BP)
    fwd, bwd = b.algorithmic_bytes(BP, K, C, B, ny, nx)
    assert fwd == 12 * BP + 12 * BP + 4 * C * K + 4 * C * B * ny * nx
    assert bwd == 12 * BP + 4 * C * B * ny * nx + 4 * C * BP
    assert abs(fwd / 1e6 - 508) < 2 and abs(bwd / 1e6 - 649) < 1      # the worked example in BASELINE.md


def test_roofline_entry_and_pmc_summary():
    b = _bench()
    r = b.roofline_entry("k", 430263488, 0.0969, ("vp_fwd_seg_gather",))
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] - 430263488 / 0.0969e-3 / 1e9) < 1e-6
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # traffic comes from the committed rocprofv3 --pmc summary and is close to the algorithmic bytes
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_hotpath_cfg2_pmc.json")))
    assert r["traffic"] == pmc["kernels"]["vp_fwd_seg_gather"]["traffic_bytes"]
    assert 0.9 < r["traffic"] / 430263488 < 1.2


def test_help_and_defaults():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout


def test_backend_choice_is_the_same_on_every_rank():
    """One GPU per rank -> RCCL; fewer GPUs than ranks (rehearsal on a 1-GPU box) -> gloo on EVERY rank
    (a per-rank decision made rank 0 pick RCCL and rank 1 gloo: the rendezvous hung)."""
    import bench
    assert [bench.choose_backend(8, r, 8) for r in range(8)] == [("nccl", r) for r in range(8)]
    assert [bench.choose_backend(2, r, 1) for r in range(2)] == [("gloo", 0), ("gloo", 0)]
    assert [bench.choose_backend(4, r, 2) for r in range(4)] == [("gloo", 0), ("gloo", 1), ("gloo", 0), ("gloo", 1)]
    assert bench.choose_backend(1, 0, 1) == ("nccl", 0)
    assert bench.choose_backend(2, 1, 8, "gloo") == ("gloo", 1)
