"""CPU: bench.py's bookkeeping -- the algorithmic-byte formulas of BASELINE.md section 2 /
SURVEY.md section 8d, the roofline objects, the committed PMC traffic summaries, and the launcher:
`python bench.py --gpus N` starts its own N ranks before any GPU call (rehearsed here on gloo)."""
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
    # bf16 feature storage ("bf16 features (cfg5): 2*C*K"): only the feature terms halve
    fwd16, bwd16 = b.algorithmic_bytes(BP, K, C, B, ny, nx, feat_bytes=2)
    assert fwd - fwd16 == 2 * C * K and bwd - bwd16 == 2 * C * BP


def test_lift_splat_and_lidar_bytes():
    b = _bench()
    BP, C, B, ny, nx = 1892352, 80, 4, 128, 128
    K, BN_HW = 1137000, 24 * 16 * 44
    fwd, bwd, l2f, l2b = b.lift_splat_bytes(BP, K, C, B, BN_HW, ny, nx, pos_memo=True)
    # 12BP geom + 12BP pos_memo + 4BP depth + 4*C*B*N*HW context + 4*C*B*ny*nx out
    assert fwd == 12 * BP + 12 * BP + 4 * BP + 4 * C * BN_HW + 4 * C * B * ny * nx
    # the frustum-tile kernels write / read no pos_memo (the backward redoes the kept test from geom)
    assert b.lift_splat_bytes(BP, K, C, B, BN_HW, ny, nx)[0] == fwd - 12 * BP
    assert bwd == 12 * BP + 4 * BP + 4 * C * BN_HW + 4 * C * B * ny * nx + 4 * BP + 4 * C * BN_HW
    assert l2f - fwd == K * 4 * C and l2b - bwd == K * 4 * C
    vox, scat, scat_bwd = b.lidar_bytes(5, 160000, 90000, 5, 64, 4, 512, 512)
    assert vox == 4 * 5 * 160000 + 16 * 90000 + 4 * 90000 + 4 * 5 * 90000          # SURVEY 8d, voxels not materialised
    assert scat == 4 * 64 * 90000 + 16 * 90000 + 4 * 64 * 4 * 512 * 512
    assert scat_bwd == 8 * 64 * 90000 + 16 * 90000
    assert b.lidar_bytes(5, 160000, 90000, 5, 64, 4, 512, 512, voxels_T=15)[0] - vox == 4 * 15 * 5 * 90000


def test_roofline_entry_and_pmc_summaries():
    b = _bench()
    r = b.roofline_entry("k", 430263488, 0.0969, traffic=None, l2_bytes=2 * 430263488)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["traffic"] is None
    assert abs(r["achieved"] - 430263488 / 0.0969e-3 / 1e9) < 1e-6
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["l2_side"]["achieved"] - 2 * r["achieved"]) < 1e-6 and r["l2_side"]["peak"] == 34500.0
    # traffic is reported only for a configuration whose PMC summary is committed, never borrowed from another shape
    assert b.pmc_traffic("no_such_config", ("vp_fwd_seg_gather",)) is None
    seen = 0
    rounds = sorted({n[:3] for n in os.listdir(os.path.join(ROOT, "profiles")) if n[:1] == "r" and n[1:3].isdigit() and "_pmc_" in n})
    newest = rounds[-1]                                                  # the newest round's summaries are the ones bench.py quotes
    for name in os.listdir(os.path.join(ROOT, "profiles")):
        if name.startswith(newest + "_pmc_") and name.endswith(".json"):
            cfg = name[len(newest + "_pmc_"):-len(".json")]
            kernels = json.load(open(os.path.join(ROOT, "profiles", name)))["kernels"]
            for k, e in kernels.items():
                if "traffic_bytes" in e:
                    assert b.pmc_traffic(cfg, (k,)) == e["traffic_bytes"] > 0
                    seen += 1
    assert seen > 10
    # the headline configuration's summary holds the kernels the step runs (round 5 on: the plan form, which has no fill and no atomics)
    k4 = json.load(open(os.path.join(ROOT, "profiles", newest + "_pmc_cfg4.json")))["kernels"]
    assert {"lift_splat_forward_plan", "lss_plan_probe", "lift_splat_backward_column_camera", "vox_emit"} <= set(k4)
    assert {"vox_cells", "vox_own"} <= set(k4) or {"vox_link", "vox_heads"} <= set(k4)      # (region-owner voxelizer from ABI 11 on)
    assert k4["lift_splat_forward_plan"]["launches"] >= 12 and k4["lift_splat_forward_plan"]["atomic_bytes"] == 0.0
    # a kernel the newest summary does not hold is "not measured" (None), never borrowed from an older round's file
    assert b.pmc_traffic("cfg4", ("scatter_write_nhwc_table_kernel",)) is None or "scatter_write_nhwc_table_kernel" in k4


def test_help_and_defaults():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout
    b = _bench()
    a = b.parse([])
    assert (a.gpus, a.config, a.mode) == (1, "cfg4", "train")           # BASELINE configs[3] is the headline workload
    assert "configs[3]" in b.WORKLOADS["cfg4"] and "LiDAR" in b.WORKLOADS["cfg4"]


def test_backend_choice_is_the_same_on_every_rank():
    """One GPU per rank -> RCCL; fewer GPUs than ranks (rehearsal on a 1-GPU box) -> gloo on EVERY rank
    (a per-rank decision made rank 0 pick RCCL and rank 1 gloo: the rendezvous hung)."""
    import bench
    assert [bench.choose_backend(8, r, 8) for r in range(8)] == [("nccl", r) for r in range(8)]
    assert [bench.choose_backend(2, r, 1) for r in range(2)] == [("gloo", 0), ("gloo", 0)]
    assert [bench.choose_backend(4, r, 2) for r in range(4)] == [("gloo", 0), ("gloo", 1), ("gloo", 0), ("gloo", 1)]
    assert bench.choose_backend(1, 0, 1) == ("nccl", 0)
    assert bench.choose_backend(2, 1, 8, "gloo") == ("gloo", 1)


def _run_bench(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True,
                          env=env, timeout=600)


def test_bench_spawns_its_own_ranks_end_to_end_on_gloo():
    """`python bench.py --gpus 2` with no launcher: the parent starts two rank processes (no GPU call, no torch import
    in the parent), they rendezvous on 127.0.0.1 (gloo here), run warm-up + timed DDP steps, take the max over ranks,
    and the parent relays rank 0's single JSON line.  --device cpu = launcher rehearsal (dense head only)."""
    out = _run_bench("--gpus", "2", "--config", "tiny", "--device", "cpu", "--steps", "3", "--warmup", "1")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["ranks"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["mode"] == "rehearsal-cpu" and d["config"]["global_batch"] == 4 and d["value"] > 0
    assert abs(d["value"] - 4 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]


def test_bench_parent_never_imports_torch_before_spawning():
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def _late_imports")]
    assert "\nimport torch" not in head and "\nfrom torch" not in head
    main = src[src.index("def main("):]
    assert main.index("spawn_ranks(") < main.index("_late_imports(")


def test_bench_exit_code_reports_a_failed_rank():
    out = _run_bench("--gpus", "2", "--config", "no_such_config", "--device", "cpu", "--steps", "1", "--warmup", "0")
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_shipped_find_db_reports_whether_miopen_used_it():
    """miopen_db.status(): matched only when MIOpen's version equals the shipped files' and MIOpen wrote no files of its own name
    into the directory (another build ignores the shipped ones silently; bench.py then reports miopen_shipped_find_db false)."""
    code = (
        "import os, sys; os.environ.pop('MIOPEN_USER_DB_PATH', None)\n"
        "from mm_training_amd import miopen_db as m\n"
        "assert m.status(warn=False)['matched'] is None\n"
        "assert m.enable(); s = m.status(warn=False); assert s['enabled'] and s['matched'] and s['shipped_build'].startswith('3.5.0-'), s\n"
        "open(os.path.join(m._state['dir'], 'gfx950100.HIP.9_9_9_other.ufdb.txt'), 'w').write('x')\n"
        "s = m.status(); assert s['matched'] is False and s['foreign_files'] == ['gfx950100.HIP.9_9_9_other.ufdb.txt'], s\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr
    assert "NOT used by this MIOpen" in out.stderr
