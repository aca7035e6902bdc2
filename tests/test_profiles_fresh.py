"""RELEASE GATE (opt-in: MMT_RELEASE_GATE=1): the committed profiler summaries describe the committed kernels.
profiles/README.md says which command produced each file; this check fails when any HIP source was committed AFTER the
newest round's kernel statistics / PMC summaries (the round-2 review found three kernel changes behind a profiles directory
that claimed to be current).  It is not a unit test -- every local kernel edit, comment-only commit or rebase would turn
the default suite red until the GPU profiles are collected again -- so it only runs when asked for, at the end of a round
(tools/collect_profiles.sh, copy to profiles/, commit, then `MMT_RELEASE_GATE=1 python -m pytest tests/test_profiles_fresh.py`)."""
import glob
import os
import subprocess

import pytest

import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(os.environ.get("MMT_RELEASE_GATE", "0") != "1", reason="release gate: set MMT_RELEASE_GATE=1")


def newest_round():
    """'rNN' of the newest round that has records under profiles/."""
    rounds = sorted({m.group(1) for f in os.listdir(os.path.join(ROOT, "profiles")) for m in [re.match(r"(r\d\d)_", f)] if m})
    return rounds[-1] if rounds else None



def _commit_time(path):
    out = subprocess.run(["git", "log", "-1", "--format=%ct", "--", path], cwd=ROOT, capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip("no git history here")
    return int(out.stdout.strip()) if out.stdout.strip() else None


def test_profiles_are_not_older_than_the_kernels():
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("not a git checkout")
    ROUND = newest_round()
    assert ROUND is not None, "no rNN_* records under profiles/"
    records = sorted(glob.glob(os.path.join(ROOT, "profiles", f"{ROUND}_*kernel_stats.csv")) +
                     glob.glob(os.path.join(ROOT, "profiles", f"{ROUND}_pmc_*.json")))
    assert records, f"no {ROUND} kernel statistics / PMC summaries under profiles/"
    rec_times = {os.path.basename(r): _commit_time(os.path.relpath(r, ROOT)) for r in records}
    assert all(t is not None for t in rec_times.values()), f"uncommitted profile records: {[k for k, v in rec_times.items() if v is None]}"
    oldest = min(rec_times.values())
    late = []
    for src in sorted(glob.glob(os.path.join(ROOT, "mm_training_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "mm_training_amd", "csrc", "*.h"))):
        t = _commit_time(os.path.relpath(src, ROOT))
        if t is not None and t > oldest:
            late.append((os.path.basename(src), t - oldest))
    assert not late, f"kernel sources committed after the {ROUND} profile records (seconds later): {late}; re-run tools/collect_profiles.sh"
    # and the working tree holds no uncommitted kernel change
    dirty = subprocess.run(["git", "status", "--porcelain", "--", "mm_training_amd/csrc"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    assert not dirty, f"uncommitted kernel changes: {dirty}"
