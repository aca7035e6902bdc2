"""The shipped MIOpen user find / perf database (text, ~200 KB; README.md here says what produced it).

`enable()` -- before the first convolution of the process -- gives the dense nets of the BASELINE configurations MIOpen's measured
solver choices instead of its immediate-mode heuristics (configs[3]: 60.9 against 49.6 samples/s; configs[4]: 44.9 against 38.0;
configs[2]: 166.0 against 131.6).  `bench.py` calls it; a training script that uses `mm_training_amd.dp.TrainStep` on those shapes
should too, together with `torch.backends.cudnn.benchmark = True` (PyTorch then asks MIOpen's find API, which is answered from the
database without running anything; a shape that is not in it is timed once, in hybrid find mode).
"""
import os
import re
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
_state = {"dir": None, "files": []}


def enable():
    """Copy the database to a private directory (MIOpen rewrites the files) and point MIOPEN_USER_DB_PATH at it.  Returns False and
    changes nothing when the user has set MIOPEN_USER_DB_PATH already or the files are missing."""
    files = [f for f in os.listdir(HERE) if f.endswith(".txt")]
    if "MIOPEN_USER_DB_PATH" in os.environ or not files:
        return False
    dst = tempfile.mkdtemp(prefix="mmt_miopen_db_")
    for f in files:
        shutil.copy(os.path.join(HERE, f), os.path.join(dst, f))
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    _state["dir"], _state["files"] = dst, sorted(files)
    # HYBRID find mode: a find-DB hit returns the tuned solver without running anything, a miss times the
    # applicable solvers once (seconds) instead of trusting the immediate-mode heuristic.  The reference "naive"
    # solvers (tens of ms per call, never chosen) are excluded from that timing: they alone cost ~15 s of warm-up
    # per process (profiles/r01_miopen_find_modes.txt).
    os.environ.setdefault("MIOPEN_FIND_MODE", "3")
    for d in ("FWD", "BWD", "WRW"):
        os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + d, "0")
    return True


def status(warn=True):
    """Did MIOpen actually read the shipped database?  Its file names carry MIOpen's build string
    (`gfx950100.HIP.3_5_0_<build>.ufdb.txt`): another build ignores them, times every shape itself (or, in immediate mode, falls
    back to its heuristics: ~50 instead of ~61 samples/s at BASELINE configs[3]) and writes files of ITS name into the directory.
    Call after the first training steps.  Returns {"enabled", "matched", "shipped_build", "miopen_version", "foreign_files"};
    `matched` is None when `enable()` did not run.  A mismatch is reported on stderr (once) -- the numbers of that process are
    not the tuned ones."""
    out = {"enabled": _state["dir"] is not None, "matched": None, "shipped_build": None, "miopen_version": None, "foreign_files": []}
    if _state["dir"] is None:
        return out
    m = re.match(r"[^.]+\.HIP\.(\d+)_(\d+)_(\d+)_(.+?)\.u", _state["files"][0])
    shipped = (int(m.group(1)), int(m.group(2)), int(m.group(3))) if m else None
    out["shipped_build"] = "%d.%d.%d-%s" % (shipped + (m.group(4),)) if m else None
    version_ok = True
    try:
        import torch
        v = torch.backends.cudnn.version()                    # MIOpen's version on ROCm: major * 1e6 + minor * 1e3 + patch
        if v:
            cur = (v // 1000000, (v // 1000) % 1000, v % 1000)
            out["miopen_version"] = "%d.%d.%d" % cur
            version_ok = shipped is None or cur == shipped
    except Exception:
        pass
    try:
        now = sorted(f for f in os.listdir(_state["dir"]) if f.endswith(".txt"))
    except OSError:
        now = list(_state["files"])
    out["foreign_files"] = [f for f in now if f not in _state["files"]]
    out["matched"] = bool(version_ok and not out["foreign_files"])
    if warn and not out["matched"] and not _state.get("warned"):
        _state["warned"] = True
        print("mm_training_amd.miopen_db: the shipped find database (%s) was NOT used by this MIOpen (%s; files it wrote itself: %s) -- "
              "convolution solvers are untuned in this process" % (out["shipped_build"], out["miopen_version"], out["foreign_files"] or "none"),
              file=sys.stderr, flush=True)
    return out
