"""The shipped MIOpen user find / perf database (text, ~200 KB; README.md here says what produced it).

`enable()` -- before the first convolution of the process -- gives the dense nets of the BASELINE configurations MIOpen's measured
solver choices instead of its immediate-mode heuristics (configs[3]: 60.9 against 49.6 samples/s; configs[4]: 44.9 against 38.0;
configs[2]: 166.0 against 131.6).  `bench.py` calls it; a training script that uses `mm_training_amd.dp.TrainStep` on those shapes
should too, together with `torch.backends.cudnn.benchmark = True` (PyTorch then asks MIOpen's find API, which is answered from the
database without running anything; a shape that is not in it is timed once, in hybrid find mode).
"""
import os
import shutil
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


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
    # HYBRID find mode: a find-DB hit returns the tuned solver without running anything, a miss times the
    # applicable solvers once (seconds) instead of trusting the immediate-mode heuristic.  The reference "naive"
    # solvers (tens of ms per call, never chosen) are excluded from that timing: they alone cost ~15 s of warm-up
    # per process (profiles/r01_miopen_find_modes.txt).
    os.environ.setdefault("MIOPEN_FIND_MODE", "3")
    for d in ("FWD", "BWD", "WRW"):
        os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + d, "0")
    return True
