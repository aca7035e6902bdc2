"""Build recipe for the native library: hipcc -> mm_training_amd/libmmt_hip.so (gfx950).

In-tree on purpose: the .so is git-ignored but travels with the repo snapshot to
the GPU box.  `python -m mm_training_amd.build` or `__graft_entry__.build()`.
"""
import glob
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_NAME = "libmmt_hip.so"
LIB_PATH = os.path.join(PKG_DIR, LIB_NAME)
ARCH = "gfx950"

HIPCC_FLAGS = [
    "-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}",
    # no fast-math anywhere: the integer index path needs IEEE fp32 divide and
    # un-contracted multiply/add (bit-exact with the oracle)
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall", "-Wno-unused-result",
]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libmmt_hip.so)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(PKG_DIR, "..", "include", "*.h")) + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    """Compile every HIP source into one shared library. Returns the .so path."""
    if not force and not is_stale():
        return LIB_PATH
    cmd = [find_hipcc()] + HIPCC_FLAGS + list(extra_flags) + sources() + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
