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


def _obj_path(src):
    return os.path.join(CSRC, "build", os.path.basename(src)[:-4] + ".o")


def _obj_stale(src, headers_mtime):
    obj = _obj_path(src)
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return os.path.getmtime(src) > t or headers_mtime > t


def build(force=False, verbose=False, extra_flags=(), jobs=None):
    """Compile every HIP source (one object per file, in parallel, rebuilt only when the file or a
    header changed) and link them into one shared library. Returns the .so path."""
    if not force and not is_stale() and not extra_flags:
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    hipcc = find_hipcc()
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG_DIR, "..", "include", "*.h")) + \
        [os.path.abspath(__file__)]
    hm = max(os.path.getmtime(h) for h in hdrs)
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags)
    todo = [src for src in sources() if force or extra_flags or _obj_stale(src, hm)]

    def compile_one(src):
        cmd = [hipcc] + cflags + ["-c", src, "-o", _obj_path(src)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}"] + [_obj_path(s) for s in sources()] + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
