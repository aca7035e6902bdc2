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


def _obj_dir(extra_flags=()):
    """Objects of the regular build live in csrc/build; a build with extra flags (diagnostic variants such as
    -DLSS_STAMPS) gets a directory of its own, named by a hash of the flags, so it can never be mistaken for -- or
    overwrite -- the objects the product library is linked from."""
    if not extra_flags:
        return os.path.join(CSRC, "build")
    import hashlib
    tag = hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:10]
    return os.path.join(CSRC, "build", "flags_" + tag)


def _obj_path(src, extra_flags=()):
    return os.path.join(_obj_dir(extra_flags), os.path.basename(src)[:-4] + ".o")


def variant_lib_path(extra_flags):
    """Where build(extra_flags=...) writes its library (never LIB_PATH)."""
    return os.path.join(_obj_dir(extra_flags), LIB_NAME)


def _obj_stale(src, headers_mtime, extra_flags=()):
    obj = _obj_path(src, extra_flags)
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return os.path.getmtime(src) > t or headers_mtime > t


def build(force=False, verbose=False, extra_flags=(), jobs=None):
    """Compile every HIP source (one object per file, in parallel, rebuilt only when the file or a
    header changed) and link them into one shared library. Returns the .so path: LIB_PATH for the regular
    build, variant_lib_path(extra_flags) for a build with extra flags (load it with MMT_HIP_LIB)."""
    extra_flags = tuple(extra_flags)
    out = variant_lib_path(extra_flags) if extra_flags else LIB_PATH
    have_objs = all(os.path.exists(_obj_path(s, extra_flags)) for s in sources())
    if not force and not extra_flags and not is_stale() and have_objs:
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    hipcc = find_hipcc()
    os.makedirs(_obj_dir(extra_flags), exist_ok=True)
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG_DIR, "..", "include", "*.h")) + \
        [os.path.abspath(__file__)]
    hm = max(os.path.getmtime(h) for h in hdrs)
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags)
    todo = [src for src in sources() if force or _obj_stale(src, hm, extra_flags)]

    def compile_one(src):
        cmd = [hipcc] + cflags + ["-c", src, "-o", _obj_path(src, extra_flags)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}"] + [_obj_path(s, extra_flags) for s in sources()] + ["-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
