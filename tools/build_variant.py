#!/usr/bin/env python3
"""Diagnostic builds of libmmt_hip.so for interleaved A/B runs on the GPU box (tools/ab_libs.py, tools/kbench_fused.py):
   python tools/build_variant.py <name> <source.hip> -DFLAG [-DFLAG2 ...]
recompiles ONE source with extra flags, links it with the objects of the regular build and writes
mm_training_amd/variants/libmmt_<name>.so (git-ignored, travels with gpurun)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_training_amd import build as B  # noqa: E402


def main():
    name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build()
    src = os.path.join(B.CSRC, os.path.basename(src))
    out_dir = os.path.join(B.PKG_DIR, "variants")
    os.makedirs(out_dir, exist_ok=True)
    obj = os.path.join(out_dir, f"{name}_{os.path.basename(src)[:-4]}.o")
    cflags = [f for f in B.HIPCC_FLAGS if f != "-shared"]
    subprocess.check_call([B.find_hipcc()] + cflags + flags + ["-c", src, "-o", obj])
    objs = [obj if os.path.basename(s) == os.path.basename(src) else B._obj_path(s) for s in B.sources()]
    lib = os.path.join(out_dir, f"libmmt_{name}.so")
    subprocess.check_call([B.find_hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}"] + objs + ["-o", lib])
    print(lib)


if __name__ == "__main__":
    main()
