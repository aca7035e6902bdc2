"""CPU: libmmt_hip.so loads and exports exactly the C ABI declared in include/mmt_hip.h;
argument errors are rejected on the host before anything touches a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mmt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(mmt_lib):
    lib = mmt_lib.lib()
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mmt_hip.h but not exported"
    # and the Python binding table covers every declared symbol
    assert sorted(mmt_lib.SIGNATURES) == names


def test_abi_version(mmt_lib):
    text = open(os.path.join(ROOT, "include", "mmt_hip.h")).read()
    ver = int(re.search(r"#define MMT_ABI_VERSION (\d+)", text).group(1))
    assert mmt_lib.lib().mmt_abi_version() == ver


def test_host_side_argument_checks(mmt_lib):
    lib = mmt_lib.lib()
    # NULL pointers / bad shapes are refused before any HIP call (safe without a GPU)
    assert lib.mmt_voxel_pooling_forward(1, 8, 4, 2, 2, 1, None, None, None, None, None) == -1
    assert b"NULL" in lib.mmt_last_error()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.mmt_voxel_pooling_forward(0, 8, 4, 2, 2, 1, p, p, p, p, None) == -2
    assert lib.mmt_voxel_pooling_forward_ex(1, 8, 4, 2, 2, 1, p, p, p, p, 7, None) == -4
    assert lib.mmt_voxel_pooling_forward_ex(1, 8, 4, 2, 2, 1, p, p, p, p, 0x100, None) == -4
    assert lib.mmt_voxel_pooling_forward(70000, 70000, 4, 2, 2, 1, p, p, p, p, None) == -3
    assert lib.mmt_voxel_pooling_backward(1, 8, 4, 2, 2, None, p, 1, 1, 1, 1, p, None, 0, None) == -1
    assert lib.mmt_voxel_pooling_backward_workspace_elems(2, 10, 4, 3, 5) == 2 * 5 * 3 * 4 + 20
    assert lib.mmt_simple_vfe(5, 15, 5, 6, p, p, p, None) == -2
    with pytest.raises(mmt_lib.MmtError, match="NULL"):
        mmt_lib.call("mmt_quantize_geometry", 4, None, None, None, None, None)


def test_python_mirror_rejects_cpu_tensors(mmt_lib):
    import torch
    from mm_training_amd.ops.voxel_pooling import voxel_pooling
    geom = torch.zeros(1, 4, 3, dtype=torch.int32)
    feats = torch.zeros(1, 4, 8)
    # same failure mode as CHECK_CUDA in voxel_pooling_forward.cpp:10-16; no CPU fallback
    with pytest.raises(RuntimeError, match="CUDA"):
        voxel_pooling(geom, feats, torch.tensor([2, 2, 1]))
