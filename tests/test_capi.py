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
    assert lib.mmt_voxel_pooling_forward_ex(1, 8, 4, 2, 2, 1, p, p, p, p, 0x100, None) == -4          # chunk of 4 points: < 64
    assert lib.mmt_voxel_pooling_forward_ex(1, 8, 4, 2, 2, 1, p, p, p, p, 3 | 0x20 | (58 << 8), None) == -4   # excludes CHUNK_1024
    assert lib.mmt_voxel_pooling_forward_ex(1, 8, 4, 2, 2, 1, p, p, p, p, 0x10000, None) == -4        # unknown bit
    assert lib.mmt_voxel_pooling_forward(70000, 70000, 4, 2, 2, 1, p, p, p, p, None) == -3
    assert lib.mmt_voxel_pooling_backward(1, 8, 4, 2, 2, None, p, 1, 1, 1, 1, p, None, 0, None) == -1
    assert lib.mmt_voxel_pooling_backward_workspace_elems(2, 10, 4, 3, 5) == 2 * 5 * 3 * 4 + 20
    assert lib.mmt_arm_kernel_timing(p, None) == -1               # both events or none
    assert lib.mmt_arm_kernel_timing(None, None) == 0
    assert lib.mmt_timing_event_create(None) == -1
    assert lib.mmt_timing_elapsed_ms(None, None, None) == -1
    assert lib.mmt_simple_vfe(5, 15, 5, 6, p, p, p, None) == -2
    with pytest.raises(mmt_lib.MmtError, match="NULL"):
        mmt_lib.call("mmt_quantize_geometry", 4, None, None, None, None, None)


def test_host_side_argument_checks_of_the_fused_lift_splat(mmt_lib):
    """mmt_lss_splat_forward / _backward: shapes, flags and strides are refused on the host before any launch."""
    lib = mmt_lib.lib()
    buf = (ctypes.c_float * 1024)()
    p = ctypes.cast(ctypes.addressof(buf) + (-ctypes.addressof(buf)) % 16, ctypes.c_void_p)       # 16-byte aligned
    PM, TILES, COLUMN = 0x100, 0x200, 0x400
    fwd = lambda C, flags: lib.mmt_lss_splat_forward(1, 1, 4, 2, 2, C, 4, 4, 1, p, p, p, p, None, flags, None)
    assert fwd(6, 0) == -2                                   # C % 4
    assert fwd(16, 0x2000) == -4 and b"unknown flag" in lib.mmt_last_error()          # (0x800 = MMT_LSS_ZERO_OUTPUT since ABI 6)
    assert fwd(16, COLUMN) == -4                             # the column kernel is a backward kernel
    assert lib.mmt_lss_splat_forward(0, 1, 4, 2, 2, 16, 4, 4, 1, p, p, p, p, None, 0, None) == -2
    assert lib.mmt_lss_splat_forward(1, 1, 4, 2, 2, 16, 4, 4, 1, None, p, p, p, None, PM, None) == -1
    bwd = lambda C, sc, flags: lib.mmt_lss_splat_backward(1, 1, 4, 2, 2, C, 4, 4, 1, p, p, p, p, 16 * C, sc, 4 * C, C, p, p, flags, None)
    assert bwd(24, 1, 0) == -2                               # C % 16
    assert bwd(16, 2, 0) == -2 and b"channels-last" in lib.mmt_last_error()
    assert bwd(16, 1, 0x1000) == -4
    assert bwd(16, 1, PM | TILES | COLUMN | 0x800) == -4
    assert lib.mmt_lss_splat_backward(1, 1, 4, 2, 2, 16, 4, 4, 1, p, p, p, None, 256, 1, 64, 16, p, p, COLUMN, None) == -1
    # camera form (ABI 6): NULL geometry operands, the tile-kernel flag and unsupported widths are refused before any launch
    f3 = (ctypes.c_float * 3)(0.4, 0.4, 0.4)
    cam = lambda C, flags, combine=p: lib.mmt_lss_splat_forward_cam(1, 1, 4, 2, 2, C, 4, 4, 1, combine, p, p, p, f3, f3, p, p, p, None, None, None, 0, flags, None)
    assert cam(64, 0, None) == -1 and cam(64, TILES) == -4 and cam(48, 0) == -2 and cam(64, 0x2000) == -4
    assert cam(64, 0x1000) == -1                                  # MMT_LSS_SUMMARY_CACHED without a column summary
    assert lib.mmt_lss_splat_backward_cam(1, 1, 4, 2, 2, 64, 4, 4, 1, p, p, p, p, f3, None, p, p, p, 1024, 1, 256, 64, p, p, None, None, 0, None) == -1
    assert lib.mmt_lss_splat_backward_cam(1, 1, 4, 2, 2, 64, 4, 4, 1, p, p, p, p, f3, f3, p, p, p, 1024, 1, 256, 64, p, p, None, None, TILES, None) == -4
    assert lib.mmt_lss_camera_form_supported(4, 6, 112, 16, 44, 80) == 1 and lib.mmt_lss_camera_form_supported(1, 1, 8, 4, 4, 48) == 0
    assert lib.mmt_lss_exclusive_cache_bytes(6, 128, 128, 4) == 4 * (64 + 8 * 136 + 4 * (4 + 96 + 128 * 128)) and lib.mmt_lss_exclusive_cache_bytes(6, 128, 0, 4) == 0
    assert lib.mmt_lss_last_kernel_family(0) & 0xF in (0, 1, 2) and lib.mmt_lss_last_kernel_family(1) in (0, 1, 2, 3, 0x11, 0x13)


def test_host_side_argument_checks_of_the_additional_entry_points(mmt_lib):
    """Cached plan, BEV warp, depth labels, CenterPoint targets, BatchNorm: refused on the host, no launch."""
    lib = mmt_lib.lib()
    buf = (ctypes.c_float * 256)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.mmt_voxel_pooling_plan_elems(0, 8, 2, 2) == -1
    assert lib.mmt_voxel_pooling_plan_elems(2, 100, 4, 4) > 2 * 100
    assert lib.mmt_voxel_pooling_plan_workspace_bytes(2, 100, 4, 4) > 3 * 4 * 200
    assert lib.mmt_voxel_pooling_plan_build(1, 8, 2, 2, 1, None, None, p, 10, p, 10, None) == -1
    assert lib.mmt_voxel_pooling_plan_build(1, 8, 2, 2, 1, p, None, p, 10, p, 1 << 20, None) == -5      # plan too small
    assert lib.mmt_voxel_pooling_forward_planned(1, 8, 6, 2, 2, p, 4, 0, 0, p, p, 6, None, 0, None) == -2   # C % 4
    assert lib.mmt_voxel_pooling_forward_planned(1, 8, 8, 2, 2, p, 3, 0, 0, p, p, 8, None, 0, None) == -2   # items < cells
    assert lib.mmt_bev_warp_affine(1, 4, 4, 6, p, p, 6, p, 6, None) == -2
    assert lib.mmt_bev_warp_affine(1, 4, 4, 8, p, None, 8, p, 8, None) == -1
    assert lib.mmt_bev_warp_affine_backward(1, 4, 4, 8, p, p, 4, p, 8, None) == -2                      # stride < C
    assert lib.mmt_depth_labels_workspace_elems(2, 6, 256, 704, 16) == 2 * 6 * 16 * 44
    assert lib.mmt_depth_labels(1, 2, 5, 10, 60, 96, 16, 2.0, 0.5, 112, p, p, p, p, p, p, 1 << 20, p, p, None) == -2   # H % ds
    assert lib.mmt_depth_labels(1, 2, 5, 10, 64, 96, 16, 2.0, 0.5, 112, p, p, p, p, p, p, 3, p, p, None) == -5       # workspace
    assert lib.mmt_depth_labels(1, 2, 5, 10, 64, 96, 16, 2.0, 0.5, 112, p, p, p, p, p, p, 1 << 20, None, None, None) == -1
    i32 = (ctypes.c_int32 * 2)(0, 1)
    ptrs = (ctypes.c_void_p * 2)(p.value, p.value)
    cp = lambda tasks, hm: lib.mmt_centerpoint_targets(
        1, tasks, ctypes.cast(i32, ctypes.c_void_p), ctypes.cast(i32, ctypes.c_void_p), 8, 4, 16, 16, 0.0, 0.0, 0.2, 0.2,
        4, 0.1, 2, 1, p, p, p, hm, ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(ptrs, ctypes.c_void_p),
        ctypes.cast(ptrs, ctypes.c_void_p), None)
    assert cp(9, ctypes.cast(ptrs, ctypes.c_void_p)) == -2        # more than 8 tasks
    assert cp(1, None) == -1
    assert lib.mmt_bn_workspace_elems(64) >= 4 * 64 and lib.mmt_bn_workspace_elems(0) == -1
    assert lib.mmt_bn_relu_forward(16, 6, p, None, p, p, p, p, 0.1, 1e-5, 1, p, p, p, None) == -2        # C % 4
    assert lib.mmt_bn_relu_forward(16, 1028, p, None, p, p, p, p, 0.1, 1e-5, 1, p, p, p, None) == -2     # unsupported C
    assert lib.mmt_bn_relu_forward(16, 8, None, None, p, p, p, p, 0.1, 1e-5, 1, p, p, p, None) == -1
    assert lib.mmt_bn_relu_backward(16, 8, p, None, p, p, 1, 1, p, p, p, p, p, None) == -1                # y needed
    assert b"y" in lib.mmt_last_error()


def test_python_mirror_rejects_cpu_tensors(mmt_lib):
    import torch
    from mm_training_amd.ops.voxel_pooling import voxel_pooling
    geom = torch.zeros(1, 4, 3, dtype=torch.int32)
    feats = torch.zeros(1, 4, 8)
    # same failure mode as CHECK_CUDA in voxel_pooling_forward.cpp:10-16; no CPU fallback
    with pytest.raises(RuntimeError, match="CUDA"):
        voxel_pooling(geom, feats, torch.tensor([2, 2, 1]))


def test_missing_library_fails_loudly():
    """No CPU fallback: without libmmt_hip.so the product path raises ImportError (it never routes
    through the oracle or a torch restatement)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from mm_training_amd import _lib\n"
            "try:\n"
            "    _lib.lib()\n"
            "except ImportError as e:\n"
            "    assert 'no CPU fallback' in str(e); print('IMPORT_ERROR_OK')\n" % root)
    env = dict(os.environ, MMT_HIP_LIB="/nonexistent/libmmt_hip.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "IMPORT_ERROR_OK" in out.stdout, out.stdout + out.stderr
    # and the product modules never import the oracle
    import re
    for dirpath, _, files in os.walk(os.path.join(root, "mm_training_amd")):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import oracle|from oracle)", text, re.M), os.path.join(dirpath, f)


def test_host_side_argument_checks_of_the_strided_pillar_scatter(mmt_lib):
    """mmt_pillar_scatter_nhwc[_table]_strided[_backward] (ABI 8): strides that do not divide the grid, a row stride below C or
    not a multiple of 4, NULL operands -- refused on the host before any launch."""
    lib = mmt_lib.lib()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.mmt_pillar_scatter_nhwc_table_strided(8, 1, 16, 16, 100, 4, 4, None, p, p, 8, None) == -1
    assert lib.mmt_pillar_scatter_nhwc_table_strided(8, 1, 16, 16, 100, 3, 4, p, p, p, 8, None) == -2 and b"divide" in lib.mmt_last_error()
    assert lib.mmt_pillar_scatter_nhwc_table_strided(8, 1, 16, 16, 100, 4, 4, p, p, p, 4, None) == -2      # row stride < C
    assert lib.mmt_pillar_scatter_nhwc_table_strided(8, 1, 16, 16, 100, 4, 4, p, p, p, 10, None) == -2     # not a multiple of 4
    assert lib.mmt_pillar_scatter_nhwc_table_strided(6, 1, 16, 16, 100, 4, 4, p, p, p, 8, None) == -2      # C % 4
    assert lib.mmt_pillar_scatter_nhwc_table_strided(8, 1, 16, 16, 0, 4, 4, p, p, p, 8, None) == -2        # max_voxels <= 0 (ABI 11: no upper limit, the directory holds full 32-bit voxel ids)
    assert lib.mmt_pillar_scatter_nhwc_strided(4, 8, 1, 16, 16, 4, 4, p, p, p, 8, None, None) == -1
    assert lib.mmt_pillar_scatter_nhwc_strided(4, 8, 1, 16, 16, 0, 4, p, p, p, 8, p, None) == -2
    assert lib.mmt_pillar_scatter_nhwc_strided_backward(0, 8, 1, 16, 16, 4, 4, None, 8, None, None, None, None) == 0   # nothing to do
    assert lib.mmt_pillar_scatter_nhwc_strided_backward(4, 8, 1, 16, 16, 4, 4, p, 8, None, None, p, None) == -1
    assert lib.mmt_pillar_scatter_nhwc_strided_backward(4, 8, 1, 16, 16, 5, 4, p, 8, p, None, p, None) == -2


def test_voxelizer_size_functions(mmt_lib):
    """mmt_voxelize_table_elems / _scratch_elems / _workspace_elems (ABI 11: they take the point count and max_points): zero for bad
    arguments, growing with the cloud, the table below the dense 8-bytes-per-cell table of the chain form, the workspace their sum."""
    lib = mmt_lib.lib()
    g512, g_native = mmt_lib.int3([512, 512, 1]), mmt_lib.int3([2048, 256, 1])
    assert lib.mmt_voxelize_table_elems(0, g512, 1000) == 0 and lib.mmt_voxelize_table_elems(4, None, 1000) == 0
    assert lib.mmt_voxelize_scratch_elems(4, g512, -1, 15) == 0 and lib.mmt_voxelize_scratch_elems(4, g512, 1000, 0) == 0
    t1, t2 = lib.mmt_voxelize_table_elems(4, g512, 160000), lib.mmt_voxelize_table_elems(4, g512, 320000)
    assert t2 - t1 == 160000                                             # one voxel id per point on top of the directory
    cells = 4 * 512 * 512
    assert cells < t1 - 160000 < cells * 1.05 < 2 * cells                # bits + ordinals + one head per cell
    s1 = lib.mmt_voxelize_scratch_elems(4, g512, 160000, 15)
    assert 14 * 160000 < s1 < 16 * 160000 + 4 * 700 * 128                # the heads' lists dominate
    assert lib.mmt_voxelize_scratch_elems(4, g512, 160000, 1) < 160000 + 4 * 700 * 128
    assert lib.mmt_voxelize_workspace_elems(8, 320000, g_native, 15) == lib.mmt_voxelize_table_elems(8, g_native, 320000) + \
        lib.mmt_voxelize_scratch_elems(8, g_native, 320000, 15) + 2
