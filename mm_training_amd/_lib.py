"""ctypes binding of libmmt_hip.so (the C ABI declared in include/mmt_hip.h).

There is deliberately no CPU fallback: if the library is missing or a launch
fails, callers get an exception.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# MMT_HIP_LIB: load another build of the same library (interleaved A/B of two builds: tools/ab_libs.py)
LIB_PATH = os.environ.get("MMT_HIP_LIB") or os.path.join(_PKG, "libmmt_hip.so")

_c_int = ctypes.c_int
_c_i64 = ctypes.c_int64
_c_ptr = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/mmt_hip.h one to one
SIGNATURES = {
    "mmt_abi_version": (_c_int, []),
    "mmt_last_error": (ctypes.c_char_p, []),
    "mmt_timing_event_create": (_c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "mmt_timing_event_destroy": (_c_int, [_c_ptr]),
    "mmt_timing_elapsed_ms": (_c_int, [_c_ptr, _c_ptr, ctypes.POINTER(ctypes.c_float)]),
    "mmt_arm_kernel_timing": (_c_int, [_c_ptr, _c_ptr]),
    "mmt_voxel_pooling_forward": (_c_int, [_c_int] * 6 + [_c_ptr] * 4 + [_c_ptr]),
    "mmt_voxel_pooling_forward_ex": (_c_int, [_c_int] * 6 + [_c_ptr] * 4 + [_c_int, _c_ptr]),
    "mmt_voxel_pooling_backward_workspace_elems": (_c_i64, [_c_int] * 5),
    "mmt_voxel_pooling_backward": (_c_int, [_c_int] * 5 + [_c_ptr, _c_ptr] + [_c_i64] * 4 + [_c_ptr, _c_ptr, _c_i64, _c_ptr]),
    "mmt_voxel_pooling_forward_bf16": (_c_int, [_c_int] * 6 + [_c_ptr] * 4 + [_c_int, _c_ptr]),
    "mmt_voxel_pooling_backward_bf16": (_c_int, [_c_int] * 5 + [_c_ptr, _c_ptr] + [_c_i64] * 4 + [_c_ptr, _c_ptr, _c_i64, _c_ptr]),
    "mmt_lift_features_bf16": (_c_int, [_c_int] * 4 + [_c_ptr] * 3 + [_c_ptr]),
    "mmt_lift_features_backward_bf16": (_c_int, [_c_int] * 4 + [_c_ptr] * 5 + [_c_ptr]),
    "mmt_lift_splat_forward_bf16": (_c_int, [_c_int] * 8 + [_c_ptr] * 5 + [_c_int, _c_ptr]),
    "mmt_lift_splat_backward_bf16": (_c_int, [_c_int] * 7 + [_c_ptr] * 4 + [_c_i64] * 4 + [_c_ptr, _c_ptr, _c_ptr]),
    "mmt_lss_splat_forward": (_c_int, [_c_int] * 9 + [_c_ptr] * 5 + [_c_int, _c_ptr]),
    "mmt_lss_splat_forward_bf16": (_c_int, [_c_int] * 9 + [_c_ptr] * 5 + [_c_int, _c_ptr]),
    "mmt_lss_splat_backward": (_c_int, [_c_int] * 9 + [_c_ptr] * 4 + [_c_i64] * 4 + [_c_ptr, _c_ptr, _c_int, _c_ptr]),
    "mmt_lss_splat_backward_bf16": (_c_int, [_c_int] * 9 + [_c_ptr] * 4 + [_c_i64] * 4 + [_c_ptr, _c_ptr, _c_int, _c_ptr]),
    "mmt_lss_splat_forward_cam": (_c_int, [_c_int] * 9 + [_c_ptr] * 12 + [_c_i64, _c_int, _c_ptr]),
    "mmt_lss_splat_forward_cam_bf16": (_c_int, [_c_int] * 9 + [_c_ptr] * 12 + [_c_i64, _c_int, _c_ptr]),
    "mmt_lss_exclusive_cache_bytes": (_c_i64, [_c_int] * 4),
    "mmt_lss_splat_backward_cam": (_c_int, [_c_int] * 9 + [_c_ptr] * 9 + [_c_i64] * 4 + [_c_ptr] * 4 + [_c_int, _c_ptr]),
    "mmt_lss_splat_backward_cam_bf16": (_c_int, [_c_int] * 9 + [_c_ptr] * 9 + [_c_i64] * 4 + [_c_ptr] * 4 + [_c_int, _c_ptr]),
    "mmt_lss_plan_supported": (_c_int, [_c_int] * 9),
    "mmt_lss_plan_cache_bytes": (_c_i64, [_c_int] * 7),
    "mmt_lss_plan_prepare": (_c_int, [_c_int] * 8 + [_c_ptr] * 7 + [_c_i64, _c_ptr]),
    "mmt_depth_softmax_forward_plan_prepare": (_c_int, [_c_i64, _c_int, _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_int]
                                               + [_c_int] * 7 + [_c_ptr] * 7 + [_c_i64, _c_ptr]),
    "mmt_lss_splat_forward_plan": (_c_int, [_c_int] * 9 + [_c_ptr] * 11 + [_c_i64, _c_int, _c_ptr]),
    "mmt_lss_splat_forward_plan_bf16": (_c_int, [_c_int] * 9 + [_c_ptr] * 11 + [_c_i64, _c_int, _c_ptr]),
    "mmt_lss_plan_cache_counters": (_c_int, [_c_ptr, _c_i64, _c_ptr, _c_ptr]),
    "mmt_lss_plan_cache_layout": (_c_int, [_c_int] * 6 + [_c_i64, _c_ptr]),
    "mmt_lss_last_kernel_family": (_c_int, [_c_int]),
    "mmt_lss_camera_form_supported": (_c_int, [_c_int] * 6),
    "mmt_lss_exclusive_cache_used": (_c_int, [_c_int] * 6),
    "mmt_quantize_geometry": (_c_int, [_c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    "mmt_frustum_geometry": (_c_int, [_c_int, _c_i64] + [_c_ptr] * 6 + [_c_ptr]),
    "mmt_depth_softmax_forward": (_c_int, [_c_i64, _c_int, _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_int, _c_ptr]),
    "mmt_depth_softmax_backward": (_c_int, [_c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_i64, _c_ptr, _c_int, _c_ptr]),
    "mmt_lift_features": (_c_int, [_c_int] * 4 + [_c_ptr] * 3 + [_c_ptr]),
    "mmt_lift_features_backward": (_c_int, [_c_int] * 4 + [_c_ptr] * 5 + [_c_ptr]),
    "mmt_lift_splat_forward": (_c_int, [_c_int] * 8 + [_c_ptr] * 5 + [_c_int, _c_ptr]),
    "mmt_lift_splat_backward": (_c_int, [_c_int] * 7 + [_c_ptr] * 4 + [_c_i64] * 4 + [_c_ptr, _c_ptr, _c_ptr]),
    "mmt_voxel_pooling_plan_elems": (_c_i64, [_c_int] * 4),
    "mmt_voxel_pooling_plan_workspace_bytes": (_c_i64, [_c_int] * 4),
    "mmt_voxel_pooling_plan_build": (_c_int, [_c_int] * 5 + [_c_ptr] * 3 + [_c_i64, _c_ptr, _c_i64, _c_ptr]),
    "mmt_voxel_pooling_plan_info": (_c_int, [_c_ptr, _c_ptr, _c_ptr]),
    "mmt_voxel_pooling_forward_planned": (_c_int, [_c_int] * 5 + [_c_ptr] + [_c_int] * 3 + [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr]),
    "mmt_depth_labels_workspace_elems": (_c_i64, [_c_int] * 5),
    "mmt_depth_labels": (_c_int, [_c_int] * 7 + [ctypes.c_float, ctypes.c_float, _c_int] + [_c_ptr] * 6 + [_c_i64, _c_ptr, _c_ptr, _c_ptr]),
    "mmt_depth_labels_flipped": (_c_int, [_c_int] * 7 + [ctypes.c_float, ctypes.c_float, _c_int] + [_c_ptr] * 6 + [_c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    "mmt_hflip": (_c_int, [_c_i64, _c_int, _c_int, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    "mmt_normalize_flip_images": (_c_int, [_c_i64, _c_int, _c_int, _c_int, _c_ptr, ctypes.c_float, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr]),
    "mmt_centerpoint_targets": (_c_int, [_c_int, _c_int, _c_ptr, _c_ptr] + [_c_int] * 4 + [ctypes.c_float] * 4 + [_c_int, ctypes.c_float, _c_int, _c_int] + [_c_ptr] * 7 + [_c_ptr]),
    "mmt_bev_warp_affine": (_c_int, [_c_int] * 4 + [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr]),
    "mmt_bev_warp_affine_backward": (_c_int, [_c_int] * 4 + [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr]),
    "mmt_bev_warp_affine_backward_assign": (_c_int, [_c_int] * 4 + [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr]),
    "mmt_bn_workspace_elems": (_c_i64, [_c_int]),
    "mmt_bn_relu_forward": (_c_int, [_c_i64, _c_int] + [_c_ptr] * 6 + [ctypes.c_float, ctypes.c_float, _c_int] + [_c_ptr] * 3 + [_c_ptr]),
    "mmt_bn_relu_backward": (_c_int, [_c_i64, _c_int] + [_c_ptr] * 4 + [_c_int, _c_int] + [_c_ptr] * 5 + [_c_ptr]),
    "mmt_bn_relu_forward_ex": (_c_int, [_c_i64, _c_int] + [_c_ptr] * 6 + [ctypes.c_float, ctypes.c_float, _c_int] + [_c_ptr] * 3 + [_c_int, _c_ptr]),
    "mmt_bn_relu_backward_ex": (_c_int, [_c_i64, _c_int] + [_c_ptr] * 4 + [_c_int, _c_int] + [_c_ptr] * 5 + [_c_int, _c_ptr]),
    "mmt_add_n": (_c_int, [_c_int, _c_ptr, _c_i64, _c_ptr, _c_ptr]),
    "mmt_head_loss_partials": (_c_int, [_c_int] * 5),
    "mmt_head_loss_forward_backward": (_c_int, [_c_int] * 5 + [_c_ptr] * 7 + [ctypes.c_float, _c_ptr, _c_ptr, _c_int, _c_ptr]),
    "mmt_heads_final_workspace_elems": (_c_i64, [_c_int, _c_int, _c_int]),
    "mmt_heads_final_forward": (_c_int, [_c_int] * 4 + [_c_ptr] * 5 + [_c_int, _c_ptr]),
    "mmt_heads_final_backward": (_c_int, [_c_int] * 4 + [_c_ptr] * 8 + [_c_int, _c_ptr]),
    "mmt_bn_relu_inference": (_c_int, [_c_i64, _c_int] + [_c_ptr] * 6 + [ctypes.c_float, _c_int, _c_ptr, _c_ptr, _c_int, _c_ptr]),
    "mmt_channel_blocks_split": (_c_int, [_c_i64, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr]),
    "mmt_channel_blocks_gather": (_c_int, [_c_i64, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr]),
    "mmt_clip_adamw_step": (_c_int, [_c_int, _c_int] + [_c_ptr] * 8 + [ctypes.c_double] * 5 + [_c_i64, ctypes.c_float, _c_ptr, _c_ptr, _c_ptr]),
    "mmt_bn_relu_backward_ex2": (_c_int, [_c_i64, _c_int] + [_c_ptr] * 5 + [_c_i64] + [_c_ptr] + [_c_int, _c_int] + [_c_ptr] * 5 + [_c_int, _c_ptr]),
    "mmt_dcn_im2col": (_c_int, [_c_int] * 5 + [_c_ptr] * 3 + [_c_ptr]),
    "mmt_dcn_col2im": (_c_int, [_c_int] * 5 + [_c_ptr] * 5 + [_c_ptr]),
    "mmt_dcn_col2im_workspace_elems": (_c_i64, [_c_int] * 3),
    "mmt_dcn_col2im_sorted": (_c_int, [_c_int] * 5 + [_c_ptr] * 6 + [_c_i64, _c_ptr]),
    "mmt_dcn_mfma_supported": (_c_int, [_c_int] * 6),
    "mmt_dcn_backward_form": (_c_int, [_c_int] * 6),
    "mmt_dcn_mfma_workspace_bytes": (_c_i64, [_c_int] * 6),
    "mmt_dcn_forward": (_c_int, [_c_int] * 6 + [_c_ptr] * 5 + [_c_i64, _c_int, _c_ptr]),
    "mmt_dcn_backward": (_c_int, [_c_int] * 6 + [_c_ptr] * 8 + [_c_i64, _c_ptr]),
    "mmt_voxelize_workspace_elems": (_c_i64, [_c_int, _c_i64, _c_ptr, _c_int]),
    "mmt_voxelize_fused_launch": (_c_int, [_c_int]),
    "mmt_voxelize_table_elems": (_c_i64, [_c_int, _c_ptr, _c_i64]),
    "mmt_voxelize_scratch_elems": (_c_i64, [_c_int, _c_ptr, _c_i64, _c_int]),
    "mmt_hard_voxelize_mean": (_c_int, [_c_int, _c_i64, _c_int] + [_c_ptr] * 5 + [_c_int, _c_int, _c_int] + [_c_ptr] * 7 + [_c_ptr]),
    "mmt_hard_voxelize": (_c_int, [_c_int, _c_i64, _c_int] + [_c_ptr] * 5 + [_c_int, _c_int] + [_c_ptr] * 5 + [_c_ptr]),
    "mmt_compact_voxels": (_c_int, [_c_int] * 3 + [_c_ptr] * 8 + [_c_ptr]),
    "mmt_simple_vfe": (_c_int, [_c_i64, _c_int, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    "mmt_pillar_scatter": (_c_int, [_c_i64] + [_c_int] * 4 + [_c_ptr] * 4 + [_c_ptr]),
    "mmt_pillar_scatter_backward": (_c_int, [_c_i64] + [_c_int] * 4 + [_c_ptr] * 4 + [_c_ptr]),
    "mmt_pillar_scatter_nhwc_table": (_c_int, [_c_int] * 5 + [_c_ptr] * 3 + [_c_ptr]),
    "mmt_pillar_scatter_nhwc_unique_backward": (_c_int, [_c_i64] + [_c_int] * 4 + [_c_ptr] * 3 + [_c_ptr]),
    "mmt_pillar_scatter_nhwc_table_strided": (_c_int, [_c_int] * 7 + [_c_ptr] * 3 + [_c_i64, _c_ptr]),
    "mmt_pillar_scatter_nhwc_strided": (_c_int, [_c_i64] + [_c_int] * 6 + [_c_ptr] * 3 + [_c_i64, _c_ptr, _c_ptr]),
    "mmt_pillar_scatter_nhwc_strided_backward": (_c_int, [_c_i64] + [_c_int] * 6 + [_c_ptr, _c_i64] + [_c_ptr] * 3 + [_c_ptr]),
    "mmt_pillar_scatter_nhwc": (_c_int, [_c_i64] + [_c_int] * 4 + [_c_ptr] * 4 + [_c_ptr]),
    "mmt_pillar_scatter_nhwc_backward": (_c_int, [_c_i64] + [_c_int] * 4 + [_c_ptr] * 4 + [_c_ptr]),
}

DTYPE_F32, DTYPE_BF16 = 0, 1   # MMT_DTYPE_*
# flags of mmt_voxel_pooling_forward_ex (include/mmt_hip.h)
VP_ALGO_AUTO = 0
VP_ALGO_ROW_ATOMIC = 1
VP_WRITE_DROPPED = 0x10
LSS_PIXEL_MAJOR = 0x100       # mmt_lss_splat_*: geom / depth / grad_depth in [B*N, fH, fW, D(, 3)] order
LSS_TILE_KERNELS = 0x200      # mmt_lss_splat_*: frustum-tile kernels instead of the ray walks
LSS_COLUMN_BACKWARD = 0x400   # mmt_lss_splat_backward*: matrix-core column kernel (level rigs)
LSS_ZERO_OUTPUT = 0x800       # mmt_lss_splat_forward*: the call zero-fills the BEV map itself (write-through stores)
LSS_SUMMARY_CACHED = 0x1000   # mmt_lss_splat_forward_cam*: read the column summary instead of computing the geometry
LSS_STATS_SLOTS = 64          # column_stats of mmt_lss_splat_backward_cam*: int64 [2 * LSS_STATS_SLOTS], (mismatching, kept) pairs
LSS_PLAN_PREPARED = 0x2000    # mmt_lss_splat_forward_plan*: mmt_lss_plan_prepare already ran for this batch
LSS_PLAN_BRUTE = 0x4000       # mmt_lss_splat_forward_plan*: brute-force path for every sample (tests)
LSS_FAMILY = {0: "none", 1: "ray", 2: "tile", 3: "column", 4: "plan"}     # mmt_lss_last_kernel_family() & 0xF; | 0x10 = camera form
LSS_FAMILY_REGISTER, LSS_FAMILY_EXCLUSIVE, LSS_FAMILY_BLOCK = 0x20, 0x40, 0x80      # forward: register walk / an exclusive-cell cache was used / block walk

_lib = None


class MmtError(RuntimeError):
    """A libmmt_hip entry point returned non-zero."""


def lib():
    """Load libmmt_hip.so (once). Raises if it has not been built."""
    global _lib
    if _lib is None:
        # torch ships its own libamdhip64.so.7; it must be in the process BEFORE this
        # library is loaded so both share one HIP runtime (streams, device pointers).
        # Loaded the other way round, libmmt_hip binds /opt/rocm's copy and every launch
        # fails with "no ROCm-capable device is detected".
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m mm_training_amd.build` "
                "(hipcc, gfx950). mm_training_amd has no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


# ---- cheap host-side plumbing for the per-layer wrappers (ops/bn_relu.py, ops/conv_overlap.py: ~250 calls each per training step;
# BASELINE configs[4] is bound by the host's issue rate, tools/scratch/host_vs_gpu.py) -------------------------------------------
class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def raw_stream(device=None):
    """Handle of the current HIP stream of `device` (None: the current device) -- `torch.cuda.current_stream(device).cuda_stream`
    without building the Stream object (8 us -> under 1 us)."""
    import torch
    index = device.index if device is not None and device.index is not None else torch._C._cuda_getDevice()
    return torch._C._cuda_getCurrentRawStream(index)


def on_device(device):
    """`with on_device(t.device):` = `with torch.cuda.device(t.device):` that costs nothing when `device` is the current one already
    (the usual case: one process per GPU; the autograd engine sets the device for its backward thread)."""
    import torch
    if device.index is None or torch._C._cuda_getDevice() == device.index:
        return _NO_GUARD
    return torch.cuda.device(device)


_bound_apply = {}


def apply_function(fn_class, *args):
    """`fn_class.apply(*args)` for a plain torch.autograd.Function (no setup_context; not for use under a functorch transform):
    the C++ `apply` bound to the class once, past Function.apply's per-call signature inspection and wrapper scan (2-3 us of host
    time per layer)."""
    f = _bound_apply.get(fn_class)
    if f is None:
        import torch
        f = _bound_apply[fn_class] = torch._C._FunctionBase.__dict__["apply"].__get__(None, fn_class)
    return f(*args)


def call(name, *args):
    """Call an int-returning entry point; raise MmtError with the library's message."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        msg = lib().mmt_last_error()
        raise MmtError(f"{name} failed (code {rc}): {msg.decode() if msg else ''}")
    return rc


# ---- measurement support (bench.py's live roofline figures) -------------------------------------------
# When TIMING is a dict, every entry-point call made through timed_call() carries a pair of HIP events ATTACHED
# TO ITS DISPATCHES (mmt_arm_kernel_timing -> hipExtLaunchKernel start / stop events: the kernels' own duration on
# the device, what rocprofv3 --kernel-trace reports; events recorded around a launch would add the dispatch
# latency), and the (start, end) pairs are appended under `kind`.  `start.elapsed_time(end)` gives milliseconds
# after a stream synchronisation.  None (the default) = plain calls.
TIMING = None


class KernelEvent:
    """A HIP event owned by libmmt_hip (same `elapsed_time` call as torch.cuda.Event)."""

    def __init__(self):
        h = ctypes.c_void_p()
        call("mmt_timing_event_create", ctypes.byref(h))
        self.handle = h.value

    def elapsed_time(self, end):
        ms = ctypes.c_float()
        call("mmt_timing_elapsed_ms", self.handle, end.handle, ctypes.byref(ms))
        return float(ms.value)

    def __del__(self):
        try:
            if self.handle:
                lib().mmt_timing_event_destroy(self.handle)
        except Exception:
            pass


def timed_call(kind, name, *args):
    """call(name, *args); with TIMING enabled the call's kernels are bracketed by dispatch-attached events."""
    if TIMING is None:
        return call(name, *args)
    start, end = KernelEvent(), KernelEvent()
    call("mmt_arm_kernel_timing", start.handle, end.handle)
    try:
        rc = call(name, *args)
    finally:
        call("mmt_arm_kernel_timing", None, None)      # never leave it armed (a timed entry point consumes it anyway)
    TIMING.setdefault(kind, []).append((start, end))
    return rc


def mean_ms(pairs):
    """Average duration in ms of a list of (start, end) event pairs (after a synchronisation)."""
    return sum(s.elapsed_time(e) for s, e in pairs) / len(pairs)


def float3(values):
    """Host float[3] argument."""
    return (ctypes.c_float * 3)(*[float(v) for v in values])


def int3(values):
    return (ctypes.c_int32 * 3)(*[int(v) for v in values])


# ---- small host-side integer arrays (per-sample row offsets) on the device ---------------------------------------------
# The offsets depend on the point / box counts of the batch only.  `torch.tensor(list).to(device)` is a pageable host-to-device
# copy: the host waits for it (and it shows up as a Memcpy HtoD in the middle of the launch queue).  Identical count patterns
# reuse one read-only device tensor; a new pattern pays one blocking copy (it is then shared by calls on any stream).
_INT_ARRAYS = {}


def device_ints(values, device):
    """int32 CUDA tensor holding `values` (a list of Python ints); cached per (device, values) -- treat it as read-only."""
    import torch
    key = (str(device), tuple(int(v) for v in values))
    t = _INT_ARRAYS.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            return torch.tensor(key[1], dtype=torch.int32).to(device, non_blocking=True)     # (a node of the graph, as before)
        if len(_INT_ARRAYS) >= 256:
            _INT_ARRAYS.pop(next(iter(_INT_ARRAYS)))
        # a blocking copy, once per pattern: the tensor is shared by later calls on ANY stream, so it must be complete before it is
        # handed out (an asynchronous copy would be ordered on the creating stream only)
        t = _INT_ARRAYS[key] = torch.tensor(key[1], dtype=torch.int32, device=device)
    return t
