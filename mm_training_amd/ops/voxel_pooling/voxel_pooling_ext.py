"""Extension module of the op, same entry point as the reference's pybind module
``ops.voxel_pooling.voxel_pooling_ext`` (ops/voxel_pooling/src/voxel_pooling_forward.cpp:24-41,
built by setup.py:60-67) -- here a thin Python binding of the C ABI in
``libmmt_hip.so`` (include/mmt_hip.h) instead of an ATen/pybind11 shim.

``voxel_pooling_forward_wrapper`` keeps the reference's 10-argument signature,
argument meaning, ownership (caller allocates; ``output_features`` is accumulated
into, ``pos_memo`` rows of kept points are overwritten) and checked error cases
(RuntimeError for non-CUDA / non-contiguous geom or features, and for a dtype other
than int32 / float32).  It launches on the current stream and does not synchronise.
A failed launch raises instead of calling exit(-1) (voxel_pooling_forward_cuda.cu:51-55).

``voxel_pooling_backward_wrapper`` is new: the reference's backward is pure ATen
(ops/voxel_pooling/voxel_pooling.py:58-69).
"""
import torch

from ... import _lib


def _check_input(t, name, dtype):
    # CHECK_INPUT (voxel_pooling_forward.cpp:10-16) + the data_ptr<T>() dtype check (:28-31)
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDAtensor ")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous ")
    if t.dtype != dtype:
        raise RuntimeError(f"expected scalar type {dtype} but found {t.dtype} for {name}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Per-launch timing for bench.py's roofline figures lives in mm_training_amd._lib (TIMING / timed_call):
# the forward's default SEG_GATHER launch and every backward launch carry dispatch-attached events; the
# non-default forward algorithms do not take them and are bracketed on the stream instead.
def _timed_call(kind, *args, dispatch_events=False):
    if _lib.TIMING is None:
        return _lib.call(*args)
    if dispatch_events:
        return _lib.timed_call(kind, *args)
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    start.record()
    _lib.call(*args)
    end.record()
    _lib.TIMING.setdefault(kind, []).append((start, end))


def voxel_pooling_forward_wrapper(batch_size, num_points, num_channels, num_voxel_x,
                                  num_voxel_y, num_voxel_z, geom_xyz_tensor,
                                  input_features_tensor, output_features_tensor,
                                  pos_memo_tensor, flags=_lib.VP_ALGO_AUTO):
    _check_input(geom_xyz_tensor, "geom_xyz_tensor", torch.int32)
    _check_input(input_features_tensor, "input_features_tensor", torch.float32)
    if output_features_tensor.dtype != torch.float32 or pos_memo_tensor.dtype != torch.int32:
        raise RuntimeError("output_features must be float32 and pos_memo int32")
    if not (output_features_tensor.is_cuda and pos_memo_tensor.is_cuda):
        raise RuntimeError("output_features and pos_memo must be CUDA tensors")
    B, P, C = int(batch_size), int(num_points), int(num_channels)
    nx, ny, nz = int(num_voxel_x), int(num_voxel_y), int(num_voxel_z)
    # the reference never validates these; a wrong size here would be an out-of-bounds
    # device access, so refuse it on the host
    if geom_xyz_tensor.numel() != B * P * 3 or input_features_tensor.numel() != B * P * C:
        raise RuntimeError("geom_xyz / input_features do not match (batch_size, num_points, num_channels)")
    if output_features_tensor.numel() != B * ny * nx * C or not output_features_tensor.is_contiguous():
        raise RuntimeError("output_features must be a contiguous [B, ny, nx, C] tensor")
    if pos_memo_tensor.numel() != B * P * 3 or not pos_memo_tensor.is_contiguous():
        raise RuntimeError("pos_memo must be a contiguous [B, P, 3] tensor")
    # the default SEG_GATHER launch is the one that carries dispatch events (see mmt_arm_kernel_timing)
    default_path = (int(flags) & 0xF) in (0, 3) and not (int(flags) & 0x60) and C % 4 == 0 and C <= 256 \
        and input_features_tensor.data_ptr() % 16 == 0
    with torch.cuda.device(input_features_tensor.device):
        _timed_call("forward", "mmt_voxel_pooling_forward_ex", B, P, C, nx, ny, nz,
                    geom_xyz_tensor.data_ptr(), input_features_tensor.data_ptr(),
                    output_features_tensor.data_ptr(), pos_memo_tensor.data_ptr(),
                    int(flags), _stream(), dispatch_events=default_path)
    return 1


def backward_workspace_elems(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y):
    return int(_lib.lib().mmt_voxel_pooling_backward_workspace_elems(
        int(batch_size), int(num_points), int(num_channels), int(num_voxel_x), int(num_voxel_y)))


def voxel_pooling_backward_wrapper(batch_size, num_points, num_channels, num_voxel_x,
                                   num_voxel_y, pos_memo_tensor, grad_output_tensor,
                                   grad_input_tensor, workspace_tensor=None):
    """grad_input[B,P,C] <- gather of grad_output (indexed [B,C,ny,nx], any strides).

    workspace_tensor (optional, float32): see mmt_voxel_pooling_backward in include/mmt_hip.h;
    `backward_workspace_elems` gives the size that enables every fast path."""
    _check_input(pos_memo_tensor, "pos_memo_tensor", torch.int32)
    _check_input(grad_input_tensor, "grad_input_tensor", torch.float32)
    if not grad_output_tensor.is_cuda or grad_output_tensor.dtype != torch.float32:
        raise RuntimeError("grad_output_tensor must be a float32 CUDAtensor ")
    B, P, C = int(batch_size), int(num_points), int(num_channels)
    nx, ny = int(num_voxel_x), int(num_voxel_y)
    if tuple(grad_output_tensor.shape) != (B, C, ny, nx):
        raise RuntimeError(f"grad_output must have shape {(B, C, ny, nx)}, got {tuple(grad_output_tensor.shape)}")
    if pos_memo_tensor.numel() != B * P * 3 or grad_input_tensor.numel() != B * P * C:
        raise RuntimeError("pos_memo / grad_input do not match (batch_size, num_points, num_channels)")
    sb, sc, sy, sx = grad_output_tensor.stride()
    ws, ws_elems = 0, 0
    if workspace_tensor is not None:
        if (not workspace_tensor.is_cuda or workspace_tensor.dtype != torch.float32
                or not workspace_tensor.is_contiguous()):
            raise RuntimeError("workspace must be a contiguous float32 CUDA tensor")
        ws, ws_elems = workspace_tensor.data_ptr(), workspace_tensor.numel()
    with torch.cuda.device(grad_input_tensor.device):
        _timed_call("backward", "mmt_voxel_pooling_backward", B, P, C, nx, ny, pos_memo_tensor.data_ptr(),
                  grad_output_tensor.data_ptr(), sb, sc, sy, sx, grad_input_tensor.data_ptr(),
                  ws, ws_elems, _stream(), dispatch_events=True)
    return 1


# ---- bf16 feature storage (include/mmt_hip.h "bf16 feature storage").  The reference's
# extension rejects anything but float32 (voxel_pooling_forward.cpp:28-31), and so do the two wrappers above; these
# are additional entry points with the same argument order: bf16 rows in, fp32 accumulate, fp32 BEV out.

def voxel_pooling_forward_wrapper_bf16(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y, num_voxel_z,
                                       geom_xyz_tensor, input_features_tensor, output_features_tensor, pos_memo_tensor,
                                       flags=_lib.VP_WRITE_DROPPED):
    _check_input(geom_xyz_tensor, "geom_xyz_tensor", torch.int32)
    _check_input(input_features_tensor, "input_features_tensor", torch.bfloat16)
    if output_features_tensor.dtype != torch.float32 or pos_memo_tensor.dtype != torch.int32:
        raise RuntimeError("output_features must be float32 and pos_memo int32")
    if not (output_features_tensor.is_cuda and pos_memo_tensor.is_cuda):
        raise RuntimeError("output_features and pos_memo must be CUDA tensors")
    B, P, C = int(batch_size), int(num_points), int(num_channels)
    nx, ny, nz = int(num_voxel_x), int(num_voxel_y), int(num_voxel_z)
    if geom_xyz_tensor.numel() != B * P * 3 or input_features_tensor.numel() != B * P * C:
        raise RuntimeError("geom_xyz / input_features do not match (batch_size, num_points, num_channels)")
    if output_features_tensor.numel() != B * ny * nx * C or not output_features_tensor.is_contiguous():
        raise RuntimeError("output_features must be a contiguous [B, ny, nx, C] tensor")
    if pos_memo_tensor.numel() != B * P * 3 or not pos_memo_tensor.is_contiguous():
        raise RuntimeError("pos_memo must be a contiguous [B, P, 3] tensor")
    with torch.cuda.device(input_features_tensor.device):
        _lib.timed_call("forward", "mmt_voxel_pooling_forward_bf16", B, P, C, nx, ny, nz, geom_xyz_tensor.data_ptr(),
                        input_features_tensor.data_ptr(), output_features_tensor.data_ptr(), pos_memo_tensor.data_ptr(),
                        int(flags), _stream())
    return 1


def voxel_pooling_backward_wrapper_bf16(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y, pos_memo_tensor,
                                        grad_output_tensor, grad_input_tensor, workspace_tensor=None):
    """grad_input bf16 [B,P,C] <- gather of the fp32 grad_output (indexed [B,C,ny,nx], any strides), rounded to nearest even."""
    _check_input(pos_memo_tensor, "pos_memo_tensor", torch.int32)
    _check_input(grad_input_tensor, "grad_input_tensor", torch.bfloat16)
    if not grad_output_tensor.is_cuda or grad_output_tensor.dtype != torch.float32:
        raise RuntimeError("grad_output_tensor must be a float32 CUDAtensor ")
    B, P, C = int(batch_size), int(num_points), int(num_channels)
    nx, ny = int(num_voxel_x), int(num_voxel_y)
    if tuple(grad_output_tensor.shape) != (B, C, ny, nx):
        raise RuntimeError(f"grad_output must have shape {(B, C, ny, nx)}, got {tuple(grad_output_tensor.shape)}")
    if pos_memo_tensor.numel() != B * P * 3 or grad_input_tensor.numel() != B * P * C:
        raise RuntimeError("pos_memo / grad_input do not match (batch_size, num_points, num_channels)")
    sb, sc, sy, sx = grad_output_tensor.stride()
    ws, ws_elems = 0, 0
    if workspace_tensor is not None:
        if not workspace_tensor.is_cuda or workspace_tensor.dtype != torch.float32 or not workspace_tensor.is_contiguous():
            raise RuntimeError("workspace must be a contiguous float32 CUDA tensor")
        ws, ws_elems = workspace_tensor.data_ptr(), workspace_tensor.numel()
    with torch.cuda.device(grad_input_tensor.device):
        _lib.timed_call("backward", "mmt_voxel_pooling_backward_bf16", B, P, C, nx, ny, pos_memo_tensor.data_ptr(),
                        grad_output_tensor.data_ptr(), sb, sc, sy, sx, grad_input_tensor.data_ptr(), ws, ws_elems, _stream())
    return 1
