"""LiDAR / radar branch behind the three calls models/bev_depth.py:181-183 makes on
``self.lidar_encoder`` (an mmdet3d MVXFasterRCNN in the reference, built from
``lidar_conf`` exps/conf_aim.py:192-213):

    voxels, num_points, coors = lidar_encoder.voxelize(list_of_point_clouds)
    voxel_feats = lidar_encoder.pts_voxel_encoder(voxels, num_points, coors)
    lidar_bev   = lidar_encoder.pts_middle_encoder(voxel_feats, coors, batch_size)

Same names, argument meaning and return layouts (voxels [M,T,F] fp32 zero padded,
num_points [M] int32, coors [M,4] int32 = (b,z,y,x)); the arithmetic runs in the HIP
kernels of libmmt_hip.so (lidar_voxelize.hip).  The middle encoder is the
pillar-scatter (mmdet3d PointPillarsScatter) BASELINE.json names, not the reference
config's SparseEncoder (3-D sparse convolution: out of scope).
"""
import torch
from torch import nn
from torch.autograd import Function

from .. import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def grid_size(point_cloud_range, voxel_size):
    """mmcv Voxelization.__init__: round((max - min) / voxel_size) as fp32 tensors."""
    r = torch.tensor(point_cloud_range, dtype=torch.float32)
    v = torch.tensor(voxel_size, dtype=torch.float32)
    return [int(x) for x in torch.round((r[3:] - r[:3]) / v).long()]


# Per-cloud cell directories of the voxelizer (mmt_hard_voxelize_mean leaves one in its table; the table-form pillar
# scatter reads it).  Allocated once per (device, stream, size) and reused: nothing is cleared per step (a table that no
# voxelization has written reads as "no directory": the scatter answers with NaN).
_TABLES = {}


def _table_elems(B, grid_c, N):
    """Table size for clouds of up to N points, N rounded up to 64 k so that a stream of ragged clouds keeps one table."""
    return _lib.lib().mmt_voxelize_table_elems(B, grid_c, (int(N) + 65535) & ~65535)


def _voxel_table(dev, elems, tables=None):
    """`tables`: the dict that owns the persistent tables -- a LidarEncoder passes its own, so that the table its scatter
    reads (pillar_scatter_from_table) can only hold that encoder's last voxelization; the module-level dict serves
    the function-style callers, which never read a table back later."""
    if torch.cuda.is_current_stream_capturing():
        # inside a hipGraph capture the table belongs to the graph: allocated (and zero-filled) by the graph itself
        return torch.zeros((int(elems),), dtype=torch.int32, device=dev)
    tables = _TABLES if tables is None else tables
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, int(elems))
    t = tables.get(key)
    if t is None:
        if len(tables) >= 16:
            tables.pop(next(iter(tables)))
        t = torch.zeros((int(elems),), dtype=torch.int32, device=dev)
        tables[key] = t
    return t


def _batch_points(points_list):
    if len(points_list) == 0:
        raise ValueError("empty batch")
    dev = points_list[0].device
    for p in points_list:
        if not p.is_cuda:
            raise RuntimeError("points must be a CUDAtensor ")
        if p.dtype != torch.float32 or p.dim() != 2:
            raise RuntimeError("each point cloud must be a float32 [N, F] tensor")
    B = len(points_list)
    sizes = [int(p.shape[0]) for p in points_list]
    points = torch.cat([p.contiguous() for p in points_list], 0) if B > 1 else points_list[0].contiguous()
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + n)
    offsets = _lib.device_ints(offs, dev)
    return points, offsets, offs[-1]


def hard_voxelize_mean_batch(points_list, voxel_size, point_cloud_range, max_num_points, max_voxels,
                             num_features, materialize_voxels=True, return_table=False, tables=None):
    """Batched hard voxelization fused with the HardSimpleVFE mean (mmt_hard_voxelize_mean: three kernels,
    no clearing pass, no host sync).  Fixed-capacity layout: returns (voxels | None, num_points, coors,
    voxel_count, mean [B*max_voxels, num_features]); unused rows have coors = -1, num_points = 0, mean = 0."""
    points, offsets, N = _batch_points(points_list)
    dev = points.device
    B, F = len(points_list), points.shape[1]
    grid = grid_size(point_cloud_range, voxel_size)
    grid_c = _lib.int3(grid)
    T, V = int(max_num_points), int(max_voxels)
    voxels = torch.empty((B * V, T, F), dtype=torch.float32, device=dev) if materialize_voxels else None
    coors = torch.empty((B * V, 4), dtype=torch.int32, device=dev)
    num_points = torch.empty((B * V,), dtype=torch.int32, device=dev)
    voxel_count = torch.empty((B,), dtype=torch.int32, device=dev)
    mean = torch.empty((B * V, int(num_features)), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        table = _voxel_table(dev, _table_elems(B, grid_c, N), tables)
        scratch = torch.empty((_lib.lib().mmt_voxelize_scratch_elems(B, grid_c, N, T),), dtype=torch.int32, device=dev)
        _lib.timed_call("voxelize", "mmt_hard_voxelize_mean", B, N, F, points.data_ptr(), offsets.data_ptr(),
                        _lib.float3(voxel_size), _lib.float3(point_cloud_range[:3]), grid_c, T, V, int(num_features),
                        voxels.data_ptr() if materialize_voxels else 0, coors.data_ptr(), num_points.data_ptr(),
                        voxel_count.data_ptr(), mean.data_ptr(), table.data_ptr(), scratch.data_ptr(), _stream())
    if return_table:     # for pillar_scatter_from_table: the scatter walks the cell directory this call left in the table
        return voxels, num_points, coors, voxel_count, mean, table
    return voxels, num_points, coors, voxel_count, mean


def hard_voxelize_batch(points_list, voxel_size, point_cloud_range, max_num_points, max_voxels,
                        compact=True):
    """Batched hard voxelization (one kernel sequence for the whole batch).

    compact=True returns the reference layout (concatenated, M = sum of per-sample voxel
    counts; costs ONE device->host copy of B ints).  compact=False returns the
    fixed-capacity layout [B*max_voxels, ...] plus the per-sample counts, no host sync:
    unused rows have coors = -1 and num_points = 0."""
    points, offsets, N = _batch_points(points_list)
    dev = points.device
    B, F = len(points_list), points.shape[1]
    grid = grid_size(point_cloud_range, voxel_size)
    grid_c = _lib.int3(grid)
    T = int(max_num_points)
    V = int(max_voxels)
    voxels = torch.empty((B * V, T, F), dtype=torch.float32, device=dev)
    coors = torch.empty((B * V, 4), dtype=torch.int32, device=dev)
    num_points = torch.empty((B * V,), dtype=torch.int32, device=dev)
    voxel_count = torch.empty((B,), dtype=torch.int32, device=dev)
    ws_elems = _lib.lib().mmt_voxelize_workspace_elems(B, N, grid_c, T)
    workspace = torch.empty((ws_elems,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.timed_call("voxelize", "mmt_hard_voxelize", B, N, F, points.data_ptr(), offsets.data_ptr(),
                        _lib.float3(voxel_size), _lib.float3(point_cloud_range[:3]), grid_c, T, V,
                        voxels.data_ptr(), coors.data_ptr(), num_points.data_ptr(), voxel_count.data_ptr(),
                        workspace.data_ptr(), _stream())
    if not compact:
        return voxels, num_points, coors, voxel_count
    counts = voxel_count.cpu()                      # the one host sync of the drop-in path
    dst = torch.zeros(B + 1, dtype=torch.int32)
    dst[1:] = torch.cumsum(counts, 0)
    M = int(dst[-1])
    voxels_out = torch.empty((M, T, F), dtype=torch.float32, device=dev)
    coors_out = torch.empty((M, 4), dtype=torch.int32, device=dev)
    num_out = torch.empty((M,), dtype=torch.int32, device=dev)
    if M > 0:
        dst_dev = dst.to(dev)
        with torch.cuda.device(dev):
            _lib.call("mmt_compact_voxels", B, V, T * F, voxel_count.data_ptr(), dst_dev.data_ptr(),
                      voxels.data_ptr(), coors.data_ptr(), num_points.data_ptr(),
                      voxels_out.data_ptr(), coors_out.data_ptr(), num_out.data_ptr(), _stream())
    return voxels_out, num_out, coors_out


def simple_vfe(voxels, num_points, num_features):
    """HardSimpleVFE: voxels[:, :, :num_features].sum(1) / num_points -> [M, num_features]."""
    if not voxels.is_cuda:
        raise RuntimeError("voxels must be a CUDAtensor ")
    if voxels.dtype != torch.float32 or num_points.dtype != torch.int32:
        raise RuntimeError("simple_vfe: voxels must be float32 and num_points int32")
    M, T, F = voxels.shape
    out = torch.empty((M, num_features), dtype=torch.float32, device=voxels.device)
    if M:
        with torch.cuda.device(voxels.device):
            _lib.timed_call("vfe", "mmt_simple_vfe", M, T, F, int(num_features), voxels.contiguous().data_ptr(),
                      num_points.contiguous().data_ptr(), out.data_ptr(), _stream())
    return out


def _fp32_rows(feats, name):
    """The scatter kernels read fp32 rows.  Under autocast a learned pillar MLP hands over bf16 features (nn.Linear is on
    autocast's cast list): torch.amp.custom_fwd casts them back, and this check turns any other way of getting a
    non-fp32 tensor here into an exception instead of an out-of-bounds read on the device."""
    if feats.dtype != torch.float32:
        raise RuntimeError(f"expected scalar type Float but found {feats.dtype} for {name}")
    return feats.contiguous()


class _PillarScatter(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, feats, coors, batch_size, ny, nx, channels_last):
        if not feats.is_cuda:
            raise RuntimeError("voxel_features must be a CUDAtensor ")
        feats = _fp32_rows(feats, "voxel_features")
        coors = coors.contiguous()
        if coors.dtype != torch.int32:
            coors = coors.int()
        M, C = feats.shape
        nhwc = bool(channels_last) and C % 4 == 0
        if nhwc:   # same values, channels_last memory: the canvas is returned as a [B, C, ny, nx] view
            canvas = torch.empty((batch_size, ny, nx, C), dtype=torch.float32, device=feats.device)
        else:
            canvas = torch.empty((batch_size, C, ny, nx), dtype=torch.float32, device=feats.device)
        cell_map = torch.empty((batch_size * ny * nx,), dtype=torch.int32, device=feats.device)
        with torch.cuda.device(feats.device):
            _lib.timed_call("scatter", "mmt_pillar_scatter_nhwc" if nhwc else "mmt_pillar_scatter", M, C, batch_size, ny, nx,
                      feats.data_ptr() if M else 0, coors.data_ptr() if M else 0, canvas.data_ptr(),
                      cell_map.data_ptr(), _stream())
        ctx.save_for_backward(coors, cell_map)
        ctx.dims = (M, C, batch_size, ny, nx, nhwc)
        return canvas.permute(0, 3, 1, 2) if nhwc else canvas

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_canvas):
        coors, cell_map = ctx.saved_tensors
        M, C, B, ny, nx, nhwc = ctx.dims
        grad_canvas = grad_canvas.float()
        grad_feats = torch.empty((M, C), dtype=torch.float32, device=grad_canvas.device)
        if M:
            if nhwc:
                if not grad_canvas.is_contiguous(memory_format=torch.channels_last):
                    grad_canvas = grad_canvas.contiguous(memory_format=torch.channels_last)
            else:
                grad_canvas = grad_canvas.contiguous()
            with torch.cuda.device(grad_canvas.device):
                _lib.timed_call("scatter_backward", "mmt_pillar_scatter_nhwc_backward" if nhwc else "mmt_pillar_scatter_backward", M, C, B, ny, nx,
                          grad_canvas.data_ptr(), coors.data_ptr(), cell_map.data_ptr(), grad_feats.data_ptr(), _stream())
        return grad_feats, None, None, None, None, None


class _PillarScatterTable(Function):
    """Channels-last pillar scatter of the fixed-capacity rows of hard_voxelize_mean_batch straight from the voxelizer's
    table (mmt_pillar_scatter_nhwc_table): the rows own distinct cells, no cell -> row map is built."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, feats, coors, table, batch_size, ny, nx, max_voxels):
        feats = _fp32_rows(feats, "voxel_features")
        M, C = feats.shape
        canvas = torch.empty((batch_size, ny, nx, C), dtype=torch.float32, device=feats.device)
        with torch.cuda.device(feats.device):
            _lib.timed_call("scatter", "mmt_pillar_scatter_nhwc_table", C, batch_size, ny, nx, max_voxels, feats.data_ptr(),
                            table.data_ptr(), canvas.data_ptr(), _stream())
        ctx.save_for_backward(coors)
        ctx.dims = (M, C, batch_size, ny, nx)
        return canvas.permute(0, 3, 1, 2)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_canvas):
        (coors,) = ctx.saved_tensors
        M, C, B, ny, nx = ctx.dims
        grad_canvas = grad_canvas.float()
        grad_feats = torch.empty((M, C), dtype=torch.float32, device=grad_canvas.device)
        if not grad_canvas.is_contiguous(memory_format=torch.channels_last):
            grad_canvas = grad_canvas.contiguous(memory_format=torch.channels_last)
        with torch.cuda.device(grad_canvas.device):
            _lib.timed_call("scatter_backward", "mmt_pillar_scatter_nhwc_unique_backward", M, C, B, ny, nx, grad_canvas.data_ptr(),
                            coors.data_ptr(), grad_feats.data_ptr(), _stream())
        return grad_feats, None, None, None, None, None, None


class _PillarScatterStrided(Function):
    """Pillar scatter of the canvas cells (i * sy, j * sx) only -- what a nearest resize by the integer ratio (sy, sx) reads
    (models/bev_depth.py:188-190) -- as a channels_last [B, C, ny / sy, nx / sx] tensor.  table given: the fixed-capacity
    rows of the last hard_voxelize_mean_batch call on it (mmt_pillar_scatter_nhwc_table_strided); None: any rows, last-writer
    rule through a cell -> row map at the output resolution (mmt_pillar_scatter_nhwc_strided)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, feats, coors, table, batch_size, ny, nx, max_voxels, sy, sx):
        if not feats.is_cuda:
            raise RuntimeError("voxel_features must be a CUDAtensor ")
        feats = _fp32_rows(feats, "voxel_features")
        M, C = feats.shape
        oh, ow = ny // sy, nx // sx
        out = torch.empty((batch_size, oh, ow, C), dtype=torch.float32, device=feats.device)
        cell_map = None
        with torch.cuda.device(feats.device):
            if table is not None:
                _lib.timed_call("scatter", "mmt_pillar_scatter_nhwc_table_strided", C, batch_size, ny, nx, max_voxels, sy, sx,
                                feats.data_ptr(), table.data_ptr(), out.data_ptr(), C, _stream())
            else:
                cell_map = torch.empty((batch_size * oh * ow,), dtype=torch.int32, device=feats.device)
                _lib.timed_call("scatter", "mmt_pillar_scatter_nhwc_strided", M, C, batch_size, ny, nx, sy, sx,
                                feats.data_ptr() if M else 0, coors.data_ptr() if M else 0, out.data_ptr(), C,
                                cell_map.data_ptr(), _stream())
        ctx.save_for_backward(coors, *([cell_map] if cell_map is not None else []))
        ctx.dims = (M, C, batch_size, ny, nx, sy, sx)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        coors, *rest = ctx.saved_tensors
        M, C, B, ny, nx, sy, sx = ctx.dims
        grad_out = grad_out.float()
        if not grad_out.is_contiguous(memory_format=torch.channels_last):
            grad_out = grad_out.contiguous(memory_format=torch.channels_last)
        grad_feats = torch.empty((M, C), dtype=torch.float32, device=grad_out.device)
        with torch.cuda.device(grad_out.device):
            _lib.timed_call("scatter_backward", "mmt_pillar_scatter_nhwc_strided_backward", M, C, B, ny, nx, sy, sx,
                            grad_out.data_ptr(), C, coors.data_ptr(), rest[0].data_ptr() if rest else 0, grad_feats.data_ptr(), _stream())
        return grad_feats, None, None, None, None, None, None, None, None


def pillar_scatter_strided(voxel_features, coors, batch_size, ny, nx, stride_y, stride_x, table=None, max_voxels=0):
    """The canvas of pillar_scatter sampled at [..., ::stride_y, ::stride_x] without the canvas: channels_last
    [B, C, ny / stride_y, nx / stride_x].  `table`: see pillar_scatter_from_table."""
    sy, sx = int(stride_y), int(stride_x)
    if sy <= 0 or sx <= 0 or ny % sy or nx % sx:
        raise ValueError(f"pillar_scatter_strided: strides ({sy}, {sx}) must divide the grid ({ny}, {nx})")
    coors = coors.contiguous()
    if coors.dtype != torch.int32:
        coors = coors.int()
    return _PillarScatterStrided.apply(voxel_features, coors, table, int(batch_size), int(ny), int(nx), int(max_voxels), sy, sx)


def pillar_scatter_from_table(voxel_features, coors, table, batch_size, ny, nx, max_voxels):
    """[B*max_voxels, C] rows of the LAST hard_voxelize_mean_batch call on `table` -> channels_last [B, C, ny, nx] canvas."""
    return _PillarScatterTable.apply(voxel_features, coors.contiguous(), table, int(batch_size), int(ny), int(nx), int(max_voxels))


def pillar_scatter(voxel_features, coors, batch_size, ny, nx, channels_last=False):
    """PointPillarsScatter: dense [B, C, ny, nx] canvas from per-voxel features.  ``channels_last=True``
    returns the same tensor in torch.channels_last memory (what channels_last convolutions consume:
    no layout conversion in front of the BEV trunk, row-contiguous backward gather)."""
    return _PillarScatter.apply(voxel_features, coors, int(batch_size), int(ny), int(nx), bool(channels_last))


class _RowLinear(Function):
    """y = x @ W^T + b over MANY rows and few features (the pillar MLP: 100 000 voxel rows x 5 -> 64).  The weight gradient
    grad^T @ x is then a reduction over all rows into a 64 x 5 matrix -- ONE output tile for a GEMM library: 271 us per step at
    BASELINE configs[3] for 64 MFLOP.  Here the rows are cut into S batched pieces (one small GEMM each) and the pieces summed."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        N, O, I = g.shape[0], g.shape[1], x.shape[1]
        S = next((s for s in (128, 125, 100, 64, 50, 40, 32, 25, 20, 16, 10, 8, 5, 4, 2) if N % s == 0 and N // s >= 256), 1)
        if S > 1 and x.is_contiguous():
            gw = torch.bmm(g.view(S, N // S, O).transpose(1, 2), x.view(S, N // S, I)).sum(0)
        else:
            gw = g.t() @ x
        gx = g @ weight if ctx.needs_input_grad[0] else None
        return gx, gw, (g.sum(0) if ctx.needs_input_grad[2] else None)


class RowLinear(nn.Linear):
    """nn.Linear (same parameters and state_dict) for 2-D inputs of many rows: see _RowLinear."""

    def forward(self, x):
        if x.dim() == 2 and x.is_cuda and torch.is_grad_enabled():
            return _RowLinear.apply(x, self.weight, self.bias)
        return super().forward(x)


class LidarEncoder(nn.Module):
    """Object with the reference's three methods (models/bev_depth.py:181-183).

    pts_voxel_layer: dict(point_cloud_range, max_num_points, voxel_size, max_voxels)
    pts_voxel_encoder: dict(type='HardSimpleVFE', num_features)
    pts_middle_encoder: dict(type='PointPillarsScatter', in_channels, output_shape=[ny, nx])
    An optional learned per-pillar MLP (`pillar_channels`) lifts the VFE mean to
    `in_channels` before the scatter (PointPillars' PFN role)."""

    def __init__(self, pts_voxel_layer, pts_voxel_encoder=None, pts_middle_encoder=None,
                 pillar_channels=None, channels_last=True, **unused):
        super().__init__()
        # memory format of the scattered canvas (values and logical [B, C, ny, nx] shape are the same)
        self.channels_last = bool(channels_last)
        self.voxel_cfg = dict(pts_voxel_layer)
        mv = self.voxel_cfg.get("max_voxels", 25000)
        self.max_voxels = int(mv[0] if isinstance(mv, (tuple, list)) else mv)
        self.max_num_points = int(self.voxel_cfg["max_num_points"])
        self.voxel_size = list(self.voxel_cfg["voxel_size"])
        self.point_cloud_range = list(self.voxel_cfg["point_cloud_range"])
        self.grid = grid_size(self.point_cloud_range, self.voxel_size)
        enc = dict(pts_voxel_encoder or dict(type="HardSimpleVFE", num_features=5))
        self.num_features = int(enc.get("num_features", 5))
        mid = dict(pts_middle_encoder or {})
        self.output_shape = list(mid.get("output_shape", [self.grid[1], self.grid[0]]))
        self.in_channels = int(mid.get("in_channels", self.num_features))
        self.pillar_mlp = None
        if pillar_channels is not None or self.in_channels != self.num_features:
            out_c = int(pillar_channels or self.in_channels)
            # no normalisation layer: rows of the fixed-capacity layout that hold no voxel
            # must not influence live rows (forward_bev runs without compaction)
            self.pillar_mlp = nn.Sequential(RowLinear(self.num_features, out_c), nn.ReLU(inplace=True))
            self.in_channels = out_c
        self._tables = {}        # this encoder's persistent voxelizer tables (see _voxel_table)

    @torch.no_grad()
    def voxelize(self, points):
        """list[Tensor[Ni,F]] -> (voxels [M,T,F], num_points [M], coors [M,4]=(b,z,y,x))."""
        pts = [p.float() for p in points]            # mmdet3d forces fp32 here
        return hard_voxelize_batch(pts, self.voxel_size, self.point_cloud_range,
                                   self.max_num_points, self.max_voxels, compact=True)

    def pts_voxel_encoder(self, voxels, num_points, coors=None):
        return simple_vfe(voxels, num_points, self.num_features)

    def pts_middle_encoder(self, voxel_features, coors, batch_size):
        if self.pillar_mlp is not None:
            voxel_features = self.pillar_mlp(voxel_features)
        return pillar_scatter(voxel_features, coors, batch_size, self.output_shape[0], self.output_shape[1],
                              channels_last=self.channels_last)

    def forward_rows(self, points):
        """voxelize + mean (ONE fused call, the padded voxel tensor is not materialised) -> (MLP): the rows the scatter
        consumes, in the voxelizer's fixed-capacity layout, with NO host synchronisation.  Returns (feats [B*max_voxels, C],
        coors [B*max_voxels, 4], table or None): `table` is the voxelizer's own table when a canvas cell IS a voxel cell
        (one z layer, same y/x grid), which lets the scatter read voxel ids straight from it; empty rows carry coors = -1."""
        direct = (self.channels_last and self.grid[2] == 1 and self.output_shape == [self.grid[1], self.grid[0]]
                  and self.in_channels % 4 == 0 and self.max_voxels <= (1 << 23))
        with torch.no_grad():
            pts = [p.float() for p in points]
            _, _, coors, _, feats, table = hard_voxelize_mean_batch(
                pts, self.voxel_size, self.point_cloud_range, self.max_num_points, self.max_voxels,
                self.num_features, materialize_voxels=False, return_table=True, tables=self._tables)
        if self.pillar_mlp is not None:
            feats = self.pillar_mlp(feats)
        if not (direct and sum(int(p.shape[0]) for p in points) < (1 << 23)):
            table = None
        return feats, coors, table

    def forward_bev(self, points):
        """forward_rows -> scatter onto the full-resolution canvas (models/bev_depth.py:181-183 in three launches + one)."""
        feats, coors, table = self.forward_rows(points)
        if table is not None:
            return pillar_scatter_from_table(feats, coors, table, len(points), self.output_shape[0], self.output_shape[1], self.max_voxels)
        return pillar_scatter(feats, coors, len(points), self.output_shape[0], self.output_shape[1],
                              channels_last=self.channels_last)

    def forward_bev_strided(self, points, stride_y, stride_x):
        """forward_rows -> the canvas cells a nearest resize by (stride_y, stride_x) reads, nothing else
        (models/bev_depth.py:181-183 + :188-190): channels_last [B, C, ny / stride_y, nx / stride_x]."""
        feats, coors, table = self.forward_rows(points)
        return pillar_scatter_strided(feats, coors, len(points), self.output_shape[0], self.output_shape[1], stride_y, stride_x,
                                      table=table, max_voxels=self.max_voxels)
