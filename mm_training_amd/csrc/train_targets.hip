// Per-step label generation that the reference does in Python loops.
//
// mmt_depth_labels: LiDAR depth supervision of the camera branch,
//   exps/mm_training_aim.py:114-163 (get_depth_labels / get_depth_image: B x N_cam projections of
//   the whole point cloud, a boolean-mask select and an indexed write per camera) and :180-215
//   (get_downsampled_gt_depth: min over each downsample x downsample block, depth-bin index,
//   one-hot).  Here: one thread per (point, camera) projects and folds the point straight into
//   its feature-map cell with an integer atomicMin on the float bits (depths are > 1, so the
//   unsigned order is the float order); a second kernel turns the per-cell minimum into the bin
//   index and the one-hot row.  No intermediate H x W depth image, no host synchronisation.
//   Two points on the same PIXEL: the reference keeps whichever was written last (an arbitrary
//   one on a GPU); the block minimum is taken over all points here -- deterministic.
#include "mmt_common.h"

namespace {

constexpr int kBlock = 256;
constexpr float kNoDepth = 1e5f;   // exps/mm_training_aim.py:200-202

struct DepthArgs {
    int B, N, F;
    int H, W, ds, fH, fW, D;
    float d_lo, d_step;
    const float *points;        // [sum Ni, F]
    const int32_t *offsets;     // [B+1] row offsets into points
    const float *extr;          // [B, N, 4, 4] ego -> camera
    const float *intr;          // [B, N, 4, 4]
    const float *bda_inv;       // [B, 3, 3] inverse of the BEV-augmentation rotation
    uint32_t *cell_min;         // [B*N, fH*fW] float bits
    int32_t *bin;               // [B*N*fH*fW] or NULL
    float *onehot;              // [B*N*fH*fW, D] or NULL
};

__global__ __launch_bounds__(kBlock) void depth_fill_kernel(uint32_t *cell_min, int64_t n) {
    const uint32_t v = __float_as_uint(kNoDepth);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) cell_min[i] = v;
}

// grid.y = sample, grid.z = camera; x runs over the sample's points
__global__ __launch_bounds__(kBlock) void depth_project_kernel(DepthArgs a) {
    const int b = blockIdx.y, n = blockIdx.z;
    __shared__ float E[16], K[16], R[9];
    if (threadIdx.x < 16) {
        E[threadIdx.x] = a.extr[((int64_t)b * a.N + n) * 16 + threadIdx.x];
        K[threadIdx.x] = a.intr[((int64_t)b * a.N + n) * 16 + threadIdx.x];
    }
    if (threadIdx.x < 9) R[threadIdx.x] = a.bda_inv[b * 9 + threadIdx.x];
    __syncthreads();
    const int beg = a.offsets[b], end = a.offsets[b + 1];
    uint32_t *cells = a.cell_min + ((int64_t)b * a.N + n) * a.fH * a.fW;
    const float wmax = (float)(a.W - 1), hmax = (float)(a.H - 1);
    for (int i = beg + blockIdx.x * kBlock + threadIdx.x; i < end; i += gridDim.x * kBlock) {
        const float *p = a.points + (int64_t)i * a.F;
        const float x = p[0], y = p[1], z = p[2];
        // undo the BEV augmentation (exps/mm_training_aim.py:129-131): q = inv(R) p
        float q[3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            q[r] = __fadd_rn(__fadd_rn(__fmul_rn(R[r * 3], x), __fmul_rn(R[r * 3 + 1], y)), __fmul_rn(R[r * 3 + 2], z));
        // ego -> camera, then the pinhole projection (:143-149)
        float c[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            c[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(E[r * 4], q[0]), __fmul_rn(E[r * 4 + 1], q[1])),
                                       __fmul_rn(E[r * 4 + 2], q[2])), E[r * 4 + 3]);
        float pr[3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            pr[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(K[r * 4], c[0]), __fmul_rn(K[r * 4 + 1], c[1])),
                                        __fmul_rn(K[r * 4 + 2], c[2])), __fmul_rn(K[r * 4 + 3], c[3]));
        const float depth = c[2];
        const float u = __fdiv_rn(pr[0], pr[2]), v = __fdiv_rn(pr[1], pr[2]);
        // :150-155 (comparisons are false for NaN, like torch's)
        if (depth > 1.0f && u > 1.0f && u < wmax && v > 1.0f && v < hmax) {
            const int iu = (int)u, iv = (int)v;                       // .to(torch.long): truncation
            atomicMin(&cells[(iv / a.ds) * a.fW + iu / a.ds], __float_as_uint(depth));
        }
    }
}

__global__ __launch_bounds__(kBlock) void depth_bins_kernel(DepthArgs a, int64_t ncells) {
    const int D = a.D;
    if (a.onehot) {
        const int64_t total = ncells * D;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
            const int64_t cell = i / D;
            const int d = (int)(i - cell * D);
            const float depth = __uint_as_float(a.cell_min[cell]);
            // :207-212: (d - (lo - step)) / step, kept if in [0, D), else 0; .long() truncates
            const float g = __fdiv_rn(__fsub_rn(depth, __fsub_rn(a.d_lo, a.d_step)), a.d_step);
            const int bin = (g < (float)D && g >= 0.0f) ? (int)g : 0;
            a.onehot[i] = d == bin ? 1.0f : 0.0f;
            if (d == 0 && a.bin) a.bin[cell] = bin;
        }
    } else {
        for (int64_t cell = (int64_t)blockIdx.x * kBlock + threadIdx.x; cell < ncells; cell += (int64_t)gridDim.x * kBlock) {
            const float depth = __uint_as_float(a.cell_min[cell]);
            const float g = __fdiv_rn(__fsub_rn(depth, __fsub_rn(a.d_lo, a.d_step)), a.d_step);
            a.bin[cell] = (g < (float)D && g >= 0.0f) ? (int)g : 0;
        }
    }
}

}  // namespace

extern "C" int64_t mmt_depth_labels_workspace_elems(int B, int num_cams, int H, int W, int downsample) {
    if (B <= 0 || num_cams <= 0 || H <= 0 || W <= 0 || downsample <= 0) return -1;
    return (int64_t)B * num_cams * (H / downsample) * (W / downsample);
}

extern "C" int mmt_depth_labels(int B, int num_cams, int F, int max_points, int H, int W, int downsample,
                                float d_lo, float d_step, int D, const float *points,
                                const int32_t *point_offsets, const float *extrinsics, const float *intrinsics,
                                const float *bda_inv, int32_t *workspace, int64_t workspace_elems,
                                int32_t *depth_bin, float *onehot, void *stream) {
    MMT_REQUIRE_PTR(points);
    MMT_REQUIRE_PTR(point_offsets);
    MMT_REQUIRE_PTR(extrinsics);
    MMT_REQUIRE_PTR(intrinsics);
    MMT_REQUIRE_PTR(bda_inv);
    MMT_REQUIRE_PTR(workspace);
    if (!depth_bin && !onehot) return mmt::fail(MMT_ERR_NULL_POINTER, "depth_labels: depth_bin and onehot are both NULL");
    if (B <= 0 || num_cams <= 0 || F < 3 || max_points < 0 || H <= 0 || W <= 0 || downsample <= 0 || D <= 0 ||
        H % downsample || W % downsample || !(d_step > 0.f))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "depth_labels: bad shape (B=%d cams=%d F=%d %dx%d / %d, D=%d)", B, num_cams, F,
                         H, W, downsample, D);
    if (B > 65535 || num_cams > 65535) return mmt::fail(MMT_ERR_TOO_LARGE, "depth_labels: B or num_cams > 65535");
    const int fH = H / downsample, fW = W / downsample;
    const int64_t ncells = (int64_t)B * num_cams * fH * fW;
    if (ncells * D >= (1ll << 40)) return mmt::fail(MMT_ERR_TOO_LARGE, "depth_labels: too many label elements");
    if (workspace_elems < ncells)
        return mmt::fail(MMT_ERR_WORKSPACE, "depth_labels: workspace holds %lld elements, needs %lld",
                         (long long)workspace_elems, (long long)ncells);
    hipStream_t st = (hipStream_t)stream;
    DepthArgs a;
    a.B = B; a.N = num_cams; a.F = F; a.H = H; a.W = W; a.ds = downsample; a.fH = fH; a.fW = fW; a.D = D;
    a.d_lo = d_lo; a.d_step = d_step;
    a.points = points; a.offsets = point_offsets; a.extr = extrinsics; a.intr = intrinsics; a.bda_inv = bda_inv;
    a.cell_min = reinterpret_cast<uint32_t *>(workspace); a.bin = depth_bin; a.onehot = onehot;
    hipLaunchKernelGGL(depth_fill_kernel, dim3(mmt::stream_grid(ncells, kBlock)), dim3(kBlock), 0, st, a.cell_min, ncells);
    if (int rc = mmt::check_launch("depth_labels(fill)")) return rc;
    if (max_points > 0) {
        int gx = (int)mmt::ceil_div(max_points, kBlock);
        if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(depth_project_kernel, dim3(gx, B, num_cams), dim3(kBlock), 0, st, a);
        if (int rc = mmt::check_launch("depth_labels(project)")) return rc;
    }
    const int64_t work = onehot ? ncells * D : ncells;
    hipLaunchKernelGGL(depth_bins_kernel, dim3(mmt::stream_grid(work, kBlock)), dim3(kBlock), 0, st, a, ncells);
    return mmt::check_launch("depth_labels(bins)");
}
