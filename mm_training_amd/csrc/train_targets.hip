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
    const uint8_t *flipped;     // [B*N] or NULL: cameras whose label map is written mirrored along w (augment_images, :89-112)
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

// cell of the projected (unflipped) map that lands in OUTPUT cell `cell`: the same one, or its mirror image along w for a
// camera augment_images flipped (exps/mm_training_aim.py:105-110: hflip of the [D, fH, fW] label image of that camera)
__device__ __forceinline__ int64_t source_cell(const DepthArgs &a, int64_t cell) {
    if (a.flipped == nullptr) return cell;
    const int64_t row = cell / a.fW;                 // (camera, h)
    const int w = (int)(cell - row * a.fW);
    return a.flipped[row / a.fH] ? row * a.fW + (a.fW - 1 - w) : cell;
}

__global__ __launch_bounds__(kBlock) void depth_bins_kernel(DepthArgs a, int64_t ncells) {
    const int D = a.D;
    if (a.onehot) {
        const int64_t total = ncells * D;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
            const int64_t cell = i / D;
            const int d = (int)(i - cell * D);
            const float depth = __uint_as_float(a.cell_min[source_cell(a, cell)]);
            // :207-212: (d - (lo - step)) / step, kept if in [0, D), else 0; .long() truncates
            const float g = __fdiv_rn(__fsub_rn(depth, __fsub_rn(a.d_lo, a.d_step)), a.d_step);
            const int bin = (g < (float)D && g >= 0.0f) ? (int)g : 0;
            a.onehot[i] = d == bin ? 1.0f : 0.0f;
            if (d == 0 && a.bin) a.bin[cell] = bin;
        }
    } else {
        for (int64_t cell = (int64_t)blockIdx.x * kBlock + threadIdx.x; cell < ncells; cell += (int64_t)gridDim.x * kBlock) {
            const float depth = __uint_as_float(a.cell_min[source_cell(a, cell)]);
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
    return mmt_depth_labels_flipped(B, num_cams, F, max_points, H, W, downsample, d_lo, d_step, D, points, point_offsets, extrinsics,
                                    intrinsics, bda_inv, workspace, workspace_elems, depth_bin, onehot, nullptr, stream);
}

extern "C" int mmt_depth_labels_flipped(int B, int num_cams, int F, int max_points, int H, int W, int downsample,
                                        float d_lo, float d_step, int D, const float *points,
                                        const int32_t *point_offsets, const float *extrinsics, const float *intrinsics,
                                        const float *bda_inv, int32_t *workspace, int64_t workspace_elems,
                                        int32_t *depth_bin, float *onehot, const uint8_t *flipped, void *stream) {
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
    a.cell_min = reinterpret_cast<uint32_t *>(workspace); a.bin = depth_bin; a.onehot = onehot; a.flipped = flipped;
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

// ---------------------------------------------------------------------------
// Image augmentation of the training step, exps/mm_training_aim.py:89-112 (augment_images: per camera, with probability 1/2,
// kornia hflip of the image AND of its depth-label map; the flags go to mats['flipped']) and :510-512 (normalize_images:
// torchvision Normalize of sweep_imgs[:, :, :, :3] / 255).  The reference stacks per-image Python lists; here the flags are
// a device byte per camera and
//   mmt_hflip                  out[i, r, w, :] = in[i, r, flipped[i / group] ? W-1-w : w, :]    (any [n, rows, W, E] tensor)
//   mmt_normalize_flip_images  normalise + flip in ONE pass over the images, optionally straight into the channels-last
//                              memory the first convolution reads (the NCHW -> NHWC conversion in front of it disappears)
namespace {

__global__ __launch_bounds__(kBlock) void hflip_kernel(int64_t n, int group, int rows, int W, int E, const float *in,
                                                       const uint8_t *flipped, float *out) {
    const int64_t per_img = (int64_t)rows * W * E, total = n * per_img;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t img = i / per_img;
        int64_t src = i;
        if (flipped[img / group]) {
            const int64_t rem = i - img * per_img;
            const int64_t row = rem / ((int64_t)W * E);
            const int we = (int)(rem - row * W * E);
            const int w = we / E, e = we - w * E;
            src = img * per_img + row * W * E + (int64_t)(W - 1 - w) * E + e;
        }
        out[i] = in[src];
    }
}

struct NormArgs {
    int64_t n;
    int c_in, H, W;
    float scale, mean[3], stdv[3];
    const float *in;             // [n, c_in, H, W]; the first 3 channels are read
    const uint8_t *flipped;      // [n] or NULL
    float *out;                  // [n, 3, H, W] or, channels-last, [n, H, W, 3]
};

// one thread per output pixel: three plane reads (coalesced along w, reversed for a flipped camera), three outputs
template <bool NHWC>
__global__ __launch_bounds__(kBlock) void normalize_flip_kernel(NormArgs a) {
    const int64_t hw = (int64_t)a.H * a.W, total = a.n * hw;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t img = i / hw;
        const int64_t rem = i - img * hw;
        const int h = (int)(rem / a.W), w = (int)(rem - (int64_t)h * a.W);
        const int ws = (a.flipped != nullptr && a.flipped[img]) ? a.W - 1 - w : w;
        const float *src = a.in + (img * a.c_in) * hw + (int64_t)h * a.W + ws;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)      // (x * (1/255) - mean) / std: ATen's scalar division is a multiplication by the fp32 reciprocal
            v[c] = __fdiv_rn(__fsub_rn(__fmul_rn(src[c * hw], a.scale), a.mean[c]), a.stdv[c]);
        if (NHWC) {
            float *dst = a.out + i * 3;
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) a.out[(img * 3 + c) * hw + rem] = v[c];
        }
    }
}

// Planar (NCHW) output: four pixels of a row per thread (W % 4 == 0, 16-byte aligned planes) -- one 16-byte load per plane, reversed
// for a flipped camera, one 16-byte store per plane, and no integer division per pixel (the kernel above spends two 64-bit
// divisions on every pixel): 20.1 -> 16.6 us at [24, 3, 256, 704].  Same arithmetic, same bits.  The channels-last output keeps
// the kernel above: three variants with 16-byte accesses (four pixels per thread = 48 bytes at a 48-byte stride; a thread per
// 16-byte piece of the interleaved row reading the planes; the same through an LDS copy of the row) took 23 / 26 / 26 us against its 20.
__global__ __launch_bounds__(kBlock) void normalize_flip_rows_kernel(NormArgs a, int w4) {      // w4 = W / 4; grid.y walks the rows
    const int64_t hw = (int64_t)a.H * a.W, rows = a.n * a.H;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const int64_t img = row / a.H;
        const int h = (int)(row - img * a.H);
        const bool fl = a.flipped != nullptr && a.flipped[img];
        for (int q = blockIdx.x * kBlock + threadIdx.x; q < w4; q += gridDim.x * kBlock) {
            const int w = 4 * q, ws = fl ? a.W - 4 - w : w;
            const float *src = a.in + (img * a.c_in) * hw + (int64_t)h * a.W + ws;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float4 t = *reinterpret_cast<const float4 *>(src + c * hw);
                const float x[4] = {fl ? t.w : t.x, fl ? t.z : t.y, fl ? t.y : t.z, fl ? t.x : t.w};
                float4 v;
                v.x = __fdiv_rn(__fsub_rn(__fmul_rn(x[0], a.scale), a.mean[c]), a.stdv[c]);
                v.y = __fdiv_rn(__fsub_rn(__fmul_rn(x[1], a.scale), a.mean[c]), a.stdv[c]);
                v.z = __fdiv_rn(__fsub_rn(__fmul_rn(x[2], a.scale), a.mean[c]), a.stdv[c]);
                v.w = __fdiv_rn(__fsub_rn(__fmul_rn(x[3], a.scale), a.mean[c]), a.stdv[c]);
                *reinterpret_cast<float4 *>(a.out + (img * 3 + c) * hw + (int64_t)h * a.W + w) = v;
            }
        }
    }
}

}  // namespace

extern "C" int mmt_hflip(int64_t n, int group, int rows, int W, int elems, const float *in, const uint8_t *flipped, float *out,
                         void *stream) {
    if (n < 0 || group <= 0 || rows <= 0 || W <= 0 || elems <= 0) return mmt::fail(MMT_ERR_BAD_SHAPE, "hflip: bad sizes");
    if (n == 0) return MMT_OK;
    MMT_REQUIRE_PTR(in);
    MMT_REQUIRE_PTR(flipped);
    MMT_REQUIRE_PTR(out);
    if (in == out) return mmt::fail(MMT_ERR_BAD_SHAPE, "hflip: in-place is not supported");
    hipLaunchKernelGGL(hflip_kernel, dim3(mmt::stream_grid(n * rows * W * elems, kBlock, 256 * 32)), dim3(kBlock), 0, (hipStream_t)stream,
                       n, group, rows, W, elems, in, flipped, out);
    return mmt::check_launch("hflip");
}

extern "C" int mmt_normalize_flip_images(int64_t n_images, int channels_in, int H, int W, const float *images, float scale,
                                         const float *mean_host, const float *std_host, const uint8_t *flipped, float *out,
                                         int channels_last, void *stream) {
    if (n_images < 0 || channels_in < 3 || H <= 0 || W <= 0) return mmt::fail(MMT_ERR_BAD_SHAPE, "normalize_flip_images: bad sizes (>= 3 channels)");
    if (n_images == 0) return MMT_OK;
    MMT_REQUIRE_PTR(images);
    MMT_REQUIRE_PTR(mean_host);
    MMT_REQUIRE_PTR(std_host);
    MMT_REQUIRE_PTR(out);
    NormArgs a;
    a.n = n_images; a.c_in = channels_in; a.H = H; a.W = W; a.scale = scale;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean_host[c]; a.stdv[c] = std_host[c]; }
    a.in = images; a.flipped = flipped; a.out = out;
    const dim3 grid(mmt::stream_grid(n_images * H * W, kBlock, 256 * 64)), block(kBlock);
    if (channels_last) hipLaunchKernelGGL(normalize_flip_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
    else if (W % 4 == 0 && (((uintptr_t)images | (uintptr_t)out) & 15) == 0) {
        const int64_t rows = n_images * H;
        hipLaunchKernelGGL(normalize_flip_rows_kernel, dim3((unsigned)((W / 4 + kBlock - 1) / kBlock), (unsigned)(rows < 32768 ? rows : 32768)), block, 0,
                           (hipStream_t)stream, a, W / 4);
    } else hipLaunchKernelGGL(normalize_flip_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
    return mmt::check_launch("normalize_flip_images");
}

// ---------------------------------------------------------------------------
// mmt_centerpoint_targets: CenterPoint training targets, layers/heads/bev_depth_head.py:113-254
// (get_targets_single: a Python loop over tasks and boxes with per-box tensor construction and
// mmdet3d's gaussian_radius / draw_heatmap_gaussian).  One workgroup per (box, sample): the box's
// Gaussian window is max-combined into its class heat-map with an integer atomicMax on the float
// bits (values are >= 0), lane 0 writes the regression target row.  Slots as in the reference: a task's
// boxes densely, class after class, each class in input order; the first max_objs of them (round 5: pinned by
// tests/golden/centerpoint_targets.npz, produced by the reference's own lines -- rounds 1-4 kept box k at slot k and
// cut the SAMPLE at max_objs, which differs once a sample holds more boxes than that).
namespace {

constexpr int kMaxTasks = 8;

struct CpArgs {
    int B, T, max_objs, fx, fy, norm_bbox;
    float x0, y0, vx, vy, osf, overlap;
    int min_radius;
    int cls_begin[kMaxTasks], cls_count[kMaxTasks];
    const float *boxes;          // [sum K, 9] x,y,z,w,l,h,yaw,vx,vy
    const int32_t *labels;       // [sum K]
    const int32_t *offsets;      // [B+1]
    float *heatmap[kMaxTasks];   // [B, cls_count, fy, fx]
    float *anno[kMaxTasks];      // [B, max_objs, 10]
    int64_t *ind[kMaxTasks];     // [B, max_objs]
    uint8_t *mask[kMaxTasks];    // [B, max_objs]
};

__global__ __launch_bounds__(kBlock) void cp_clear_kernel(CpArgs a) {
    const int t = blockIdx.y;
    const int64_t nh = (int64_t)a.B * a.cls_count[t] * a.fy * a.fx, ns = (int64_t)a.B * a.max_objs;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nh; i += (int64_t)gridDim.x * kBlock) a.heatmap[t][i] = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ns * 10; i += (int64_t)gridDim.x * kBlock) a.anno[t][i] = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ns; i += (int64_t)gridDim.x * kBlock) {
        a.ind[t][i] = 0;
        a.mask[t][i] = 0;
    }
}

// mmdet3d.core.gaussian_radius((height, width), min_overlap)
__device__ __forceinline__ float cp_gaussian_radius(float h, float w, float o) {
    const float b1 = h + w, c1 = w * h * (1.f - o) / (1.f + o);
    const float r1 = (b1 + sqrtf(b1 * b1 - 4.f * c1)) / 2.f;
    const float b2 = 2.f * (h + w), c2 = (1.f - o) * w * h;
    const float r2 = (b2 + sqrtf(b2 * b2 - 16.f * c2)) / 2.f;
    const float a3 = 4.f * o, b3 = -2.f * o * (h + w), c3 = (o - 1.f) * w * h;
    const float r3 = (b3 + sqrtf(b3 * b3 - 4.f * a3 * c3)) / 2.f;
    return fminf(fminf(r1, r2), r3);
}

__global__ __launch_bounds__(64) void cp_draw_kernel(CpArgs a) {
    const int b = blockIdx.y, k = blockIdx.x;
    const int beg = a.offsets[b];
    const int nk = a.offsets[b + 1] - beg;
    if (k >= nk) return;
    const float *box = a.boxes + (int64_t)(beg + k) * 9;
    const int label = a.labels[beg + k];
    int t = -1;
    for (int i = 0; i < a.T; ++i)
        if (label >= a.cls_begin[i] && label < a.cls_begin[i] + a.cls_count[i]) t = i;
    if (t < 0) return;
    const int cls = label - a.cls_begin[t];
    // the box's slot: its place among the task's boxes in the reference's order -- class after class, inside a class in input order
    // (:141-163: task_boxes = cat over the task's classes) -- and a task takes its first max_objs boxes only (:171)
    int rank = 0;
    for (int j0 = 0; j0 < nk; j0 += 64) {
        const int j = j0 + (int)threadIdx.x;
        bool before = false;
        if (j < nk) {
            const int cj = a.labels[beg + j] - a.cls_begin[t];
            before = cj >= 0 && cj < a.cls_count[t] && (cj < cls || (cj == cls && j < k));
        }
        rank += __popcll(__ballot(before));
    }
    if (rank >= a.max_objs) return;
    const float width = box[3] / a.vx / a.osf, length = box[4] / a.vy / a.osf;      // :176-181
    if (!(width > 0.f && length > 0.f)) return;                                     // :183
    float rf = cp_gaussian_radius(length, width, a.overlap);                        // :184-186
    int radius = (rf == rf) ? (int)rf : 0;
    if (radius < a.min_radius) radius = a.min_radius;                               // :187
    const float cx = (box[0] - a.x0) / a.vx / a.osf, cy = (box[1] - a.y0) / a.vy / a.osf;   // :194-199
    const int xi = (int)cx, yi = (int)cy;                                           // .to(torch.int32)
    if (!(xi >= 0 && xi < a.fx && yi >= 0 && yi < a.fy)) return;                    // :208-210
    // draw_heatmap_gaussian: sigma = diameter / 6, window clipped to the map, max-combine
    const float sigma = (float)(2 * radius + 1) / 6.f;
    const float inv = 1.f / (2.f * sigma * sigma);
    const int left = min(xi, radius), right = min(a.fx - xi, radius + 1);
    const int top = min(yi, radius), bottom = min(a.fy - yi, radius + 1);
    const int ww = left + right, wh = top + bottom;
    float *hm = a.heatmap[t] + (((int64_t)b * a.cls_count[t] + cls) * a.fy) * a.fx;
    for (int i = threadIdx.x; i < ww * wh; i += 64) {
        const int dy = i / ww - top, dx = i - (i / ww) * ww - left;
        const float g = expf(-(float)(dx * dx + dy * dy) * inv);
        atomicMax(reinterpret_cast<unsigned int *>(hm + (int64_t)(yi + dy) * a.fx + xi + dx), __float_as_uint(g));
    }
    if (threadIdx.x == 0) {
        const int64_t slot = (int64_t)b * a.max_objs + rank;
        a.ind[t][slot] = (int64_t)yi * a.fx + xi;                                   // :218
        a.mask[t][slot] = 1;
        float *row = a.anno[t] + slot * 10;                                         // :220-234
        row[0] = cx - (float)xi; row[1] = cy - (float)yi; row[2] = box[2];
        row[3] = a.norm_bbox ? logf(box[3]) : box[3];
        row[4] = a.norm_bbox ? logf(box[4]) : box[4];
        row[5] = a.norm_bbox ? logf(box[5]) : box[5];
        row[6] = sinf(box[6]); row[7] = cosf(box[6]); row[8] = box[7]; row[9] = box[8];
    }
}

}  // namespace

extern "C" int mmt_centerpoint_targets(int B, int num_tasks, const int32_t *class_begin, const int32_t *class_count,
                                       int max_objs, int max_boxes, int fx, int fy, float x0, float y0, float vx,
                                       float vy, int out_size_factor, float gaussian_overlap, int min_radius,
                                       int norm_bbox, const float *boxes, const int32_t *labels,
                                       const int32_t *box_offsets, float *const *heatmaps, float *const *anno_boxes,
                                       int64_t *const *inds, uint8_t *const *masks, void *stream) {
    MMT_REQUIRE_PTR(class_begin);
    MMT_REQUIRE_PTR(class_count);
    MMT_REQUIRE_PTR(box_offsets);
    MMT_REQUIRE_PTR(heatmaps);
    MMT_REQUIRE_PTR(anno_boxes);
    MMT_REQUIRE_PTR(inds);
    MMT_REQUIRE_PTR(masks);
    if (B <= 0 || B > 65535 || num_tasks <= 0 || num_tasks > kMaxTasks || max_objs <= 0 || max_boxes < 0 || fx <= 0 ||
        fy <= 0 || out_size_factor <= 0 || !(vx > 0.f) || !(vy > 0.f))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "centerpoint_targets: bad shape (B=%d tasks=%d max_objs=%d map=%dx%d)", B,
                         num_tasks, max_objs, fx, fy);
    if (max_boxes > 0) {
        MMT_REQUIRE_PTR(boxes);
        MMT_REQUIRE_PTR(labels);
    }
    CpArgs a;
    a.B = B; a.T = num_tasks; a.max_objs = max_objs; a.fx = fx; a.fy = fy; a.norm_bbox = norm_bbox;
    a.x0 = x0; a.y0 = y0; a.vx = vx; a.vy = vy; a.osf = (float)out_size_factor; a.overlap = gaussian_overlap;
    a.min_radius = min_radius;
    a.boxes = boxes; a.labels = labels; a.offsets = box_offsets;
    int max_cls = 1;
    for (int t = 0; t < kMaxTasks; ++t) {
        const bool on = t < num_tasks;
        a.cls_begin[t] = on ? class_begin[t] : 0;
        a.cls_count[t] = on ? class_count[t] : 0;
        a.heatmap[t] = on ? heatmaps[t] : nullptr;
        a.anno[t] = on ? anno_boxes[t] : nullptr;
        a.ind[t] = on ? inds[t] : nullptr;
        a.mask[t] = on ? masks[t] : nullptr;
        if (on) {
            if (!heatmaps[t] || !anno_boxes[t] || !inds[t] || !masks[t] || class_count[t] <= 0)
                return mmt::fail(MMT_ERR_NULL_POINTER, "centerpoint_targets: task %d has a NULL output or no classes", t);
            if (class_count[t] > max_cls) max_cls = class_count[t];
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t biggest = (int64_t)B * max_cls * fy * fx;
    hipLaunchKernelGGL(cp_clear_kernel, dim3(mmt::stream_grid(biggest, kBlock, 1024), num_tasks), dim3(kBlock), 0, st, a);
    if (int rc = mmt::check_launch("centerpoint_targets(clear)")) return rc;
    const int nb = max_boxes;                 // (every box looks for its slot: a task keeps ITS first max_objs boxes, whatever their index in the sample)
    if (nb > 65535 * 64) return mmt::fail(MMT_ERR_TOO_LARGE, "centerpoint_targets: %d boxes in one sample", nb);
    if (nb > 0) {
        hipLaunchKernelGGL(cp_draw_kernel, dim3(nb, B), dim3(64), 0, st, a);
        return mmt::check_launch("centerpoint_targets(draw)");
    }
    return 0;
}
