// BEV-augmentation warp of the pooled camera map:
// models/bev_depth.py:69-84 (BEVDepth.bev_augment_image) =
//   mat = T(+c) @ bda_mat[:3,:3] @ T(-c),  c = ((W-1)/2, (H-1)/2)      (kornia get_affine_matrix2d x2)
//   y   = kornia.geometry.warp_affine(x, mat[:, :2, :3], dsize=(H, W))   bilinear, zeros, align_corners=True
// i.e.  y[b, :, v, u] = bilinear(x[b], M^-1 (u, v, 1))  in pixel coordinates.
// The reference builds the matrices with ~10 small tensor ops, inverts them with a batched LU
// and calls grid_sample on an NCHW tensor.  Here the pooled map is channels-last already
// ([B, H, W, C], one contiguous C-row per cell): one lane group per output cell gathers the four
// neighbour rows as float4 columns; the 2x3 matrix and its inverse are formed in-kernel.  Input
// and output rows may live inside wider channels-last buffers (row strides), so the warped map
// can be written straight into the camera|LiDAR concat buffer (models/bev_depth.py:187-192).
#include "mmt_common.h"

namespace {

constexpr int kBlock = 256;

struct WarpArgs {
    int B, H, W, C;
    int64_t in_stride, out_stride;     // floats between consecutive cells' rows
    const float *bda;                  // [B, 4, 4]
    const float *x;                    // forward: input;  backward: grad of the output
    float *y;                          // forward: output; backward: grad of the input (accumulated into)
};

// inverse of the 2x3 affine map of bev_augment_image for sample b: src = A^-1 (dst - t).
// Matrix algebra in double (a handful of flops per lane): source coordinates reach ~W, where an fp32
// ulp is 3e-5 px, and the result is cast to fp32 only once -- the same as the oracle.
__device__ __forceinline__ void inverse_affine(const float *bda, int H, int W, double (&m)[6]) {
    const double cx = (W - 1) / 2.0, cy = (H - 1) / 2.0;
    const double a = bda[0], b = bda[1], c = bda[4], d = bda[5];
    // T(+c) R T(-c): translation column = R[:2,:2] (-c) + R[:2,2] + c   (R[2,:] = (0,0,1) for a BDA matrix)
    const double tx = (a * -cx + b * -cy) + bda[2] + cx;
    const double ty = (c * -cx + d * -cy) + bda[6] + cy;
    const double det = a * d - b * c;
    const double ia = d / det, ib = -b / det, ic = -c / det, id = a / det;
    m[0] = ia; m[1] = ib; m[2] = -(ia * tx + ib * ty);
    m[3] = ic; m[4] = id; m[5] = -(ic * tx + id * ty);
}

// forward map of bev_augment_image for sample b: dst = A src + t (same double algebra as inverse_affine)
__device__ __forceinline__ void forward_affine(const float *bda, int H, int W, double (&f)[6]) {
    const double cx = (W - 1) / 2.0, cy = (H - 1) / 2.0;
    const double a = bda[0], b = bda[1], c = bda[4], d = bda[5];
    f[0] = a; f[1] = b; f[2] = (a * -cx + b * -cy) + bda[2] + cx;
    f[3] = c; f[4] = d; f[5] = (c * -cx + d * -cy) + bda[6] + cy;
}

// one lane group of C/4 lanes per output cell
// The samples' matrices are worked out ONCE per workgroup (thread b takes sample b: a dozen double operations and four double
// divisions) and read from LDS: round 2 had every thread invert its sample's matrix itself -- some 150 instructions in
// front of every 16-byte row piece, half of the forward's time and a third of the backward's (the kernels are VALU-bound).
constexpr int kWarpMaxB = 64;          // samples whose matrices a workgroup keeps in LDS (more: every thread computes its own)

__global__ __launch_bounds__(kBlock) void bev_warp_kernel(WarpArgs a) {
    __shared__ double s_m[kWarpMaxB][6];
    const bool shared = a.B <= kWarpMaxB;
    if (shared && (int)threadIdx.x < a.B) {
        double t[6];
        inverse_affine(a.bda + threadIdx.x * 16, a.H, a.W, t);
#pragma unroll
        for (int k = 0; k < 6; ++k) s_m[threadIdx.x][k] = t[k];
    }
    __syncthreads();
    const int C4 = a.C >> 2;
    const int64_t cells = (int64_t)a.B * a.H * a.W;
    const int64_t total = cells * C4;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        // (cell, piece) of the item: 32-bit divisions where the item count allows (a 64-bit division is ~150 instructions)
        int64_t cell;
        int c4, b, rem;
        if (total < (1ll << 31)) {
            const unsigned i32 = (unsigned)i, cell32 = i32 / (unsigned)C4, hw = (unsigned)(a.H * a.W);
            c4 = (int)(i32 - cell32 * (unsigned)C4); b = (int)(cell32 / hw); rem = (int)(cell32 - (unsigned)b * hw); cell = cell32;
        } else {
            cell = i / C4; c4 = (int)(i - cell * C4); b = (int)(cell / ((int64_t)a.H * a.W)); rem = (int)(cell - (int64_t)b * a.H * a.W);
        }
        const int v = rem / a.W, u = rem - v * a.W;
        double m[6];
        if (shared) {
#pragma unroll
            for (int k = 0; k < 6; ++k) m[k] = s_m[b][k];
        } else inverse_affine(a.bda + b * 16, a.H, a.W, m);
        const float sx = (float)(m[0] * u + m[1] * v + m[2]);
        const float sy = (float)(m[3] * u + m[4] * v + m[5]);
        const float fx0 = floorf(sx), fy0 = floorf(sy);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float wx1 = sx - fx0, wy1 = sy - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const float w[4] = {wy0 * wx0, wy0 * wx1, wy1 * wx0, wy1 * wx1};
        const int xs[4] = {x0, x0 + 1, x0, x0 + 1}, ys[4] = {y0, y0, y0 + 1, y0 + 1};
        const float *img = a.x + (int64_t)b * a.H * a.W * a.in_stride;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (xs[k] >= 0 && xs[k] < a.W && ys[k] >= 0 && ys[k] < a.H && sx == sx && sy == sy) {
                const float4 p = *reinterpret_cast<const float4 *>(img + ((int64_t)ys[k] * a.W + xs[k]) * a.in_stride + c4 * 4);
                acc.x += w[k] * p.x; acc.y += w[k] * p.y; acc.z += w[k] * p.z; acc.w += w[k] * p.w;
            }
        }
        *reinterpret_cast<float4 *>(a.y + cell * a.out_stride + c4 * 4) = acc;
    }
}

// Backward as a GATHER: one lane group per SOURCE cell (xs, ys).  The output cells whose bilinear footprint
// contains it are those with floor(sx) in {xs-1, xs} and floor(sy) in {ys-1, ys}; they lie inside the image
// of the box [xs-1, xs+1] x [ys-1, ys+1] under the forward map dst = A src + t, whose bounding box is
// centre +- (|A00|+|A01|, |A10|+|A11|): about 3x3 candidates for a BDA rotation / scale ~1.  Every candidate
// recomputes (sx, sy) with the forward kernel's own expression, so the accumulated terms w * g are exactly the
// forward's weights; they are summed in a fixed order in registers and added to the row once.  The first
// version scattered 4 x C scalar fp32 atomics per output cell (21 M atomics on a [4,128,128,80] map): 69 us
// inside the training step, bound by the memory-side atomic units, and not reproducible.
// a.x = grad of the warped map (row stride in_stride), a.y = grad of the source map (accumulated into).
__global__ __launch_bounds__(kBlock) void bev_warp_backward_gather(WarpArgs a) {
    __shared__ double s_m[kWarpMaxB][6], s_f[kWarpMaxB][6];
    const bool shared = a.B <= kWarpMaxB;
    if (shared && (int)threadIdx.x < a.B) {
        double t[6];
        inverse_affine(a.bda + threadIdx.x * 16, a.H, a.W, t);
#pragma unroll
        for (int k = 0; k < 6; ++k) s_m[threadIdx.x][k] = t[k];
        forward_affine(a.bda + threadIdx.x * 16, a.H, a.W, t);
#pragma unroll
        for (int k = 0; k < 6; ++k) s_f[threadIdx.x][k] = t[k];
    }
    __syncthreads();
    const int C4 = a.C >> 2;
    const int64_t cells = (int64_t)a.B * a.H * a.W;
    const int64_t total = cells * C4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x), 0, (int)((cells - 1) * a.in_stride + a.C) * 4, 0x00020000);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        int64_t cell;
        int c4, b, rem;
        if (total < (1ll << 31)) {        // (32-bit divisions: see bev_warp_kernel)
            const unsigned i32 = (unsigned)i, cell32 = i32 / (unsigned)C4, hw = (unsigned)(a.H * a.W);
            c4 = (int)(i32 - cell32 * (unsigned)C4); b = (int)(cell32 / hw); rem = (int)(cell32 - (unsigned)b * hw); cell = cell32;
        } else {
            cell = i / C4; c4 = (int)(i - cell * C4); b = (int)(cell / ((int64_t)a.H * a.W)); rem = (int)(cell - (int64_t)b * a.H * a.W);
        }
        const int ys = rem / a.W, xs = rem - ys * a.W;
        double m[6], f[6];
        if (shared) {
#pragma unroll
            for (int k = 0; k < 6; ++k) { m[k] = s_m[b][k]; f[k] = s_f[b][k]; }
        } else {
            inverse_affine(a.bda + b * 16, a.H, a.W, m);
            forward_affine(a.bda + b * 16, a.H, a.W, f);
        }
        const double uc = f[0] * xs + f[1] * ys + f[2], vc = f[3] * xs + f[4] * ys + f[5];
        const double hu = fabs(f[0]) + fabs(f[1]) + 0.01, hv = fabs(f[3]) + fabs(f[4]) + 0.01;
        // non-finite matrices contribute nothing in the forward either (sx != sx / out of range)
        if (!(fabs(uc) < 1e9) || !(fabs(vc) < 1e9) || !(hu < 1e9) || !(hv < 1e9)) continue;
        // integer points of [uc - hu, uc + hu] x [vc - hv, vc + hv] (hu, hv carry a 0.01 rounding margin)
        const int ulo = (int)fmax(0.0, ceil(uc - hu)), uhi = (int)fmin((double)(a.W - 1), floor(uc + hu));
        const int vlo = (int)fmax(0.0, ceil(vc - hv)), vhi = (int)fmin((double)(a.H - 1), floor(vc + hv));
        // Candidates are taken 4 columns of one row at a time with UNCONDITIONAL loads: a candidate that does not
        // touch this cell (or lies outside the box) gets weight 0 and an out-of-range buffer offset, for which
        // the hardware returns zeros without a memory access -- no branch around the loads (with `if (hit) load`
        // every hit paid its own L2 round trip: 38 us instead of the forward's 17).
        const unsigned row0 = (unsigned)(((int64_t)b * a.H * a.W) * a.in_stride + c4 * 4) * 4u;   // bytes, < 2^31 (host check)
        const unsigned stride_b = (unsigned)a.in_stride * 4u;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int v = vlo; v <= vhi; ++v) {
            for (int u0 = ulo; u0 <= uhi; u0 += 4) {
                float w[4];
                mmt_u32x4 g[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int u = u0 + k;
                    const float sx = (float)(m[0] * u + m[1] * v + m[2]);
                    const float sy = (float)(m[3] * u + m[4] * v + m[5]);
                    const float fx0 = floorf(sx), fy0 = floorf(sy);
                    const int x0 = (int)fx0, y0 = (int)fy0;
                    const bool hit = ((x0 == xs) || (x0 == xs - 1)) && ((y0 == ys) || (y0 == ys - 1)) && sx == sx && sy == sy && u <= uhi;
                    const float wx1 = sx - fx0, wy1 = sy - fy0;
                    const float wk = ((y0 == ys) ? 1.f - wy1 : wy1) * ((x0 == xs) ? 1.f - wx1 : wx1);
                    w[k] = hit ? wk : 0.f;
                    const unsigned off = row0 + ((unsigned)v * (unsigned)a.W + (unsigned)u) * stride_b;
                    g[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (hit && wk != 0.f) ? off : 0xFFFFFFF0u, 0, 0);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc.x += w[k] * __uint_as_float(g[k].x); acc.y += w[k] * __uint_as_float(g[k].y);
                    acc.z += w[k] * __uint_as_float(g[k].z); acc.w += w[k] * __uint_as_float(g[k].w);
                }
            }
        }
        float4 *dst = reinterpret_cast<float4 *>(a.y + cell * a.out_stride + c4 * 4);
        float4 cur = *dst;
        cur.x += acc.x; cur.y += acc.y; cur.z += acc.z; cur.w += acc.w;
        *dst = cur;
    }
}

int check(const char *who, int B, int H, int W, int C, const void *bda, const void *x, const void *y,
          int64_t in_stride, int64_t out_stride) {
    if (!bda || !x || !y) return mmt::fail(MMT_ERR_NULL_POINTER, "%s: NULL pointer", who);
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || in_stride < C || out_stride < C || (in_stride & 3) ||
        (out_stride & 3) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: need C %% 4 == 0, row strides >= C and multiples of 4, 16-byte aligned "
                                            "buffers (B=%d %dx%d C=%d)", who, B, H, W, C);
    if ((int64_t)B * H * W * (in_stride > out_stride ? in_stride : out_stride) >= (1ll << 40))
        return mmt::fail(MMT_ERR_TOO_LARGE, "%s: map too large", who);
    return 0;
}

}  // namespace

extern "C" int mmt_bev_warp_affine(int B, int H, int W, int C, const float *bda_mat, const float *input,
                                   int64_t in_row_stride, float *output, int64_t out_row_stride, void *stream) {
    if (int rc = check("bev_warp_affine", B, H, W, C, bda_mat, input, output, in_row_stride, out_row_stride)) return rc;
    WarpArgs a{B, H, W, C, in_row_stride, out_row_stride, bda_mat, input, output};
    const int64_t work = (int64_t)B * H * W * (C / 4);
    mmt::TimedSeq seq;      // (kernel timing for the bench: mmt_arm_kernel_timing)
    seq.launch(true, bev_warp_kernel, dim3(mmt::stream_grid(work, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, a);
    return mmt::check_launch("bev_warp_affine");
}

extern "C" int mmt_bev_warp_affine_backward(int B, int H, int W, int C, const float *bda_mat, const float *grad_output,
                                            int64_t grad_out_row_stride, float *grad_input,
                                            int64_t grad_in_row_stride, void *stream) {
    if (int rc = check("bev_warp_affine_backward", B, H, W, C, bda_mat, grad_output, grad_input, grad_out_row_stride,
                       grad_in_row_stride))
        return rc;
    if (((int64_t)B * H * W * grad_out_row_stride) * 4 >= (1ll << 31))
        return mmt::fail(MMT_ERR_TOO_LARGE, "bev_warp_affine_backward: grad_output spans 2 GiB or more");
    WarpArgs a{B, H, W, C, grad_out_row_stride, grad_in_row_stride, bda_mat, grad_output, grad_input};
    const int64_t work = (int64_t)B * H * W * (C / 4);
    mmt::TimedSeq seq;
    seq.launch(true, bev_warp_backward_gather, dim3(mmt::stream_grid(work, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, a);
    return mmt::check_launch("bev_warp_affine_backward");
}
