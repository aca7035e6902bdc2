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
    float *y;                          // forward: output; backward: grad of the input (accumulated into, or assigned)
    int assign;                        // backward: every row of y is written exactly once -- store it instead of adding to what is there
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
// Round 6: both kernels in TWO PHASES per workgroup of kCells cells.  Phase 1: one THREAD per cell does the cell's index
// work once -- source coordinates in double, the four taps (forward) or the candidate output cells whose footprint holds the
// cell (backward), their byte offsets and weights -- into LDS.  Phase 2: the cell's C / 4 lanes read the (offset, weight) list
// and move the rows.  Round 2 had every lane of a cell's lane group repeat the index work: a wave of 64 lanes served 3.2
// cells, and the backward's dozen candidates (two double FMAs, floors, tests each) made it VALU-bound at 0.30 of the HBM rate.
// The arithmetic per cell is unchanged (same expressions, same order of the sums): same bits as before.
constexpr int kCells = 64;             // cells per workgroup (phase 1: the first kCells threads)
constexpr int kHits = 8;               // backward: candidates with a weight kept in LDS per cell (more: the cell's lanes do it the long way)
constexpr int kWarpMaxB = 64;          // samples whose matrices a workgroup keeps in LDS (more: recomputed per cell)
constexpr unsigned kNoRow = 0xFFFFFFF0u;   // a buffer offset beyond every map: the load returns zeros without a memory access

__device__ __forceinline__ void cell_of(int64_t cell, int H, int W, int &b, int &y, int &x) {
    if (cell < (1ll << 31)) { const unsigned c = (unsigned)cell, hw = (unsigned)(H * W); b = (int)(c / hw); const unsigned r = c - (unsigned)b * hw; y = (int)(r / (unsigned)W); x = (int)(r - (unsigned)y * W); }
    else { b = (int)(cell / ((int64_t)H * W)); const int r = (int)(cell - (int64_t)b * H * W); y = r / W; x = r - y * W; }
}

__global__ __launch_bounds__(kBlock) void bev_warp_kernel(WarpArgs a) {
    __shared__ double s_m[kWarpMaxB][6];
    __shared__ unsigned s_off[kCells][4];
    __shared__ float s_w[kCells][4];
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
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x), 0, (int)((cells - 1) * a.in_stride + a.C) * 4, 0x00020000);
    for (int64_t c0 = (int64_t)blockIdx.x * kCells; c0 < cells; c0 += (int64_t)gridDim.x * kCells) {
        if (threadIdx.x < kCells) {
            const int64_t cell = c0 + threadIdx.x;
            unsigned off[4] = {kNoRow, kNoRow, kNoRow, kNoRow};
            float w[4] = {0.f, 0.f, 0.f, 0.f};
            if (cell < cells) {
                int b, v, u;
                cell_of(cell, a.H, a.W, b, v, u);
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
                const float wk[4] = {wy0 * wx0, wy0 * wx1, wy1 * wx0, wy1 * wx1};
                const int xs[4] = {x0, x0 + 1, x0, x0 + 1}, ys[4] = {y0, y0, y0 + 1, y0 + 1};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (xs[k] >= 0 && xs[k] < a.W && ys[k] >= 0 && ys[k] < a.H && sx == sx && sy == sy) {
                        off[k] = (unsigned)((((int64_t)b * a.H + ys[k]) * a.W + xs[k]) * a.in_stride) * 4u;     // (bytes < 2^32: host check)
                        w[k] = wk[k];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { s_off[threadIdx.x][k] = off[k]; s_w[threadIdx.x][k] = w[k]; }
        }
        __syncthreads();
        const int64_t left = cells - c0;
        const int ncell = left < kCells ? (int)left : kCells;
        for (int it = threadIdx.x; it < ncell * C4; it += kBlock) {
            const int cl = it / C4, c4 = it - cl * C4;
            mmt_u32x4 p[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned o = s_off[cl][k];
                p[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o == kNoRow ? kNoRow : o + (unsigned)c4 * 16u, 0, 0);
            }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float w = s_w[cl][k];
                if (s_off[cl][k] != kNoRow) {            // (a tap outside the map adds nothing -- not even 0 * x: round 2's sums exactly)
                    acc.x += w * __uint_as_float(p[k].x); acc.y += w * __uint_as_float(p[k].y);
                    acc.z += w * __uint_as_float(p[k].z); acc.w += w * __uint_as_float(p[k].w);
                }
            }
            *reinterpret_cast<float4 *>(a.y + (c0 + cl) * a.out_stride + c4 * 4) = acc;
        }
        __syncthreads();
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
struct Cand { int ulo, uhi, vlo, vhi; };      // the candidate box of a source cell (empty: vlo > vhi)

__device__ __forceinline__ Cand candidates(const double (&f)[6], int xs, int ys, int H, int W) {
    const double uc = f[0] * xs + f[1] * ys + f[2], vc = f[3] * xs + f[4] * ys + f[5];
    const double hu = fabs(f[0]) + fabs(f[1]) + 0.01, hv = fabs(f[3]) + fabs(f[4]) + 0.01;
    // non-finite matrices contribute nothing in the forward either (sx != sx / out of range)
    if (!(fabs(uc) < 1e9) || !(fabs(vc) < 1e9) || !(hu < 1e9) || !(hv < 1e9)) return Cand{0, -1, 0, -1};
    // integer points of [uc - hu, uc + hu] x [vc - hv, vc + hv] (hu, hv carry a 0.01 rounding margin)
    return Cand{(int)fmax(0.0, ceil(uc - hu)), (int)fmin((double)(W - 1), floor(uc + hu)), (int)fmax(0.0, ceil(vc - hv)), (int)fmin((double)(H - 1), floor(vc + hv))};
}
// weight of output cell (u, v) on source cell (xs, ys): the forward's own expression; 0 when the footprint does not hold the cell
__device__ __forceinline__ float cand_weight(const double (&m)[6], int u, int v, int xs, int ys) {
    const float sx = (float)(m[0] * u + m[1] * v + m[2]);
    const float sy = (float)(m[3] * u + m[4] * v + m[5]);
    const float fx0 = floorf(sx), fy0 = floorf(sy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const bool hit = ((x0 == xs) || (x0 == xs - 1)) && ((y0 == ys) || (y0 == ys - 1)) && sx == sx && sy == sy;
    const float wx1 = sx - fx0, wy1 = sy - fy0;
    const float wk = ((y0 == ys) ? 1.f - wy1 : wy1) * ((x0 == xs) ? 1.f - wx1 : wx1);
    return hit ? wk : 0.f;
}

__global__ __launch_bounds__(kBlock) void bev_warp_backward_gather(WarpArgs a) {
    __shared__ double s_m[kWarpMaxB][6], s_f[kWarpMaxB][6];
    __shared__ unsigned s_off[kCells][kHits];
    __shared__ float s_w[kCells][kHits];
    __shared__ int s_n[kCells];            // hits kept in LDS, or -1: more than kHits (the lanes walk the candidates themselves)
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
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x), 0, (int)((cells - 1) * a.in_stride + a.C) * 4, 0x00020000);
    const unsigned stride_b = (unsigned)a.in_stride * 4u;
    auto matrices = [&](int b, double (&m)[6], double (&f)[6]) __attribute__((always_inline)) {
        if (shared) {
#pragma unroll
            for (int k = 0; k < 6; ++k) { m[k] = s_m[b][k]; f[k] = s_f[b][k]; }
        } else {
            inverse_affine(a.bda + b * 16, a.H, a.W, m);
            forward_affine(a.bda + b * 16, a.H, a.W, f);
        }
    };
    for (int64_t c0 = (int64_t)blockIdx.x * kCells; c0 < cells; c0 += (int64_t)gridDim.x * kCells) {
        if (threadIdx.x < kCells) {
            const int64_t cell = c0 + threadIdx.x;
            int n = 0;
            if (cell < cells) {
                int b, ys, xs;
                cell_of(cell, a.H, a.W, b, ys, xs);
                double m[6], f[6];
                matrices(b, m, f);
                const Cand cd = candidates(f, xs, ys, a.H, a.W);
                const unsigned img = (unsigned)(((int64_t)b * a.H * a.W) * a.in_stride) * 4u;      // bytes, < 2^31 (host check)
                for (int v = cd.vlo; v <= cd.vhi && n >= 0; ++v)
                    for (int u = cd.ulo; u <= cd.uhi; ++u) {
                        const float wk = cand_weight(m, u, v, xs, ys);
                        if (wk != 0.f) {
                            if (n == kHits) { n = -1; break; }
                            s_off[threadIdx.x][n] = img + ((unsigned)v * (unsigned)a.W + (unsigned)u) * stride_b;
                            s_w[threadIdx.x][n] = wk;
                            ++n;
                        }
                    }
            }
            s_n[threadIdx.x] = n;
        }
        __syncthreads();
        const int64_t left = cells - c0;
        const int ncell = left < kCells ? (int)left : kCells;
        for (int it = threadIdx.x; it < ncell * C4; it += kBlock) {
            const int cl = it / C4, c4 = it - cl * C4;
            const int n = s_n[cl];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n >= 0) {
                // the hits four at a time with UNCONDITIONAL loads (a slot past the list: weight 0 and an out-of-range offset, for which
                // the hardware returns zeros without a memory access): with `if (hit) load` every hit paid its own L2 round trip
                for (int j0 = 0; j0 < n; j0 += 4) {
                    float w[4];
                    mmt_u32x4 g[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool on = j0 + k < n;
                        w[k] = on ? s_w[cl][(j0 + k) & (kHits - 1)] : 0.f;
                        g[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, on ? s_off[cl][(j0 + k) & (kHits - 1)] + (unsigned)c4 * 16u : kNoRow, 0, 0);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        acc.x += w[k] * __uint_as_float(g[k].x); acc.y += w[k] * __uint_as_float(g[k].y);
                        acc.z += w[k] * __uint_as_float(g[k].z); acc.w += w[k] * __uint_as_float(g[k].w);
                    }
                }
            } else {
                // a source cell under more than kHits output cells (a map zoomed in): every lane walks the candidates itself, as round 2 did
                int b, ys, xs;
                cell_of(c0 + cl, a.H, a.W, b, ys, xs);
                double m[6], f[6];
                matrices(b, m, f);
                const Cand cd = candidates(f, xs, ys, a.H, a.W);
                const unsigned row0 = (unsigned)(((int64_t)b * a.H * a.W) * a.in_stride + c4 * 4) * 4u;
                for (int v = cd.vlo; v <= cd.vhi; ++v)
                    for (int u0 = cd.ulo; u0 <= cd.uhi; u0 += 4) {
                        float w[4];
                        mmt_u32x4 g[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int u = u0 + k;
                            const float wk = u <= cd.uhi ? cand_weight(m, u, v, xs, ys) : 0.f;
                            w[k] = wk;
                            g[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, wk != 0.f ? row0 + ((unsigned)v * (unsigned)a.W + (unsigned)u) * stride_b : kNoRow, 0, 0);
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            acc.x += w[k] * __uint_as_float(g[k].x); acc.y += w[k] * __uint_as_float(g[k].y);
                            acc.z += w[k] * __uint_as_float(g[k].z); acc.w += w[k] * __uint_as_float(g[k].w);
                        }
                    }
            }
            float4 *dst = reinterpret_cast<float4 *>(a.y + (c0 + cl) * a.out_stride + c4 * 4);
            if (a.assign) *dst = acc;
            else {
                float4 cur = *dst;
                cur.x += acc.x; cur.y += acc.y; cur.z += acc.z; cur.w += acc.w;
                *dst = cur;
            }
        }
        __syncthreads();
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
    if (((int64_t)B * H * W * in_row_stride) * 4 >= (1ll << 31))     // (the rows are read through a buffer descriptor: 32-bit byte offsets)
        return mmt::fail(MMT_ERR_TOO_LARGE, "bev_warp_affine: input spans 2 GiB or more");
    WarpArgs a{B, H, W, C, in_row_stride, out_row_stride, bda_mat, input, output, 0};
    const int64_t groups = ((int64_t)B * H * W + kCells - 1) / kCells;
    mmt::TimedSeq seq;      // (kernel timing for the bench: mmt_arm_kernel_timing)
    seq.launch(true, bev_warp_kernel, dim3((unsigned)(groups < 16384 ? groups : 16384)), dim3(kBlock), 0, (hipStream_t)stream, a);
    return mmt::check_launch("bev_warp_affine");
}

static int warp_backward(const char *who, int assign, int B, int H, int W, int C, const float *bda_mat, const float *grad_output,
                         int64_t grad_out_row_stride, float *grad_input, int64_t grad_in_row_stride, void *stream) {
    if (int rc = check(who, B, H, W, C, bda_mat, grad_output, grad_input, grad_out_row_stride, grad_in_row_stride)) return rc;
    if (((int64_t)B * H * W * grad_out_row_stride) * 4 >= (1ll << 31))
        return mmt::fail(MMT_ERR_TOO_LARGE, "%s: grad_output spans 2 GiB or more", who);
    WarpArgs a{B, H, W, C, grad_out_row_stride, grad_in_row_stride, bda_mat, grad_output, grad_input, assign};
    const int64_t groups = ((int64_t)B * H * W + kCells - 1) / kCells;
    mmt::TimedSeq seq;
    seq.launch(true, bev_warp_backward_gather, dim3((unsigned)(groups < 16384 ? groups : 16384)), dim3(kBlock), 0, (hipStream_t)stream, a);
    return mmt::check_launch(who);
}

extern "C" int mmt_bev_warp_affine_backward(int B, int H, int W, int C, const float *bda_mat, const float *grad_output,
                                            int64_t grad_out_row_stride, float *grad_input,
                                            int64_t grad_in_row_stride, void *stream) {
    return warp_backward("bev_warp_affine_backward", 0, B, H, W, C, bda_mat, grad_output, grad_out_row_stride, grad_input, grad_in_row_stride, stream);
}

extern "C" int mmt_bev_warp_affine_backward_assign(int B, int H, int W, int C, const float *bda_mat, const float *grad_output,
                                                   int64_t grad_out_row_stride, float *grad_input,
                                                   int64_t grad_in_row_stride, void *stream) {
    return warp_backward("bev_warp_affine_backward_assign", 1, B, H, W, C, bda_mat, grad_output, grad_out_row_stride, grad_input, grad_in_row_stride, stream);
}
