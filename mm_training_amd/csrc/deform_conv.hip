// Deformable convolution v1 (mmcv 'DCN' = DeformConv2dPack, the one non-PyTorch dense op
// on the camera path: layers/backbones/lss_fpn.py:189-197) for MI355X.
//
// 3x3, stride 1, pad 1, dilation 1, deform_groups 1, `groups` weight groups.  Split the
// mmcv way: a deformable im2col (bilinear sampling with zero padding) followed by a plain
// grouped GEMM -- the GEMM is left to rocBLAS/hipBLASLt (MFMA) through torch.bmm, the
// data-dependent gather / scatter is hand-written here.  Everything is channels-last:
//   x       fp32 [B, H, W, C]
//   offset  fp32 [B, H, W, 18]   (dy, dx) per kernel tap, tap k = ky*3 + kx
//   col     fp32 [groups][B*H*W][9 * C/groups]   K index = k * (C/groups) + c_in_group
// so a sampling point reads 4 contiguous C-vectors and a wave's lanes run along channels
// (16 bytes per lane, no bank or address divergence inside a tap).
#include "mmt_common.h"

namespace {

struct Tap {
    float w1, w2, w3, w4;   // bilinear weights of (y0,x0) (y0,x1) (y1,x0) (y1,x1), 0 where outside
    int o1, o2, o3, o4;     // pixel indices y*W+x of the four corners (clamped when outside)
    float dy1, dy2, dy3, dy4, dx1, dx2, dx3, dx4;  // d(weight)/d(py), d(weight)/d(px)
};

// mmcv deformable_im2col_bilinear semantics: value 0 unless -1 < p < size; corners outside
// the image contribute 0.
__device__ __forceinline__ Tap make_tap(float py, float px, int H, int W) {
    Tap t;
    t.w1 = t.w2 = t.w3 = t.w4 = 0.f;
    t.dy1 = t.dy2 = t.dy3 = t.dy4 = t.dx1 = t.dx2 = t.dx3 = t.dx4 = 0.f;
    t.o1 = t.o2 = t.o3 = t.o4 = 0;
    if (!(py > -1.f && px > -1.f && py < (float)H && px < (float)W)) return t;
    const int y0 = (int)floorf(py), x0 = (int)floorf(px);
    const int y1 = y0 + 1, x1 = x0 + 1;
    const float ly = py - (float)y0, lx = px - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const bool vy0 = y0 >= 0, vy1 = y1 <= H - 1, vx0 = x0 >= 0, vx1 = x1 <= W - 1;
    const int cy0 = vy0 ? y0 : 0, cy1 = vy1 ? y1 : H - 1, cx0 = vx0 ? x0 : 0, cx1 = vx1 ? x1 : W - 1;
    t.o1 = cy0 * W + cx0; t.o2 = cy0 * W + cx1; t.o3 = cy1 * W + cx0; t.o4 = cy1 * W + cx1;
    const float m1 = (vy0 && vx0) ? 1.f : 0.f, m2 = (vy0 && vx1) ? 1.f : 0.f;
    const float m3 = (vy1 && vx0) ? 1.f : 0.f, m4 = (vy1 && vx1) ? 1.f : 0.f;
    t.w1 = hy * hx * m1; t.w2 = hy * lx * m2; t.w3 = ly * hx * m3; t.w4 = ly * lx * m4;
    // val = hy*hx*v1 + hy*lx*v2 + ly*hx*v3 + ly*lx*v4 ; d/dpy = d/dly, d/dpx = d/dlx
    t.dy1 = -hx * m1; t.dy2 = -lx * m2; t.dy3 = hx * m3; t.dy4 = lx * m4;
    t.dx1 = -hy * m1; t.dx2 = hy * m2; t.dx3 = -ly * m3; t.dx4 = ly * m4;
    return t;
}

__device__ __forceinline__ float4 fma4(float w, float4 v, float4 acc) {
    return make_float4(acc.x + w * v.x, acc.y + w * v.y, acc.z + w * v.z, acc.w + w * v.w);
}

// one wave per output position, 4 positions per workgroup
__global__ __launch_bounds__(256) void dcn_im2col_kernel(int B, int H, int W, int C, int groups,
                                                         const float *x, const float *offset, float *col) {
    const int lane = threadIdx.x & 63;
    const int64_t npos = (int64_t)B * H * W;
    const int Cg = C / groups;
    const int C4 = C >> 2;
    for (int64_t pos = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); pos < npos; pos += (int64_t)gridDim.x * 4) {
        const int b = (int)(pos / (H * W));
        const int hw = (int)(pos - (int64_t)b * H * W);
        const int h = hw / W, w = hw - h * W;
        const float *xb = x + (int64_t)b * H * W * C;
        const float *op = offset + pos * 18;
#pragma unroll 1
        for (int k = 0; k < 9; ++k) {
            const int ky = k / 3, kx = k - ky * 3;
            const Tap t = make_tap((float)(h + ky - 1) + op[2 * k], (float)(w + kx - 1) + op[2 * k + 1], H, W);
            for (int c4 = lane; c4 < C4; c4 += 64) {
                const int c = c4 * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t.w1 != 0.f) v = fma4(t.w1, *reinterpret_cast<const float4 *>(xb + (int64_t)t.o1 * C + c), v);
                if (t.w2 != 0.f) v = fma4(t.w2, *reinterpret_cast<const float4 *>(xb + (int64_t)t.o2 * C + c), v);
                if (t.w3 != 0.f) v = fma4(t.w3, *reinterpret_cast<const float4 *>(xb + (int64_t)t.o3 * C + c), v);
                if (t.w4 != 0.f) v = fma4(t.w4, *reinterpret_cast<const float4 *>(xb + (int64_t)t.o4 * C + c), v);
                const int g = c / Cg, cin = c - g * Cg;
                float *dst = col + ((int64_t)g * npos + pos) * (9 * Cg) + k * Cg + cin;
                mmt_nt_store4(v, reinterpret_cast<float4 *>(dst));
            }
        }
    }
}

// grad_col -> grad_x (scatter with fp32 atomics, zero-weight corners skipped) and
// grad_offset (channel reduction across the wave).  Here a lane owns channels
// lane, lane+64, lane+128, ... (not 4 consecutive ones): every atomic wave instruction then
// covers 256 CONTIGUOUS bytes, the shape the memory-side atomic units run at full rate
// (MI355X_MICROARCH.md "Global float atomics"); a float4-per-lane layout would spray each
// instruction over sixteen 64-byte segments.
__global__ __launch_bounds__(256) void dcn_col2im_kernel(int B, int H, int W, int C, int groups,
                                                         const float *x, const float *offset,
                                                         const float *grad_col, float *grad_x,
                                                         float *grad_offset) {
    const int lane = threadIdx.x & 63;
    const int64_t npos = (int64_t)B * H * W;
    const int Cg = C / groups;
    for (int64_t pos = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); pos < npos; pos += (int64_t)gridDim.x * 4) {
        const int b = (int)(pos / (H * W));
        const int hw = (int)(pos - (int64_t)b * H * W);
        const int h = hw / W, w = hw - h * W;
        const float *xb = x + (int64_t)b * H * W * C;
        float *gxb = grad_x + (int64_t)b * H * W * C;
        const float *op = offset + pos * 18;
#pragma unroll 1
        for (int k = 0; k < 9; ++k) {
            const int ky = k / 3, kx = k - ky * 3;
            const Tap t = make_tap((float)(h + ky - 1) + op[2 * k], (float)(w + kx - 1) + op[2 * k + 1], H, W);
            float gy = 0.f, gx = 0.f;
            const float *x1 = xb + (int64_t)t.o1 * C, *x2 = xb + (int64_t)t.o2 * C;
            const float *x3 = xb + (int64_t)t.o3 * C, *x4 = xb + (int64_t)t.o4 * C;
            float *d1 = gxb + (int64_t)t.o1 * C, *d2 = gxb + (int64_t)t.o2 * C;
            float *d3 = gxb + (int64_t)t.o3 * C, *d4 = gxb + (int64_t)t.o4 * C;
            for (int c = lane; c < C; c += 64) {
                const int g = c / Cg, cin = c - g * Cg;
                const float gc = grad_col[((int64_t)g * npos + pos) * (9 * Cg) + k * Cg + cin];
                const float v1 = x1[c], v2 = x2[c], v3 = x3[c], v4 = x4[c];
                gy += gc * (t.dy1 * v1 + t.dy2 * v2 + t.dy3 * v3 + t.dy4 * v4);
                gx += gc * (t.dx1 * v1 + t.dx2 * v2 + t.dx3 * v3 + t.dx4 * v4);
                // wave-uniform conditions: the tap is the same for all lanes
                if (t.w1 != 0.f) atomicAdd(d1 + c, t.w1 * gc);
                if (t.w2 != 0.f) atomicAdd(d2 + c, t.w2 * gc);
                if (t.w3 != 0.f) atomicAdd(d3 + c, t.w3 * gc);
                if (t.w4 != 0.f) atomicAdd(d4 + c, t.w4 * gc);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                gy += __shfl_xor(gy, o);
                gx += __shfl_xor(gx, o);
            }
            if (lane == 0) {
                grad_offset[pos * 18 + 2 * k] = gy;
                grad_offset[pos * 18 + 2 * k + 1] = gx;
            }
        }
    }
}

int check_dcn(int B, int H, int W, int C, int groups, const char *what) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || groups <= 0 || C % groups != 0 || (C / groups) % 4 != 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: need C %% groups == 0 and (C/groups) %% 4 == 0 (B=%d H=%d W=%d C=%d groups=%d)",
                         what, B, H, W, C, groups);
    if ((int64_t)B * H * W >= (1ll << 31)) return mmt::fail(MMT_ERR_TOO_LARGE, "%s: B*H*W exceeds int32", what);
    return 0;
}

}  // namespace

extern "C" int mmt_dcn_im2col(int B, int H, int W, int C, int groups, const float *x,
                              const float *offset, float *col, void *stream) {
    MMT_REQUIRE_PTR(x);
    MMT_REQUIRE_PTR(offset);
    MMT_REQUIRE_PTR(col);
    if (int rc = check_dcn(B, H, W, C, groups, "dcn_im2col")) return rc;
    if ((((uintptr_t)x | (uintptr_t)col) & 15) != 0) return mmt::fail(MMT_ERR_BAD_SHAPE, "dcn_im2col: x / col must be 16-byte aligned");
    const int64_t npos = (int64_t)B * H * W;
    hipLaunchKernelGGL(dcn_im2col_kernel, dim3(mmt::stream_grid(npos, 4, 256 * 32)), dim3(256), 0,
                       (hipStream_t)stream, B, H, W, C, groups, x, offset, col);
    return mmt::check_launch("dcn_im2col");
}

extern "C" int mmt_dcn_col2im(int B, int H, int W, int C, int groups, const float *x,
                              const float *offset, const float *grad_col, float *grad_x,
                              float *grad_offset, void *stream) {
    MMT_REQUIRE_PTR(x);
    MMT_REQUIRE_PTR(offset);
    MMT_REQUIRE_PTR(grad_col);
    MMT_REQUIRE_PTR(grad_x);
    MMT_REQUIRE_PTR(grad_offset);
    if (int rc = check_dcn(B, H, W, C, groups, "dcn_col2im")) return rc;
    if ((((uintptr_t)x | (uintptr_t)grad_col) & 15) != 0) return mmt::fail(MMT_ERR_BAD_SHAPE, "dcn_col2im: x / grad_col must be 16-byte aligned");
    const int64_t npos = (int64_t)B * H * W;
    hipLaunchKernelGGL(dcn_col2im_kernel, dim3(mmt::stream_grid(npos, 4, 256 * 32)), dim3(256), 0,
                       (hipStream_t)stream, B, H, W, C, groups, x, offset, grad_col, grad_x, grad_offset);
    return mmt::check_launch("dcn_col2im");
}
