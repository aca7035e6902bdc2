// One sampling point of the deformable convolution (mmcv 'DCN' v1, lss_fpn.py:189-197): its four bilinear corners.
// Shared by deform_conv.hip (im2col / col2im, the general fallback) and deform_conv_mfma.hip (implicit GEMM).
#pragma once
#include <hip/hip_runtime.h>

namespace mmt_dcn {

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

// the sampling point of kernel tap k (= ky*3 + kx) at output pixel (h, w): op = the pixel's 18 offsets (dy, dx per tap)
__device__ __forceinline__ Tap tap_at(int h, int w, int k, const float *op, int H, int W) {
    const int ky = k / 3, kx = k - ky * 3;
    return make_tap((float)(h + ky - 1) + op[2 * k], (float)(w + kx - 1) + op[2 * k + 1], H, W);
}

}  // namespace mmt_dcn
