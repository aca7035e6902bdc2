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

// one wave per output position, 4 positions per workgroup; three taps = twelve 16-byte corner loads in flight per lane, no
// branches (a corner outside the image is read at its clamped position and replaced by an exact 0, as mmcv's if does)
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
        for (int c4 = lane; c4 < C4; c4 += 64) {
            const int c = c4 * 4;
            const int g = c / Cg, cin = c - g * Cg;
            float *dst0 = col + ((int64_t)g * npos + pos) * (9 * Cg) + cin;
#pragma unroll 1
            for (int k0 = 0; k0 < 9; k0 += 3) {
                Tap t[3];
                float4 v[3][4];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int k = k0 + j;
                    const int ky = k / 3, kx = k - ky * 3;
                    t[j] = make_tap((float)(h + ky - 1) + op[2 * k], (float)(w + kx - 1) + op[2 * k + 1], H, W);
                    v[j][0] = *reinterpret_cast<const float4 *>(xb + (int64_t)t[j].o1 * C + c);
                    v[j][1] = *reinterpret_cast<const float4 *>(xb + (int64_t)t[j].o2 * C + c);
                    v[j][2] = *reinterpret_cast<const float4 *>(xb + (int64_t)t[j].o3 * C + c);
                    v[j][3] = *reinterpret_cast<const float4 *>(xb + (int64_t)t[j].o4 * C + c);
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                    float4 r = z;
                    r = fma4(t[j].w1, t[j].w1 != 0.f ? v[j][0] : z, r);
                    r = fma4(t[j].w2, t[j].w2 != 0.f ? v[j][1] : z, r);
                    r = fma4(t[j].w3, t[j].w3 != 0.f ? v[j][2] : z, r);
                    r = fma4(t[j].w4, t[j].w4 != 0.f ? v[j][3] : z, r);
                    mmt_nt_store4(r, reinterpret_cast<float4 *>(dst0 + (k0 + j) * Cg));
                }
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

// ---------------------------------------------------------------------------
// col2im WITHOUT global atomics (the kernel above moves 1.24 GB of fp32 atomics at the DepthNet shape: 870 us).
// With free offsets the set of (position, tap) pairs that reach a pixel is not bounded by a neighbourhood, so a
// plain gather needs the contributions sorted by destination first.  Three kernels:
//   dcn_plan_kernel        one workgroup per image: every (position, tap) has up to 4 bilinear corners; the corners
//                          with non-zero weight are binned by destination pixel in LDS (count -> scan -> counting
//                          sort) and leave as a per-pixel list of (source position, tap, weight) entries;
//   dcn_offset_grad_kernel grad_offset[pos, tap] = sum_c grad_col * d(sample)/d(py, px): a streaming read of grad_col
//                          with the four x rows of the tap gathered from L2, channel reduction across the wave;
//   dcn_col2im_gather      grad_x[pixel, group] = sum over the pixel's list of  weight * grad_col[source, tap, group]:
//                          one lane group (Cg/4 lanes, a float4 column each) per (pixel, weight group), rows summed in
//                          registers, ONE plain store per output row -- grad_x is overwritten (no zero-fill needed).
// The order of a pixel's list is the LDS arrival order, so the fp32 sum order may differ between runs (as with the
// atomics before).
constexpr int kPlanThreads = 1024;
constexpr int kPlanMaxHW = 4096;     // destination bins held in LDS

struct DcnEntry { int src_tap; float w; };    // (source position inside the image << 4) | tap

__global__ __launch_bounds__(kPlanThreads) void dcn_plan_kernel(int H, int W, const float *offset, int32_t *bin_off,
                                                                DcnEntry *entries) {
    __shared__ int cnt[kPlanMaxHW];
    __shared__ int off[kPlanMaxHW + 1];
    __shared__ int wsum[kPlanThreads / 64];
    const int HW = H * W;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *ob = offset + (int64_t)b * HW * 18;
    for (int i = tid; i < HW; i += kPlanThreads) cnt[i] = 0;
    __syncthreads();
    for (int e = tid; e < HW * 9; e += kPlanThreads) {
        const int pos = e / 9, k = e - pos * 9;
        const int h = pos / W, w = pos - h * W, ky = k / 3, kx = k - ky * 3;
        const Tap t = make_tap((float)(h + ky - 1) + ob[e * 2], (float)(w + kx - 1) + ob[e * 2 + 1], H, W);
        if (t.w1 != 0.f) atomicAdd(&cnt[t.o1], 1);
        if (t.w2 != 0.f) atomicAdd(&cnt[t.o2], 1);
        if (t.w3 != 0.f) atomicAdd(&cnt[t.o3], 1);
        if (t.w4 != 0.f) atomicAdd(&cnt[t.o4], 1);
    }
    __syncthreads();
    // exclusive scan of the HW counts: kPer consecutive bins per thread, wave scan, then the wave totals
    constexpr int kPer = kPlanMaxHW / kPlanThreads;
    int loc[kPer], sum = 0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int idx = tid * kPer + i;
        loc[i] = idx < HW ? cnt[idx] : 0;
        sum += loc[i];
    }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    int run = base + incl - sum;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int idx = tid * kPer + i;
        if (idx < HW) { off[idx] = run; cnt[idx] = 0; }
        run += loc[i];
    }
    if (tid == kPlanThreads - 1) off[HW] = run;
    __syncthreads();
    for (int i = tid; i <= HW; i += kPlanThreads) bin_off[(int64_t)b * (HW + 1) + i] = off[i];
    DcnEntry *eb = entries + (int64_t)b * HW * 36;
    for (int e = tid; e < HW * 9; e += kPlanThreads) {
        const int pos = e / 9, k = e - pos * 9;
        const int h = pos / W, w = pos - h * W, ky = k / 3, kx = k - ky * 3;
        const Tap t = make_tap((float)(h + ky - 1) + ob[e * 2], (float)(w + kx - 1) + ob[e * 2 + 1], H, W);
        const int st = (pos << 4) | k;
        if (t.w1 != 0.f) eb[off[t.o1] + atomicAdd(&cnt[t.o1], 1)] = DcnEntry{st, t.w1};
        if (t.w2 != 0.f) eb[off[t.o2] + atomicAdd(&cnt[t.o2], 1)] = DcnEntry{st, t.w2};
        if (t.w3 != 0.f) eb[off[t.o3] + atomicAdd(&cnt[t.o3], 1)] = DcnEntry{st, t.w3};
        if (t.w4 != 0.f) eb[off[t.o4] + atomicAdd(&cnt[t.o4], 1)] = DcnEntry{st, t.w4};
    }
}

// one wave per (position, tap): lanes own float4 columns of the C channels
__global__ __launch_bounds__(256) void dcn_offset_grad_kernel(int B, int H, int W, int C, int groups, const float *x,
                                                              const float *offset, const float *grad_col,
                                                              float *grad_offset) {
    const int lane = threadIdx.x & 63;
    const int64_t npos = (int64_t)B * H * W;
    const int64_t nitems = npos * 9;
    const int Cg = C / groups, C4 = C >> 2;
    for (int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < nitems; it += (int64_t)gridDim.x * 4) {
        const int64_t pos = it / 9;
        const int k = (int)(it - pos * 9);
        const int b = (int)(pos / (H * W));
        const int hw = (int)(pos - (int64_t)b * H * W);
        const int h = hw / W, w = hw - h * W, ky = k / 3, kx = k - ky * 3;
        const Tap t = make_tap((float)(h + ky - 1) + offset[it * 2], (float)(w + kx - 1) + offset[it * 2 + 1], H, W);
        const float *xb = x + (int64_t)b * H * W * C;
        const float *x1 = xb + (int64_t)t.o1 * C, *x2 = xb + (int64_t)t.o2 * C;
        const float *x3 = xb + (int64_t)t.o3 * C, *x4 = xb + (int64_t)t.o4 * C;
        float gy = 0.f, gx = 0.f;
        for (int c4 = lane; c4 < C4; c4 += 64) {
            const int c = c4 * 4;
            const int g = c / Cg, cin = c - g * Cg;
            const float4 gc = *reinterpret_cast<const float4 *>(grad_col + ((int64_t)g * npos + pos) * (9 * Cg) + k * Cg + cin);
            const float4 v1 = *reinterpret_cast<const float4 *>(x1 + c), v2 = *reinterpret_cast<const float4 *>(x2 + c);
            const float4 v3 = *reinterpret_cast<const float4 *>(x3 + c), v4 = *reinterpret_cast<const float4 *>(x4 + c);
            // explicit FMAs (the library is built with -ffp-contract=off for the bit-exact geometry path)
#define MMT_DOT4(d1, d2, d3, d4, comp) __builtin_fmaf(d1, v1.comp, __builtin_fmaf(d2, v2.comp, __builtin_fmaf(d3, v3.comp, d4 * v4.comp)))
            gy = __builtin_fmaf(gc.x, MMT_DOT4(t.dy1, t.dy2, t.dy3, t.dy4, x), gy);
            gy = __builtin_fmaf(gc.y, MMT_DOT4(t.dy1, t.dy2, t.dy3, t.dy4, y), gy);
            gy = __builtin_fmaf(gc.z, MMT_DOT4(t.dy1, t.dy2, t.dy3, t.dy4, z), gy);
            gy = __builtin_fmaf(gc.w, MMT_DOT4(t.dy1, t.dy2, t.dy3, t.dy4, w), gy);
            gx = __builtin_fmaf(gc.x, MMT_DOT4(t.dx1, t.dx2, t.dx3, t.dx4, x), gx);
            gx = __builtin_fmaf(gc.y, MMT_DOT4(t.dx1, t.dx2, t.dx3, t.dx4, y), gx);
            gx = __builtin_fmaf(gc.z, MMT_DOT4(t.dx1, t.dx2, t.dx3, t.dx4, z), gx);
            gx = __builtin_fmaf(gc.w, MMT_DOT4(t.dx1, t.dx2, t.dx3, t.dx4, w), gx);
#undef MMT_DOT4
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            gy += __shfl_xor(gy, o);
            gx += __shfl_xor(gx, o);
        }
        if (lane == 0) {
            grad_offset[it * 2] = gy;
            grad_offset[it * 2 + 1] = gx;
        }
    }
}

// LPG lanes (a float4 column each) per (destination pixel, weight group); kU list entries in flight
template <int LPG>
__global__ __launch_bounds__(256) void dcn_col2im_gather(int B, int HW, int C, int groups, const float *grad_col,
                                                         const int32_t *bin_off, const DcnEntry *entries, float *grad_x) {
    constexpr int kU = 4;
    const int Cg = C / groups;
    const int64_t npos = (int64_t)B * HW;
    const int64_t nitems = npos * groups;
    const int grp = threadIdx.x / LPG, li = threadIdx.x - grp * LPG;
    constexpr int kGroups = 256 / LPG;
    for (int64_t item = (int64_t)blockIdx.x * kGroups + grp; item < nitems; item += (int64_t)gridDim.x * kGroups) {
        const int64_t pix = item / groups;             // b * HW + destination pixel
        const int g = (int)(item - pix * groups);
        const int b = (int)(pix / HW);
        const int dest = (int)(pix - (int64_t)b * HW);
        const int beg = bin_off[(int64_t)b * (HW + 1) + dest], end = bin_off[(int64_t)b * (HW + 1) + dest + 1];
        const DcnEntry *eb = entries + (int64_t)b * HW * 36;
        const float *gcb = grad_col + ((int64_t)g * npos + (int64_t)b * HW) * (9 * Cg) + li * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // the list entries of the NEXT kU rows are requested before the current rows are waited for: one dependent round
        // trip per kU rows instead of two
        DcnEntry en[kU], nx[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) en[u] = eb[(beg + u) < end ? (beg + u) : (end > beg ? end - 1 : beg)];
        for (int j = beg; j < end; j += kU) {
            float4 v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u)
                v[u] = *reinterpret_cast<const float4 *>(gcb + ((int64_t)(en[u].src_tap >> 4) * 9 + (en[u].src_tap & 15)) * Cg);
#pragma unroll
            for (int u = 0; u < kU; ++u) nx[u] = eb[(j + kU + u) < end ? (j + kU + u) : (end - 1)];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const float w = (j + u) < end ? en[u].w : 0.f;
                acc = fma4(w, v[u], acc);
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) en[u] = nx[u];
        }
        *reinterpret_cast<float4 *>(grad_x + pix * C + g * Cg + li * 4) = acc;
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

extern "C" int64_t mmt_dcn_col2im_workspace_elems(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t HW = (int64_t)H * W;
    return (int64_t)B * (HW + 1) + 2 * (int64_t)B * HW * 36 + 4;
}

extern "C" int mmt_dcn_col2im_sorted(int B, int H, int W, int C, int groups, const float *x,
                                     const float *offset, const float *grad_col, float *grad_x,
                                     float *grad_offset, int32_t *workspace, int64_t workspace_elems, void *stream) {
    MMT_REQUIRE_PTR(x);
    MMT_REQUIRE_PTR(offset);
    MMT_REQUIRE_PTR(grad_col);
    MMT_REQUIRE_PTR(grad_x);
    MMT_REQUIRE_PTR(grad_offset);
    MMT_REQUIRE_PTR(workspace);
    if (int rc = check_dcn(B, H, W, C, groups, "dcn_col2im_sorted")) return rc;
    if ((((uintptr_t)x | (uintptr_t)grad_col | (uintptr_t)grad_x) & 15) != 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "dcn_col2im_sorted: x / grad_col / grad_x must be 16-byte aligned");
    const int HW = H * W;
    const int lpg = (C / groups) / 4;
    if (HW > kPlanMaxHW || lpg > 64 || (lpg & (lpg - 1)) != 0 || B > 65535 || HW >= (1 << 27))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "dcn_col2im_sorted: needs H*W <= %d and C/groups/4 a power of two <= 64 (H*W=%d, C/groups=%d); "
                                            "use mmt_dcn_col2im", kPlanMaxHW, HW, C / groups);
    if (workspace_elems < mmt_dcn_col2im_workspace_elems(B, H, W))
        return mmt::fail(MMT_ERR_WORKSPACE, "dcn_col2im_sorted: workspace too small (%lld < %lld elements)", (long long)workspace_elems,
                         (long long)mmt_dcn_col2im_workspace_elems(B, H, W));
    hipStream_t st = (hipStream_t)stream;
    int32_t *bin_off = workspace;
    int64_t eoff = (int64_t)B * (HW + 1);
    eoff += eoff & 1;                                      // 8-byte aligned entries
    DcnEntry *entries = reinterpret_cast<DcnEntry *>(workspace + eoff);
    const int64_t npos = (int64_t)B * HW;
    hipLaunchKernelGGL(dcn_plan_kernel, dim3(B), dim3(kPlanThreads), 0, st, H, W, offset, bin_off, entries);
    hipLaunchKernelGGL(dcn_offset_grad_kernel, dim3(mmt::stream_grid(npos * 9, 4, 256 * 64)), dim3(256), 0, st, B, H, W, C, groups, x,
                       offset, grad_col, grad_offset);
    const int64_t nitems = npos * groups;
#define MMT_DCN_GATHER(L)                                                                                                 \
    hipLaunchKernelGGL((dcn_col2im_gather<L>), dim3(mmt::stream_grid(nitems, 256 / L, 256 * 64)), dim3(256), 0, st, B, HW, C, \
                       groups, grad_col, (const int32_t *)bin_off, (const DcnEntry *)entries, grad_x)
    switch (lpg) {
        case 1: MMT_DCN_GATHER(1); break;
        case 2: MMT_DCN_GATHER(2); break;
        case 4: MMT_DCN_GATHER(4); break;
        case 8: MMT_DCN_GATHER(8); break;
        case 16: MMT_DCN_GATHER(16); break;
        case 32: MMT_DCN_GATHER(32); break;
        default: MMT_DCN_GATHER(64); break;
    }
#undef MMT_DCN_GATHER
    return mmt::check_launch("dcn_col2im_sorted");
}
