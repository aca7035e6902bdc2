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
#include "dcn_tap.h"

namespace {

using mmt_dcn::Tap;
using mmt_dcn::make_tap;

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
//   dcn_plan_kernel        one workgroup per image: every (position, tap) has up to 4 bilinear corners inside the image;
//                          they are binned by destination pixel in LDS (count -> scan -> counting sort) and leave as a
//                          per-pixel list of (source position, corner, tap, weight) entries.  Corners whose WEIGHT is zero
//                          stay in the list when their coordinate derivative is not (at integer sampling positions -- zero
//                          offsets, how training starts -- three of the four corners carry the offset gradient);
//   dcn_col2im_gather      one lane group (Cg/4 lanes, a float4 column each) per (destination pixel, weight group) walks the
//                          pixel's list and reads each grad_col row ONCE for both results:
//                            grad_x[pixel, group]  = sum of  weight * grad_col[source, tap, group]  (registers, one plain
//                                                    store per output row: grad_x is overwritten, no zero-fill needed);
//                            dot[source, tap, corner, group] = < grad_col[source, tap, group], x[pixel, group] >  -- the
//                                                    channel sums the offset gradient is made of (the pixel IS that corner);
//   dcn_offset_reduce      grad_offset[pos, tap] = sum over the 4 corners of d(weight)/d(py, px) * (sum over groups of dot).
// Round 2 computed the offset gradient in a kernel of its own that streamed grad_col a second time and pulled the four x
// rows of every (position, tap) through L1 (1.25 GB): 107 of the 257 us.
// The order of a pixel's list is the LDS arrival order, so the fp32 sum order may differ between runs (as with the
// atomics before).
constexpr int kDcnMaxGroups = 8;     // weight groups the workspace of the sorted backward is sized for
constexpr int kPlanThreads = 1024;
constexpr int kPlanMaxHW = 4096;     // destination bins held in LDS

struct DcnEntry { int src_tap; float w; };    // (source position inside the image << 6) | (corner << 4) | tap

// corner i of a tap takes part in the backward: inside the image, and a non-zero weight or a non-zero derivative
__device__ __forceinline__ bool tap_corner_live(const Tap &t, int i) {
    const float w = i == 0 ? t.w1 : i == 1 ? t.w2 : i == 2 ? t.w3 : t.w4;
    const float dy = i == 0 ? t.dy1 : i == 1 ? t.dy2 : i == 2 ? t.dy3 : t.dy4;
    const float dx = i == 0 ? t.dx1 : i == 1 ? t.dx2 : i == 2 ? t.dx3 : t.dx4;
    return w != 0.f || dy != 0.f || dx != 0.f;
}

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
        if (tap_corner_live(t, 0)) atomicAdd(&cnt[t.o1], 1);
        if (tap_corner_live(t, 1)) atomicAdd(&cnt[t.o2], 1);
        if (tap_corner_live(t, 2)) atomicAdd(&cnt[t.o3], 1);
        if (tap_corner_live(t, 3)) atomicAdd(&cnt[t.o4], 1);
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
        const int st = (pos << 6) | k;
        if (tap_corner_live(t, 0)) eb[off[t.o1] + atomicAdd(&cnt[t.o1], 1)] = DcnEntry{st, t.w1};
        if (tap_corner_live(t, 1)) eb[off[t.o2] + atomicAdd(&cnt[t.o2], 1)] = DcnEntry{st | (1 << 4), t.w2};
        if (tap_corner_live(t, 2)) eb[off[t.o3] + atomicAdd(&cnt[t.o3], 1)] = DcnEntry{st | (2 << 4), t.w3};
        if (tap_corner_live(t, 3)) eb[off[t.o4] + atomicAdd(&cnt[t.o4], 1)] = DcnEntry{st | (3 << 4), t.w4};
    }
}

// LPG lanes (a float4 column each) per (destination pixel, weight group); kU = 4 list entries in flight.
// dots [B*HW*9*4][groups]: < grad_col row, x row of the destination pixel > per (source position, tap, corner, group) -- the
// four entries of a trip are reduced together: after two exchange steps lane (l & 3) of every quad holds the quad's sum for
// entry l & 3, three more steps fold the quads of the lane group (LPG >= 4; smaller groups reduce entry by entry).
// Placement: the up-to-four destination pixels of a (source, tap) are a 2 x 2 block of neighbours, and each of them reads the
// same grad_col row.  All workgroups of one (image, 8-row band) run on ONE XCD (blockIdx & 7 selects the unit modulo 8), and
// inside a band the items are ordered (row pair, x, row in pair, group), so the vertical neighbours sit in the same workgroup and the
// horizontal ones in the next: the second to fourth read of a row is an L1 / L2 hit instead of a trip to another XCD's
// share of the Infinity Cache (HBM-side traffic of this kernel: 820 MB for 311 MB of rows before, PMC).
template <int LPG>
__global__ __launch_bounds__(256) void dcn_col2im_gather(int B, int H, int W, int C, int groups, int blocks_per_unit, const float *x,
                                                         const float *grad_col, const int32_t *bin_off, const DcnEntry *entries,
                                                         float *grad_x, float *dots) {
    constexpr int kU = 4;
    const int Cg = C / groups, HW = H * W;
    const int64_t npos = (int64_t)B * HW;
    const int grp = threadIdx.x / LPG, li = threadIdx.x - grp * LPG;
    constexpr int kGroups = 256 / LPG;
    {
        // unit = (image, band of kBand rows); the units are dealt to the XCDs round robin, all workgroups of a unit to one XCD
        constexpr int kBand = 8;
        const int bands = (H + kBand - 1) / kBand;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int unit = (idx / blocks_per_unit) * 8 + xcd;
        if (unit >= B * bands) return;                        // whole workgroup (padding of the unit -> XCD deal)
        const int b = unit / bands, band = unit - b * bands;
        const int k = idx - (idx / blocks_per_unit) * blocks_per_unit;
        const int64_t item = (int64_t)k * kGroups + grp;     // inside the band: ((y / 2 * W + x) * 2 + (y & 1)) * groups + g
        const int g = (int)(item % groups);
        const int64_t pp = item / groups;
        const int ysub = (int)(pp & 1);
        const int64_t q = pp >> 1;
        const int xx = (int)(q % W), yb = (int)(q / W) * 2 + ysub;
        const int y = yb < kBand ? band * kBand + yb : H;     // past the band: idle
        const bool on = y < H;
        const int dest = on ? y * W + xx : 0;
        const int64_t pix = (int64_t)b * HW + dest;           // b * HW + destination pixel
        const int beg = on ? bin_off[(int64_t)b * (HW + 1) + dest] : 0, end = on ? bin_off[(int64_t)b * (HW + 1) + dest + 1] : 0;
        const DcnEntry *eb = entries + (int64_t)b * HW * 36;
        const float *gcb = grad_col + ((int64_t)g * npos + (int64_t)b * HW) * (9 * Cg) + li * 4;
        const float4 xd = *reinterpret_cast<const float4 *>(x + pix * C + g * Cg + li * 4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // the longest list among the lane groups of the wave decides the trip count (shorter ones run on clamped entries)
        int trips = (end - beg + kU - 1) / kU;
#pragma unroll
        for (int m = LPG; m < 64; m <<= 1) { const int o = __shfl_xor(trips, m); trips = o > trips ? o : trips; }
        // the list entries of the NEXT kU rows are requested before the current rows are waited for: one dependent round
        // trip per kU rows instead of two
        DcnEntry en[kU], nx[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) en[u] = eb[(beg + u) < end ? (beg + u) : (end > beg ? end - 1 : beg)];
        for (int tr = 0; tr < trips; ++tr) {
            const int j = beg + tr * kU;
            float4 v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                // (a lane group whose list is shorter than the wave's trip count -- or empty -- runs on clamped or, past the
                // image's last entry, unwritten entries: whatever they hold, the row address is forced inside grad_col)
                const int src = min((unsigned)en[u].src_tap >> 6, (unsigned)(HW - 1)), tap = min(en[u].src_tap & 15, 8);
                v[u] = *reinterpret_cast<const float4 *>(gcb + ((int64_t)src * 9 + tap) * Cg);
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int jn = j + kU + u;
                nx[u] = eb[jn < end ? jn : (end > beg ? end - 1 : beg)];
            }
            float d[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const bool live = (j + u) < end;
                acc = fma4(live ? en[u].w : 0.f, v[u], acc);
                d[u] = live ? __builtin_fmaf(v[u].x, xd.x, __builtin_fmaf(v[u].y, xd.y, __builtin_fmaf(v[u].z, xd.z, v[u].w * xd.w))) : 0.f;
            }
            if constexpr (LPG >= 4) {
                // 4 values per lane -> 1: lane l keeps entry (l & 3)
                float a0 = (li & 1) ? d[1] : d[0], b0 = (li & 1) ? d[0] : d[1];
                float a1 = (li & 1) ? d[3] : d[2], b1 = (li & 1) ? d[2] : d[3];
                a0 += __shfl_xor(b0, 1);
                a1 += __shfl_xor(b1, 1);
                float e0 = (li & 2) ? a1 : a0, f0 = (li & 2) ? a0 : a1;
                e0 += __shfl_xor(f0, 2);
#pragma unroll
                for (int m = 4; m < LPG; m <<= 1) e0 += __shfl_xor(e0, m);
                const int u = li & 3;
                if (li < 4 && (j + u) < end) {
                    const int st = u == 0 ? en[0].src_tap : u == 1 ? en[1].src_tap : u == 2 ? en[2].src_tap : en[3].src_tap;
                    dots[((((int64_t)b * HW + (st >> 6)) * 9 + (st & 15)) * 4 + ((st >> 4) & 3)) * groups + g] = e0;
                }
            } else {
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    float e0 = d[u];
#pragma unroll
                    for (int m = 1; m < LPG; m <<= 1) e0 += __shfl_xor(e0, m);
                    if (li == 0 && (j + u) < end)
                        dots[((((int64_t)b * HW + (en[u].src_tap >> 6)) * 9 + (en[u].src_tap & 15)) * 4 + ((en[u].src_tap >> 4) & 3)) * groups + g] = e0;
                }
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) en[u] = nx[u];
        }
        if (on) *reinterpret_cast<float4 *>(grad_x + pix * C + g * Cg + li * 4) = acc;
    }
}

// grad_offset[pos, tap] = sum over the live corners of (d weight / d py, d weight / d px) * (sum over the groups of dots)
__global__ __launch_bounds__(256) void dcn_offset_reduce(int B, int H, int W, int groups, const float *offset, const float *dots,
                                                         float *grad_offset) {
    const int64_t nitems = (int64_t)B * H * W * 9;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < nitems; it += (int64_t)gridDim.x * 256) {
        const int64_t pos = it / 9;
        const int k = (int)(it - pos * 9);
        const int hw = (int)(pos % (H * W));
        const int h = hw / W, w = hw - h * W, ky = k / 3, kx = k - ky * 3;
        const Tap t = make_tap((float)(h + ky - 1) + offset[it * 2], (float)(w + kx - 1) + offset[it * 2 + 1], H, W);
        float gy = 0.f, gx = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (tap_corner_live(t, c)) {
                float sum = 0.f;
                for (int g = 0; g < groups; ++g) sum += dots[(it * 4 + c) * groups + g];
                const float dy = c == 0 ? t.dy1 : c == 1 ? t.dy2 : c == 2 ? t.dy3 : t.dy4;
                const float dx = c == 0 ? t.dx1 : c == 1 ? t.dx2 : c == 2 ? t.dx3 : t.dx4;
                gy = __builtin_fmaf(dy, sum, gy);
                gx = __builtin_fmaf(dx, sum, gx);
            }
        }
        grad_offset[it * 2] = gy;
        grad_offset[it * 2 + 1] = gx;
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
    // bin offsets + 8-byte entries (<= 36 per position) + the dot products [B*HW*36][groups <= kDcnMaxGroups]
    return (int64_t)B * (HW + 1) + 2 * (int64_t)B * HW * 36 + (int64_t)B * HW * 36 * kDcnMaxGroups + 8;
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
    if (HW > kPlanMaxHW || lpg > 64 || (lpg & (lpg - 1)) != 0 || B > 65535 || HW >= (1 << 25) || groups > kDcnMaxGroups)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "dcn_col2im_sorted: needs H*W <= %d, C/groups/4 a power of two <= 64 and groups <= %d (H*W=%d, C/groups=%d, "
                                            "groups=%d); use mmt_dcn_col2im", kPlanMaxHW, kDcnMaxGroups, HW, C / groups, groups);
    if (workspace_elems < mmt_dcn_col2im_workspace_elems(B, H, W))
        return mmt::fail(MMT_ERR_WORKSPACE, "dcn_col2im_sorted: workspace too small (%lld < %lld elements)", (long long)workspace_elems,
                         (long long)mmt_dcn_col2im_workspace_elems(B, H, W));
    hipStream_t st = (hipStream_t)stream;
    int32_t *bin_off = workspace;
    int64_t eoff = (int64_t)B * (HW + 1);
    eoff += eoff & 1;                                      // 8-byte aligned entries
    DcnEntry *entries = reinterpret_cast<DcnEntry *>(workspace + eoff);
    float *dots = reinterpret_cast<float *>(workspace + eoff + 2 * (int64_t)B * HW * 36);
    const int64_t npos = (int64_t)B * HW;
    hipLaunchKernelGGL(dcn_plan_kernel, dim3(B), dim3(kPlanThreads), 0, st, H, W, offset, bin_off, entries);
    // items of one (image, 8-row band) unit: 4 row pairs x columns x 2 x groups
    const int64_t items_per_unit = (int64_t)4 * W * 2 * groups;
    const int units = B * ((H + 7) / 8);
#define MMT_DCN_GATHER(L)                                                                                                 \
    {                                                                                                                     \
        const int bpi = (int)mmt::ceil_div(items_per_unit, 256 / L);                                                      \
        hipLaunchKernelGGL((dcn_col2im_gather<L>), dim3((unsigned)(8 * ((units + 7) / 8) * bpi)), dim3(256), 0, st, B, H, W, C, \
                           groups, bpi, x, grad_col, (const int32_t *)bin_off, (const DcnEntry *)entries, grad_x, dots);   \
    }
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
    hipLaunchKernelGGL(dcn_offset_reduce, dim3(mmt::stream_grid(npos * 9, 256, 256 * 16)), dim3(256), 0, st, B, H, W, groups, offset,
                       (const float *)dots, grad_offset);
    return mmt::check_launch("dcn_col2im_sorted");
}
