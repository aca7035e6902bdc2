// Training-mode BatchNorm2d fused with the residual add and the ReLU that follow it in the dense
// nets (ResNet-50 / SECONDFPN / DepthNet / BEV trunk of layers/backbones/lss_fpn.py and
// layers/heads/bev_depth_head.py -- mmcv/mmdet modules in the reference), for channels-last fp32
// activations [R = N*H*W rows, C channels].
//
// These layers are pure HBM streaming (0.25 flop/byte).  Unfused (MIOpen BN + ATen add / relu) one
// conv-bn-relu costs 5 passes over the activation forward and 8 backward; fused:
//   forward : bn_stats (1 read)  ->  bn_finalize (C values)  ->  bn_apply (1 read [+ residual] , 1 write)
//   backward: bn_bwd_reduce (dy, x [, y -> dres])  ->  bn_bwd_finalize  ->  bn_bwd_dx (dy | dres, x -> dx)
// i.e. 3 + 5 passes.  The ReLU mask is recomputed from x (plain variant) or read from the saved
// output y (residual variant); nothing but mean / rstd is saved beyond what autograd keeps anyway.
//
// Layout: one float4 (4 channels) per lane, TPR = C/4 lanes per row (or 256 lanes looping over
// C/1024 column blocks when C > 1024), 256/TPR rows per workgroup and trip; every thread keeps
// its channels for the whole kernel, so per-channel statistics accumulate in registers and meet
// in LDS once per workgroup and leave as one partial row per workgroup, summed (in double) by the
// finalize kernel.  Loads in the row loops are
// unconditional (clamped row index, zero weight) so several rows stay in flight per lane.
// Activation type AT = float or bf16_t (the *_ex entry points; bf16 = the autocast region of BASELINE configs[4]): x, residual, y
// and their gradients are stored in AT (4 channels per lane = 16 or 8 bytes), the statistics, the affine coefficients and all
// arithmetic stay fp32 -- what autocast does for batch_norm -- so the bf16 nets no longer go through MIOpen's NHWC batch norm.
#include "mmt_common.h"

namespace {

constexpr int kBlock = 256;

struct BnGeom {
    int C4;        // float4 columns per row
    int tpr;       // threads per row (<= 256)
    int kc;        // column blocks of tpr lanes per row (grid.y of the reduce kernels; C4 = tpr * kc)
    int rpi;       // rows per workgroup and trip
};

struct BnArgs {
    int64_t R;
    int C;
    BnGeom g;
    int relu, has_res;
    float momentum, eps;
    const void *x, *res, *y_in, *dy;   // activations, AT
    int64_t dy_pitch4;                 // rows of dy are this many quadruples apart (C4: dense; more: dy is a channel slice of a wider channels-last tensor,
                                       // e.g. the gradient of one input of a torch.cat -- read in place instead of through a copy)
    const void *dy3;                   // backward, nullable (only with dy2): a third one
    const void *dy2;                   // backward, nullable: a second gradient of y (the output was handed out twice: conv path + identity of the next block), added on load
    const float *weight, *bias;
    float *running_mean, *running_var;
    float *acc;            // [blocks, 2*C] per-workgroup partial sums (scratch)
    float *save_mean, *save_rstd;
    float *scale, *shift;  // [C] each (workspace): y = x * scale + shift
    void *y, *dx, *dres;               // activations, AT
    float *dweight, *dbias;
    float *coef;           // [2*C] backward: mean(dy'), mean(dy' * xhat)
};

constexpr int kMaxKC = 8;  // C <= 8192

// the four channels of float4 column `i` of an activation tensor
template <typename AT>
__device__ __forceinline__ float4 ld4(const void *p, int64_t i) {
    if constexpr (sizeof(AT) == 4) {
        return reinterpret_cast<const float4 *>(p)[i];
    } else {
        const uint2 t = reinterpret_cast<const uint2 *>(p)[i];
        return make_float4(bf16_lo(t.x), bf16_hi(t.x), bf16_lo(t.y), bf16_hi(t.y));
    }
}
// the same in two steps: the raw 16 / 8 bytes stay in flight (a bf16 quadruple is two registers, not four), converted when consumed
template <typename AT> struct Raw4 { typedef float4 type; };
template <> struct Raw4<bf16_t> { typedef uint2 type; };
template <typename AT>
__device__ __forceinline__ typename Raw4<AT>::type ldraw(const void *p, int64_t i) {
    return reinterpret_cast<const typename Raw4<AT>::type *>(p)[i];
}
__device__ __forceinline__ float4 cvt4(const float4 &v) { return v; }
__device__ __forceinline__ float4 cvt4(const uint2 &t) { return make_float4(bf16_lo(t.x), bf16_hi(t.x), bf16_lo(t.y), bf16_hi(t.y)); }
// rows (reduce) / elements (map) a lane keeps in flight: the same BYTES for either activation type (four 16-byte or eight 8-byte
// loads per tensor).  Measured flat against four for bf16 (tools/kbench_bn.py --bf16: 3.0-3.5 TB/s either way at the configs[4] shapes);
// kept because the raw form costs the bf16 kernels no more registers than the fp32 ones use.
template <typename AT> struct InFlight { static constexpr int value = sizeof(AT) == 2 ? 8 : 4; };

template <typename AT>
__device__ __forceinline__ void st4(void *p, int64_t i, float4 v) {
    if constexpr (sizeof(AT) == 4) reinterpret_cast<float4 *>(p)[i] = v;
    else reinterpret_cast<uint2 *>(p)[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
}

// Sum of (a, b) per channel over this workgroup's rows -> partial[blockIdx][0:C], [C:2C] (plain stores:
// 2*C same-address atomics per workgroup cost 150 us per launch, measured).
// MODE 0: a = x, b = x*x.   MODE 1: a = dy', b = dy' * xhat  (dy' = relu-masked dy).
template <int MODE, typename AT, int NG>     // NG: gradients of y to add up (1; 2 or 3 for a forked output)
__global__ __launch_bounds__(kBlock) void bn_reduce_kernel(BnArgs a, int64_t rows_per_block) {
    __shared__ float4 red[2][kBlock];
    const BnGeom g = a.g;
    const int tid = threadIdx.x;
    const int rsub = tid / g.tpr;
    const int col = tid - rsub * g.tpr + blockIdx.y * g.tpr;      // grid.y = column block (C > 1024)
    const bool live = rsub < g.rpi;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = (r_begin + rows_per_block) < a.R ? (r_begin + rows_per_block) : a.R;
    float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
    float4 sc = sa, sh = sa, mean = sa, rstd = sa;
    if (MODE == 1) {
        sc = *reinterpret_cast<const float4 *>(a.scale + col * 4);
        sh = *reinterpret_cast<const float4 *>(a.shift + col * 4);
        mean = *reinterpret_cast<const float4 *>(a.save_mean + col * 4);
        rstd = *reinterpret_cast<const float4 *>(a.save_rstd + col * 4);
    }
    if (live && r_begin < r_end) {
        const int64_t last = r_end - 1;
        constexpr int kRows = InFlight<AT>::value;
        typedef typename Raw4<AT>::type RawT;
        for (int64_t r0 = r_begin + rsub; r0 < r_end; r0 += (int64_t)kRows * g.rpi) {
            RawT rx[kRows], rd[kRows], rd2[NG > 1 ? kRows : 1], rd3[NG > 2 ? kRows : 1], ry[kRows];
            float wgt[kRows];
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const int64_t r = r0 + (int64_t)u * g.rpi;
                wgt[u] = r < r_end ? 1.f : 0.f;
                const int64_t off = (r < r_end ? r : last) * g.C4 + col;
                rx[u] = ldraw<AT>(a.x, off);
                if (MODE == 1) {
                    rd[u] = ldraw<AT>(a.dy, (r < r_end ? r : last) * a.dy_pitch4 + col);
                    if (NG > 1) rd2[NG > 1 ? u : 0] = ldraw<AT>(a.dy2, off);
                    if (NG > 2) rd3[NG > 2 ? u : 0] = ldraw<AT>(a.dy3, off);
                    if (a.has_res) ry[u] = ldraw<AT>(a.y_in, off);
                }
            }
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const float4 cx = cvt4(rx[u]);
                if (MODE == 0) {
                    const float w = wgt[u];
                    sa.x += w * cx.x; sa.y += w * cx.y; sa.z += w * cx.z; sa.w += w * cx.w;
                    sb.x += w * cx.x * cx.x; sb.y += w * cx.y * cx.y;
                    sb.z += w * cx.z * cx.z; sb.w += w * cx.w * cx.w;
                } else {
                    float4 d = cvt4(rd[u]);
                    if (NG > 1) { const float4 t = cvt4(rd2[NG > 1 ? u : 0]); d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w; }
                    if (NG > 2) { const float4 t = cvt4(rd3[NG > 2 ? u : 0]); d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w; }
                    d.x *= wgt[u]; d.y *= wgt[u]; d.z *= wgt[u]; d.w *= wgt[u];
                    if (a.relu) {
                        if (a.has_res) {
                            const float4 cy = cvt4(ry[u]);
                            if (!(cy.x > 0.f)) d.x = 0.f;
                            if (!(cy.y > 0.f)) d.y = 0.f;
                            if (!(cy.z > 0.f)) d.z = 0.f;
                            if (!(cy.w > 0.f)) d.w = 0.f;
                        } else {
                            if (!(cx.x * sc.x + sh.x > 0.f)) d.x = 0.f;
                            if (!(cx.y * sc.y + sh.y > 0.f)) d.y = 0.f;
                            if (!(cx.z * sc.z + sh.z > 0.f)) d.z = 0.f;
                            if (!(cx.w * sc.w + sh.w > 0.f)) d.w = 0.f;
                        }
                    }
                    if (a.dres && wgt[u] != 0.f) {      // residual variant: the masked gradient IS grad_residual
                        const int64_t r = r0 + (int64_t)u * g.rpi;
                        st4<AT>(a.dres, r * g.C4 + col, d);
                    }
                    sa.x += d.x; sa.y += d.y; sa.z += d.z; sa.w += d.w;
                    sb.x += d.x * ((cx.x - mean.x) * rstd.x);
                    sb.y += d.y * ((cx.y - mean.y) * rstd.y);
                    sb.z += d.z * ((cx.z - mean.z) * rstd.z);
                    sb.w += d.w * ((cx.w - mean.w) * rstd.w);
                }
            }
        }
    }
    // rows of the workgroup meet in LDS, then one partial row (this column block of it) per workgroup
    red[0][tid] = sa;
    red[1][tid] = sb;
    __syncthreads();
    if (rsub == 0) {
        float4 ta = red[0][tid], tb = red[1][tid];
        for (int j = 1; j < g.rpi; ++j) {
            const float4 pa = red[0][tid + j * g.tpr], pb = red[1][tid + j * g.tpr];
            ta.x += pa.x; ta.y += pa.y; ta.z += pa.z; ta.w += pa.w;
            tb.x += pb.x; tb.y += pb.y; tb.z += pb.z; tb.w += pb.w;
        }
        float *p = a.acc + (int64_t)blockIdx.x * 2 * a.C;          // this workgroup's partial sums
        *reinterpret_cast<float4 *>(p + col * 4) = ta;
        *reinterpret_cast<float4 *>(p + a.C + col * 4) = tb;
    }
}

// Sum of the workgroups' partial rows, 16 channels per workgroup: 16 groups of 16 lanes take every 16th partial row each (64-byte
// pieces of rows that sit in L2, 8 independent loads in flight), double accumulation, LDS combine.  Returns the totals to the first
// 16 threads.  256 threads, not 1024 (as up to round 4): the kernel sits between two streaming passes of the main stream while
// the weight-gradient stream keeps the CUs full of long-running workgroups, and a workgroup that needs 16 free wave slots on ONE
// CU waited for them -- 75 us per launch behind DepthNet's 512-channel weight gradients against 5 us alone (tools/glue_audit.py).
constexpr int kFinBlock = 256;
constexpr int kFinCh = 16;
__device__ __forceinline__ bool bn_block_sums(const BnArgs &a, int nblocks, int *c_out, double *s0, double *s1) {
    __shared__ double red[2][kFinBlock];
    constexpr int NSEG = kFinBlock / kFinCh;
    const int lane = threadIdx.x & (kFinCh - 1), seg = threadIdx.x / kFinCh;
    const int c = blockIdx.x * kFinCh + lane;
    double t0 = 0.0, t1 = 0.0;
    if (c < a.C) {
        const int last = nblocks - 1;
        for (int b0 = seg; b0 < nblocks; b0 += 8 * NSEG) {
            float v0[8], v1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + u * NSEG;
                const float *p = a.acc + (int64_t)(b < nblocks ? b : last) * 2 * a.C;
                v0[u] = p[c];
                v1[u] = p[a.C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (b0 + u * NSEG < nblocks) { t0 += (double)v0[u]; t1 += (double)v1[u]; }
        }
    }
    red[0][threadIdx.x] = t0;
    red[1][threadIdx.x] = t1;
    __syncthreads();
    *c_out = c;
    if (seg != 0 || c >= a.C) return false;
    for (int j = 1; j < NSEG; ++j) { t0 += red[0][lane + kFinCh * j]; t1 += red[1][lane + kFinCh * j]; }
    *s0 = t0; *s1 = t1;
    return true;
}

// forward statistics -> mean / rstd / affine coefficients / running statistics (torch semantics:
// biased variance for the normalisation, unbiased for running_var)
__global__ __launch_bounds__(kFinBlock) void bn_finalize_kernel(BnArgs a, int nblocks) {
    int c;
    double sum, sumsq;
    if (!bn_block_sums(a, nblocks, &c, &sum, &sumsq)) return;
    const double n = (double)a.R;
    const double mean = sum / n;
    double var = sumsq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float w = a.weight ? a.weight[c] : 1.f, b = a.bias ? a.bias[c] : 0.f;
    a.save_mean[c] = (float)mean;
    a.save_rstd[c] = rstd;
    a.scale[c] = w * rstd;
    a.shift[c] = b - (float)mean * (w * rstd);
    if (a.running_mean) {
        a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
    }
}

__global__ __launch_bounds__(kFinBlock) void bn_bwd_finalize_kernel(BnArgs a, int nblocks) {
    int c;
    double sdy, sdyx;
    if (!bn_block_sums(a, nblocks, &c, &sdy, &sdyx)) return;
    if (a.dbias) a.dbias[c] = (float)sdy;
    if (a.dweight) a.dweight[c] = (float)sdyx;
    a.coef[c] = (float)(sdy / (double)a.R);
    a.coef[a.C + c] = (float)(sdyx / (double)a.R);
}

// MODE 0: y = relu(x * scale + shift [+ res]).   MODE 1: dx = scale * (dy' - c1 - xhat * c2) [, dres = dy'].
template <int MODE, typename AT, int NG>
__global__ __launch_bounds__(kBlock) void bn_map_kernel(BnArgs a) {
    const int C4 = a.g.C4;
    const int64_t total = a.R * C4;
    constexpr int U = InFlight<AT>::value;
    typedef typename Raw4<AT>::type RawT;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < total; i0 += U * stride) {
        RawT rx[U], rr[U], rd[U], rd2[NG > 1 ? U : 1], rd3[NG > 2 ? U : 1];
        int64_t idx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            idx[u] = i < total ? i : total - 1;              // clamped: unconditional loads
            rx[u] = ldraw<AT>(a.x, idx[u]);
            if (MODE == 0) {
                if (a.has_res) rr[u] = ldraw<AT>(a.res, idx[u]);
            } else {
                rd[u] = ldraw<AT>(a.dy, a.dy_pitch4 == C4 ? idx[u] : (idx[u] / C4) * a.dy_pitch4 + idx[u] % C4);
                if (NG > 1) rd2[NG > 1 ? u : 0] = ldraw<AT>(a.dy2, idx[u]);
                if (NG > 2) rd3[NG > 2 ? u : 0] = ldraw<AT>(a.dy3, idx[u]);
                if (a.has_res) rr[u] = ldraw<AT>(a.y_in, idx[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= total) break;
            const float4 cx = cvt4(rx[u]);
            const int c = (total < (1ll << 31) ? (int)((unsigned)i % (unsigned)C4) : (int)(i % C4)) * 4;
            const float4 sc = *reinterpret_cast<const float4 *>(a.scale + c);
            const float4 sh = *reinterpret_cast<const float4 *>(a.shift + c);
            if (MODE == 0) {
                float4 y = make_float4(cx.x * sc.x + sh.x, cx.y * sc.y + sh.y, cx.z * sc.z + sh.z, cx.w * sc.w + sh.w);
                if (a.has_res) { const float4 cr = cvt4(rr[u]); y.x += cr.x; y.y += cr.y; y.z += cr.z; y.w += cr.w; }
                if (a.relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
                st4<AT>(a.y, i, y);
            } else {
                const float4 mean = *reinterpret_cast<const float4 *>(a.save_mean + c);
                const float4 rstd = *reinterpret_cast<const float4 *>(a.save_rstd + c);
                const float4 c1 = *reinterpret_cast<const float4 *>(a.coef + c);
                const float4 c2 = *reinterpret_cast<const float4 *>(a.coef + a.C + c);
                float4 d = cvt4(rd[u]);
                if (NG > 1) { const float4 t = cvt4(rd2[NG > 1 ? u : 0]); d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w; }
                if (NG > 2) { const float4 t = cvt4(rd3[NG > 2 ? u : 0]); d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w; }
                if (a.relu) {
                    if (a.has_res) {
                        const float4 cr = cvt4(rr[u]);
                        if (!(cr.x > 0.f)) d.x = 0.f;
                        if (!(cr.y > 0.f)) d.y = 0.f;
                        if (!(cr.z > 0.f)) d.z = 0.f;
                        if (!(cr.w > 0.f)) d.w = 0.f;
                    } else {
                        if (!(cx.x * sc.x + sh.x > 0.f)) d.x = 0.f;
                        if (!(cx.y * sc.y + sh.y > 0.f)) d.y = 0.f;
                        if (!(cx.z * sc.z + sh.z > 0.f)) d.z = 0.f;
                        if (!(cx.w * sc.w + sh.w > 0.f)) d.w = 0.f;
                    }
                }
                if (a.dres) st4<AT>(a.dres, i, d);
                float4 o;
                o.x = sc.x * (d.x - c1.x - (cx.x - mean.x) * rstd.x * c2.x);
                o.y = sc.y * (d.y - c1.y - (cx.y - mean.y) * rstd.y * c2.y);
                o.z = sc.z * (d.z - c1.z - (cx.z - mean.z) * rstd.z * c2.z);
                o.w = sc.w * (d.w - c1.w - (cx.w - mean.w) * rstd.w * c2.w);
                st4<AT>(a.dx, i, o);
            }
        }
    }
}

int geometry(const char *who, int64_t R, int C, BnGeom *g) {
    if (R <= 0 || C <= 0 || C % 4) return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: need R > 0 and C %% 4 == 0 (R=%lld C=%d)", who, (long long)R, C);
    const int C4 = C / 4;
    g->C4 = C4;
    // wider rows: column blocks of 256, 128 or 64 lanes (C = 1536 = 3 x 128 lanes: the task heads' 24 x 64 channels normalised in one go)
    g->tpr = 0;
    if (C4 <= kBlock) { g->tpr = C4; g->kc = 1; }
    else
        for (int t = kBlock; t >= 64 && !g->tpr; t /= 2)
            if (C4 % t == 0 && C4 / t <= kMaxKC) { g->tpr = t; g->kc = C4 / t; }
    if (!g->tpr) return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: C=%d is not supported (C <= 1024, or a multiple of 256 with at most %d column blocks of 1024 / 512 / 256 channels)", who, C, kMaxKC);
    g->rpi = kBlock / g->tpr;
    if (R * C4 >= (1ll << 40)) return mmt::fail(MMT_ERR_TOO_LARGE, "%s: activation too large", who);
    return 0;
}

constexpr int kMaxPartialBlocks = 512;

void reduce_grid(const BnGeom &g, int64_t R, int rows_in_flight, int *blocks, int64_t *rows_per_block) {
    // every workgroup owns whole trips of rows and at least 64 rows, so the partial rows it writes
    // (2*C floats) stay below ~6 % of what it reads; at most kMaxPartialBlocks workgroups
    const int64_t trip = (int64_t)g.rpi * rows_in_flight;
    int64_t nb = R / 64;
    // ... unless that leaves most of the chip idle (ResNet-50's last stage at BASELINE configs[3]: R = 4 224 rows of 2 048 channels = 66 x 2
    // workgroups, 18.6 us for 35-69 MB): then down to 16 rows per workgroup, up to 512 workgroups over the column blocks
    const int64_t fill = 512 / g.kc, fine = R / 16;
    if (nb < fill) nb = fine < fill ? (fine > nb ? fine : nb) : fill;
    if (nb < 1) nb = 1;
    if (nb > kMaxPartialBlocks) nb = kMaxPartialBlocks;
    const int64_t rpb = mmt::ceil_div(mmt::ceil_div(R, nb), trip) * trip;
    *rows_per_block = rpb;
    *blocks = (int)mmt::ceil_div(R, rpb);
}

}  // namespace

extern "C" int64_t mmt_bn_workspace_elems(int C) { return C > 0 ? (2ll * kMaxPartialBlocks + 2) * C : -1; }

namespace {

template <typename AT>
void launch_forward(const BnArgs &a, int blocks, int64_t rpb, hipStream_t st, int *rc) {
    hipLaunchKernelGGL((bn_reduce_kernel<0, AT, 1>), dim3(blocks, a.g.kc), dim3(kBlock), 0, st, a, rpb);
    if ((*rc = mmt::check_launch("bn_relu_forward(stats)"))) return;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((int)mmt::ceil_div(a.C, kFinCh)), dim3(kFinBlock), 0, st, a, blocks);
    if ((*rc = mmt::check_launch("bn_relu_forward(finalize)"))) return;
    hipLaunchKernelGGL((bn_map_kernel<0, AT, 1>), dim3(mmt::stream_grid(mmt::ceil_div(a.R * a.g.C4, InFlight<AT>::value), kBlock)), dim3(kBlock), 0, st, a);
    *rc = mmt::check_launch("bn_relu_forward(apply)");
}

template <typename AT, int NG>
void launch_backward_ng(BnArgs a, int blocks, int64_t rpb, hipStream_t st, int *rc) {
    hipLaunchKernelGGL((bn_reduce_kernel<1, AT, NG>), dim3(blocks, a.g.kc), dim3(kBlock), 0, st, a, rpb);
    if ((*rc = mmt::check_launch("bn_relu_backward(reduce)"))) return;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((int)mmt::ceil_div(a.C, kFinCh)), dim3(kFinBlock), 0, st, a, blocks);
    if ((*rc = mmt::check_launch("bn_relu_backward(finalize)"))) return;
    const dim3 grid(mmt::stream_grid(mmt::ceil_div(a.R * a.g.C4, InFlight<AT>::value), kBlock));
    if (a.dres) {
        // the reduce pass has written grad_residual = masked grad_y (the sum of its gradients): the dx pass reads that instead of
        // grad_y + y (7 passes instead of 8 for the residual variant)
        a.dy = a.dres; a.dy_pitch4 = a.g.C4; a.dy2 = nullptr; a.dy3 = nullptr; a.dres = nullptr; a.relu = 0; a.has_res = 0;
        hipLaunchKernelGGL((bn_map_kernel<1, AT, 1>), grid, dim3(kBlock), 0, st, a);
    } else {
        hipLaunchKernelGGL((bn_map_kernel<1, AT, NG>), grid, dim3(kBlock), 0, st, a);
    }
    *rc = mmt::check_launch("bn_relu_backward(dx)");
}

template <typename AT>
void launch_backward(const BnArgs &a, int blocks, int64_t rpb, hipStream_t st, int *rc) {
    if (a.dy3) launch_backward_ng<AT, 3>(a, blocks, rpb, st, rc);
    else if (a.dy2) launch_backward_ng<AT, 2>(a, blocks, rpb, st, rc);
    else launch_backward_ng<AT, 1>(a, blocks, rpb, st, rc);
}

}  // namespace

extern "C" int mmt_bn_relu_forward_ex(int64_t R, int C, const void *x, const void *residual, const float *weight,
                                      const float *bias, float *running_mean, float *running_var, float momentum,
                                      float eps, int relu, float *workspace, float *save, void *y, int act_dtype, void *stream) {
    MMT_REQUIRE_PTR(x);
    MMT_REQUIRE_PTR(workspace);
    MMT_REQUIRE_PTR(save);
    MMT_REQUIRE_PTR(y);
    if (act_dtype != MMT_DTYPE_F32 && act_dtype != MMT_DTYPE_BF16)
        return mmt::fail(MMT_ERR_BAD_FLAG, "bn_relu_forward: unknown activation dtype %d", act_dtype);
    float *save_mean = save, *save_rstd = save + C;
    BnArgs a = {};
    if (int rc = geometry("bn_relu_forward", R, C, &a.g)) return rc;
    const uintptr_t act_mask = act_dtype == MMT_DTYPE_F32 ? 15 : 7;      // one lane moves 4 channels: 16 or 8 bytes
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & act_mask) || (((uintptr_t)workspace | (uintptr_t)save) & 15))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "bn_relu_forward: buffers must be 16-byte aligned (bf16 activations: 8)");
    a.R = R; a.C = C; a.relu = relu; a.has_res = residual != nullptr; a.momentum = momentum; a.eps = eps;
    a.x = x; a.res = residual; a.weight = weight; a.bias = bias; a.running_mean = running_mean; a.running_var = running_var;
    a.acc = workspace + 2 * C; a.coef = workspace; a.scale = save + 2 * C; a.shift = save + 3 * C;
    a.save_mean = save_mean; a.save_rstd = save_rstd; a.y = y;
    int blocks; int64_t rpb;
    reduce_grid(a.g, R, act_dtype == MMT_DTYPE_F32 ? InFlight<float>::value : InFlight<bf16_t>::value, &blocks, &rpb);
    int rc = 0;
    if (act_dtype == MMT_DTYPE_F32) launch_forward<float>(a, blocks, rpb, (hipStream_t)stream, &rc);
    else launch_forward<bf16_t>(a, blocks, rpb, (hipStream_t)stream, &rc);
    return rc;
}

namespace {
// eval mode: y = x * scale + shift from the RUNNING statistics (nothing to reduce, nothing updated)
__global__ __launch_bounds__(256) void bn_eval_coefficients(BnArgs a) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= a.C) return;
    const float rstd = 1.f / sqrtf(a.running_var[c] + a.eps);
    const float w = a.weight ? a.weight[c] : 1.f, b = a.bias ? a.bias[c] : 0.f;
    a.scale[c] = w * rstd;
    a.shift[c] = b - a.running_mean[c] * (w * rstd);
}
}  // namespace

extern "C" int mmt_bn_relu_inference(int64_t R, int C, const void *x, const void *residual, const float *weight, const float *bias,
                                     const float *running_mean, const float *running_var, float eps, int relu, float *workspace,
                                     void *y, int act_dtype, void *stream) {
    MMT_REQUIRE_PTR(x);
    MMT_REQUIRE_PTR(running_mean);
    MMT_REQUIRE_PTR(running_var);
    MMT_REQUIRE_PTR(workspace);
    MMT_REQUIRE_PTR(y);
    if (act_dtype != MMT_DTYPE_F32 && act_dtype != MMT_DTYPE_BF16)
        return mmt::fail(MMT_ERR_BAD_FLAG, "bn_relu_inference: unknown activation dtype %d", act_dtype);
    BnArgs a = {};
    if (int rc = geometry("bn_relu_inference", R, C, &a.g)) return rc;
    const uintptr_t act_mask = act_dtype == MMT_DTYPE_F32 ? 15 : 7;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & act_mask) || ((uintptr_t)workspace & 15))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "bn_relu_inference: buffers must be 16-byte aligned (bf16 activations: 8)");
    a.R = R; a.C = C; a.relu = relu; a.has_res = residual != nullptr; a.eps = eps;
    a.x = x; a.res = residual; a.weight = weight; a.bias = bias;
    a.running_mean = const_cast<float *>(running_mean); a.running_var = const_cast<float *>(running_var);
    a.scale = workspace; a.shift = workspace + C; a.y = y;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_eval_coefficients, dim3((int)mmt::ceil_div(C, 256)), dim3(256), 0, st, a);
    if (int rc = mmt::check_launch("bn_relu_inference(coefficients)")) return rc;
    if (act_dtype == MMT_DTYPE_F32)
        hipLaunchKernelGGL((bn_map_kernel<0, float, 1>), dim3(mmt::stream_grid(mmt::ceil_div(R * a.g.C4, InFlight<float>::value), kBlock)), dim3(kBlock), 0, st, a);
    else
        hipLaunchKernelGGL((bn_map_kernel<0, bf16_t, 1>), dim3(mmt::stream_grid(mmt::ceil_div(R * a.g.C4, InFlight<bf16_t>::value), kBlock)), dim3(kBlock), 0, st, a);
    return mmt::check_launch("bn_relu_inference(apply)");
}

extern "C" int mmt_bn_relu_forward(int64_t R, int C, const float *x, const float *residual, const float *weight,
                                   const float *bias, float *running_mean, float *running_var, float momentum,
                                   float eps, int relu, float *workspace, float *save, float *y, void *stream) {
    return mmt_bn_relu_forward_ex(R, C, x, residual, weight, bias, running_mean, running_var, momentum, eps, relu, workspace, save, y,
                                  MMT_DTYPE_F32, stream);
}

extern "C" int mmt_bn_relu_backward_ex2(int64_t R, int C, const void *x, const void *y, const void *grad_y, const void *grad_y2,
                                        const void *grad_y3, int64_t grad_y_row_stride, const float *save, int relu, int has_residual, float *workspace,
                                        void *grad_x, void *grad_residual, float *grad_weight, float *grad_bias,
                                        int act_dtype, void *stream) {
    MMT_REQUIRE_PTR(x);
    MMT_REQUIRE_PTR(grad_y);
    MMT_REQUIRE_PTR(save);
    MMT_REQUIRE_PTR(workspace);
    if (act_dtype != MMT_DTYPE_F32 && act_dtype != MMT_DTYPE_BF16)
        return mmt::fail(MMT_ERR_BAD_FLAG, "bn_relu_backward: unknown activation dtype %d", act_dtype);
    const float *save_mean = save, *save_rstd = save + C;
    MMT_REQUIRE_PTR(grad_x);
    if (relu && has_residual) MMT_REQUIRE_PTR(y);
    if (has_residual) MMT_REQUIRE_PTR(grad_residual);
    BnArgs a = {};
    if (int rc = geometry("bn_relu_backward", R, C, &a.g)) return rc;
    const uintptr_t act_mask = act_dtype == MMT_DTYPE_F32 ? 15 : 7;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)grad_y | (uintptr_t)grad_y2 | (uintptr_t)grad_y3 | (uintptr_t)grad_x | (uintptr_t)grad_residual) & act_mask) ||
        (((uintptr_t)workspace | (uintptr_t)save) & 15))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "bn_relu_backward: buffers must be 16-byte aligned (bf16 activations: 8)");
    a.R = R; a.C = C; a.relu = relu; a.has_res = (relu && has_residual) ? 1 : 0;
    a.x = x; a.y_in = y; a.dy = grad_y; a.dy2 = grad_y2; a.dy3 = grad_y2 ? grad_y3 : nullptr;
    if (grad_y_row_stride != 0 && (grad_y_row_stride < C || (grad_y_row_stride & 3)))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "bn_relu_backward: grad_y_row_stride must be 0 (dense), or >= C and a multiple of 4");
    a.dy_pitch4 = grad_y_row_stride ? grad_y_row_stride / 4 : a.g.C4;
    a.acc = workspace + 2 * C; a.coef = workspace;
    a.scale = const_cast<float *>(save) + 2 * C; a.shift = const_cast<float *>(save) + 3 * C;
    a.save_mean = const_cast<float *>(save_mean); a.save_rstd = const_cast<float *>(save_rstd);
    a.dx = grad_x; a.dres = has_residual ? grad_residual : nullptr; a.dweight = grad_weight; a.dbias = grad_bias;
    int blocks; int64_t rpb;
    reduce_grid(a.g, R, act_dtype == MMT_DTYPE_F32 ? InFlight<float>::value : InFlight<bf16_t>::value, &blocks, &rpb);
    int rc = 0;
    if (act_dtype == MMT_DTYPE_F32) launch_backward<float>(a, blocks, rpb, (hipStream_t)stream, &rc);
    else launch_backward<bf16_t>(a, blocks, rpb, (hipStream_t)stream, &rc);
    return rc;
}

extern "C" int mmt_bn_relu_backward_ex(int64_t R, int C, const void *x, const void *y, const void *grad_y,
                                       const float *save, int relu, int has_residual, float *workspace,
                                       void *grad_x, void *grad_residual, float *grad_weight, float *grad_bias,
                                       int act_dtype, void *stream) {
    return mmt_bn_relu_backward_ex2(R, C, x, y, grad_y, nullptr, nullptr, 0, save, relu, has_residual, workspace, grad_x, grad_residual, grad_weight,
                                    grad_bias, act_dtype, stream);
}

extern "C" int mmt_bn_relu_backward(int64_t R, int C, const float *x, const float *y, const float *grad_y,
                                    const float *save, int relu, int has_residual, float *workspace,
                                    float *grad_x, float *grad_residual, float *grad_weight, float *grad_bias,
                                    void *stream) {
    return mmt_bn_relu_backward_ex(R, C, x, y, grad_y, save, relu, has_residual, workspace, grad_x, grad_residual, grad_weight, grad_bias,
                                   MMT_DTYPE_F32, stream);
}
