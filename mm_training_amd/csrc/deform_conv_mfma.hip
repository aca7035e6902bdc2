// Deformable 3x3 convolution (mmcv 'DCN' v1 = DeformConv2dPack, layers/backbones/lss_fpn.py:189-197) as IMPLICIT GEMMs on the
// fp32 matrix cores of gfx950 -- no [B*H*W, 9*C] column buffer in memory (deform_conv.hip's im2col + vendor GEMM wrote, read
// and re-read 311 MB of fp32 columns per pass at the DepthNet shape [24, 512, 16, 44], groups 4).
//
//   stride 1, pad 1, dilation 1, deform_groups 1, `groups` weight groups.  Channels-last:
//     x [B,H,W,C]   offset [B,H,W,18] (dy, dx per tap k = ky*3+kx)   weight [O, C/groups, 3, 3] (the torch layout)
//     out / grad_out [B,H,W,O]
//   A[p][(tap,c)] = bilinear sample of x[.., c] at (pixel p + tap + offset[p][tap])     -- never stored
//
//   forward   out[p][o]        = sum_(tap,c) A[p][(tap,c)] * w[o][c][tap]            dcn_fwd_mfma
//   wgrad     gw[o][c][tap]    = sum_p       A[p][(tap,c)] * go[p][o]                dcn_wgrad_mfma (+ dcn_wgrad_reduce)
//   dgrad     gc[p][(tap,c)]   = sum_o       go[p][o] * w[o][c][tap]                 dcn_dgrad_mfma; gc stays in registers:
//             grad_x[q][c]    += corner weight * gc   (LDS window of the image, ds_add_f32; written once)
//             grad_offset[p][tap] = sum_c gc * d(sample)/d(py, px)                   (partials per 16-channel chunk + reduce)
//
// All three use exact-fp32 MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain, MI355X_MICROARCH.md
// "Matrix cores"); the bound is the 157 TFLOP/s fp32 matrix rate (19.9 GFLOP per GEMM at the DepthNet shape = 127 us), not
// HBM: x + offsets + weights + out are ~77 MB.
#include "mmt_common.h"
#include "dcn_tap.h"
#include <stdlib.h>
#include <type_traits>

namespace {

using mmt_dcn::Tap;
using mmt_dcn::tap_at;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kKC = 32;        // input channels per K-step of the forward
constexpr int kLd = kKC + 4;   // LDS row stride of its tiles (floats): 144-byte rows, ds_read_b128 lane groups hit 16 distinct slots

struct TapRec { float w[4]; int o[4]; };   // bilinear weights + GLOBAL pixel indices (b*H*W + y*W + x) of the 4 corners

// blocks that share operands on one XCD (blockIdx % 8 labels the blocks that share an L2): bijective for any n
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, xcd = bid & 7, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

__device__ __forceinline__ float4 corner_mix(const float (&w)[4], const float4 (&v)[4]) {
    // a corner outside the image is read at its clamped position and replaced by an exact 0 (mmcv's `if`), so a
    // non-finite neighbour cannot leak through a zero weight
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 r = z;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 u = w[i] != 0.f ? v[i] : z;
        r.x = __builtin_fmaf(w[i], u.x, r.x);
        r.y = __builtin_fmaf(w[i], u.y, r.y);
        r.z = __builtin_fmaf(w[i], u.z, r.z);
        r.w = __builtin_fmaf(w[i], u.w, r.w);
    }
    return r;
}

__device__ __forceinline__ TapRec tap_record(int pos, int npos, int HW, int H, int W, int tap, const float *offset) {
    TapRec r;
    if (pos < npos) {
        const int b = pos / HW, hw = pos - b * HW, h = hw / W, w = hw - h * W;
        const Tap t = tap_at(h, w, tap, offset + (size_t)pos * 18, H, W);
        const int base = b * HW;
        r.w[0] = t.w1; r.w[1] = t.w2; r.w[2] = t.w3; r.w[3] = t.w4;
        r.o[0] = base + t.o1; r.o[1] = base + t.o2; r.o[2] = base + t.o3; r.o[3] = base + t.o4;
    } else {
        r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0.f;
        r.o[0] = r.o[1] = r.o[2] = r.o[3] = 0;
    }
    return r;
}

// weight [O][Cg][9] -> wf [g][tap][o][c] (forward: B^T rows, c contiguous) and wd [g][tap][c][o] (dgrad: A rows, o contiguous)
__global__ __launch_bounds__(256) void dcn_pack_weights(int O, int Cg, int groups, const float *weight, float *wf, float *wd) {
    const int Og = O / groups;
    const int64_t n = (int64_t)O * Cg * 9;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int tap = (int)(i % 9);
        const int64_t oc = i / 9;
        const int c = (int)(oc % Cg), o = (int)(oc / Cg), g = o / Og, og = o - g * Og;
        const float v = weight[i];
        if (wf) wf[(((int64_t)g * 9 + tap) * Og + og) * Cg + c] = v;
        if (wd) wd[(((int64_t)g * 9 + tap) * Cg + c) * Og + og] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// forward: a workgroup owns BM = 32*WM output pixels x BN output channels of one weight group; K-steps of (tap, 32 channels).
// Per K-step the bilinear taps of x are sampled straight into the LDS A tile (what dcn_im2col_kernel wrote to memory), the weight
// tile streams into the B tile; WM x WN waves, wave (wm, wn) multiplies rows 32*wm.. by columns wn*BN/WN.. with
// v_mfma_f32_32x32x2_f32.  Both tiles are [row][k] with k contiguous: lane (i, h) reads k = 8j + 4h .. + 3 with one ds_read_b128
// and feeds MFMA t with element t -- the k order inside a K-step is permuted identically for A and B.  Double-buffered: the
// global loads of step s+1 are in flight under the MFMAs of step s, one barrier per step.  The sampling points of a tap are
// computed once per (pixel, tap) by the first BM threads (offsets prefetched a tap ahead) and shared through LDS.
template <int WM, int WN, int BN>
__global__ __launch_bounds__(WM * WN * 64) void dcn_fwd_mfma(int H, int W, int C, int O, int groups, int npos, const float *__restrict__ x,
                                                             const float *__restrict__ offset, const float *__restrict__ wf,
                                                             float *__restrict__ out) {
    constexpr int BM = WM * 32, NT = WM * WN * 64, NB = BN / WN / 32;
    constexpr int PPP = NT / 8;                  // pixels per fill pass (8 lanes x float4 per pixel)
    constexpr int AP = BM / PPP;                 // fill passes
    constexpr int NBL = (BN * 8 + NT - 1) / NT;  // B-tile float4 per thread
    static_assert(BM % PPP == 0 && BM <= NT, "fill mapping");
    extern __shared__ __align__(16) float smem[];
    float *As = smem;                                   // [2][BM][kLd]
    float *Bs = As + 2 * BM * kLd;                      // [2][BN][kLd]
    TapRec *taps = reinterpret_cast<TapRec *>(Bs + 2 * BN * kLd);   // [9][BM]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int Cg = C / groups, Og = O / groups, ntn = Og / BN, HW = H * W;
    // Which tile.  Workgroup i runs on XCD i % 8 (a label for "shares an L2"): with groups | 8 a weight group's tiles take 8 / groups
    // XCDs, so an L2 holds ONE group's weights (0.6 MB at the DepthNet shape) next to its tiles' pixels instead of all four groups'
    // (2.4 MB of its 4 MB).
    int nt, g, mt;
    {
        const int mtiles = (npos + BM - 1) / BM;
        if (8 % groups == 0) {
            const int xps = 8 / groups, x = blockIdx.x & 7, q = blockIdx.x >> 3;          // grid = 8 * ceil(mtiles * ntn / xps)
            g = x / xps;
            const int u = q * xps + x % xps;
            nt = u % ntn; mt = u / ntn;
            if (mt >= mtiles) return;
        } else {
            int bid = blockIdx.x;
            nt = bid % ntn; bid /= ntn;
            g = bid % groups; mt = bid / groups;
        }
    }
    const int pos0 = mt * BM;
    // K-steps: channel chunk OUTER, tap INNER -- the nine taps of a pixel read the same 128-byte pieces of its 3 x 3 neighbours'
    // rows, so a piece is re-used within 9 consecutive steps (out of L2, and often L1) instead of once every Cg/32 steps
    const int nchunk = Cg / kKC, nsteps = 9 * nchunk;
    const float *wfg = wf + ((size_t)g * 9 * Og + (size_t)nt * BN) * Cg;
    const float *xg = x + g * Cg + (tid & 7) * 4;

    f32x16 acc[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    // two register stages: the loads of step s+2 are issued at the top of step s (a whole step ahead of their use)
    struct Stage {
        float4 va[AP][4];
        f32x4 vb[NBL];          // (an ext vector: a float4 struct copied global -> array -> LDS stays a memcpy through scratch)
        float wa[AP][4];
    };
    Stage sa, sb;

    // the sampling points of all 9 taps of the tile's pixels, once: record (tap, pixel) by thread e = tid, tid + NT, ...
    for (int e = tid; e < 9 * BM; e += NT) {
        const int tap = e / BM, pl = e - tap * BM, pos = pos0 + pl;
        TapRec r;
        if (pos < npos) {
            const int b = pos / HW, hw = pos - b * HW, h = hw / W, w = hw - h * W;
            const Tap t = tap_at(h, w, tap, offset + (size_t)pos * 18, H, W);
            const int base = b * HW;
            r.w[0] = t.w1; r.w[1] = t.w2; r.w[2] = t.w3; r.w[3] = t.w4;
            // byte offsets of the corner rows in x (B*H*W*C*4 < 2^32: dcn_shape)
            r.o[0] = (base + t.o1) * C * 4; r.o[1] = (base + t.o2) * C * 4; r.o[2] = (base + t.o3) * C * 4; r.o[3] = (base + t.o4) * C * 4;
        } else {
            r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0.f;
            r.o[0] = r.o[1] = r.o[2] = r.o[3] = 0;
        }
        taps[e] = r;
    }
    const unsigned lane_bytes = (tid & 7) * 16;
    auto issue_loads = [&](int s, Stage &st) {
        const int ch = s / 9, tap = s - ch * 9;
        const TapRec *tp = taps + tap * BM + (tid >> 3);
        const char *xc = reinterpret_cast<const char *>(x + g * Cg + ch * kKC);      // wave-uniform base + 32-bit per-lane offset
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const TapRec r = tp[p * PPP];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                st.wa[p][i] = r.w[i];
                st.va[p][i] = *reinterpret_cast<const float4 *>(xc + ((unsigned)r.o[i] + lane_bytes));
            }
        }
        const float *ws = wfg + (size_t)tap * Og * Cg + ch * kKC;
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
            const int e = min(tid + i * NT, BN * 8 - 1);      // (a thread past the tile re-reads its last float4)
            st.vb[i] = *reinterpret_cast<const f32x4 *>(ws + (unsigned)((e >> 3) * Cg + (e & 7) * 4));
        }
    };
    auto write_lds = [&](int buf, Stage &st) {
        float *ad = As + ((size_t)buf * BM + (tid >> 3)) * kLd + (tid & 7) * 4;
#pragma unroll
        for (int p = 0; p < AP; ++p) *reinterpret_cast<float4 *>(ad + p * PPP * kLd) = corner_mix(st.wa[p], st.va[p]);
        float *bd = Bs + (size_t)buf * BN * kLd;
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
            const int e = min(tid + i * NT, BN * 8 - 1);      // (threads past the tile store the same float4 again)
            *reinterpret_cast<f32x4 *>(bd + (e >> 3) * kLd + (e & 7) * 4) = st.vb[i];
        }
    };
    auto compute = [&](int buf, int j0, int j1) {
        const float *ar = As + ((size_t)buf * BM + wm * 32 + l31) * kLd + 4 * hh;
        const float *br = Bs + ((size_t)buf * BN + wn * (BN / WN) + l31) * kLd + 4 * hh;
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(ar + 8 * j);
            f32x4 b[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) b[n] = *reinterpret_cast<const f32x4 *>(br + n * 32 * kLd + 8 * j);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[n][t], acc[n], 0, 0, 0);
        }
    };
    // step s: `done` held step s (already in LDS; free for step s+2), `next` holds step s+1 (its loads issued a step ago).
    // FULL = neither of the two is past the end (no branches in the body).
    auto step = [&](int s, Stage &done, Stage &next, auto full) {
        constexpr bool FULL = decltype(full)::value;
        if (FULL || s + 2 < nsteps) issue_loads(s + 2, done);
        compute(s & 1, 0, kKC / 16);
        if (FULL || s + 1 < nsteps) write_lds((s + 1) & 1, next);
        compute(s & 1, kKC / 16, kKC / 8);
        __syncthreads();
    };

    __syncthreads();
    issue_loads(0, sa);
    issue_loads(1, sb);
    write_lds(0, sa);
    __syncthreads();
    int s = 0;
    for (; s + 3 < nsteps; s += 2) {
        step(s, sa, sb, std::true_type{});
        step(s + 1, sb, sa, std::true_type{});
    }
    for (; s < nsteps; s += 2) {
        step(s, sa, sb, std::false_type{});
        if (s + 1 < nsteps) step(s + 1, sb, sa, std::false_type{});
    }
    float *ob = out + g * Og + nt * BN + wn * (BN / WN) + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int pos = pos0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (pos < npos) {
#pragma unroll
            for (int n = 0; n < NB; ++n) ob[(size_t)pos * O + n * 32] = acc[n][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// weight gradient: a workgroup owns (weight group, tap, TC input channels x TO output channels) and one slice of the pixels; the
// reduction runs over the pixels, 32 per K-step: the same sampler fills As[p][c], grad_out rows fill Gs[p][o]; both tiles have the
// reduction index as the ROW, so lane (i, h) reads element [2s+h][i] with a conflict-free ds_read_b32.  4 waves as 2 x 2.
// Partial sums of the slices go to a slab [nsplit][groups][9][Cg][Og]; dcn_wgrad_reduce sums them into the torch layout.
template <int TC, int TO>
__global__ __launch_bounds__(256) void dcn_wgrad_mfma(int H, int W, int C, int O, int groups, int npos, int nsplit,
                                                      const float *__restrict__ x, const float *__restrict__ offset,
                                                      const float *__restrict__ go, float *__restrict__ slab) {
    constexpr int MI = TC / 64, NI = TO / 64;
    constexpr int LPP = TC / 4, PPP = 256 / LPP, PASSES = 32 / PPP;     // A fill: lanes per pixel, pixels per pass
    constexpr int GL = TO / 32;                                         // grad_out float4 per thread
    extern __shared__ __align__(16) float smem[];
    float *As = smem;                               // [2][32][TC]
    float *Gs = As + 2 * 32 * TC;                   // [2][32][TO]
    TapRec *taps = reinterpret_cast<TapRec *>(Gs + 2 * 32 * TO);   // [2][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int Cg = C / groups, Og = O / groups, nct = Cg / TC, nto = Og / TO, HW = H * W;
    // the 9 * nct * nto workgroups of one (pixel slice, weight group) read the same x and grad_out rows: consecutive ids, one XCD
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int ot = bid % nto; bid /= nto;
    const int ct = bid % nct; bid /= nct;
    const int tap = bid % 9; bid /= 9;
    const int g = bid % groups, split = bid / groups;
    const int nblk = (npos + 31) / 32;
    const int blk0 = (int)((int64_t)split * nblk / nsplit), blk1 = (int)((int64_t)(split + 1) * nblk / nsplit);
    const int nsteps = blk1 - blk0;
    const float *xg = x + g * Cg + ct * TC + (tid % LPP) * 4;
    const float *gg = go + g * Og + ot * TO;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    float4 va[PASSES][4], vg[GL];
    float wa[PASSES][4];

    auto write_taps = [&](int s) {
        if (tid < 32) taps[(s & 1) * 32 + tid] = tap_record((blk0 + s) * 32 + tid, npos, HW, H, W, tap, offset);
    };
    auto issue_loads = [&](int s) {
        const TapRec *tp = taps + (s & 1) * 32 + tid / LPP;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const TapRec r = tp[p * PPP];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wa[p][i] = r.w[i];
                va[p][i] = *reinterpret_cast<const float4 *>(xg + (size_t)r.o[i] * C);
            }
        }
#pragma unroll
        for (int i = 0; i < GL; ++i) {
            const int e = tid + i * 256, row = e / (TO / 4), c4 = (e % (TO / 4)) * 4;
            const int pos = (blk0 + s) * 32 + row;
            // pixels past the end contribute nothing: their grad_out row is 0 (their A row is a finite clamped sample)
            vg[i] = pos < npos ? *reinterpret_cast<const float4 *>(gg + (size_t)pos * O + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto write_lds = [&](int buf) {
        float *ad = As + ((size_t)buf * 32 + tid / LPP) * TC + (tid % LPP) * 4;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) *reinterpret_cast<float4 *>(ad + p * PPP * TC) = corner_mix(wa[p], va[p]);
        float *gd = Gs + (size_t)buf * 32 * TO;
#pragma unroll
        for (int i = 0; i < GL; ++i) {
            const int e = tid + i * 256;
            *reinterpret_cast<float4 *>(gd + (e / (TO / 4)) * TO + (e % (TO / 4)) * 4) = vg[i];
        }
    };
    auto compute = [&](int buf) {
        const float *ar = As + ((size_t)buf * 32 + hh) * TC + wm * (TC / 2) + l31;
        const float *br = Gs + ((size_t)buf * 32 + hh) * TO + wn * (TO / 2) + l31;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float a[MI], b[NI];
#pragma unroll
            for (int m = 0; m < MI; ++m) a[m] = ar[2 * s * TC + m * 32];
#pragma unroll
            for (int n = 0; n < NI; ++n) b[n] = br[2 * s * TO + n * 32];
#pragma unroll
            for (int m = 0; m < MI; ++m)
#pragma unroll
                for (int n = 0; n < NI; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[m][n], 0, 0, 0);
        }
    };

    if (nsteps > 0) {
        write_taps(0);
        if (nsteps > 1) write_taps(1);
        __syncthreads();
        issue_loads(0);
        write_lds(0);
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const bool more = s + 1 < nsteps;
            if (more) issue_loads(s + 1);
            compute(s & 1);
            if (more) write_lds((s + 1) & 1);
            if (s + 2 < nsteps) write_taps(s + 2);      // buffer s & 1: last read at the top of step s - 1
            __syncthreads();
        }
    }
    float *sb = slab + ((((size_t)split * groups + g) * 9 + tap) * Cg + ct * TC + wm * (TC / 2)) * Og + ot * TO + wn * (TO / 2) + l31;
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                sb[(size_t)row * Og + n * 32] = acc[m][n][r];
            }
}

// slab [nsplit][g][tap][c][o] -> grad_weight [O][Cg][3][3], slices summed in slice order (deterministic)
__global__ __launch_bounds__(256) void dcn_wgrad_reduce(int nsplit, int groups, int Cg, int Og, const float *slab, float *grad_weight) {
    const int64_t n = (int64_t)groups * 9 * Cg * Og;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += slab[(int64_t)k * n + i];
        const int o = (int)(i % Og);
        int64_t r = i / Og;
        const int c = (int)(r % Cg); r /= Cg;
        const int tap = (int)(r % 9), g = (int)(r / 9);
        grad_weight[(((int64_t)g * Og + o) * Cg + c) * 9 + tap] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// data + offset gradient: a workgroup owns (image, band of rows, weight group, 16 input channels).  A wave owns 32 pixels of the
// band (two 16-pixel sub-tiles): their grad_out rows sit in registers as the B operand for all 9 taps; per tap the [16 c][OG]
// weight tile is the A operand (LDS, shared by the waves), v_mfma_f32_16x16x4_f32 leaves gc[c][p] with the PIXEL on the lane
// and the lane's 4 channels in registers -- so the sampling point of (pixel, tap) is per-lane state, the dot products with the
// four corner rows of x (the offset gradient) are per-lane fmaf chains, and the scatter of weight * gc into grad_x goes to an
// LDS image of the band's rows (+ halo) with ds_add_f32.  Corners outside that window (large offsets) take a global atomic.
// bands == 1: the window is the whole image, every contribution lands in it, the slice of grad_x is written once with plain
// stores (no zero-fill, no global atomics).  Otherwise the windows are flushed with global atomics into a zero-filled grad_x.
// part [C/16][9][2][npos]: per-chunk (d/dy, d/dx) sums; dcn_offset_reduce_parts adds the chunks.
constexpr int kWinLd = 17;       // floats per window pixel (16 channels + 1: consecutive pixels on consecutive banks)

template <int OG>
__global__ __launch_bounds__(768) void dcn_dgrad_mfma(int H, int W, int C, int O, int groups, int npos, int bands, int band_rows, int halo,
                                                      const float *__restrict__ x, const float *__restrict__ offset,
                                                      const float *__restrict__ wd, const float *__restrict__ go,
                                                      float *__restrict__ grad_x, float *__restrict__ part) {
    constexpr int KQ = OG / 4;        // output channels (k values) per lane quarter
    constexpr int WLD = OG + 2;       // weight-tile row stride: ds_read_b64 of lane (c, kq) at c*WLD + kq*KQ is conflict-free
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int nth = blockDim.x, nw = nth >> 6;
    const int Cg = C / groups, HW = H * W, nch = Cg / 16;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int chunk = bid % nch; bid /= nch;
    const int g = bid % groups; bid /= groups;
    const int band = bid % bands, b = bid / bands;
    const int row0 = band * band_rows, row1 = min(H, row0 + band_rows);
    const int wrow0 = max(0, row0 - halo), wrow1 = min(H, row1 + halo);
    const int wpix0 = wrow0 * W, wnpix = (wrow1 - wrow0) * W;
    const int p0 = row0 * W, p1 = row1 * W;
    const int ntiles = (p1 - p0 + 31) / 32, rounds = (ntiles + nw - 1) / nw;
    const int cb = g * Cg + chunk * 16;
    const int cglob = g * nch + chunk;
    float *win = smem;                                   // [wnpix][kWinLd]
    float *wl = smem + (((size_t)(2 * halo + band_rows) * W * kWinLd + 3) & ~(size_t)3);   // [2][16][WLD]
    const float *wdg = wd + ((size_t)g * 9 * Cg + chunk * 16) * OG;
    const float *gob = go + (size_t)b * HW * O + g * OG + kq * KQ;
    const float *xb = x + (size_t)b * HW * C + cb + 4 * kq;
    const float *ofb = offset + (size_t)b * HW * 18;
    float *gxb = grad_x + (size_t)b * HW * C + cb;

    for (int i = tid; i < wnpix * kWinLd; i += nth) win[i] = 0.f;
    // weight tile of tap 0: 16 rows x OG/4 float4, at most two per thread (>= 256 threads)
    bool wfill[2];
    int wsrc[2], wdst[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = tid + k * nth;
        wfill[k] = e < 4 * OG;
        const int wc = e / (OG / 4), wo = (e % (OG / 4)) * 4;
        wsrc[k] = wc * OG + wo;
        wdst[k] = wc * WLD + wo;
        if (wfill[k]) {
            const float4 v = *reinterpret_cast<const float4 *>(wdg + wsrc[k]);
            float *d = wl + wdst[k];
            *reinterpret_cast<f32x2 *>(d) = f32x2{v.x, v.y};
            *reinterpret_cast<f32x2 *>(d + 2) = f32x2{v.z, v.w};
        }
    }
    __syncthreads();

    int buf = 0;
    for (int rd = 0; rd < rounds; ++rd) {
        const int tile = rd * nw + wave;
        const bool active = tile < ntiles;
        float gq[2][KQ];
        int pp[2];
        bool valid[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            pp[sub] = p0 + tile * 32 + sub * 16 + l15;
            valid[sub] = active && pp[sub] < p1;
            if (!valid[sub]) pp[sub] = p0;
            const float *src = gob + (size_t)pp[sub] * O;
#pragma unroll
            for (int j = 0; j < KQ / 4; ++j) {
                const float4 v = valid[sub] ? *reinterpret_cast<const float4 *>(src + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
                gq[sub][4 * j] = v.x; gq[sub][4 * j + 1] = v.y; gq[sub][4 * j + 2] = v.z; gq[sub][4 * j + 3] = v.w;
            }
        }
        for (int tap = 0; tap < 9; ++tap) {
            const bool pre = tap < 8 || rd + 1 < rounds;      // the next tap's weight tile (tap 0 again for the next round)
            float4 vw[2];
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (pre && wfill[k]) vw[k] = *reinterpret_cast<const float4 *>(wdg + (size_t)(tap == 8 ? 0 : tap + 1) * Cg * OG + wsrc[k]);
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            const float *ar = wl + buf * 16 * WLD + l15 * WLD + kq * KQ;
#pragma unroll
            for (int s = 0; s < KQ / 2; ++s) {
                const f32x2 a = *reinterpret_cast<const f32x2 *>(ar + 2 * s);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], gq[0][2 * s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], gq[1][2 * s], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], gq[0][2 * s + 1], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], gq[1][2 * s + 1], acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int p = pp[sub];
                const int h = p / W, w = p - h * W;
                const Tap t = tap_at(h, w, tap, ofb + (size_t)p * 18, H, W);
                const float tw[4] = {t.w1, t.w2, t.w3, t.w4}, tdy[4] = {t.dy1, t.dy2, t.dy3, t.dy4}, tdx[4] = {t.dx1, t.dx2, t.dx3, t.dx4};
                const int to[4] = {t.o1, t.o2, t.o3, t.o4};
                const f32x4 a4 = acc[sub];
                float4 xv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const float4 *>(xb + (size_t)to[i] * C);
                float gy = 0.f, gx = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // the corner takes part when it is inside the image: a non-zero weight or a non-zero coordinate derivative
                    const bool live = valid[sub] && (tw[i] != 0.f || tdy[i] != 0.f || tdx[i] != 0.f);
                    float d = __builtin_fmaf(a4[0], xv[i].x, __builtin_fmaf(a4[1], xv[i].y, __builtin_fmaf(a4[2], xv[i].z, a4[3] * xv[i].w)));
                    d = live ? d : 0.f;
                    gy = __builtin_fmaf(tdy[i], d, gy);
                    gx = __builtin_fmaf(tdx[i], d, gx);
                    if (valid[sub] && tw[i] != 0.f) {
                        const int wp = to[i] - wpix0;
                        if ((unsigned)wp < (unsigned)wnpix) {
                            float *dst = win + wp * kWinLd + 4 * kq;
#pragma unroll
                            for (int r = 0; r < 4; ++r) unsafeAtomicAdd(dst + r, tw[i] * a4[r]);
                        } else {
                            float *dst = gxb + (size_t)to[i] * C + 4 * kq;
#pragma unroll
                            for (int r = 0; r < 4; ++r) unsafeAtomicAdd(dst + r, tw[i] * a4[r]);
                        }
                    }
                }
                gy += __shfl_xor(gy, 16); gx += __shfl_xor(gx, 16);
                gy += __shfl_xor(gy, 32); gx += __shfl_xor(gx, 32);
                if (kq == 0 && valid[sub]) {
                    float *pd = part + ((size_t)(cglob * 9 + tap) * 2) * npos + (size_t)b * HW + p;
                    pd[0] = gy;
                    pd[npos] = gx;
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (pre && wfill[k]) {
                    float *d = wl + (buf ^ 1) * 16 * WLD + wdst[k];
                    *reinterpret_cast<f32x2 *>(d) = f32x2{vw[k].x, vw[k].y};
                    *reinterpret_cast<f32x2 *>(d + 2) = f32x2{vw[k].z, vw[k].w};
                }
            buf ^= 1;
            __syncthreads();
        }
    }
    // flush the window: 4 threads (a float4 each) per pixel
    for (int i = tid; i < wnpix * 4; i += nth) {
        const int wp = i >> 2, j = (i & 3) * 4;
        const float *s = win + wp * kWinLd + j;
        float *d = gxb + (size_t)(wpix0 + wp) * C + j;
        if (bands == 1) {
            *reinterpret_cast<float4 *>(d) = make_float4(s[0], s[1], s[2], s[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) unsafeAtomicAdd(d + r, s[r]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The data gradient WITHOUT float atomics.  ds_add_f32 runs at one LANE per 3 cycles on gfx950 (192 cycles per wave
// instruction whatever the addresses: tools/ubench/lds_atomic.hip) -- the LDS-window scatter of dcn_dgrad_mfma spends 1.3 ms
// of its 1.7 ms in them at the DepthNet shape.  Here the scatter is turned into a gather through per-destination lists:
//   dcn_plan_taps      one workgroup per (tap, image): every source pixel's sampling point has up to 4 corners with a non-zero
//                      weight; they are binned by (round of the source, destination pixel) in LDS (count -> scan -> counting
//                      sort, integer LDS atomics) and leave as lists of (source pixel, weight).  A "round" = the nw*32 source
//                      pixels the gather kernel's waves hold at a time.
//   dcn_dgrad_gather   as dcn_dgrad_mfma up to gc[c][p] in registers and the offset-gradient dots; then each wave stages its
//                      gc tile in LDS ([pixel of the round][16 channels], double-buffered), and after the barrier every thread
//                      walks the lists of the destinations it owns -- (pixel, 4 channels) items, fixed for the whole kernel,
//                      sums in registers -- reading the staged rows.  grad_x is written once at the end with plain stores.
// The gather of (round, tap) is issued after the MFMAs + staging of the NEXT tap, so that waves drift apart and one wave's list
// walk runs under another's matrix work.  Needs H*W*4 items <= 4 per thread (H*W <= 64 * waves <= 768) and rounds*H*W bins in
// LDS; larger images take dcn_dgrad_mfma (the general, banded form).
// what the offset gradient needs of a sampling point: the corner pixels, d(weight)/d(py, px), and which corners take part
// (inside the image: a non-zero weight or a non-zero coordinate derivative)
struct CornerInfo { int o[4]; float dy[4], dx[4]; unsigned live; };

__device__ __forceinline__ CornerInfo corner_info(const Tap &t, bool valid) {
    CornerInfo c;
    c.o[0] = t.o1; c.o[1] = t.o2; c.o[2] = t.o3; c.o[3] = t.o4;
    c.dy[0] = t.dy1; c.dy[1] = t.dy2; c.dy[2] = t.dy3; c.dy[3] = t.dy4;
    c.dx[0] = t.dx1; c.dx[1] = t.dx2; c.dx[2] = t.dx3; c.dx[3] = t.dx4;
    const float w[4] = {t.w1, t.w2, t.w3, t.w4};
    c.live = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) c.live |= (valid && (w[i] != 0.f || c.dy[i] != 0.f || c.dx[i] != 0.f)) ? (1u << i) : 0u;
    return c;
}

// (d/dy, d/dx) of one (pixel, tap) over this lane quarter's 4 channels, summed over the 4 quarters; lane quarter 0 stores
__device__ __forceinline__ void offset_dots(const CornerInfo &c, f32x4 a4, const float4 (&xv)[4], bool store, float *pd, int npos) {
    float gy = 0.f, gx = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float d = __builtin_fmaf(a4[0], xv[i].x, __builtin_fmaf(a4[1], xv[i].y, __builtin_fmaf(a4[2], xv[i].z, a4[3] * xv[i].w)));
        d = (c.live >> i) & 1 ? d : 0.f;
        gy = __builtin_fmaf(c.dy[i], d, gy);
        gx = __builtin_fmaf(c.dx[i], d, gx);
    }
    gy += __shfl_xor(gy, 16); gx += __shfl_xor(gx, 16);
    gy += __shfl_xor(gy, 32); gx += __shfl_xor(gx, 32);
    if (store) {
        pd[0] = gy;
        pd[npos] = gx;
    }
}

struct PlanEntry { int src; float w; };     // source pixel inside the image, bilinear weight of the corner that lands here

constexpr int kPlanThreads = 1024;
constexpr int kPlanMaxBins = 8192;

__global__ __launch_bounds__(kPlanThreads) void dcn_plan_taps(int H, int W, int ppr, int rounds, const float *offset, int32_t *bin_off,
                                                              PlanEntry *entries) {
    __shared__ int cnt[kPlanMaxBins];
    __shared__ int off[kPlanMaxBins + 1];
    __shared__ int wsum[kPlanThreads / 64];
    const int HW = H * W, nbins = rounds * HW;
    const int tap = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *ob = offset + (int64_t)b * HW * 18;
    for (int i = tid; i < nbins; i += kPlanThreads) cnt[i] = 0;
    __syncthreads();
    for (int p = tid; p < HW; p += kPlanThreads) {
        const int h = p / W, w = p - h * W;
        const Tap t = tap_at(h, w, tap, ob + (int64_t)p * 18, H, W);
        const int rb = (p / ppr) * HW;
        if (t.w1 != 0.f) atomicAdd(&cnt[rb + t.o1], 1);
        if (t.w2 != 0.f) atomicAdd(&cnt[rb + t.o2], 1);
        if (t.w3 != 0.f) atomicAdd(&cnt[rb + t.o3], 1);
        if (t.w4 != 0.f) atomicAdd(&cnt[rb + t.o4], 1);
    }
    __syncthreads();
    // exclusive scan of the bin counts: kPer consecutive bins per thread, wave scan, then the wave totals
    constexpr int kPer = kPlanMaxBins / kPlanThreads;
    int loc[kPer], sum = 0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int idx = tid * kPer + i;
        loc[i] = idx < nbins ? cnt[idx] : 0;
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
        if (idx < nbins) { off[idx] = run; cnt[idx] = 0; }
        run += loc[i];
    }
    if (tid == kPlanThreads - 1) off[nbins] = run;
    __syncthreads();
    int32_t *bo = bin_off + ((int64_t)b * 9 + tap) * (nbins + 1);
    for (int i = tid; i <= nbins; i += kPlanThreads) bo[i] = off[i];
    PlanEntry *eb = entries + ((int64_t)b * 9 + tap) * HW * 4;
    for (int p = tid; p < HW; p += kPlanThreads) {
        const int h = p / W, w = p - h * W;
        const Tap t = tap_at(h, w, tap, ob + (int64_t)p * 18, H, W);
        const int rb = (p / ppr) * HW;
        if (t.w1 != 0.f) eb[off[rb + t.o1] + atomicAdd(&cnt[rb + t.o1], 1)] = PlanEntry{p, t.w1};
        if (t.w2 != 0.f) eb[off[rb + t.o2] + atomicAdd(&cnt[rb + t.o2], 1)] = PlanEntry{p, t.w2};
        if (t.w3 != 0.f) eb[off[rb + t.o3] + atomicAdd(&cnt[rb + t.o3], 1)] = PlanEntry{p, t.w3};
        if (t.w4 != 0.f) eb[off[rb + t.o4] + atomicAdd(&cnt[rb + t.o4], 1)] = PlanEntry{p, t.w4};
    }
}

constexpr int kGcLd = 20;        // floats per staged pixel row (16 channels, 80-byte rows: 16-byte aligned float4 slots)
constexpr int kItems = 1;        // destination pixels per thread (all 16 channels of the chunk each)

template <int OG>
__global__ __launch_bounds__(768) void dcn_dgrad_gather(int H, int W, int C, int O, int groups, int npos, int rounds,
                                                        const float *__restrict__ x, const float *__restrict__ offset,
                                                        const float *__restrict__ wd, const float *__restrict__ go,
                                                        const int32_t *__restrict__ bin_off, const PlanEntry *__restrict__ entries,
                                                        float *__restrict__ grad_x, float *__restrict__ part) {
    constexpr int KQ = OG / 4;
    constexpr int WLD = OG + 2;
    constexpr int kPieces = 16 * OG / 64;       // 256-byte pieces of a weight tile (one LDS-DMA wave instruction each)
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int nth = blockDim.x, nw = nth >> 6, ppr = nw * 32;
    const int Cg = C / groups, HW = H * W, nch = Cg / 16;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int chunk = bid % nch; bid /= nch;
    const int g = bid % groups, b = bid / groups;
    const int ntiles = (HW + 31) / 32;
    const int cb = g * Cg + chunk * 16;
    const int cglob = g * nch + chunk;
    float *gcs = smem;                                   // [2][ppr][kGcLd]
    float *wl = smem + 2 * ppr * kGcLd;                  // [2][16][WLD]
    const float *wdg = wd + ((size_t)g * 9 * Cg + chunk * 16) * OG;
    const float *gob = go + (size_t)b * HW * O + g * OG + kq * KQ;
    // corner rows of x: wave-uniform base + 32-bit byte offsets (B*H*W*C*4 < 2^32: dcn_shape)
    const char *xbase = reinterpret_cast<const char *>(x + (size_t)b * HW * C + cb);
    const unsigned xstride = (unsigned)C * 4, xlane = 16u * kq;
    const float *ofb = offset + (size_t)b * HW * 18;
    const int nbins = rounds * HW;

    // the destinations this thread owns: item k = pixel tid + k*nth, all 16 channels of the chunk.  Consecutive lanes own
    // consecutive pixels, so the lists of a wave's 64 pixels of item k are ONE contiguous run of the entry array.
    float4 sums[kItems][4];
#pragma unroll
    for (int k = 0; k < kItems; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) sums[k][c] = make_float4(0.f, 0.f, 0.f, 0.f);

    // weight tile of tap t -> wl[buf]: LDS-DMA, 64 lanes x 4 bytes per instruction (a 256-byte piece of a row; the rows are padded
    // by 8 bytes, so wider pieces would cross the pad), the pieces dealt to the waves
    auto fill_weights = [&](int tap, int buf) {
        for (int pc = wave; pc < kPieces; pc += nw) {
            const int row = pc / (OG / 64), hf = pc - row * (OG / 64);
            const float *src = wdg + ((size_t)tap * Cg + row) * OG + hf * 64 + lane;
            float *dst = wl + (buf * 16 + row) * WLD + hf * 64;
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)dst, 4, 0, 0);
        }
    };
    fill_weights(0, 0);

    // list bounds of unit (rd, tap) for this thread's items; pixels past the image get the (empty) end of the round's run
    int lo[kItems], hi[kItems];
    auto bounds = [&](int rd, int tap) {
        const int32_t *bo = bin_off + ((size_t)b * 9 + tap) * (nbins + 1) + rd * HW;
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const int q = min(tid + k * nth, HW);
            lo[k] = bo[q];
            hi[k] = bo[min(q + 1, HW)];
        }
    };
    // the wave's run of entries per item: lane l holds entry run_lo + l (runs longer than 64 entries: the rest is read in place)
    int esrc[kItems];
    float ew[kItems];
    auto prefetch_entries = [&](int tap) {
        const PlanEntry *eb = entries + ((size_t)b * 9 + tap) * HW * 4;
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const int rlo = __builtin_amdgcn_readfirstlane(lo[k]), rhi = __builtin_amdgcn_readlane(hi[k], 63);
            const int e = min(rlo + lane, max(rhi - 1, rlo));
            const PlanEntry en = rhi > rlo ? eb[e] : PlanEntry{0, 0.f};
            esrc[k] = en.src;
            ew[k] = en.w;
        }
    };
    auto gather_item = [&](int k, int rd, int tap, int sbuf) {
        const PlanEntry *eb = entries + ((size_t)b * 9 + tap) * HW * 4;
        const float *gb = gcs + (sbuf * ppr - rd * ppr) * kGcLd;
        const int rlo = __builtin_amdgcn_readfirstlane(lo[k]);
        const int n = hi[k] - lo[k];
        for (int j = 0; __any(j < n); ++j) {
            const int idx = lo[k] + j - rlo;
            int src = __shfl(esrc[k], idx & 63);
            float w = __shfl(ew[k], idx & 63);
            if (j < n) {
                if (idx >= 64) { const PlanEntry en = eb[lo[k] + j]; src = en.src; w = en.w; }
                const float *row = gb + src * kGcLd;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 v = *reinterpret_cast<const float4 *>(row + 4 * c);
                    sums[k][c].x = __builtin_fmaf(w, v.x, sums[k][c].x);
                    sums[k][c].y = __builtin_fmaf(w, v.y, sums[k][c].y);
                    sums[k][c].z = __builtin_fmaf(w, v.z, sums[k][c].z);
                    sums[k][c].w = __builtin_fmaf(w, v.w, sums[k][c].w);
                }
            }
        }
    };

    __syncthreads();
    int buf = 0;
    for (int rd = 0; rd < rounds; ++rd) {
        const int tile = rd * nw + wave;
        const bool active = tile < ntiles;
        float gq[2][KQ];
        int phw[2];               // the sub-tile's pixel of this lane as (row << 16) | column (registers are what this kernel is short of)
        bool valid[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            int p = tile * 32 + sub * 16 + l15;
            valid[sub] = active && p < HW;
            if (!valid[sub]) p = 0;
            const int h = p / W;
            phw[sub] = (h << 16) | (p - h * W);
            const float *src = gob + (size_t)p * O;
#pragma unroll
            for (int j = 0; j < KQ / 4; ++j) {
                const float4 v = valid[sub] ? *reinterpret_cast<const float4 *>(src + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
                gq[sub][4 * j] = v.x; gq[sub][4 * j + 1] = v.y; gq[sub][4 * j + 2] = v.z; gq[sub][4 * j + 3] = v.w;
            }
        }
        float2 onext[2];        // the two pixels' offsets of the tap about to be processed (loaded a unit ahead)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) onext[sub] = *reinterpret_cast<const float2 *>(ofb + (size_t)((phw[sub] >> 16) * W + (phw[sub] & 0xFFFF)) * 18);
        for (int tap = 0; tap < 9; ++tap) {
            const int u = rd * 9 + tap;
            const int prd = tap == 0 ? rd - 1 : rd, ptap = tap == 0 ? 8 : tap - 1;     // the unit whose staged tile is gathered now
            // the next tap's weight tile (tap 0 again for the next round) into the other buffer: last read before the barrier
            if (tap < 8 || rd + 1 < rounds) fill_weights(tap == 8 ? 0 : tap + 1, buf ^ 1);
            if (u > 0) prefetch_entries(ptap);          // (the bounds of the previous unit were loaded at its end)
            const int ky = tap / 3, kx = tap - ky * 3;
            // sub-tile 0: sampling point and corner rows before the matrix work (the rows arrive under it)
            CornerInfo c0 = corner_info(mmt_dcn::make_tap((float)((phw[0] >> 16) + ky - 1) + onext[0].x, (float)((phw[0] & 0xFFFF) + kx - 1) + onext[0].y, H, W), valid[0]);
            float4 xv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const float4 *>(xbase + ((unsigned)c0.o[i] * xstride + xlane));
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            const float *ar = wl + buf * 16 * WLD + l15 * WLD + kq * KQ;
#pragma unroll
            for (int s = 0; s < KQ / 2; ++s) {
                const f32x2 a = *reinterpret_cast<const f32x2 *>(ar + 2 * s);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], gq[0][2 * s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], gq[1][2 * s], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], gq[0][2 * s + 1], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], gq[1][2 * s + 1], acc[1], 0, 0, 0);
            }
            float *pd = part + ((size_t)(cglob * 9 + tap) * 2) * npos + (size_t)b * HW;
            // stage gc[pixel of the round][4 channels of this lane] (pixels past the image stage zeros: never listed)
            float *sg = gcs + (buf * ppr + wave * 32 + l15) * kGcLd + 4 * kq;
            *reinterpret_cast<f32x4 *>(sg) = acc[0];
            *reinterpret_cast<f32x4 *>(sg + 16 * kGcLd) = acc[1];
            offset_dots(c0, acc[0], xv, kq == 0 && valid[0], pd + ((phw[0] >> 16) * W + (phw[0] & 0xFFFF)), npos);
            // sub-tile 1: its corner rows arrive under the previous unit's gather
            CornerInfo c1 = corner_info(mmt_dcn::make_tap((float)((phw[1] >> 16) + ky - 1) + onext[1].x, (float)((phw[1] & 0xFFFF) + kx - 1) + onext[1].y, H, W), valid[1]);
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const float4 *>(xbase + ((unsigned)c1.o[i] * xstride + xlane));
            if (u > 0) {
#pragma unroll
                for (int k = 0; k < kItems; ++k) gather_item(k, prd, ptap, buf ^ 1);
            }
            offset_dots(c1, acc[1], xv, kq == 0 && valid[1], pd + ((phw[1] >> 16) * W + (phw[1] & 0xFFFF)), npos);
            if (tap < 8) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
                    onext[sub] = *reinterpret_cast<const float2 *>(ofb + (size_t)((phw[sub] >> 16) * W + (phw[sub] & 0xFFFF)) * 18 + 2 * (tap + 1));
            }
            bounds(rd, tap);                 // for the gather of THIS unit, one unit from now
            buf ^= 1;
            __syncthreads();
        }
    }
    prefetch_entries(8);
#pragma unroll
    for (int k = 0; k < kItems; ++k) gather_item(k, rounds - 1, 8, buf ^ 1);
    float *gxb = grad_x + (size_t)b * HW * C + cb;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int q = tid + k * nth;
        if (q < HW) {
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<float4 *>(gxb + (size_t)q * C + 4 * c) = sums[k][c];
        }
    }
}

// grad_offset[pos][2*tap + yx] = sum over the 16-channel chunks of part[chunk][tap][yx][pos], in chunk order
__global__ __launch_bounds__(256) void dcn_offset_reduce_parts(int npos, int nparts, const float *part, float *grad_offset) {
    const int64_t n = (int64_t)npos * 18;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int pos = (int)(i % npos), j = (int)(i / npos);       // j = tap*2 + yx
        float s = 0.f;
        for (int k = 0; k < nparts; ++k) s += part[((int64_t)k * 18 + j) * npos + pos];
        grad_offset[(int64_t)pos * 18 + j] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
struct DcnShape {
    int B, H, W, C, O, groups, Cg, Og, npos;
};

constexpr int kLdsBudget = 160 * 1024;
constexpr int kMaxWaves = 12;       // dcn_dgrad_mfma: __launch_bounds__(768)

int dcn_shape(int B, int H, int W, int C, int O, int groups, DcnShape *s, const char *what, bool quiet) {
    auto bad = [&](int code, const char *msg) { return quiet ? code : mmt::fail(code, "%s: %s (B=%d H=%d W=%d C=%d O=%d groups=%d)", what, msg, B, H, W, C, O, groups); };
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || O <= 0 || groups <= 0 || C % groups != 0 || O % groups != 0) return bad(MMT_ERR_BAD_SHAPE, "bad sizes");
    const int Cg = C / groups, Og = O / groups;
    if (Cg % 64 != 0 || (Og != 64 && Og != 128)) return bad(MMT_ERR_BAD_SHAPE, "the matrix-core form needs C/groups % 64 == 0 and O/groups in {64, 128}");
    if ((int64_t)B * H * W * (int64_t)(C > O ? C : O) >= (1ll << 30)) return bad(MMT_ERR_TOO_LARGE, "B*H*W*max(C,O)*4 bytes exceeds 32 bits");
    // the window of the data gradient: at least 1 row + 2 halo rows of W pixels next to the weight tiles
    if ((int64_t)3 * W * kWinLd * 4 + 2 * 16 * (Og + 2) * 4 + 64 > kLdsBudget) return bad(MMT_ERR_BAD_SHAPE, "W too large for the LDS window");
    *s = DcnShape{B, H, W, C, O, groups, Cg, Og, B * H * W};
    return 0;
}

struct DgradPlan { int bands, band_rows, halo, nw; size_t lds; };

DgradPlan dgrad_plan(const DcnShape &s) {
    DgradPlan p;
    const size_t wl = (size_t)2 * 16 * (s.Og + 2) * 4 + 64;
    const int rows_max = (int)((kLdsBudget - wl) / ((size_t)s.W * kWinLd * 4));
    if (s.H <= rows_max) { p.bands = 1; p.band_rows = s.H; p.halo = 0; }
    else {
        p.halo = rows_max >= 8 ? 2 : 1;
        const int r = rows_max - 2 * p.halo;
        p.bands = (s.H + r - 1) / r;
        p.band_rows = (s.H + p.bands - 1) / p.bands;      // even bands
        p.bands = (s.H + p.band_rows - 1) / p.band_rows;
    }
    p.lds = ((((size_t)(2 * p.halo + p.band_rows) * s.W * kWinLd + 3) & ~(size_t)3) * 4) + (size_t)2 * 16 * (s.Og + 2) * 4;
    // waves per workgroup: the count that wastes the fewest wave slots over the band's 32-pixel tiles (larger on ties)
    const int ntiles = (p.band_rows * s.W + 31) / 32;
    int best = 4; double beff = 0.0;
    for (int nw = 4; nw <= kMaxWaves; ++nw) {
        const int rounds = (ntiles + nw - 1) / nw;
        const double eff = (double)ntiles / ((double)rounds * nw);
        if (eff >= beff - 1e-9) { beff = eff; best = nw; }
    }
    p.nw = best;
    return p;
}

// the gather form of the data gradient: waves per workgroup (fewest wasted wave slots over the image's 32-pixel tiles, every
// (pixel, 4 channels) destination owned by a thread), rounds, LDS; ok = the image fits it
struct GatherPlan { bool ok; int nw, rounds; size_t lds; };

GatherPlan gather_plan(const DcnShape &s) {
    GatherPlan p{false, 0, 0, 0};
    const int HW = s.H * s.W, ntiles = (HW + 31) / 32;
    double beff = 0.0;
    for (int nw = 4; nw <= kMaxWaves; ++nw) {
        if (HW > kItems * nw * 64) continue;
        const int rounds = (ntiles + nw - 1) / nw;
        const double eff = (double)ntiles / ((double)rounds * nw);
        if (eff > beff + 1e-9) { beff = eff; p.nw = nw; p.rounds = rounds; }
    }
    if (p.nw == 0 || (int64_t)p.rounds * HW > kPlanMaxBins) return p;
    p.lds = (size_t)2 * p.nw * 32 * kGcLd * 4 + (size_t)2 * 16 * (s.Og + 2) * 4;
    p.ok = p.lds <= (size_t)kLdsBudget;
    return p;
}

int wgrad_splits(const DcnShape &s) {
    // workgroups = splits * groups * 9 * (Cg/TC) * (Og/TO); three fit a CU: aim at one full round of 768
    const int tc = 64, to = s.Og % 128 == 0 ? 128 : 64;
    const int tiles = s.groups * 9 * (s.Cg / tc) * (s.Og / to);
    int splits = 768 / tiles;
    const int nblk = (s.npos + 31) / 32;
    if (splits > nblk / 4) splits = nblk / 4;       // at least 4 K-steps per slice
    if (splits < 1) splits = 1;
    return splits;
}

struct DcnWorkspace { size_t wf, wd, slab, part, bins, entries, total; int splits; };

DcnWorkspace dcn_workspace(const DcnShape &s) {
    DcnWorkspace w;
    const size_t wbytes = (size_t)s.O * s.Cg * 9 * 4;
    w.splits = wgrad_splits(s);
    w.wf = 0;
    w.wd = wbytes;
    w.slab = 2 * wbytes;
    w.part = w.slab + (size_t)w.splits * wbytes;
    w.bins = w.part + (size_t)(s.C / 16) * 18 * s.npos * 4;
    const GatherPlan gp = gather_plan(s);
    const size_t HW = (size_t)s.H * s.W;
    w.entries = w.bins + (gp.ok ? (((size_t)s.B * 9 * (gp.rounds * HW + 1) * 4 + 15) & ~(size_t)15) : 0);
    w.total = w.entries + (gp.ok ? (size_t)s.B * 9 * HW * 4 * sizeof(PlanEntry) : 0);
    return w;
}

template <typename K>
int set_lds(K kernel, size_t bytes, const char *what) {
    if (bytes > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return mmt::fail((int)e, "%s: hipFuncSetAttribute(%zu bytes of LDS): %s", what, bytes, hipGetErrorString(e));
    }
    return 0;
}

// forward configurations: (waves along the pixels, waves along the output channels)
struct FwdCfg { int wm, wn; };
constexpr FwdCfg kFwdCfgs[] = {{3, 2}, {2, 2}, {4, 1}, {4, 2}};

size_t fwd_lds(int wm, int bn) { return (size_t)2 * (32 * wm + bn) * kLd * 4 + (size_t)9 * 32 * wm * sizeof(TapRec); }

// default: 2 x 2 waves on a 64-pixel x 128-channel tile, two workgroups per CU (measured at [24,512,16,44] and [12,512,32,88],
// tools/kbench_dcn.py: 267 / 452 us against 300-350 / 490-580 for the taller tiles -- the step's barrier, tap records and LDS
// fill are paid per workgroup-step, and smaller workgroups interleave them better)
int fwd_config(const DcnShape &, int) { return 1; }

}  // namespace

extern "C" int mmt_dcn_mfma_supported(int B, int H, int W, int C, int O, int groups) {
    DcnShape s;
    return dcn_shape(B, H, W, C, O, groups, &s, "dcn_mfma_supported", true) == 0 ? 1 : 0;
}

extern "C" int mmt_dcn_backward_form(int B, int H, int W, int C, int O, int groups) {
    DcnShape s;
    if (dcn_shape(B, H, W, C, O, groups, &s, "dcn_backward_form", true) != 0) return 0;
    return gather_plan(s).ok ? 2 : 1;
}

extern "C" int64_t mmt_dcn_mfma_workspace_bytes(int B, int H, int W, int C, int O, int groups) {
    DcnShape s;
    if (dcn_shape(B, H, W, C, O, groups, &s, "dcn_mfma_workspace_bytes", true) != 0) return 0;
    return (int64_t)dcn_workspace(s).total;
}

extern "C" int mmt_dcn_forward(int B, int H, int W, int C, int O, int groups, const float *x, const float *offset, const float *weight,
                               float *out, void *workspace, int64_t workspace_bytes, int fwd_config_override, void *stream) {
    MMT_REQUIRE_PTR(x);
    MMT_REQUIRE_PTR(offset);
    MMT_REQUIRE_PTR(weight);
    MMT_REQUIRE_PTR(out);
    MMT_REQUIRE_PTR(workspace);
    DcnShape s;
    if (int rc = dcn_shape(B, H, W, C, O, groups, &s, "dcn_forward", false)) return rc;
    if ((((uintptr_t)x | (uintptr_t)weight | (uintptr_t)out | (uintptr_t)workspace) & 15) != 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "dcn_forward: x / weight / out / workspace must be 16-byte aligned");
    const DcnWorkspace ws = dcn_workspace(s);
    if (workspace_bytes < (int64_t)ws.total)
        return mmt::fail(MMT_ERR_WORKSPACE, "dcn_forward: workspace too small (%lld < %lld bytes)", (long long)workspace_bytes, (long long)ws.total);
    hipStream_t st = (hipStream_t)stream;
    float *wf = reinterpret_cast<float *>((char *)workspace + ws.wf);
    mmt::TimedSeq seq;
    seq.launch(false, dcn_pack_weights, dim3(mmt::stream_grid((int64_t)s.O * s.Cg * 9, 256, 1024)), dim3(256), 0, st, s.O, s.Cg, s.groups, weight, wf,
               (float *)nullptr);
    const int bn = s.Og % 128 == 0 ? 128 : 64;
    const int cfg = fwd_config_override >= 1 && fwd_config_override <= 4 ? fwd_config_override - 1 : fwd_config(s, bn);
    const int wm = kFwdCfgs[cfg].wm, wn = kFwdCfgs[cfg].wn, bm = 32 * wm;
    const size_t lds = fwd_lds(wm, bn);
    const int tiles_per_group = ((s.npos + bm - 1) / bm) * (s.Og / bn);
    const unsigned grid = 8 % s.groups == 0 ? (unsigned)(8 * ((tiles_per_group + 8 / s.groups - 1) / (8 / s.groups))) : (unsigned)(tiles_per_group * s.groups);
#define MMT_DCN_FWD(WM, WN, BN)                                                                                               \
    {                                                                                                                         \
        if (int rc = set_lds(dcn_fwd_mfma<WM, WN, BN>, lds, "dcn_forward")) return rc;                                        \
        seq.launch(true, dcn_fwd_mfma<WM, WN, BN>, dim3(grid), dim3(WM * WN * 64), lds, st, H, W, C, O, groups, s.npos, x,    \
                   offset, (const float *)wf, out);                                                                           \
    }
    if (bn == 128) {
        if (cfg == 0) MMT_DCN_FWD(3, 2, 128) else if (cfg == 1) MMT_DCN_FWD(2, 2, 128) else if (cfg == 2) MMT_DCN_FWD(4, 1, 128) else MMT_DCN_FWD(4, 2, 128)
    } else {
        if (cfg == 0) MMT_DCN_FWD(3, 2, 64) else if (cfg == 1) MMT_DCN_FWD(2, 2, 64) else if (cfg == 2) MMT_DCN_FWD(4, 1, 64) else MMT_DCN_FWD(4, 2, 64)
    }
#undef MMT_DCN_FWD
    return mmt::check_launch("dcn_forward");
}

extern "C" int mmt_dcn_backward(int B, int H, int W, int C, int O, int groups, const float *x, const float *offset, const float *weight,
                                const float *grad_out, float *grad_x, float *grad_offset, float *grad_weight, void *workspace,
                                int64_t workspace_bytes, void *stream) {
    MMT_REQUIRE_PTR(x);
    MMT_REQUIRE_PTR(offset);
    MMT_REQUIRE_PTR(weight);
    MMT_REQUIRE_PTR(grad_out);
    MMT_REQUIRE_PTR(grad_x);
    MMT_REQUIRE_PTR(grad_offset);
    MMT_REQUIRE_PTR(grad_weight);
    MMT_REQUIRE_PTR(workspace);
    DcnShape s;
    if (int rc = dcn_shape(B, H, W, C, O, groups, &s, "dcn_backward", false)) return rc;
    if ((((uintptr_t)x | (uintptr_t)weight | (uintptr_t)grad_out | (uintptr_t)grad_x | (uintptr_t)workspace) & 15) != 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "dcn_backward: x / weight / grad_out / grad_x / workspace must be 16-byte aligned");
    const DcnWorkspace ws = dcn_workspace(s);
    if (workspace_bytes < (int64_t)ws.total)
        return mmt::fail(MMT_ERR_WORKSPACE, "dcn_backward: workspace too small (%lld < %lld bytes)", (long long)workspace_bytes, (long long)ws.total);
    hipStream_t st = (hipStream_t)stream;
    float *wd = reinterpret_cast<float *>((char *)workspace + ws.wd);
    float *slab = reinterpret_cast<float *>((char *)workspace + ws.slab);
    float *part = reinterpret_cast<float *>((char *)workspace + ws.part);
    const GatherPlan gp = gather_plan(s);
    const bool general = !gp.ok || (getenv("MMT_DCN_DGRAD_GENERAL") && atoi(getenv("MMT_DCN_DGRAD_GENERAL")));
    const DgradPlan dp = dgrad_plan(s);
    mmt::TimedSeq seq;
    seq.launch(false, dcn_pack_weights, dim3(mmt::stream_grid((int64_t)s.O * s.Cg * 9, 256, 1024)), dim3(256), 0, st, s.O, s.Cg, s.groups, weight,
               (float *)nullptr, wd);
    int32_t *bins = reinterpret_cast<int32_t *>((char *)workspace + ws.bins);
    PlanEntry *entries = reinterpret_cast<PlanEntry *>((char *)workspace + ws.entries);
    if (!general) {
        seq.launch(false, dcn_plan_taps, dim3(9, s.B), dim3(kPlanThreads), 0, st, H, W, gp.nw * 32, gp.rounds, offset, bins, entries);
    } else if (dp.bands > 1) {
        const hipError_t e = hipMemsetAsync(grad_x, 0, (size_t)s.npos * s.C * 4, st);
        if (e != hipSuccess) return mmt::fail((int)e, "dcn_backward: hipMemsetAsync: %s", hipGetErrorString(e));
    }
    // weight gradient
    {
        // 64 input x 128 output channels per workgroup (48 KB of LDS: three per CU) and 768 / tiles pixel slices measured best at
        // [24,512,16,44] g=4 (tools/scratch/dcn_wgrad_tiles.py: 128 x 128 / 512 slots +60 us)
        const int tc = 64, to = s.Og % 128 == 0 ? 128 : 64;
        const size_t lds = (size_t)2 * 32 * (tc + to) * 4 + 2 * 32 * sizeof(TapRec);
        const unsigned grid = (unsigned)(ws.splits * s.groups * 9 * (s.Cg / tc) * (s.Og / to));
#define MMT_DCN_WG(TC, TO)                                                                                                    \
    {                                                                                                                         \
        if (int rc = set_lds(dcn_wgrad_mfma<TC, TO>, lds, "dcn_backward")) return rc;                                         \
        seq.launch(false, dcn_wgrad_mfma<TC, TO>, dim3(grid), dim3(256), lds, st, H, W, C, O, groups, s.npos, ws.splits, x, offset, \
                   grad_out, slab);                                                                                           \
    }
        if (to == 128) MMT_DCN_WG(64, 128) else MMT_DCN_WG(64, 64)
#undef MMT_DCN_WG
        seq.launch(false, dcn_wgrad_reduce, dim3(mmt::stream_grid((int64_t)s.O * s.Cg * 9, 256, 2048)), dim3(256), 0, st, ws.splits, s.groups, s.Cg,
                   s.Og, (const float *)slab, grad_weight);
    }
    // data + offset gradient
    {
        if (!general) {
            const unsigned grid = (unsigned)(s.B * s.groups * (s.Cg / 16));
            if (s.Og == 128) {
                if (int rc = set_lds(dcn_dgrad_gather<128>, gp.lds, "dcn_backward")) return rc;
                seq.launch(false, dcn_dgrad_gather<128>, dim3(grid), dim3(gp.nw * 64), gp.lds, st, H, W, C, O, groups, s.npos, gp.rounds, x, offset,
                           (const float *)wd, grad_out, (const int32_t *)bins, (const PlanEntry *)entries, grad_x, part);
            } else {
                if (int rc = set_lds(dcn_dgrad_gather<64>, gp.lds, "dcn_backward")) return rc;
                seq.launch(false, dcn_dgrad_gather<64>, dim3(grid), dim3(gp.nw * 64), gp.lds, st, H, W, C, O, groups, s.npos, gp.rounds, x, offset,
                           (const float *)wd, grad_out, (const int32_t *)bins, (const PlanEntry *)entries, grad_x, part);
            }
        } else {
            const unsigned grid = (unsigned)(s.B * dp.bands * s.groups * (s.Cg / 16));
            if (s.Og == 128) {
                if (int rc = set_lds(dcn_dgrad_mfma<128>, dp.lds, "dcn_backward")) return rc;
                seq.launch(false, dcn_dgrad_mfma<128>, dim3(grid), dim3(dp.nw * 64), dp.lds, st, H, W, C, O, groups, s.npos, dp.bands, dp.band_rows, dp.halo,
                           x, offset, (const float *)wd, grad_out, grad_x, part);
            } else {
                if (int rc = set_lds(dcn_dgrad_mfma<64>, dp.lds, "dcn_backward")) return rc;
                seq.launch(false, dcn_dgrad_mfma<64>, dim3(grid), dim3(dp.nw * 64), dp.lds, st, H, W, C, O, groups, s.npos, dp.bands, dp.band_rows, dp.halo,
                           x, offset, (const float *)wd, grad_out, grad_x, part);
            }
        }
        seq.launch(true, dcn_offset_reduce_parts, dim3(mmt::stream_grid((int64_t)s.npos * 18, 256, 2048)), dim3(256), 0, st, s.npos, s.C / 16,
                   (const float *)part, grad_offset);
    }
    return mmt::check_launch("dcn_backward");
}
