// The task heads' FINAL convolutions (layers/heads/bev_depth_head.py; the reference's SeparateHead: per branch a ConvModule and then
// `Conv2d(64, classes, 3, padding=1, bias=True)` with classes = 1..3): 24 branches, each reading ITS 64 channels of one wide
// channels-last map [B, H, W, NB * 64] (the output of the fused first layer, `_forward_tasks_fused`) and writing 1..4 channels of one
// narrow map [B, H, W, KT] (KT = the branches' channel counts added up, 44 at the BASELINE configurations).  As dense convolutions
// these are 24 launches per direction of problems with N = 1..3 output channels -- a GEMM shape no matrix-core kernel likes (MIOpen:
// 25 / 26 / 36 us each, forward / data gradient / weight gradient, for 9.4 MFLOP per pixel row) -- and pure streaming: 402 MB of
// activations in, 11.5 MB out.  Here: lane = input channel, wave = (branch, image row, 32 pixels), a 3 x 3 window of the lane's channel
// slides along the row in registers (3 new loads per pixel), the 1..4 output sums meet by a wave reduction.  Three kernels:
//   thin_conv_fwd       out[p, off_j + k] = bias + sum_{tap, c} z[p + tap, j * 64 + c] * w[off_j + k][tap][c]
//   thin_conv_bwd_data  gz[p, j * 64 + c] = sum_{tap, k} gout[p - tap, off_j + k] * w[off_j + k][tap][c]
//   thin_conv_wgrad     gw[off_j + k][tap][c] = sum_p gout[p, off_j + k] * z[p + tap, j * 64 + c]   (+ gbias), per-workgroup partials
//                       in a fixed order + thin_conv_wgrad_reduce (no atomics: the same bits every run)
// Weights / bias / their gradients are fp32 ([KT][9][64] = the memory of a channels-last [KT, 64, 3, 3] parameter); activations
// fp32 or bf16 (fp32 arithmetic either way).
#include "mmt_common.h"

namespace {

constexpr int kThinSeg = 32;        // pixels of a row per wave
constexpr int kThinMaxBranches = 32;
constexpr int kThinMaxK = 4;
constexpr int kThinRowsPerBlock = 8;

struct ThinArgs {
    const void *z;
    const float *w, *bias;
    void *out;
    const void *gout;
    void *gz;
    float *partial, *gw, *gbias;
    int B, H, W, NB, KT, chunks;
    unsigned char off[kThinMaxBranches], k[kThinMaxBranches];
};

template <typename AT> __device__ __forceinline__ float ldact(const void *p, int64_t i);
template <> __device__ __forceinline__ float ldact<float>(const void *p, int64_t i) { return static_cast<const float *>(p)[i]; }
template <> __device__ __forceinline__ float ldact<bf16_t>(const void *p, int64_t i) {
    return __uint_as_float((unsigned)static_cast<const bf16_t *>(p)[i] << 16);
}
template <typename AT> __device__ __forceinline__ void stact(void *p, int64_t i, float v);
template <> __device__ __forceinline__ void stact<float>(void *p, int64_t i, float v) { static_cast<float *>(p)[i] = v; }
template <> __device__ __forceinline__ void stact<bf16_t>(void *p, int64_t i, float v) {
    static_cast<bf16_t *>(p)[i] = (bf16_t)(pack_bf16x2(v, 0.f) & 0xFFFFu);
}

// XCD-aware order of the (image row, segment) workgroups: consecutive workgroup ids land on different XCDs (id % 8), so XCD x takes
// the x-th EIGHTH of the rows as one contiguous band -- the three waves that read an input row (outputs y - 1, y, y + 1) then share
// one L2 instead of pulling the row into two
__device__ __forceinline__ int band_order(int id, int n) { return n % 8 == 0 ? (id % 8) * (n / 8) + id / 8 : id; }

// sum over the 64 lanes on the DPP path (row_shr 1, 2, 4, 8 inside the rows of 16 lanes, row_bcast:15 / :31 across them: lane 63
// ends with the total), handed to every lane through an SGPR
__device__ __forceinline__ float wave_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// The three input rows of an output row: base pointers and validity are the same for the whole wave (SGPRs); a lane's load is then
// `row + (xc * C + chan)` with a 32-bit offset -- the 64-bit address arithmetic per load was what bound the first version (VALU).
template <typename AT>
struct Rows {
    const AT *p[3];
    bool ok[3];
    int C, W;
    __device__ __forceinline__ Rows(const ThinArgs &a, int b, int y) {
        C = a.NB * 64; W = a.W;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = y + r - 1;
            ok[r] = yy >= 0 && yy < a.H;
            const int yc = yy < 0 ? 0 : (yy >= a.H ? a.H - 1 : yy);
            p[r] = static_cast<const AT *>(a.z) + (int64_t)(b * a.H + yc) * a.W * C;
        }
    }
    // the lane's channel at column xx of row r, zero outside the image (clamped address, unconditional load)
    __device__ __forceinline__ float at(int r, int xx, int chan) const {
        const bool in = ok[r] && xx >= 0 && xx < W;
        const int xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
        float v;
        if constexpr (sizeof(AT) == 4) v = p[r][xc * C + chan];
        else v = __uint_as_float((unsigned)p[r][xc * C + chan] << 16);
        return in ? v : 0.f;
    }
};

constexpr int kThinTrip = 4;        // pixels per trip of a wave: their 3 x 4 new window entries are requested together

// The bodies are compiled per number of output channels K of the branch (1..4; the kernel switches once per wave): the inner sums
// then hold exactly 9 K fused multiply-adds per pixel.  (With a run-time K the compiler computed all four channels and masked:
// 144 multiplies + 144 adds + 144 selects per trip of 4 pixels where 9 K x 4 fused operations do.)
template <int K>
__device__ __forceinline__ void load_weights(const ThinArgs &a, int off, int lane, float (&wr)[K][9]) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[k][t] = a.w[((off + k) * 9 + t) * 64 + lane];
}

template <typename AT, int K>
__device__ __forceinline__ void fwd_body(const ThinArgs &a, int b, int y, int x0, int x1, int off, int chan, int lane) {
    float wr[K][9], bk[K];
    load_weights<K>(a, off, lane, wr);
#pragma unroll
    for (int k = 0; k < K; ++k) bk[k] = a.bias ? a.bias[off + k] : 0.f;      // (in registers: a load per output in the loop was a round trip each)
    const Rows<AT> rows(a, b, y);
    float win[3][kThinTrip + 2], nxt[3][kThinTrip];                            // nxt: the next trip's new columns, requested one trip ahead
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        win[r][kThinTrip] = rows.at(r, x0 - 1, chan);
        win[r][kThinTrip + 1] = rows.at(r, x0, chan);
#pragma unroll
        for (int p = 0; p < kThinTrip; ++p) nxt[r][p] = rows.at(r, x0 + 1 + p, chan);
    }
    for (int x = x0; x < x1; x += kThinTrip) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            win[r][0] = win[r][kThinTrip];
            win[r][1] = win[r][kThinTrip + 1];
#pragma unroll
            for (int p = 0; p < kThinTrip; ++p) {
                win[r][2 + p] = nxt[r][p];
                nxt[r][p] = rows.at(r, x + kThinTrip + 1 + p, chan);          // (clamped inside: a trip past the segment reads valid memory)
            }
        }
#pragma unroll
        for (int p = 0; p < kThinTrip; ++p) {
            if (x + p >= x1) break;                                            // (uniform)
            const int64_t o = ((int64_t)(b * a.H + y) * a.W + x + p) * a.KT + off;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) s = fmaf(win[r][p + c], wr[k][r * 3 + c], s);
                s = wave_sum(s);
                if (lane == 0) stact<AT>(a.out, o + k, s + bk[k]);
            }
        }
    }
}

// grid: (B * H * segments, ceil(NB / 4)); wave = one branch
template <typename AT>
__global__ __launch_bounds__(256) void thin_conv_fwd(ThinArgs a) {
    const int lane = threadIdx.x & 63, j = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
    if (j >= a.NB) return;
    const int segs = (a.W + kThinSeg - 1) / kThinSeg;
    int t = band_order(blockIdx.x, gridDim.x);
    const int xs = t % segs; t /= segs;
    const int y = t % a.H, b = t / a.H;
    const int kj = a.k[j], off = a.off[j], chan = j * 64 + lane;
    const int x0 = xs * kThinSeg, x1 = min(x0 + kThinSeg, a.W);
    switch (kj) {
        case 1: fwd_body<AT, 1>(a, b, y, x0, x1, off, chan, lane); break;
        case 2: fwd_body<AT, 2>(a, b, y, x0, x1, off, chan, lane); break;
        case 3: fwd_body<AT, 3>(a, b, y, x0, x1, off, chan, lane); break;
        default: fwd_body<AT, 4>(a, b, y, x0, x1, off, chan, lane); break;
    }
}

// gz[p, chan] = sum over the 9 output pixels that read p, and the branch's K output channels.  The gradients of a trip's 4 pixels
// (3 rows x 6 columns x K values, the same for every lane) come in by TWO lane-parallel loads -- lane l fetches position l / 4 (and
// 16 + l / 4), channel l % 4 -- and reach the arithmetic through SGPRs (v_readlane); requested one trip ahead.
template <typename AT, int K>
__device__ __forceinline__ void bwd_data_body(const ThinArgs &a, int b, int y, int x0, int x1, int off, int chan, int lane) {
    float wr[K][9];
    load_weights<K>(a, off, lane, wr);
    auto fetch = [&](int x, float &va, float &vb) {
        const int k = lane & 3;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int pos = half * 16 + (lane >> 2);
            const int r = pos / (kThinTrip + 2), ci = pos - r * (kThinTrip + 2);
            const int yy = y - 1 + r, xx = x - 1 + ci;
            const bool in = pos < 3 * (kThinTrip + 2) && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W && k < K;
            const int yc = min(max(yy, 0), a.H - 1), xc = min(max(xx, 0), a.W - 1);
            const float v = ldact<AT>(a.gout, ((int64_t)(b * a.H + yc) * a.W + xc) * a.KT + off + (k < K ? k : 0));
            (half ? vb : va) = in ? v : 0.f;
        }
    };
    float na, nb;
    fetch(x0, na, nb);
    for (int x = x0; x < x1; x += kThinTrip) {
        const float va = na, vb = nb;
        fetch(x + kThinTrip, na, nb);                          // (clamped: always valid memory)
        float gw[3][kThinTrip + 2][K];                          // gout[y - 1 + r][x - 1 + ci][off + k]
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int ci = 0; ci < kThinTrip + 2; ++ci)
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int pos = r * (kThinTrip + 2) + ci;
                    gw[r][ci][k] = __int_as_float(pos < 16 ? __builtin_amdgcn_readlane(__float_as_int(va), pos * 4 + k)
                                                           : __builtin_amdgcn_readlane(__float_as_int(vb), (pos - 16) * 4 + k));
                }
#pragma unroll
        for (int p = 0; p < kThinTrip; ++p) {
            if (x + p >= x1) break;
            float g = 0.f;
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)                      // out[yy, xx] read in[yy + ty - 1, xx + tx - 1]
#pragma unroll
                for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                    for (int k = 0; k < K; ++k) g = fmaf(gw[2 - ty][p + 2 - tx][k], wr[k][ty * 3 + tx], g);
            stact<AT>(a.gz, ((int64_t)(b * a.H + y) * a.W + x + p) * ((int64_t)a.NB * 64) + chan, g);
        }
    }
}

template <typename AT>
__global__ __launch_bounds__(256) void thin_conv_bwd_data(ThinArgs a) {
    const int lane = threadIdx.x & 63, j = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
    if (j >= a.NB) return;
    const int segs = (a.W + kThinSeg - 1) / kThinSeg;
    int t = band_order(blockIdx.x, gridDim.x);
    const int xs = t % segs; t /= segs;
    const int y = t % a.H, b = t / a.H;
    const int kj = a.k[j], off = a.off[j], chan = j * 64 + lane;
    const int x0 = xs * kThinSeg, x1 = min(x0 + kThinSeg, a.W);
    switch (kj) {
        case 1: bwd_data_body<AT, 1>(a, b, y, x0, x1, off, chan, lane); break;
        case 2: bwd_data_body<AT, 2>(a, b, y, x0, x1, off, chan, lane); break;
        case 3: bwd_data_body<AT, 3>(a, b, y, x0, x1, off, chan, lane); break;
        default: bwd_data_body<AT, 4>(a, b, y, x0, x1, off, chan, lane); break;
    }
}

// one image row of one branch: acc[k][tap] += gout[p, off + k] * z[p + tap, chan] over the row's pixels (+ the bias sums)
template <typename AT, int K>
__device__ __forceinline__ void wgrad_row(const ThinArgs &a, int q, int off, int chan, int lane, float (&acc)[kThinMaxK][9], float (&bacc)[kThinMaxK]) {
    const int b = q / a.H, y = q % a.H;
    const Rows<AT> rows(a, b, y);
    auto fetch_g = [&](int x) {
        const int p = (lane >> 2) & (kThinTrip - 1), k = lane & 3;            // (lanes 0..15 matter)
        const bool in = x + p < a.W && k < K;
        const float v = ldact<AT>(a.gout, ((int64_t)q * a.W + (x + p < a.W ? x + p : a.W - 1)) * a.KT + off + (k < K ? k : 0));
        return in ? v : 0.f;
    };
    float win[3][kThinTrip + 2], nxt[3][kThinTrip], ng = fetch_g(0);            // one trip ahead
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        win[r][kThinTrip] = rows.at(r, -1, chan);
        win[r][kThinTrip + 1] = rows.at(r, 0, chan);
#pragma unroll
        for (int p = 0; p < kThinTrip; ++p) nxt[r][p] = rows.at(r, 1 + p, chan);
    }
    for (int x = 0; x < a.W; x += kThinTrip) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            win[r][0] = win[r][kThinTrip];
            win[r][1] = win[r][kThinTrip + 1];
#pragma unroll
            for (int p = 0; p < kThinTrip; ++p) {
                win[r][2 + p] = nxt[r][p];
                nxt[r][p] = rows.at(r, x + kThinTrip + 1 + p, chan);
            }
        }
        const float vv = ng;
        ng = fetch_g(x + kThinTrip);
#pragma unroll
        for (int p = 0; p < kThinTrip; ++p)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vv), p * 4 + k));
                bacc[k] += g;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[k][r * 3 + c] = fmaf(g, win[r][p + c], acc[k][r * 3 + c]);
            }
    }
}

// grid: (chunks of kThinRowsPerBlock image rows, NB); the 4 waves of a workgroup take every 4th row of the chunk, whole rows
template <typename AT>
__global__ __launch_bounds__(256) void thin_conv_wgrad(ThinArgs a) {
    __shared__ float red[4][kThinMaxK * 9 + kThinMaxK][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = blockIdx.y;
    const int kj = a.k[j], off = a.off[j], chan = j * 64 + lane;
    float acc[kThinMaxK][9], bacc[kThinMaxK];
#pragma unroll
    for (int k = 0; k < kThinMaxK; ++k) {
        bacc[k] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[k][t] = 0.f;
    }
    const int rows = a.B * a.H;
    for (int q = blockIdx.x * kThinRowsPerBlock + wave; q < min((int)(blockIdx.x + 1) * kThinRowsPerBlock, rows); q += 4) {
        switch (kj) {
            case 1: wgrad_row<AT, 1>(a, q, off, chan, lane, acc, bacc); break;
            case 2: wgrad_row<AT, 2>(a, q, off, chan, lane, acc, bacc); break;
            case 3: wgrad_row<AT, 3>(a, q, off, chan, lane, acc, bacc); break;
            default: wgrad_row<AT, 4>(a, q, off, chan, lane, acc, bacc); break;
        }
    }
#pragma unroll
    for (int k = 0; k < kThinMaxK; ++k) {
#pragma unroll
        for (int t = 0; t < 9; ++t) red[wave][k * 9 + t][lane] = acc[k][t];
        red[wave][kThinMaxK * 9 + k][lane] = bacc[k];
    }
    __syncthreads();
    // partial[(j * chunks + chunk)][40][64]: rows 0..35 = (k, tap) x channel, rows 36..39 = the bias sums (every lane holds the same)
    float *p = a.partial + ((int64_t)j * a.chunks + blockIdx.x) * ((kThinMaxK * 9 + kThinMaxK) * 64);
    for (int i = threadIdx.x; i < (kThinMaxK * 9 + kThinMaxK) * 64; i += 256) {
        const int row = i >> 6, l = i & 63;
        p[i] = (red[0][row][l] + red[1][row][l]) + (red[2][row][l] + red[3][row][l]);
    }
}

// one thread per element of gw ([KT][9][64]) and of gbias ([KT]): the chunks' partials in order
__global__ __launch_bounds__(256) void thin_conv_wgrad_reduce(ThinArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nw = a.KT * 576;
    if (i >= nw + a.KT) return;
    const int kk = i < nw ? i / 576 : i - nw;               // output channel
    int j = 0;
    while (j + 1 < a.NB && a.off[j + 1] <= kk) ++j;
    const int k = kk - a.off[j];
    const int row = i < nw ? k * 9 + (i % 576) / 64 : kThinMaxK * 9 + k, l = i < nw ? i & 63 : 0;
    const float *p = a.partial + (int64_t)j * a.chunks * ((kThinMaxK * 9 + kThinMaxK) * 64) + row * 64 + l;
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < a.chunks; ++c) s += p[(int64_t)c * ((kThinMaxK * 9 + kThinMaxK) * 64)];
    if (i < nw) a.gw[i] = s;
    else a.gbias[kk] = s;
}

int fill(const char *who, ThinArgs *a, int B, int H, int W, int NB, const unsigned char *k_host) {
    if (B < 1 || H < 1 || W < 1 || NB < 1 || NB > kThinMaxBranches) return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: B, H, W >= 1 and 1 <= branches <= %d", who, kThinMaxBranches);
    if (!k_host) return mmt::fail(MMT_ERR_NULL_POINTER, "%s: k_host is NULL", who);
    if ((int64_t)B * H * W * NB * 64 >= (1ll << 40)) return mmt::fail(MMT_ERR_TOO_LARGE, "%s: map too large", who);
    int kt = 0;
    for (int j = 0; j < NB; ++j) {
        if (k_host[j] < 1 || k_host[j] > kThinMaxK) return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: branch %d has %d output channels (1..%d)", who, j, (int)k_host[j], kThinMaxK);
        a->off[j] = (unsigned char)kt;
        a->k[j] = k_host[j];
        kt += k_host[j];
    }
    a->B = B; a->H = H; a->W = W; a->NB = NB; a->KT = kt;
    a->chunks = (int)mmt::ceil_div((int64_t)B * H, kThinRowsPerBlock);
    return 0;
}

}  // namespace

extern "C" int64_t mmt_heads_final_workspace_elems(int B, int H, int NB) {
    if (B < 1 || H < 1 || NB < 1 || NB > kThinMaxBranches) return -1;
    return (int64_t)NB * mmt::ceil_div((int64_t)B * H, kThinRowsPerBlock) * ((kThinMaxK * 9 + kThinMaxK) * 64);
}

extern "C" int mmt_heads_final_forward(int B, int H, int W, int NB, const unsigned char *k_host, const void *z, const float *weight,
                                       const float *bias, void *out, int act_dtype, void *stream) {
    MMT_REQUIRE_PTR(z);
    MMT_REQUIRE_PTR(weight);
    MMT_REQUIRE_PTR(out);
    if (act_dtype != MMT_DTYPE_F32 && act_dtype != MMT_DTYPE_BF16) return mmt::fail(MMT_ERR_BAD_FLAG, "heads_final_forward: unknown activation dtype %d", act_dtype);
    ThinArgs a = {};
    if (int rc = fill("heads_final_forward", &a, B, H, W, NB, k_host)) return rc;
    a.z = z; a.w = weight; a.bias = bias; a.out = out;
    const dim3 grid(B * H * (int)mmt::ceil_div(W, kThinSeg), (int)mmt::ceil_div(NB, 4));
    if (act_dtype == MMT_DTYPE_F32) hipLaunchKernelGGL(thin_conv_fwd<float>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(thin_conv_fwd<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return mmt::check_launch("heads_final_forward");
}

extern "C" int mmt_heads_final_backward(int B, int H, int W, int NB, const unsigned char *k_host, const void *z, const float *weight,
                                        const void *grad_out, void *grad_z, float *grad_weight, float *grad_bias, float *workspace,
                                        int act_dtype, void *stream) {
    MMT_REQUIRE_PTR(z);
    MMT_REQUIRE_PTR(weight);
    MMT_REQUIRE_PTR(grad_out);
    if (act_dtype != MMT_DTYPE_F32 && act_dtype != MMT_DTYPE_BF16) return mmt::fail(MMT_ERR_BAD_FLAG, "heads_final_backward: unknown activation dtype %d", act_dtype);
    if ((grad_weight || grad_bias) && !(grad_weight && grad_bias && workspace))
        return mmt::fail(MMT_ERR_NULL_POINTER, "heads_final_backward: grad_weight, grad_bias and workspace come together");
    ThinArgs a = {};
    if (int rc = fill("heads_final_backward", &a, B, H, W, NB, k_host)) return rc;
    a.z = z; a.w = weight; a.gout = grad_out; a.gz = grad_z; a.gw = grad_weight; a.gbias = grad_bias; a.partial = workspace;
    hipStream_t st = (hipStream_t)stream;
    if (grad_z) {
        const dim3 grid(B * H * (int)mmt::ceil_div(W, kThinSeg), (int)mmt::ceil_div(NB, 4));
        if (act_dtype == MMT_DTYPE_F32) hipLaunchKernelGGL(thin_conv_bwd_data<float>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(thin_conv_bwd_data<bf16_t>, grid, dim3(256), 0, st, a);
        if (int rc = mmt::check_launch("heads_final_backward(data)")) return rc;
    }
    if (grad_weight) {
        const dim3 grid(a.chunks, NB);
        if (act_dtype == MMT_DTYPE_F32) hipLaunchKernelGGL(thin_conv_wgrad<float>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(thin_conv_wgrad<bf16_t>, grid, dim3(256), 0, st, a);
        if (int rc = mmt::check_launch("heads_final_backward(weights)")) return rc;
        hipLaunchKernelGGL(thin_conv_wgrad_reduce, dim3((int)mmt::ceil_div(a.KT * 576 + a.KT, 256)), dim3(256), 0, st, a);
        if (int rc = mmt::check_launch("heads_final_backward(reduce)")) return rc;
    }
    return MMT_OK;
}
