// Fused lift-splat backward, COLUMN form on the matrix cores for MI355X (gfx950).
//
//   grad_depth[t]        = < grad_out[cell(t), :], context[pix(t), :] >
//   grad_context[pix, :] = sum over the pixel's depth bins of depth[t] * grad_out[cell(t), :]
//
// The pixels of one image column at one depth bin lie above each other in the world: for a level camera they share their
// BEV cell.  Take the column's 16 pixels x D bins and let G[d, :] be the BEV-gradient row of bin d's cell; then
//   grad_depth  [16 x D] = Ctx [16 x C] * G^T [C x D]          grad_context [16 x C] = Dep [16 x D] * G [D x C]
// two small fp32 GEMMs per column that read every gradient row ONCE per column (the ray walk reads it once per pixel,
// 16 cycles of the texture path each) and need no cross-lane reductions.  v_mfma_f32_16x16x4_f32 is exact fp32 (a k-ordered
// fmaf chain), so the result meets the same tolerance as the walk.
//
// Which cell "the column" has at a bin is decided from the data: ref[d] = the smallest row offset among the kept pixels of
// the bin.  Pixels whose own cell differs (a camera that is not level, arbitrary geometry) are MISMATCHES: they are left out
// of the GEMMs (depth 0, no grad_depth store) and handled one by one afterwards -- correct for any geometry, fast for a
// frustum.  A workgroup = (camera, column, block of 16 image rows), 2 waves; a wave owns every other batch of 16 bins.
#include "mmt_camera.h"

namespace {

constexpr int kColBlock = 128;
constexpr int kBins = 16;                       // bins per batch = N of the grad_depth tile = K of the grad_context product
constexpr unsigned kOut = 0x80000000u;          // row offset of a dropped point: beyond num_records (< 2^30), loads zeros

struct ColArgs {
    int BN, N, D, fH, fW, C, nx, ny, nz;
    int pm, rblocks;
    int split, seg;               // column segments per camera, columns per segment (workgroup -> XCD balance, see the kernel)
    int vec;                      // grad_depth may leave as 16-byte (fp32) / 8-byte (bf16) vectors: pixel-major, D % 4 == 0, aligned
    const int32_t *geom;          // geom form; camera form (template CAM): the cell comes from the matrix (mmt_camera.h)
    const float *combine, *fu, *fv, *fd;
    mmt::CamGrid q;
    const int2 *summary;          // camera form, nullable: the forward's column summary (mmt_camera.h CamGeom::summary)
    unsigned long long *stats;    // nullable: [0] += points handled by the mismatch pass, [1] += kept points (cumulative)
    const void *depth;
    const void *context;
    const float *grad_out;
    int64_t sb, sy, sx;
    int span_bytes;
    void *grad_depth;
    float *grad_context;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int64_t col_point(const ColArgs &a, int bn, int row, int col, int d) {
    return a.pm ? (((int64_t)bn * a.fH + row) * a.fW + col) * a.D + d : (((int64_t)bn * a.D + d) * a.fH + row) * a.fW + col;
}
template <typename FT> __device__ __forceinline__ float ld_scalar(const FT *p);
template <> __device__ __forceinline__ float ld_scalar<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ld_scalar<bf16_t>(const bf16_t *p) { return __uint_as_float((unsigned)*p << 16); }
template <typename FT> __device__ __forceinline__ void st_scalar(FT *p, float v);
template <> __device__ __forceinline__ void st_scalar<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_scalar<bf16_t>(bf16_t *p, float v) { *p = (bf16_t)(pack_bf16x2(v, 0.f) & 0xFFFFu); }

__device__ __forceinline__ float quad_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
    return v;
}

// LDS (dynamic): dep [16][Dp] fp32 | ref [Dp] int | flag [Dp] words (2 bits per row) | G tiles of the 2 waves [2][16][CP] fp32
// (20 KB at D = 112, C = 80; the kernel takes 156 VGPRs -- the geometry phase's 16-row arrays -- so a CU holds six workgroups
// of two waves: 1 536 slots, every workgroup of the cfg4 launch (1 056) is resident at once, the 2 112 of configs[4] take two
// rounds.  Capping the registers at 128 spills and was slower; round 2 also kept every point's own row offset in LDS, 7 KB
// that only the mismatch pass reads: it recomputes them now)
// (CP = C + 4; after the products: the grad_context total and the lists / partial sums of the mismatch pass)
// NT = C / 16 (N tiles of the grad_context product); C / 4 <= 64 lanes move one row.
template <typename FT, int NT, bool CAM>
#ifndef COL_MIN_WAVES
#define COL_MIN_WAVES 1
#endif
__global__ __launch_bounds__(kColBlock, COL_MIN_WAVES) void lss_col_bwd(ColArgs a) {
    extern __shared__ __align__(16) unsigned char col_lds[];
    constexpr int C = 16 * NT, C4 = C / 4, CP = C + 4;
    constexpr int NV = (16 * C4) / 64;                       // 16-byte vectors of a G tile per lane
    // Workgroup -> (camera, column, row block).  Workgroup i runs on XCD i % 8; a camera's neighbouring columns read neighbouring
    // gradient rows, so they share an XCD -- but the XCDs must also carry the same number of workgroups: with BN cameras dealt
    // whole, BASELINE configs[4] (12 cameras) put two cameras on four XCDs and one on the other four (352 against 176 workgroups;
    // the kernel lasted as long as the loaded half).  A camera is cut into `split` segments of columns so that the units
    // (camera, segment) are a multiple of 8 wherever that is possible: 12 cameras -> 24 half cameras, three per XCD.
    const int L = blockIdx.x, xcd = L & 7, i0 = L >> 3;
    const int per = a.seg * a.rblocks;
    const int q = i0 / per, r = i0 - q * per;
    const int unit = q * 8 + xcd;
    if (unit >= a.BN * a.split) return;
    const int bn = unit / a.split, sg = unit - bn * a.split;
    const int col = sg * a.seg + r / a.rblocks, rb = r % a.rblocks;
    if (col >= a.fW) return;
    const int row0 = rb * 16;
    const int nrow = (a.fH - row0) < 16 ? (a.fH - row0) : 16;
    const int D = a.D, HW = a.fH * a.fW;
    const int nb = (D + kBins - 1) / kBins, Dp = nb * kBins;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = bn / a.N;
    const FT *depth = reinterpret_cast<const FT *>(a.depth);
    const FT *context = reinterpret_cast<const FT *>(a.context);
    FT *grad_depth = reinterpret_cast<FT *>(a.grad_depth);

    float *dep = reinterpret_cast<float *>(col_lds);                   // [16][Dp] depth value (0 for dropped points)
    int *ref = reinterpret_cast<int *>(dep + 16 * Dp);                 // [Dp] byte offset of the COLUMN's row per bin, or kOut
    unsigned *flag = reinterpret_cast<unsigned *>(ref + Dp);           // [Dp] bits [2 u, 2 u + 2) of a bin's word: row u is 0 dropped, 1 in the column's cell, 2 mismatch
    float *gw0 = reinterpret_cast<float *>(flag + Dp);                 // (ONE word per bin: sixteen byte stores per thread kept the CU's LDS port busy -- twelve waves do phase A at once)
    auto flag_of = [&](int row, int bin) __attribute__((always_inline)) { return (int)((flag[bin] >> (2 * row)) & 3u); };
    float *gw1 = gw0 + 16 * CP;
    float *gw = wave == 0 ? gw0 : gw1;
    unsigned *own = reinterpret_cast<unsigned *>(gw1 + 16 * CP);       // [16][Dp] a pixel's OWN row offset, written where a block's rows are looked at one by one:
                                                                       // what the mismatch pass walks (it recomputed the geometry per entry -- a load of the bin's
                                                                       // depth coordinate and two divisions in front of every gradient row: 12 of 28 us at 0.5 degrees of pitch)
    __shared__ int nmis, nkept;
    if (tid == 0) { nmis = 0; nkept = 0; }
    __syncthreads();
#ifdef LSS_STAMPS   // diagnostic build: grad_context receives 4 s_memtime stamps per workgroup instead of its rows
    unsigned long long *cstamps = reinterpret_cast<unsigned long long *>(a.grad_context) + (int64_t)blockIdx.x * 8;
#define COL_STAMP(i) do { if (threadIdx.x == 0) cstamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define COL_STAMP(i) do { } while (0)
#endif
    COL_STAMP(0);

    // ---- the block's 16 context rows -> wave 0's G tile (free until the products start), as whole 16-byte vectors; the A
    // operand of the grad_depth product is read out of it after phase A's barrier (round 2: 20 strided dword loads per lane).
    // The loads are ISSUED here and parked in LDS behind phase A (round 5 waited for them in front of it: a memory round trip
    // or two before phase A's own loads were even issued -- 3 of the workgroup's 14 us in the stamps)
    constexpr int VEC = sizeof(FT) == 2 ? 8 : 4, CV = C / VEC, NCI = (16 * CV + kColBlock - 1) / kColBlock;
    uint4 creg[NCI];
#pragma unroll
    for (int i = 0; i < NCI; ++i) {
        const int e = tid + i * kColBlock, ec = e < 16 * CV ? e : 16 * CV - 1;
        const int prow = ec / CV, cvi = ec - prow * CV;
        creg[i] = *reinterpret_cast<const uint4 *>(context + ((int64_t)bn * HW + (row0 + (prow < nrow ? prow : 0)) * a.fW + col) * C + cvi * VEC);
    }
    auto park_context = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NCI; ++i) {
            const int e = tid + i * kColBlock;
            if (e < 16 * CV) {
                const int prow = e / CV, cvi = e - prow * CV;
                float *dst = gw0 + prow * CP + cvi * VEC;
                const uint4 r = creg[i];
                if constexpr (sizeof(FT) == 2) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(bf16_lo(r.x), bf16_hi(r.x), bf16_lo(r.y), bf16_hi(r.y));
                    *reinterpret_cast<float4 *>(dst + 4) = make_float4(bf16_lo(r.z), bf16_hi(r.z), bf16_lo(r.w), bf16_hi(r.w));
                } else {
                    *reinterpret_cast<float4 *>(dst) = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
                }
            }
        }
    };

    // ---- phase A: a thread takes one bin of all 16 rows (coalesced along the bins in the pixel-major order), all 32 loads in
    // flight at once: kept test, the column's cell = the smallest row offset among the kept pixels, who shares it, who does not
    const int64_t tcol = col_point(a, bn, row0, col, 0);               // point (row0, col, bin 0); rows / bins are strides away
    const int64_t rstep = a.pm ? (int64_t)a.fW * D : a.fW;
    const int64_t dstep = a.pm ? 1 : (int64_t)HW;
    // phase A's loads of this thread's first (at D <= 128: only) bin go out FIRST: the camera operands below are waited for before
    // they are used, and with the depths behind them that wait was a memory round trip in front of phase A's own (stamps: 2.4 us)
    float dv0[16];
    int2 sv0 = make_int2(-1, 0);
    {
        const int64_t tb = tcol + (tid < D ? tid : 0) * dstep;
#pragma unroll
        for (int u = 0; u < 16; ++u) dv0[u] = ld_scalar<FT>(depth + tb + (u < nrow ? u : 0) * rstep);
        if constexpr (CAM) {
            if (a.summary) sv0 = a.summary[(((int64_t)bn * a.rblocks + rb) * a.fW + col) * D + (tid < D ? tid : 0)];
        }
    }
    float cm[12], cv[16];
    float cu = 0.f;
    bool cv_sorted = false;
    if constexpr (CAM) {       // camera form: the column's cells from the matrix (mmt_camera.h: end-row shortcut, exact range thresholds)
#pragma unroll
        for (int k = 0; k < 12; ++k) cm[k] = a.combine[bn * 16 + k];
        cu = a.fu[col];
        // the block's 16 row coordinates: ONE vector load (lane u takes row u's) handed out with v_readlane -- sixteen uniform loads
        // compile to sixteen s_load_dword, each waited for before the next is issued (2 us in front of phase A's loads in the stamps)
        const int fvl = __float_as_int(a.fv[row0 + ((lane & 15) < nrow ? (lane & 15) : nrow - 1)]);
#pragma unroll
        for (int u = 0; u < 16; ++u) cv[u] = __int_as_float(__builtin_amdgcn_readlane(fvl, u));
        cv_sorted = mmt_rows_sorted<16>(cv, nrow);
    }
    int tmis = 0, tkept = 0;
    for (int bin = tid; bin < Dp; bin += kColBlock) {
        int gx[16], gy[16], gz[16];
        float dv[16];
        const int64_t tb = tcol + (bin < D ? bin : 0) * dstep;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int64_t t = tb + (u < nrow ? u : 0) * rstep;
            if constexpr (!CAM) { gx[u] = a.geom[t * 3]; gy[u] = a.geom[t * 3 + 1]; gz[u] = a.geom[t * 3 + 2]; }
            dv[u] = bin == tid ? dv0[u] : ld_scalar<FT>(depth + t);
        }
#ifdef LSS_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); COL_STAMP(4);
#endif
        unsigned o[16];
        unsigned m = kOut;
        if constexpr (CAM) {
            // (the column's geometry is made only where it is used: with the forward's summary and a level camera nothing of it runs)
            auto column = [&]() __attribute__((always_inline)) { return mmt_cam_column_make(cm, cu, a.fd[bin < D ? bin : 0]); };
            bool uniform, in0;
            int x0, y0;
            unsigned zmask;
            if (a.summary) {      // written by the forward: two dwords per bin instead of the geometry
                const int2 sv = bin == tid ? sv0 : a.summary[(((int64_t)bn * a.rblocks + rb) * a.fW + col) * D + (bin < D ? bin : 0)];
                in0 = sv.x >= 0; x0 = sv.x & 0xFFFF; y0 = sv.x >> 16;
                zmask = (unsigned)sv.y & 0xFFFFu;
                uniform = __all((sv.y & mmt::kSummaryUniform) != 0);
            } else {
                zmask = mmt_cam_column_cells<16>(column(), cv, nrow, cv_sorted, a.q, a.nx, a.ny, a.nz, uniform, in0, x0, y0);
            }
            if (bin >= D) zmask = 0u;
            if (uniform) {
                // every wave of a level camera comes here: the rows that pass z all sit in the column's cell -- nobody mismatches, and what
                // the general tail below works out per row (sixteen offsets, comparisons, flags: 250 VALU operations a thread, with three
                // waves per SIMD at it together 4 k cycles in the stamps) is a bit of zmask
                const unsigned zm = in0 ? zmask : 0u;
                ref[bin] = zm ? (int)(unsigned)(((int)(b * a.sb) + y0 * (int)a.sy + x0 * (int)a.sx) * 4) : (int)kOut;
                unsigned fw = zm;                                   // bit u -> bits [2 u, 2 u + 2) = 1
                fw = (fw | (fw << 8)) & 0x00FF00FFu; fw = (fw | (fw << 4)) & 0x0F0F0F0Fu; fw = (fw | (fw << 2)) & 0x33333333u; fw = (fw | (fw << 1)) & 0x55555555u;
                flag[bin] = fw;
#pragma unroll
                for (int u = 0; u < 16; ++u) dep[u * Dp + bin] = ((zm >> u) & 1u) ? dv[u] : 0.f;
                tkept += __popc(zm);
                continue;
            } else {
                const mmt_cam_column cc = column();
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    int x, y;
                    const bool in = mmt_cam_row_xy(cc, cv[u], a.q, a.nx, a.ny, x, y);
                    o[u] = (in && ((zmask >> u) & 1u)) ? (unsigned)(((int)(b * a.sb) + y * (int)a.sy + x * (int)a.sx) * 4) : kOut;
                    m = o[u] < m ? o[u] : m;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                // in range <=> (unsigned)v < n for all three, one comparison each; no branches
                const bool keep = (u < nrow) & (bin < D) & ((unsigned)gx[u] < (unsigned)a.nx) & ((unsigned)gy[u] < (unsigned)a.ny) & ((unsigned)gz[u] < (unsigned)a.nz);
                o[u] = keep ? (unsigned)(((int)(b * a.sb) + gy[u] * (int)a.sy + gx[u] * (int)a.sx) * 4) : kOut;
                m = o[u] < m ? o[u] : m;
            }
        }
        ref[bin] = (int)m;
        int mis = 0, kept = 0;
        unsigned fword = 0u;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int f = o[u] == kOut ? 0 : (o[u] == m ? 1 : 2);
            own[u * Dp + bin] = o[u];
            fword |= (unsigned)f << (2 * u);
            dep[u * Dp + bin] = f ? dv[u] : 0.f;
            mis += f == 2;
            kept += f != 0;
        }
        flag[bin] = fword;
        tmis += mis; tkept += kept;
    }
    COL_STAMP(7);
    {   // one LDS atomic per wave and counter (every lane adding to the same word is 64 serialised read-modify-writes)
        int pk = (tmis << 16) | tkept;                                  // (<= 7 * 16 points per lane, 64 lanes: 13 bits each)
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) pk += __shfl_xor(pk, m);
        if (lane == 0) {
            if (pk >> 16) atomicAdd(&nmis, pk >> 16);
            if (a.stats && (pk & 0xFFFF)) atomicAdd(&nkept, pk & 0xFFFF);
        }
    }
    park_context();
    COL_STAMP(5);
    __syncthreads();
    COL_STAMP(6);
    // Ctx[pixel = lane & 15][channel C4 * (lane >> 4) + j], in registers for the rest of the kernel.  The products sum over all
    // channels, so which of them a (register, lane quarter) pair stands for is free as long as both operands agree: a quarter
    // takes C4 CONSECUTIVE channels and both operands are read from LDS as 16-byte vectors (round 5: channel 4 j + quarter, a
    // dword per read: 20 reads per operand and batch; the rows' stride of C + 4 floats keeps eight lanes' vectors on distinct banks)
    static_assert(C4 % 4 == 0, "16-byte operand reads");
    float ctxA[C4];
    {
        const int prow = lane & 15;
#pragma unroll
        for (int j = 0; j < C4; j += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(gw0 + prow * CP + C4 * (lane >> 4) + j);
            ctxA[j] = prow < nrow ? v.x : 0.f; ctxA[j + 1] = prow < nrow ? v.y : 0.f; ctxA[j + 2] = prow < nrow ? v.z : 0.f; ctxA[j + 3] = prow < nrow ? v.w : 0.f;
        }
    }
    __syncthreads();                      // wave 0 stages its first G tile into gw0 next
    COL_STAMP(1);

    // ---- the two products, a batch of 16 bins at a time
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.grad_out), 0, a.span_bytes, 0x00020000);
    f32x4 accC[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) accC[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mmt_u32x4 pre[NV];
    auto fetch = [&](int bb) __attribute__((always_inline)) {           // the batch's 16 rows -> registers
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = lane + 64 * v;
            const int rr = idx / C4, c4 = idx - rr * C4;
            pre[v] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)ref[bb * kBins + rr] + c4 * 16, 0, 0);
        }
    };
    if (wave < nb) fetch(wave);
    for (int bb = wave; bb < nb; bb += kColBlock / 64) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = lane + 64 * v;
            const int rr = idx / C4, c4 = idx - rr * C4;
            *reinterpret_cast<mmt_u32x4 *>(gw + rr * CP + c4 * 4) = pre[v];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (bb + kColBlock / 64 < nb) fetch(bb + kColBlock / 64);       // the next batch's rows are on their way during the MFMAs
        // grad_depth tile, TRANSPOSED [16 bins x 16 pixels] = G [16 x C] * Ctx^T: K = C.  The operands of the two matrices have the same
        // lane layout (row / column = lane & 15, k = lane >> 4), so which one is "A" is free -- and with the bins as rows a lane ends
        // up with four consecutive BINS of one pixel: its 16 bytes of grad_depth, without the turn through LDS round 5 made
        // (4 LDS writes, a read and two wave barriers per batch).  Same products, same k order: the same bits.
        f32x4 accD = {0.f, 0.f, 0.f, 0.f}, accD2 = {0.f, 0.f, 0.f, 0.f};
        {
            const float *gB = gw + (lane & 15) * CP + C4 * (lane >> 4);
            float bv[C4];
#pragma unroll
            for (int j = 0; j < C4; j += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(gB + j);
                bv[j] = v.x; bv[j + 1] = v.y; bv[j + 2] = v.z; bv[j + 3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < C4; j += 2) {
                accD = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j], ctxA[j], accD, 0, 0, 0);
                accD2 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j + 1], ctxA[j + 1], accD2, 0, 0, 0);
            }
        }
        // grad_context partial [16 pixels x C]: K = the batch's 16 bins
        {
            const float *dA = dep + (lane & 15) * Dp + bb * kBins + (lane >> 4);
            const unsigned *fA = flag + bb * kBins + (lane >> 4);
            // N tile n of the product <-> channels: the tiles of a full group of four interleaved -- tile 4 g + i is channel
            // 64 g + 4 (lane & 15) + i, so a lane reads its four tiles' B elements as ONE 16-byte vector --, a leftover tile
            // (C = 80: the fifth) plain: channel 64 g + (lane & 15)
            const float *gC = gw + (lane >> 4) * CP;
            float av[4], cv[4][NT];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                av[j] = ((fA[4 * j] >> (2 * (lane & 15))) & 3u) == 1u ? dA[4 * j] : 0.f;      // mismatches are left to the pass below
#pragma unroll
                for (int g4 = 0; g4 < NT / 4; ++g4) {
                    const float4 v = *reinterpret_cast<const float4 *>(gC + 4 * j * CP + 64 * g4 + 4 * (lane & 15));
                    cv[j][4 * g4] = v.x; cv[j][4 * g4 + 1] = v.y; cv[j][4 * g4 + 2] = v.z; cv[j][4 * g4 + 3] = v.w;
                }
#pragma unroll
                for (int n = 4 * (NT / 4); n < NT; ++n) cv[j][n] = gC[4 * j * CP + 64 * (NT / 4) + 16 * (n - 4 * (NT / 4)) + (lane & 15)];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int n = 0; n < NT; ++n) accC[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], cv[j][n], accC[n], 0, 0, 0);
        }
        // grad_depth: the lane holds pixel lane & 15, bins bb * 16 + 4 * (lane >> 4) .. + 3: one 16-byte (bf16: 8-byte) store in the
        // pixel-major order (4-byte stores cost the fabric ~6x as much per byte: 8.5 of this kernel's 23 us in a round-2 ablation
        // build).  Mismatching elements are written as they come and overwritten by the pass below (after a workgroup barrier).
        {
            const int prow = lane & 15, b4 = bb * kBins + 4 * (lane >> 4);
            const uint4 fw = *reinterpret_cast<const uint4 *>(flag + b4);        // the four bins' flag words (b4 is a multiple of 4)
            const unsigned sh = 2u * prow;
            const int f0 = (int)((fw.x >> sh) & 3u), f1 = (int)((fw.y >> sh) & 3u), f2 = (int)((fw.z >> sh) & 3u), f3 = (int)((fw.w >> sh) & 3u);
            const float v0 = f0 == 1 ? accD[0] + accD2[0] : 0.f, v1 = f1 == 1 ? accD[1] + accD2[1] : 0.f;
            const float v2 = f2 == 1 ? accD[2] + accD2[2] : 0.f, v3 = f3 == 1 ? accD[3] + accD2[3] : 0.f;
            if (a.vec && bb * kBins + kBins <= D) {
                if (prow < nrow) {
                    FT *dst = grad_depth + tcol + prow * rstep + b4;
                    if constexpr (sizeof(FT) == 2) {
                        uint2 pk;
                        pk.x = pack_bf16x2(v0, v1); pk.y = pack_bf16x2(v2, v3);
                        *reinterpret_cast<uint2 *>(dst) = pk;
                    } else {
                        *reinterpret_cast<float4 *>(dst) = make_float4(v0, v1, v2, v3);
                    }
                }
            } else if (prow < nrow) {
                if (b4 < D && f0 != 2) st_scalar<FT>(grad_depth + tcol + prow * rstep + (int64_t)b4 * dstep, v0);
                if (b4 + 1 < D && f1 != 2) st_scalar<FT>(grad_depth + tcol + prow * rstep + (int64_t)(b4 + 1) * dstep, v1);
                if (b4 + 2 < D && f2 != 2) st_scalar<FT>(grad_depth + tcol + prow * rstep + (int64_t)(b4 + 2) * dstep, v2);
                if (b4 + 3 < D && f3 != 2) st_scalar<FT>(grad_depth + tcol + prow * rstep + (int64_t)(b4 + 3) * dstep, v3);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                                 // the tile is rewritten by the next batch
    }
    COL_STAMP(2);
    // ---- grad_context: the two waves' partial sums meet in LDS
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float *row = gw + (4 * (lane >> 4) + i) * CP;
#pragma unroll
        for (int g4 = 0; g4 < NT / 4; ++g4)
            *reinterpret_cast<float4 *>(row + 64 * g4 + 4 * (lane & 15)) = make_float4(accC[4 * g4][i], accC[4 * g4 + 1][i], accC[4 * g4 + 2][i], accC[4 * g4 + 3][i]);
#pragma unroll
        for (int n = 4 * (NT / 4); n < NT; ++n) row[64 * (NT / 4) + 16 * (n - 4 * (NT / 4)) + (lane & 15)] = accC[n][i];
    }
    __syncthreads();
    float *sum = gw0;                                                    // wave 0's tile becomes the total
    for (int e = tid; e < 16 * C4; e += kColBlock) {
        const int prow = e / C4, c4 = e - prow * C4;
        float4 s0 = *reinterpret_cast<const float4 *>(sum + prow * CP + c4 * 4);
        const float4 s1 = *reinterpret_cast<const float4 *>(gw1 + prow * CP + c4 * 4);
        s0.x += s1.x; s0.y += s1.y; s0.z += s1.z; s0.w += s1.w;
        *reinterpret_cast<float4 *>(sum + prow * CP + c4 * 4) = s0;
    }
    __syncthreads();

    // ---- mismatches (3.8 % of the kept points on the reference's nuScenes calibration, none on a level rig): a lane group
    // (C/4 lanes, a float4 column each) per pixel gathers the pixel's mismatching bins into a list and walks it, four
    // BEV-gradient rows in flight -- the ray walk of lift_splat_tile.hip without its pipeline
    if (nmis > 0) {
        constexpr int G = 64 / C4, NGR = (kColBlock / 64) * G, Q = C4 / 4;
        const int g = lane / C4, li = lane - g * C4;
        const int grp = wave * G + (g < G ? g : 0);
        int *klist = reinterpret_cast<int *>(gw1) + grp * Dp;            // bins of the pixel that need the pass
        float *part = gw1 + NGR * Dp + grp * 4 * Q;                      // quad partials of 4 dot products
        const unsigned long long gmask = C4 >= 64 ? ~0ull : ((1ull << C4) - 1ull);
        for (int p0 = 0; p0 < 16; p0 += NGR) {
            const int prow = p0 + wave * G + g;
            const bool act = g < G && prow < nrow;
            const int prc = act ? prow : 0;
            const int64_t pix = (int64_t)bn * HW + (row0 + prc) * a.fW + col;
            float4 cx = make_float4(0.f, 0.f, 0.f, 0.f), acc = cx;
            if (act) {
                if constexpr (sizeof(FT) == 2) {
                    const uint2 rr = *reinterpret_cast<const uint2 *>(context + pix * C + li * 4);
                    cx = make_float4(bf16_lo(rr.x), bf16_hi(rr.x), bf16_lo(rr.y), bf16_hi(rr.y));
                } else {
                    cx = *reinterpret_cast<const float4 *>(context + pix * C + li * 4);
                }
            }
            int cnt = 0;
            for (int d0 = 0; d0 < D; d0 += C4) {
                const int bin = d0 + li;
                const bool mm = act && bin < D && flag_of(prc, bin < D ? bin : 0) == 2;
                const unsigned long long seg = (__ballot(mm) >> (g < G ? g * C4 : 0)) & gmask;
                if (mm) klist[cnt + __popcll(seg & ((1ull << li) - 1ull))] = bin;
                cnt += __popcll(seg);
            }
            int most = act ? cnt : 0;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) { const int o_ = __shfl_xor(most, m); most = o_ > most ? o_ : most; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int e0 = 0; e0 < most; e0 += 4) {
                mmt_u32x4 v[4];
                float dv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool on = act && (e0 + u) < cnt;
                    const int bin = on ? klist[e0 + u] : 0;
                    const unsigned o = on ? own[prc * Dp + bin] : kOut;        // the point's OWN row, left by phase A
                    dv[u] = on ? dep[prc * Dp + bin] : 0.f;
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + (g < G ? (unsigned)li * 16u : 0x40000000u), 0, 0);
                }
                float mine = 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float g0 = __uint_as_float(v[u].x), g1 = __uint_as_float(v[u].y), g2 = __uint_as_float(v[u].z), g3 = __uint_as_float(v[u].w);
                    acc.x += g0 * dv[u]; acc.y += g1 * dv[u]; acc.z += g2 * dv[u]; acc.w += g3 * dv[u];
                    const float qs = quad_sum(g0 * cx.x + g1 * cx.y + g2 * cx.z + g3 * cx.w);
                    if ((li & 3) == u) mine = qs;
                }
                if (g < G) part[(li & 3) * Q + (li >> 2)] = mine;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (act && li < 4 && (e0 + li) < cnt) {
                    float s2 = 0.f;
                    for (int k = 0; k < Q; ++k) s2 += part[li * Q + k];
                    st_scalar<FT>(grad_depth + tcol + prow * rstep + klist[e0 + li] * dstep, s2);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (act) {
                float4 s0 = *reinterpret_cast<const float4 *>(sum + prow * CP + li * 4);
                s0.x += acc.x; s0.y += acc.y; s0.z += acc.z; s0.w += acc.w;
                *reinterpret_cast<float4 *>(sum + prow * CP + li * 4) = s0;
            }
        }
        __syncthreads();
    }
    COL_STAMP(3);         // (after the mismatch pass)
#ifndef LSS_STAMPS
    for (int e = tid; e < 16 * C4; e += kColBlock) {
        const int prow = e / C4, c4 = e - prow * C4;
        if (prow < nrow) {
            const int64_t pix = (int64_t)bn * HW + (row0 + prow) * a.fW + col;
            *reinterpret_cast<float4 *>(a.grad_context + pix * C + c4 * 4) = *reinterpret_cast<const float4 *>(sum + prow * CP + c4 * 4);
        }
    }
#endif
    // How well the geometry suits this kernel, for the caller's choice between it and the ray walk (read back lazily).  A
    // pseudo-random 1-in-8 SAMPLE of the workgroups reports (same-address atomics retire one at a time at the memory side:
    // with every workgroup reporting, the kernel's tail grew by 3 us inside the training step), spread over
    // MMT_LSS_STATS_SLOTS pairs, as the last thing the workgroup does.
    if (a.stats && tid == 0 && ((blockIdx.x * 0x9E3779B1u) >> 29) == 0u) {
        unsigned long long *slot = a.stats + 2 * (blockIdx.x & (MMT_LSS_STATS_SLOTS - 1));
        if (nmis) atomicAdd(slot, (unsigned long long)nmis);
        if (nkept) atomicAdd(slot + 1, (unsigned long long)nkept);
    }
}

template <typename FT>
int launch(const char *what, int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz, const int32_t *geom, const mmt::CamGeom *cam,
           const FT *depth, const FT *context, const float *grad_out, int64_t sb, int64_t sy, int64_t sx, int64_t span, FT *grad_depth,
           float *grad_context, unsigned long long *stats, int pm, hipStream_t st) {
    ColArgs a = {};
    a.BN = B * N; a.N = N; a.D = D; a.fH = fH; a.fW = fW; a.C = C; a.nx = nx; a.ny = ny; a.nz = nz;
    a.pm = pm; a.rblocks = (fH + 15) / 16;
    a.vec = pm && (D % 4) == 0 && (((uintptr_t)grad_depth) & 15) == 0;
    a.geom = geom; a.depth = depth; a.context = context; a.grad_out = grad_out; a.sb = sb; a.sy = sy; a.sx = sx;
    if (cam) {
        a.combine = cam->combine; a.fu = cam->fu; a.fv = cam->fv; a.fd = cam->fd; a.q = cam->q;
        a.summary = reinterpret_cast<const int2 *>(cam->summary);
    }
    a.stats = stats;
    a.span_bytes = (int)(span * 4);
    a.grad_depth = grad_depth; a.grad_context = grad_context;
    const int Dp = ((D + kBins - 1) / kBins) * kBins;
    const size_t lds = (size_t)16 * Dp * 4 + (size_t)Dp * 4 + (size_t)Dp * 4 + (size_t)2 * 16 * (C + 4) * 4 + (size_t)16 * Dp * 4;
    {
        int gcd = a.BN, e = 8;
        while (e) { const int t = gcd % e; gcd = e; e = t; }
        a.split = 8 / gcd;                                   // BN * split is a multiple of 8
        if (a.split > fW) a.split = 1;
        a.seg = (fW + a.split - 1) / a.split;
    }
    const int64_t grid = 8ll * ((a.BN * a.split + 7) / 8) * a.seg * a.rblocks;
    mmt::TimedSeq seq;
    const dim3 g((unsigned)grid), blk(kColBlock);
    if (cam) {
        if (C == 80) seq.launch(true, lss_col_bwd<FT, 5, true>, g, blk, lds, st, a);
        else if (C == 64) seq.launch(true, lss_col_bwd<FT, 4, true>, g, blk, lds, st, a);
        else seq.launch(true, lss_col_bwd<FT, 8, true>, g, blk, lds, st, a);
    } else {
        if (C == 80) seq.launch(true, lss_col_bwd<FT, 5, false>, g, blk, lds, st, a);
        else if (C == 64) seq.launch(true, lss_col_bwd<FT, 4, false>, g, blk, lds, st, a);
        else seq.launch(true, lss_col_bwd<FT, 8, false>, g, blk, lds, st, a);
    }
    return mmt::check_launch(what);
}

}  // namespace

namespace mmt {

// true when the column kernel takes this shape (C in {64, 80, 128}; LDS within 64 KB; a gradient map below 1 GiB)
bool lss_col_backward_fits(int D, int fH, int fW, int C, int64_t span, int64_t grid_units) {
    const int Dp = ((D + kBins - 1) / kBins) * kBins;
    const size_t lds = (size_t)16 * Dp * 4 + (size_t)Dp * 4 + (size_t)Dp * 4 + (size_t)2 * 16 * (C + 4) * 4 + (size_t)16 * Dp * 4;
    const int C4 = C / 4, NGR = (kColBlock / 64) * (64 / C4);
    if (NGR * Dp + NGR * C4 > 16 * (C + 4)) return false;            // the lists of the mismatch pass live in wave 1's tile
    return (C == 64 || C == 80 || C == 128) && lds <= 64 * 1024 && span * 4 < (1ll << 30) && grid_units < (1ll << 28);
}
int lss_col_backward_f32(const char *what, int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz, const int32_t *geom,
                         const CamGeom *cam, const float *depth, const float *context, const float *grad_out, int64_t sb, int64_t sy,
                         int64_t sx, int64_t span, float *grad_depth, float *grad_context, unsigned long long *stats, int pm, hipStream_t st) {
    return launch<float>(what, B, N, D, fH, fW, C, nx, ny, nz, geom, cam, depth, context, grad_out, sb, sy, sx, span, grad_depth, grad_context, stats, pm, st);
}
int lss_col_backward_bf16(const char *what, int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz, const int32_t *geom,
                          const CamGeom *cam, const bf16_t *depth, const bf16_t *context, const float *grad_out, int64_t sb, int64_t sy,
                          int64_t sx, int64_t span, bf16_t *grad_depth, float *grad_context, unsigned long long *stats, int pm, hipStream_t st) {
    return launch<bf16_t>(what, B, N, D, fH, fW, C, nx, ny, nz, geom, cam, depth, context, grad_out, sb, sy, sx, span, grad_depth, grad_context, stats, pm, st);
}

}  // namespace mmt
