// Fused lift + voxel_pooling over a camera frustum for MI355X (gfx950): the entry points
// mmt_lss_splat_forward / _backward and two of their three kernel families.
//
//   out[b, cell(t), :] += depth[t] * context[pix(t), :]        (lss_fpn.py:441-464 in one pass)
//
// * RAY WALKS (default; second half of this file): no hash, no sort.  Forward: a workgroup owns one image column, its lane
//   groups walk (depth bin, image row) and sum depth * context in registers while the BEV cell stays the same, flushing whole
//   64-byte segments.  Backward: a lane group owns one pixel and walks its kept depth bins; no atomics.
// * FRUSTUM TILES (MMT_LSS_TILE_KERNELS; first half): the second generation.  The first fused kernel (voxel_pooling.hip,
//   vp_fwd_seg_gather<FUSED>) cuts the point list into chunks of CONSECUTIVE points -- ten rows of 44 neighbouring pixels at
//   one depth: 464 different context rows per chunk, gathered from L2 one by one.  Here a workgroup owns a FRUSTUM TILE
//   instead: TPW image columns x all fH rows x Dt depth bins.  The pixels of one image column see the same ray in BEV, and
//   consecutive depth bins of a ray fall into the same or the next cell, so the tile's <= 512 points hit a dozen cells and
//   use only TPW*fH different context rows: the context tile is loaded ONCE into LDS (10-20 KB), the points are sorted by
//   cell in LDS, every lane group sums one cell's rows out of LDS in registers, and the tile leaves as one run of global
//   fp32 atomics per touched cell.  Less sensitive to geometry whose columns are not level (the hash does not care).
// * MATRIX CORES (MMT_LSS_COLUMN_BACKWARD): lift_splat_col.hip, backward only.
// All are correct for ANY geometry; the frustum structure only makes them fast.
//
// Workgroups of one camera are dealt to the same XCD (blockIdx & 7), so the geom / depth lines shared by neighbouring
// columns and the camera's context rows are fetched into one L2 -- placement affects speed only.
#include <type_traits>

#include "mmt_camera.h"

namespace {

constexpr int kBlock = 256;
constexpr int kPts = 512;          // points per workgroup (TP pixels x Dt depth bins)
constexpr int kHT = 2 * kPts;      // hash entries (load factor <= 0.5)
constexpr int kHTLog2 = 10;
constexpr int kEmpty = -1;

struct TileArgs {
    int N, D, fH, fW, C, nx, ny, nz;
    int TPW, TP, Dt;               // columns per tile, pixels per tile (TPW * fH), depth bins per tile
    int wtiles, dtiles, dt_per_grp, ngrp_per_cam, NG;
    const int32_t *geom;           // [B*N, D, fH, fW, 3]
    const void *depth;             // [B*N, D, fH*fW]      fp32 or bf16
    const void *context;           // [B*N, fH*fW, C]      fp32 or bf16, channels-last
    float *out;                    // [B, ny, nx, C]       accumulated into
    int32_t *pos_memo;             // [B*N*D*fH*fW, 3] or NULL (same point order as geom)
    int write_dropped;
    int pm;                        // 1: geom / depth / grad_depth / pos_memo are PIXEL-major [B*N, fH, fW, D(, 3)] (the nets'
                                   //    channels-last order: a tile reads Dt consecutive depth bins per pixel = whole
                                   //    64- / 192-byte runs), 0: the reference's frustum order [B*N, D, fH, fW(, 3)]
    // backward
    const float *grad_out;         // [B, ny, nx, C] view: element strides sb, sy, sx (channel stride 1)
    int64_t sb, sy, sx;
    void *grad_depth;              // [B*N, D, fH*fW]  fp32 or bf16, every element written
    float *grad_context;           // [B*N, fH*fW, C]  fp32, ACCUMULATED INTO with atomics (caller zero-fills)
};

// which (camera, depth tile, column tile) a workgroup owns; false = nothing (grid padding)
__device__ __forceinline__ bool locate(const TileArgs &a, int L, int *bn, int *dtile, int *wtile) {
    const int x = L & 7, i = L >> 3;
    const int per = a.dt_per_grp * a.wtiles;
    const int gl = i / per, r = i - gl * per;
    const int g = gl * 8 + x;
    if (g >= a.NG) return false;
    *bn = g / a.ngrp_per_cam;
    const int dgrp = g - *bn * a.ngrp_per_cam;
    *dtile = dgrp * a.dt_per_grp + r / a.wtiles;
    *wtile = r % a.wtiles;
    return *dtile < a.dtiles;
}

template <typename FT> struct Elem;
template <> struct Elem<float> {
    static constexpr int VEC = 4;       // channels per 16-byte lane vector
    __device__ __forceinline__ static float scalar(const float *p) { return *p; }
    __device__ __forceinline__ static void to_lds(const float *src, float *dst) {   // one 16-byte vector of a context row
        *reinterpret_cast<float4 *>(dst) = *reinterpret_cast<const float4 *>(src);
    }
};
template <> struct Elem<bf16_t> {
    static constexpr int VEC = 8;
    __device__ __forceinline__ static float scalar(const bf16_t *p) { return __uint_as_float((unsigned)*p << 16); }
    __device__ __forceinline__ static void to_lds(const bf16_t *src, float *dst) {   // up-cast once, rows live in LDS as fp32
        const uint4 r = *reinterpret_cast<const uint4 *>(src);
        *reinterpret_cast<float4 *>(dst) = make_float4(bf16_lo(r.x), bf16_hi(r.x), bf16_lo(r.y), bf16_hi(r.y));
        *reinterpret_cast<float4 *>(dst + 4) = make_float4(bf16_lo(r.z), bf16_hi(r.z), bf16_lo(r.w), bf16_hi(r.w));
    }
};

// LDS (dynamic): ctx [TP][C] fp32
template <typename FT, int C4T>
__global__ __launch_bounds__(kBlock) void lss_splat_fwd_tile(TileArgs a) {
    extern __shared__ __align__(16) float ctx_lds[];
    constexpr int PPT = kPts / kBlock;
    constexpr int NW = kBlock / 64;
    __shared__ __align__(16) int tab_key[kHT];           // hash keys; dead after the sort: reused as the flush staging rows
    __shared__ unsigned short tab_slot[kHT];
    __shared__ int slot_key[kPts];
    __shared__ int slot_cnt[kPts];
    __shared__ unsigned short slot_off[kPts + 2];
    __shared__ __align__(8) int2 rec[kPts];              // cell-sorted point records: (depth bits, pixel)
    __shared__ int nslots, next_slot;
    static_assert(kHT * 4 >= NW * 256 * 4, "staging rows must fit in the dead hash table");

    int bn, dtile, wtile;
    if (!locate(a, blockIdx.x, &bn, &dtile, &wtile)) return;
#ifdef LSS_STAMPS   // diagnostic build: pos_memo receives 8 s_memtime stamps per workgroup instead of its rows
    unsigned long long *stamps = reinterpret_cast<unsigned long long *>(a.pos_memo) + (int64_t)blockIdx.x * 8;
    a.pos_memo = nullptr;
#define LSS_STAMP(i) do { if (threadIdx.x == 0 && stamps) stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LSS_STAMP(i) do { } while (0)
#endif
    LSS_STAMP(0);
    const int C = a.C;
    const int C4 = C4T > 0 ? C4T : C >> 2;
    const int G = 64 / C4;                    // lane groups per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HW = a.fH * a.fW;
    const int w0 = wtile * a.TPW, d0 = dtile * a.Dt;
    const int b = bn / a.N;
    const FT *depth = reinterpret_cast<const FT *>(a.depth);
    const FT *context = reinterpret_cast<const FT *>(a.context);

    // ---- requests first: geometry + depth of this thread's points
    int gx[PPT], gy[PPT], gz[PPT], pix[PPT];
    int64_t tg[PPT];
    float dep[PPT];
    bool valid[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int lp = tid + k * kBlock;
        int dd, pp;
        if (a.pm) { pp = lp / a.Dt; dd = lp - pp * a.Dt; } else { dd = lp / a.TP; pp = lp - dd * a.TP; }
        const int hh = pp / a.TPW, ww = pp - hh * a.TPW;
        valid[k] = dd < a.Dt && pp < a.TP && (d0 + dd) < a.D && (w0 + ww) < a.fW;
        pix[k] = pp;
        const int dc = valid[k] ? d0 + dd : d0, wc = valid[k] ? w0 + ww : w0, hc = pp < a.TP ? hh : 0;
        const int64_t t = a.pm ? (((int64_t)bn * a.fH + hc) * a.fW + wc) * a.D + dc
                               : (((int64_t)bn * a.D + dc) * a.fH + hc) * a.fW + wc;
        tg[k] = t;
        gx[k] = a.geom[t * 3]; gy[k] = a.geom[t * 3 + 1]; gz[k] = a.geom[t * 3 + 2];
        dep[k] = Elem<FT>::scalar(depth + t);
    }
    for (int i = tid; i < kHT; i += kBlock) tab_key[i] = kEmpty;
    for (int i = tid; i < kPts; i += kBlock) slot_cnt[i] = 0;
    if (tid == 0) { nslots = 0; next_slot = 0; }
    // context tile -> LDS (fp32): TP rows of C channels
    {
        constexpr int VEC = Elem<FT>::VEC;
        const int CV = C / VEC;
        for (int i = tid; i < a.TP * CV; i += kBlock) {
            const int pp = i / CV, cv = i - pp * CV;
            const int hh = pp / a.TPW, ww = pp - hh * a.TPW;
            if (w0 + ww < a.fW)
                Elem<FT>::to_lds(context + ((int64_t)bn * HW + hh * a.fW + w0 + ww) * C + cv * VEC, ctx_lds + pp * C + cv * VEC);
        }
    }
    __syncthreads();
    LSS_STAMP(1);

    // ---- A1: bounds test, hash insert of the cell key; pos_memo straight from registers (optional output)
    int ent[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        int e = -1;
        if (valid[k]) {
            const int x = gx[k], y = gy[k], z = gz[k];
            const bool kept = !(x < 0 || x >= a.nx || y < 0 || y >= a.ny || z < 0 || z >= a.nz);
            if (kept) {
                const int key = (b * a.ny + y) * a.nx + x;
                unsigned h = ((unsigned)key * 2654435761u) >> (32 - kHTLog2);
                for (int probe = 0; probe < kHT; ++probe) {   // never fills: <= kPts keys in 2*kPts entries
                    const int prev = atomicCAS(&tab_key[h], kEmpty, key);
                    if (prev == kEmpty) {
                        const int s = atomicAdd(&nslots, 1);
                        tab_slot[h] = (unsigned short)s;
                        slot_key[s] = key;
                        e = (int)h;
                        break;
                    }
                    if (prev == key) { e = (int)h; break; }
                    h = (h + 1) & (kHT - 1);
                }
            }
            if (a.pos_memo && (kept || a.write_dropped)) {
                int32_t *pm = a.pos_memo + tg[k] * 3;
                pm[0] = kept ? b : -1; pm[1] = kept ? y : -1; pm[2] = kept ? x : -1;
            }
        }
        ent[k] = e;
    }
    __syncthreads();
    LSS_STAMP(2);

    // ---- A2: per-slot counts (rank of the point inside its cell's list)
    int slot[PPT], rank[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        slot[k] = -1;
        rank[k] = 0;
        if (ent[k] >= 0) {
            slot[k] = tab_slot[ent[k]];
            rank[k] = atomicAdd(&slot_cnt[slot[k]], 1);
        }
    }
    __syncthreads();

    // ---- A3: exclusive scan of the counts by wave 0
    const int ns = nslots;
    if (wave == 0) {
        constexpr int PER = kPts / 64;
        int loc[PER];
        int sum = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = lane * PER + i;
            loc[i] = idx < ns ? slot_cnt[idx] : 0;
            sum += loc[i];
        }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        int run = incl - sum;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = lane * PER + i;
            if (idx <= ns) slot_off[idx] = (unsigned short)run;
            run += loc[i];
        }
        if (lane == 63) slot_off[ns] = (unsigned short)incl;      // ns == kPts: idx never reaches it above
    }
    __syncthreads();
    LSS_STAMP(3);

    // ---- A4: every point writes its record (depth, pixel) at its place in the cell-sorted list
#pragma unroll
    for (int k = 0; k < PPT; ++k)
        if (slot[k] >= 0) rec[slot_off[slot[k]] + rank[k]] = make_int2(__float_as_int(dep[k]), pix[k]);
    __syncthreads();
    LSS_STAMP(4);

    // ---- B/C: every lane group sums ONE cell's rows (out of the LDS context tile) in registers; the G rows of a wave meet
    // in a staging row (the dead hash table) and leave as contiguous runs of global fp32 atomics
    constexpr int kRows = 4;
    const int g = lane / C4;
    const int li = lane - g * C4;
    const bool active = g < G;
    float *st = reinterpret_cast<float *>(tab_key) + wave * 256;
    for (;;) {
        int s0 = 0;
        if (lane == 0) s0 = atomicAdd(&next_slot, G);
        s0 = __builtin_amdgcn_readfirstlane(s0);
        if (s0 >= ns) break;
        const int s = s0 + g;
        int beg = 0, end = 0;
        if (active && s < ns) { beg = slot_off[s]; end = slot_off[s + 1]; }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = beg; j < end; j += kRows) {
            int2 r[kRows];
            float4 v[kRows];
#pragma unroll
            for (int u = 0; u < kRows; ++u) r[u] = rec[(j + u) < end ? (j + u) : (end - 1)];
#pragma unroll
            for (int u = 0; u < kRows; ++u) v[u] = *reinterpret_cast<const float4 *>(ctx_lds + r[u].y * C + li * 4);
#pragma unroll
            for (int u = 0; u < kRows; ++u) {   // product rounded to fp32 first (= the materialised lift), then added
                const float dv = (j + u) < end ? __int_as_float(r[u].x) : 0.f;
                acc.x += __fmul_rn(dv, v[u].x); acc.y += __fmul_rn(dv, v[u].y);
                acc.z += __fmul_rn(dv, v[u].z); acc.w += __fmul_rn(dv, v[u].w);
            }
        }
        if (active) *reinterpret_cast<float4 *>(st + g * C + li * 4) = acc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < G * C; e += 64) {
            const int gg = e / C;
            const int ss = s0 + gg;
            if (ss < ns) atomicAdd(a.out + (int64_t)slot_key[ss] * C + (e - gg * C), st[e]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    LSS_STAMP(5);
#undef LSS_STAMP
}

// ---------------------------------------------------------------------------
// Backward on the same frustum tiles:
//   grad_depth[t]        = < grad_out[cell(t), :], context[pix(t), :] >          (0 for dropped points)
//   grad_context[pix, :] += sum over the tile's depth bins of depth[t] * grad_out[cell(t), :]
// The first-generation kernel (voxel_pooling.hip, lift_splat_backward_kernel) gathers one 320-byte BEV-gradient row per
// kept point from L2 (364 MB at cfg2).  A tile touches a dozen or two cells: their rows are loaded ONCE into LDS beside the
// context tile, the kept test is redone from geom (no pos_memo), every lane group owns a pixel at a time and walks its
// Dt depth bins out of LDS -- no sort is needed, only each point's row slot.  grad_depth is a plain store (a point
// belongs to one tile); grad_context receives one run of fp32 atomics per (pixel, depth tile) -- D / Dt partial sums
// per pixel, 38 MB of atomic traffic at cfg2 for a 5.4 MB tensor.
constexpr int kGradRows = 24;      // BEV-gradient rows held in LDS per workgroup; further cells are read from L2

// LDS (dynamic): ctx [TP][C] fp32 | rows [kGradRows][C] fp32
template <typename FT, int C4T>
__global__ __launch_bounds__(kBlock) void lss_splat_bwd_tile(TileArgs a) {
    extern __shared__ __align__(16) float dyn_lds[];
    constexpr int PPT = kPts / kBlock;
    constexpr int NW = kBlock / 64;
    __shared__ __align__(16) int tab_key[kHT];           // hash keys; dead after the slot lookup: reused as the flush staging rows
    __shared__ unsigned short tab_slot[kHT];
    __shared__ int slot_addr[kPts];                      // element offset of the slot's BEV-gradient row
    __shared__ short pt_slot[kPts];                      // row slot of a point, -1 = dropped / outside the tile
    __shared__ float pt_depth[kPts];
    __shared__ float gd[kPts];
    __shared__ int nslots;
    static_assert(kHT * 4 >= NW * 256 * 4, "staging rows must fit in the dead hash table");

    int bn, dtile, wtile;
    if (!locate(a, blockIdx.x, &bn, &dtile, &wtile)) return;
    const int C = a.C;
    const int C4 = C4T > 0 ? C4T : C >> 2;
    const int G = 64 / C4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HW = a.fH * a.fW;
    const int w0 = wtile * a.TPW, d0 = dtile * a.Dt;
    const int b = bn / a.N;
    const FT *depth = reinterpret_cast<const FT *>(a.depth);
    const FT *context = reinterpret_cast<const FT *>(a.context);
    float *ctx_lds = dyn_lds;
    float *rows = dyn_lds + a.TP * C;

    int gx[PPT], gy[PPT], gz[PPT];
    int64_t tg[PPT];
    float dep[PPT];
    bool valid[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int lp = tid + k * kBlock;
        int dd, pp;
        if (a.pm) { pp = lp / a.Dt; dd = lp - pp * a.Dt; } else { dd = lp / a.TP; pp = lp - dd * a.TP; }
        const int hh = pp / a.TPW, ww = pp - hh * a.TPW;
        valid[k] = dd < a.Dt && pp < a.TP && (d0 + dd) < a.D && (w0 + ww) < a.fW;
        const int dc = valid[k] ? d0 + dd : d0, wc = valid[k] ? w0 + ww : w0, hc = pp < a.TP ? hh : 0;
        const int64_t t = a.pm ? (((int64_t)bn * a.fH + hc) * a.fW + wc) * a.D + dc
                               : (((int64_t)bn * a.D + dc) * a.fH + hc) * a.fW + wc;
        tg[k] = t;
        gx[k] = a.geom[t * 3]; gy[k] = a.geom[t * 3 + 1]; gz[k] = a.geom[t * 3 + 2];
        dep[k] = Elem<FT>::scalar(depth + t);
    }
    for (int i = tid; i < kHT; i += kBlock) tab_key[i] = kEmpty;
    if (tid == 0) nslots = 0;
    {
        constexpr int VEC = Elem<FT>::VEC;
        const int CV = C / VEC;
        for (int i = tid; i < a.TP * CV; i += kBlock) {
            const int pp = i / CV, cv = i - pp * CV;
            const int hh = pp / a.TPW, ww = pp - hh * a.TPW;
            if (w0 + ww < a.fW)
                Elem<FT>::to_lds(context + ((int64_t)bn * HW + hh * a.fW + w0 + ww) * C + cv * VEC, ctx_lds + pp * C + cv * VEC);
        }
    }
    __syncthreads();

    // ---- every kept point finds / creates the slot of its cell (slot = order of first insertion)
    int ent[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int lp = tid + k * kBlock;
        int e = -1;
        pt_depth[lp] = dep[k];
        if (valid[k]) {
            const int x = gx[k], y = gy[k], z = gz[k];
            if (!(x < 0 || x >= a.nx || y < 0 || y >= a.ny || z < 0 || z >= a.nz)) {
                const int key = (b * a.ny + y) * a.nx + x;
                unsigned h = ((unsigned)key * 2654435761u) >> (32 - kHTLog2);
                for (int probe = 0; probe < kHT; ++probe) {   // never fills: <= kPts keys in 2*kPts entries
                    const int prev = atomicCAS(&tab_key[h], kEmpty, key);
                    if (prev == kEmpty) {
                        const int s_ = atomicAdd(&nslots, 1);
                        tab_slot[h] = (unsigned short)s_;
                        slot_addr[s_] = (int)(b * a.sb + y * a.sy + x * a.sx);
                        e = (int)h;
                        break;
                    }
                    if (prev == key) { e = (int)h; break; }
                    h = (h + 1) & (kHT - 1);
                }
            }
        }
        ent[k] = e;
    }
    __syncthreads();
    const int ns = nslots;
#pragma unroll
    for (int k = 0; k < PPT; ++k) pt_slot[tid + k * kBlock] = ent[k] >= 0 ? (short)tab_slot[ent[k]] : (short)-1;
    // the BEV-gradient rows of the tile's cells -> LDS, once
    {
        const int nrow = ns < kGradRows ? ns : kGradRows;
        for (int i = tid; i < nrow * C4; i += kBlock) {
            const int s_ = i / C4, c4 = i - s_ * C4;
            *reinterpret_cast<float4 *>(rows + s_ * C + c4 * 4) = *reinterpret_cast<const float4 *>(a.grad_out + slot_addr[s_] + c4 * 4);
        }
    }
    __syncthreads();

    // ---- one lane group per pixel at a time: Dt depth bins out of LDS
    const int g = lane / C4;
    const int li = lane - g * C4;
    float *st = reinterpret_cast<float *>(tab_key) + wave * 256;       // hash keys are dead now
    const int NGR = NW * G;
    for (int p0 = 0; p0 < a.TP; p0 += NGR) {            // uniform trip count for the whole workgroup (wave barriers inside)
        const int pp = p0 + wave * G + g;
        const bool act = g < G && pp < a.TP;
        const int hh = act ? pp / a.TPW : 0, ww = act ? pp - hh * a.TPW : 0;
        const bool inimg = act && (w0 + ww) < a.fW;
        float4 cx = make_float4(0.f, 0.f, 0.f, 0.f), acc = cx;
        if (inimg) cx = *reinterpret_cast<const float4 *>(ctx_lds + pp * C + li * 4);
        if (inimg) {
            for (int dd = 0; dd < a.Dt; ++dd) {
                const int lp = a.pm ? pp * a.Dt + dd : dd * a.TP + pp;
                const int s_ = pt_slot[lp];
                float total = 0.f;
                if (s_ >= 0) {
                    const float dv = pt_depth[lp];
                    const float4 gr = s_ < kGradRows ? *reinterpret_cast<const float4 *>(rows + s_ * C + li * 4)
                                                     : *reinterpret_cast<const float4 *>(a.grad_out + slot_addr[s_] + li * 4);
                    acc.x += gr.x * dv; acc.y += gr.y * dv; acc.z += gr.z * dv; acc.w += gr.w * dv;
                    float q = gr.x * cx.x + gr.y * cx.y + gr.z * cx.z + gr.w * cx.w;
                    q += __shfl_xor(q, 1);
                    q += __shfl_xor(q, 2);                  // quad sum in all 4 lanes (lane groups start on quad boundaries)
                    total = q;
                    for (int o = 4; o < C4; o += 4) total += __shfl_down(q, o);   // lane li == 0: all C4/4 quads of the group
                }
                if (li == 0) gd[lp] = total;
            }
        }
        // grad_context: the G rows of a wave meet in a staging row and leave as contiguous fp32 atomics
        if (act) *reinterpret_cast<float4 *>(st + g * C + li * 4) = acc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < G * C; e += 64) {
            const int gg = e / C;
            const int pq = p0 + wave * G + gg;
            if (pq < a.TP) {
                const int h2 = pq / a.TPW, w2 = pq - h2 * a.TPW;
                if (w0 + w2 < a.fW) atomicAdd(a.grad_context + ((int64_t)bn * HW + h2 * a.fW + w0 + w2) * C + (e - gg * C), st[e]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();

    // ---- grad_depth: a point belongs to exactly one tile
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        if (valid[k]) {
            const float v = gd[tid + k * kBlock];
            if constexpr (sizeof(FT) == 2) reinterpret_cast<bf16_t *>(a.grad_depth)[tg[k]] = (bf16_t)(pack_bf16x2(v, 0.f) & 0xFFFFu);
            else reinterpret_cast<float *>(a.grad_depth)[tg[k]] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// Third generation: RAY WALKS.  No hash, no sort, one workgroup barrier (forward) or none (backward).
//
// The points of one image column at one depth bin sit above each other in the world: for a (nearly) level camera they
// fall into the same BEV cell, and the next depth bin of that ray falls into the same or the neighbouring cell.  A walk
// over (depth bin, image row) of ONE column therefore sees the same cell dozens of times in a row, and a plain
// run-length sum in registers already removes ~95 % of the atomic rows -- for any geometry the result is the same, an
// unstructured one merely flushes more often.
//
// ---------------------------------------------------------------------------
// EXCLUSIVE-CELL CACHE of the register-walk forward (camera form + MMT_LSS_ZERO_OUTPUT; nullable).  With the walk down to
// 7 us the forward waits for the memory-side atomic units: 78 k runs x 320 B = 25 MB at ~1.25 TB/s.  But 43 % of the runs
// (cfg4 rig) go to cells that NO other run of the launch touches -- far cells seen by one column of one camera -- and a cell
// with a single contributor needs no atomic: a plain store into the zero-filled row does.  Which cells those are is a
// function of a sample's camera matrices and of the launch shape alone, so it is learnt on the device, per SAMPLE (a
// shuffled loader repeats rigs, not batches), with no host involvement, no synchronisation and no workgroup waiting for
// another:
//   * the cache is a direct-mapped table of `slots` calibrations: a sample's N matrices, a 64-bit hash of them, a stage,
//     and one int32 STATE per BEV cell of a sample's map;
//   * every workgroup of the forward hashes its sample's matrices itself (96 words, one wave each), reads the slot the hash
//     points at and compares the matrices bit for bit -- two rounds of loads behind which its geometry phase runs anyway.
//     All workgroups of a sample read the same words, and nothing changes them during the launch, so they agree;
//   * stage 1, MARK: every run leaves its id -- (camera, column, slab, lane group, first bin, first row), the same in every
//     launch of this shape whatever the sample's place in the batch -- in state[cell] with a plain store: some id survives;
//   * stage 2, VERIFY: a run that finds another id than its own stores -1.  Afterwards state[cell] > 0 says "one run only";
//   * stage 3, USE: the geometry phase tags the keys of such cells, and a tagged run leaves as plain stores.
//   * the first column workgroup of a sample posts what should happen next -- "claim this slot for my matrices" on a miss,
//     "next stage" on a hit -- in a MAILBOX; the zero-fill kernel in front of the NEXT forward (one workgroup) commits the
//     mail to the table.  A claim costs one call, so a calibration is used from its fourth call on.
// Learning costs a 4-byte load or store per run (no same-address atomic chains: a counter per cell was measured at +14 us
// per call).  Samples of one call that share their matrices post and learn the same things twice, harmlessly.  A
// calibration whose slot another one takes starts over when it comes back.  The header remembers the launch shape (a
// host-side signature) and a hash of the frustum axes' contents: a change empties the table.
// One cache per stream: calls that share it must be stream-ordered (each forward is preceded by its zero-fill).
constexpr int kExclFlag = 1 << 30;       // in a key: the cell has a single contributing run
constexpr int kExclMaxB = 8;             // samples per call that take part (the rest run without the cache)
constexpr int kExclMaxN = 8;             // cameras per sample (more: the cache is not used)
constexpr int kExclMailWords = 8 + kExclMaxN * 16;     // per sample: [0] new stage (0 = no mail), [1] slot, [2..3] hash, [8..] matrices (claims)
constexpr int kExclHeaderWords = 64 + kExclMaxB * kExclMailWords;    // [0..1] shape signature, [3..4] hash of the frustum axes, [64..] the mailbox
constexpr int kExclMetaWords = 4;        // per slot: hash lo, hash hi, stage (0 free, 1 claimed: mark, 2 marked: verify, 3 verified: use), -
constexpr int kExclMaxSlots = 1 << 16;
struct ExclShape { int slots, N, cells; unsigned sig_lo, sig_hi; };      // cells = ny * nx of one sample
struct ExclCall {                        // what the commit step needs (zero-fill kernel argument)
    int32_t *cache;                      // nullptr: no cache in this call
    const float *fu, *fv, *fd;
    int fW, fH, D;
    ExclShape xs;
};
__host__ __device__ inline int64_t excl_slot_words(int N, int cells) { return (int64_t)kExclMetaWords + (int64_t)N * 16 + cells; }
__device__ __forceinline__ int32_t *excl_mail(int32_t *c, int b) { return c + 64 + b * kExclMailWords; }
__device__ __forceinline__ int32_t *excl_meta(int32_t *c, int s) { return c + kExclHeaderWords + (int64_t)s * kExclMetaWords; }
__device__ __forceinline__ int32_t *excl_mats(int32_t *c, const ExclShape &x, int s) {
    return c + kExclHeaderWords + (int64_t)x.slots * kExclMetaWords + (int64_t)s * x.N * 16;
}
__device__ __forceinline__ int32_t *excl_state(int32_t *c, const ExclShape &x, int s) {
    return c + kExclHeaderWords + (int64_t)x.slots * (kExclMetaWords + x.N * 16) + (int64_t)s * x.cells;
}

__device__ __forceinline__ unsigned excl_mix(unsigned v) {      // a 32-bit finaliser (murmur3)
    v ^= v >> 16; v *= 0x85EBCA6Bu; v ^= v >> 13; v *= 0xC2B2AE35u; v ^= v >> 16;
    return v;
}
__device__ __forceinline__ void excl_hash_word(unsigned w, int i, unsigned &h0, unsigned &h1) {
    h0 ^= excl_mix(w + 0x9E3779B9u * (unsigned)(i + 1));
    h1 ^= excl_mix((w ^ 0x7F4A7C15u) + 0x85EBCA77u * (unsigned)(i + 1));
}

// What ONE WAVE of a forward workgroup finds out about its sample (every wave does it for itself: no barrier).  nw <= 128.
// excl_probe_begin issues the loads (the matrices' words, then -- the slot follows from their hash -- the slot's metadata
// and matrices); excl_probe_end, called after the geometry phase, compares.  In between the slot is known and its states can
// be read speculatively.
// Round 4: the table is 2-WAY SET-ASSOCIATIVE (slots 2s and 2s + 1 form set s; a table of one slot stays direct-mapped): two
// calibrations whose hashes meet in one set no longer take the slot from each other on every visit -- with a few hundred
// calibrations in a shuffled loader and 1024 slots, dozens of pairs collided and never got past the claim.  A miss claims
// the free way, else the way that has learnt less (ties: a hash bit).  Header words [32..34] count, per sample and call,
// hits (stage 3: the cache was used), learning calls (stages 1 / 2) and misses -- mmt_lss_exclusive_cache_bytes' caller reads
// them back (bench.py reports them).
// (scalars, no arrays: an indexed member sends the whole struct to scratch memory -- 18 MB of private-segment traffic per launch)
struct ExclProbe { int mode, slot, way_free; unsigned h0, h1, w0, w1, a0, a1, b0, b1; int4 mta, mtb; int base, ways; bool match; };
__device__ __forceinline__ bool excl_meta_is(const int4 &mt, unsigned h0, unsigned h1) { return (unsigned)mt.x == h0 && (unsigned)mt.y == h1 && mt.z > 0; }
__device__ __forceinline__ ExclProbe excl_probe_begin(int32_t *cache, const ExclShape &x, const float *combine, int b) {
    const int lane = threadIdx.x & 63, nw = x.N * 16;
    const unsigned *m = reinterpret_cast<const unsigned *>(combine) + (int64_t)b * nw;
    ExclProbe p;
    p.w0 = lane < nw ? m[lane] : 0u; p.w1 = lane + 64 < nw ? m[lane + 64] : 0u;
    unsigned h0 = 0, h1 = 0;
    if (lane < nw) excl_hash_word(p.w0, lane, h0, h1);
    if (lane + 64 < nw) excl_hash_word(p.w1, lane + 64, h0, h1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { h0 ^= __shfl_xor(h0, o); h1 ^= __shfl_xor(h1, o); }
    p.h0 = h0; p.h1 = h1 | 1u;                                      // (a free slot's hash words are 0)
    p.ways = x.slots >= 2 ? 2 : 1;
    p.base = (int)((h0 ^ (p.h1 * 0x9E3779B1u)) % (unsigned)(x.slots / p.ways)) * p.ways;
    const int sla = p.base, slb = p.base + p.ways - 1;              // (one way: both name the same slot)
    p.mta = *reinterpret_cast<const int4 *>(excl_meta(cache, sla));
    p.mtb = *reinterpret_cast<const int4 *>(excl_meta(cache, slb));
    const unsigned *sma = reinterpret_cast<const unsigned *>(excl_mats(cache, x, sla));
    const unsigned *smb = reinterpret_cast<const unsigned *>(excl_mats(cache, x, slb));
    p.a0 = lane < nw ? sma[lane] : 0u; p.a1 = lane + 64 < nw ? sma[lane + 64] : 0u;
    p.b0 = lane < nw ? smb[lane] : 0u; p.b1 = lane + 64 < nw ? smb[lane + 64] : 0u;
    p.slot = p.base; p.way_free = 0;
    p.match = false; p.mode = 0;
    return p;
}
// the way whose metadata carries this sample's hash (known one round trip after excl_probe_begin: the states of that slot are
// then read speculatively; whether its matrices really are the sample's is settled in excl_probe_end)
__device__ __forceinline__ int excl_probe_slot(const ExclProbe &p) {
    return (p.ways == 2 && !excl_meta_is(p.mta, p.h0, p.h1) && excl_meta_is(p.mtb, p.h0, p.h1)) ? p.base + 1 : p.base;
}
__device__ __forceinline__ void excl_probe_end(ExclProbe &p) {
    const bool second = excl_probe_slot(p) != p.base;
    const bool same_a = __all(p.a0 == p.w0 && p.a1 == p.w1), same_b = __all(p.b0 == p.w0 && p.b1 == p.w1);
    const bool is_a = excl_meta_is(p.mta, p.h0, p.h1), is_b = excl_meta_is(p.mtb, p.h0, p.h1);
    p.slot = p.base + (second ? 1 : 0);
    p.match = second ? (same_b && is_b) : (same_a && is_a);
    p.mode = p.match ? (second ? p.mtb.z : p.mta.z) : 0;
    // where a miss claims: a free way first, else the way that has learnt less, else a hash bit
    if (p.ways == 2) {
        const int st0 = p.mta.z, st1 = p.mtb.z;
        p.way_free = st0 <= 0 ? 0 : (st1 <= 0 ? 1 : (st0 != st1 ? (st1 < st0 ? 1 : 0) : (int)((p.h1 >> 1) & 1u)));
    }
}
// The sample's first column workgroup (wave 0) posts the mail of this call.
__device__ __forceinline__ void excl_post(int32_t *cache, const ExclShape &x, const float *combine, int b, const ExclProbe &p) {
    const int lane = threadIdx.x & 63, nw = x.N * 16;
    int32_t *mail = excl_mail(cache, b);
    const int slot = p.match ? p.slot : p.base + p.way_free;
    if (!p.match) {                                                  // claim the slot
        const unsigned *m = reinterpret_cast<const unsigned *>(combine) + (int64_t)b * nw;
        for (int i = lane; i < nw; i += 64) mail[8 + i] = (int)m[i];
        if (lane == 0) { mail[1] = slot; mail[2] = (int)p.h0; mail[3] = (int)p.h1; mail[0] = 1; }
    } else if (p.mode < 3) {                                         // this call marks (verifies): next stage
        if (lane == 0) { mail[1] = slot; mail[2] = (int)p.h0; mail[3] = (int)p.h1; mail[0] = p.mode + 1; }
    }
    if (lane == 0) {
        cache[8 + b] = slot; cache[24 + b] = p.mode;                 // (for the curious: what this call did with sample b)
        atomicAdd(cache + (p.mode == 3 ? 32 : (p.mode > 0 ? 33 : 34)), 1);     // hits / learning calls / misses, cumulative
    }
}

// The commit step: one workgroup (256 threads) of the zero-fill kernel in front of every forward.
__device__ void excl_commit(const ExclCall &k) {
    __shared__ unsigned s_axes[2 + 2 * (kBlock / 64)];
    __shared__ int s_fresh;
    int32_t *cache = k.cache;
    const ExclShape x = k.xs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = x.N * 16;
    // (everything the usual call needs in ONE round of loads: this workgroup runs beside the fill's 21 MB of stores)
    int hdr = 0, stage_new = 0;
    if (tid < 5) hdr = cache[tid];
    if (tid < kExclMaxB) stage_new = excl_mail(cache, tid)[0];
    {   // the frustum axes' contents (a cache serves one frustum)
        unsigned a0 = 0, a1 = 0;
        for (int i = tid; i < k.fW + k.fH + k.D; i += kBlock) {
            const float *src = i < k.fW ? k.fu + i : (i < k.fW + k.fH ? k.fv + (i - k.fW) : k.fd + (i - k.fW - k.fH));
            excl_hash_word(__float_as_uint(*src), i, a0, a1);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a0 ^= __shfl_xor(a0, o); a1 ^= __shfl_xor(a1, o); }
        if (lane == 0) { s_axes[2 + wave * 2] = a0; s_axes[3 + wave * 2] = a1; }
    }
    __syncthreads();
    if (wave == 0) {
        unsigned a0 = 0, a1 = 0;
        for (int w = 0; w < kBlock / 64; ++w) { a0 ^= s_axes[2 + w * 2]; a1 ^= s_axes[3 + w * 2]; }
        const unsigned want = lane == 0 ? x.sig_lo : (lane == 1 ? x.sig_hi : (lane == 3 ? a0 : a1));
        const bool differs = (lane < 5 && lane != 2) && (unsigned)hdr != want;
        const bool any = __any(differs);
        if (lane == 0) { s_axes[0] = a0; s_axes[1] = a1; s_fresh = any ? 1 : 0; }
    }
    __syncthreads();
    if (s_fresh) {                                                   // another launch shape, other axes, a new cache: empty table, mail dropped
        for (int64_t i = tid; i < (int64_t)x.slots * kExclMetaWords; i += kBlock) cache[kExclHeaderWords + i] = 0;
        if (tid < kExclMaxB) excl_mail(cache, tid)[0] = 0;
        if (tid == 0) { cache[0] = (int)x.sig_lo; cache[1] = (int)x.sig_hi; cache[3] = (int)s_axes[0]; cache[4] = (int)s_axes[1]; }
        return;
    }
    if (wave != 0) return;
    // wave 0: the mail in sample order (a later claim of a slot overrides an earlier letter about it)
    for (int b = 0; b < kExclMaxB; ++b) {
        const int sn = __shfl(stage_new, b);
        if (sn == 0) continue;
        int32_t *mail = excl_mail(cache, b);
        const int slot = mail[1];
        const unsigned h0 = (unsigned)mail[2], h1 = (unsigned)mail[3];
        if (slot >= 0 && slot < x.slots) {
            int32_t *mt = excl_meta(cache, slot);
            if (sn == 1) {
                unsigned *sm = reinterpret_cast<unsigned *>(excl_mats(cache, x, slot));
                for (int i = lane; i < nw; i += 64) sm[i] = (unsigned)mail[8 + i];
                if (lane == 0) { mt[0] = (int)h0; mt[1] = (int)h1; mt[2] = 1; mt[3] = 0; }
            } else {
                const bool owner = (unsigned)mt[0] == h0 && (unsigned)mt[1] == h1 && mt[2] > 0;
                if (lane == 0 && owner) mt[2] = sn;
            }
        }
        if (lane == 0) mail[0] = 0;
    }
}

// Forward: a workgroup owns one image column of one camera (optionally a depth slab of it).  Its geometry is reduced to
// (cell, depth) records in LDS by all threads at once (coalesced in the pixel-major layout), the column's fH context
// rows are staged in LDS as fp32, then every lane group walks `kd` consecutive depth bins x fH rows, summing
// depth * context into 4 registers per lane (lane li holds channels li, li + C/4, li + 2C/4, li + 3C/4, so that a
// flush is 4 atomic instructions of C/4 CONTIGUOUS floats each) and flushing when the cell changes.
//
// Backward: a lane group owns one PIXEL and walks its D depth bins: context row and grad_context sum in registers,
// BEV-gradient rows straight from L2 through a range-checked buffer descriptor (dropped points read zeros), four rows
// in flight; grad_depth = <g_row, ctx_row> by quad DPP sums whose C/16 partials meet in a small per-group LDS buffer
// once per C/4 depth bins.  A pixel belongs to one lane group, so grad_context is a plain store: no atomics, no
// zero-fill, no barrier.  The 4*G pixels of a workgroup are neighbours in one column (same ray => same BEV rows, L1 hits).
struct RayArgs {
    int BN, N, D, fH, fW, C, nx, ny, nz;
    int pm, write_dropped;
    int dsplit, dspan, kd;         // forward: depth slabs per ray, bins per slab, bins per lane group
    int compact;                   // forward: walk only the depth bins that hold a kept point (long rays: most far bins are empty)
    int reg;                       // forward: the register walk (lss_ray_fwd_reg: fH <= 16, C <= 80, short rays)
    int blk;                       // forward, camera form: the block walk (lss_ray_fwd_blk: every other shape)
    int32_t *excl;                 // forward (register walk, camera form), nullable: the exclusive-cell cache
    ExclShape xs;
    int wpc;                       // backward: workgroups per camera
    int split, wps;                // backward: segments per camera, workgroups per segment (XCD balance)
    const int32_t *geom;           // geom form: int32 voxel indices per point; camera form (template CAM): unused
    const float *combine, *fu, *fv, *fd;   // camera form: [B*N, 16] matrices and the frustum's three axes (mmt_camera.h)
    mmt::CamGrid q;
    int2 *summary;                 // camera form, nullable: column summary (mmt_camera.h CamGeom::summary)
    int summary_cached;            // forward: 1 = read the summary instead of computing (and writing) it
    const void *depth;
    const void *context;
    float *out;
    int32_t *pos_memo;
    const float *grad_out;
    int64_t sb, sy, sx;
    int span_bytes;
    void *grad_depth;
    float *grad_context;
};

__device__ __forceinline__ int64_t ray_point(const RayArgs &a, int bn, int row, int col, int d) {
    return a.pm ? (((int64_t)bn * a.fH + row) * a.fW + col) * a.D + d : (((int64_t)bn * a.D + d) * a.fH + row) * a.fW + col;
}

// LDS (dynamic): ctx [fH][C] fp32 | rec [fH][dspan] (cell or -1, depth bits)
// S = C / 16: a lane group is 16 lanes, lane li holds channels li, li + 16, ... (S registers), so that one atomic
// instruction of a group is one whole, aligned 64-byte segment of a BEV row -- the memory-side atomic units work in 64-byte
// requests, and 80-byte pieces (C/4 lanes x 4 registers) or 16-byte lane strides cost 1.6x / 4x as many of them
// (tools/ubench/atomic_rows.hip).  The 16 lane groups of a workgroup split the depth bins of the column among them.
template <typename FT, int S, bool CAM>
__global__ __launch_bounds__(kBlock) void lss_ray_fwd(RayArgs a) {
    extern __shared__ __align__(16) float ray_lds[];
    constexpr int C = 16 * S;
    const int L = blockIdx.x, xcd = L & 7, i = L >> 3;
    const int per = a.fW * a.dsplit;
    const int q = i / per, r = i - q * per;
    const int bn = q * 8 + xcd;                       // the columns of one camera share an XCD (context rows, geom lines)
    if (bn >= a.BN) return;
    const int col = r / a.dsplit, slab = r - col * a.dsplit;
    const int d0 = slab * a.dspan;
    const int dn = (a.D - d0) < a.dspan ? (a.D - d0) : a.dspan;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fH = a.fH, HW = a.fH * a.fW;
    const int b = bn / a.N;
    const FT *depth = reinterpret_cast<const FT *>(a.depth);
    const FT *context = reinterpret_cast<const FT *>(a.context);
#ifdef LSS_STAMPS   // diagnostic build: pos_memo receives 4 s_memtime stamps per workgroup (per wave 0 .. 3: stamp 2) instead of its rows
    unsigned long long *fstamps = reinterpret_cast<unsigned long long *>(a.pos_memo) + (int64_t)blockIdx.x * 8;
    a.pos_memo = nullptr;
#define FWD_STAMP(i) do { if (threadIdx.x == 0 && fstamps) fstamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FWD_STAMP(i) do { } while (0)
#endif
    FWD_STAMP(0);
    // padded row strides: the 4 lane groups of a wave read the same context row (broadcast) but records 2 * kd dwords apart
    constexpr int CP = C + 4;
    const int dnp = a.dspan | 1;
    float *ctx = ray_lds;
    int2 *rec = reinterpret_cast<int2 *>(ray_lds + fH * CP);
    // which depth bins of this column hold a kept point at all (the walk skips the others: beyond the grid the far bins of a
    // ray are empty in every row -- 84 % of all points at the reference's native 409-bin frustum), and their compacted list
    int *binkept = reinterpret_cast<int *>(rec + (size_t)fH * dnp);
    int *klist = binkept + a.dspan;
    __shared__ int s_nkb;

    // ---- all threads: geometry -> (cell, depth) records
    // camera form: the voxel index of a point is computed here from the camera's matrix and the frustum axes, with the
    // arithmetic of mmt_frustum_geometry (mmt_camera.h) -- no geom tensor is read
    const int npts = fH * dn;
    if constexpr (CAM) {
        // a thread takes ONE depth bin of a block of 16 image rows, 8 rows at a time (two passes keep the register count at
        // that of the walk): for a fixed (camera, column, bin) the coordinates are monotone in the row, so two (x, y)
        // quantisations and 8 exact z range tests settle a pass (mmt_camera.h)
        float cm[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) cm[k] = a.combine[bn * 16 + k];
        const float cu = a.fu[col];
        const int64_t rstep = a.pm ? (int64_t)a.fW * a.D : a.fW;              // points between consecutive image rows
#pragma unroll 1
        for (int r0 = 0; r0 < fH; r0 += 16) {
            const int nr = (fH - r0) < 16 ? (fH - r0) : 16;
#pragma unroll 1
            for (int dd = tid; dd < dn; dd += kBlock) {
                const mmt_cam_column cc = mmt_cam_column_make(cm, cu, a.fd[d0 + dd]);
                int2 *sum = a.summary ? a.summary + (((int64_t)bn * ((fH + 15) >> 4) + (r0 >> 4)) * a.fW + col) * a.D + d0 + dd : nullptr;
                const bool cached = sum && a.summary_cached;                   // the calibration is the one the summary was written for
                int2 sv = make_int2(-1, 0);
                if (cached) sv = *sum;
                unsigned zm16 = 0;
                bool uni16 = true, in00 = false, any = false;
                int x00 = 0, y00 = 0;
#pragma unroll 1
                for (int hb = 0; hb < 16; hb += 8) {
                    if (hb >= nr) break;
                    const int nh = (nr - hb) < 8 ? (nr - hb) : 8;
                    const int64_t t0 = ray_point(a, bn, r0 + hb, col, d0 + dd);
                    float dv[8], cv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        dv[u] = Elem<FT>::scalar(depth + t0 + (u < nh ? u : nh - 1) * rstep);
                        cv[u] = a.fv[r0 + hb + (u < nh ? u : nh - 1)];
                    }
                    bool uniform, in0;
                    int x0, y0;
                    unsigned zmask;
                    if (cached) {
                        in0 = sv.x >= 0; x0 = sv.x & 0xFFFF; y0 = sv.x >> 16;
                        zmask = ((unsigned)sv.y >> hb) & 0xFFu;
                        uniform = __all((sv.y & mmt::kSummaryUniform) != 0);
                    } else {
                        zmask = mmt_cam_column_cells<8>(cc, cv, nh, mmt_rows_sorted<8>(cv, nh), a.q, a.nx, a.ny, a.nz, uniform, in0, x0, y0);
                        zm16 |= zmask << hb;
                        if (hb == 0) { uni16 = uniform; in00 = in0; x00 = x0; y00 = y0; }
                        else uni16 = uni16 && uniform && in0 == in00 && x0 == x00 && y0 == y00;
                    }
                    const int cell0 = in0 ? (b * a.ny + y0) * a.nx + x0 : -1;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (u < nh) {
                            int gx = x0, gy = y0, cell = cell0;
                            if (!uniform) cell = mmt_cam_row_xy(cc, cv[u], a.q, a.nx, a.ny, gx, gy) ? (b * a.ny + gy) * a.nx + gx : -1;
                            const bool keep = ((zmask >> u) & 1u) && cell >= 0;
                            any = any || keep;
                            rec[(r0 + hb + u) * dnp + dd] = make_int2(keep ? cell : -1, keep ? __float_as_int(dv[u]) : 0);
                            if (a.pos_memo) {
                                const int64_t t = t0 + u * rstep;
                                if (keep) {
                                    a.pos_memo[t * 3] = b; a.pos_memo[t * 3 + 1] = gy; a.pos_memo[t * 3 + 2] = gx;
                                } else if (a.write_dropped) {
                                    a.pos_memo[t * 3] = -1; a.pos_memo[t * 3 + 1] = -1; a.pos_memo[t * 3 + 2] = -1;
                                }
                            }
                        }
                    }
                }
                if (sum && !cached) *sum = make_int2(in00 ? ((y00 << 16) | x00) : -1, (int)zm16 | (uni16 ? mmt::kSummaryUniform : 0));
                if (a.compact) binkept[dd] = (r0 == 0 ? 0 : binkept[dd]) | (any ? 1 : 0);     // this thread owns bin dd in every row block
            }
        }
    } else {
    if (a.compact) {
        for (int dd = tid; dd < dn; dd += kBlock) binkept[dd] = 0;
        __syncthreads();
    }
    for (int p0 = tid; p0 < npts; p0 += kBlock * 4) {
        int gx[4], gy[4], gz[4];
        float dv[4];
        int64_t t[4];
        int row_of[4], dd_of[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * kBlock;
            const int pc = p < npts ? p : 0;
            const int row = pc / dn, dd = pc - row * dn;
            row_of[u] = row; dd_of[u] = dd;
            t[u] = ray_point(a, bn, row, col, d0 + dd);
            gx[u] = a.geom[t[u] * 3]; gy[u] = a.geom[t[u] * 3 + 1]; gz[u] = a.geom[t[u] * 3 + 2];
            dv[u] = Elem<FT>::scalar(depth + t[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * kBlock;
            if (p < npts) {
                const bool keep = !(gx[u] < 0 || gx[u] >= a.nx || gy[u] < 0 || gy[u] >= a.ny || gz[u] < 0 || gz[u] >= a.nz);
                rec[row_of[u] * dnp + dd_of[u]] = make_int2(keep ? (b * a.ny + gy[u]) * a.nx + gx[u] : -1, keep ? __float_as_int(dv[u]) : 0);
                if (keep && a.compact) binkept[dd_of[u]] = 1;  // (every writer writes the same value)
                if (a.pos_memo) {
                    if (keep) {
                        a.pos_memo[t[u] * 3] = b; a.pos_memo[t[u] * 3 + 1] = gy[u]; a.pos_memo[t[u] * 3 + 2] = gx[u];
                    } else if (a.write_dropped) {
                        a.pos_memo[t[u] * 3] = -1; a.pos_memo[t[u] * 3 + 1] = -1; a.pos_memo[t[u] * 3 + 2] = -1;
                    }
                }
            }
        }
    }
    }
    {
        constexpr int VEC = Elem<FT>::VEC;
        constexpr int CV = C / VEC;
        for (int e = tid; e < fH * CV; e += kBlock) {
            const int row = e / CV, cv = e - row * CV;
            Elem<FT>::to_lds(context + ((int64_t)bn * HW + row * a.fW + col) * C + cv * VEC, ctx + row * CP + cv * VEC);
        }
    }
    __syncthreads();
    // long rays (a.compact; the reference's native 409-bin frustum: 84 % of the points lie beyond the grid): the bins with a
    // kept point, in depth order (wave 0: ballot + popcount), so that the lane groups share the work that exists.  Short rays
    // keep the plain split -- the extra barrier costs them more than the few empty bins (cfg4: +3 us).
    int nkb = dn;
    if (a.compact) {
        if (wave == 0) {
            int n = 0;
            for (int base = 0; base < dn; base += 64) {
                const bool k = (base + lane) < dn && binkept[base + lane] != 0;
                const unsigned long long m = __ballot(k);
                if (k) klist[n + __popcll(m & ((1ull << lane) - 1ull))] = base + lane;
                n += __popcll(m);
            }
            if (lane == 0) s_nkb = n;
        }
        __syncthreads();
        nkb = s_nkb;
    }
    FWD_STAMP(1);

    // ---- 16 lane groups (4 per wave): an equal share of the (kept) depth bins x all fH rows each, run-length sums in
    // registers (no barrier below)
    const int g = lane >> 4, li = lane & 15;
    const int share = a.compact ? (nkb + kBlock / 16 - 1) / (kBlock / 16) : a.kd;
    // (Rotating the lane groups' shares by the column index -- so that the nearest bins of a camera's 44 columns, which fall
    // into the same few cells, do not send their atomics together -- was measured: no difference, 34.5 us either way.)
    const int ds = (wave * 4 + g) * share;
    const int de = (ds + share) < nkb ? (ds + share) : nkb;
    float acc[S];
#pragma unroll
    for (int j = 0; j < S; ++j) acc[j] = 0.f;
    int cur = -1;
    const float *cl = ctx + li;
    auto flush = [&]() __attribute__((always_inline)) {
        float *o = a.out + (int64_t)cur * C + li;
#pragma unroll
        for (int j = 0; j < S; ++j) unsafeAtomicAdd(o + 16 * j, acc[j]);
    };
    // SKIP (long rays only): four rows that are dropped in every lane group of the wave -- the rows above / below the z range
    // at the far bins, 3/4 of a 44-row column at the reference's native frustum -- are passed over before their context rows
    // are read.  The test sits between the record reads and the context reads, which costs a short ray more than it saves
    // (cfg4: +3 us), so it is compiled only into the long-ray path.
    auto walk = [&](auto skip) __attribute__((always_inline)) {
        constexpr bool SKIP = decltype(skip)::value;
        for (int e = ds; e < de; ++e) {
            const int dd = SKIP ? klist[e] : e;
            for (int r0 = 0; r0 < fH; r0 += 4) {
                int2 kr[4];
                float c[4][S];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = (r0 + u) < fH ? (r0 + u) : (fH - 1);
                    kr[u] = rec[row * dnp + dd];
                    if ((r0 + u) >= fH) kr[u] = make_int2(-1, 0);
                    if constexpr (!SKIP) {
#pragma unroll
                        for (int j = 0; j < S; ++j) c[u][j] = cl[row * CP + 16 * j];
                    }
                }
                if constexpr (SKIP) {
                    if (!__any((kr[0].x & kr[1].x & kr[2].x & kr[3].x) >= 0)) continue;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int row = (r0 + u) < fH ? (r0 + u) : (fH - 1);
#pragma unroll
                        for (int j = 0; j < S; ++j) c[u][j] = cl[row * CP + 16 * j];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int key = kr[u].x;
                    if (key >= 0 && key != cur) {
                        if (cur >= 0) flush();
                        cur = key;
#pragma unroll
                        for (int j = 0; j < S; ++j) acc[j] = 0.f;
                    }
                    const float dv = __int_as_float(kr[u].y);      // 0 for dropped points
#pragma unroll
                    for (int j = 0; j < S; ++j) acc[j] = __builtin_fmaf(dv, c[u][j], acc[j]);
                }
            }
        }
    };
    if (a.compact) walk(std::true_type{});
    else walk(std::false_type{});
    if (cur >= 0) flush();
#ifdef LSS_STAMPS
    if (lane == 0 && fstamps) fstamps[2 + wave] = __builtin_amdgcn_s_memtime();      // end of the walk (atomics issued, not retired)
#endif
}

// ---------------------------------------------------------------------------
// Forward for columns of up to 16 image rows (every BASELINE config but cfg5): the REGISTER walk.
// The walk of lss_ray_fwd is bound by instruction issue, not by memory: a wave64 instruction holds its SIMD16 for four
// cycles, and per point it spends ~17 of them (record read, key compare, S context reads, S multiply-adds, loop) --
// 16 us of the kernel's 34 at cfg4 (measured with the walk compiled out / the flush compiled out).  Here
//   * a lane group keeps the column's WHOLE context tile in registers (16 rows x S channels: loaded from LDS once, not once
//     per depth bin), so a point costs its multiply-adds only -- two channels per v_pk_fma_f32;
//   * the geometry phase, which knows it anyway, leaves ONE key per (depth bin): the cell shared by all kept rows of the bin
//     (the rule on a level rig), "no kept row" (the bin is skipped) or "mixed" (the rows are walked one by one from LDS, as
//     before); the 16 depths of a bin arrive as four 16-byte LDS reads;
//   * 128 threads per column (8 lane groups): the geometry phase has one thread per depth bin and 112 bins -- the second
//     half of a 256-thread workgroup idled through it -- and ~110 VGPRs x 2 waves leave room for every column of the
//     launch to be resident at once.
// The sums are formed in the order of lss_ray_fwd (bins ascending, rows ascending, one fused multiply-add per point).
constexpr int kRegBlock = 128;
constexpr int kBinStride = 20;      // LDS floats per depth bin: 16 rows + 4 of padding (16-byte vectors of neighbouring bins on different banks)
typedef float mmt_v2f __attribute__((ext_vector_type(2)));

__host__ __device__ inline size_t ray_fwd_reg_lds(int C, int dspan) { return (size_t)16 * (C + 4) * 4 + (size_t)dspan * (kBinStride * 8 + 4); }

template <typename FT, int S, bool CAM>
__global__ __launch_bounds__(kRegBlock, 4) void lss_ray_fwd_reg(RayArgs a) {
    extern __shared__ __align__(16) float ray_lds[];
    constexpr int C = 16 * S, CP = C + 4, NT = kRegBlock, BS = kBinStride, NP = S / 2;
    const int L = blockIdx.x;
    const int xcd = L & 7, i = L >> 3;
    const int per = a.fW * a.dsplit;
    const int q = i / per, r = i - q * per;
    const int bn = q * 8 + xcd;                       // the columns of one camera share an XCD (context rows, geom lines)
    if (bn >= a.BN) return;
    const int col = r / a.dsplit, slab = r - col * a.dsplit;
    const int d0 = slab * a.dspan;
    const int dn = (a.D - d0) < a.dspan ? (a.D - d0) : a.dspan;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fH = a.fH, HW = a.fH * a.fW;            // fH <= 16
    const int b = bn / a.N;
    const FT *depth = reinterpret_cast<const FT *>(a.depth);
    const FT *context = reinterpret_cast<const FT *>(a.context);
    float *ctx = ray_lds;                                             // [fH][CP] fp32 context rows
    float *dep = ray_lds + 16 * CP;                                   // [dspan][BS]: depths of a bin's rows, 0 for a dropped or missing row
    int *keyrow = reinterpret_cast<int *>(dep + (size_t)a.dspan * BS); // [dspan][BS]: cell (| kExclFlag) of a row, -1 dropped
    int *bkey = keyrow + (size_t)a.dspan * BS;                        // [dspan]: the bin's key (see above): cell, -1 nothing kept, -2 mixed
#ifdef LSS_STAMPS
    unsigned long long *fstamps = reinterpret_cast<unsigned long long *>(a.pos_memo) + (int64_t)blockIdx.x * 8;
    a.pos_memo = nullptr;
#endif
    FWD_STAMP(0);
    // exclusive-cell cache: this wave looks its sample up (loads only; resolved behind the geometry phase).  The slot is
    // known early, so the geometry phase reads its states SPECULATIVELY -- tags that turn out not to be ours are stripped
    ExclProbe probe = {};
    const int32_t *excl_st = nullptr;                 // the slot's int32 per cell, indexed by the call's cell id
    bool excl_on = false;
    if constexpr (CAM) {
        excl_on = a.excl != nullptr && b < kExclMaxB;
        if (excl_on) probe = excl_probe_begin(a.excl, a.xs, a.combine, b);      // (loads only; excl_st follows at its first use)
    }

    // ---- geometry -> per-bin depths and keys
    if constexpr (CAM) {
        float cm[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) cm[k] = a.combine[bn * 16 + k];
        const float cu = a.fu[col];
        const int64_t rstep = a.pm ? (int64_t)a.fW * a.D : a.fW;              // points between consecutive image rows
#pragma unroll 1
        for (int dd = tid; dd < dn; dd += NT) {
            const mmt_cam_column cc = mmt_cam_column_make(cm, cu, a.fd[d0 + dd]);
            int2 *sum = a.summary ? a.summary + ((int64_t)bn * a.fW + col) * a.D + d0 + dd : nullptr;      // one block of rows: fH <= 16
            const bool cached = sum && a.summary_cached;
            int2 sv = make_int2(-1, 0);
            if (cached) sv = *sum;
            unsigned zm16 = 0;
            bool uni16 = cached ? (sv.y & mmt::kSummaryUniform) != 0 : true, in00 = false;
            int x00 = 0, y00 = 0, ukey = -1;                                  // ukey: key of the first kept row
            int ustate = 0;                                                   // its cell's state (read as soon as the cell is known)
            bool uasked = false;
#pragma unroll 1
            for (int hb = 0; hb < 16; hb += 8) {
                float dvs[8];
                int ks[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { dvs[u] = 0.f; ks[u] = -1; }
                if (hb < fH) {
                    const int nh = (fH - hb) < 8 ? (fH - hb) : 8;
                    const int64_t t0 = ray_point(a, bn, hb, col, d0 + dd);
                    float dv[8], cv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        dv[u] = Elem<FT>::scalar(depth + t0 + (u < nh ? u : nh - 1) * rstep);
                        cv[u] = a.fv[hb + (u < nh ? u : nh - 1)];
                    }
                    bool uniform, in0;
                    int x0, y0;
                    unsigned zmask;
                    if (cached) {
                        in0 = sv.x >= 0; x0 = sv.x & 0xFFFF; y0 = sv.x >> 16;
                        zmask = ((unsigned)sv.y >> hb) & 0xFFu;
                        uniform = __all((sv.y & mmt::kSummaryUniform) != 0);
                    } else {
                        zmask = mmt_cam_column_cells<8>(cc, cv, nh, mmt_rows_sorted<8>(cv, nh), a.q, a.nx, a.ny, a.nz, uniform, in0, x0, y0);
                        zm16 |= zmask << hb;
                        if (hb == 0) { uni16 = uniform; in00 = in0; x00 = x0; y00 = y0; }
                        else uni16 = uni16 && uniform && in0 == in00 && x0 == x00 && y0 == y00;
                    }
                    const int cell0 = in0 ? (b * a.ny + y0) * a.nx + x0 : -1;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (u < nh) {
                            int gx = x0, gy = y0, cell = cell0;
                            if (!uniform) cell = mmt_cam_row_xy(cc, cv[u], a.q, a.nx, a.ny, gx, gy) ? (b * a.ny + gy) * a.nx + gx : -1;
                            const bool keep = ((zmask >> u) & 1u) && cell >= 0;
                            if (keep) {
                                dvs[u] = dv[u]; ks[u] = cell;
                                if (ukey < 0) ukey = ks[u];
                            }
                            if (a.pos_memo) {
                                const int64_t t = t0 + u * rstep;
                                if (keep) {
                                    a.pos_memo[t * 3] = b; a.pos_memo[t * 3 + 1] = gy; a.pos_memo[t * 3 + 2] = gx;
                                } else if (a.write_dropped) {
                                    a.pos_memo[t * 3] = -1; a.pos_memo[t * 3 + 1] = -1; a.pos_memo[t * 3 + 2] = -1;
                                }
                            }
                        }
                    }
                }
                if (excl_on && !uasked && ukey >= 0) {                                                // (consumed behind the next pass)
                    if (excl_st == nullptr) excl_st = excl_state(a.excl, a.xs, excl_probe_slot(probe)) - (int64_t)b * a.xs.cells;
                    ustate = excl_st[ukey]; uasked = true;
                }
                float4 *dq = reinterpret_cast<float4 *>(dep + dd * BS + hb);
                int4 *kq = reinterpret_cast<int4 *>(keyrow + dd * BS + hb);
                dq[0] = make_float4(dvs[0], dvs[1], dvs[2], dvs[3]); dq[1] = make_float4(dvs[4], dvs[5], dvs[6], dvs[7]);
                kq[0] = make_int4(ks[0], ks[1], ks[2], ks[3]); kq[1] = make_int4(ks[4], ks[5], ks[6], ks[7]);
            }
            if (sum && !cached) *sum = make_int2(in00 ? ((y00 << 16) | x00) : -1, (int)zm16 | (uni16 ? mmt::kSummaryUniform : 0));
            if (excl_on && ukey >= 0 && !uni16) {                             // mixed bin (rare): the rows' tags one by one
                if (excl_st == nullptr) excl_st = excl_state(a.excl, a.xs, excl_probe_slot(probe)) - (int64_t)b * a.xs.cells;
                for (int row = 0; row < fH; ++row) {
                    const int kk = keyrow[dd * BS + row];
                    if (kk >= 0 && excl_st[kk] > 0) keyrow[dd * BS + row] = kk | kExclFlag;
                }
            }
            bkey[dd] = ukey < 0 ? -1 : (uni16 ? (ukey | (ustate > 0 ? kExclFlag : 0)) : -2);
        }
    } else {
        const int npts = fH * dn;
        for (int idx = tid; idx < dn * 16; idx += NT) {                       // rows past fH: nothing
            const int dd = idx >> 4, row = idx & 15;
            if (row >= fH) { dep[dd * BS + row] = 0.f; keyrow[dd * BS + row] = -1; }
        }
        for (int p0 = tid; p0 < npts; p0 += NT * 4) {
            int gx[4], gy[4], gz[4];
            float dv[4];
            int64_t t[4];
            int row_of[4], dd_of[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + u * NT;
                const int pc = p < npts ? p : 0;
                const int row = pc / dn, dd = pc - row * dn;
                row_of[u] = row; dd_of[u] = dd;
                t[u] = ray_point(a, bn, row, col, d0 + dd);
                gx[u] = a.geom[t[u] * 3]; gy[u] = a.geom[t[u] * 3 + 1]; gz[u] = a.geom[t[u] * 3 + 2];
                dv[u] = Elem<FT>::scalar(depth + t[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + u * NT;
                if (p < npts) {
                    const bool keep = !(gx[u] < 0 || gx[u] >= a.nx || gy[u] < 0 || gy[u] >= a.ny || gz[u] < 0 || gz[u] >= a.nz);
                    dep[dd_of[u] * BS + row_of[u]] = keep ? dv[u] : 0.f;
                    keyrow[dd_of[u] * BS + row_of[u]] = keep ? (b * a.ny + gy[u]) * a.nx + gx[u] : -1;
                    if (a.pos_memo) {
                        if (keep) {
                            a.pos_memo[t[u] * 3] = b; a.pos_memo[t[u] * 3 + 1] = gy[u]; a.pos_memo[t[u] * 3 + 2] = gx[u];
                        } else if (a.write_dropped) {
                            a.pos_memo[t[u] * 3] = -1; a.pos_memo[t[u] * 3 + 1] = -1; a.pos_memo[t[u] * 3 + 2] = -1;
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int dd = tid; dd < dn; dd += NT) {                              // the bin's key from its rows' keys
            int ukey = -1;
            bool same = true;
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4) {
                const int4 k4 = *reinterpret_cast<const int4 *>(keyrow + dd * BS + v4 * 4);
                const int kk[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (kk[u] >= 0) {
                        if (ukey < 0) ukey = kk[u];
                        else same = same && kk[u] == ukey;
                    }
                }
            }
            bkey[dd] = ukey < 0 ? -1 : (same ? ukey : -2);
        }
    }
    {
        constexpr int VEC = Elem<FT>::VEC;
        constexpr int CV = C / VEC;
        for (int e = tid; e < fH * CV; e += NT) {
            const int row = e / CV, cv = e - row * CV;
            Elem<FT>::to_lds(context + ((int64_t)bn * HW + row * a.fW + col) * C + cv * VEC, ctx + row * CP + cv * VEC);
        }
    }
    __syncthreads();
    // ---- exclusive-cell cache: the probe's answer.  Stage 3 (verified): the tags stand.  Anything else: they were read from
    // another calibration's (or an unfinished) table -- strip them; stages 1 / 2 mark / verify in the flush below
    int excl_mode = 0;
    if constexpr (CAM) {
        if (excl_on) {
            excl_probe_end(probe);
            excl_mode = probe.mode;
            excl_st = excl_state(a.excl, a.xs, probe.slot) - (int64_t)b * a.xs.cells;       // (the slot the mark / verify stores go to)
            if (wave == 0 && bn == b * a.N && col == 0 && slab == 0) excl_post(a.excl, a.xs, a.combine, b, probe);
            if (excl_mode != 3) {
                for (int dd = tid; dd < dn; dd += NT) {
                    const int k = bkey[dd];
                    if (k >= 0) bkey[dd] = k & (kExclFlag - 1);
                    else if (k == -2)
                        for (int row = 0; row < fH; ++row) { const int kk = keyrow[dd * BS + row]; if (kk >= 0) keyrow[dd * BS + row] = kk & (kExclFlag - 1); }
                }
                __syncthreads();
            }
        }
    }
    FWD_STAMP(1);
#ifdef LSS_EXP_NOWALK
    if (a.D != 12345) return;
#endif

    // ---- 8 lane groups: an equal share of the depth bins each; the context tile in registers
    const int g = lane >> 4, li = lane & 15;
    const int share = (dn + NT / 16 - 1) / (NT / 16);
    // Shares end where a run ends: a nominal boundary that falls inside a run (the bin before it has the same key) moves on to
    // the run's last bin, so no cell's run is cut in two between neighbouring lane groups -- each cut was one more flush (one
    // more atomic row, or a cell that could have been a plain store) per boundary: 7 of the ~52 flushes of a column at
    // BASELINE configs[3].  The adjusted boundaries are a function of the keys alone: every group finds its own and its
    // successor's without a barrier (a run is 1-3 bins).
    auto run_start = [&](int p) __attribute__((always_inline)) {
        if (p >= dn) return dn;
        while (p > 0 && p < dn) {
            const int k = bkey[p];
            if (!(k >= 0 && k == bkey[p - 1])) break;
            ++p;
        }
        return p;
    };
    const int ds = run_start((wave * 4 + g) * share);
    const int de = run_start((wave * 4 + g + 1) * share);
    const float *cl = ctx + li;
    mmt_v2f c2[16][NP > 0 ? NP : 1];
    float c1[16];
#pragma unroll
    for (int row = 0; row < 16; ++row) {
        const bool live = row < fH;
        const float *cr = cl + (live ? row : 0) * CP;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const float lo = cr[16 * (2 * p)], hi = cr[16 * (2 * p + 1)];
            c2[row][p].x = live ? lo : 0.f; c2[row][p].y = live ? hi : 0.f;
        }
        if constexpr (S & 1) { const float v = cr[16 * (S - 1)]; c1[row] = live ? v : 0.f; }
    }
    mmt_v2f acc2[NP > 0 ? NP : 1];
    float acc1 = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) acc2[p] = mmt_v2f{0.f, 0.f};
    int cur = -1, run_at = 0;                        // run_at: (bin, row) the current run began at
    // a run's id: the same in every launch of this shape, whatever the sample's place in the batch
    const int run_base = (((((bn - b * a.N) * a.fW + col) * a.dsplit + slab) * (NT / 16) + wave * 4 + g) * a.dspan) * 16 + 1;
    auto flush = [&]() __attribute__((always_inline)) {
#ifdef LSS_EXP_NOFLUSH
        if (acc1 != 12345.f) return;
#endif
        const int cellid = cur & (kExclFlag - 1);
        float *o = a.out + (int64_t)cellid * C + li;
        if (cur & kExclFlag) {                      // the only run of the launch into this cell: the zero-filled row is simply written
#pragma unroll
            for (int p = 0; p < NP; ++p) { o[16 * (2 * p)] = acc2[p].x; o[16 * (2 * p + 1)] = acc2[p].y; }
            if constexpr (S & 1) o[16 * (S - 1)] = acc1;
        } else {
#pragma unroll
            for (int p = 0; p < NP; ++p) { unsafeAtomicAdd(o + 16 * (2 * p), acc2[p].x); unsafeAtomicAdd(o + 16 * (2 * p + 1), acc2[p].y); }
            if constexpr (S & 1) unsafeAtomicAdd(o + 16 * (S - 1), acc1);
            if constexpr (CAM) {
                if (li == 0 && (excl_mode == 1 || excl_mode == 2)) {
                    int32_t *st = const_cast<int32_t *>(excl_st) + cellid;
                    const int id = run_base + run_at;
                    if (excl_mode == 1) *st = id;                                   // MARK: some run's id survives
                    else if (*st != id) *st = -1;                                   // VERIFY: the cell has another run
                }
            }
        }
    };
    auto renew = [&](int key, int at) __attribute__((always_inline)) {
        if (cur >= 0) flush();
        cur = key; run_at = at;
#pragma unroll
        for (int p = 0; p < NP; ++p) acc2[p] = mmt_v2f{0.f, 0.f};
        acc1 = 0.f;
    };
#pragma unroll 1
    for (int e = ds; e < de; ++e) {
        const int k = bkey[e];
        if (k == -1) continue;
        if (k >= 0) {
            if (k != cur) renew(k, e * 16);
            const float4 *dp = reinterpret_cast<const float4 *>(dep + e * BS);
            const float4 d0v = dp[0], d1v = dp[1], d2v = dp[2], d3v = dp[3];
            const float dv[16] = {d0v.x, d0v.y, d0v.z, d0v.w, d1v.x, d1v.y, d1v.z, d1v.w, d2v.x, d2v.y, d2v.z, d2v.w, d3v.x, d3v.y, d3v.z, d3v.w};
#pragma unroll
            for (int row = 0; row < 16; ++row) {
                const mmt_v2f d2 = {dv[row], dv[row]};
#pragma unroll
                for (int p = 0; p < NP; ++p) acc2[p] = __builtin_elementwise_fma(d2, c2[row][p], acc2[p]);
                if constexpr (S & 1) acc1 = __builtin_fmaf(dv[row], c1[row], acc1);
            }
        } else {                                    // mixed bin (rare): row by row, context from LDS
#pragma unroll 1
            for (int row = 0; row < fH; ++row) {
                const int key = keyrow[e * BS + row];
                const float dv = dep[e * BS + row];
                if (key >= 0 && key != cur) renew(key, e * 16 + row);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    acc2[p].x = __builtin_fmaf(dv, cl[row * CP + 16 * (2 * p)], acc2[p].x);
                    acc2[p].y = __builtin_fmaf(dv, cl[row * CP + 16 * (2 * p + 1)], acc2[p].y);
                }
                if constexpr (S & 1) acc1 = __builtin_fmaf(dv, cl[row * CP + 16 * (S - 1)], acc1);
            }
        }
    }
    if (cur >= 0) flush();
#ifdef LSS_STAMPS
    if (lane == 0 && fstamps) fstamps[2 + wave] = __builtin_amdgcn_s_memtime();
#endif
}

// ---------------------------------------------------------------------------
// Forward for the camera form's other shapes -- columns of more than 16 image rows (BASELINE configs[4]: 32) and long rays
// (the reference's native frustum: 409 bins x 44 rows) -- the BLOCK walk.  The register walk's arithmetic with the context
// rows read from LDS: the geometry phase leaves ONE key per (depth bin, block of 16 rows) -- the cell all kept rows of the
// block share / nothing kept (the block costs the walk one LDS read; at the native frustum 84 % of the points lie beyond
// the grid) / mixed (walked row by row: the rows' cells are computed again from the camera's matrix, rare) -- and a block's
// 16 depths as four 16-byte LDS vectors; a point then costs its context reads and packed multiply-adds (~9 wave-
// instructions against ~17 of lss_ray_fwd's record walk, which was bound by instruction issue).  Same order of additions
// as lss_ray_fwd: bins ascending, rows ascending.  256 threads: 16 lane groups share the (kept) depth bins.
constexpr int kBlkStride = 20;      // LDS floats per (bin, row block): 16 rows + padding

__host__ __device__ inline size_t ray_fwd_blk_lds(int fH, int C, int dspan) {
    const int RB = (fH + 15) / 16;
    return (size_t)fH * (C + 4) * 4 + (size_t)dspan * RB * (kBlkStride * 4 + 4) + (size_t)dspan * 8;
}

template <typename FT, int S>
__global__ __launch_bounds__(kBlock) void lss_ray_fwd_blk(RayArgs a) {
    extern __shared__ __align__(16) float ray_lds[];
    constexpr int C = 16 * S, CP = C + 4, BS = kBlkStride, NP = S / 2;
    const int L = blockIdx.x, xcd = L & 7, i = L >> 3;
    const int per = a.fW * a.dsplit;
    const int q = i / per, r = i - q * per;
    const int bn = q * 8 + xcd;                       // the columns of one camera share an XCD
    if (bn >= a.BN) return;
    const int col = r / a.dsplit, slab = r - col * a.dsplit;
    const int d0 = slab * a.dspan;
    const int dn = (a.D - d0) < a.dspan ? (a.D - d0) : a.dspan;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fH = a.fH, HW = a.fH * a.fW, RB = (fH + 15) >> 4;
    const int b = bn / a.N;
    const FT *depth = reinterpret_cast<const FT *>(a.depth);
    const FT *context = reinterpret_cast<const FT *>(a.context);
    float *ctx = ray_lds;                                              // [fH][CP] fp32 context rows
    float *dep = ray_lds + fH * CP;                                    // [dspan][RB][BS]: depths of a block's rows, 0 for a dropped / missing row
    int *bkey = reinterpret_cast<int *>(dep + (size_t)a.dspan * RB * BS);   // [dspan][RB]: cell, -1 nothing kept, -2 mixed
    int *binkept = bkey + a.dspan * RB;                                // [dspan]: some block of the bin holds a kept row
    int *klist = binkept + a.dspan;                                    // the kept bins in depth order (long rays)
    __shared__ int s_nkb;
    float cm[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) cm[k] = a.combine[bn * 16 + k];
    const float cu = a.fu[col];
    const int64_t rstep = a.pm ? (int64_t)a.fW * a.D : a.fW;              // points between consecutive image rows

    // ---- geometry -> per-block depths and keys: a thread takes one (depth bin, row block) at a time, 8 rows per pass
    for (int dd = tid; dd < dn; dd += kBlock) binkept[dd] = 0;
    __syncthreads();
#pragma unroll 1
    for (int it = tid; it < dn * RB; it += kBlock) {
        const int rb = it / dn, dd = it - rb * dn;
        const int r0 = rb * 16;
        const int nr = (fH - r0) < 16 ? (fH - r0) : 16;
        {
            const mmt_cam_column cc = mmt_cam_column_make(cm, cu, a.fd[d0 + dd]);
            int2 *sum = a.summary ? a.summary + (((int64_t)bn * RB + rb) * a.fW + col) * a.D + d0 + dd : nullptr;
            const bool cached = sum && a.summary_cached;
            int2 sv = make_int2(-1, 0);
            if (cached) sv = *sum;
            unsigned zm16 = 0;
            bool uni16 = cached ? (sv.y & mmt::kSummaryUniform) != 0 : true, in00 = false;
            int x00 = 0, y00 = 0, ukey = -1;
#pragma unroll 1
            for (int hb = 0; hb < 16; hb += 8) {
                float dvs[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) dvs[u] = 0.f;
                if (hb < nr) {
                    const int nh = (nr - hb) < 8 ? (nr - hb) : 8;
                    const int64_t t0 = ray_point(a, bn, r0 + hb, col, d0 + dd);
                    float dv[8], cv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        dv[u] = Elem<FT>::scalar(depth + t0 + (u < nh ? u : nh - 1) * rstep);
                        cv[u] = a.fv[r0 + hb + (u < nh ? u : nh - 1)];
                    }
                    bool uniform, in0;
                    int x0, y0;
                    unsigned zmask;
                    if (cached) {
                        in0 = sv.x >= 0; x0 = sv.x & 0xFFFF; y0 = sv.x >> 16;
                        zmask = ((unsigned)sv.y >> hb) & 0xFFu;
                        uniform = __all((sv.y & mmt::kSummaryUniform) != 0);
                    } else {
                        zmask = mmt_cam_column_cells<8>(cc, cv, nh, mmt_rows_sorted<8>(cv, nh), a.q, a.nx, a.ny, a.nz, uniform, in0, x0, y0);
                        zm16 |= zmask << hb;
                        if (hb == 0) { uni16 = uniform; in00 = in0; x00 = x0; y00 = y0; }
                        else uni16 = uni16 && uniform && in0 == in00 && x0 == x00 && y0 == y00;
                    }
                    const int cell0 = in0 ? (b * a.ny + y0) * a.nx + x0 : -1;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (u < nh) {
                            int gx = x0, gy = y0, cell = cell0;
                            if (!uniform) cell = mmt_cam_row_xy(cc, cv[u], a.q, a.nx, a.ny, gx, gy) ? (b * a.ny + gy) * a.nx + gx : -1;
                            const bool keep = ((zmask >> u) & 1u) && cell >= 0;
                            if (keep) {
                                dvs[u] = dv[u];
                                if (ukey < 0) ukey = cell;
                            }
                            if (a.pos_memo) {
                                const int64_t t = t0 + u * rstep;
                                if (keep) {
                                    a.pos_memo[t * 3] = b; a.pos_memo[t * 3 + 1] = gy; a.pos_memo[t * 3 + 2] = gx;
                                } else if (a.write_dropped) {
                                    a.pos_memo[t * 3] = -1; a.pos_memo[t * 3 + 1] = -1; a.pos_memo[t * 3 + 2] = -1;
                                }
                            }
                        }
                    }
                }
                float4 *dq = reinterpret_cast<float4 *>(dep + ((size_t)dd * RB + rb) * BS + hb);
                dq[0] = make_float4(dvs[0], dvs[1], dvs[2], dvs[3]); dq[1] = make_float4(dvs[4], dvs[5], dvs[6], dvs[7]);
            }
            if (sum && !cached) *sum = make_int2(in00 ? ((y00 << 16) | x00) : -1, (int)zm16 | (uni16 ? mmt::kSummaryUniform : 0));
            bkey[dd * RB + rb] = ukey < 0 ? -1 : (uni16 ? ukey : -2);
            if (ukey >= 0) binkept[dd] = 1;                                           // (every writer writes the same value)
        }
    }
    {
        constexpr int VEC = Elem<FT>::VEC;
        constexpr int CV = C / VEC;
        for (int e = tid; e < fH * CV; e += kBlock) {
            const int row = e / CV, cv = e - row * CV;
            Elem<FT>::to_lds(context + ((int64_t)bn * HW + row * a.fW + col) * C + cv * VEC, ctx + row * CP + cv * VEC);
        }
    }
    __syncthreads();
    // the bins with a kept point, in depth order (wave 0: ballot + popcount): the lane groups share the work that exists
    if (wave == 0) {
        int n = 0;
        for (int base = 0; base < dn; base += 64) {
            const bool k = (base + lane) < dn && binkept[base + lane] != 0;
            const unsigned long long m = __ballot(k);
            if (k) klist[n + __popcll(m & ((1ull << lane) - 1ull))] = base + lane;
            n += __popcll(m);
        }
        if (lane == 0) s_nkb = n;
    }
    __syncthreads();
    const int nkb = s_nkb;

    // ---- 16 lane groups: an equal share of the kept bins x all row blocks each
    const int g = lane >> 4, li = lane & 15;
    const int share = (nkb + kBlock / 16 - 1) / (kBlock / 16);
    // (Shares that end where a run ends -- lss_ray_fwd_reg's rule -- were measured here and dropped: +3 us at BASELINE
    // configs[4] (44.0 -> 47.0 us, tools/kbench_camera.py interleaved).  70 kept bins over 16 lane groups are 4-5 bins each;
    // moving a boundary to the next run end makes the longest share 6-7 bins, and the longest share is the kernel's walk time.)
    const int ds = (wave * 4 + g) * share;
    const int de = (ds + share) < nkb ? (ds + share) : nkb;
    const float *cl = ctx + li;
    mmt_v2f acc2[NP > 0 ? NP : 1];
    float acc1 = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) acc2[p] = mmt_v2f{0.f, 0.f};
    int cur = -1;
    auto flush = [&]() __attribute__((always_inline)) {
        float *o = a.out + (int64_t)cur * C + li;
#pragma unroll
        for (int p = 0; p < NP; ++p) { unsafeAtomicAdd(o + 16 * (2 * p), acc2[p].x); unsafeAtomicAdd(o + 16 * (2 * p + 1), acc2[p].y); }
        if constexpr (S & 1) unsafeAtomicAdd(o + 16 * (S - 1), acc1);
    };
    auto renew = [&](int key) __attribute__((always_inline)) {
        if (cur >= 0) flush();
        cur = key;
#pragma unroll
        for (int p = 0; p < NP; ++p) acc2[p] = mmt_v2f{0.f, 0.f};
        acc1 = 0.f;
    };
    auto add_row = [&](float dv, int row) __attribute__((always_inline)) {
        const float *cr = cl + row * CP;
        const mmt_v2f d2 = {dv, dv};
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const mmt_v2f c2 = {cr[16 * (2 * p)], cr[16 * (2 * p + 1)]};
            acc2[p] = __builtin_elementwise_fma(d2, c2, acc2[p]);
        }
        if constexpr (S & 1) acc1 = __builtin_fmaf(dv, cr[16 * (S - 1)], acc1);
    };
#pragma unroll 1
    for (int e = ds; e < de; ++e) {
        const int dd = klist[e];
#pragma unroll 1
        for (int rb = 0; rb < RB; ++rb) {
            const int k = bkey[dd * RB + rb];
            if (k == -1) continue;
            const int r0 = rb * 16;
            const int nr = (fH - r0) < 16 ? (fH - r0) : 16;
            const float4 *dp = reinterpret_cast<const float4 *>(dep + ((size_t)dd * RB + rb) * BS);
            const float4 d0v = dp[0], d1v = dp[1], d2v = dp[2], d3v = dp[3];
            const float dv[16] = {d0v.x, d0v.y, d0v.z, d0v.w, d1v.x, d1v.y, d1v.z, d1v.w, d2v.x, d2v.y, d2v.z, d2v.w, d3v.x, d3v.y, d3v.z, d3v.w};
            if (k >= 0) {
                if (k != cur) renew(k);
                if (nr == 16) {
#pragma unroll
                    for (int row = 0; row < 16; ++row) add_row(dv[row], r0 + row);
                } else {
#pragma unroll
                    for (int row = 0; row < 16; ++row)
                        if (row < nr) add_row(dv[row], r0 + row);
                }
            } else {                                    // mixed block (rare): the rows' cells again from the matrix.  A kept row whose
                                                        // depth is exactly 0 adds nothing and is passed over
                const mmt_cam_column cc = mmt_cam_column_make(cm, cu, a.fd[d0 + dd]);
#pragma unroll
                for (int row = 0; row < 16; ++row) {
                    if (row < nr && dv[row] != 0.f) {
                        int gx, gy;
                        mmt_cam_row_xy(cc, a.fv[r0 + row], a.q, a.nx, a.ny, gx, gy);
                        const int key = (b * a.ny + gy) * a.nx + gx;
                        if (key != cur) renew(key);
                        add_row(dv[row], r0 + row);
                    }
                }
            }
        }
    }
    if (cur >= 0) flush();
}

__device__ __forceinline__ float quad_sum(float v) {   // sum over the 4 lanes of a quad, in all 4 (DPP quad_perm, no LDS)
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));   // [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));   // [2,3,0,1]
    return v;
}

constexpr int kRayBins = 16;        // depth bins per reduction batch of the backward walk (4 sub-batches of 4)

// LDS (dynamic), per lane group: off [Dp] int (byte offset of the BEV-gradient row, or an out-of-range value in the padding) |
// dep [Dp] fp32 | kbin [Dp] int | part [kRayBins][C4/4] fp32
template <typename FT, int C4T, bool CAM>
__global__ __launch_bounds__(kBlock, 6) void lss_ray_bwd(RayArgs a) {
    extern __shared__ __align__(16) float ray_lds[];
    const int L = blockIdx.x, xcd = L & 7, i = L >> 3;
    // (camera, segment of its workgroups) units dealt to the XCDs, a multiple of 8 of them where possible, so that every XCD carries
    // the same number of workgroups (lift_splat_col.hip: 12 cameras dealt whole put twice the work on four of the eight XCDs)
    const int q = i / a.wps, wi = i - q * a.wps;
    const int unit = q * 8 + xcd;
    if (unit >= a.BN * a.split) return;
    const int bn = unit / a.split, w = (unit - bn * a.split) * a.wps + wi;
    if (w >= a.wpc) return;
    const int C = a.C, D = a.D;
    const int C4 = C4T > 0 ? C4T : C >> 2;
    const int G = 64 / C4, Q = C4 >> 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fH = a.fH, HW = a.fH * a.fW;
    const int b = bn / a.N;
    const int g = lane / C4, li = lane - g * C4;
    const int slot = wave * G + (g < G ? g : 0);
    const int pq = w * (kBlock / 64) * G + slot;                  // pixel, columns first: col * fH + row
    const bool act = g < G && pq < HW;
    const int col = act ? pq / fH : 0, row = act ? pq - col * fH : 0;
    const FT *depth = reinterpret_cast<const FT *>(a.depth);
    const FT *context = reinterpret_cast<const FT *>(a.context);
    const int Dp = ((D + kRayBins - 1) & ~(kRayBins - 1)) + 8;    // list entries the walk may ask for (it prefetches into the next batch)
    const int stride = 3 * Dp + kRayBins * Q;
    int *off = reinterpret_cast<int *>(ray_lds) + slot * stride;  // the pixel's KEPT bins, in depth order: row offset,
    float *dep = reinterpret_cast<float *>(off + Dp);             // depth value,
    int *kbin = reinterpret_cast<int *>(dep + Dp);                // bin number
    float *part = reinterpret_cast<float *>(kbin + Dp);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.grad_out), 0, a.span_bytes, 0x00020000);
    const unsigned kOut = 0x80000000u;                            // beyond num_records (< 2^30): the load returns zeros

#ifdef LSS_STAMPS   // diagnostic build: grad_context receives 4 s_memtime stamps per workgroup instead of its rows
    unsigned long long *rstamps = reinterpret_cast<unsigned long long *>(a.grad_context) + (int64_t)blockIdx.x * 4;
#define RAY_STAMP(i) do { if (threadIdx.x == 0) rstamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RAY_STAMP(i) do { } while (0)
#endif
    RAY_STAMP(0);
    // ---- the pixel's D points: kept test, row offset, depth (bins past D and the bins of an idle lane group: nothing)
    const int64_t t0 = ray_point(a, bn, row, col, 0);              // the pixel's bin 0; bin d sits dstep points further
    const int dstep = a.pm ? 1 : HW;
    const int64_t pix = (int64_t)bn * HW + row * a.fW + col;
    float4 cx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {
        if constexpr (sizeof(FT) == 2) {
            const uint2 rr = *reinterpret_cast<const uint2 *>(context + pix * C + li * 4);
            cx = make_float4(bf16_lo(rr.x), bf16_hi(rr.x), bf16_lo(rr.y), bf16_hi(rr.y));
        } else {
            cx = *reinterpret_cast<const float4 *>(context + pix * C + li * 4);
        }
    }
    // Dropped bins (40 % on the analytic rig: beyond the grid or outside the z range) get their zero gradient here and are
    // left out of the walk: every row load of the walk, in or out of range, costs the texture path its 16 cycles.
    int nk = 0;                                                    // kept bins of this lane group's pixel
    {
        constexpr int PA = 6;                                      // bins per lane in flight: the whole ray at D = 112, C = 80
        const unsigned long long gmask = C4 >= 64 ? ~0ull : ((1ull << C4) - 1ull);
        float cm[12];
        float cu = 0.f, cv = 0.f;
        if constexpr (CAM) {                                       // camera form: cells from the matrix (mmt_camera.h), no geom tensor
#pragma unroll
            for (int k = 0; k < 12; ++k) cm[k] = a.combine[bn * 16 + k];
            cu = a.fu[col]; cv = a.fv[row];
        }
        for (int dbase = 0; dbase < D; dbase += PA * C4) {
            int gx[PA], gy[PA], gz[PA];
            float dv[PA];
            float fdd[PA];
#pragma unroll
            for (int u = 0; u < PA; ++u) {
                const int d = dbase + u * C4 + li;
                const int dc = (act && d < D) ? d : 0;
                const int64_t t = t0 + (int64_t)dc * dstep;
                if constexpr (CAM) {
                    fdd[u] = a.fd[dc];
                } else {
                    gx[u] = a.geom[t * 3]; gy[u] = a.geom[t * 3 + 1]; gz[u] = a.geom[t * 3 + 2];
                }
                dv[u] = Elem<FT>::scalar(depth + t);
            }
            if constexpr (CAM) {
                if (a.summary) {       // the forward's column summary: the kept test and the cell cost two dwords per bin
                    const int2 *sum = a.summary + (((int64_t)bn * ((fH + 15) >> 4) + (row >> 4)) * a.fW + col) * (int64_t)D;
                    int2 sv[PA];
#pragma unroll
                    for (int u = 0; u < PA; ++u) {
                        const int d = dbase + u * C4 + li;
                        sv[u] = sum[(act && d < D) ? d : 0];
                    }
                    bool all_uniform = true;
#pragma unroll
                    for (int u = 0; u < PA; ++u) all_uniform = all_uniform && (sv[u].y & mmt::kSummaryUniform);
                    if (__all(all_uniform)) {
#pragma unroll
                        for (int u = 0; u < PA; ++u) {
                            const bool k = sv[u].x >= 0 && ((sv[u].y >> (row & 15)) & 1);
                            gx[u] = k ? (sv[u].x & 0xFFFF) : -1; gy[u] = sv[u].x >> 16; gz[u] = 0;
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < PA; ++u) mmt_cam_cell(cm, cu, cv, fdd[u], a.q, gx[u], gy[u], gz[u]);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < PA; ++u) mmt_cam_cell(cm, cu, cv, fdd[u], a.q, gx[u], gy[u], gz[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < PA; ++u) {
                const int d = dbase + u * C4 + li;
                const bool in = act && d < D;
                const bool keep = in && !(gx[u] < 0 || gx[u] >= a.nx || gy[u] < 0 || gy[u] >= a.ny || gz[u] < 0 || gz[u] >= a.nz);
                const unsigned long long seg = (__ballot(keep) >> (g < G ? g * C4 : 0)) & gmask;   // this group's lanes
                if (keep) {
                    const int pos = nk + __popcll(seg & ((1ull << li) - 1ull));
                    off[pos] = (int)((b * a.sb + gy[u] * a.sy + gx[u] * a.sx) * 4);
                    dep[pos] = dv[u];
                    kbin[pos] = d;
                } else if (in) {
                    const int64_t t = t0 + (int64_t)d * dstep;
                    if constexpr (sizeof(FT) == 2) reinterpret_cast<bf16_t *>(a.grad_depth)[t] = (bf16_t)0;
                    else reinterpret_cast<float *>(a.grad_depth)[t] = 0.f;
                }
                nk += __popcll(seg);
            }
        }
        if (g < G)
            for (int i2 = nk + li; i2 < Dp; i2 += C4) { off[i2] = (int)kOut; dep[i2] = 0.f; }      // padding: loads nothing
    }
    // batches of the longest list in the wave (the shorter ones run into their padding)
    int nb = g < G ? (nk + kRayBins - 1) / kRayBins : 0;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int o_ = __shfl_xor(nb, m); nb = o_ > nb ? o_ : nb; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    RAY_STAMP(1);

    const unsigned lane_off = g < G ? (unsigned)li * 16u : 0x40000000u;   // idle lanes: beyond the buffer whatever the bin
    // four bins = four row loads per sub-batch, two sub-batches in flight (software pipeline: the loads of the next one are
    // issued before the current one is consumed; the scheduling barriers keep the compiler from hoisting further loads and
    // paying for them in registers -- 57 VGPRs with one sub-batch, 136 with five, and occupancy matters more here)
    // The walk is VALU-bound (every wave-instruction occupies its SIMD for 4 cycles and a wave spends ~30 of them per bin in a
    // naive form), so: packed fp32 FMAs, no per-bin selects (a dropped bin's offset lies beyond the buffer, adding the lane's
    // 16 * li keeps it there), and a sub-batch in which no lane group of the wave has a kept bin skips its arithmetic.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 cxl = {cx.x, cx.y}, cxh = {cx.z, cx.w};
    f32x2 accl = {0.f, 0.f}, acch = {0.f, 0.f};
    auto load4 = [&](int dbase, mmt_u32x4 (&v)[4], float (&dv)[4], bool &any) __attribute__((always_inline)) {
        const int4 o4 = *reinterpret_cast<const int4 *>(off + dbase);        // dbase is a multiple of 4: one 16-byte LDS read each
        const float4 d4 = *reinterpret_cast<const float4 *>(dep + dbase);
        const unsigned o[4] = {(unsigned)o4.x, (unsigned)o4.y, (unsigned)o4.z, (unsigned)o4.w};
        dv[0] = d4.x; dv[1] = d4.y; dv[2] = d4.z; dv[3] = d4.w;
        any = __any((o[0] & o[1] & o[2] & o[3]) != kOut);                   // kOut is a single bit: all four dropped <=> the AND keeps it
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o[u] + lane_off, 0, 0);
    };
    auto comp4 = [&](int u0, const mmt_u32x4 (&v)[4], const float (&dv)[4], bool any) __attribute__((always_inline)) {
        float mine = 0.f;
        if (any) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x2 gl = {__uint_as_float(v[u].x), __uint_as_float(v[u].y)}, gh = {__uint_as_float(v[u].z), __uint_as_float(v[u].w)};
                const f32x2 dd = {dv[u], dv[u]};
                accl = __builtin_elementwise_fma(gl, dd, accl);
                acch = __builtin_elementwise_fma(gh, dd, acch);
                const f32x2 t = __builtin_elementwise_fma(gh, cxh, gl * cxl);
                const float qs = quad_sum(t.x + t.y);
                if ((li & 3) == u) mine = qs;
            }
        }
        if (g < G) part[(u0 + (li & 3)) * Q + (li >> 2)] = mine;      // partial of bin u0 + (li & 3), quad li >> 2
        asm volatile("" : "+v"(accl), "+v"(acch));   // the sums are due HERE: without this the compiler sinks them to the end of the
                                                     // batch and keeps 16 rows alive (150 VGPRs instead of 70)
    };
    mmt_u32x4 va[4], vb[4];
    float da[4], db[4];
    bool sa, sb_;
    // The 4 waves of a workgroup (12 pixels of one column: the same BEV rows) start a quarter of the ray apart and wrap
    // around: 4x as many DIFFERENT rows are in flight per workgroup, and a row that one wave had to wait for from HBM is in
    // L2 when the next wave arrives there (cold grad_out inside the training step: 41 -> RAYROT us; warm: 28.6).
#ifdef RAY_NO_ROTATE
    const int b0 = 0;
#else
    const int b0 = (wave * nb) / (kBlock / 64);
#endif
    int d0 = b0 * kRayBins;
    __builtin_amdgcn_sched_barrier(0);
    load4(d0, va, da, sa);
    __builtin_amdgcn_sched_barrier(0);             // issue order = use order, or the first wait of the loop has to drain everything
    load4(d0 + 4, vb, db, sb_);
#pragma unroll 1
    for (int it = 0; it < nb; ++it) {               // bins past D load nothing and count for nothing
        int dn_ = d0 + kRayBins;                   // the batch after this one; after the last one: the padding (nothing)
        if (dn_ >= nb * kRayBins) dn_ = 0;
        if (it == nb - 1) dn_ = nb * kRayBins;
        __builtin_amdgcn_sched_barrier(0);
        comp4(0, va, da, sa);
        __builtin_amdgcn_sched_barrier(0);
        load4(d0 + 8, va, da, sa);
        __builtin_amdgcn_sched_barrier(0);
        comp4(4, vb, db, sb_);
        __builtin_amdgcn_sched_barrier(0);
        load4(d0 + 12, vb, db, sb_);
        __builtin_amdgcn_sched_barrier(0);
        comp4(8, va, da, sa);
        __builtin_amdgcn_sched_barrier(0);
        load4(dn_, va, da, sa);                             // the next batch is on its way while this one is reduced
        __builtin_amdgcn_sched_barrier(0);
        comp4(12, vb, db, sb_);
        __builtin_amdgcn_sched_barrier(0);
        load4(dn_ + 4, vb, db, sb_);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int bin = li; bin < kRayBins; bin += C4) {
            const int e = d0 + bin;                         // list entry
            if (g < G && e < nk) {
                float s = 0.f;
                for (int k = 0; k < Q; ++k) s += part[bin * Q + k];
                const int64_t t = t0 + (int64_t)kbin[e] * dstep;
                if constexpr (sizeof(FT) == 2) reinterpret_cast<bf16_t *>(a.grad_depth)[t] = (bf16_t)(pack_bf16x2(s, 0.f) & 0xFFFFu);
                else reinterpret_cast<float *>(a.grad_depth)[t] = s;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        d0 = dn_;
    }
    RAY_STAMP(2);
#ifndef LSS_STAMPS
    if (act) *reinterpret_cast<float4 *>(a.grad_context + pix * C + li * 4) = make_float4(accl.x, accl.y, acch.x, acch.y);
#endif
}

// forward ray walk: depth slabs per ray (LDS within 64 KB; whole rays unless that leaves the chip short of workgroups) and
// the depth bins of each of the 16 lane groups
size_t ray_fwd_lds(const RayArgs &r, int dspan) {
    if (r.reg) return ray_fwd_reg_lds(r.C, dspan);
    if (r.blk) return ray_fwd_blk_lds(r.fH, r.C, dspan);
    return (size_t)r.fH * (r.C + 4) * 4 + (size_t)r.fH * (dspan | 1) * 8 + (size_t)dspan * 8;
}

bool pick_ray_forward(RayArgs *r) {
    int dsplit = 1;
    static const char *env = getenv("MMT_RAY_DSPLIT");     // experiments only
    if (env && atoi(env) > 0) dsplit = atoi(env);
    else while ((int64_t)r->BN * r->fW * dsplit < 1024 && (r->D + dsplit) / (dsplit + 1) >= 16) ++dsplit;
    if (dsplit > r->D) dsplit = r->D;
    r->compact = (r->D >= 160 || r->fH > 32) ? 1 : 0;              // long rays / tall columns: most (bin, row) pairs lie outside the grid
    static const char *no_reg = getenv("MMT_RAY_NO_REG");          // experiments only
    r->reg = (!r->compact && r->fH <= 16 && r->C <= 80 && !(no_reg && atoi(no_reg) > 0)) ? 1 : 0;
    static const char *no_blk = getenv("MMT_RAY_NO_BLK");          // experiments only
    r->blk = (!r->reg && r->combine != nullptr && !(no_blk && atoi(no_blk) > 0)) ? 1 : 0;
    for (;; ++dsplit) {
        const int dspan = (r->D + dsplit - 1) / dsplit;
        const size_t lds = ray_fwd_lds(*r, dspan);
        if (lds <= 64 * 1024) {
            r->dspan = dspan;
            r->dsplit = (r->D + dspan - 1) / dspan;
            r->kd = (dspan + kBlock / 16 - 1) / (kBlock / 16);
            return true;
        }
        if (dspan == 1) return false;
    }
}

// tile shape for a feature map: whole image columns, about 32 pixels x 16 depth bins
void pick_tile(int fH, int fW, int D, int BN, TileArgs *a) {
    int tpw = 32 / fH;
    if (tpw < 1) tpw = 1;
    if (tpw > fW) tpw = fW;
    a->TPW = tpw;
    a->TP = tpw * fH;
    int dt = kPts / a->TP;
    if (dt > D) dt = D;
    a->Dt = dt;
    a->wtiles = (fW + tpw - 1) / tpw;
    a->dtiles = (D + dt - 1) / dt;
    // depth groups: enough (camera, depth-group) units to spread over the 8 XCDs evenly
    int ngrp = 1;
    while (BN * ngrp < 64 && ngrp < a->dtiles) ++ngrp;
    a->dt_per_grp = (a->dtiles + ngrp - 1) / ngrp;
    a->ngrp_per_cam = (a->dtiles + a->dt_per_grp - 1) / a->dt_per_grp;
    a->NG = BN * a->ngrp_per_cam;
}

// which kernel family the process's last forward ([0]) / backward ([1]) call launched (mmt_lss_last_kernel_family); process-wide
// on purpose: an autograd backward runs on the engine's thread, the caller asks from its own
volatile int g_last_family[2] = {0, 0};

// zero-fill of the BEV map in front of the forward (MMT_LSS_ZERO_OUTPUT): `sc1` stores leave no line behind in the XCD L2s,
// so the memory-side atomics that follow do not wait for freshly written lines to be evicted (a torch.zeros right before
// the launch cost the forward 2.2 us, tools/kbench_fused.py `after_zero_fill`)
__global__ __launch_bounds__(kBlock) void lss_zero_fill(float4 *p, int64_t n4, ExclCall xc) {
    const int extra = xc.cache != nullptr ? 1 : 0;           // the grid's FIRST workgroup commits the exclusive-cell cache's mail and fills nothing
    if (extra && blockIdx.x == 0) { excl_commit(xc); return; }
    const int64_t stride = (int64_t)(gridDim.x - extra) * kBlock;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7FFFFFFF, 0x00020000);
    const mmt_u32x4 z = {0u, 0u, 0u, 0u};
    for (int64_t i = (int64_t)(blockIdx.x - extra) * kBlock + threadIdx.x; i < n4; i += stride)
        __builtin_amdgcn_raw_buffer_store_b128(z, rsrc, (unsigned)(i * 16), 0, 16);      // aux 16 = sc1
}

static int fill_blocks() {
    static const char *env = getenv("MMT_FILL_BLOCKS");        // experiments only
    return (env && atoi(env) > 0) ? atoi(env) : 2048;
}

template <typename FT>
int forward_impl(const char *what, int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz, const int32_t *geom,
                 const mmt::CamGeom *cam, const FT *depth, const FT *context, float *out, int32_t *pos_memo, int flags, hipStream_t st) {
    if (B <= 0 || N <= 0 || D <= 0 || fH <= 0 || fW <= 0 || C <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: non-positive size", what);
    constexpr int VEC = Elem<FT>::VEC;
    if (C % VEC != 0 || C % 4 != 0 || C > 256 || (((uintptr_t)context & 15) != 0))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: needs C %% %d == 0, C <= 256 and a 16-byte aligned context", what, VEC);
    if (fH > kPts) return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: fH=%d exceeds the %d-point tile; use mmt_lift_splat_forward", what, fH, kPts);
    const int64_t P = (int64_t)N * D * fH * fW, BP = (int64_t)B * P;
    if (BP >= (1ll << 31) || (int64_t)B * ny * nx >= (1ll << 31) || (int64_t)B * N * fH * fW * C >= (1ll << 31))
        return mmt::fail(MMT_ERR_TOO_LARGE, "%s: index range exceeds int32", what);
    if (flags & ~(MMT_VP_WRITE_DROPPED | MMT_LSS_PIXEL_MAJOR | MMT_LSS_TILE_KERNELS | MMT_LSS_ZERO_OUTPUT))
        return mmt::fail(MMT_ERR_BAD_FLAG, "%s: unknown flag bits 0x%x", what, flags);
    if (cam && (flags & MMT_LSS_TILE_KERNELS))
        return mmt::fail(MMT_ERR_BAD_FLAG, "%s: the camera form has no frustum-tile kernels (MMT_LSS_TILE_KERNELS)", what);
    const int64_t out_elems = (int64_t)B * ny * nx * C;
    if ((flags & MMT_LSS_ZERO_OUTPUT) && (((uintptr_t)out & 15) != 0 || out_elems * 4 >= (1ll << 31)))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: MMT_LSS_ZERO_OUTPUT needs a 16-byte aligned map below 2 GiB", what);
    mmt::TimedSeq seq;   // armed by mmt_arm_kernel_timing (bench only)
    ExclCall xc = {};
    auto zero_fill = [&]() {
        if (flags & MMT_LSS_ZERO_OUTPUT)
            seq.launch(false, lss_zero_fill, dim3((unsigned)mmt::stream_grid(out_elems / 4, kBlock, fill_blocks()) + (xc.cache ? 1u : 0u)), dim3(kBlock), 0, st,
                       reinterpret_cast<float4 *>(out), out_elems / 4, xc);
    };
    if (!(flags & MMT_LSS_TILE_KERNELS) && (C == 64 || C == 80 || C == 128)) {   // other widths: the tile kernels
        RayArgs r = {};
        r.BN = B * N; r.N = N; r.D = D; r.fH = fH; r.fW = fW; r.C = C; r.nx = nx; r.ny = ny; r.nz = nz;
        r.pm = (flags & MMT_LSS_PIXEL_MAJOR) ? 1 : 0;
        r.write_dropped = (flags & MMT_VP_WRITE_DROPPED) ? 1 : 0;
        r.geom = geom; r.depth = depth; r.context = context; r.out = out; r.pos_memo = pos_memo;
        if (cam) {
            r.combine = cam->combine; r.fu = cam->fu; r.fv = cam->fv; r.fd = cam->fd; r.q = cam->q;
            r.summary = reinterpret_cast<int2 *>(cam->summary); r.summary_cached = cam->summary_cached;
        }
        if (pick_ray_forward(&r)) {
            const size_t lds = ray_fwd_lds(r, r.dspan);
            const int64_t grid = 8ll * ((r.BN + 7) / 8) * fW * r.dsplit;
            if (grid >= (1ll << 31)) return mmt::fail(MMT_ERR_TOO_LARGE, "%s: grid too large", what);
            // the exclusive-cell cache: register walk + in-library zero-fill only (the fill kernel commits the cache's mail);
            // other shapes ignore it.  Run ids and tagged cell ids must fit 31 / 30 bits.
            if (cam && cam->excl && r.reg && (flags & MMT_LSS_ZERO_OUTPUT) && N <= kExclMaxN && (int64_t)B * ny * nx < kExclFlag &&
                (int64_t)N * fW * r.dsplit * (kRegBlock / 16) * r.dspan * 16 < (1ll << 31) - 2) {
                ExclShape xs = {};
                xs.N = N; xs.cells = ny * nx;
                const int64_t fit = (cam->excl_bytes / 4 - kExclHeaderWords) / excl_slot_words(N, xs.cells);
                if (fit < 1) return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: the exclusive-cell cache holds no slot (%lld bytes; mmt_lss_exclusive_cache_bytes)", what,
                                              (long long)cam->excl_bytes);
                xs.slots = fit > kExclMaxSlots ? kExclMaxSlots : (int)fit;
                // the launch shape the learnt states depend on (run decomposition and cell numbering)
                int words[17] = {N, D, fH, fW, nx, ny, nz, r.dsplit, r.dspan, kRegBlock + (3 << 16) /* table layout v3: run decomposition with shares ending at run ends; 2-way sets */, xs.slots};
                memcpy(words + 11, cam->q.lo, 12);
                memcpy(words + 14, cam->q.vs, 12);
                uint64_t h = 0xCBF29CE484222325ull;
                for (int w : words) { h ^= (uint32_t)w; h *= 0x100000001B3ull; h ^= h >> 29; }
                xs.sig_lo = (unsigned)h | 1u; xs.sig_hi = (unsigned)(h >> 32);
                r.excl = cam->excl; r.xs = xs;
                xc.cache = cam->excl; xc.fu = cam->fu; xc.fv = cam->fv; xc.fd = cam->fd; xc.fW = fW; xc.fH = fH; xc.D = D; xc.xs = xs;
            }
            zero_fill();
            const dim3 g((unsigned)grid), blk(kBlock);
            if (r.reg) {
                const dim3 rb(kRegBlock);
                if (cam) {
                    if (C == 80) seq.launch(true, lss_ray_fwd_reg<FT, 5, true>, g, rb, lds, st, r);
                    else seq.launch(true, lss_ray_fwd_reg<FT, 4, true>, g, rb, lds, st, r);
                } else {
                    if (C == 80) seq.launch(true, lss_ray_fwd_reg<FT, 5, false>, g, rb, lds, st, r);
                    else seq.launch(true, lss_ray_fwd_reg<FT, 4, false>, g, rb, lds, st, r);
                }
            } else if (r.blk) {
                if (C == 80) seq.launch(true, lss_ray_fwd_blk<FT, 5>, g, blk, lds, st, r);
                else if (C == 64) seq.launch(true, lss_ray_fwd_blk<FT, 4>, g, blk, lds, st, r);
                else seq.launch(true, lss_ray_fwd_blk<FT, 8>, g, blk, lds, st, r);
            } else if (cam) {
                if (C == 80) seq.launch(true, lss_ray_fwd<FT, 5, true>, g, blk, lds, st, r);
                else if (C == 64) seq.launch(true, lss_ray_fwd<FT, 4, true>, g, blk, lds, st, r);
                else seq.launch(true, lss_ray_fwd<FT, 8, true>, g, blk, lds, st, r);
            } else {
                if (C == 80) seq.launch(true, lss_ray_fwd<FT, 5, false>, g, blk, lds, st, r);
                else if (C == 64) seq.launch(true, lss_ray_fwd<FT, 4, false>, g, blk, lds, st, r);
                else seq.launch(true, lss_ray_fwd<FT, 8, false>, g, blk, lds, st, r);
            }
            g_last_family[0] = MMT_LSS_FAMILY_RAY | (cam ? MMT_LSS_FAMILY_CAMERA : 0) | (r.reg ? MMT_LSS_FAMILY_REGISTER : 0) |
                               (r.excl ? MMT_LSS_FAMILY_EXCLUSIVE : 0) | (r.blk ? MMT_LSS_FAMILY_BLOCK : 0);
            return mmt::check_launch(what);
        }
    }
    if (cam)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: the camera form needs C in {64, 80, 128} and a column of fH=%d rows within 64 KB of LDS; "
                         "use the geom form (mmt_frustum_geometry + mmt_lss_splat_forward)", what, fH);
    TileArgs a;
    a.N = N; a.D = D; a.fH = fH; a.fW = fW; a.C = C; a.nx = nx; a.ny = ny; a.nz = nz;
    pick_tile(fH, fW, D, B * N, &a);
    a.geom = geom; a.depth = depth; a.context = context; a.out = out; a.pos_memo = pos_memo;
    a.write_dropped = (flags & MMT_VP_WRITE_DROPPED) ? 1 : 0;
    a.pm = (flags & MMT_LSS_PIXEL_MAJOR) ? 1 : 0;
    a.grad_out = nullptr; a.sb = a.sy = a.sx = 0; a.grad_depth = nullptr; a.grad_context = nullptr;
    const size_t lds = (size_t)a.TP * C * 4;
    if (lds > 96 * 1024) return mmt::fail(MMT_ERR_TOO_LARGE, "%s: context tile of %d pixels x %d channels exceeds LDS", what, a.TP, C);
    const int64_t grid = 8ll * ((a.NG + 7) / 8) * a.dt_per_grp * a.wtiles;
    if (grid >= (1ll << 31)) return mmt::fail(MMT_ERR_TOO_LARGE, "%s: grid too large", what);
    zero_fill();
    if (C == 80) seq.launch(true, lss_splat_fwd_tile<FT, 20>, dim3((unsigned)grid), dim3(kBlock), lds, st, a);
    else if (C == 64) seq.launch(true, lss_splat_fwd_tile<FT, 16>, dim3((unsigned)grid), dim3(kBlock), lds, st, a);
    else seq.launch(true, lss_splat_fwd_tile<FT, 0>, dim3((unsigned)grid), dim3(kBlock), lds, st, a);
    g_last_family[0] = MMT_LSS_FAMILY_TILE;
    return mmt::check_launch(what);
}

template <typename FT>
int backward_impl(const char *what, int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz, const int32_t *geom,
                  const mmt::CamGeom *cam, const FT *depth, const FT *context, const float *grad_out, int64_t sb, int64_t sc, int64_t sy,
                  int64_t sx, FT *grad_depth, float *grad_context, unsigned long long *stats, int flags, hipStream_t st) {
    if (flags & ~(MMT_LSS_PIXEL_MAJOR | MMT_LSS_TILE_KERNELS | MMT_LSS_COLUMN_BACKWARD))
        return mmt::fail(MMT_ERR_BAD_FLAG, "%s: unknown flag bits 0x%x", what, flags);
    if (B <= 0 || N <= 0 || D <= 0 || fH <= 0 || fW <= 0 || C <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: non-positive size", what);
    constexpr int VEC = Elem<FT>::VEC;
    if (C % VEC != 0 || C % 16 != 0 || C > 256)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: needs C %% 16 == 0 and C <= 256 (C=%d)", what, C);
    if (fH > kPts) return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: fH=%d exceeds the %d-point tile; use mmt_lift_splat_backward", what, fH, kPts);
    const int64_t span = (B - 1) * sb + (ny - 1) * sy + (nx - 1) * sx + C;
    if (sc != 1 || sb % 4 || sy % 4 || sx % 4 || sb < 0 || sy < 0 || sx < 0 || span >= (1ll << 31) ||
        (((uintptr_t)grad_out | (uintptr_t)context | (uintptr_t)grad_context) & 15) != 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: grad_out must be channels-last (stride_c == 1), 16-byte aligned, < 2^31 elements", what);
    if ((int64_t)B * N * D * fH * fW >= (1ll << 31) || (int64_t)B * N * fH * fW * C >= (1ll << 31))
        return mmt::fail(MMT_ERR_TOO_LARGE, "%s: index range exceeds int32", what);
    if (cam && (flags & MMT_LSS_TILE_KERNELS))
        return mmt::fail(MMT_ERR_BAD_FLAG, "%s: the camera form has no frustum-tile kernels (MMT_LSS_TILE_KERNELS)", what);
    {
        const bool want_col = (flags & MMT_LSS_COLUMN_BACKWARD) != 0;       // the matrix-core column kernel (lift_splat_col.hip)
        if (want_col && !(flags & MMT_LSS_TILE_KERNELS) && mmt::lss_col_backward_fits(D, fH, fW, C, span, (int64_t)B * N * fW * ((fH + 15) / 16))) {
            const int pm = (flags & MMT_LSS_PIXEL_MAJOR) ? 1 : 0;
            g_last_family[1] = MMT_LSS_FAMILY_COLUMN | (cam ? MMT_LSS_FAMILY_CAMERA : 0);
            if constexpr (sizeof(FT) == 2)
                return mmt::lss_col_backward_bf16(what, B, N, D, fH, fW, C, nx, ny, nz, geom, cam, depth, context, grad_out, sb, sy, sx, span, grad_depth, grad_context, stats, pm, st);
            else
                return mmt::lss_col_backward_f32(what, B, N, D, fH, fW, C, nx, ny, nz, geom, cam, depth, context, grad_out, sb, sy, sx, span, grad_depth, grad_context, stats, pm, st);
        }
    }
    {
        const int C4 = C / 4, NGR = (kBlock / 64) * (64 / C4);
        const size_t lds = (size_t)NGR * (3 * (((D + kRayBins - 1) & ~(kRayBins - 1)) + 8) + kRayBins * (C4 / 4)) * 4;
        if (!(flags & MMT_LSS_TILE_KERNELS) && lds <= 64 * 1024 && span * 4 < (1ll << 30)) {
            RayArgs r = {};
            r.BN = B * N; r.N = N; r.D = D; r.fH = fH; r.fW = fW; r.C = C; r.nx = nx; r.ny = ny; r.nz = nz;
            r.pm = (flags & MMT_LSS_PIXEL_MAJOR) ? 1 : 0;
            r.geom = geom; r.depth = depth; r.context = context;
            if (cam) {
                r.combine = cam->combine; r.fu = cam->fu; r.fv = cam->fv; r.fd = cam->fd; r.q = cam->q;
                r.summary = reinterpret_cast<int2 *>(cam->summary);
            }
            r.grad_out = grad_out; r.sb = sb; r.sy = sy; r.sx = sx; r.span_bytes = (int)(span * 4);
            r.grad_depth = grad_depth; r.grad_context = grad_context;
            r.wpc = (fH * fW + NGR - 1) / NGR;
            {
                int gcd = r.BN, e = 8;
                while (e) { const int t = gcd % e; gcd = e; e = t; }
                r.split = 8 / gcd;                              // BN * split is a multiple of 8
                if (r.split > r.wpc) r.split = 1;
                r.wps = (r.wpc + r.split - 1) / r.split;
            }
            const int64_t grid = 8ll * ((r.BN * r.split + 7) / 8) * r.wps;
            if (grid >= (1ll << 31)) return mmt::fail(MMT_ERR_TOO_LARGE, "%s: grid too large", what);
            const size_t lds_req = lds;
            const dim3 g((unsigned)grid), blk(kBlock);
            mmt::TimedSeq seq;
            if (cam) {
                if (C == 80) seq.launch(true, lss_ray_bwd<FT, 20, true>, g, blk, lds_req, st, r);
                else if (C == 64) seq.launch(true, lss_ray_bwd<FT, 16, true>, g, blk, lds_req, st, r);
                else seq.launch(true, lss_ray_bwd<FT, 0, true>, g, blk, lds_req, st, r);
            } else {
                if (C == 80) seq.launch(true, lss_ray_bwd<FT, 20, false>, g, blk, lds_req, st, r);
                else if (C == 64) seq.launch(true, lss_ray_bwd<FT, 16, false>, g, blk, lds_req, st, r);
                else seq.launch(true, lss_ray_bwd<FT, 0, false>, g, blk, lds_req, st, r);
            }
            g_last_family[1] = MMT_LSS_FAMILY_RAY | (cam ? MMT_LSS_FAMILY_CAMERA : 0);
            return mmt::check_launch(what);
        }
    }
    if (cam)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: the camera form needs a ray of D=%d bins within 64 KB of LDS per workgroup; use the geom form", what, D);
    // frustum-tile kernel: partial sums per depth tile meet in grad_context through fp32 atomics
    if (const hipError_t e = hipMemsetAsync(grad_context, 0, (size_t)B * N * fH * fW * C * 4, st); e != hipSuccess)
        return mmt::fail((int)e, "%s: zero-fill of grad_context: %s", what, hipGetErrorString(e));
    TileArgs a;
    a.N = N; a.D = D; a.fH = fH; a.fW = fW; a.C = C; a.nx = nx; a.ny = ny; a.nz = nz;
    pick_tile(fH, fW, D, B * N, &a);
    a.geom = geom; a.depth = depth; a.context = context; a.out = nullptr; a.pos_memo = nullptr; a.write_dropped = 0;
    a.pm = (flags & MMT_LSS_PIXEL_MAJOR) ? 1 : 0;
    a.grad_out = grad_out; a.sb = sb; a.sy = sy; a.sx = sx; a.grad_depth = grad_depth; a.grad_context = grad_context;
    const size_t lds = ((size_t)a.TP + kGradRows) * C * 4;
    if (lds > 96 * 1024) return mmt::fail(MMT_ERR_TOO_LARGE, "%s: context tile of %d pixels x %d channels exceeds LDS", what, a.TP, C);
    const int64_t grid = 8ll * ((a.NG + 7) / 8) * a.dt_per_grp * a.wtiles;
    if (grid >= (1ll << 31)) return mmt::fail(MMT_ERR_TOO_LARGE, "%s: grid too large", what);
    mmt::TimedSeq seq;   // armed by mmt_arm_kernel_timing (bench only)
    if (C == 80) seq.launch(true, lss_splat_bwd_tile<FT, 20>, dim3((unsigned)grid), dim3(kBlock), lds, st, a);
    else if (C == 64) seq.launch(true, lss_splat_bwd_tile<FT, 16>, dim3((unsigned)grid), dim3(kBlock), lds, st, a);
    else seq.launch(true, lss_splat_bwd_tile<FT, 0>, dim3((unsigned)grid), dim3(kBlock), lds, st, a);
    g_last_family[1] = MMT_LSS_FAMILY_TILE;
    return mmt::check_launch(what);
}

}  // namespace

extern "C" int mmt_lss_splat_backward(int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz,
                                      const int32_t *geom, const float *depth, const float *context,
                                      const float *grad_out, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                                      float *grad_depth, float *grad_context, int flags, void *stream) {
    MMT_REQUIRE_PTR(geom);
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(grad_out);
    MMT_REQUIRE_PTR(grad_depth);
    MMT_REQUIRE_PTR(grad_context);
    return backward_impl<float>("lss_splat_backward", B, N, D, fH, fW, C, nx, ny, nz, geom, nullptr, depth, context, grad_out, sb, sc, sy, sx,
                                grad_depth, grad_context, nullptr, flags, (hipStream_t)stream);
}

extern "C" int mmt_lss_splat_backward_bf16(int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz,
                                           const int32_t *geom, const uint16_t *depth, const uint16_t *context,
                                           const float *grad_out, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                                           uint16_t *grad_depth, float *grad_context, int flags, void *stream) {
    MMT_REQUIRE_PTR(geom);
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(grad_out);
    MMT_REQUIRE_PTR(grad_depth);
    MMT_REQUIRE_PTR(grad_context);
    return backward_impl<bf16_t>("lss_splat_backward_bf16", B, N, D, fH, fW, C, nx, ny, nz, geom, nullptr, depth, context, grad_out, sb, sc,
                                 sy, sx, grad_depth, grad_context, nullptr, flags, (hipStream_t)stream);
}

extern "C" int mmt_lss_splat_forward(int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz,
                                     const int32_t *geom, const float *depth, const float *context, float *out,
                                     int32_t *pos_memo, int flags, void *stream) {
    MMT_REQUIRE_PTR(geom);
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(out);
    return forward_impl<float>("lss_splat_forward", B, N, D, fH, fW, C, nx, ny, nz, geom, nullptr, depth, context, out, pos_memo, flags,
                               (hipStream_t)stream);
}

extern "C" int mmt_lss_splat_forward_bf16(int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz,
                                          const int32_t *geom, const uint16_t *depth, const uint16_t *context, float *out,
                                          int32_t *pos_memo, int flags, void *stream) {
    MMT_REQUIRE_PTR(geom);
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(out);
    return forward_impl<bf16_t>("lss_splat_forward_bf16", B, N, D, fH, fW, C, nx, ny, nz, geom, nullptr, depth, context, out, pos_memo,
                                flags, (hipStream_t)stream);
}

// ---- camera form (ABI 6): the voxel index of a frustum point is computed inside the kernels (mmt_camera.h)
namespace {
int make_cam(const char *what, const float *combine, const float *fu, const float *fv, const float *fd, const float *vc, const float *vs,
             int nx, int ny, int nz, int32_t *summary, int summary_cached, mmt::CamGeom *cam) {
    if (!combine || !fu || !fv || !fd || !vc || !vs) return mmt::fail(MMT_ERR_NULL_POINTER, "%s: a camera-geometry pointer is NULL", what);
    if (summary_cached && !summary) return mmt::fail(MMT_ERR_NULL_POINTER, "%s: MMT_LSS_SUMMARY_CACHED without a column summary", what);
    if (summary && (nx > 32767 || ny > 32767 || ((uintptr_t)summary & 7) != 0))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: the column summary needs an 8-byte aligned buffer and a grid below 32768 x 32768", what);
    cam->combine = combine; cam->fu = fu; cam->fv = fv; cam->fd = fd;
    cam->summary = summary; cam->summary_cached = summary_cached;
    cam->excl = nullptr; cam->excl_bytes = 0;
    mmt::make_cam_grid(vc, vs, &cam->q);
    if (nx > 0 && ny > 0 && nz > 0) mmt::make_cam_range(&cam->q, nx, ny, nz);      // (non-positive sizes are refused further down)
    return MMT_OK;
}
int attach_excl(const char *what, int32_t *cache, int64_t bytes, mmt::CamGeom *cam) {
    if (!cache) return MMT_OK;
    if (((uintptr_t)cache & 15) != 0 || bytes < (kExclHeaderWords + 16) * 4)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: the exclusive-cell cache needs a 16-byte aligned buffer of mmt_lss_exclusive_cache_bytes() bytes", what);
    cam->excl = cache; cam->excl_bytes = bytes;
    return MMT_OK;
}
}  // namespace

// bytes of an exclusive-cell cache for `slots` calibrations of N cameras on an nx x ny map (int32 words, zero-initialised once)
extern "C" int64_t mmt_lss_exclusive_cache_bytes(int N, int nx, int ny, int slots) {
    if (N <= 0 || nx <= 0 || ny <= 0 || slots <= 0 || (int64_t)nx * ny >= kExclFlag) return 0;
    if (slots > kExclMaxSlots) slots = kExclMaxSlots;
    return 4 * ((int64_t)kExclHeaderWords + (int64_t)slots * excl_slot_words(N, nx * ny));
}

extern "C" int mmt_lss_splat_forward_cam(int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz, const float *combine,
        const float *frustum_u, const float *frustum_v, const float *frustum_d, const float *voxel_coord_host,
        const float *voxel_size_host, const float *depth, const float *context, float *out, int32_t *pos_memo,
        int32_t *column_summary, int32_t *exclusive_cache, int64_t exclusive_cache_bytes, int flags, void *stream) {
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(out);
    mmt::CamGeom cam;
    if (const int rc = make_cam("lss_splat_forward_cam", combine, frustum_u, frustum_v, frustum_d, voxel_coord_host, voxel_size_host, nx, ny, nz,
                                column_summary, (flags & MMT_LSS_SUMMARY_CACHED) ? 1 : 0, &cam)) return rc;
    if (const int rc = attach_excl("lss_splat_forward_cam", exclusive_cache, exclusive_cache_bytes, &cam)) return rc;
    return forward_impl<float>("lss_splat_forward_cam", B, N, D, fH, fW, C, nx, ny, nz, nullptr, &cam, depth, context, out, pos_memo,
                                flags & ~MMT_LSS_SUMMARY_CACHED, (hipStream_t)stream);
}

extern "C" int mmt_lss_splat_forward_cam_bf16(int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz, const float *combine,
        const float *frustum_u, const float *frustum_v, const float *frustum_d, const float *voxel_coord_host,
        const float *voxel_size_host, const uint16_t *depth, const uint16_t *context, float *out, int32_t *pos_memo,
        int32_t *column_summary, int32_t *exclusive_cache, int64_t exclusive_cache_bytes, int flags, void *stream) {
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(out);
    mmt::CamGeom cam;
    if (const int rc = make_cam("lss_splat_forward_cam_bf16", combine, frustum_u, frustum_v, frustum_d, voxel_coord_host, voxel_size_host, nx, ny, nz,
                                column_summary, (flags & MMT_LSS_SUMMARY_CACHED) ? 1 : 0, &cam)) return rc;
    if (const int rc = attach_excl("lss_splat_forward_cam_bf16", exclusive_cache, exclusive_cache_bytes, &cam)) return rc;
    return forward_impl<bf16_t>("lss_splat_forward_cam_bf16", B, N, D, fH, fW, C, nx, ny, nz, nullptr, &cam, depth, context, out, pos_memo,
                                flags & ~MMT_LSS_SUMMARY_CACHED, (hipStream_t)stream);
}

extern "C" int mmt_lss_splat_backward_cam(int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz, const float *combine,
        const float *frustum_u, const float *frustum_v, const float *frustum_d, const float *voxel_coord_host,
        const float *voxel_size_host, const float *depth, const float *context, const float *grad_out, int64_t sb, int64_t sc,
        int64_t sy, int64_t sx, float *grad_depth, float *grad_context, const int32_t *column_summary, uint64_t *column_stats,
        int flags, void *stream) {
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(grad_out);
    MMT_REQUIRE_PTR(grad_depth);
    MMT_REQUIRE_PTR(grad_context);
    mmt::CamGeom cam;
    if (const int rc = make_cam("lss_splat_backward_cam", combine, frustum_u, frustum_v, frustum_d, voxel_coord_host, voxel_size_host, nx, ny, nz,
                                const_cast<int32_t *>(column_summary), 0, &cam)) return rc;
    return backward_impl<float>("lss_splat_backward_cam", B, N, D, fH, fW, C, nx, ny, nz, nullptr, &cam, depth, context, grad_out, sb, sc, sy, sx,
                                 grad_depth, grad_context, reinterpret_cast<unsigned long long *>(column_stats), flags, (hipStream_t)stream);
}

extern "C" int mmt_lss_splat_backward_cam_bf16(int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz, const float *combine,
        const float *frustum_u, const float *frustum_v, const float *frustum_d, const float *voxel_coord_host,
        const float *voxel_size_host, const uint16_t *depth, const uint16_t *context, const float *grad_out, int64_t sb, int64_t sc,
        int64_t sy, int64_t sx, uint16_t *grad_depth, float *grad_context, const int32_t *column_summary, uint64_t *column_stats,
        int flags, void *stream) {
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(grad_out);
    MMT_REQUIRE_PTR(grad_depth);
    MMT_REQUIRE_PTR(grad_context);
    mmt::CamGeom cam;
    if (const int rc = make_cam("lss_splat_backward_cam_bf16", combine, frustum_u, frustum_v, frustum_d, voxel_coord_host, voxel_size_host, nx, ny, nz,
                                const_cast<int32_t *>(column_summary), 0, &cam)) return rc;
    return backward_impl<bf16_t>("lss_splat_backward_cam_bf16", B, N, D, fH, fW, C, nx, ny, nz, nullptr, &cam, depth, context, grad_out, sb, sc, sy, sx,
                                 grad_depth, grad_context, reinterpret_cast<unsigned long long *>(column_stats), flags, (hipStream_t)stream);
}

extern "C" int mmt_lss_last_kernel_family(int backward) { return g_last_family[backward ? 1 : 0]; }
void mmt::lss_note_forward_family(int family) { g_last_family[0] = family; }

// 1 when mmt_lss_splat_forward_cam* of this shape uses an exclusive-cell cache it is handed (with MMT_LSS_ZERO_OUTPUT): the
// kernel choice of forward_impl -- today the register walk.  A caller allocates the cache only then.
extern "C" int mmt_lss_exclusive_cache_used(int B, int N, int D, int fH, int fW, int C) {
    if (B <= 0 || N <= 0 || N > kExclMaxN || D <= 0 || fH <= 0 || fW <= 0 || !(C == 64 || C == 80 || C == 128) || fH > kPts) return 0;
    RayArgs r = {};
    r.BN = B * N; r.N = N; r.D = D; r.fH = fH; r.fW = fW; r.C = C;
    r.combine = reinterpret_cast<const float *>(16);
    if (!pick_ray_forward(&r)) return 0;
    return r.reg ? 1 : 0;
}

// 1 when both directions have a camera-form kernel for this shape (the gates of forward_impl / backward_impl)
extern "C" int mmt_lss_camera_form_supported(int B, int N, int D, int fH, int fW, int C) {
    if (B <= 0 || N <= 0 || D <= 0 || fH <= 0 || fW <= 0 || !(C == 64 || C == 80 || C == 128) || fH > kPts) return 0;
    RayArgs r = {};
    r.BN = B * N; r.N = N; r.D = D; r.fH = fH; r.fW = fW; r.C = C;
    r.combine = reinterpret_cast<const float *>(16);         // (camera form: the kernel choice of forward_impl)
    if (!pick_ray_forward(&r)) return 0;
    const int C4 = C / 4, NGR = (kBlock / 64) * (64 / C4);
    const size_t lds = (size_t)NGR * (3 * (((D + kRayBins - 1) & ~(kRayBins - 1)) + 8) + kRayBins * (C4 / 4)) * 4;
    return lds <= 64 * 1024 ? 1 : 0;
}
