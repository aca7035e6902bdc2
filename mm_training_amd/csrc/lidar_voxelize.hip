// LiDAR / radar half of the hot path for MI355X (gfx950, wave64):
// hard voxelization (deterministic, first-point order), per-voxel mean, pillar scatter.
//
// Replaces the third-party ops the reference reaches at models/bev_depth.py:181-183
// (mmcv-full 1.7.0 ops.Voxelization, mmdet3d 1.0.0rc4 HardSimpleVFE and
// PointPillarsScatter).  mmcv's deterministic GPU path compares every point with
// every earlier point (O(N^2)) and numbers voxels in a single thread.
//
// Round 5: the REGION-OWNER form.  Rounds 1-4 threaded every point onto its cell's chain with one scattered 64-bit
// atomicExch on a dense per-sample table in memory (160 k returning device-scope atomics at ~20 G/s = 8 us before anything
// else happened), decided "first of its cell" by chain walks, and left the voxel ids in the table with one scattered write
// per voxel: 28 us and 29 MB of traffic for 7 MB of points and outputs.  Here nothing is exchanged through memory atomics:
//   1. cells : per point (coalesced): cell id -> the REGION it falls into (a run of R = 1024..4096 consecutive cells; one
//              byte -- two when a grid has more than 254 regions) and the cell's position inside the region (16 bits);
//   2. own   : one workgroup per (sample, region) streams the sample's region bytes (40 KB for 40 k points: every load is
//              in flight at once), keeps the indices of ITS points in LDS, reads their cell positions (one gather) and
//              settles every cell of the region in LDS: rounds of "smallest index not yet taken" (an LDS atomicMin per
//              live point and round) give the cell's points in ascending order -- the first is the voxel's head, the
//              next max_points - 1 go to the head's list in memory, the rest are dropped -- order-independent, so
//              bit-reproducible.  It leaves: a count byte at the head's point (0 = not a head), and the region's slice
//              of the sample's cell directory: an occupancy bit per cell, the ordinal of every 64-cell block's first
//              occupied cell, and the head of every occupied cell in ordinal order -- all coalesced;
//              It also leaves, per tile of 256 points, how many of ITS heads lie in front of that tile (a histogram over the
//              heads' tiles + a prefix in LDS; row t, column r of a small table);
//   3. emit  : per 256-point tile: row t of that table summed over the regions = the heads before the tile, the last row =
//              the heads of the sample (clouds of more than 2 M points: every tile counts the sample's count bytes
//              itself), a ballot + popcount over the tile's own count bytes the rank inside the tile:
//              voxel id of a head = heads before it  ==> voxels numbered in order of their first point, exactly like
//              the sequential algorithm; heads past max_voxels are dropped.  A head reads its list (contiguous), the
//              tile then writes its voxels' rows -- a CONTIGUOUS block of the output -- the zero padding, coors,
//              num_points and the HardSimpleVFE mean (summed in slot order) as flat coalesced stores, and the voxel id
//              at the head's point index (the pillar scatter's last hop: cell -> bit -> ordinal -> head -> voxel id).
//              Rows past a sample's voxel count are marked empty (coors -1, num_points 0, mean 0) by the same kernel.
// The workgroups of a sample's regions sit on 8 / B XCDs (B | 8) or one (8 | B): what they all read is fetched into
// that many L2s, not eight.
#include <stdlib.h>
#include <atomic>

#include "mmt_common.h"

namespace {

constexpr int kTile = 256;            // points per tile / threads per workgroup (4 waves)
constexpr int kTileWaves = kTile / 64;
constexpr int kIdxBits = 24;          // point index inside its sample (host checks N < 2^24)
constexpr int kMaxBatchLds = 255;     // sample offsets cached in LDS up to this batch size
constexpr int kInf = 0x7fffffff;
constexpr int kEntries = 4096;        // vox_own: indices of the region's points waiting in LDS
constexpr int kStreamDepth = 10;      // 16-byte loads of region ids in flight per thread (10 x 256 x 16 points: a 40 k cloud in one round trip;
                                      // 20 measured the same at 2 x 80 k points)
constexpr int kSlowChunk = kTile * 8; // points per step when a batch has to be done again step by step (vox_own)
constexpr int kMinRegion = 1024, kMaxRegion = 4096;
constexpr int kCellTile = 1024;       // points per cells workgroup of the fused cells + own launch
constexpr int kSc1 = 16;              // cache policy of a buffer load / store: sc1 (write-through / not served from this CU's L1)
constexpr int kTableMagic = 0x32584f56;   // "VOX2": the table holds a cell directory of the region-owner form

// the cell directory inside the caller's table (int32 units from its start; the table is 8-byte aligned)
struct Directory {
    int R, shift, NR;                 // cells per region (a power of two), log2 R, regions per sample
    int64_t bits, pref, head_of, vidp, elems_without_vidp;
};

inline Directory directory_of(int B, int64_t cells) {
    Directory d;
    d.R = kMinRegion;
    // regions large enough for a byte to name them and for two workgroups per CU (one per sample and region) to cover the batch
    while (d.R < kMaxRegion && (mmt::ceil_div(cells, d.R) > 254 || (int64_t)B * mmt::ceil_div(cells, d.R) > 512)) d.R *= 2;
    d.shift = 0;
    while ((1 << d.shift) < d.R) ++d.shift;
    d.NR = (int)mmt::ceil_div(cells, d.R);
    const int64_t padded = (int64_t)B * d.NR * d.R;
    d.bits = 4;                                   // uint64 [B * NR * R / 64]
    d.pref = d.bits + padded / 32;                // uint16 [B * NR * R / 64]
    d.head_of = d.pref + padded / 128;            // int32  [B * NR * R]   (a region's slice: its occupied cells' heads, in cell order)
    d.vidp = d.head_of + padded;                  // int32  [total points] voxel id of a head point (-1: dropped by the voxel cap)
    d.elems_without_vidp = d.vidp;
    return d;
}

struct VoxArgs {
    int F, max_points, max_voxels, nf;
    int gx, gy, gz;
    float vs[3], rmin[3];
    const float *points;
    const int32_t *offsets;        // [B+1]
    int R, shift, NR, wide;        // directory_of; wide: region ids are 16 bits
    int64_t cells;
    // scratch, indexed by a point's FLAG POSITION fpos(b, i) = align16(offsets[b]) + 16 b + i (every sample's slice starts on a 16-byte boundary)
    void *reg;                     // uint8 / uint16 region of the point (all ones: outside the grid)
    unsigned short *cloc;          // the cell's position inside its region
    unsigned char *flag;           // head: min(points of its cell, max_points); anything else: 0
    unsigned short *tcum;          // [B][ntiles + 1][NR] heads of region r among the sample's points before tile t (NULL: emit counts the count bytes itself)
    int32_t *lists;                // [N * (max_points - 1)] at a head's point: the cell's 2nd, 3rd, ... point (index inside the sample)
    // table
    int32_t *header;
    unsigned long long *bits;
    unsigned short *pref;
    int32_t *head_of;
    int32_t *vidp;
    int ntiles;
    unsigned *cdone;               // fused cells + own launch: [ceil(N / kCellTile) + B] a cells workgroup's word = the launch's token when its bytes are out
    unsigned *epoch;               // the word the launch's token is made of (epoch + 1); vox_emit, the next launch, advances it
    unsigned long long *stamps;    // -DVOX_STAMPS builds only (tools/scratch/vox_stamps.py): phase time stamps of the first 1024 workgroups
    float *voxels;                 // may be NULL (only the mean is wanted)
    int32_t *coors;
    int32_t *num_points;
    int32_t *voxel_count;
    float *mean;                   // may be NULL
};

#ifdef VOX_STAMPS
#define VSTAMP(k) do { if (a.stamps && threadIdx.x == 0) a.stamps[(blockIdx.x & 1023) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define VSTAMP(k) do { } while (0)
#endif
__device__ __forceinline__ int cell_coord(float p, float rmin, float vs) {
    // mmcv: int c = floor((p - range_min) / voxel_size)  (fp32; saturating convert, NaN -> 0)
    return (int)floorf(__fdiv_rn(__fsub_rn(p, rmin), vs));
}

// sample of global point index g (offsets [B+1] ascending): B is small, a linear scan of the LDS copy
__device__ __forceinline__ int sample_of_point(const int *offs, int B, int g) {
    int b = 0;
    while (b + 1 < B && g >= offs[b + 1]) ++b;
    return b;
}

// inclusive prefix sum across the 64 lanes of a wave on the DPP path (no LDS round trips): row_shr 1, 2, 4, 8 inside the rows of
// 16 lanes, then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast:15) and lane 31 into rows 2 and 3 (row_bcast:31)
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    return v;
}

__device__ __forceinline__ int64_t flag_base(int beg, int b) { return (((int64_t)beg + 15) & ~(int64_t)15) + 16 * (int64_t)b; }

// (sample, tile) of a workgroup in the flattened tile space: sample b owns max(ceil(n_b / kTile), 1) consecutive
// workgroups (an empty sample keeps one: its rows still have to be marked empty).  Returns false past the last tile.
__device__ __forceinline__ bool locate_tile(const int32_t *offsets, int B, int wg, int *b_out, int *tile_out) {
    int first = 0;
    for (int b = 0; b < B; ++b) {
        const int n = offsets[b + 1] - offsets[b];
        int t = (n + kTile - 1) / kTile;
        t = t > 0 ? t : 1;
        if (wg < first + t) { *b_out = b; *tile_out = wg - first; return true; }
        first += t;
    }
    return false;
}

// The same from ONE round trip (B <= 63; every wave for itself, same answer): lane l takes sample l's offsets, a prefix sum over the
// samples' tile counts on the DPP path finds the sample whose range holds the workgroup.  Also hands back the sample's first point
// and point count.
__device__ __forceinline__ bool locate_tile_wave(const int32_t *offsets, int B, int wg, int *b_out, int *tile_out, int *beg_out, int *n_out) {
    const int lane = threadIdx.x & 63;
    const int o0 = offsets[lane < B ? lane : B], o1 = offsets[lane + 1 < B ? lane + 1 : B];
    const int n = lane < B ? o1 - o0 : 0;
    int t = (n + kTile - 1) / kTile;
    t = lane < B ? (t > 0 ? t : 1) : 0;
    const int incl = wave_incl_scan(t);
    const unsigned long long m = __ballot(lane < B && wg >= incl - t && wg < incl);
    if (m == 0ull) return false;
    const int src = __ffsll((long long)m) - 1;
    *b_out = src;
    *tile_out = wg - __builtin_amdgcn_readlane(incl - t, src);
    *beg_out = __builtin_amdgcn_readlane(o0, src);
    *n_out = __builtin_amdgcn_readlane(n, src);
    return true;
}

template <typename RT>
__global__ __launch_bounds__(kTile) void vox_cells(VoxArgs a, int B, int total) {
    __shared__ int offs[kMaxBatchLds + 1];
    for (int i = threadIdx.x; i <= B && i <= kMaxBatchLds; i += kTile) offs[i] = a.offsets[i];
    __syncthreads();
    RT *reg = reinterpret_cast<RT *>(a.reg);
    for (int g = blockIdx.x * kTile + threadIdx.x; g < total; g += gridDim.x * kTile) {
        int b, beg;
        if (B <= kMaxBatchLds) { b = sample_of_point(offs, B, g); beg = offs[b]; }
        else { b = 0; while (b + 1 < B && g >= a.offsets[b + 1]) ++b; beg = a.offsets[b]; }
        const float *p = a.points + (int64_t)g * a.F;
        const int cx = cell_coord(p[0], a.rmin[0], a.vs[0]);
        const int cy = cell_coord(p[1], a.rmin[1], a.vs[1]);
        const int cz = cell_coord(p[2], a.rmin[2], a.vs[2]);
        unsigned r = ~0u, loc = 0;
        if (!(cx < 0 || cx >= a.gx || cy < 0 || cy >= a.gy || cz < 0 || cz >= a.gz)) {
            const int cell = (cz * a.gy + cy) * a.gx + cx;
            r = (unsigned)cell >> a.shift;
            loc = (unsigned)cell & (unsigned)(a.R - 1);
        }
        const int64_t f = flag_base(beg, b) + (g - beg);
        reg[f] = (RT)r;
        a.cloc[f] = (unsigned short)loc;
        a.flag[f] = 0;
        if (g + 1 == (B <= kMaxBatchLds ? offs[b + 1] : a.offsets[b + 1]))      // the sample's last point also fills the slice up to its
            for (int64_t q = f + 1; q & 15; ++q) { reg[q] = (RT)~0u; a.flag[q] = 0; }   // 16-element boundary: no region, not a head
    }
}

// which of the points of a streamed dword belong to the region (rrrr: its id in every byte / half): bit 8k+7 (bytes) / 16k+15
// (halves) of the result for point k.  Exact (no borrow runs from a matching byte into the next), five operations
__device__ __forceinline__ unsigned match_bytes(unsigned w, unsigned rrrr) {
    const unsigned x = w ^ rrrr;
    return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
}
__device__ __forceinline__ unsigned match_halves(unsigned w, unsigned rr) {
    const unsigned x = w ^ rr;
    return ~(((x & 0x7fff7fffu) + 0x7fff7fffu) | x) & 0x80008000u;
}

// ---- the fused cells + own launch (round 6) -------------------------------------------------------------------------------
// vox_cells and vox_own as ONE launch: the first workgroups do the cells pass -- a workgroup per (sample, tile of kCellTile
// points) --, the ones behind them are the region owners.  An owner needs the region bytes and cell positions of its whole
// sample; it sets up its LDS while the cells workgroups of its sample work, then waits for them: every cells workgroup
// writes its bytes with write-through (sc1) 16-byte stores, drains them (vmcnt(0) in every storing wave, then the
// workgroup's barrier) and leaves the launch's token in a word of its own; an owner's first wave polls its sample's words
// with sc1 loads, the workgroup's barrier releases the other waves, and every load of the handed-over bytes is an sc1
// load (MI355X_MICROARCH.md, "Inter-workgroup visibility": flag per storing workgroup, sc1 on both sides, no fences).
// Workgroups are dispatched in index order and an owner only waits for workgroups in front of it, so the wait ends whatever
// the residency is; it is bounded all the same (a wait that runs out leaves the header without its magic: the table is
// refused by everything that reads it).  The token is the value of a word in the scratch + 1, advanced by vox_emit -- the
// launch behind this one -- so a captured graph replays with fresh tokens, and whatever an uninitialised scratch holds
// cannot look "done" unless a word happens to hold exactly the token.
__device__ __forceinline__ int cell_tiles(int n) { const int t = (n + kCellTile - 1) / kCellTile; return t > 0 ? t : 1; }
// the word cells workgroup w leaves when its bytes are out: made of the launch's epoch AND of w, so that a scratch buffer that
// holds the same value everywhere (fresh zeros; an older call's 0xFF region bytes where the epoch word now lies and its zeroed
// count bytes where the words now lie -- tests/soak/fuzz_lidar.py found exactly that) cannot look "done" in more than one place
__device__ __forceinline__ unsigned cells_token(unsigned epoch, int w) { return (epoch + 1u) * 0x9E3779B1u + (unsigned)w * 0x85EBCA6Bu + 0x2545F491u; }

// first cells workgroup and number of cells workgroups of sample b (every wave for itself, same answer)
__device__ __forceinline__ void cells_range(const int32_t *offsets, int B, int b, int *first_out, int *count_out) {
    const int lane = threadIdx.x & 63;
    int first = 0, count = 1;
    for (int b0 = 0; b0 <= b; b0 += 64) {
        const int s = b0 + lane;
        const int n = s < B ? offsets[s + 1] - offsets[s] : 0;
        const int t = s < B && s <= b ? cell_tiles(n) : 0;
        const int incl = wave_incl_scan(t);
        if (b < b0 + 64) {
            first += __builtin_amdgcn_readlane(incl - t, b - b0);
            count = __builtin_amdgcn_readlane(t, b - b0);
        } else first += __builtin_amdgcn_readlane(incl, 63);
    }
    *first_out = first; *count_out = count;
}

// LDS (the owners' dynamic block, free in a cells workgroup): region ids [kCellTile] RT | cell positions [kCellTile] u16
template <typename RT>
__device__ __forceinline__ void cells_role(const VoxArgs &a, int B, int widx) {
    extern __shared__ __align__(16) int lds[];
    RT *l_reg = reinterpret_cast<RT *>(lds);
    unsigned short *l_loc = reinterpret_cast<unsigned short *>(l_reg + kCellTile);
    const unsigned epoch = __hip_atomic_load(a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (sample, tile) of this workgroup: sample b owns cell_tiles(n_b) consecutive workgroups
    int b = -1, tile = 0, beg = 0, n = 0;
    {
        int first = 0;
        for (int s = 0; s < B; ++s) {              // (B is small; the offsets are uniform: scalar loads)
            const int o0 = a.offsets[s], ns = a.offsets[s + 1] - o0, t = cell_tiles(ns);
            if (widx < first + t) { b = s; tile = widx - first; beg = o0; n = ns; break; }
            first += t;
        }
    }
    if (b < 0) return;
    const int p0 = tile * kCellTile;
    const int npad = (n + 15) & ~15;               // the sample's slice is filled up to its 16-element boundary: no region, not a head
    const int m = npad - p0 < kCellTile ? npad - p0 : kCellTile;        // elements this workgroup leaves (a multiple of 16; 0 for an empty sample)
#pragma unroll
    for (int it = 0; it < kCellTile / kTile; ++it) {
        const int k = it * kTile + threadIdx.x, i = p0 + k;
        unsigned r = ~0u, loc = 0;
        if (i < n) {
            const float *p = a.points + (int64_t)(beg + i) * a.F;
            const int cx = cell_coord(p[0], a.rmin[0], a.vs[0]);
            const int cy = cell_coord(p[1], a.rmin[1], a.vs[1]);
            const int cz = cell_coord(p[2], a.rmin[2], a.vs[2]);
            if (!(cx < 0 || cx >= a.gx || cy < 0 || cy >= a.gy || cz < 0 || cz >= a.gz)) {
                const int cell = (cz * a.gy + cy) * a.gx + cx;
                r = (unsigned)cell >> a.shift;
                loc = (unsigned)cell & (unsigned)(a.R - 1);
            }
        }
        l_reg[k] = (RT)r;
        l_loc[k] = (unsigned short)loc;
    }
    __syncthreads();
    // out as 16-byte write-through stores: region ids, cell positions, zeroed count bytes
    const int64_t f0 = flag_base(beg, b) + p0;
    const __amdgpu_buffer_rsrc_t rs_reg = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<RT *>(a.reg) + f0, 0, m * (int)sizeof(RT), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_loc = __builtin_amdgcn_make_buffer_rsrc(a.cloc + f0, 0, m * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_flg = __builtin_amdgcn_make_buffer_rsrc(a.flag + f0, 0, m, 0x00020000);
    const mmt_u32x4 zero = {0u, 0u, 0u, 0u};
    for (int c = threadIdx.x; c < m * (int)sizeof(RT) / 16; c += kTile)
        __builtin_amdgcn_raw_buffer_store_b128(reinterpret_cast<const mmt_u32x4 *>(l_reg)[c], rs_reg, c * 16, 0, kSc1);
    for (int c = threadIdx.x; c < m * 2 / 16; c += kTile)
        __builtin_amdgcn_raw_buffer_store_b128(reinterpret_cast<const mmt_u32x4 *>(l_loc)[c], rs_loc, c * 16, 0, kSc1);
    for (int c = threadIdx.x; c < m / 16; c += kTile) __builtin_amdgcn_raw_buffer_store_b128(zero, rs_flg, c * 16, 0, kSc1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(a.cdone + widx, cells_token(epoch, widx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS: head [R] | cur [R] | eidx [kEntries] | enext [kEntries] | filled (u8) [R]
// FUSED: a region owner of the fused cells + own launch (waits for its sample's cells workgroups, reads what they left with sc1 loads)
template <typename RT, bool FUSED>
__device__ __forceinline__ void own_role(const VoxArgs &a, int B, int widx) {
    extern __shared__ __align__(16) int lds[];
    __shared__ int s_count, s_over;
    __shared__ int wpop[kMaxRegion / 64], wpre[kMaxRegion / 64];
    constexpr bool kWide = sizeof(RT) == 2;
    const int R = a.R, T = a.max_points;
    int *head = lds;
    int *cur = head + R;
    int *eidx = cur + R;
    int *enext = eidx + kEntries;
    unsigned char *filled = reinterpret_cast<unsigned char *>(enext + kEntries);
    // (sample, region) of this workgroup.  Workgroup i runs on XCD i % 8: with B | 8 a sample's regions take 8 / B XCDs, with
    // 8 | B one -- the region bytes and cell positions every workgroup of a sample reads then sit in that many L2s
    VSTAMP(0);
    int b, r;
    {
        const int x = widx & 7, q = widx >> 3;
        if (B <= 8 && 8 % B == 0) { const int xps = 8 / B; b = x / xps; r = q * xps + x % xps; }
        else if (B % 8 == 0) { const int per = B / 8; b = x + 8 * (q % per); r = q / per; }
        else { b = widx / a.NR; r = widx - b * a.NR; }
        if (b >= B || r >= a.NR) return;
    }
    const int beg = a.offsets[b], n = a.offsets[b + 1] - beg;
    const int64_t fb = flag_base(beg, b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int npad_ = (n + 15) & ~15;
    // (fused launch: the sample's region ids and cell positions through buffer descriptors -- sc1 loads)
    const __amdgpu_buffer_rsrc_t rs_reg = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<RT *>(a.reg) + fb, 0, npad_ * (int)sizeof(RT), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cloc = __builtin_amdgcn_make_buffer_rsrc(a.cloc + fb, 0, npad_ * 2, 0x00020000);
    for (int c = threadIdx.x * 4; c < R; c += kTile * 4) {         // (16-byte LDS stores; R is a multiple of 1024)
        *reinterpret_cast<int4 *>(head + c) = make_int4(kInf, kInf, kInf, kInf);
        *reinterpret_cast<int4 *>(cur + c) = make_int4(kInf, kInf, kInf, kInf);
        *reinterpret_cast<int *>(filled + c) = 0;
    }
    if (threadIdx.x == 0) { s_count = 0; s_over = 0; }
    if (FUSED && wave == 0) {
        // the sample's cells workgroups: all of their words must hold this launch's token
        const unsigned epoch = __hip_atomic_load(a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), token = epoch + 1u;
        int first, count;
        cells_range(a.offsets, B, b, &first, &count);
        int spins = 0;
        for (int c0 = 0; c0 < count; c0 += 64) {
            const int wi = first + (c0 + lane < count ? c0 + lane : count - 1);
            const unsigned *w = a.cdone + wi;
            const unsigned want = cells_token(epoch, wi);
            while (!__all(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want)) {
                if (++spins > (1 << 20)) break;                    // (~0.1 s: never met; the table is left without its magic)
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (spins > (1 << 20) && lane == 0) a.header[1] = (int32_t)(token ^ 0xDEADDEADu);      // vox_emit does not sign a table that carries this launch's mark
    }
    __syncthreads();

    VSTAMP(1);
    // settle what waits in eidx[0, count): every thread takes its share into registers and reads the cell positions (one gather);
    // an LDS atomicExch per point threads it onto its cell's chain (arrival order: any); every point then walks its cell's chain
    // once and knows its rank among the cell's points (those with a smaller index), the cell's smallest index and its count --
    // order-independent, so bit-reproducible.  Rank 0 of a cell nobody had reached before is the voxel's head; ranks below
    // max_points go to the head's list; the rest is dropped.  (A cell with very many points costs its points a long walk each;
    // the walk of a dropped point ends as soon as it has seen max_points smaller indices.)
    auto flush = [&](int count) {
        constexpr int kPer = kEntries / kTile;
        int idx[kPer], cl[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) idx[j] = j * kTile + threadIdx.x < count ? eidx[j * kTile + threadIdx.x] : -1;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {                             // (unconditional: one round trip for all)
            if (FUSED) cl[j] = (int)__builtin_amdgcn_raw_buffer_load_b16(rs_cloc, (unsigned)(idx[j] >= 0 ? idx[j] : 0) * 2u, 0, kSc1);
            else cl[j] = (int)a.cloc[fb + (idx[j] >= 0 ? idx[j] : 0)];
        }
        VSTAMP(5);
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            if (j * kTile >= count) break;
            if (idx[j] >= 0) enext[j * kTile + threadIdx.x] = atomicExch(&cur[cl[j]], j * kTile + threadIdx.x);
        }
        __syncthreads();
        int first[kPer];                                           // >= 0: this point is the smallest of its cell in this call -> the count to leave
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            first[j] = -1;
            if (j * kTile >= count) break;
            if (idx[j] < 0) continue;
            const int c = cl[j], had = filled[c];
            if (had >= T) continue;                                // (the cell already has its max_points: nothing of this call is kept)
            int rank = 0, mn = idx[j], cnt = 0;
            for (int e = cur[c]; e != kInf; e = enext[e]) {
                const int v = eidx[e];
                rank += v < idx[j] ? 1 : 0;
                mn = v < mn ? v : mn;
                ++cnt;
                if (had + rank >= T) break;                        // dropped whatever follows
            }
            const int r = had + rank;
            if (r < T) {
                const int h = had > 0 ? head[c] : mn;
                if (r > 0) a.lists[(int64_t)(beg + h) * (T - 1) + (r - 1)] = idx[j];
                if (rank == 0) first[j] = had + cnt < T ? had + cnt : T;      // (rank 0 walked the whole chain: cnt is the cell's count)
            }
        }
        __syncthreads();                                           // every walk is over: the cells' states may change
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            if (j * kTile >= count) break;
            if (idx[j] < 0) continue;
            const int c = cl[j];
            if (first[j] >= 0) {
                if (filled[c] == 0) head[c] = idx[j];
                filled[c] = (unsigned char)first[j];
            }
            cur[c] = kInf;                                         // (every point of the cell writes the same value)
        }
        __syncthreads();
    };

    // ---- stream the sample's region ids: 16 bytes per thread and step (16 points; 8 when the ids are 16 bits wide),
    // kDepth steps in flight.  A step's four dwords become ONE mask of the points that fall into region r; a thread counts
    // its matches of the whole batch, reserves their places in the list with one LDS atomic and writes them (a region's share of
    // 40 k points is a few hundred indices: the first match of a step without a loop, further ones in a loop few waves enter).
    // Should the list overflow -- a cloud packed into few regions -- the batch is done again step by step, settling the list
    // whenever the next step could overflow it (later steps only hold larger indices: the cells' lists stay ascending).
    const RT *reg = reinterpret_cast<const RT *>(a.reg) + fb;
    const int npad = (n + 15) & ~15;                               // (vox_cells filled the slice up to here: no region)
    const unsigned rrrr = kWide ? (unsigned)r * 0x00010001u : (unsigned)r * 0x01010101u;
    constexpr int kDepth = kStreamDepth;
    constexpr int kPts = kWide ? 8 : 16;                           // points per 16-byte load
    constexpr int kChunk = kTile * kPts;
    // point of mask bit j inside its step: bytes -- bit 8k + d is byte k of dword d; halves -- bit 16k + d is half k of dword d
    auto point_of = [](int j) { return kWide ? (j & 15) * 2 + (j >> 4) : (j & 7) * 4 + (j >> 3); };
    for (int base = 0; base < n; base += kChunk * kDepth) {
        unsigned w[kDepth];
        {
            uint4 v[kDepth];
#pragma unroll
            for (int u = 0; u < kDepth; ++u) {               // unconditional loads (clamped address): all in flight together
                const int p = base + u * kChunk + threadIdx.x * kPts;
                if (FUSED) {
                    const mmt_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs_reg, (unsigned)(p < npad ? p : npad - kPts) * (unsigned)sizeof(RT), 0, kSc1);
                    v[u] = make_uint4(q.x, q.y, q.z, q.w);
                } else v[u] = *reinterpret_cast<const uint4 *>(reg + (p < npad ? p : npad - kPts));
            }
#pragma unroll
            for (int u = 0; u < kDepth; ++u) {
                if (base + u * kChunk + (int)threadIdx.x * kPts >= npad) v[u] = make_uint4(~0u, ~0u, ~0u, ~0u);
                if (kWide) w[u] = (match_halves(v[u].x, rrrr) >> 15) | (match_halves(v[u].y, rrrr) >> 14) | (match_halves(v[u].z, rrrr) >> 13) | (match_halves(v[u].w, rrrr) >> 12);
                else w[u] = (match_bytes(v[u].x, rrrr) >> 7) | (match_bytes(v[u].y, rrrr) >> 6) | (match_bytes(v[u].z, rrrr) >> 5) | (match_bytes(v[u].w, rrrr) >> 4);
            }
        }
        VSTAMP(6);
        int before = 0;
        if (base > 0) { before = s_count; __syncthreads(); }      // (uniform: nobody appends between the last barrier and this one)
        int mine = 0;
#pragma unroll
        for (int u = 0; u < kDepth; ++u) mine += __popc(w[u]);
        // places in the list: a prefix sum inside the wave, ONE LDS atomic per wave (256 returning atomics on one LDS word cost 3 us)
        const int incl = wave_incl_scan(mine);
        int wbase = 0;
        if (lane == 63 && incl > 0) wbase = atomicAdd(&s_count, incl);
        int pos = __builtin_amdgcn_readlane(wbase, 63) + incl - mine;
        if (pos + mine <= kEntries) {
            unsigned rest = 0;
#pragma unroll
            for (int u = 0; u < kDepth; ++u) {               // a step's first match
                if (w[u]) {
                    eidx[pos++] = base + u * kChunk + threadIdx.x * kPts + point_of(__ffs((int)w[u]) - 1);
                    w[u] &= w[u] - 1;
                }
                rest |= w[u];
            }
            if (__any(rest != 0)) {
#pragma unroll
                for (int u = 0; u < kDepth; ++u)
                    while (w[u]) {
                        eidx[pos++] = base + u * kChunk + threadIdx.x * kPts + point_of(__ffs((int)w[u]) - 1);
                        w[u] &= w[u] - 1;
                    }
            }
        }
        // did everything fit?  One barrier decides for everybody: the count only grows, and the thread whose reservation came
        // last reads the full count afterwards, so the OR of "what I see is past the end" is the exact answer
        VSTAMP(7);
        if (s_count > kEntries) s_over = 1;
        __syncthreads();
        if (s_over) {
            __syncthreads();
            if (threadIdx.x == 0) { s_count = before; s_over = 0; }
            __syncthreads();
            const int end = base + kChunk * kDepth < n ? base + kChunk * kDepth : n;
            for (int p0 = base; p0 < end; p0 += kSlowChunk) {     // ascending steps of 8 points per thread, read again
                if (__syncthreads_or(s_count > kEntries - kSlowChunk ? 1 : 0)) {     // could this step overflow the list?  (same argument)
                    flush(s_count);
                    if (threadIdx.x == 0) s_count = 0;
                    __syncthreads();
                }
                const int p = p0 + threadIdx.x * 8;
                if (p >= npad) continue;                           // (no barrier below)
                unsigned v[kWide ? 4 : 2];
                if (kWide) {
                    uint4 q;
                    if (FUSED) { const mmt_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_reg, (unsigned)p * 2u, 0, kSc1); q = make_uint4(t.x, t.y, t.z, t.w); }
                    else q = *reinterpret_cast<const uint4 *>(reg + p);
                    v[0] = q.x; v[1] = q.y; v[kWide ? 2 : 0] = q.z; v[kWide ? 3 : 1] = q.w;
                } else {
                    uint2 q;
                    if (FUSED) { typedef unsigned u32x2 __attribute__((ext_vector_type(2))); const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs_reg, (unsigned)p, 0, kSc1); q = make_uint2(t.x, t.y); }
                    else q = *reinterpret_cast<const uint2 *>(reg + p);
                    v[0] = q.x; v[1] = q.y;
                }
#pragma unroll
                for (int d = 0; d < (kWide ? 4 : 2); ++d) {
                    unsigned m = kWide ? match_halves(v[d], rrrr) : match_bytes(v[d], rrrr);
                    while (m) {
                        const int k = (__ffs((int)m) - 1) >> (kWide ? 4 : 3);
                        m &= m - 1;
                        eidx[atomicAdd(&s_count, 1)] = p + d * (kWide ? 2 : 4) + k;
                    }
                }
            }
            __syncthreads();
        }
    }
    VSTAMP(2);
    {
        const int count = s_count;
        if (count > 0) flush(count);
    }
    // heads per 256-point tile (for vox_emit: the heads in front of a tile): histogram in the list's LDS, which is free now
    int *thist = eidx;
    const int my_tiles = (n + kTile - 1) / kTile;
    if (a.tcum)
        for (int t = threadIdx.x; t <= my_tiles; t += kTile) thist[t] = 0;
    __syncthreads();
    VSTAMP(3);

    // ---- the region's slice of the directory: occupancy bits, block ordinals, heads in cell order; the count byte at every head
    const int nwords = R >> 6;
    const int64_t region = (int64_t)b * a.NR + r;
    // A thread takes 8 consecutive cells per pass (a wave 512: eight 64-cell blocks, a block = 8 lanes): one LDS round trip for
    // the heads and counts, the block's occupancy word by three DPP ORs across its 8 lanes, a cell's ordinal from the block's
    // ordinal (a scan over the region's <= 64 blocks by one wave) + the occupied cells before it inside the word.
    const int passes = R > kTile * 8 ? R / (kTile * 8) : 1;        // 1 (half of the threads idle in a region of 1024 cells), 2 or 4
    unsigned long long word[kMaxRegion / (kTile * 8)];
    int hd[kMaxRegion / (kTile * 8)][8];
    unsigned long long fl8[kMaxRegion / (kTile * 8)];
#pragma unroll
    for (int ps = 0; ps < kMaxRegion / (kTile * 8); ++ps) {
        if (ps >= passes) break;
        const int c0 = (ps * kTile + threadIdx.x) * 8;
        const bool in = c0 < R;                                    // (whole blocks of 8 lanes are in or out)
        const int4 none = make_int4(kInf, kInf, kInf, kInf);
        const int4 h0 = in ? *reinterpret_cast<const int4 *>(head + c0) : none, h1 = in ? *reinterpret_cast<const int4 *>(head + c0 + 4) : none;
        fl8[ps] = in ? *reinterpret_cast<const unsigned long long *>(filled + c0) : 0ull;
        hd[ps][0] = h0.x; hd[ps][1] = h0.y; hd[ps][2] = h0.z; hd[ps][3] = h0.w;
        hd[ps][4] = h1.x; hd[ps][5] = h1.y; hd[ps][6] = h1.z; hd[ps][7] = h1.w;
        unsigned m8 = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            m8 |= (hd[ps][k] != kInf ? 1u : 0u) << k;
            if (a.tcum && hd[ps][k] != kInf) atomicAdd(&thist[hd[ps][k] >> 8], 1);
        }
        const int sub = lane & 7;
        unsigned lo = sub < 4 ? m8 << (8 * sub) : 0u, hi = sub >= 4 ? m8 << (8 * (sub - 4)) : 0u;
        // OR across the 8 lanes of the block: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror
        lo |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xF, 0xF, false); hi |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xF, 0xF, false);
        lo |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0x4E, 0xF, 0xF, false); hi |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0x4E, 0xF, 0xF, false);
        lo |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0x141, 0xF, 0xF, false); hi |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0x141, 0xF, 0xF, false);
        word[ps] = ((unsigned long long)hi << 32) | lo;
        if (sub == 0 && in) {
            const int wi = (ps * kTile + threadIdx.x) >> 3;         // the block's index inside the region
            wpop[wi] = __popcll(word[ps]);
            a.bits[region * nwords + wi] = word[ps];
        }
    }
    __syncthreads();
    if (wave == 0) {
        const int v = lane < nwords ? wpop[lane] : 0;
        const int incl = wave_incl_scan(v);
        if (lane < nwords) { wpre[lane] = incl - v; a.pref[region * nwords + lane] = (unsigned short)(incl - v); }
    } else if (wave == 1 && a.tcum) {
        // exclusive prefix over the tiles (64 at a time, carried on), left as row t of the sample's table, column r
        unsigned short *row = a.tcum + (int64_t)b * (a.ntiles + 1) * a.NR + r;
        int carry = 0;
        for (int t0 = 0; t0 <= my_tiles; t0 += 64) {
            const int t = t0 + lane;
            const int v = t <= my_tiles ? thist[t] : 0;
            const int incl = wave_incl_scan(v);
            if (t <= my_tiles) row[(int64_t)t * a.NR] = (unsigned short)(carry + incl - v);
            carry += __builtin_amdgcn_readlane(incl, 63);
        }
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < kMaxRegion / (kTile * 8); ++ps) {
        if (ps >= passes) break;
        const int wi = (ps * kTile + threadIdx.x) >> 3, sub = lane & 7;
        int ord = (wi < nwords ? wpre[wi] : 0) + __popcll(word[ps] & ((1ull << (8 * sub)) - 1ull));
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (hd[ps][k] != kInf) {
                a.head_of[region * R + ord++] = beg + hd[ps][k];
                a.flag[fb + hd[ps][k]] = (unsigned char)(fl8[ps] >> (8 * k));
            }
    }
    VSTAMP(4);
}

template <typename RT>
__global__ __launch_bounds__(kTile) void vox_own(VoxArgs a, int B) { own_role<RT, false>(a, B, (int)blockIdx.x); }

// the first ncells workgroups do the cells pass, the rest are the region owners
template <typename RT>
__global__ __launch_bounds__(kTile) void vox_cells_own(VoxArgs a, int B, int ncells) {
    if ((int)blockIdx.x < ncells) cells_role<RT>(a, B, (int)blockIdx.x);
    else own_role<RT, true>(a, B, (int)blockIdx.x - ncells);
}

// nonzero bytes of a dword of count bytes (each < 128: max_points <= 127), added to acc
__device__ __forceinline__ int nonzero_bytes(unsigned v, int acc) { return acc + __popc((v + 0x7f7f7f7fu) & 0x80808080u); }

// LDS: lists [kTile][T] sorted point indices | cnt [kTile] | cellv [kTile].  FT = compile-time F (0 = any)
template <int FT>
__global__ __launch_bounds__(kTile) void vox_emit(VoxArgs a, int B) {
    extern __shared__ __align__(16) int lds[];
    __shared__ int s_before[kTileWaves], s_all[kTileWaves], s_heads[kTileWaves];
    const int T = a.max_points, F = FT > 0 ? FT : a.F, V = a.max_voxels;
    const int TF = T * F;
    int *lists = lds;
    int *cnt = lists + kTile * T;
    int *cellv = cnt + kTile;
    VSTAMP(8);
    int b, tile, beg, n;
    if (B <= 63) {
        if (!locate_tile_wave(a.offsets, B, blockIdx.x, &b, &tile, &beg, &n)) return;
    } else {
        if (!locate_tile(a.offsets, B, blockIdx.x, &b, &tile)) return;
        beg = a.offsets[b]; n = a.offsets[b + 1] - beg;
    }
    VSTAMP(9);
    const int my_tiles = (n + kTile - 1) / kTile;
    const int64_t fb = flag_base(beg, b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // the fused cells + own launch in front of this one made its token of *epoch + 1: sign the table unless an owner gave up
        // waiting (own_role), and advance the word for the next voxelization (a captured graph replays with fresh tokens)
        const unsigned e = *a.epoch;
        a.header[0] = (unsigned)a.header[1] == ((e + 1u) ^ 0xDEADDEADu) ? 0 : kTableMagic;
        *a.epoch = e + 1u;
    }

    // this point's count byte and cell: requested together with the stream of the sample's count bytes
    const int i = tile * kTile + threadIdx.x;
    int fl = 0, my_cell = -1;
    if (i < n) {
        fl = a.flag[fb + i];
        const unsigned rg = a.wide ? reinterpret_cast<const unsigned short *>(a.reg)[fb + i] : reinterpret_cast<const unsigned char *>(a.reg)[fb + i];
        my_cell = (int)((rg << a.shift) | a.cloc[fb + i]);
    }
    // heads before this tile / in the whole sample: the nonzero count bytes (16 per load, kFlagDepth loads in flight; the slice
    // is 16-byte aligned)
    int before = 0, all = 0;
    if (a.tcum) {                                               // vox_own left the heads in front of every tile, per region: two rows to add up
        const unsigned short *rows = a.tcum + (int64_t)b * (a.ntiles + 1) * a.NR;
        for (int q = threadIdx.x; q < a.NR; q += kTile) {
            before += rows[(int64_t)tile * a.NR + q];
            all += rows[(int64_t)my_tiles * a.NR + q];
        }
    } else {
        constexpr int kFlagDepth = 10;                          // 10 x 256 x 16 bytes: a 40 k cloud in one round trip
        const uint4 *fw = reinterpret_cast<const uint4 *>(a.flag + fb);
        const int nw = (n + 15) >> 4, bw = tile * (kTile / 16);
        const int rot = (int)((unsigned)(tile * 3) % kFlagDepth);
        auto frot = [&](int u) { const int x = u + rot; return x >= kFlagDepth ? x - kFlagDepth : x; };
        for (int w0 = 0; w0 < nw; w0 += kTile * kFlagDepth) {
            uint4 v[kFlagDepth];
#pragma unroll
            for (int u = 0; u < kFlagDepth; ++u) {                  // (every tile of the sample reads these bytes: each starts at another step)
                const int w = w0 + frot(u) * kTile + threadIdx.x;
                v[u] = fw[w < nw ? w : nw - 1];
            }
#pragma unroll
            for (int u = 0; u < kFlagDepth; ++u)
                if (w0 + frot(u) * kTile + (int)threadIdx.x >= nw) v[u] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int u = 0; u < kFlagDepth; ++u) {                  // (vox_cells zeroed the bytes between the sample's end and the 16-byte boundary)
                const int c = nonzero_bytes(v[u].x, nonzero_bytes(v[u].y, nonzero_bytes(v[u].z, nonzero_bytes(v[u].w, 0))));
                all += c;
                if (w0 + frot(u) * kTile + (int)threadIdx.x < bw) before += c;
            }
        }
    }
    VSTAMP(13);
    const bool h = fl != 0;
    const unsigned long long m = __ballot(h);
    for (int o = 32; o > 0; o >>= 1) {
        before += __shfl_down(before, o);
        all += __shfl_down(all, o);
    }
    if (lane == 0) { s_before[wave] = before; s_all[wave] = all; s_heads[wave] = __popcll(m); }
    __syncthreads();
    int off = 0, total = 0, woff = 0, nheads = 0;
#pragma unroll
    for (int w = 0; w < kTileWaves; ++w) {
        off += s_before[w];
        total += s_all[w];
        if (w < wave) woff += s_heads[w];
        nheads += s_heads[w];
    }
    const int M = total < V ? total : V;                // voxels of this sample
    if (tile == 0 && threadIdx.x == 0) a.voxel_count[b] = M;
    const int nown = (off + nheads <= V) ? nheads : (V - off > 0 ? V - off : 0);   // heads of this tile below the cap

    VSTAMP(10);
    // ---- every head: its voxel id (heads before it), its list into LDS
    if (h) {
        const int r = woff + __popcll(m & ((1ull << lane) - 1ull));
        a.vidp[beg + i] = r < nown ? off + r : -1;
        if (r < nown) {
            int *L = lists + r * T;
            L[0] = i;
            const int32_t *src = a.lists + (int64_t)(beg + i) * (T - 1);
            for (int k = 1; k < fl; ++k) L[k] = src[k - 1];
            cnt[r] = fl;
            cellv[r] = my_cell;
        }
    }
    __syncthreads();

    VSTAMP(11);
    // ---- the tile's voxels are rows [off, off + nown) of the sample: flat, coalesced stores.  (row, element) of a
    // flat index advance incrementally (no integer division per element: (slot, column) of an element come from qmap)
    const int64_t row0 = (int64_t)b * V + off;
    if (a.voxels) {
        // one (voxel, slot) per thread and trip, four trips requested together: the point's F floats are stored as one
        // contiguous F*4-byte piece (the slots of a tile's voxels are one contiguous block of the output)
        float *dst = a.voxels + row0 * TF;
        const int total = nown * T;
        constexpr int kU = 4, kFmax = FT > 0 ? FT : 1;
        for (int e0 = threadIdx.x; e0 < total; e0 += kTile * kU) {
            if (FT > 0) {
                float v[kU][kFmax];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int e = e0 + u * kTile;
                    const int r = e < total ? e / T : 0, t = e < total ? e - r * T : 0;
                    const bool live = e < total && t < cnt[r];
                    const float *src = a.points + (int64_t)(beg + (live ? lists[r * T + t] : 0)) * F;
#pragma unroll
                    for (int f = 0; f < kFmax; ++f) v[u][f] = live ? src[f] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int e = e0 + u * kTile;
                    if (e < total) {
#pragma unroll
                        for (int f = 0; f < kFmax; ++f) dst[(int64_t)e * F + f] = v[u][f];
                    }
                }
            } else {
                for (int u = 0; u < kU; ++u) {
                    const int e = e0 + u * kTile;
                    if (e >= total) break;
                    const int r = e / T, t = e - r * T;
                    const bool live = t < cnt[r];
                    const float *src = a.points + (int64_t)(beg + (live ? lists[r * T + t] : 0)) * F;
                    for (int f = 0; f < F; ++f) dst[(int64_t)e * F + f] = live ? src[f] : 0.f;
                }
            }
        }
    }
    if (a.mean) {
        const int nf = a.nf;
        if (FT > 0) {
            // a voxel per thread: its points' rows four at a time (all requested before any is used), added in slot order
            constexpr int kFmax = FT > 0 ? FT : 1;
            for (int r = threadIdx.x; r < nown; r += kTile) {
                const int c = cnt[r];
                const int *L = lists + r * T;
                float acc[kFmax];
#pragma unroll
                for (int f = 0; f < kFmax; ++f) acc[f] = 0.f;
                for (int t0 = 0; t0 < c; t0 += 4) {
                    float v[4][kFmax];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float *src = a.points + (int64_t)(beg + L[t0 + u < c ? t0 + u : 0]) * FT;
#pragma unroll
                        for (int f = 0; f < kFmax; ++f) v[u][f] = src[f];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (t0 + u < c) {
#pragma unroll
                            for (int f = 0; f < kFmax; ++f) acc[f] = __fadd_rn(acc[f], v[u][f]);
                        }
                }
                float *dst = a.mean + (row0 + r) * nf;
#pragma unroll
                for (int f = 0; f < kFmax; ++f)
                    if (f < nf) dst[f] = __fdiv_rn(acc[f], (float)c);     // zero-padded slots add nothing; c >= 1
            }
        } else {
            float *dst = a.mean + row0 * nf;
            const int total = nown * nf;
            int r = threadIdx.x / nf, k = threadIdx.x - r * nf;
            const int dr = kTile / nf, dk = kTile - dr * nf;
            for (int e = threadIdx.x; e < total; e += kTile) {
                const int c = cnt[r];
                const int *L = lists + r * T;
                float sum = 0.f;
                for (int t = 0; t < c; ++t) sum = __fadd_rn(sum, a.points[(int64_t)(beg + L[t]) * F + k]);
                dst[e] = __fdiv_rn(sum, (float)c);
                r += dr; k += dk;
                if (k >= nf) { k -= nf; ++r; }
            }
        }
    }
    for (int r = threadIdx.x; r < nown; r += kTile) {
        const int cell = cellv[r];
        const int plane = a.gx * a.gy;
        const int z = cell / plane, rem = cell - z * plane;
        const int y = rem / a.gx;
        reinterpret_cast<int4 *>(a.coors)[row0 + r] = make_int4(b, z, y, rem - y * a.gx);
        a.num_points[row0 + r] = cnt[r];
    }

    VSTAMP(12);
    // ---- rows past the sample's voxel count: marked empty; each tile takes an equal share of them
    const int tiles = my_tiles > 0 ? my_tiles : 1;
    const int dead = V - M;
    const int per = (dead + tiles - 1) / tiles;
    const int d0 = M + tile * per;
    const int d1 = (d0 + per) < V ? (d0 + per) : V;
    if (d0 < d1) {
        const int64_t r0 = (int64_t)b * V + d0;
        const int nd = d1 - d0;
        for (int e = threadIdx.x; e < nd * 4; e += kTile) a.coors[r0 * 4 + e] = -1;
        for (int e = threadIdx.x; e < nd; e += kTile) a.num_points[r0 + e] = 0;
        if (a.mean)
            for (int e = threadIdx.x; e < nd * a.nf; e += kTile) a.mean[r0 * a.nf + e] = 0.f;
    }
}

__global__ __launch_bounds__(256) void compact_kernel(int max_voxels, int row_elems,
                                                      const int32_t *voxel_count,
                                                      const int32_t *dst_offsets, const float *voxels,
                                                      const int32_t *coors, const int32_t *num_points,
                                                      float *voxels_out, int32_t *coors_out,
                                                      int32_t *num_points_out) {
    const int b = blockIdx.y;
    const int M = voxel_count[b];
    const int64_t dst0 = dst_offsets[b];
    const int64_t total = (int64_t)M * row_elems;
    const float *src = voxels + (int64_t)b * max_voxels * row_elems;
    float *dst = voxels_out + dst0 * row_elems;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) {
        const int64_t s = (int64_t)b * max_voxels + i;
        num_points_out[dst0 + i] = num_points[s];
        for (int k = 0; k < 4; ++k) coors_out[(dst0 + i) * 4 + k] = coors[s * 4 + k];
    }
}

__global__ __launch_bounds__(256) void simple_vfe_kernel(int64_t M, int T, int F, int nf,
                                                         const float *voxels,
                                                         const int32_t *num_points, float *out) {
    const int64_t total = M * nf;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / nf;
        const int k = (int)(i - m * nf);
        const float *v = voxels + m * T * F + k;
        float s = 0.f;
        for (int t = 0; t < T; ++t) s = __fadd_rn(s, v[t * F]);  // slot order, like sum(dim=1)
        const int np = num_points[m];
        out[i] = np > 0 ? __fdiv_rn(s, (float)np) : 0.f;
    }
}

// cell -> row map: highest row index wins (sequential last-writer semantics)
// (sy, sx) > 1: the map of the canvas SAMPLED at cells (i * sy, j * sx) -- [B, ny / sy, nx / sx] entries; only the voxels
// on sampled cells enter it (mmt_pillar_scatter_nhwc_strided)
__global__ __launch_bounds__(256) void scatter_map_kernel(int64_t M, int B, int ny, int nx, int sy, int sx,
                                                          const int32_t *coors, int32_t *map) {
    const int oh = ny / sy, ow = nx / sx;
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
        const int b = coors[m * 4], y = coors[m * 4 + 2], x = coors[m * 4 + 3];
        if (b < 0 || b >= B || y < 0 || y >= ny || x < 0 || x >= nx) continue;
        if (y % sy || x % sx) continue;
        atomicMax(&map[((int64_t)b * oh + y / sy) * ow + x / sx], (int)m);
    }
}

// canvas[b,c,y,x] written exactly once, 16 bytes per lane along x.  A lane owns 4 consecutive
// cells for a block of kChanBlock channels: the 16-byte slice of the cell->row map is read once
// per channel block (not once per channel) and the stores of a wave stay within kChanBlock
// DRAM pages.  Empty cells -- the vast majority -- never touch the feature matrix; the canvas
// memset is folded into the same pass.
constexpr int kChanBlock = 4;

template <bool VEC4>
__global__ __launch_bounds__(256) void scatter_write_kernel(int C, int B, int HW, const float *feats,
                                                            const int32_t *map, float *canvas) {
    const int XV = VEC4 ? 4 : 1;
    const int per_c = HW / XV;
    const int cblocks = (C + kChanBlock - 1) / kChanBlock;
    const int64_t per_b = (int64_t)cblocks * per_c;
    const int64_t total = (int64_t)B * per_b;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / per_b);
        const int64_t r = i - b * per_b;
        const int cb = (int)(r / per_c);
        const int s = (int)(r - (int64_t)cb * per_c) * XV;
        const int c0 = cb * kChanBlock;
        const int32_t *mp = map + (int64_t)b * HW + s;
        float *dst = canvas + ((int64_t)b * C + c0) * HW + s;
        if (VEC4) {
            const int4 m4 = *reinterpret_cast<const int4 *>(mp);
            const bool empty = (m4.x & m4.y & m4.z & m4.w) == -1;
#pragma unroll
            for (int k = 0; k < kChanBlock; ++k) {
                const int c = c0 + k;
                if (c >= C) break;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!empty) {
                    if (m4.x >= 0) v.x = feats[(int64_t)m4.x * C + c];
                    if (m4.y >= 0) v.y = feats[(int64_t)m4.y * C + c];
                    if (m4.z >= 0) v.z = feats[(int64_t)m4.z * C + c];
                    if (m4.w >= 0) v.w = feats[(int64_t)m4.w * C + c];
                }
                mmt_nt_store4(v, reinterpret_cast<float4 *>(dst + (int64_t)k * HW));
            }
        } else {
            const int m = mp[0];
            for (int k = 0; k < kChanBlock && c0 + k < C; ++k)
                dst[(int64_t)k * HW] = m >= 0 ? feats[(int64_t)m * C + c0 + k] : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void scatter_backward_kernel(int64_t M, int C, int B, int ny, int nx,
                                                               const float *grad_canvas,
                                                               const int32_t *coors,
                                                               const int32_t *map, float *grad_feats) {
    const int64_t total = M * C;
    const int64_t HW = (int64_t)ny * nx;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / C;
        const int c = (int)(i - m * C);
        const int b = coors[m * 4], y = coors[m * 4 + 2], x = coors[m * 4 + 3];
        float v = 0.f;
        if (!(b < 0 || b >= B || y < 0 || y >= ny || x < 0 || x >= nx)) {
            const int64_t cell = (int64_t)y * nx + x;
            if (map[b * HW + cell] == (int)m) v = grad_canvas[((int64_t)b * C + c) * HW + cell];
        }
        grad_feats[i] = v;
    }
}

}  // namespace

namespace {

#ifdef VOX_STAMPS
unsigned long long *g_vox_stamps = nullptr;
#endif
int64_t vox_tiles(int64_t N) { return mmt::ceil_div(N > 0 ? N : 1, kTile); }
std::atomic<int> g_vox_fused{[] { const char *e = getenv("MMT_VOX_FUSED"); return e && atoi(e) != 0 ? 1 : 0; }()};

int vox_check(const char *what, int B, int64_t N, int F, const int32_t *grid_host, int max_points, int max_voxels, int nf) {
    if (B <= 0 || B > 65535 || N < 0 || F < 3 || max_points <= 0 || max_points > 127 || max_voxels <= 0 || nf < 0 || nf > F)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: bad sizes (B=%d N=%lld F=%d T=%d max_voxels=%d nf=%d)", what, B, (long long)N, F, max_points, max_voxels, nf);
    if (grid_host[0] <= 0 || grid_host[1] <= 0 || grid_host[2] <= 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: non-positive grid", what);
    const int64_t cells = (int64_t)grid_host[0] * grid_host[1] * grid_host[2];
    if (cells * B >= (1ll << 31) || cells >= (1ll << 31) || N >= (1ll << kIdxBits))
        return mmt::fail(MMT_ERR_TOO_LARGE, "%s: B*cells exceeds int32 or N >= 2^24 points", what);
    if (mmt::ceil_div(cells, kMaxRegion) > 65534)
        return mmt::fail(MMT_ERR_TOO_LARGE, "%s: more than 65534 regions of %d cells", what, kMaxRegion);
    if ((size_t)(kTile * (int64_t)max_points + 2 * kTile) * 4 > 150 * 1024)
        return mmt::fail(MMT_ERR_TOO_LARGE, "%s: max_points=%d too large for the LDS index lists", what, max_points);
    return 0;
}

int64_t align4(int64_t ints) { return (ints + 3) & ~(int64_t)3; }

// scratch (int32 units; every part starts on a 16-byte boundary of the aligned base): lists | region ids | cell positions | count bytes
constexpr int64_t kMaxTcumTiles = 2 * kEntries - 1;            // vox_own's tile histogram lives in the list's LDS (2 x kEntries ints)
struct ScratchLayout { int64_t lists, reg, cloc, flag, tcum, cdone, elems; };
ScratchLayout scratch_layout(int B, int64_t N, int max_points, bool wide, int NR) {
    ScratchLayout l;
    const int64_t slots = N + 32 * (int64_t)B + 32;          // flag positions (flag_base: every sample's slice 16-byte aligned)
    l.lists = 0;
    l.reg = l.lists + align4(N * (max_points > 1 ? max_points - 1 : 0));
    l.cloc = l.reg + align4(mmt::ceil_div(slots * (wide ? 2 : 1), 4));
    l.flag = l.cloc + align4(mmt::ceil_div(slots * 2, 4));
    l.tcum = l.flag + align4(mmt::ceil_div(slots, 4));
    l.cdone = l.tcum + align4(mmt::ceil_div((int64_t)B * (vox_tiles(N) + 1) * NR, 2));       // [0]: the epoch word; [4 ..]: the cells workgroups' words
    l.elems = l.cdone + align4(4 + mmt::ceil_div(N > 0 ? N : 1, kCellTile) + B) + 4;          // + 4: the base is aligned up to 16 bytes
    return l;
}

// the three kernels of one voxelization (table: 8-byte aligned, any contents; scratch: any contents)
int vox_run(const char *what, int B, int64_t N, int F, const float *points, const int32_t *point_offsets,
            const float *voxel_size_host, const float *range_min_host, const int32_t *grid_host, int max_points,
            int max_voxels, int nf, float *voxels, int32_t *coors, int32_t *num_points, int32_t *voxel_count,
            float *mean, int32_t *table, int32_t *scratch, hipStream_t st) {
    VoxArgs a;
    a.F = F; a.max_points = max_points; a.max_voxels = max_voxels; a.nf = nf;
    a.gx = grid_host[0]; a.gy = grid_host[1]; a.gz = grid_host[2];
    for (int k = 0; k < 3; ++k) { a.vs[k] = voxel_size_host[k]; a.rmin[k] = range_min_host[k]; }
    a.points = points; a.offsets = point_offsets;
    a.ntiles = (int)vox_tiles(N);
    a.stamps = nullptr;
#ifdef VOX_STAMPS
    { static unsigned long long *buf = nullptr; if (!buf) (void)hipMalloc(&buf, 1024 * 16 * 8); a.stamps = buf; g_vox_stamps = buf; }
#endif
    a.cells = (int64_t)a.gx * a.gy * a.gz;
    const Directory d = directory_of(B, a.cells);
    a.R = d.R; a.shift = d.shift; a.NR = d.NR; a.wide = d.NR > 254 ? 1 : 0;
    a.header = table;
    a.bits = reinterpret_cast<unsigned long long *>(table + d.bits);
    a.pref = reinterpret_cast<unsigned short *>(table + d.pref);
    a.head_of = table + d.head_of;
    a.vidp = table + d.vidp;
    int32_t *base = reinterpret_cast<int32_t *>(((uintptr_t)scratch + 15) & ~(uintptr_t)15);
    const ScratchLayout l = scratch_layout(B, N, max_points, a.wide != 0, d.NR);
    a.lists = base + l.lists;
    a.reg = base + l.reg;
    a.cloc = reinterpret_cast<unsigned short *>(base + l.cloc);
    a.flag = reinterpret_cast<unsigned char *>(base + l.flag);
    a.tcum = vox_tiles(N) <= kMaxTcumTiles ? reinterpret_cast<unsigned short *>(base + l.tcum) : nullptr;
    a.epoch = reinterpret_cast<unsigned *>(base + l.cdone);
    a.cdone = a.epoch + 4;
    a.voxels = voxels; a.coors = coors; a.num_points = num_points; a.voxel_count = voxel_count; a.mean = mean;
    mmt::TimedSeq seq;   // armed by mmt_arm_kernel_timing (bench only)
    const unsigned gpts = (unsigned)mmt::stream_grid(N > 0 ? N : 1, kTile, 4096);
    const unsigned gtiles = (unsigned)(vox_tiles(N) + B);       // flattened (sample, tile) space, see locate_tile
    // (sample, region) workgroups; the XCD-aware numbering of vox_own covers B * NR rounded up to whole rows of 8
    unsigned gown = (unsigned)((int64_t)B * d.NR);
    if (B <= 8 && 8 % B == 0) gown = 8u * (unsigned)mmt::ceil_div(d.NR, 8 / B);
    const size_t lds_own = (size_t)d.R * 9 + (size_t)kEntries * 8;
    const bool fused = g_vox_fused.load(std::memory_order_relaxed) != 0;      // (off by default: measured slower than the three launches, see mmt_voxelize_fused_launch)
    const int ncells = (int)(mmt::ceil_div(N > 0 ? N : 1, kCellTile) + B);       // cells workgroups of the fused launch: sample b has max(ceil(n_b / kCellTile), 1)
    if (fused) {
        if (a.wide) seq.launch(false, vox_cells_own<unsigned short>, dim3(ncells + gown), dim3(kTile), lds_own, st, a, B, ncells);
        else seq.launch(false, vox_cells_own<unsigned char>, dim3(ncells + gown), dim3(kTile), lds_own, st, a, B, ncells);
    } else if (a.wide) {
        seq.launch(false, vox_cells<unsigned short>, dim3(gpts), dim3(kTile), 0, st, a, B, (int)N);
        seq.launch(false, vox_own<unsigned short>, dim3(gown), dim3(kTile), lds_own, st, a, B);
    } else {
        seq.launch(false, vox_cells<unsigned char>, dim3(gpts), dim3(kTile), 0, st, a, B, (int)N);
        seq.launch(false, vox_own<unsigned char>, dim3(gown), dim3(kTile), lds_own, st, a, B);
    }
    const size_t lds = (size_t)(kTile * (int64_t)max_points + 2 * kTile) * 4;
    if (F == 5) seq.launch(true, vox_emit<5>, dim3(gtiles), dim3(kTile), lds, st, a, B);
    else if (F == 8) seq.launch(true, vox_emit<8>, dim3(gtiles), dim3(kTile), lds, st, a, B);
    else if (F == 4) seq.launch(true, vox_emit<4>, dim3(gtiles), dim3(kTile), lds, st, a, B);
    else seq.launch(true, vox_emit<0>, dim3(gtiles), dim3(kTile), lds, st, a, B);
    return mmt::check_launch(what);
}

}  // namespace

#ifdef VOX_STAMPS
extern "C" int mmt_vox_debug_stamps(unsigned long long *host) { if (!g_vox_stamps) return 1; (void)hipDeviceSynchronize(); return (int)hipMemcpy(host, g_vox_stamps, 1024 * 16 * 8, hipMemcpyDeviceToHost); }
#endif

extern "C" int mmt_voxelize_fused_launch(int on) {
    if (on < 0) return g_vox_fused.load(std::memory_order_relaxed);
    return g_vox_fused.exchange(on ? 1 : 0, std::memory_order_relaxed);
}

extern "C" int64_t mmt_voxelize_table_elems(int B, const int32_t *grid, int64_t total_points) {
    if (B <= 0 || grid == nullptr || total_points < 0) return 0;
    const int64_t cells = (int64_t)grid[0] * grid[1] * grid[2];
    if (cells <= 0) return 0;
    return directory_of(B, cells).elems_without_vidp + total_points + 4;
}

extern "C" int64_t mmt_voxelize_scratch_elems(int B, const int32_t *grid, int64_t total_points, int max_points) {
    if (B <= 0 || grid == nullptr || total_points < 0 || max_points <= 0) return 0;
    const int64_t cells = (int64_t)grid[0] * grid[1] * grid[2];
    if (cells <= 0) return 0;
    return scratch_layout(B, total_points, max_points, true, directory_of(B, cells).NR).elems;
}

extern "C" int64_t mmt_voxelize_workspace_elems(int B, int64_t total_points, const int32_t *grid, int max_points) {
    if (B <= 0 || total_points < 0 || grid == nullptr || max_points <= 0) return 0;
    return mmt_voxelize_table_elems(B, grid, total_points) + mmt_voxelize_scratch_elems(B, grid, total_points, max_points) + 2;
}

extern "C" int mmt_hard_voxelize_mean(int B, int64_t N, int F, const float *points,
                                      const int32_t *point_offsets, const float *voxel_size_host,
                                      const float *range_min_host, const int32_t *grid_host,
                                      int max_points, int max_voxels, int num_features, float *voxels,
                                      int32_t *coors, int32_t *num_points, int32_t *voxel_count,
                                      float *mean, int32_t *table, int32_t *scratch, void *stream) {
    MMT_REQUIRE_PTR(point_offsets);
    MMT_REQUIRE_PTR(voxel_size_host);
    MMT_REQUIRE_PTR(range_min_host);
    MMT_REQUIRE_PTR(grid_host);
    MMT_REQUIRE_PTR(coors);
    MMT_REQUIRE_PTR(num_points);
    MMT_REQUIRE_PTR(voxel_count);
    MMT_REQUIRE_PTR(table);
    MMT_REQUIRE_PTR(scratch);
    if (N > 0) MMT_REQUIRE_PTR(points);
    int rc = vox_check("hard_voxelize_mean", B, N, F, grid_host, max_points, max_voxels, num_features);
    if (rc) return rc;
    if (mean != nullptr && num_features <= 0) return mmt::fail(MMT_ERR_BAD_SHAPE, "hard_voxelize_mean: mean requested with num_features=%d", num_features);
    if ((uintptr_t)table & 7) return mmt::fail(MMT_ERR_WORKSPACE, "hard_voxelize_mean: table must be 8-byte aligned");
    return vox_run("hard_voxelize_mean", B, N, F, points, point_offsets, voxel_size_host, range_min_host, grid_host,
                   max_points, max_voxels, num_features, voxels, coors, num_points, voxel_count, mean, table, scratch, (hipStream_t)stream);
}

extern "C" int mmt_hard_voxelize(int B, int64_t N, int F, const float *points,
                                 const int32_t *point_offsets, const float *voxel_size_host,
                                 const float *range_min_host, const int32_t *grid_host,
                                 int max_points, int max_voxels, float *voxels, int32_t *coors,
                                 int32_t *num_points, int32_t *voxel_count, int32_t *workspace,
                                 void *stream) {
    MMT_REQUIRE_PTR(point_offsets);
    MMT_REQUIRE_PTR(voxel_size_host);
    MMT_REQUIRE_PTR(range_min_host);
    MMT_REQUIRE_PTR(grid_host);
    MMT_REQUIRE_PTR(voxels);
    MMT_REQUIRE_PTR(coors);
    MMT_REQUIRE_PTR(num_points);
    MMT_REQUIRE_PTR(voxel_count);
    MMT_REQUIRE_PTR(workspace);
    if (N > 0) MMT_REQUIRE_PTR(points);
    int rc = vox_check("hard_voxelize", B, N, F, grid_host, max_points, max_voxels, 0);
    if (rc) return rc;
    // the cell directory and the scratch both live in the caller's workspace (any contents: every word read is written first)
    int32_t *tab = workspace + (((uintptr_t)workspace & 7) ? 1 : 0);
    const int64_t tab_elems = mmt_voxelize_table_elems(B, grid_host, N);
    return vox_run("hard_voxelize", B, N, F, points, point_offsets, voxel_size_host, range_min_host, grid_host,
                   max_points, max_voxels, 0, voxels, coors, num_points, voxel_count, nullptr, tab, tab + tab_elems, (hipStream_t)stream);
}

extern "C" int mmt_compact_voxels(int B, int max_voxels, int row_elems, const int32_t *voxel_count,
                                  const int32_t *dst_offsets, const float *voxels,
                                  const int32_t *coors, const int32_t *num_points, float *voxels_out,
                                  int32_t *coors_out, int32_t *num_points_out, void *stream) {
    MMT_REQUIRE_PTR(voxel_count);
    MMT_REQUIRE_PTR(dst_offsets);
    MMT_REQUIRE_PTR(voxels);
    MMT_REQUIRE_PTR(coors);
    MMT_REQUIRE_PTR(num_points);
    if (B <= 0 || B > 65535 || max_voxels <= 0 || row_elems <= 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "compact_voxels: bad sizes");
    // outputs may be NULL only when every sample is empty; the kernel never touches them then
    hipLaunchKernelGGL(compact_kernel, dim3(256, B), dim3(256), 0, (hipStream_t)stream, max_voxels,
                       row_elems, voxel_count, dst_offsets, voxels, coors, num_points, voxels_out,
                       coors_out, num_points_out);
    return mmt::check_launch("compact_voxels");
}

extern "C" int mmt_simple_vfe(int64_t M, int T, int F, int nf, const float *voxels,
                              const int32_t *num_points, float *out, void *stream) {
    if (M == 0) return MMT_OK;
    MMT_REQUIRE_PTR(voxels);
    MMT_REQUIRE_PTR(num_points);
    MMT_REQUIRE_PTR(out);
    if (M < 0 || T <= 0 || F <= 0 || nf <= 0 || nf > F)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "simple_vfe: bad sizes (M=%lld T=%d F=%d nf=%d)", (long long)M, T, F, nf);
    mmt::TimedSeq seq;
    seq.launch(true, simple_vfe_kernel, dim3(mmt::stream_grid(M * nf, 256)), dim3(256), 0,
               (hipStream_t)stream, M, T, F, nf, voxels, num_points, out);
    return mmt::check_launch("simple_vfe");
}

// Channels-last canvas [B, ny, nx, C] (what the channels_last BEV convolutions consume): a cell is one
// contiguous C-float row, so the scatter writes and the backward gathers whole rows -- one lane group of
// C/4 lanes per cell / voxel, 16 bytes per lane, no strided accesses (the NCHW backward gathers C values
// with a stride of ny*nx floats per voxel).
__global__ __launch_bounds__(256) void scatter_write_nhwc_kernel(int C4, int64_t cells, const float *feats,
                                                                 const int32_t *map, float *canvas) {
    // one lane group of C4 lanes per cell, kCells cells per group and trip: the map entries of a trip are
    // loaded first (unconditional), the rare owner rows (5 % of the cells) are fetched per lane
    constexpr int kCells = 4;
    const int lane_in = threadIdx.x % C4;                 // C4 divides 256 or the tail lanes idle
    const int groups_per_block = 256 / C4;
    const int grp = threadIdx.x / C4;
    if (grp >= groups_per_block) return;
    const int64_t ngroups = (int64_t)gridDim.x * groups_per_block;
    for (int64_t c0 = ((int64_t)blockIdx.x * groups_per_block + grp) * kCells; c0 < cells; c0 += ngroups * kCells) {
        int m[kCells];
#pragma unroll
        for (int u = 0; u < kCells; ++u) m[u] = map[(c0 + u) < cells ? (c0 + u) : (cells - 1)];
#pragma unroll
        for (int u = 0; u < kCells; ++u) {
            if (c0 + u >= cells) break;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m[u] >= 0) v = reinterpret_cast<const float4 *>(feats)[(int64_t)m[u] * C4 + lane_in];
            mmt_nt_store4(v, reinterpret_cast<float4 *>(canvas) + (c0 + u) * C4 + lane_in);
        }
    }
}

// fill of the cell -> row map (inside the timed kernel sequence, unlike a memset node)
__global__ __launch_bounds__(256) void fill_i32_kernel(int64_t n, int32_t value, int32_t *dst) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = value;
}

// Backward of the channels-last scatter: grad_feats[m,:] = grad_canvas[cell(m),:] for rows that own their cell.
// One lane group of C/4 lanes per voxel row, kRows rows in flight per group: the coors of the rows are loaded
// first (one int4 each, clamped index), then their map entries, then the gradient rows through a range-checked
// buffer descriptor (a row that owns nothing uses an out-of-range offset: zeros, no branch), so the three
// dependent loads of a row overlap with those of its neighbours instead of forming one serial chain per lane.
template <int kRows>
__global__ __launch_bounds__(256) void scatter_backward_nhwc_kernel(int64_t M, int C4, int B, int ny, int nx,
                                                                    const float *grad_canvas, const int32_t *coors,
                                                                    const int32_t *map, float *grad_feats,
                                                                    unsigned span_bytes) {
    const int gpb = 256 / C4;                       // lane groups per workgroup
    const int grp = threadIdx.x / C4, li = threadIdx.x - grp * C4;
    if (grp >= gpb) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(grad_canvas), 0, (int)span_bytes, 0x00020000);
    const int4 *co4 = reinterpret_cast<const int4 *>(coors);
    const int64_t step = (int64_t)gridDim.x * gpb * kRows;
    for (int64_t m0 = ((int64_t)blockIdx.x * gpb + grp) * kRows; m0 < M; m0 += step) {
        int4 co[kRows];
#pragma unroll
        for (int u = 0; u < kRows; ++u) co[u] = co4[(m0 + u) < M ? (m0 + u) : (M - 1)];
        int64_t cell[kRows];
        int own[kRows];
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
            const int b = co[u].x, y = co[u].z, x = co[u].w;
            const bool ok = !(b < 0 || b >= B || y < 0 || y >= ny || x < 0 || x >= nx);
            cell[u] = ok ? ((int64_t)b * ny + y) * nx + x : 0;
            own[u] = ok ? map[cell[u]] : -1;
        }
        mmt_u32x4 v[kRows];
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
            const bool mine = own[u] == (int)(m0 + u) && (m0 + u) < M;
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, mine ? (unsigned)((cell[u] * C4 + li) << 4) : 0xFFFFFFF0u, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < kRows; ++u)
            if (m0 + u < M) reinterpret_cast<mmt_u32x4 *>(grad_feats)[(m0 + u) * C4 + li] = v[u];
    }
}

// the same for a gradient of 4 GiB or more (beyond a buffer descriptor's range): plain loads
__global__ __launch_bounds__(256) void scatter_backward_nhwc_big_kernel(int64_t M, int C4, int B, int ny, int nx,
                                                                        const float *grad_canvas, const int32_t *coors,
                                                                        const int32_t *map, float *grad_feats) {
    const int64_t total = M * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / C4;
        const int c4 = (int)(i - m * C4);
        const int b = coors[m * 4], y = coors[m * 4 + 2], x = coors[m * 4 + 3];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(b < 0 || b >= B || y < 0 || y >= ny || x < 0 || x >= nx)) {
            const int64_t cell = ((int64_t)b * ny + y) * nx + x;
            if (map[cell] == (int)m) v = reinterpret_cast<const float4 *>(grad_canvas)[cell * C4 + c4];
        }
        reinterpret_cast<float4 *>(grad_feats)[i] = v;
    }
}

extern "C" int mmt_pillar_scatter(int64_t M, int C, int B, int ny, int nx, const float *feats,
                                  const int32_t *coors, float *canvas, int32_t *workspace,
                                  void *stream) {
    MMT_REQUIRE_PTR(canvas);
    MMT_REQUIRE_PTR(workspace);
    if (M > 0) { MMT_REQUIRE_PTR(feats); MMT_REQUIRE_PTR(coors); }
    if (M < 0 || C <= 0 || B <= 0 || ny <= 0 || nx <= 0 || M >= (1ll << 31))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "pillar_scatter: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int64_t HW = (int64_t)ny * nx;
    if (HW * B >= (1ll << 31)) return mmt::fail(MMT_ERR_TOO_LARGE, "pillar_scatter: B*ny*nx exceeds int32");
    mmt::TimedSeq seq;   // armed by mmt_arm_kernel_timing (bench only)
    seq.launch(false, fill_i32_kernel, dim3(mmt::stream_grid(B * HW, 256, 2048)), dim3(256), 0, st, B * HW, (int32_t)-1, workspace);
    if (M > 0)
        seq.launch(false, scatter_map_kernel, dim3(mmt::stream_grid(M, 256)), dim3(256), 0, st, M, B, ny, nx, 1, 1, coors, workspace);
    const bool vec4 = (HW % 4 == 0) && (((uintptr_t)canvas & 15) == 0) && (((uintptr_t)workspace & 15) == 0);
    const int64_t work = (int64_t)B * ((C + kChanBlock - 1) / kChanBlock) * (vec4 ? HW / 4 : HW);
    if (vec4) seq.launch(true, scatter_write_kernel<true>, dim3(mmt::stream_grid(work, 256, 256 * 32)), dim3(256), 0, st, C, B, (int)HW, feats, (const int32_t *)workspace, canvas);
    else seq.launch(true, scatter_write_kernel<false>, dim3(mmt::stream_grid(work, 256, 256 * 32)), dim3(256), 0, st, C, B, (int)HW, feats, (const int32_t *)workspace, canvas);
    return mmt::check_launch("pillar_scatter");
}

extern "C" int mmt_pillar_scatter_backward(int64_t M, int C, int B, int ny, int nx,
                                           const float *grad_canvas, const int32_t *coors,
                                           const int32_t *workspace, float *grad_feats, void *stream) {
    if (M == 0) return MMT_OK;
    MMT_REQUIRE_PTR(grad_canvas);
    MMT_REQUIRE_PTR(coors);
    MMT_REQUIRE_PTR(workspace);
    MMT_REQUIRE_PTR(grad_feats);
    if (M < 0 || C <= 0 || B <= 0 || ny <= 0 || nx <= 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "pillar_scatter_backward: bad sizes");
    mmt::TimedSeq seq;
    seq.launch(true, scatter_backward_kernel, dim3(mmt::stream_grid(M * C, 256)), dim3(256), 0,
               (hipStream_t)stream, M, C, B, ny, nx, grad_canvas, coors, workspace, grad_feats);
    return mmt::check_launch("pillar_scatter_backward");
}

extern "C" int mmt_pillar_scatter_nhwc(int64_t M, int C, int B, int ny, int nx, const float *feats,
                                       const int32_t *coors, float *canvas, int32_t *workspace, void *stream) {
    MMT_REQUIRE_PTR(canvas);
    MMT_REQUIRE_PTR(workspace);
    if (M > 0) { MMT_REQUIRE_PTR(feats); MMT_REQUIRE_PTR(coors); }
    if (M < 0 || C <= 0 || C % 4 || C > 1024 || B <= 0 || ny <= 0 || nx <= 0 || M >= (1ll << 31) || ((uintptr_t)canvas & 15) ||
        ((uintptr_t)feats & 15))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "pillar_scatter_nhwc: bad sizes (C %% 4 == 0, C <= 1024, 16-byte aligned buffers)");
    hipStream_t st = (hipStream_t)stream;
    const int64_t cells = (int64_t)B * ny * nx;
    if (cells >= (1ll << 31)) return mmt::fail(MMT_ERR_TOO_LARGE, "pillar_scatter_nhwc: B*ny*nx exceeds int32");
    mmt::TimedSeq seq;   // armed by mmt_arm_kernel_timing (bench only)
    seq.launch(false, fill_i32_kernel, dim3(mmt::stream_grid(cells, 256, 2048)), dim3(256), 0, st, cells, (int32_t)-1, workspace);
    if (M > 0)
        seq.launch(false, scatter_map_kernel, dim3(mmt::stream_grid(M, 256)), dim3(256), 0, st, M, B, ny, nx, 1, 1, coors, workspace);
    seq.launch(true, scatter_write_nhwc_kernel, dim3(mmt::stream_grid(mmt::ceil_div(cells, 4 * (256 / (C / 4))) * 256, 256, 256 * 32)), dim3(256), 0, st,
               C / 4, cells, feats, (const int32_t *)workspace, canvas);
    return mmt::check_launch("pillar_scatter_nhwc");
}

// Pillar scatter straight from the voxelizer's table (the model path: LidarEncoder.forward_bev).  The rows of the
// fixed-capacity layout own DISTINCT cells, and the voxelization has left a cell directory in the table (vox_own: occupancy
// bit, ordinal, head point; vox_emit: the head's voxel id), so the canvas pass needs neither the cell -> row map, nor its fill,
// nor the scatter_map kernel: cell -> bit -> ordinal -> head -> voxel id; a head that lost to the voxel cap reads as an empty cell.
struct DirView {
    const int32_t *header;
    const unsigned long long *bits;
    const unsigned short *pref;
    const int32_t *head_of;
    const int32_t *vidp;
    int R, shift, NR;
};

inline DirView dir_view(const int32_t *table, int B, int64_t cells) {
    const Directory d = directory_of(B, cells);
    DirView v;
    v.header = table;
    v.bits = reinterpret_cast<const unsigned long long *>(table + d.bits);
    v.pref = reinterpret_cast<const unsigned short *>(table + d.pref);
    v.head_of = table + d.head_of;
    v.vidp = table + d.vidp;
    v.R = d.R; v.shift = d.shift; v.NR = d.NR;
    return v;
}

// first hop: the cell's 64-cell block of the directory; second: its head (-1: nobody there); third: the head's voxel id
__device__ __forceinline__ int64_t dir_block(const DirView &d, int b, int cell) {
    return ((int64_t)b * d.NR + (cell >> d.shift)) * (d.R >> 6) + ((cell & (d.R - 1)) >> 6);
}
__device__ __forceinline__ int dir_head(const DirView &d, int b, int cell, unsigned long long bits, int pref) {
    const int k = cell & 63;
    if (!((bits >> k) & 1ull)) return -1;
    const int ord = pref + __popcll(bits & ((1ull << k) - 1ull));
    return d.head_of[((int64_t)b * d.NR + (cell >> d.shift)) * d.R + ord];
}

__global__ __launch_bounds__(256) void scatter_write_nhwc_table_kernel(int C4, int B, int64_t cells_per_sample, int V,
                                                                       const float *feats, DirView dir, float *canvas) {
    constexpr int kCells = 4;
    const int lane_in = threadIdx.x % C4;
    const int groups_per_block = 256 / C4;
    const int grp = threadIdx.x / C4;
    if (grp >= groups_per_block) return;
    // a table no voxelization of this form has written: pairing cells with rows from its contents would be SILENTLY wrong, so
    // the canvas is filled with NaN instead -- loud in the first loss that sees it
    const bool valid = dir.header[0] == kTableMagic;
    const int64_t cells = (int64_t)B * cells_per_sample;
    const int64_t ngroups = (int64_t)gridDim.x * groups_per_block;
    for (int64_t c0 = ((int64_t)blockIdx.x * groups_per_block + grp) * kCells; c0 < cells; c0 += ngroups * kCells) {
        unsigned long long bits[kCells];
        int pref[kCells], bb[kCells], cc[kCells], hd[kCells], vid[kCells];
#pragma unroll
        for (int u = 0; u < kCells; ++u) {
            const int64_t c = (c0 + u) < cells ? (c0 + u) : (cells - 1);
            bb[u] = (int)(c / cells_per_sample);
            cc[u] = (int)(c - (int64_t)bb[u] * cells_per_sample);
            const int64_t w = dir_block(dir, bb[u], cc[u]);
            bits[u] = dir.bits[w];
            pref[u] = dir.pref[w];
        }
#pragma unroll
        for (int u = 0; u < kCells; ++u) hd[u] = dir_head(dir, bb[u], cc[u], bits[u], pref[u]);
#pragma unroll
        for (int u = 0; u < kCells; ++u) vid[u] = hd[u] >= 0 ? dir.vidp[hd[u]] : -1;
#pragma unroll
        for (int u = 0; u < kCells; ++u) {
            if (c0 + u >= cells) break;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (vid[u] >= 0 && vid[u] < V) v = reinterpret_cast<const float4 *>(feats)[((int64_t)bb[u] * V + vid[u]) * C4 + lane_in];
            if (!valid) v = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
            mmt_nt_store4(v, reinterpret_cast<float4 *>(canvas) + (c0 + u) * C4 + lane_in);
        }
    }
}

// backward for rows that own distinct cells: grad_feats[m,:] = grad_canvas[cell(m),:] (0 for rows without a cell)
template <int kRows>
__global__ __launch_bounds__(256) void scatter_backward_nhwc_unique_kernel(int64_t M, int C4, int B, int ny, int nx,
                                                                           const float *grad_canvas, const int32_t *coors,
                                                                           float *grad_feats, unsigned span_bytes) {
    const int gpb = 256 / C4;
    const int grp = threadIdx.x / C4, li = threadIdx.x - grp * C4;
    if (grp >= gpb) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(grad_canvas), 0, (int)span_bytes, 0x00020000);
    const int4 *co4 = reinterpret_cast<const int4 *>(coors);
    const int64_t step = (int64_t)gridDim.x * gpb * kRows;
    for (int64_t m0 = ((int64_t)blockIdx.x * gpb + grp) * kRows; m0 < M; m0 += step) {
        int4 co[kRows];
#pragma unroll
        for (int u = 0; u < kRows; ++u) co[u] = co4[(m0 + u) < M ? (m0 + u) : (M - 1)];
        mmt_u32x4 v[kRows];
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
            const int b = co[u].x, y = co[u].z, x = co[u].w;
            const bool ok = !(b < 0 || b >= B || y < 0 || y >= ny || x < 0 || x >= nx) && (m0 + u) < M;
            const int64_t cell = ((int64_t)b * ny + y) * nx + x;
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? (unsigned)((cell * C4 + li) << 4) : 0xFFFFFFF0u, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < kRows; ++u)
            if (m0 + u < M) reinterpret_cast<mmt_u32x4 *>(grad_feats)[(m0 + u) * C4 + li] = v[u];
    }
}

extern "C" int mmt_pillar_scatter_nhwc_table(int C, int B, int ny, int nx, int max_voxels, const float *feats,
                                             const int32_t *table, float *canvas, void *stream) {
    MMT_REQUIRE_PTR(feats);
    MMT_REQUIRE_PTR(table);
    MMT_REQUIRE_PTR(canvas);
    if (C <= 0 || C % 4 || C > 1024 || B <= 0 || ny <= 0 || nx <= 0 || max_voxels <= 0 ||
        ((uintptr_t)canvas & 15) || ((uintptr_t)feats & 15) || ((uintptr_t)table & 7))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "pillar_scatter_nhwc_table: bad sizes (C %% 4 == 0, C <= 1024, aligned buffers)");
    const int64_t cells = (int64_t)B * ny * nx;
    if (cells >= (1ll << 31)) return mmt::fail(MMT_ERR_TOO_LARGE, "pillar_scatter_nhwc_table: B*ny*nx exceeds int32");
    mmt::TimedSeq seq;   // armed by mmt_arm_kernel_timing (bench only)
    seq.launch(true, scatter_write_nhwc_table_kernel, dim3(mmt::stream_grid(mmt::ceil_div(cells, 4 * (256 / (C / 4))) * 256, 256, 256 * 32)),
               dim3(256), 0, (hipStream_t)stream, C / 4, B, (int64_t)ny * nx, max_voxels, feats, dir_view(table, B, (int64_t)ny * nx), canvas);
    return mmt::check_launch("pillar_scatter_nhwc_table");
}

extern "C" int mmt_pillar_scatter_nhwc_unique_backward(int64_t M, int C, int B, int ny, int nx, const float *grad_canvas,
                                                       const int32_t *coors, float *grad_feats, void *stream) {
    if (M == 0) return MMT_OK;
    MMT_REQUIRE_PTR(grad_canvas);
    MMT_REQUIRE_PTR(coors);
    MMT_REQUIRE_PTR(grad_feats);
    const int64_t span = (int64_t)B * ny * nx * C * 4;
    if (M < 0 || C <= 0 || C % 4 || C > 1024 || B <= 0 || ny <= 0 || nx <= 0 || ((uintptr_t)grad_canvas & 15) || ((uintptr_t)grad_feats & 15) ||
        ((uintptr_t)coors & 15) || span >= (1ll << 32) - 16)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "pillar_scatter_nhwc_unique_backward: bad sizes (C %% 4 == 0, C <= 1024, 16-byte aligned buffers, canvas < 4 GiB)");
    constexpr int kRows = 4;
    const int C4 = C / 4, gpb = 256 / C4;
    mmt::TimedSeq seq;
    seq.launch(true, scatter_backward_nhwc_unique_kernel<kRows>, dim3(mmt::stream_grid(mmt::ceil_div(M, (int64_t)gpb * kRows) * 256, 256, 256 * 16)),
               dim3(256), 0, (hipStream_t)stream, M, C4, B, ny, nx, grad_canvas, coors, grad_feats, (unsigned)span);
    return mmt::check_launch("pillar_scatter_nhwc_unique_backward");
}

extern "C" int mmt_pillar_scatter_nhwc_backward(int64_t M, int C, int B, int ny, int nx, const float *grad_canvas,
                                                const int32_t *coors, const int32_t *workspace, float *grad_feats,
                                                void *stream) {
    if (M == 0) return MMT_OK;
    MMT_REQUIRE_PTR(grad_canvas);
    MMT_REQUIRE_PTR(coors);
    MMT_REQUIRE_PTR(workspace);
    MMT_REQUIRE_PTR(grad_feats);
    if (M < 0 || C <= 0 || C % 4 || C > 1024 || B <= 0 || ny <= 0 || nx <= 0 || ((uintptr_t)grad_canvas & 15) || ((uintptr_t)grad_feats & 15) ||
        ((uintptr_t)coors & 15))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "pillar_scatter_nhwc_backward: bad sizes (C %% 4 == 0, C <= 1024, 16-byte aligned buffers)");
    const int C4 = C / 4;
    const int64_t span = (int64_t)B * ny * nx * C * 4;
    mmt::TimedSeq seq;
    if (span < (1ll << 32) - 16) {
        constexpr int kRows = 4;
        const int gpb = 256 / C4;
        const int grid = mmt::stream_grid(mmt::ceil_div(M, (int64_t)gpb * kRows) * 256, 256, 256 * 16);
        seq.launch(true, scatter_backward_nhwc_kernel<kRows>, dim3(grid), dim3(256), 0, (hipStream_t)stream, M, C4, B, ny, nx,
                   grad_canvas, coors, workspace, grad_feats, (unsigned)span);
    } else {
        seq.launch(true, scatter_backward_nhwc_big_kernel, dim3(mmt::stream_grid(M * C4, 256)), dim3(256), 0,
                   (hipStream_t)stream, M, C4, B, ny, nx, grad_canvas, coors, workspace, grad_feats);
    }
    return mmt::check_launch("pillar_scatter_nhwc_backward");
}

// ---------------------------------------------------------------------------------------------------------------------
// Pillar scatter AT THE RESOLUTION THE FUSION LAYER CONSUMES (round 4).  models/bev_depth.py:188-190 nearest-resizes the
// pillar canvas onto the camera BEV grid before the channel concat (:189): with an integer ratio (sy, sx) -- 512 x 512
// pillars of 0.2 m onto 128 x 128 cells of 0.8 m: 4 x 4 -- torch's 'nearest' samples canvas cell (i * sy, j * sx) for output
// cell (i, j) and nothing else, so 15 / 16 of the canvas (268 MB at BASELINE configs[3]) was written, read once and thrown
// away, and its gradient (another 268 MB, almost all zeros) was written by upsample_nearest2d_backward just to be gathered
// from.  These entry points scatter only the sampled cells, straight into the camera|LiDAR concat buffer (rows of
// `out_row_stride` floats, `out` already advanced to the LiDAR channel offset), and the backward gathers
// grad[b, y / sy, x / sx, :] for the voxels on sampled cells (zeros for the others).  Bit-identical to the full-resolution
// scatter followed by [..., ::sy, ::sx] (tests/test_lidar_strided_gpu.py).
namespace {

struct StridedDims {
    int C4, B, ny, nx, sy, sx, oh, ow;
    int64_t row_stride4;           // float4 units between consecutive output cells' rows
};

// table form: one lane group of C4 lanes per OUTPUT cell, kCells cells per group and trip (directory hops first, then rows)
__global__ __launch_bounds__(256) void scatter_write_strided_table_kernel(StridedDims d, int V, const float *feats, DirView dir, float *out) {
    constexpr int kCells = 4;
    const int lane_in = threadIdx.x % d.C4;
    const int groups_per_block = 256 / d.C4;
    const int grp = threadIdx.x / d.C4;
    if (grp >= groups_per_block) return;
    const bool valid = dir.header[0] == kTableMagic;     // see scatter_write_nhwc_table_kernel
    const int64_t per_b = (int64_t)d.oh * d.ow, cells = (int64_t)d.B * per_b;
    const int64_t ngroups = (int64_t)gridDim.x * groups_per_block;
    for (int64_t c0 = ((int64_t)blockIdx.x * groups_per_block + grp) * kCells; c0 < cells; c0 += ngroups * kCells) {
        unsigned long long bits[kCells];
        int pref[kCells], bb[kCells], cc[kCells], hd[kCells], vid[kCells];
#pragma unroll
        for (int u = 0; u < kCells; ++u) {
            const int64_t oc = (c0 + u) < cells ? (c0 + u) : (cells - 1);
            const int b = (int)(oc / per_b);
            const int r = (int)(oc - (int64_t)b * per_b);
            const int i = r / d.ow, j = r - i * d.ow;
            bb[u] = b;
            cc[u] = (i * d.sy) * d.nx + j * d.sx;
            const int64_t w = dir_block(dir, b, cc[u]);
            bits[u] = dir.bits[w];
            pref[u] = dir.pref[w];
        }
#pragma unroll
        for (int u = 0; u < kCells; ++u) hd[u] = dir_head(dir, bb[u], cc[u], bits[u], pref[u]);
#pragma unroll
        for (int u = 0; u < kCells; ++u) vid[u] = hd[u] >= 0 ? dir.vidp[hd[u]] : -1;
#pragma unroll
        for (int u = 0; u < kCells; ++u) {
            if (c0 + u >= cells) break;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (vid[u] >= 0 && vid[u] < V) v = reinterpret_cast<const float4 *>(feats)[((int64_t)bb[u] * V + vid[u]) * d.C4 + lane_in];
            if (!valid) v = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
            reinterpret_cast<float4 *>(out)[(c0 + u) * d.row_stride4 + lane_in] = v;
        }
    }
}

// map form: the cell -> row map is already at the output resolution (scatter_map_kernel with strides)
__global__ __launch_bounds__(256) void scatter_write_strided_kernel(int C4, int64_t cells, int64_t row_stride4, const float *feats,
                                                                    const int32_t *map, float *out) {
    constexpr int kCells = 4;
    const int lane_in = threadIdx.x % C4;
    const int groups_per_block = 256 / C4;
    const int grp = threadIdx.x / C4;
    if (grp >= groups_per_block) return;
    const int64_t ngroups = (int64_t)gridDim.x * groups_per_block;
    for (int64_t c0 = ((int64_t)blockIdx.x * groups_per_block + grp) * kCells; c0 < cells; c0 += ngroups * kCells) {
        int m[kCells];
#pragma unroll
        for (int u = 0; u < kCells; ++u) m[u] = map[(c0 + u) < cells ? (c0 + u) : (cells - 1)];
#pragma unroll
        for (int u = 0; u < kCells; ++u) {
            if (c0 + u >= cells) break;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m[u] >= 0) v = reinterpret_cast<const float4 *>(feats)[(int64_t)m[u] * C4 + lane_in];
            reinterpret_cast<float4 *>(out)[(c0 + u) * row_stride4 + lane_in] = v;
        }
    }
}

// backward of both forms: grad_feats[m,:] = grad[b, y / sy, x / sx, :] for a row on a sampled cell (that owns it: `map`
// non-NULL = the map form's last-writer rule; NULL = rows own distinct cells), zeros otherwise.  Same structure as
// scatter_backward_nhwc_unique_kernel: coors of kRows rows first, then the gradient rows through a range-checked buffer
// descriptor (a row that takes nothing uses an out-of-range offset -> zeros without a memory access, no branch).
template <int kRows>
__global__ __launch_bounds__(256) void scatter_backward_strided_kernel(int64_t M, StridedDims d, const float *grad, const int32_t *coors,
                                                                       const int32_t *map, float *grad_feats, unsigned span_bytes) {
    const int gpb = 256 / d.C4;
    const int grp = threadIdx.x / d.C4, li = threadIdx.x - grp * d.C4;
    if (grp >= gpb) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(grad), 0, (int)span_bytes, 0x00020000);
    const int4 *co4 = reinterpret_cast<const int4 *>(coors);
    const int64_t step = (int64_t)gridDim.x * gpb * kRows;
    for (int64_t m0 = ((int64_t)blockIdx.x * gpb + grp) * kRows; m0 < M; m0 += step) {
        int4 co[kRows];
#pragma unroll
        for (int u = 0; u < kRows; ++u) co[u] = co4[(m0 + u) < M ? (m0 + u) : (M - 1)];
        int64_t ocell[kRows];
        bool ok[kRows];
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
            const int b = co[u].x, y = co[u].z, x = co[u].w;
            ok[u] = !(b < 0 || b >= d.B || y < 0 || y >= d.ny || x < 0 || x >= d.nx) && (m0 + u) < M && (y % d.sy == 0) && (x % d.sx == 0);
            ocell[u] = ok[u] ? ((int64_t)b * d.oh + y / d.sy) * d.ow + x / d.sx : 0;
        }
        if (map != nullptr) {
            int own[kRows];
#pragma unroll
            for (int u = 0; u < kRows; ++u) own[u] = map[ocell[u]];
#pragma unroll
            for (int u = 0; u < kRows; ++u) ok[u] = ok[u] && own[u] == (int)(m0 + u);
        }
        mmt_u32x4 v[kRows];
#pragma unroll
        for (int u = 0; u < kRows; ++u)
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok[u] ? (unsigned)((ocell[u] * d.row_stride4 + li) << 4) : 0xFFFFFFF0u, 0, 0);
#pragma unroll
        for (int u = 0; u < kRows; ++u)
            if (m0 + u < M) reinterpret_cast<mmt_u32x4 *>(grad_feats)[(m0 + u) * d.C4 + li] = v[u];
    }
}

int strided_check(const char *who, int C, int B, int ny, int nx, int sy, int sx, int64_t row_stride, const void *buf, StridedDims *d) {
    if (C <= 0 || C % 4 || C > 1024 || B <= 0 || ny <= 0 || nx <= 0 || sy <= 0 || sx <= 0 || ny % sy || nx % sx)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: bad sizes (C %% 4 == 0, C <= 1024, strides must divide the grid: C=%d B=%d %dx%d / %dx%d)",
                         who, C, B, ny, nx, sy, sx);
    if (row_stride < C || (row_stride & 3) || ((uintptr_t)buf & 15))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: the row stride must be >= C and a multiple of 4 floats, the buffer 16-byte aligned", who);
    if ((int64_t)B * ny * nx >= (1ll << 31)) return mmt::fail(MMT_ERR_TOO_LARGE, "%s: B*ny*nx exceeds int32", who);
    d->C4 = C / 4; d->B = B; d->ny = ny; d->nx = nx; d->sy = sy; d->sx = sx; d->oh = ny / sy; d->ow = nx / sx;
    d->row_stride4 = row_stride / 4;
    return 0;
}

}  // namespace

extern "C" int mmt_pillar_scatter_nhwc_table_strided(int C, int B, int ny, int nx, int max_voxels, int stride_y, int stride_x,
                                                     const float *feats, const int32_t *table, float *out,
                                                     int64_t out_row_stride, void *stream) {
    MMT_REQUIRE_PTR(feats);
    MMT_REQUIRE_PTR(table);
    MMT_REQUIRE_PTR(out);
    StridedDims d;
    if (int rc = strided_check("pillar_scatter_nhwc_table_strided", C, B, ny, nx, stride_y, stride_x, out_row_stride, out, &d)) return rc;
    if (max_voxels <= 0 || ((uintptr_t)feats & 15) || ((uintptr_t)table & 7))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "pillar_scatter_nhwc_table_strided: max_voxels > 0, aligned feats / table");
    const int64_t cells = (int64_t)B * d.oh * d.ow;
    mmt::TimedSeq seq;
    seq.launch(true, scatter_write_strided_table_kernel, dim3(mmt::stream_grid(mmt::ceil_div(cells, 4 * (256 / d.C4)) * 256, 256, 256 * 32)),
               dim3(256), 0, (hipStream_t)stream, d, max_voxels, feats, dir_view(table, B, (int64_t)ny * nx), out);
    return mmt::check_launch("pillar_scatter_nhwc_table_strided");
}

extern "C" int mmt_pillar_scatter_nhwc_strided(int64_t M, int C, int B, int ny, int nx, int stride_y, int stride_x, const float *feats,
                                               const int32_t *coors, float *out, int64_t out_row_stride, int32_t *workspace,
                                               void *stream) {
    MMT_REQUIRE_PTR(out);
    MMT_REQUIRE_PTR(workspace);
    if (M > 0) { MMT_REQUIRE_PTR(feats); MMT_REQUIRE_PTR(coors); }
    StridedDims d;
    if (int rc = strided_check("pillar_scatter_nhwc_strided", C, B, ny, nx, stride_y, stride_x, out_row_stride, out, &d)) return rc;
    if (M < 0 || M >= (1ll << 31) || ((uintptr_t)feats & 15)) return mmt::fail(MMT_ERR_BAD_SHAPE, "pillar_scatter_nhwc_strided: bad M / unaligned feats");
    hipStream_t st = (hipStream_t)stream;
    const int64_t cells = (int64_t)B * d.oh * d.ow;
    mmt::TimedSeq seq;
    seq.launch(false, fill_i32_kernel, dim3(mmt::stream_grid(cells, 256, 2048)), dim3(256), 0, st, cells, (int32_t)-1, workspace);
    if (M > 0)
        seq.launch(false, scatter_map_kernel, dim3(mmt::stream_grid(M, 256)), dim3(256), 0, st, M, B, ny, nx, stride_y, stride_x, coors, workspace);
    seq.launch(true, scatter_write_strided_kernel, dim3(mmt::stream_grid(mmt::ceil_div(cells, 4 * (256 / d.C4)) * 256, 256, 256 * 32)), dim3(256), 0, st,
               d.C4, cells, d.row_stride4, feats, (const int32_t *)workspace, out);
    return mmt::check_launch("pillar_scatter_nhwc_strided");
}

extern "C" int mmt_pillar_scatter_nhwc_strided_backward(int64_t M, int C, int B, int ny, int nx, int stride_y, int stride_x,
                                                        const float *grad_out, int64_t grad_row_stride, const int32_t *coors,
                                                        const int32_t *workspace, float *grad_feats, void *stream) {
    if (M == 0) return MMT_OK;
    MMT_REQUIRE_PTR(grad_out);
    MMT_REQUIRE_PTR(coors);
    MMT_REQUIRE_PTR(grad_feats);
    StridedDims d;
    if (int rc = strided_check("pillar_scatter_nhwc_strided_backward", C, B, ny, nx, stride_y, stride_x, grad_row_stride, grad_out, &d)) return rc;
    // bytes from grad_out to the end of the last sampled cell's C-float slice (the buffer descriptor's range)
    const int64_t span = (((int64_t)B * d.oh * d.ow - 1) * grad_row_stride + C) * 4;
    if (M < 0 || ((uintptr_t)grad_feats & 15) || ((uintptr_t)coors & 15) || span >= (1ll << 32) - 16)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "pillar_scatter_nhwc_strided_backward: bad M, unaligned buffers or a gradient of 4 GiB or more");
    constexpr int kRows = 4;
    const int gpb = 256 / d.C4;
    mmt::TimedSeq seq;
    seq.launch(true, scatter_backward_strided_kernel<kRows>, dim3(mmt::stream_grid(mmt::ceil_div(M, (int64_t)gpb * kRows) * 256, 256, 256 * 16)),
               dim3(256), 0, (hipStream_t)stream, M, d, grad_out, coors, workspace, grad_feats, (unsigned)span);
    return mmt::check_launch("pillar_scatter_nhwc_strided_backward");
}
