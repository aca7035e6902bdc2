// LiDAR / radar half of the hot path for MI355X (gfx950, wave64):
// hard voxelization (deterministic, first-point order), per-voxel mean, pillar scatter.
//
// Replaces the third-party ops the reference reaches at models/bev_depth.py:181-183
// (mmcv-full 1.7.0 ops.Voxelization, mmdet3d 1.0.0rc4 HardSimpleVFE and
// PointPillarsScatter).  mmcv's deterministic GPU path compares every point with
// every earlier point (O(N^2)) and numbers voxels in a single thread; here three
// kernels, no memset, ONE scattered atomic per point (the scattered device-scope
// atomic -- ~20 G/s chip-wide, MI355X_MICROARCH.md "Global float atomics" -- is what
// bounds the first kernel, so there is exactly one):
//   1. link  : per point, cell id; one 64-bit atomicExch on the cell's entry of a dense
//              per-sample table threads the point onto its cell's chain.  Entries carry a
//              GENERATION stamp in their upper 40 bits: an entry of an older call reads as
//              "empty", so the table is never cleared (the round-1 version spent three
//              memsets of dense per-cell tables, 34 MB, on 6 MB of points);
//   2. heads : per 256-point tile, "is this point the first of its cell" by a chain walk
//              that stops at the first earlier point; wave ballot + popcount -> rank of the
//              head inside its tile and the tile's head count;
//   3. emit  : per tile, offset = sum of the preceding tiles' head counts; voxel id of a
//              head = offset + rank  ==> voxels numbered in order of their first point,
//              exactly like the sequential algorithm; heads past max_voxels are dropped.
//              Each head walks its cell's chain once, keeps the max_points smallest point
//              indices sorted (LDS) and the tile then writes its voxels' rows -- a
//              CONTIGUOUS block of the output, voxel ids of a tile are consecutive -- the
//              zero padding, coors, num_points and the HardSimpleVFE mean (summed in slot
//              order) as flat coalesced stores.  Rows past a sample's voxel count are
//              marked empty (coors -1, num_points 0, mean 0) by the same kernel.
// All samples of the batch go through each kernel together (flattened (sample, tile) space).
// Round 3: a point whose cell holds nothing else (the table entry is the point itself and its link is empty -- the common
// case: 93 % of the occupied 0.2 m cells of a 40 k-point cloud) is recognised in step 2 from ONE scattered read and flagged,
// so step 3 emits it without touching the table or the chain links again (the two steps used to walk every chain twice).
// A variant that merged steps 2 and 3 into one kernel (tile counts published in status words, decoupled look-back over
// the earlier tiles) was built and measured: 38 us against 28 us at 4 x 40 k points -- every tile then pays several
// round trips of uncached polls on its critical path -- and it hung once under rocprofv3; it was dropped.
#include "mmt_common.h"

namespace {

constexpr int kTile = 256;            // points per tile / threads per workgroup (4 waves)
constexpr int kTileWaves = kTile / 64;
constexpr int kIdxBits = 24;          // point index inside its sample (host checks N < 2^24)
constexpr unsigned long long kIdxMask = (1ull << kIdxBits) - 1ull;
constexpr unsigned long long kOwnedBit = 1ull << (kIdxBits - 1);   // vox_emit: the entry now holds the cell's VOXEL ID (points < 2^23)
constexpr int kMaxBatchLds = 255;     // sample offsets cached in LDS up to this batch size
constexpr unsigned kMarkBit = 1u << 30;     // vox_link -> vox_heads, in a point's hrank word: a later point of this generation landed in its cell
constexpr int kSingleBit = 1 << 20;   // hrank: the head's chain is the head alone (vox_emit then needs neither the table nor the links)

struct VoxArgs {
    int F, max_points, max_voxels, nf;
    int gx, gy, gz;
    float vs[3], rmin[3];
    const float *points;
    const int32_t *offsets;        // [B+1]
    unsigned long long *table;     // [2 + B*cells]: [0] generation counter, [1] pad, then one entry per cell
    int32_t *cell_of_point;        // [N]
    int32_t *next;                 // [N] chain link (point index inside the sample, -1 = end)
    int32_t *hrank;                // [N] rank of a head point among the heads of its tile, -1 = not a head
    int32_t *tile_counts;          // [B*ntiles]
    int32_t *gen_word;             // [1] the call's generation (low 30 bits) as vox_link used it, for vox_heads (table[0] changes under its feet)
    int ntiles;
    float *voxels;                 // may be NULL (only the mean is wanted)
    int32_t *coors;
    int32_t *num_points;
    int32_t *voxel_count;
    float *mean;                   // may be NULL
    int mark_owned;                // vox_emit leaves (generation | owned | voxel id) in the entries of the cells it emitted
};

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

__global__ __launch_bounds__(kTile) void vox_link(VoxArgs a, int B, int total) {
    __shared__ int offs[kMaxBatchLds + 1];
    for (int i = threadIdx.x; i <= B && i <= kMaxBatchLds; i += kTile) offs[i] = a.offsets[i];
    __syncthreads();
    const int64_t cells = (int64_t)a.gx * a.gy * a.gz;
    // every workgroup reads the same value: the counter is advanced by the NEXT kernel of the call
    const unsigned long long gen = a.table[0] + 1ull;
    if (blockIdx.x == 0 && threadIdx.x == 0) a.gen_word[0] = (int)(kMarkBit | ((unsigned)gen & (kMarkBit - 1u)));
    for (int g = blockIdx.x * kTile + threadIdx.x; g < total; g += gridDim.x * kTile) {
        int b, beg;
        if (B <= kMaxBatchLds) { b = sample_of_point(offs, B, g); beg = offs[b]; }
        else { b = 0; while (b + 1 < B && g >= a.offsets[b + 1]) ++b; beg = a.offsets[b]; }
        const int i = g - beg;
        const float *p = a.points + (int64_t)g * a.F;
        const int cx = cell_coord(p[0], a.rmin[0], a.vs[0]);
        const int cy = cell_coord(p[1], a.rmin[1], a.vs[1]);
        const int cz = cell_coord(p[2], a.rmin[2], a.vs[2]);
        int cell = -1, nxt = -1;
        if (!(cx < 0 || cx >= a.gx || cy < 0 || cy >= a.gy || cz < 0 || cz >= a.gz)) {
            cell = (cz * a.gy + cy) * a.gx + cx;
            const unsigned long long old = atomicExch(&a.table[2 + (int64_t)b * cells + cell], (gen << kIdxBits) | (unsigned long long)i);
            if ((old >> kIdxBits) == gen) {
                nxt = (int)(old & kIdxMask);
                // tell the point that was here before that it has company: vox_heads then recognises a cell's ONLY point from
                // two coalesced words (its own link is empty and nobody marked it) without reading the table at all.  The mark
                // carries the generation, so the word (the point's hrank slot, any contents before) is never cleared; a stale
                // word that happens to equal the mark only sends a singleton down the general path.
                a.hrank[beg + nxt] = (int)(kMarkBit | ((unsigned)gen & (kMarkBit - 1u)));
            }
        }
        a.cell_of_point[g] = cell;
        a.next[g] = nxt;
    }
}

__global__ __launch_bounds__(kTile) void vox_heads(VoxArgs a, int B) {
    __shared__ int wc[kTileWaves];
    if (blockIdx.x == 0 && threadIdx.x == 0) {                        // vox_link of this call is done
        a.table[0] += 1ull;
        a.table[1] = a.mark_owned ? (1ull << 32) : 0ull;             // header word 1, bit 32: vox_emit leaves voxel ids in this generation's entries
    }
    int b, tile;
    if (!locate_tile(a.offsets, B, blockIdx.x, &b, &tile)) return;
    const int beg = a.offsets[b], n = a.offsets[b + 1] - beg;
    if (tile * kTile >= n) {
        if (threadIdx.x == 0) a.tile_counts[b * a.ntiles + tile] = 0;
        return;
    }
    const int64_t cells = (int64_t)a.gx * a.gy * a.gz;
    const unsigned long long *tab = a.table + 2 + (int64_t)b * cells;
    const int i = tile * kTile + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool h = false, single = false;
    if (i < n) {
        const int cell = a.cell_of_point[beg + i];
        const int own_next = a.next[beg + i];                // coalesced; spares the scattered link read of a point's own entry
        const int mark = a.hrank[beg + i];                   // coalesced: vox_link's "a later point joined your cell" (or anything else)
        if (cell >= 0) {
            const int joined = a.gen_word[0];             // (written by vox_link: the mark of this call)
            if (own_next < 0 && mark != joined) {
                // first into its cell and nobody after it: the cell's only point (87 % of the points of a 40 k cloud on a
                // 0.2 m grid) -- a head, decided without touching the table or the links
                h = true; single = true;
            } else {
                // head <=> no point of the chain comes earlier in the cloud (the chain holds every point of the cell)
                h = true;
                for (int j = (int)(tab[cell] & kIdxMask); j >= 0; j = (j == i) ? own_next : a.next[beg + j])
                    if (j < i) { h = false; break; }
            }
        }
    }
    const unsigned long long m = __ballot(h);
    if (lane == 0) wc[wave] = __popcll(m);
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kTileWaves; ++w) {
        const int c = wc[w];
        if (w < wave) woff += c;
        total += c;
    }
    // rank of a head inside its tile (< 256), bit kSingleBit: its cell holds no other point
    if (i < n) a.hrank[beg + i] = h ? ((woff + __popcll(m & ((1ull << lane) - 1ull))) | (single ? kSingleBit : 0)) : -1;
    if (threadIdx.x == 0) a.tile_counts[b * a.ntiles + tile] = total;
}

// LDS: lists [kTile][T] sorted point indices | cnt [kTile] | cellv [kTile].  FT = compile-time F (0 = any)
template <int FT>
__global__ __launch_bounds__(kTile) void vox_emit(VoxArgs a, int B) {
    extern __shared__ __align__(16) int lds[];
    __shared__ int s_off, s_total, s_mine;
    const int T = a.max_points, F = FT > 0 ? FT : a.F, V = a.max_voxels;
    const int TF = T * F;
    int *lists = lds;
    int *cnt = lists + kTile * T;
    int *cellv = cnt + kTile;
    int b, tile;
    if (!locate_tile(a.offsets, B, blockIdx.x, &b, &tile)) return;
    const int beg = a.offsets[b], n = a.offsets[b + 1] - beg;
    const int my_tiles = (n + kTile - 1) / kTile;
    const int64_t cells = (int64_t)a.gx * a.gy * a.gz;
    const unsigned long long *tab = a.table + 2 + (int64_t)b * cells;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // this point's head rank and cell: requested BEFORE the prefix over the tile counts (they do not depend on it; behind the
    // barrier they were one more round trip on the kernel's critical path)
    const int i = tile * kTile + threadIdx.x;
    int hr = -1, my_cell = -1;
    if (i < n) { hr = a.hrank[beg + i]; my_cell = a.cell_of_point[beg + i]; }
    // heads before this tile / in the whole sample (a few hundred tile counts at most)
    if (wave == 0) {
        int before = 0, all = 0, mine = 0;
        for (int t = lane; t < my_tiles; t += 64) {
            const int c = a.tile_counts[b * a.ntiles + t];
            all += c;
            if (t < tile) before += c;
            if (t == tile) mine = c;
        }
        for (int o = 32; o > 0; o >>= 1) {
            before += __shfl_down(before, o);
            all += __shfl_down(all, o);
            mine += __shfl_down(mine, o);
        }
        if (lane == 0) { s_off = before; s_total = all; s_mine = mine; }
    }
    __syncthreads();
    const int off = s_off;
    const int M = s_total < V ? s_total : V;            // voxels of this sample
    if (tile == 0 && threadIdx.x == 0) a.voxel_count[b] = M;
    const int nheads = s_mine;
    const int nown = (off + nheads <= V) ? nheads : (V - off > 0 ? V - off : 0);   // heads of this tile below the cap

    // ---- every owning head walks its chain once: the T smallest point indices, sorted, in LDS
    if (i < n) {
        const int r = hr < 0 ? -1 : (hr & (kSingleBit - 1));
        if (r >= 0 && r < nown) {
            const int cell = my_cell;
            int *L = lists + r * T;
            int c = 0;
            if (hr & kSingleBit) {                          // the whole chain, known since vox_heads
                L[0] = i;
                c = 1;
            } else {
                for (int j = (int)(tab[cell] & kIdxMask); j >= 0; j = a.next[beg + j]) {
                    if (c == T && j > L[T - 1]) continue;
                    int k = c < T ? c : T - 1;              // insertion position search from the top
                    while (k > 0 && L[k - 1] > j) { L[k] = L[k - 1]; --k; }
                    L[k] = j;
                    if (c < T) ++c;
                }
            }
            cnt[r] = c;
            cellv[r] = cell;
            // the chain has been read: the cell's entry now names its voxel, (generation | owned | voxel id), for the pillar
            // scatter that follows (mmt_pillar_scatter_nhwc_table reads the table instead of building a cell -> row map)
            if (a.mark_owned) a.table[2 + (int64_t)b * cells + cell] = (a.table[0] << kIdxBits) | kOwnedBit | (unsigned long long)(off + r);
        }
    }
    __syncthreads();

    // ---- the tile's voxels are rows [off, off + nown) of the sample: flat, coalesced stores.  (row, element) of a
    // flat index advance incrementally (no integer division per element: (slot, column) of an element come from qmap)
    const int64_t row0 = (int64_t)b * V + off;
    if (a.voxels) {
        // one (voxel, slot) per thread and trip: the point's F floats are requested together and stored as one
        // contiguous F*4-byte piece (the slots of a tile's voxels are one contiguous block of the output)
        float *dst = a.voxels + row0 * TF;
        const int total = nown * T;
        int r = threadIdx.x / T, t = threadIdx.x - r * T;
        const int dr = kTile / T, dt = kTile - dr * T;
        for (int e = threadIdx.x; e < total; e += kTile) {
            const bool live = t < cnt[r];
            const float *src = a.points + (int64_t)(beg + (live ? lists[r * T + t] : 0)) * F;
            float *d = dst + (int64_t)e * F;
            if (FT > 0) {
                float v[FT > 0 ? FT : 1];
#pragma unroll
                for (int f = 0; f < FT; ++f) v[f] = live ? src[f] : 0.f;
#pragma unroll
                for (int f = 0; f < FT; ++f) d[f] = v[f];
            } else {
                for (int f = 0; f < F; ++f) d[f] = live ? src[f] : 0.f;
            }
            r += dr; t += dt;
            if (t >= T) { t -= T; ++r; }
        }
    }
    if (a.mean) {
        const int nf = a.nf;
        float *dst = a.mean + row0 * nf;
        const int total = nown * nf;
        int r = threadIdx.x / nf, k = threadIdx.x - r * nf;
        const int dr = kTile / nf, dk = kTile - dr * nf;
        for (int e = threadIdx.x; e < total; e += kTile) {
            const int c = cnt[r];
            const int *L = lists + r * T;
            float sum = 0.f;
            for (int t = 0; t < c; ++t) sum = __fadd_rn(sum, a.points[(int64_t)(beg + L[t]) * F + k]);
            dst[e] = __fdiv_rn(sum, (float)c);          // zero-padded slots add nothing; c >= 1
            r += dr; k += dk;
            if (k >= nf) { k -= nf; ++r; }
        }
    }
    for (int e = threadIdx.x; e < nown * 4; e += kTile) {
        const int r = e >> 2, k = e & 3;
        const int cell = cellv[r];
        int v = b;
        if (k == 1) v = cell / (a.gx * a.gy);
        else if (k == 2) v = (cell / a.gx) % a.gy;
        else if (k == 3) v = cell % a.gx;
        a.coors[row0 * 4 + e] = v;
    }
    for (int e = threadIdx.x; e < nown; e += kTile) a.num_points[row0 + e] = cnt[e];

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

int64_t vox_tiles(int64_t N) { return mmt::ceil_div(N > 0 ? N : 1, kTile); }

int vox_check(const char *what, int B, int64_t N, int F, const int32_t *grid_host, int max_points, int max_voxels, int nf) {
    if (B <= 0 || B > 65535 || N < 0 || F < 3 || max_points <= 0 || max_voxels <= 0 || nf < 0 || nf > F)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: bad sizes (B=%d N=%lld F=%d T=%d max_voxels=%d nf=%d)", what, B, (long long)N, F, max_points, max_voxels, nf);
    if (grid_host[0] <= 0 || grid_host[1] <= 0 || grid_host[2] <= 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: non-positive grid", what);
    const int64_t cells = (int64_t)grid_host[0] * grid_host[1] * grid_host[2];
    if (cells * B >= (1ll << 31) || cells >= (1ll << 31) || N >= (1ll << kIdxBits))
        return mmt::fail(MMT_ERR_TOO_LARGE, "%s: B*cells exceeds int32 or N >= 2^24 points", what);
    if ((size_t)(kTile * (int64_t)max_points + 2 * kTile) * 4 > 150 * 1024)
        return mmt::fail(MMT_ERR_TOO_LARGE, "%s: max_points=%d too large for the LDS index lists", what, max_points);
    return 0;
}

// the three kernels of one voxelization (table: 8-byte aligned, generation-stamped; scratch: any contents)
int vox_run(const char *what, int B, int64_t N, int F, const float *points, const int32_t *point_offsets,
            const float *voxel_size_host, const float *range_min_host, const int32_t *grid_host, int max_points,
            int max_voxels, int nf, float *voxels, int32_t *coors, int32_t *num_points, int32_t *voxel_count,
            float *mean, unsigned long long *table, int32_t *scratch, hipStream_t st) {
    VoxArgs a;
    a.F = F; a.max_points = max_points; a.max_voxels = max_voxels; a.nf = nf;
    a.gx = grid_host[0]; a.gy = grid_host[1]; a.gz = grid_host[2];
    for (int k = 0; k < 3; ++k) { a.vs[k] = voxel_size_host[k]; a.rmin[k] = range_min_host[k]; }
    a.points = points; a.offsets = point_offsets;
    a.ntiles = (int)vox_tiles(N);
    a.table = table;
    a.cell_of_point = scratch;
    a.next = a.cell_of_point + N;
    a.hrank = a.next + N;
    a.tile_counts = a.hrank + N;
    a.gen_word = a.tile_counts + (int64_t)B * a.ntiles;          // (mmt_voxelize_scratch_elems leaves 16 spare words)
    a.voxels = voxels; a.coors = coors; a.num_points = num_points; a.voxel_count = voxel_count; a.mean = mean;
    a.mark_owned = (N < (1ll << (kIdxBits - 1))) ? 1 : 0;
    mmt::TimedSeq seq;   // armed by mmt_arm_kernel_timing (bench only)
    const unsigned gpts = (unsigned)mmt::stream_grid(N > 0 ? N : 1, kTile, 4096);
    const unsigned gtiles = (unsigned)(vox_tiles(N) + B);       // flattened (sample, tile) space, see locate_tile
    seq.launch(false, vox_link, dim3(gpts), dim3(kTile), 0, st, a, B, (int)N);
    seq.launch(false, vox_heads, dim3(gtiles), dim3(kTile), 0, st, a, B);
    const size_t lds = (size_t)(kTile * (int64_t)max_points + 2 * kTile) * 4;
    if (F == 5) seq.launch(true, vox_emit<5>, dim3(gtiles), dim3(kTile), lds, st, a, B);
    else if (F == 8) seq.launch(true, vox_emit<8>, dim3(gtiles), dim3(kTile), lds, st, a, B);
    else if (F == 4) seq.launch(true, vox_emit<4>, dim3(gtiles), dim3(kTile), lds, st, a, B);
    else seq.launch(true, vox_emit<0>, dim3(gtiles), dim3(kTile), lds, st, a, B);
    return mmt::check_launch(what);
}

}  // namespace

extern "C" int64_t mmt_voxelize_table_elems(int B, const int32_t *grid) {
    if (B <= 0 || grid == nullptr) return 0;
    return 4 + 2 * (int64_t)B * grid[0] * grid[1] * grid[2];
}

extern "C" int64_t mmt_voxelize_scratch_elems(int B, int64_t total_points) {
    if (B <= 0 || total_points < 0) return 0;
    return 3 * total_points + (int64_t)B * vox_tiles(total_points) + 16;
}

extern "C" int64_t mmt_voxelize_workspace_elems(int B, int64_t total_points, const int32_t *grid) {
    if (B <= 0 || total_points < 0 || grid == nullptr) return 0;
    return mmt_voxelize_table_elems(B, grid) + mmt_voxelize_scratch_elems(B, total_points) + 2;
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
                   max_points, max_voxels, num_features, voxels, coors, num_points, voxel_count, mean,
                   reinterpret_cast<unsigned long long *>(table), scratch, (hipStream_t)stream);
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
    // stateless form: the table lives in the caller's scratch workspace (any contents), so it is cleared here;
    // mmt_hard_voxelize_mean with a persistent table skips this memset
    int32_t *tab = workspace + (((uintptr_t)workspace & 7) ? 1 : 0);
    const int64_t tab_elems = mmt_voxelize_table_elems(B, grid_host);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(tab, 0, sizeof(int32_t) * (size_t)tab_elems, st);
    if (e != hipSuccess) return mmt::fail((int)e, "hard_voxelize: hipMemsetAsync failed: %s", hipGetErrorString(e));
    return vox_run("hard_voxelize", B, N, F, points, point_offsets, voxel_size_host, range_min_host, grid_host,
                   max_points, max_voxels, 0, voxels, coors, num_points, voxel_count, nullptr,
                   reinterpret_cast<unsigned long long *>(tab), tab + tab_elems, st);
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
// fixed-capacity layout own DISTINCT cells, and vox_emit has left (generation | owned | voxel id) in the table entry of every
// cell it emitted, so the canvas pass needs neither the cell -> row map, nor its fill, nor the scatter_map kernel: an entry
// of another generation (or a chain head that lost to the voxel cap) reads as an empty cell.
__global__ __launch_bounds__(256) void scatter_write_nhwc_table_kernel(int C4, int B, int64_t cells_per_sample, int V,
                                                                       const float *feats, const unsigned long long *table,
                                                                       float *canvas) {
    constexpr int kCells = 4;
    const int lane_in = threadIdx.x % C4;
    const int groups_per_block = 256 / C4;
    const int grp = threadIdx.x / C4;
    if (grp >= groups_per_block) return;
    const unsigned long long gen = table[0];            // the voxelization that ran last on this table
    // header word 1, bit 32: that voxelization left voxel ids in its entries.  Without it (a cloud of 2^23 points or more: the
    // id field is too narrow) the entries still hold chain heads; reading those as voxel ids would pair cells with the wrong
    // rows SILENTLY, so the canvas is filled with NaN instead -- loud in the first loss that sees it.
    const bool ids_valid = (table[1] >> 32) & 1ull;
    const int64_t cells = (int64_t)B * cells_per_sample;
    const int64_t ngroups = (int64_t)gridDim.x * groups_per_block;
    for (int64_t c0 = ((int64_t)blockIdx.x * groups_per_block + grp) * kCells; c0 < cells; c0 += ngroups * kCells) {
        unsigned long long e[kCells];
#pragma unroll
        for (int u = 0; u < kCells; ++u) e[u] = table[2 + ((c0 + u) < cells ? (c0 + u) : (cells - 1))];
#pragma unroll
        for (int u = 0; u < kCells; ++u) {
            if (c0 + u >= cells) break;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((e[u] >> kIdxBits) == gen && (e[u] & kOwnedBit)) {
                const int64_t row = (c0 + u) / cells_per_sample * V + (int64_t)(e[u] & (kOwnedBit - 1));
                v = reinterpret_cast<const float4 *>(feats)[row * C4 + lane_in];
            }
            if (!ids_valid) v = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
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
    if (C <= 0 || C % 4 || C > 1024 || B <= 0 || ny <= 0 || nx <= 0 || max_voxels <= 0 || max_voxels > (1 << (kIdxBits - 1)) ||
        ((uintptr_t)canvas & 15) || ((uintptr_t)feats & 15) || ((uintptr_t)table & 7))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "pillar_scatter_nhwc_table: bad sizes (C %% 4 == 0, C <= 1024, aligned buffers, max_voxels <= 2^23)");
    const int64_t cells = (int64_t)B * ny * nx;
    if (cells >= (1ll << 31)) return mmt::fail(MMT_ERR_TOO_LARGE, "pillar_scatter_nhwc_table: B*ny*nx exceeds int32");
    mmt::TimedSeq seq;   // armed by mmt_arm_kernel_timing (bench only)
    seq.launch(true, scatter_write_nhwc_table_kernel, dim3(mmt::stream_grid(mmt::ceil_div(cells, 4 * (256 / (C / 4))) * 256, 256, 256 * 32)),
               dim3(256), 0, (hipStream_t)stream, C / 4, B, (int64_t)ny * nx, max_voxels, feats,
               reinterpret_cast<const unsigned long long *>(table), canvas);
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

// table form: one lane group of C4 lanes per OUTPUT cell, kCells cells per group and trip (entries first, then rows)
__global__ __launch_bounds__(256) void scatter_write_strided_table_kernel(StridedDims d, int V, const float *feats,
                                                                          const unsigned long long *table, float *out) {
    constexpr int kCells = 4;
    const int lane_in = threadIdx.x % d.C4;
    const int groups_per_block = 256 / d.C4;
    const int grp = threadIdx.x / d.C4;
    if (grp >= groups_per_block) return;
    const unsigned long long gen = table[0];
    const bool ids_valid = (table[1] >> 32) & 1ull;     // see scatter_write_nhwc_table_kernel
    const int64_t per_b = (int64_t)d.oh * d.ow, cells = (int64_t)d.B * per_b;
    const int64_t src_per_b = (int64_t)d.ny * d.nx;
    const int64_t ngroups = (int64_t)gridDim.x * groups_per_block;
    for (int64_t c0 = ((int64_t)blockIdx.x * groups_per_block + grp) * kCells; c0 < cells; c0 += ngroups * kCells) {
        unsigned long long e[kCells];
        int bb[kCells];
#pragma unroll
        for (int u = 0; u < kCells; ++u) {
            const int64_t oc = (c0 + u) < cells ? (c0 + u) : (cells - 1);
            const int b = (int)(oc / per_b);
            const int r = (int)(oc - (int64_t)b * per_b);
            const int i = r / d.ow, j = r - i * d.ow;
            bb[u] = b;
            e[u] = table[2 + (int64_t)b * src_per_b + (int64_t)(i * d.sy) * d.nx + j * d.sx];
        }
#pragma unroll
        for (int u = 0; u < kCells; ++u) {
            if (c0 + u >= cells) break;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((e[u] >> kIdxBits) == gen && (e[u] & kOwnedBit)) {
                const int64_t row = (int64_t)bb[u] * V + (int64_t)(e[u] & (kOwnedBit - 1));
                v = reinterpret_cast<const float4 *>(feats)[row * d.C4 + lane_in];
            }
            if (!ids_valid) v = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
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
    if (max_voxels <= 0 || max_voxels > (1 << (kIdxBits - 1)) || ((uintptr_t)feats & 15) || ((uintptr_t)table & 7))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "pillar_scatter_nhwc_table_strided: max_voxels <= 2^23, aligned feats / table");
    const int64_t cells = (int64_t)B * d.oh * d.ow;
    mmt::TimedSeq seq;
    seq.launch(true, scatter_write_strided_table_kernel, dim3(mmt::stream_grid(mmt::ceil_div(cells, 4 * (256 / d.C4)) * 256, 256, 256 * 32)),
               dim3(256), 0, (hipStream_t)stream, d, max_voxels, feats, reinterpret_cast<const unsigned long long *>(table), out);
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
