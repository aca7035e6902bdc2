// voxel_pooling forward / backward for MI355X (gfx950, wave64).
//
// Replaces ops/voxel_pooling/src/voxel_pooling_forward_cuda.cu:9-56 (forward kernel +
// launcher) and ops/voxel_pooling/voxel_pooling.py:58-69 (backward, pure ATen in the
// reference).  Not a translation of the CUDA kernel: the reference maps one THREAD to
// a point and loops over channels with strided loads + one contended fp32 atomic per
// channel.  The default forward (ALGO_AUTO = SEG_GATHER, vp_fwd_seg_gather) gives a
// workgroup a chunk of consecutive points, sorts the chunk by BEV cell in LDS (hash ->
// count -> wave scan -> counting sort), lets every lane group of a wave sum one cell's
// feature rows (whole 16-byte-per-lane rows, rows of dropped points never fetched) in
// fp32 REGISTERS, and sends one contiguous run of global fp32 atomics per (chunk, cell).
// The backward is a two-pass gather (row offsets + cache warm-up, then a software-
// pipelined, XCD-partitioned row gather through a range-checked buffer descriptor with
// non-temporal 16-byte stores).  Feature storage is a template argument: fp32 (the
// reference's type) or bf16 (SURVEY 5.6 / BASELINE configs[4]; fp32 accumulation).
// ROW_ATOMIC / LDS_ATOMIC / STREAM are earlier algorithms kept selectable (odd channel
// counts fall back to them) and as measured comparison points.
// This file also holds the first-generation fused lift-splat pair (mmt_lift_splat_*);
// the frustum-tile pair the model runs is in lift_splat_tile.hip.
//
// HBM traffic per call (algorithmic): 12*BP geom + 12*BP pos_memo + 4*C*K features
// of kept points + 4*C*B*ny*nx BEV rows.
#include <hip/hip_ext.h>

#include "mmt_common.h"

namespace {

constexpr int kBlock = 256;       // 4 waves
constexpr int kEmpty = -1;        // hash-table empty marker (cell keys are >= 0)
constexpr int kDropped = -1;      // pt_slot: point outside the grid
constexpr int kOverflow = -2;     // pt_slot: no LDS row available -> direct row atomics
constexpr int kHashSize = 512;    // entries, power of two
constexpr int kChunk = 512;       // points per chunk (2 per thread)
constexpr int kLongSlot = 24;     // seg_gather: cells with more points per chunk are walked by a whole wave

struct VpArgs {
    int64_t BP;  // B*P
    int P, C, nx, ny, nz;
    const int32_t *geom;
    const void *feats;     // fp32 or bf16 rows (the kernel's FT template argument says which)
    float *out;
    int32_t *pos_memo;
    int write_dropped;
    int nslot;    // LDS BEV rows per workgroup
    int nchunks;
    // fused lift-splat (feats == nullptr): row(t) = depth[t] * context[pix(t), :]
    const void *depth;     // [B*P] in point order (= [B*N, D, HW]), fp32 or bf16
    const void *context;   // [B*N, HW, C] channels-last, fp32 or bf16
    int DHW, HW;           // D*HW points per camera, HW pixels per camera
};

__device__ __forceinline__ bool in_grid(int x, int y, int z, int nx, int ny, int nz) {
    // voxel_pooling_forward_cuda.cu:24-26 (negated)
    return !(x < 0 || x >= nx || y < 0 || y >= ny || z < 0 || z >= nz);
}

__device__ __forceinline__ void write_pos(int32_t *pos_memo, int64_t t, int b, int y, int x) {
    // voxel_pooling_forward_cuda.cu:27-29
    pos_memo[t * 3] = b;
    pos_memo[t * 3 + 1] = y;
    pos_memo[t * 3 + 2] = x;
}

// ---------------------------------------------------------------------------
// ALGO_ROW_ATOMIC: every kept point adds its row to the BEV with global fp32
// atomics; lanes run along the channel axis so each wave instruction covers
// contiguous bytes (the shape the memory-side atomic units want).
template <int VEC>
__global__ __launch_bounds__(kBlock) void vp_fwd_row_atomic(VpArgs a) {
    const int CV = a.C / VEC;
    const int64_t total = a.BP * CV;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
        const int64_t t = i / CV;
        const int cv = (int)(i - t * CV);
        const int x = a.geom[t * 3], y = a.geom[t * 3 + 1], z = a.geom[t * 3 + 2];
        const bool kept = in_grid(x, y, z, a.nx, a.ny, a.nz);
        const int b = (int)(t / a.P);
        if (cv == 0) {
            if (kept) write_pos(a.pos_memo, t, b, y, x);
            else if (a.write_dropped) write_pos(a.pos_memo, t, -1, -1, -1);
        }
        if (!kept) continue;
        float *o = a.out + (((int64_t)b * a.ny + y) * a.nx + x) * a.C + cv * VEC;
        const float *f = static_cast<const float *>(a.feats) + t * a.C + cv * VEC;
        if (VEC == 4) {
            const float4 v = *reinterpret_cast<const float4 *>(f);
            atomicAdd(o, v.x);
            atomicAdd(o + 1, v.y);
            atomicAdd(o + 2, v.z);
            atomicAdd(o + 3, v.w);
        } else {
            atomicAdd(o, f[0]);
        }
    }
}

// ---------------------------------------------------------------------------
// ALGO_LDS_ATOMIC (kept for comparison and for C % 4 != 0): LDS-staged BEV-tile combine
// with LDS float atomics.
//
// A workgroup owns chunks of kChunk consecutive points.  Per chunk:
//  A. index pass: read geom, bounds test, write pos_memo, insert the cell key
//     (b*ny+y)*nx+x into an LDS hash table; the inserting lane claims the next free
//     LDS BEV row ("slot").
//  B. stream pass: the chunk's [points, C] feature block is read as flat float4s
//     (fully coalesced; rows of dropped points are not fetched) and added into the
//     slot rows with LDS float atomics.
//  C. flush: every used slot is added to the global BEV as one coalesced row of
//     global fp32 atomics.
// Points whose cell finds no slot (more distinct cells in the chunk than LDS rows)
// fall back to direct row atomics, so any geometry is handled.
//
// LDS row layout for VEC==4: channel 4j+k is stored at k*C4 + j, so that the four
// ds_add_f32 of a lane's float4 hit consecutive banks across lanes (conflict-free
// within a row); the flush undoes the permutation.
template <int VEC, int C4T>
__global__ __launch_bounds__(kBlock) void vp_fwd_lds_combine(VpArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int C = a.C;
    const int CV = (VEC == 4) ? (C4T > 0 ? C4T : C / 4) : C;
    float *acc = reinterpret_cast<float *>(smem);               // [nslot][C]
    int *tab = reinterpret_cast<int *>(acc + (size_t)a.nslot * C);  // [kHashSize] keys
    int *ent_slot = tab + kHashSize;                            // [kHashSize] slot of entry
    int *slot_key = ent_slot + kHashSize;                       // [nslot]
    int *pt_slot = slot_key + a.nslot;                          // [kChunk]
    int *counter = pt_slot + kChunk;                            // [1]

    const int tid = threadIdx.x;
    constexpr int PPT = kChunk / kBlock;  // points per thread

    for (int chunk = blockIdx.x; chunk < a.nchunks; chunk += gridDim.x) {
        const int64_t base = (int64_t)chunk * kChunk;
        const int npts = (int)((a.BP - base) < kChunk ? (a.BP - base) : kChunk);

        for (int i = tid; i < kHashSize; i += kBlock) tab[i] = kEmpty;
        if (tid == 0) *counter = 0;
        __syncthreads();

        // ---- A: index pass
        int ent[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int lp = tid + k * kBlock;
            int e = kDropped;
            if (lp < npts) {
                const int64_t t = base + lp;
                const int x = a.geom[t * 3], y = a.geom[t * 3 + 1], z = a.geom[t * 3 + 2];
                if (in_grid(x, y, z, a.nx, a.ny, a.nz)) {
                    const int b = (int)(t / a.P);
                    write_pos(a.pos_memo, t, b, y, x);
                    const int key = (b * a.ny + y) * a.nx + x;
                    unsigned h = ((unsigned)key * 2654435761u) >> (32 - 9);  // log2(kHashSize)=9
                    e = kOverflow;
                    for (int probe = 0; probe < kHashSize; ++probe) {
                        const int prev = atomicCAS(&tab[h], kEmpty, key);
                        if (prev == kEmpty) {
                            const int s = atomicAdd(counter, 1);
                            if (s < a.nslot) { ent_slot[h] = s; slot_key[s] = key; }
                            else ent_slot[h] = kOverflow;
                            e = (int)h;
                            break;
                        }
                        if (prev == key) { e = (int)h; break; }
                        h = (h + 1) & (kHashSize - 1);
                    }
                } else if (a.write_dropped) {
                    write_pos(a.pos_memo, t, -1, -1, -1);
                }
            }
            ent[k] = e;
        }
        __syncthreads();
        const int nused = (*counter < a.nslot) ? *counter : a.nslot;
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            pt_slot[tid + k * kBlock] = ent[k] >= 0 ? ent_slot[ent[k]] : ent[k];
        for (int i = tid; i < nused * C; i += kBlock) acc[i] = 0.0f;
        __syncthreads();

        // ---- B: stream pass
        const int nvec = npts * CV;
        if (VEC == 4) {
            const float4 *src = reinterpret_cast<const float4 *>(static_cast<const float *>(a.feats) + base * C);
            constexpr int U = 4;
            for (int i0 = tid; i0 < nvec; i0 += kBlock * U) {
                float4 v[U];
                int slot[U], cv[U], row[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u * kBlock;
                    slot[u] = kDropped;
                    if (i < nvec) {
                        row[u] = i / CV;
                        cv[u] = i - row[u] * CV;
                        slot[u] = pt_slot[row[u]];
                        if (slot[u] != kDropped) v[u] = src[i];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (slot[u] >= 0) {
                        float *r = acc + slot[u] * C + cv[u];
                        atomicAdd(r, v[u].x);
                        atomicAdd(r + CV, v[u].y);
                        atomicAdd(r + 2 * CV, v[u].z);
                        atomicAdd(r + 3 * CV, v[u].w);
                    } else if (slot[u] == kOverflow) {
                        const int64_t t = base + row[u];
                        const int x = a.geom[t * 3], y = a.geom[t * 3 + 1];
                        const int b = (int)(t / a.P);
                        float *o = a.out + (((int64_t)b * a.ny + y) * a.nx + x) * C + cv[u] * 4;
                        atomicAdd(o, v[u].x);
                        atomicAdd(o + 1, v[u].y);
                        atomicAdd(o + 2, v[u].z);
                        atomicAdd(o + 3, v[u].w);
                    }
                }
            }
        } else {
            const float *src = static_cast<const float *>(a.feats) + base * C;
            for (int i = tid; i < nvec; i += kBlock) {
                const int row = i / C;
                const int c = i - row * C;
                const int slot = pt_slot[row];
                if (slot >= 0) {
                    atomicAdd(acc + slot * C + c, src[i]);
                } else if (slot == kOverflow) {
                    const int64_t t = base + row;
                    const int x = a.geom[t * 3], y = a.geom[t * 3 + 1];
                    const int b = (int)(t / a.P);
                    atomicAdd(a.out + (((int64_t)b * a.ny + y) * a.nx + x) * C + c, src[i]);
                }
            }
        }
        __syncthreads();

        // ---- C: flush used slots as coalesced rows of global atomics
        for (int i = tid; i < nused * C; i += kBlock) {
            const int s = i / C;
            const int c = i - s * C;
            const int pos = (VEC == 4) ? ((c & 3) * CV + (c >> 2)) : c;
            atomicAdd(a.out + (int64_t)slot_key[s] * C + c, acc[s * C + pos]);
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------
// Feature storage types.  A lane owns one 16-byte vector of a feature row: 4 fp32 channels or 8 bf16 channels
// (SURVEY 5.6 / BASELINE configs[4]: bf16 STORAGE, fp32 accumulate -- every sum below is fp32 either way).

// A gathered row vector stays in its 16-byte load form (Raw) until it is added (bf16: still packed, so 8 rows in flight
// cost 32 registers, not 64).
template <typename FT> struct RowVec;
template <> struct RowVec<float> {
    static constexpr int VEC = 4;
    typedef float4 Raw;
    __device__ __forceinline__ static Raw zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ static Raw load(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    __device__ __forceinline__ static void unpack(const Raw &r, float (&o)[4]) { o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w; }
    __device__ __forceinline__ static float scalar(const float *p) { return *p; }
};
template <> struct RowVec<bf16_t> {
    static constexpr int VEC = 8;
    typedef uint4 Raw;
    __device__ __forceinline__ static Raw zero() { return make_uint4(0u, 0u, 0u, 0u); }
    __device__ __forceinline__ static Raw load(const bf16_t *p) { return *reinterpret_cast<const uint4 *>(p); }
    __device__ __forceinline__ static void unpack(const Raw &r, float (&o)[8]) {
        o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xFFFF0000u);
        o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xFFFF0000u);
        o[4] = __uint_as_float(r.z << 16); o[5] = __uint_as_float(r.z & 0xFFFF0000u);
        o[6] = __uint_as_float(r.w << 16); o[7] = __uint_as_float(r.w & 0xFFFF0000u);
    }
    __device__ __forceinline__ static float scalar(const bf16_t *p) { return __uint_as_float((unsigned)*p << 16); }
};

// ---------------------------------------------------------------------------
// ALGO_AUTO: chunk-local sort by BEV cell + register accumulation.
//
// A workgroup owns one chunk of CHUNK consecutive points.
//  A. index pass: read geom, bounds test, write pos_memo, insert the cell key
//     (b*ny+y)*nx+x into an LDS hash table (the inserting lane numbers the cell:
//     "slot"), count points per slot, wave-scan the counts, and scatter the local
//     point ids into a per-slot list (a counting sort of the chunk by cell).
//  B. gather pass: the wave is split into G = 64 / (C/VEC) lane groups, each lane owning one
//     16-byte column of the feature row (4 fp32 or 8 bf16 channels).  Every lane group takes ONE
//     cell of the chunk at a time and walks its list alone, loading whole rows (C*sizeof(FT)
//     contiguous bytes; rows of dropped points are never fetched) and summing them in fp32
//     REGISTERS -- no LDS or global atomics in the loop; the few long lists are walked by a
//     whole wave (every G-th row per group).
//  C. the G partial rows meet in an LDS staging row and leave the wave as contiguous runs of
//     global fp32 atomics, one run per (chunk, cell).
// The BEV tile of the chunk therefore lives in registers + a staging row; HBM sees
// each kept feature row once and one atomic row per (chunk, cell).
template <typename FT, int CVT, int CHUNK, bool FUSED>
__global__ __launch_bounds__(kBlock) void vp_fwd_seg_gather(VpArgs a) {
    using RV = RowVec<FT>;
    constexpr int VEC = RV::VEC;
    constexpr int HT = CHUNK * 2;             // hash entries (load factor <= 0.5)
    constexpr int HT_LOG2 = (CHUNK == 512) ? 10 : 11;
    static_assert(CHUNK == 512 || CHUNK == 1024, "chunk size");
    constexpr int PPT = CHUNK / kBlock;
    constexpr int NW = kBlock / 64;
    __shared__ __align__(16) int tab[2 * HT];    // hash keys | slot ids; reused for pos_memo (12*CHUNK B)
    int *tab_key = tab, *tab_slot = tab + HT;
    static_assert(2 * HT * 4 >= CHUNK * 12, "pos_memo block must fit in the hash table's LDS");
    __shared__ int slot_key[CHUNK];
    __shared__ int slot_cnt[CHUNK];
    __shared__ int slot_off[CHUNK + 1];
    __shared__ unsigned short sorted[CHUNK];
    __shared__ __align__(16) float stage[NW][64 * VEC];     // G*C <= 64*VEC floats per wave
    __shared__ int nslots, next_slot, next_long, nlong;
    __shared__ unsigned short long_list[CHUNK / kLongSlot + 1];
    // fused lift-splat: per point of the chunk its depth probability and its pixel's context row
    __shared__ float pt_depth[FUSED ? CHUNK : 1];
    __shared__ int pt_pix[FUSED ? CHUNK : 1];

    const int C = a.C;
    const int CV = CVT > 0 ? CVT : C / VEC;   // lanes per row
    const int G = 64 / CV;                    // lane groups per wave (C <= 64*VEC)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a.nchunks carries the points per workgroup (<= CHUNK, multiple of 4): the host sizes it so
    // that the grid is a whole number of rounds of the resident workgroup slots (no half-empty tail)
    const int cp = a.nchunks > 0 ? a.nchunks : CHUNK;
    const int64_t base = (int64_t)blockIdx.x * cp;
    const int npts = (int)((a.BP - base) < cp ? (a.BP - base) : cp);

    // all geom rows of the lane are requested up front, ahead of the LDS initialisation and its barrier, (clamped index, so the
    // loads are unconditional): one memory round trip per workgroup instead of PPT serial ones
    int gx[PPT], gy[PPT], gz[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int lp = tid + k * kBlock;
        const int64_t t = base + (lp < npts ? lp : npts - 1);
        gx[k] = a.geom[t * 3]; gy[k] = a.geom[t * 3 + 1]; gz[k] = a.geom[t * 3 + 2];
    }

    for (int i = tid; i < HT; i += kBlock) tab_key[i] = kEmpty;
    for (int i = tid; i < CHUNK; i += kBlock) slot_cnt[i] = 0;
    if (tid == 0) { nslots = 0; next_slot = 0; next_long = 0; nlong = 0; }
    __syncthreads();

    // ---- A1: bounds test, hash insert.  pos_memo rows are kept in registers: in WRITE_DROPPED
    // mode the chunk's whole 12*CHUNK-byte pos_memo block is written later as full 16-byte
    // coalesced stores (stride-12 dword stores leave partially written L2 lines behind, which
    // cost a fetch-on-write: 23 MB of avoidable HBM reads at cfg2).
    int ent[PPT];
    int pm[PPT][3];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int lp = tid + k * kBlock;
        int e = -1;
        pm[k][0] = pm[k][1] = pm[k][2] = -1;
        if (lp < npts) {
            const int64_t t = base + lp;
            if (FUSED) {
                const unsigned cam = (unsigned)t / (unsigned)a.DHW;
                const unsigned rem = (unsigned)t - cam * (unsigned)a.DHW;
                pt_pix[lp] = (int)(cam * (unsigned)a.HW + rem % (unsigned)a.HW);
                pt_depth[lp] = RV::scalar(reinterpret_cast<const FT *>(a.depth) + t);
            }
            const int x = gx[k], y = gy[k], z = gz[k];
            if (in_grid(x, y, z, a.nx, a.ny, a.nz)) {
                const int b = (int)((unsigned)t / (unsigned)a.P);
                pm[k][0] = b; pm[k][1] = y; pm[k][2] = x;
                const int key = (b * a.ny + y) * a.nx + x;
                unsigned h = ((unsigned)key * 2654435761u) >> (32 - HT_LOG2);
                for (int probe = 0; probe < HT; ++probe) {  // never fills: <= CHUNK keys in 2*CHUNK entries
                    const int prev = atomicCAS(&tab_key[h], kEmpty, key);
                    if (prev == kEmpty) {
                        const int s = atomicAdd(&nslots, 1);
                        tab_slot[h] = s;
                        slot_key[s] = key;
                        e = (int)h;
                        break;
                    }
                    if (prev == key) { e = (int)h; break; }
                    h = (h + 1) & (HT - 1);
                }
            }
        }
        ent[k] = e;
    }
    __syncthreads();

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

    // ---- A3: exclusive scan of the counts by wave 0 (CHUNK/64 entries per lane)
    const int ns = nslots;
    if (wave == 0) {
        constexpr int PER = CHUNK / 64;
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
            if (idx <= ns) slot_off[idx] = run;
            run += loc[i];
        }
        // ns == CHUNK (every point of a full chunk in its own cell): idx never reaches ns above
        if (lane == 63) slot_off[ns] = incl;
    }
    __syncthreads();

    // ---- A4: counting-sort scatter of local point ids; the hash table is dead now, so its
    // LDS holds the chunk's pos_memo block for the coalesced write-out
#pragma unroll
    for (int k = 0; k < PPT; ++k)
        if (slot[k] >= 0) sorted[slot_off[slot[k]] + rank[k]] = (unsigned short)(tid + k * kBlock);
    // cells with more than kLongSlot points of this chunk (at most CHUNK / kLongSlot of them)
    for (int i = tid; i < ns; i += kBlock)
        if (slot_cnt[i] > kLongSlot) long_list[atomicAdd(&nlong, 1)] = (unsigned short)i;
    if (a.write_dropped) {
        int *pml = tab;                          // [CHUNK*3] ints
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int lp = tid + k * kBlock;
            pml[lp * 3] = pm[k][0]; pml[lp * 3 + 1] = pm[k][1]; pml[lp * 3 + 2] = pm[k][2];
        }
    }
    __syncthreads();
    // (pos_memo is written here and not in A1: a global store ahead of a barrier makes the barrier's release
    // fence wait for every outstanding memory operation)
    if (!a.write_dropped) {
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            if (pm[k][0] != -1) write_pos(a.pos_memo, base + tid + k * kBlock, pm[k][0], pm[k][1], pm[k][2]);
    }
    if (a.write_dropped) {
        const int *pml = tab;
        int32_t *dstp = a.pos_memo + base * 3;   // 12*CHUNK*blockIdx bytes: 16-byte aligned
        const int nint = npts * 3;
        if ((nint & 3) == 0) {
            for (int i = tid; i < (nint >> 2); i += kBlock)
                reinterpret_cast<int4 *>(dstp)[i] = reinterpret_cast<const int4 *>(pml)[i];
        } else {
            for (int i = tid; i < nint; i += kBlock) dstp[i] = pml[i];
        }
    }

    // ---- B/C: register accumulation, staged flush.
    // Pass 1: every lane group owns ONE cell (slot) at a time -- G cells per wave in flight, up to
    // kRows rows each -- because the typical slot is short (36 slots / ~280 kept rows per chunk at
    // cfg2): one slot per WAVE left two thirds of the load slots empty and paid the ticket /
    // staging overhead per slot instead of per G slots.  Pass 2: the few long slots are walked by
    // a whole wave (G groups, every G-th row) so no group serialises hundreds of rows.
    constexpr int kRows = VEC == 8 ? 6 : 8;      // rows in flight per lane group (bf16: 6 keeps the kernel at 6 waves / SIMD)
    const int g = lane / CV;
    const int li = lane - g * CV;
    const bool active = g < G;
    const FT *fbase = (FUSED ? reinterpret_cast<const FT *>(a.context) : reinterpret_cast<const FT *>(a.feats) + base * C) + li * VEC;
    float *st = stage[wave];
    for (;;) {
        int s0 = 0;
        if (lane == 0) s0 = atomicAdd(&next_slot, G);
        s0 = __builtin_amdgcn_readfirstlane(s0);
        if (s0 >= ns) break;
        const int s = s0 + g;
        int beg = 0, end = 0;
        if (active && s < ns) {
            beg = slot_off[s];
            end = slot_off[s + 1];
            if (end - beg > kLongSlot) end = beg;      // left to pass 2
        }
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int j = beg; j < end; j += kRows) {
            typename RV::Raw v[kRows];
            float dv[kRows];
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                v[u] = RV::zero();
                dv[u] = 0.f;
                if (j + u < end) {
                    const int p = sorted[j + u];
                    if (FUSED) {
                        dv[u] = pt_depth[p];
                        v[u] = RV::load(fbase + (int64_t)pt_pix[p] * C);
                    } else {
                        v[u] = RV::load(fbase + p * C);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                float f[VEC];
                RV::unpack(v[u], f);
#pragma unroll
                for (int e = 0; e < VEC; ++e)   // fused: product rounded to fp32 first (= the materialised lift), then added
                    acc[e] += FUSED ? __fmul_rn(dv[u], f[e]) : f[e];
            }
        }
        if (active) {
#pragma unroll
            for (int e = 0; e < VEC; e += 4)
                *reinterpret_cast<float4 *>(st + g * C + li * VEC + e) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // G rows leave the wave as contiguous runs of global fp32 atomics
        for (int e = lane; e < G * C; e += 64) {
            const int gg = e / C;
            const int ss = s0 + gg;
            if (ss < ns) {
                const int n = slot_off[ss + 1] - slot_off[ss];
                if (n <= kLongSlot) atomicAdd(a.out + (int64_t)slot_key[ss] * C + (e - gg * C), st[e]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const int nl = nlong;
    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(&next_long, 1);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= nl) break;
        const int s = long_list[t];
        const int beg = slot_off[s], end = slot_off[s + 1];
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        if (active) {
            for (int j = beg + g; j < end; j += 4 * G) {
                typename RV::Raw v[4];
                float dv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jj = j + u * G;
                    v[u] = RV::zero();
                    dv[u] = 0.f;
                    if (jj < end) {
                        const int p = sorted[jj];
                        if (FUSED) {
                            dv[u] = pt_depth[p];
                            v[u] = RV::load(fbase + (int64_t)pt_pix[p] * C);
                        } else {
                            v[u] = RV::load(fbase + p * C);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float f[VEC];
                    RV::unpack(v[u], f);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] += FUSED ? __fmul_rn(dv[u], f[e]) : f[e];
                }
            }
#pragma unroll
            for (int e = 0; e < VEC; e += 4)
                *reinterpret_cast<float4 *>(st + g * C + li * VEC + e) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float *orow = a.out + (int64_t)slot_key[s] * C;
        for (int e = lane; e < C; e += 64) {
            float sum = st[e];
            for (int gg = 1; gg < G; ++gg) sum += st[gg * C + e];
            atomicAdd(orow + e, sum);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}


// ---------------------------------------------------------------------------
// ALGO_STREAM: chunk sorted by BEV cell, then a BALANCED stream over the sorted list.
//
// Same index pass as SEG_GATHER (hash -> slot, count, wave scan, counting sort), but
// the gather no longer iterates cell by cell.  The K kept points of the chunk, in
// cell-sorted order, are split into equal contiguous segments, one per lane group
// (4 waves x G groups); a group streams its segment with 4 whole-row loads in flight
// and keeps the running sum of the current cell in registers.  Only when the cell id
// changes does it hand the finished row to an LDS row buffer: a plain ds_write if the
// cell lies entirely inside the group's segment (the common case), ds_add_f32 for the
// <= (groups-1) cells that straddle a segment boundary.  After one barrier the row
// buffer leaves the workgroup as contiguous runs of global fp32 atomics.
// Per-cell overhead (ticket, staging, wave barriers) is gone and every group has the
// same number of rows, so a chunk with one hot cell costs the same as an even one.
constexpr int kStreamRowCap = 64;  // LDS row buffer rows; further cells flush straight to HBM

template <int C4T, int CHUNK>
__global__ __launch_bounds__(kBlock) void vp_fwd_stream(VpArgs a) {
    constexpr int HT = CHUNK * 2;
    constexpr int HT_LOG2 = (CHUNK == 512) ? 10 : 11;
    static_assert(CHUNK == 512 || CHUNK == 1024, "chunk size");
    constexpr int PPT = CHUNK / kBlock;
    constexpr int NW = kBlock / 64;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // [rowbuf: kStreamRowCap*C floats] aliases nothing; tables follow
    const int C = a.C;
    float *rowbuf = reinterpret_cast<float *>(smem_raw);
    int *tab_key = reinterpret_cast<int *>(rowbuf + kStreamRowCap * C);
    unsigned short *tab_slot = reinterpret_cast<unsigned short *>(tab_key + HT);
    int *slot_key = reinterpret_cast<int *>(tab_slot + HT);
    unsigned short *slot_cnt = reinterpret_cast<unsigned short *>(slot_key + CHUNK);
    unsigned short *slot_off = slot_cnt + CHUNK;            // [CHUNK + 8] (padded: keeps `stage` 16-byte aligned)
    unsigned short *sorted = slot_off + CHUNK + 8;
    unsigned short *sorted_slot = sorted + CHUNK;
    float *stage = reinterpret_cast<float *>(sorted_slot + CHUNK);  // [NW][256] overflow staging
    __shared__ int nslots;

    const int C4 = C4T > 0 ? C4T : C >> 2;
    const int G = 64 / C4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * CHUNK;
    const int npts = (int)((a.BP - base) < CHUNK ? (a.BP - base) : CHUNK);

    for (int i = tid; i < HT; i += kBlock) tab_key[i] = kEmpty;
    for (int i = tid; i < CHUNK; i += kBlock) slot_cnt[i] = 0;
    if (tid == 0) nslots = 0;
    __syncthreads();

    // ---- A1: bounds test, pos_memo, hash insert
    int ent[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int lp = tid + k * kBlock;
        int e = -1;
        if (lp < npts) {
            const int64_t t = base + lp;
            const int x = a.geom[t * 3], y = a.geom[t * 3 + 1], z = a.geom[t * 3 + 2];
            if (in_grid(x, y, z, a.nx, a.ny, a.nz)) {
                const int b = (int)((unsigned)t / (unsigned)a.P);
                write_pos(a.pos_memo, t, b, y, x);
                const int key = (b * a.ny + y) * a.nx + x;
                unsigned h = ((unsigned)key * 2654435761u) >> (32 - HT_LOG2);
                for (int probe = 0; probe < HT; ++probe) {
                    const int prev = atomicCAS(&tab_key[h], kEmpty, key);
                    if (prev == kEmpty) {
                        const int sidx = atomicAdd(&nslots, 1);
                        tab_slot[h] = (unsigned short)sidx;
                        slot_key[sidx] = key;
                        e = (int)h;
                        break;
                    }
                    if (prev == key) { e = (int)h; break; }
                    h = (h + 1) & (HT - 1);
                }
            } else if (a.write_dropped) {
                write_pos(a.pos_memo, t, -1, -1, -1);
            }
        }
        ent[k] = e;
    }
    __syncthreads();

    // ---- A2: per-slot counts. 16-bit counters are packed two per dword: add through the
    // containing dword (a count never exceeds CHUNK <= 1024, so no carry into the neighbour)
    int slot[PPT], rank[PPT];
    unsigned *cnt32 = reinterpret_cast<unsigned *>(slot_cnt);
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        slot[k] = -1;
        rank[k] = 0;
        if (ent[k] >= 0) {
            slot[k] = tab_slot[ent[k]];
            const int sh = (slot[k] & 1) * 16;
            const unsigned old = atomicAdd(&cnt32[slot[k] >> 1], 1u << sh);
            rank[k] = (int)((old >> sh) & 0xFFFFu);
        }
    }
    __syncthreads();

    // ---- A3: exclusive scan of the counts by wave 0
    const int ns = nslots;
    if (wave == 0) {
        constexpr int PER = CHUNK / 64;
        int loc[PER];
        int sum = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = lane * PER + i;
            loc[i] = idx < ns ? (int)slot_cnt[idx] : 0;
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
        if (lane == 63) {
            slot_off[ns] = (unsigned short)incl;          // needed when ns == CHUNK (idx never reaches it)
            slot_off[CHUNK + 1] = (unsigned short)incl;   // total kept
        }
    }
    __syncthreads();

    // ---- A4: counting-sort scatter (point id and its slot, cell-sorted order)
#pragma unroll
    for (int k = 0; k < PPT; ++k)
        if (slot[k] >= 0) {
            const int pos = slot_off[slot[k]] + rank[k];
            sorted[pos] = (unsigned short)(tid + k * kBlock);
            sorted_slot[pos] = (unsigned short)slot[k];
        }
    const int nbuf = ns < kStreamRowCap ? ns : kStreamRowCap;
    for (int i = tid; i < nbuf * C; i += kBlock) rowbuf[i] = 0.f;
    __syncthreads();

    // ---- B: balanced stream over the sorted list
    const int nkept = slot_off[CHUNK + 1];
    const int g = lane / C4;
    const int li = lane - g * C4;
    if (g < G && nkept > 0) {
        const int NG = NW * G;
        const int gid = wave * G + g;
        const int L = (nkept + NG - 1) / NG;
        const int jb = gid * L;
        const int je = (jb + L) < nkept ? (jb + L) : nkept;
        const float *fbase = static_cast<const float *>(a.feats) + base * C + li * 4;
        float *st = stage + wave * 256 + g * C;       // this group's private staging row
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int cur = jb < je ? (int)sorted_slot[jb] : -1;

        auto flush = [&](int s_) {
            if (s_ < kStreamRowCap) {
                float *r = rowbuf + s_ * C + li * 4;
                const bool interior = (int)slot_off[s_] >= jb && (int)slot_off[s_ + 1] <= je;
                if (interior) {
                    *reinterpret_cast<float4 *>(r) = acc;
                } else {
                    atomicAdd(r, acc.x); atomicAdd(r + 1, acc.y); atomicAdd(r + 2, acc.z); atomicAdd(r + 3, acc.w);
                }
            } else {
                // row buffer full: transpose through the private staging row so each
                // atomic instruction of the group covers 80 contiguous bytes
                *reinterpret_cast<float4 *>(st + li * 4) = acc;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                float *orow = a.out + (int64_t)slot_key[s_] * C;
                for (int e = li; e < C; e += C4) atomicAdd(orow + e, st[e]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
        };

        for (int j = jb; j < je; j += 4) {
            float4 v[4];
            int sl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jj = j + u;
                sl[u] = -1;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (jj < je) {
                    sl[u] = sorted_slot[jj];
                    v[u] = *reinterpret_cast<const float4 *>(fbase + (int)sorted[jj] * C);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (sl[u] >= 0) {
                    if (sl[u] != cur) {
                        flush(cur);
                        acc = make_float4(0.f, 0.f, 0.f, 0.f);
                        cur = sl[u];
                    }
                    acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
                }
            }
        }
        if (cur >= 0) flush(cur);
    }
    __syncthreads();

    // ---- C: the buffered rows leave as contiguous runs of global fp32 atomics
    for (int i = tid; i < nbuf * C; i += kBlock) {
        const int s_ = i / C;
        atomicAdd(a.out + (int64_t)slot_key[s_] * C + (i - s_ * C), rowbuf[i]);
    }
}


// ---------------------------------------------------------------------------
// Backward: grad_in[t,:] = grad_out[b,:,y,x] (kept) or 0.  Pure gather: the BEV
// gradient (B*ny*nx*C fp32, 21 MB at cfg2) is served from L2 / Infinity Cache, the
// [B*P, C] result is streamed out with non-temporal 16-byte stores so it does not
// evict the gradient rows.
struct VpBwdArgs {
    int64_t BP;
    int C, nx, ny;
    const int32_t *pos_memo;
    const float *grad_out;
    int64_t sb, sc, sy, sx;
    void *grad_in;       // fp32 or bf16 [BP, C] (the kernel's OT template argument says which)
    int64_t span_bytes;  // bytes spanned by the grad_out view (buffer descriptor range)
    const uint32_t *row_off;  // optional [BP] byte offset of each point's BEV-gradient row (prepared), or NULL
    uint32_t *row_off_out;
};

// Backward, pass 1 (tiny): pos_memo -> per-point byte offset of its BEV-gradient row
// (0xFFFFFFF0 = dropped), and pull the gradient on-die before the write-heavy pass starts.
// Why a separate pass: with ~93 % of pass 2's HBM traffic being writes, a read that misses the
// Infinity Cache queues behind the write drain (measured: the same gather kernel runs 100 us
// with pos_memo / grad_out cache-warm and 170 us cold, although only 44 MB of its 650 MB are
// reads).  Pass 1 reads them while the memory system is idle and leaves 4 B/point in L2 /
// Infinity Cache.  Same XCD-contiguous partition of the points as pass 2.
//
// The pass is a short dependent chain on a small problem, i.e. bound by memory latency: each lane
// takes kPrepU points per trip and requests all of their pos_memo rows together (clamped index,
// unconditional loads; one point per trip cost ~4 serial round trips per lane).
// The gradient is warmed by a LINEAR sweep of its span (16 B per lane, coalesced), not by gathering
// the rows the points use: the gather (5 sector touches per kept point, 5.7 M scattered L2 requests
// at cfg2) cost 9 us and bought nothing -- pass 2 runs equally fast when its first touch of a row
// is an Infinity-Cache hit instead of an L2 hit (interleaved A/B, us, gather touches / sweep / no
// warm-up: cfg2 138.8 / 130.4 / 129.9, after a 1 GiB cache flush 138.3 / 133.5 / 130.9; cfg5 246.8 /
// 234.8 / 233.1, flushed 245.8 / 233.0 / 240.2; inside the cfg-2 training step 137.9 / 131.5 / 130.0).
// The sweep costs 1-3 us and covers the cold-gradient case; it is skipped when the view spans much more
// memory than it uses (a thin channel slice of a wide buffer).
constexpr int kPrepU = 4;

__global__ __launch_bounds__(kBlock) void vp_bwd_prepare(VpBwdArgs a, int rows_per_xcd, int sweep) {
    const int xcd = blockIdx.x & 7;
    const int64_t r_begin = (int64_t)xcd * rows_per_xcd;
    const int64_t r_end = (r_begin + rows_per_xcd) < a.BP ? (r_begin + rows_per_xcd) : a.BP;
    const int64_t step = (int64_t)(gridDim.x >> 3) * kBlock * kPrepU;
    for (int64_t t0 = r_begin + (int64_t)(blockIdx.x >> 3) * kBlock * kPrepU + threadIdx.x; t0 < r_end; t0 += step) {
        int b[kPrepU], y[kPrepU], x[kPrepU];
#pragma unroll
        for (int u = 0; u < kPrepU; ++u) {
            const int64_t t = t0 + (int64_t)u * kBlock;
            const int64_t tc = t < r_end ? t : r_end - 1;     // clamped: every load is unconditional
            b[u] = a.pos_memo[tc * 3];
            y[u] = a.pos_memo[tc * 3 + 1];
            x[u] = a.pos_memo[tc * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < kPrepU; ++u) {
            const int64_t t = t0 + (int64_t)u * kBlock;
            const int64_t tc = t < r_end ? t : r_end - 1;     // clamped lanes rewrite the last point's own value
            const unsigned o = (unsigned)(b[u] * a.sb + y[u] * a.sy + x[u] * a.sx) * 4u;
            a.row_off_out[tc] = (b[u] != -1) ? o : 0xFFFFFFF0u;
        }
    }
    if (sweep) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(a.grad_out), 0, (int)a.span_bytes, 0x00020000);
        const int64_t nvec = a.span_bytes >> 4;
        const int64_t nthr = (int64_t)gridDim.x * kBlock;
        unsigned sink = 0u;
        for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < nvec; i0 += 4 * nthr) {
            mmt_u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // past the end: out-of-range offset, the hardware returns zeros without an access
                const int64_t i = i0 + u * nthr;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, i < nvec ? (unsigned)(i << 4) : 0xFFFFFFF0u, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) sink ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
        if (sink == 0x9E3779B9u && a.BP < 0) a.row_off_out[0] = 0;  // never true: keeps the sweep alive
    }
}

// Software-pipelined: the dependent chain pos_memo -> BEV-gradient row -> store is
// what bounds this kernel when the caches are cold (a bare 605 MB memset runs at
// ~8 TB/s on this chip, a naive load-load-store loop at ~3.5 TB/s because every trip
// waits two full HBM read latencies behind a saturated write queue).  So each lane
// keeps the row offsets of tile t+2 and the gathered vectors of tile t+1 in flight
// while it stores tile t.
// OT = float: grad_in fp32, a lane's 16-byte output vector is one float4 of the gradient row;
// OT = bf16_t: grad_in bf16 (bf16 storage path), a lane's 16-byte output vector holds 8 channels = two float4
// of the fp32 gradient row, rounded to nearest even on the way out.
template <typename OT, int CVT>
__global__ __launch_bounds__(kBlock) void vp_bwd_rows_vec(VpBwdArgs a) {
    constexpr int VEC = 16 / (int)sizeof(OT);   // channels per lane vector
    constexpr int NL = VEC / 4;                 // 16-byte gradient loads per lane vector
    const int CV = CVT > 0 ? CVT : a.C / VEC;
    const int64_t total = a.BP * CV;
    mmt_u32x4 *dst = reinterpret_cast<mmt_u32x4 *>(a.grad_in);
    constexpr int U = 4;
    constexpr int64_t kTileVecs = (int64_t)kBlock * U;
    // XCD-aware tile order: workgroups b, b+8, b+16.. share an XCD and its 4 MiB L2, so
    // each XCD walks ONE contiguous eighth of the points; the BEV-gradient rows those
    // points gather (a few cameras' footprint, ~2-3 MB) then stay L2-resident instead of
    // all 8 L2s thrashing over the whole 21 MB gradient.  Placement only affects speed.
    const int64_t ntiles = (total + kTileVecs - 1) / kTileVecs;
    const int64_t per_xcd = (ntiles + 7) / 8;
    const int xcd = blockIdx.x & 7;
    const int64_t t_begin = xcd * per_xcd;
    const int64_t t_end = (t_begin + per_xcd) < ntiles ? (t_begin + per_xcd) : ntiles;
    const int64_t t_first = t_begin + (blockIdx.x >> 3);
    const int64_t t_step = gridDim.x >> 3;

    // Only FULL tiles go through the pipelined loop and every load / store in it is
    // unconditional (dropped points read row 0 and select zero afterwards): hipcc can
    // then use counted s_waitcnt vmcnt(N) and really keep two tiles of loads in flight;
    // a predicated load would force vmcnt(0) and collapse the pipeline.
    const int64_t nfull = total / kTileVecs;
    const int64_t my_end = t_end < nfull ? t_end : nfull;

    // The gather goes through a buffer descriptor: a dropped point uses an out-of-range
    // byte offset and the hardware range check returns zeros -- no select, no branch.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.grad_out), 0, (int)a.span_bytes, 0x00020000);

    // BYTE offset of the lane's first float4 inside grad_out, or 0xFFFFFFF0 (dropped point)
    auto load_offsets = [&](int64_t tile, unsigned (&off)[U]) {
        tile = tile < my_end ? tile : t_first;  // prefetch past the end re-reads a valid tile
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = tile * kTileVecs + threadIdx.x + (int64_t)u * kBlock;
            const int64_t t = i / CV;
            const int cv = (int)(i - t * CV);
            if (a.row_off) {   // prepared by vp_bwd_prepare (wave-uniform branch)
                const unsigned ro = a.row_off[t];
                off[u] = ro == 0xFFFFFFF0u ? ro : ro + cv * (VEC * 4u);
            } else {
                const int b = a.pos_memo[t * 3];
                const int y = a.pos_memo[t * 3 + 1], x = a.pos_memo[t * 3 + 2];
                const unsigned o = ((unsigned)(b * a.sb + y * a.sy + x * a.sx) + cv * VEC) * 4u;
                off[u] = (b != -1) ? o : 0xFFFFFFF0u;
            }
        }
    };
    auto gather = [&](const unsigned (&off)[U], mmt_u32x4 (&v)[U][NL]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int l = 0; l < NL; ++l)   // the out-of-range marker must stay out of range (no wrap-around)
                v[u][l] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (l == 0 || off[u] == 0xFFFFFFF0u) ? off[u] : off[u] + 16u * l, 0, 0);
        }
    };
    auto pack = [&](const mmt_u32x4 (&v)[NL]) -> mmt_u32x4 {
        if constexpr (NL == 1) {
            return v[0];
        } else {
            mmt_u32x4 r;
            r.x = pack_bf16x2(__uint_as_float(v[0].x), __uint_as_float(v[0].y));
            r.y = pack_bf16x2(__uint_as_float(v[0].z), __uint_as_float(v[0].w));
            r.z = pack_bf16x2(__uint_as_float(v[1].x), __uint_as_float(v[1].y));
            r.w = pack_bf16x2(__uint_as_float(v[1].z), __uint_as_float(v[1].w));
            return r;
        }
    };

    if (t_first < my_end) {
        unsigned off1[U], off2[U];
        mmt_u32x4 g0[U][NL], g1[U][NL];
        load_offsets(t_first, off1);
        gather(off1, g0);
        load_offsets(t_first + t_step, off1);
        auto store_tile = [&](int64_t tile, const mmt_u32x4 (&v)[U][NL]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = tile * kTileVecs + threadIdx.x + (int64_t)u * kBlock;
                __builtin_nontemporal_store(pack(v[u]), dst + i);
            }
        };
        // unrolled by two with the register sets swapping roles: no copies, so nothing
        // waits on the loads issued in the same trip
        for (int64_t tile = t_first; tile < my_end; tile += 2 * t_step) {
            load_offsets(tile + 2 * t_step, off2);
            gather(off1, g1);
            store_tile(tile, g0);
            if (tile + t_step >= my_end) break;
            load_offsets(tile + 3 * t_step, off1);
            gather(off2, g0);
            store_tile(tile + t_step, g1);
        }
    }
    // the single partial tile at the very end
    if (blockIdx.x == 0 && nfull * kTileVecs < total) {
        for (int64_t i = nfull * kTileVecs + threadIdx.x; i < total; i += kBlock) {
            const int64_t t = i / CV;
            const int cv = (int)(i - t * CV);
            const int b = a.pos_memo[t * 3];
            mmt_u32x4 v[NL];
#pragma unroll
            for (int l = 0; l < NL; ++l) v[l] = mmt_u32x4{0u, 0u, 0u, 0u};
            if (b != -1) {
                const int y = a.pos_memo[t * 3 + 1], x = a.pos_memo[t * 3 + 2];
                const mmt_u32x4 *src = reinterpret_cast<const mmt_u32x4 *>(a.grad_out + b * a.sb + y * a.sy + x * a.sx + cv * VEC);
#pragma unroll
                for (int l = 0; l < NL; ++l) v[l] = src[l];
            }
            dst[i] = pack(v);
        }
    }
}

// any strides, any C (slow path: one element per lane)
template <typename OT>
__global__ __launch_bounds__(kBlock) void vp_bwd_strided(VpBwdArgs a) {
    const int64_t total = a.BP * a.C;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    OT *gi = reinterpret_cast<OT *>(a.grad_in);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
        const int64_t t = i / a.C;
        const int c = (int)(i - t * a.C);
        const int b = a.pos_memo[t * 3];
        float v = 0.f;
        if (b != -1) {
            const int y = a.pos_memo[t * 3 + 1], x = a.pos_memo[t * 3 + 2];
            v = a.grad_out[b * a.sb + c * a.sc + y * a.sy + x * a.sx];
        }
        if constexpr (sizeof(OT) == 4) gi[i] = v;
        else gi[i] = (OT)(pack_bf16x2(v, 0.f) & 0xFFFFu);
    }
}

// [B,C,ny,nx] (element strides) -> channels-last workspace [B,ny*nx,C] through a
// 32x33 LDS tile: reads run along the spatial axis, writes along the channel axis.
__global__ __launch_bounds__(kBlock) void vp_to_channels_last(int C, int ny, int nx,
                                                              const float *src, int64_t sb,
                                                              int64_t sc, int64_t sy, int64_t sx,
                                                              float *dst) {
    __shared__ float tile[32][33];
    const int HW = ny * nx;
    const int b = blockIdx.z;
    const int s0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, s = s0 + tx;
        float v = 0.f;
        if (c < C && s < HW) {
            const int y = s / nx, x = s - y * nx;
            v = src[b * sb + c * sc + y * sy + x * sx];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int s = s0 + r, c = c0 + tx;
        if (c < C && s < HW) dst[((int64_t)b * HW + s) * C + c] = tile[tx][r];
    }
}

// ---------------------------------------------------------------------------
// Fused lift-splat: voxel_pooling of features that are never
// materialised,  row(t) = depth[t] * context[pix(t), :]  (lss_fpn.py:441-464 in one pass).
// Forward = vp_fwd_seg_gather<.., FUSED=true> (the gather reads context rows, 5 MB and
// L2-resident, instead of 364 MB of lifted rows).  Backward below is pixel-major and needs
// no atomics:  grad_context[pix,:] = sum_d depth[t] * grad_out[cell(t),:]
//              grad_depth[t]       = < grad_out[cell(t),:], context[pix,:] >
// One lane group (C/4 lanes, a float4 column each) owns one pixel and walks its D depth
// bins; grad_out rows are L2 gathers through a range-checked buffer descriptor (dropped
// points read zeros), 4 bins in flight.
template <typename FT, int C4T>
__global__ __launch_bounds__(kBlock) void lift_splat_backward_kernel(
    int D, int HW, int C, const int32_t *pos_memo, const FT *depth, const FT *context,
    const float *grad_out, int64_t sb, int64_t sy, int64_t sx, int64_t span_bytes,
    FT *grad_depth, FT *grad_context) {
    constexpr bool kBf16 = sizeof(FT) == 2;
    extern __shared__ __align__(16) float lds[];
    const int C4 = C4T > 0 ? C4T : C >> 2;
    const int G = 64 / C4;
    const int NG = (kBlock / 64) * G;            // pixels per workgroup
    float *gd = lds;                             // [D][NG] grad_depth accumulators
    const int bn = blockIdx.y;
    const int s0 = blockIdx.x * NG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane / C4, li = lane - grp * C4;
    const int j = wave * G + grp;
    const bool active = grp < G && (s0 + j) < HW;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(grad_out), 0, (int)span_bytes, 0x00020000);
    for (int i = tid; i < D * NG; i += kBlock) gd[i] = 0.f;
    const int64_t pix = (int64_t)bn * HW + s0 + (active ? j : 0);
    float4 cx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        if constexpr (kBf16) {
            const uint2 r = *reinterpret_cast<const uint2 *>(context + pix * C + li * 4);
            cx = make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xFFFF0000u),
                             __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xFFFF0000u));
        } else {
            cx = *reinterpret_cast<const float4 *>(context + pix * C + li * 4);
        }
    }
    __syncthreads();

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t t0 = (int64_t)bn * D * HW + s0 + (active ? j : 0);   // point index of bin 0
    for (int d0 = 0; d0 < D; d0 += 4) {
        mmt_u32x4 v[4];
        float dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int d = d0 + u;
            unsigned off = 0xFFFFFFF0u;
            dv[u] = 0.f;
            if (active && d < D) {
                const int64_t t = t0 + (int64_t)d * HW;
                const int b = pos_memo[t * 3];
                const int y = pos_memo[t * 3 + 1], x = pos_memo[t * 3 + 2];
                if (b != -1) off = ((unsigned)(b * sb + y * sy + x * sx) + li * 4) * 4u;
                dv[u] = RowVec<FT>::scalar(depth + t);
            }
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int d = d0 + u;
            if (d < D) {
                const float gx_ = __uint_as_float(v[u].x), gy_ = __uint_as_float(v[u].y);
                const float gz_ = __uint_as_float(v[u].z), gw_ = __uint_as_float(v[u].w);
                acc.x += gx_ * dv[u]; acc.y += gy_ * dv[u]; acc.z += gz_ * dv[u]; acc.w += gw_ * dv[u];
                float dot = gx_ * cx.x + gy_ * cx.y + gz_ * cx.z + gw_ * cx.w;
                dot += __shfl_xor(dot, 1);
                dot += __shfl_xor(dot, 2);
                if (active && (li & 3) == 0) atomicAdd(&gd[d * NG + j], dot);
            }
        }
    }
    if (active) {
        if constexpr (kBf16) {
            uint2 r;
            r.x = pack_bf16x2(acc.x, acc.y);
            r.y = pack_bf16x2(acc.z, acc.w);
            *reinterpret_cast<uint2 *>(grad_context + pix * C + li * 4) = r;
        } else {
            *reinterpret_cast<float4 *>(grad_context + pix * C + li * 4) = acc;
        }
    }
    __syncthreads();
    for (int i = tid; i < D * NG; i += kBlock) {
        const int d = i / NG, jj = i - d * NG;
        if ((s0 + jj) < HW) {
            if constexpr (kBf16) grad_depth[((int64_t)bn * D + d) * HW + s0 + jj] = (FT)(pack_bf16x2(gd[i], 0.f) & 0xFFFFu);
            else grad_depth[((int64_t)bn * D + d) * HW + s0 + jj] = gd[i];
        }
    }
}

// Points per workgroup for the chunked forward kernels: at most `max_chunk`, and such that the
// grid is (just under) a whole number of rounds of the 256 CUs x 8 resident workgroups --
// 3696 workgroups of 512 points are 1.8 rounds, i.e. the second round runs 80 % empty-handed.
int balanced_chunk_points(int64_t BP, int max_chunk) {
    const int64_t slots = 256 * 8;
    const int64_t rounds = mmt::ceil_div(BP, (int64_t)max_chunk * slots);
    int64_t cp = mmt::ceil_div(BP, rounds * slots);
    cp = (cp + 3) & ~3ll;
    if (cp > max_chunk) cp = max_chunk;
    if (cp < 64) cp = 64 < max_chunk ? 64 : max_chunk;
    return (int)cp;
}

template <int VEC>
int launch_lds_combine(const VpArgs &a, int grid, size_t lds, hipStream_t st) {
    if (VEC == 4) {
        if (a.C == 80) hipLaunchKernelGGL((vp_fwd_lds_combine<4, 20>), dim3(grid), dim3(kBlock), lds, st, a);
        else if (a.C == 64) hipLaunchKernelGGL((vp_fwd_lds_combine<4, 16>), dim3(grid), dim3(kBlock), lds, st, a);
        else hipLaunchKernelGGL((vp_fwd_lds_combine<4, 0>), dim3(grid), dim3(kBlock), lds, st, a);
    } else {
        hipLaunchKernelGGL((vp_fwd_lds_combine<1, 0>), dim3(grid), dim3(kBlock), lds, st, a);
    }
    return mmt::check_launch("voxel_pooling_forward(lds_combine)");
}

// SEG_GATHER launch for a storage type: compile-time lanes-per-row for the common channel counts
template <typename FT>
void launch_seg_gather(mmt::TimedSeq &seq, const VpArgs &a, bool big, bool fused, hipStream_t st) {
    constexpr int VEC = RowVec<FT>::VEC;
    const dim3 grid((unsigned)mmt::ceil_div(a.BP, a.nchunks)), block(kBlock);
#define MMT_LAUNCH_SEG(CVT)                                                                         \
    do {                                                                                            \
        if (fused) seq.launch(true, vp_fwd_seg_gather<FT, CVT, 512, true>, grid, block, 0, st, a);  \
        else if (big) seq.launch(true, vp_fwd_seg_gather<FT, CVT, 1024, false>, grid, block, 0, st, a); \
        else seq.launch(true, vp_fwd_seg_gather<FT, CVT, 512, false>, grid, block, 0, st, a);       \
    } while (0)
    if (a.C == 80) MMT_LAUNCH_SEG(80 / VEC);
    else if (a.C == 64) MMT_LAUNCH_SEG(64 / VEC);
    else MMT_LAUNCH_SEG(0);
#undef MMT_LAUNCH_SEG
}

// Backward for an output type (float / bf16_t): optional layout pass, optional prepare pass, main gather pass.
template <typename OT>
int backward_impl(const char *what, int B, int P, int C, int nx, int ny, const int32_t *pos_memo, const float *grad_out,
                  int64_t sb, int64_t sc, int64_t sy, int64_t sx, void *grad_in, float *workspace,
                  int64_t workspace_elems, hipStream_t st) {
    constexpr int VEC = 16 / (int)sizeof(OT);
    if (workspace == nullptr) workspace_elems = 0;
    const int64_t bev_elems = (int64_t)B * ny * nx * C;
    const int64_t BP = (int64_t)B * P;
    VpBwdArgs a;
    a.BP = BP; a.C = C; a.nx = nx; a.ny = ny;
    a.pos_memo = pos_memo; a.grad_out = grad_out; a.grad_in = grad_in;
    a.sb = sb; a.sc = sc; a.sy = sy; a.sx = sx;
    a.row_off = nullptr; a.row_off_out = nullptr;
    // armed by mmt_arm_kernel_timing (bench only): the start event rides on the first kernel of this call, the stop
    // event on the last one
    mmt::TimedSeq seq;
    const bool can_vec = C % VEC == 0 && (((uintptr_t)grad_in & 15) == 0);

    if (sc != 1 && can_vec && workspace_elems >= bev_elems) {
        dim3 grid((unsigned)mmt::ceil_div((int64_t)ny * nx, 32), (unsigned)mmt::ceil_div(C, 32), (unsigned)B);
        seq.launch(false, vp_to_channels_last, grid, dim3(kBlock), 0, st, C, ny, nx, grad_out, sb, sc, sy, sx, workspace);
        int rc = mmt::check_launch(what);
        if (rc) return rc;
        a.grad_out = workspace;
        a.sc = 1; a.sx = C; a.sy = (int64_t)nx * C; a.sb = (int64_t)ny * nx * C;
    }
    const int64_t span = (B - 1) * a.sb + (ny - 1) * a.sy + (nx - 1) * a.sx + C;
    const bool vec = can_vec && a.sc == 1 && a.sb % 4 == 0 && a.sy % 4 == 0 && a.sx % 4 == 0 &&
                     a.sb >= 0 && a.sy >= 0 && a.sx >= 0 && span < (1ll << 29) && (((uintptr_t)a.grad_out & 15) == 0);
    a.span_bytes = span * 4;
    if (vec && workspace_elems >= bev_elems + BP && (((uintptr_t)workspace & 3) == 0)) {
        // pass 1: row offsets + cache warm-up (see vp_bwd_prepare)
        a.row_off_out = reinterpret_cast<uint32_t *>(workspace + bev_elems);
        const int rows_per_xcd = (int)mmt::ceil_div(BP, 8);
        int pgrid = mmt::stream_grid(mmt::ceil_div(BP, kPrepU), kBlock, 256 * 8);
        pgrid = (pgrid + 7) & ~7;
        const int sweep = span <= 4 * bev_elems;   // a thin slice of a much wider buffer: not worth reading the whole span
        seq.launch(false, vp_bwd_prepare, dim3(pgrid), dim3(kBlock), 0, st, a, rows_per_xcd, sweep);
        int rc = mmt::check_launch(what);
        if (rc) return rc;
        a.row_off = a.row_off_out;
    }
    if (vec) {
        int grid = mmt::stream_grid(mmt::ceil_div(BP * (C / VEC), 4), kBlock, 256 * 16);
        grid = (grid + 7) & ~7;  // whole groups of 8 (one workgroup per XCD)
        if (C == 80) seq.launch(true, vp_bwd_rows_vec<OT, 80 / VEC>, dim3(grid), dim3(kBlock), 0, st, a);
        else if (C == 64) seq.launch(true, vp_bwd_rows_vec<OT, 64 / VEC>, dim3(grid), dim3(kBlock), 0, st, a);
        else seq.launch(true, vp_bwd_rows_vec<OT, 0>, dim3(grid), dim3(kBlock), 0, st, a);
        return mmt::check_launch(what);
    }
    const int grid = mmt::stream_grid(BP * C, kBlock);
    seq.launch(true, vp_bwd_strided<OT>, dim3(grid), dim3(kBlock), 0, st, a);
    return mmt::check_launch(what);
}

int forward_check(const char *what, int B, int P, int C, int nx, int ny, int nz) {
    if (B <= 0 || P <= 0 || C <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: non-positive size (B=%d P=%d C=%d grid=%dx%dx%d)", what, B, P, C, nx, ny, nz);
    if ((int64_t)B * P >= (1ll << 31) || (int64_t)B * ny * nx >= (1ll << 31))
        return mmt::fail(MMT_ERR_TOO_LARGE, "%s: B*P or B*ny*nx exceeds int32", what);
    return 0;
}

}  // namespace

extern "C" int mmt_voxel_pooling_forward_ex(int B, int P, int C, int nx, int ny, int nz,
                                            const int32_t *geom, const float *feats, float *out,
                                            int32_t *pos_memo, int flags, void *stream) {
    MMT_REQUIRE_PTR(geom);
    MMT_REQUIRE_PTR(feats);
    MMT_REQUIRE_PTR(out);
    MMT_REQUIRE_PTR(pos_memo);
    if (int rc = forward_check("voxel_pooling_forward", B, P, C, nx, ny, nz)) return rc;
    const int64_t BP = (int64_t)B * P;
    int algo = flags & MMT_VP_ALGO_MASK;
    if (algo > MMT_VP_ALGO_STREAM)
        return mmt::fail(MMT_ERR_BAD_FLAG, "voxel_pooling_forward: unknown algorithm %d", algo);
    // MMT_VP_WAVE_PER_SLOT (the pre-ABI-3 gather schedule, kept for A/B runs until ABI 4) is accepted and ignored
    if (flags & ~(MMT_VP_ALGO_MASK | MMT_VP_WRITE_DROPPED | MMT_VP_CHUNK_1024 | MMT_VP_WAVE_PER_SLOT | MMT_VP_CHUNK_POINTS_MASK))
        return mmt::fail(MMT_ERR_BAD_FLAG, "voxel_pooling_forward: unknown flag bits 0x%x", flags);
    const int chunk_points = ((flags & MMT_VP_CHUNK_POINTS_MASK) >> 8) * 4;
    if (chunk_points != 0 && (chunk_points < 64 || chunk_points > 512 || (flags & MMT_VP_CHUNK_1024)))
        return mmt::fail(MMT_ERR_BAD_FLAG, "voxel_pooling_forward: MMT_VP_CHUNK_POINTS must be 64..512 (got %d) and excludes MMT_VP_CHUNK_1024", chunk_points);
    hipStream_t st = (hipStream_t)stream;

    VpArgs a;
    a.BP = BP; a.P = P; a.C = C; a.nx = nx; a.ny = ny; a.nz = nz;
    a.geom = geom; a.feats = feats; a.out = out; a.pos_memo = pos_memo;
    a.write_dropped = (flags & MMT_VP_WRITE_DROPPED) ? 1 : 0;
    a.nslot = 0; a.nchunks = 0;
    a.depth = nullptr; a.context = nullptr; a.DHW = 1; a.HW = 1;
    // float4 paths need 16-byte aligned rows
    const bool vec4 = (C % 4 == 0) && (((uintptr_t)feats & 15) == 0);
    const bool seg_ok = vec4 && C <= 256;
    if (algo == MMT_VP_ALGO_AUTO) algo = seg_ok ? MMT_VP_ALGO_SEG_GATHER : MMT_VP_ALGO_LDS_ATOMIC;
    if ((algo == MMT_VP_ALGO_SEG_GATHER || algo == MMT_VP_ALGO_STREAM) && !seg_ok) algo = MMT_VP_ALGO_LDS_ATOMIC;

    if (algo == MMT_VP_ALGO_SEG_GATHER) {
        const bool big = (flags & MMT_VP_CHUNK_1024) != 0;
        const int chunk = big ? 1024 : (chunk_points ? chunk_points : balanced_chunk_points(BP, 512));
        a.nchunks = chunk;
        mmt::TimedSeq seq;   // armed by mmt_arm_kernel_timing (bench only)
        launch_seg_gather<float>(seq, a, big, false, (hipStream_t)stream);
        return mmt::check_launch("voxel_pooling_forward(seg_gather)");
    }
    { hipEvent_t t0, t1; mmt::take_timing_events(&t0, &t1); }   // the other algorithms consume and ignore an armed timing

    if (algo == MMT_VP_ALGO_STREAM) {
        const bool big = (flags & MMT_VP_CHUNK_1024) != 0;
        const int chunk = big ? 1024 : 512;
        const int64_t nchunks = mmt::ceil_div(BP, chunk);
        const dim3 grid((unsigned)nchunks), block(kBlock);
        // rowbuf + tab_key + tab_slot + slot_key + slot_cnt + slot_off + sorted + sorted_slot + stage
        const size_t lds = (size_t)kStreamRowCap * C * 4 + (size_t)chunk * 2 * 4 + (size_t)chunk * 2 * 2 +
                           (size_t)chunk * 4 + (size_t)chunk * 2 + (size_t)(chunk + 8) * 2 + (size_t)chunk * 2 * 2 +
                           (size_t)(kBlock / 64) * 256 * 4 + 64;
#define MMT_LAUNCH_STREAM(C4T)                                                                    \
    do {                                                                                          \
        if (big) hipLaunchKernelGGL((vp_fwd_stream<C4T, 1024>), grid, block, lds, st, a);         \
        else hipLaunchKernelGGL((vp_fwd_stream<C4T, 512>), grid, block, lds, st, a);              \
    } while (0)
        if (C == 80) MMT_LAUNCH_STREAM(20);
        else if (C == 64) MMT_LAUNCH_STREAM(16);
        else MMT_LAUNCH_STREAM(0);
#undef MMT_LAUNCH_STREAM
        return mmt::check_launch("voxel_pooling_forward(stream)");
    }

    if (algo == MMT_VP_ALGO_ROW_ATOMIC) {
        const int64_t work = BP * (vec4 ? C / 4 : C);
        const int grid = mmt::stream_grid(work, kBlock);
        if (vec4) hipLaunchKernelGGL((vp_fwd_row_atomic<4>), dim3(grid), dim3(kBlock), 0, st, a);
        else hipLaunchKernelGGL((vp_fwd_row_atomic<1>), dim3(grid), dim3(kBlock), 0, st, a);
        return mmt::check_launch("voxel_pooling_forward(row_atomic)");
    }

    // LDS_ATOMIC. LDS budget: ~40 KiB of BEV rows per workgroup -> 3 workgroups per CU.
    int nslot = (40 * 1024) / (C * 4);
    if (nslot > 256) nslot = 256;
    if (nslot < 4) {  // rows too long for the LDS tile: use plain row atomics
        return mmt_voxel_pooling_forward_ex(B, P, C, nx, ny, nz, geom, feats, out, pos_memo,
                                            (flags & ~MMT_VP_ALGO_MASK) | MMT_VP_ALGO_ROW_ATOMIC, stream);
    }
    a.nslot = nslot;
    a.nchunks = (int)mmt::ceil_div(BP, kChunk);
    const size_t lds = (size_t)nslot * C * 4 + (size_t)(2 * kHashSize + nslot + kChunk + 4) * 4;
    const int grid = a.nchunks < 256 * 32 ? a.nchunks : 256 * 32;
    return vec4 ? launch_lds_combine<4>(a, grid, lds, st) : launch_lds_combine<1>(a, grid, lds, st);
}

extern "C" int mmt_voxel_pooling_forward(int B, int P, int C, int nx, int ny, int nz,
                                         const int32_t *geom, const float *feats, float *out,
                                         int32_t *pos_memo, void *stream) {
    return mmt_voxel_pooling_forward_ex(B, P, C, nx, ny, nz, geom, feats, out, pos_memo,
                                        MMT_VP_ALGO_AUTO, stream);
}

extern "C" int64_t mmt_voxel_pooling_backward_workspace_elems(int B, int P, int C, int nx, int ny) {
    if (B <= 0 || P <= 0 || C <= 0 || nx <= 0 || ny <= 0) return 0;
    return (int64_t)B * ny * nx * C + (int64_t)B * P;
}

extern "C" int mmt_voxel_pooling_backward(int B, int P, int C, int nx, int ny,
                                          const int32_t *pos_memo, const float *grad_out,
                                          int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                                          float *grad_in, float *workspace,
                                          int64_t workspace_elems, void *stream) {
    MMT_REQUIRE_PTR(pos_memo);
    MMT_REQUIRE_PTR(grad_out);
    MMT_REQUIRE_PTR(grad_in);
    if (B <= 0 || P <= 0 || C <= 0 || nx <= 0 || ny <= 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "voxel_pooling_backward: non-positive size");
    if ((int64_t)B * P >= (1ll << 31)) return mmt::fail(MMT_ERR_TOO_LARGE, "voxel_pooling_backward: B*P exceeds int32");
    return backward_impl<float>("voxel_pooling_backward", B, P, C, nx, ny, pos_memo, grad_out, sb, sc, sy, sx, grad_in,
                                workspace, workspace_elems, (hipStream_t)stream);
}

// ---- bf16 feature storage (SURVEY 5.6 / BASELINE configs[4]): bf16 rows in, fp32 accumulate, fp32 BEV out;
// the backward rounds the gathered fp32 gradient rows to bf16 (nearest even) on the way out.
extern "C" int mmt_voxel_pooling_forward_bf16(int B, int P, int C, int nx, int ny, int nz,
                                              const int32_t *geom, const uint16_t *feats, float *out,
                                              int32_t *pos_memo, int flags, void *stream) {
    MMT_REQUIRE_PTR(geom);
    MMT_REQUIRE_PTR(feats);
    MMT_REQUIRE_PTR(out);
    MMT_REQUIRE_PTR(pos_memo);
    if (int rc = forward_check("voxel_pooling_forward_bf16", B, P, C, nx, ny, nz)) return rc;
    if (C % 8 != 0 || C > 512 || (((uintptr_t)feats & 15) != 0))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "voxel_pooling_forward_bf16: needs C %% 8 == 0, C <= 512 and 16-byte aligned rows (C=%d)", C);
    if (flags & ~(MMT_VP_WRITE_DROPPED | MMT_VP_CHUNK_POINTS_MASK))
        return mmt::fail(MMT_ERR_BAD_FLAG, "voxel_pooling_forward_bf16: unknown flag bits 0x%x (only WRITE_DROPPED / CHUNK_POINTS)", flags);
    const int chunk_points = ((flags & MMT_VP_CHUNK_POINTS_MASK) >> 8) * 4;
    if (chunk_points != 0 && (chunk_points < 64 || chunk_points > 512))
        return mmt::fail(MMT_ERR_BAD_FLAG, "voxel_pooling_forward_bf16: MMT_VP_CHUNK_POINTS must be 64..512 (got %d)", chunk_points);
    VpArgs a;
    a.BP = (int64_t)B * P; a.P = P; a.C = C; a.nx = nx; a.ny = ny; a.nz = nz;
    a.geom = geom; a.feats = feats; a.out = out; a.pos_memo = pos_memo;
    a.write_dropped = (flags & MMT_VP_WRITE_DROPPED) ? 1 : 0;
    a.nslot = 0;
    a.depth = nullptr; a.context = nullptr; a.DHW = 1; a.HW = 1;
    a.nchunks = chunk_points ? chunk_points : balanced_chunk_points(a.BP, 512);
    mmt::TimedSeq seq;
    launch_seg_gather<bf16_t>(seq, a, false, false, (hipStream_t)stream);
    return mmt::check_launch("voxel_pooling_forward_bf16");
}

extern "C" int mmt_voxel_pooling_backward_bf16(int B, int P, int C, int nx, int ny,
                                               const int32_t *pos_memo, const float *grad_out,
                                               int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                                               uint16_t *grad_in, float *workspace,
                                               int64_t workspace_elems, void *stream) {
    MMT_REQUIRE_PTR(pos_memo);
    MMT_REQUIRE_PTR(grad_out);
    MMT_REQUIRE_PTR(grad_in);
    if (B <= 0 || P <= 0 || C <= 0 || nx <= 0 || ny <= 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "voxel_pooling_backward_bf16: non-positive size");
    if ((int64_t)B * P >= (1ll << 31)) return mmt::fail(MMT_ERR_TOO_LARGE, "voxel_pooling_backward_bf16: B*P exceeds int32");
    return backward_impl<bf16_t>("voxel_pooling_backward_bf16", B, P, C, nx, ny, pos_memo, grad_out, sb, sc, sy, sx, grad_in,
                                 workspace, workspace_elems, (hipStream_t)stream);
}

namespace {

template <typename FT>
int lift_splat_forward_impl(const char *what, int B, int N, int D, int HW, int C, int nx, int ny, int nz, const int32_t *geom,
                            const FT *depth, const FT *context, float *out, int32_t *pos_memo, int flags, hipStream_t st) {
    constexpr int VEC = RowVec<FT>::VEC;
    if (B <= 0 || N <= 0 || D <= 0 || HW <= 0 || C <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: non-positive size", what);
    if (C % VEC != 0 || C > 64 * VEC || (((uintptr_t)context & 15) != 0))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: needs C %% %d == 0, C <= %d and a 16-byte aligned context", what, VEC, 64 * VEC);
    const int64_t P = (int64_t)N * D * HW, BP = (int64_t)B * P;
    if (BP >= (1ll << 31) || (int64_t)B * ny * nx >= (1ll << 31) || (int64_t)B * N * HW * C >= (1ll << 31))
        return mmt::fail(MMT_ERR_TOO_LARGE, "%s: index range exceeds int32", what);
    if (flags & ~MMT_VP_WRITE_DROPPED) return mmt::fail(MMT_ERR_BAD_FLAG, "%s: unknown flag bits 0x%x", what, flags);
    VpArgs a;
    a.BP = BP; a.P = (int)P; a.C = C; a.nx = nx; a.ny = ny; a.nz = nz;
    a.geom = geom; a.feats = nullptr; a.out = out; a.pos_memo = pos_memo;
    a.write_dropped = (flags & MMT_VP_WRITE_DROPPED) ? 1 : 0;
    a.nslot = 0;
    a.depth = depth; a.context = context; a.DHW = D * HW; a.HW = HW;
    a.nchunks = balanced_chunk_points(BP, 512);
    mmt::TimedSeq seq;   // armed by mmt_arm_kernel_timing (bench only)
    launch_seg_gather<FT>(seq, a, false, true, st);
    return mmt::check_launch(what);
}

template <typename FT>
int lift_splat_backward_impl(const char *what, int B, int N, int D, int HW, int C, int nx, int ny, const int32_t *pos_memo,
                             const FT *depth, const FT *context, const float *grad_out, int64_t sb, int64_t sc, int64_t sy,
                             int64_t sx, FT *grad_depth, FT *grad_context, hipStream_t st) {
    if (B <= 0 || N <= 0 || D <= 0 || HW <= 0 || C <= 0 || nx <= 0 || ny <= 0 || B * N > 65535)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: bad sizes", what);
    if (C % 16 != 0 || C > 256)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: needs C %% 16 == 0 and C <= 256", what);
    const int64_t span = (B - 1) * sb + (ny - 1) * sy + (nx - 1) * sx + C;
    if (sc != 1 || sb % 4 || sy % 4 || sx % 4 || sb < 0 || sy < 0 || sx < 0 || span >= (1ll << 29) ||
        (((uintptr_t)grad_out | (uintptr_t)context | (uintptr_t)grad_context) & 15) != 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: grad_out must be channels-last (stride_c == 1), 16-byte aligned", what);
    if ((int64_t)B * N * D * HW >= (1ll << 31)) return mmt::fail(MMT_ERR_TOO_LARGE, "%s: B*P exceeds int32", what);
    const int C4 = C / 4;
    const int NG = (kBlock / 64) * (64 / C4);
    const size_t lds = (size_t)D * NG * 4;
    if (lds > 64 * 1024) return mmt::fail(MMT_ERR_TOO_LARGE, "%s: D too large for the LDS tile", what);
    dim3 grid((unsigned)mmt::ceil_div(HW, NG), (unsigned)(B * N));
    // pos_memo's batch index b is the SAMPLE index; cameras of one sample share it
    mmt::TimedSeq seq;   // armed by mmt_arm_kernel_timing (bench only)
#define MMT_LSB(C4T) seq.launch(true, lift_splat_backward_kernel<FT, C4T>, grid, dim3(kBlock), lds, st, D, HW, C, pos_memo, depth, context, grad_out, sb, sy, sx, (int64_t)(span * 4), grad_depth, grad_context)
    if (C == 80) MMT_LSB(20);
    else if (C == 64) MMT_LSB(16);
    else MMT_LSB(0);
#undef MMT_LSB
    return mmt::check_launch(what);
}

}  // namespace

extern "C" int mmt_lift_splat_forward(int B, int N, int D, int HW, int C, int nx, int ny, int nz,
                                      const int32_t *geom, const float *depth, const float *context,
                                      float *out, int32_t *pos_memo, int flags, void *stream) {
    MMT_REQUIRE_PTR(geom);
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(out);
    MMT_REQUIRE_PTR(pos_memo);
    return lift_splat_forward_impl<float>("lift_splat_forward", B, N, D, HW, C, nx, ny, nz, geom, depth, context, out, pos_memo,
                                          flags, (hipStream_t)stream);
}

extern "C" int mmt_lift_splat_forward_bf16(int B, int N, int D, int HW, int C, int nx, int ny, int nz,
                                           const int32_t *geom, const uint16_t *depth, const uint16_t *context,
                                           float *out, int32_t *pos_memo, int flags, void *stream) {
    MMT_REQUIRE_PTR(geom);
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(out);
    MMT_REQUIRE_PTR(pos_memo);
    return lift_splat_forward_impl<bf16_t>("lift_splat_forward_bf16", B, N, D, HW, C, nx, ny, nz, geom, depth, context, out,
                                           pos_memo, flags, (hipStream_t)stream);
}

extern "C" int mmt_lift_splat_backward(int B, int N, int D, int HW, int C, int nx, int ny,
                                       const int32_t *pos_memo, const float *depth,
                                       const float *context, const float *grad_out, int64_t sb,
                                       int64_t sc, int64_t sy, int64_t sx, float *grad_depth,
                                       float *grad_context, void *stream) {
    MMT_REQUIRE_PTR(pos_memo);
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(grad_out);
    MMT_REQUIRE_PTR(grad_depth);
    MMT_REQUIRE_PTR(grad_context);
    return lift_splat_backward_impl<float>("lift_splat_backward", B, N, D, HW, C, nx, ny, pos_memo, depth, context, grad_out,
                                           sb, sc, sy, sx, grad_depth, grad_context, (hipStream_t)stream);
}

extern "C" int mmt_lift_splat_backward_bf16(int B, int N, int D, int HW, int C, int nx, int ny,
                                            const int32_t *pos_memo, const uint16_t *depth,
                                            const uint16_t *context, const float *grad_out, int64_t sb,
                                            int64_t sc, int64_t sy, int64_t sx, uint16_t *grad_depth,
                                            uint16_t *grad_context, void *stream) {
    MMT_REQUIRE_PTR(pos_memo);
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(grad_out);
    MMT_REQUIRE_PTR(grad_depth);
    MMT_REQUIRE_PTR(grad_context);
    return lift_splat_backward_impl<bf16_t>("lift_splat_backward_bf16", B, N, D, HW, C, nx, ny, pos_memo, depth, context,
                                            grad_out, sb, sc, sy, sx, grad_depth, grad_context, (hipStream_t)stream);
}
