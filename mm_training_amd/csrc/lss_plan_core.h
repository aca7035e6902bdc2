// Output-stationary plan of the fused lift-splat forward (SURVEY section 8 rows f1 + f3; lss_fpn.py:441-464 with the cells of
// :328-361 / :461-462 known in advance): the INTEGER logic that turns a calibration's column summary into jobs.
//
// For a fixed calibration (a sample's camera matrices, the frustum axes, the grid) the BEV cell of every frustum point is
// fixed, so which points feed a cell can be worked out once.  The plan groups a cell's points into RUNS -- up to kRunBins
// consecutive depth bins of one 16-row block of one image column that share the cell (16-bit row masks: z range, and for a
// camera that is not level the rows that fall into THIS cell) -- and cuts the BEV map into JOBS: a contiguous range of the
// cells of one 8 x 8 tile with at most kMaxRuns runs.  The forward kernel (lift_splat_plan.hip) gives a job to a workgroup:
// lane groups take the job's (column, row block) PAIRS -- context rows in registers, one partial row per run into LDS --
// and then every cell's partial rows are summed in plan order and STORED.  No zero fill, no atomics, bit-reproducible.
//
// Everything here is plain index arithmetic written as phases of strided loops (`for (i = tid; i < n; i += nthreads)`), so
// the same source runs as one workgroup per sample on the device (phases separated by __syncthreads()) and single-threaded
// on the host, where tests/test_lss_plan_core.py builds it with g++ and checks the plan against the oracle's geometry.  The
// geometry itself (summary, cells of the rows of a mixed block) is the device's (mmt_camera.h) and enters through `RowCells`.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PLAN_HD __host__ __device__ __forceinline__
#else
#define PLAN_HD static inline
#endif

namespace mmt {
namespace plan {

constexpr int kTile = 8;                 // BEV tile edge in cells
constexpr int kTileCells = kTile * kTile;
#ifndef PLAN_MAX_RUNS
#define PLAN_MAX_RUNS 96
#endif
#ifndef PLAN_MAX_PAIR_RUNS
#define PLAN_MAX_PAIR_RUNS 4
#endif
constexpr int kMaxRuns = PLAN_MAX_RUNS;  // runs (= partial rows in LDS) per job (<= 255: a record's cell_begin is bytes)
constexpr int kMaxPairRuns = PLAN_MAX_PAIR_RUNS;   // runs per pair record (4: a wave runs as long as the longest of its four pairs -- 6 / 8 / 12 measured 4 % slower at BASELINE configs[3])
constexpr int kRunBins = 4;              // depth bins per run
constexpr int kWindowBins = 8;           // depth bins a lane group loads per PAIR and image row (one 16-byte load in bf16, two in fp32): every run of
                                         // the pair takes its bins out of that window (round 5 loaded four bins per RUN: 2.3 runs per pair at BASELINE
                                         // configs[4], every load a cache line per image row -- 12 of the kernel's 46 us there)
constexpr int kGroups = 8;               // record groups of a plan: neighbouring tiles (one XCD's share at batch 1), heaviest tiles first inside
constexpr int kSummaryUniformBit = 0x10000;   // = mmt::kSummaryUniform (mmt_camera.h)

// job record: header | cell_begin[kTileCells + 1] u8 | pairs[kMaxRuns] | runs[kMaxRuns]
constexpr int kJobCellBeginOff = 16;
constexpr int kJobPairsOff = 96;
constexpr int kJobRunsOff = kJobPairsOff + 8 * kMaxRuns;
constexpr int kJobBytes = kJobRunsOff + 24 * kMaxRuns;      // 3168 = 198 x 16

struct Dims {
    int N, D, fH, fW, nb;                // cameras, depth bins, image rows / columns of the feature map, 16-row blocks per column
    int nx, ny;                          // BEV grid
    int tiles_x, tiles_y, ntiles;        // 8 x 8 tiles (the last ones may overhang the grid)
    int ncells_tm;                       // ntiles * 64: cells in TILE-MAJOR order (tile * 64 + the cell's place on the tile's Z curve)
    int strips;                          // N * nb * fW: (camera, row block, column) = the summary's first three axes
    int runs_cap, jobs_cap;              // capacity of the slot / scratch arrays; beyond them a calibration is "unplannable"
};

PLAN_HD void make_dims(int N, int D, int fH, int fW, int nx, int ny, Dims *d, int runs_cap = 0) {
    d->N = N; d->D = D; d->fH = fH; d->fW = fW; d->nb = (fH + 15) / 16;
    d->nx = nx; d->ny = ny;
    d->tiles_x = (nx + kTile - 1) / kTile; d->tiles_y = (ny + kTile - 1) / kTile;
    d->ntiles = d->tiles_x * d->tiles_y;
    d->ncells_tm = d->ntiles * kTileCells;
    d->strips = N * d->nb * fW;
    // a level rig has at most one run per (strip, bin) -- fewer: consecutive bins share cells; a rig that is not level adds a
    // run per extra cell of a mixed block.  Twice the level bound covers every rig met so far (the reference's nuScenes
    // calibration has 4 % mixed blocks); what exceeds it is served by the forward's brute-force path.
    const long long rc = 2ll * d->strips * D;
    d->runs_cap = runs_cap > 0 ? runs_cap : (int)(rc < 64 ? 64 : rc);       // (an explicit capacity: tests)
    // a tile's greedy cut closes a job only when the next cell would overflow it, so two consecutive jobs of a tile hold more
    // than kMaxRuns runs together: jobs <= ntiles + 2 * runs / kMaxRuns; a cell with more runs than a job holds becomes a CHAIN
    // of records, at most one of them not full: + runs / kMaxRuns
    d->jobs_cap = d->ntiles + 3 * (d->runs_cap / kMaxRuns) + 2;
}

// limits of the record fields (checked by the entry points): col 16 bits, row block 5 bits, bin 11 bits, cells 28 bits
PLAN_HD bool dims_ok(const Dims &d) {
    return d.N > 0 && d.D >= kRunBins && d.D <= 2047 && d.fH > 0 && d.fW > 0 && d.nb <= 32 && (long long)d.N * d.fW <= 65535 && d.nx > 0 && d.ny > 0 &&
           d.nx <= 32767 && d.ny <= 32767 && (long long)d.ntiles * kTileCells < (1ll << 28) && 2ll * d.strips * d.D < (1ll << 30);
}

struct RunTmp {                          // a run on its way into the plan (build scratch), 16 bytes
    uint32_t cell_len;                   // tile-major cell | (bins - 1) << 28 | shift << 30 (bins counted from the LOADED bin, below)
    uint32_t key;                        // column << 16 | row block << 11 | first bin: the order of a cell's partial sums
    uint16_t mask[kRunBins];             // bit i: row i of the block at bin (first - shift + j) belongs to this run
};
struct JobDesc { int32_t tile, c0, ncells, run_begin, nruns, chain, pad1, pad2; };   // build scratch, 32 bytes
// chain: a cell fed by more runs than a job holds is a chain of consecutive records for that ONE cell; the workgroup that meets
// the head walks the whole chain and sums the records' partial rows in order; a workgroup that meets a link skips it
constexpr uint32_t kChainNone = 0, kChainHead = 1, kChainLink = 2, kChainMore = 4;      // kChainMore: the next record continues this chain
struct JobHeader { uint16_t ncells, npairs, nruns, c0; int32_t tile; uint32_t chain; };
struct PairRec { uint16_t col; uint8_t rb; uint8_t nruns; uint16_t run0; uint16_t w0; };         // col = camera * fW + column; w0: first bin of the pair's depth window (even)
struct RunRec { uint16_t d0; uint8_t len; uint8_t pslot; uint32_t cell_local; uint64_t wlo, whi; };   // pslot: partial row; cell_local: index in the job;
// wlo / whi: eight 16-bit row masks, bits [16 k, 16 k + 16) = the rows of the block that take window bin w0 + k (d0 / len: the run's loaded bins, for the record)
static_assert(sizeof(PairRec) == 8 && sizeof(RunRec) == 24, "record layout");

// A tile's 64 cells are numbered along a Z curve (x bits at the even, y bits at the odd positions), so that a job -- a range of
// consecutive cells -- is a compact block (2 x 2, 4 x 2, 4 x 4, ...) and not a strip: a column's ray then crosses several of
// the job's cells and its 16 context rows, loaded once per job, serve them all (near the cameras, where a cell is fed by a
// dozen columns and a job holds a handful of cells, a row-major strip is crossed in one or two cells only).
PLAN_HD int z_of(int lx, int ly) {
    return (lx & 1) | ((ly & 1) << 1) | ((lx & 2) << 1) | ((ly & 2) << 2) | ((lx & 4) << 2) | ((ly & 4) << 3);
}
PLAN_HD int z_x(int l) { return (l & 1) | ((l >> 1) & 2) | ((l >> 2) & 4); }
PLAN_HD int z_y(int l) { return ((l >> 1) & 1) | ((l >> 2) & 2) | ((l >> 3) & 4); }
PLAN_HD int cell_tm(const Dims &d, int x, int y) {
    return ((y / kTile) * d.tiles_x + x / kTile) * kTileCells + z_of(x % kTile, y % kTile);
}
PLAN_HD void cell_xy(const Dims &d, int ctm, int *x, int *y) {
    const int tile = ctm / kTileCells, l = ctm % kTileCells;
    *x = (tile % d.tiles_x) * kTile + z_x(l);
    *y = (tile / d.tiles_x) * kTile + z_y(l);
}

// Build scratch of ONE sample (device: a slice of the plan cache; host: plain arrays)
struct Scratch {
    int32_t *cell_off;                   // [ncells_tm + 1]  counts, then exclusive prefix
    int32_t *cursor;                     // [ncells_tm]
    RunTmp *runs;                        // [runs_cap], sorted by (cell, key) when phase_sort_cells is through
    int32_t *tile_jobs;                  // [ntiles + 1] records per tile
    int32_t *tile_perm;                  // [ntiles] the tiles in the order their records are laid out: eight groups of neighbouring tiles, heaviest first inside a group
    int32_t *perm_jobs;                  // [ntiles + 1] records per tile in that order, then exclusive prefix
    int32_t *tile_base;                  // [ntiles] first record of a tile
    JobDesc *jobs;                       // [jobs_cap]
    uint8_t *order;                      // [jobs_cap * kMaxRuns]
    int32_t *partial;                    // [nthreads_max] scan partials
    int32_t *status;                     // [16]: nruns, njobs, unplannable (0 / 1), first record of group g at [3 + g] (g = 0..kGroups)
};

PLAN_HD long long scratch_bytes(const Dims &d, int nthreads_max) {
    auto up = [](long long v) { return (v + 255) & ~255ll; };
    return up(4ll * (d.ncells_tm + 1)) + up(4ll * d.ncells_tm) + up(16ll * d.runs_cap) + 4 * up(4ll * (d.ntiles + 1)) + up(32ll * d.jobs_cap) +
           up((long long)d.jobs_cap * kMaxRuns) + up(4ll * nthreads_max) + 256;
}
PLAN_HD void scratch_carve(const Dims &d, int nthreads_max, void *base, Scratch *s) {
    auto up = [](long long v) { return (v + 255) & ~255ll; };
    char *p = static_cast<char *>(base);
    s->cell_off = reinterpret_cast<int32_t *>(p); p += up(4ll * (d.ncells_tm + 1));
    s->cursor = reinterpret_cast<int32_t *>(p); p += up(4ll * d.ncells_tm);
    s->runs = reinterpret_cast<RunTmp *>(p); p += up(16ll * d.runs_cap);
    s->tile_jobs = reinterpret_cast<int32_t *>(p); p += up(4ll * (d.ntiles + 1));
    s->tile_perm = reinterpret_cast<int32_t *>(p); p += up(4ll * (d.ntiles + 1));
    s->perm_jobs = reinterpret_cast<int32_t *>(p); p += up(4ll * (d.ntiles + 1));
    s->tile_base = reinterpret_cast<int32_t *>(p); p += up(4ll * (d.ntiles + 1));
    s->jobs = reinterpret_cast<JobDesc *>(p); p += up(32ll * d.jobs_cap);
    s->order = reinterpret_cast<uint8_t *>(p); p += up((long long)d.jobs_cap * kMaxRuns);
    s->partial = reinterpret_cast<int32_t *>(p); p += up(4ll * nthreads_max);
    s->status = reinterpret_cast<int32_t *>(p);
}

#if defined(__HIP_DEVICE_COMPILE__)
PLAN_HD int plan_atomic_inc(int32_t *p) { return atomicAdd(p, 1); }
#else
PLAN_HD int plan_atomic_inc(int32_t *p) { return (*p)++; }
#endif

// ---- runs of one strip (camera, row block, column): consecutive bins whose kept rows share a cell ----------------------------
// summary: int32 pairs [strips * D] of ONE sample (mmt_camera.h `column summary`): [0] = (y << 16 | x) of the block's first
// row or -1, [1] = z mask | kSummaryUniformBit.  A block without the bit is MIXED: rc(strip, bin, row) gives the packed cell
// of each of its rows (or -1 outside the grid) and the block yields one single-bin run per distinct cell.
// sink(col, rb, d0, len, packed_cell, masks) -- masks: the run's (up to) four 16-bit row masks packed into 64 bits, first bin lowest
template <class RowCells, class Sink>
PLAN_HD void emit_strip(const Dims &d, const int32_t *summary, int s, RowCells &rc, Sink &sink) {
    const int w = s % d.fW, rb = (s / d.fW) % d.nb, n = s / (d.fW * d.nb);
    const int col = n * d.fW + w;
    int cur = -1, cur_d0 = 0, cur_len = 0;
    uint16_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    auto flush = [&]() {
        if (cur >= 0 && cur_len > 0) sink(col, rb, cur_d0, cur_len, cur, (uint64_t)m0 | ((uint64_t)m1 << 16) | ((uint64_t)m2 << 32) | ((uint64_t)m3 << 48));
        cur = -1; cur_len = 0; m0 = m1 = m2 = m3 = 0;
    };
    for (int bin = 0; bin < d.D; ++bin) {
        const int32_t sx = summary[((long long)s * d.D + bin) * 2], sy = summary[((long long)s * d.D + bin) * 2 + 1];
        const unsigned zm = (unsigned)sy & 0xFFFFu;
        if (sy & kSummaryUniformBit) {
            const int cell = (zm != 0u && sx >= 0) ? sx : -1;
            if (cell < 0) { flush(); continue; }
            if (cell != cur || cur_len >= kRunBins) { flush(); cur = cell; cur_d0 = bin; }
            if (cur_len == 0) m0 = (uint16_t)zm; else if (cur_len == 1) m1 = (uint16_t)zm; else if (cur_len == 2) m2 = (uint16_t)zm; else m3 = (uint16_t)zm;
            ++cur_len;
        } else {
            flush();
            int cells[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) cells[i] = ((zm >> i) & 1u) ? rc(s, bin, i) : -1;
            unsigned left = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) left |= cells[i] >= 0 ? (1u << i) : 0u;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if ((left >> i) & 1u) {
                    unsigned m = 0;
#pragma unroll
                    for (int j = 0; j < 16; ++j) m |= (j >= i && ((left >> j) & 1u) && cells[j] == cells[i]) ? (1u << j) : 0u;
                    left &= ~m;
                    sink(col, rb, bin, 1, cells[i], (uint64_t)(m & 0xFFFFu));
                }
            }
        }
    }
    flush();
}

// phase 0: clear the counters
PLAN_HD void phase_clear(const Dims &d, Scratch &s, int tid, int nt) {
    for (int i = tid; i <= d.ncells_tm; i += nt) s.cell_off[i] = 0;
    for (int i = tid; i < d.ncells_tm; i += nt) s.cursor[i] = 0;
    if (tid == 0) for (int i = 0; i < 16; ++i) s.status[i] = 0;
}
// phase 1: runs per cell
template <class RowCells>
PLAN_HD void phase_count(const Dims &d, Scratch &s, const int32_t *summary, RowCells &rc, int tid, int nt) {
    for (int st = tid; st < d.strips; st += nt) {
        auto sink = [&](int, int, int, int, int cell, uint64_t) { plan_atomic_inc(&s.cell_off[cell_tm(d, cell & 0xFFFF, cell >> 16)]); };
        emit_strip(d, summary, st, rc, sink);
    }
}
// exclusive prefix of a[0..n) in place, a[n] = the total: three sub-phases (a barrier between them on the device)
PLAN_HD void scan_a(int32_t *a, int n, int32_t *partial, int tid, int nt) {
    const int ch = (n + nt - 1) / nt;
    int sum = 0;
    for (int i = tid * ch; i < n && i < (tid + 1) * ch; ++i) sum += a[i];
    partial[tid] = sum;
}
PLAN_HD void scan_b(int32_t *a, int n, int32_t *partial, int tid, int nt) {
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < nt; ++t) { const int v = partial[t]; partial[t] = run; run += v; }
        a[n] = run;
    }
}
PLAN_HD void scan_c(int32_t *a, int n, int32_t *partial, int tid, int nt) {
    const int ch = (n + nt - 1) / nt;
    int run = partial[tid];
    for (int i = tid * ch; i < n && i < (tid + 1) * ch; ++i) { const int v = a[i]; a[i] = run; run += v; }
}
// phase 3 (after the scan of cell_off): too many runs?  (every thread reads the same verdict afterwards)
PLAN_HD void phase_check_runs(const Dims &d, Scratch &s, int tid) {
    if (tid == 0) {
        s.status[0] = s.cell_off[d.ncells_tm];
        if (s.cell_off[d.ncells_tm] > d.runs_cap) s.status[2] = 1;
    }
}
// phase 4: the runs into their cells' ranges (order inside a cell: whatever the atomics give; phase 5 sorts it)
template <class RowCells>
PLAN_HD void phase_place(const Dims &d, Scratch &s, const int32_t *summary, RowCells &rc, int tid, int nt) {
    for (int st = tid; st < d.strips; st += nt) {
        auto sink = [&](int col, int rb, int d0, int len, int cell, uint64_t mm) {
            const int c = cell_tm(d, cell & 0xFFFF, cell >> 16);
            const int pos = s.cell_off[c] + plan_atomic_inc(&s.cursor[c]);
            // the kernel loads the kRunBins depths [d0 - sh, d0 - sh + kRunBins) of every row: a run near the end of the ray is
            // shifted back so that the load stays inside the pixel's D bins (masks move with it; the key keeps the true bin)
            const int sh = d0 + kRunBins > d.D ? d0 + kRunBins - d.D : 0;
            RunTmp r;
            r.cell_len = (uint32_t)c | ((uint32_t)(len + sh - 1) << 28) | ((uint32_t)sh << 30);
            r.key = ((uint32_t)col << 16) | ((uint32_t)rb << 11) | (uint32_t)d0;
            const uint64_t ms = mm << (16 * sh);          // (sh <= 3; the bins shifted out are past the run's length: zero masks)
            r.mask[0] = (uint16_t)ms; r.mask[1] = (uint16_t)(ms >> 16); r.mask[2] = (uint16_t)(ms >> 32); r.mask[3] = (uint16_t)(ms >> 48);
            s.runs[pos] = r;
        };
        emit_strip(d, summary, st, rc, sink);
    }
}
// phase 5: a cell's runs by key (camera, column, row block, first bin): the order its partial rows are summed in, the
// same on every build of the same calibration
PLAN_HD void phase_sort_cells(const Dims &d, Scratch &s, int tid, int nt) {
    for (int c = tid; c < d.ncells_tm; c += nt) {
        const int lo = s.cell_off[c], hi = s.cell_off[c + 1];
        for (int i = lo + 1; i < hi; ++i) {
            const RunTmp r = s.runs[i];
            int j = i - 1;
            while (j >= lo && s.runs[j].key > r.key) { s.runs[j + 1] = s.runs[j]; --j; }
            s.runs[j + 1] = r;
        }
    }
}
// the greedy cut of one tile: jobs = contiguous cell ranges with at most kMaxRuns runs (an empty tile is one job that
// stores zeros); a cell with more runs than that is a chain of single-cell records.  emit(job, c0, ncells, run_begin, nruns, chain).
// Returns the number of records.
template <class Emit>
PLAN_HD int cut_tile(const Scratch &s, int tile, Emit &emit) {
    const int32_t *off = s.cell_off + tile * kTileCells;
    int jobs = 0, c0 = 0;
    for (int c = 0; c < kTileCells; ++c) {
        const int n = off[c + 1] - off[c];
        if (n > kMaxRuns) {
            if (c > c0) { emit(jobs, c0, c - c0, off[c0], off[c] - off[c0], (int)kChainNone); ++jobs; }
            for (int r = 0; r < n; r += kMaxRuns) {
                const int m = (n - r) < kMaxRuns ? (n - r) : kMaxRuns;
                emit(jobs, c, 1, off[c] + r, m, (int)((r == 0 ? kChainHead : kChainLink) | (r + m < n ? kChainMore : 0u)));
                ++jobs;
            }
            c0 = c + 1;
        } else if (c > c0 && off[c + 1] - off[c0] > kMaxRuns) {
            emit(jobs, c0, c - c0, off[c0], off[c] - off[c0], (int)kChainNone); ++jobs; c0 = c;
        }
    }
    if (c0 < kTileCells) { emit(jobs, c0, kTileCells - c0, off[c0], off[kTileCells] - off[c0], (int)kChainNone); ++jobs; }
    return jobs;
}
// phase 6: records per tile
PLAN_HD void phase_count_jobs(const Dims &d, Scratch &s, int tid, int nt) {
    for (int t = tid; t < d.ntiles; t += nt) {
        auto none = [](int, int, int, int, int, int) {};
        s.tile_jobs[t] = cut_tile(s, t, none);
    }
}
// phase 7: the order the records are laid out in.  The forward deals a calibration's records to its workgroups in index
// order, and workgroups start in index order: a tile that takes long (near the cameras a cell is fed by dozens of columns)
// must not be met last.  kGroups groups of neighbouring tiles (a group's context rows meet in one XCD's L2); inside a group
// the tiles by falling number of runs (ties: by tile index), so a group's heavy records come first.
PLAN_HD int group_begin(const Dims &d, int g) { return (int)((long long)g * d.ntiles / kGroups); }
PLAN_HD void phase_tile_order(const Dims &d, Scratch &s, int tid, int nt) {
    for (int g = tid; g < kGroups; g += nt) {
        const int lo = group_begin(d, g), hi = group_begin(d, g + 1);
        for (int i = lo; i < hi; ++i) {
            const int t = i, runs = s.cell_off[(t + 1) * kTileCells] - s.cell_off[t * kTileCells];
            int j = i - 1;
            while (j >= lo) {
                const int tj = s.tile_perm[j], rj = s.cell_off[(tj + 1) * kTileCells] - s.cell_off[tj * kTileCells];
                if (rj >= runs) break;
                s.tile_perm[j + 1] = tj; --j;
            }
            s.tile_perm[j + 1] = t;
        }
    }
}
PLAN_HD void phase_perm_gather(const Dims &d, Scratch &s, int tid, int nt) {
    for (int i = tid; i < d.ntiles; i += nt) s.perm_jobs[i] = s.tile_jobs[s.tile_perm[i]];
    if (tid == 0) s.perm_jobs[d.ntiles] = 0;
}
// phase 8 (after the scan of perm_jobs): first record of every tile and group; too many records?
PLAN_HD void phase_tile_bases(const Dims &d, Scratch &s, int tid, int nt) {
    for (int i = tid; i < d.ntiles; i += nt) s.tile_base[s.tile_perm[i]] = s.perm_jobs[i];
    if (tid == 0) {
        s.status[1] = s.perm_jobs[d.ntiles];
        if (s.perm_jobs[d.ntiles] > d.jobs_cap) s.status[2] = 1;
        for (int g = 0; g <= kGroups; ++g) s.status[3 + g] = s.perm_jobs[group_begin(d, g)];
    }
}
// phase 9: job descriptors
PLAN_HD void phase_write_jobs(const Dims &d, Scratch &s, int tid, int nt) {
    for (int t = tid; t < d.ntiles; t += nt) {
        const int base = s.tile_base[t];
        auto put = [&](int j, int c0, int nc, int rb, int nr, int chain) {
            JobDesc jd; jd.tile = t; jd.c0 = c0; jd.ncells = nc; jd.run_begin = rb; jd.nruns = nr; jd.chain = chain; jd.pad1 = jd.pad2 = 0;
            s.jobs[base + j] = jd;
        };
        cut_tile(s, t, put);
    }
}
// phase 10: the job records the forward kernel reads.  A job's runs are contiguous in the cell-sorted list; their position
// there is the partial row (`pslot`), `cell_begin` delimits the cells' partial rows.  The record lists the runs PAIR-major
// (a pair = consecutive runs, by key, of one (column, row block), at most kMaxPairRuns of them, whose bins fit one window of kWindowBins
// bins), the pairs by falling number of runs.
PLAN_HD void phase_records(const Dims &, Scratch &s, uint8_t *records, int njobs, int tid, int nt) {
    for (int j = tid; j < njobs; j += nt) {
        const JobDesc jd = s.jobs[j];
        uint8_t *rec = records + (long long)j * kJobBytes;
        const RunTmp *runs = s.runs + jd.run_begin;
        uint8_t *order = s.order + (long long)j * kMaxRuns;
        const int ctm0 = jd.tile * kTileCells + jd.c0;
        for (int i = 0; i < jd.nruns; ++i) {          // rank by (key, position): keys repeat across the cells of a mixed block
            const uint32_t k = runs[i].key;
            int rank = 0;
            for (int q = 0; q < jd.nruns; ++q) { const uint32_t kq = runs[q].key; rank += (kq < k || (kq == k && q < i)) ? 1 : 0; }
            order[rank] = (uint8_t)i;
        }
        uint8_t *cb = rec + kJobCellBeginOff;
        if (jd.chain != (int)kChainNone) { cb[0] = 0; for (int c = 1; c <= kTileCells; ++c) cb[c] = (uint8_t)jd.nruns; }      // one cell, this record's share of its runs
        else {
            for (int c = 0; c <= jd.ncells; ++c) cb[c] = (uint8_t)(s.cell_off[ctm0 + c] - jd.run_begin);
            for (int c = jd.ncells + 1; c <= kTileCells; ++c) cb[c] = (uint8_t)jd.nruns;
        }
        PairRec *pairs = reinterpret_cast<PairRec *>(rec + kJobPairsOff);
        RunRec *rr = reinterpret_cast<RunRec *>(rec + kJobRunsOff);
        // The pairs are found in key order (a pair = consecutive runs of one (column, row block) that fit a window) but LISTED by falling
        // number of runs: the forward hands four consecutive pairs to the four lane groups of a wave, and the wave loops as long as the
        // longest of them -- pairs of equal length side by side waste nothing (listed in key order a wave ran 3.6 iterations for a mean
        // of 2.9 runs per pair at BASELINE configs[3]).  One scan per length: nothing to store between the passes.
        auto run_rec = [&](int p) {
            const int i = order[p];
            const RunTmp r = runs[i];
            RunRec o;
            o.d0 = (uint16_t)((r.key & 0x7FFu) - (r.cell_len >> 30)); o.len = (uint8_t)(((r.cell_len >> 28) & 3u) + 1); o.pslot = (uint8_t)i;
            o.cell_local = (r.cell_len & 0x0FFFFFFFu) - (uint32_t)ctm0;
            o.wlo = 0ull; o.whi = 0ull;
            return o;
        };
        int npairs = 0, nout = 0;
        for (int want = kMaxPairRuns; want >= 1; --want) {
            int pstart = 0;
            while (pstart < jd.nruns) {
                // the pair that starts at key rank pstart: runs of its (column, row block) while they are at most kMaxPairRuns and fit the
                // window of kWindowBins bins from the even bin at or below the first run's (a dword-aligned address in bf16 too)
                const RunRec first = run_rec(pstart);
                const uint32_t pk = runs[order[pstart]].key >> 11;
                const int w0 = (int)first.d0 & ~1;
                int pend = pstart + 1;
                while (pend < jd.nruns && pend - pstart < kMaxPairRuns && (runs[order[pend]].key >> 11) == pk) {
                    const RunRec o = run_rec(pend);
                    if ((int)o.d0 + (int)o.len - w0 > kWindowBins) break;
                    ++pend;
                }
                if (pend - pstart == want) {
                    PairRec pr; pr.col = (uint16_t)(pk >> 5); pr.rb = (uint8_t)(pk & 31u); pr.nruns = (uint8_t)want; pr.run0 = (uint16_t)nout; pr.w0 = (uint16_t)w0;
                    pairs[npairs++] = pr;
                    for (int p = pstart; p < pend; ++p) {
                        RunRec o = run_rec(p);
                        const RunTmp r = runs[order[p]];
                        // the run's four masks moved to its place in the window (a 128-bit shift by 16 bits per bin; masks past the run's length are zero)
                        const uint64_t m = (uint64_t)r.mask[0] | ((uint64_t)r.mask[1] << 16) | ((uint64_t)r.mask[2] << 32) | ((uint64_t)r.mask[3] << 48);
                        const int off = (int)o.d0 - w0;
                        o.wlo = off < 4 ? (m << (16 * off)) : 0ull;
                        o.whi = off == 0 ? 0ull : (off < 4 ? (m >> (64 - 16 * off)) : (m << (16 * (off - 4))));
                        rr[nout++] = o;
                    }
                }
                pstart = pend;
            }
        }
        JobHeader h; h.ncells = (uint16_t)jd.ncells; h.npairs = (uint16_t)npairs; h.nruns = (uint16_t)jd.nruns; h.c0 = (uint16_t)jd.c0; h.tile = jd.tile; h.chain = (uint32_t)jd.chain;
        *reinterpret_cast<JobHeader *>(rec) = h;
    }
}

}  // namespace plan
}  // namespace mmt
