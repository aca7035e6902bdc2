// Output-stationary fused lift-splat forward ("plan form", ABI 10; SURVEY section 8 rows f1 + f3): replaces
// layers/backbones/lss_fpn.py:328-361 (get_geometry), :461-462 (quantise), :441-464 (lift, voxel_pooling) like the camera
// form of lift_splat_tile.hip, but with the roles turned round: the OUTPUT is stationary.
//
// The camera-form ray walks are input-stationary: a workgroup owns an image column, and the BEV rows it produces meet in
// memory through fp32 atomics on a zero-filled map (12 MB of atomics + a 21 MB fill per launch at BASELINE configs[3]; sums in
// arrival order).  Which points feed a cell depends on the calibration only, so here it is worked out ONCE per calibration,
// on the device, into a plan (lss_plan_core.h): runs (<= 4 consecutive depth bins of a 16-row block of one image column that
// share a cell, with their row masks) grouped into jobs (a contiguous range of the cells of an 8 x 8 BEV tile, <= 96 runs).
//   lss_plan_probe   one workgroup: hashes every sample's matrices, finds (or claims) its slot in the caller's plan cache,
//                    leaves a verdict per sample and a to-do list of calibrations to learn
//   lss_plan_build   one workgroup per calibration to learn: column summary (the geometry of mmt_camera.h), runs, jobs, records
//   lss_plan_fwd     a workgroup per job: lane groups take the job's (column, row block) pairs -- the pair's 16 context rows
//                    in registers, a run's depths summed per image row, one partial row per run into LDS -- then every cell's
//                    partial rows are summed in plan order and STORED; cells nobody reaches are stored as zeros.
// No zero fill, no atomics, every output element written exactly once, bit-identical from launch to launch.  A calibration
// whose plan would not fit its slot (a rig rolled so far that most blocks of rows straddle cells) is served by the same
// kernel's brute-force path from the slot's column summary: slow, still deterministic, still exact.
#include <stdlib.h>
#include <atomic>
#include <string.h>

#include "lss_plan_core.h"
#include "mmt_camera.h"

namespace {

using namespace mmt::plan;
#include "depth_softmax_body.h"

constexpr int kPlanMaxB = 64;            // samples per call
constexpr int kPlanMaxN = 16;            // cameras per sample
constexpr int kBuildPar = 8;             // calibrations learnt concurrently (scratch areas)
constexpr int kBuildThreads = 1024;
#ifndef PLAN_FWD_THREADS
#define PLAN_FWD_THREADS 256
#endif
constexpr int kFwdThreads = PLAN_FWD_THREADS;     // 16 lanes per lane group
constexpr int kFwdGroups = kFwdThreads / 16;
constexpr unsigned kPlanMagic = 0x4E4C504Du;      // "MPLN"
constexpr int kStateEmpty = 0, kStateReady = 1, kStateBrute = 2;

// ---- the caller's plan cache: header | verdicts | to-do list | slots | build scratch --------------------------------------
constexpr int64_t kHdrBytes = 8192;
constexpr int64_t kVerdictOff = 256;     // Verdict [kPlanMaxB]
constexpr int64_t kTodoOff = 256 + 64 * kPlanMaxB;       // int4 [kPlanMaxB]: sample, slot, hash lo, hash hi
constexpr int64_t kDupOff = kTodoOff + 16 * kPlanMaxB;   // int [kPlanMaxB]: the earlier sample with the same matrices, or -1
constexpr int64_t kSlotMetaBytes = 2048; // SlotMeta

// what the forward goes by for one sample of the call: its calibration's slot, how many units (records of a learnt plan; tiles
// for the brute-force path) in which of the kGroups groups
struct Verdict { int slot, units, state, rep; int gstart[kGroups + 1]; int pad[3]; };
static_assert(sizeof(Verdict) == 64, "Verdict");

struct CacheHeader {
    unsigned magic, sig_lo, sig_hi, axes_lo, axes_hi, clock, nslots, todo_count;
    unsigned hits, built, brute, resets, calls, stale, snaps, snap_next;   // stale: forwards that found a verdict nobody prepared (see lss_plan_fwd);
};                                                                         // snaps / snap_next: batch snapshots of the lookup's fast path
constexpr int64_t kFlagOff = 128;        // unsigned inside the header block: the token of the lookup launch whose probe is done
constexpr int kSnaps = 4;                // batches (matrices + verdicts) the lookup recognises without probing the slots
struct SnapHdr { unsigned valid, B, words, n_hit; };
struct SlotMeta { unsigned hash_lo, hash_hi; int state, njobs, nruns; unsigned stamp; int gstart[kGroups + 1]; int pad[1]; float mats[kPlanMaxN * 16]; };

struct Layout {
    Dims d;
    int64_t summary_off, records_off, slot_bytes, scratch_bytes, slots_off, scratch_off, total;
    int nslots;
};
static inline int64_t up256(int64_t v) { return (v + 255) & ~255ll; }
void make_layout(int N, int D, int fH, int fW, int nx, int ny, int slots, Layout *l) {
    make_dims(N, D, fH, fW, nx, ny, &l->d);
    l->summary_off = kSlotMetaBytes;
    l->records_off = l->summary_off + up256(8ll * l->d.strips * D);
    l->slot_bytes = l->records_off + up256((int64_t)l->d.jobs_cap * kJobBytes);
    l->scratch_bytes = up256(scratch_bytes(l->d, kBuildThreads));
    l->nslots = slots;
    l->slots_off = kHdrBytes;
    l->scratch_off = l->slots_off + (int64_t)slots * l->slot_bytes;
    l->total = l->scratch_off + (int64_t)kBuildPar * l->scratch_bytes;
}

struct PlanArgs {                        // what the three kernels share
    unsigned char *cache;
    int64_t summary_off, records_off, slot_bytes, scratch_bytes, slots_off, scratch_off;
    int64_t snap_stride;                 // bytes per batch snapshot at the head of the build scratch (0: no room, no fast path)
    unsigned token;                      // of this lookup launch (host counter, never 0): what workgroup 0 publishes and the builders wait for
    Dims d;
    int B, nz, nslots;
    unsigned sig_lo, sig_hi;
    const float *combine, *fu, *fv, *fd;
    mmt::CamGrid q;
};

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// ---- probe ------------------------------------------------------------------------------------------------------------------
// One workgroup.  Per sample: a 64-bit hash of its N matrices; a sample whose matrices equal an earlier sample's (bit for
// bit) shares that sample's verdict; otherwise its slot is looked up among the cache's slots (hash, then the matrices bit for
// bit); a miss claims an empty slot, else the least recently used one that this call does not use, and goes on the to-do
// list.  The frustum axes' contents and the launch shape sign the table: a change empties it.  Everything that reads memory
// is spread over the workgroup (the samples over its waves), the decisions that depend on each other are made by one wave
// from LDS: two or three rounds of loads in all.
__device__ __forceinline__ void verdict_from_slot(Verdict *v, int slot, const SlotMeta *sm, int rep, const Dims &d) {
    v->slot = slot; v->units = sm->njobs; v->state = sm->state; v->rep = rep;
    for (int g = 0; g <= kGroups; ++g) v->gstart[g] = sm->state == kStateReady ? sm->gstart[g] : group_begin(d, g);
}

__device__ __forceinline__ void plan_probe(const PlanArgs &a, int *s_result /* LDS: [0] calibrations to learn, [1] table was reset, [2] distinct hits */) {
    __shared__ unsigned long long s_hash[kPlanMaxB];
    __shared__ unsigned long long s_axes;
    __shared__ unsigned s_shash_lo[256], s_shash_hi[256], s_stamp[256];
    __shared__ int s_state[256], s_used[256], s_slot[kPlanMaxB], s_rep[kPlanMaxB], s_njobs[256], s_gs[256][kGroups + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwav = nthr >> 6;
    CacheHeader *hdr = reinterpret_cast<CacheHeader *>(a.cache);
    Verdict *verdict = reinterpret_cast<Verdict *>(a.cache + kVerdictOff);
    int4 *todo = reinterpret_cast<int4 *>(a.cache + kTodoOff);
    int *dup = reinterpret_cast<int *>(a.cache + kDupOff);
    const int words = a.d.N * 16;
    const unsigned *mats = reinterpret_cast<const unsigned *>(a.combine);
    if (tid == 0) s_axes = 0ull;
    if (tid < kPlanMaxB) s_hash[tid] = 0ull;
    __syncthreads();
    {   // order-independent hashes (position-keyed terms, summed): the axes' contents, every sample's matrices
        unsigned long long part = 0ull;
        const int na = a.d.fW + a.d.fH + a.d.D;
        for (int i = tid; i < na; i += nthr) {
            const float v = i < a.d.fW ? a.fu[i] : (i < a.d.fW + a.d.fH ? a.fv[i - a.d.fW] : a.fd[i - a.d.fW - a.d.fH]);
            part += mix64(((unsigned long long)i << 32) | __float_as_uint(v));
        }
        atomicAdd(&s_axes, part);
        for (int i = tid; i < a.B * words; i += nthr) {
            const int b = i / words, k = i - b * words;
            atomicAdd(&s_hash[b], mix64(((unsigned long long)k << 32) | mats[i]));
        }
    }
    const unsigned h_magic = hdr->magic, h_sig_lo = hdr->sig_lo, h_sig_hi = hdr->sig_hi, h_ax_lo = hdr->axes_lo, h_ax_hi = hdr->axes_hi,
                   h_nslots = hdr->nslots, h_clock = hdr->clock, h_resets = hdr->resets;
    __syncthreads();
    const unsigned ax_lo = (unsigned)s_axes, ax_hi = (unsigned)(s_axes >> 32);
    const int reset = (h_magic != kPlanMagic || h_sig_lo != a.sig_lo || h_sig_hi != a.sig_hi || h_ax_lo != ax_lo || h_ax_hi != ax_hi ||
                       h_nslots != (unsigned)a.nslots) ? 1 : 0;
    for (int s = tid; s < a.nslots; s += nthr) {
        SlotMeta *sm = reinterpret_cast<SlotMeta *>(a.cache + a.slots_off + (int64_t)s * a.slot_bytes);
        if (reset) sm->state = kStateEmpty;
        s_shash_lo[s] = sm->hash_lo; s_shash_hi[s] = sm->hash_hi; s_stamp[s] = sm->stamp;
        const int st_ = reset ? kStateEmpty : sm->state;
        s_state[s] = st_; s_njobs[s] = sm->njobs;
#pragma unroll
        for (int g = 0; g <= kGroups; ++g) s_gs[s][g] = st_ == kStateReady ? sm->gstart[g] : group_begin(a.d, g);
        s_used[s] = 0;
    }
    __syncthreads();
    auto same_words = [&](const unsigned *p, const unsigned *q) {      // one wave, the words spread over its lanes
        bool eq = true;
        for (int i = lane; i < words; i += 64) eq = eq && p[i] == q[i];
        return __all(eq) != 0;
    };
    // pass 1, a sample per wave: the earlier sample with the same matrices, else the slot that holds them (independent of the
    // other samples' outcomes)
    for (int b = wave; b < a.B; b += nwav) {
        const unsigned long long h = s_hash[b];
        const unsigned *mb = mats + (int64_t)b * words;
        int rep = -1;
        for (int e = 0; e < b && rep < 0; ++e)
            if (s_hash[e] == h && same_words(mb, mats + (int64_t)e * words)) rep = e;
        int slot = -1;
        if (rep < 0) {
            for (int s = 0; s < a.nslots && slot < 0; ++s) {
                if (s_state[s] != kStateEmpty && s_shash_lo[s] == (unsigned)h && s_shash_hi[s] == (unsigned)(h >> 32)) {
                    const SlotMeta *sm = reinterpret_cast<const SlotMeta *>(a.cache + a.slots_off + (int64_t)s * a.slot_bytes);
                    if (same_words(mb, reinterpret_cast<const unsigned *>(sm->mats))) slot = s;
                }
            }
        }
        if (lane == 0) { s_rep[b] = rep; s_slot[b] = slot; dup[b] = rep; }
    }
    __syncthreads();
    if (tid >= 64) return;               // (the caller's next barrier waits for wave 0)
    // wave 0 decides the rest from LDS, every lane with the same (uniform) values; lane 0 writes
    const unsigned clock = reset ? 1u : h_clock + 1u;
    unsigned n_hit = 0, n_todo = 0;
    for (int b = 0; b < a.B; ++b) {
        const int slot = s_slot[b];
        if (s_rep[b] < 0 && slot >= 0) {
            SlotMeta *sm = reinterpret_cast<SlotMeta *>(a.cache + a.slots_off + (int64_t)slot * a.slot_bytes);
            if (lane == 0) {                   // (everything it needs is in LDS: stores only)
                Verdict v;
                v.slot = slot; v.units = s_njobs[slot]; v.state = s_state[slot]; v.rep = b;
#pragma unroll
                for (int g = 0; g <= kGroups; ++g) v.gstart[g] = s_gs[slot][g];
                v.pad[0] = v.pad[1] = v.pad[2] = 0;
                verdict[b] = v;
                sm->stamp = clock;
            }
            s_used[slot] = 1;
            ++n_hit;
        }
    }
    // pass 2: a slot for every calibration to learn -- an empty one, else the least recently used one this call does not use
    // (the entry points demand nslots >= B, so there is one)
    for (int b = 0; b < a.B; ++b) {
        if (s_rep[b] >= 0 || s_slot[b] >= 0) continue;
        const unsigned long long h = s_hash[b];
        int victim = -1;
        for (int s = 0; s < a.nslots && victim < 0; ++s) if (s_state[s] == kStateEmpty && !s_used[s]) victim = s;
        if (victim < 0) {
            unsigned best = 0xFFFFFFFFu;
            for (int s = 0; s < a.nslots; ++s) if (!s_used[s] && (victim < 0 || s_stamp[s] < best)) { victim = s; best = s_stamp[s]; }
        }
        s_used[victim] = 1; s_state[victim] = kStateEmpty;
        if (lane == 0) {
            SlotMeta *sm = reinterpret_cast<SlotMeta *>(a.cache + a.slots_off + (int64_t)victim * a.slot_bytes);
            sm->state = kStateEmpty;           // (nothing of it is served while it is being learnt)
            todo[n_todo] = make_int4(b, victim, (int)(unsigned)h, (int)(unsigned)(h >> 32));
            verdict[b].slot = victim; verdict[b].units = 0; verdict[b].state = kStateEmpty; verdict[b].rep = b;
        }
        ++n_todo;
    }
    // duplicates of a hit copy its verdict now (the build writes those of a sample it learns); lane 0 wrote them itself
    if (lane == 0) {
        for (int b = 0; b < a.B; ++b) {
            const int rep = s_rep[b];
            if (rep >= 0 && s_slot[rep] >= 0) {
                const int slot = s_slot[rep];
                Verdict v;
                v.slot = slot; v.units = s_njobs[slot]; v.state = s_state[slot]; v.rep = rep;
#pragma unroll
                for (int g = 0; g <= kGroups; ++g) v.gstart[g] = s_gs[slot][g];
                v.pad[0] = v.pad[1] = v.pad[2] = 0;
                verdict[b] = v;
            }
        }
        hdr->magic = kPlanMagic; hdr->sig_lo = a.sig_lo; hdr->sig_hi = a.sig_hi; hdr->axes_lo = ax_lo; hdr->axes_hi = ax_hi;
        hdr->clock = clock; hdr->nslots = (unsigned)a.nslots; hdr->todo_count = n_todo;
        if (reset) { hdr->hits = 0; hdr->built = 0; hdr->brute = 0; hdr->resets = h_magic == kPlanMagic ? h_resets + 1 : 0; hdr->calls = 0; hdr->stale = 0; }
        hdr->hits += n_hit; hdr->built += n_todo; hdr->calls += 1;
        s_result[0] = (int)n_todo; s_result[1] = reset; s_result[2] = (int)n_hit;
    }
}

// ---- build ------------------------------------------------------------------------------------------------------------------
struct DevRowCells {                     // the cells of a mixed block's rows, from the matrices (mmt_camera.h: bit-identical to the kernels')
    const PlanArgs &a;
    int b;
    int last_s, last_bin;
    mmt_cam_column cc;
    __device__ int operator()(int s, int bin, int row) {
        const Dims &d = a.d;
        const int w = s % d.fW, rb = (s / d.fW) % d.nb, n = s / (d.fW * d.nb);
        if (s != last_s || bin != last_bin) {
            float cm[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) cm[k] = a.combine[((int64_t)b * d.N + n) * 16 + k];
            cc = mmt_cam_column_make(cm, a.fu[w], a.fd[bin]);
            last_s = s; last_bin = bin;
        }
        const int r = rb * 16 + row;
        int gx, gy;
        const bool in = mmt_cam_row_xy(cc, a.fv[r < d.fH ? r : d.fH - 1], a.q, d.nx, d.ny, gx, gy);
        return in ? ((gy << 16) | gx) : -1;
    }
};

template <int NT>       // threads of the workgroup (1 024 in a launch of its own; 256 as a rider of the depth softmax: same plan, four times as long)
__device__ __forceinline__ void plan_build(const PlanArgs &a, int bi, int nb) {
    const CacheHeader *hdr = reinterpret_cast<const CacheHeader *>(a.cache);
    const int4 *todo = reinterpret_cast<const int4 *>(a.cache + kTodoOff);
    const int *dup = reinterpret_cast<const int *>(a.cache + kDupOff);
    Verdict *verdict = reinterpret_cast<Verdict *>(a.cache + kVerdictOff);
    const int ntodo = (int)hdr->todo_count;
    const int tid = threadIdx.x, nt = NT;
    const Dims &d = a.d;
    Scratch sc;
    scratch_carve(d, kBuildThreads, a.cache + a.scratch_off + (int64_t)bi * a.scratch_bytes, &sc);
    for (int k = bi; k < ntodo; k += nb) {
        const int4 td = todo[k];
        const int b = td.x, slot = td.y;
        unsigned char *sbase = a.cache + a.slots_off + (int64_t)slot * a.slot_bytes;
        int2 *summary = reinterpret_cast<int2 *>(sbase + a.summary_off);
        // ---- the column summary: (cell of the first row | z mask | "one cell" bit) per (strip, bin)
        const int total = d.strips * d.D;
        for (int base = 0; base < total; base += nt) {
            const int e = base + tid;
            const bool valid = e < total;
            const int ec = valid ? e : total - 1;
            const int s = ec / d.D, bin = ec - s * d.D;
            const int w = s % d.fW, rb = (s / d.fW) % d.nb, n = s / (d.fW * d.nb);
            float cm[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) cm[q] = a.combine[((int64_t)b * d.N + n) * 16 + q];
            const mmt_cam_column cc = mmt_cam_column_make(cm, a.fu[w], a.fd[bin]);
            const int r0 = rb * 16;
            const int nr = (d.fH - r0) < 16 ? (d.fH - r0) : 16;
            unsigned zm16 = 0;
            bool uni16 = true, in00 = false;
            int x00 = 0, y00 = 0;
#pragma unroll 1
            for (int hb = 0; hb < 16; hb += 8) {
                if (hb < nr) {
                    const int nh = (nr - hb) < 8 ? (nr - hb) : 8;
                    float cv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) cv[u] = a.fv[r0 + hb + (u < nh ? u : nh - 1)];
                    bool uniform, in0;
                    int x0, y0;
                    const unsigned zmask = mmt_cam_column_cells<8>(cc, cv, nh, mmt_rows_sorted<8>(cv, nh), a.q, d.nx, d.ny, a.nz, uniform, in0, x0, y0);
                    bool mine = uniform;       // `uniform` is a wave-wide verdict; when it fails, this block's own rows decide its bit
                    if (!uniform) {
                        mine = true;
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            int gx, gy;
                            mmt_cam_row_xy(cc, cv[u], a.q, d.nx, d.ny, gx, gy);
                            mine = mine && gx == x0 && gy == y0;
                        }
                        mine = mine || zmask == 0u;
                    }
                    zm16 |= zmask << hb;
                    if (hb == 0) { uni16 = mine; in00 = in0; x00 = x0; y00 = y0; }
                    else uni16 = uni16 && mine && in0 == in00 && x0 == x00 && y0 == y00;      // (the rule of the forward kernels that write a summary)
                }
            }
            if (valid) summary[e] = make_int2(in00 ? ((y00 << 16) | x00) : -1, (int)zm16 | (uni16 ? mmt::kSummaryUniform : 0));
        }
        __syncthreads();
        const int32_t *sum32 = reinterpret_cast<const int32_t *>(summary);
        DevRowCells rc{a, b, -1, -1, {}};
        phase_clear(d, sc, tid, nt);
        __syncthreads();
        phase_count(d, sc, sum32, rc, tid, nt);
        __syncthreads();
        scan_a(sc.cell_off, d.ncells_tm, sc.partial, tid, nt); __syncthreads();
        scan_b(sc.cell_off, d.ncells_tm, sc.partial, tid, nt); __syncthreads();
        scan_c(sc.cell_off, d.ncells_tm, sc.partial, tid, nt); __syncthreads();
        phase_check_runs(d, sc, tid);
        __syncthreads();
        int njobs = 0;
        if (!sc.status[2]) {
            phase_place(d, sc, sum32, rc, tid, nt);
            __syncthreads();
            phase_sort_cells(d, sc, tid, nt);
            __syncthreads();
            phase_count_jobs(d, sc, tid, nt);
            __syncthreads();
            phase_tile_order(d, sc, tid, nt);
            __syncthreads();
            phase_perm_gather(d, sc, tid, nt);
            __syncthreads();
            scan_a(sc.perm_jobs, d.ntiles, sc.partial, tid, nt); __syncthreads();
            scan_b(sc.perm_jobs, d.ntiles, sc.partial, tid, nt); __syncthreads();
            scan_c(sc.perm_jobs, d.ntiles, sc.partial, tid, nt); __syncthreads();
            phase_tile_bases(d, sc, tid, nt);
            __syncthreads();
            if (!sc.status[2]) {
                phase_write_jobs(d, sc, tid, nt);
                __syncthreads();
                njobs = sc.status[1];
                phase_records(d, sc, sbase + a.records_off, njobs, tid, nt);
            }
        }
        __syncthreads();
        const int state = sc.status[2] ? kStateBrute : kStateReady;
        const int units = state == kStateReady ? njobs : d.ntiles;
        SlotMeta *sm = reinterpret_cast<SlotMeta *>(sbase);
        for (int i = tid; i < d.N * 16; i += nt) sm->mats[i] = a.combine[(int64_t)b * d.N * 16 + i];
        if (tid == 0) {
            sm->hash_lo = (unsigned)td.z; sm->hash_hi = (unsigned)td.w; sm->njobs = units; sm->nruns = sc.status[0]; sm->stamp = hdr->clock;
            for (int g = 0; g <= kGroups; ++g) sm->gstart[g] = state == kStateReady ? sc.status[3 + g] : group_begin(d, g);
            sm->state = state;
        }
        __syncthreads();
        for (int e = tid; e < a.B; e += nt)
            if (e == b || dup[e] == b) verdict_from_slot(&verdict[e], slot, sm, b, d);
        __syncthreads();
    }
}

// ---- lookup: probe + build in ONE launch, with a fast path for batches seen before ---------------------------------------
// Workgroup 0 looks the batch up; workgroups 1.. wait for its verdict (an agent-scope flag: release by lane 0 after the
// workgroup's barrier, relaxed polls + one acquire on the other side; MI355X_MICROARCH.md "Inter-workgroup visibility"; the flag's
// value is a per-launch token from the host, so nothing has to be re-armed and an uninitialised cache cannot look "done") and learn
// what is on the to-do list -- in the steady state nothing: they leave at once, and no empty build kernel is launched any more.
// FAST PATH: the matrices and verdicts of the last kSnaps fully-known batches are kept at the head of the build scratch
// (a build invalidates them: it overwrites the scratch and may re-assign slots).  A batch whose matrices equal a snapshot's bit
// for bit -- under an unchanged table signature and unchanged frustum axes -- gets that snapshot's verdicts copied back: two
// dependent rounds of loads instead of the probe's five or six.  Round 5's two launches took 10.7 + 3.6 us per step.
__device__ __forceinline__ unsigned char *snap_base(const PlanArgs &a, int k) { return a.cache + a.scratch_off + (int64_t)k * a.snap_stride; }

template <int NT>
__device__ __forceinline__ void plan_lookup_body(const PlanArgs &a, const int wg, const int nwg) {      // workgroup wg (NT threads) of the nwg that do the lookup
    unsigned *flags = reinterpret_cast<unsigned *>(a.cache + kFlagOff);
    const int tid = threadIdx.x, nthr = NT;
    constexpr int VW = (kPlanMaxB * (int)(sizeof(Verdict) / 4) + NT - 1) / NT;     // verdict words per thread
    if (wg != 0) {
        // ---- a builder: wait for workgroup 0's word -- the launch's token: there is a to-do list (acquire, then build); token + 1:
        // nothing to learn, leave without reading anything.  (Bounded: a wait that runs out leaves the list unbuilt, which the forward counts as stale.)
        __shared__ int s_go;
        if (tid == 0) {
            int spins = 0;
            unsigned v = __hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (v != a.token && v != a.token + 1u && spins < (1 << 22)) {
                __builtin_amdgcn_s_sleep(4); ++spins;
                v = __hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (v == a.token) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_go = v == a.token ? 1 : 0;
        }
        __syncthreads();
        if (s_go) plan_build<NT>(a, wg - 1, nwg - 1);
        return;
    }
    // ---- workgroup 0
    __shared__ unsigned long long s_axpart[NT / 64];
    __shared__ unsigned s_badpart[NT / 64];
    __shared__ int s_res[4];             // plan_probe's result
    CacheHeader *hdr = reinterpret_cast<CacheHeader *>(a.cache);
    Verdict *verdict = reinterpret_cast<Verdict *>(a.cache + kVerdictOff);
    const int words = a.B * a.d.N * 16, vwords = a.B * (int)(sizeof(Verdict) / 4);
    const unsigned *mats = reinterpret_cast<const unsigned *>(a.combine);
    const int64_t voff = 256 + (((int64_t)words * 4 + 255) & ~255ll);                  // a snapshot: [SnapHdr 256 B][matrices][verdicts]
    const bool have_snaps = a.snap_stride > 0 && voff + (int64_t)vwords * 4 <= a.snap_stride;
    // ONE round of loads, none depending on another: the header, the axes, the batch's matrices, and all kSnaps snapshots'
    // headers, matrices and verdicts whether valid or not (the addresses are inside the scratch either way; what a void one holds is not used)
    const unsigned h_magic = hdr->magic, h_sig_lo = hdr->sig_lo, h_sig_hi = hdr->sig_hi, h_ax_lo = hdr->axes_lo, h_ax_hi = hdr->axes_hi,
                   h_nslots = hdr->nslots, h_snaps = hdr->snaps, h_next = hdr->snap_next, h_clock = hdr->clock, h_hits = hdr->hits, h_calls = hdr->calls;
    unsigned bad = 0u, sn_hit[kSnaps] = {0u, 0u, 0u, 0u};
    unsigned vw[kSnaps][VW] = {};
    if (have_snaps) {
        unsigned long long part = 0ull;
        const int na = a.d.fW + a.d.fH + a.d.D;
        for (int i = tid; i < na; i += nthr) {
            const float v = i < a.d.fW ? a.fu[i] : (i < a.d.fW + a.d.fH ? a.fv[i - a.d.fW] : a.fd[i - a.d.fW - a.d.fH]);
            part += mix64(((unsigned long long)i << 32) | __float_as_uint(v));
        }
#pragma unroll
        for (int k = 0; k < kSnaps; ++k) {
            const SnapHdr sh = *reinterpret_cast<const SnapHdr *>(snap_base(a, k));
            if (sh.valid != 1u || sh.B != (unsigned)a.B || sh.words != (unsigned)words) bad |= 1u << k;
            sn_hit[k] = sh.n_hit;
#pragma unroll
            for (int j = 0; j < VW; ++j)
                if (tid + j * NT < vwords) vw[k][j] = reinterpret_cast<const unsigned *>(snap_base(a, k) + voff)[tid + j * NT];
        }
        for (int i = tid; i < words; i += nthr) {
            const unsigned m = mats[i];
#pragma unroll
            for (int k = 0; k < kSnaps; ++k) if (reinterpret_cast<const unsigned *>(snap_base(a, k) + 256)[i] != m) bad |= 1u << k;
        }
        // the workgroup's sum / or: per wave by cross-lane steps, the waves' parts through LDS (one barrier, nothing to initialise)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { part += __shfl_xor(part, o); bad |= (unsigned)__shfl_xor((int)bad, o); }
        if ((tid & 63) == 0) { s_axpart[tid >> 6] = part; s_badpart[tid >> 6] = bad; }
    }
    if (tid == 0) { s_res[0] = 0; s_res[1] = 0; s_res[2] = 0; }
    __syncthreads();
    int hit = -1;
    if (have_snaps) {
        unsigned long long ax = 0ull;
        unsigned badall = 0u;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { ax += s_axpart[w]; badall |= s_badpart[w]; }
        const bool table_ok = h_magic == kPlanMagic && h_sig_lo == a.sig_lo && h_sig_hi == a.sig_hi && h_nslots == (unsigned)a.nslots &&
                              (unsigned)ax == h_ax_lo && (unsigned)(ax >> 32) == h_ax_hi;
        const int nsn = table_ok ? (int)(h_snaps < (unsigned)kSnaps ? h_snaps : (unsigned)kSnaps) : 0;
#pragma unroll
        for (int k = kSnaps - 1; k >= 0; --k) if (k < nsn && !((badall >> k) & 1u)) hit = k;
    }
    if (hit >= 0) {
        // the builders have nothing to read: let them go before the verdicts are even written
        if (tid == 0 && nwg > 1) __hip_atomic_store(&flags[0], a.token + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned nh = sn_hit[0];
#pragma unroll
        for (int k = 1; k < kSnaps; ++k) if (hit == k) nh = sn_hit[k];
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            unsigned v = vw[0][j];
#pragma unroll
            for (int k = 1; k < kSnaps; ++k) if (hit == k) v = vw[k][j];
            if (tid + j * NT < vwords) reinterpret_cast<unsigned *>(verdict)[tid + j * NT] = v;
        }
        if (tid == 0) { hdr->todo_count = 0; hdr->clock = h_clock + 1u; hdr->hits = h_hits + nh; hdr->calls = h_calls + 1u; }
        return;
    }
    plan_probe(a, s_res);
    __syncthreads();
    const int n_todo = s_res[0];
    if (have_snaps) {
        const bool table_was_ok = h_magic == kPlanMagic && h_sig_lo == a.sig_lo && h_sig_hi == a.sig_hi && h_nslots == (unsigned)a.nslots;
        if ((n_todo > 0 || s_res[1]) && tid == 0) { hdr->snaps = 0; hdr->snap_next = 0; }    // a build follows (or the table was reset): every snapshot is void
        if (n_todo == 0) {
            // every calibration of the batch is known: remember the batch
            const int k = (s_res[1] ? 0 : (int)h_next) % kSnaps;
            unsigned *sm = reinterpret_cast<unsigned *>(snap_base(a, k) + 256);
            unsigned *sv = reinterpret_cast<unsigned *>(snap_base(a, k) + voff);
            const unsigned *dv = reinterpret_cast<const unsigned *>(verdict);
            for (int i = tid; i < words; i += nthr) sm[i] = mats[i];
            for (int i = tid; i < vwords; i += nthr) sv[i] = dv[i];
            if (tid == 0) {
                *reinterpret_cast<SnapHdr *>(snap_base(a, k)) = SnapHdr{1u, (unsigned)a.B, (unsigned)words, (unsigned)s_res[2]};
                const unsigned base_n = (s_res[1] || !table_was_ok) ? 0u : h_snaps;
                hdr->snaps = base_n > (unsigned)(k + 1) ? base_n : (unsigned)(k + 1);
                hdr->snap_next = (unsigned)(k + 1) % kSnaps;
            }
        }
    }
    if (nwg == 1) return;
    if (n_todo == 0) {
        if (tid == 0) __hip_atomic_store(&flags[0], a.token + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // ---- publish the to-do list: every storing wave drains, the workgroup meets, lane 0 releases at agent scope and raises the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&flags[0], a.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (a token per launch: no re-arming, and whatever an
                                                                                                // uninitialised cache holds here is not it)
    }
}

__global__ __launch_bounds__(kBuildThreads) void lss_plan_lookup(PlanArgs a) { plan_lookup_body<kBuildThreads>(a, (int)blockIdx.x, (int)gridDim.x); }

// ---- the lookup as a rider of the depth softmax ---------------------------------------------------------------------------------
// A lookup of known calibrations is one round of memory loads in ONE workgroup: 4.8 us as a launch of its own on an idle card,
// 6-10 us inside a training step -- against a forward of 27 us.  The depth softmax (csrc/depth_softmax.hip) runs between the
// moment the matrices exist and the forward anyway, for 6.4 us over a thousand workgroups: this kernel is that softmax -- its grid,
// its 256 threads, its rows' code -- with the lookup's 1 + g workgroups in FRONT (they start first; what they do ends long before the
// softmax's rows do), so the steady state has no launch for the lookup at all.  The lookup and the build run with 256 threads here
// (the plan does not depend on the thread count; a batch with calibrations to learn makes this launch as long as the build:
// ~6 ms per calibration instead of 1.6, once).  Why 256: the same rows in workgroups of 1 024 threads take 7.2-7.6 us instead
// of 6.2 (tools/ubench/softmax_shape.hip), while the build's 128 registers, 16 KB of LDS and private segment cost the 256-thread
// shape 0.3 us.
constexpr int kRiderThreads = 256;
__global__ __launch_bounds__(kRiderThreads, 4) void lss_plan_lookup_softmax(PlanArgs a, SoftmaxArgs s, int variant, int nlookup) {
    if ((int)blockIdx.x < nlookup) { plan_lookup_body<kRiderThreads>(a, (int)blockIdx.x, nlookup); return; }
    const int blk = (int)blockIdx.x - nlookup, nblk = (int)gridDim.x - nlookup, tid = threadIdx.x;
#define MMT_RIDER_CASE(id, LT, UT, NV) case id: softmax_fwd_rows<LT, UT, 4, NV>(s, blk, nblk, tid, kRiderThreads); break;
    switch (variant) {          // (logits type, depth_used type, 16-byte pieces per lane)
        MMT_RIDER_CASE(0, float, float, 2) MMT_RIDER_CASE(1, float, float, 4) MMT_RIDER_CASE(2, float, float, 8)
        MMT_RIDER_CASE(3, float, bf16_t, 2) MMT_RIDER_CASE(4, float, bf16_t, 4) MMT_RIDER_CASE(5, float, bf16_t, 8)
        MMT_RIDER_CASE(6, bf16_t, float, 2) MMT_RIDER_CASE(7, bf16_t, float, 4) MMT_RIDER_CASE(8, bf16_t, float, 8)
        MMT_RIDER_CASE(9, bf16_t, bf16_t, 2) MMT_RIDER_CASE(10, bf16_t, bf16_t, 4) MMT_RIDER_CASE(11, bf16_t, bf16_t, 8)
        default: break;
    }
#undef MMT_RIDER_CASE
}

// ---- forward ----------------------------------------------------------------------------------------------------------------
struct FwdArgs {
    PlanArgs p;
    const void *depth, *context;
    float *out;
    int2 *summary_out;                    // nullable: the batch's column summary [B*N, nb, fW, D] for the backward
    int C, W, xps;                        // channels; workgroups per sample; XCDs per sample (0: plain mapping)
    int force_brute;
    unsigned depth_bytes, ctx_bytes;
};

template <int H>
__device__ __forceinline__ float row_bcast(float v) {     // lane H of every row of 16 lanes, to all 16 (DPP row_newbcast)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + H, 0xF, 0xF, false));
}

typedef float plan_v2f __attribute__((ext_vector_type(2)));
typedef float plan_f4u __attribute__((ext_vector_type(4), aligned(4)));      // four floats at a 4-byte aligned address

// a lane group's share of a row of C = 16 * S channels: lane li holds channels [4 li, 4 li + 4) (+ [64 + 4 li, ..) for S = 8) and 64 + li (S = 5)

template <int S> struct Acc {
    static constexpr int NQ = S / 4, N1 = S % 4;          // float4 pieces, single floats (S in {4, 5, 8})
    plan_v2f q[NQ * 2];
    float s[N1 > 0 ? N1 : 1];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < NQ * 2; ++i) q[i] = plan_v2f{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < (N1 > 0 ? N1 : 1); ++i) s[i] = 0.f;
    }
    __device__ __forceinline__ void fma(float w, const Acc &c) {
        const plan_v2f ww = {w, w};
#pragma unroll
        for (int i = 0; i < NQ * 2; ++i) q[i] = __builtin_elementwise_fma(c.q[i], ww, q[i]);
#pragma unroll
        for (int i = 0; i < N1; ++i) s[i] = __builtin_fmaf(w, c.s[i], s[i]);
    }
    __device__ __forceinline__ void add(const Acc &c) {
#pragma unroll
        for (int i = 0; i < NQ * 2; ++i) q[i] += c.q[i];
#pragma unroll
        for (int i = 0; i < N1; ++i) s[i] += c.s[i];
    }
};

// context row piece of lane li: elements at `row` (FT *), fp32 in registers
template <typename FT, int S>
__device__ __forceinline__ void load_ctx(const FT *row, int li, Acc<S> &c) {
    constexpr int NQ = S / 4, N1 = S % 4;
    if constexpr (sizeof(FT) == 4) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const float4 v = *reinterpret_cast<const float4 *>(row + 64 * i + 4 * li);
            c.q[2 * i] = plan_v2f{v.x, v.y}; c.q[2 * i + 1] = plan_v2f{v.z, v.w};
        }
#pragma unroll
        for (int i = 0; i < N1; ++i) c.s[i] = row[64 * NQ + 16 * i + li];
    } else {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const uint2 v = *reinterpret_cast<const uint2 *>(row + 64 * i + 4 * li);
            c.q[2 * i] = plan_v2f{bf16_lo(v.x), bf16_hi(v.x)}; c.q[2 * i + 1] = plan_v2f{bf16_lo(v.y), bf16_hi(v.y)};
        }
#pragma unroll
        for (int i = 0; i < N1; ++i) c.s[i] = __uint_as_float((unsigned)row[64 * NQ + 16 * i + li] << 16);
    }
}
template <int S>
__device__ __forceinline__ void store_row(float *row, int li, const Acc<S> &c) {      // fp32 row in LDS or global, same channel layout
    constexpr int NQ = S / 4, N1 = S % 4;
#pragma unroll
    for (int i = 0; i < NQ; ++i)
        *reinterpret_cast<float4 *>(row + 64 * i + 4 * li) = make_float4(c.q[2 * i].x, c.q[2 * i].y, c.q[2 * i + 1].x, c.q[2 * i + 1].y);
#pragma unroll
    for (int i = 0; i < N1; ++i) row[64 * NQ + 16 * i + li] = c.s[i];
}
template <int S>
__device__ __forceinline__ void load_row(const float *row, int li, Acc<S> &c) {
    load_ctx<float, S>(row, li, c);
}

// The same piece through a buffer descriptor: byte offset = voff (per lane: the pair's first row) + soff (wave-uniform: the row
// inside the block).  One 32-bit VGPR addresses all 16 rows of a pair -- sixteen 64-bit row pointers cost the kernel 30 VGPRs
// and a wave per SIMD.
template <typename FT, int S>
__device__ __forceinline__ void load_ctx_buf(const __amdgpu_buffer_rsrc_t &rs, unsigned voff, unsigned soff, int li, Acc<S> &c) {
    constexpr int NQ = S / 4, N1 = S % 4;
    if constexpr (sizeof(FT) == 4) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const mmt_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (unsigned)(256 * i + 16 * li), soff, 0);
            c.q[2 * i] = plan_v2f{__uint_as_float(v.x), __uint_as_float(v.y)}; c.q[2 * i + 1] = plan_v2f{__uint_as_float(v.z), __uint_as_float(v.w)};
        }
#pragma unroll
        for (int i = 0; i < N1; ++i)
#ifdef PLAN_EXP_NOEXTRA      // ablation build: the channels past the float4 pieces are not loaded (results are wrong)
            c.s[i] = __uint_as_float(voff + soff);
#else
            c.s[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff + (unsigned)(256 * NQ + 64 * i + 4 * li), soff, 0));
#endif
    } else {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + (unsigned)(128 * i + 8 * li), soff, 0);
            c.q[2 * i] = plan_v2f{bf16_lo(v.x), bf16_hi(v.x)}; c.q[2 * i + 1] = plan_v2f{bf16_lo(v.y), bf16_hi(v.y)};
        }
#pragma unroll
        for (int i = 0; i < N1; ++i)
#ifdef PLAN_EXP_NOEXTRA
            c.s[i] = __uint_as_float(voff + soff);
#else
            c.s[i] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, voff + (unsigned)(128 * NQ + 32 * i + 2 * li), soff, 0) << 16);
#endif
    }
}

// the depths of 4 consecutive bins of one pixel, starting at element e of the depth tensor.  The plan shifts a run at the end
// of a ray back (lss_plan_core.h, phase_place), so the four bins always lie inside the pixel's D bins.
template <typename FT>
__device__ __forceinline__ float4 load_depth4(const __amdgpu_buffer_rsrc_t &rs, unsigned e) {
    if constexpr (sizeof(FT) == 4) {
        const mmt_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, e * 4u, 0, 0);        // (dword-aligned, not 16-byte aligned)
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    } else {
        // bf16: e may be odd (a 2-byte aligned address).  Even: the 8 bytes themselves.  Odd: the two dwords from e - 1 and the
        // fourth element on its own (every access inside the pixel's four bins, or one element in front of them)
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const bool odd = (e & 1u) != 0u;
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (e & ~1u) * 2u, 0, 0);
        const unsigned v2 = odd ? (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, (e + 3u) * 2u, 0, 0) : 0u;
        const unsigned a0 = odd ? (v.x >> 16) | (v.y << 16) : v.x, a1 = odd ? (v.y >> 16) | (v2 << 16) : v.y;
        return make_float4(bf16_lo(a0), bf16_hi(a0), bf16_lo(a1), bf16_hi(a1));
    }
}

// the depths of kWindowBins consecutive bins of one pixel row from element e of the depth tensor on (bf16: e is even), as fp32.
// A window that leaves the pixel's D bins reads the next pixel's (or, past the tensor, the descriptor's zeros): the masks of
// the runs never select those.
template <typename FT>
__device__ __forceinline__ void load_window(const __amdgpu_buffer_rsrc_t &rs, unsigned e, float (&win)[kWindowBins]) {
    static_assert(kWindowBins == 8, "two 16-byte loads (fp32) / one (bf16)");
    if constexpr (sizeof(FT) == 4) {
        const mmt_u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, e * 4u, 0, 0);          // (dword-aligned, not 16-byte aligned)
        const mmt_u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, e * 4u + 16u, 0, 0);
        win[0] = __uint_as_float(a.x); win[1] = __uint_as_float(a.y); win[2] = __uint_as_float(a.z); win[3] = __uint_as_float(a.w);
        win[4] = __uint_as_float(b.x); win[5] = __uint_as_float(b.y); win[6] = __uint_as_float(b.z); win[7] = __uint_as_float(b.w);
    } else {
        // bf16: w0 is even, and so is the pixel's first element unless D is odd -- then e may sit at an odd 2-byte address: the eight
        // elements from e - 1 on (dword-aligned) and the ninth on its own, moved down by one
        const bool odd = (e & 1u) != 0u;
        const mmt_u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (e & ~1u) * 2u, 0, 0);
        const unsigned last = odd ? (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, (e + 7u) * 2u, 0, 0) : 0u;
        const unsigned x0 = odd ? (a.x >> 16) | (a.y << 16) : a.x, x1 = odd ? (a.y >> 16) | (a.z << 16) : a.y;
        const unsigned x2 = odd ? (a.z >> 16) | (a.w << 16) : a.z, x3 = odd ? (a.w >> 16) | (last << 16) : a.w;
        win[0] = bf16_lo(x0); win[1] = bf16_hi(x0); win[2] = bf16_lo(x1); win[3] = bf16_hi(x1);
        win[4] = bf16_lo(x2); win[5] = bf16_hi(x2); win[6] = bf16_lo(x3); win[7] = bf16_hi(x3);
    }
}

#define PLAN_FOR16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// 4 waves per SIMD for C <= 80 (a 128-register cap: two loop-invariant values are spilled and reloaded once per unit, outside the
// pair loop) against 3 without: 22.8 against 27.0 us at BASELINE configs[3] (profiles/r05_kbench_camera_cfg4.json)
#ifndef PLAN_FWD_WAVES
#define PLAN_FWD_WAVES 4
#endif
template <typename FT, int S>
__global__ __launch_bounds__(kFwdThreads, (S <= 5 ? PLAN_FWD_WAVES : 1)) void lss_plan_fwd(FwdArgs a) {
    extern __shared__ __align__(16) unsigned char plan_lds[];
    constexpr int C = 16 * S;
    const PlanArgs &p = a.p;
    const Dims &d = p.d;
    const int tid = threadIdx.x, g = tid >> 4, li = tid & 15;
#ifdef PLAN_STAMPS     // diagnostic build (tools/build_variant.py ... -DPLAN_STAMPS): the column-summary argument receives 8 words per workgroup
    unsigned long long *stamps = reinterpret_cast<unsigned long long *>(a.summary_out) + (int64_t)blockIdx.x * 16;
    a.summary_out = nullptr;
    int st_units = 0;
#define PLAN_STAMP(i) do { if (tid == 0 && stamps && (st_units == 0 || (i) >= 5)) stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
    if (tid == 0 && stamps) { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); stamps[7] = xcc & 15u; }
#else
#define PLAN_STAMP(i) do { } while (0)
#endif
    PLAN_STAMP(0);
    // ---- which sample, which units (records of a learnt plan / tiles for the brute-force path).  A calibration's units come
    // in kGroups groups of neighbouring tiles, heaviest first inside a group.  With 8 % B == 0 the sample's workgroups sit on
    // xps = 8 / B XCDs and each XCD takes kGroups / xps whole groups (their context rows meet in its L2); its workgroups walk
    // those groups interleaved, so that every group's heavy units are dealt first.
    int b, first, stride, g0, ng;
    const int wg = blockIdx.x;
    if (a.xps > 0) {
        const int xcd = wg & 7;
        b = xcd / a.xps;
        ng = kGroups / a.xps; g0 = (xcd % a.xps) * ng;
        first = wg >> 3; stride = a.W / a.xps;
    } else {
        b = wg / a.W; first = wg - b * a.W; stride = a.W;
        g0 = 0; ng = kGroups;
    }
    const int *vd = reinterpret_cast<const int *>(p.cache + kVerdictOff) + b * (int)(sizeof(Verdict) / 4);       // (read word by word: a struct indexed at run time would live in scratch memory)
    const int slot = vd[0], vstate = vd[2];
    if ((unsigned)slot >= (unsigned)p.nslots) return;       // (a cache nobody prepared: nothing to go by -- the entry point's contract, not a fault)
    if (vstate == kStateEmpty) {
        // MMT_LSS_PLAN_PREPARED with a verdict that no lookup of THIS batch left: the slot's summary was never built for these
        // matrices, and serving the sample from it would write a wrong map without any sign.  The sample's part of the map is
        // left as the caller allocated it and the header counts the event (mmt_lss_plan_cache_counters, counters[6]; LSSFPN's
        // lazy read-back raises on it): loud instead of silently wrong.
        if (tid == 0 && first == 0 && g0 == 0) atomicAdd(&reinterpret_cast<CacheHeader *>(p.cache)->stale, 1u);
        return;
    }
    const bool brute = vstate != kStateReady || a.force_brute;
    __shared__ int s_gs[kGroups + 1];                        // first unit of every group (indexed at run time: LDS, not registers)
    if (tid <= kGroups) s_gs[tid] = brute ? group_begin(d, tid) : vd[4 + tid];
    __syncthreads();
    int gmax = 0;
    for (int i = 0; i < ng; ++i) { const int len = s_gs[g0 + i + 1] - s_gs[g0 + i]; gmax = len > gmax ? len : gmax; }
    const int nunits = gmax * ng;                           // virtual units: u -> group u % ng, rank u / ng (holes where a group is shorter)
    auto unit_of = [&](int u) __attribute__((always_inline)) {
        const int gi = u % ng, r = u / ng;
        const int lo = s_gs[g0 + gi], len = s_gs[g0 + gi + 1] - lo;
        return r < len ? lo + r : -1;
    };
    const unsigned char *sbase = p.cache + p.slots_off + (int64_t)slot * p.slot_bytes;
    const FT *context = reinterpret_cast<const FT *>(a.context);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.depth), 0, a.depth_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.context), 0, a.ctx_bytes, 0x00020000);

    // ---- the batch's column summary for the backward: every workgroup of the sample copies a slice of the slot's
    if (a.summary_out != nullptr) {
        const int64_t n8 = (int64_t)d.strips * d.D;
        const int wl = a.xps > 0 ? (g0 / ng) * stride + first : first;
        const int2 *src = reinterpret_cast<const int2 *>(sbase + p.summary_off);
        int2 *dst = a.summary_out + (int64_t)b * n8;
        for (int64_t i = (int64_t)wl * kFwdThreads + tid; i < n8; i += (int64_t)a.W * kFwdThreads) dst[i] = src[i];
    }

    PLAN_STAMP(1);
    if (!brute) {
        // LDS: two record buffers (the next unit's record is fetched while this one is worked on) | the partial rows
        float *partial = reinterpret_cast<float *>(plan_lds + 2 * kJobBytes);
        const unsigned char *records = sbase + p.records_off;
        int ucur = first;
        auto next_unit = [&]() __attribute__((always_inline)) {        // the next unit of this workgroup's share, -1 when it is through
            while (ucur < nunits) {
                const int j = unit_of(ucur);
                ucur += stride;
                if (j >= 0) return j;
            }
            return -1;
        };
        int job = next_unit();
        if (job >= 0 && tid < kJobBytes / 16) reinterpret_cast<uint4 *>(plan_lds)[tid] = reinterpret_cast<const uint4 *>(records + (int64_t)job * kJobBytes)[tid];
        int cur = 0;
        bool in_chain = false;
        Acc<S> chain_acc;                      // a cell fed by more runs than a record holds: summed over its chain of records
        chain_acc.zero();
        while (job >= 0) {
            __syncthreads();                   // the record in buffer `cur` is complete; the other buffer and the partial rows are free
            const unsigned char *rec_lds = plan_lds + cur * kJobBytes;
            PLAN_STAMP(2);
            const JobHeader hdr = *reinterpret_cast<const JobHeader *>(rec_lds);
            if ((hdr.chain & kChainLink) && !in_chain) {            // a link of a chain met as a unit: the workgroup that met the head does it
                job = next_unit();
                __syncthreads();
                if (job >= 0 && tid < kJobBytes / 16) reinterpret_cast<uint4 *>(plan_lds + cur * kJobBytes)[tid] = reinterpret_cast<const uint4 *>(records + (int64_t)job * kJobBytes)[tid];
                continue;
            }
            const bool more = (hdr.chain & kChainMore) != 0u;
            const int njob = more ? job + 1 : next_unit();
            const uint8_t *cb = rec_lds + kJobCellBeginOff;
            const PairRec *pairs = reinterpret_cast<const PairRec *>(rec_lds + kJobPairsOff);
            const RunRec *runs = reinterpret_cast<const RunRec *>(rec_lds + kJobRunsOff);
            // ---- phase 1: a lane group per pair -- 16 context rows in registers, one partial row per run
#ifdef PLAN_EXP_NOPAIRS    // ablation build: no pairs at all (results are wrong)
            if (a.W == 123457)
#endif
#pragma unroll 1
            for (int pi = g; pi < hdr.npairs; pi += kFwdThreads / 16) {
                const PairRec pr = pairs[pi];
                const int n = pr.col / d.fW, w = pr.col - n * d.fW;
                const int bn = b * d.N + n, r0 = pr.rb * 16;
                const int myrow = (r0 + li) < d.fH ? (r0 + li) : d.fH - 1;
                const unsigned pix = (unsigned)((((int64_t)bn * d.fH + myrow) * d.fW + w) * d.D);
                // the pair's depth window: bins [w0, w0 + kWindowBins) of this lane's image row (w0 even: a dword-aligned address in bf16
                // too).  ONE load per pair where round 5 made one (bf16: two) per run; issued in front of the context rows (loads return
                // in order, and the weights are needed first).
                const unsigned e0 = pix + pr.w0;
                float win[kWindowBins];
#ifdef PLAN_EXP_NODEPTH    // ablation build: no depth traffic (results are wrong)
#pragma unroll
                for (int k = 0; k < kWindowBins; ++k) win[k] = __uint_as_float(0x3f800000u | ((e0 + k) & 0xFFFFu));
#else
                load_window<FT>(drs, e0, win);
#endif
                Acc<S> ctx[16];
                const unsigned row_bytes = (unsigned)(d.fW * C) * (unsigned)sizeof(FT);
                const unsigned cvoff = (unsigned)((((int64_t)bn * d.fH + r0) * d.fW + w) * C) * (unsigned)sizeof(FT);
#pragma unroll
                for (int h = 0; h < 16; ++h) {
#ifdef PLAN_EXP_NOCTX      // (ablation build: no context traffic, results are wrong)
                    ctx[h].zero(); ctx[h].q[0].x = __uint_as_float(cvoff + h);
#else
                    load_ctx_buf<FT, S>(crs, cvoff, (unsigned)h * row_bytes, li, ctx[h]);       // (rows past the image: zeroed below)
#endif
                }
#ifdef PLAN_STAMPS
                if (pi == g) { PLAN_STAMP(8); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PLAN_STAMP(9); }
#endif
                if (r0 + 16 > d.fH) {          // rows past the image: their weights are zero, and so must their context be (0 * NaN)
#pragma unroll
                    for (int h = 0; h < 16; ++h) if (r0 + h >= d.fH) ctx[h].zero();
                }
                // the runs: a run's weight of image row li = the window's bins under the run's row masks (eight 16-bit masks at the
                // window's positions, shifted there by the builder: fixed positions, no register indexed at run time); the next run's
                // words are fetched from the record while this one's sixteen rows are multiplied
                const RunRec *rr = runs + pr.run0;
                uint2 mlo = *reinterpret_cast<const uint2 *>(&rr->wlo), mhi = *reinterpret_cast<const uint2 *>(&rr->whi);
                int pslot = rr->pslot;
#pragma unroll 1     // (unrolled four times with early exits: 2 % faster at BASELINE configs[3], 6 % slower at configs[4])
                for (int r = 0; r < pr.nruns; ++r) {
                    const unsigned t0 = mlo.x >> li, t1 = mlo.y >> li, t2 = mhi.x >> li, t3 = mhi.y >> li;
                    const int ps = pslot;
                    if (r + 1 < pr.nruns) {
                        ++rr;
                        mlo = *reinterpret_cast<const uint2 *>(&rr->wlo); mhi = *reinterpret_cast<const uint2 *>(&rr->whi);
                        pslot = rr->pslot;
                    }
                    float wr = (t0 & 1u) ? win[0] : 0.f;
                    wr += (t0 & 0x10000u) ? win[1] : 0.f;
                    wr += (t1 & 1u) ? win[2] : 0.f;
                    wr += (t1 & 0x10000u) ? win[3] : 0.f;
                    wr += (t2 & 1u) ? win[4] : 0.f;
                    wr += (t2 & 0x10000u) ? win[5] : 0.f;
                    wr += (t3 & 1u) ? win[6] : 0.f;
                    wr += (t3 & 0x10000u) ? win[7] : 0.f;
                    Acc<S> acc;
                    acc.zero();
#ifdef PLAN_EXP_NOFMA      // ablation build: one row's products instead of sixteen (results are wrong)
#define PLAN_STEP(h) if (h == 0) acc.fma(row_bcast<h>(wr), ctx[h]); else acc.q[0].x += ctx[h].q[0].x + ctx[h].q[1].y + ctx[h].s[0];
#else
#define PLAN_STEP(h) acc.fma(row_bcast<h>(wr), ctx[h]);
#endif
                    PLAN_FOR16(PLAN_STEP)
#undef PLAN_STEP
                    store_row<S>(partial + ps * C, li, acc);
                }
            }
#ifdef PLAN_STAMPS
            PLAN_STAMP(10);
#endif
            // the next record: requested now (behind the pairs, whose registers it would otherwise take four of), parked in LDS
            // when this unit's cells are stored
            uint4 pre = make_uint4(0u, 0u, 0u, 0u);
            const bool have = njob >= 0 && tid < kJobBytes / 16;
            if (have) pre = reinterpret_cast<const uint4 *>(records + (int64_t)njob * kJobBytes)[tid];
            __syncthreads();
            PLAN_STAMP(3);
#ifdef PLAN_STAMPS
            if (tid == 0 && stamps && st_units == 0) stamps[6] = ((unsigned long long)hdr.npairs << 32) | hdr.nruns;
#endif
            // ---- phase 2: a lane group per cell -- its partial rows in plan order, one store
            const int tx0 = (hdr.tile % d.tiles_x) * kTile, ty0 = (hdr.tile / d.tiles_x) * kTile;
            if (hdr.chain != kChainNone) {
                if (g == 0) {
                    for (int q = 0; q < hdr.nruns; ++q) { Acc<S> t; load_row<S>(partial + q * C, li, t); chain_acc.add(t); }
                    const int x = tx0 + z_x(hdr.c0), y = ty0 + z_y(hdr.c0);
                    if (!more && x < d.nx && y < d.ny) store_row<S>(a.out + (((int64_t)b * d.ny + y) * d.nx + x) * C, li, chain_acc);
                }
            } else {
#pragma unroll 1
                for (int c = g; c < hdr.ncells; c += kFwdThreads / 16) {
                    const int l = hdr.c0 + c;
                    const int x = tx0 + z_x(l), y = ty0 + z_y(l);
                    Acc<S> acc;
                    acc.zero();
                    for (int q = cb[c]; q < cb[c + 1]; ++q) { Acc<S> t; load_row<S>(partial + q * C, li, t); acc.add(t); }
#ifdef PLAN_EXP_NOSTORE    // ablation build: no output (results are wrong)
                    if (acc.q[0].x == 1234.5f)
#endif
                    if (x < d.nx && y < d.ny) store_row<S>(a.out + (((int64_t)b * d.ny + y) * d.nx + x) * C, li, acc);
                }
            }
            PLAN_STAMP(4);
#ifdef PLAN_STAMPS
            ++st_units;
#endif
            in_chain = more;
            if (!more) chain_acc.zero();
            if (have) reinterpret_cast<uint4 *>(plan_lds + (cur ^ 1) * kJobBytes)[tid] = pre;
            job = njob;
            cur ^= 1;
        }
        PLAN_STAMP(5);
        return;
    }

    // ---- brute force: a calibration without a plan (its runs or jobs overflow the slot), straight from the slot's summary.
    // A workgroup takes a tile; a lane group owns the cells whose index in the tile is g + 16 * pass, one pass at a time;
    // the entries of the summary that can reach the tile are compacted in index order, so the sums are ordered as well.
    if (tid == 0 && first == 0 && g0 == 0) atomicAdd(&reinterpret_cast<CacheHeader *>(p.cache)->brute, 1u);
    int *list = reinterpret_cast<int *>(plan_lds);
    int *wcount = list + kFwdThreads;
    const int2 *summary = reinterpret_cast<const int2 *>(sbase + p.summary_off);
    const int total = d.strips * d.D;
    const int wave = tid >> 6, lane = tid & 63;
    for (int u = first; u < nunits; u += stride) {
        const int tile = unit_of(u);
        if (tile < 0) continue;
        const int tx0 = (tile % d.tiles_x) * kTile, ty0 = (tile / d.tiles_x) * kTile;
#pragma unroll 1
        for (int pass = 0; pass < kTileCells / kFwdGroups; ++pass) {
            const int mylocal = g + kFwdGroups * pass;
            Acc<S> acc;
            acc.zero();
#pragma unroll 1
            for (int base = 0; base < total; base += kFwdThreads) {
                const int e = base + tid;
                bool cand = false;
                if (e < total) {
                    const int2 sv = summary[e];
                    const unsigned zm = (unsigned)sv.y & 0xFFFFu;
                    if (sv.y & mmt::kSummaryUniform) {
                        if (zm != 0u && sv.x >= 0) {
                            const int lx = (sv.x & 0xFFFF) - tx0, ly = (sv.x >> 16) - ty0;
                            cand = (unsigned)lx < (unsigned)kTile && (unsigned)ly < (unsigned)kTile && (ly * kTile + lx) / kFwdGroups == pass;
                        }
                    } else cand = zm != 0u;
                }
                const unsigned long long m = __ballot(cand);
                __syncthreads();                 // (the list of the previous round has been consumed)
                if (lane == 0) wcount[wave] = __popcll(m);
                __syncthreads();
                int off = 0, ncand = 0;
#pragma unroll
                for (int q = 0; q < kFwdThreads / 64; ++q) { const int cq = wcount[q]; off += q < wave ? cq : 0; ncand += cq; }
                if (cand) list[off + __popcll(m & ((1ull << lane) - 1ull))] = e;
                __syncthreads();
#pragma unroll 1
                for (int q = 0; q < ncand; ++q) {
                    const int e2 = list[q];
                    const int2 sv = summary[e2];
                    const unsigned zm = (unsigned)sv.y & 0xFFFFu;
                    const int s = e2 / d.D, bin = e2 - s * d.D;
                    const int w = s % d.fW, rb = (s / d.fW) % d.nb, n = s / (d.fW * d.nb);
                    const int bn = b * d.N + n, r0 = rb * 16;
                    const int myrow = (r0 + li) < d.fH ? (r0 + li) : d.fH - 1;
                    const int bin0 = bin + kRunBins > d.D ? d.D - kRunBins : bin;       // (the four bins stay inside the pixel's D)
                    const float4 dq = load_depth4<FT>(drs, (unsigned)((((int64_t)bn * d.fH + myrow) * d.fW + w) * d.D + bin0));
                    const int dsel = bin - bin0;
                    const float dbin = dsel == 0 ? dq.x : (dsel == 1 ? dq.y : (dsel == 2 ? dq.z : dq.w));
                    const float dep = ((zm >> li) & 1u) ? dbin : 0.f;
                    if (sv.y & mmt::kSummaryUniform) {
                        const int local = ((sv.x >> 16) - ty0) * kTile + ((sv.x & 0xFFFF) - tx0);
                        if (local == mylocal) {
#define PLAN_STEP(h) { if (r0 + h < d.fH) { Acc<S> cr; load_ctx<FT, S>(context + (((int64_t)bn * d.fH + r0 + h) * d.fW + w) * C, li, cr); acc.fma(row_bcast<h>(dep), cr); } }
                            PLAN_FOR16(PLAN_STEP)
#undef PLAN_STEP
                        }
                    } else {
                        float cm[12];
#pragma unroll
                        for (int k = 0; k < 12; ++k) cm[k] = p.combine[(int64_t)bn * 16 + k];
                        const mmt_cam_column cc = mmt_cam_column_make(cm, p.fu[w], p.fd[bin]);
#define PLAN_STEP(h) { if (((zm >> h) & 1u) && r0 + h < d.fH) { int gx, gy; const bool in = mmt_cam_row_xy(cc, p.fv[r0 + h], p.q, d.nx, d.ny, gx, gy); \
                            const float dh = row_bcast<h>(dep); \
                            if (in && (gy - ty0) * kTile + (gx - tx0) == mylocal && (unsigned)(gx - tx0) < (unsigned)kTile && (unsigned)(gy - ty0) < (unsigned)kTile) { \
                                Acc<S> cr; load_ctx<FT, S>(context + (((int64_t)bn * d.fH + r0 + h) * d.fW + w) * C, li, cr); acc.fma(dh, cr); } } }
                        PLAN_FOR16(PLAN_STEP)
#undef PLAN_STEP
                    }
                }
            }
            const int x = tx0 + (mylocal & 7), y = ty0 + (mylocal >> 3);
            if (x < d.nx && y < d.ny) store_row<S>(a.out + (((int64_t)b * d.ny + y) * d.nx + x) * C, li, acc);
        }
    }
}

int plan_shape_ok(const char *what, int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz, bool quiet) {
    Dims d;
    make_dims(N > 0 ? N : 1, D > 0 ? D : 1, fH > 0 ? fH : 1, fW > 0 ? fW : 1, nx > 0 ? nx : 1, ny > 0 ? ny : 1, &d);
    const bool ok = B > 0 && B <= kPlanMaxB && N > 0 && N <= kPlanMaxN && D > 0 && fH > 0 && fW > 0 && nx > 0 && ny > 0 && nz > 0 && dims_ok(d) &&
                    (C == 64 || C == 80 || C == 128) && (int64_t)B * N * fH * fW * D * 4 < (1ll << 31) && (int64_t)B * N * fH * fW * C * 4 < (1ll << 32) &&      /* context rows: 32-bit BYTE offsets (fp32 worst case) */
                    (int64_t)B * ny * nx * C < (1ll << 31);
    if (!ok && !quiet)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: the plan form takes B <= %d, N <= %d, C in {64, 80, 128}, D <= 2047, fH <= 512, N * fW <= 65535, tensors "
                         "below 2^31 elements and a context tensor below 2^30 (B=%d N=%d D=%d fH=%d fW=%d C=%d grid %d x %d x %d)", what, kPlanMaxB, kPlanMaxN, B, N, D, fH, fW, C, nx, ny, nz);
    return ok ? MMT_OK : MMT_ERR_BAD_SHAPE;
}

int fill_plan_args(const char *what, int B, int N, int D, int fH, int fW, int nx, int ny, int nz, const float *combine, const float *fu, const float *fv,
                   const float *fd, const float *vc, const float *vs, void *cache, int64_t cache_bytes, PlanArgs *p) {
    if (!combine || !fu || !fv || !fd || !vc || !vs || !cache) return mmt::fail(MMT_ERR_NULL_POINTER, "%s: a NULL pointer among the geometry operands / the plan cache", what);
    if (((uintptr_t)cache & 255) != 0) return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: the plan cache must be 256-byte aligned", what);
    Layout l;
    make_layout(N, D, fH, fW, nx, ny, 1, &l);
    const int64_t fixed = kHdrBytes + (int64_t)kBuildPar * l.scratch_bytes;
    const int64_t fit = (cache_bytes - fixed) / l.slot_bytes;
    if (cache_bytes < fixed || fit < B)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: the plan cache (%lld bytes) holds %lld calibrations, the call has %d samples (mmt_lss_plan_cache_bytes)", what,
                         (long long)cache_bytes, (long long)(cache_bytes < fixed ? 0 : fit), B);
    const int slots = fit > 256 ? 256 : (int)fit;
    make_layout(N, D, fH, fW, nx, ny, slots, &l);
    p->cache = static_cast<unsigned char *>(cache);
    p->summary_off = l.summary_off; p->records_off = l.records_off; p->slot_bytes = l.slot_bytes; p->scratch_bytes = l.scratch_bytes;
    p->slots_off = l.slots_off; p->scratch_off = l.scratch_off;
    {   // batch snapshots of the lookup's fast path: header (256) + matrices + verdicts per batch, at the head of the build scratch
        // (sized for the largest batch this cache takes: as many samples as it has slots, its N cameras)
        const int64_t bmax = slots < kPlanMaxB ? slots : kPlanMaxB;
        const int64_t stride = 256 + up256(bmax * N * 64) + up256(bmax * (int64_t)sizeof(Verdict));
        p->snap_stride = (int64_t)kSnaps * stride <= l.scratch_bytes ? stride : 0;
    }
    p->d = l.d; p->B = B; p->nz = nz; p->nslots = slots;
    p->combine = combine; p->fu = fu; p->fv = fv; p->fd = fd;
    mmt::make_cam_grid(vc, vs, &p->q);
    mmt::make_cam_range(&p->q, nx, ny, nz);
    // what the learnt plans depend on besides the matrices and the axes' contents (hashed on the device)
    int words[20] = {N, D, fH, fW, nx, ny, nz, kMaxRuns, kMaxPairRuns, kRunBins, kTile, kJobBytes, slots, 2 /* layout version */, kWindowBins};
    memcpy(words + 14, p->q.lo, 12);
    memcpy(words + 17, p->q.vs, 12);
    uint64_t h = 0xCBF29CE484222325ull;
    for (int w : words) { h ^= (uint32_t)w; h *= 0x100000001B3ull; h ^= h >> 29; }
    p->sig_lo = (unsigned)h | 1u; p->sig_hi = (unsigned)(h >> 32);
    return MMT_OK;
}

unsigned next_token() {
    static std::atomic<unsigned> counter{0x5EED0001u};
    return counter.fetch_add(2u);                  // (odd: never 0, never the previous launch's; the launch also uses token + 1)
}

void launch_prepare(mmt::TimedSeq &seq, const PlanArgs &p, hipStream_t st, bool last) {
    PlanArgs q = p;
    q.token = next_token();
    const int g = p.B < kBuildPar ? p.B : kBuildPar;
    seq.launch(last, lss_plan_lookup, dim3((unsigned)(1 + g)), dim3(kBuildThreads), 0, st, q);
}

static int plan_fwd_wgs() {
    // 2 048 over the batch: at BASELINE configs[3] (1 280 records) the same as 1 024 (22.1 / 22.2 us), at configs[4] (1 844 records: a
    // record per workgroup) 46.2 against 50.2 us, at the reference's native frustum 52.4 against 60.2 (tools/scratch/plan_wgs.sh)
    static const char *env = getenv("MMT_PLAN_WGS");          // experiments only
    return (env && atoi(env) > 0) ? atoi(env) : 2048 * 256 / kFwdThreads;
}

template <typename FT>
int plan_forward_impl(const char *what, int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz, const float *combine, const float *fu,
                      const float *fv, const float *fd, const float *vc, const float *vs, const FT *depth, const FT *context, float *out,
                      int32_t *column_summary, void *cache, int64_t cache_bytes, int flags, hipStream_t st) {
    if (flags & ~(MMT_LSS_PIXEL_MAJOR | MMT_LSS_PLAN_PREPARED | MMT_LSS_PLAN_BRUTE))
        return mmt::fail(MMT_ERR_BAD_FLAG, "%s: unknown flag bits 0x%x", what, flags);
    if (!(flags & MMT_LSS_PIXEL_MAJOR)) return mmt::fail(MMT_ERR_BAD_FLAG, "%s: the plan form reads depth pixel-major (MMT_LSS_PIXEL_MAJOR)", what);
    if (const int rc = plan_shape_ok(what, B, N, D, fH, fW, C, nx, ny, nz, false)) return rc;
    if ((((uintptr_t)context | (uintptr_t)out) & 15) != 0 || ((uintptr_t)depth & 3) != 0 || (column_summary && ((uintptr_t)column_summary & 7) != 0))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: context / out must be 16-byte aligned, the column summary 8-byte, depth 4-byte aligned", what);
    FwdArgs a = {};
    if (const int rc = fill_plan_args(what, B, N, D, fH, fW, nx, ny, nz, combine, fu, fv, fd, vc, vs, cache, cache_bytes, &a.p)) return rc;
    a.depth = depth; a.context = context; a.out = out; a.summary_out = reinterpret_cast<int2 *>(column_summary);
    a.C = C; a.force_brute = (flags & MMT_LSS_PLAN_BRUTE) ? 1 : 0;
    a.depth_bytes = (unsigned)((int64_t)B * N * fH * fW * D * (int64_t)sizeof(FT));
    a.ctx_bytes = (unsigned)((int64_t)B * N * fH * fW * C * (int64_t)sizeof(FT));
    const Dims &d = a.p.d;
    // persistent workgroups: two rounds of what the chip holds at once (4 per CU: 35 KB of LDS, 128 VGPRs), every one walks its
    // share of the sample's units with the next record in flight
    int W = plan_fwd_wgs() / B;
    if (W > d.jobs_cap) W = d.jobs_cap;
    W = (W + 7) & ~7;
    if (W < 8) W = 8;
    a.W = W;
    a.xps = (B <= 8 && 8 % B == 0) ? 8 / B : 0;
    mmt::TimedSeq seq;
    if (!(flags & MMT_LSS_PLAN_PREPARED)) launch_prepare(seq, a.p, st, false);
    const size_t lds = 2 * (size_t)kJobBytes + (size_t)kMaxRuns * C * 4;
    const dim3 grid((unsigned)((int64_t)B * W)), blk(kFwdThreads);
    if (C == 80) seq.launch(true, lss_plan_fwd<FT, 5>, grid, blk, lds, st, a);
    else if (C == 64) seq.launch(true, lss_plan_fwd<FT, 4>, grid, blk, lds, st, a);
    else seq.launch(true, lss_plan_fwd<FT, 8>, grid, blk, lds, st, a);
    mmt::lss_note_forward_family(MMT_LSS_FAMILY_PLAN | MMT_LSS_FAMILY_CAMERA);
    return mmt::check_launch(what);
}

}  // namespace

extern "C" int mmt_lss_plan_supported(int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz) {
    return plan_shape_ok("lss_plan_supported", B, N, D, fH, fW, C, nx, ny, nz, true) == MMT_OK ? 1 : 0;
}

extern "C" int64_t mmt_lss_plan_cache_bytes(int N, int D, int fH, int fW, int nx, int ny, int slots) {
    if (N <= 0 || N > kPlanMaxN || D <= 0 || fH <= 0 || fW <= 0 || nx <= 0 || ny <= 0 || slots <= 0) return 0;
    Layout l;
    make_layout(N, D, fH, fW, nx, ny, slots > 256 ? 256 : slots, &l);
    if (!dims_ok(l.d)) return 0;
    return l.total;
}

extern "C" int mmt_lss_plan_prepare(int B, int N, int D, int fH, int fW, int nx, int ny, int nz, const float *combine, const float *frustum_u,
                                    const float *frustum_v, const float *frustum_d, const float *voxel_coord_host, const float *voxel_size_host,
                                    void *plan_cache, int64_t plan_cache_bytes, void *stream) {
    if (const int rc = plan_shape_ok("lss_plan_prepare", B, N, D, fH, fW, 64, nx, ny, nz, false)) return rc;
    PlanArgs p = {};
    if (const int rc = fill_plan_args("lss_plan_prepare", B, N, D, fH, fW, nx, ny, nz, combine, frustum_u, frustum_v, frustum_d, voxel_coord_host,
                                      voxel_size_host, plan_cache, plan_cache_bytes, &p)) return rc;
    mmt::TimedSeq seq;
    launch_prepare(seq, p, (hipStream_t)stream, true);
    return mmt::check_launch("lss_plan_prepare");
}

extern "C" int mmt_depth_softmax_forward_plan_prepare(int64_t pixels, int D, const void *logits, int64_t logit_row_stride, int logits_dtype,
                                                      float *probs, const float *oracle, int64_t oracle_row_stride, void *depth_used, int used_dtype,
                                                      int B, int N, int fH, int fW, int nx, int ny, int nz, const float *combine,
                                                      const float *frustum_u, const float *frustum_v, const float *frustum_d,
                                                      const float *voxel_coord_host, const float *voxel_size_host, void *plan_cache,
                                                      int64_t plan_cache_bytes, void *stream) {
    const char *who = "depth_softmax_forward_plan_prepare";
    if (int rc = softmax_common_check(who, pixels, D, logits_dtype, used_dtype)) return rc;
    if (const int rc = plan_shape_ok(who, B, N, D, fH, fW, 64, nx, ny, nz, false)) return rc;
    if (pixels != (int64_t)B * N * fH * fW)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: %lld softmax rows for a batch of %d x %d cameras of %d x %d pixels", who, (long long)pixels, B, N, fH, fW);
    MMT_REQUIRE_PTR(logits);
    MMT_REQUIRE_PTR(probs);
    if (logit_row_stride < D || (oracle != nullptr && oracle_row_stride < D)) return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: row strides must be >= D", who);
    if (oracle != nullptr && depth_used == nullptr) return mmt::fail(MMT_ERR_NULL_POINTER, "%s: an oracle needs a depth_used output", who);
    PlanArgs p = {};
    if (const int rc = fill_plan_args(who, B, N, D, fH, fW, nx, ny, nz, combine, frustum_u, frustum_v, frustum_d, voxel_coord_host, voxel_size_host,
                                      plan_cache, plan_cache_bytes, &p)) return rc;
    const bool lb = logits_dtype == MMT_DTYPE_BF16, ub = used_dtype == MMT_DTYPE_BF16;
    const bool vec4 = D % 4 == 0 && softmax_aligned(logits, logit_row_stride, lb ? 2 : 4) && softmax_aligned(probs, D, 4) &&
                      softmax_aligned(oracle, oracle_row_stride, 4) && softmax_aligned(depth_used, D, ub ? 2 : 4);
    if (!vec4)      // rows the 16-byte pieces cannot take: the caller makes the two launches (mmt_depth_softmax_rides_lookup says so beforehand)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: rows of D %% 4 == 0 elements on 16-byte (fp32) / 8-byte (bf16) boundaries only -- call mmt_lss_plan_prepare and "
                         "mmt_depth_softmax_forward instead", who);
    const int pieces = (int)mmt::ceil_div(D, kGroup * 4);
    const int variant = (lb ? 6 : 0) + (ub ? 3 : 0) + (pieces <= 2 ? 0 : (pieces <= 4 ? 1 : 2));
    SoftmaxArgs a{pixels, D, logits, logit_row_stride, probs, oracle, oracle_row_stride, depth_used, nullptr, nullptr, nullptr};
    p.token = next_token();
    static const char *dbg = getenv("MMT_RIDER_NOLOOKUP");          // experiments only: the softmax in this kernel's shape, no lookup
    const int nlookup = (dbg && dbg[0] == '1') ? 0 : 1 + (B < kBuildPar ? B : kBuildPar);
    // the softmax's share of the grid: what the stand-alone kernel launches (csrc/depth_softmax.hip)
    const int nsm = mmt::stream_grid(pixels * kGroup, kRiderThreads, 256 * 32);
    mmt::TimedSeq seq;
    seq.launch(true, lss_plan_lookup_softmax, dim3((unsigned)(nlookup + nsm)), dim3(kRiderThreads), 0, (hipStream_t)stream, p, a, variant, nlookup);
    return mmt::check_launch(who);
}

extern "C" int mmt_lss_plan_cache_layout(int N, int D, int fH, int fW, int nx, int ny, int64_t plan_cache_bytes, int64_t *layout_host) {
    MMT_REQUIRE_PTR(layout_host);
    if (N <= 0 || N > kPlanMaxN || D <= 0 || fH <= 0 || fW <= 0 || nx <= 0 || ny <= 0) return mmt::fail(MMT_ERR_BAD_SHAPE, "lss_plan_cache_layout: bad shape");
    Layout l;
    make_layout(N, D, fH, fW, nx, ny, 1, &l);
    const int64_t fixed = kHdrBytes + (int64_t)kBuildPar * l.scratch_bytes;
    int64_t fit = plan_cache_bytes < fixed ? 0 : (plan_cache_bytes - fixed) / l.slot_bytes;
    if (fit > 256) fit = 256;
    const int64_t v[12] = {fit, l.slots_off, l.slot_bytes, l.summary_off, l.records_off, kVerdictOff, l.d.jobs_cap, l.d.runs_cap, kJobBytes, kPlanMaxB, l.d.strips, l.d.ntiles};
    static_assert(sizeof(Verdict) == 64, "tests read 16 words per verdict");
    memcpy(layout_host, v, sizeof(v));
    return MMT_OK;
}

extern "C" int mmt_lss_plan_cache_counters(const void *plan_cache, int64_t plan_cache_bytes, int64_t *counters_host, void *stream) {
    MMT_REQUIRE_PTR(plan_cache);
    MMT_REQUIRE_PTR(counters_host);
    if (plan_cache_bytes < kHdrBytes) return mmt::fail(MMT_ERR_BAD_SHAPE, "lss_plan_cache_counters: not a plan cache (%lld bytes)", (long long)plan_cache_bytes);
    CacheHeader h;
    if (const hipError_t e = hipMemcpyAsync(&h, plan_cache, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream); e != hipSuccess)
        return mmt::fail((int)e, "lss_plan_cache_counters: %s", hipGetErrorString(e));
    if (const hipError_t e = hipStreamSynchronize((hipStream_t)stream); e != hipSuccess)
        return mmt::fail((int)e, "lss_plan_cache_counters: %s", hipGetErrorString(e));
    const bool live = h.magic == kPlanMagic;
    const int64_t v[8] = {live ? h.hits : 0, live ? h.built : 0, live ? h.brute : 0, live ? h.resets : 0, live ? h.calls : 0, live ? h.nslots : 0,
                          live ? h.stale : 0, 0};
    memcpy(counters_host, v, sizeof(v));
    return MMT_OK;
}

extern "C" int mmt_lss_splat_forward_plan(int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz, const float *combine,
                                          const float *frustum_u, const float *frustum_v, const float *frustum_d, const float *voxel_coord_host,
                                          const float *voxel_size_host, const float *depth, const float *context, float *out,
                                          int32_t *column_summary, void *plan_cache, int64_t plan_cache_bytes, int flags, void *stream) {
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(out);
    return plan_forward_impl<float>("lss_splat_forward_plan", B, N, D, fH, fW, C, nx, ny, nz, combine, frustum_u, frustum_v, frustum_d,
                                    voxel_coord_host, voxel_size_host, depth, context, out, column_summary, plan_cache, plan_cache_bytes, flags,
                                    (hipStream_t)stream);
}

extern "C" int mmt_lss_splat_forward_plan_bf16(int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz, const float *combine,
                                               const float *frustum_u, const float *frustum_v, const float *frustum_d,
                                               const float *voxel_coord_host, const float *voxel_size_host, const uint16_t *depth,
                                               const uint16_t *context, float *out, int32_t *column_summary, void *plan_cache,
                                               int64_t plan_cache_bytes, int flags, void *stream) {
    MMT_REQUIRE_PTR(depth);
    MMT_REQUIRE_PTR(context);
    MMT_REQUIRE_PTR(out);
    return plan_forward_impl<bf16_t>("lss_splat_forward_plan_bf16", B, N, D, fH, fW, C, nx, ny, nz, combine, frustum_u, frustum_v, frustum_d,
                                     voxel_coord_host, voxel_size_host, depth, context, out, column_summary, plan_cache, plan_cache_bytes, flags,
                                     (hipStream_t)stream);
}
