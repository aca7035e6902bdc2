// Host build of the plan builder's integer core (mm_training_amd/csrc/lss_plan_core.h) for tests/test_lss_plan_core.py:
// the phases the device runs as one workgroup per sample, here single-threaded (tid 0 of 1), on a summary and row cells
// that the TEST derives from the oracle's geometry.  Test infrastructure only -- the product library never runs this.
#include <stdlib.h>
#include <string.h>

#include "../../mm_training_amd/csrc/lss_plan_core.h"

using namespace mmt::plan;

struct HostRowCells {
    const int32_t *cells;   // [strips * D * 16] packed (y << 16 | x) or -1
    int D;
    int operator()(int s, int bin, int row) const { return cells[((long long)s * D + bin) * 16 + row]; }
};

extern "C" int plan_host_dims(int N, int D, int fH, int fW, int nx, int ny, int runs_cap, int32_t *out /* 14 ints: Dims */) {
    Dims d;
    make_dims(N, D, fH, fW, nx, ny, &d, runs_cap);
    memcpy(out, &d, sizeof(d));
    return dims_ok(d) ? 1 : 0;
}

extern "C" int plan_host_job_bytes(void) { return kJobBytes; }

// Returns njobs (>= 0), or -1 unplannable; records: [jobs_cap * kJobBytes]; status: 16 ints (nruns, njobs, unplannable, group starts [9], ...)
extern "C" int plan_host_build(int N, int D, int fH, int fW, int nx, int ny, int runs_cap_override, const int32_t *summary, const int32_t *rowcells,
                               uint8_t *records, int32_t *status) {
    Dims d;
    make_dims(N, D, fH, fW, nx, ny, &d, runs_cap_override);
    void *mem = calloc(1, (size_t)scratch_bytes(d, 1));
    Scratch s;
    scratch_carve(d, 1, mem, &s);
    HostRowCells rc{rowcells, D};
    phase_clear(d, s, 0, 1);
    phase_count(d, s, summary, rc, 0, 1);
    scan_a(s.cell_off, d.ncells_tm, s.partial, 0, 1); scan_b(s.cell_off, d.ncells_tm, s.partial, 0, 1); scan_c(s.cell_off, d.ncells_tm, s.partial, 0, 1);
    phase_check_runs(d, s, 0);
    int njobs = -1;
    if (!s.status[2]) {
        phase_place(d, s, summary, rc, 0, 1);
        phase_sort_cells(d, s, 0, 1);
        phase_count_jobs(d, s, 0, 1);
        phase_tile_order(d, s, 0, 1);
        phase_perm_gather(d, s, 0, 1);
        scan_a(s.perm_jobs, d.ntiles, s.partial, 0, 1); scan_b(s.perm_jobs, d.ntiles, s.partial, 0, 1); scan_c(s.perm_jobs, d.ntiles, s.partial, 0, 1);
        phase_tile_bases(d, s, 0, 1);
        if (!s.status[2]) {
            phase_write_jobs(d, s, 0, 1);
            njobs = s.status[1];
            phase_records(d, s, records, njobs, 0, 1);
        }
    }
    memcpy(status, s.status, 64);
    free(mem);
    return njobs;
}
