// Camera geometry of the LSS frustum (lss_fpn.py:328-361 + the quantise :461-462), shared by mmt_frustum_geometry
// (bev_geometry.hip: the kernel that WRITES the int32 voxel indices) and the "camera form" of the fused lift-splat
// kernels (lift_splat_tile.hip / lift_splat_col.hip: the cell of a point is computed in the kernel, no geom tensor).
// One set of device functions so that the two agree bit for bit by construction.
#pragma once
#include <string.h>

#include "mmt_common.h"

namespace mmt {

// Quantisation grid, fp32 values exactly as the reference's buffers yield them (lss_fpn.py:278-285, :461-462):
// lo = fp32(voxel_coord - fp32(voxel_size / 2)), vs = voxel_size; inv = fp32(1 / vs) serves the fast path of cam_quant.
struct CamGrid {
    float lo[3];
    float vs[3];
    float inv[3];
    // Exact range test without a quantisation (set by make_cam_range): the index of coordinate value r lies in [0, n) iff
    // !(r < tmin) && !(r >= tmax), where tmin / tmax are the SMALLEST fp32 values whose index is >= 0 / >= n.  The quantise
    // (fp32 subtract, correctly rounded divide, saturating truncation) is monotone in r for a positive finite voxel size, so
    // the two thresholds describe the kept set exactly; a NaN coordinate has index 0 (the device's conversion) and passes
    // both comparisons, as it must.  range_ok = 0 (degenerate grid): the kernels test the quantised index instead.
    float tmin[3], tmax[3];
    int range_ok;
};

// the device's quantise on the host: IEEE fp32 subtract and divide, truncation, saturating, NaN -> 0
static inline long long quantize_host(float v, float lo, float vs) {
    volatile float a = v - lo;
    volatile float q = a / vs;
    if (q != q) return 0;
    if (q >= 2147483648.0f) return 2147483647ll;
    if (q <= -2147483648.0f) return -2147483648ll;
    return (long long)(int)q;
}

// smallest fp32 value (in the order -inf < ... < -0 < +0 < ... < +inf) whose index is >= n; +inf if none is
static inline float first_value_with_index_at_least(float lo, float vs, long long n) {
    auto value = [](uint32_t key) { uint32_t b = (key & 0x80000000u) ? (key ^ 0x80000000u) : ~key; float f; memcpy(&f, &b, 4); return f; };
    uint32_t lo_k = 0x007FFFFFu;        // key of -inf (bits 0xFF800000 -> ~bits)
    uint32_t hi_k = 0xFF800000u;        // key of +inf (bits 0x7F800000 ^ 0x80000000)
    if (quantize_host(value(hi_k), lo, vs) < n) return value(hi_k);
    while (lo_k < hi_k) {               // invariant: index(value(hi_k)) >= n
        const uint32_t mid = lo_k + (hi_k - lo_k) / 2;
        if (quantize_host(value(mid), lo, vs) >= n) hi_k = mid; else lo_k = mid + 1;
    }
    return value(hi_k);
}

static inline void make_cam_grid(const float *vc, const float *vs, CamGrid *q) {
    for (int a = 0; a < 3; ++a) {
        volatile float half = vs[a] / 2.0f;   // fp32, as torch computes voxel_size / 2.0
        volatile float lo = vc[a] - half;
        volatile float inv = 1.0f / vs[a];
        q->lo[a] = lo;
        q->vs[a] = vs[a];
        q->inv[a] = inv;
        q->tmin[a] = q->tmax[a] = 0.f;
    }
    q->range_ok = 0;
}

// thresholds of the exact range test for a grid of (nx, ny, nz) voxels
static inline void make_cam_range(CamGrid *q, int nx, int ny, int nz) {
    const int n[3] = {nx, ny, nz};
    q->range_ok = 1;
    for (int a = 0; a < 3; ++a) {
        const float lo = q->lo[a], vs = q->vs[a];
        if (!(vs > 0.f) || !(vs < 3.0e38f) || !(lo > -3.0e38f && lo < 3.0e38f) || n[a] <= 0) { q->range_ok = 0; return; }
        q->tmin[a] = first_value_with_index_at_least(lo, vs, 0);
        q->tmax[a] = first_value_with_index_at_least(lo, vs, n[a]);
        if (!(q->tmin[a] <= q->tmax[a])) { q->range_ok = 0; return; }
    }
}

// The camera form's operands (device pointers; host-side bundle of the entry points' arguments).
struct CamGeom {
    const float *combine;    // [B*N, 16]: sensor2ego @ inverse(intrin), row-major 4x4 (lss_fpn.py:339-352)
    const float *fu;         // [fW]  frustum[..., 0]: image x of a column      (lss_fpn.py:318-320)
    const float *fv;         // [fH]  frustum[..., 1]: image y of a row         (:321-323)
    const float *fd;         // [D]   frustum[..., 2]: depth of a bin           (:314-316); frustum[..., 3] == 1
    CamGrid q;
    // COLUMN SUMMARY (nullable): int32 [B*N, ceil(fH/16), fW, D, 2] -- what mmt_cam_column_cells finds out about a block of
    // 16 image rows of one column at one depth bin, 8 bytes per block (0.5 byte per point):
    //   [0] = (y0 << 16) | x0 of the block's first row, or -1 when that cell lies outside the grid
    //   [1] = bit i: the z index of row i is in range; bit 16: every row of the block shares (x0, y0)
    // The forward WRITES it while it computes the geometry (summary_cached = 0); the backward kernels -- and a forward whose
    // calibration has not changed (summary_cached = 1) -- READ it instead of computing: their kept tests and cells cost two
    // dwords per block then.  Blocks without bit 16 are evaluated row by row from the matrices wherever they are met.
    int32_t *summary;
    int summary_cached;
    int32_t *excl;           // nullable: exclusive-cell cache of the forward (lift_splat_tile.hip), persistent, zero-initialised once
    int64_t excl_bytes;
};
constexpr int kSummaryUniform = 0x10000;

}  // namespace mmt

// fp32 subtract, correctly rounded IEEE divide, truncate toward zero (v_cvt_i32_f32 saturates and maps NaN to 0, as the
// reference's device does): the reference expression, operation by operation.
__device__ __forceinline__ int mmt_quantize_exact(float v, float lo, float vs) {
    return (int)__fdiv_rn(__fsub_rn(v, lo), vs);
}

// The same integer without the division on (nearly) every call.  a = v - lo as above; qf = RN(a * RN(1 / vs)) differs from
// RN(a / vs) by less than |qf| * 2^-22 (three roundings of 2^-24 relative each), so whenever qf is further than
// |qf| * 2^-21 from the nearest integer the two lie strictly between the same pair of integers and truncate alike.
// Otherwise (about one value in 10^4; also NaN, infinities, |qf| >= 2^23 where every float is an integer: the
// comparison is false) the exact form decides.  The branch is wave-uniform, so the division costs nothing unless a lane
// of the wave needs it.
__device__ __forceinline__ int mmt_quantize_fast(float v, float lo, float vs, float inv) {
    const float a = __fsub_rn(v, lo);
    const float qf = __fmul_rn(a, inv);
    const bool sure = __builtin_fabsf(__fsub_rn(qf, __builtin_rintf(qf))) > __fmul_rn(__builtin_fabsf(qf), 0x1p-21f);
    int n = (int)qf;
    if (!__all(sure)) {
        const int e = (int)__fdiv_rn(a, vs);
        n = sure ? n : e;
    }
    return n;
}

// xyz = (combine @ (u*d, v*d, d, w))[:3]: explicit *_rn operations forbid FMA contraction, so the result equals the
// k-ordered fp32 dot product of the oracle (and of torch's matmul on 4-vectors) bit for bit.  m = the first 12 entries
// of the camera's row-major 4x4.
__device__ __forceinline__ void mmt_cam_xyz(const float (&m)[12], float u, float v, float d, float w, float (&r)[3]) {
    const float p0 = __fmul_rn(u, d), p1 = __fmul_rn(v, d);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float acc = __fmul_rn(m[k * 4], p0);
        acc = __fadd_rn(acc, __fmul_rn(m[k * 4 + 1], p1));
        acc = __fadd_rn(acc, __fmul_rn(m[k * 4 + 2], d));
        acc = __fadd_rn(acc, __fmul_rn(m[k * 4 + 3], w));
        r[k] = acc;
    }
}

// Voxel index of frustum point (u, v, d) of the camera with matrix m: what mmt_frustum_geometry writes for it.
__device__ __forceinline__ void mmt_cam_cell(const float (&m)[12], float u, float v, float d, const mmt::CamGrid &q, int &gx, int &gy,
                                             int &gz) {
    float r[3];
    mmt_cam_xyz(m, u, v, d, 1.0f, r);
    gx = mmt_quantize_fast(r[0], q.lo[0], q.vs[0], q.inv[0]);
    gy = mmt_quantize_fast(r[1], q.lo[1], q.vs[1], q.inv[1]);
    gz = mmt_quantize_fast(r[2], q.lo[2], q.vs[2], q.inv[2]);
}

// ---- a block of up to NR image rows of ONE column at ONE depth bin ------------------------------------------------------
// For fixed (camera, u, d) every coordinate of the point is a monotone function of the row's v: fl(v*d), its product with a
// matrix entry, the sums with row-independent terms and the quantise are all monotone (rounding never reverses an order),
// in one direction or the other.  Hence, for rows sorted by v,
//   * if the first and the last row of the block fall into the same (x, y) cell, every row in between does;
//   * the rows whose z index is in range form one interval.
// A level camera (what get_geometry's rigs are, to within their calibration) takes the first case for every block: two
// (x, y) quantisations per block instead of 2 * NR.  z is evaluated per row, but only its range test -- two comparisons
// against the exact thresholds of CamGrid -- not its index.  Blocks whose ends differ (or an unsorted v, or a degenerate
// grid) evaluate every row; the choice is wave-uniform.  Either way the result equals mmt_cam_cell per row, bit for bit.
//   v[i]     : frustum_v of the block's rows; rows i >= nr must repeat v[nr - 1]
struct mmt_cam_column {            // row-independent terms of one (camera, column, depth bin)
    float A[3], Bd[3], m1[3], m3[3], d;
    __device__ __forceinline__ float coord(int k, float v) const {      // row k of mmt_cam_xyz with w = 1 (m3 * 1 == m3 exactly)
        return __fadd_rn(__fadd_rn(__fadd_rn(A[k], __fmul_rn(m1[k], __fmul_rn(v, d))), Bd[k]), m3[k]);
    }
};
__device__ __forceinline__ mmt_cam_column mmt_cam_column_make(const float (&m)[12], float u, float d) {
    mmt_cam_column c;
    const float p0 = __fmul_rn(u, d);
#pragma unroll
    for (int k = 0; k < 3; ++k) { c.A[k] = __fmul_rn(m[k * 4], p0); c.Bd[k] = __fmul_rn(m[k * 4 + 2], d); c.m1[k] = m[k * 4 + 1]; c.m3[k] = m[k * 4 + 3]; }
    c.d = d;
    return c;
}
// (x, y) voxel index of the row with frustum_v = v; in range?
__device__ __forceinline__ bool mmt_cam_row_xy(const mmt_cam_column &c, float v, const mmt::CamGrid &q, int nx, int ny, int &gx, int &gy) {
    gx = mmt_quantize_fast(c.coord(0, v), q.lo[0], q.vs[0], q.inv[0]);
    gy = mmt_quantize_fast(c.coord(1, v), q.lo[1], q.vs[1], q.inv[1]);
    return (unsigned)gx < (unsigned)nx && (unsigned)gy < (unsigned)ny;
}
// Rows of the block whose z index is in range (bit i), and whether the whole block shares the (x, y) cell (x0, y0) of its
// first row.  `uniform` false: the caller evaluates mmt_cam_row_xy per row (the decision is wave-uniform).
template <int NR>
__device__ __forceinline__ unsigned mmt_cam_column_cells(const mmt_cam_column &c, const float (&v)[NR], int nr, bool v_sorted,
                                                         const mmt::CamGrid &q, int nx, int ny, int nz, bool &uniform, bool &in0, int &x0, int &y0) {
    unsigned zmask = 0;
    if (q.range_ok) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const float z = c.coord(2, v[i]);
            zmask |= (!(z < q.tmin[2]) && !(z >= q.tmax[2])) ? (1u << i) : 0u;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int gz = mmt_quantize_fast(c.coord(2, v[i]), q.lo[2], q.vs[2], q.inv[2]);
            zmask |= ((unsigned)gz < (unsigned)nz) ? (1u << i) : 0u;
        }
    }
    zmask &= nr >= 32 ? ~0u : ((1u << nr) - 1u);
    int x1, y1;
    in0 = mmt_cam_row_xy(c, v[0], q, nx, ny, x0, y0);
    mmt_cam_row_xy(c, v[NR - 1], q, nx, ny, x1, y1);
    uniform = __all((v_sorted && x0 == x1 && y0 == y1) || zmask == 0u);
    return zmask;
}

// true when v[0..nr) is sorted (either direction): the precondition of the end-row shortcut above
template <int NR>
__device__ __forceinline__ bool mmt_rows_sorted(const float (&v)[NR], int nr) {
    bool up = true, down = true;
#pragma unroll
    for (int i = 0; i + 1 < NR; ++i) {
        if (i + 1 < nr) { up = up && v[i] <= v[i + 1]; down = down && v[i] >= v[i + 1]; }
    }
    return up || down;
}
