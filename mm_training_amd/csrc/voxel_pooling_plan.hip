// Cached-plan voxel_pooling forward for MI355X (gfx950, wave64)  --  SURVEY section 8, row f3.
//
// The point -> BEV-cell assignment of voxel_pooling
// (ops/voxel_pooling/src/voxel_pooling_forward_cuda.cu:19-29) depends only on geom_xyz, i.e.
// on the camera calibration (layers/backbones/lss_fpn.py:328-361,461-462; BDA is disabled in
// the reference's get_geometry, :355-360).  While the calibration is unchanged the sort of the
// points by cell can be done ONCE ("plan"), after which the forward is a pure segmented gather:
//
//   plan build (amortised):  cell key per point -> stable radix sort by cell (rocPRIM; keeps
//       ascending point order inside a cell) -> one pass over the cell histogram that cuts
//       every cell's list into items of <= kSeg rows.  pos_memo is produced here as well.
//   forward (per step):      one lane group (C/4 lanes, one float4 column each) per item:
//       rows summed in registers in plan order, result stored with plain 16-byte stores.
//       Cells with one item are written straight to the BEV (empty cells: zeros, so the caller
//       does not pre-zero), cells with several items go through a small partial-row buffer
//       that a second tiny kernel folds in item order.
//
// No atomics, no hash table, no geom read, no pos_memo write per step; the result is
// bit-reproducible run to run (fixed summation order).
//
// HBM traffic per forward (algorithmic): 4*C*K feature rows of kept points + 4*K row ids
// + 16 B per item + 4*C*B*ny*nx BEV rows written once.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "mmt_common.h"

namespace {

constexpr int kBlock = 256;
#ifndef MMT_PLAN_SEG
#define MMT_PLAN_SEG 32
#endif
constexpr int kSeg = MMT_PLAN_SEG;   // rows per item (multiple of kInFlight)
constexpr int kInFlight = 8;      // feature rows in flight per lane group
constexpr int kScanBlock = 1024;
constexpr int kPlanMagic = 0x4d4d5450;  // "MMTP"

// plan header (int32 words at the start of the plan buffer)
enum { H_MAGIC = 0, H_B, H_P, H_NX, H_NY, H_SEG, H_NITEMS, H_KEPT, H_NMULTI, H_NPARTIAL, H_WORDS = 16 };

struct PlanLayout {
    int64_t order, items, multi, total;   // offsets in int32 words
    int64_t max_items, max_multi;
};

PlanLayout plan_layout(int64_t BP, int64_t NC) {
    PlanLayout L;
    L.max_items = NC + BP / kSeg + 1;       // sum over cells of max(1, ceil(n/kSeg)) <= NC + BP/kSeg
    L.max_multi = BP / (kSeg + 1) + 1;      // cells holding more than kSeg points
    L.order = H_WORDS;
    L.items = (L.order + BP + 3) & ~3ll;    // int4 aligned
    L.multi = L.items + 4 * L.max_items;
    L.total = L.multi + 4 * L.max_multi;
    return L;
}

int key_bits(int64_t NC) {
    int b = 1;
    while ((1ll << b) <= NC) ++b;           // NC itself is the "dropped" key
    return b;
}

// ---------------------------------------------------------------- plan build
__global__ __launch_bounds__(kBlock) void plan_keys_kernel(int64_t BP, int P, int nx, int ny, int nz, unsigned NC,
                                                           const int32_t *__restrict__ geom,
                                                           unsigned *__restrict__ keys, unsigned *__restrict__ vals,
                                                           int *__restrict__ count, int32_t *__restrict__ pos_memo,
                                                           int32_t *__restrict__ hdr) {
    if (blockIdx.x == 0 && threadIdx.x < H_WORDS) {
        const int B = (int)(BP / P);
        const int v[H_WORDS] = {kPlanMagic, B, P, nx, ny, kSeg, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        hdr[threadIdx.x] = v[threadIdx.x];
    }
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < BP; t += (int64_t)gridDim.x * kBlock) {
        const int x = geom[t * 3], y = geom[t * 3 + 1], z = geom[t * 3 + 2];
        // voxel_pooling_forward_cuda.cu:24-26 (negated)
        const bool keep = !(x < 0 || x >= nx || y < 0 || y >= ny || z < 0 || z >= nz);
        const int b = (int)((unsigned)t / (unsigned)P);
        unsigned key = NC;
        if (keep) {
            key = (unsigned)((b * ny + y) * nx + x);
            atomicAdd(&count[key], 1);
        }
        keys[t] = key;
        vals[t] = (unsigned)t;
        if (pos_memo) {  // voxel_pooling_forward_cuda.cu:27-29; dropped rows get the -1 pre-fill of voxel_pooling.py:40
            pos_memo[t * 3] = keep ? b : -1;
            pos_memo[t * 3 + 1] = keep ? y : -1;
            pos_memo[t * 3 + 2] = keep ? x : -1;
        }
    }
}

// One workgroup walks the cell histogram: exclusive scans of (points, items, multi cells,
// partial rows) and the item / multi-cell descriptors, in cell order.
__global__ __launch_bounds__(kScanBlock) void plan_items_kernel(int NC, const int *__restrict__ count,
                                                                int4 *__restrict__ items, int4 *__restrict__ multi,
                                                                int32_t *__restrict__ hdr) {
    __shared__ int4 wsum[kScanBlock / 64];
    __shared__ int4 carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (NC + kScanBlock - 1) / kScanBlock;
    const int c0 = tid * per, c1 = min(NC, c0 + per);
    int4 loc = make_int4(0, 0, 0, 0);   // points, items, multi cells, partial rows
    for (int c = c0; c < c1; ++c) {
        const int n = count[c];
        const int ns = n > kSeg ? (n + kSeg - 1) / kSeg : 1;
        loc.x += n; loc.y += ns;
        if (ns > 1) { loc.z += 1; loc.w += ns; }
    }
    int4 inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int vx = __shfl_up(inc.x, o), vy = __shfl_up(inc.y, o), vz = __shfl_up(inc.z, o), vw = __shfl_up(inc.w, o);
        if (lane >= o) { inc.x += vx; inc.y += vy; inc.z += vz; inc.w += vw; }
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    if (tid == 0) {
        int4 run = make_int4(0, 0, 0, 0);
        for (int w = 0; w < kScanBlock / 64; ++w) {
            const int4 v = wsum[w];
            wsum[w] = run;
            run.x += v.x; run.y += v.y; run.z += v.z; run.w += v.w;
        }
        carry = run;
    }
    __syncthreads();
    int4 base = wsum[wave];
    base.x += inc.x - loc.x; base.y += inc.y - loc.y; base.z += inc.z - loc.z; base.w += inc.w - loc.w;
    for (int c = c0; c < c1; ++c) {
        const int n = count[c];
        const int ns = n > kSeg ? (n + kSeg - 1) / kSeg : 1;
        if (ns == 1) {
            items[base.y] = make_int4(c, base.x, n, 0);                       // -> BEV row c
        } else {
            multi[base.z] = make_int4(c, base.w, ns, 0);
            for (int j = 0; j < ns; ++j)
                items[base.y + j] = make_int4(base.w + j, base.x + j * kSeg, min(kSeg, n - j * kSeg), 1);  // -> partial row
            base.z += 1; base.w += ns;
        }
        base.x += n; base.y += ns;
    }
    if (tid == 0) {
        hdr[H_KEPT] = carry.x; hdr[H_NITEMS] = carry.y; hdr[H_NMULTI] = carry.z; hdr[H_NPARTIAL] = carry.w;
    }
}

// ---------------------------------------------------------------- planned forward
struct PlannedArgs {
    int C, nitems, nmulti;
    int order_len;                 // B*P entries in `order`
    int64_t feat_bytes;
    int64_t out_stride;            // floats between consecutive BEV cells' rows (>= C)
    const int32_t *order;
    const int4 *items;
    const int4 *multi;
    const float *feats;
    float *out;
    float *partial;                // [npartial, C]
};

// Every load in the loop is UNCONDITIONAL (a predicated load makes hipcc wait with vmcnt(0) before
// the next one, i.e. one row in flight per group): positions past the item's end are clamped to its
// last row for the row-id load, and their feature load goes through a buffer descriptor with an
// out-of-range offset, which returns zeros without touching memory (BUF; feature matrix < 4 GiB),
// or re-reads the last row and is masked afterwards.
template <int C4T, bool BUF>
__global__ __launch_bounds__(kBlock) void vp_planned_items(PlannedArgs a) {
    constexpr int U = kInFlight;
    const int C = a.C;
    const int C4 = C4T > 0 ? C4T : C >> 2;
    const int G = 64 / C4;
    const int lane = threadIdx.x & 63;
    const int g = lane / C4, li = lane - g * C4;
    if (g >= G) return;
    const int groups_per_block = (kBlock / 64) * G;
    const int wave_in_block = threadIdx.x >> 6;
    const int64_t ngroups = (int64_t)gridDim.x * groups_per_block;
    const float *fcol = a.feats + li * 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.feats), 0, BUF ? (int)(unsigned)a.feat_bytes : 0, 0x00020000);
    const unsigned row_bytes = (unsigned)C * 4u, col_bytes = (unsigned)li * 16u;
    const int last_pos = a.order_len - 1;
    int64_t it = (int64_t)blockIdx.x * groups_per_block + wave_in_block * G + g;
    if (it >= a.nitems) return;
    int4 dn = a.items[it];                     // (destination row, first sorted position, rows, to partial?)
    for (; it < a.nitems; it += ngroups) {
        const int4 d = dn;
        if (it + ngroups < a.nitems) dn = a.items[it + ngroups];   // next descriptor rides under this item's rows
        const int len = d.z;
        const int last = d.y + (len > 0 ? len - 1 : 0);
        int ids[U];
#pragma unroll
        for (int u = 0; u < U; ++u) ids[u] = a.order[min(min(d.y + u, last), last_pos)];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j0 = 0; j0 < len; j0 += U) {
            int nxt[U];
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = a.order[min(min(d.y + j0 + U + u, last), last_pos)];
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool valid = j0 + u < len;
                if (BUF) {
                    const unsigned off = valid ? (unsigned)ids[u] * row_bytes + col_bytes : 0xFFFFFFF0u;
                    const mmt_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
                    v[u] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
                } else {
                    v[u] = *reinterpret_cast<const float4 *>(fcol + (int64_t)ids[u] * C);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!BUF && !(j0 + u < len)) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) ids[u] = nxt[u];
        }
        float *dst = d.w ? a.partial + (int64_t)d.x * C : a.out + (int64_t)d.x * a.out_stride;
        *reinterpret_cast<float4 *>(dst + li * 4) = acc;
    }
}

// cells cut into several items: fold their partial rows in item order (one wave per cell)
template <int C4T>
__global__ __launch_bounds__(kBlock) void vp_planned_fold(PlannedArgs a) {
    __shared__ __align__(16) float stage[kBlock / 64][256];
    const int C = a.C;
    const int C4 = C4T > 0 ? C4T : C >> 2;
    const int G = 64 / C4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / C4, li = lane - g * C4;
    float *st = stage[wave];
    for (int m = blockIdx.x * (kBlock / 64) + wave; m < a.nmulti; m += gridDim.x * (kBlock / 64)) {
        const int4 d = a.multi[m];             // (cell, first partial row, rows)
        if (g < G) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j = g; j < d.z; j += G) {
                const float4 v = *reinterpret_cast<const float4 *>(a.partial + (int64_t)(d.y + j) * C + li * 4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            *reinterpret_cast<float4 *>(st + g * C + li * 4) = acc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float *orow = a.out + (int64_t)d.x * a.out_stride;
        for (int e = lane; e < C; e += 64) {
            float sum = st[e];
            for (int gg = 1; gg < G; ++gg) sum += st[gg * C + e];
            orow[e] = sum;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

int check_shape(const char *who, int B, int P, int nx, int ny, int64_t *BP, int64_t *NC) {
    if (B <= 0 || P <= 0 || nx <= 0 || ny <= 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "%s: non-positive size (B=%d P=%d grid=%dx%d)", who, B, P, nx, ny);
    *BP = (int64_t)B * P;
    *NC = (int64_t)B * ny * nx;
    if (*BP >= (1ll << 31) || *NC >= (1ll << 30))
        return mmt::fail(MMT_ERR_TOO_LARGE, "%s: B*P exceeds int32 or B*ny*nx exceeds 2^30", who);
    return 0;
}

size_t sort_temp_bytes(int64_t BP, int bits) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned *)nullptr, (unsigned *)nullptr,
                                    (const unsigned *)nullptr, (unsigned *)nullptr, (size_t)BP, 0, bits,
                                    (hipStream_t)0);
    return (bytes + 255) & ~(size_t)255;
}

}  // namespace

extern "C" int64_t mmt_voxel_pooling_plan_elems(int B, int P, int nx, int ny) {
    int64_t BP, NC;
    if (check_shape("voxel_pooling_plan_elems", B, P, nx, ny, &BP, &NC)) return -1;
    return plan_layout(BP, NC).total;
}

extern "C" int64_t mmt_voxel_pooling_plan_workspace_bytes(int B, int P, int nx, int ny) {
    int64_t BP, NC;
    if (check_shape("voxel_pooling_plan_workspace_bytes", B, P, nx, ny, &BP, &NC)) return -1;
    // cell histogram | keys in | keys out | point ids in | rocPRIM temporary storage
    const int64_t cnt = ((NC * 4 + 255) / 256) * 256, arr = ((BP * 4 + 255) / 256) * 256;
    return cnt + 3 * arr + (int64_t)sort_temp_bytes(BP, key_bits(NC));
}

extern "C" int mmt_voxel_pooling_plan_build(int B, int P, int nx, int ny, int nz, const int32_t *geom,
                                            int32_t *pos_memo, int32_t *plan, int64_t plan_elems,
                                            void *workspace, int64_t workspace_bytes, void *stream) {
    MMT_REQUIRE_PTR(geom);
    MMT_REQUIRE_PTR(plan);
    MMT_REQUIRE_PTR(workspace);
    int64_t BP, NC;
    if (int rc = check_shape("voxel_pooling_plan_build", B, P, nx, ny, &BP, &NC)) return rc;
    if (nz <= 0) return mmt::fail(MMT_ERR_BAD_SHAPE, "voxel_pooling_plan_build: non-positive nz=%d", nz);
    const PlanLayout L = plan_layout(BP, NC);
    if (plan_elems < L.total)
        return mmt::fail(MMT_ERR_WORKSPACE, "voxel_pooling_plan_build: plan holds %lld int32, needs %lld",
                         (long long)plan_elems, (long long)L.total);
    const int bits = key_bits(NC);
    const int64_t cnt = ((NC * 4 + 255) / 256) * 256, arr = ((BP * 4 + 255) / 256) * 256;
    size_t tmp_bytes = sort_temp_bytes(BP, bits);
    if (workspace_bytes < cnt + 3 * arr + (int64_t)tmp_bytes)
        return mmt::fail(MMT_ERR_WORKSPACE, "voxel_pooling_plan_build: workspace holds %lld bytes, needs %lld",
                         (long long)workspace_bytes, (long long)(cnt + 3 * arr + (int64_t)tmp_bytes));
    if (((uintptr_t)plan & 15) || ((uintptr_t)workspace & 15))
        return mmt::fail(MMT_ERR_BAD_SHAPE, "voxel_pooling_plan_build: plan and workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    int *count = (int *)ws;
    unsigned *keys_in = (unsigned *)(ws + cnt), *keys_out = (unsigned *)(ws + cnt + arr);
    unsigned *vals_in = (unsigned *)(ws + cnt + 2 * arr);
    void *tmp = ws + cnt + 3 * arr;

    hipError_t e = hipMemsetAsync(count, 0, (size_t)NC * 4, st);
    if (e != hipSuccess) return mmt::fail((int)e, "voxel_pooling_plan_build: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(plan_keys_kernel, dim3(mmt::stream_grid(BP, kBlock)), dim3(kBlock), 0, st, BP, P, nx, ny, nz,
                       (unsigned)NC, geom, keys_in, vals_in, count, pos_memo, plan);
    if (int rc = mmt::check_launch("voxel_pooling_plan_build(keys)")) return rc;
    e = rocprim::radix_sort_pairs(tmp, tmp_bytes, (const unsigned *)keys_in, keys_out, (const unsigned *)vals_in,
                                  (unsigned *)(plan + L.order), (size_t)BP, 0, bits, st);
    if (e != hipSuccess) return mmt::fail((int)e, "voxel_pooling_plan_build: sort: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(plan_items_kernel, dim3(1), dim3(kScanBlock), 0, st, (int)NC, count,
                       (int4 *)(plan + L.items), (int4 *)(plan + L.multi), plan);
    return mmt::check_launch("voxel_pooling_plan_build(items)");
}

extern "C" int mmt_voxel_pooling_plan_info(const int32_t *plan, int32_t *info_host, void *stream) {
    MMT_REQUIRE_PTR(plan);
    MMT_REQUIRE_PTR(info_host);
    hipStream_t st = (hipStream_t)stream;
    int32_t hdr[H_WORDS];
    hipError_t e = hipMemcpyAsync(hdr, plan, sizeof(hdr), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return mmt::fail((int)e, "voxel_pooling_plan_info: %s", hipGetErrorString(e));
    if (hdr[H_MAGIC] != kPlanMagic) return mmt::fail(MMT_ERR_BAD_FLAG, "voxel_pooling_plan_info: not a plan buffer");
    info_host[0] = hdr[H_NITEMS]; info_host[1] = hdr[H_KEPT]; info_host[2] = hdr[H_NMULTI]; info_host[3] = hdr[H_NPARTIAL];
    return 0;
}

extern "C" int mmt_voxel_pooling_forward_planned(int B, int P, int C, int nx, int ny, const int32_t *plan,
                                                 int num_items, int num_multi, int num_partial,
                                                 const float *feats, float *out, int64_t out_row_stride,
                                                 float *partial, int64_t partial_elems, void *stream) {
    MMT_REQUIRE_PTR(plan);
    MMT_REQUIRE_PTR(feats);
    MMT_REQUIRE_PTR(out);
    int64_t BP, NC;
    if (int rc = check_shape("voxel_pooling_forward_planned", B, P, nx, ny, &BP, &NC)) return rc;
    if (C <= 0 || C % 4 != 0 || C > 256)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "voxel_pooling_forward_planned: C=%d must be a multiple of 4, <= 256", C);
    if (((uintptr_t)feats & 15) || ((uintptr_t)out & 15) || (out_row_stride & 3) || out_row_stride < C)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "voxel_pooling_forward_planned: feats/out must be 16-byte aligned, "
                                            "out_row_stride a multiple of 4 and >= C");
    const PlanLayout L = plan_layout(BP, NC);
    if (num_items < NC || num_items > L.max_items || num_multi < 0 || num_multi > L.max_multi || num_partial < 0)
        return mmt::fail(MMT_ERR_BAD_SHAPE, "voxel_pooling_forward_planned: counts (%d items, %d multi) do not belong "
                                            "to a plan of this shape", num_items, num_multi);
    if (num_multi > 0) {
        MMT_REQUIRE_PTR(partial);
        if (partial_elems < (int64_t)num_partial * C || ((uintptr_t)partial & 15))
            return mmt::fail(MMT_ERR_WORKSPACE, "voxel_pooling_forward_planned: partial buffer holds %lld floats, needs %lld",
                             (long long)partial_elems, (long long)num_partial * C);
    }
    hipStream_t st = (hipStream_t)stream;
    PlannedArgs a;
    a.C = C; a.nitems = num_items; a.nmulti = num_multi; a.out_stride = out_row_stride;
    a.order_len = (int)BP; a.feat_bytes = BP * C * 4;
    a.order = plan + L.order;
    a.items = (const int4 *)(plan + L.items);
    a.multi = (const int4 *)(plan + L.multi);
    a.feats = feats; a.out = out; a.partial = partial;
    const int G = 64 / (C / 4);
    const int per_block = (kBlock / 64) * G;
    int64_t blocks = mmt::ceil_div(num_items, per_block);
    // lane groups loop over their items (next descriptor prefetched); the grid size itself is not
    // critical: 2048 ... one-item-per-group all measure 80 +- 4 us at cfg2 (run-to-run noise is larger)
    if (blocks > 8192) blocks = 8192;
    const dim3 grid((unsigned)(blocks < 1 ? 1 : blocks)), block(kBlock);
    const bool buf = a.feat_bytes < (1ll << 32) - 16;   // buffer descriptors address 32-bit byte offsets
    if (C == 80 && buf) hipLaunchKernelGGL((vp_planned_items<20, true>), grid, block, 0, st, a);
    else if (C == 64 && buf) hipLaunchKernelGGL((vp_planned_items<16, true>), grid, block, 0, st, a);
    else if (buf) hipLaunchKernelGGL((vp_planned_items<0, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((vp_planned_items<0, false>), grid, block, 0, st, a);
    if (int rc = mmt::check_launch("voxel_pooling_forward_planned(items)")) return rc;
    if (num_multi > 0) {
        const dim3 fgrid((unsigned)mmt::ceil_div(num_multi, kBlock / 64));
        if (C == 80) hipLaunchKernelGGL((vp_planned_fold<20>), fgrid, block, 0, st, a);
        else if (C == 64) hipLaunchKernelGGL((vp_planned_fold<16>), fgrid, block, 0, st, a);
        else hipLaunchKernelGGL((vp_planned_fold<0>), fgrid, block, 0, st, a);
        return mmt::check_launch("voxel_pooling_forward_planned(fold)");
    }
    return 0;
}
