// Shared host-side helpers for libmmt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mmt_hip.h"

namespace mmt {

// thread-local message behind mmt_last_error()
char *error_buffer();
int fail(int code, const char *fmt, ...);

// post-launch check: returns 0 or the hipError_t (recorded in the error buffer)
int check_launch(const char *what);

// Kernel timing for the bench (mmt_arm_kernel_timing): the events armed by the caller, handed to the NEXT
// timed launch sequence of this thread and cleared.  Both are null when nothing is armed.
void take_timing_events(hipEvent_t *start, hipEvent_t *stop);

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// The kernels of ONE C-ABI call as a timed sequence: takes whatever mmt_arm_kernel_timing armed (nothing, normally);
// the start event rides on the first dispatch made through launch(), the stop event on the one marked `last`
// (hipExtLaunchKernel start / stop events = the kernels' own duration on the device, no dispatch latency).
struct TimedSeq {
    hipEvent_t start = nullptr, stop = nullptr;
    TimedSeq() { take_timing_events(&start, &stop); }
    template <typename F, typename... Args>
    void launch(bool last, F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
        hipEvent_t s = start, e = last ? stop : nullptr;
        start = nullptr;
        if (s || e) hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, s, e, 0, args...);
        else hipLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, args...);
    }
};

// Grid for streaming kernels: enough workgroups to fill 256 CUs several times over,
// capped so very large problems grid-stride instead of queueing >100k tiny blocks.
static inline int stream_grid(int64_t work_items, int block, int max_blocks = 256 * 16) {
    int64_t g = ceil_div(work_items, block);
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

}  // namespace mmt

// 16-byte non-temporal store (streaming output that must not evict reused lines)
typedef float mmt_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int mmt_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mmt_nt_store4(float4 v, float4 *p) {
    mmt_f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<mmt_f32x4 *>(p));
}

// bf16 storage (SURVEY 5.6 / BASELINE configs[4]: bf16 STORAGE, fp32 accumulate): raw bits + conversions
typedef unsigned short bf16_t;
// fp32 pair -> two bf16 in one dword (low half = first), round to nearest even, NaN stays NaN (the plain cast:
// v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __bf16 mmt_bf16x2 __attribute__((ext_vector_type(2)));
    typedef float mmt_f32x2 __attribute__((ext_vector_type(2)));
    const mmt_f32x2 f = {lo, hi};
    const mmt_bf16x2 h = __builtin_convertvector(f, mmt_bf16x2);
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float bf16_lo(unsigned pair) { return __uint_as_float(pair << 16); }
__device__ __forceinline__ float bf16_hi(unsigned pair) { return __uint_as_float(pair & 0xFFFF0000u); }

// fused lift-splat backward on the matrix cores (lift_splat_col.hip), called from mmt_lss_splat_backward[_cam]
namespace mmt {
struct CamGeom;   // mmt_camera.h; NULL = geom form
void lss_note_forward_family(int family);   // what mmt_lss_last_kernel_family(0) reports (lift_splat_tile.hip; set by lift_splat_plan.hip too)
bool lss_col_backward_fits(int D, int fH, int fW, int C, int64_t span, int64_t grid_units);
int lss_col_backward_f32(const char *what, int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz, const int32_t *geom,
                         const CamGeom *cam, const float *depth, const float *context, const float *grad_out, int64_t sb, int64_t sy,
                         int64_t sx, int64_t span, float *grad_depth, float *grad_context, unsigned long long *stats, int pm, hipStream_t st);
int lss_col_backward_bf16(const char *what, int B, int N, int D, int fH, int fW, int C, int nx, int ny, int nz, const int32_t *geom,
                          const CamGeom *cam, const bf16_t *depth, const bf16_t *context, const float *grad_out, int64_t sb, int64_t sy,
                          int64_t sx, int64_t span, bf16_t *grad_depth, float *grad_context, unsigned long long *stats, int pm, hipStream_t st);
}  // namespace mmt

#define MMT_REQUIRE_PTR(p)                                                        \
    do {                                                                          \
        if ((p) == nullptr) return mmt::fail(MMT_ERR_NULL_POINTER, "%s: %s is NULL", __func__, #p); \
    } while (0)
