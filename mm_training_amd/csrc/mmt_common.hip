// Error plumbing of libmmt_hip.so.
#include "mmt_common.h"

namespace mmt {

char *error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) return 0;
    return fail((int)err, "%s: HIP launch failed: %s", what, hipGetErrorString(err));
}

static thread_local hipEvent_t g_timing_start = nullptr, g_timing_stop = nullptr;

void take_timing_events(hipEvent_t *start, hipEvent_t *stop) {
    *start = g_timing_start;
    *stop = g_timing_stop;
    g_timing_start = g_timing_stop = nullptr;
}

}  // namespace mmt

extern "C" int mmt_timing_event_create(void **event) {
    if (!event) return mmt::fail(MMT_ERR_NULL_POINTER, "timing_event_create: event is NULL");
    hipEvent_t e;
    hipError_t err = hipEventCreate(&e);
    if (err != hipSuccess) return mmt::fail((int)err, "timing_event_create: %s", hipGetErrorString(err));
    *event = (void *)e;
    return 0;
}

extern "C" int mmt_timing_event_destroy(void *event) {
    if (!event) return 0;
    hipError_t err = hipEventDestroy((hipEvent_t)event);
    return err == hipSuccess ? 0 : mmt::fail((int)err, "timing_event_destroy: %s", hipGetErrorString(err));
}

extern "C" int mmt_timing_elapsed_ms(void *start, void *stop, float *ms) {
    if (!start || !stop || !ms) return mmt::fail(MMT_ERR_NULL_POINTER, "timing_elapsed_ms: NULL argument");
    hipError_t err = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    return err == hipSuccess ? 0 : mmt::fail((int)err, "timing_elapsed_ms: %s", hipGetErrorString(err));
}

extern "C" int mmt_arm_kernel_timing(void *start, void *stop) {
    if ((start == nullptr) != (stop == nullptr)) return mmt::fail(MMT_ERR_NULL_POINTER, "arm_kernel_timing: need both events or none");
    mmt::g_timing_start = (hipEvent_t)start;
    mmt::g_timing_stop = (hipEvent_t)stop;
    return 0;
}

extern "C" int mmt_abi_version(void) { return MMT_ABI_VERSION; }
extern "C" const char *mmt_last_error(void) { return mmt::error_buffer(); }
