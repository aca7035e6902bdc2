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

}  // namespace mmt

extern "C" int mmt_abi_version(void) { return MMT_ABI_VERSION; }
extern "C" const char *mmt_last_error(void) { return mmt::error_buffer(); }
