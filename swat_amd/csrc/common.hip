// swat_amd/csrc/common.hip — error plumbing + version for libswat_hip.so.
#include "common.hpp"
#include "../../include/swat_hip.h"
#include <stdarg.h>

namespace swat {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    set_error("%s: %s", what, hipGetErrorString(e));
    return -(int)e;
}
}  // namespace swat

SWAT_API int swat_version(void) { return 1; }
SWAT_API const char* swat_last_error(void) { return swat::g_err; }
