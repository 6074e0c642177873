// runtime.cpp -- error plumbing of the C ABI.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void ew_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

ew_status ew_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ew_set_error("%s: %s", what, hipGetErrorString(e));
        return EW_ERR_HIP;
    }
    return EW_OK;
}

extern "C" const char* ew_last_error(void) { return g_err; }
extern "C" int ew_abi_version(void) { return EW_ABI_VERSION; }
