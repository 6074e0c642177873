// runtime.cpp -- error plumbing of the C ABI.
#include "common.h"
#include <atomic>
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

// ---- CU budget of the persistent kernels + CU-masked streams (ABI 8) ----
// Process-wide and atomic: the header's use case -- two CU-masked streams -- is multi-threaded, but the budget is a property of the
// CONFIGURATION (how the chip is partitioned), not of a launch: set it before any stream of the partition launches work.  A change that races a
// launch on another thread could size that launch's grid and stream-K split from two different values (the value is read more than once per
// launch); the atomic only guarantees that every read sees one of the values that were set.
static std::atomic<int> g_cu_budget{256};
int ew_cu_budget() { return g_cu_budget.load(std::memory_order_relaxed); }
extern "C" int ew_get_cu_budget(void) { return ew_cu_budget(); }
extern "C" int ew_set_cu_budget(int n) {
    if (n >= 8 && n <= 256 && n % 8 == 0) return g_cu_budget.exchange(n);
    return ew_cu_budget();
}
extern "C" void* ew_stream_create_cu_mask(int first_cu, int n_cus) {
    if (first_cu < 0 || n_cus <= 0 || first_cu + n_cus > 256) { ew_set_error("ew_stream_create_cu_mask: CU range [%d, %d) outside [0, 256)", first_cu, first_cu + n_cus); return nullptr; }
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = first_cu; c < first_cu + n_cus; ++c) mask[c >> 5] |= 1u << (c & 31);
    hipStream_t s = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e != hipSuccess) { (void)hipGetLastError(); ew_set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e)); return nullptr; }
    return (void*)s;
}
extern "C" ew_status ew_stream_destroy(void* stream) {
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) { (void)hipGetLastError(); ew_set_error("hipStreamDestroy: %s", hipGetErrorString(e)); return EW_ERR_HIP; }
    return EW_OK;
}
