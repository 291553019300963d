// wmx_core.hip -- error plumbing and device queries for libwmix_amd.so.
#include <atomic>
#include <cstdlib>
#include "wmx_internal.h"

namespace wmx {

static thread_local char g_err[512] = "";

// see runtime_exiting() in wmx_internal.h
static std::atomic<bool> g_exiting{false};
static void mark_exiting() { g_exiting.store(true); }
static const int g_exit_hook = (atexit(mark_exiting), 0);  // registered at load time: behind the HIP runtime's handlers, so it runs first
__attribute__((destructor)) static void on_unload() { g_exiting.store(true); }
bool runtime_exiting() { return g_exiting.load(); }

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    set_error("HIP error %d (%s) at %s:%d in %s", (int)e, hipGetErrorString(e), file, line, what);
    (void)hipGetLastError();  // clear the sticky per-thread error
    return e == hipErrorNoDevice ? WMX_ENODEV : -(int)e;
}

int current_device() {
    int d = -1;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) {
        hip_fail(e, "hipGetDevice", __FILE__, __LINE__);
        return -1;
    }
    return d;
}

}  // namespace wmx

extern "C" {

const char *wmx_last_error(void) { return wmx::g_err; }

int wmx_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        wmx::hip_fail(e, "hipGetDeviceCount", __FILE__, __LINE__);
        return 0;
    }
    return n;
}

int wmx_version(void) { return 310; }  // 310: platform setters, rwTest, the PCM pipeline (round 5, second half)

// "default" for the product build; otherwise the developer flags it was made with (the Makefile's EXTRA), prefixed "TIMING-ONLY
// (wrong results): " when one of them is a timing experiment's switch.  See build_flags.h.
const char *wmx_build_info(void) {
#ifdef WMX_TIMING_ONLY_BUILD
    return "TIMING-ONLY (wrong results): " WMX_BUILD_EXTRA;
#else
    return WMX_BUILD_EXTRA[0] ? WMX_BUILD_EXTRA : "default";
#endif
}

// `int device` is the first member of every wmx_* handle struct
int wmx_handle_device(const void *handle) { return handle ? *static_cast<const int *>(handle) : WMX_EINVAL; }

}  // extern "C"
