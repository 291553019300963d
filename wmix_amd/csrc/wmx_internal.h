// wmx_internal.h -- shared host-side helpers for libwmix_amd.so (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/wmix_amd.h"

#define WMX_HIP_RC(expr)                                                      \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess) return wmx::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

namespace wmx {

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Device affinity of a handle.  The reference is one thread per device; a C host of ours runs one thread per GPU
// (SURVEY 8b "Threading") and must not depend on a thread-local current device: every handle
// records the device that was current when it was created and every entry point runs on it, restoring the caller's
// device afterwards.  hipGetDevice is a thread-local read; the switch happens only when the two differ.
int current_device();  // -1 (and the error set) when there is no usable device
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int dev) {
        if (dev < 0) return;
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~DeviceScope() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

// Grid size for a grid-stride, HBM-bound kernel: enough workgroups to fill
// 256 CUs x 8 (guide: Guideline 11), never more than the work needs.
inline unsigned stream_grid(size_t work_items, unsigned block) {
    size_t need = (work_items + block - 1) / block;
    size_t cap = 256u * 8u;
    if (need < 1) need = 1;
    return (unsigned)(need < cap ? need : cap);
}

// Device-resident schedules (gather lists, load schedules) the host builds per format.  An entry, once uploaded, is
// never overwritten or freed while other work may run: a kernel of an earlier call that still reads it on another
// stream stays valid (round-1 ADVICE: the scratch buffers used to be rewritten by the next call).  Formats are few --
// a daemon has a handful -- so entries simply accumulate; past kMax the cache drains the device before it frees them.
// The upload is a blocking copy, once per format.
struct SchedCache {
    struct Entry {
        uint64_t k0, k1;
        void *p;
        size_t n;  // elements
    };
    static constexpr size_t kMax = 64;
    std::vector<Entry> e;
    bool leak;  // thread_local instances: the HIP runtime may be gone when the thread or process ends
    explicit SchedCache(bool leak_at_exit = false) : leak(leak_at_exit) {}
    SchedCache(const SchedCache &) = delete;
    SchedCache &operator=(const SchedCache &) = delete;
    const Entry *find(uint64_t k0, uint64_t k1) const {
        for (const Entry &x : e)
            if (x.k0 == k0 && x.k1 == k1) return &x;
        return nullptr;
    }
    // returns 0 and *out on success, a WMX error otherwise
    int add(uint64_t k0, uint64_t k1, const void *host, size_t bytes, size_t n, const Entry **out) {
        if (e.size() >= kMax) {
            WMX_HIP_RC(hipDeviceSynchronize());
            clear();
        }
        void *p = nullptr;
        if (bytes) {
            WMX_HIP_RC(hipMalloc(&p, bytes));
            hipError_t er = hipMemcpy(p, host, bytes, hipMemcpyHostToDevice);
            if (er != hipSuccess) {
                (void)hipFree(p);
                return hip_fail(er, "hipMemcpy(schedule)", __FILE__, __LINE__);
            }
        }
        e.push_back(Entry{k0, k1, p, n});
        *out = &e.back();
        return 0;
    }
    void clear() {
        for (Entry &x : e)
            if (x.p) (void)hipFree(x.p);
        e.clear();
    }
    ~SchedCache() {
        if (!leak) clear();
    }
};

// wmix_pcm_zoom's cursor walk (src/wmix.c:139-222) as a gather list: out int16 i <- in int16 idx[i]; identical formats
// give the identity (the reference's memcpy branch).  Defined in mix.hip.
void zoom_gather_list(int inChn, int inFreq, uint32_t inLen, int outChn, int outFreq, std::vector<int32_t> &idx);

#ifdef __HIPCC__
// Lane-per-stream kernels (VAD, AGC) walk their state field by field along a sequential dependency chain, one wave
// per SIMD: every first touch of a field would expose a full HBM round trip.  touch_line() requests a line up front
// (result discarded; loads return in order, so ordinary loads behind the touches are still waited for correctly);
// the chain then runs against L2.
// The load's result arrives asynchronously, so its destination must be a register the compiler keeps reserved until
// the wave has waited for every touch: `sink` is threaded through all touches as a read-write operand and pinned by
// touch_done() at the end of the kernel.
__device__ __forceinline__ void touch_line(const void *p, int &sink) {
    asm volatile("global_load_ubyte %0, %1, off" : "+v"(sink) : "v"(p) : "memory");
}
__device__ __forceinline__ void touch_done(int &sink) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) : : "memory"); }
#endif

}  // namespace wmx

#define WMX_HIP(expr)                                                         \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess) return wmx::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define WMX_LAUNCH_CHECK() WMX_HIP(hipGetLastError())

// first statement of every entry point that takes a handle (null-safe: the null check follows it)
#define WMX_ON_DEVICE(h)                              \
    wmx::DeviceScope _dev_scope((h) ? (h)->device : -1); \
    WMX_HIP(_dev_scope.err)
