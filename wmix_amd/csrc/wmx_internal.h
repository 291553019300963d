// wmx_internal.h -- shared host-side helpers for libwmix_amd.so (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/wmix_amd.h"

namespace wmx {

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Grid size for a grid-stride, HBM-bound kernel: enough workgroups to fill
// 256 CUs x 8 (guide: Guideline 11), never more than the work needs.
inline unsigned stream_grid(size_t work_items, unsigned block) {
    size_t need = (work_items + block - 1) / block;
    size_t cap = 256u * 8u;
    if (need < 1) need = 1;
    return (unsigned)(need < cap ? need : cap);
}

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
