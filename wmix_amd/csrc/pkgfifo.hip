// pkgfifo.hip -- the packet FIFOs that hand the AEC its delayed far-end, device resident.
//
// Replaces recordPkgBuff_add/get and playPkgBuff_add/get (src/wmix.c:432-526; geometry src/wmixConf.h:112-141) for n
// streams per call: every slot holds one interval's packet of every stream ([slot][stream][pkg_bytes] in HBM), `add`
// writes the slot under the cursor, `get(delayms)` copies out the packet the reference's index arithmetic selects.
// That arithmetic is data independent and runs on the host exactly as written -- including what it does for
// delayms = AEC_INTERVALMS, the only value the daemon passes -- and the kernels are plain coalesced copies.
#include "wmx_internal.h"

struct wmx_pkgfifo {
    int device;  // the HIP device the state lives on (current device at create); every entry point switches to it
    int n_streams, n_slots, pkg_bytes, interval_ms, frame_bytes, count;
    uint8_t *d_slots;
};

namespace wmx {
namespace {

// dst[s][0..lead) = tail of lead_slot, dst[s][lead..pkg) = head of slot (src/wmix.c:511-523); 4 bytes per thread
__global__ void pkgfifo_copy_kernel(uint8_t *dst, long dst_stride, const uint8_t *a, int a_off, int lead, const uint8_t *b, int pkg,
                                    long slot_stride, int n_streams) {
    const long total = (long)n_streams * pkg;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long s = i / pkg;
        const int j = (int)(i - s * pkg);
        dst[s * dst_stride + j] = j < lead ? a[s * slot_stride + a_off + j] : b[s * slot_stride + (j - lead)];
    }
}

}  // namespace
}  // namespace wmx

using namespace wmx;

extern "C" {

int wmx_pkgfifo_create(wmx_pkgfifo **out, int n_streams, int n_slots, int pkg_bytes, int interval_ms, int frame_bytes) {
    if (!out || n_streams < 1 || n_slots < 2 || pkg_bytes < 1 || interval_ms < 1 || frame_bytes < 1 || pkg_bytes % frame_bytes) {
        set_error("wmx_pkgfifo_create: bad geometry");
        return WMX_EINVAL;
    }
    wmx_pkgfifo *h = new wmx_pkgfifo{wmx::current_device(), n_streams, n_slots, pkg_bytes, interval_ms, frame_bytes, 0, nullptr};
    if (h->device < 0) {
        delete h;
        return WMX_ENODEV;
    }
    const size_t bytes = (size_t)n_slots * n_streams * pkg_bytes;
    if (hipMalloc(&h->d_slots, bytes) != hipSuccess) {
        delete h;
        set_error("wmx_pkgfifo_create: no HIP device / out of memory");
        return WMX_ENODEV;
    }
    const hipError_t e = hipMemset(h->d_slots, 0, bytes);  // the reference's FIFOs are zero-initialised statics
    if (e != hipSuccess) {
        const int rc = hip_fail(e, "hipMemset(slots)", __FILE__, __LINE__);
        wmx_pkgfifo_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

int wmx_pkgfifo_destroy(wmx_pkgfifo *h) {
    WMX_ON_DEVICE(h);
    if (!h) return 0;
    (void)hipFree(h->d_slots);
    delete h;
    return 0;
}

int wmx_pkgfifo_add(wmx_pkgfifo *h, const uint8_t *d_pkgs, long stride, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h || !d_pkgs || (h->n_streams > 1 && stride < h->pkg_bytes)) {
        set_error("wmx_pkgfifo_add: bad arguments");
        return WMX_EINVAL;
    }
    uint8_t *slot = h->d_slots + (size_t)h->count * h->n_streams * h->pkg_bytes;
    WMX_HIP(hipMemcpy2DAsync(slot, h->pkg_bytes, d_pkgs, h->n_streams > 1 ? (size_t)stride : (size_t)h->pkg_bytes, h->pkg_bytes, h->n_streams,
                             hipMemcpyDeviceToDevice, as_stream(stream)));
    if (++h->count >= h->n_slots) h->count = 0;  // src/wmix.c:489-491
    return 0;
}

int wmx_pkgfifo_get(wmx_pkgfifo *h, uint8_t *d_out, long stride, int delayms, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h || !d_out || delayms < 0 || (h->n_streams > 1 && stride < h->pkg_bytes)) {
        set_error("wmx_pkgfifo_get: bad arguments");
        return WMX_EINVAL;
    }
    // src/wmix.c:494-510, as written
    const int frames = h->pkg_bytes / h->frame_bytes;
    int k = h->count - delayms / h->interval_ms;
    const int lead = (int)((float)((delayms % h->interval_ms) * frames) / h->interval_ms) * h->frame_bytes;
    if (k >= h->n_slots)
        k = h->n_slots;
    else if (k < 0)
        k = 0;
    k = h->count - k;
    if (k >= h->n_slots)
        k -= h->n_slots;
    else if (k < 0)
        k += h->n_slots;
    // `_playPkgBuff[k - 1] - byteCount` (k == 0: `[NUM - 1] - byteCount`) points at the tail of the slot before that one
    const int lead_slot = (k == 0 ? h->n_slots - 1 : k - 1) - 1;
    if (lead > 0 && lead_slot < 0) {
        set_error("wmx_pkgfifo_get: delay %d ms selects bytes in front of the FIFO (undefined in the reference too)", delayms);
        return WMX_EINVAL;
    }
    const size_t slot_bytes = (size_t)h->n_streams * h->pkg_bytes;
    const uint8_t *b = h->d_slots + (size_t)k * slot_bytes;
    const uint8_t *a = lead > 0 ? h->d_slots + (size_t)lead_slot * slot_bytes : b;
    const long total = (long)h->n_streams * h->pkg_bytes;
    hipLaunchKernelGGL(pkgfifo_copy_kernel, dim3(stream_grid((size_t)total, 256)), dim3(256), 0, as_stream(stream), d_out,
                       h->n_streams > 1 ? stride : (long)h->pkg_bytes, a, h->pkg_bytes - lead, lead, b, h->pkg_bytes, (long)h->pkg_bytes,
                       h->n_streams);
    WMX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
