// compat_webrtc.hip -- the reference's per-handle wrapper API (src/webrtc.h:32-61) exported
// unchanged over HOST buffers, as thin adapters over a batch of ONE stream: each call stages
// the caller's int16 buffer in HBM, launches the batched kernel and copies the result back.
// This is what lets the wmix daemon link against libwmix_amd.so instead of src/webrtc.c + the
// five libwebrtc*.so; throughput comes from the wmx_* batch API, not from here.
//
// Semantics kept from src/webrtc.c: *_init returns NULL for unsupported rates; handles keep the
// caller's `bool *debug` and print only when it is set; in == out aliasing is fine; frameNum is
// in frames (chn samples each) and must be a multiple of the packet size.
#include <cstdlib>
#include "wmx_internal.h"
#include "../../include/wmix_compat.h"

namespace {

struct DevBuf {
    int16_t *p = nullptr;
    size_t cap = 0;  // int16 elements
    bool ensure(size_t n) {
        if (n <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        if (hipMalloc(&p, n * sizeof(int16_t)) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        cap = n;
        return true;
    }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};

struct NsHandleCompat {
    wmx_ns *batch;
    int chn, freq, pkg;
    bool *debug;
    DevBuf buf;
};

}  // namespace

extern "C" {

// src/webrtc.c:560-602
void *ns_init(int chn, int freq, bool *debug) {
    if (freq > 32000 || freq % 8000 != 0) return NULL;
    wmx_ns *b = nullptr;
    if (wmx_ns_create(&b, 1, chn, freq) != 0) {
        if (debug && *debug) printf("WebRtcNs_Create failed !! (%s)\r\n", wmx_last_error());
        return NULL;
    }
    NsHandleCompat *h = new NsHandleCompat();
    h->batch = b;
    h->chn = chn;
    h->freq = freq;
    h->pkg = freq / 1000 * 10;
    h->debug = debug;
    if (debug && *debug) printf("ns_init: chn/%d freq/%d intervalMs/%d pkgFrame/%d x %d\r\n", chn, freq, 10, h->pkg, chn);
    return h;
}

// src/webrtc.c:612-644
void ns_process(void *fp, int16_t *frame, int16_t *frameOut, int frameNum) {
    NsHandleCompat *h = static_cast<NsHandleCompat *>(fp);
    const int per_pkt = h->pkg * h->chn;
    const int total = frameNum * h->chn;
    const int n_packets = (total + per_pkt - 1) / per_pkt;  // the reference loop runs while cLen < realFrameLen
    if (n_packets <= 0) return;
    const size_t n = (size_t)n_packets * per_pkt;
    bool ok = h->buf.ensure(n);
    ok = ok && hipMemcpy(h->buf.p, frame, (size_t)total * sizeof(int16_t), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && wmx_ns_process(h->batch, h->buf.p, h->buf.p, n_packets, 0, per_pkt, nullptr) == 0;
    ok = ok && hipMemcpy(frameOut, h->buf.p, (size_t)total * sizeof(int16_t), hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        fprintf(stderr, "wmix_amd: ns_process failed on the GPU: %s\n", wmx_last_error());
    }
}

// src/webrtc.c:649-661
void ns_release(void *fp) {
    NsHandleCompat *h = static_cast<NsHandleCompat *>(fp);
    if (!h) return;
    wmx_ns_destroy(h->batch);
    if (h->debug && *h->debug) printf("ns_release\r\n");
    delete h;
}

}  // extern "C"
