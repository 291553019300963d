// compat_webrtc.hip -- the reference's per-handle wrapper API (src/webrtc.h:32-61) exported
// unchanged over HOST buffers, as thin adapters over a batch of ONE stream: each call stages
// the caller's int16 buffer in mapped pinned memory, launches the batched kernel on the HANDLE'S OWN
// non-blocking stream and waits for that stream alone (never the NULL stream: wmx_internal.h).
// This is what lets the wmix daemon link against libwmix_amd.so instead of src/webrtc.c + the
// five libwebrtc*.so; throughput comes from the wmx_* batch API, not from here.
//
// Semantics kept from src/webrtc.c: *_init returns NULL for unsupported rates; handles keep the
// caller's `bool *debug` and print only when it is set; in == out aliasing is fine; frameNum is
// in frames (chn samples each) and must be a multiple of the packet size.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include "wmx_internal.h"
#include "../../include/wmix_compat.h"

namespace {

// The handle's staging buffer: PINNED HOST memory mapped into the device's address space.  A legacy call moves a few hundred bytes;
// two runtime copies around the launch cost more than the work (round 5: ~ 45 us per call, of which the kernel ~ 8).  The kernels
// read the packet straight out of this buffer and write the result back into it over PCIe -- two plain memcpy calls on the host,
// one launch, one synchronisation.
struct DevBuf {
    int16_t *p = nullptr;  // the device's view (what the wmx_* entry points get)
    int16_t *host = nullptr;
    size_t cap = 0;  // int16 elements
    bool ensure(size_t n) {
        if (n <= cap) return true;
        if (host) (void)hipHostFree(host);
        host = p = nullptr;
        cap = 0;
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, n * sizeof(int16_t), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (hp) (void)hipHostFree(hp);
            return false;
        }
        host = static_cast<int16_t *>(hp);
        p = static_cast<int16_t *>(dp);
        cap = n;
        return true;
    }
    bool in(const int16_t *src, size_t n_elems) {
        memcpy(host, src, n_elems * sizeof(int16_t));
        return true;
    }
    // the launches of this call (on the handle's stream) have finished and their writes are in host memory
    bool out(int16_t *dst, size_t n_elems, hipStream_t s) {
        if (hipStreamSynchronize(s) != hipSuccess) return false;
        memcpy(dst, host, n_elems * sizeof(int16_t));
        return true;
    }
    ~DevBuf() {
        if (host && !wmx::runtime_exiting()) (void)hipHostFree(host);
    }
};

struct NsHandleCompat {
    hipStream_t s = nullptr;  // this handle's launch stream (wmx_internal.h: legacy_stream_create)
    wmx_ns *batch;    // float NS (the reference's default build) ...
    wmx_nsx *batchx;  // ... or the fixed-point NSX (its MAKE_WEBRTC_NSX build): exactly one is set
    int chn, freq, pkg;
    bool *debug;
    DevBuf buf;
};

struct VadHandleCompat {
    hipStream_t s = nullptr;  // this handle's launch stream (wmx_internal.h: legacy_stream_create)
    wmx_vad *batch;
    int chn, freq, pkg;
    bool *debug;
    DevBuf buf;
};

struct AgcHandleCompat {
    hipStream_t s = nullptr;  // this handle's launch stream (wmx_internal.h: legacy_stream_create)
    // agc_addition comes from the daemon's message thread while the record thread is inside agc_process (src/wmix.c:1070 beside :684-690);
    // the reference has no lock there and lives with it, the batch's host-side tables need one
    std::mutex mu;
    wmx_agc *batch;
    int chn, freq, pkg;
    bool *debug;
    DevBuf buf;
};

struct AecHandleCompat {
    hipStream_t s = nullptr;
    wmx_aec *batch;    // float AEC (the reference's default build) ...
    wmx_aecm *batchm;  // ... or the fixed-point AECM (its `#undef MAKE_WEBRTC_AEC` build): exactly one is set
    int chn, freq, pkg;
    bool *debug;
    DevBuf far, near;
};

// shared body of aec_setFrameFar / aec_process / aec_process2 (src/webrtc.c:286-483)
int aec_run_host(AecHandleCompat *h, int mode, int16_t *far, int16_t *nearp, int16_t *out, int frameNum, int delayms) {
    const int per_pkt = h->pkg * h->chn, total = frameNum * h->chn;
    const int n_packets = (total + per_pkt - 1) / per_pkt;
    if (n_packets <= 0) return 0;
    const size_t n = (size_t)n_packets * per_pkt;
    bool ok = true;
    if (mode & 1) ok = ok && h->far.ensure(n) && h->far.in(far, (size_t)total);
    if (mode & 2) ok = ok && h->near.ensure(n) && h->near.in(nearp, (size_t)total);
    int rc = -1;
    if (ok) {
        rc = h->batchm ? wmx_aecm_run(h->batchm, mode, (mode & 1) ? h->far.p : nullptr, per_pkt, (mode & 2) ? h->near.p : nullptr,
                                      (mode & 2) ? h->near.p : nullptr, n_packets, 0, per_pkt, delayms, h->s)
                       : wmx_aec_run(h->batch, mode, (mode & 1) ? h->far.p : nullptr, per_pkt, (mode & 2) ? h->near.p : nullptr,
                                     (mode & 2) ? h->near.p : nullptr, n_packets, 0, per_pkt, delayms, h->s);
        if (rc == 0 || rc == -1) {
            // rc == -1: the reference returned mid-buffer; packets before the offending one were written
            if (mode & 2) {
                if (!h->near.out(out, (size_t)total, h->s)) rc = -1;
            } else if (hipStreamSynchronize(h->s) != hipSuccess) {  // aec_setFrameFar: the far buffer is reused by the next call
                rc = -1;
            }
        }
    }
    if (rc != 0) {
        (void)hipGetLastError();
        if (h->debug && *h->debug) printf("WebRtcAecX_Process failed !!, ret %d \r\n", rc);
        // the reference's own -1 (a delay outside [0, 500]) is the caller's business; a HIP or WMX_E* failure is the operator's
        if (rc != -1) fprintf(stderr, "wmix_amd: aec_process failed on the GPU (rc %d): %s\n", rc, wmx_last_error());
    }
    return rc == 0 ? 0 : -1;  // every failure is the reference's -1 to the daemon (HIP errors live at WMX_EHIP_BASE - e, never at -1)
}

}  // namespace

extern "C" {

// src/webrtc.c:217-274
void *aec_init(int chn, int freq, int intervalMs, bool *debug) {
    if (freq > 16000 || freq % 8000 != 0) return NULL;
    // The reference chooses between WebRtcAec_* and WebRtcAecm_* by a source edit (`#undef MAKE_WEBRTC_AEC`,
    // src/webrtc.c:168-191).  One library serves both builds: WMIX_AMD_AECM=1 in the daemon's environment is that switch.
    const char *sw = getenv("WMIX_AMD_AECM");
    const bool mobile = sw && sw[0] == '1';
    wmx_aec *b = nullptr;
    wmx_aecm *bm = nullptr;
    if ((mobile ? wmx_aecm_create(&bm, 1, chn, freq, intervalMs) : wmx_aec_create(&b, 1, chn, freq, intervalMs)) != 0) {
        if (debug && *debug) printf("WebRtcAecX_Create failed !! (%s)\r\n", wmx_last_error());
        return NULL;
    }
    AecHandleCompat *h = new AecHandleCompat();
    h->batch = b;
    h->batchm = bm;
    h->chn = chn;
    h->freq = freq;
    h->pkg = (mobile ? wmx_aecm_packet_samples(bm) : wmx_aec_packet_samples(b)) / chn;
    h->debug = debug;
    h->s = wmx::legacy_stream_create();
    if (debug && *debug)
        printf("aec_init: chn/%d freq/%d intervalMs/%d pkgFrame/%d x %d\r\n", chn, freq, h->pkg / (freq / 1000), h->pkg, chn);
    return h;
}

// src/webrtc.c:286-323
int aec_setFrameFar(void *fp, int16_t *frameFar, int frameNum) {
    return aec_run_host(static_cast<AecHandleCompat *>(fp), 1, frameFar, nullptr, nullptr, frameNum, 0);
}

// src/webrtc.c:337-395.  Note: when the reference aborts mid-buffer its frameOut keeps whatever the caller had
// there for the unprocessed packets; with in == out (the daemon's use) that is the unprocessed input, same here.
int aec_process(void *fp, int16_t *frameNear, int16_t *frameOut, int frameNum, int delayms) {
    return aec_run_host(static_cast<AecHandleCompat *>(fp), 2, nullptr, frameNear, frameOut, frameNum, delayms);
}

// src/webrtc.c:410-483
int aec_process2(void *fp, int16_t *frameFar, int16_t *frameNear, int16_t *frameOut, int frameNum, int delayms) {
    return aec_run_host(static_cast<AecHandleCompat *>(fp), 3, frameFar, frameNear, frameOut, frameNum, delayms);
}

// src/webrtc.c:488-505
void aec_release(void *fp) {
    AecHandleCompat *h = static_cast<AecHandleCompat *>(fp);
    if (!h) return;
    if (h->batch) wmx_aec_destroy(h->batch);
    if (h->batchm) wmx_aecm_destroy(h->batchm);
    wmx::legacy_stream_destroy(h->s);
    if (h->debug && *h->debug) printf("aec_release\r\n");
    delete h;
}

// src/webrtc.c:40-82
void *vad_init(int chn, int freq, int intervalMs, bool *debug) {
    if (freq > 32000 || freq % 8000 != 0) return NULL;
    wmx_vad *b = nullptr;
    if (wmx_vad_create(&b, 1, chn, freq, intervalMs) != 0) {
        if (debug && *debug) printf("WebRtcVad_Create failed !! (%s)\r\n", wmx_last_error());
        return NULL;
    }
    VadHandleCompat *h = new VadHandleCompat();
    h->batch = b;
    h->chn = chn;
    h->freq = freq;
    h->pkg = wmx_vad_packet_samples(b) / chn;
    h->debug = debug;
    h->s = wmx::legacy_stream_create();
    if (debug && *debug) printf("vad_init: chn/%d freq/%d intervalMs/%d pkgFrame/%d\r\n", chn, freq, h->pkg / (freq / 1000), h->pkg);
    return h;
}

// src/webrtc.c:91-151: one call = ceil(frameNum / pkgFrame) decisions, all on packet 0
void vad_process(void *fp, int16_t *frame, int frameNum) {
    VadHandleCompat *h = static_cast<VadHandleCompat *>(fp);
    const int packets = (frameNum + h->pkg - 1) / h->pkg;
    if (packets <= 0) return;
    const size_t n = (size_t)packets * h->pkg * h->chn, given = (size_t)frameNum * h->chn;
    bool ok = h->buf.ensure(n);
    ok = ok && h->buf.in(frame, given);
    ok = ok && wmx_vad_process(h->batch, h->buf.p, packets, 1, 0, (long)n, h->s) == 0;
    ok = ok && h->buf.out(frame, given, h->s);
    if (!ok) {
        (void)hipGetLastError();
        fprintf(stderr, "wmix_amd: vad_process failed on the GPU: %s\n", wmx_last_error());
    }
}

// src/webrtc.c:156-164
void vad_release(void *fp) {
    VadHandleCompat *h = static_cast<VadHandleCompat *>(fp);
    if (!h) return;
    wmx_vad_destroy(h->batch);
    wmx::legacy_stream_destroy(h->s);
    if (h->debug && *h->debug) printf("vad_release\r\n");
    delete h;
}

// src/webrtc.c:694-753
void *agc_init(int chn, int freq, int intervalMs, int value, bool *debug) {
    if (freq > 32000 || freq % 8000 != 0) return NULL;
    wmx_agc *b = nullptr;
    if (wmx_agc_create(&b, 1, chn, freq, intervalMs, value) != 0) {
        if (debug && *debug) printf("WebRtcAgc_set_config failed !! (%s)\r\n", wmx_last_error());
        return NULL;
    }
    AgcHandleCompat *h = new AgcHandleCompat();
    h->batch = b;
    h->chn = chn;
    h->freq = freq;
    h->pkg = wmx_agc_packet_samples(b) / chn;
    h->debug = debug;
    h->s = wmx::legacy_stream_create();
    if (debug && *debug)
        printf("agc_init: chn/%d freq/%d intervalMs/%d pkgFrame/%d x %d\r\n", chn, freq, freq <= 16000 ? 10 : 5, h->pkg, chn);
    return h;
}

// src/webrtc.c:767-819: 0 on success, -1 on failure
int agc_process(void *fp, int16_t *frame, int16_t *frameOut, int frameNum) {
    AgcHandleCompat *h = static_cast<AgcHandleCompat *>(fp);
    const int per_pkt = h->pkg * h->chn, total = frameNum * h->chn;
    const int n_packets = (total + per_pkt - 1) / per_pkt;
    if (n_packets <= 0) return 0;
    const size_t n = (size_t)n_packets * per_pkt;
    std::lock_guard<std::mutex> lock(h->mu);
    bool ok = h->buf.ensure(n);
    ok = ok && h->buf.in(frame, (size_t)total);
    ok = ok && wmx_agc_process(h->batch, h->buf.p, h->buf.p, n_packets, 0, per_pkt, h->s) == 0;
    ok = ok && h->buf.out(frameOut, (size_t)total, h->s);
    if (!ok) {
        (void)hipGetLastError();
        if (h->debug && *h->debug) printf("WebRtcAgc_Process failed !!, ret %d \r\n", -1);
        fprintf(stderr, "wmix_amd: agc_process failed on the GPU: %s\n", wmx_last_error());
        return -1;
    }
    return 0;
}

// src/webrtc.c:824-839
void agc_addition(void *fp, uint8_t value) {
    AgcHandleCompat *h = static_cast<AgcHandleCompat *>(fp);
    std::lock_guard<std::mutex> lock(h->mu);
    // ordered on the handle's own stream (nothing else of the device is waited for): the new table is in place when the call returns
    const int32_t all = 0;
    int ret = wmx_agc_set_gain_streams(h->batch, &all, 1, (int)value, h->s);
    if (ret == 0 && hipStreamSynchronize(h->s) != hipSuccess) {
        (void)hipGetLastError();
        ret = -1;
    }
    if (ret != 0 && h->debug && *h->debug) printf("WebRtcAgc_set_config failed !!, ret %d \r\n", -1);
}

// src/webrtc.c:844-860
void agc_release(void *fp) {
    AgcHandleCompat *h = static_cast<AgcHandleCompat *>(fp);
    if (!h) return;
    wmx_agc_destroy(h->batch);
    wmx::legacy_stream_destroy(h->s);
    if (h->debug && *h->debug) printf("agc_release\r\n");
    delete h;
}

// src/webrtc.c:560-602
void *ns_init(int chn, int freq, bool *debug) {
    if (freq > 32000 || freq % 8000 != 0) return NULL;
    // The reference chooses between WebRtcNs_* and WebRtcNsx_* at build time (#define MAKE_WEBRTC_NSX, src/webrtc.c:512-521).
    // One library serves both builds: WMIX_AMD_NSX=1 in the daemon's environment is that switch.
    const char *sw = getenv("WMIX_AMD_NSX");
    const bool fixed = sw && sw[0] == '1';
    wmx_ns *b = nullptr;
    wmx_nsx *bx = nullptr;
    if ((fixed ? wmx_nsx_create(&bx, 1, chn, freq) : wmx_ns_create(&b, 1, chn, freq)) != 0) {
        if (debug && *debug) printf("WebRtcNs_Create failed !! (%s)\r\n", wmx_last_error());
        return NULL;
    }
    NsHandleCompat *h = new NsHandleCompat();
    h->batch = b;
    h->batchx = bx;
    h->chn = chn;
    h->freq = freq;
    h->pkg = freq / 1000 * 10;
    h->debug = debug;
    h->s = wmx::legacy_stream_create();
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
    ok = ok && h->buf.in(frame, (size_t)total);
    ok = ok && (h->batchx ? wmx_nsx_process(h->batchx, h->buf.p, h->buf.p, n_packets, 0, per_pkt, h->s)
                          : wmx_ns_process(h->batch, h->buf.p, h->buf.p, n_packets, 0, per_pkt, h->s)) == 0;
    ok = ok && h->buf.out(frameOut, (size_t)total, h->s);
    if (!ok) {
        (void)hipGetLastError();
        fprintf(stderr, "wmix_amd: ns_process failed on the GPU: %s\n", wmx_last_error());
    }
}

// src/webrtc.c:649-661
void ns_release(void *fp) {
    NsHandleCompat *h = static_cast<NsHandleCompat *>(fp);
    if (!h) return;
    if (h->batch) wmx_ns_destroy(h->batch);
    if (h->batchx) wmx_nsx_destroy(h->batchx);
    wmx::legacy_stream_destroy(h->s);
    if (h->debug && *h->debug) printf("ns_release\r\n");
    delete h;
}

}  // extern "C"
