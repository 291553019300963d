/* oracle/ref_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Our own glue compiled INTO oracle/_ref/libwmixref.so next to the unmodified
 * reference sources.  It contains no reference code; it
 *   (1) pins WebRTC's run-time kernel selection to the generic C path
 *       (cpu_features.cc:71-72 exports the two function
 *       pointers; aec_core.c:1451-1455 and aec_rdft.c:574-577 consult them), and
 *   (2) offers whole-run drivers so a Python test can push a [frames x pkt]
 *       buffer through the reference wrappers (src/webrtc.h:32-61) in one call.
 */
#include <stdint.h>
#include <stdbool.h>
#include <stdlib.h>
#include <string.h>
#include "webrtc.h"

extern int (*WebRtc_GetCPUInfo)(int);
extern int (*WebRtc_GetCPUInfoNoASM)(int);

__attribute__((constructor)) static void ref_pin_ctor(void) { WebRtc_GetCPUInfo = WebRtc_GetCPUInfoNoASM; }

int ref_pin_generic_c(void)
{
    WebRtc_GetCPUInfo = WebRtc_GetCPUInfoNoASM;
    return WebRtc_GetCPUInfo(0) == 0 && WebRtc_GetCPUInfo(1) == 0;
}

static bool g_dbg = false;

/* Each driver: `n_calls` wrapper calls of `frames_per_call` frames (a frame =
 * chn int16).  in/out are contiguous; out may alias in. Returns 0 or the first
 * error code. */
int ref_run_ns(int chn, int freq, const int16_t *in, int16_t *out, int frames_per_call, int n_calls)
{
    void *h = ns_init(chn, freq, &g_dbg);
    if (!h) return -100;
    size_t step = (size_t)frames_per_call * chn;
    if (out != in) memcpy(out, in, step * n_calls * sizeof(int16_t));
    for (int i = 0; i < n_calls; i++) ns_process(h, out + i * step, out + i * step, frames_per_call);
    ns_release(h);
    return 0;
}

/* the same wrapper built with -DMAKE_WEBRTC_NSX (oracle/Makefile): the reference's fixed-point noise suppressor */
void *nsx_ns_init(int chn, int freq, bool *debug);
void nsx_ns_process(void *fp, int16_t *frame, int16_t *frameOut, int frameNum);
void nsx_ns_release(void *fp);

int ref_run_nsx(int chn, int freq, const int16_t *in, int16_t *out, int frames_per_call, int n_calls)
{
    void *h = nsx_ns_init(chn, freq, &g_dbg);
    if (!h) return -100;
    size_t step = (size_t)frames_per_call * chn;
    if (out != in) memcpy(out, in, step * n_calls * sizeof(int16_t));
    for (int i = 0; i < n_calls; i++) nsx_ns_process(h, out + i * step, out + i * step, frames_per_call);
    nsx_ns_release(h);
    return 0;
}

int ref_run_agc(int chn, int freq, int value, const int16_t *in, int16_t *out, int frames_per_call, int n_calls)
{
    void *h = agc_init(chn, freq, 10, value, &g_dbg);
    if (!h) return -100;
    size_t step = (size_t)frames_per_call * chn;
    if (out != in) memcpy(out, in, step * n_calls * sizeof(int16_t));
    int rc = 0;
    for (int i = 0; i < n_calls && rc == 0; i++) rc = agc_process(h, out + i * step, out + i * step, frames_per_call);
    agc_release(h);
    return rc;
}

int ref_run_vad(int chn, int freq, int interval_ms, const int16_t *in, int16_t *out, int frames_per_call, int n_calls)
{
    void *h = vad_init(chn, freq, interval_ms, &g_dbg);
    if (!h) return -100;
    size_t step = (size_t)frames_per_call * chn;
    if (out != in) memcpy(out, in, step * n_calls * sizeof(int16_t));
    for (int i = 0; i < n_calls; i++) vad_process(h, out + i * step, frames_per_call);
    vad_release(h);
    return 0;
}

int ref_run_aec(int chn, int freq, int interval_ms, const int16_t *far, const int16_t *near, int16_t *out,
                int frames_per_call, int n_calls, int delay_ms)
{
    void *h = aec_init(chn, freq, interval_ms, &g_dbg);
    if (!h) return -100;
    size_t step = (size_t)frames_per_call * chn;
    int rc = 0;
    for (int i = 0; i < n_calls && rc == 0; i++)
        rc = aec_process2(h, (int16_t *)far + i * step, (int16_t *)near + i * step, out + i * step, frames_per_call, delay_ms);
    aec_release(h);
    return rc;
}

int ref_run_aec_delays(int chn, int freq, int interval_ms, const int16_t *far, const int16_t *near, int16_t *out,
                       int frames_per_call, int n_calls, const int32_t *delay_ms)
{
    void *h = aec_init(chn, freq, interval_ms, &g_dbg);
    if (!h) return -100;
    size_t step = (size_t)frames_per_call * chn;
    int rc = 0;
    for (int i = 0; i < n_calls && rc == 0; i++)
        rc = aec_process2(h, (int16_t *)far + i * step, (int16_t *)near + i * step, out + i * step, frames_per_call, delay_ms[i]);
    aec_release(h);
    return rc;
}

/* the same wrapper built with the reference's AECM switch (oracle/Makefile, oracle/aecm_switch/): WebRtcAecm_* */
void *aecm_aec_init(int chn, int freq, int intervalMs, bool *debug);
int aecm_aec_setFrameFar(void *fp, int16_t *frameFar, int frameNum);
int aecm_aec_process(void *fp, int16_t *frameNear, int16_t *frameOut, int frameNum, int delayms);
int aecm_aec_process2(void *fp, int16_t *frameFar, int16_t *frameNear, int16_t *frameOut, int frameNum, int delayms);
void aecm_aec_release(void *fp);

/* split = 0: aec_process2 per call; split = 1: aec_setFrameFar then aec_process (the two-call form, src/webrtc.c:286-395) */
int ref_run_aecm(int chn, int freq, int interval_ms, const int16_t *far, const int16_t *near, int16_t *out,
                 int frames_per_call, int n_calls, int delay_ms, int split)
{
    void *h = aecm_aec_init(chn, freq, interval_ms, &g_dbg);
    if (!h) return -100;
    size_t step = (size_t)frames_per_call * chn;
    int rc = 0;
    for (int i = 0; i < n_calls && rc == 0; i++) {
        if (split) {
            rc = aecm_aec_setFrameFar(h, (int16_t *)far + i * step, frames_per_call);
            if (rc == 0) rc = aecm_aec_process(h, (int16_t *)near + i * step, out + i * step, frames_per_call, delay_ms);
        } else {
            rc = aecm_aec_process2(h, (int16_t *)far + i * step, (int16_t *)near + i * step, out + i * step, frames_per_call, delay_ms);
        }
    }
    aecm_aec_release(h);
    return rc;
}

/* The daemon's record chain (src/wmix.c:613-709): NS -> AEC -> AGC -> VAD, all in place. */
int ref_run_chain_iv(int chn, int freq, int interval_ms, int agc_value, unsigned stages, const int16_t *far, const int16_t *near,
                     int16_t *out, int frames_per_call, int n_calls)
{
    void *ns = (stages & 1) ? ns_init(chn, freq, &g_dbg) : NULL;
    void *aec = (stages & 2) ? aec_init(chn, freq, interval_ms, &g_dbg) : NULL;
    void *agc = (stages & 4) ? agc_init(chn, freq, interval_ms, agc_value, &g_dbg) : NULL;
    void *vad = (stages & 8) ? vad_init(chn, freq, interval_ms, &g_dbg) : NULL;
    size_t step = (size_t)frames_per_call * chn;
    int rc = 0;
    memcpy(out, near, step * n_calls * sizeof(int16_t));
    for (int i = 0; i < n_calls && rc == 0; i++) {
        int16_t *p = out + i * step;
        if (ns) ns_process(ns, p, p, frames_per_call);
        if (aec) rc = aec_process2(aec, (int16_t *)far + i * step, p, p, frames_per_call, 0);
        if (agc && rc == 0) rc = agc_process(agc, p, p, frames_per_call);
        if (vad && rc == 0) vad_process(vad, p, frames_per_call);
    }
    if (ns) ns_release(ns);
    if (aec) aec_release(aec);
    if (agc) agc_release(agc);
    if (vad) vad_release(vad);
    return rc;
}

int ref_run_chain(int chn, int freq, int agc_value, unsigned stages, const int16_t *far, const int16_t *near,
                  int16_t *out, int frames_per_call, int n_calls)
{
    return ref_run_chain_iv(chn, freq, 10, agc_value, stages, far, near, out, frames_per_call, n_calls);
}
