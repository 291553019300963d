/* oracle/orc_chain.c -- TEST INFRASTRUCTURE ONLY.
 * The daemon's record chain, src/wmix.c:613-709: NS -> AEC -> AGC -> VAD, all in place on the
 * same packet buffer.  stages bitmask: 1 NS, 2 AEC, 4 AGC, 8 VAD. */
#include <stdlib.h>
#include <string.h>
#include "orc_aec.h"
#include "orc_agc.h"
#include "orc_ns.h"
#include "orc_vad.h"

/* interval_ms: what the daemon hands aec_init / agc_init / vad_init (WMIX_INTERVAL_MS = 20, src/wmixConf.h:112, src/wmix.c:636,684,703) */
int orc_run_chain_iv(int chn, int freq, int interval_ms, int agc_value, unsigned stages, const int16_t *far, const int16_t *nearp,
                     int16_t *out, int frames_per_call, int n_calls)
{
    orc_ns *ns = (stages & 1) ? orc_ns_init(chn, freq) : NULL;
    orc_aec *aec = (stages & 2) ? orc_aec_init(chn, freq, interval_ms) : NULL;
    orc_agc *agc = (stages & 4) ? orc_agc_init(chn, freq, interval_ms, agc_value) : NULL;
    orc_vad *vad = (stages & 8) ? orc_vad_init(chn, freq, interval_ms) : NULL;
    size_t step = (size_t)frames_per_call * chn;
    int rc = 0;
    memcpy(out, nearp, step * n_calls * sizeof(int16_t));
    for (int i = 0; i < n_calls && rc == 0; i++) {
        int16_t *p = out + i * step;
        if (ns) orc_ns_run(ns, p, p, frames_per_call);
        if (aec) rc = orc_aec_process2(aec, far + i * step, p, p, frames_per_call, 0);
        if (agc && rc == 0) rc = orc_agc_run(agc, p, p, frames_per_call);
        if (vad && rc == 0) orc_vad_run(vad, p, frames_per_call);
    }
    if (ns) orc_ns_release(ns);
    if (aec) orc_aec_release(aec);
    if (agc) orc_agc_release(agc);
    if (vad) orc_vad_release(vad);
    return rc;
}

int orc_run_chain(int chn, int freq, int agc_value, unsigned stages, const int16_t *far, const int16_t *nearp, int16_t *out,
                  int frames_per_call, int n_calls)
{
    return orc_run_chain_iv(chn, freq, 10, agc_value, stages, far, nearp, out, frames_per_call, n_calls);
}
