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

/* the same heartbeat as a handle set that lives across calls (what the daemon keeps in wmix->webrtcPoint[], src/wmix.c:613-709):
 * one call = one package through the enabled stages, in place */
typedef struct {
    orc_ns *ns;
    orc_aec *aec;
    orc_agc *agc;
    orc_vad *vad;
    int chn;
} orc_chain;

orc_chain *orc_chain_open(int chn, int freq, int interval_ms, int agc_value, unsigned stages)
{
    orc_chain *c = (orc_chain *)calloc(1, sizeof(*c));
    if (!c) return NULL;
    c->chn = chn;
    c->ns = (stages & 1) ? orc_ns_init(chn, freq) : NULL;
    c->aec = (stages & 2) ? orc_aec_init(chn, freq, interval_ms) : NULL;
    c->agc = (stages & 4) ? orc_agc_init(chn, freq, interval_ms, agc_value) : NULL;
    c->vad = (stages & 8) ? orc_vad_init(chn, freq, interval_ms) : NULL;
    return c;
}

int orc_chain_step(orc_chain *c, const int16_t *far, int16_t *pcm, int frames)
{
    int rc = 0;
    if (c->ns) orc_ns_run(c->ns, pcm, pcm, frames);
    if (c->aec) rc = orc_aec_process2(c->aec, far, pcm, pcm, frames, 0);
    if (c->agc && rc == 0) rc = orc_agc_run(c->agc, pcm, pcm, frames);
    if (c->vad && rc == 0) orc_vad_run(c->vad, pcm, frames);
    return rc;
}

void orc_chain_close(orc_chain *c)
{
    if (!c) return;
    if (c->ns) orc_ns_release(c->ns);
    if (c->aec) orc_aec_release(c->aec);
    if (c->agc) orc_agc_release(c->agc);
    if (c->vad) orc_vad_release(c->vad);
    free(c);
}
