/* oracle/orc_aec.h -- TEST INFRASTRUCTURE ONLY. See orc_aec.c. */
#ifndef ORC_AEC_H
#define ORC_AEC_H
#include <stdint.h>
#include "orc_fft.h"

#define ORC_AEC_FAR_BLOCKS 250 /* kBufSizePartitions, aec_core.c:38 */

typedef struct { /* RingBuffer, ring_buffer.c:25-32 */
    float *data;
    int count, esize, rd, wr, diff_wrap;
} orc_ring;

typedef struct {
    /* ---- AecCore (aec_core_internal.h:52-169), normal mode, one band */
    int fs, mult, nlp_mode;
    float mu, err_thr;
    orc_ring near_fr, out_fr, far_buf, far_buf_w, far_pre;
    float near_store[144], out_store[144], farpre_store[128 + 320];
    float far_store[ORC_AEC_FAR_BLOCKS * 130], farw_store[ORC_AEC_FAR_BLOCKS * 130];
    float dBuf[128], eBuf[128], outBuf[64];
    float xPow[65], dPow[65], dMinPow[65], dInitMinPow[65];
    int noise_is_init, noise_ctr;
    float xf[2][12][65], wf[2][12][65];
    float sde[65][2], sxd[65][2], sx[65], sd[65], se[65];
    float xfwBuf[32][130];
    float hNlFbMin, hNlFbLocalMin, hNlXdAvgMin, overDrive, overDriveSm;
    int hNlNewMin, hNlMinCtr, delayIdx, delay_est_ctr, xf_pos;
    short stNearState, echoState, divergeState;
    int system_delay, core_known_delay;
    uint32_t seed;
    /* ---- Aec wrapper (echo_cancellation_internal.h:17-65) */
    int rate_factor, bufSizeStart, knownDelay, sum, timeForDelayChange, startup_phase, checkBuffSize, farend_started;
    short counter, firstVal, checkBufSizeCtr, msInSndCardBuf, filtDelay, lastDelayDiff;
    /* ---- wmix wrapper (src/webrtc.c:196-207) */
    int chn, pkg;
} orc_aec;

void orc_aec_core_setup(orc_aec *a, int fs);
int orc_aec_buffer_farend(orc_aec *a, const float *far, int n);
int orc_aec_process(orc_aec *a, const float *nearend, float *out, int n, int ms_in_snd_card_buf);
orc_aec *orc_aec_init(int chn, int freq, int interval_ms);
int orc_aec_process2(orc_aec *a, const int16_t *far, const int16_t *nearp, int16_t *out, int frame_num, int delay_ms);
void orc_aec_release(orc_aec *a);
void orc_aec_probe(const orc_aec *a, int32_t *ints8, float *floats7);
int orc_run_aec(int chn, int freq, int interval_ms, const int16_t *far, const int16_t *nearp, int16_t *out, int frames_per_call,
                int n_calls, int delay_ms);
int orc_run_aec_seeded(int chn, int freq, int interval_ms, const int16_t *far, const int16_t *nearp, int16_t *out, int frames_per_call,
                       int n_calls, int delay_ms, uint32_t seed);
int orc_run_aec_delays(int chn, int freq, int interval_ms, const int16_t *far, const int16_t *nearp, int16_t *out, int frames_per_call,
                       int n_calls, const int32_t *delay_ms);
#endif
