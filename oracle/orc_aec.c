/* oracle/orc_aec.c -- TEST INFRASTRUCTURE ONLY (CPU checker; never on the product path).
 *
 * Restatement of wmix's acoustic echo canceller (float AEC, "normal" 12-partition mode,
 * NLP aggressive, no skew compensation / metrics / delay logging):
 *   aec_init / aec_setFrameFar / aec_process / aec_process2     src/webrtc.c:217-505
 *   WebRtcAec_Init / set_config / BufferFarend / Process        W:modules/audio_processing/aec/echo_cancellation.c:179-444
 *   ProcessNormal, EstBufDelayNormal                            W:...echo_cancellation.c:599-747,821-872
 *   InitAec, BufferFarendPartition, MoveFarReadPtr, ProcessFrames  W:...aec_core.c:1527-1850
 *   ProcessBlock, NonLinearProcessing and their helpers         W:...aec_core.c:148-547,831-1351
 *   ring buffer                                                 W:common_audio/ring_buffer.c
 *   WebRtcSpl_RandUArray                                        W:common_audio/signal_processing/randomization_functions.c:94-112
 * The 128-point transforms are orc_fft.c's aec flavour (frozen rdft_w table).
 * Pinned against oracle/_ref (generic-C kernels) in tests/test_aec_oracle.py.
 * Compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "orc_aec.h"

#define PART 64
#define PART1 65
#define PART2 128
#define FRAME 80
#define NPART 12

/* ------------------------------------------------------------------ ring buffer (ring_buffer.c) */
static void ring_init(orc_ring *r, float *storage, int count, int esize)
{
    r->data = storage;
    r->count = count;
    r->esize = esize;
    r->rd = r->wr = 0;
    r->diff_wrap = 0;
    memset(storage, 0, sizeof(float) * (size_t)count * esize);
}
static int ring_avail_read(const orc_ring *r) { return r->diff_wrap ? r->count - r->rd + r->wr : r->wr - r->rd; }
static int ring_avail_write(const orc_ring *r) { return r->count - ring_avail_read(r); }

static int ring_move_read(orc_ring *r, int n)
{
    int freee = ring_avail_write(r), readable = ring_avail_read(r), pos = r->rd;
    if (n > readable) n = readable;
    if (n < -freee) n = -freee;
    pos += n;
    if (pos > r->count) {
        pos -= r->count;
        r->diff_wrap = 0;
    }
    if (pos < 0) {
        pos += r->count;
        r->diff_wrap = 1;
    }
    r->rd = pos;
    return n;
}

static int ring_write(orc_ring *r, const float *src, int n)
{
    int freee = ring_avail_write(r), w = freee < n ? freee : n, left = w, margin = r->count - r->wr;
    if (w > margin) {
        memcpy(r->data + (size_t)r->wr * r->esize, src, sizeof(float) * (size_t)margin * r->esize);
        r->wr = 0;
        left -= margin;
        r->diff_wrap = 1;
    }
    memcpy(r->data + (size_t)r->wr * r->esize, src + (size_t)(w - left) * r->esize, sizeof(float) * (size_t)left * r->esize);
    r->wr += left;
    return w;
}

/* copies out (the reference sometimes hands back a pointer into the ring; values are the same) */
static int ring_read(orc_ring *r, float *dst, int n)
{
    int readable = ring_avail_read(r), k = readable < n ? readable : n, margin = r->count - r->rd;
    if (k > margin) {
        memcpy(dst, r->data + (size_t)r->rd * r->esize, sizeof(float) * (size_t)margin * r->esize);
        memcpy(dst + (size_t)margin * r->esize, r->data, sizeof(float) * (size_t)(k - margin) * r->esize);
    } else {
        memcpy(dst, r->data + (size_t)r->rd * r->esize, sizeof(float) * (size_t)k * r->esize);
    }
    ring_move_read(r, k);
    return k;
}

/* ------------------------------------------------------------------ constant tables (aec_core.c:49-103) */
static float g_hanning[65], g_weight[65], g_overdrive[65];
static int g_tables_ready;
static void build_tables(void)
{
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < 65; i++) {
        g_hanning[i] = (float)sin(pi * i / 128.0); /* sqrt(hanning(63)) half, 14-decimal literals */
        g_overdrive[i] = (float)(floor((sqrt(i / 64.0) + 1.0) * 1e4 + 0.5) / 1e4);
        g_weight[i] = i == 0 ? 0.f : (float)(floor((0.3 * sqrt((i - 1) / 63.0) + 0.1) * 1e4 + 0.5) / 1e4);
    }
    g_tables_ready = 1;
}

/* ------------------------------------------------------------------ core init (aec_core.c:1527-1688 + SetConfigCore) */
static void core_init(orc_aec *a, int fs)
{
    if (fs == 8000) {
        a->mu = 0.6f;
        a->err_thr = 2e-6f;
    } else {
        a->mu = 0.5f;
        a->err_thr = 1.5e-6f;
    }
    a->mult = fs / 8000; /* num_bands == 1 for 8 / 16 kHz */
    ring_init(&a->near_fr, a->near_store, FRAME + PART, 1);
    ring_init(&a->out_fr, a->out_store, FRAME + PART, 1);
    ring_init(&a->far_buf, a->far_store, ORC_AEC_FAR_BLOCKS, 2 * PART1);
    ring_init(&a->far_buf_w, a->farw_store, ORC_AEC_FAR_BLOCKS, 2 * PART1);
    a->system_delay = 0;
    a->core_known_delay = 0;
    a->noise_ctr = 0;
    for (int i = 0; i < PART1; i++) {
        a->dMinPow[i] = 1.0e6f;
        a->sd[i] = 1;
        a->sx[i] = 1;
    }
    a->hNlFbMin = 1;
    a->hNlFbLocalMin = 1;
    a->hNlXdAvgMin = 1;
    a->overDrive = 2;
    a->overDriveSm = 2;
    a->seed = 777;
    a->nlp_mode = 2; /* kAecNlpAggressive, src/webrtc.c:224 */
}

/* aec_core.c:831-854 */
static void time_to_freq(float *t, float f[2][PART1], int window)
{
    if (window)
        for (int i = 0; i < PART; i++) {
            t[i] *= g_hanning[i];
            t[PART + i] *= g_hanning[PART - i];
        }
    orc_aec_rdft(1, t);
    f[1][0] = 0;
    f[1][PART] = 0;
    f[0][0] = t[0];
    f[0][PART] = t[1];
    for (int i = 1; i < PART; i++) {
        f[0][i] = t[2 * i];
        f[1][i] = t[2 * i + 1];
    }
}

/* aec_core.c:1709-1717 */
static int move_far_read(orc_aec *a, int n)
{
    int moved = ring_move_read(&a->far_buf_w, n);
    ring_move_read(&a->far_buf, n);
    a->system_delay -= moved * PART;
    return moved;
}

/* aec_core.c:1690-1707 */
static void buffer_far_partition(orc_aec *a, const float *farend)
{
    float fft[PART2], xf[2][PART1];
    if (ring_avail_write(&a->far_buf) < 1) move_far_read(a, 1);
    memcpy(fft, farend, sizeof(fft));
    time_to_freq(fft, xf, 0);
    ring_write(&a->far_buf, &xf[0][0], 1);
    memcpy(fft, farend, sizeof(fft));
    time_to_freq(fft, xf, 1);
    ring_write(&a->far_buf_w, &xf[0][0], 1);
}

/* echo_cancellation.c:278-339 */
int orc_aec_buffer_farend(orc_aec *a, const float *far, int n)
{
    if (n != 80 && n != 160) return -1;
    a->farend_started = 1;
    a->system_delay += n;
    ring_write(&a->far_pre, far, n);
    while (ring_avail_read(&a->far_pre) >= PART2) {
        float tmp[PART2];
        ring_read(&a->far_pre, tmp, PART2);
        buffer_far_partition(a, tmp);
        ring_move_read(&a->far_pre, -PART);
    }
    return 0;
}

static int cmp_float(const void *x, const void *y)
{
    float a = *(const float *)x, b = *(const float *)y;
    return (a > b) - (a < b);
}

/* aec_core.c:911-1141 */
static void nlp(orc_aec *a, float *output)
{
    float efw[2][PART1], dfw[2][PART1], xfw[2][PART1], fft[PART2], cohde[PART1], cohxd[PART1], hNl[PART1], pref[24];
    float hNlFb = 0, hNlFbLow = 0;
    const int prefSize = 24 / a->mult, minPref = 4 / a->mult, delayInterval = 10 * a->mult;
    static const float kTargetSupp[3] = {-6.9f, -11.5f, -18.4f}, kMinOverDrive[3] = {1.0f, 2.0f, 5.0f};
    const float g0 = 0.9f, g1 = a->mult == 1 ? 0.1f : 0.07f;
    const float gc0 = a->mult == 1 ? 0.9f : 0.93f; /* kNormalSmoothingCoefficients[mult-1] */
    a->delay_est_ctr++;
    if (a->delay_est_ctr == delayInterval) a->delay_est_ctr = 0;
    (void)g0;
    /* newest windowed far spectrum into the history (aec_core.c:946-948) */
    {
        float blk[2 * PART1];
        ring_read(&a->far_buf_w, blk, 1);
        memcpy(a->xfwBuf[0], blk, sizeof(blk));
    }
    /* SubbandCoherence aec_core.c:412-450 */
    if (a->delay_est_ctr == 0) {
        float best = 0;
        int d = 0;
        for (int p = 0; p < NPART; p++) {
            float en = 0;
            for (int j = 0; j < PART1; j++) en += a->wf[0][p][j] * a->wf[0][p][j] + a->wf[1][p][j] * a->wf[1][p][j];
            if (en > best) {
                best = en;
                d = p;
            }
        }
        a->delayIdx = d;
    }
    memcpy(xfw, a->xfwBuf[a->delayIdx], sizeof(xfw));
    for (int i = 0; i < PART; i++) {
        fft[i] = a->dBuf[i] * g_hanning[i];
        fft[PART + i] = a->dBuf[PART + i] * g_hanning[PART - i];
    }
    time_to_freq(fft, dfw, 0);
    for (int i = 0; i < PART; i++) {
        fft[i] = a->eBuf[i] * g_hanning[i];
        fft[PART + i] = a->eBuf[PART + i] * g_hanning[PART - i];
    }
    time_to_freq(fft, efw, 0);
    /* SmoothedPSD aec_core.c:333-386 */
    {
        float sdSum = 0, seSum = 0;
        for (int i = 0; i < PART1; i++) {
            a->sd[i] = gc0 * a->sd[i] + g1 * (dfw[0][i] * dfw[0][i] + dfw[1][i] * dfw[1][i]);
            a->se[i] = gc0 * a->se[i] + g1 * (efw[0][i] * efw[0][i] + efw[1][i] * efw[1][i]);
            float xx = xfw[0][i] * xfw[0][i] + xfw[1][i] * xfw[1][i];
            a->sx[i] = gc0 * a->sx[i] + g1 * (xx > 15.f ? xx : 15.f);
            a->sde[i][0] = gc0 * a->sde[i][0] + g1 * (dfw[0][i] * efw[0][i] + dfw[1][i] * efw[1][i]);
            a->sde[i][1] = gc0 * a->sde[i][1] + g1 * (dfw[0][i] * efw[1][i] - dfw[1][i] * efw[0][i]);
            a->sxd[i][0] = gc0 * a->sxd[i][0] + g1 * (dfw[0][i] * xfw[0][i] + dfw[1][i] * xfw[1][i]);
            a->sxd[i][1] = gc0 * a->sxd[i][1] + g1 * (dfw[0][i] * xfw[1][i] - dfw[1][i] * xfw[0][i]);
            sdSum += a->sd[i];
            seSum += a->se[i];
        }
        a->divergeState = (a->divergeState ? 1.05f : 1.0f) * seSum > sdSum;
        if (a->divergeState) memcpy(efw, dfw, sizeof(efw));
        if (seSum > (19.95f * sdSum)) memset(a->wf, 0, sizeof(a->wf));
    }
    for (int i = 0; i < PART1; i++) {
        cohde[i] = (a->sde[i][0] * a->sde[i][0] + a->sde[i][1] * a->sde[i][1]) / (a->sd[i] * a->se[i] + 1e-10f);
        cohxd[i] = (a->sxd[i][0] * a->sxd[i][0] + a->sxd[i][1] * a->sxd[i][1]) / (a->sx[i] * a->sd[i] + 1e-10f);
    }
    float hNlXdAvg = 0, hNlDeAvg = 0;
    for (int i = minPref; i < prefSize + minPref; i++) hNlXdAvg += cohxd[i];
    hNlXdAvg /= prefSize;
    hNlXdAvg = 1 - hNlXdAvg;
    for (int i = minPref; i < prefSize + minPref; i++) hNlDeAvg += cohde[i];
    hNlDeAvg /= prefSize;
    if (hNlXdAvg < 0.75f && hNlXdAvg < a->hNlXdAvgMin) a->hNlXdAvgMin = hNlXdAvg;
    if (hNlDeAvg > 0.98f && hNlXdAvg > 0.9f)
        a->stNearState = 1;
    else if (hNlDeAvg < 0.95f || hNlXdAvg < 0.8f)
        a->stNearState = 0;
    if (a->hNlXdAvgMin == 1) {
        a->echoState = 0;
        a->overDrive = kMinOverDrive[a->nlp_mode];
        if (a->stNearState == 1) {
            memcpy(hNl, cohde, sizeof(hNl));
            hNlFb = hNlDeAvg;
            hNlFbLow = hNlDeAvg;
        } else {
            for (int i = 0; i < PART1; i++) hNl[i] = 1 - cohxd[i];
            hNlFb = hNlXdAvg;
            hNlFbLow = hNlXdAvg;
        }
    } else {
        if (a->stNearState == 1) {
            a->echoState = 0;
            memcpy(hNl, cohde, sizeof(hNl));
            hNlFb = hNlDeAvg;
            hNlFbLow = hNlDeAvg;
        } else {
            a->echoState = 1;
            for (int i = 0; i < PART1; i++) hNl[i] = cohde[i] < 1 - cohxd[i] ? cohde[i] : 1 - cohxd[i];
            memcpy(pref, &hNl[minPref], sizeof(float) * prefSize);
            qsort(pref, prefSize, sizeof(float), cmp_float);
            hNlFb = pref[(int)floor(0.75f * (prefSize - 1))];
            hNlFbLow = pref[(int)floor(0.5f * (prefSize - 1))];
        }
    }
    if (hNlFbLow < 0.6f && hNlFbLow < a->hNlFbLocalMin) {
        a->hNlFbLocalMin = hNlFbLow;
        a->hNlFbMin = hNlFbLow;
        a->hNlNewMin = 1;
        a->hNlMinCtr = 0;
    }
    {
        float t = a->hNlFbLocalMin + 0.0008f / a->mult;
        a->hNlFbLocalMin = t < 1 ? t : 1;
        t = a->hNlXdAvgMin + 0.0006f / a->mult;
        a->hNlXdAvgMin = t < 1 ? t : 1;
    }
    if (a->hNlNewMin == 1) a->hNlMinCtr++;
    if (a->hNlMinCtr == 2) {
        a->hNlNewMin = 0;
        a->hNlMinCtr = 0;
        float od = kTargetSupp[a->nlp_mode] / ((float)log(a->hNlFbMin + 1e-10f) + 1e-10f);
        a->overDrive = od > kMinOverDrive[a->nlp_mode] ? od : kMinOverDrive[a->nlp_mode];
    }
    if (a->overDrive < a->overDriveSm)
        a->overDriveSm = 0.99f * a->overDriveSm + 0.01f * a->overDrive;
    else
        a->overDriveSm = 0.9f * a->overDriveSm + 0.1f * a->overDrive;
    /* OverdriveAndSuppress aec_core.c:272-293 */
    for (int i = 0; i < PART1; i++) {
        if (hNl[i] > hNlFb) hNl[i] = g_weight[i] * hNlFb + (1 - g_weight[i]) * hNl[i];
        hNl[i] = powf(hNl[i], a->overDriveSm * g_overdrive[i]);
        efw[0][i] *= hNl[i];
        efw[1][i] *= hNl[i];
        efw[1][i] *= -1;
    }
    /* ComfortNoise aec_core.c:462-547 (num_bands == 1) */
    {
        const float *noisePow = a->noise_is_init ? a->dInitMinPow : a->dMinPow;
        const float pi2 = 6.28318530717959f;
        float rnd[PART], u[PART1][2];
        for (int i = 0; i < PART; i++) {
            a->seed = (a->seed * 69069u + 1u) & 0x7FFFFFFFu;
            rnd[i] = ((float)(int16_t)(a->seed >> 16)) / 32768;
        }
        u[0][0] = 0;
        u[0][1] = 0;
        for (int i = 1; i < PART1; i++) {
            float tmp = pi2 * rnd[i - 1], noise = sqrtf(noisePow[i]);
            u[i][0] = noise * cosf(tmp);
            u[i][1] = -noise * sinf(tmp);
        }
        u[PART][1] = 0;
        for (int i = 0; i < PART1; i++) {
            float v = 1 - hNl[i] * hNl[i];
            float tmp = sqrtf(v > 0 ? v : 0);
            efw[0][i] += tmp * u[i][0];
            efw[1][i] += tmp * u[i][1];
        }
    }
    fft[0] = efw[0][0];
    fft[1] = efw[0][PART];
    for (int i = 1; i < PART; i++) {
        fft[2 * i] = efw[0][i];
        fft[2 * i + 1] = -efw[1][i];
    }
    orc_aec_rdft(-1, fft);
    {
        const float scale = 2.0f / PART2;
        for (int i = 0; i < PART; i++) {
            fft[i] *= scale;
            fft[i] = fft[i] * g_hanning[i] + a->outBuf[i];
            fft[PART + i] *= scale;
            a->outBuf[i] = fft[PART + i] * g_hanning[PART - i];
            output[i] = fft[i] > 32767.f ? 32767.f : (fft[i] < -32768.f ? -32768.f : fft[i]);
        }
    }
    memcpy(a->dBuf, a->dBuf + PART, sizeof(float) * PART);
    memcpy(a->eBuf, a->eBuf + PART, sizeof(float) * PART);
    memmove(a->xfwBuf[1], a->xfwBuf[0], sizeof(a->xfwBuf) - sizeof(a->xfwBuf[0]));
}

/* aec_core.c:1143-1351 */
static void process_block(orc_aec *a)
{
    float nearend[PART], fft[PART2], xf[2 * PART1], df[2][PART1], yf[2][PART1], ef[2][PART1], y[PART], e[PART], output[PART];
    const int noiseInitBlocks = 500 * a->mult;
    ring_read(&a->near_fr, nearend, PART);
    memcpy(a->dBuf + PART, nearend, sizeof(nearend));
    ring_read(&a->far_buf, xf, 1);
    memcpy(fft, a->dBuf, sizeof(fft));
    time_to_freq(fft, df, 0);
    for (int i = 0; i < PART1; i++) {
        float fs = (xf[i] * xf[i]) + (xf[PART1 + i] * xf[PART1 + i]);
        a->xPow[i] = 0.9f * a->xPow[i] + 0.1f * NPART * fs;
        float ns = df[0][i] * df[0][i] + df[1][i] * df[1][i];
        a->dPow[i] = 0.9f * a->dPow[i] + 0.1f * ns;
    }
    if (a->noise_ctr > 50) {
        for (int i = 0; i < PART1; i++) {
            if (a->dPow[i] < a->dMinPow[i])
                a->dMinPow[i] = (a->dPow[i] + 0.1f * (a->dMinPow[i] - a->dPow[i])) * 1.0002f;
            else
                a->dMinPow[i] *= 1.0002f;
        }
    }
    if (a->noise_ctr < noiseInitBlocks) {
        a->noise_ctr++;
        for (int i = 0; i < PART1; i++) {
            if (a->dMinPow[i] > a->dInitMinPow[i])
                a->dInitMinPow[i] = 0.999f * a->dInitMinPow[i] + 0.001f * a->dMinPow[i];
            else
                a->dInitMinPow[i] = a->dMinPow[i];
        }
        a->noise_is_init = 1;
    } else {
        a->noise_is_init = 0;
    }
    a->xf_pos--;
    if (a->xf_pos == -1) a->xf_pos = NPART - 1;
    memcpy(a->xf[0][a->xf_pos], xf, sizeof(float) * PART1);
    memcpy(a->xf[1][a->xf_pos], xf + PART1, sizeof(float) * PART1);
    memset(yf, 0, sizeof(yf));
    /* FilterFar aec_core.c:148-170 */
    for (int p = 0; p < NPART; p++) {
        int xp = p + a->xf_pos;
        if (xp >= NPART) xp -= NPART;
        for (int j = 0; j < PART1; j++) {
            yf[0][j] += a->xf[0][xp][j] * a->wf[0][p][j] - a->xf[1][xp][j] * a->wf[1][p][j];
            yf[1][j] += a->xf[0][xp][j] * a->wf[1][p][j] + a->xf[1][xp][j] * a->wf[0][p][j];
        }
    }
    fft[0] = yf[0][0];
    fft[1] = yf[0][PART];
    for (int i = 1; i < PART; i++) {
        fft[2 * i] = yf[0][i];
        fft[2 * i + 1] = yf[1][i];
    }
    orc_aec_rdft(-1, fft);
    {
        const float scale = 2.0f / PART2;
        for (int i = 0; i < PART; i++) y[i] = fft[PART + i] * scale;
    }
    for (int i = 0; i < PART; i++) e[i] = nearend[i] - y[i];
    memcpy(a->eBuf + PART, e, sizeof(e));
    memset(fft, 0, sizeof(float) * PART);
    memcpy(fft + PART, e, sizeof(e));
    orc_aec_rdft(1, fft);
    ef[1][0] = 0;
    ef[1][PART] = 0;
    ef[0][0] = fft[0];
    ef[0][PART] = fft[1];
    for (int i = 1; i < PART; i++) {
        ef[0][i] = fft[2 * i];
        ef[1][i] = fft[2 * i + 1];
    }
    /* ScaleErrorSignal aec_core.c:172-194 */
    for (int i = 0; i < PART1; i++) {
        ef[0][i] /= (a->xPow[i] + 1e-10f);
        ef[1][i] /= (a->xPow[i] + 1e-10f);
        float abs_ef = sqrtf(ef[0][i] * ef[0][i] + ef[1][i] * ef[1][i]);
        if (abs_ef > a->err_thr) {
            abs_ef = a->err_thr / (abs_ef + 1e-10f);
            ef[0][i] *= abs_ef;
            ef[1][i] *= abs_ef;
        }
        ef[0][i] *= a->mu;
        ef[1][i] *= a->mu;
    }
    /* FilterAdaptation aec_core.c:222-270 */
    for (int p = 0; p < NPART; p++) {
        int xp = p + a->xf_pos;
        if (xp >= NPART) xp -= NPART;
        for (int j = 0; j < PART; j++) {
            float xr = a->xf[0][xp][j], xi = -a->xf[1][xp][j];
            fft[2 * j] = xr * ef[0][j] - xi * ef[1][j];
            fft[2 * j + 1] = xr * ef[1][j] + xi * ef[0][j];
        }
        fft[1] = a->xf[0][xp][PART] * ef[0][PART] - (-a->xf[1][xp][PART]) * ef[1][PART];
        orc_aec_rdft(-1, fft);
        memset(fft + PART, 0, sizeof(float) * PART);
        {
            const float scale = 2.0f / PART2;
            for (int j = 0; j < PART; j++) fft[j] *= scale;
        }
        orc_aec_rdft(1, fft);
        a->wf[0][p][0] += fft[0];
        a->wf[0][p][PART] += fft[1];
        for (int j = 1; j < PART; j++) {
            a->wf[0][p][j] += fft[2 * j];
            a->wf[1][p][j] += fft[2 * j + 1];
        }
    }
    nlp(a, output);
    ring_write(&a->out_fr, output, PART);
}

/* aec_core.c:1719-1850 (num_bands = 1, reported delays enabled) */
static void process_frames(orc_aec *a, const float *nearend, int n, int known_delay, float *out)
{
    for (int j = 0; j < n; j += FRAME) {
        ring_write(&a->near_fr, nearend + j, FRAME);
        if (a->system_delay < FRAME) move_far_read(a, -(a->mult + 1));
        {
            int move = (a->core_known_delay - known_delay - 32) / PART;
            int moved = ring_move_read(&a->far_buf, move);
            ring_move_read(&a->far_buf_w, move);
            a->core_known_delay -= moved * PART;
        }
        while (ring_avail_read(&a->near_fr) >= PART) process_block(a);
        a->system_delay -= FRAME;
        int avail = ring_avail_read(&a->out_fr);
        if (avail < FRAME) ring_move_read(&a->out_fr, avail - FRAME);
        ring_read(&a->out_fr, out + j, FRAME);
    }
}

/* echo_cancellation.c:821-872 */
static void est_buf_delay(orc_aec *a)
{
    int nSamp = a->msInSndCardBuf * 8 * a->rate_factor;
    int cur = nSamp - a->system_delay;
    cur += FRAME * a->rate_factor;
    if (cur < PART) cur += move_far_read(a, 1) * PART;
    a->filtDelay = a->filtDelay < 0 ? 0 : a->filtDelay;
    {
        short f = (short)(0.8 * a->filtDelay + 0.2 * cur);
        a->filtDelay = f > 0 ? f : 0;
    }
    int diff = a->filtDelay - a->knownDelay;
    if (diff > 224) {
        if (a->lastDelayDiff < 96)
            a->timeForDelayChange = 0;
        else
            a->timeForDelayChange++;
    } else if (diff < 96 && a->knownDelay > 0) {
        if (a->lastDelayDiff > 224)
            a->timeForDelayChange = 0;
        else
            a->timeForDelayChange++;
    } else {
        a->timeForDelayChange = 0;
    }
    a->lastDelayDiff = (short)diff;
    if (a->timeForDelayChange > 25) {
        int k = (int)a->filtDelay - 160;
        a->knownDelay = k > 0 ? k : 0;
    }
}

/* echo_cancellation.c:341-409 + ProcessNormal :599-747.  in/out may alias. */
int orc_aec_process(orc_aec *a, const float *nearend, float *out, int n, int ms_in_snd_card_buf)
{
    int ret = 0;
    if (n != 80 && n != 160) return -1;
    short ms = (short)ms_in_snd_card_buf;
    if (ms < 0) {
        ms = 0;
        ret = -1;
    } else if (ms > 500) {
        ret = -1;
    }
    ms = ms > 500 ? 500 : ms;
    ms = (short)(ms + 10);
    a->msInSndCardBuf = ms;
    short nBlocks10ms = (short)(n / (FRAME * a->rate_factor));
    if (a->startup_phase) {
        if (nearend != out) memcpy(out, nearend, sizeof(float) * n);
        if (a->checkBuffSize) {
            a->checkBufSizeCtr++;
            if (a->counter == 0) {
                a->firstVal = a->msInSndCardBuf;
                a->sum = 0;
            }
            double lim = 0.2 * a->msInSndCardBuf;
            if (abs(a->firstVal - a->msInSndCardBuf) < (lim > 8 ? lim : 8)) {
                a->sum += a->msInSndCardBuf;
                a->counter++;
            } else {
                a->counter = 0;
            }
            if (a->counter * nBlocks10ms >= 6) {
                int v = (3 * a->sum * a->rate_factor * 8) / (4 * a->counter * PART);
                a->bufSizeStart = v < 62 ? v : 62;
                a->checkBuffSize = 0;
            }
            if (a->checkBufSizeCtr * nBlocks10ms > 50) {
                int v = (a->msInSndCardBuf * a->rate_factor * 3) / 40;
                a->bufSizeStart = v < 62 ? v : 62;
                a->checkBuffSize = 0;
            }
        }
        if (!a->checkBuffSize) {
            int overhead = a->system_delay / PART - a->bufSizeStart;
            if (overhead == 0) {
                a->startup_phase = 0;
            } else if (overhead > 0) {
                move_far_read(a, overhead);
                a->startup_phase = 0;
            }
        }
    } else {
        est_buf_delay(a);
        process_frames(a, nearend, n, a->knownDelay, out);
    }
    return ret;
}

/* echo_cancellation.c:121-275 (Create + Init + set_config) */
void orc_aec_core_setup(orc_aec *a, int fs)
{
    if (!g_tables_ready) build_tables();
    memset(a, 0, sizeof(*a));
    a->fs = fs;
    core_init(a, fs);
    ring_init(&a->far_pre, a->farpre_store, PART2 + 320, 1);
    ring_move_read(&a->far_pre, -PART);
    a->rate_factor = fs / 8000;
    a->checkBuffSize = 1;
    a->startup_phase = 1;
    a->filtDelay = -1;
}

/* ------------------------------------------------------------------ wmix wrapper, src/webrtc.c:217-505 */
orc_aec *orc_aec_init(int chn, int freq, int interval_ms)
{
    if (freq > 16000 || freq % 8000 != 0) return NULL;
    orc_aec *a = calloc(1, sizeof(*a));
    orc_aec_core_setup(a, freq);
    a->chn = chn;
    a->pkg = freq / 1000 * ((freq <= 8000 && interval_ms % 20 == 0) ? 20 : 10);
    return a;
}

int orc_aec_process2(orc_aec *a, const int16_t *far, const int16_t *nearp, int16_t *out, int frame_num, int delay_ms)
{
    int total = frame_num * a->chn, step = a->pkg * a->chn;
    float f[160], in[160], o[160];
    for (int done = 0; done < total; done += step) {
        for (int i = 0; i < a->pkg; i++) {
            f[i] = (float)far[i * a->chn];
            in[i] = (float)nearp[i * a->chn];
        }
        far += step;
        nearp += step;
        int r = orc_aec_buffer_farend(a, f, a->pkg);
        if (r != 0) return r;
        r = orc_aec_process(a, in, o, a->pkg, delay_ms);
        if (r != 0) return r;
        for (int i = 0; i < a->pkg; i++)
            for (int c = 0; c < a->chn; c++) *out++ = (int16_t)o[i];
    }
    return 0;
}

void orc_aec_release(orc_aec *a) { free(a); }

int orc_run_aec(int chn, int freq, int interval_ms, const int16_t *far, const int16_t *nearp, int16_t *out, int frames_per_call,
                int n_calls, int delay_ms)
{
    orc_aec *a = orc_aec_init(chn, freq, interval_ms);
    if (!a) return -100;
    size_t step = (size_t)frames_per_call * chn;
    int rc = 0;
    for (int i = 0; i < n_calls && rc == 0; i++)
        rc = orc_aec_process2(a, far + i * step, nearp + i * step, out + i * step, frames_per_call, delay_ms);
    orc_aec_release(a);
    return rc;
}

/* the same for a handle whose comfort-noise generator stands at `seed` when the run starts (aec->seed, aec_core.c:1670, is 777 for a
 * new handle and moves by 64 draws per block): what a handle that has already lived for many blocks draws from here on.  Test
 * infrastructure for the kernels' noise table far from its start; nothing else of the handle's state is aged. */
int orc_run_aec_seeded(int chn, int freq, int interval_ms, const int16_t *far, const int16_t *nearp, int16_t *out, int frames_per_call,
                       int n_calls, int delay_ms, uint32_t seed)
{
    orc_aec *a = orc_aec_init(chn, freq, interval_ms);
    if (!a) return -100;
    a->seed = seed & 0x7FFFFFFFu;
    size_t step = (size_t)frames_per_call * chn;
    int rc = 0;
    for (int i = 0; i < n_calls && rc == 0; i++)
        rc = orc_aec_process2(a, far + i * step, nearp + i * step, out + i * step, frames_per_call, delay_ms);
    orc_aec_release(a);
    return rc;
}

/* the same with the delay the caller reports changing from call to call (aec_process2's delayms, src/webrtc.c:410-483) */
int orc_run_aec_delays(int chn, int freq, int interval_ms, const int16_t *far, const int16_t *nearp, int16_t *out, int frames_per_call,
                       int n_calls, const int32_t *delay_ms)
{
    orc_aec *a = orc_aec_init(chn, freq, interval_ms);
    if (!a) return -100;
    size_t step = (size_t)frames_per_call * chn;
    int rc = 0;
    for (int i = 0; i < n_calls && rc == 0; i++)
        rc = orc_aec_process2(a, far + i * step, nearp + i * step, out + i * step, frames_per_call, delay_ms[i]);
    orc_aec_release(a);
    return rc;
}


/* State probe (SURVEY 8c "state probes for debugging"): the decisions of NonLinearProcessing and the scalars they are
 * taken on (aec_core.c:911-1141), so that two runs can be compared decision by decision.
 * ints: stNearState, echoState, divergeState, delayIdx, hNlNewMin, hNlMinCtr, noise_ctr, system_delay
 * floats: hNlFbMin, hNlFbLocalMin, hNlXdAvgMin, overDrive, overDriveSm, sum(sd), sum(se) */
void orc_aec_probe(const orc_aec *a, int32_t *ints8, float *floats7)
{
    ints8[0] = a->stNearState;
    ints8[1] = a->echoState;
    ints8[2] = a->divergeState;
    ints8[3] = a->delayIdx;
    ints8[4] = a->hNlNewMin;
    ints8[5] = a->hNlMinCtr;
    ints8[6] = a->noise_ctr;
    ints8[7] = a->system_delay;
    floats7[0] = a->hNlFbMin;
    floats7[1] = a->hNlFbLocalMin;
    floats7[2] = a->hNlXdAvgMin;
    floats7[3] = a->overDrive;
    floats7[4] = a->overDriveSm;
    float sd = 0.f, se = 0.f;
    for (int i = 0; i < 65; i++) {
        sd += a->sd[i];
        se += a->se[i];
    }
    floats7[5] = sd;
    floats7[6] = se;
}
