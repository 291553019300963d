/* oracle/orc_ns.c -- TEST INFRASTRUCTURE ONLY (CPU checker; never on the product path).
 *
 * Restatement of the reference's float noise suppressor as wmix drives it:
 *   ns_init / ns_process / ns_release        src/webrtc.c:560-661
 *   WebRtcNs_InitCore / set_policy_core      W:modules/audio_processing/ns/ns_core.c:74-214,1013-1041
 *   WebRtcNs_AnalyzeCore                     W:...ns_core.c:1043-1181
 *   WebRtcNs_ProcessCore                     W:...ns_core.c:1183-1415
 * (W: = inside pkg/webrtc_cut.tar.gz, webrtc_cut/webrtc/...).  Pinned bit-exact
 * against oracle/_ref (the real sources) in tests/test_ns_oracle.py and against
 * tests/golden/ns_*.npz.  Compile with -ffp-contract=off; double-precision libm
 * calls are kept double exactly where the reference has them.
 *
 * Organisation differs from the reference on purpose: one routine per dataflow
 * phase, each written as "per-bin map" + "ordered reduction", which is also how
 * the HIP kernel is organised; every float expression keeps the reference's
 * operand order and rounding points.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "orc_fft.h"
#include "orc_ns.h"

#define STARTUP_SHORT 50   /* defines.h:22 END_STARTUP_SHORT */
#define STARTUP_LONG 200   /* defines.h:21 END_STARTUP_LONG */
#define HIST_BINS 1000     /* defines.h:45 HIST_PAR_EST */
#define UPDATE_WINDOW 500  /* ns_core.c:188 modelUpdatePars[1] */
#define START_BAND 5       /* ns_core.c:1045 kStartBand */

static float sat16(float v) /* WEBRTC_SPL_SAT(32767, v, -32768) */
{
    return v > 32767.f ? 32767.f : (v < -32768.f ? -32768.f : v);
}

/* windows_private.h:64,94 kBlocks80w128 / kBlocks160w256: a sine ramp of 48 (96)
 * samples, 32 (64) ones, mirrored ramp; the table entries are the 8-decimal literals,
 * i.e. float(round(sin * 1e8) / 1e8)  (checked against the header in the container). */
void orc_ns_window(int ana_len, float *w)
{
    int ramp = ana_len == 128 ? 48 : 96;
    const double half_pi = 1.5707963267948966;
    for (int i = 0; i < ana_len; i++) {
        double v;
        if (i < ramp)
            v = sin(half_pi * i / ramp);
        else if (i < ana_len - ramp)
            v = 1.0;
        else
            v = sin(half_pi * (ana_len - i) / ramp);
        w[i] = (float)(floor(v * 1e8 + 0.5) / 1e8);
    }
}

/* ns_core.c:74-214 + set_policy_core(mode 2) :1013-1041 (src/webrtc.c:532,577) */
void orc_ns_core_init(orc_ns_core *s, int fs)
{
    memset(s, 0, sizeof(*s));
    s->fs = fs;
    if (fs == 8000) {
        s->block_len = 80;
        s->ana_len = 128;
    } else {
        s->block_len = 160;
        s->ana_len = 256;
    }
    s->magn_len = s->ana_len / 2 + 1;
    orc_ns_window(s->ana_len, s->window);
    orc_fft_init(&s->fft, s->ana_len);
    for (int i = 0; i < 3 * ORC_NS_MAXBINS; i++) {
        s->lquantile[i] = 8.f;
        s->density[i] = 0.3f;
    }
    for (int i = 0; i < 3; i++) s->counter[i] = (int)floor((float)(STARTUP_LONG * (i + 1)) / (float)3);
    for (int i = 0; i < ORC_NS_MAXBINS; i++) {
        s->smooth[i] = 1.f;
        s->log_lrt_avg[i] = 0.5f;
    }
    s->prior_speech_prob = 0.5f;
    s->feat_flatness = 0.5f;
    s->feat_lrt = 0.5f;
    s->feat_diff = 0.5f;
    s->block_ind = -1;
    s->thr_lrt = 0.5f;
    s->thr_flat = 0.5f;
    s->thr_diff = 0.5f;
    s->w_lrt = 1.f;
    s->update_flag = 2;
    s->window_countdown = UPDATE_WINDOW;
    /* policy 2 */
    s->overdrive = 1.1f;
    s->denoise_bound = 0.125f;
    s->gainmap = 1;
}

static void shift_in(float *buf, int buf_len, const float *frame, int n)
{
    memmove(buf, buf + n, sizeof(float) * (buf_len - n));
    if (frame)
        memcpy(buf + buf_len - n, frame, sizeof(float) * n);
    else
        memset(buf + buf_len - n, 0, sizeof(float) * n);
}

/* ns_core.c:1135 Windowing + :1119 Energy */
static float window_and_energy(const orc_ns_core *s, const float *buf, float *out)
{
    float e = 0.f;
    for (int i = 0; i < s->ana_len; i++) out[i] = s->window[i] * buf[i];
    for (int i = 0; i < s->ana_len; i++) e += out[i] * out[i];
    return e;
}

/* ns_core.c:886-911 */
static void spectrum(const orc_ns_core *s, float *t, float *re, float *im, float *magn)
{
    int M = s->magn_len;
    orc_rdft_forward(&s->fft, t);
    im[0] = 0;
    re[0] = t[0];
    magn[0] = (float)(fabs(re[0]) + 1.f);
    im[M - 1] = 0;
    re[M - 1] = t[1];
    magn[M - 1] = (float)(fabs(re[M - 1]) + 1.f);
    for (int i = 1; i < M - 1; i++) {
        re[i] = t[2 * i];
        im[i] = t[2 * i + 1];
        magn[i] = sqrtf(re[i] * re[i] + im[i] * im[i]) + 1.f;
    }
}

/* ns_core.c:217-285 */
static void quantile_noise(orc_ns_core *s, const float *lmagn, float *noise)
{
    int M = s->magn_len, off = 0;
    if (s->updates < STARTUP_LONG) s->updates++;
    for (int k = 0; k < 3; k++) {
        off = k * M;
        float cnt1 = (float)(s->counter[k] + 1), cnt = (float)s->counter[k];
        for (int i = 0; i < M; i++) {
            float dens = s->density[off + i], delta;
            if (dens > 1.0)
                delta = 40.f * 1.f / dens;
            else
                delta = 40.f;
            if (lmagn[i] > s->lquantile[off + i])
                s->lquantile[off + i] += 0.25f * delta / cnt1;
            else
                s->lquantile[off + i] -= (1.f - 0.25f) * delta / cnt1;
            if (fabs(lmagn[i] - s->lquantile[off + i]) < 0.01f)
                s->density[off + i] = (cnt * s->density[off + i] + 1.f / (2.f * 0.01f)) / cnt1;
        }
        if (s->counter[k] >= STARTUP_LONG) {
            s->counter[k] = 0;
            if (s->updates >= STARTUP_LONG)
                for (int i = 0; i < M; i++) s->quantile[i] = (float)exp(s->lquantile[off + i]);
        }
        s->counter[k]++;
    }
    if (s->updates < STARTUP_LONG)
        for (int i = 0; i < M; i++) s->quantile[i] = (float)exp(s->lquantile[off + i]);
    for (int i = 0; i < M; i++) noise[i] = s->quantile[i];
}

/* ns_core.c:523-556 (lmagn[i] == (float)log(magn[i]) is shared with quantile_noise) */
static void spectral_flatness(orc_ns_core *s, const float *magn, const float *lmagn)
{
    int M = s->magn_len;
    float num = 0.0, den = s->sum_magn;
    den -= magn[0];
    for (int i = 1; i < M; i++) {
        if (magn[i] > 0.0) {
            num += lmagn[i];
        } else {
            s->feat_flatness -= 0.3f * s->feat_flatness;
            return;
        }
    }
    den = den / M;
    num = num / M;
    float tmp = (float)exp(num) / den;
    s->feat_flatness += 0.3f * (tmp - s->feat_flatness);
}

/* ns_core.c:595-634 */
static void spectral_difference(orc_ns_core *s, const float *magn)
{
    int M = s->magn_len;
    float avg_pause = 0.0, avg_magn = s->sum_magn, cov = 0.0, var_pause = 0.0, var_magn = 0.0;
    for (int i = 0; i < M; i++) avg_pause += s->magn_avg_pause[i];
    avg_pause = avg_pause / ((float)M);
    avg_magn = avg_magn / ((float)M);
    for (int i = 0; i < M; i++) {
        cov += (magn[i] - avg_magn) * (s->magn_avg_pause[i] - avg_pause);
        var_pause += (s->magn_avg_pause[i] - avg_pause) * (s->magn_avg_pause[i] - avg_pause);
        var_magn += (magn[i] - avg_magn) * (magn[i] - avg_magn);
    }
    cov = cov / ((float)M);
    var_pause = var_pause / ((float)M);
    var_magn = var_magn / ((float)M);
    s->feat_energy_acc += s->signal_energy;
    float d = var_magn - (cov * cov) / (var_pause + 0.0001f);
    d = (float)(d / (s->feat_energy_norm + 0.0001f));
    s->feat_diff += 0.3f * (d - s->feat_diff);
}

/* ns_core.c:293-518: histogram update (flag 0) */
static void hist_add(orc_ns_core *s)
{
    const float bin_lrt = 0.1f, bin_flat = 0.05f, bin_diff = 0.1f;
    if ((s->feat_lrt < HIST_BINS * bin_lrt) && (s->feat_lrt >= 0.0)) s->hist_lrt[(int)(s->feat_lrt / bin_lrt)]++;
    if ((s->feat_flatness < HIST_BINS * bin_flat) && (s->feat_flatness >= 0.0))
        s->hist_flat[(int)(s->feat_flatness / bin_flat)]++;
    if ((s->feat_diff < HIST_BINS * bin_diff) && (s->feat_diff >= 0.0)) s->hist_diff[(int)(s->feat_diff / bin_diff)]++;
}

static void two_peaks(const int *hist, float bin, int *w1, int *w2, float *p1, float *p2)
{
    int max1 = 0, max2 = 0;
    *w1 = *w2 = 0;
    *p1 = *p2 = 0.0;
    for (int i = 0; i < HIST_BINS; i++) {
        float mid = ((float)i + 0.5f) * bin;
        if (hist[i] > max1) {
            max2 = max1;
            *w2 = *w1;
            *p2 = *p1;
            max1 = hist[i];
            *w1 = hist[i];
            *p1 = mid;
        } else if (hist[i] > max2) {
            max2 = hist[i];
            *w2 = hist[i];
            *p2 = mid;
        }
    }
}

/* ns_core.c:336-517: threshold extraction (flag 1) */
static void hist_extract(orc_ns_core *s)
{
    const float bin_lrt = 0.1f, bin_flat = 0.05f, bin_diff = 0.1f;
    const float thres_fluct = 0.05f, max_lrt = 1.f, min_lrt = 0.2f, max_flat = 0.95f, min_flat = 0.1f;
    const float max_diff = 1.f, min_diff = 0.16f, f1 = 1.2f, f2 = 0.9f;
    const float lim_space_flat = 2 * bin_flat, lim_space_diff = 2 * bin_diff, lim_w = 0.5f, thres_pos_flat = 0.6f;
    const int thres_weight = (int)(0.3 * UPDATE_WINDOW);
    float avg = 0.0, avg_compl = 0.0, avg_sq = 0.0;
    int num = 0;
    for (int i = 0; i < HIST_BINS; i++) {
        float mid = ((float)i + 0.5f) * bin_lrt;
        if (mid <= 1.f) {
            avg += s->hist_lrt[i] * mid;
            num += s->hist_lrt[i];
        }
        avg_sq += s->hist_lrt[i] * mid * mid;
        avg_compl += s->hist_lrt[i] * mid;
    }
    if (num > 0) avg = avg / ((float)num);
    avg_compl = avg_compl / ((float)UPDATE_WINDOW);
    avg_sq = avg_sq / ((float)UPDATE_WINDOW);
    float fluct = avg_sq - avg * avg_compl;
    if (fluct < thres_fluct) {
        s->thr_lrt = max_lrt;
    } else {
        s->thr_lrt = f1 * avg;
        if (s->thr_lrt < min_lrt) s->thr_lrt = min_lrt;
        if (s->thr_lrt > max_lrt) s->thr_lrt = max_lrt;
    }
    int w1f, w2f, w1d, w2d;
    float p1f, p2f, p1d, p2d;
    two_peaks(s->hist_flat, bin_flat, &w1f, &w2f, &p1f, &p2f);
    two_peaks(s->hist_diff, bin_diff, &w1d, &w2d, &p1d, &p2d);
    int use_flat = 1, use_diff = 1;
    if ((fabs(p2f - p1f) < lim_space_flat) && (w2f > lim_w * w1f)) {
        w1f += w2f;
        p1f = 0.5f * (p1f + p2f);
    }
    if (w1f < thres_weight || p1f < thres_pos_flat) use_flat = 0;
    if (use_flat == 1) {
        s->thr_flat = f2 * p1f;
        if (s->thr_flat < min_flat) s->thr_flat = min_flat;
        if (s->thr_flat > max_flat) s->thr_flat = max_flat;
    }
    if ((fabs(p2d - p1d) < lim_space_diff) && (w2d > lim_w * w1d)) {
        w1d += w2d;
        p1d = 0.5f * (p1d + p2d);
    }
    s->thr_diff = f1 * p1d;
    if (w1d < thres_weight) use_diff = 0;
    if (s->thr_diff < min_diff) s->thr_diff = min_diff;
    if (s->thr_diff > max_diff) s->thr_diff = max_diff;
    if (fluct < thres_fluct) use_diff = 0;
    float fsum = (float)(1 + use_flat + use_diff);
    s->w_lrt = 1.f / fsum;
    s->w_flat = ((float)use_flat) / fsum;
    s->w_diff = ((float)use_diff) / fsum;
    memset(s->hist_lrt, 0, sizeof(s->hist_lrt));
    memset(s->hist_flat, 0, sizeof(s->hist_flat));
    memset(s->hist_diff, 0, sizeof(s->hist_diff));
}

/* ns_core.c:755-791 */
static void feature_update(orc_ns_core *s, const float *magn, const float *lmagn, int flag)
{
    spectral_flatness(s, magn, lmagn);
    spectral_difference(s, magn);
    if (flag >= 1) {
        s->window_countdown--;
        if (s->window_countdown > 0) hist_add(s);
        if (s->window_countdown == 0) {
            hist_extract(s);
            s->window_countdown = UPDATE_WINDOW;
            if (flag == 1) {
                s->update_flag = 0;
            } else {
                s->feat_energy_acc = s->feat_energy_acc / ((float)UPDATE_WINDOW);
                s->feat_energy_norm = 0.5f * (s->feat_energy_acc + s->feat_energy_norm);
                s->feat_energy_acc = 0.f;
            }
        }
    }
}

/* ns_core.c:642-749 */
static void speech_prob(orc_ns_core *s, const float *snr_prior, const float *snr_post)
{
    int M = s->magn_len;
    float ksum = 0.0;
    for (int i = 0; i < M; i++) {
        float t1 = 1.f + 2.f * snr_prior[i];
        float t2 = 2.f * snr_prior[i] / (t1 + 0.0001f);
        float bessel = (snr_post[i] + 1.f) * t2;
        s->log_lrt_avg[i] += 0.5f * (bessel - (float)log(t1) - s->log_lrt_avg[i]);
        ksum += s->log_lrt_avg[i];
    }
    ksum = (float)ksum / (M);
    s->feat_lrt = ksum;
    float width = 4.0f;
    if (ksum < s->thr_lrt) width = 2.f * 4.0f;
    float ind0 = 0.5f * ((float)tanh(width * (ksum - s->thr_lrt)) + 1.f);
    float t = s->feat_flatness;
    width = 4.0f;
    if (t > s->thr_flat) width = 2.f * 4.0f; /* sgnMap == 1 always (priorModelPars[2] = 1) */
    float ind1 = 0.5f * ((float)tanh((float)1 * width * (s->thr_flat - t)) + 1.f);
    t = s->feat_diff;
    width = 4.0f;
    if (t < s->thr_diff) width = 2.f * 4.0f;
    float ind2 = 0.5f * ((float)tanh(width * (t - s->thr_diff)) + 1.f);
    float ind = s->w_lrt * ind0 + s->w_flat * ind1 + s->w_diff * ind2;
    s->prior_speech_prob += 0.1f * (ind - s->prior_speech_prob);
    if (s->prior_speech_prob > 1.f) s->prior_speech_prob = 1.f;
    if (s->prior_speech_prob < 0.01f) s->prior_speech_prob = 0.01f;
    float gain_prior = (1.f - s->prior_speech_prob) / (s->prior_speech_prob + 0.0001f);
    for (int i = 0; i < M; i++) {
        float inv = (float)exp(-s->log_lrt_avg[i]);
        inv = (float)gain_prior * inv;
        s->speech_prob[i] = 1.f / (1.f + inv);
    }
}

/* ns_core.c:800-846.  gamma of bin i-1 carries into bin i's first estimate. */
static void update_noise(orc_ns_core *s, const float *magn, float *noise)
{
    int M = s->magn_len;
    float gamma = 0.9f;
    for (int i = 0; i < M; i++) {
        float ps = s->speech_prob[i], pn = 1.f - ps;
        float tmp = gamma * s->noise_prev[i] + (1.f - gamma) * (pn * magn[i] + ps * s->noise_prev[i]);
        float gamma_old = gamma;
        gamma = 0.9f;
        if (ps > 0.2f) gamma = 0.99f;
        if (ps < 0.2f) s->magn_avg_pause[i] += 0.05f * (magn[i] - s->magn_avg_pause[i]);
        if (gamma == gamma_old) {
            noise[i] = tmp;
        } else {
            noise[i] = gamma * s->noise_prev[i] + (1.f - gamma) * (pn * magn[i] + ps * s->noise_prev[i]);
            if (tmp < noise[i]) noise[i] = tmp;
        }
    }
}

/* ns_core.c:1043-1181 */
void orc_ns_analyze(orc_ns_core *s, const float *frame)
{
    int M = s->magn_len;
    float win[ORC_NS_MAXLEN], re[ORC_NS_MAXBINS], im[ORC_NS_MAXBINS], magn[ORC_NS_MAXBINS], lmagn[ORC_NS_MAXBINS];
    float noise[ORC_NS_MAXBINS], snr_prior[ORC_NS_MAXBINS], snr_post[ORC_NS_MAXBINS];
    int flag = s->update_flag;
    shift_in(s->analyze_buf, s->ana_len, frame, s->block_len);
    float energy = window_and_energy(s, s->analyze_buf, win);
    if (energy == 0.0) return;
    s->block_ind++;
    spectrum(s, win, re, im, magn);
    float signal_energy = 0.f, sum_magn = 0.f;
    float sum_log_i = 0.0, sum_log_i_sq = 0.0, sum_log_magn = 0.0, sum_log_i_log_magn = 0.0;
    for (int i = 0; i < M; i++) {
        lmagn[i] = (float)log(magn[i]);
        signal_energy += re[i] * re[i] + im[i] * im[i];
        sum_magn += magn[i];
        if (s->block_ind < STARTUP_SHORT && i >= START_BAND) {
            float li = log((float)i);
            sum_log_i += li;
            sum_log_i_sq += li * li;
            sum_log_magn += lmagn[i];
            sum_log_i_log_magn += li * lmagn[i];
        }
    }
    signal_energy = signal_energy / ((float)M);
    s->signal_energy = signal_energy;
    s->sum_magn = sum_magn;
    quantile_noise(s, lmagn, noise);
    if (s->block_ind < STARTUP_SHORT) {
        float pnum = 0.0, pexp = 0.0;
        s->white_level += sum_magn / ((float)M) * s->overdrive;
        float t1 = sum_log_i_sq * ((float)(M - START_BAND));
        t1 -= (sum_log_i * sum_log_i);
        float t2 = (sum_log_i_sq * sum_log_magn - sum_log_i * sum_log_i_log_magn);
        float t3 = t2 / t1;
        if (t3 < 0.f) t3 = 0.f;
        s->pink_num += t3;
        t2 = (sum_log_i * sum_log_magn);
        t2 -= ((float)(M - START_BAND)) * sum_log_i_log_magn;
        t3 = t2 / t1;
        if (t3 < 0.f) t3 = 0.f;
        if (t3 > 1.f) t3 = 1.f;
        s->pink_exp += t3;
        if (s->pink_exp > 0.f) {
            pnum = exp(s->pink_num / (float)(s->block_ind + 1));
            pnum *= (float)(s->block_ind + 1);
            pexp = s->pink_exp / (float)(s->block_ind + 1);
        }
        for (int i = 0; i < M; i++) {
            if (s->pink_exp == 0.f) {
                s->parametric_noise[i] = s->white_level;
            } else {
                float band = (float)(i < START_BAND ? START_BAND : i);
                s->parametric_noise[i] = pnum / pow(band, pexp);
            }
            noise[i] *= (s->block_ind);
            t2 = s->parametric_noise[i] * (STARTUP_SHORT - s->block_ind);
            noise[i] += (t2 / (float)(s->block_ind + 1));
            noise[i] /= STARTUP_SHORT;
        }
    }
    if (s->block_ind < STARTUP_LONG) {
        s->feat_energy_norm *= s->block_ind;
        s->feat_energy_norm += signal_energy;
        s->feat_energy_norm /= (s->block_ind + 1);
    }
    /* ComputeSnr ns_core.c:566-588 */
    for (int i = 0; i < M; i++) {
        float prev = s->magn_prev_analyze[i] / (s->noise_prev[i] + 0.0001f) * s->smooth[i];
        snr_post[i] = 0.f;
        if (magn[i] > noise[i]) snr_post[i] = magn[i] / (noise[i] + 0.0001f) - 1.f;
        snr_prior[i] = 0.98f * prev + (1.f - 0.98f) * snr_post[i];
    }
    feature_update(s, magn, lmagn, flag);
    speech_prob(s, snr_prior, snr_post);
    update_noise(s, magn, noise);
    memcpy(s->noise, noise, sizeof(float) * M);
    memcpy(s->magn_prev_analyze, magn, sizeof(float) * M);
}

/* ns_core.c:1183-1415.  in/out: [num_bands][block_len] */
void orc_ns_process(orc_ns_core *s, const float *const *in, int num_bands, float *const *out)
{
    int M = s->magn_len, L = s->ana_len, B = s->block_len;
    float win[ORC_NS_MAXLEN], re[ORC_NS_MAXBINS], im[ORC_NS_MAXBINS], magn[ORC_NS_MAXBINS];
    float filt[ORC_NS_MAXBINS], fout[160];
    int hb = num_bands > 1, delta_hb = hb ? M / 4 : 1;
    shift_in(s->data_buf, L, in[0], B);
    for (int b = 1; b < num_bands; b++) shift_in(s->data_buf_hb[b - 1], L, in[b], B);
    float energy1 = window_and_energy(s, s->data_buf, win);
    if (energy1 == 0.0) {
        for (int i = 0; i < B; i++) fout[i] = s->synt_buf[i];
        shift_in(s->synt_buf, L, NULL, B);
        for (int i = 0; i < B; i++) out[0][i] = sat16(fout[i]);
        for (int b = 1; b < num_bands; b++)
            for (int j = 0; j < B; j++) out[b][j] = sat16(s->data_buf_hb[b - 1][j]);
        return;
    }
    spectrum(s, win, re, im, magn);
    if (s->block_ind < STARTUP_SHORT)
        for (int i = 0; i < M; i++) s->init_magn_est[i] += magn[i];
    for (int i = 0; i < M; i++) {
        /* ComputeDdBasedWienerFilter ns_core.c:985-1007 */
        float prev = s->magn_prev_process[i] / (s->noise_prev[i] + 0.0001f) * s->smooth[i];
        float cur = 0.f;
        if (magn[i] > s->noise[i]) cur = magn[i] / (s->noise[i] + 0.0001f) - 1.f;
        float snr = 0.98f * prev + (1.f - 0.98f) * cur;
        float f = snr / (s->overdrive + snr);
        if (f < s->denoise_bound) f = s->denoise_bound;
        if (f > 1.f) f = 1.f;
        if (s->block_ind < STARTUP_SHORT) {
            float ft = (s->init_magn_est[i] - s->overdrive * s->parametric_noise[i]);
            ft /= (s->init_magn_est[i] + 0.0001f);
            if (ft < s->denoise_bound) ft = s->denoise_bound;
            if (ft > 1.f) ft = 1.f;
            f *= (s->block_ind);
            ft *= (STARTUP_SHORT - s->block_ind);
            f += ft;
            f /= (STARTUP_SHORT);
        }
        filt[i] = f;
        s->smooth[i] = f;
        re[i] *= s->smooth[i];
        im[i] *= s->smooth[i];
    }
    memcpy(s->magn_prev_process, magn, sizeof(float) * M);
    memcpy(s->noise_prev, s->noise, sizeof(float) * M);
    /* IFFT ns_core.c:923-944 */
    win[0] = re[0];
    win[1] = re[M - 1];
    for (int i = 1; i < M - 1; i++) {
        win[2 * i] = re[i];
        win[2 * i + 1] = im[i];
    }
    orc_rdft_inverse(&s->fft, win);
    for (int i = 0; i < L; i++) win[i] *= 2.f / L;
    float factor = 1.f;
    if (s->gainmap == 1 && s->block_ind > STARTUP_LONG) {
        float factor1 = 1.f, factor2 = 1.f, energy2 = 0.f;
        for (int i = 0; i < L; i++) energy2 += win[i] * win[i];
        float gain = (float)sqrt(energy2 / (energy1 + 1.f));
        if (gain > 0.5f) {
            factor1 = 1.f + 1.3f * (gain - 0.5f);
            if (gain * factor1 > 1.f) factor1 = 1.f / gain;
        }
        if (gain < 0.5f) {
            if (gain <= s->denoise_bound) gain = s->denoise_bound;
            factor2 = 1.f - 0.3f * (0.5f - gain);
        }
        factor = s->prior_speech_prob * factor1 + (1.f - s->prior_speech_prob) * factor2;
    }
    for (int i = 0; i < L; i++) win[i] = s->window[i] * win[i];
    for (int i = 0; i < L; i++) s->synt_buf[i] += factor * win[i];
    for (int i = 0; i < B; i++) fout[i] = s->synt_buf[i];
    shift_in(s->synt_buf, L, NULL, B);
    for (int i = 0; i < B; i++) out[0][i] = sat16(fout[i]);
    if (hb) {
        /* ns_core.c:1362-1414 */
        float avg_prob = 0.0, sum_a = 0, sum_p = 0, avg_gain = 0.0;
        for (int i = M - delta_hb - 1; i < M - 1; i++) avg_prob += s->speech_prob[i];
        avg_prob = avg_prob / ((float)delta_hb);
        for (int i = 0; i < M; i++) {
            sum_a += s->magn_prev_analyze[i];
            sum_p += s->magn_prev_process[i];
        }
        avg_prob *= sum_p / sum_a;
        for (int i = M - delta_hb - 1; i < M - 1; i++) avg_gain += s->smooth[i];
        avg_gain = avg_gain / ((float)(delta_hb));
        float tmp = 2.f * avg_prob - 1.f;
        float gain_mod = 0.5f * (1.f + (float)tanh(1.0f * tmp));
        float g = 0.5f * gain_mod + 0.5f * avg_gain;
        if (avg_prob >= 0.5f) g = 0.25f * gain_mod + 0.75f * avg_gain;
        g = g * 1.0f;
        if (g < s->denoise_bound) g = s->denoise_bound;
        if (g > 1.f) g = 1.f;
        for (int b = 1; b < num_bands; b++)
            for (int j = 0; j < B; j++) out[b][j] = sat16(g * s->data_buf_hb[b - 1][j]);
    }
    (void)filt;
}

/* ------------------------------------------------------------------ wmix wrapper
 * src/webrtc.c:560-661.  pkgFrame = freq/1000*10 but the core only consumes
 * block_len (=160 at 32 kHz) samples of it; out[][] beyond block_len stays at its
 * calloc zero; chn is passed as num_bands (quirk 2). */
orc_ns *orc_ns_init(int chn, int freq)
{
    if (freq > 32000 || freq % 8000 != 0) return NULL;
    orc_ns *h = calloc(1, sizeof(*h));
    orc_ns_core_init(&h->core, freq);
    h->chn = chn;
    h->freq = freq;
    h->pkg = freq / 1000 * 10;
    return h;
}

void orc_ns_run(orc_ns *h, const int16_t *frame, int16_t *frame_out, int frame_num)
{
    int total = frame_num * h->chn, step = h->pkg * h->chn;
    for (int done = 0; done < total; done += step) {
        float *ip[2] = {h->in[0], h->in[1]}, *op[2] = {h->out[0], h->out[1]};
        for (int i = 0; i < h->pkg; i++)
            for (int c = 0; c < h->chn; c++) h->in[c][i] = (float)(*frame++);
        orc_ns_analyze(&h->core, h->in[0]);
        orc_ns_process(&h->core, (const float *const *)ip, h->chn, op);
        for (int i = 0; i < h->pkg; i++)
            for (int c = 0; c < h->chn; c++) *frame_out++ = (int16_t)h->out[c][i];
    }
}

void orc_ns_release(orc_ns *h) { free(h); }

int orc_run_ns(int chn, int freq, const int16_t *in, int16_t *out, int frames_per_call, int n_calls)
{
    orc_ns *h = orc_ns_init(chn, freq);
    if (!h) return -100;
    size_t step = (size_t)frames_per_call * chn;
    if (out != in) memcpy(out, in, step * n_calls * sizeof(int16_t));
    for (int i = 0; i < n_calls; i++) orc_ns_run(h, out + i * step, out + i * step, frames_per_call);
    orc_ns_release(h);
    return 0;
}
