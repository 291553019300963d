/* oracle/orc_aecm.c -- TEST INFRASTRUCTURE ONLY (checker; never linked into or called by the product).
 *
 * CPU restatement of the fixed-point echo canceller the reference selects by un-commenting `#undef MAKE_WEBRTC_AEC`
 * (src/webrtc.c:168-191): aec_init / aec_setFrameFar / aec_process / aec_process2 / aec_release over WebRtcAecm_*.
 * Follows, function by function:
 *   W:modules/audio_processing/aecm/echo_control_mobile.c  Init :177, BufferFarend :233, Process :277, EstBufDelay :633,
 *       DelayComp :693
 *   W:modules/audio_processing/aecm/aecm_core.c   InitCore :401, ProcessFrame :569, AsymFilt :668, LogOfEnergyInQ8 :709,
 *       CalcEnergies :730, CalcStepSize :858, UpdateChannel :902, CalcSuppressionGain :1118, UpdateFarHistory :172,
 *       AlignedFarend :188, Buffer/FetchFarFrame :1187-1249 (an identity while the core's knownDelay stays 0, which it does)
 *   W:modules/audio_processing/aecm/aecm_core_c.c TimeToFrequencyDomain :171, WindowAndFFT :68, InverseFFTAndWindow :98,
 *       ProcessBlock :280, ComfortNoise :641
 *   W:modules/audio_processing/utility/delay_estimator.c  AddBinaryFarSpectrum :257, ProcessBinarySpectrum :393,
 *       MeanEstimatorFix :672;  delay_estimator_wrapper.c BinarySpectrumFix :52, AddFarSpectrumFix :200, ProcessFix :414
 *       (robust validation is switched off by WebRtcAecm_CreateCore, aecm_core.c:268, so its float histogram never moves)
 *   W:common_audio/ring_buffer.c, W:common_audio/signal_processing/{real_fft,complex_fft,spl_sqrt_floor,
 *       randomization_functions}.c (the FFT lives in orc_nsx.c)
 * Pinned bit-exact against those very functions compiled from the tarball behind the reference's own wrapper
 * (oracle/_ref/libwmixref.so: src/webrtc.c built with the AECM switch, see oracle/Makefile and oracle/aecm_switch/;
 * tests/test_aecm_oracle.py) and by tests/golden/aecm_golden.npz.  Integer path: the bar is bit-exactness.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "orc_aecm.h"
#include "orc_fx_tables.h"
#include "orc_nsx.h" /* orc_spl_real_fft / orc_spl_real_ifft */

#define FRAME 80
#define PART 64
#define PART1 65
#define PART2 128
#define FAR_BUF_LEN 256

/* ---------------------------------------------------------------- SPL primitives */
static int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static int32_t wshl(int32_t a, int s) { return (int32_t)((uint32_t)a << s); }
static int32_t shift32(int32_t x, int c) { return c >= 0 ? wshl(x, c) : (x >> -c); }
static int16_t sat16(int32_t v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : (int16_t)v); }
static int norm_u32(uint32_t a) { return a ? __builtin_clz(a) : 0; }
static int norm_w32(int32_t a)
{
    if (a == 0) return 0;
    if (a < 0) a = ~a;
    return a ? __builtin_clz((uint32_t)a) - 1 : 31;
}
static int norm_w16(int16_t a)
{
    if (a == 0) return 0;
    int v = a < 0 ? (int16_t)~a : a;
    return v ? __builtin_clz((uint32_t)v) - 17 : 15;
}
static int32_t add_sat32(int32_t a, int32_t b)
{
    int32_t s = wadd(a, b);
    if (a < 0) {
        if (b < 0 && s >= 0) s = (int32_t)0x80000000;
    } else if (b > 0 && s < 0) {
        s = 0x7FFFFFFF;
    }
    return s;
}
static int32_t div_w32_w16(int32_t num, int16_t den) { return den ? num / den : 0x7FFFFFFF; }
static int32_t sqrt_floor(int32_t value)
{
    int32_t root = 0;
    for (int n = 15; n >= 0; n--) {
        const int32_t t = wshl(root + (1 << n), n);
        if (value >= t) {
            value -= t;
            root |= 2 << n;
        }
    }
    return root >> 1;
}
static int16_t max_abs16(const int16_t *v, int n)
{
    int m = 0;
    for (int i = 0; i < n; i++) {
        const int a = abs((int)v[i]);
        if (a > m) m = a;
    }
    return (int16_t)(m > 32767 ? 32767 : m);
}
static int popcount32(uint32_t v) { return __builtin_popcount(v); }

/* ---------------------------------------------------------------- int16 ring buffer (ring_buffer.c) */
static void r16_init(orc_r16 *r, int16_t *storage, int count)
{
    r->data = storage;
    r->count = count;
    r->rd = r->wr = 0;
    r->diff_wrap = 0;
    memset(storage, 0, sizeof(int16_t) * (size_t)count);
}
static int r16_avail_read(const orc_r16 *r) { return r->diff_wrap ? r->count - r->rd + r->wr : r->wr - r->rd; }
static int r16_avail_write(const orc_r16 *r) { return r->count - r16_avail_read(r); }
static int r16_move_read(orc_r16 *r, int n)
{
    const int freee = r16_avail_write(r), readable = r16_avail_read(r);
    int pos = r->rd;
    if (n > readable) n = readable;
    if (n < -freee) n = -freee;
    pos += n;
    if (pos > r->count) pos -= r->count, r->diff_wrap = 0;
    if (pos < 0) pos += r->count, r->diff_wrap = 1;
    r->rd = pos;
    return n;
}
static int r16_write(orc_r16 *r, const int16_t *src, int n)
{
    const int freee = r16_avail_write(r), w = freee < n ? freee : n, margin = r->count - r->wr;
    int left = w;
    if (w > margin) {
        memcpy(r->data + r->wr, src, sizeof(int16_t) * (size_t)margin);
        r->wr = 0;
        left -= margin;
        r->diff_wrap = 1;
    }
    memcpy(r->data + r->wr, src + (w - left), sizeof(int16_t) * (size_t)left);
    r->wr += left;
    return w;
}
static int r16_read(orc_r16 *r, int16_t *dst, int n)
{
    const int readable = r16_avail_read(r), k = readable < n ? readable : n, margin = r->count - r->rd;
    if (k > margin) {
        memcpy(dst, r->data + r->rd, sizeof(int16_t) * (size_t)margin);
        memcpy(dst + margin, r->data, sizeof(int16_t) * (size_t)(k - margin));
    } else {
        memcpy(dst, r->data + r->rd, sizeof(int16_t) * (size_t)k);
    }
    r16_move_read(r, k);
    return k;
}

/* ---------------------------------------------------------------- init */
static void init_echo_path(orc_aecm *a, const int16_t *path)  /* InitEchoPathCore, aecm_core.c:291-307 */
{
    for (int i = 0; i < PART1; i++) {
        a->ch_stored[i] = path[i];
        a->ch_adapt16[i] = path[i];
        a->ch_adapt32[i] = wshl(path[i], 16);
    }
    a->mse_adapt_old = 1000;
    a->mse_stored_old = 1000;
    a->mse_threshold = 0x7FFFFFFF;
    a->mse_channel_count = 0;
}

static int core_init(orc_aecm *a, int fs)  /* InitCore, aecm_core.c:401-546; delay estimator inits */
{
    if (fs != 8000 && fs != 16000) return -1;
    a->mult = (int16_t)((int16_t)fs / 8000);
    r16_init(&a->far_fr, a->far_fr_store, FRAME + PART);
    r16_init(&a->near_fr, a->near_fr_store, FRAME + PART);
    r16_init(&a->out_fr, a->out_fr_store, FRAME + PART);
    a->seed = 666;
    a->far_history_pos = 100;
    a->nlp_flag = 1;
    a->fixed_delay = -1;
    init_echo_path(a, fs == 8000 ? fx_aecm_channel_8k : fx_aecm_channel_16k);
    int32_t t32 = PART1 * PART1;
    int16_t t16 = PART1;
    int i;
    for (i = 0; i < (PART1 >> 1) - 1; i++) {
        a->noise_est[i] = wshl(t32, 8);
        t16--;
        t32 -= (int32_t)((t16 << 1) + 1);
    }
    for (; i < PART1; i++) a->noise_est[i] = wshl(t32, 8);
    a->far_energy_min = 32767;
    a->far_energy_max = -32768;
    a->far_energy_vad = 1025;
    a->first_vad = 1;
    a->sup_gain = 256;
    a->sup_gain_old = 256;
    a->sup_a = 3072;
    a->sup_d = 256;
    a->sup_diff_ab = 3072 - 1536;
    a->sup_diff_bd = 1536 - 256;
    a->cng_mode = 1;
    /* WebRtc_InitBinaryDelayEstimator, delay_estimator.c:360-379 */
    for (i = 0; i <= ORC_AECM_MAX_DELAY; i++) a->mean_bit_counts[i] = 20 << 9;
    a->minimum_probability = 32 << 9;
    a->last_delay_probability = 32 << 9;
    a->last_delay = -2;
    return 0;
}

/* ---------------------------------------------------------------- binary delay estimator */
static void mean_estimator(int32_t v, int factor, int32_t *mean)  /* delay_estimator.c:672-684 */
{
    int32_t d = wsub(v, *mean);
    d = d < 0 ? -((-d) >> factor) : d >> factor;
    *mean = wadd(*mean, d);
}

static uint32_t binary_spectrum(const uint16_t *spec, int32_t *thr, int q, int *initialized)  /* wrapper :52-78 */
{
    uint32_t out = 0;
    if (!*initialized)
        for (int i = 12; i <= 43; i++)
            if (spec[i] > 0) {
                thr[i] = wshl((int32_t)spec[i], 15 - q) >> 1;
                *initialized = 1;
            }
    for (int i = 12; i <= 43; i++) {
        const int32_t s = wshl((int32_t)spec[i], 15 - q);
        mean_estimator(s, 6, &thr[i]);
        if (s > thr[i]) out |= 1u << (i - 12);
    }
    return out;
}

static int process_binary_spectrum(orc_aecm *a, uint32_t near_bin)  /* delay_estimator.c:393-487, robust validation off */
{
    int candidate = -1;
    int32_t best = 32 << 9, worst = 0;
    for (int i = 0; i < ORC_AECM_MAX_DELAY; i++) {
        const int32_t bc = popcount32(near_bin ^ a->bin_far_hist[i]) << 9;
        if (a->far_bit_counts[i] > 0) mean_estimator(bc, 13 - ((3 * a->far_bit_counts[i]) >> 4), &a->mean_bit_counts[i]);
    }
    for (int i = 0; i < ORC_AECM_MAX_DELAY; i++) {
        if (a->mean_bit_counts[i] < best) best = a->mean_bit_counts[i], candidate = i;
        if (a->mean_bit_counts[i] > worst) worst = a->mean_bit_counts[i];
    }
    const int32_t depth = worst - best;
    if (a->minimum_probability > 8704 && depth > 2816) {
        int32_t thr = best + 1024;
        if (thr < 8704) thr = 8704;
        if (a->minimum_probability > thr) a->minimum_probability = thr;
    }
    a->last_delay_probability++;
    if (depth > 1024 && (best < a->minimum_probability || best < a->last_delay_probability)) {
        a->last_delay = candidate;
        if (best < a->last_delay_probability) a->last_delay_probability = best;
    }
    return a->last_delay;
}

/* ---------------------------------------------------------------- block processing */
/* TimeToFrequencyDomain + WindowAndFFT, aecm_core_c.c:68-96 / 171-278: returns the scaling (Q-domain) */
static int time_to_freq(const int16_t *td, int16_t *re, int16_t *im, uint16_t *mag, uint32_t *sum)
{
    int16_t win[PART2], spec[PART2 + 2];
    const int q = norm_w16(max_abs16(td, PART2));
    for (int i = 0; i < PART; i++) {
        int16_t s = (int16_t)wshl(td[i], q);
        win[i] = (int16_t)((s * fx_aecm_sqrt_hanning[i]) >> 14);
        s = (int16_t)wshl(td[i + PART], q);
        win[PART + i] = (int16_t)((s * fx_aecm_sqrt_hanning[PART - i]) >> 14);
    }
    orc_spl_real_fft(7, win, spec);
    for (int i = 0; i < PART; i++) re[i] = spec[2 * i], im[i] = (int16_t)-spec[2 * i + 1];
    re[PART] = spec[2 * PART];
    im[0] = 0;
    im[PART] = 0;
    mag[0] = (uint16_t)(re[0] >= 0 ? re[0] : -re[0]);
    mag[PART] = (uint16_t)(re[PART] >= 0 ? re[PART] : -re[PART]);
    *sum = (uint32_t)mag[0] + (uint32_t)mag[PART];
    for (int i = 1; i < PART; i++) {
        if (re[i] == 0) {
            mag[i] = (uint16_t)(im[i] >= 0 ? im[i] : -im[i]);
        } else if (im[i] == 0) {
            mag[i] = (uint16_t)(re[i] >= 0 ? re[i] : -re[i]);
        } else {
            const int16_t ar = (int16_t)(re[i] >= 0 ? re[i] : -re[i]), ai = (int16_t)(im[i] >= 0 ? im[i] : -im[i]);
            mag[i] = (uint16_t)sqrt_floor(add_sat32(ar * ar, ai * ai));
        }
        *sum += (uint32_t)mag[i];
    }
    return q;
}

static int16_t asym_filt(int16_t old, int16_t in, int16_t step_pos, int16_t step_neg)  /* aecm_core.c:668-690 */
{
    if (old == 32767 || old == -32768) return in;
    int16_t r = old;
    if (old > in)
        r = (int16_t)(r - ((old - in) >> step_neg));
    else
        r = (int16_t)(r + ((in - old) >> step_pos));
    return r;
}

static int16_t log_energy_q8(uint32_t energy, int q)  /* aecm_core.c:709-721 */
{
    int16_t l = 7 << 7;
    if (energy > 0) {
        const int zeros = norm_u32(energy);
        const int16_t frac = (int16_t)(((energy << zeros) & 0x7FFFFFFF) >> 23);
        l = (int16_t)(l + ((31 - zeros) << 8) + frac - (q << 8));
    }
    return l;
}

/* CalcEnergies, aecm_core.c:730-851 */
static void calc_energies(orc_aecm *a, const uint16_t *far_spec, int16_t far_q, uint32_t near_energy, int32_t *echo_est)
{
    uint32_t e_adapt = 0, e_stored = 0, e_far = 0;
    int16_t inc_max = 4, dec_max = 11, inc_min = 11, dec_min = 3;
    memmove(a->near_log + 1, a->near_log, sizeof(int16_t) * 63);
    a->near_log[0] = log_energy_q8(near_energy, a->dfa_noisy_q);
    for (int i = 0; i < PART1; i++) {
        echo_est[i] = (int32_t)a->ch_stored[i] * far_spec[i];
        e_far += (uint32_t)far_spec[i];
        e_adapt += (uint32_t)(a->ch_adapt16[i] * far_spec[i]);
        e_stored += (uint32_t)echo_est[i];
    }
    memmove(a->echo_adapt_log + 1, a->echo_adapt_log, sizeof(int16_t) * 63);
    memmove(a->echo_stored_log + 1, a->echo_stored_log, sizeof(int16_t) * 63);
    a->far_log = log_energy_q8(e_far, far_q);
    a->echo_adapt_log[0] = log_energy_q8(e_adapt, 12 + far_q);
    a->echo_stored_log[0] = log_energy_q8(e_stored, 12 + far_q);
    if (a->far_log > 1025) {
        if (a->startup_state == 0) inc_max = 2, dec_min = 2, inc_min = 8;
        a->far_energy_min = asym_filt(a->far_energy_min, a->far_log, inc_min, dec_min);
        a->far_energy_max = asym_filt(a->far_energy_max, a->far_log, inc_max, dec_max);
        a->far_energy_maxmin = (int16_t)(a->far_energy_max - a->far_energy_min);
        int16_t t = (int16_t)(2560 - a->far_energy_min);
        t = t > 0 ? (int16_t)((t * 230) >> 9) : 0;
        t = (int16_t)(t + 230);
        if ((a->startup_state == 0) | (a->vad_update_count > 1024)) {
            a->far_energy_vad = (int16_t)(a->far_energy_min + t);
        } else if (a->far_energy_vad > a->far_log) {
            a->far_energy_vad = (int16_t)(a->far_energy_vad + ((a->far_log + t - a->far_energy_vad) >> 6));
            a->vad_update_count = 0;
        } else {
            a->vad_update_count++;
        }
        a->far_energy_mse = (int16_t)(a->far_energy_vad + (1 << 8));
    }
    if (a->far_log > a->far_energy_vad) {
        if ((a->startup_state == 0) | (a->far_energy_maxmin > 929)) a->current_vad = 1;
    } else {
        a->current_vad = 0;
    }
    if (a->current_vad && a->first_vad) {
        a->first_vad = 0;
        if (a->echo_adapt_log[0] > a->near_log[0]) {
            for (int i = 0; i < PART1; i++) a->ch_adapt16[i] >>= 3;
            a->echo_adapt_log[0] = (int16_t)(a->echo_adapt_log[0] - (3 << 8));
            a->first_vad = 1;
        }
    }
}

static int16_t calc_step_size(const orc_aecm *a)  /* aecm_core.c:858-891 */
{
    int16_t mu = 1;
    if (!a->current_vad) {
        mu = 0;
    } else if (a->startup_state > 0) {
        if (a->far_energy_min >= a->far_energy_max) {
            mu = 10;
        } else {
            const int16_t t16 = (int16_t)(a->far_log - a->far_energy_min);
            int32_t t32 = t16 * 9;
            t32 = div_w32_w16(t32, a->far_energy_maxmin);
            mu = (int16_t)(10 - 1 - (int16_t)t32);
        }
        if (mu < 1) mu = 1;
    }
    return mu;
}

static void store_adaptive_channel(orc_aecm *a, const uint16_t *far_spec, int32_t *echo_est)
{
    for (int i = 0; i < PART1; i++) {
        a->ch_stored[i] = a->ch_adapt16[i];
        echo_est[i] = (int32_t)a->ch_stored[i] * far_spec[i];
    }
}

/* UpdateChannel, aecm_core.c:902-1109 */
static void update_channel(orc_aecm *a, const uint16_t *far_spec, int16_t far_q, const uint16_t *dfa, int16_t mu, int32_t *echo_est)
{
    if (mu) {
        for (int i = 0; i < PART1; i++) {
            const int16_t zeros_ch = (int16_t)norm_u32((uint32_t)a->ch_adapt32[i]), zeros_far = (int16_t)norm_u32((uint32_t)far_spec[i]);
            uint32_t u1;
            int16_t shift_ch_far;
            if (zeros_ch + zeros_far > 31) {
                u1 = (uint32_t)a->ch_adapt32[i] * far_spec[i];
                shift_ch_far = 0;
            } else {
                shift_ch_far = (int16_t)(32 - zeros_ch - zeros_far);
                u1 = (uint32_t)wmul(a->ch_adapt32[i] >> shift_ch_far, far_spec[i]);
            }
            int16_t zeros_num = (int16_t)norm_u32(u1);
            const int16_t zeros_dfa = (int16_t)(dfa[i] ? norm_u32((uint32_t)dfa[i]) : 32);
            const int16_t t16 = (int16_t)(zeros_dfa - 2 + a->dfa_noisy_q - 28 - far_q + shift_ch_far);
            int16_t xfa_q, dfa_q;
            if (zeros_num > t16 + 1) {
                xfa_q = t16;
                dfa_q = (int16_t)(zeros_dfa - 2);
            } else {
                xfa_q = (int16_t)(zeros_num - 2);
                dfa_q = (int16_t)(28 + far_q - a->dfa_noisy_q - shift_ch_far + xfa_q);
            }
            /* WEBRTC_SPL_SHIFT_W32 on unsigned operands: a logical right shift when the count is negative */
            u1 = xfa_q >= 0 ? u1 << xfa_q : u1 >> -xfa_q;
            const uint32_t u2 = dfa_q >= 0 ? (uint32_t)dfa[i] << dfa_q : (uint32_t)dfa[i] >> -dfa_q;
            const int32_t err = (int32_t)u2 - (int32_t)u1;
            zeros_num = (int16_t)norm_w32(err);
            if (err && far_spec[i] > (16 << far_q)) {
                int32_t upd;
                int16_t shift_num;
                if (zeros_num + zeros_far > 31) {
                    upd = err > 0 ? (int32_t)((uint32_t)err * far_spec[i]) : -(int32_t)((uint32_t)(-err) * far_spec[i]);
                    shift_num = 0;
                } else {
                    shift_num = (int16_t)(32 - (zeros_num + zeros_far));
                    upd = err > 0 ? wmul(err >> shift_num, far_spec[i]) : -wmul(-err >> shift_num, far_spec[i]);
                }
                upd = div_w32_w16(upd, (int16_t)(i + 1));
                const int16_t shift2 = (int16_t)(shift_num + shift_ch_far - xfa_q - mu - ((30 - zeros_far) << 1));
                if (norm_w32(upd) < shift2)
                    upd = 0x7FFFFFFF;
                else
                    upd = shift32(upd, shift2);
                a->ch_adapt32[i] = add_sat32(a->ch_adapt32[i], upd);
                if (a->ch_adapt32[i] < 0) a->ch_adapt32[i] = 0;
                a->ch_adapt16[i] = (int16_t)(a->ch_adapt32[i] >> 16);
            }
        }
    }
    if ((a->startup_state == 0) & a->current_vad) {
        store_adaptive_channel(a, far_spec, echo_est);
        return;
    }
    if (a->far_log < a->far_energy_mse)
        a->mse_channel_count = 0;
    else
        a->mse_channel_count++;
    if (a->mse_channel_count >= 20 + 10) {
        int32_t mse_stored = 0, mse_adapt = 0;
        for (int i = 0; i < 20; i++) {
            int32_t d = (int32_t)a->echo_stored_log[i] - (int32_t)a->near_log[i];
            mse_stored += d >= 0 ? d : -d;
            d = (int32_t)a->echo_adapt_log[i] - (int32_t)a->near_log[i];
            mse_adapt += d >= 0 ? d : -d;
        }
        if (((mse_stored << 5) < (29 * mse_adapt)) & ((wshl(a->mse_stored_old, 5)) < wmul(29, a->mse_adapt_old))) {
            for (int i = 0; i < PART1; i++) {  /* ResetAdaptiveChannelC */
                a->ch_adapt16[i] = a->ch_stored[i];
                a->ch_adapt32[i] = wshl(a->ch_stored[i], 16);
            }
        } else if (((29 * mse_stored) > (mse_adapt << 5)) & (mse_adapt < a->mse_threshold) & (a->mse_adapt_old < a->mse_threshold)) {
            store_adaptive_channel(a, far_spec, echo_est);
            if (a->mse_threshold == 0x7FFFFFFF) {
                a->mse_threshold = wadd(mse_adapt, a->mse_adapt_old);
            } else {
                const int scaled = wmul(a->mse_threshold, 5) / 8;
                a->mse_threshold = wadd(a->mse_threshold, wmul(mse_adapt - scaled, 205) >> 8);
            }
        }
        a->mse_channel_count = 0;
        a->mse_stored_old = mse_stored;
        a->mse_adapt_old = mse_adapt;
    }
}

static int16_t calc_suppression_gain(orc_aecm *a)  /* aecm_core.c:1118-1185 */
{
    int16_t sup = 256;
    if (!a->current_vad) {
        sup = 0;
    } else {
        const int16_t t = (int16_t)(a->near_log[0] - a->echo_stored_log[0] - 0);
        const int16_t dE = (int16_t)(t >= 0 ? t : -t);
        if (dE < 400) {
            if (dE < 200) {
                int32_t v = a->sup_diff_ab * dE;
                v += 200 >> 1;
                sup = (int16_t)(a->sup_a - (int16_t)div_w32_w16(v, 200));
            } else {
                int32_t v = a->sup_diff_bd * (400 - dE);
                v += (400 - 200) >> 1;
                sup = (int16_t)(a->sup_d + (int16_t)div_w32_w16(v, 400 - 200));
            }
        } else {
            sup = a->sup_d;
        }
    }
    const int16_t t = sup > a->sup_gain_old ? sup : a->sup_gain_old;
    a->sup_gain_old = sup;
    a->sup_gain = (int16_t)(a->sup_gain + (int16_t)((t - a->sup_gain) >> 4));
    return a->sup_gain;
}

/* ComfortNoise, aecm_core_c.c:641-771 */
static void comfort_noise(orc_aecm *a, const uint16_t *dfa, int16_t *ere, int16_t *eim, const int16_t *lambda)
{
    int16_t noise[PART1], rnd[PART];
    const int16_t shift = (int16_t)(15 - a->dfa_clean_q);
    int16_t min_track;
    if (a->noise_est_ctr < 100) {
        a->noise_est_ctr++;
        min_track = 6;
    } else {
        min_track = 9;
    }
    for (int i = 0; i < PART1; i++) {
        const int32_t v = wshl((int32_t)dfa[i], shift);
        int32_t *ne = &a->noise_est[i];
        if (v < *ne) {
            a->noise_low_ctr[i] = 0;
            if (*ne < (1 << min_track)) {
                a->noise_high_ctr[i]++;
                if (a->noise_high_ctr[i] >= 5) {
                    (*ne)--;
                    a->noise_high_ctr[i] = 0;
                }
            } else {
                *ne -= (*ne - v) >> min_track;
            }
        } else {
            a->noise_high_ctr[i] = 0;
            if ((*ne >> 19) > 0) {
                *ne >>= 11;
                *ne = wmul(*ne, 2049);
            } else if ((*ne >> 11) > 0) {
                *ne = wmul(*ne, 2049);
                *ne >>= 11;
            } else {
                a->noise_low_ctr[i]++;
                if (a->noise_low_ctr[i] >= 5) {
                    *ne += (*ne >> 9) + 1;
                    a->noise_low_ctr[i] = 0;
                }
            }
        }
    }
    for (int i = 0; i < PART1; i++) {
        int32_t v = a->noise_est[i] >> shift;
        if (v > 32767) {
            v = 32767;
            a->noise_est[i] = wshl(v, shift);
        }
        const int16_t t = (int16_t)(16384 - lambda[i]);
        noise[i] = (int16_t)((t * (int16_t)v) >> 14);
    }
    for (int i = 0; i < PART; i++) {  /* WebRtcSpl_RandUArray */
        a->seed = (a->seed * 69069u + 1u) & 0x7FFFFFFFu;
        rnd[i] = (int16_t)(a->seed >> 16);
    }
    for (int i = 0; i < PART1; i++) {
        int16_t ur = 0, ui = 0;
        if (i > 0) {
            const int16_t idx = (int16_t)((359 * rnd[i - 1]) >> 15);
            ur = (int16_t)((noise[i] * fx_aecm_cos[idx]) >> 13);
            ui = (int16_t)((-noise[i] * fx_aecm_sin[idx]) >> 13);
        }
        if (i == PART) ui = 0;
        ere[i] = sat16((int32_t)ere[i] + ur);
        eim[i] = sat16((int32_t)eim[i] + ui);
    }
}

/* ProcessBlock, aecm_core_c.c:280-639 (nearendClean == NULL: the wrapper never passes one) */
static int process_block(orc_aecm *a, const int16_t *far, const int16_t *near, int16_t *out)
{
    uint16_t xfa[PART1], dfa[PART1];
    int16_t dre[PART1], dim[PART1], ere[PART1], eim[PART1], hnl[PART1];
    int32_t echo_est[PART1];
    uint32_t xfa_sum, dfa_sum;
    if (a->startup_state < 2) a->startup_state = (int16_t)((a->tot_count >= 512) + (a->tot_count >= 1024));
    memcpy(a->x_buf + PART, far, sizeof(int16_t) * PART);
    memcpy(a->d_buf + PART, near, sizeof(int16_t) * PART);
    int far_q = time_to_freq(a->x_buf, dre, dim, xfa, &xfa_sum);
    const int zeros_d = time_to_freq(a->d_buf, dre, dim, dfa, &dfa_sum);
    a->dfa_noisy_q_old = a->dfa_noisy_q;
    a->dfa_noisy_q = (int16_t)zeros_d;
    a->dfa_clean_q_old = a->dfa_noisy_q_old;
    a->dfa_clean_q = a->dfa_noisy_q;

    /* UpdateFarHistory + the far half of the delay estimator */
    a->far_history_pos++;
    if (a->far_history_pos >= ORC_AECM_MAX_DELAY) a->far_history_pos = 0;
    a->far_q_domains[a->far_history_pos] = far_q;
    memcpy(&a->far_history[a->far_history_pos * PART1], xfa, sizeof(uint16_t) * PART1);
    if (far_q > 15) return -1;
    {
        const uint32_t b = binary_spectrum(xfa, a->mean_far, far_q, &a->far_initialized);
        memmove(&a->bin_far_hist[1], &a->bin_far_hist[0], sizeof(uint32_t) * (ORC_AECM_MAX_DELAY - 1));
        a->bin_far_hist[0] = b;
        memmove(&a->far_bit_counts[1], &a->far_bit_counts[0], sizeof(int) * (ORC_AECM_MAX_DELAY - 1));
        a->far_bit_counts[0] = popcount32(b);
    }
    if (zeros_d > 15) return -1;
    int delay = process_binary_spectrum(a, binary_spectrum(dfa, a->mean_near, zeros_d, &a->near_initialized));
    if (delay == -1) return -1;
    if (delay == -2) delay = 0;
    if (a->fixed_delay >= 0) delay = a->fixed_delay;
    int pos = a->far_history_pos - delay;
    if (pos < 0) pos += ORC_AECM_MAX_DELAY;
    far_q = a->far_q_domains[pos];
    const uint16_t *far_spec = &a->far_history[pos * PART1];
    const int16_t zeros_x = (int16_t)far_q;

    calc_energies(a, far_spec, zeros_x, dfa_sum, echo_est);
    const int16_t mu = calc_step_size(a);
    a->tot_count++;
    update_channel(a, far_spec, zeros_x, dfa, mu, echo_est);
    const int16_t sup_gain = calc_suppression_gain(a);

    /* Wiener filter coefficients, :434-545 */
    int16_t num_pos = 0;
    for (int i = 0; i < PART1; i++) {
        const int32_t d = wsub(echo_est[i], a->echo_filt[i]);
        a->echo_filt[i] = wadd(a->echo_filt[i], wmul(d, 50) >> 8);
        const int16_t zeros32 = (int16_t)(norm_w32(a->echo_filt[i]) + 1);
        int16_t zeros16 = (int16_t)(norm_w16(sup_gain) + 1);
        uint32_t gained;
        int16_t res_diff;
        if (zeros32 + zeros16 > 16) {
            gained = (uint32_t)a->echo_filt[i] * (uint16_t)sup_gain;
            res_diff = 14 - 12 - 8;
            res_diff = (int16_t)(res_diff + (a->dfa_clean_q - zeros_x));
        } else {
            const int16_t t = (int16_t)(17 - zeros32 - zeros16);
            res_diff = (int16_t)(14 + t - 12 - 8);
            res_diff = (int16_t)(res_diff + (a->dfa_clean_q - zeros_x));
            if (zeros32 > t)
                gained = (uint32_t)a->echo_filt[i] * (uint16_t)(sup_gain >> t);
            else
                gained = (uint32_t)wmul(a->echo_filt[i] >> t, sup_gain);
        }
        zeros16 = (int16_t)norm_w16(a->near_filt[i]);
        const int16_t dq = (int16_t)(a->dfa_clean_q - a->dfa_clean_q_old);
        int16_t t1, t2, q_diff;
        if (zeros16 < dq && a->near_filt[i]) {
            t1 = (int16_t)wshl(a->near_filt[i], zeros16);
            q_diff = (int16_t)(zeros16 - dq);
            t2 = (int16_t)(dfa[i] >> -q_diff);
        } else {
            t1 = (int16_t)(dq < 0 ? a->near_filt[i] >> -dq : wshl(a->near_filt[i], dq));
            q_diff = 0;
            t2 = (int16_t)dfa[i];
        }
        const int32_t nd = (int32_t)(t2 - t1);
        t2 = (int16_t)(nd >> 4);
        t2 = (int16_t)(t2 + t1);
        zeros16 = (int16_t)norm_w16(t2);
        if ((t2) & (-q_diff > zeros16))
            a->near_filt[i] = 32767;
        else
            a->near_filt[i] = (int16_t)(q_diff < 0 ? wshl(t2, -q_diff) : t2 >> q_diff);
        if (gained == 0) {
            hnl[i] = 16384;
        } else if (a->near_filt[i] == 0) {
            hnl[i] = 0;
        } else {
            gained += (uint32_t)(a->near_filt[i] >> 1);
            const uint32_t q = gained / (uint16_t)a->near_filt[i];
            /* WEBRTC_SPL_SHIFT_W32 on an unsigned operand, result read as int32 */
            const int32_t r = (int32_t)(res_diff >= 0 ? q << res_diff : q >> -res_diff);
            if (r > 16384) {
                hnl[i] = 0;
            } else if (r < 0) {
                hnl[i] = 16384;
            } else {
                hnl[i] = (int16_t)(16384 - (int16_t)r);
                if (hnl[i] < 0) hnl[i] = 0;
            }
        }
        if (hnl[i]) num_pos++;
    }
    if (a->mult == 2) {  /* wideband: square, and cap the upper bands by the mean of bands 4..24 */
        int32_t avg = 0;
        for (int i = 0; i < PART1; i++) hnl[i] = (int16_t)((hnl[i] * hnl[i]) >> 14);
        for (int i = 4; i <= 24; i++) avg += (int32_t)hnl[i];
        avg /= 24 - 4 + 1;
        for (int i = 24; i < PART1; i++)
            if (hnl[i] > (int16_t)avg) hnl[i] = (int16_t)avg;
    }
    for (int i = 0; i < PART1; i++) {
        if (a->nlp_flag) {
            if (hnl[i] > 16384)
                hnl[i] = 16384;
            else if (hnl[i] < 3277)
                hnl[i] = 0;
            const int16_t nlp_gain = num_pos < 3 ? 0 : 16384;
            if (!(hnl[i] == 16384 && nlp_gain == 16384)) hnl[i] = (int16_t)((hnl[i] * nlp_gain) >> 14);
        }
        ere[i] = (int16_t)(((int32_t)dre[i] * hnl[i] + 8192) >> 14);
        eim[i] = (int16_t)(((int32_t)dim[i] * hnl[i] + 8192) >> 14);
    }
    if (a->cng_mode == 1) comfort_noise(a, dfa, ere, eim, hnl);

    /* InverseFFTAndWindow, :98-169 */
    {
        int16_t spec[PART2 + 2], td[PART2];
        for (int i = 0; i <= PART; i++) spec[2 * i] = ere[i], spec[2 * i + 1] = (int16_t)-eim[i];
        const int sc = orc_spl_real_ifft(7, spec, td);
        for (int i = 0; i < PART; i++) {
            td[i] = (int16_t)(((int32_t)td[i] * fx_aecm_sqrt_hanning[i] + 8192) >> 14);
            int32_t v = shift32((int32_t)td[i], sc - a->dfa_clean_q);
            out[i] = sat16(v + a->out_buf[i]);
            v = (td[PART + i] * fx_aecm_sqrt_hanning[PART - i]) >> 14;
            v = shift32(v, sc - a->dfa_clean_q);
            a->out_buf[i] = sat16(v);
        }
    }
    memcpy(a->x_buf, a->x_buf + PART, sizeof(int16_t) * PART);
    memcpy(a->d_buf, a->d_buf + PART, sizeof(int16_t) * PART);
    return 0;
}

/* ProcessFrame, aecm_core.c:569-664 (Buffer/FetchFarFrame is an identity: the core's knownDelay is never set) */
static int process_frame(orc_aecm *a, const int16_t *far, const int16_t *near, int16_t *out)
{
    int16_t out_block[PART], far_block[PART], near_block[PART];
    r16_write(&a->far_fr, far, FRAME);
    r16_write(&a->near_fr, near, FRAME);
    while (r16_avail_read(&a->far_fr) >= PART) {
        r16_read(&a->far_fr, far_block, PART);
        r16_read(&a->near_fr, near_block, PART);
        if (process_block(a, far_block, near_block, out_block) == -1) return -1;
        r16_write(&a->out_fr, out_block, PART);
    }
    const int size = r16_avail_read(&a->out_fr);
    if (size < FRAME) r16_move_read(&a->out_fr, size - FRAME);
    r16_read(&a->out_fr, out, FRAME);
    return 0;
}

/* ---------------------------------------------------------------- echo_control_mobile.c */
static void delay_comp(orc_aecm *a)  /* :693-720 */
{
    const int n_far = r16_avail_read(&a->farend);
    const int n_snd = a->ms_in_snd * 8 * a->mult;
    const int delay_new = n_snd - n_far;
    if (delay_new > FAR_BUF_LEN - FRAME * a->mult) {
        int add = (n_snd >> 1) - n_far > FRAME ? (n_snd >> 1) - n_far : FRAME;
        add = add < 10 * FRAME ? add : 10 * FRAME;
        r16_move_read(&a->farend, -add);
        a->delay_change = 1;
    }
}

static void est_buf_delay(orc_aecm *a, short ms)  /* :633-691 */
{
    const short n_far = (short)r16_avail_read(&a->farend);
    const short n_snd = (short)(ms * 8 * a->mult);
    short delay_new = (short)(n_snd - n_far);
    if (delay_new < FRAME) {
        r16_move_read(&a->farend, FRAME);
        delay_new = (short)(delay_new + FRAME);
    }
    const int f = (8 * a->filt_delay + 2 * delay_new) / 10;
    a->filt_delay = (short)(0 > f ? 0 : f);
    const short diff = (short)(a->filt_delay - a->known_delay);
    if (diff > 224) {
        if (a->last_delay_diff < 96)
            a->time_for_delay_change = 0;
        else
            a->time_for_delay_change++;
    } else if (diff < 96 && a->known_delay > 0) {
        if (a->last_delay_diff > 224)
            a->time_for_delay_change = 0;
        else
            a->time_for_delay_change++;
    } else {
        a->time_for_delay_change = 0;
    }
    a->last_delay_diff = diff;
    if (a->time_for_delay_change > 25) a->known_delay = (int)a->filt_delay - 160 > 0 ? (int)a->filt_delay - 160 : 0;
}

int orc_aecm_buffer_farend(orc_aecm *a, const int16_t *far, int n)  /* :233-275 */
{
    if (n != 80 && n != 160) return -1;
    if (!a->ec_startup) delay_comp(a);
    r16_write(&a->farend, far, n);
    return 0;
}

int orc_aecm_process(orc_aecm *a, const int16_t *near, int16_t *out, int n, int ms)  /* :277-482 */
{
    int ret = 0;
    if (n != 80 && n != 160) return -1;
    if (ms < 0)
        ms = 0, ret = -1;
    else if (ms > 500)
        ms = 500, ret = -1;
    ms += 10;
    a->ms_in_snd = (short)ms;
    const short n_frames = (short)(n / FRAME), n_blocks = (short)(n_frames / a->mult);
    if (a->ec_startup) {
        if (out != near) memcpy(out, near, sizeof(short) * (size_t)n);
        const short filled = (short)((short)r16_avail_read(&a->farend) / FRAME);
        if (a->check_buff_size) {
            a->check_buf_size_ctr++;
            if (a->counter == 0) {
                a->first_val = a->ms_in_snd;
                a->sum = 0;
            }
            const double lim = 0.2 * a->ms_in_snd > 8 ? 0.2 * a->ms_in_snd : 8;
            if (abs(a->first_val - a->ms_in_snd) < lim) {
                a->sum = (short)(a->sum + a->ms_in_snd);
                a->counter++;
            } else {
                a->counter = 0;
            }
            if (a->counter * n_blocks >= 6) {
                const int v = (3 * a->sum * a->mult) / (a->counter * 40);
                a->buf_size_start = (short)(v < 50 ? v : 50);
                a->check_buff_size = 0;
            }
            if (a->check_buf_size_ctr * n_blocks > 50) {
                const int v = (3 * a->ms_in_snd * a->mult) / 40;
                a->buf_size_start = (short)(v < 50 ? v : 50);
                a->check_buff_size = 0;
            }
        }
        if (!a->check_buff_size) {
            if (filled == a->buf_size_start) {
                a->ec_startup = 0;
            } else if (filled > a->buf_size_start) {
                r16_move_read(&a->farend, r16_avail_read(&a->farend) - (int)a->buf_size_start * FRAME);
                a->ec_startup = 0;
            }
        }
    } else {
        for (short i = 0; i < n_frames; i++) {
            int16_t far[FRAME];
            const short filled = (short)((short)r16_avail_read(&a->farend) / FRAME);
            if (filled > 0) {
                r16_read(&a->farend, far, FRAME);
                memcpy(a->farend_old[i], far, sizeof(far));
            } else {
                memcpy(far, a->farend_old[i], sizeof(far));
            }
            if ((i == 0 && a->fs == 8000) || (i == 1 && a->fs == 16000)) est_buf_delay(a, a->ms_in_snd);
            if (process_frame(a, far, near + FRAME * i, out + FRAME * i) == -1) return -1;
        }
    }
    return ret;
}

/* ---------------------------------------------------------------- the wrapper with the AECM switch, src/webrtc.c:217-505 */
orc_aecm *orc_aecm_init(int chn, int freq, int interval_ms)
{
    if (freq > 16000 || freq % 8000 != 0) return NULL;
    orc_aecm *a = (orc_aecm *)calloc(1, sizeof(orc_aecm));
    if (!a) return NULL;
    a->fs = freq;
    if (core_init(a, freq) != 0) {
        free(a);
        return NULL;
    }
    r16_init(&a->farend, a->farend_store, 50 * FRAME);
    a->delay_change = 1;
    a->check_buff_size = 1;
    a->ec_startup = 1;
    a->chn = chn;
    a->pkg = freq / 1000 * ((freq <= 8000 && interval_ms % 20 == 0) ? 20 : 10);
    return a;
}

/* mode bit 1: aec_setFrameFar, bit 2: aec_process, 3: aec_process2 (far then near per packet) */
int orc_aecm_run(orc_aecm *a, int mode, const int16_t *far, const int16_t *nearp, int16_t *out, int frame_num, int delay_ms)
{
    int16_t f[160], in[160], o[160];
    const int total = frame_num * a->chn, per = a->pkg * a->chn;
    for (int c = 0; c < total; c += per) {
        for (int p = 0; p < a->pkg; p++) {
            if (mode & 1) f[p] = far[(size_t)c + (size_t)p * a->chn];
            if (mode & 2) in[p] = nearp[(size_t)c + (size_t)p * a->chn];
        }
        if (mode & 1) {
            const int rc = orc_aecm_buffer_farend(a, f, a->pkg);
            if (rc != 0) return rc;
        }
        if (mode & 2) {
            memset(o, 0, sizeof(o));
            const int rc = orc_aecm_process(a, in, o, a->pkg, delay_ms);
            if (rc != 0) return rc;
            for (int p = 0; p < a->pkg; p++)
                for (int ch = 0; ch < a->chn; ch++) out[(size_t)c + (size_t)p * a->chn + ch] = o[p];
        }
    }
    return 0;
}

void orc_aecm_release(orc_aecm *a) { free(a); }

int orc_run_aecm(int chn, int freq, int interval_ms, const int16_t *far, const int16_t *nearp, int16_t *out, int frames_per_call,
                 int n_calls, int delay_ms, int split)
{
    orc_aecm *a = orc_aecm_init(chn, freq, interval_ms);
    if (!a) return -100;
    const size_t step = (size_t)frames_per_call * (size_t)chn;
    int rc = 0;
    for (int i = 0; i < n_calls && rc == 0; i++) {
        if (split) {
            rc = orc_aecm_run(a, 1, far + i * step, NULL, NULL, frames_per_call, delay_ms);
            if (rc == 0) rc = orc_aecm_run(a, 2, NULL, nearp + i * step, out + i * step, frames_per_call, delay_ms);
        } else {
            rc = orc_aecm_run(a, 3, far + i * step, nearp + i * step, out + i * step, frames_per_call, delay_ms);
        }
    }
    orc_aecm_release(a);
    return rc;
}
