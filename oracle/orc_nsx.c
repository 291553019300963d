/* oracle/orc_nsx.c -- TEST INFRASTRUCTURE ONLY (checker; never linked into or called by the product).
 *
 * CPU restatement of the fixed-point noise suppressor the reference selects with MAKE_WEBRTC_NSX
 * (src/webrtc.c:512-521: ns_init / ns_process / ns_release over WebRtcNsx_Create / Init / set_policy(2) / Process).
 * Follows, function by function:
 *   W:modules/audio_processing/ns/nsx_core.c      InitCore :631, set_policy_core :786, NoiseEstimationC :334,
 *       UpdateNoiseEstimate :303, CalcParametricNoiseEstimate :586, FeatureParameterExtraction :821,
 *       ComputeSpectralFlatness :1022, ComputeSpectralDifference :1091, DataAnalysis :1184, DataSynthesis :1421,
 *       ProcessCore :1501, AnalysisUpdateC :524, SynthesisUpdateC :491, PrepareSpectrumC :456, DenormalizeC :477
 *   W:modules/audio_processing/ns/nsx_core_c.c    SpeechNoiseProb :26
 *   W:common_audio/signal_processing/real_fft.c :46-100, complex_fft.c :30-296 (mode 1), complex_bit_reverse.c,
 *       energy.c, get_scaling_square.c, spl_sqrt_floor.c, division_operations.c, min_max_operations.c, spl_inl.h
 * Pinned bit-exact against those very functions compiled from the tarball (oracle/_ref/libwmixref.so, which also holds
 * src/webrtc.c built a second time with -DMAKE_WEBRTC_NSX; tests/test_nsx_oracle.py) and by tests/golden/nsx_golden.npz.
 * Everything is integer: the bar is bit-exactness.  Constant tables: oracle/orc_fx_tables.h (generated, numbers only).
 *
 * Written bin-wise (one helper per per-bin step) because the GPU kernel maps one lane to one bin; signed overflow the
 * reference leaves to two's complement is spelled with unsigned arithmetic.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "orc_fx_tables.h"
#include "orc_nsx.h"

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
static int size_in_bits(uint32_t n) { return n ? 32 - __builtin_clz(n) : 0; }
static int32_t mul_rsft_round(int16_t a, int16_t b, int c) { return ((int32_t)a * b + ((int32_t)1 << (c - 1))) >> c; }
static int32_t div_w32_w16(int32_t num, int16_t den) { return den ? num / den : 0x7FFFFFFF; }
static uint32_t div_u32_u16(uint32_t num, uint16_t den) { return den ? num / den : 0xFFFFFFFFu; }

/* log2 of a non-zero magnitude in Q8 through the 256-entry fraction table (nsx_core.c:362-370 and five more sites) */
static int16_t log2_q8(uint32_t v)
{
    const int zeros = norm_u32(v);
    const int frac = (int)(((v << zeros) & 0x7FFFFFFF) >> 23);
    return (int16_t)(((31 - zeros) << 8) + fx_nsx_log_frac[frac]);
}

/* spl_sqrt_floor.c:48-75 */
static int32_t sqrt_floor(int32_t value)
{
    int32_t root = 0;
    for (int n = 15; n >= 0; n--) {
        const int32_t t = root + (1 << n);
        if (value >= wshl(t, n)) {
            value -= wshl(t, n);
            root |= 2 << n;
        }
    }
    return root >> 1;
}

/* energy.c:20-37 + get_scaling_square.c:20-45 */
static int32_t energy(const int16_t *v, int n, int *scale)
{
    int smax = -1;
    for (int i = 0; i < n; i++) {
        const int16_t a = (int16_t)(v[i] > 0 ? v[i] : -v[i]);
        if (a > smax) smax = a;
    }
    int sc = 0;
    if (smax != 0) {
        const int nbits = size_in_bits((uint32_t)n), t = norm_w32(wmul(smax, smax));
        sc = t > nbits ? 0 : nbits - t;
    }
    int32_t en = 0;
    for (int i = 0; i < n; i++) en = wadd(en, ((int32_t)v[i] * v[i]) >> sc);
    *scale = sc;
    return en;
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

/* ---------------------------------------------------------------- SPL complex FFT (radix-2 DIT, per-stage scaling) */
static void bit_reverse(int16_t *x, int stages)
{
    const int n = 1 << stages;
    for (int i = 0; i < n; i++) {
        int r = 0;
        for (int b = 0; b < stages; b++) r |= ((i >> b) & 1) << (stages - 1 - b);
        if (r > i) {
            int32_t t;
            memcpy(&t, x + 2 * i, 4);
            memcpy(x + 2 * i, x + 2 * r, 4);
            memcpy(x + 2 * r, &t, 4);
        }
    }
}

/* complex_fft.c mode 1: forward halves every stage (rounding 1 before the Q14 shift, 16384 after); the inverse picks a
 * shift of 0..2 per stage from the largest |value| of the whole array and returns the number of shifts it applied */
static int cfft(int16_t *x, int stages, int inverse)
{
    const int n = 1 << stages;
    int scale = 0, k = 9;
    for (int l = 1; l < n; l <<= 1, k--) {
        int shift = inverse ? 0 : 1;
        int32_t round2 = inverse ? 8192 : 16384;
        if (inverse) {
            const int32_t m = max_abs16(x, 2 * n);
            if (m > 13573) shift++, scale++, round2 <<= 1;
            if (m > 27146) shift++, scale++, round2 <<= 1;
        }
        for (int m = 0; m < l; m++) {
            const int j0 = m << k;
            const int16_t wr = fx_spl_sin1024[j0 + 256], wi = (int16_t)(inverse ? fx_spl_sin1024[j0] : -fx_spl_sin1024[j0]);
            for (int i = m; i < n; i += 2 * l) {
                const int j = i + l;
                const int32_t tr = ((int32_t)wr * x[2 * j] - (int32_t)wi * x[2 * j + 1] + 1) >> 1;
                const int32_t ti = ((int32_t)wr * x[2 * j + 1] + (int32_t)wi * x[2 * j] + 1) >> 1;
                const int32_t qr = wshl(x[2 * i], 14), qi = wshl(x[2 * i + 1], 14);
                x[2 * j] = (int16_t)((qr - tr + round2) >> (shift + 14));
                x[2 * j + 1] = (int16_t)((qi - ti + round2) >> (shift + 14));
                x[2 * i] = (int16_t)((qr + tr + round2) >> (shift + 14));
                x[2 * i + 1] = (int16_t)((qi + ti + round2) >> (shift + 14));
            }
        }
    }
    return scale;
}

void orc_spl_real_fft(int order, const int16_t *in, int16_t *out)
{
    int16_t buf[2 << 10];
    const int n = 1 << order;
    for (int i = 0; i < n; i++) buf[2 * i] = in[i], buf[2 * i + 1] = 0;
    bit_reverse(buf, order);
    cfft(buf, order, 0);
    memcpy(out, buf, sizeof(int16_t) * (n + 2));
}

int orc_spl_real_ifft(int order, const int16_t *in, int16_t *out)
{
    int16_t buf[2 << 10];
    const int n = 1 << order;
    memcpy(buf, in, sizeof(int16_t) * (n + 2));
    for (int i = n + 2; i < 2 * n; i += 2) buf[i] = in[2 * n - i], buf[i + 1] = (int16_t)-in[2 * n - i + 1];
    bit_reverse(buf, order);
    const int sc = cfft(buf, order, 1);
    for (int i = 0; i < n; i++) out[i] = buf[2 * i];
    return sc;
}

/* ---------------------------------------------------------------- init */
int orc_nsx_core_init(orc_nsx_core *s, int fs, int mode)
{
    if (fs != 8000 && fs != 16000 && fs != 32000 && fs != 48000) return -1;
    if (mode < 0 || mode > 3) return -1;
    memset(s, 0, sizeof(*s));
    s->fs = fs;
    const int nb = fs == 8000;
    s->block = nb ? 80 : 160;
    s->ana = nb ? 128 : 256;
    s->stages = nb ? 7 : 8;
    s->window = nb ? fx_nsx_window128 : fx_nsx_window256;
    s->thr_lrt = nb ? 131072 : 212644;
    s->max_lrt = nb ? 0x0040000 : 0x0080000;
    s->min_lrt = nb ? 52429 : 104858;
    s->ana2 = s->ana / 2;
    s->nbins = s->ana2 + 1;
    for (int i = 0; i < 3 * ORC_NSX_BINS; i++) s->lq[i] = 2048, s->dens[i] = 153;
    for (int i = 0; i < 3; i++) s->counter[i] = (int16_t)((int16_t)(200 * (i + 1)) / 3);
    for (int i = 0; i < ORC_NSX_BINS; i++) s->filt[i] = 16384;
    s->prior_nonspeech = 8192;
    s->thr_diff = 50;
    s->thr_flat = 20480;
    s->feat_lrt = s->thr_lrt;
    s->feat_flat = s->thr_flat;
    s->feat_diff = s->thr_diff;
    s->w_lrt = 6;
    s->block_index = -1;
    s->model_update = 1 << 9;
    s->min_norm = 15;
    /* set_policy_core(mode) */
    static const uint16_t od[4] = {256, 256, 282, 320}, db[4] = {8192, 4096, 2048, 1475};
    s->overdrive = od[mode];
    s->denoise_bound = db[mode];
    s->gain_map = mode != 0;
    s->factor2 = mode == 1 ? fx_nsx_factor2_mode1 : (mode == 3 ? fx_nsx_factor2_mode3 : fx_nsx_factor2_mode2);
    return 0;
}

/* ---------------------------------------------------------------- quantile noise estimate */
/* UpdateNoiseEstimate, nsx_core.c:303-331 */
static void update_noise_estimate(orc_nsx_core *s, int offset)
{
    int16_t mx = -32768;
    for (int i = 0; i < s->nbins; i++)
        if (s->lq[offset + i] > mx) mx = s->lq[offset + i];
    s->q_noise = 14 - (int)mul_rsft_round(11819, mx, 21);
    for (int i = 0; i < s->nbins; i++) {
        const int32_t e = 11819 * s->lq[offset + i];
        int32_t m = 0x00200000 | (e & 0x001FFFFF);
        int16_t sh = (int16_t)(e >> 21);
        sh = (int16_t)(sh - 21);
        sh = (int16_t)(sh + (int16_t)s->q_noise);
        m = sh < 0 ? m >> -sh : wshl(m, sh);
        s->quant[i] = sat16(m);
    }
}

/* NoiseEstimationC, nsx_core.c:334-453 */
static void noise_estimation(orc_nsx_core *s, const uint16_t *magn, uint32_t *noise, int16_t *q_noise)
{
    int16_t lmagn[ORC_NSX_BINS];
    const int tabind = s->stages - s->norm_data;
    const int16_t logval = (int16_t)(tabind < 0 ? -fx_nsx_log_table[-tabind] : fx_nsx_log_table[tabind]);
    for (int i = 0; i < s->nbins; i++) {
        if (magn[i]) {
            const int16_t l2 = log2_q8(magn[i]);
            lmagn[i] = (int16_t)((l2 * 22713) >> 15);
            lmagn[i] = (int16_t)(lmagn[i] + logval);
        } else {
            lmagn[i] = logval;
        }
    }
    int offset = 0;
    for (int e = 0; e < 3; e++) {
        offset = e * s->nbins;
        const int16_t counter = s->counter[e], count_div = fx_nsx_counter_div[counter];
        const int16_t count_prod = (int16_t)(counter * count_div);
        for (int i = 0; i < s->nbins; i++) {
            int16_t *lq = &s->lq[offset + i], *dn = &s->dens[offset + i];
            int16_t delta;
            if (*dn > 512)
                delta = (int16_t)(2621440 >> (14 - norm_w16(*dn)));
            else
                delta = s->block_index < 200 ? 1024 : 5120;
            int16_t step = (int16_t)((delta * count_div) >> 14);
            if (lmagn[i] > *lq) {
                step = (int16_t)(step + 2);
                *lq = (int16_t)(*lq + step / 4);
            } else {
                step = (int16_t)(step + 1);
                *lq = (int16_t)(*lq - (int16_t)((step / 2) * 3 / 2));
                if (*lq < logval) *lq = logval;
            }
            const int d = lmagn[i] - *lq;
            if ((d >= 0 ? d : -d) < 3) {
                const int16_t a = (int16_t)mul_rsft_round(*dn, count_prod, 15), b = (int16_t)mul_rsft_round(21845, count_div, 15);
                *dn = (int16_t)(a + b);
            }
        }
        if (counter >= 200) {
            s->counter[e] = 0;
            if (s->block_index >= 200) update_noise_estimate(s, offset);
        }
        s->counter[e]++;
    }
    if (s->block_index < 200) update_noise_estimate(s, offset);
    for (int i = 0; i < s->nbins; i++) noise[i] = (uint32_t)s->quant[i];
    *q_noise = (int16_t)s->q_noise;
}

/* CalcParametricNoiseEstimate, nsx_core.c:586-628 */
static void parametric_noise(const orc_nsx_core *s, int16_t exp_avg, int32_t num_avg, int bin, uint32_t *est, uint32_t *est_avg)
{
    int32_t t = num_avg - ((exp_avg * fx_nsx_log_index[bin]) >> 15);
    t += (s->min_norm - s->stages) * 2048;
    if (t > 0) {
        const int16_t ip = (int16_t)(t >> 11), fp = (int16_t)(t & 0x7ff);
        int32_t b = (fp >> 10) ? 2048 - (((2048 - fp) * 1244) >> 10) : (fp * 804) >> 10;
        b = shift32(b, ip - 11);
        *est_avg = (uint32_t)wshl(1, ip) + (uint32_t)b;
        *est = *est_avg * (uint32_t)(s->block_index + 1);
    }
}

/* ---------------------------------------------------------------- features */
/* ComputeSpectralFlatness, nsx_core.c:1022-1084 */
static void spectral_flatness(orc_nsx_core *s, const uint16_t *magn)
{
    uint32_t num = 0;
    const uint32_t den = s->sum_magn - (uint32_t)magn[0];
    for (int i = 1; i < s->nbins; i++) {
        if (!magn[i]) {
            s->feat_flat -= (s->feat_flat * (uint32_t)4915) >> 14;
            return;
        }
        num += (uint32_t)log2_q8(magn[i]);
    }
    const int zeros = norm_u32(den);
    const int frac = (int)(((den << zeros) & 0x7FFFFFFF) >> 23);
    const int32_t lden = ((31 - zeros) << 8) + fx_nsx_log_frac[frac];
    int32_t lf = (int32_t)num;
    lf = wadd(lf, wshl(s->stages - 1, s->stages + 7));
    lf = wsub(lf, wshl(lden, s->stages - 1));
    lf = wshl(lf, 10 - s->stages);
    const int32_t mant = 0x00020000 | ((lf >= 0 ? lf : -lf) & 0x0001FFFF);
    const int16_t ip = (int16_t)(7 - (lf >> 17));
    const int32_t cur = ip > 0 ? mant >> ip : wshl(mant, -ip);
    int32_t d = wsub(cur, (int32_t)s->feat_flat);
    d = wmul(d, 4915);
    s->feat_flat += (uint32_t)(d >> 14);
}

/* ComputeSpectralDifference, nsx_core.c:1091-1181 */
static void spectral_difference(orc_nsx_core *s, const uint16_t *magn)
{
    int32_t avg_pause = 0, mx = 0, mn = s->pause[0];
    for (int i = 0; i < s->nbins; i++) {
        avg_pause = wadd(avg_pause, s->pause[i]);
        if (s->pause[i] > mx) mx = s->pause[i];
        if (s->pause[i] < mn) mn = s->pause[i];
    }
    avg_pause >>= s->stages - 1;
    const int32_t avg_magn = (int32_t)(s->sum_magn >> (s->stages - 1));
    const int32_t dev = mx - avg_pause > avg_pause - mn ? mx - avg_pause : avg_pause - mn;
    int n_shifts = 10 + s->stages - norm_w32(dev);
    if (n_shifts < 0) n_shifts = 0;
    uint32_t var_magn = 0, var_pause = 0;
    int32_t cov = 0;
    for (int i = 0; i < s->nbins; i++) {
        const int16_t dm = (int16_t)((int32_t)magn[i] - avg_magn);
        const int32_t dp = wsub(s->pause[i], avg_pause);
        var_magn += (uint32_t)(dm * dm);
        cov = wadd(cov, wmul(dp, dm));
        const int32_t r = dp >> n_shifts;
        var_pause += (uint32_t)wmul(r, r);
    }
    /* the count reaches 35 for near-silent input: undefined in C; the reference's x86 shift takes it modulo 32 */
    s->cur_avg_energy += s->magn_energy >> ((2 * s->norm_data + s->stages - 1) & 31);
    uint32_t diff = var_magn;
    if (var_pause && cov) {
        uint32_t c = (uint32_t)(cov >= 0 ? cov : -cov);
        const int norm = norm_u32(c) - 16;
        c = norm > 0 ? c << norm : c >> -norm;
        const uint32_t c2 = c * c;
        n_shifts += norm;
        n_shifts *= 2;
        if (n_shifts < 0) {
            var_pause >>= -n_shifts;
            n_shifts = 0;
        }
        if (var_pause > 0) {
            const uint32_t q = (c2 / var_pause) >> n_shifts;
            diff -= diff < q ? diff : q;
        } else {
            diff = 0;
        }
    }
    const uint32_t cur = diff >> (2 * s->norm_data);
    if (s->feat_diff > cur)
        s->feat_diff -= ((s->feat_diff - cur) * (uint32_t)77) >> 8;
    else
        s->feat_diff += ((cur - s->feat_diff) * (uint32_t)77) >> 8;
}

/* the two largest histogram peaks (position 2i+1, weight) and their merge, nsx_core.c:913-945 / 963-993 */
static void two_peaks(const int16_t *hist, uint32_t *pos1, int *w1)
{
    int max1 = 0, max2 = 0, wa = 0, wb = 0;
    uint32_t pa = 0, pb = 0;
    for (int i = 0; i < ORC_NSX_HIST; i++) {
        if (hist[i] > max1) {
            max2 = max1, wb = wa, pb = pa;
            max1 = hist[i], wa = hist[i], pa = (uint32_t)(2 * i + 1);
        } else if (hist[i] > max2) {
            max2 = hist[i], wb = hist[i], pb = (uint32_t)(2 * i + 1);
        }
    }
    if (pa - pb < 4 && wb * 2 > wa) {
        wa += wb;
        pa = (pa + pb) >> 1;
    }
    *pos1 = pa;
    *w1 = wa;
}

/* FeatureParameterExtraction, nsx_core.c:821-1017 */
static void feature_parameters(orc_nsx_core *s, int extract)
{
    if (!extract) {
        uint32_t h = (uint32_t)s->feat_lrt;
        if (h < ORC_NSX_HIST) s->hist_lrt[h]++;
        h = (s->feat_flat * 5) >> 8;
        if (h < ORC_NSX_HIST) s->hist_flat[h]++;
        h = ORC_NSX_HIST;
        if (s->time_avg_energy > 0) h = ((s->feat_diff * 5) >> s->stages) / s->time_avg_energy;
        if (h < ORC_NSX_HIST) s->hist_diff[h]++;
        return;
    }
    int use_diff = 1;
    int32_t avg = 0, avg_sq = 0;
    int16_t count = 0;
    int i;
    for (i = 0; i < 10; i++) {
        const int16_t j = (int16_t)(2 * i + 1);
        const int32_t t = s->hist_lrt[i] * j;
        avg = wadd(avg, t);
        count = (int16_t)(count + s->hist_lrt[i]);
        avg_sq = wadd(avg_sq, wmul(t, j));
    }
    int32_t avg_all = avg;
    for (; i < ORC_NSX_HIST; i++) {
        const int16_t j = (int16_t)(2 * i + 1);
        const int32_t t = s->hist_lrt[i] * j;
        avg_all = wadd(avg_all, t);
        avg_sq = wadd(avg_sq, wmul(t, j));
    }
    const int32_t fluct = wsub(wmul(avg_sq, count), wmul(avg, avg_all)), thr_fluct = 10240 * count;
    const uint32_t six_avg = 6 * (uint32_t)avg;
    if (fluct < thr_fluct || count == 0 || six_avg > (uint32_t)(100 * count)) {
        s->thr_lrt = s->max_lrt;
    } else {
        const int32_t t = (int32_t)((six_avg << (9 + s->stages)) / (uint32_t)count / 25);
        s->thr_lrt = t > s->max_lrt ? s->max_lrt : (t < s->min_lrt ? s->min_lrt : t);
    }
    if (fluct < thr_fluct) use_diff = 0;

    uint32_t pos;
    int weight;
    two_peaks(s->hist_flat, &pos, &weight);
    int use_flat = 1;
    if (weight < 154 || pos < 24) {
        use_flat = 0;
    } else {
        const uint32_t t = 922 * pos;
        s->thr_flat = t > 38912 ? 38912 : (t < 4096 ? 4096 : t);
    }
    if (use_diff) {
        two_peaks(s->hist_diff, &pos, &weight);
        const uint32_t t = 6 * pos;
        s->thr_diff = t > 100 ? 100 : (t < 16 ? 16 : t);
        if (weight < 154) use_diff = 0;
    }
    const int share = 6 / (1 + use_flat + use_diff);
    s->w_lrt = (int16_t)share;
    s->w_flat = (int16_t)(use_flat * share);
    s->w_diff = (int16_t)(use_diff * share);
    memset(s->hist_lrt, 0, sizeof(s->hist_lrt));
    memset(s->hist_diff, 0, sizeof(s->hist_diff));
    memset(s->hist_flat, 0, sizeof(s->hist_flat));
}

/* the sigmoid map 0.5*(1 + tanh) through the 17-entry table, nsx_core_c.c:104-116 / 137-149 / 185-199.  x is Q14;
 * the table index is the int16 of bits 14..29.  Indicator 0 tests 0 <= index < 16; indicators 1 and 2 only index < 16
 * (a negative index would read in front of the table there: it needs x >= 2^29, which their inputs cannot reach, so
 * it is treated like index >= 16).  Indicator 2 rounds the interpolation, the others truncate. */
static int16_t indicator(uint32_t x_q14, int positive, int rounded)
{
    int16_t ind = (int16_t)(positive ? 16384 : 0);
    const int16_t idx = (int16_t)(x_q14 >> 14);
    if (idx < 16 && idx >= 0) {
        int16_t v = fx_nsx_indicator[idx];
        const int16_t d = (int16_t)(fx_nsx_indicator[idx + 1] - fx_nsx_indicator[idx]), frac = (int16_t)(x_q14 & 0x3fff);
        v = (int16_t)(v + (int16_t)(rounded ? mul_rsft_round(d, frac, 14) : (d * frac) >> 14));
        ind = (int16_t)(positive ? 8192 + v : 8192 - v);
    }
    return ind;
}

/* SpeechNoiseProb, nsx_core_c.c:26-260 */
static void speech_noise_prob(orc_nsx_core *s, uint16_t *nonspeech, const uint32_t *prior_snr, const uint32_t *post_snr)
{
    int32_t lrt_sum = 0;
    for (int i = 0; i < s->nbins; i++) {
        int32_t bessel = (int32_t)post_snr[i];
        const int nt = norm_u32(post_snr[i]);
        const uint32_t num = post_snr[i] << nt;
        const uint32_t den = nt > 10 ? prior_snr[i] << (nt - 11) : prior_snr[i] >> (11 - nt);
        bessel = den > 0 ? wsub(bessel, (int32_t)(num / den)) : 0;
        const int zeros = norm_u32(prior_snr[i]);
        int32_t f = (int32_t)(((prior_snr[i] << zeros) & 0x7FFFFFFF) >> 19);
        int32_t t = wmul(wmul(f, f), -43) >> 19;
        t += ((int16_t)f * 5412) >> 12;
        f = t + 37;
        t = (int32_t)(((31 - zeros) << 12) + f) - (11 << 12);
        const int32_t log_prior = wmul(t, 178) >> 8;
        const int32_t half = wadd(log_prior, s->lrt_avg[i]) / 2;
        s->lrt_avg[i] = wadd(s->lrt_avg[i], wsub(bessel, half));
        lrt_sum = wadd(lrt_sum, s->lrt_avg[i]);
    }
    s->feat_lrt = wmul(lrt_sum, 10) >> (s->stages + 11);

    /* indicator 0: average LRT */
    int32_t d0 = wsub(lrt_sum, s->thr_lrt);
    int sh = 7 - s->stages, pos = 1;
    if (d0 < 0) pos = 0, d0 = -d0, sh++;
    d0 = shift32(d0, sh);
    int32_t ind_prior = s->w_lrt * indicator((uint32_t)d0, pos, 0);
    /* indicator 1: spectral flatness */
    if (s->w_flat) {
        const uint32_t a = s->feat_flat * (uint32_t)400;
        uint32_t d = s->thr_flat - a;
        sh = 4, pos = 1;
        if (s->thr_flat < a) pos = 0, d = a - s->thr_flat, sh++;
        ind_prior += s->w_flat * indicator(div_u32_u16(d << sh, 25), pos, 0);
    }
    /* indicator 2: template spectral difference */
    if (s->w_diff) {
        uint32_t a = 0;
        if (s->feat_diff) {
            int nt = norm_u32(s->feat_diff);
            if (20 - s->stages < nt) nt = 20 - s->stages;
            a = s->feat_diff << nt;
            const uint32_t e = s->time_avg_energy >> (20 - s->stages - nt);
            a = e > 0 ? a / e : 0x7fffffffu;
        }
        const uint32_t thr = (s->thr_diff << 17) / 25;
        uint32_t d = a - thr;
        sh = 1, pos = 1;
        if (d & 0x80000000u) pos = 0, d = thr - a, sh--;
        ind_prior += s->w_diff * indicator(d >> sh, pos, 1);
    }
    const int16_t ind16 = (int16_t)((98307 - ind_prior) / 6);
    const int16_t dprior = (int16_t)(ind16 - s->prior_nonspeech);
    s->prior_nonspeech = (int16_t)(s->prior_nonspeech + (int16_t)((1638 * dprior) >> 14));

    memset(nonspeech, 0, sizeof(uint16_t) * (size_t)s->nbins);
    if (s->prior_nonspeech > 0) {
        for (int i = 0; i < s->nbins; i++) {
            if (s->lrt_avg[i] >= 65300) continue;
            const int32_t e = wmul(s->lrt_avg[i], 23637) >> 14;
            int16_t ip = (int16_t)(e >> 12);
            if (ip < -8) ip = -8;
            const int16_t fr = (int16_t)(e & 0xfff);
            int32_t p = (fr * fr * 44) >> 19;
            p += (fr * 84) >> 7;
            int32_t inv = wadd(wshl(1, 8 + ip), shift32(p, ip - 4));
            const int n1 = norm_w32(inv), n2 = norm_w16((int16_t)(16384 - s->prior_nonspeech));
            if (n1 + n2 < 7) continue;
            if (n1 + n2 < 15) {
                inv >>= 15 - n2 - n1;
                inv = shift32(wmul(inv, 16384 - s->prior_nonspeech), 7 - n1 - n2);
            } else {
                inv = wmul(inv, 16384 - s->prior_nonspeech) >> 8;
            }
            nonspeech[i] = (uint16_t)(((int32_t)s->prior_nonspeech << 8) / wadd(s->prior_nonspeech, inv));
        }
    }
}

/* ---------------------------------------------------------------- analysis */
/* DataAnalysis, nsx_core.c:1184-1419 */
static void data_analysis(orc_nsx_core *s, const int16_t *speech, uint16_t *magn)
{
    int16_t win[ORC_NSX_ANA] = {0}, norm[ORC_NSX_ANA], spec[ORC_NSX_ANA + 2];
    const int keep = s->ana - s->block, h = s->ana2;
    memmove(s->ana_buf, s->ana_buf + s->block, sizeof(int16_t) * (size_t)keep);
    memcpy(s->ana_buf + keep, speech, sizeof(int16_t) * (size_t)s->block);
    for (int i = 0; i < s->ana; i++) win[i] = (int16_t)mul_rsft_round(s->window[i], s->ana_buf[i], 14);
    s->energy_in = energy(win, s->ana, &s->scale_energy_in);
    s->zero_input = 0;
    const int16_t mx = max_abs16(win, s->ana);
    s->norm_data = norm_w16(mx);
    if (mx == 0) {
        s->zero_input = 1;
        return;
    }
    const int net_norm = s->stages - s->norm_data;
    int rs_magn = s->norm_data - s->min_norm;
    const int rs_init = -rs_magn > 0 ? -rs_magn : 0;
    s->min_norm -= rs_init;
    if (rs_magn < 0) rs_magn = 0;
    for (int i = 0; i < s->ana; i++) norm[i] = (int16_t)wshl(win[i], s->norm_data);
    orc_spl_real_fft(s->stages, norm, spec);

    s->im[0] = 0, s->im[h] = 0;
    s->re[0] = spec[0], s->re[h] = spec[s->ana];
    s->magn_energy = (uint32_t)(s->re[0] * s->re[0]);
    s->magn_energy += (uint32_t)(s->re[h] * s->re[h]);
    magn[0] = (uint16_t)(s->re[0] >= 0 ? s->re[0] : -s->re[0]);
    magn[h] = (uint16_t)(s->re[h] >= 0 ? s->re[h] : -s->re[h]);
    s->sum_magn = (uint32_t)magn[0] + (uint32_t)magn[h];
    const int startup = s->block_index < 50;  /* END_STARTUP_SHORT; block_index is the PREVIOUS block's here */
    int32_t sum_log = 0, sum_ilog = 0;
    if (startup) {
        s->init_magn[0] >>= rs_init;
        s->init_magn[h] >>= rs_init;
        s->init_magn[0] += (uint32_t)(magn[0] >> rs_magn);
        s->init_magn[h] += (uint32_t)(magn[h] >> rs_magn);
        const int16_t l2 = (int16_t)(magn[h] ? log2_q8(magn[h]) : 0);
        sum_log = l2;
        sum_ilog = (fx_nsx_log_index[h] * l2) >> 3;
    }
    for (int i = 1; i < h; i++) {
        s->re[i] = spec[2 * i];
        s->im[i] = (int16_t)-spec[2 * i + 1];
        uint32_t e = (uint32_t)(spec[2 * i] * spec[2 * i]);
        e += (uint32_t)(spec[2 * i + 1] * spec[2 * i + 1]);
        s->magn_energy += e;
        magn[i] = (uint16_t)sqrt_floor((int32_t)e);
        s->sum_magn += (uint32_t)magn[i];
        if (startup) {
            s->init_magn[i] >>= rs_init;
            s->init_magn[i] += (uint32_t)(magn[i] >> rs_magn);
            if (i >= 5) {
                const int16_t l2 = (int16_t)(magn[i] ? log2_q8(magn[i]) : 0);
                sum_log += l2;
                sum_ilog += (fx_nsx_log_index[i] * l2) >> 3;
            }
        }
    }
    if (!startup) return;

    /* white-noise level and the pink-noise fit (least squares of log-magnitude over log-frequency), :1330-1417 */
    s->white >>= rs_init;
    uint32_t w = (s->sum_magn * (uint32_t)s->overdrive) >> (s->stages + 8);
    w >>= rs_magn;
    s->white += w;

    int16_t det = fx_nsx_determinant[5], sum_i = fx_nsx_sum_log_index[5], sum_i2 = fx_nsx_sum_sq_log_index[5];
    if (s->fs == 8000) {
        int32_t t = det;
        t += (fx_nsx_sum_log_index[65] * sum_i) >> 9;
        t -= (fx_nsx_sum_log_index[65] * fx_nsx_sum_log_index[65]) >> 10;
        t -= (int32_t)sum_i2 << 4;
        t -= ((s->nbins - 5) * fx_nsx_sum_sq_log_index[65]) >> 2;
        det = (int16_t)t;
        sum_i = (int16_t)(sum_i - fx_nsx_sum_log_index[65]);
        sum_i2 = (int16_t)(sum_i2 - fx_nsx_sum_sq_log_index[65]);
    }
    int zeros = 16 - norm_w32(sum_log);
    if (zeros < 0) zeros = 0;
    const uint16_t sum_log_u16 = (uint16_t)(wshl(sum_log, 1) >> zeros);
    int32_t num = (int32_t)sum_i2 * sum_log_u16;
    uint32_t ilog = (uint32_t)(sum_ilog >> 12);
    uint16_t si = (uint16_t)((uint16_t)sum_i << 1);
    if ((uint32_t)sum_i > ilog)
        si = (uint16_t)(si >> zeros);
    else
        ilog >>= zeros;
    num = wsub(num, (int32_t)(ilog * (uint32_t)si));
    det = (int16_t)(det >> zeros);
    num = div_w32_w16(num, det);
    num = wadd(num, wshl(net_norm, 11));
    if (num < 0) num = 0;
    s->pink_num = wadd(s->pink_num, num);

    int32_t ex = (int32_t)sum_i * sum_log_u16;
    int32_t t = sum_ilog >> (3 + zeros);
    t = wmul(t, s->nbins - 5);
    ex = wsub(ex, t);
    if (ex > 0) {
        const int32_t q = div_w32_w16(ex, det);
        s->pink_exp += q > 16384 ? 16384 : (q < 0 ? 0 : q);
    }
}

/* ---------------------------------------------------------------- synthesis */
/* DataSynthesis, nsx_core.c:1421-1499 */
static void data_synthesis(orc_nsx_core *s, int16_t *out)
{
    const int keep = s->ana - s->block, h = s->ana2;
    if (!s->zero_input) {
        int16_t spec[ORC_NSX_ANA + 2], td[ORC_NSX_ANA];
        for (int i = 0; i < s->nbins; i++) {
            s->re[i] = (int16_t)((s->re[i] * (int16_t)s->filt[i]) >> 14);
            s->im[i] = (int16_t)((s->im[i] * (int16_t)s->filt[i]) >> 14);
        }
        for (int i = 0; i <= h; i++) spec[2 * i] = s->re[i], spec[2 * i + 1] = (int16_t)-s->im[i];
        const int sc = orc_spl_real_ifft(s->stages, spec, td);
        for (int i = 0; i < s->ana; i++) s->re[i] = sat16(shift32((int32_t)td[i], sc - s->norm_data));

        int16_t gain = 8192;
        if (s->gain_map == 1 && s->block_index > 200 && s->energy_in > 0) {
            int sc_out = 0;
            int32_t e_out = energy(s->re, s->ana, &sc_out);
            if (sc_out == 0 && !(e_out & 0x7f800000))
                e_out = shift32(e_out, 8 + sc_out - s->scale_energy_in);
            else
                /* the count is 8 + sc_out - sc_in >= -1; -1 (sc_in = 9, a near-full-scale block whose output peak fell
                 * below 1448) is an undefined shift in the reference, which then trips its assert(energyIn > 0): the
                 * restatement and the kernel leave the gain at 1.0 there instead of aborting */
                s->energy_in = 8 + sc_out - s->scale_energy_in >= 0 ? s->energy_in >> (8 + sc_out - s->scale_energy_in) : 0;
            if (s->energy_in > 0) {
                int16_t ratio = (int16_t)(wadd(e_out, s->energy_in / 2) / s->energy_in);
                ratio = ratio > 256 ? 256 : (ratio < 0 ? 0 : ratio);
                const int16_t g1 = fx_nsx_factor1[ratio], g2 = s->factor2[ratio];
                const int16_t a = (int16_t)(((16384 - s->prior_nonspeech) * g1) >> 14), b = (int16_t)((s->prior_nonspeech * g2) >> 14);
                gain = (int16_t)(a + b);
            }
        }
        for (int i = 0; i < s->ana; i++) {
            const int16_t w = (int16_t)mul_rsft_round(s->window[i], s->re[i], 14);
            const int16_t g = sat16(mul_rsft_round(w, gain, 13));
            s->syn_buf[i] = sat16((int32_t)s->syn_buf[i] + g);
        }
    }
    memcpy(out, s->syn_buf, sizeof(int16_t) * (size_t)s->block);
    memmove(s->syn_buf, s->syn_buf + s->block, sizeof(int16_t) * (size_t)keep);
    memset(s->syn_buf + keep, 0, sizeof(int16_t) * (size_t)s->block);
}

static void high_band_shift(orc_nsx_core *s, int b, const int16_t *in)
{
    const int keep = s->ana - s->block;
    memmove(s->hb[b], s->hb[b] + s->block, sizeof(int16_t) * (size_t)keep);
    memcpy(s->hb[b] + keep, in, sizeof(int16_t) * (size_t)s->block);
}

/* ---------------------------------------------------------------- ProcessCore, nsx_core.c:1501-2116 */
void orc_nsx_core_process(orc_nsx_core *s, const int16_t *const *in, int num_bands, int16_t *const *out)
{
    uint16_t magn[ORC_NSX_BINS], prev_noise16[ORC_NSX_BINS], nonspeech[ORC_NSX_BINS], filt_tmp[ORC_NSX_BINS];
    uint32_t noise[ORC_NSX_BINS], post_snr[ORC_NSX_BINS], prior_snr[ORC_NSX_BINS], prev_near[ORC_NSX_BINS];
    int16_t q_noise;
    const int n_hb = num_bands > 1 ? num_bands - 1 : 0;

    data_analysis(s, in[0], magn);
    if (s->zero_input) {
        data_synthesis(s, out[0]);
        for (int b = 0; b < n_hb; b++) {
            high_band_shift(s, b, in[1 + b]);
            memcpy(out[1 + b], s->hb[b], sizeof(int16_t) * (size_t)s->block);
        }
        return;
    }
    s->block_index++;
    const int16_t q_magn = (int16_t)(s->norm_data - s->stages);
    spectral_flatness(s, magn);
    noise_estimation(s, magn, noise, &q_noise);
    for (int i = 0; i < s->nbins; i++) prev_noise16[i] = (uint16_t)(s->prev_noise[i] >> 11);

    if (s->block_index < 50) {
        /* start-up: blend the quantile estimate with a white / pink parametric model, :1596-1709 */
        const int qd = (int)q_noise < s->min_norm - s->stages ? (int)q_noise : s->min_norm - s->stages;
        int16_t exp_avg = 0;
        int32_t num_avg = 0;
        uint32_t est = 0, est_avg = 0;
        if (s->pink_exp) {
            exp_avg = (int16_t)div_w32_w16(s->pink_exp, (int16_t)(s->block_index + 1));
            num_avg = div_w32_w16(s->pink_num, (int16_t)(s->block_index + 1));
            parametric_noise(s, exp_avg, num_avg, 5, &est, &est_avg);
        } else {
            est = s->white;
            est_avg = est / (uint32_t)(s->block_index + 1);
        }
        for (int i = 0; i < s->nbins; i++) {
            if (s->pink_exp && i >= 5) {
                est = 0, est_avg = 0;
                parametric_noise(s, exp_avg, num_avg, i, &est, &est_avg);
            }
            filt_tmp[i] = s->denoise_bound;
            if (s->init_magn[i]) {
                const uint32_t od = est * (uint32_t)s->overdrive;
                uint32_t numer = s->init_magn[i] << 8;
                if (numer > od) {
                    numer -= od;
                    int sh = norm_u32(numer);
                    sh = sh > 6 ? 6 : (sh < 0 ? 0 : sh);
                    numer <<= sh;
                    uint32_t den = s->init_magn[i] >> (6 - sh);
                    if (den == 0) den = 1;
                    const uint32_t q = numer / den;
                    filt_tmp[i] = (uint16_t)(q > 16384 ? 16384 : (q < (uint32_t)s->denoise_bound ? (uint32_t)s->denoise_bound : q));
                }
            }
            uint32_t a = noise[i] >> (q_noise - qd);
            uint32_t b = est_avg >> (s->min_norm - s->stages - qd);
            int sh = 0;
            if (a & 0xfc000000) a >>= 6, b >>= 6, sh = 6;
            a *= (uint32_t)s->block_index;
            b *= (uint32_t)(50 - s->block_index);
            noise[i] = div_u32_u16(a + b, 50);
            noise[i] <<= sh;
        }
        q_noise = (int16_t)qd;
    }
    if (s->block_index < 200) {
        s->time_avg_energy_tmp += s->magn_energy >> ((2 * s->norm_data + s->stages - 1) & 31); /* as above */
        s->time_avg_energy = div_u32_u16(s->time_avg_energy_tmp, (uint16_t)(s->block_index + 1));
    }

    /* step 1: decision-directed prior / post SNR from the quantile estimate, :1722-1782 */
    const uint32_t sat_max = 1048575;
    int post_shifts = 6 + q_magn - q_noise, n_shifts = 5 - s->prev_q_magn + s->prev_q_noise;
    for (int i = 0; i < s->nbins; i++) {
        post_snr[i] = 2048;
        uint32_t m = (uint32_t)magn[i] << 6;
        const uint32_t nz = post_shifts < 0 ? noise[i] >> -post_shifts : noise[i] << post_shifts;
        if (m > nz) {
            m <<= 11;
            if (nz > 0) {
                m /= nz;
                post_snr[i] = sat_max < m ? sat_max : m;
            } else {
                post_snr[i] = sat_max;
            }
        }
        const uint32_t near_est = (uint32_t)(s->prev_magn[i] * s->filt[i]);
        uint32_t a = near_est << 3;
        const uint32_t b = s->prev_noise[i] >> (n_shifts & 31); /* negative counts happen; x86 takes them modulo 32 */
        if (b > 0) {
            a /= b;
            a = sat_max < a ? sat_max : a;
        } else {
            a = sat_max;
        }
        prev_near[i] = a;
        const uint32_t p = prev_near[i] * (uint32_t)2007 + (post_snr[i] - 2048) * (uint32_t)41 + 512;
        prior_snr[i] = 2048 + (p >> 10);
    }

    /* step 2: speech / noise likelihood, :1784-1838 */
    spectral_difference(s, magn);
    s->cnt_thr++;
    const int flag = s->cnt_thr == s->model_update;
    feature_parameters(s, flag);
    if (flag) {
        s->cnt_thr = 0;
        s->cur_avg_energy >>= 9;
        const uint32_t mean = (s->cur_avg_energy + s->time_avg_energy + 1) >> 1;
        if (mean != s->time_avg_energy && s->feat_diff && s->time_avg_energy > 0) {
            int nrm = 0;
            uint32_t a = mean, b = s->feat_diff;
            while (0xFFFF0000 & a) a >>= 1, nrm++;
            while (0xFFFF0000 & b) b >>= 1, nrm++;
            uint32_t p = a * b;
            p /= s->time_avg_energy;
            if (norm_u32(p) < nrm)
                s->feat_diff = 0x007FFFFF;
            else
                s->feat_diff = 0x007FFFFF < p << nrm ? 0x007FFFFF : p << nrm;
        }
        s->time_avg_energy = mean;
        s->cur_avg_energy = 0;
    }
    speech_noise_prob(s, nonspeech, prior_snr, post_snr);

    /* noise update, :1840-1945 */
    uint16_t gamma = 26;
    uint32_t max_noise = 0;
    post_shifts = s->prev_q_noise - q_magn;
    n_shifts = s->prev_q_magn - q_magn;
    for (int i = 0; i < s->nbins; i++) {
        const uint32_t m = post_shifts < 0 ? (uint32_t)(magn[i] >> -post_shifts) : (uint32_t)magn[i] << post_shifts;
        int sign;
        uint32_t d;
        if (prev_noise16[i] > m)
            sign = -1, d = prev_noise16[i] - m;
        else
            sign = 1, d = m - prev_noise16[i];
        uint32_t upd = s->prev_noise[i], dp = 0;
        if (d && nonspeech[i]) {
            dp = d * (uint32_t)nonspeech[i];
            const uint32_t st = (0x7c000000 & dp) ? (dp >> 5) * gamma : (dp * gamma) >> 5;
            upd = sign > 0 ? upd + st : upd - st;
        }
        const uint16_t prev_gamma = gamma;
        gamma = nonspeech[i] < 205 ? 3 : 26;
        if (prev_gamma != gamma) {
            const uint32_t st = (0x7c000000 & dp) ? (dp >> 5) * gamma : (dp * gamma) >> 5;
            const uint32_t alt = sign > 0 ? s->prev_noise[i] + st : s->prev_noise[i] - st;
            if (upd > alt) upd = alt;
        }
        noise[i] = upd;
        if (upd > max_noise) max_noise = upd;

        int32_t pz = shift32(s->pause[i], -n_shifts);
        if (nonspeech[i] > 205) {
            int32_t t;
            if (n_shifts < 0) {
                t = wsub((int32_t)magn[i], pz);
                t = wmul(t, 13);
                t = wadd(t, 128) >> 8;
            } else {
                t = wsub(wshl((int32_t)magn[i], n_shifts), s->pause[i]);
                t = wmul(t, 13);
                t = wadd(t, wshl(128, n_shifts)) >> (8 + n_shifts);
            }
            pz = wadd(pz, t);
        }
        s->pause[i] = pz;
    }
    const int norm_max = norm_u32(max_noise);
    q_noise = (int16_t)(s->prev_q_noise + norm_max - 5);

    /* step 3: Wiener gain from the updated noise, :1947-2013 */
    n_shifts = s->prev_q_noise + 11 - q_magn;
    for (int i = 0; i < s->nbins; i++) {
        uint32_t cur = 0, m, nz;
        if (n_shifts < 0) {
            m = (uint32_t)magn[i];
            nz = noise[i] << -n_shifts;
        } else if (n_shifts > 17) {
            m = (uint32_t)magn[i] << 17;
            nz = noise[i] >> (n_shifts - 17);
        } else {
            m = (uint32_t)magn[i] << n_shifts;
            nz = noise[i];
        }
        if (m > nz) {
            uint32_t a = m - nz;
            int nr = norm_u32(a);
            if (nr > 11) nr = 11;
            a <<= nr;
            const uint32_t b = nz >> (11 - nr);
            if (b > 0) a /= b;
            cur = sat_max < a ? sat_max : a;
        }
        const uint32_t prior = prev_near[i] * (uint32_t)2007 + cur * (uint32_t)41;
        const uint32_t den = s->overdrive + ((prior + 8192) >> 14);
        const uint16_t g = (uint16_t)((prior + den / 2) / den);
        s->filt[i] = g > 16384 ? 16384 : (g < s->denoise_bound ? s->denoise_bound : g);
        if (s->block_index < 50) {
            uint32_t a = (uint32_t)(s->filt[i] * s->block_index);
            a += (uint32_t)(filt_tmp[i] * (50 - s->block_index));
            s->filt[i] = (uint16_t)div_u32_u16(a, 50);
        }
    }
    s->prev_q_noise = q_noise;
    s->prev_q_magn = q_magn;
    for (int i = 0; i < s->nbins; i++) {
        s->prev_noise[i] = norm_max > 5 ? noise[i] << (norm_max - 5) : noise[i] >> (5 - norm_max);
        s->prev_magn[i] = magn[i];
    }
    data_synthesis(s, out[0]);

    /* high bands: time-domain gain from the top quarter of the low band, :2026-2115 */
    if (n_hb > 0) {
        for (int b = 0; b < n_hb; b++) high_band_shift(s, b, in[1 + b]);
        uint32_t gsum = 0;
        uint16_t psum = 0;
        for (int i = s->ana2 - (s->ana2 >> 2); i < s->ana2; i++) {
            psum = (uint16_t)(psum + nonspeech[i]);
            gsum += (uint32_t)s->filt[i];
        }
        const int16_t avg_prob = (int16_t)(4096 - (psum >> (s->stages - 7)));
        const int16_t avg_gain = (int16_t)(gsum >> (s->stages - 3));
        const int16_t gmod = avg_prob < 3607 ? avg_prob : 3607;
        int16_t g;
        if (avg_prob < 2048) {
            g = (int16_t)((gmod << 1) + (avg_gain >> 1));
        } else {
            g = (int16_t)((3 * avg_gain) >> 2);
            g = (int16_t)(g + gmod);
        }
        g = g > 16384 ? 16384 : (g < (int16_t)s->denoise_bound ? (int16_t)s->denoise_bound : g);
        for (int b = 0; b < n_hb; b++)
            for (int j = 0; j < s->block; j++) out[1 + b][j] = (int16_t)((g * s->hb[b][j]) >> 14);
    }
}

/* ---------------------------------------------------------------- the wrapper with MAKE_WEBRTC_NSX, src/webrtc.c:560-661 */
orc_nsx *orc_nsx_init(int chn, int freq)
{
    if (freq > 32000 || freq % 8000 != 0) return NULL;
    orc_nsx *h = (orc_nsx *)calloc(1, sizeof(orc_nsx));
    if (!h) return NULL;
    if (orc_nsx_core_init(&h->core, freq, 2) != 0) {
        free(h);
        return NULL;
    }
    h->chn = chn;
    h->freq = freq;
    h->pkg = freq / 1000 * 10;
    return h;
}

void orc_nsx_run(orc_nsx *h, const int16_t *frame, int16_t *frame_out, int frame_num)
{
    const int total = frame_num * h->chn, per = h->pkg * h->chn;
    const int16_t *ip[2] = {h->in[0], h->in[1]};
    int16_t *op[2] = {h->out[0], h->out[1]};
    for (int c = 0; c < total; c += per) {
        for (int p = 0; p < h->pkg; p++)
            for (int ch = 0; ch < h->chn; ch++) h->in[ch][p] = *frame++;
        orc_nsx_core_process(&h->core, ip, h->chn, op);  /* chn passed as num_bands, SURVEY quirk 2 */
        for (int p = 0; p < h->pkg; p++)
            for (int ch = 0; ch < h->chn; ch++) *frame_out++ = h->out[ch][p];
    }
}

void orc_nsx_release(orc_nsx *h) { free(h); }

int orc_run_nsx(int chn, int freq, const int16_t *in, int16_t *out, int frames_per_call, int n_calls)
{
    orc_nsx *h = orc_nsx_init(chn, freq);
    if (!h) return -100;
    const size_t step = (size_t)frames_per_call * (size_t)chn;
    if (out != in) memcpy(out, in, step * (size_t)n_calls * sizeof(int16_t));
    for (int i = 0; i < n_calls; i++) orc_nsx_run(h, out + i * step, out + i * step, frames_per_call);
    orc_nsx_release(h);
    return 0;
}
