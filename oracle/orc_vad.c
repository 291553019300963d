/* oracle/orc_vad.c -- TEST INFRASTRUCTURE ONLY (CPU checker; never on the product path).
 *
 * Restatement of wmix's voice-activity gate:
 *   vad_init / vad_process / vad_release       src/webrtc.c:40-164
 *   WebRtcVad_Process -> CalcVad{32,16,8}khz   W:common_audio/vad/webrtc_vad.c:71-104, vad_core.c:623-674
 *   WebRtcVad_Downsampling, FindMinimum        W:common_audio/vad/vad_sp.c:27-177
 *   WebRtcVad_CalculateFeatures (+filters)     W:common_audio/vad/vad_filterbank.c:41-333
 *   WebRtcVad_GaussianProbability              W:common_audio/vad/vad_gmm.c:30-83
 *   GmmProbability, InitCore, set_mode_core(3) W:common_audio/vad/vad_core.c:124-593
 *   WebRtcSpl_Energy / GetScalingSquare / NormW32 / NormU32 / DivW32W16 (signal_processing)
 * All integer: pinned bit-exact against oracle/_ref and the upstream known-answer values
 * (vad_filterbank_unittest.cc, vad_gmm_unittest.cc, vad_sp_unittest.cc) in tests/test_vad_oracle.py.
 * Signed overflow is made explicit (wrap through uint32_t) wherever the reference relies on
 * two's-complement behaviour.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "orc_vad.h"

#define NCH 6

static int32_t wrap_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static int32_t wrap_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static int32_t wrap_shl(int32_t a, int s) { return (int32_t)((uint32_t)a << s); }

/* spl_inl.h:105-124 / 126-141 */
int orc_norm_w32(int32_t a)
{
    if (a == 0) return 0;
    if (a < 0) a = ~a;
    return a == 0 ? 31 : __builtin_clz((uint32_t)a) - 1;
}
int orc_norm_u32(uint32_t a) { return a == 0 ? 0 : __builtin_clz(a); }

/* division_operations.c:38-47 */
int32_t orc_div_w32_w16(int32_t num, int16_t den) { return den != 0 ? (int32_t)(num / den) : (int32_t)0x7FFFFFFF; }

/* spl_inl.h:88-103 */
static int size_in_bits(uint32_t n) { return n == 0 ? 0 : 32 - __builtin_clz(n); }

/* energy.c:20-39 + get_scaling_square.c:20-47 */
static uint32_t band_energy(const int16_t *v, int len, int *scale)
{
    int nbits = size_in_bits((uint32_t)len);
    int16_t smax = -1;
    for (int i = 0; i < len; i++) {
        int16_t sabs = (int16_t)(v[i] > 0 ? v[i] : -v[i]); /* -(-32768) wraps back to -32768 */
        if (sabs > smax) smax = sabs;
    }
    int t = orc_norm_w32((int32_t)smax * smax);
    int scaling = (smax == 0) ? 0 : ((t > nbits) ? 0 : nbits - t);
    uint32_t en = 0;
    for (int i = 0; i < len; i++) en += (uint32_t)(((int32_t)v[i] * v[i]) >> scaling);
    *scale = scaling;
    return en;
}

/* vad_sp.c:27-54 */
void orc_vad_downsample(const int16_t *in, int16_t *out, int32_t *st, int in_len)
{
    int32_t s1 = st[0], s2 = st[1];
    for (int n = 0; n < (in_len >> 1); n++) {
        int16_t t1 = (int16_t)((s1 >> 1) + ((5243 * in[0]) >> 14));
        s1 = (int32_t)in[0] - ((5243 * t1) >> 12);
        int16_t t2 = (int16_t)((s2 >> 1) + ((1392 * in[1]) >> 14));
        s2 = (int32_t)in[1] - ((1392 * t2) >> 12);
        *out++ = (int16_t)(t1 + t2);
        in += 2;
    }
    st[0] = s1;
    st[1] = s2;
}

/* vad_filterbank.c:83-118 */
static void allpass(const int16_t *in, int n, int16_t coef, int16_t *state, int16_t *out)
{
    int32_t s32 = wrap_shl(*state, 16);
    for (int i = 0; i < n; i++) {
        int32_t t32 = wrap_add(s32, coef * in[0]);
        int16_t t16 = (int16_t)(t32 >> 16);
        *out++ = t16;
        s32 = wrap_shl((int32_t)in[0], 14);
        s32 = wrap_sub(s32, coef * t16);
        s32 = wrap_shl(s32, 1);
        in += 2;
    }
    *state = (int16_t)(s32 >> 16);
}

/* vad_filterbank.c:121-145 */
static void split(const int16_t *in, int len, int16_t *up, int16_t *lo, int16_t *hp, int16_t *lp)
{
    int half = len >> 1;
    allpass(in, half, 20972, up, hp);
    allpass(in + 1, half, 5571, lo, lp);
    for (int i = 0; i < half; i++) {
        int16_t t = hp[i];
        hp[i] = (int16_t)(hp[i] - lp[i]);
        lp[i] = (int16_t)(lp[i] + t);
    }
}

/* vad_filterbank.c:41-80 */
static void highpass(const int16_t *in, int n, int16_t *st, int16_t *out)
{
    for (int i = 0; i < n; i++) {
        int32_t t = 6631 * in[i];
        t += -13262 * st[0];
        t += 6631 * st[1];
        st[1] = st[0];
        st[0] = in[i];
        t -= -7756 * st[2];
        t -= 5620 * st[3];
        st[3] = st[2];
        st[2] = (int16_t)(t >> 14);
        out[i] = st[2];
    }
}

/* vad_filterbank.c:155-243 */
static void log_energy(const int16_t *in, int n, int16_t offset, int16_t *total, int16_t *out)
{
    int rsh = 0;
    uint32_t energy = band_energy(in, n, &rsh);
    if (energy == 0) {
        *out = offset;
        return;
    }
    int norm = 17 - orc_norm_u32(energy);
    int16_t log2e = 14336;
    rsh += norm;
    if (norm < 0)
        energy <<= -norm;
    else
        energy >>= norm;
    log2e = (int16_t)(log2e + (int16_t)((energy & 0x3FFF) >> 4));
    int16_t le = (int16_t)(((24660 * log2e) >> 19) + ((rsh * 24660) >> 9));
    if (le < 0) le = 0;
    *out = (int16_t)(le + offset);
    if (*total <= 10) {
        if (rsh >= 0)
            *total = (int16_t)(*total + 10 + 1);
        else
            *total = (int16_t)(*total + (int16_t)(energy >> -rsh));
    }
}

/* vad_filterbank.c:246-333 */
int16_t orc_vad_features(orc_vad_core *s, const int16_t *in, int len, int16_t *f)
{
    static const int16_t off[6] = {368, 368, 272, 176, 176, 176};
    int16_t total = 0, hp120[120], lp120[120], hp60[60], lp60[60];
    int half = len >> 1, n = half;
    split(in, len, &s->upper_state[0], &s->lower_state[0], hp120, lp120);
    split(hp120, n, &s->upper_state[1], &s->lower_state[1], hp60, lp60);
    n >>= 1;
    log_energy(hp60, n, off[5], &total, &f[5]);
    log_energy(lp60, n, off[4], &total, &f[4]);
    n = half;
    split(lp120, n, &s->upper_state[2], &s->lower_state[2], hp60, lp60);
    n >>= 1;
    log_energy(hp60, n, off[3], &total, &f[3]);
    split(lp60, n, &s->upper_state[3], &s->lower_state[3], hp120, lp120);
    n >>= 1;
    log_energy(hp120, n, off[2], &total, &f[2]);
    split(lp120, n, &s->upper_state[4], &s->lower_state[4], hp60, lp60);
    n >>= 1;
    log_energy(hp60, n, off[1], &total, &f[1]);
    highpass(lp60, n, s->hp_filter_state, hp120);
    log_energy(hp120, n, off[0], &total, &f[0]);
    return total;
}

/* vad_gmm.c:30-83 */
int32_t orc_vad_gauss(int16_t input, int16_t mean, int16_t std, int16_t *delta)
{
    int16_t exp_value = 0;
    int32_t t32 = (int32_t)131072 + (int32_t)(std >> 1);
    int16_t inv_std = (int16_t)orc_div_w32_w16(t32, std);
    int16_t t16 = (int16_t)(inv_std >> 2);
    int16_t inv_std2 = (int16_t)((t16 * t16) >> 2);
    t16 = (int16_t)(input * 8);
    t16 = (int16_t)(t16 - mean);
    *delta = (int16_t)((inv_std2 * t16) >> 10);
    t32 = (*delta * t16) >> 9;
    if (t32 < 22005) {
        t16 = (int16_t)((5909 * t32) >> 12);
        t16 = (int16_t)-t16;
        exp_value = (int16_t)(0x0400 | (t16 & 0x03FF));
        t16 ^= (int16_t)0xFFFF;
        t16 >>= 10;
        t16 += 1;
        exp_value >>= t16;
    }
    return inv_std * exp_value;
}

/* vad_sp.c:59-177 */
int16_t orc_vad_find_min(orc_vad_core *s, int16_t v, int ch)
{
    int16_t *age = &s->index_vector[ch << 4], *low = &s->low_value_vector[ch << 4];
    int pos = -1;
    for (int i = 0; i < 16; i++) {
        if (age[i] != 100) {
            age[i]++;
        } else {
            for (int j = i; j < 15; j++) { /* the reference also reads [j+1] at j == 15 and overwrites it below */
                low[j] = low[j + 1];
                age[j] = age[j + 1];
            }
            age[15] = 101;
            low[15] = 10000;
        }
    }
    /* the reference's hand-unrolled binary search == first index whose value is larger */
    if (v < low[15]) {
        pos = 0;
        while (!(v < low[pos])) pos++;
    }
    if (pos > -1) {
        for (int i = 15; i > pos; i--) {
            low[i] = low[i - 1];
            age[i] = age[i - 1];
        }
        low[pos] = v;
        age[pos] = 1;
    }
    int16_t median = 1600, alpha = 0;
    if (s->frame_counter > 2)
        median = low[2];
    else if (s->frame_counter > 0)
        median = low[0];
    if (s->frame_counter > 0) alpha = (median < s->mean_value[ch]) ? 6553 : 32439;
    int32_t t = (alpha + 1) * s->mean_value[ch];
    t += (32767 - alpha) * median;
    t += 16384;
    s->mean_value[ch] = (int16_t)(t >> 15);
    return s->mean_value[ch];
}

static const int16_t kNoiseW[12] = {34, 62, 72, 66, 53, 25, 94, 66, 56, 62, 75, 103};
static const int16_t kSpeechW[12] = {48, 82, 45, 87, 50, 47, 80, 46, 83, 41, 78, 81};
static const int16_t kNoiseMeans0[12] = {6738, 4892, 7065, 6715, 6771, 3369, 7646, 3863, 7820, 7266, 5020, 4362};
static const int16_t kSpeechMeans0[12] = {8306, 10085, 10078, 11823, 11843, 6309, 9473, 9571, 10879, 7581, 8180, 7483};
static const int16_t kNoiseStds0[12] = {378, 1064, 493, 582, 688, 593, 474, 697, 475, 688, 421, 455};
static const int16_t kSpeechStds0[12] = {555, 505, 567, 524, 585, 1231, 509, 828, 492, 1540, 1079, 850};
static const int16_t kSpecW[6] = {6, 8, 10, 12, 14, 16};
static const int16_t kMinDiff[6] = {544, 544, 576, 576, 576, 576};
static const int16_t kMaxSpeech[6] = {11392, 11392, 11520, 11520, 11520, 11520};
static const int16_t kMaxNoise[6] = {9216, 9088, 8960, 8832, 8704, 8576};
static const int16_t kMinMean[2] = {640, 768};

/* vad_core.c:108-118 */
static int32_t weighted_avg(int16_t *data, int16_t offset, const int16_t *w)
{
    int32_t acc = 0;
    for (int k = 0; k < 2; k++) {
        data[k * NCH] = (int16_t)(data[k * NCH] + offset);
        acc += data[k * NCH] * w[k * NCH];
    }
    return acc;
}

/* vad_core.c:124-479; mode 3 thresholds vad_core.c:88-91 */
static int16_t gmm(orc_vad_core *s, int16_t *feat, int16_t total_power, int frame_len)
{
    static const int16_t oh1[3] = {6, 3, 2}, oh2[3] = {9, 5, 3}, loc[3] = {94, 94, 94}, glob[3] = {1100, 1050, 1100};
    int idx = frame_len == 80 ? 0 : (frame_len == 160 ? 1 : 2);
    int16_t vadflag = 0, dN[12], dS[12], ngpr[12] = {0}, sgpr[12] = {0};
    int32_t sum_llr = 0;
    if (total_power > 10) {
        for (int c = 0; c < NCH; c++) {
            int32_t h0t = 0, h1t = 0, np[2], sp[2];
            for (int k = 0; k < 2; k++) {
                int g = c + k * NCH;
                np[k] = kNoiseW[g] * orc_vad_gauss(feat[c], s->noise_means[g], s->noise_stds[g], &dN[g]);
                h0t += np[k];
                sp[k] = kSpeechW[g] * orc_vad_gauss(feat[c], s->speech_means[g], s->speech_stds[g], &dS[g]);
                h1t += sp[k];
            }
            int16_t sh0 = (int16_t)orc_norm_w32(h0t), sh1 = (int16_t)orc_norm_w32(h1t);
            if (h0t == 0) sh0 = 31;
            if (h1t == 0) sh1 = 31;
            int16_t llr = (int16_t)(sh0 - sh1);
            sum_llr += (int32_t)(llr * kSpecW[c]);
            if ((llr * 4) > loc[idx]) vadflag = 1;
            int16_t h0 = (int16_t)(h0t >> 12);
            if (h0 > 0) {
                int32_t t = wrap_shl((int32_t)(np[0] & 0xFFFFF000), 2);
                ngpr[c] = (int16_t)orc_div_w32_w16(t, h0);
                ngpr[c + NCH] = (int16_t)(16384 - ngpr[c]);
            } else {
                ngpr[c] = 16384;
            }
            int16_t h1 = (int16_t)(h1t >> 12);
            if (h1 > 0) {
                int32_t t = wrap_shl((int32_t)(sp[0] & 0xFFFFF000), 2);
                sgpr[c] = (int16_t)orc_div_w32_w16(t, h1);
                sgpr[c + NCH] = (int16_t)(16384 - sgpr[c]);
            }
        }
        vadflag |= (sum_llr >= glob[idx]);
        int16_t maxspe = 12800;
        for (int c = 0; c < NCH; c++) {
            int16_t fmin = orc_vad_find_min(s, feat[c], c);
            int32_t ngm = weighted_avg(&s->noise_means[c], 0, &kNoiseW[c]);
            int16_t t1 = (int16_t)(ngm >> 6);
            for (int k = 0; k < 2; k++) {
                int g = c + k * NCH;
                int16_t nmk = s->noise_means[g], smk = s->speech_means[g], nsk = s->noise_stds[g], ssk = s->speech_stds[g];
                int16_t nmk2 = nmk, t16;
                if (!vadflag) {
                    int16_t delt = (int16_t)((ngpr[g] * dN[g]) >> 11);
                    nmk2 = (int16_t)(nmk + (int16_t)((delt * 655) >> 22));
                }
                int16_t ndelt = (int16_t)((fmin << 4) - t1);
                int16_t nmk3 = (int16_t)(nmk2 + (int16_t)((ndelt * 154) >> 9));
                t16 = (int16_t)((k + 5) << 7);
                if (nmk3 < t16) nmk3 = t16;
                t16 = (int16_t)((72 + k - c) << 7);
                if (nmk3 > t16) nmk3 = t16;
                s->noise_means[g] = nmk3;
                if (vadflag) {
                    int16_t delt = (int16_t)((sgpr[g] * dS[g]) >> 11);
                    t16 = (int16_t)((delt * 6554) >> 21);
                    int16_t smk2 = (int16_t)(smk + ((t16 + 1) >> 1));
                    int16_t maxmu = (int16_t)(maxspe + 640);
                    if (smk2 < kMinMean[k]) smk2 = kMinMean[k];
                    if (smk2 > maxmu) smk2 = maxmu;
                    s->speech_means[g] = smk2;
                    t16 = (int16_t)((smk + 4) >> 3);
                    t16 = (int16_t)(feat[c] - t16);
                    int32_t a = (dS[g] * t16) >> 3;
                    int32_t b = a - 4096;
                    t16 = (int16_t)(sgpr[g] >> 2);
                    a = t16 * b;
                    b = a >> 4;
                    if (b > 0) {
                        t16 = (int16_t)orc_div_w32_w16(b, (int16_t)(ssk * 10));
                    } else {
                        t16 = (int16_t)orc_div_w32_w16(-b, (int16_t)(ssk * 10));
                        t16 = (int16_t)-t16;
                    }
                    t16 = (int16_t)(t16 + 128);
                    ssk = (int16_t)(ssk + (t16 >> 8));
                    if (ssk < 384) ssk = 384;
                    s->speech_stds[g] = ssk;
                } else {
                    t16 = (int16_t)(feat[c] - (nmk >> 3));
                    int32_t a = (dN[g] * t16) >> 3;
                    a -= 4096;
                    t16 = (int16_t)((ngpr[g] + 2) >> 2);
                    int32_t b = (int32_t)((uint32_t)(int32_t)t16 * (uint32_t)a); /* wraps in the reference (vad_core.c:395) */
                    a = b >> 14;
                    if (a > 0) {
                        t16 = (int16_t)orc_div_w32_w16(a, nsk);
                    } else {
                        t16 = (int16_t)orc_div_w32_w16(-a, nsk);
                        t16 = (int16_t)-t16;
                    }
                    t16 = (int16_t)(t16 + 32);
                    nsk = (int16_t)(nsk + (t16 >> 6));
                    if (nsk < 384) nsk = 384;
                    s->noise_stds[g] = nsk;
                }
            }
            ngm = weighted_avg(&s->noise_means[c], 0, &kNoiseW[c]);
            int32_t sgm = weighted_avg(&s->speech_means[c], 0, &kSpeechW[c]);
            int16_t diff = (int16_t)((int16_t)(sgm >> 9) - (int16_t)(ngm >> 9));
            if (diff < kMinDiff[c]) {
                int16_t t16 = (int16_t)(kMinDiff[c] - diff);
                int16_t u1 = (int16_t)((13 * t16) >> 2), u2 = (int16_t)((3 * t16) >> 2);
                sgm = weighted_avg(&s->speech_means[c], u1, &kSpeechW[c]);
                ngm = weighted_avg(&s->noise_means[c], (int16_t)-u2, &kNoiseW[c]);
            }
            maxspe = kMaxSpeech[c];
            int16_t t2 = (int16_t)(sgm >> 7);
            if (t2 > maxspe) {
                t2 = (int16_t)(t2 - maxspe);
                for (int k = 0; k < 2; k++) s->speech_means[c + k * NCH] = (int16_t)(s->speech_means[c + k * NCH] - t2);
            }
            t2 = (int16_t)(ngm >> 7);
            if (t2 > kMaxNoise[c]) {
                t2 = (int16_t)(t2 - kMaxNoise[c]);
                for (int k = 0; k < 2; k++) s->noise_means[c + k * NCH] = (int16_t)(s->noise_means[c + k * NCH] - t2);
            }
        }
        s->frame_counter++;
    }
    if (!vadflag) {
        if (s->over_hang > 0) {
            vadflag = (int16_t)(2 + s->over_hang);
            s->over_hang--;
        }
        s->num_of_speech = 0;
    } else {
        s->num_of_speech++;
        if (s->num_of_speech > 6) {
            s->num_of_speech = 6;
            s->over_hang = oh2[idx];
        } else {
            s->over_hang = oh1[idx];
        }
    }
    return vadflag;
}

/* vad_core.c:482-531 */
void orc_vad_core_init(orc_vad_core *s)
{
    memset(s, 0, sizeof(*s));
    for (int i = 0; i < 12; i++) {
        s->noise_means[i] = kNoiseMeans0[i];
        s->speech_means[i] = kSpeechMeans0[i];
        s->noise_stds[i] = kNoiseStds0[i];
        s->speech_stds[i] = kSpeechStds0[i];
    }
    for (int i = 0; i < 96; i++) s->low_value_vector[i] = 10000;
    for (int i = 0; i < NCH; i++) s->mean_value[i] = 1600;
}

/* webrtc_vad.c:71-104 + vad_core.c:623-674.  frame_len in samples at fs. returns 0/1 or -1. */
int orc_vad_core_process(orc_vad_core *s, int fs, const int16_t *frame, int frame_len)
{
    int16_t wb[480], nb[240], feat[NCH];
    int ms10 = fs / 100;
    if ((fs != 8000 && fs != 16000 && fs != 32000) || (frame_len != ms10 && frame_len != 2 * ms10 && frame_len != 3 * ms10))
        return -1;
    const int16_t *p = frame;
    int len = frame_len;
    if (fs == 32000) {
        orc_vad_downsample(p, wb, &s->ds_state[2], len);
        len /= 2;
        p = wb;
    }
    if (fs >= 16000) {
        orc_vad_downsample(p, nb, &s->ds_state[0], len);
        len /= 2;
        p = nb;
    }
    int16_t total = orc_vad_features(s, p, len, feat);
    int v = gmm(s, feat, total, len);
    return v > 0 ? 1 : v;
}

/* ------------------------------------------------------------------ wmix wrapper, src/webrtc.c:40-164 */
orc_vad *orc_vad_init(int chn, int freq, int interval_ms)
{
    if (freq > 32000 || freq % 8000 != 0) return NULL;
    orc_vad *h = calloc(1, sizeof(*h));
    orc_vad_core_init(&h->core);
    h->chn = chn;
    h->freq = freq;
    h->interval_ms = (freq <= 16000 && interval_ms % 20 == 0) ? 20 : 10;
    h->pkg = freq / 1000 * h->interval_ms;
    h->reduce = 4;
    return h;
}

void orc_vad_run(orc_vad *h, int16_t *frame, int frame_num)
{
    int real = frame_num * h->chn, n = real;
    if (h->chn > 1) {
        int i = 0;
        n = 0;
        while (i < real) {
            int32_t acc = 0;
            for (int c = 0; c < h->chn; c++) acc += frame[i++];
            frame[n++] = (int16_t)(acc / h->chn);
        }
    }
    for (int done = 0; done < n; done += h->pkg) {
        /* quirk 1: always packet 0 */
        int r = orc_vad_core_process(&h->core, h->freq, frame, h->pkg);
        if (r < 0) return;
        if (r == 0) {
            if (h->reduce < 4) h->reduce += 1;
        } else {
            if (h->reduce > 0) h->reduce -= 1;
        }
        for (int i = done; i < h->pkg; i++) frame[i] = (int16_t)(frame[i] >> h->reduce);
    }
    if (h->chn > 1) {
        int i = real - 1;
        for (n -= 1; i >= 0; n--)
            for (int c = 0; c < h->chn; c++) frame[i--] = frame[n];
    }
}

void orc_vad_release(orc_vad *h) { free(h); }

int orc_run_vad(int chn, int freq, int interval_ms, const int16_t *in, int16_t *out, int frames_per_call, int n_calls)
{
    orc_vad *h = orc_vad_init(chn, freq, interval_ms);
    if (!h) return -100;
    size_t step = (size_t)frames_per_call * chn;
    if (out != in) memcpy(out, in, step * n_calls * sizeof(int16_t));
    for (int i = 0; i < n_calls; i++) orc_vad_run(h, out + i * step, frames_per_call);
    orc_vad_release(h);
    return 0;
}
