/* oracle/orc_agc.c -- TEST INFRASTRUCTURE ONLY (CPU checker; never on the product path).
 *
 * Restatement of wmix's automatic gain control (adaptive-digital mode, target 0 dBFS,
 * limiter off):
 *   agc_init / agc_process / agc_addition / agc_release   src/webrtc.c:694-860
 *   WebRtcAgc_Init / set_config / UpdateAgcThresholds     W:modules/audio_processing/agc/legacy/analog_agc.c:1361-1533,1231-1286,424-472
 *   WebRtcAgc_Process                                     W:...analog_agc.c:1134-1229
 *   WebRtcAgc_CalculateGainTable / InitDigital / InitVad  W:...digital_agc.c:61-282,606-631
 *   WebRtcAgc_ProcessDigital / WebRtcAgc_ProcessVad       W:...digital_agc.c:294-604,633-771
 *   WebRtcSpl_DownsampleBy2, WebRtcSpl_Sqrt, NormU32/W32  W:common_audio/signal_processing/{resample_by_2.c:70-124,spl_sqrt.c}
 *
 * WebRtcAgc_ProcessAnalog (analog_agc.c:639-1132) also runs in the reference, but with
 * inMicLevel = 0 it only updates analog-side bookkeeping whose outputs wmix discards and it
 * cannot fail; it is not restated.  tests/test_agc_oracle.py pins this
 * file bit-exact against oracle/_ref, which does run it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "orc_agc.h"
#include "orc_vad.h" /* orc_norm_w32 / orc_norm_u32 / orc_div_w32_w16 */

static int16_t sat16(int32_t v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : (int16_t)v); }
static int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static int32_t wneg(int32_t a) { return (int32_t)(0u - (uint32_t)a); } /* -a; INT32_MIN stays INT32_MIN, as on the reference's CPU */
static int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static int32_t wshl(int32_t a, int s) { return (int32_t)((uint32_t)a << s); }
static int32_t shift_w32(int32_t x, int c) { return c >= 0 ? wshl(x, c) : (x >> (-c)); }

/* digital_agc.h:24 AGC_SCALEDIFF32(A,B,C) = C + (B>>16)*A + (((0xFFFF & B)*A) >> 16), all int32 */
static int32_t scalediff32(int32_t A, int32_t B, int32_t C)
{
    return wadd(wadd(C, wmul(B >> 16, A)), wmul(0x0000FFFF & B, A) >> 16);
}
/* digital_agc.h:22 AGC_MUL32(A,B) = (B>>13)*A + (((0x1FFF & B)*A) >> 13) */
static int32_t mul32(int32_t A, int32_t B) { return wadd(wmul(B >> 13, A), wmul(0x00001FFF & B, A) >> 13); }

/* signal_processing_library.h:78 WEBRTC_SPL_SCALEDIFF32: the low half is multiplied as uint32 */
static int32_t spl_scalediff32(int32_t A, int32_t B, int32_t C)
{
    return (int32_t)((uint32_t)C + (uint32_t)wmul(B >> 16, A) + (((uint32_t)(0x0000FFFF & B) * (uint32_t)A) >> 16));
}

/* resample_by_2.c:70-124 (generic C path) */
static void downsample_by2(const int16_t *in, int len, int16_t *out, int32_t *st)
{
    static const uint16_t ap1[3] = {3284, 24441, 49528}, ap2[3] = {12199, 37471, 60255};
    for (int i = len >> 1; i > 0; i--) {
        int32_t in32 = wshl((int32_t)(*in++), 10), diff, t1, t2;
        diff = wsub(in32, st[1]);
        t1 = spl_scalediff32(ap2[0], diff, st[0]);
        st[0] = in32;
        diff = wsub(t1, st[2]);
        t2 = spl_scalediff32(ap2[1], diff, st[1]);
        st[1] = t1;
        diff = wsub(t2, st[3]);
        st[3] = spl_scalediff32(ap2[2], diff, st[2]);
        st[2] = t2;
        in32 = wshl((int32_t)(*in++), 10);
        diff = wsub(in32, st[5]);
        t1 = spl_scalediff32(ap1[0], diff, st[4]);
        st[4] = in32;
        diff = wsub(t1, st[6]);
        t2 = spl_scalediff32(ap1[1], diff, st[5]);
        st[5] = t1;
        diff = wsub(t2, st[7]);
        st[7] = spl_scalediff32(ap1[2], diff, st[6]);
        st[6] = t2;
        *out++ = sat16(wadd(wadd(st[3], st[7]), 1024) >> 11);
    }
}

/* spl_sqrt.c: WebRtcSpl_SqrtLocal */
static int32_t sqrt_local(int32_t in)
{
    int32_t B = in / 2, A, x2;
    B = wsub(B, 0x40000000);
    int16_t x_half = (int16_t)(B >> 16), t16;
    B = wadd(B, 0x40000000);
    B = wadd(B, 0x40000000);
    x2 = wmul(wmul(x_half, x_half), 2);
    A = wneg(x2);
    B = wadd(B, A >> 1);
    A >>= 16;
    A = wmul(wmul(A, A), 2);
    t16 = (int16_t)(A >> 16);
    B = wadd(B, wmul(-20480 * t16, 2));
    A = wmul(x_half * t16, 2);
    t16 = (int16_t)(A >> 16);
    B = wadd(B, wmul(28672 * t16, 2));
    t16 = (int16_t)(x2 >> 16);
    A = wmul(x_half * t16, 2);
    B = wadd(B, A >> 1);
    B = wadd(B, 32768);
    return B;
}

/* spl_sqrt.c: WebRtcSpl_Sqrt */
int32_t orc_spl_sqrt(int32_t value)
{
    int32_t A = value;
    if (A == 0) return 0;
    int16_t sh = (int16_t)orc_norm_w32(A);
    A = wshl(A, sh);
    if (A < (0x7FFFFFFF - 32767))
        A = A + 32768;
    else
        A = 0x7FFFFFFF;
    int16_t x_norm = (int16_t)(A >> 16), nshift = (int16_t)(sh / 2), t16;
    A = wshl((int32_t)x_norm, 16);
    A = A >= 0 ? A : wneg(A);
    A = sqrt_local(A);
    if (2 * nshift == sh) {
        t16 = (int16_t)(A >> 16);
        A = wmul(23170 * t16, 2);
        A = wadd(A, 32768);
        A = A & 0x7fff0000;
        A >>= 15;
    } else {
        A >>= 16;
    }
    A = A & 0x0000ffff;
    A >>= nshift;
    return A;
}

/* digital_agc.c:606-631 */
static void vad_init(orc_agc_vad *v)
{
    memset(v, 0, sizeof(*v));
    v->mean_long = 15 << 10;
    v->var_long = 500 << 8;
    v->mean_short = 15 << 10;
    v->var_short = 500 << 8;
    v->counter = 3;
}

/* digital_agc.c:633-771 */
static int16_t vad_process(orc_agc_vad *v, const int16_t *in, int n)
{
    int32_t nrg = 0, t32, t32b;
    int16_t hp = v->hp_state, buf1[8], buf2[4];
    for (int sub = 0; sub < 10; sub++) {
        if (n == 160) {
            for (int k = 0; k < 8; k++) buf1[k] = (int16_t)(((int32_t)in[2 * k] + (int32_t)in[2 * k + 1]) >> 1);
            in += 16;
            downsample_by2(buf1, 8, buf2, v->down_state);
        } else {
            downsample_by2(in, 8, buf2, v->down_state);
            in += 8;
        }
        for (int k = 0; k < 4; k++) {
            int32_t out = buf2[k] + hp;
            t32 = 600 * out;
            hp = (int16_t)((t32 >> 10) - buf2[k]);
            nrg = wadd(nrg, wmul(out, out) >> 6); /* wraps for |out| > 46 340 (full-scale input), digital_agc.c:633 */
        }
    }
    v->hp_state = hp;
    /* the hand-written count-leading-zeros of digital_agc.c:685-708 (nrg == 0 gives 31) */
    int16_t zeros = nrg == 0 ? 31 : (int16_t)__builtin_clz((uint32_t)nrg);
    int16_t dB = (int16_t)((15 - zeros) * 2048);
    if (v->counter < 250) v->counter++;
    t32 = v->mean_short * 15 + dB;
    v->mean_short = (int16_t)(t32 >> 4);
    t32 = (dB * dB) >> 12;
    t32 += v->var_short * 15;
    v->var_short = t32 / 16;
    t32 = v->mean_short * v->mean_short;
    t32 = wsub(wshl(v->var_short, 12), t32);
    v->std_short = (int16_t)orc_spl_sqrt(t32);
    t32 = v->mean_long * v->counter + dB;
    v->mean_long = (int16_t)(t32 / sat16((int32_t)v->counter + 1));
    t32 = (dB * dB) >> 12;
    t32 += v->var_long * v->counter;
    v->var_long = orc_div_w32_w16(t32, sat16((int32_t)v->counter + 1));
    t32 = v->mean_long * v->mean_long;
    t32 = wsub(wshl(v->var_long, 12), t32);
    v->std_long = (int16_t)orc_spl_sqrt(t32);
    int16_t t16 = 3 << 12;
    t32 = t16 * (int16_t)(dB - v->mean_long);
    t32 = orc_div_w32_w16(t32, v->std_long);
    t32b = (int32_t)v->log_ratio * (uint16_t)(13 << 12);
    t32 += t32b >> 10;
    v->log_ratio = (int16_t)(t32 >> 6);
    if (v->log_ratio > 2048) v->log_ratio = 2048;
    if (v->log_ratio < -2048) v->log_ratio = -2048;
    return v->log_ratio;
}

/* digital_agc.c:61-257.  kGenFuncTable[i] = round(256*log2(1+e^i)) (digital_agc.c:38-56,
 * reproduced exactly by this formula -- checked against the header in the container). */
int orc_agc_gain_table(int32_t *table, int16_t comp_gain_db, int16_t target_dbfs, int limiter, int16_t analog_target)
{
    uint16_t gen[128];
    for (int i = 0; i < 128; i++) gen[i] = (uint16_t)floor(256.0 * log2(1.0 + exp((double)i)) + 0.5);
    const uint16_t kLog10 = 54426, kLog10_2 = 49321, kLogE_1 = 23637;
    const int16_t kCompRatio = 3;
    int16_t limiterOffset = 0;
    int32_t t32 = (comp_gain_db - analog_target) * (kCompRatio - 1);
    int16_t t16 = (int16_t)(analog_target - target_dbfs);
    t16 = (int16_t)(t16 + (int16_t)((t32 + (kCompRatio >> 1)) / kCompRatio));
    int16_t maxGain = t16 > (analog_target - target_dbfs) ? t16 : (int16_t)(analog_target - target_dbfs);
    t32 = maxGain * kCompRatio;
    int16_t zeroGainLvl = comp_gain_db;
    zeroGainLvl = (int16_t)(zeroGainLvl - (int16_t)((t32 + ((kCompRatio - 1) >> 1)) / (kCompRatio - 1)));
    if ((comp_gain_db <= analog_target) && limiter) zeroGainLvl = (int16_t)(zeroGainLvl + (analog_target - comp_gain_db + 1));
    t32 = comp_gain_db * (kCompRatio - 1);
    int16_t diffGain = (int16_t)((t32 + (kCompRatio >> 1)) / kCompRatio);
    if (diffGain < 0 || diffGain >= 128) return -1;
    int16_t limiterLvlX = (int16_t)(analog_target - limiterOffset);
    int16_t limiterIdx = (int16_t)(2 + (int16_t)(((int32_t)limiterLvlX << 13) / (int16_t)(kLog10_2 / 2)));
    int32_t limiterLvl = target_dbfs + (int16_t)((limiterOffset + (kCompRatio >> 1)) / kCompRatio);
    uint16_t constMaxGain = gen[diffGain];
    const int16_t constLinApprox = 22817;
    int32_t den = 20 * (int32_t)constMaxGain;
    for (int16_t i = 0; i < 32; i++) {
        t16 = (int16_t)((kCompRatio - 1) * (i - 1));
        t32 = (int32_t)t16 * kLog10_2 + 1;
        int32_t inLevel = orc_div_w32_w16(t32, kCompRatio);
        inLevel = ((int32_t)diffGain << 14) - inLevel;
        uint32_t absIn = (uint32_t)(inLevel >= 0 ? inLevel : -inLevel);
        uint16_t intPart = (uint16_t)(absIn >> 14), fracPart = (uint16_t)(absIn & 0x3FFF);
        if (intPart + 1 >= 128) return -1; /* the reference reads past kGenFuncTable for gains 187..190 dB: refused */
        uint16_t tU16 = (uint16_t)(gen[intPart + 1] - gen[intPart]);
        uint32_t u1 = (uint32_t)tU16 * fracPart, u2;
        u1 += (uint32_t)gen[intPart] << 14;
        uint32_t logApprox = u1 >> 8;
        if (inLevel < 0) {
            int zeros = orc_norm_u32(absIn), zerosScale = 0;
            if (zeros < 15) {
                u2 = absIn >> (15 - zeros);
                u2 = u2 * kLogE_1;
                if (zeros < 9) {
                    zerosScale = 9 - zeros;
                    u1 >>= zerosScale;
                } else {
                    u2 >>= zeros - 9;
                }
            } else {
                u2 = absIn * kLogE_1;
                u2 >>= 6;
            }
            logApprox = 0;
            if (u2 < u1) logApprox = (u1 - u2) >> (8 - zerosScale);
        }
        int32_t numFIX = wshl(maxGain * constMaxGain, 6);
        numFIX = wsub(numFIX, wmul((int32_t)logApprox, diffGain));
        int zeros;
        if (numFIX > (den >> 8))
            zeros = orc_norm_w32(numFIX);
        else
            zeros = orc_norm_w32(den) + 8;
        numFIX = wshl(numFIX, zeros);
        int32_t d = shift_w32(den, zeros - 8);
        if (numFIX < 0) /* the reference lets this wrap (digital_agc.c:196-200) */
            numFIX = wsub(numFIX, d / 2);
        else
            numFIX = wadd(numFIX, d / 2);
        int32_t y32 = numFIX / d;
        if (limiter && (i < limiterIdx)) {
            t32 = (int32_t)(int16_t)(i - 1) * kLog10_2;
            t32 -= limiterLvl * 16384;
            y32 = orc_div_w32_w16(t32 + 10, 20);
        }
        if (y32 > 39000) {
            t32 = wadd(wmul(y32 >> 1, kLog10), 4096);
            t32 >>= 13;
        } else {
            t32 = wadd(wmul(y32, kLog10), 8192);
            t32 >>= 14;
        }
        t32 += 16 << 14;
        if (t32 > 0) {
            intPart = (uint16_t)(int16_t)(t32 >> 14);
            fracPart = (uint16_t)(t32 & 0x3FFF);
            int32_t t2;
            if ((fracPart >> 13) != 0) {
                t16 = (int16_t)((2 << 14) - constLinApprox);
                t2 = (1 << 14) - fracPart;
                t2 *= t16;
                t2 >>= 13;
                t2 = (1 << 14) - t2;
            } else {
                t16 = (int16_t)(constLinApprox - (1 << 14));
                t2 = (fracPart * t16) >> 13;
            }
            fracPart = (uint16_t)t2;
            table[i] = wadd(wshl(1, intPart), shift_w32(fracPart, intPart - 14));
        } else {
            table[i] = 0;
        }
    }
    return 0;
}

/* analog_agc.c:438-444 (UpdateAgcThresholds, adaptive-digital mode) */
static int16_t analog_target_for(int16_t comp_gain_db)
{
    int16_t t = (int16_t)((5 * comp_gain_db) + 5);
    t = (int16_t)((int32_t)t / 11);
    int16_t a = (int16_t)(4 + t);
    return a < 4 ? 4 : a;
}

/* analog_agc.c:1231-1286 set_config + digital_agc.c:259-282 InitDigital */
int orc_agc_core_set_gain(orc_agc_core *s, int16_t comp_gain_db)
{
    return orc_agc_gain_table(s->gain_table, comp_gain_db, 0, 0, analog_target_for(comp_gain_db));
}

int orc_agc_core_init(orc_agc_core *s, int fs, int16_t comp_gain_db)
{
    memset(s, 0, sizeof(*s));
    s->fs = fs;
    s->capacitor_slow = 134217728;
    s->gain = 65536;
    vad_init(&s->vad_near);
    /* WebRtcAgc_Init first applies the default config (9 dB, target 3, limiter on: analog_agc.c:1500-1507),
     * then agc_init overrides it with set_config; only the last table survives. */
    orc_agc_gain_table(s->gain_table, 9, 3, 1, analog_target_for(9));
    return orc_agc_core_set_gain(s, comp_gain_db);
}

/* digital_agc.c:294-604, num_bands = 1, lowlevelSignal = 0, agcMode = adaptive digital, far-end VAD idle */
int orc_agc_core_process(orc_agc_core *s, const int16_t *in, int16_t *out)
{
    int32_t gains[11], env[10], t32, cur = 0, gain32, delta;
    int16_t L, L2, zeros = 0, frac = 0, decay;
    if (s->fs == 8000) {
        L = 8;
        L2 = 3;
    } else {
        L = 16;
        L2 = 4;
    }
    if (in != out) memcpy(out, in, 10 * L * sizeof(int16_t));
    int16_t logratio = vad_process(&s->vad_near, out, L * 10);
    if (logratio > 1024)
        decay = -65;
    else if (logratio < 0)
        decay = 0;
    else
        decay = (int16_t)(((0 - logratio) * 65) >> 10);
    if (s->vad_near.std_long < 4000)
        decay = 0;
    else if (s->vad_near.std_long < 8096)
        decay = (int16_t)(((s->vad_near.std_long - 4000) * decay) >> 12);
    for (int k = 0; k < 10; k++) {
        int32_t mx = 0;
        for (int n = 0; n < L; n++) {
            int32_t nrg = out[k * L + n] * out[k * L + n];
            if (nrg > mx) mx = nrg;
        }
        env[k] = mx;
    }
    gains[0] = s->gain;
    for (int k = 0; k < 10; k++) {
        s->capacitor_fast = scalediff32(-1000, s->capacitor_fast, s->capacitor_fast);
        if (env[k] > s->capacitor_fast) s->capacitor_fast = env[k];
        if (env[k] > s->capacitor_slow)
            s->capacitor_slow = scalediff32(500, wsub(env[k], s->capacitor_slow), s->capacitor_slow);
        else
            s->capacitor_slow = scalediff32(decay, s->capacitor_slow, s->capacitor_slow);
        cur = s->capacitor_fast > s->capacitor_slow ? s->capacitor_fast : s->capacitor_slow;
        zeros = (int16_t)orc_norm_u32((uint32_t)cur);
        if (cur == 0) zeros = 31;
        t32 = wshl(cur, zeros) & 0x7FFFFFFF;
        frac = (int16_t)(t32 >> 19);
        t32 = wmul(wsub(s->gain_table[zeros - 1], s->gain_table[zeros]), frac);
        gains[k + 1] = wadd(s->gain_table[zeros], t32 >> 12);
    }
    zeros = (int16_t)((zeros << 9) - (frac >> 3));
    int16_t zeros_fast = (int16_t)orc_norm_u32((uint32_t)s->capacitor_fast);
    if (s->capacitor_fast == 0) zeros_fast = 31;
    t32 = wshl(s->capacitor_fast, zeros_fast) & 0x7FFFFFFF;
    zeros_fast = (int16_t)(zeros_fast << 9);
    zeros_fast = (int16_t)(zeros_fast - (int16_t)(t32 >> 22));
    int16_t gate = (int16_t)(1000 + zeros_fast - zeros - s->vad_near.std_short), gain_adj;
    if (gate < 0) {
        s->gate_prev = 0;
    } else {
        t32 = s->gate_prev * 7;
        gate = (int16_t)((gate + t32) >> 3);
        s->gate_prev = gate;
    }
    if (gate > 0) {
        gain_adj = gate < 2500 ? (int16_t)((2500 - gate) >> 5) : 0;
        for (int k = 0; k < 10; k++) {
            if (wsub(gains[k + 1], s->gain_table[0]) > 8388608) {
                t32 = wsub(gains[k + 1], s->gain_table[0]) >> 8;
                t32 = wmul(t32, 178 + gain_adj);
            } else {
                t32 = wmul(wsub(gains[k + 1], s->gain_table[0]), 178 + gain_adj);
                t32 >>= 8;
            }
            gains[k + 1] = wadd(s->gain_table[0], t32);
        }
    }
    for (int k = 0; k < 10; k++) {
        zeros = 10;
        if (gains[k + 1] > 47453132) zeros = (int16_t)(16 - orc_norm_w32(gains[k + 1]));
        gain32 = (gains[k + 1] >> zeros) + 1;
        gain32 = wmul(gain32, gain32);
        while (mul32((env[k] >> 12) + 1, gain32) > shift_w32((int32_t)32767, 2 * (1 - zeros + 10))) {
            if (gains[k + 1] > 8388607)
                gains[k + 1] = (gains[k + 1] / 256) * 253;
            else
                gains[k + 1] = (gains[k + 1] * 253) / 256;
            gain32 = (gains[k + 1] >> zeros) + 1;
            gain32 = wmul(gain32, gain32);
        }
    }
    for (int k = 1; k < 10; k++)
        if (gains[k] > gains[k + 1]) gains[k] = gains[k + 1];
    s->gain = gains[10];
    delta = wshl(wsub(gains[1], gains[0]), 4 - L2);
    gain32 = wshl(gains[0], 4);
    for (int n = 0; n < L; n++) {
        t32 = wmul(out[n], wadd(gain32, 127) >> 7);
        int32_t o = t32 >> 16;
        if (o > 4095)
            out[n] = 32767;
        else if (o < -4096)
            out[n] = -32768;
        else
            out[n] = (int16_t)(wmul(out[n], gain32 >> 4) >> 16);
        gain32 = wadd(gain32, delta);
    }
    for (int k = 1; k < 10; k++) {
        delta = wshl(wsub(gains[k + 1], gains[k]), 4 - L2);
        gain32 = wshl(gains[k], 4);
        for (int n = 0; n < L; n++) {
            out[k * L + n] = (int16_t)(wmul(out[k * L + n], gain32 >> 4) >> 16);
            gain32 = wadd(gain32, delta);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ wmix wrapper, src/webrtc.c:694-860 */
orc_agc *orc_agc_init(int chn, int freq, int interval_ms, int value)
{
    (void)interval_ms;
    if (freq > 32000 || freq % 8000 != 0) return NULL;
    orc_agc *h = calloc(1, sizeof(*h));
    if (orc_agc_core_init(&h->core, freq, (int16_t)value) != 0) { /* set_config failed -> agc_init returns NULL */
        free(h);
        return NULL;
    }
    h->chn = chn;
    h->freq = freq;
    h->pkg = freq / 1000 * (freq <= 16000 ? 10 : 5); /* 5 ms packets at 32 kHz (quirk 4) */
    return h;
}

int orc_agc_run(orc_agc *h, const int16_t *frame, int16_t *frame_out, int frame_num)
{
    int total = frame_num * h->chn, step = h->pkg * h->chn;
    int16_t in[160], out[160];
    for (int done = 0; done < total; done += step) {
        for (int i = 0; i < h->pkg; i++) {
            int32_t acc = 0;
            for (int c = 0; c < h->chn; c++) acc += *frame++;
            in[i] = (int16_t)(acc / h->chn);
        }
        /* WebRtcAgc_Process: fs 8000 needs 80 samples, else 160 (analog_agc.c:1153-1170) */
        if (h->pkg != (h->freq == 8000 ? 80 : 160)) return -1;
        orc_agc_core_process(&h->core, in, out);
        for (int i = 0; i < h->pkg; i++)
            for (int c = 0; c < h->chn; c++) *frame_out++ = out[i];
    }
    return 0;
}

void orc_agc_addition(orc_agc *h, uint8_t value) { orc_agc_core_set_gain(&h->core, (int16_t)value); }
void orc_agc_release(orc_agc *h) { free(h); }

int orc_run_agc(int chn, int freq, int value, const int16_t *in, int16_t *out, int frames_per_call, int n_calls)
{
    orc_agc *h = orc_agc_init(chn, freq, 10, value);
    if (!h) return -100;
    size_t step = (size_t)frames_per_call * chn;
    if (out != in) memcpy(out, in, step * n_calls * sizeof(int16_t));
    int rc = 0;
    for (int i = 0; i < n_calls && rc == 0; i++) rc = orc_agc_run(h, out + i * step, out + i * step, frames_per_call);
    orc_agc_release(h);
    return rc;
}
