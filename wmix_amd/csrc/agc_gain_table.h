// agc_gain_table.h -- host side of the batched AGC: the 32-entry Q16 gain table WebRtcAgc_CalculateGainTable builds at
// agc_init / agc_addition time (W:modules/audio_processing/agc/legacy/digital_agc.c:61-257) and the analog target level of
// UpdateAgcThresholds (analog_agc.c:438-444).  Integer arithmetic only, no HIP: agc.hip includes it, and so does the
// sanitizer driver tools_dev/san/host_ctl_san.cpp (ASan + UBSan over every compression gain).
#pragma once
#include <cmath>
#include <cstdint>

namespace wmx {
namespace {

// ---- host: WebRtcAgc_CalculateGainTable (digital_agc.c:61-257), integer arithmetic only ----
inline int32_t h_wshl(int32_t a, int s) { return (int32_t)((uint32_t)a << s); }
inline int32_t h_wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
inline int32_t h_shift(int32_t x, int c) { return c >= 0 ? h_wshl(x, c) : (x >> (-c)); }
inline int h_norm_w32(int32_t a) {
    if (a == 0) return 0;
    if (a < 0) a = ~a;
    return a == 0 ? 31 : __builtin_clz((uint32_t)a) - 1;
}
inline int h_norm_u32(uint32_t a) { return a == 0 ? 0 : __builtin_clz(a); }
inline int32_t h_div(int32_t num, int16_t den) { return den != 0 ? num / den : 0x7FFFFFFF; }

int host_gain_table(int32_t *table, int16_t comp_db, int16_t target_dbfs, bool limiter, int16_t analog_target) {
    // kGenFuncTable[i] = round(256 * log2(1 + e^i)) (digital_agc.c:38-56)
    uint16_t gen[128];
    for (int i = 0; i < 128; i++) gen[i] = (uint16_t)floor(256.0 * log2(1.0 + exp((double)i)) + 0.5);
    const uint16_t kLog10 = 54426, kLog10_2 = 49321, kLogE_1 = 23637;
    const int16_t kCompRatio = 3, constLinApprox = 22817;
    const int16_t limiterOffset = 0;
    int32_t t32 = (comp_db - analog_target) * (kCompRatio - 1);
    int16_t t16 = (int16_t)(analog_target - target_dbfs);
    t16 = (int16_t)(t16 + (int16_t)((t32 + (kCompRatio >> 1)) / kCompRatio));
    const int16_t maxGain = t16 > (analog_target - target_dbfs) ? t16 : (int16_t)(analog_target - target_dbfs);
    t32 = comp_db * (kCompRatio - 1);
    const int16_t diffGain = (int16_t)((t32 + (kCompRatio >> 1)) / kCompRatio);
    if (diffGain < 0 || diffGain >= 128) return -1;
    const int16_t limiterLvlX = (int16_t)(analog_target - limiterOffset);
    const int16_t limiterIdx = (int16_t)(2 + (int16_t)(((int32_t)limiterLvlX << 13) / (int16_t)(kLog10_2 / 2)));
    const int32_t limiterLvl = target_dbfs + (int16_t)((limiterOffset + (kCompRatio >> 1)) / kCompRatio);
    const uint16_t constMaxGain = gen[diffGain];
    const int32_t den = 20 * (int32_t)constMaxGain;
    for (int16_t i = 0; i < 32; i++) {
        t16 = (int16_t)((kCompRatio - 1) * (i - 1));
        t32 = (int32_t)t16 * kLog10_2 + 1;
        int32_t inLevel = h_div(t32, kCompRatio);
        inLevel = ((int32_t)diffGain << 14) - inLevel;
        const uint32_t absIn = (uint32_t)(inLevel >= 0 ? inLevel : -inLevel);
        uint16_t intPart = (uint16_t)(absIn >> 14), fracPart = (uint16_t)(absIn & 0x3FFF);
        // compression gains 187..190 dB pass the reference's diffGain < 128 test and then index kGenFuncTable[128..130]: it
        // reads past its table (digital_agc.c:133-136).  Refused here (and in the oracle) instead of reproduced.
        if (intPart + 1 >= 128) return -1;
        const uint16_t tU16 = (uint16_t)(gen[intPart + 1] - gen[intPart]);
        uint32_t u1 = (uint32_t)tU16 * fracPart, u2;
        u1 += (uint32_t)gen[intPart] << 14;
        uint32_t logApprox = u1 >> 8;
        if (inLevel < 0) {
            const int zeros = h_norm_u32(absIn);
            int zerosScale = 0;
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
        int32_t numFIX = h_wshl(maxGain * constMaxGain, 6);
        numFIX = (int32_t)((uint32_t)numFIX - (uint32_t)h_wmul((int32_t)logApprox, diffGain));
        int zeros;
        if (numFIX > (den >> 8))
            zeros = h_norm_w32(numFIX);
        else
            zeros = h_norm_w32(den) + 8;
        numFIX = h_wshl(numFIX, zeros);
        const int32_t d = h_shift(den, zeros - 8);
        // the reference lets this wrap (digital_agc.c:196-200); spelled out: signed overflow is undefined, and UBSan says so
        if (numFIX < 0)
            numFIX = (int32_t)((uint32_t)numFIX - (uint32_t)(d / 2));
        else
            numFIX = (int32_t)((uint32_t)numFIX + (uint32_t)(d / 2));
        int32_t y32 = numFIX / d;
        if (limiter && (i < limiterIdx)) {
            t32 = (int32_t)(int16_t)(i - 1) * kLog10_2;
            t32 -= limiterLvl * 16384;
            y32 = h_div(t32 + 10, 20);
        }
        if (y32 > 39000) {
            t32 = (int32_t)((uint32_t)h_wmul(y32 >> 1, kLog10) + 4096u);
            t32 >>= 13;
        } else {
            t32 = (int32_t)((uint32_t)h_wmul(y32, kLog10) + 8192u);
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
            table[i] = (int32_t)((uint32_t)h_wshl(1, intPart) + (uint32_t)h_shift(fracPart, intPart - 14));
        } else {
            table[i] = 0;
        }
    }
    return 0;
}

// analog_agc.c:438-444 (UpdateAgcThresholds, adaptive-digital mode)
int16_t analog_target_for(int16_t comp_db) {
    int16_t t = (int16_t)((5 * comp_db) + 5);
    t = (int16_t)((int32_t)t / 11);
    const int16_t a = (int16_t)(4 + t);
    return a < 4 ? (int16_t)4 : a;
}

}  // namespace
}  // namespace wmx
