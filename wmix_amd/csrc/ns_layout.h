// ns_layout.h -- HBM layout of one stream's noise-suppressor state (32-bit words).
//
// One contiguous block of WORDS floats per stream; every per-bin array is padded to MP
// (a multiple of 4 words) so it starts 16-byte aligned and a wave reads it with 256-byte
// coalesced accesses.  Field names cite NoiseSuppressionC (W:.../ns/ns_core.h:52-114).
// The three feature histograms (3 x 1000 counters, touched 3 times per frame and scanned
// every 500 frames) live in a separate uint16 buffer.
#pragma once

namespace wmx {

template <int L>  // L = anaLen: 128 (8 kHz) or 256 (16 / 32 kHz)
struct NsLayout {
    static constexpr int M = L / 2 + 1;            // magnLen
    static constexpr int MP = (M + 3) & ~3;        // padded bin-array length
    static constexpr int B = (L == 128) ? 80 : 160;  // blockLen
    static constexpr int SLOTS = (M + 63) / 64;    // bins per lane

    // sliding time-domain buffers
    static constexpr int IN_BUF = 0;               // analyzeBuf == dataBuf (see ns.hip header)
    static constexpr int SYNT_BUF = IN_BUF + L;    // syntBuf
    static constexpr int HB_BUF = SYNT_BUF + L;    // dataBufHB[0] (2-channel streams only)
    // per-bin arrays
    static constexpr int DENSITY = HB_BUF + L;             // [3][MP]
    static constexpr int LQUANTILE = DENSITY + 3 * MP;     // [3][MP]
    static constexpr int QUANTILE = LQUANTILE + 3 * MP;
    static constexpr int SMOOTH = QUANTILE + MP;
    static constexpr int NOISE_PREV = SMOOTH + MP;
    static constexpr int MAGN_PREV = NOISE_PREV + MP;      // magnPrevAnalyze == magnPrevProcess
    static constexpr int LOG_LRT = MAGN_PREV + MP;         // logLrtTimeAvg
    static constexpr int MAGN_AVG_PAUSE = LOG_LRT + MP;
    static constexpr int INIT_MAGN = MAGN_AVG_PAUSE + MP;  // initMagnEst
    // scalars (ints stored as their bit patterns)
    static constexpr int SCALARS = INIT_MAGN + MP;
    static constexpr int S_COUNTER = SCALARS + 0;      // counter[3]
    static constexpr int S_UPDATES = SCALARS + 3;
    static constexpr int S_BLOCK_IND = SCALARS + 4;
    static constexpr int S_UPDATE_FLAG = SCALARS + 5;  // modelUpdatePars[0]
    static constexpr int S_COUNTDOWN = SCALARS + 6;    // modelUpdatePars[3]
    static constexpr int S_THR_LRT = SCALARS + 7;      // priorModelPars[0]
    static constexpr int S_THR_FLAT = SCALARS + 8;     // priorModelPars[1]
    static constexpr int S_THR_DIFF = SCALARS + 9;     // priorModelPars[3]
    static constexpr int S_W_LRT = SCALARS + 10;       // priorModelPars[4]
    static constexpr int S_W_FLAT = SCALARS + 11;      // priorModelPars[5]
    static constexpr int S_W_DIFF = SCALARS + 12;      // priorModelPars[6]
    static constexpr int S_PRIOR = SCALARS + 13;       // priorSpeechProb
    static constexpr int S_FEAT_FLAT = SCALARS + 14;   // featureData[0]
    static constexpr int S_FEAT_LRT = SCALARS + 15;    // featureData[3]
    static constexpr int S_FEAT_DIFF = SCALARS + 16;   // featureData[4]
    static constexpr int S_FEAT_NORM = SCALARS + 17;   // featureData[5]
    static constexpr int S_FEAT_ACC = SCALARS + 18;    // featureData[6]
    static constexpr int S_WHITE = SCALARS + 19;       // whiteNoiseLevel
    static constexpr int S_PINK_NUM = SCALARS + 20;    // pinkNoiseNumerator
    static constexpr int S_PINK_EXP = SCALARS + 21;    // pinkNoiseExp
    static constexpr int WORDS = SCALARS + 24;
};

}  // namespace wmx
