// nsx.hip -- batched FIXED-POINT noise suppressor for gfx950: one wavefront per stream.
//
// What the reference runs when src/webrtc.c is built with its MAKE_WEBRTC_NSX switch (src/webrtc.c:512-521):
// ns_process() hands every 10 ms packet to WebRtcNsx_Process -> WebRtcNsx_ProcessCore
// (W:modules/audio_processing/ns/nsx_core.c:1501-2116; DataAnalysis :1184, DataSynthesis :1421, NoiseEstimationC :334,
// ComputeSpectralFlatness :1022, ComputeSpectralDifference :1091, FeatureParameterExtraction :821;
// SpeechNoiseProb W:.../ns/nsx_core_c.c:26-260) over the SPL radix-2 fixed-point FFT
// (W:common_audio/signal_processing/real_fft.c:46-100, complex_fft.c:30-296 mode 1).  Policy 2, like ns_init sets it.
//
// Mapping.  65 / 129 bins: lane k owns bins k, k+64 (lane 0 also bin 128).  Everything is integer, and every sum over
// bins or samples is a two's-complement sum, so lane partials + a butterfly give the reference's bits in any order --
// there is no ordered / unordered switch here.  A stream's state (5.6 KB at 16 kHz: two 256-sample int16 buffers, seven
// int16 and four int32 per-bin arrays, 25 scalars) is one contiguous block in HBM, pulled into LDS with 16-byte
// accesses at the start of a launch and written back at its end; the 3 x 1000 int16 feature histograms stay in HBM
// (three increments per frame, a full scan every 512 blocks).  The FFT is the reference's own dataflow -- bit reversal,
// then one radix-2 pass per stage with its per-stage rounding and, in the inverse, its data-dependent shift from the
// largest |value| of the whole array -- on packed int16 pairs in LDS, two butterflies per lane and stage.
// Per-stream scalars are wave-uniform (read from the LDS state block through readfirstlane where they are used; results
// of wave reductions likewise), so the start-up phases, the zero-input path and the 512-block threshold update never
// diverge inside a wave.
//
// Integer path: bit-exact against the reference (tests/test_nsx_gpu.py).
#include <vector>
#include "wmx_internal.h"
#include "spl_fx.h"
#include "fx_tables.h"

namespace wmx {
namespace {

// Workgroup shape: waves (= streams) per workgroup, and the occupancy the register budget is set for.  Four waves and
// 32 368 B of LDS per workgroup (16 kHz mono: 7 128 B per stream + 3.8 KB of tables) let five workgroups share a CU's
// 160 KB -- 5 waves per SIMD; the 8 kHz formats fit 8 workgroups and are register-limited to 5 per SIMD.
// A workgroup's waves are dealt round-robin to the CU's four SIMDs starting at the same one, so a wave count that is not a
// multiple of 4 piles the surplus on SIMD 0 and the CU stops accepting workgroups early: 5-, 9- and 10-wave workgroups
// (sized to fill the LDS with 15 / 18 / 20 waves) all ran at 10 waves per CU, 0.65-0.73 ms against 0.455 ms for 4 x 4.
#ifndef WMX_NSX_WPB
#define WMX_NSX_WPB 4
#endif
template <int ANA, int CHN>
struct NsxShape {
    static constexpr int WPB = WMX_NSX_WPB, WPE = ANA == 256 && CHN == 2 ? 4 : 5;  // 16 kHz 2-channel: 34 KB of LDS, 4 workgroups
};
constexpr int kNsxHist = 1000;  // HIST_PAR_EST, nsx_defines.h:45

// ---------------------------------------------------------------- constants (one copy per workgroup in LDS)
struct alignas(16) NsxConsts {
    SplTwiddles tw;  // packed twiddle pairs of the SPL FFT (spl_fx.h), both directions
    int16_t window[256];
    int16_t log_frac[256];
    int16_t counter_div[202];
    int16_t log_index[130];
    int16_t factor1[258];
    int16_t factor2[258];
    int16_t indicator[18];
    int16_t log_table[10];
    // kSumLogIndex / kSumSquareLogIndex / kDeterminantEstMatrix are only ever read at the start band (5) and, for the
    // 129-bin format, at 65 (nsx_core.c:1330-1365): five values instead of three tables
    int16_t sum_log_index_5, sum_log_index_65, sum_sq_log_index_5, sum_sq_log_index_65, determinant_5;
    int16_t pad[5];
};
static_assert(sizeof(NsxConsts) % 16 == 0, "NsxConsts is copied in 16-byte pieces");

// ---------------------------------------------------------------- per-stream state block (int32 words)
template <int ANA>
struct NsxLayout {
    static constexpr int BINS = ANA / 2 + 1, BP = ANA / 2 + 2;
    static constexpr int ANA_BUF = 0;                    // int16[ANA]   analysisBuffer
    static constexpr int SYN_BUF = ANA_BUF + ANA / 2;    // int16[ANA]   synthesisBuffer
    static constexpr int FILT = SYN_BUF + ANA / 2;       // uint16[BP]   noiseSupFilter (Q14)
    static constexpr int LQ = FILT + BP / 2;             // int16[3][BP] noiseEstLogQuantile
    static constexpr int DENS = LQ + 3 * BP / 2;         // int16[3][BP] noiseEstDensity
    static constexpr int QUANT = DENS + 3 * BP / 2;      // int16[BP]    noiseEstQuantile
    static constexpr int PMAGN = QUANT + BP / 2;         // uint16[BP]   prevMagnU16
    static constexpr int LRT = PMAGN + BP / 2;           // int32[BP]    logLrtTimeAvgW32
    static constexpr int PAUSE = LRT + BP;               // int32[BP]    avgMagnPause
    static constexpr int INITM = PAUSE + BP;             // uint32[BP]   initMagnEst
    static constexpr int PNOISE = INITM + BP;            // uint32[BP]   prevNoiseU32
    static constexpr int SCAL = PNOISE + BP;             // 35 scalar words (X_COUNT used)
    static constexpr int HB = SCAL + 35;                 // int16[ANA]   dataBufHBFX[0] (2-channel streams only)
    static constexpr int WORDS_MONO = HB, WORDS_2CH = HB + ANA / 2;
    static_assert(WORDS_MONO % 4 == 0 && WORDS_2CH % 4 == 0, "16-byte state copies");
};
enum NsxScalar {
    X_COUNTER0 = 0, X_COUNTER1, X_COUNTER2, X_QNOISE, X_FEAT_LRT, X_THR_LRT, X_W_LRT, X_W_DIFF, X_W_FLAT, X_FEAT_DIFF, X_THR_DIFF,
    X_FEAT_FLAT, X_THR_FLAT, X_CUR_AVG_E, X_TIME_AVG_E, X_TIME_AVG_E_TMP, X_WHITE, X_PINK_NUM, X_PINK_EXP, X_MIN_NORM, X_PRIOR,
    X_BLOCK_INDEX, X_CNT_THR, X_PREV_QNOISE, X_PREV_QMAGN, X_COUNT
};

template <int ANA, int CHN>
struct alignas(16) NsxWave {
    static constexpr int BP = NsxLayout<ANA>::BP;
    int32_t st[CHN == 2 ? NsxLayout<ANA>::WORDS_2CH : NsxLayout<ANA>::WORDS_MONO];
    // FFT work array, one packed complex (re | im << 16) per word.  Three tenants, never at the same time: the windowed
    // frame td[ANA] (int16, in the first half) until the forward transform's bit-reversed load; the transform itself, whose
    // first ANA/2 + 1 words then stay as the frame's spectrum (inst->real / inst->imag) until the inverse transform's load;
    // and the inverse transform, whose output becomes td[] again (inst->real).  Each hand-over reads into registers, then
    // writes after a wave_sync.
    int32_t cx[ANA];
    __device__ __forceinline__ int16_t *td() { return reinterpret_cast<int16_t *>(cx); }
    uint16_t magn[BP], nsp[BP];  // read across lanes (magn[0], nsp[b - 1]); the other per-bin intermediates are registers
};

__device__ __forceinline__ int16_t log2_q8(const NsxConsts &K, uint32_t v) {  // nsx_core.c:362-370
    const int zeros = norm_u32(v);
    const int frac = (int)(((v << zeros) & 0x7FFFFFFF) >> 23);
    return (int16_t)(((31 - zeros) << 8) + K.log_frac[frac]);
}
// energy.c + get_scaling_square.c over the ANA int16 samples of v (LDS): returns the energy, *scale the right shift used
template <int ANA>
__device__ int32_t wave_energy(const int16_t *v, int lane, int *scale) {
    int smax = -1;
    for (int i = lane; i < ANA; i += 64) {
        const int16_t a = (int16_t)(v[i] > 0 ? v[i] : -v[i]);  // -(-32768) wraps to -32768 like the reference's int16
        smax = a > smax ? a : smax;
    }
    smax = wave_max(smax);
    int sc = 0;
    if (smax != 0) {
        constexpr int nbits = ANA == 256 ? 9 : 8;  // GetSizeInBits(ANA)
        const int t = norm_w32(wmul(smax, smax));
        sc = t > nbits ? 0 : nbits - t;
    }
    uint32_t en = 0;
    for (int i = lane; i < ANA; i += 64) en += (uint32_t)(((int32_t)v[i] * v[i]) >> sc);
    *scale = sc;
    return (int32_t)wave_sum(en);
}

// nsx_core_c.c:104-116 / 137-149 / 185-199 (see oracle/orc_nsx.c indicator())
__device__ __forceinline__ int16_t indicator(const NsxConsts &K, uint32_t x_q14, int positive, int rounded) {
    int16_t ind = (int16_t)(positive ? 16384 : 0);
    const int16_t idx = (int16_t)(x_q14 >> 14);
    if (idx < 16 && idx >= 0) {
        int16_t v = K.indicator[idx];
        const int16_t d = (int16_t)(K.indicator[idx + 1] - K.indicator[idx]), frac = (int16_t)(x_q14 & 0x3fff);
        v = (int16_t)(v + (int16_t)(rounded ? mul_rsft_round(d, frac, 14) : (d * frac) >> 14));
        ind = (int16_t)(positive ? 8192 + v : 8192 - v);
    }
    return ind;
}

// CalcParametricNoiseEstimate, nsx_core.c:586-628 (leaves est / est_avg untouched when the exponent is <= 0)
__device__ __forceinline__ void parametric_noise(const NsxConsts &K, int min_norm, int stages, int block_index, int16_t exp_avg,
                                                 int32_t num_avg, int bin, uint32_t &est, uint32_t &est_avg) {
    int32_t t = num_avg - ((exp_avg * K.log_index[bin]) >> 15);
    t += (min_norm - stages) << 11;
    if (t > 0) {
        const int16_t ip = (int16_t)(t >> 11), fp = (int16_t)(t & 0x7ff);
        int32_t b = (fp >> 10) ? 2048 - (((2048 - fp) * 1244) >> 10) : (fp * 804) >> 10;
        b = shift_w32(b, ip - 11);
        est_avg = (uint32_t)wshl(1, ip) + (uint32_t)b;
        est = est_avg * (uint32_t)(block_index + 1);
    }
}

// the two largest histogram peaks by (count descending, index ascending) -- what the reference's sequential scan with
// strict comparisons keeps (nsx_core.c:913-931) -- and their merge (:935-940).  hist: this stream's 1000 int16 counters.
__device__ __forceinline__ uint32_t hist_word(const int16_t *hist, int w) {
    // the counters are incremented by L2 atomics; read them at L2 as well (a plain load could hit a stale L1 line
    // from the previous window's scan when one launch spans more than 512 blocks)
    return __hip_atomic_load(reinterpret_cast<const uint32_t *>(hist) + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the 500 words of one histogram, eight per lane (word lane + 64 q), requested TOGETHER: the scan is one L2 round trip, not eight
// dependent ones -- it decides how long the slowest wave of a launch lives when the streams of a batch have different ages and
// some wave of every launch is in its 512-block update (tools_dev/ns_age_exp.py nsx: 0.38 -> 0.48 ms per launch before)
// (in two batches of four: eight values at once cost the kernels their register budget -- scratch spills)
constexpr int kNsxHistBatch = 4, kNsxHistBatches = ((kNsxHist / 2 + 63) / 64 + kNsxHistBatch - 1) / kNsxHistBatch;
__device__ __forceinline__ void hist_fetch(const int16_t *hist, int lane, int batch, uint32_t (&pr)[kNsxHistBatch]) {
#pragma unroll
    for (int q = 0; q < kNsxHistBatch; q++) {
        const int w = lane + 64 * (kNsxHistBatch * batch + q);
        const uint32_t v = hist_word(hist, w < kNsxHist / 2 ? w : 0);
        pr[q] = w < kNsxHist / 2 ? v : 0u;  // past the end: empty bins (they add nothing and are never peaks)
    }
}
__device__ void two_peaks(const int16_t *hist, int lane, uint32_t &pos1, int &w1) {
    uint32_t k1 = 0, k2 = 0;  // key = count << 16 | (0xFFFF - index); 0 = no peak (counts of 0 never become peaks)
#pragma unroll 1
    for (int batch = 0; batch < kNsxHistBatches; batch++) {
    uint32_t pr[kNsxHistBatch];
    hist_fetch(hist, lane, batch, pr);
#pragma unroll
    for (int q = 0; q < kNsxHistBatch; q++) {
        const int w = lane + 64 * (kNsxHistBatch * batch + q);
        const uint32_t pair = pr[q];
#pragma unroll
        for (int hlf = 0; hlf < 2; hlf++) {
            const int c = (int16_t)(hlf ? pair >> 16 : pair & 0xffff), i = 2 * w + hlf;
            const uint32_t key = c > 0 ? ((uint32_t)c << 16) | (uint32_t)(0xFFFF - i) : 0u;
            if (key > k1) {
                k2 = k1;
                k1 = key;
            } else if (key > k2) {
                k2 = key;
            }
        }
    }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a1 = (uint32_t)__shfl_xor((int)k1, o, 64), a2 = (uint32_t)__shfl_xor((int)k2, o, 64);
        const uint32_t hi = a1 > k1 ? a1 : k1, lo = a1 > k1 ? k1 : a1, other = a2 > k2 ? a2 : k2;
        k1 = hi;
        k2 = lo > other ? lo : other;
    }
    k1 = (uint32_t)uni((int)k1);
    k2 = (uint32_t)uni((int)k2);
    int wa = (int)(k1 >> 16), wb = (int)(k2 >> 16);
    uint32_t pa = k1 ? 2u * (0xFFFFu - (k1 & 0xFFFFu)) + 1u : 0u, pb = k2 ? 2u * (0xFFFFu - (k2 & 0xFFFFu)) + 1u : 0u;
    if (pa - pb < 4 && wb * 2 > wa) {
        wa += wb;
        pa = (pa + pb) >> 1;
    }
    pos1 = pa;
    w1 = wa;
}

// pass r of a per-bin loop: bin b = lane + 64 r (unrolled: r is a compile-time constant in the body)
#define FOR_BINS(r, b) _Pragma("unroll") for (int r = 0; r < NP; r++) if (const int b = lane + 64 * r; b < BINS)

// ---------------------------------------------------------------- one 10 ms block of one stream (ProcessCore)
// in0 / out0: low band (channel 0), in1 / out1: the "high band" (channel 1 of a 2-channel stream);
// element stride CHN.  sc[]: the stream's scalars in its LDS state block, read wave-uniformly where used (LdsScal, spl_fx.h).
#ifdef WMX_NSX_PROF  // developer build only (make EXTRA=-DWMX_NSX_PROF): cycles per phase of nsx_block, summed over waves
// (one slot per SIMD-sized group of waves and phase, plain stores: atomics on sixteen shared words queued every wave's next
// load behind everybody's counters -- the phase with the first global load read 70 %)
__device__ unsigned long long g_nsx_prof[1024 * 16];
#define NSX_PROF(i)                                                                              \
    do {                                                                                         \
        const long long t_now = clock64();                                                       \
        prof_acc[i] += (unsigned long long)(t_now - prof_t0);                                    \
        prof_t0 = clock64();                                                                     \
    } while (0)
#else
#define NSX_PROF(i)
#endif

template <int ANA, int CHN>
__device__ void nsx_block(NsxWave<ANA, CHN> &W, const NsxConsts &K, const LdsScal sc, int16_t *hist, const int16_t *in, int16_t *out,
                          int lane, int overdrive, int denoise_bound) {
    using Y = NsxLayout<ANA>;
#ifdef WMX_NSX_PROF
    unsigned long long prof_acc[16] = {0};
    long long prof_t0 = clock64();
#endif
    constexpr int BINS = Y::BINS, H = ANA / 2, BLOCK = ANA == 256 ? 160 : 80, KEEP = ANA - BLOCK, STAGES = ANA == 256 ? 8 : 7;
    constexpr int32_t kMaxLrt = ANA == 256 ? 0x0080000 : 0x0040000, kMinLrt = ANA == 256 ? 104858 : 52429;
    int16_t *ana = reinterpret_cast<int16_t *>(&W.st[Y::ANA_BUF]), *syn = reinterpret_cast<int16_t *>(&W.st[Y::SYN_BUF]);
    uint16_t *filt = reinterpret_cast<uint16_t *>(&W.st[Y::FILT]), *pmagn = reinterpret_cast<uint16_t *>(&W.st[Y::PMAGN]);
    int16_t *lq = reinterpret_cast<int16_t *>(&W.st[Y::LQ]), *dens = reinterpret_cast<int16_t *>(&W.st[Y::DENS]);
    int16_t *quant = reinterpret_cast<int16_t *>(&W.st[Y::QUANT]);
    int32_t *lrt = &W.st[Y::LRT], *pause = &W.st[Y::PAUSE];
    uint32_t *initm = reinterpret_cast<uint32_t *>(&W.st[Y::INITM]), *pnoise = reinterpret_cast<uint32_t *>(&W.st[Y::PNOISE]);
    int16_t *hb = reinterpret_cast<int16_t *>(&W.st[Y::HB]);
    constexpr int BP = Y::BP;
    int16_t *td = W.td();
    // the frame's per-bin intermediates stay in registers: pass r of a per-bin loop handles bin lane + 64 r (the last
    // pass is lane 0's Nyquist bin), every loop is unrolled over the passes, so the indices are compile-time constants
    constexpr int NP = (BINS + 63) / 64;
    uint32_t t_noise[NP], t_post[NP], t_prior[NP], t_pnear[NP];
    uint16_t t_ftmp[NP], t_pn16[NP];
    int16_t t_lmagn[NP];
#pragma unroll
    for (int r = 0; r < NP; r++) t_noise[r] = t_post[r] = t_prior[r] = t_pnear[r] = 0, t_ftmp[r] = t_pn16[r] = 0, t_lmagn[r] = 0;

    // ---- DataAnalysis, nsx_core.c:1184-1419: shift in the packet, window (AnalysisUpdateC :524-541)
    {
        int16_t keep[(KEEP + 63) / 64], hkeep[(KEEP + 63) / 64];
#pragma unroll
        for (int r = 0; r < (KEEP + 63) / 64; r++) {
            const int i = lane + 64 * r;
            keep[r] = i < KEEP ? ana[i + BLOCK] : (int16_t)0;
            if (CHN == 2) hkeep[r] = i < KEEP ? hb[i + BLOCK] : (int16_t)0;
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r < (KEEP + 63) / 64; r++) {
            const int i = lane + 64 * r;
            if (i < KEEP) {
                ana[i] = keep[r];
                if (CHN == 2) hb[i] = hkeep[r];
            }
        }
        for (int i = lane; i < BLOCK; i += 64) {
            ana[KEEP + i] = in[(long)i * CHN];
            if (CHN == 2) hb[KEEP + i] = in[(long)i * CHN + 1];
        }
        wave_sync();
    }
    NSX_PROF(11);
    for (int i = lane; i < ANA; i += 64) td[i] = (int16_t)mul_rsft_round(K.window[i], ana[i], 14);
    wave_sync();
    int scale_energy_in;
    int32_t energy_in = wave_energy<ANA>(td, lane, &scale_energy_in);
    int mxabs = 0;
    for (int i = lane; i < ANA; i += 64) {
        const int a = td[i] < 0 ? -(int)td[i] : (int)td[i];
        mxabs = a > mxabs ? a : mxabs;
    }
    mxabs = wave_max(mxabs);
    if (mxabs > 32767) mxabs = 32767;
    const int norm_data = norm_w16((int16_t)mxabs);
    const bool zero_input = mxabs == 0;
    NSX_PROF(12);

    if (!zero_input) {
        const int net_norm = STAGES - norm_data;
        int rs_magn = norm_data - sc[X_MIN_NORM];
        const int rs_init = -rs_magn > 0 ? -rs_magn : 0;
        sc[X_MIN_NORM] -= rs_init;
        if (rs_magn < 0) rs_magn = 0;
        // NormalizeRealBufferC + real_fft.c:46-70: zero imaginary parts, bit reversal, forward transform
        {
            int16_t tv[ANA / 64];
#pragma unroll
            for (int r = 0; r < ANA / 64; r++) tv[r] = td[lane + 64 * r];
            wave_sync();  // td[] lives in cx
#pragma unroll
            for (int r = 0; r < ANA / 64; r++) W.cx[bitrev<STAGES>(lane + 64 * r)] = (int32_t)(uint16_t)(int16_t)wshl(tv[r], norm_data);
        }
        wave_sync();
        if constexpr (STAGES == 7)
            spl_cfft128<false>(W.cx, K.tw, lane);
        else
            spl_cfft256<false>(W.cx, K.tw, lane);
        NSX_PROF(13);
        // spectrum, magnitudes, sums (:1231-1264 / :1266-1328)
        const bool startup = sc[X_BLOCK_INDEX] < 50;  // the previous block's index: it is advanced below
        uint32_t e_sum = 0, m_sum = 0;
        int32_t sum_log = 0, sum_ilog = 0;
        FOR_BINS(r, b) {
            const int32_t x = W.cx[b];
            int16_t re = lo16(x), im = (int16_t)-hi16(x);
            uint32_t e;
            uint16_t mg;
            if (b == 0 || b == H) {
                im = 0;
                e = (uint32_t)(re * re);
                mg = (uint16_t)(re >= 0 ? re : -re);
            } else {
                e = (uint32_t)(re * re) + (uint32_t)(hi16(x) * hi16(x));
                mg = (uint16_t)sqrt_floor((int32_t)e);
            }
            W.cx[b] = pack16(re, im);  // the spectrum stays in place
            W.magn[b] = mg;
            e_sum += e;
            m_sum += mg;
            if (startup) {
                initm[b] = (initm[b] >> rs_init) + (uint32_t)(mg >> rs_magn);
                if (b >= 5) {
                    const int16_t l2 = (int16_t)(mg ? log2_q8(K, mg) : 0);
                    sum_log += l2;
                    sum_ilog += (K.log_index[b] * l2) >> 3;
                }
            }
        }
        const uint32_t magn_energy = wave_sum(e_sum), sum_magn = wave_sum(m_sum);
        if (startup) {
            sum_log = (int32_t)wave_sum((uint32_t)sum_log);
            sum_ilog = (int32_t)wave_sum((uint32_t)sum_ilog);
            // white-noise level and pink-noise fit, :1330-1417
            sc[X_WHITE] = (int32_t)((uint32_t)sc[X_WHITE] >> rs_init);
            uint32_t w = (sum_magn * (uint32_t)overdrive) >> (STAGES + 8);
            w >>= rs_magn;
            sc[X_WHITE] = (int32_t)((uint32_t)sc[X_WHITE] + w);
            int16_t det = K.determinant_5, sum_i = K.sum_log_index_5, sum_i2 = K.sum_sq_log_index_5;
            if (ANA == 128) {
                int32_t t = det;
                t += (K.sum_log_index_65 * sum_i) >> 9;
                t -= (K.sum_log_index_65 * K.sum_log_index_65) >> 10;
                t -= (int32_t)sum_i2 << 4;
                t -= ((BINS - 5) * K.sum_sq_log_index_65) >> 2;
                det = (int16_t)t;
                sum_i = (int16_t)(sum_i - K.sum_log_index_65);
                sum_i2 = (int16_t)(sum_i2 - K.sum_sq_log_index_65);
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
            sc[X_PINK_NUM] = wadd(sc[X_PINK_NUM], num);
            int32_t ex = (int32_t)sum_i * sum_log_u16;
            int32_t t = sum_ilog >> (3 + zeros);
            t = wmul(t, BINS - 5);
            ex = wsub(ex, t);
            if (ex > 0) {
                const int32_t q = div_w32_w16(ex, det);
                sc[X_PINK_EXP] += q > 16384 ? 16384 : (q < 0 ? 0 : q);
            }
        }
        wave_sync();

        NSX_PROF(0);
        // ---- ProcessCore, :1590 onwards
        sc[X_BLOCK_INDEX]++;
        const int block_index = sc[X_BLOCK_INDEX];
        const int16_t q_magn = (int16_t)(norm_data - STAGES);

        // ComputeSpectralFlatness, :1022-1084
        {
            uint32_t num = 0;
            int any_zero = 0;
            FOR_BINS(r, b) {
                if (b >= 1) {
                    const uint16_t mg = W.magn[b];
                    if (mg)
                        num += (uint32_t)log2_q8(K, mg);
                    else
                        any_zero = 1;
                }
            }
            uint32_t feat = (uint32_t)sc[X_FEAT_FLAT];
            if (wave_any(any_zero)) {
                feat -= (feat * (uint32_t)4915) >> 14;
            } else {
                num = wave_sum(num);
                const uint32_t den = sum_magn - (uint32_t)W.magn[0];
                const int zeros = norm_u32(den);
                const int frac = (int)(((den << zeros) & 0x7FFFFFFF) >> 23);
                const int32_t lden = ((31 - zeros) << 8) + K.log_frac[frac];
                int32_t lf = (int32_t)num;
                lf = wadd(lf, wshl(STAGES - 1, STAGES + 7));
                lf = wsub(lf, wshl(lden, STAGES - 1));
                lf = wshl(lf, 10 - STAGES);
                const int32_t mant = 0x00020000 | ((lf >= 0 ? lf : -lf) & 0x0001FFFF);
                const int16_t ip = (int16_t)(7 - (lf >> 17));
                const int32_t cur = ip > 0 ? mant >> ip : wshl(mant, -ip);
                int32_t d = wsub(cur, (int32_t)feat);
                d = wmul(d, 4915);
                feat += (uint32_t)(d >> 14);
            }
            sc[X_FEAT_FLAT] = (int32_t)feat;
        }

        NSX_PROF(1);
        // NoiseEstimationC, :334-453
        int16_t q_noise;
        {
            const int tabind = STAGES - norm_data;
            const int16_t logval = (int16_t)(tabind < 0 ? -K.log_table[-tabind] : K.log_table[tabind]);
            FOR_BINS(r, b) {
                const uint16_t mg = W.magn[b];
                int16_t lm = logval;
                if (mg) {
                    lm = (int16_t)((log2_q8(K, mg) * 22713) >> 15);
                    lm = (int16_t)(lm + logval);
                }
                t_lmagn[r] = lm;
            }
            int update_off = -1;
#pragma unroll
            for (int e = 0; e < 3; e++) {
                const int16_t counter = (int16_t)sc[X_COUNTER0 + e], count_div = K.counter_div[counter];
                const int16_t count_prod = (int16_t)(counter * count_div);
                FOR_BINS(r, b) {
                    int16_t q = lq[e * BP + b], dn = dens[e * BP + b];
                    const int16_t lm = t_lmagn[r];
                    int16_t delta;
                    if (dn > 512)
                        delta = (int16_t)(2621440 >> (14 - norm_w16(dn)));
                    else
                        delta = (int16_t)(block_index < 200 ? 1024 : 5120);
                    int16_t step = (int16_t)((delta * count_div) >> 14);
                    if (lm > q) {
                        step = (int16_t)(step + 2);
                        q = (int16_t)(q + step / 4);
                    } else {
                        step = (int16_t)(step + 1);
                        q = (int16_t)(q - (int16_t)((step / 2) * 3 / 2));
                        if (q < logval) q = logval;
                    }
                    const int d = lm - q;
                    if ((d >= 0 ? d : -d) < 3) {
                        const int16_t a = (int16_t)mul_rsft_round(dn, count_prod, 15), c = (int16_t)mul_rsft_round(21845, count_div, 15);
                        dn = (int16_t)(a + c);
                    }
                    lq[e * BP + b] = q;
                    dens[e * BP + b] = dn;
                }
                if (counter >= 200) {
                    sc[X_COUNTER0 + e] = 0;
                    if (block_index >= 200) update_off = e;
                }
                sc[X_COUNTER0 + e]++;
            }
            if (block_index < 200) update_off = 2;
            if (update_off >= 0) {  // UpdateNoiseEstimate, :303-331 (at most one estimator per block reaches its period)
                const int16_t *q = lq + update_off * BP;
                int32_t mx = -32768;
                FOR_BINS(r, b) mx = q[b] > mx ? q[b] : mx;
                mx = wave_max(mx);
                sc[X_QNOISE] = 14 - (int)mul_rsft_round(11819, (int16_t)mx, 21);
                FOR_BINS(r, b) {
                    const int32_t ee = 11819 * q[b];
                    int32_t m = 0x00200000 | (ee & 0x001FFFFF);
                    int16_t sh = (int16_t)(ee >> 21);
                    sh = (int16_t)(sh - 21);
                    sh = (int16_t)(sh + (int16_t)sc[X_QNOISE]);
                    m = sh < 0 ? m >> -sh : wshl(m, sh);
                    quant[b] = sat_w16(m);
                }
            }
            FOR_BINS(r, b) {
                t_noise[r] = (uint32_t)quant[b];
                t_pn16[r] = (uint16_t)(pnoise[b] >> 11);
            }
            q_noise = (int16_t)sc[X_QNOISE];
        }

        NSX_PROF(2);
        // start-up blend with the white / pink parametric model, :1596-1709
        if (block_index < 50) {
            const int qd = (int)q_noise < sc[X_MIN_NORM] - STAGES ? (int)q_noise : sc[X_MIN_NORM] - STAGES;
            int16_t exp_avg = 0;
            int32_t num_avg = 0;
            uint32_t est0 = 0, est0_avg = 0;
            if (sc[X_PINK_EXP]) {
                exp_avg = (int16_t)div_w32_w16(sc[X_PINK_EXP], (int16_t)(block_index + 1));
                num_avg = div_w32_w16(sc[X_PINK_NUM], (int16_t)(block_index + 1));
                parametric_noise(K, sc[X_MIN_NORM], STAGES, block_index, exp_avg, num_avg, 5, est0, est0_avg);
            } else {
                est0 = (uint32_t)sc[X_WHITE];
                est0_avg = est0 / (uint32_t)(block_index + 1);
            }
            FOR_BINS(r, b) {
                uint32_t est = est0, est_avg = est0_avg;
                if (sc[X_PINK_EXP] && b >= 5) {
                    est = 0, est_avg = 0;
                    parametric_noise(K, sc[X_MIN_NORM], STAGES, block_index, exp_avg, num_avg, b, est, est_avg);
                }
                uint16_t ft = (uint16_t)denoise_bound;
                const uint32_t im = initm[b];
                if (im) {
                    const uint32_t od = est * (uint32_t)overdrive;
                    uint32_t numer = im << 8;
                    if (numer > od) {
                        numer -= od;
                        int sh = norm_u32(numer);
                        sh = sh > 6 ? 6 : sh;
                        numer <<= sh;
                        uint32_t den = im >> (6 - sh);
                        if (den == 0) den = 1;
                        const uint32_t q = numer / den;
                        ft = (uint16_t)(q > 16384 ? 16384 : (q < (uint32_t)denoise_bound ? (uint32_t)denoise_bound : q));
                    }
                }
                t_ftmp[r] = ft;
                uint32_t a = t_noise[r] >> (q_noise - qd);
                uint32_t c = est_avg >> (sc[X_MIN_NORM] - STAGES - qd);
                int sh = 0;
                if (a & 0xfc000000) a >>= 6, c >>= 6, sh = 6;
                a *= (uint32_t)block_index;
                c *= (uint32_t)(50 - block_index);
                t_noise[r] = ((a + c) / 50u) << sh;
            }
            q_noise = (int16_t)qd;
        }
        if (block_index < 200) {
            sc[X_TIME_AVG_E_TMP] = (int32_t)((uint32_t)sc[X_TIME_AVG_E_TMP] + (magn_energy >> ((2 * norm_data + STAGES - 1) & 31)));
            sc[X_TIME_AVG_E] = (int32_t)div_u32_u16((uint32_t)sc[X_TIME_AVG_E_TMP], (uint16_t)(block_index + 1));
        }

        NSX_PROF(3);
        // step 1: decision-directed prior / post SNR, :1722-1782
        const uint32_t sat_max = 1048575;
        {
            const int post_shifts = 6 + q_magn - q_noise, n_shifts = 5 - sc[X_PREV_QMAGN] + sc[X_PREV_QNOISE];
            FOR_BINS(r, b) {
                uint32_t post = 2048;
                uint32_t m = (uint32_t)W.magn[b] << 6;
                const uint32_t nz = post_shifts < 0 ? t_noise[r] >> -post_shifts : t_noise[r] << post_shifts;
                if (m > nz) {
                    m <<= 11;
                    if (nz > 0) {
                        m /= nz;
                        post = sat_max < m ? sat_max : m;
                    } else {
                        post = sat_max;
                    }
                }
                uint32_t a = (uint32_t)(pmagn[b] * filt[b]) << 3;
                const uint32_t c = pnoise[b] >> (n_shifts & 31);  // the count can be negative or > 31: the reference's x86 shift takes it modulo 32, and so does v_lshrrev
                if (c > 0) {
                    a /= c;
                    a = sat_max < a ? sat_max : a;
                } else {
                    a = sat_max;
                }
                t_post[r] = post;
                t_pnear[r] = a;
                const uint32_t p = a * 2007u + (post - 2048) * 41u + 512;
                t_prior[r] = 2048 + (p >> 10);
            }
        }

        NSX_PROF(4);
        // ComputeSpectralDifference, :1091-1181
        {
            int32_t s_p = 0, mx = 0, mn = pause[0];
            FOR_BINS(r, b) {
                const int32_t p = pause[b];
                s_p = wadd(s_p, p);
                mx = p > mx ? p : mx;
                mn = p < mn ? p : mn;
            }
            int32_t avg_pause = (int32_t)wave_sum((uint32_t)s_p);
            mx = wave_max(mx);
            mn = wave_min(mn);
            avg_pause >>= STAGES - 1;
            const int32_t avg_magn = (int32_t)(sum_magn >> (STAGES - 1));
            const int32_t dev = mx - avg_pause > avg_pause - mn ? mx - avg_pause : avg_pause - mn;
            int n_shifts = 10 + STAGES - norm_w32(dev);
            if (n_shifts < 0) n_shifts = 0;
            uint32_t v_m = 0, v_p = 0, cv = 0;
            FOR_BINS(r, b) {
                const int16_t dm = (int16_t)((int32_t)W.magn[b] - avg_magn);
                const int32_t dp = wsub(pause[b], avg_pause);
                v_m += (uint32_t)(dm * dm);
                cv += (uint32_t)wmul(dp, dm);
                const int32_t r = dp >> n_shifts;
                v_p += (uint32_t)wmul(r, r);
            }
            const uint32_t var_magn = wave_sum(v_m);
            uint32_t var_pause = wave_sum(v_p);
            const int32_t cov = (int32_t)wave_sum(cv);
            sc[X_CUR_AVG_E] = (int32_t)((uint32_t)sc[X_CUR_AVG_E] + (magn_energy >> ((2 * norm_data + STAGES - 1) & 31)));
            uint32_t diff = var_magn;
            if (var_pause && cov) {
                uint32_t c = (uint32_t)(cov >= 0 ? cov : -cov);
                const int norm = norm_u32(c) - 16;
                c = norm > 0 ? c << norm : c >> -norm;
                const uint32_t c2 = c * c;
                n_shifts += norm;
                n_shifts <<= 1;
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
            const uint32_t cur = diff >> (2 * norm_data);
            uint32_t fd = (uint32_t)sc[X_FEAT_DIFF];
            if (fd > cur)
                fd -= ((fd - cur) * 77u) >> 8;
            else
                fd += ((cur - fd) * 77u) >> 8;
            sc[X_FEAT_DIFF] = (int32_t)fd;
        }

        NSX_PROF(5);
        // FeatureParameterExtraction, :821-1017
        sc[X_CNT_THR]++;
        const bool flag = sc[X_CNT_THR] == 512;
        if (!flag) {
            if (lane < 3) {
                uint32_t h;
                if (lane == 0) {
                    h = (uint32_t)sc[X_FEAT_LRT];
                } else if (lane == 1) {
                    h = ((uint32_t)sc[X_FEAT_FLAT] * 5) >> 8;
                } else {
                    h = kNsxHist;
                    if ((uint32_t)sc[X_TIME_AVG_E] > 0) h = (((uint32_t)sc[X_FEAT_DIFF] * 5) >> STAGES) / (uint32_t)sc[X_TIME_AVG_E];
                }
                if (h < (uint32_t)kNsxHist) {
                    // counters are int16 pairs in 32-bit words; a window holds at most 511 increments, so no carry
                    uint32_t *word = reinterpret_cast<uint32_t *>(hist + lane * kNsxHist) + (h >> 1);
                    __hip_atomic_fetch_add(word, (h & 1) ? 0x10000u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        } else {
            // The window's increments have landed: they are this wave's own (lanes 0..2, L2 atomics), nobody else touches the
            // stream's histograms, and a wave's memory operations on one address are performed in order -- waiting for its
            // outstanding ones is all it takes.  (An agent-scope acq_rel fence stood here: an L2 write-back per update, 2 ms
            // when every stream of a 65 536-stream batch updated in the same launch.)
            // The immediate is the gfx9 encoding of vmcnt(0) (bits 3:0 and 15:14; expcnt and lgkmcnt left at their maxima), and on
            // gfx9 returning atomics and stores both count in vmcnt.  gfx10+ counts stores in vscnt and lays the fields out
            // differently: this line must not be compiled for anything but gfx9 (round-4 ADVICE).
#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "nsx.hip: s_waitcnt immediate 0x0f70 is the gfx9 encoding of vmcnt(0)"
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // compiler-level ordering edge: nothing moves across the wait
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int16_t *h_lrt = hist, *h_flat = hist + kNsxHist, *h_diff = hist + 2 * kNsxHist;
            int use_diff = 1;
            uint32_t a0 = 0, a1 = 0, a2 = 0, cn = 0;
#pragma unroll 1
            for (int batch = 0; batch < kNsxHistBatches; batch++) {
            uint32_t pr[kNsxHistBatch];
            hist_fetch(h_lrt, lane, batch, pr);
#pragma unroll
            for (int q = 0; q < kNsxHistBatch; q++) {
                const int w = lane + 64 * (kNsxHistBatch * batch + q);
                const uint32_t pair = pr[q];
#pragma unroll
                for (int hlf = 0; hlf < 2; hlf++) {
                    const int i = 2 * w + hlf;
                    const int16_t c = (int16_t)(hlf ? pair >> 16 : pair & 0xffff), j = (int16_t)(2 * i + 1);
                    const int32_t t = c * j;
                    if (i < 10) {
                        a0 += (uint32_t)t;
                        cn += (uint32_t)(int32_t)c;
                    }
                    a1 += (uint32_t)t;
                    a2 += (uint32_t)wmul(t, j);
                }
            }
            }
            const int32_t avg = (int32_t)wave_sum(a0), avg_all = (int32_t)wave_sum(a1), avg_sq = (int32_t)wave_sum(a2);
            const int16_t count = (int16_t)wave_sum(cn);
            const int32_t fluct = wsub(wmul(avg_sq, count), wmul(avg, avg_all)), thr_fluct = 10240 * count;
            const uint32_t six_avg = 6 * (uint32_t)avg;
            if (fluct < thr_fluct || count == 0 || six_avg > (uint32_t)(100 * count)) {
                sc[X_THR_LRT] = kMaxLrt;
            } else {
                const int32_t t = (int32_t)((six_avg << (9 + STAGES)) / (uint32_t)count / 25);
                sc[X_THR_LRT] = t > kMaxLrt ? kMaxLrt : (t < kMinLrt ? kMinLrt : t);
            }
            if (fluct < thr_fluct) use_diff = 0;
            uint32_t pos;
            int weight;
            two_peaks(h_flat, lane, pos, weight);
            int use_flat = 1;
            if (weight < 154 || pos < 24) {
                use_flat = 0;
            } else {
                const uint32_t t = 922 * pos;
                sc[X_THR_FLAT] = (int32_t)(t > 38912 ? 38912 : (t < 4096 ? 4096 : t));
            }
            if (use_diff) {
                two_peaks(h_diff, lane, pos, weight);
                const uint32_t t = 6 * pos;
                sc[X_THR_DIFF] = (int32_t)(t > 100 ? 100 : (t < 16 ? 16 : t));
                if (weight < 154) use_diff = 0;
            }
            const int share = 6 / (1 + use_flat + use_diff);
            sc[X_W_LRT] = share;
            sc[X_W_FLAT] = use_flat * share;
            sc[X_W_DIFF] = use_diff * share;
            uint32_t *hw = reinterpret_cast<uint32_t *>(hist);
            for (int i = lane; i < 3 * kNsxHist / 2; i += 64) hw[i] = 0;
            // the window's normalisation of the spectral difference, :1800-1836
            sc[X_CNT_THR] = 0;
            sc[X_CUR_AVG_E] = (int32_t)((uint32_t)sc[X_CUR_AVG_E] >> 9);
            const uint32_t tavg = (uint32_t)sc[X_TIME_AVG_E];
            const uint32_t mean = ((uint32_t)sc[X_CUR_AVG_E] + tavg + 1) >> 1;
            if (mean != tavg && sc[X_FEAT_DIFF] && tavg > 0) {
                int nrm = 0;
                uint32_t a = mean, c = (uint32_t)sc[X_FEAT_DIFF];
                while (0xFFFF0000 & a) a >>= 1, nrm++;
                while (0xFFFF0000 & c) c >>= 1, nrm++;
                uint32_t p = a * c;
                p /= tavg;
                if (norm_u32(p) < nrm)
                    sc[X_FEAT_DIFF] = 0x007FFFFF;
                else
                    sc[X_FEAT_DIFF] = (int32_t)(0x007FFFFFu < p << nrm ? 0x007FFFFFu : p << nrm);
            }
            sc[X_TIME_AVG_E] = (int32_t)mean;
            sc[X_CUR_AVG_E] = 0;
        }

        NSX_PROF(6);
        // SpeechNoiseProb, nsx_core_c.c:26-260
        {
            uint32_t ls = 0;
            FOR_BINS(r, b) {
                const uint32_t post = t_post[r], prior = t_prior[r];
                int32_t bessel = (int32_t)post;
                const int nt = norm_u32(post);
                const uint32_t num = post << nt;
                const uint32_t den = nt > 10 ? prior << (nt - 11) : prior >> (11 - nt);
                bessel = den > 0 ? wsub(bessel, (int32_t)(num / den)) : 0;
                const int zeros = norm_u32(prior);
                int32_t f = (int32_t)(((prior << zeros) & 0x7FFFFFFF) >> 19);
                int32_t t = wmul(wmul(f, f), -43) >> 19;
                t += ((int16_t)f * 5412) >> 12;
                f = t + 37;
                t = (int32_t)(((31 - zeros) << 12) + f) - (11 << 12);
                const int32_t log_prior = wmul(t, 178) >> 8;
                const int32_t half = wadd(log_prior, lrt[b]) / 2;
                const int32_t nl = wadd(lrt[b], wsub(bessel, half));
                lrt[b] = nl;
                ls += (uint32_t)nl;
            }
            const int32_t lrt_sum = (int32_t)wave_sum(ls);
            sc[X_FEAT_LRT] = wmul(lrt_sum, 10) >> (STAGES + 11);
            int32_t d0 = wsub(lrt_sum, sc[X_THR_LRT]);
            int sh = 7 - STAGES, pos = 1;
            if (d0 < 0) pos = 0, d0 = -d0, sh++;
            d0 = shift_w32(d0, sh);
            int32_t ind_prior = sc[X_W_LRT] * indicator(K, (uint32_t)d0, pos, 0);
            if (sc[X_W_FLAT]) {
                const uint32_t a = (uint32_t)sc[X_FEAT_FLAT] * 400u, thr = (uint32_t)sc[X_THR_FLAT];
                uint32_t d = thr - a;
                sh = 4, pos = 1;
                if (thr < a) pos = 0, d = a - thr, sh++;
                ind_prior += sc[X_W_FLAT] * indicator(K, (d << sh) / 25u, pos, 0);
            }
            if (sc[X_W_DIFF]) {
                uint32_t a = 0;
                const uint32_t fd = (uint32_t)sc[X_FEAT_DIFF];
                if (fd) {
                    int nt = norm_u32(fd);
                    if (20 - STAGES < nt) nt = 20 - STAGES;
                    a = fd << nt;
                    const uint32_t e = (uint32_t)sc[X_TIME_AVG_E] >> (20 - STAGES - nt);
                    a = e > 0 ? a / e : 0x7fffffffu;
                }
                const uint32_t thr = ((uint32_t)sc[X_THR_DIFF] << 17) / 25;
                uint32_t d = a - thr;
                sh = 1, pos = 1;
                if (d & 0x80000000u) pos = 0, d = thr - a, sh--;
                ind_prior += sc[X_W_DIFF] * indicator(K, d >> sh, pos, 1);
            }
            const int16_t ind16 = (int16_t)((98307 - ind_prior) / 6);
            const int16_t dprior = (int16_t)(ind16 - (int16_t)sc[X_PRIOR]);
            sc[X_PRIOR] = (int16_t)((int16_t)sc[X_PRIOR] + (int16_t)((1638 * dprior) >> 14));
            const int32_t prior_ns = sc[X_PRIOR];
            FOR_BINS(r, b) {
                uint16_t ns = 0;
                const int32_t la = lrt[b];
                if (prior_ns > 0 && la < 65300) {
                    const int32_t e = wmul(la, 23637) >> 14;
                    int16_t ip = (int16_t)(e >> 12);
                    if (ip < -8) ip = -8;
                    const int16_t fr = (int16_t)(e & 0xfff);
                    int32_t p = (fr * fr * 44) >> 19;
                    p += (fr * 84) >> 7;
                    int32_t inv = wadd(wshl(1, 8 + ip), shift_w32(p, ip - 4));
                    const int n1 = norm_w32(inv), n2 = norm_w16((int16_t)(16384 - prior_ns));
                    if (n1 + n2 >= 7) {
                        if (n1 + n2 < 15) {
                            inv >>= 15 - n2 - n1;
                            inv = shift_w32(wmul(inv, 16384 - prior_ns), 7 - n1 - n2);
                        } else {
                            inv = wmul(inv, 16384 - prior_ns) >> 8;
                        }
                        ns = (uint16_t)((prior_ns << 8) / wadd(prior_ns, inv));
                    }
                }
                W.nsp[b] = ns;
            }
        }
        wave_sync();

        NSX_PROF(7);
        // noise update, :1840-1945 (gamma of a bin is set by the speech probability of the bin before it)
        int norm_max;
        {
            uint32_t mx = 0;
            const int post_shifts = sc[X_PREV_QNOISE] - q_magn, n_shifts = sc[X_PREV_QMAGN] - q_magn;
            FOR_BINS(r, b) {
                const uint16_t mg = W.magn[b], ns = W.nsp[b], pn = t_pn16[r];
                const uint32_t m = post_shifts < 0 ? (uint32_t)(mg >> -post_shifts) : (uint32_t)mg << post_shifts;
                int sign;
                uint32_t d;
                if (pn > m)
                    sign = -1, d = pn - m;
                else
                    sign = 1, d = m - pn;
                const uint32_t pv = pnoise[b];
                uint32_t upd = pv, dp = 0;
                const uint32_t gamma_in = (b == 0 || W.nsp[b - 1] >= 205) ? 26u : 3u;
                if (d && ns) {
                    dp = d * (uint32_t)ns;
                    const uint32_t st = (0x7c000000 & dp) ? (dp >> 5) * gamma_in : (dp * gamma_in) >> 5;
                    upd = sign > 0 ? upd + st : upd - st;
                }
                const uint32_t gamma = ns < 205 ? 3u : 26u;
                if (gamma_in != gamma) {
                    const uint32_t st = (0x7c000000 & dp) ? (dp >> 5) * gamma : (dp * gamma) >> 5;
                    const uint32_t alt = sign > 0 ? pv + st : pv - st;
                    if (upd > alt) upd = alt;
                }
                t_noise[r] = upd;
                mx = upd > mx ? upd : mx;
                int32_t pz = shift_w32(pause[b], -n_shifts);
                if (ns > 205) {
                    int32_t t;
                    if (n_shifts < 0) {
                        t = wsub((int32_t)mg, pz);
                        t = wmul(t, 13);
                        t = wadd(t, 128) >> 8;
                    } else {
                        t = wsub(wshl((int32_t)mg, n_shifts), pause[b]);
                        t = wmul(t, 13);
                        t = wadd(t, wshl(128, n_shifts)) >> (8 + n_shifts);
                    }
                    pz = wadd(pz, t);
                }
                pause[b] = pz;
            }
            norm_max = norm_u32(wave_umax(mx));
        }
        q_noise = (int16_t)(sc[X_PREV_QNOISE] + norm_max - 5);

        NSX_PROF(8);
        // step 3: Wiener gain from the updated noise, :1947-2013; previous-frame arrays, :2015-2029
        {
            const int n_shifts = sc[X_PREV_QNOISE] + 11 - q_magn;
            FOR_BINS(r, b) {
                const uint16_t mg = W.magn[b];
                const uint32_t nu = t_noise[r];
                uint32_t cur = 0, m, nz;
                if (n_shifts < 0) {
                    m = (uint32_t)mg;
                    nz = nu << -n_shifts;
                } else if (n_shifts > 17) {
                    m = (uint32_t)mg << 17;
                    nz = nu >> (n_shifts - 17);
                } else {
                    m = (uint32_t)mg << n_shifts;
                    nz = nu;
                }
                if (m > nz) {
                    uint32_t a = m - nz;
                    int nr = norm_u32(a);
                    if (nr > 11) nr = 11;
                    a <<= nr;
                    const uint32_t c = nz >> (11 - nr);
                    if (c > 0) a /= c;
                    cur = sat_max < a ? sat_max : a;
                }
                const uint32_t prior = t_pnear[r] * 2007u + cur * 41u;
                const uint32_t den = (uint32_t)overdrive + ((prior + 8192) >> 14);
                const uint16_t g = (uint16_t)((prior + den / 2) / den);
                uint16_t fl = g > 16384 ? (uint16_t)16384 : (g < denoise_bound ? (uint16_t)denoise_bound : g);
                if (block_index < 50) {
                    uint32_t a = (uint32_t)(fl * block_index);
                    a += (uint32_t)(t_ftmp[r] * (50 - block_index));
                    fl = (uint16_t)(a / 50u);
                }
                filt[b] = fl;
                pnoise[b] = norm_max > 5 ? nu << (norm_max - 5) : nu >> (5 - norm_max);
                pmagn[b] = mg;
            }
        }
        sc[X_PREV_QNOISE] = q_noise;
        sc[X_PREV_QMAGN] = q_magn;
        wave_sync();

        NSX_PROF(9);
        // ---- DataSynthesis, :1421-1499: PrepareSpectrumC :456-474, inverse transform, DenormalizeC :477-488
        int32_t t_spec[NP];
        FOR_BINS(r, b) t_spec[r] = W.cx[b];
        wave_sync();  // the spectrum lives in cx
        FOR_BINS(r, b) {
            const int32_t x = t_spec[r];
            const int16_t f = (int16_t)filt[b];
            const int16_t re = (int16_t)((lo16(x) * f) >> 14), im = (int16_t)((hi16(x) * f) >> 14);
            // real_fft.c:72-100: the packed spectrum carries -imag; the upper half is its conjugate mirror
            const int32_t v = pack16(re, (int16_t)-im);
            W.cx[bitrev<STAGES>(b)] = v;
            if (b > 0 && b < H) W.cx[bitrev<STAGES>(ANA - b)] = pack16(re, (int16_t) - (int16_t)-im);
        }
        wave_sync();
        int out_scale;
        if constexpr (STAGES == 7)
            out_scale = spl_cfft128<true>(W.cx, K.tw, lane);
        else
            out_scale = spl_cfft256<true>(W.cx, K.tw, lane);
        NSX_PROF(14);
        {
            int16_t tv[ANA / 64];
#pragma unroll
            for (int r = 0; r < ANA / 64; r++) tv[r] = sat_w16(shift_w32((int32_t)lo16(W.cx[lane + 64 * r]), out_scale - norm_data));
            wave_sync();  // td[] lives in cx
#pragma unroll
            for (int r = 0; r < ANA / 64; r++) td[lane + 64 * r] = tv[r];
        }
        wave_sync();
        int16_t gain = 8192;
        if (block_index > 200 && energy_in > 0) {  // gainMap == 1 for policy 2
            int sc_out = 0;
            int32_t e_out = wave_energy<ANA>(td, lane, &sc_out);
            if (sc_out == 0 && !(e_out & 0x7f800000))
                e_out = shift_w32(e_out, 8 + sc_out - scale_energy_in);
            else  // a negative count is an undefined shift + failed assert in the reference; unity gain here (orc_nsx.c)
                energy_in = 8 + sc_out - scale_energy_in >= 0 ? energy_in >> (8 + sc_out - scale_energy_in) : 0;
            if (energy_in > 0) {
                int16_t ratio = (int16_t)(wadd(e_out, energy_in / 2) / energy_in);
                ratio = ratio > 256 ? (int16_t)256 : (ratio < 0 ? (int16_t)0 : ratio);
                const int16_t g1 = K.factor1[ratio], g2 = K.factor2[ratio];
                const int16_t a = (int16_t)(((16384 - sc[X_PRIOR]) * g1) >> 14), c = (int16_t)((sc[X_PRIOR] * g2) >> 14);
                gain = (int16_t)(a + c);
            }
        }
        // SynthesisUpdateC, :491-521
        for (int i = lane; i < ANA; i += 64) {
            const int16_t w = (int16_t)mul_rsft_round(K.window[i], td[i], 14);
            const int16_t g = sat_w16(mul_rsft_round(w, gain, 13));
            syn[i] = sat_w16((int32_t)syn[i] + g);
        }
        wave_sync();
    }

    // read out the finished segment and slide the synthesis buffer (both the normal and the zero-input path)
    {
        int16_t keep[(KEEP + 63) / 64];
        for (int i = lane; i < BLOCK; i += 64) out[(long)i * CHN] = syn[i];
#pragma unroll
        for (int r = 0; r < (KEEP + 63) / 64; r++) {
            const int i = lane + 64 * r;
            keep[r] = i < KEEP ? syn[i + BLOCK] : (int16_t)0;
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r < (KEEP + 63) / 64; r++) {
            const int i = lane + 64 * r;
            if (i < KEEP) syn[i] = keep[r];
        }
        for (int i = lane; i < BLOCK; i += 64) syn[KEEP + i] = 0;
    }
    // high band: delayed copy (zero input) or time-domain gain from the top quarter of the low band, :2026-2115
    if (CHN == 2) {
        int16_t g = 16384;
        if (!zero_input) {
            uint32_t gs = 0, ps = 0;
            for (int b = lane; b < H; b += 64)
                if (b >= H - (H >> 2)) {
                    ps += W.nsp[b];
                    gs += filt[b];
                }
            const uint16_t psum = (uint16_t)wave_sum(ps);
            const uint32_t gsum = wave_sum(gs);
            const int16_t avg_prob = (int16_t)(4096 - (psum >> (STAGES - 7)));
            const int16_t avg_gain = (int16_t)(gsum >> (STAGES - 3));
            const int16_t gmod = avg_prob < 3607 ? avg_prob : (int16_t)3607;
            if (avg_prob < 2048) {
                g = (int16_t)((gmod << 1) + (avg_gain >> 1));
            } else {
                g = (int16_t)((3 * avg_gain) >> 2);
                g = (int16_t)(g + gmod);
            }
            g = g > 16384 ? (int16_t)16384 : (g < (int16_t)denoise_bound ? (int16_t)denoise_bound : g);
        }
        for (int i = lane; i < BLOCK; i += 64) out[(long)i * CHN + 1] = zero_input ? hb[i] : (int16_t)((g * hb[i]) >> 14);
    }
    wave_sync();
    NSX_PROF(10);
#ifdef WMX_NSX_PROF
    if (lane < 16) {  // racy read-modify-write among the waves that share a slot: good enough for a profile
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) v = lane == i ? prof_acc[i] : v;
        g_nsx_prof[(blockIdx.x & 1023) * 16 + lane] += v;
    }
#endif
}

#ifdef WMX_NSX_PROF
extern "C" int wmx_debug_nsx_prof(unsigned long long *out16, int reset) {
    (void)hipDeviceSynchronize();
    static unsigned long long all[1024 * 16];
    (void)hipMemcpyFromSymbol(all, HIP_SYMBOL(g_nsx_prof), sizeof(all));
    for (int i = 0; i < 16; i++) {
        out16[i] = 0;
        for (int g = 0; g < 1024; g++) out16[i] += all[g * 16 + i];
    }
    if (reset) {
        static const unsigned long long z[1024 * 16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_nsx_prof), z, sizeof(z));
    }
    return 0;
}
#endif

// One wave per stream, NsxShape::WPB streams per workgroup, all packets of the launch with the state in LDS.
template <int ANA, int CHN>
__global__ __launch_bounds__((64 * NsxShape<ANA, CHN>::WPB)) __attribute__((amdgpu_waves_per_eu(NsxShape<ANA, CHN>::WPE, NsxShape<ANA, CHN>::WPE))) void nsx_kernel(int32_t *__restrict__ state, int16_t *__restrict__ hist,
                                                                     const NsxConsts *__restrict__ consts, const int16_t *in, int16_t *out,
                                                                     int n_streams, int n_packets, long stream_stride, long packet_stride,
                                                                     int pkg, int overdrive, int denoise_bound, const uint8_t *__restrict__ active) {
    using Y = NsxLayout<ANA>;
    constexpr int WORDS = CHN == 2 ? Y::WORDS_2CH : Y::WORDS_MONO, BLOCK = ANA == 256 ? 160 : 80, WPB = NsxShape<ANA, CHN>::WPB;
    __shared__ NsxConsts K;
    __shared__ NsxWave<ANA, CHN> WS[WPB];
    {
        const int4 *src = reinterpret_cast<const int4 *>(consts);
        int4 *dst = reinterpret_cast<int4 *>(&K);
        for (int i = threadIdx.x; i < (int)(sizeof(NsxConsts) / 16); i += blockDim.x) dst[i] = src[i];
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long s = (long)blockIdx.x * WPB + wave;
    NsxWave<ANA, CHN> &W = WS[wave];
    const bool live = stream_active(active, (int)s, n_streams);  // no stream, or one that is switched off: nothing touched
    int32_t *st = state + (live ? s : 0) * (long)WORDS;
    if (live) {
        const int4 *src = reinterpret_cast<const int4 *>(st);
        int4 *dst = reinterpret_cast<int4 *>(W.st);
        for (int i = lane; i < WORDS / 4; i += 64) dst[i] = src[i];
    }
    __syncthreads();
    if (!live) return;
    const LdsScal sc{&W.st[Y::SCAL]};
    int16_t *hs = hist + s * (long)(3 * kNsxHist);
    for (int p = 0; p < n_packets; p++) {
        const int16_t *ip = in + s * stream_stride + (long)p * packet_stride;
        int16_t *op = out + s * stream_stride + (long)p * packet_stride;
        nsx_block<ANA, CHN>(W, K, sc, hs, ip, op, lane, overdrive, denoise_bound);
        // 32 kHz: the wrapper's packet is 320 frames but the core consumes 160; the rest of the output packet is
        // the wrapper's calloc zeros (src/webrtc.c:577 vs nsx_core.c:655)
        for (int i = BLOCK * CHN + lane; i < pkg * CHN; i += 64) op[i] = 0;
    }
    wave_sync();
    {
        int4 *dst = reinterpret_cast<int4 *>(st);
        const int4 *src = reinterpret_cast<const int4 *>(W.st);
        for (int i = lane; i < WORDS / 4; i += 64) dst[i] = src[i];
    }
}

__global__ void nsx_fill_state(int32_t *state, const int32_t *tmpl, int words, int n_streams) {
    const size_t total = (size_t)words * n_streams;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        state[i] = tmpl[i % words];
}

template <int ANA>
void nsx_template(std::vector<int32_t> &st, int chn) {
    using Y = NsxLayout<ANA>;
    st.assign(chn == 2 ? Y::WORDS_2CH : Y::WORDS_MONO, 0);
    int16_t *lq = reinterpret_cast<int16_t *>(&st[Y::LQ]), *dens = reinterpret_cast<int16_t *>(&st[Y::DENS]);
    uint16_t *filt = reinterpret_cast<uint16_t *>(&st[Y::FILT]);
    // WebRtcNsx_InitCore, nsx_core.c:631-784
    for (int i = 0; i < 3 * Y::BP; i++) lq[i] = 2048, dens[i] = 153;
    for (int i = 0; i < Y::BP; i++) filt[i] = 16384;
    int32_t *sc = &st[Y::SCAL];
    for (int i = 0; i < 3; i++) sc[X_COUNTER0 + i] = (int16_t)((int16_t)(200 * (i + 1)) / 3);
    sc[X_PRIOR] = 8192;
    sc[X_THR_LRT] = ANA == 256 ? 212644 : 131072;
    sc[X_THR_DIFF] = 50;
    sc[X_THR_FLAT] = 20480;
    sc[X_FEAT_LRT] = sc[X_THR_LRT];
    sc[X_FEAT_FLAT] = sc[X_THR_FLAT];
    sc[X_FEAT_DIFF] = sc[X_THR_DIFF];
    sc[X_W_LRT] = 6;
    sc[X_BLOCK_INDEX] = -1;
    sc[X_MIN_NORM] = 15;
}

}  // namespace
}  // namespace wmx

// ------------------------------------------------------------------------------------ host
struct wmx_nsx {
    int device;  // the HIP device the state lives on (current device at create); every entry point switches to it
    int n_streams, chn, freq, ana, pkg, words;
    int overdrive, denoise_bound;
    int32_t *d_state;
    int16_t *d_hist;
    wmx::NsxConsts *d_consts;
    int32_t *d_tmpl;  // the state ns_init gives a stream (reset_streams refills from it)
    wmx::StreamLife life;
};

extern "C" {

int wmx_nsx_destroy(wmx_nsx *h) {
    WMX_ON_DEVICE(h);
    if (!h) return 0;
    if (h->d_state) (void)hipFree(h->d_state);
    if (h->d_hist) (void)hipFree(h->d_hist);
    if (h->d_consts) (void)hipFree(h->d_consts);
    if (h->d_tmpl) (void)hipFree(h->d_tmpl);
    h->life.release();
    delete h;
    return 0;
}

// ns_release + ns_init for the listed streams in the reference's MAKE_WEBRTC_NSX build (src/webrtc.c:512-521, 560-602)
int wmx_nsx_reset_streams(wmx_nsx *h, const int32_t *idx, int n, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h || n < 0 || (n > 0 && !idx)) return WMX_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = wmx::as_stream(stream);
    const int32_t *d_idx = nullptr;
    const int rc = h->life.upload(idx, n, h->n_streams, s, &d_idx);
    if (rc != 0) return rc;
    const unsigned grid = (unsigned)(n < 4096 ? n : 4096);
    hipLaunchKernelGGL((wmx::fill_rows_idx<int32_t>), dim3(grid), dim3(256), 0, s, h->d_state, (const int32_t *)h->d_tmpl, h->words, d_idx, n);
    hipLaunchKernelGGL((wmx::fill_rows_idx<int16_t>), dim3(grid), dim3(256), 0, s, h->d_hist, (const int16_t *)nullptr, 3 * wmx::kNsxHist, d_idx, n);
    WMX_LAUNCH_CHECK();
    return h->life.done(s);
}

int wmx_nsx_set_active(wmx_nsx *h, const uint8_t *host_mask, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    return h->life.set_active(h->n_streams, host_mask, wmx::as_stream(stream));
}

// stream migration: [header | state words | 3 x 1000 histogram counters]
static constexpr uint32_t kNsxBlobVersion = 1;  // bump when the meaning of a state word changes (wmx_internal.h: blob_layout)
int wmx_nsx_stream_state_bytes(const wmx_nsx *h) { return h ? (int)(sizeof(wmx::BlobHeader) + h->words * 4 + 3 * wmx::kNsxHist * 2) : WMX_EINVAL; }

int wmx_nsx_export_stream(wmx_nsx *h, int stream_index, void *host_blob) {
    WMX_ON_DEVICE(h);
    if (!h || !host_blob || stream_index < 0 || stream_index >= h->n_streams) return WMX_EINVAL;
    WMX_HIP(hipDeviceSynchronize());
    const size_t hb = 3 * wmx::kNsxHist * 2;
    char *p = static_cast<char *>(host_blob);
    wmx::blob_begin(p, wmx::blob_tag("NSX "), wmx::blob_layout((uint32_t)h->words, kNsxBlobVersion), (uint32_t)(h->words * 4 + hb));
    p += sizeof(wmx::BlobHeader);
    WMX_HIP(hipMemcpy(p, h->d_state + (size_t)stream_index * h->words, (size_t)h->words * 4, hipMemcpyDeviceToHost));
    WMX_HIP(hipMemcpy(p + (size_t)h->words * 4, h->d_hist + (size_t)stream_index * 3 * wmx::kNsxHist, hb, hipMemcpyDeviceToHost));
    return 0;
}

int wmx_nsx_import_stream(wmx_nsx *h, int stream_index, const void *host_blob) {
    WMX_ON_DEVICE(h);
    if (!h || !host_blob || stream_index < 0 || stream_index >= h->n_streams) return WMX_EINVAL;
    const size_t hb = 3 * wmx::kNsxHist * 2;
    const int rc = wmx::blob_check(host_blob, wmx::blob_tag("NSX "), wmx::blob_layout((uint32_t)h->words, kNsxBlobVersion), (uint32_t)(h->words * 4 + hb));
    if (rc) return rc;
    WMX_HIP(hipDeviceSynchronize());
    const char *p = static_cast<const char *>(host_blob) + sizeof(wmx::BlobHeader);
    WMX_HIP(hipMemcpy(h->d_state + (size_t)stream_index * h->words, p, (size_t)h->words * 4, hipMemcpyHostToDevice));
    WMX_HIP(hipMemcpy(h->d_hist + (size_t)stream_index * 3 * wmx::kNsxHist, p + (size_t)h->words * 4, hb, hipMemcpyHostToDevice));
    return 0;
}

int wmx_nsx_create(wmx_nsx **out, int n_streams, int chn, int freq) {
    using namespace wmx;
    if (!out) return WMX_EINVAL;
    *out = nullptr;
    // ns_init (src/webrtc.c:563-564) + WebRtcNsx_InitCore (nsx_core.c:641-645); in[2] / out[2] => chn <= 2
    if ((freq != 8000 && freq != 16000 && freq != 32000) || chn < 1 || chn > 2 || n_streams < 1) {
        set_error("wmx_nsx_create: unsupported n_streams=%d chn=%d freq=%d", n_streams, chn, freq);
        return WMX_EINVAL;
    }
    wmx_nsx *h = new wmx_nsx();
    if ((h->device = current_device()) < 0) {
        delete h;
        return WMX_ENODEV;
    }
    h->n_streams = n_streams;
    h->chn = chn;
    h->freq = freq;
    h->ana = freq == 8000 ? 128 : 256;
    h->pkg = freq / 1000 * 10;
    h->overdrive = 282;       // WebRtcNsx_set_policy_core(2), nsx_core.c:804-808 (NS_AGGRESSIVE 2, src/webrtc.c:532)
    h->denoise_bound = 2048;
    h->d_state = nullptr;
    h->d_hist = nullptr;
    h->d_consts = nullptr;
    h->d_tmpl = nullptr;
    std::vector<int32_t> st;
    if (h->ana == 128)
        nsx_template<128>(st, chn);
    else
        nsx_template<256>(st, chn);
    h->words = (int)st.size();
    NsxConsts *K = new NsxConsts();
    memset(K, 0, sizeof(*K));
    spl_twiddles(fx_spl_sin1024, &K->tw);
    if (h->ana == 256)
        memcpy(K->window, fx_nsx_window256, sizeof(fx_nsx_window256));
    else
        memcpy(K->window, fx_nsx_window128, sizeof(fx_nsx_window128));
    memcpy(K->log_frac, fx_nsx_log_frac, sizeof(fx_nsx_log_frac));
    memcpy(K->counter_div, fx_nsx_counter_div, sizeof(fx_nsx_counter_div));
    memcpy(K->log_index, fx_nsx_log_index, sizeof(fx_nsx_log_index));
    memcpy(K->factor1, fx_nsx_factor1, sizeof(fx_nsx_factor1));
    memcpy(K->factor2, fx_nsx_factor2_mode2, sizeof(fx_nsx_factor2_mode2));
    memcpy(K->indicator, fx_nsx_indicator, sizeof(fx_nsx_indicator));
    memcpy(K->log_table, fx_nsx_log_table, sizeof(fx_nsx_log_table));
    K->sum_log_index_5 = fx_nsx_sum_log_index[5], K->sum_log_index_65 = fx_nsx_sum_log_index[65];
    K->sum_sq_log_index_5 = fx_nsx_sum_sq_log_index[5], K->sum_sq_log_index_65 = fx_nsx_sum_sq_log_index[65];
    K->determinant_5 = fx_nsx_determinant[5];
    hipError_t e;
#define NSX_TRY(x)                                        \
    if ((e = (x)) != hipSuccess) {                        \
        const int rc = hip_fail(e, #x, __FILE__, __LINE__); \
        wmx_nsx_destroy(h);                               \
        delete K;                                         \
        return rc;                                        \
    }
    NSX_TRY(hipMalloc(&h->d_state, (size_t)h->words * n_streams * sizeof(int32_t)));
    NSX_TRY(hipMalloc(&h->d_hist, (size_t)3 * kNsxHist * n_streams * sizeof(int16_t)));
    NSX_TRY(hipMalloc(&h->d_consts, sizeof(NsxConsts)));
    NSX_TRY(hipMalloc(&h->d_tmpl, st.size() * sizeof(int32_t)));
    NSX_TRY(hipMemcpy(h->d_consts, K, sizeof(NsxConsts), hipMemcpyHostToDevice));
    NSX_TRY(hipMemcpy(h->d_tmpl, st.data(), st.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    NSX_TRY(hipMemset(h->d_hist, 0, (size_t)3 * kNsxHist * n_streams * sizeof(int16_t)));
    hipLaunchKernelGGL(nsx_fill_state, dim3(1024), dim3(256), 0, nullptr, h->d_state, h->d_tmpl, h->words, n_streams);
    NSX_TRY(hipGetLastError());
    NSX_TRY(hipDeviceSynchronize());
#undef NSX_TRY
    delete K;
    *out = h;
    return 0;
}

int wmx_nsx_packet_samples(const wmx_nsx *h) { return h ? h->pkg * h->chn : WMX_EINVAL; }
int wmx_nsx_state_bytes(const wmx_nsx *h) { return h ? h->words * 4 : WMX_EINVAL; }

int wmx_nsx_process(wmx_nsx *h, const int16_t *d_in, int16_t *d_out, int n_packets, long stream_stride, long packet_stride, void *stream) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || n_packets < 0) {
        set_error("wmx_nsx_process: bad argument");
        return WMX_EINVAL;
    }
    if (n_packets == 0) return 0;
    if (!d_in || !d_out) {
        set_error("wmx_nsx_process: null buffer");
        return WMX_EINVAL;
    }
    const long per_pkt = (long)h->pkg * h->chn;
    if (packet_stride < per_pkt || (h->n_streams > 1 && stream_stride < per_pkt)) {
        // a packet must not overlap its neighbours (the kernel would write across rows)
        set_error("wmx_nsx_process: strides (%ld, %ld) smaller than a packet (%ld samples)", stream_stride, packet_stride, per_pkt);
        return WMX_EINVAL;
    }
    hipStream_t s = as_stream(stream);
#define NSX_LAUNCH(A, C)                                                                                                    \
    hipLaunchKernelGGL((nsx_kernel<A, C>), dim3((unsigned)((h->n_streams + NsxShape<A, C>::WPB - 1) / NsxShape<A, C>::WPB)),  \
                       dim3(64 * NsxShape<A, C>::WPB), 0, s, h->d_state, h->d_hist, h->d_consts, d_in, d_out, h->n_streams, \
                       n_packets, stream_stride, packet_stride, h->pkg, h->overdrive, h->denoise_bound, h->life.d_active)
    if (h->ana == 128) {
        if (h->chn == 1)
            NSX_LAUNCH(128, 1);
        else
            NSX_LAUNCH(128, 2);
    } else {
        if (h->chn == 1)
            NSX_LAUNCH(256, 1);
        else
            NSX_LAUNCH(256, 2);
    }
#undef NSX_LAUNCH
    WMX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
