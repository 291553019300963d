// ns.hip -- batched float noise suppressor for gfx950: one wavefront per stream.
//
// Replaces, for many independent streams per launch, what wmix's ns_process() does per
// 10 ms packet (src/webrtc.c:612-644): WebRtcNs_Analyze(in[0]) followed by
// WebRtcNs_Process(in, chn, out) of the vendored float NS
// (W:modules/audio_processing/ns/ns_core.c:1043-1181 AnalyzeCore, :1183-1415 ProcessCore,
// policy 2 set by ns_init, src/webrtc.c:532,577).
//
// Mapping.  A stream's spectrum has 65 (8 kHz) or 129 (16/32 kHz) bins; lane k of the
// wave owns bins k, k+64 (and lane 0 bin 128).  Per-bin state is read from / written to
// HBM exactly once per frame with 256-byte coalesced accesses; the FFT work array, the frame's
// per-bin intermediates and the reduction staging live in LDS (8.4 KB per wave), the FFT tables,
// window and libm tables once per 4-wave workgroup.  The kernel fits 128 VGPRs and runs 4 waves per
// SIMD; the per-bin state a frame needs is requested from HBM at the start of the frame and consumed
// phases later (its loops were latency bound before that), the independent ordered sums of a phase
// advance as parallel lane chains, and what is left is VALU issue.
// All per-stream scalars and control flow (start-up phases, zero-energy early-out,
// histogram windows) are wave-uniform, so streams in different states never diverge
// inside a wave.
//
// Because wmix always hands the SAME frame to Analyze and Process, analyzeBuf == dataBuf
// and magnPrevAnalyze == magnPrevProcess at all times (both start at zero and receive
// identical updates, ns_core.c:1069,1228 and :1180,1298), so one window+FFT serves both
// halves and one copy of each is kept in the state.  `noise`, `speechProb` and
// `parametricNoise` are only consumed inside the frame that produces them and are not
// state here.
//
// Numerics.  Same float expressions as the reference, -ffp-contract=off, the reference's
// double-precision libm calls evaluated in double (log / exp by the table-driven routines of
// libm_dev.h, same floats as glibc's over every argument class swept; tanh / pow by ocml).  Sums over
// bins/samples are where a parallel machine wants a different order, and does not get it: the
// kernel adds in the reference's index order (bit-exact against the CPU path).  (Rounds 1-5 also
// carried a variant in which each lane added its own elements and the wave combined them by a
// butterfly -- an ulp of difference in the sums, which NS feeds back into decisions: up to 13 LSB
// on a few samples, outside north_star's +-1 LSB.  It is gone since round 6: no entry point
// of this library produces anything but the reference's result.)
#include <cmath>
#include <vector>
#include "wmx_internal.h"
#include "fft_ooura.h"
#include "libm_dev.h"
#include "ns_layout.h"

namespace wmx {
namespace {

constexpr int kStartupShort = 50;   // defines.h:22
constexpr int kStartupLong = 200;   // defines.h:21
constexpr int kHistBins = 1000;     // defines.h:45
constexpr int kUpdateWindow = 500;  // ns_core.c:188
constexpr int kStartBand = 5;       // ns_core.c:1045

// Block-shared constants (one copy per 4-wave workgroup) and one working set per wave / stream.
constexpr int kNsWavesPerBlock = 4;

template <int L>
struct NsConstLds {
    FftTables tab;                // Ooura tables for n = L
    float window[L];              // hybrid Hanning window (windows_private.h:64,94)
    float logi[NsLayout<L>::MP];  // (float)log((float)i); [MP-2], [MP-1] hold the two data-independent start-up sums
    NsLibmTables lm;              // tables of fast_log_ge1 / fast_exp (libm_dev.h)
};
template <int L>
struct alignas(16) NsWaveLds {
    static constexpr int MP = NsLayout<L>::MP;
    float fa[L];  // FFT work array / packed spectrum / time-domain output staging
    // per-bin intermediates of the frame that another lane reads (ordered sums walk whole arrays, sprob[b - 1], magn[0]) or
    // that a rolled loop indexes (noise).  The ones only their own lane touches -- the spectrum, the previous-frame SNR
    // estimate, the instantaneous SNR -- are registers of ns_frame (t_re / t_im / t_prev / t_snrq, index = the group k of
    // the unrolled per-bin loops): 5 776 B per stream instead of 8 416, five workgroups per CU instead of four.
    float magn[MP], lmagn[MP], noise[MP], snrp[MP], sprob[MP], pause[MP];
    float r0[MP], r1[MP], r2[MP];  // staging of terms for the ordered sums; r0..r1 double as a L-float time-domain stage
#ifdef WMX_NS_PROF
    unsigned long long prof[16];
#endif
};
#ifdef WMX_NS_PROF  // developer build only (make EXTRA=-DWMX_NS_PROF): cycles per phase of ns_frame, summed over waves
__device__ unsigned long long g_ns_prof[16];
#define NS_PROF(i)                                                         \
    do {                                                                   \
        const long long t_now = clock64();                                 \
        if (lane == 0) W.prof[i] += (unsigned long long)(t_now - prof_t0); \
        prof_t0 = clock64();                                               \
    } while (0)
#else
#define NS_PROF(i)
#endif
static_assert(2 * NsLayout<256>::MP >= 256 && 2 * NsLayout<128>::MP >= 128, "r0..r1 must hold L floats");

// WEBRTC_SPL_SAT as the reference spells it: a NaN fails both comparisons and passes through (and the conversion to int16 behind
// it makes it 0, on x86 as on gfx950).  NaNs do reach this in the reference's own runs (the AEC's first blocks), so the one-instruction
// median v_med3_f32(v, -32768, 32767), which answers a NaN with -32768, is not a substitute -- tried, aec_golden caught it.
__device__ __forceinline__ float sat16f(float v) { return v > 32767.f ? 32767.f : (v < -32768.f ? -32768.f : v); }

// sum of x[lo..hi) in the reference's index order; every lane adds the same LDS-broadcast values (16-byte reads, eight terms in
// flight) and gets the result
__device__ __forceinline__ float sum_range(const float *x, int lo, int hi, int lane) {
    float acc = 0.f;
    int i = lo;
#pragma unroll 1
    for (; i < hi && (i & 3); i++) acc += x[i];
#pragma unroll 4
    for (; i + 8 <= hi; i += 8) {
        const float4 a = *reinterpret_cast<const float4 *>(x + i), b = *reinterpret_cast<const float4 *>(x + i + 4);
        acc += a.x;
        acc += a.y;
        acc += a.z;
        acc += a.w;
        acc += b.x;
        acc += b.y;
        acc += b.z;
        acc += b.w;
    }
#pragma unroll 1
    for (; i < hi; i++) acc += x[i];
    return acc;
}

// Several ordered sums at once: lane k (k < n) adds the MP4*4 floats at its own pointer in index order, all chains
// advancing in the same instructions; sum k is then read from lane k.  The arrays are zero outside the reference's
// summation range (x + 0.0f == x), so every chain runs the same full length.
template <int MP4>
__device__ __forceinline__ float sum_lanes(const float *mine) {
    float acc = 0.f;
#pragma unroll 4
    for (int i = 0; i < MP4; i += 2) {
        const float4 a = *reinterpret_cast<const float4 *>(mine + 4 * i);
        acc += a.x;
        acc += a.y;
        acc += a.z;
        acc += a.w;
        if (i + 1 < MP4) {
            const float4 b = *reinterpret_cast<const float4 *>(mine + 4 * i + 4);
            acc += b.x;
            acc += b.y;
            acc += b.z;
            acc += b.w;
        }
    }
    return acc;
}
// second call site of the 32 kHz path: kept out of line so that its fp64 temporaries do not raise the kernel's pressure
__device__ __noinline__ static float tanh_call(float x, const NsLibmTables *m) { return fast_tanh(x, *m); }
__device__ __forceinline__ float lane_value(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }

template <int L, int CHN>
__device__ void ns_frame(const NsConstLds<L> &K, NsWaveLds<L> &W, float *__restrict__ st, unsigned short *__restrict__ hist,
                         const int16_t *in, int16_t *out, const int pkg, const int lane_in) {
    using Y = NsLayout<L>;
    constexpr int chn = CHN;  // 1 or 2 (ns_init takes nothing else): compile-time, so the mono kernel carries no high band
    // Lane-derived addresses are loop invariant; left alone the compiler hoists them out of the packet loop and
    // keeps them in VGPRs for the whole kernel.  Re-deriving them per phase is a few VALU ops and frees the registers.
    int lane = lane_in;
#define NS_RELANE() asm volatile("" : "+v"(lane))
    NS_RELANE();
#ifdef WMX_NS_PROF
    long long prof_t0 = clock64();
#endif
    constexpr int M = Y::M, B = Y::B, NT = L / 64, NC = L / 2;
    // the stream's state block: ONE scalar base for the launch, every access `base + (4 * lane + constant)` in the SGPR-base +
    // 32-bit-VGPR-offset form of global_load / global_store (wmx_internal.h: global_row, row_ld, row_st) -- as `st[i]` with a
    // 64-bit `st` the compiler builds a per-lane 64-bit address for every array (v_lshl_add_u64, v_add_co / v_addc_co pairs)
    struct StF {
        GlobalF g;
        unsigned i;
        __device__ __forceinline__ operator float() const { return row_ld(g, i); }
        __device__ __forceinline__ void operator=(float v) const { row_st(g, i, v); }
    };
    struct StI {
        GlobalF g;
        unsigned i;
        __device__ __forceinline__ operator int() const { return __float_as_int(row_ld(g, i)); }
        __device__ __forceinline__ void operator=(int v) const { row_st(g, i, __int_as_float(v)); }
    };
    const GlobalF st_row = global_row(st, 0);
#define ST(idx) (StF{st_row, (unsigned)(idx)})
#define STI(idx) (StI{st_row, (unsigned)(idx)})
    float *tdst = W.r0;  // L floats spanning r0..r1

    // ---- load the packet (channel 0 = low band, channel 1 = "high band") and slide the analysis
    //      buffer (UpdateBuffer, ns_core.c:855-873).  All loads before the stores.
    // the stream's 24 scalar state words in one coalesced load; read back lane by lane (every field is read at most
    // once per frame, before it is written)
    const float scv = lane < 24 ? ST(Y::SCALARS + lane) : 0.f;
#define SCF(f) lane_value(scv, Y::f - Y::SCALARS)
#define SCI(f) __float_as_int(lane_value(scv, Y::f - Y::SCALARS))
    float buf[NT], hb[NT], synt[NT];
#pragma unroll
    for (int k = 0; k < NT; k++) {
        const int i = lane + 64 * k;
        // No load under a lane-dependent branch (such a load is waited for on the spot, one HBM round trip each): which
        // of the 64-sample groups hold old samples, new samples or both is a compile-time fact; the mixed group
        // fetches both with clamped addresses and selects.
        constexpr int KEEP = L - B;  // samples carried over from the previous frame
        const bool all_old = 64 * k + 63 < KEEP, all_new = 64 * k >= KEEP;
        const bool is_old = all_old || (!all_new && i < KEEP);
        float o = 0.f, oh = 0.f;
        int16_t n = 0, nh = 0;
        if (!all_new) o = ST(Y::IN_BUF + (is_old ? i + B : 0));
        if (!all_old) n = in[(is_old ? 0 : i - KEEP) * chn];
        buf[k] = is_old ? o : (float)n;
        // syntBuf: its first KEEP samples carry the overlap, the rest is the zeros UpdateBuffer shifted in (ns_core.c:855-873): known, not
        // stored, not read -- 1 280 bytes per frame less traffic together with the store below
        synt[k] = (64 * k < KEEP && i < KEEP) ? ST(Y::SYNT_BUF + (i < KEEP ? i : 0)) : 0.f;
        hb[k] = 0.f;
        if (chn == 2) {
            if (!all_new) oh = ST(Y::HB_BUF + (is_old ? i + B : 0));
            if (!all_old) nh = in[(is_old ? 0 : i - KEEP) * chn + 1];
            hb[k] = is_old ? oh : (float)nh;
        }
    }
    // Per-bin state this frame will need, requested from HBM right behind the time-domain buffers (loads return in order) and consumed
    // several phases later (window, energy sum, FFT, spectrum loop and the ordered sums run in between): the loops below
    // never wait on HBM latency.  (A zero-energy frame does not use them; it is the rare case.)
#if defined(WMX_NS_EXP) && WMX_NS_EXP == 1  // timing-only experiment (WRONG results): the per-bin loops without their lone Nyquist pass
    constexpr int NI = M / 64;
#else
    constexpr int NI = (M + 63) / 64;  // bins per lane: 3 (M = 129) or 2 (M = 65), the last one lane 0 only
#endif
    float pf_quant[NI], pf_dens[NI][3], pf_lq[NI][3], pf_pause[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) {
        // unconditional loads: lanes past the last bin fetch bin M - 1 again (same value in every lane of that group), so
        // nothing below needs a lane test around its arithmetic
        const int b = lane + 64 * k < M ? lane + 64 * k : M - 1;
        pf_pause[k] = ST(Y::MAGN_AVG_PAUSE + b);
        pf_quant[k] = ST(Y::QUANTILE + b);
#pragma unroll
        for (int q = 0; q < 3; q++) {
            pf_dens[k][q] = ST(Y::DENSITY + q * Y::MP + b);
            pf_lq[k][q] = ST(Y::LQUANTILE + q * Y::MP + b);
        }
    }
    wave_sync();  // other lanes' stores below overwrite what this lane just loaded: keep the compiler from interleaving them
#pragma unroll
    for (int k = 0; k < NT; k++) {
        // only what the NEXT frame reads back: its KEEP old samples are this buffer's last KEEP (positions >= B); the first B stay
        // whatever they were -- 640 bytes per frame that the kernel, which runs at three quarters of the device-copy rate, need not write
        const int i = lane + 64 * k;
        if (64 * k + 63 < B) continue;  // a compile-time fact per group
        if (i >= B) {
            ST(Y::IN_BUF + i) = buf[k];
            if (chn == 2) ST(Y::HB_BUF + i) = hb[k];
        }
    }
    // window + energy (ns_core.c:1071-1072 / 1241-1242)
#pragma unroll
    for (int k = 0; k < NT; k++) {
        const int i = lane + 64 * k;
        const float w = K.window[i] * buf[k];
        W.fa[i] = w;
        tdst[i] = w * w;
    }
    wave_sync();
    const float energy1 = sum_range(tdst, 0, L, lane);
    wave_sync();
    NS_PROF(0);

    float hb_gain = 1.f;
    const bool zero_frame = (energy1 == 0.0f);
    float t_re[(Y::M + 63) / 64], t_im[(Y::M + 63) / 64], t_prev[(Y::M + 63) / 64], t_snrq[(Y::M + 63) / 64];

    if (!zero_frame) {
        // ===================================================== Analyze (ns_core.c:1085-1180)
        const int block_ind = SCI(S_BLOCK_IND) + 1;
        STI(Y::S_BLOCK_IND) = block_ind;
        const int update_flag = SCI(S_UPDATE_FLAG);
        const bool startup = block_ind < kStartupShort;
        const float overdrive = 1.1f, denoise_bound = 0.125f;  // policy 2, ns_core.c:1030-1033

        NS_RELANE();
        rdft_forward<NC>(W.fa, &K.tab, lane);
        if (lane < Y::MP - M) W.r0[M + lane] = 0.f;  // the window-energy stage above spilled into r0's zero tail
        NS_PROF(1);

        NS_RELANE();
        // ---- spectrum, magnitude, log-magnitude (FFT() ns_core.c:886-911; :228, :1095)
#pragma unroll
        for (int k = 0; k < NI; k++) {
            // No lane-dependent branch around the arithmetic: groups that are full (a compile-time fact) run straight-line,
            // and in the last group (bin M - 1 alone) every lane evaluates that bin and lane 0 stores it -- the log chains of
            // the groups then sit in one basic block and overlap, instead of one serial pass for lane 0.
            const int b0 = lane + 64 * k;
            const bool ok = (64 * k + 63 < M) || b0 < M;
            const int b = ok ? b0 : M - 1;
            float re, im, mg;
            if (k == NI - 1) {  // only bin M - 1 (Nyquist): real
                re = W.fa[1];
                im = 0.f;
                mg = fabsf(re) + 1.f;
            } else {
                const float a0 = W.fa[2 * b], a1 = W.fa[2 * b + 1];
                const bool dc = k == 0 && b == 0;  // bin 0 is real; fa[1] holds the Nyquist value
                re = a0;
                im = dc ? 0.f : a1;
                const float g = sqrtf(re * re + im * im) + 1.f;
                mg = dc ? fabsf(re) + 1.f : g;
            }
            const float lm = fast_log_ge1(mg, K.lm);
            const float pz = pf_pause[k];
            t_re[k] = re;
            t_im[k] = im;
            if (!ok) continue;
            W.magn[b] = mg;
            W.lmagn[b] = lm;
            W.pause[b] = pz;
            W.r0[b] = re * re + im * im;
            W.r1[b] = b >= kStartBand ? K.logi[b] * lm : 0.f;  // terms of the start-up sums (i >= 5, ns_core.c:1092)
            W.r2[b] = b >= 1 ? lm : 0.f;                       // spectral flatness skips bin 0 (ns_core.c:540)
            W.snrp[b] = b >= kStartBand ? lm : 0.f;            // snrp is free until ComputeSnr
        }
        wave_sync();
        NS_PROF(2);
        // ordered reductions over the bins (ns_core.c:1089-1101, :540, :608)
        float signal_energy, sum_magn, flat_num, avg_pause, sum_log_magn = 0.f, sum_log_i_log_magn = 0.f;
        {
            const float *mine =
                lane == 1 ? W.magn : (lane == 2 ? W.r2 : (lane == 3 ? W.pause : (lane == 4 ? W.snrp : (lane == 5 ? W.r1 : W.r0))));
            const float acc = sum_lanes<Y::MP / 4>(mine);
            signal_energy = lane_value(acc, 0);
            sum_magn = lane_value(acc, 1);
            flat_num = lane_value(acc, 2);
            avg_pause = lane_value(acc, 3);
            if (startup) {
                sum_log_magn = lane_value(acc, 4);
                sum_log_i_log_magn = lane_value(acc, 5);
            }
        }
        const float magn0 = W.magn[0];
        wave_sync();
        NS_PROF(3);
        signal_energy = signal_energy / ((float)M);

        NS_RELANE();
        // ---- NoiseEstimation (ns_core.c:217-285)
        int updates = SCI(S_UPDATES);
        if (updates < kStartupLong) updates++;
        STI(Y::S_UPDATES) = updates;
        const int cnt0 = SCI(S_COUNTER + 0), cnt1 = SCI(S_COUNTER + 1), cnt2 = SCI(S_COUNTER + 2);
        // second prefetch wave: what ComputeSnr and SpeechNoiseProb read (consumed after the quantile update and the
        // next ordered sums)
        float pf_nprev[NI], pf_mprev[NI], pf_smooth[NI], pf_lrt[NI];
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int b = lane + 64 * k < M ? lane + 64 * k : M - 1;
            pf_nprev[k] = ST(Y::NOISE_PREV + b);
            pf_mprev[k] = ST(Y::MAGN_PREV + b);
            pf_smooth[k] = ST(Y::SMOOTH + b);
            pf_lrt[k] = ST(Y::LOG_LRT + b);
        }
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int b0 = lane + 64 * k;
            const bool ok = (64 * k + 63 < M) || b0 < M;  // compile-time true except in the last group (bin M - 1 alone)
            const int b = ok ? b0 : M - 1;
            const float lm = W.lmagn[b];
            float quant = pf_quant[k], lq = 0.f;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int cnt = q == 0 ? cnt0 : (q == 1 ? cnt1 : cnt2);
                const float c1 = (float)(cnt + 1), cf = (float)cnt;
                float dens = pf_dens[k][q];
                lq = pf_lq[k][q];
                float delta;
                // div_ordinary (libm_dev.h): 1 < dens <= 50, 1 <= c1 <= 201, 0.2 <= numerators <= 10 050
                if (dens > 1.0f)
                    delta = div_ordinary(40.f * 1.f, dens);
                else
                    delta = 40.f;
                // lq += 0.25 delta / c1  or  lq -= 0.75 delta / c1 (ns_core.c:247-253) as ONE division: a quotient changes sign with its
                // numerator and nothing else, and x - y is x + (-y): same bits, one correction sequence instead of two
                lq += div_ordinary(lm > lq ? 0.25f * delta : -((1.f - 0.25f) * delta), c1);
                if (fabsf(lm - lq) < 0.01f) {
                    dens = div_ordinary(cf * dens + 1.f / (2.f * 0.01f), c1);
                    if (ok) ST(Y::DENSITY + q * Y::MP + b) = dens;
                }
                if (ok) ST(Y::LQUANTILE + q * Y::MP + b) = lq;
                if (cnt >= kStartupLong && updates >= kStartupLong) quant = fast_exp(lq, K.lm);
            }
            if (updates < kStartupLong) quant = fast_exp(lq, K.lm);  // lq of the last estimator
            if (ok) {
                // the noise quantile changes in a stream's first 200 frames and then when one of the three staggered counters comes
                // round (3 frames in 200): the other frames need not store what they loaded
                if (updates < kStartupLong || cnt0 >= kStartupLong || cnt1 >= kStartupLong || cnt2 >= kStartupLong) ST(Y::QUANTILE + b) = quant;
                W.noise[b] = quant;
            }
        }
        STI(Y::S_COUNTER + 0) = cnt0 >= kStartupLong ? 1 : cnt0 + 1;
        STI(Y::S_COUNTER + 1) = cnt1 >= kStartupLong ? 1 : cnt1 + 1;
        STI(Y::S_COUNTER + 2) = cnt2 >= kStartupLong ? 1 : cnt2 + 1;

        NS_RELANE();
        NS_PROF(4);
        // ---- start-up white/pink parametric noise model (ns_core.c:1108-1160); parametricNoise kept in r1
        if (startup) {
            const float white = SCF(S_WHITE) + sum_magn / ((float)M) * overdrive;
            ST(Y::S_WHITE) = white;
            const float sum_log_i = K.logi[Y::MP - 2], sum_log_i_sq = K.logi[Y::MP - 1];
            float t1 = sum_log_i_sq * ((float)(M - kStartBand));
            t1 -= (sum_log_i * sum_log_i);
            float t2 = (sum_log_i_sq * sum_log_magn - sum_log_i * sum_log_i_log_magn);
            float t3 = t2 / t1;
            if (t3 < 0.f) t3 = 0.f;
            const float pink_num = SCF(S_PINK_NUM) + t3;
            ST(Y::S_PINK_NUM) = pink_num;
            t2 = (sum_log_i * sum_log_magn);
            t2 -= ((float)(M - kStartBand)) * sum_log_i_log_magn;
            t3 = t2 / t1;
            if (t3 < 0.f) t3 = 0.f;
            if (t3 > 1.f) t3 = 1.f;
            const float pink_exp = SCF(S_PINK_EXP) + t3;
            ST(Y::S_PINK_EXP) = pink_exp;
            float pnum = 0.0f, pexp = 0.0f;
            if (pink_exp > 0.f) {
                pnum = exp_d(pink_num / (float)(block_ind + 1));
                pnum *= (float)(block_ind + 1);
                pexp = pink_exp / (float)(block_ind + 1);
            }
#pragma unroll 1
            for (int b = lane; b < M; b += 64) {
                float pn;
                if (pink_exp == 0.f) {
                    pn = white;
                } else {
                    const float band = (float)(b < kStartBand ? kStartBand : b);
                    pn = div_pow_d(pnum, band, pexp);
                }
                W.r1[b] = pn;
                float nz = W.noise[b];
                nz *= (float)(block_ind);
                const float t = pn * (float)(kStartupShort - block_ind);
                nz += (t / (float)(block_ind + 1));
                nz /= (float)kStartupShort;
                W.noise[b] = nz;
            }
        }
        // normalisation of the spectral-difference feature (ns_core.c:1163-1167)
        float feat_norm = SCF(S_FEAT_NORM);
        if (block_ind < kStartupLong) {
            feat_norm *= (float)block_ind;
            feat_norm += signal_energy;
            feat_norm /= (float)(block_ind + 1);
        }

        NS_RELANE();
        // ---- ComputeSnr (ns_core.c:566-588); prev is reused by the Wiener filter (:996).  Same loop: the terms of
        //      the spectral-difference sums (ns_core.c:612-620), which need avg_pause / avg_magn only.
        avg_pause = avg_pause / ((float)M);
        const float avg_magn = sum_magn / ((float)M);
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int b0 = lane + 64 * k;
            const bool ok = (64 * k + 63 < M) || b0 < M;  // see the noise estimation loop
            const int b = ok ? b0 : M - 1;
            const float mg = W.magn[b], nz = W.noise[b];
            const float np = pf_nprev[k];
            // div_ordinary: magnitudes are 0 (a stream's first frame) or in [1, 2^24] (|X| + 1 of 256 int16 samples), noise
            // estimates positive and below 2^27, so the denominators lie in [1e-4, 2^27]
            const float pe = div_ordinary(pf_mprev[k], np + 0.0001f) * pf_smooth[k];
            float sq = 0.f;
            if (mg > nz) sq = div_ordinary(mg, nz + 0.0001f) - 1.f;
            const float dm = mg - avg_magn, dp = W.pause[b] - avg_pause;
            t_prev[k] = pe;
            t_snrq[k] = sq;
            if (!ok) continue;
            W.snrp[b] = 0.98f * pe + (1.f - 0.98f) * sq;
            W.r0[b] = dm * dp;
            W.r2[b] = dp * dp;
            W.lmagn[b] = dm * dm;  // lmagn is dead from here on
        }
        wave_sync();
        NS_PROF(5);
        // ---- FeatureUpdate: spectral flatness (ns_core.c:523-556), difference (:595-634)
        float feat_flat = SCF(S_FEAT_FLAT);
        {
            float den = sum_magn;
            den -= magn0;
            den = den / (float)M;
            const float num = flat_num / (float)M;
            const float tmp = fast_exp(num, K.lm) / den;
            feat_flat += 0.3f * (tmp - feat_flat);
        }
        float cov, var_pause, var_magn;
        {
            const float acc = sum_lanes<Y::MP / 4>(lane == 1 ? W.r2 : (lane == 2 ? W.lmagn : W.r0));
            cov = lane_value(acc, 0);
            var_pause = lane_value(acc, 1);
            var_magn = lane_value(acc, 2);
        }
        wave_sync();
        NS_PROF(6);
        cov = cov / ((float)M);
        var_pause = var_pause / ((float)M);
        var_magn = var_magn / ((float)M);
        float feat_acc = SCF(S_FEAT_ACC) + signal_energy;
        float feat_diff = SCF(S_FEAT_DIFF);
        {
            float d = var_magn - (cov * cov) / (var_pause + 0.0001f);
            d = d / (feat_norm + 0.0001f);
            feat_diff += 0.3f * (d - feat_diff);
        }
        float feat_lrt = SCF(S_FEAT_LRT);
        float thr_lrt = SCF(S_THR_LRT), thr_flat = SCF(S_THR_FLAT), thr_diff = SCF(S_THR_DIFF);
        float w_lrt = SCF(S_W_LRT), w_flat = SCF(S_W_FLAT), w_diff = SCF(S_W_DIFF);
        // histograms + threshold extraction every 500 blocks (ns_core.c:293-518, :765-790)
        if (update_flag >= 1) {
            const int countdown = SCI(S_COUNTDOWN) - 1;
            if (countdown > 0) {
                if (lane == 0) {
                    // counter++ as a no-return 32-bit atomic on the dword that holds the uint16 (a counter never exceeds
                    // the 500-frame window, so the low half cannot carry into the high one): nothing waits for HBM
                    auto bump = [&](int i) { atomicAdd(reinterpret_cast<unsigned *>(hist) + (i >> 1), (i & 1) ? 0x10000u : 1u); };
                    if ((feat_lrt < kHistBins * 0.1f) && (feat_lrt >= 0.0f)) bump((int)(feat_lrt / 0.1f));
                    if ((feat_flat < kHistBins * 0.05f) && (feat_flat >= 0.0f)) bump(kHistBins + (int)(feat_flat / 0.05f));
                    if ((feat_diff < kHistBins * 0.1f) && (feat_diff >= 0.0f)) bump(2 * kHistBins + (int)(feat_diff / 0.1f));
                }
                STI(Y::S_COUNTDOWN) = countdown;
            } else {
                // The three histograms are walked in bin order (the float sums are order-sensitive), but only their
                // NON-ZERO bins: an empty bin adds +0 to every sum and can never be a peak (h > max needs h > 0), and 500
                // samples leave most of the 1000 bins empty.  64 bins per coalesced load, occupied ones found by ballot.
                // (a macro, not a lambda: with the accumulators captured by reference through two lambda levels the
                // compiler kept them in scratch memory)
                // The histograms reach the wave in ONE round trip: 500 dwords each as eight coalesced loads per lane, issued together,
                // then staged through the (dead at this point) FFT work array 2 L bins at a time and walked from LDS.
                // Walking it with one dependent 64-bin load per step cost 48 serial HBM round trips per update: ~150 us for the
                // one stream in 500 that updates in a given frame -- invisible while every stream of a batch has the same age
                // (one slow launch in 500), but with streams of all ages (handles created at different ticks, the normal case
                // for many handles) some wave of EVERY launch is in its update, and the launch lasts as long as its slowest wave:
                // ns_kernel 0.30 -> 0.45 ms at 65 536 streams joined over 256 ticks (tools_dev/ns_age_exp.py, round 4).
                // all three histograms are requested up front (24 registers for the length of the update): one round trip in all
                unsigned hc_[3][8];
#pragma unroll
                for (int w_ = 0; w_ < 3; w_++) {
                    const unsigned *hsrc_ = reinterpret_cast<const unsigned *>(hist + w_ * kHistBins);
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const int d_ = 64 * q + lane;
                        const unsigned v_ = hsrc_[d_ < kHistBins / 2 ? d_ : 0];
                        hc_[w_][q] = d_ < kHistBins / 2 ? v_ : 0u;  // bins past 999 are staged as zeros
                    }
                }
#define NS_WALK(which, ...)                                                                                        \
    {                                                                                                              \
        constexpr int CH_ = 2 * L, QPC_ = L / 64, NCH_ = 8 / QPC_; /* bins staged at a time (the work array holds L floats) */ \
        const unsigned(&c_)[8] = hc_[which];                                                                       \
        _Pragma("unroll") for (int part_ = 0; part_ < NCH_; part_++) {                                             \
            unsigned *stage_ = reinterpret_cast<unsigned *>(W.fa);                                                 \
            _Pragma("unroll") for (int q = 0; q < QPC_; q++) stage_[64 * q + lane] = c_[QPC_ * part_ + q];         \
            wave_sync();                                                                                           \
            const unsigned short *h16_ = reinterpret_cast<const unsigned short *>(W.fa);                           \
            _Pragma("unroll 1") for (int base = CH_ * part_; base < CH_ * part_ + CH_ && base < kHistBins; base += 64) { \
                const int hv = (int)h16_[base - CH_ * part_ + lane];                                               \
                unsigned long long m = __builtin_amdgcn_ballot_w64(hv != 0);                                       \
                while (m) {                                                                                        \
                    const int j = __builtin_ctzll(m);                                                              \
                    m &= m - 1;                                                                                    \
                    const int i = base + j, h = __builtin_amdgcn_readlane(hv, j);                                  \
                    __VA_ARGS__                                                                                    \
                }                                                                                                  \
            }                                                                                                      \
            wave_sync();                                                                                           \
        }                                                                                                          \
    }
                static_assert(L == 128 || L == 256, "the staging above is sized for these");
                float avg = 0.0f, avg_compl = 0.0f, avg_sq = 0.0f;
                int num = 0;
                NS_WALK(0, {
                    const float mid = ((float)i + 0.5f) * 0.1f;
                    if (mid <= 1.f) {
                        avg += h * mid;
                        num += h;
                    }
                    avg_sq += h * mid * mid;
                    avg_compl += h * mid;
                })
                if (num > 0) avg = avg / ((float)num);
                avg_compl = avg_compl / ((float)kUpdateWindow);
                avg_sq = avg_sq / ((float)kUpdateWindow);
                const float fluct = avg_sq - avg * avg_compl;
                if (fluct < 0.05f) {
                    thr_lrt = 1.f;
                } else {
                    thr_lrt = 1.2f * avg;
                    if (thr_lrt < 0.2f) thr_lrt = 0.2f;
                    if (thr_lrt > 1.f) thr_lrt = 1.f;
                }
                // the two peaks of the flatness / difference histograms
                int w1a = 0, w2a = 0, w1b = 0, w2b = 0;
                float p1a = 0.0f, p2a = 0.0f, p1b = 0.0f, p2b = 0.0f;
                {
                    int max1 = 0, max2 = 0;
                    NS_WALK(1, {
                        const float mid = ((float)i + 0.5f) * 0.05f;
                        if (h > max1) {
                            max2 = max1, w2a = w1a, p2a = p1a;
                            max1 = h, w1a = h, p1a = mid;
                        } else if (h > max2) {
                            max2 = h, w2a = h, p2a = mid;
                        }
                    })
                }
                {
                    int max1 = 0, max2 = 0;
                    NS_WALK(2, {
                        const float mid = ((float)i + 0.5f) * 0.1f;
                        if (h > max1) {
                            max2 = max1, w2b = w1b, p2b = p1b;
                            max1 = h, w1b = h, p1b = mid;
                        } else if (h > max2) {
                            max2 = h, w2b = h, p2b = mid;
                        }
                    })
                }
#undef NS_WALK
                const int thres_weight = 150;  // (int)(0.3 * 500)
                int use_flat = 1, use_diff = 1;
                if ((fabsf(p2a - p1a) < 2 * 0.05f) && (w2a > 0.5f * w1a)) {
                    w1a += w2a;
                    p1a = 0.5f * (p1a + p2a);
                }
                if (w1a < thres_weight || p1a < 0.6f) use_flat = 0;
                if (use_flat == 1) {
                    thr_flat = 0.9f * p1a;
                    if (thr_flat < 0.1f) thr_flat = 0.1f;
                    if (thr_flat > 0.95f) thr_flat = 0.95f;
                }
                if ((fabsf(p2b - p1b) < 2 * 0.1f) && (w2b > 0.5f * w1b)) {
                    w1b += w2b;
                    p1b = 0.5f * (p1b + p2b);
                }
                thr_diff = 1.2f * p1b;
                if (w1b < thres_weight) use_diff = 0;
                if (thr_diff < 0.16f) thr_diff = 0.16f;
                if (thr_diff > 1.f) thr_diff = 1.f;
                if (fluct < 0.05f) use_diff = 0;
                const float fsum = (float)(1 + use_flat + use_diff);
                w_lrt = 1.f / fsum;
                w_flat = ((float)use_flat) / fsum;
                w_diff = ((float)use_diff) / fsum;
                wave_sync();
                for (int i = lane; i < 3 * kHistBins; i += 64) hist[i] = 0;
                STI(Y::S_COUNTDOWN) = kUpdateWindow;
                if (update_flag == 1) {
                    STI(Y::S_UPDATE_FLAG) = 0;
                } else {
                    feat_acc = feat_acc / ((float)kUpdateWindow);
                    feat_norm = 0.5f * (feat_acc + feat_norm);
                    feat_acc = 0.f;
                }
                ST(Y::S_THR_LRT) = thr_lrt;
                ST(Y::S_THR_FLAT) = thr_flat;
                ST(Y::S_THR_DIFF) = thr_diff;
                ST(Y::S_W_LRT) = w_lrt;
                ST(Y::S_W_FLAT) = w_flat;
                ST(Y::S_W_DIFF) = w_diff;
            }
        }
        ST(Y::S_FEAT_FLAT) = feat_flat;
        ST(Y::S_FEAT_DIFF) = feat_diff;
        ST(Y::S_FEAT_NORM) = feat_norm;
        ST(Y::S_FEAT_ACC) = feat_acc;

        NS_RELANE();
        // ---- SpeechNoiseProb (ns_core.c:642-749)
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int b0 = lane + 64 * k;
            const bool ok = (64 * k + 63 < M) || b0 < M;  // see the noise estimation loop
            const int b = ok ? b0 : M - 1;
            const float sp = W.snrp[b];
            const float t1 = 1.f + 2.f * sp;
            const float t2 = div_ordinary(2.f * sp, t1 + 0.0001f);  // 0 <= sp < 2^38 (a magnitude over a denominator >= 1e-4), t1 >= 1
            const float bessel = (t_snrq[k] + 1.f) * t2;
            float v = pf_lrt[k];
            v += 0.5f * (bessel - fast_log_ge1(t1, K.lm) - v);
            if (!ok) continue;
            ST(Y::LOG_LRT + b) = v;
            W.r0[b] = v;
        }
        wave_sync();
        NS_PROF(7);
        float ksum = sum_range(W.r0, 0, M, lane);
        ksum = ksum / (float)(M);
        feat_lrt = ksum;
        ST(Y::S_FEAT_LRT) = feat_lrt;
        float prior = SCF(S_PRIOR);
        {
            float width = 4.0f;
            if (ksum < thr_lrt) width = 2.f * 4.0f;
            const float a0 = width * (ksum - thr_lrt);
            width = 4.0f;
            if (feat_flat > thr_flat) width = 2.f * 4.0f;
            const float a1 = (float)1 * width * (thr_flat - feat_flat);
            width = 4.0f;
            if (feat_diff < thr_diff) width = 2.f * 4.0f;
            const float a2 = width * (feat_diff - thr_diff);
            // the three indicator tanh's are wave-uniform scalars: evaluate them in lanes 0..2 of one call
            const float th = fast_tanh(lane == 1 ? a1 : (lane == 2 ? a2 : a0), K.lm);
            const float ind0 = 0.5f * (lane_value(th, 0) + 1.f);
            const float ind1 = 0.5f * (lane_value(th, 1) + 1.f);
            const float ind2 = 0.5f * (lane_value(th, 2) + 1.f);
            const float ind = w_lrt * ind0 + w_flat * ind1 + w_diff * ind2;
            prior += 0.1f * (ind - prior);
            if (prior > 1.f) prior = 1.f;
            if (prior < 0.01f) prior = 0.01f;
            ST(Y::S_PRIOR) = prior;
        }
        const float gain_prior = (1.f - prior) / (prior + 0.0001f);
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int b0 = lane + 64 * k;
            const bool ok = (64 * k + 63 < M) || b0 < M;
            const int b = ok ? b0 : M - 1;
            float inv = fast_exp(-W.r0[b], K.lm);
            inv = gain_prior * inv;
            const float spb = 1.f / (1.f + inv);
            if (ok) W.sprob[b] = spb;
        }
        wave_sync();
        NS_PROF(8);
        NS_RELANE();
        // ---- UpdateNoiseEstimate (ns_core.c:800-846): bin i starts from the gamma chosen at bin i-1.
        //      Then Process: initMagnEst, DD Wiener filter + flooring / start-up blend (ns_core.c:1277-1315), IFFT packing.
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int b0 = lane + 64 * k;
            const bool ok = (64 * k + 63 < M) || b0 < M;  // see the noise estimation loop
            const int b = ok ? b0 : M - 1;
            const float ps = W.sprob[b], pn = 1.f - ps, mg = W.magn[b], np = pf_nprev[k];
            float gamma_old = 0.9f;
            if (b > 0 && W.sprob[b - 1] > 0.2f) gamma_old = 0.99f;
            const float tmp = gamma_old * np + (1.f - gamma_old) * (pn * mg + ps * np);
            float gamma = 0.9f;
            if (ps > 0.2f) gamma = 0.99f;
            if (ps < 0.2f) {
                float pz = W.pause[b];
                pz += 0.05f * (mg - pz);
                if (ok) ST(Y::MAGN_AVG_PAUSE + b) = pz;
            }
            float nz;
            if (gamma == gamma_old) {
                nz = tmp;
            } else {
                nz = gamma * np + (1.f - gamma) * (pn * mg + ps * np);
                if (tmp < nz) nz = tmp;
            }
            // ---- Process
            float init_est = 0.f;
            if (startup) {
                init_est = ST(Y::INIT_MAGN + b) + mg;
                if (ok) ST(Y::INIT_MAGN + b) = init_est;
            }
            float cur = 0.f;
            if (mg > nz) cur = div_ordinary(mg, nz + 0.0001f) - 1.f;
            const float snr = 0.98f * t_prev[k] + (1.f - 0.98f) * cur;
            float f = div_ordinary(snr, overdrive + snr);  // 0 <= snr < 2^38, overdrive >= 1
            if (f < denoise_bound) f = denoise_bound;
            if (f > 1.f) f = 1.f;
            if (startup) {
                float ft = (init_est - overdrive * W.r1[b]);
                ft /= (init_est + 0.0001f);
                if (ft < denoise_bound) ft = denoise_bound;
                if (ft > 1.f) ft = 1.f;
                f *= (float)(block_ind);
                ft *= (float)(kStartupShort - block_ind);
                f += ft;
                f /= (float)(kStartupShort);
            }
            const float re = t_re[k] * f, im = t_im[k] * f;
            if (!ok) continue;
            ST(Y::SMOOTH + b) = f;
            ST(Y::MAGN_PREV + b) = mg;
            ST(Y::NOISE_PREV + b) = nz;
            W.snrp[b] = f;  // the filter, for the high-band gain
            if (b == 0)
                W.fa[0] = re;
            else if (b == M - 1)
                W.fa[1] = re;
            else {
                W.fa[2 * b] = re;
                W.fa[2 * b + 1] = im;
            }
        }
        wave_sync();
        NS_PROF(9);
        NS_RELANE();
        rdft_inverse<NC>(W.fa, &K.tab, lane);
        NS_PROF(10);
        float td[NT];
#pragma unroll
        for (int k = 0; k < NT; k++) {
            const int i = lane + 64 * k;
            td[k] = W.fa[i] * (2.f / L);
            tdst[i] = td[k] * td[k];
        }
        wave_sync();
        NS_PROF(11);
        float factor = 1.f;
        if (block_ind > kStartupLong) {  // gainmap == 1 for policy 2
            const float energy2 = sum_range(tdst, 0, L, lane);
            float gain = sqrtf(energy2 / (energy1 + 1.f));
            float factor1 = 1.f, factor2 = 1.f;
            if (gain > 0.5f) {
                factor1 = 1.f + 1.3f * (gain - 0.5f);
                if (gain * factor1 > 1.f) factor1 = 1.f / gain;
            }
            if (gain < 0.5f) {
                if (gain <= denoise_bound) gain = denoise_bound;
                factor2 = 1.f - 0.3f * (0.5f - gain);
            }
            factor = prior * factor1 + (1.f - prior) * factor2;
        }
#pragma unroll
        for (int k = 0; k < NT; k++) {
            const int i = lane + 64 * k;
            const float w = K.window[i] * td[k];
            synt[k] += factor * w;
        }
        // ---- high band: time-domain gain (ns_core.c:1362-1414), only when chn == 2
        if (chn == 2) {
            constexpr int D = M / 4;
            float avg_prob = sum_range(W.sprob, M - D - 1, M - 1, lane);
            avg_prob = avg_prob / ((float)D);
            // magnPrevAnalyze == magnPrevProcess here, so sumMagnProcess / sumMagnAnalyze is x / x
            avg_prob *= sum_magn / sum_magn;
            float avg_gain = sum_range(W.snrp, M - D - 1, M - 1, lane);
            avg_gain = avg_gain / ((float)D);
            const float t = 2.f * avg_prob - 1.f;
            const float gain_mod = 0.5f * (1.f + tanh_call(1.0f * t, &K.lm));
            float g = 0.5f * gain_mod + 0.5f * avg_gain;
            if (avg_prob >= 0.5f) g = 0.25f * gain_mod + 0.75f * avg_gain;
            g = g * 1.0f;
            if (g < denoise_bound) g = denoise_bound;
            if (g > 1.f) g = 1.f;
            hb_gain = g;
        }
        wave_sync();
        NS_PROF(12);
    }

        NS_RELANE();
    // ---- read out the finished segment, slide the synthesis buffer (ns_core.c:1245-1251 / 1347-1359)
#pragma unroll
    for (int k = 0; k < NT; k++) {
        const int i = lane + 64 * k;
        if (i >= B) ST(Y::SYNT_BUF + i - B) = synt[k];
        tdst[i] = sat16f(synt[k]);
        // HB output: the OLDEST block of the (already slid) high-band buffer, times the gain (zero-energy frames pass
        // it through unscaled, ns_core.c:1255-1265)
        if (chn == 2) W.fa[i] = zero_frame ? sat16f(hb[k]) : sat16f(hb_gain * hb[k]);
    }
    wave_sync();
    // interleave + (int16_t) cast (src/webrtc.c:640-642).  Samples beyond the core's block length (32 kHz: 160..319)
    // are the wrapper's calloc zeros.
    for (int i = lane; i < pkg; i += 64) {
        const float lo = (i < B) ? tdst[i] : 0.f;
        out[i * chn] = (int16_t)lo;
        if (chn == 2) out[i * chn + 1] = (int16_t)((i < B) ? W.fa[i] : 0.f);
    }
    wave_sync();
    NS_PROF(13);
#undef NS_RELANE
#undef SCF
#undef SCI
#undef ST
#undef STI
}

// Register budget = the occupancy the LDS allows: 16 kHz / 32 kHz streams (28.8 KB per workgroup) five workgroups per CU, 8 kHz
// streams (16.7 KB) nine, more than the seven waves per SIMD a 72-register budget gives.
#ifndef WMX_NS_W128
#define WMX_NS_W128 7  // 6: 0.409, 7: 0.393, 8: 0.389 ms (131 072 streams; at 8 the 2-channel variant spills)
#endif
#ifndef WMX_NS_W256
#define WMX_NS_W256 5
#endif
template <int L>
struct NsOcc {
    static constexpr int kWaves = L == 128 ? WMX_NS_W128 : WMX_NS_W256;
};
template <int L, int CHN>
__global__ __launch_bounds__(64 * kNsWavesPerBlock) __attribute__((amdgpu_waves_per_eu(NsOcc<L>::kWaves, NsOcc<L>::kWaves))) void ns_kernel(float *__restrict__ state, unsigned short *__restrict__ hists,
                                                                      const float *__restrict__ consts, const int16_t *in, int16_t *out,
                                                                      int n_streams, int n_packets, long stream_stride,
                                                                      long packet_stride, int pkg, const uint8_t *__restrict__ active) {
    using Y = NsLayout<L>;
    __shared__ NsConstLds<L> K;
    __shared__ NsWaveLds<L> Wv[kNsWavesPerBlock];
    // constants: FftTables | window[L] | logi[MP] | libm tables (the host builds the same struct)
    {
        float *dst = reinterpret_cast<float *>(&K);
        constexpr int NCONST = sizeof(NsConstLds<L>) / 4, NIT = (NCONST + 64 * kNsWavesPerBlock - 1) / (64 * kNsWavesPerBlock);
        // every load requested before the first store (a rolled copy loop waits for each L2 round trip in turn)
        float c[NIT];
#pragma unroll
        for (int k = 0; k < NIT; k++) {
            const int i = threadIdx.x + 64 * kNsWavesPerBlock * k;
            c[k] = consts[i < NCONST ? i : 0];
        }
#pragma unroll
        for (int k = 0; k < NIT; k++) {
            const int i = threadIdx.x + 64 * kNsWavesPerBlock * k;
            if (i < NCONST) dst[i] = c[k];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: state pointers become scalar bases
    const int sidx = blockIdx.x * kNsWavesPerBlock + wave;  // one stream per wave; no block-level barrier below
    if (!stream_active(active, sidx, n_streams)) return;  // no stream, or one that is switched off: state and PCM rows untouched
    float *st = state + (size_t)sidx * Y::WORDS;
    unsigned short *hist = hists + (size_t)sidx * 3 * kHistBins;
    {
        // the tail [M, MP) of every per-bin array is summed by sum_lanes and must be zero; nothing below writes it
        float *wb = Wv[wave].magn;
        constexpr int NARR = 9;  // magn .. r2
        if (lane < NARR * (Y::MP - Y::M)) wb[(lane / (Y::MP - Y::M)) * Y::MP + Y::M + lane % (Y::MP - Y::M)] = 0.f;
    }
#ifdef WMX_NS_PROF
    if (lane < 16) Wv[wave].prof[lane] = 0;
#endif
    for (int p = 0; p < n_packets; p++) {
        const size_t off = (size_t)sidx * stream_stride + (size_t)p * packet_stride;
        ns_frame<L, CHN>(K, Wv[wave], st, hist, in + off, out + off, pkg, lane);
    }
#ifdef WMX_NS_PROF
    if (lane < 16) atomicAdd(&g_ns_prof[lane], Wv[wave].prof[lane]);
#endif
}
#ifdef WMX_NS_PROF
extern "C" int wmx_debug_ns_prof(unsigned long long *out16, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_ns_prof), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ns_prof), z, sizeof(z));
    }
    return 0;
}
#endif

}  // namespace
}  // namespace wmx

// ------------------------------------------------------------------------------------ host
struct wmx_ns {
    int device;  // the HIP device the state lives on (current device at create); every entry point switches to it
    int n_streams, chn, freq, L, pkg;
    float *d_state;
    unsigned short *d_hist;
    float *d_consts;
    float *d_tmpl;  // the state ns_init gives a stream (reset_streams refills from it)
    size_t words;
    wmx::StreamLife life;
};

namespace {

template <int L>
void build_ns_template(std::vector<float> &st, std::vector<float> &consts) {
    using Y = wmx::NsLayout<L>;
    st.assign(Y::WORDS, 0.f);
    int *sti = reinterpret_cast<int *>(st.data());
    // WebRtcNs_InitCore ns_core.c:74-214
    for (int q = 0; q < 3; q++)
        for (int b = 0; b < Y::M; b++) {
            st[Y::LQUANTILE + q * Y::MP + b] = 8.f;
            st[Y::DENSITY + q * Y::MP + b] = 0.3f;
        }
    for (int q = 0; q < 3; q++) sti[Y::S_COUNTER + q] = (int)floor((float)(200 * (q + 1)) / (float)3);
    for (int b = 0; b < Y::M; b++) {
        st[Y::SMOOTH + b] = 1.f;
        st[Y::LOG_LRT + b] = 0.5f;
    }
    sti[Y::S_UPDATES] = 0;
    sti[Y::S_BLOCK_IND] = -1;
    sti[Y::S_UPDATE_FLAG] = 2;
    sti[Y::S_COUNTDOWN] = 500;
    st[Y::S_PRIOR] = 0.5f;
    st[Y::S_FEAT_FLAT] = 0.5f;
    st[Y::S_FEAT_LRT] = 0.5f;
    st[Y::S_FEAT_DIFF] = 0.5f;
    st[Y::S_THR_LRT] = 0.5f;
    st[Y::S_THR_FLAT] = 0.5f;
    st[Y::S_THR_DIFF] = 0.5f;
    st[Y::S_W_LRT] = 1.f;
    // constants block: FftTables | window | log table (+ the two data-independent start-up sums)
    static_assert(offsetof(wmx::NsConstLds<L>, lm) == (wmx::kFftTableWords + L + Y::MP) * 4, "constants block layout");
    consts.assign(sizeof(wmx::NsConstLds<L>) / 4, 0.f);
    wmx::ns_libm_tables(reinterpret_cast<wmx::NsLibmTables *>(consts.data() + wmx::kFftTableWords + L + Y::MP));
    wmx::FftTables tab;
    wmx::fft_tables_ooura(L, &tab);
    std::memcpy(consts.data(), &tab, sizeof(tab));
    float *win = consts.data() + wmx::kFftTableWords;
    const int ramp = (L == 128) ? 48 : 96;
    const double half_pi = 1.5707963267948966;
    for (int i = 0; i < L; i++) {
        // windows_private.h:64,94: 8-decimal literals of a sine ramp / flat top / mirrored ramp
        double v = (i < ramp) ? sin(half_pi * i / ramp) : (i < L - ramp ? 1.0 : sin(half_pi * (L - i) / ramp));
        win[i] = (float)(floor(v * 1e8 + 0.5) / 1e8);
    }
    float *logi = win + L;
    volatile float acc1 = 0.0f, acc2 = 0.0f;
    for (int i = 0; i < Y::M; i++) {
        float li = (i == 0) ? 0.f : (float)log((double)(float)i);  // ns_core.c:1093
        logi[i] = li;
        if (i >= 5) {
            acc1 = acc1 + li;
            volatile float sq = li * li;
            acc2 = acc2 + sq;
        }
    }
    logi[Y::MP - 2] = acc1;  // sum_log_i        (ns_core.c:1094)
    logi[Y::MP - 1] = acc2;  // sum_log_i_square (ns_core.c:1095)
}

__global__ void ns_fill_state(float *state, const float *tmpl, int words, int n_streams) {
    const size_t total = (size_t)words * n_streams;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        state[i] = tmpl[i % words];
}

}  // namespace

extern "C" {

int wmx_ns_create(wmx_ns **out, int n_streams, int chn, int freq) {
    if (!out) return WMX_EINVAL;
    *out = nullptr;
    // ns_init: freq <= 32000 and a multiple of 8000 (src/webrtc.c:563-564); in[2]/out[2] => chn <= 2
    if ((freq != 8000 && freq != 16000 && freq != 32000) || chn < 1 || chn > 2 || n_streams < 1) {
        wmx::set_error("wmx_ns_create: unsupported n_streams=%d chn=%d freq=%d", n_streams, chn, freq);
        return WMX_EINVAL;
    }
    wmx_ns *h = new wmx_ns();
    if ((h->device = wmx::current_device()) < 0) {
        delete h;
        return WMX_ENODEV;
    }
    h->n_streams = n_streams;
    h->chn = chn;
    h->freq = freq;
    h->L = (freq == 8000) ? 128 : 256;
    h->pkg = freq / 1000 * 10;
    std::vector<float> st, consts;
    if (h->L == 128)
        build_ns_template<128>(st, consts);
    else
        build_ns_template<256>(st, consts);
    h->words = st.size();
    hipError_t e;
#define NS_TRY(x)                                                  \
    if ((e = (x)) != hipSuccess) {                                 \
        int rc = wmx::hip_fail(e, #x, __FILE__, __LINE__);         \
        wmx_ns_destroy(h);                                         \
        return rc;                                                 \
    }
    NS_TRY(hipMalloc(&h->d_state, h->words * sizeof(float) * (size_t)n_streams));
    NS_TRY(hipMalloc(&h->d_hist, (size_t)n_streams * 3 * 1000 * sizeof(unsigned short)));
    NS_TRY(hipMalloc(&h->d_consts, consts.size() * sizeof(float)));
    NS_TRY(hipMalloc(&h->d_tmpl, st.size() * sizeof(float)));
    NS_TRY(hipMemcpy(h->d_consts, consts.data(), consts.size() * sizeof(float), hipMemcpyHostToDevice));
    NS_TRY(hipMemcpy(h->d_tmpl, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice));
    NS_TRY(hipMemset(h->d_hist, 0, (size_t)n_streams * 3 * 1000 * sizeof(unsigned short)));
    hipLaunchKernelGGL(ns_fill_state, dim3(1024), dim3(256), 0, nullptr, h->d_state, h->d_tmpl, (int)h->words, n_streams);
    NS_TRY(hipGetLastError());
    NS_TRY(hipDeviceSynchronize());
#undef NS_TRY
    *out = h;
    return 0;
}

int wmx_ns_destroy(wmx_ns *h) {
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

// ns_release + ns_init for the listed streams (src/webrtc.c:560-602, 646-661): WebRtcNs_InitCore state, empty histograms
int wmx_ns_reset_streams(wmx_ns *h, const int32_t *idx, int n, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h || n < 0 || (n > 0 && !idx)) return WMX_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = wmx::as_stream(stream);
    const int32_t *d_idx = nullptr;
    const int rc = h->life.upload(idx, n, h->n_streams, s, &d_idx);
    if (rc != 0) return rc;
    const unsigned grid = (unsigned)(n < 4096 ? n : 4096);
    hipLaunchKernelGGL((wmx::fill_rows_idx<float>), dim3(grid), dim3(256), 0, s, h->d_state, (const float *)h->d_tmpl, (int)h->words, d_idx, n);
    hipLaunchKernelGGL((wmx::fill_rows_idx<unsigned short>), dim3(grid), dim3(256), 0, s, h->d_hist, (const unsigned short *)nullptr, 3000, d_idx, n);
    WMX_LAUNCH_CHECK();
    return h->life.done(s);
}

int wmx_ns_set_active(wmx_ns *h, const uint8_t *host_mask, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    return h->life.set_active(h->n_streams, host_mask, wmx::as_stream(stream));
}

// stream migration: [header | state words | 3 x 1000 histogram counters]
static constexpr uint32_t kNsBlobVersion = 1;  // bump when the meaning of a state word changes (wmx_internal.h: blob_layout)
int wmx_ns_stream_state_bytes(const wmx_ns *h) { return h ? (int)(sizeof(wmx::BlobHeader) + h->words * 4 + 3000 * 2) : WMX_EINVAL; }

int wmx_ns_export_stream(wmx_ns *h, int stream_index, void *host_blob) {
    WMX_ON_DEVICE(h);
    if (!h || !host_blob || stream_index < 0 || stream_index >= h->n_streams) return WMX_EINVAL;
    WMX_HIP(hipDeviceSynchronize());
    char *p = static_cast<char *>(host_blob);
    wmx::blob_begin(p, wmx::blob_tag("NS  "), wmx::blob_layout((uint32_t)h->L, kNsBlobVersion), (uint32_t)(h->words * 4 + 6000));
    p += sizeof(wmx::BlobHeader);
    WMX_HIP(hipMemcpy(p, h->d_state + (size_t)stream_index * h->words, h->words * 4, hipMemcpyDeviceToHost));
    WMX_HIP(hipMemcpy(p + h->words * 4, h->d_hist + (size_t)stream_index * 3000, 6000, hipMemcpyDeviceToHost));
    return 0;
}

int wmx_ns_import_stream(wmx_ns *h, int stream_index, const void *host_blob) {
    WMX_ON_DEVICE(h);
    if (!h || !host_blob || stream_index < 0 || stream_index >= h->n_streams) return WMX_EINVAL;
    const int rc = wmx::blob_check(host_blob, wmx::blob_tag("NS  "), wmx::blob_layout((uint32_t)h->L, kNsBlobVersion), (uint32_t)(h->words * 4 + 6000));
    if (rc) return rc;
    WMX_HIP(hipDeviceSynchronize());
    const char *p = static_cast<const char *>(host_blob) + sizeof(wmx::BlobHeader);
    WMX_HIP(hipMemcpy(h->d_state + (size_t)stream_index * h->words, p, h->words * 4, hipMemcpyHostToDevice));
    WMX_HIP(hipMemcpy(h->d_hist + (size_t)stream_index * 3000, p + h->words * 4, 6000, hipMemcpyHostToDevice));
    return 0;
}

int wmx_ns_packet_samples(const wmx_ns *h) { return h ? h->pkg * h->chn : WMX_EINVAL; }
int wmx_ns_state_words(const wmx_ns *h) { return h ? (int)h->words : WMX_EINVAL; }

int wmx_ns_export_state(const wmx_ns *h, int stream_index, float *host_words, unsigned short *host_hist) {
    WMX_ON_DEVICE(h);
    if (!h || stream_index < 0 || stream_index >= h->n_streams) return WMX_EINVAL;
    WMX_HIP(hipDeviceSynchronize());
    if (host_words)
        WMX_HIP(hipMemcpy(host_words, h->d_state + (size_t)stream_index * h->words, h->words * sizeof(float), hipMemcpyDeviceToHost));
    if (host_hist)
        WMX_HIP(hipMemcpy(host_hist, h->d_hist + (size_t)stream_index * 3000, 3000 * sizeof(unsigned short), hipMemcpyDeviceToHost));
    return 0;
}

int wmx_ns_process(wmx_ns *h, const int16_t *d_in, int16_t *d_out, int n_packets, long stream_stride,
                   long packet_stride, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h || n_packets < 0) {
        wmx::set_error("wmx_ns_process: bad argument");
        return WMX_EINVAL;
    }
    if (n_packets == 0) return 0;  // frameNum == 0: the reference's loop does not run, whatever the pointers are
    if (!d_in || !d_out) {
        wmx::set_error("wmx_ns_process: null buffer");
        return WMX_EINVAL;
    }
    const int per_pkt = h->pkg * h->chn;
    if (packet_stride < per_pkt || (h->n_streams > 1 && stream_stride < per_pkt)) {
        // a packet must not overlap its neighbours; streams may be packet- or stream-major
        wmx::set_error("wmx_ns_process: strides (%ld, %ld) smaller than a packet (%d samples)", stream_stride, packet_stride, per_pkt);
        return WMX_EINVAL;
    }
    const unsigned grid = (unsigned)((h->n_streams + wmx::kNsWavesPerBlock - 1) / wmx::kNsWavesPerBlock);
    hipStream_t s = wmx::as_stream(stream);
#define NS_LAUNCH(LL, CC)                                                                                                  \
    hipLaunchKernelGGL((wmx::ns_kernel<LL, CC>), dim3(grid), dim3(64 * wmx::kNsWavesPerBlock), 0, s, h->d_state, h->d_hist, \
                       h->d_consts, d_in, d_out, h->n_streams, n_packets, stream_stride, packet_stride, h->pkg, h->life.d_active)
    if (h->L == 128) {
        if (h->chn == 1)
            NS_LAUNCH(128, 1);
        else
            NS_LAUNCH(128, 2);
    } else {
        if (h->chn == 1)
            NS_LAUNCH(256, 1);
        else
            NS_LAUNCH(256, 2);
    }
#undef NS_LAUNCH
    WMX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// Developer / test hook: div_ordinary (libm_dev.h) beside the compiler's a / b on the device, element by element.
namespace wmx {
namespace {
__global__ void div_debug_kernel(const float *a, const float *b, float *q_ordinary, float *q_ieee, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        q_ordinary[i] = div_ordinary(a[i], b[i]);
        q_ieee[i] = a[i] / b[i];
    }
}
}  // namespace
}  // namespace wmx
extern "C" int wmx_debug_div(const float *d_a, const float *d_b, float *d_q_ordinary, float *d_q_ieee, size_t n, void *stream) {
    if (!d_a || !d_b || !d_q_ordinary || !d_q_ieee) return WMX_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(wmx::div_debug_kernel, dim3(4096), dim3(256), 0, wmx::as_stream(stream), d_a, d_b, d_q_ordinary, d_q_ieee, n);
    WMX_LAUNCH_CHECK();
    return 0;
}
// the same function compiled for the host (reciprocal estimate = the rounded 1 / b, moved by `seed_ulps` ulps: the device's
// v_rcp_f32 is good to one ulp, and the quotient must not depend on which estimate it was)
extern "C" int wmx_debug_div_host(const float *a, const float *b, float *q, size_t n) {
    if (!a || !b || !q) return WMX_EINVAL;
    for (size_t i = 0; i < n; i++) q[i] = wmx::div_ordinary(a[i], b[i]);
    return 0;
}

// Host-side evaluation of the NS's table-driven log / exp (libm_dev.h) -- the same source the kernels compile, run on
// the CPU so that the `-m "not gpu"` tests can sweep millions of arguments against glibc.  kind 0: log (x >= 1), 1: exp, 2: tanh.
extern "C" int wmx_debug_ns_libm(int kind, const float *x, float *y, size_t n) {
    static wmx::NsLibmTables tab;
    static bool init = false;
    if (!init) {
        wmx::ns_libm_tables(&tab);
        init = true;
    }
    if (!x || !y || kind < 0 || kind > 2) return WMX_EINVAL;
    for (size_t i = 0; i < n; i++)
        y[i] = kind == 0 ? wmx::fast_log_ge1(x[i], tab) : (kind == 1 ? wmx::fast_exp(x[i], tab) : wmx::fast_tanh(x[i], tab));
    return 0;
}
