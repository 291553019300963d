// aecm.hip -- batched FIXED-POINT echo canceller (AECM) for gfx950: one wavefront per stream.
//
// What the reference runs when src/webrtc.c is built with its AECM switch (`#undef MAKE_WEBRTC_AEC`, src/webrtc.c:168-191):
// aec_setFrameFar / aec_process / aec_process2 over WebRtcAecm_BufferFarend / WebRtcAecm_Process
// (W:modules/audio_processing/aecm/echo_control_mobile.c), WebRtcAecm_ProcessFrame / ProcessBlock
// (aecm_core.c:569-664, aecm_core_c.c:280-639: TimeToFrequencyDomain, CalcEnergies, CalcStepSize, UpdateChannel,
// CalcSuppressionGain, the Wiener / NLP gains, ComfortNoise, InverseFFTAndWindow) and the binary-spectrum delay estimator
// (W:modules/audio_processing/utility/delay_estimator.c:393-487, delay_estimator_wrapper.c:52-78).
//
// Split of the work, as in aec.hip.  The control plane (start-up phase, far-end buffer bookkeeping, 80 -> 64 re-blocking)
// is data independent and runs on the host (aecm_ctl.h -> AecmPlan).  `aecm_far_kernel` (one wave per batch) does
// everything that depends on the SHARED far-end only: far ring, re-blocking, the far spectrum |X| with its Q-domain, the
// binary far spectrum; it appends them to a 256-slot history that replaces the reference's per-handle far_history /
// binary_far_history (index = absolute block number, the reference's `history_pos - delay`).  `aecm_near_kernel` (one
// wave per stream, 4 streams per workgroup) does the per-stream part: near spectrum, the delay estimate (100 history
// comparisons = two per lane, argmin by a wave reduction that keeps the reference's first-minimum rule), channel
// estimation, gains, comfort noise, inverse transform.  Lane k owns bin k; lane 0 also bin 64.  Every sum is an integer
// sum (order-free), so the kernel is bit-exact by construction; the FFT is the SPL radix-2 fixed-point transform of
// spl_fx.h, order 7.  State: 3.4 KB per stream, one contiguous block moved through LDS.
//
// One reference quirk is pinned to its well-defined reading: at 8 kHz with 20 ms packets the second 10 ms frame of a
// packet replays farendOld[1], which WebRtcAecm_Init leaves uninitialised (it clears 160 bytes, not 160 samples,
// echo_control_mobile.c:211); here it starts as zeros, what a fresh heap gives (tests/golden/make_aecm_golden.py).
#include <unordered_map>
#include <vector>
#include "wmx_internal.h"
#include "spl_fx.h"
#include "aecm_ctl.h"
#include "fx_tables.h"

namespace wmx {
namespace {

constexpr int kAecmWavesPerBlock = 4;
constexpr int kAecmBP = 68;        // 65 bins padded
constexpr int kAecmMaxDelay = 100;  // MAX_DELAY, aecm_defines.h:26

struct alignas(16) AecmConsts {
    SplTwiddles tw;  // packed twiddle pairs of the SPL FFT (spl_fx.h), both directions
    int16_t sqrt_hanning[66];
    int16_t cos360[360], sin360[360];
    int16_t pad[2];
    uint32_t lcg_a[64], lcg_c[64];  // seed after k+1 steps = lcg_a[k] * seed + lcg_c[k]  (mod 2^31)
};
static_assert(sizeof(AecmConsts) % 16 == 0, "AecmConsts is copied in 16-byte pieces");

// shared far-end state of a batch (device memory)
struct AecmFarBufs {
    int16_t *ring;      // [kAecmFarRing]           farendBuf
    int16_t *old;       // [2][80]                  farendOld
    int16_t *frame;     // [kAecmFrameRing]         farFrameBuf
    int16_t *x_prev;    // [64]                     xBuf[0..63]
    int32_t *mean_far;  // [32] + initialised flag  mean_far_spectrum of bins 12..43, far_spectrum_initialized
    uint16_t *hist;     // [kAecmHist][kAecmBP]     far spectra by absolute block number
    int32_t *hist_q;    // [kAecmHist]              their Q-domains
    uint32_t *hist_bin; // [kAecmHist]              binary far spectra
    size_t group_bytes; // bytes between the buffers of consecutive cohorts (all eight live in one slab per cohort)
};
#ifdef __HIPCC__
// the buffers of cohort g (wave-uniform g: pointer arithmetic on scalars)
__device__ __forceinline__ AecmFarBufs far_cohort(const AecmFarBufs &F, int g) {
    const size_t o = (size_t)g * F.group_bytes;
    auto sh = [o](auto *p) { return reinterpret_cast<decltype(p)>(reinterpret_cast<char *>(p) + o); };
    return AecmFarBufs{sh(F.ring), sh(F.old), sh(F.frame), sh(F.x_prev), sh(F.mean_far), sh(F.hist), sh(F.hist_q), sh(F.hist_bin), F.group_bytes};
}
#endif

// ---------------------------------------------------------------- per-stream state block (int32 words)
enum AecmLayout : int {
    A_D_PREV = 0,                    // int16[64]  dBufNoisy[0..63]
    A_OUT_BUF = A_D_PREV + 32,       // int16[64]  outBuf
    A_NEAR_RING = A_OUT_BUF + 32,    // int16[144] nearNoisyFrameBuf
    A_OUT_RING = A_NEAR_RING + 72,   // int16[144] outFrameBuf
    A_CH_STORED = A_OUT_RING + 72,   // int16[68]
    A_CH_ADAPT16 = A_CH_STORED + 34, // int16[68]
    A_NEAR_FILT = A_CH_ADAPT16 + 34, // int16[68]
    A_NOISE_LO = A_NEAR_FILT + 34,   // int16[68]  noiseEstTooLowCtr
    A_NOISE_HI = A_NOISE_LO + 34,    // int16[68]  noiseEstTooHighCtr
    A_NEAR_LOG = A_NOISE_HI + 34,    // int16[64]  nearLogEnergy        (rings: logical [i] = physical [(head + i) & 63])
    A_EADAPT_LOG = A_NEAR_LOG + 32,  // int16[64]  echoAdaptLogEnergy
    A_ESTORED_LOG = A_EADAPT_LOG + 32,  // int16[64] echoStoredLogEnergy
    A_CH_ADAPT32 = A_ESTORED_LOG + 32,  // int32[68]
    A_ECHO_FILT = A_CH_ADAPT32 + 68,    // int32[68]
    A_NOISE_EST = A_ECHO_FILT + 68,     // int32[68]
    A_MEAN_NEAR = A_NOISE_EST + 68,     // int32[32] mean_near_spectrum of bins 12..43
    A_MEAN_BITS = A_MEAN_NEAR + 32,     // int32[104] mean_bit_counts
    A_SCAL = A_MEAN_BITS + 104,         // 34 scalar words (M_COUNT used), read in place through LdsScal
    A_WORDS = A_SCAL + 34
};
static_assert(A_WORDS % 4 == 0, "16-byte state copies");
enum AecmScalar {
    M_FIRST_VAD = 0, M_CUR_VAD, M_DFA_Q, M_TOT_COUNT, M_FAR_LOG, M_E_MIN, M_E_MAX, M_E_MAXMIN, M_E_VAD, M_E_MSE, M_VAD_UPD, M_STARTUP,
    M_MSE_COUNT, M_SUP_GAIN, M_SUP_GAIN_OLD, M_MSE_ADAPT_OLD, M_MSE_STORED_OLD, M_MSE_THR, M_NOISE_CTR, M_SEED, M_NEAR_INIT, M_MIN_PROB,
    M_LAST_PROB, M_LAST_DELAY, M_LOG_HEAD, M_COUNT
};

struct alignas(16) AecmWave {
    int32_t st[A_WORDS];
    int32_t cx[128];        // FFT work array (packed complex)
    int32_t dfw[kAecmBP];   // near spectrum (re | im << 16); the filtered spectrum overwrites it bin by bin
    int32_t echo_est[kAecmBP];
    uint16_t dfa[kAecmBP], xfa[kAecmBP];
    int16_t hnl[kAecmBP];
};

__device__ __forceinline__ int32_t add_sat32(int32_t a, int32_t b) {  // spl_inl.h:38-55
    int32_t s = wadd(a, b);
    if (a < 0) {
        if (b < 0 && s >= 0) s = (int32_t)0x80000000;
    } else if (b > 0 && s < 0) {
        s = 0x7FFFFFFF;
    }
    return s;
}
__device__ __forceinline__ int16_t log_energy_q8(uint32_t energy, int q) {  // aecm_core.c:709-721
    int16_t l = 7 << 7;
    if (energy > 0) {
        const int zeros = norm_u32(energy);
        const int16_t frac = (int16_t)(((energy << zeros) & 0x7FFFFFFF) >> 23);
        l = (int16_t)(l + ((31 - zeros) << 8) + frac - (q << 8));
    }
    return l;
}
__device__ __forceinline__ int16_t asym_filt(int16_t old, int16_t in, int step_pos, int step_neg) {  // aecm_core.c:668-690
    if (old == 32767 || old == -32768) return in;
    return old > in ? (int16_t)(old - ((old - in) >> step_neg)) : (int16_t)(old + ((in - old) >> step_pos));
}
__device__ __forceinline__ void mean_estimator(int32_t v, int factor, int32_t &mean) {  // delay_estimator.c:672-684
    int32_t d = wsub(v, mean);
    d = d < 0 ? -((-d) >> factor) : d >> factor;
    mean = wadd(mean, d);
}

// TimeToFrequencyDomain + WindowAndFFT (aecm_core_c.c:68-96, 171-278) of the 128 samples (t0, t1): leaves the spectrum in
// spec[] (re | im << 16, imag already sign-flipped like freq_signal) and |X| in mag[]; returns the scaling; *sum = sum |X|
__device__ int time_to_freq(AecmWave &W, const AecmConsts &K, int lane, int16_t t0, int16_t t1, int32_t *spec, uint16_t *mag,
                            uint32_t *sum) {
    // t0 / t1: samples lane and 64 + lane of the 128-sample block pair (registers: the pair never lives in LDS)
    int mx = t0 < 0 ? -(int)t0 : (int)t0;
    {
        const int a = t1 < 0 ? -(int)t1 : (int)t1;
        mx = a > mx ? a : mx;
    }
    mx = wave_max(mx);
    if (mx > 32767) mx = 32767;
    const int q = norm_w16((int16_t)mx);
    W.cx[bitrev<7>(lane)] = (int32_t)(uint16_t)(int16_t)(((int16_t)wshl(t0, q) * K.sqrt_hanning[lane]) >> 14);
    W.cx[bitrev<7>(64 + lane)] = (int32_t)(uint16_t)(int16_t)(((int16_t)wshl(t1, q) * K.sqrt_hanning[64 - lane]) >> 14);
    wave_sync();
    spl_cfft128<false>(W.cx, K.tw, lane);
    uint32_t part = 0;
    for (int b = lane; b < 65; b += 64) {
        const int32_t x = W.cx[b];
        const int16_t re = lo16(x);
        int16_t im = (int16_t)-hi16(x);
        uint16_t m;
        if (b == 0 || b == 64) {
            im = 0;
            m = (uint16_t)(re >= 0 ? re : -re);
        } else if (re == 0) {
            m = (uint16_t)(im >= 0 ? im : -im);
        } else if (im == 0) {
            m = (uint16_t)(re >= 0 ? re : -re);
        } else {
            const int16_t ar = (int16_t)(re >= 0 ? re : -re), ai = (int16_t)(im >= 0 ? im : -im);
            m = (uint16_t)sqrt_floor(add_sat32(ar * ar, ai * ai));
        }
        spec[b] = pack16(re, im);
        mag[b] = m;
        part += m;
    }
    *sum = wave_sum(part);
    wave_sync();
    return q;
}

// BinarySpectrumFix (delay_estimator_wrapper.c:52-78): lanes 12..43 own one band each; mean[] holds bands 12..43
__device__ uint32_t binary_spectrum(const uint16_t *mag, int32_t *mean, int q, int &initialized, int lane) {
    const bool mine = lane >= 12 && lane <= 43;
    const int32_t s = mine ? wshl((int32_t)mag[lane], 15 - q) : 0;
    int32_t m = mine ? mean[lane - 12] : 0;
    if (!initialized) {
        if (mine && mag[lane] > 0) m = s >> 1;
        initialized = wave_any(mine && mag[lane] > 0);
    }
    if (mine) {
        mean_estimator(s, 6, m);
        mean[lane - 12] = m;
    }
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(mine && s > m);
    return (uint32_t)(bal >> 12);
}

// ---------------------------------------------------------------- far-end kernel: one wave per batch
// plan_by_value: a one-packet launch hands its plan over as a kernel argument; this kernel, which runs in front of the near
// kernel in the same stream, stores it into plans[0] for both (no host-to-device copy of the plan in front of the launch).
// grid = number of cohorts: workgroup g (one wave) serves cohort g with plans[packet * n_cohorts + g] and the far-end
// packets at far + g * far_group_stride (0: every cohort hears the same far-end, blocked from its own start)
// What one lane of this wave has stored to global memory, the other lanes of the SAME wave read next (a workgroup here is one wave):
// a workgroup-scope fence -- the stores have left the wave (s_waitcnt vmcnt(0)); loads of the same compute unit see them, its L1 is
// write-through -- instead of the agent-scope __threadfence() that stood here: with thousands of far-end waves (one per cohort) in
// flight, five cache write-backs per packet and cohort made this kernel 0.45 ms for 4 096 far-ends, 14 x the single one.
__device__ __forceinline__ void wave_handoff_global() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

__global__ __launch_bounds__(64) void aecm_far_kernel(AecmFarBufs F_all, const AecmConsts *__restrict__ consts, AecmPlan *plans, int n_plans,
                                                     int n_classes, const int32_t *__restrict__ plan_of, const int16_t *far, long far_stride,
                                                     long far_group_stride, int chn, int plan_by_value, const AecmPlan plan_value,
                                                     const AecmPlan *__restrict__ host_plans) {
    __shared__ AecmConsts K;
    __shared__ AecmWave W;
    const int lane = threadIdx.x;
    const AecmFarBufs F = far_cohort(F_all, (int)blockIdx.x);
    if (far) far += (size_t)blockIdx.x * far_group_stride;
    // [packet][class]: cohorts whose control planes run in lockstep share one plan (control-plane classes, aec_ctl.h; wmx_aecm_run_cohorts)
    const int cls = plan_of ? plan_of[blockIdx.x] : (int)blockIdx.x;
    if (plan_by_value) {
        const int *src = reinterpret_cast<const int *>(&plan_value);
        int *dst = reinterpret_cast<int *>(plans);
        for (int i = lane; i < (int)(sizeof(AecmPlan) / 4); i += 64) dst[i] = src[i];
        wave_handoff_global();  // (the near kernel sees the stored plans through the kernel boundary)
        wave_sync();
    } else if (host_plans) {
        // several cohorts or packets: every cohort's wave fetches ITS plans from the pinned host slot (device-visible host memory,
        // one PCIe round trip for the whole grid) and stores them for the near kernel -- no copy-engine operation between the previous
        // kernel of the stream and this one (13 us per step with two cohorts, 45 with 256: the launch queue drains around a blit)
        constexpr int NW = (int)(sizeof(AecmPlan) / 4);
        for (int p = 0; p < n_plans; p++) {
            const int *src = reinterpret_cast<const int *>(host_plans + (size_t)p * n_classes + cls);  // (every cohort its own class here)
            int *dst = reinterpret_cast<int *>(plans + (size_t)p * n_classes + cls);
            for (int i = lane; i < NW; i += 64) dst[i] = __builtin_nontemporal_load(src + i);
        }
        wave_handoff_global();  // (the near kernel sees the stored plans through the kernel boundary)
        wave_sync();
    }
    plans += cls;  // [packet][class]: a launch uploads exactly packets x classes plans
    {
        const int4 *src = reinterpret_cast<const int4 *>(consts);
        int4 *dst = reinterpret_cast<int4 *>(&K);
        for (int i = lane; i < (int)(sizeof(AecmConsts) / 16); i += 64) dst[i] = src[i];
    }
    __syncthreads();
    int far_init = uni(F.mean_far[32]);
    int32_t *mean = &W.st[0];  // 32 thresholds staged in LDS for the launch
    if (lane < 32) mean[lane] = F.mean_far[lane];
    wave_sync();
    for (int p = 0; p < n_plans; p++) {
        const AecmPlan &pl = plans[(size_t)p * n_classes];
        if (pl.has_far) {
            const int16_t *src = far + (long)p * far_stride;
            for (int i = lane; i < pl.far_n; i += 64) {
                int pos = pl.far_w + i;
                pos -= pos >= kAecmFarRing ? kAecmFarRing : 0;
                F.ring[pos] = src[(long)i * chn];  // left channel only, src/webrtc.c:303-309
            }
            wave_handoff_global();
            wave_sync();
        }
        if (!pl.has_near || pl.passthrough) continue;
        for (int f = 0; f < pl.n_frames; f++) {
            const AecmFramePlan &fp = pl.fr[f];
            for (int i = lane; i < kAecmFrame; i += 64) {
                int16_t v;
                if (fp.far_src >= 0) {
                    int pos = fp.far_src + i;
                    pos -= pos >= kAecmFarRing ? kAecmFarRing : 0;
                    v = F.ring[pos];
                    F.old[fp.old_slot * kAecmFrame + i] = v;
                } else {
                    v = F.old[fp.old_slot * kAecmFrame + i];
                }
                int w = fp.ring_w + i;
                w -= w >= kAecmFrameRing ? kAecmFrameRing : 0;
                F.frame[w] = v;
            }
            wave_handoff_global();
            wave_sync();
            for (int b = 0; b < fp.n_blocks; b++) {
                int r = fp.blk_r[b] + lane;
                r -= r >= kAecmFrameRing ? kAecmFrameRing : 0;
                const int16_t nw = F.frame[r];
                uint32_t sum;
                const int far_q = time_to_freq(W, K, lane, F.x_prev[lane], nw, W.dfw, W.xfa, &sum);
                const int slot = fp.blk_t[b] & (kAecmHist - 1);
                for (int k = lane; k < 65; k += 64) F.hist[slot * kAecmBP + k] = W.xfa[k];
                const uint32_t bin = binary_spectrum(W.xfa, mean, far_q, far_init, lane);
                if (lane == 0) {
                    F.hist_q[slot] = far_q;
                    F.hist_bin[slot] = bin;
                }
                F.x_prev[lane] = nw;
                wave_handoff_global();
                wave_sync();
            }
        }
    }
    if (lane < 32) F.mean_far[lane] = mean[lane];
    if (lane == 0) F.mean_far[32] = far_init;
}

// ---------------------------------------------------------------- one 64-sample block of one stream (ProcessBlock)
// t_prev / t_new: sample `lane` of the previous and of the new near-end block on entry; on return t_prev is the new block's
// sample (dBufNoisy slides) and t_new the block's output sample.
#ifdef WMX_AECM_PROF  // developer build only (make EXTRA=-DWMX_AECM_PROF): cycles per phase of aecm_block, summed over waves
__device__ unsigned long long g_aecm_prof[1024 * 16];
#define AECM_PROF(i)                                          \
    do {                                                      \
        const long long t_now = clock64();                    \
        prof_acc[i] += (unsigned long long)(t_now - prof_t0); \
        prof_t0 = clock64();                                  \
    } while (0)
extern "C" int wmx_debug_aecm_prof(unsigned long long *out16, int reset) {
    (void)hipDeviceSynchronize();
    static unsigned long long all[1024 * 16];
    (void)hipMemcpyFromSymbol(all, HIP_SYMBOL(g_aecm_prof), sizeof(all));
    for (int i = 0; i < 16; i++) {
        out16[i] = 0;
        for (int g = 0; g < 1024; g++) out16[i] += all[g * 16 + i];
    }
    if (reset) {
        static const unsigned long long z[1024 * 16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_aecm_prof), z, sizeof(z));
    }
    return 0;
}
#else
#define AECM_PROF(i)
#endif

__device__ void aecm_block(AecmWave &W, const AecmConsts &K, const AecmFarBufs &F, const LdsScal sc, int t, int mult, int lane,
                           int16_t &t_prev, int16_t &t_new) {
#ifdef WMX_AECM_PROF
    unsigned long long prof_acc[16] = {0};
    long long prof_t0 = clock64();
#endif
    int16_t *ch_stored = reinterpret_cast<int16_t *>(&W.st[A_CH_STORED]), *ch_adapt16 = reinterpret_cast<int16_t *>(&W.st[A_CH_ADAPT16]);
    int16_t *near_filt = reinterpret_cast<int16_t *>(&W.st[A_NEAR_FILT]);
    int16_t *noise_lo = reinterpret_cast<int16_t *>(&W.st[A_NOISE_LO]), *noise_hi = reinterpret_cast<int16_t *>(&W.st[A_NOISE_HI]);
    int16_t *near_log = reinterpret_cast<int16_t *>(&W.st[A_NEAR_LOG]), *eadapt_log = reinterpret_cast<int16_t *>(&W.st[A_EADAPT_LOG]);
    int16_t *estored_log = reinterpret_cast<int16_t *>(&W.st[A_ESTORED_LOG]);
    int32_t *ch_adapt32 = &W.st[A_CH_ADAPT32], *echo_filt = &W.st[A_ECHO_FILT], *noise_est = &W.st[A_NOISE_EST];
    int32_t *mean_near = &W.st[A_MEAN_NEAR], *mean_bits = &W.st[A_MEAN_BITS];
    int16_t *out_buf = reinterpret_cast<int16_t *>(&W.st[A_OUT_BUF]);

    // the far-end's binary spectra of the last kAecmMaxDelay blocks (global memory, L2): requested here, used by the delay
    // estimator behind the near spectrum's transform instead of being waited for there
    uint32_t far_bin[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int i = lane + 64 * r;
        far_bin[r] = F.hist_bin[(t - (i < kAecmMaxDelay ? i : 0)) & (kAecmHist - 1)];  // binary_far_history[i] = the spectrum of i blocks ago
    }
    if (sc[M_STARTUP] < 2) sc[M_STARTUP] = ((uint32_t)sc[M_TOT_COUNT] >= 512) + ((uint32_t)sc[M_TOT_COUNT] >= 1024);
    // near spectrum of [previous block | new block]
    uint32_t dfa_sum;
    const int zeros_d = time_to_freq(W, K, lane, t_prev, t_new, W.dfw, W.dfa, &dfa_sum);
    const int dfa_q_old = sc[M_DFA_Q];
    sc[M_DFA_Q] = zeros_d;

    AECM_PROF(0);
    // delay estimate: BinarySpectrumFix + WebRtc_ProcessBinarySpectrum (robust validation off)
    int delay;
    {
        int near_init = sc[M_NEAR_INIT];
        const uint32_t near_bin = binary_spectrum(W.dfa, mean_near, zeros_d, near_init, lane);
        sc[M_NEAR_INIT] = near_init;
        uint32_t kmin = 0xFFFFFFFFu;
        int32_t vmax = 0;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int i = lane + 64 * r;
            if (i < kAecmMaxDelay) {
                const uint32_t fb = far_bin[r];
                const int fbc = __popc(fb);
                int32_t m = mean_bits[i];
                if (fbc > 0) mean_estimator(__popc(near_bin ^ fb) << 9, 13 - ((3 * fbc) >> 4), m);
                mean_bits[i] = m;
                // first minimum wins (strict < in index order): order by (value, index); values are < 2^15
                const uint32_t key = ((uint32_t)m << 8) | (uint32_t)i;
                kmin = key < kmin ? key : kmin;
                vmax = m > vmax ? m : vmax;
            }
        }
        kmin = wave_umin(kmin);
        vmax = wave_max(vmax);
        int32_t best = (int32_t)(kmin >> 8);
        int candidate = (int)(kmin & 0xff);
        if (best >= (32 << 9)) best = 32 << 9, candidate = -1;  // nothing below kMaxBitCountsQ9
        const int32_t depth = vmax - best;
        if (sc[M_MIN_PROB] > 8704 && depth > 2816) {
            int32_t thr = best + 1024;
            if (thr < 8704) thr = 8704;
            if (sc[M_MIN_PROB] > thr) sc[M_MIN_PROB] = thr;
        }
        sc[M_LAST_PROB]++;
        if (depth > 1024 && (best < sc[M_MIN_PROB] || best < sc[M_LAST_PROB])) {
            sc[M_LAST_DELAY] = candidate;
            if (best < sc[M_LAST_PROB]) sc[M_LAST_PROB] = best;
        }
        delay = sc[M_LAST_DELAY];
        if (delay < 0) delay = 0;  // -2: no estimate yet (aecm_core_c.c:397-400); -1 cannot reach here
    }
    AECM_PROF(1);
    // aligned far spectrum (WebRtcAecm_AlignedFarend): the block `delay` blocks ago
    const int slot = (t - delay) & (kAecmHist - 1);
    const int zeros_x = F.hist_q[slot];
    for (int b = lane; b < 65; b += 64) W.xfa[b] = F.hist[slot * kAecmBP + b];
    wave_sync();

    AECM_PROF(2);
    // ---- CalcEnergies, aecm_core.c:730-851
    const int head = (sc[M_LOG_HEAD] - 1) & 63;  // the three log histories shift by one: move the ring head instead
    sc[M_LOG_HEAD] = head;
    {
        uint32_t e_far = 0, e_adapt = 0, e_stored = 0;
        for (int b = lane; b < 65; b += 64) {
            const uint16_t x = W.xfa[b];
            const int32_t est = (int32_t)ch_stored[b] * x;
            W.echo_est[b] = est;
            e_far += x;
            e_adapt += (uint32_t)(ch_adapt16[b] * x);
            e_stored += (uint32_t)est;
        }
        e_far = wave_sum(e_far);
        e_adapt = wave_sum(e_adapt);
        e_stored = wave_sum(e_stored);
        const int16_t near0 = log_energy_q8(dfa_sum, zeros_d);
        int16_t eadapt0 = log_energy_q8(e_adapt, 12 + zeros_x);
        const int16_t estored0 = log_energy_q8(e_stored, 12 + zeros_x);
        const int16_t far_log = log_energy_q8(e_far, zeros_x);
        sc[M_FAR_LOG] = far_log;
        int16_t e_min = (int16_t)sc[M_E_MIN], e_max = (int16_t)sc[M_E_MAX], e_vad = (int16_t)sc[M_E_VAD];
        if (far_log > 1025) {
            int inc_max = 4, dec_max = 11, inc_min = 11, dec_min = 3;
            if (sc[M_STARTUP] == 0) inc_max = 2, dec_min = 2, inc_min = 8;
            e_min = asym_filt(e_min, far_log, inc_min, dec_min);
            e_max = asym_filt(e_max, far_log, inc_max, dec_max);
            sc[M_E_MAXMIN] = (int16_t)(e_max - e_min);
            int16_t tv = (int16_t)(2560 - e_min);
            tv = tv > 0 ? (int16_t)((tv * 230) >> 9) : (int16_t)0;
            tv = (int16_t)(tv + 230);
            if ((int)(sc[M_STARTUP] == 0) | (int)(sc[M_VAD_UPD] > 1024)) {
                e_vad = (int16_t)(e_min + tv);
            } else if (e_vad > far_log) {
                e_vad = (int16_t)(e_vad + ((far_log + tv - e_vad) >> 6));
                sc[M_VAD_UPD] = 0;
            } else {
                sc[M_VAD_UPD] = (int16_t)(sc[M_VAD_UPD] + 1);
            }
            sc[M_E_MSE] = (int16_t)(e_vad + (1 << 8));
            sc[M_E_MIN] = e_min;
            sc[M_E_MAX] = e_max;
            sc[M_E_VAD] = e_vad;
        }
        if (far_log > e_vad) {
            if ((int)(sc[M_STARTUP] == 0) | (int)(sc[M_E_MAXMIN] > 929)) sc[M_CUR_VAD] = 1;
        } else {
            sc[M_CUR_VAD] = 0;
        }
        if (sc[M_CUR_VAD] && sc[M_FIRST_VAD]) {
            sc[M_FIRST_VAD] = 0;
            if (eadapt0 > near0) {
                for (int b = lane; b < 65; b += 64) ch_adapt16[b] >>= 3;
                eadapt0 = (int16_t)(eadapt0 - (3 << 8));
                sc[M_FIRST_VAD] = 1;
            }
        }
        if (lane == 0) {
            near_log[head] = near0;
            eadapt_log[head] = eadapt0;
            estored_log[head] = estored0;
        }
    }
    wave_sync();

    AECM_PROF(3);
    // ---- CalcStepSize, aecm_core.c:858-891
    int16_t mu = 1;
    if (!sc[M_CUR_VAD]) {
        mu = 0;
    } else if (sc[M_STARTUP] > 0) {
        if (sc[M_E_MIN] >= sc[M_E_MAX]) {
            mu = 10;
        } else {
            const int16_t t16 = (int16_t)(sc[M_FAR_LOG] - sc[M_E_MIN]);
            mu = (int16_t)(10 - 1 - (int16_t)div_w32_w16(t16 * 9, (int16_t)sc[M_E_MAXMIN]));
        }
        if (mu < 1) mu = 1;
    }
    sc[M_TOT_COUNT] = (int32_t)((uint32_t)sc[M_TOT_COUNT] + 1);

    AECM_PROF(4);
    // ---- UpdateChannel, aecm_core.c:902-1109
    if (mu) {
        for (int b = lane; b < 65; b += 64) {
            const uint16_t x = W.xfa[b], d = W.dfa[b];
            const int32_t ca = ch_adapt32[b];
            const int16_t zeros_ch = (int16_t)norm_u32((uint32_t)ca), zeros_far = (int16_t)norm_u32((uint32_t)x);
            uint32_t u1;
            int16_t shift_ch_far;
            if (zeros_ch + zeros_far > 31) {
                u1 = (uint32_t)ca * x;
                shift_ch_far = 0;
            } else {
                shift_ch_far = (int16_t)(32 - zeros_ch - zeros_far);
                u1 = (uint32_t)wmul(ca >> shift_ch_far, x);
            }
            int16_t zeros_num = (int16_t)norm_u32(u1);
            const int16_t zeros_dfa = (int16_t)(d ? norm_u32((uint32_t)d) : 32);
            const int16_t t16 = (int16_t)(zeros_dfa - 2 + zeros_d - 28 - zeros_x + shift_ch_far);
            int16_t xfa_q, dfa_q;
            if (zeros_num > t16 + 1) {
                xfa_q = t16;
                dfa_q = (int16_t)(zeros_dfa - 2);
            } else {
                xfa_q = (int16_t)(zeros_num - 2);
                dfa_q = (int16_t)(28 + zeros_x - zeros_d - shift_ch_far + xfa_q);
            }
            u1 = xfa_q >= 0 ? u1 << xfa_q : u1 >> -xfa_q;
            const uint32_t u2 = dfa_q >= 0 ? (uint32_t)d << dfa_q : (uint32_t)d >> -dfa_q;
            const int32_t err = (int32_t)u2 - (int32_t)u1;
            zeros_num = (int16_t)norm_w32(err);
            if (err && x > (16 << zeros_x)) {
                int32_t upd;
                int16_t shift_num;
                if (zeros_num + zeros_far > 31) {
                    upd = err > 0 ? (int32_t)((uint32_t)err * x) : -(int32_t)((uint32_t)(-err) * x);
                    shift_num = 0;
                } else {
                    shift_num = (int16_t)(32 - (zeros_num + zeros_far));
                    upd = err > 0 ? wmul(err >> shift_num, x) : -wmul(-err >> shift_num, x);
                }
                upd = div_w32_w16(upd, (int16_t)(b + 1));
                const int16_t shift2 = (int16_t)(shift_num + shift_ch_far - xfa_q - mu - ((30 - zeros_far) << 1));
                upd = norm_w32(upd) < shift2 ? 0x7FFFFFFF : shift_w32(upd, shift2);
                int32_t nc = add_sat32(ca, upd);
                if (nc < 0) nc = 0;
                ch_adapt32[b] = nc;
                ch_adapt16[b] = (int16_t)(nc >> 16);
            }
        }
        wave_sync();
    }
    int store = 0, reset = 0;
    if ((int)(sc[M_STARTUP] == 0) & (int)(sc[M_CUR_VAD] != 0)) {
        store = 1;
    } else {
        if (sc[M_FAR_LOG] < sc[M_E_MSE])
            sc[M_MSE_COUNT] = 0;
        else
            sc[M_MSE_COUNT] = (int16_t)(sc[M_MSE_COUNT] + 1);
        if (sc[M_MSE_COUNT] >= 30) {
            uint32_t ms = 0, ma = 0;
            if (lane < 20) {
                const int k = (head + lane) & 63;
                int32_t d = (int32_t)estored_log[k] - (int32_t)near_log[k];
                ms = (uint32_t)(d >= 0 ? d : -d);
                d = (int32_t)eadapt_log[k] - (int32_t)near_log[k];
                ma = (uint32_t)(d >= 0 ? d : -d);
            }
            const int32_t mse_stored = (int32_t)wave_sum(ms), mse_adapt = (int32_t)wave_sum(ma);
            if (((mse_stored << 5) < (29 * mse_adapt)) & (wshl(sc[M_MSE_STORED_OLD], 5) < wmul(29, sc[M_MSE_ADAPT_OLD]))) {
                reset = 1;
            } else if ((int)((29 * mse_stored) > (mse_adapt << 5)) & (int)(mse_adapt < sc[M_MSE_THR]) & (int)(sc[M_MSE_ADAPT_OLD] < sc[M_MSE_THR])) {
                store = 1;
                if (sc[M_MSE_THR] == 0x7FFFFFFF) {
                    sc[M_MSE_THR] = wadd(mse_adapt, sc[M_MSE_ADAPT_OLD]);
                } else {
                    const int scaled = wmul(sc[M_MSE_THR], 5) / 8;
                    sc[M_MSE_THR] = wadd(sc[M_MSE_THR], wmul(mse_adapt - scaled, 205) >> 8);
                }
            }
            sc[M_MSE_COUNT] = 0;
            sc[M_MSE_STORED_OLD] = mse_stored;
            sc[M_MSE_ADAPT_OLD] = mse_adapt;
        }
    }
    if (store) {  // StoreAdaptiveChannelC, aecm_core.c:334-362
        for (int b = lane; b < 65; b += 64) {
            const int16_t c = ch_adapt16[b];
            ch_stored[b] = c;
            W.echo_est[b] = (int32_t)c * W.xfa[b];
        }
    } else if (reset) {  // ResetAdaptiveChannelC, :364-380
        for (int b = lane; b < 65; b += 64) {
            const int16_t c = ch_stored[b];
            ch_adapt16[b] = c;
            ch_adapt32[b] = wshl(c, 16);
        }
    }

    AECM_PROF(5);
    // ---- CalcSuppressionGain, aecm_core.c:1118-1185
    int16_t sup_gain;
    {
        int16_t sup = 256;
        if (!sc[M_CUR_VAD]) {
            sup = 0;
        } else {
            // nearLogEnergy[0] / echoStoredLogEnergy[0] of this block were written by lane 0 above; all lanes read them
            const int16_t tdiff = (int16_t)(near_log[head] - estored_log[head]);
            const int16_t dE = (int16_t)(tdiff >= 0 ? tdiff : -tdiff);
            if (dE < 400) {
                if (dE < 200)
                    sup = (int16_t)(3072 - (int16_t)div_w32_w16(1536 * dE + 100, 200));
                else
                    sup = (int16_t)(256 + (int16_t)div_w32_w16(1280 * (400 - dE) + 100, 200));
            } else {
                sup = 256;
            }
        }
        const int16_t tg = sup > (int16_t)sc[M_SUP_GAIN_OLD] ? sup : (int16_t)sc[M_SUP_GAIN_OLD];
        sc[M_SUP_GAIN_OLD] = sup;
        sc[M_SUP_GAIN] = (int16_t)(sc[M_SUP_GAIN] + (int16_t)((tg - sc[M_SUP_GAIN]) >> 4));
        sup_gain = (int16_t)sc[M_SUP_GAIN];
    }
    wave_sync();

    AECM_PROF(6);
    // ---- Wiener filter coefficients, aecm_core_c.c:434-545
    int pos_count = 0;
    for (int b = lane; b < 65; b += 64) {
        int32_t ef = echo_filt[b];
        ef = wadd(ef, wmul(wsub(W.echo_est[b], ef), 50) >> 8);
        echo_filt[b] = ef;
        const int16_t zeros32 = (int16_t)(norm_w32(ef) + 1);
        int16_t zeros16 = (int16_t)(norm_w16(sup_gain) + 1);
        uint32_t gained;
        int16_t res_diff;
        if (zeros32 + zeros16 > 16) {
            gained = (uint32_t)ef * (uint16_t)sup_gain;
            res_diff = (int16_t)(14 - 12 - 8 + (zeros_d - zeros_x));
        } else {
            const int16_t tt = (int16_t)(17 - zeros32 - zeros16);
            res_diff = (int16_t)(14 + tt - 12 - 8 + (zeros_d - zeros_x));
            gained = zeros32 > tt ? (uint32_t)ef * (uint16_t)(sup_gain >> tt) : (uint32_t)wmul(ef >> tt, sup_gain);
        }
        int16_t nf = near_filt[b];
        const uint16_t d = W.dfa[b];
        zeros16 = (int16_t)norm_w16(nf);
        const int16_t dq = (int16_t)(zeros_d - dfa_q_old);
        int16_t t1, t2, q_diff;
        if (zeros16 < dq && nf) {
            t1 = (int16_t)wshl(nf, zeros16);
            q_diff = (int16_t)(zeros16 - dq);
            t2 = (int16_t)(d >> -q_diff);
        } else {
            t1 = (int16_t)(dq < 0 ? nf >> -dq : wshl(nf, dq));
            q_diff = 0;
            t2 = (int16_t)d;
        }
        const int32_t nd = (int32_t)(t2 - t1);
        t2 = (int16_t)(nd >> 4);
        t2 = (int16_t)(t2 + t1);
        zeros16 = (int16_t)norm_w16(t2);
        if ((t2) & (-q_diff > zeros16))
            nf = 32767;
        else
            nf = (int16_t)(q_diff < 0 ? wshl(t2, -q_diff) : t2 >> q_diff);
        near_filt[b] = nf;
        int16_t h;
        if (gained == 0) {
            h = 16384;
        } else if (nf == 0) {
            h = 0;
        } else {
            gained += (uint32_t)(nf >> 1);
            const uint32_t qv = gained / (uint16_t)nf;
            const int32_t r = (int32_t)(res_diff >= 0 ? qv << res_diff : qv >> -res_diff);
            if (r > 16384) {
                h = 0;
            } else if (r < 0) {
                h = 16384;
            } else {
                h = (int16_t)(16384 - (int16_t)r);
                if (h < 0) h = 0;
            }
        }
        if (h) pos_count++;
        if (mult == 2) h = (int16_t)((h * h) >> 14);  // wideband: squared (aecm_core_c.c:549-553)
        W.hnl[b] = h;
    }
    const int num_pos = (int)wave_sum((uint32_t)pos_count);
    wave_sync();
    int16_t avg_hnl = 0;
    if (mult == 2) {  // cap the bands above 24 by the mean of bands 4..24, :555-571
        uint32_t a = (lane >= 4 && lane <= 24) ? (uint32_t)(int32_t)W.hnl[lane] : 0u;
        avg_hnl = (int16_t)((int32_t)wave_sum(a) / 21);
    }
    const int16_t nlp_gain = (int16_t)(num_pos < 3 ? 0 : 16384);
    for (int b = lane; b < 65; b += 64) {
        int16_t h = W.hnl[b];
        if (mult == 2 && b >= 24 && h > avg_hnl) h = avg_hnl;
        // nlpFlag is always 1 (aecm_core.c:436; WebRtcAecm_Control is never called by wmix)
        if (h > 16384)
            h = 16384;
        else if (h < 3277)
            h = 0;
        if (!(h == 16384 && nlp_gain == 16384)) h = (int16_t)((h * nlp_gain) >> 14);
        W.hnl[b] = h;
        const int32_t x = W.dfw[b];
        W.dfw[b] = pack16((int16_t)((lo16(x) * h + 8192) >> 14), (int16_t)((hi16(x) * h + 8192) >> 14));
    }

    AECM_PROF(7);
    // ---- ComfortNoise, aecm_core_c.c:641-771 (cngMode is AecmTrue, echo_control_mobile.c:223)
    {
        const int16_t shift = (int16_t)(15 - zeros_d);
        int min_track;
        if (sc[M_NOISE_CTR] < 100) {
            sc[M_NOISE_CTR]++;
            min_track = 6;
        } else {
            min_track = 9;
        }
        // the 64 draws of WebRtcSpl_RandUArray by jump-ahead: draw k = A[k] * seed + C[k] (mod 2^31)
        const uint32_t seed = (uint32_t)sc[M_SEED];
        const uint32_t my_draw = (K.lcg_a[lane] * seed + K.lcg_c[lane]) & 0x7FFFFFFFu;
        sc[M_SEED] = (int32_t)((K.lcg_a[63] * seed + K.lcg_c[63]) & 0x7FFFFFFFu);
        // bin b uses draw b - 1: lane b takes the draw of the lane below, and lane 0 -- which owns bin 64 in the second
        // pass -- that of lane 63 (the exchange sits outside the per-bin loop: every lane must take part in it)
        const uint32_t prev_draw = (uint32_t)dpp_rows<0x13C>((int)my_draw);  // wave_ror:1 -- lane b takes lane b - 1, lane 0 lane 63
        for (int b = lane; b < 65; b += 64) {
            const int32_t v = wshl((int32_t)W.dfa[b], shift);
            int32_t ne = noise_est[b];
            int16_t lo = noise_lo[b], hi = noise_hi[b];
            if (v < ne) {
                lo = 0;
                if (ne < (1 << min_track)) {
                    hi++;
                    if (hi >= 5) {
                        ne--;
                        hi = 0;
                    }
                } else {
                    ne -= (ne - v) >> min_track;
                }
            } else {
                hi = 0;
                if ((ne >> 19) > 0) {
                    ne >>= 11;
                    ne = wmul(ne, 2049);
                } else if ((ne >> 11) > 0) {
                    ne = wmul(ne, 2049);
                    ne >>= 11;
                } else {
                    lo++;
                    if (lo >= 5) {
                        ne += (ne >> 9) + 1;
                        lo = 0;
                    }
                }
            }
            int32_t nv = ne >> shift;
            if (nv > 32767) {
                nv = 32767;
                ne = wshl(nv, shift);
            }
            noise_est[b] = ne;
            noise_lo[b] = lo;
            noise_hi[b] = hi;
            const int16_t noise = (int16_t)(((int16_t)(16384 - W.hnl[b]) * (int16_t)nv) >> 14);
            int16_t ur = 0, ui = 0;
            if (b > 0) {
                const int16_t idx = (int16_t)((359 * (int16_t)(prev_draw >> 16)) >> 15);
                ur = (int16_t)((noise * K.cos360[idx]) >> 13);
                ui = (int16_t)((-noise * K.sin360[idx]) >> 13);
            }
            if (b == 64) ui = 0;
            const int32_t e = W.dfw[b];
            W.dfw[b] = pack16(sat_w16((int32_t)lo16(e) + ur), sat_w16((int32_t)hi16(e) + ui));
        }
    }
    wave_sync();

    AECM_PROF(8);
    // ---- InverseFFTAndWindow, aecm_core_c.c:98-169
    for (int b = lane; b < 65; b += 64) {
        const int32_t e = W.dfw[b];
        const int16_t re = lo16(e), nim = (int16_t)-hi16(e);
        W.cx[bitrev<7>(b)] = pack16(re, nim);
        if (b > 0 && b < 64) W.cx[bitrev<7>(128 - b)] = pack16(re, (int16_t)-nim);
    }
    wave_sync();
    const int out_scale = spl_cfft128<true>(W.cx, K.tw, lane);
    {
        const int i = lane;
        const int16_t a = (int16_t)(((int32_t)lo16(W.cx[i]) * K.sqrt_hanning[i] + 8192) >> 14);
        int32_t v = shift_w32((int32_t)a, out_scale - zeros_d);
        const int16_t o = sat_w16(v + out_buf[i]);
        v = (lo16(W.cx[64 + i]) * K.sqrt_hanning[64 - i]) >> 14;
        v = shift_w32(v, out_scale - zeros_d);
        out_buf[i] = sat_w16(v);
        t_prev = t_new;  // dBufNoisy slides: the new block becomes the previous one
        t_new = o;       // the block's output, picked up by the caller
    }
    wave_sync();
    AECM_PROF(9);
#ifdef WMX_AECM_PROF
    if (lane < 16) {  // racy read-modify-write among the waves that share a slot: good enough for a profile
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) v = lane == i ? prof_acc[i] : v;
        g_aecm_prof[(blockIdx.x & 1023) * 16 + lane] += v;
    }
#endif
}

// One wave per stream, kAecmWavesPerBlock streams per workgroup.
#ifndef WMX_AECM_WPE
#define WMX_AECM_WPE 7  // 0.63 / 0.53 / 0.48 / 0.445 ms at 3 / 4 / 5 / 6 waves per SIMD when first measured; 22.6 KB of LDS per workgroup allow 7
#endif
__global__ __launch_bounds__(64 * kAecmWavesPerBlock) __attribute__((amdgpu_waves_per_eu(WMX_AECM_WPE, WMX_AECM_WPE))) void aecm_near_kernel(int32_t *__restrict__ state, AecmFarBufs F_all,
                                                                            const AecmConsts *__restrict__ consts,
                                                                            const AecmPlan *__restrict__ plans, int n_plans, int n_classes, const int32_t *__restrict__ plan_of, const int16_t *near,
                                                                            int16_t *out, int n_streams, long stream_stride, long packet_stride,
                                                                            int chn, int pkg, int mult, const int *__restrict__ stream_cohort,
                                                                            const uint8_t *__restrict__ active) {
    __shared__ AecmConsts K;
    __shared__ AecmWave WS[kAecmWavesPerBlock];
    {
        const int4 *src = reinterpret_cast<const int4 *>(consts);
        int4 *dst = reinterpret_cast<int4 *>(&K);
        for (int i = threadIdx.x; i < (int)(sizeof(AecmConsts) / 16); i += blockDim.x) dst[i] = src[i];
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long s = (long)blockIdx.x * kAecmWavesPerBlock + wave;
    AecmWave &W = WS[wave];
    const bool live = stream_active(active, (int)s, n_streams);  // no stream, or one that is switched off: nothing touched
    int32_t *st = state + (live ? s : 0) * (long)A_WORDS;
    if (live) {
        const int4 *src = reinterpret_cast<const int4 *>(st);
        int4 *dst = reinterpret_cast<int4 *>(W.st);
        for (int i = lane; i < A_WORDS / 4; i += 64) dst[i] = src[i];
    }
    __syncthreads();
    if (!live) return;
    // the cohort this stream belongs to (wave-uniform): its far-end history and its plans
    const int grp = stream_cohort ? __builtin_amdgcn_readfirstlane(stream_cohort[s]) : 0;
    const AecmFarBufs F = far_cohort(F_all, grp);
    plans += plan_of ? __builtin_amdgcn_readfirstlane(plan_of[grp]) : grp;  // [packet][class]
    const LdsScal sc{&W.st[A_SCAL]};
    int16_t *near_ring = reinterpret_cast<int16_t *>(&W.st[A_NEAR_RING]), *out_ring = reinterpret_cast<int16_t *>(&W.st[A_OUT_RING]);
    int16_t *d_prev = reinterpret_cast<int16_t *>(&W.st[A_D_PREV]);
    for (int p = 0; p < n_plans; p++) {
        const AecmPlan &pl = plans[(size_t)p * n_classes];
        if (!pl.has_near) continue;
        const int16_t *ip = near + s * stream_stride + (long)p * packet_stride;
        int16_t *op = out + s * stream_stride + (long)p * packet_stride;
        if (pl.passthrough) {  // start-up: out = near (echo_control_mobile.c:341-352), left channel to every channel
            if (pl.discard_out) continue;
            for (int i = lane; i < pkg; i += 64) {
                const int16_t v = ip[(long)i * chn];
                for (int c = 0; c < chn; c++) op[(long)i * chn + c] = v;
            }
            continue;
        }
        // the whole packet is read before anything is written (in-place calls)
        int16_t in0[2], in1[2];
#pragma unroll
        for (int f = 0; f < 2; f++) {
            in0[f] = (f < pl.n_frames) ? ip[(long)(f * kAecmFrame + lane) * chn] : (int16_t)0;
            in1[f] = (f < pl.n_frames && lane < kAecmFrame - 64) ? ip[(long)(f * kAecmFrame + 64 + lane) * chn] : (int16_t)0;
        }
#pragma unroll
        for (int f = 0; f < 2; f++) {
            if (f >= pl.n_frames) break;
            const AecmFramePlan &fp = pl.fr[f];
            {
                int w = fp.ring_w + lane;
                w -= w >= kAecmFrameRing ? kAecmFrameRing : 0;
                near_ring[w] = in0[f];
                if (lane < kAecmFrame - 64) {
                    int w2 = fp.ring_w + 64 + lane;
                    w2 -= w2 >= kAecmFrameRing ? kAecmFrameRing : 0;
                    near_ring[w2] = in1[f];
                }
            }
            wave_sync();
            for (int b = 0; b < fp.n_blocks; b++) {
                int r = fp.blk_r[b] + lane;
                r -= r >= kAecmFrameRing ? kAecmFrameRing : 0;
                int16_t t_prev = d_prev[lane], t_new = near_ring[r];
                aecm_block(W, K, F, sc, fp.blk_t[b], mult, lane, t_prev, t_new);
                d_prev[lane] = t_prev;
                int ow = fp.blk_out_w[b] + lane;
                ow -= ow >= kAecmFrameRing ? kAecmFrameRing : 0;
                out_ring[ow] = t_new;
                wave_sync();
            }
            if (!pl.discard_out)
                for (int i = lane; i < kAecmFrame; i += 64) {
                    int r = fp.out_r + i;
                    r -= r >= kAecmFrameRing ? kAecmFrameRing : 0;
                    const int16_t v = out_ring[r];
                    for (int c = 0; c < chn; c++) op[(long)(f * kAecmFrame + i) * chn + c] = v;
                }
        }
    }
    wave_sync();
    {
        int4 *dst = reinterpret_cast<int4 *>(st);
        const int4 *src = reinterpret_cast<const int4 *>(W.st);
        for (int i = lane; i < A_WORDS / 4; i += 64) dst[i] = src[i];
    }
}

__global__ void aecm_set_cohort(int *stream_cohort, const int32_t *idx, int n_idx, int cohort) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_idx) stream_cohort[idx[j]] = cohort;
}

__global__ void aecm_fill_state(int32_t *state, const int32_t *tmpl, int words, int n_streams) {
    const size_t total = (size_t)words * n_streams;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        state[i] = tmpl[i % words];
}


// ---------------------------------------------------------------- coalescing control cohorts (wmx_aecm_coalesce; aec.hip has the long story)
constexpr int kAecmCoMax = 32;
struct AecmPairChecks {
    AecmPairCheck p[kAecmCoMax];
};
// flags[pair] = 1 when every far-end buffer of cohort b equals its rotation of cohort a's, bit for bit (the running thresholds of the
// binary far spectrum are integer IIRs from each cohort's own start: equal when they are, not before)
__global__ __launch_bounds__(256) void aecm_cohort_equal(AecmFarBufs F_all, AecmPairChecks pairs, int *flags) {
    const AecmPairCheck pc = pairs.p[blockIdx.x];
    const AecmFarBufs A = far_cohort(F_all, pc.a), B = far_cohort(F_all, pc.b);
    unsigned diff = 0;
    auto rows = [&](auto *a, auto *b, int n_rows, int row_len, int d_row) {
        for (int i = threadIdx.x; i < n_rows * row_len; i += 256) {
            const int r = i / row_len, c = i - r * row_len;
            int rb = r + d_row;
            rb -= rb >= n_rows ? n_rows : 0;
            diff |= (unsigned)(a[i] ^ b[rb * row_len + c]);
        }
    };
    rows(A.ring, B.ring, kAecmFarRing, 1, pc.d_ring);
    rows(A.frame, B.frame, kAecmFrameRing, 1, pc.d_frame);
    rows(A.old, B.old, 2 * kAecmFrame, 1, 0);
    rows(A.x_prev, B.x_prev, kAecmPart, 1, 0);
    rows(A.mean_far, B.mean_far, 33, 1, 0);
    rows(A.hist, B.hist, kAecmHist, kAecmBP, pc.d_hist);
    rows(A.hist_q, B.hist_q, kAecmHist, 1, pc.d_hist);
    rows(A.hist_bin, B.hist_bin, kAecmHist, 1, pc.d_hist);
    const int any = __syncthreads_or(diff != 0);
    if (threadIdx.x == 0) flags[blockIdx.x] = any ? 0 : 1;
}
// one wave per stream: members of a `from` cohort get their near and out frame rings rotated to the positions of the cohort they join
__global__ __launch_bounds__(256) void aecm_merge_streams(int32_t *state, int *stream_cohort, int n_streams, AecmPairChecks pairs, int n_pairs) {
    const int s = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= n_streams) return;
    const int c = stream_cohort[s];
    int k = -1;
    for (int i = 0; i < n_pairs; i++)
        if (pairs.p[i].b == c) k = i;
    if (k < 0) return;
    int16_t *st = reinterpret_cast<int16_t *>(state + (size_t)s * A_WORDS);
    // position p of b's ring is position p - d of a's: contents move by ring - d
    const int d[2] = {kAecmFrameRing - pairs.p[k].d_frame, kAecmFrameRing - pairs.p[k].d_out};
    const int base[2] = {2 * A_NEAR_RING, 2 * A_OUT_RING};  // int16 offsets
    int16_t v[2][3];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) v[r][j] = lane + 64 * j < kAecmFrameRing ? st[base[r] + lane + 64 * j] : (int16_t)0;
    __builtin_amdgcn_s_waitcnt(0);  // every load of the wave before its first store: the rotation is in place
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = lane + 64 * j;
            if (i < kAecmFrameRing) st[base[r] + (i + d[r]) % kAecmFrameRing] = v[r][j];
        }
    if (lane == 0) stream_cohort[s] = pairs.p[k].a;
}
__global__ void aecm_clamp_cohort(int *stream_cohort, int n_streams, int n_cohorts) {
    const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (s < n_streams && stream_cohort[s] >= n_cohorts) stream_cohort[s] = 0;
}
}  // namespace
}  // namespace wmx

// ------------------------------------------------------------------------------------ host
struct wmx_aecm {
    int device;  // the HIP device the state lives on (current device at create); every entry point switches to it
    int n_streams, chn, freq, pkg;
    int n_cohorts;
    std::vector<wmx::AecmCtl> ctl;  // one control plane per cohort -- kept up to date for the LEADERS of the control-plane classes only
    // control-plane classes, as in aec.hip (wmx_aec::lead; the bookkeeping is aec_ctl.h's, shared): cohorts started together and
    // called alike run ONE plane and get ONE plan per packet, however many far-ends they hear
    std::vector<int32_t> lead;         // [n_cohorts]
    std::vector<int32_t> cls_leader;   // [n_cls]
    std::vector<int32_t> h_plan_of[2]; // alternating sources of the asynchronous upload
    int h_plan_of_sel;
    int32_t *d_plan_of;                // [cap_cohorts]
    bool cls_dirty;
    int32_t *d_state;
    int32_t *d_tmpl;        // the state aec_init gives a stream (reset_streams refills from it)
    int *d_stream_cohort;   // [n_streams] cohort of each stream, or nullptr with one cohort
    wmx::StreamLife life;
    wmx::AecmConsts *d_consts;
    void *d_far;  // one allocation carved into AecmFarBufs
    wmx::AecmFarBufs far;
    wmx::AecmPlan *d_plans[2];  // double-buffered: a chunk's plans stay untouched while its kernels may still run
    wmx::AecmPlan *h_plans[2];  // pinned mirrors: the asynchronous copy reads them in place
    hipEvent_t plan_free[2];    // recorded behind the kernels that read d_plans[i]; waited for before it (and its mirror) is rewritten
    bool plan_used[2];
    int plan_sel;
    int cap_cohorts;            // cohorts the far slabs and plan slots are allocated for; grows by doubling (wmx_aecm_add_cohort)
    size_t far_bytes;           // bytes of one cohort's far-end slab
    std::vector<uint8_t> live;  // [n_cohorts] 0: retired, never called, the id is handed out again
    std::vector<int> rc_g;             // per-call scratch kept with the handle
    std::vector<int32_t> same_delay;
    // wmx_aecm_coalesce: the pairs whose device comparison is in flight (`b` < 0: dropped)
    wmx::AecmPairChecks co_pairs;
    int co_n;
    bool co_inflight;
    int *d_co_flags, *h_co_flags;
    hipEvent_t co_done;
    long co_calls;
    std::vector<long> co_retry_at;
    long last_far_group_stride;
    // the far kernel on a side stream beside whatever the caller launches in front of the near kernel (wmx::aecm_fork_far; aec.hip)
    hipStream_t side;
    hipEvent_t ev_fork, ev_join;
    bool fork_pending;
};
static void aecm_co_drop(wmx_aecm *h, int cohort) {
    for (int i = 0; i < h->co_n; i++)
        if (h->co_pairs.p[i].a == cohort || h->co_pairs.p[i].b == cohort) h->co_pairs.p[i].b = -1;
}

extern "C" {

int wmx_aecm_destroy(wmx_aecm *h) {
    WMX_ON_DEVICE(h);
    if (!h) return 0;
    if (h->d_state) (void)hipFree(h->d_state);
    if (h->d_consts) (void)hipFree(h->d_consts);
    if (h->d_far) (void)hipFree(h->d_far);
    if (h->d_plan_of) (void)hipFree(h->d_plan_of);
    if (h->d_plans[0]) (void)hipFree(h->d_plans[0]);
    if (h->h_plans[0]) (void)hipHostFree(h->h_plans[0]);
    if (h->d_tmpl) (void)hipFree(h->d_tmpl);
    if (h->d_stream_cohort) (void)hipFree(h->d_stream_cohort);
    if (h->d_co_flags) (void)hipFree(h->d_co_flags);
    if (h->h_co_flags) (void)hipHostFree(h->h_co_flags);
    if (h->co_done) (void)hipEventDestroy(h->co_done);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    h->life.release();
    for (int i = 0; i < 2; i++)
        if (h->plan_free[i]) (void)hipEventDestroy(h->plan_free[i]);
    delete h;
    return 0;
}

static void aecm_carve_far(wmx_aecm *h) {  // the far-end allocation (32-bit arrays first: alignment); one slab per cohort
    using namespace wmx;
    char *p = static_cast<char *>(h->d_far);
    h->far.mean_far = reinterpret_cast<int32_t *>(p);
    p += sizeof(int32_t) * 36;
    h->far.hist_q = reinterpret_cast<int32_t *>(p);
    p += sizeof(int32_t) * kAecmHist;
    h->far.hist_bin = reinterpret_cast<uint32_t *>(p);
    p += sizeof(uint32_t) * kAecmHist;
    h->far.hist = reinterpret_cast<uint16_t *>(p);
    p += sizeof(uint16_t) * (size_t)kAecmHist * kAecmBP;
    h->far.ring = reinterpret_cast<int16_t *>(p);
    p += sizeof(int16_t) * kAecmFarRing;
    h->far.old = reinterpret_cast<int16_t *>(p);
    p += sizeof(int16_t) * 2 * kAecmFrame;
    h->far.frame = reinterpret_cast<int16_t *>(p);
    p += sizeof(int16_t) * kAecmFrameRing;
    h->far.x_prev = reinterpret_cast<int16_t *>(p);
    h->far.group_bytes = h->far_bytes;
}

// far slabs and plan slots for `cap` cohorts (existing slabs carried over); everything new is allocated before anything old is let go
static int aecm_reserve(wmx_aecm *h, int cap) {
    using namespace wmx;
    if (cap <= h->cap_cohorts) return 0;
    WMX_HIP_RC(hipDeviceSynchronize());
    int ncap = h->cap_cohorts > 0 ? h->cap_cohorts : 1;
    while (ncap < cap) ncap *= 2;
    void *nf = nullptr;
    AecmPlan *nd = nullptr, *nh = nullptr;
    int32_t *npo = nullptr;
    const size_t plan_bytes = 2 * (size_t)ncap * kAecmMaxPktPerLaunch * sizeof(AecmPlan);
    hipError_t e = hipMalloc(&nf, h->far_bytes * (size_t)ncap);
    if (e == hipSuccess) e = hipMalloc(&npo, sizeof(int32_t) * (size_t)ncap);
    if (e == hipSuccess) e = hipMalloc(&nd, plan_bytes);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&nh), plan_bytes, hipHostMallocDefault);
    if (e == hipSuccess && h->d_far) e = hipMemcpy(nf, h->d_far, h->far_bytes * (size_t)h->cap_cohorts, hipMemcpyDeviceToDevice);
    if (e == hipSuccess)
        e = hipMemset(static_cast<char *>(nf) + h->far_bytes * (size_t)h->cap_cohorts, 0, h->far_bytes * (size_t)(ncap - h->cap_cohorts));
    if (e != hipSuccess) {
        if (nf) (void)hipFree(nf);
        if (nd) (void)hipFree(nd);
        if (nh) (void)hipHostFree(nh);
        if (npo) (void)hipFree(npo);
        return hip_fail(e, "growing the cohort buffers", __FILE__, __LINE__);
    }
    if (h->d_far) (void)hipFree(h->d_far);
    if (h->d_plan_of) (void)hipFree(h->d_plan_of);
    h->d_plan_of = npo;
    h->cls_dirty = true;  // the new array holds nothing yet
    if (h->d_plans[0]) (void)hipFree(h->d_plans[0]);
    if (h->h_plans[0]) (void)hipHostFree(h->h_plans[0]);
    h->d_far = nf;
    h->d_plans[0] = nd;
    h->d_plans[1] = nd + (size_t)ncap * kAecmMaxPktPerLaunch;
    h->h_plans[0] = nh;
    h->h_plans[1] = nh + (size_t)ncap * kAecmMaxPktPerLaunch;
    h->plan_used[0] = h->plan_used[1] = false;  // drained above
    h->cap_cohorts = ncap;
    aecm_carve_far(h);
    return 0;
}

int wmx_aecm_create(wmx_aecm **out, int n_streams, int chn, int freq, int interval_ms) {
    return wmx_aecm_create_cohorts(out, n_streams, chn, freq, interval_ms, 1);
}

int wmx_aecm_create_cohorts(wmx_aecm **out, int n_streams, int chn, int freq, int interval_ms, int n_cohorts) {
    using namespace wmx;
    if (!out) return WMX_EINVAL;
    *out = nullptr;
    if (n_cohorts < 1) {
        set_error("wmx_aecm_create_cohorts: n_cohorts=%d", n_cohorts);
        return WMX_EINVAL;
    }
    // aec_init (src/webrtc.c:220-221) + WebRtcAecm_Init (echo_control_mobile.c:186-190)
    if ((freq != 8000 && freq != 16000) || chn < 1 || n_streams < 1) {
        set_error("wmx_aecm_create: unsupported n_streams=%d chn=%d freq=%d", n_streams, chn, freq);
        return WMX_EINVAL;
    }
    wmx_aecm *h = new wmx_aecm();
    if ((h->device = current_device()) < 0) {
        delete h;
        return WMX_ENODEV;
    }
    h->n_streams = n_streams;
    h->chn = chn;
    h->freq = freq;
    h->pkg = freq / 1000 * ((freq <= 8000 && interval_ms % 20 == 0) ? 20 : 10);  // src/webrtc.c:239-248
    h->n_cohorts = n_cohorts;
    h->ctl.resize((size_t)n_cohorts);
    for (AecmCtl &c : h->ctl) c.init(freq);
    h->lead.assign((size_t)n_cohorts, 0);  // made together, equal planes: one class led by cohort 0 until something tells them apart
    h->h_plan_of_sel = 0;
    h->d_plan_of = nullptr;
    h->cls_dirty = true;
    h->d_tmpl = nullptr;
    h->d_stream_cohort = nullptr;
    h->co_n = 0;
    h->co_inflight = false;
    h->d_co_flags = h->h_co_flags = nullptr;
    h->co_done = nullptr;
    h->co_calls = 0;
    h->co_retry_at.assign((size_t)n_cohorts, 0);
    h->last_far_group_stride = 0;
    h->side = nullptr;
    h->ev_fork = h->ev_join = nullptr;
    h->fork_pending = false;
    h->d_state = nullptr;
    h->d_consts = nullptr;
    h->d_far = nullptr;
    h->d_plans[0] = h->d_plans[1] = nullptr;
    h->h_plans[0] = h->h_plans[1] = nullptr;
    h->plan_free[0] = h->plan_free[1] = nullptr;
    h->plan_used[0] = h->plan_used[1] = false;
    h->plan_sel = 0;
    h->cap_cohorts = 0;
    h->live.assign((size_t)n_cohorts, 1);

    AecmConsts *K = new AecmConsts();
    memset(K, 0, sizeof(*K));
    spl_twiddles(fx_spl_sin1024, &K->tw);
    memcpy(K->sqrt_hanning, fx_aecm_sqrt_hanning, sizeof(fx_aecm_sqrt_hanning));
    memcpy(K->cos360, fx_aecm_cos, sizeof(fx_aecm_cos));
    memcpy(K->sin360, fx_aecm_sin, sizeof(fx_aecm_sin));
    {
        uint32_t a = 1, c = 0;  // x -> 69069 x + 1 composed k + 1 times (randomization_functions.c:92-96)
        for (int k = 0; k < 64; k++) {
            a = a * 69069u;
            c = c * 69069u + 1u;
            K->lcg_a[k] = a;
            K->lcg_c[k] = c;
        }
    }
    // initial per-stream state: WebRtcAecm_InitCore aecm_core.c:401-546, WebRtc_InitBinaryDelayEstimator
    std::vector<int32_t> st(A_WORDS, 0);
    {
        int16_t *cs = reinterpret_cast<int16_t *>(&st[A_CH_STORED]), *c16 = reinterpret_cast<int16_t *>(&st[A_CH_ADAPT16]);
        const int16_t *path = freq == 8000 ? fx_aecm_channel_8k : fx_aecm_channel_16k;
        for (int i = 0; i < 65; i++) {
            cs[i] = c16[i] = path[i];
            st[A_CH_ADAPT32 + i] = (int32_t)((uint32_t)(int32_t)path[i] << 16);
        }
        int32_t t32 = 65 * 65;
        int16_t t16 = 65;
        int i;
        for (i = 0; i < (65 >> 1) - 1; i++) {
            st[A_NOISE_EST + i] = (int32_t)((uint32_t)t32 << 8);
            t16--;
            t32 -= (int32_t)((t16 << 1) + 1);
        }
        for (; i < 65; i++) st[A_NOISE_EST + i] = (int32_t)((uint32_t)t32 << 8);
        for (i = 0; i <= kAecmMaxDelay; i++) st[A_MEAN_BITS + i] = 20 << 9;
        int32_t *sc = &st[A_SCAL];
        sc[M_FIRST_VAD] = 1;
        sc[M_E_MIN] = 32767;
        sc[M_E_MAX] = -32768;
        sc[M_E_VAD] = 1025;
        sc[M_SUP_GAIN] = 256;
        sc[M_SUP_GAIN_OLD] = 256;
        sc[M_MSE_ADAPT_OLD] = 1000;
        sc[M_MSE_STORED_OLD] = 1000;
        sc[M_MSE_THR] = 0x7FFFFFFF;
        sc[M_SEED] = 666;
        sc[M_MIN_PROB] = 32 << 9;
        sc[M_LAST_PROB] = 32 << 9;
        sc[M_LAST_DELAY] = -2;
    }
    const size_t far_raw = sizeof(int16_t) * (kAecmFarRing + 2 * kAecmFrame + kAecmFrameRing + 64 + 8) + sizeof(int32_t) * (36 + kAecmHist) +
                           sizeof(uint16_t) * (size_t)kAecmHist * kAecmBP + sizeof(uint32_t) * kAecmHist + 64;
    h->far_bytes = (far_raw + 255) / 256 * 256;  // per cohort
    hipError_t e;
#define AECM_TRY(x)                                         \
    if ((e = (x)) != hipSuccess) {                          \
        const int rc = hip_fail(e, #x, __FILE__, __LINE__); \
        wmx_aecm_destroy(h);                                \
        delete K;                                           \
        return rc;                                          \
    }
    AECM_TRY(hipMalloc(&h->d_state, (size_t)A_WORDS * n_streams * sizeof(int32_t)));
    AECM_TRY(hipMalloc(&h->d_consts, sizeof(AecmConsts)));
    {
        const int rc = aecm_reserve(h, n_cohorts);
        if (rc != 0) {
            wmx_aecm_destroy(h);
            delete K;
            return rc;
        }
    }
    if (n_cohorts > 1) {
        AECM_TRY(hipMalloc(&h->d_stream_cohort, sizeof(int) * n_streams));
        AECM_TRY(hipMemset(h->d_stream_cohort, 0, sizeof(int) * n_streams));
    }
    AECM_TRY(hipEventCreateWithFlags(&h->plan_free[0], hipEventDisableTiming));
    AECM_TRY(hipEventCreateWithFlags(&h->plan_free[1], hipEventDisableTiming));
    AECM_TRY(hipMalloc(&h->d_tmpl, A_WORDS * sizeof(int32_t)));
    AECM_TRY(hipMemcpy(h->d_consts, K, sizeof(AecmConsts), hipMemcpyHostToDevice));
    AECM_TRY(hipMemcpy(h->d_tmpl, st.data(), A_WORDS * sizeof(int32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(aecm_fill_state, dim3(1024), dim3(256), 0, nullptr, h->d_state, h->d_tmpl, (int)A_WORDS, n_streams);
    AECM_TRY(hipGetLastError());
    AECM_TRY(hipDeviceSynchronize());
#undef AECM_TRY
    delete K;
    *out = h;
    return 0;
}

int wmx_aecm_packet_samples(const wmx_aecm *h) { return h ? h->pkg * h->chn : WMX_EINVAL; }
int wmx_aecm_state_bytes(const wmx_aecm *h) { return h ? (int)wmx::A_WORDS * 4 : WMX_EINVAL; }

// Same contract as wmx_aec_run: mode bit 1 = aec_setFrameFar, bit 2 = aec_process, 3 = aec_process2.
int wmx_aecm_run(wmx_aecm *h, int mode, const int16_t *d_far, long far_packet_stride, const int16_t *d_near, int16_t *d_out, int n_packets,
                 long stream_stride, long packet_stride, int delay_ms, void *stream) {
    if (!h) {
        wmx::set_error("wmx_aecm_run: bad argument");
        return WMX_EINVAL;
    }
    h->same_delay.assign((size_t)h->n_cohorts, delay_ms);
    return wmx_aecm_run_cohorts(h, mode, d_far, far_packet_stride, 0, d_near, d_out, n_packets, stream_stride, packet_stride,
                                h->same_delay.data(), nullptr, nullptr, stream);
}

// Same contract as wmx_aec_run_cohorts (include/wmix_amd.h): one reported delay, one on/off byte and one return code per cohort.
int wmx_aecm_run_cohorts(wmx_aecm *h, int mode, const int16_t *d_far, long far_packet_stride, long far_group_stride, const int16_t *d_near,
                         int16_t *d_out, int n_packets, long stream_stride, long packet_stride, const int32_t *delay_ms,
                         const uint8_t *cohort_on, int32_t *cohort_rc, void *stream) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    using wmx::aec_ctl;
    // a fork point belongs to THIS call, whichever way it ends (as in wmx_aec_run_cohorts)
    const bool fork_here = h && h->fork_pending;
    if (h) h->fork_pending = false;
    if (!h || n_packets < 0 || (mode & 3) == 0 || !delay_ms) {
        set_error("wmx_aecm_run: bad argument");
        return WMX_EINVAL;
    }
    const int G = h->n_cohorts;
    if (cohort_rc)
        for (int g = 0; g < G; g++) cohort_rc[g] = 0;
    if (n_packets == 0) return 0;
    if (((mode & 1) && !d_far) || ((mode & 2) && (!d_near || !d_out))) {
        set_error("wmx_aecm_run: null buffer");
        return WMX_EINVAL;
    }
    const long per_pkt = (long)h->pkg * h->chn;
    if ((mode & 2) && (packet_stride < per_pkt || (h->n_streams > 1 && stream_stride < per_pkt))) {
        set_error("wmx_aecm_run: strides (%ld, %ld) smaller than a packet (%ld samples)", stream_stride, packet_stride, per_pkt);
        return WMX_EINVAL;
    }
    if ((mode & 1) && far_packet_stride < per_pkt && n_packets > 1) {
        set_error("wmx_aecm_run: far stride %ld smaller than a packet (%ld samples)", far_packet_stride, per_pkt);
        return WMX_EINVAL;
    }
    hipStream_t s = as_stream(stream);
    std::vector<int> &rc_g = h->rc_g;
    rc_g.assign((size_t)G, 0);
    int rc_first = 0, running = 0;
    for (int g = 0; g < G; g++) running += (h->live[(size_t)g] && (!cohort_on || cohort_on[g])) ? 1 : 0;
    h->last_far_group_stride = (mode & 1) ? far_group_stride : h->last_far_group_stride;
    // pairs whose comparison is in flight (wmx_aecm_coalesce) stay candidates only while the two cohorts are called identically
    for (int i = 0; i < h->co_n; i++) {
        AecmPairCheck &pc = h->co_pairs.p[i];
        if (pc.b < 0) continue;
        const bool on_a = !cohort_on || cohort_on[pc.a], on_b = !cohort_on || cohort_on[pc.b];
        if (on_a != on_b || (on_a && delay_ms[pc.a] != delay_ms[pc.b]) || ((mode & 1) && far_group_stride != 0)) pc.b = -1;
    }
    // control-plane classes (aec.hip / aec_ctl.h): a follower that is called differently from its leader in THIS call takes a plane of
    // its own first; then one plane and one plan per class and packet
    if (G > 1 && (cohort_on || delay_ms != h->same_delay.data())) wmx::aec_classes_split(h, delay_ms, cohort_on);
    bool classes_moved = false;
    if (h->cls_dirty) {
        // uploaded in `s`, behind every launch that still reads the old classes; a far kernel forked onto the side stream would not
        // wait for it: this one launch keeps the far kernel in line
        h->h_plan_of_sel ^= 1;
        std::vector<int32_t> &po = h->h_plan_of[h->h_plan_of_sel];
        wmx::aec_classes_list(h, h->cls_leader, po);
        if (G > 1) WMX_HIP(hipMemcpyAsync(h->d_plan_of, po.data(), sizeof(int32_t) * (size_t)G, hipMemcpyHostToDevice, s));
        h->cls_dirty = false;
        classes_moved = true;
    }
    const int C = (int)h->cls_leader.size();
    const int32_t *plan_of = G > 1 ? h->d_plan_of : nullptr;
    running = 0;
    for (int c = 0; c < C; c++) {
        const int g = h->cls_leader[(size_t)c];
        running += (h->live[(size_t)g] && (!cohort_on || cohort_on[g])) ? 1 : 0;
    }
    for (int done = 0; done < n_packets && running > 0;) {
        int chunk = n_packets - done;
        if (chunk > kAecmMaxPktPerLaunch) chunk = kAecmMaxPktPerLaunch;
        // the next plan slot: its pinned host half and its device half are rewritten only after the kernels that read the device
        // half last have finished, whatever stream they ran on
        const int sel = h->plan_sel;
        h->plan_sel ^= 1;
        if (h->plan_used[sel]) WMX_HIP(hipEventSynchronize(h->plan_free[sel]));
        AecmPlan *hp = h->h_plans[sel], *dp = h->d_plans[sel];  // [packet][class], C apart: chunk x C plans
        int any = 0;
        for (int c = 0; c < C; c++) {
            const int g = h->cls_leader[(size_t)c];  // the class's one control plane
            const bool on = h->live[(size_t)g] && (!cohort_on || cohort_on[g]) && rc_g[g] == 0;
            for (int k = 0; k < chunk; k++) {
                AecmPlan &pl = hp[(size_t)k * C + c];
                memset(&pl, 0, sizeof(pl));
                if (!on || rc_g[g] != 0) continue;  // has_far = has_near = 0: both kernels skip the packet for this class's cohorts
                any = 1;
                if (mode & 1) {
                    const int r = h->ctl[(size_t)g].buffer_farend(h->pkg, &pl);
                    if (r != 0) {
                        pl.has_far = 0;
                        rc_g[g] = r;
                        continue;
                    }
                }
                if (mode & 2) {
                    const int r = h->ctl[(size_t)g].process(h->pkg, delay_ms[g], &pl);
                    if (r != 0) {
                        // WebRtcAecm_Process has PROCESSED the packet with the delay clamped (state advances) and returns -1; the
                        // wmix wrapper then returns without copying its output (src/webrtc.c:382-387).  Same here: the kernel
                        // runs the packet but leaves the caller's buffer alone; later packets are not touched.
                        pl.discard_out = 1;
                        rc_g[g] = r;
                    }
                }
            }
            if (on && rc_g[g] != 0) {
                running--;
                if (rc_first == 0) rc_first = rc_g[g];
            }
        }
        if (any) {
            const int by_value = (chunk == 1 && C == 1 && G == 1) ? 1 : 0;
            // the far kernel on the side stream when the caller forked it (first chunk of the call only)
            const bool forked = fork_here && done == 0 && (mode & 2) && !classes_moved;
            hipStream_t fs = forked ? h->side : s;
            if (forked) WMX_HIP(hipStreamWaitEvent(fs, h->ev_fork, 0));
            // every cohort its own class: each far wave fetches ITS plans from the pinned host slot (no copy-engine operation in front of
            // the launch); classes shared by several cohorts: one small upload, and every member reads the class's plan on the device
            const bool from_host = !by_value && C == G;
            if (!by_value && !from_host) WMX_HIP(hipMemcpyAsync(dp, hp, (size_t)C * chunk * sizeof(AecmPlan), hipMemcpyHostToDevice, fs));
            hipLaunchKernelGGL(aecm_far_kernel, dim3((unsigned)G), dim3(64), 0, fs, h->far, h->d_consts, dp, chunk, C, plan_of,
                               d_far ? d_far + (size_t)done * far_packet_stride : nullptr, far_packet_stride, far_group_stride, h->chn, by_value,
                               hp[0], from_host ? hp : nullptr);
            WMX_LAUNCH_CHECK();
            if (forked) {
                WMX_HIP(hipEventRecord(h->ev_join, fs));
                WMX_HIP(hipStreamWaitEvent(s, h->ev_join, 0));
            }
            if (mode & 2) {
                const unsigned grid = (unsigned)((h->n_streams + kAecmWavesPerBlock - 1) / kAecmWavesPerBlock);
                hipLaunchKernelGGL(aecm_near_kernel, dim3(grid), dim3(64 * kAecmWavesPerBlock), 0, s, h->d_state, h->far, h->d_consts, dp, chunk, C,
                                   plan_of, d_near + (size_t)done * packet_stride, d_out + (size_t)done * packet_stride, h->n_streams,
                                   stream_stride, packet_stride, h->chn, h->pkg, h->freq / 8000, h->d_stream_cohort, h->life.d_active);
                WMX_LAUNCH_CHECK();
            }
            WMX_HIP(hipEventRecord(h->plan_free[sel], s));
            h->plan_used[sel] = true;
        }
        done += chunk;
    }
    if (cohort_rc)
        for (int g = 0; g < G; g++) cohort_rc[g] = (h->live[(size_t)g] && (!cohort_on || cohort_on[g])) ? rc_g[(size_t)h->lead[(size_t)g]] : 0;
    return rc_first;
}

}  // extern "C"
// Library-internal (wmx_internal.h): from this point of `stream` on, the far-end packets of the NEXT wmx_aecm_run_* call on this
// handle are in place; its far kernel may start here, on the handle's side stream, beside whatever the caller launches on `stream`
// between now and that call (wmx_chain_process: the NSX).  The near kernel still runs on `stream`, behind the far kernel.
int wmx::aecm_fork_far(wmx_aecm *h, hipStream_t s) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    if (!h->side) {
        WMX_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
        WMX_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        WMX_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    WMX_HIP(hipEventRecord(h->ev_fork, s));
    h->fork_pending = true;
    return 0;
}
void wmx::aecm_cancel_fork(wmx_aecm *h) {
    if (h) h->fork_pending = false;
}
extern "C" {

// stream / cohort migration, as for the float AEC
static constexpr uint32_t kAecmBlobVersion = 1;  // bump when the meaning of a state word changes (wmx_internal.h: blob_layout)
int wmx_aecm_stream_state_bytes(const wmx_aecm *h) { return h ? (int)(sizeof(wmx::BlobHeader) + wmx::A_WORDS * 4) : WMX_EINVAL; }
int wmx_aecm_cohort_state_bytes(const wmx_aecm *h) {
    return h ? (int)(sizeof(wmx::BlobHeader) + sizeof(wmx::AecmCtl) + h->far.group_bytes) : WMX_EINVAL;
}

int wmx_aecm_export_stream(wmx_aecm *h, int stream_index, void *host_blob) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || !host_blob || stream_index < 0 || stream_index >= h->n_streams) return WMX_EINVAL;
    WMX_HIP(hipDeviceSynchronize());
    char *p = static_cast<char *>(host_blob);
    blob_begin(p, blob_tag("AECM"), blob_layout((uint32_t)h->freq, kAecmBlobVersion), A_WORDS * 4);
    WMX_HIP(hipMemcpy(p + sizeof(BlobHeader), h->d_state + (size_t)stream_index * A_WORDS, A_WORDS * 4, hipMemcpyDeviceToHost));
    return 0;
}

int wmx_aecm_import_stream(wmx_aecm *h, int stream_index, const void *host_blob, int cohort) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || !host_blob || stream_index < 0 || stream_index >= h->n_streams || cohort < -1 || cohort >= h->n_cohorts) return WMX_EINVAL;
    const int rc = blob_check(host_blob, blob_tag("AECM"), blob_layout((uint32_t)h->freq, kAecmBlobVersion), A_WORDS * 4);
    if (rc) return rc;
    WMX_HIP(hipDeviceSynchronize());
    WMX_HIP(hipMemcpy(h->d_state + (size_t)stream_index * A_WORDS, static_cast<const char *>(host_blob) + sizeof(BlobHeader), A_WORDS * 4,
                      hipMemcpyHostToDevice));
    if (cohort >= 0 && h->d_stream_cohort) WMX_HIP(hipMemcpy(h->d_stream_cohort + stream_index, &cohort, sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

int wmx_aecm_export_cohort(wmx_aecm *h, int cohort, void *host_blob) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || !host_blob || cohort < 0 || cohort >= h->n_cohorts) return WMX_EINVAL;
    WMX_HIP(hipDeviceSynchronize());
    char *p = static_cast<char *>(host_blob);
    blob_begin(p, blob_tag("AEMc"), blob_layout((uint32_t)h->freq, kAecmBlobVersion), (uint32_t)(sizeof(AecmCtl) + h->far.group_bytes));
    p += sizeof(BlobHeader);
    memcpy(p, &wmx::aec_ctl(h, cohort), sizeof(AecmCtl));
    WMX_HIP(hipMemcpy(p + sizeof(AecmCtl), static_cast<char *>(h->d_far) + (size_t)cohort * h->far.group_bytes, h->far.group_bytes,
                      hipMemcpyDeviceToHost));
    return 0;
}

int wmx_aecm_import_cohort(wmx_aecm *h, int cohort, const void *host_blob) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || !host_blob || cohort < 0 || cohort >= h->n_cohorts) return WMX_EINVAL;
    const int rc = blob_check(host_blob, blob_tag("AEMc"), blob_layout((uint32_t)h->freq, kAecmBlobVersion), (uint32_t)(sizeof(AecmCtl) + h->far.group_bytes));
    if (rc) return rc;
    WMX_HIP(hipDeviceSynchronize());
    const char *p = static_cast<const char *>(host_blob) + sizeof(BlobHeader);
    wmx::aec_ctl_own(h, cohort);
    memcpy(&h->ctl[(size_t)cohort], p, sizeof(AecmCtl));
    wmx::aec_ctl_join(h, cohort);
    aecm_co_drop(h, cohort);
    WMX_HIP(hipMemcpy(static_cast<char *>(h->d_far) + (size_t)cohort * h->far.group_bytes, p + sizeof(AecmCtl), h->far.group_bytes,
                      hipMemcpyHostToDevice));
    return 0;
}

int wmx_aecm_cohorts(const wmx_aecm *h) { return h ? h->n_cohorts : WMX_EINVAL; }
int wmx_aecm_cohort_key(const wmx_aecm *h, int cohort, int32_t *key8) {
    if (!h || !key8 || cohort < 0 || cohort >= h->n_cohorts) return WMX_EINVAL;
    wmx::AecmCoKey k;
    if (!h->live[(size_t)cohort] || !wmx::aecm_co_key(wmx::aec_ctl(h, cohort), &k)) return 1;  // retired, or still in its start-up
    for (int i = 0; i < 8; i++) key8[i] = k.v[i];
    return 0;
}
int wmx_aecm_live_cohorts(const wmx_aecm *h) {
    if (!h) return WMX_EINVAL;
    int n = 0;
    for (uint8_t l : h->live) n += l ? 1 : 0;
    return n;
}

// wmx_aec_coalesce for the fixed-point canceller (include/wmix_amd.h): completes the merges whose device comparison came back equal,
// then proposes up to max_pairs new pairs and launches their comparison behind the work already in `stream`.
int wmx_aecm_coalesce(wmx_aecm *h, int max_pairs, int32_t *merged_from, int32_t *merged_into, int cap, int *n_merged, void *stream) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (n_merged) *n_merged = 0;
    if (!h || max_pairs < 0 || cap < 0 || (cap > 0 && (!merged_from || !merged_into))) return WMX_EINVAL;
    hipStream_t s = as_stream(stream);
    h->co_calls++;
    if (h->n_cohorts < 2 || !h->d_stream_cohort) return 0;
    if (!h->d_co_flags) {
        WMX_HIP(hipMalloc(&h->d_co_flags, sizeof(int) * kAecmCoMax));
        WMX_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->h_co_flags), sizeof(int) * kAecmCoMax, hipHostMallocDefault));
        WMX_HIP(hipEventCreateWithFlags(&h->co_done, hipEventDisableTiming));
    }
    int merged = 0;
    if (h->co_inflight) {
        const hipError_t q = hipEventQuery(h->co_done);
        if (q == hipErrorNotReady) return 0;
        if (q != hipSuccess) return hip_fail(q, "hipEventQuery(co_done)", __FILE__, __LINE__);
        h->co_inflight = false;
        AecmPairChecks go;
        int n_go = 0;
        for (int i = 0; i < h->co_n; i++) {
            AecmPairCheck pc = h->co_pairs.p[i];
            if (pc.b < 0) continue;
            AecmCoKey ka, kb;
            const bool ok = h->h_co_flags[i] == 1 && h->live[(size_t)pc.a] && h->live[(size_t)pc.b] &&
                            aecm_co_key(aec_ctl(h, pc.a), &ka) && aecm_co_key(aec_ctl(h, pc.b), &kb) && ka == kb;
            if (!ok) {
                h->co_retry_at[(size_t)pc.b] = h->co_calls + 64;
                continue;
            }
            if (merged >= cap) continue;
            aecm_co_pair(aec_ctl(h, pc.a), aec_ctl(h, pc.b), pc.a, pc.b, &pc);
            go.p[n_go++] = pc;
            merged_from[merged] = pc.b;
            merged_into[merged] = pc.a;
            merged++;
        }
        h->co_n = 0;
        if (n_go > 0) {
            hipLaunchKernelGGL(aecm_merge_streams, dim3((unsigned)((h->n_streams + 3) / 4)), dim3(256), 0, s, h->d_state, h->d_stream_cohort,
                               h->n_streams, go, n_go);
            WMX_LAUNCH_CHECK();
            for (int i = 0; i < n_go; i++) {
                wmx::aec_ctl_own(h, go.p[i].b);  // a retired cohort leads nobody
                h->live[(size_t)go.p[i].b] = 0;
            }
            int nc = h->n_cohorts;
            while (nc > 1 && !h->live[(size_t)nc - 1]) nc--;
            if (nc < h->n_cohorts) {
                hipLaunchKernelGGL(aecm_clamp_cohort, dim3((unsigned)((h->n_streams + 255) / 256)), dim3(256), 0, s, h->d_stream_cohort, h->n_streams, nc);
                WMX_LAUNCH_CHECK();
                h->n_cohorts = nc;
                h->ctl.resize((size_t)nc);
                h->lead.resize((size_t)nc);
                h->cls_dirty = true;
                h->live.resize((size_t)nc);
                h->co_retry_at.resize((size_t)nc);
            }
        }
    }
    if (n_merged) *n_merged = merged;
    if (max_pairs == 0 || h->last_far_group_stride != 0) return 0;
    if (max_pairs > kAecmCoMax) max_pairs = kAecmCoMax;
    std::unordered_multimap<uint64_t, int> leads;
    leads.reserve((size_t)h->n_cohorts);
    int n = 0;
    for (int g = 0; g < h->n_cohorts && n < max_pairs; g++) {
        if (!h->live[(size_t)g]) continue;
        AecmCoKey k, kl;
        if (!aecm_co_key(aec_ctl(h, g), &k)) continue;
        uint64_t hash = 1469598103934665603ull;
        for (int v : k.v) hash = (hash ^ (uint32_t)v) * 1099511628211ull;
        int lead = -1;
        const auto range = leads.equal_range(hash);
        for (auto it = range.first; it != range.second && lead < 0; ++it)
            if (aecm_co_key(aec_ctl(h, it->second), &kl) && kl == k) lead = it->second;
        if (lead < 0) {
            leads.emplace(hash, g);
            continue;
        }
        if (h->co_retry_at[(size_t)g] > h->co_calls) continue;
        aecm_co_pair(aec_ctl(h, lead), aec_ctl(h, g), lead, g, &h->co_pairs.p[n++]);
    }
    h->co_n = n;
    if (n == 0) return 0;
    hipLaunchKernelGGL(aecm_cohort_equal, dim3((unsigned)n), dim3(256), 0, s, h->far, h->co_pairs, h->d_co_flags);
    WMX_LAUNCH_CHECK();
    WMX_HIP(hipMemcpyAsync(h->h_co_flags, h->d_co_flags, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s));
    WMX_HIP(hipEventRecord(h->co_done, s));
    h->co_inflight = true;
    return 0;
}


// aec_init for a cohort's shared part in the AECM build: control plane, far-end ring, farendOld, far spectrum history
int wmx_aecm_reset_cohort(wmx_aecm *h, int cohort, void *stream) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || cohort < 0 || cohort >= h->n_cohorts) return WMX_EINVAL;
    wmx::aec_ctl_own(h, cohort);
    h->ctl[(size_t)cohort].init(h->freq);
    wmx::aec_ctl_join(h, cohort);  // cohorts restarted at the same point of the packet sequence run one control plane
    aecm_co_drop(h, cohort);
    WMX_HIP(hipMemsetAsync(static_cast<char *>(h->d_far) + (size_t)cohort * h->far.group_bytes, 0, h->far.group_bytes, as_stream(stream)));
    return 0;
}

// A new cohort (a join time of its own), as wmx_aec_add_cohort: a retired id when there is one, else the next, buffers doubling
int wmx_aecm_add_cohort(wmx_aecm *h, int *cohort, void *stream) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || !cohort) return WMX_EINVAL;
    int id = -1;
    for (int g = 0; g < h->n_cohorts; g++)
        if (!h->live[(size_t)g]) {
            id = g;
            break;
        }
    if (id < 0) {
        id = h->n_cohorts;
        const int rc = aecm_reserve(h, id + 1);
        if (rc != 0) return rc;
        h->ctl.resize((size_t)id + 1);
        h->lead.push_back(id);
        h->cls_dirty = true;
        h->live.push_back(1);
        h->co_retry_at.push_back(0);
        h->n_cohorts = id + 1;
    }
    if (h->n_cohorts > 1 && !h->d_stream_cohort) {  // so far every stream was in cohort 0 by construction
        WMX_HIP(hipMalloc(&h->d_stream_cohort, sizeof(int) * h->n_streams));
        WMX_HIP(hipMemsetAsync(h->d_stream_cohort, 0, sizeof(int) * h->n_streams, as_stream(stream)));
    }
    h->live[(size_t)id] = 1;
    *cohort = id;
    return wmx_aecm_reset_cohort(h, id, stream);
}

int wmx_aecm_retire_cohort(wmx_aecm *h, int cohort) {
    if (!h || cohort < 0 || cohort >= h->n_cohorts) return WMX_EINVAL;
    wmx::aec_ctl_own(h, cohort);  // a retired cohort leads nobody
    h->live[(size_t)cohort] = 0;
    aecm_co_drop(h, cohort);
    return 0;
}

// aec_release + aec_init for the listed streams in the AECM build; cohort >= 0 also makes them members of that cohort
int wmx_aecm_reset_streams(wmx_aecm *h, const int32_t *idx, int n, int cohort, void *stream) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || n < 0 || (n > 0 && !idx) || cohort < -1 || cohort >= h->n_cohorts) return WMX_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = as_stream(stream);
    const int32_t *d_idx = nullptr;
    const int rc = h->life.upload(idx, n, h->n_streams, s, &d_idx);
    if (rc != 0) return rc;
    hipLaunchKernelGGL((fill_rows_idx<int32_t>), dim3((unsigned)(n < 4096 ? n : 4096)), dim3(256), 0, s, h->d_state, (const int32_t *)h->d_tmpl,
                       (int)A_WORDS, d_idx, n);
    if (cohort >= 0 && h->d_stream_cohort)
        hipLaunchKernelGGL(aecm_set_cohort, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h->d_stream_cohort, d_idx, n, cohort);
    WMX_LAUNCH_CHECK();
    return h->life.done(s);
}

int wmx_aecm_set_active(wmx_aecm *h, const uint8_t *host_mask, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    return h->life.set_active(h->n_streams, host_mask, wmx::as_stream(stream));
}

}  // extern "C"
