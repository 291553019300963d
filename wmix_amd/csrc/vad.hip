// vad.hip -- batched voice-activity gate for gfx950: one LANE per stream (vad_kernel), and for the mono 10 ms packets of the
// batched chain one lane per stream in FOUR waves that split a packet's work (vad_pipe_kernel, further down).
//
// Replaces, for many independent streams per launch, wmix's vad_process() (src/webrtc.c:91-151)
// over WebRtcVad_Process in mode 3 (W:common_audio/vad/webrtc_vad.c:71-104, vad_core.c:124-674,
// vad_filterbank.c:41-333, vad_gmm.c:30-83, vad_sp.c:27-177).  The whole path is a chain of
// short integer recurrences (all-pass decimators, GMM update, a 16-entry order-statistics
// filter), so there is nothing to parallelise inside a stream: 64 streams ride one wavefront,
// per-stream state is laid out field-major ([field][stream]) so every state access of the wave
// is one 128/256-byte line, and the sub-band signals of the 6-band split tree live in LDS as
// [sample][lane] int16 (bank-conflict free).  At one wave per SIMD the kernel's time is the length of
// one stream's dependency chain, so state is kept off the memory latency path: the order-statistics
// vectors (192 of the 263 int16 fields, walked element by element by FindMinimum) are staged in LDS for
// the launch, the other 76 state words live in registers (one batch of loads at entry, one of stores at
// the end), and a mono single-packet call also fetches its packet as uint4 rows in one batch, analyses
// and attenuates it in registers and writes it back the same way (other shapes: the packet's lines are
// pulled into L2 up front and read from memory).
// vad_pipe_kernel: the filter bank and the packet in one wave, two GMM channels in each of the other three, features and
// likelihood ratios through LDS at two workgroup barriers -- four waves per SIMD instead of one, half the time.
// Results are bit-exact with the reference; signed overflow that the reference leaves to
// two's-complement wrap is spelled out.
//
// Kept reference quirks: a call of several packets always analyses packet 0
// and only packet 0 is attenuated; multi-channel input is mean-downmixed in place and expanded
// backwards afterwards.
#include <cstdlib>
#include <vector>
#include "wmx_internal.h"
#include "spl_dev.h"

namespace wmx {
namespace {

// ---- field-major state: int16 words and int32 words (VadInstT, vad_core.h:27-56; Vad_Struct.reduce)
enum : int {
    V16_NOISE_MEANS = 0,
    V16_SPEECH_MEANS = 12,
    V16_NOISE_STDS = 24,
    V16_SPEECH_STDS = 36,
    V16_OVER_HANG = 48,
    V16_NUM_SPEECH = 49,
    V16_AGE = 50,         // index_vector[96]
    V16_LOW = 146,        // low_value_vector[96]
    V16_MEAN_VALUE = 242,
    V16_UPPER = 248,
    V16_LOWER = 253,
    V16_HP = 258,
    V16_REDUCE = 262,
    V16_WORDS = 263,
    V32_DS = 0,  // downsampling_filter_states[4]
    V32_FRAME_COUNTER = 4,
    V32_WORDS = 5,
};

constexpr int kVadMinFields = V16_MEAN_VALUE - V16_AGE;  // 192
constexpr int kVadRegFields = V16_WORDS - (V16_MEAN_VALUE - V16_AGE);  // 71: everything but the order statistics
struct VadRef {
    // The lane's own copy of the stream's state for the length of the launch: the 71 scalar int16 fields and the five
    // int32 words in REGISTERS (every index below is a compile-time constant once the channel / Gaussian loops are
    // unrolled), the order statistics in LDS.  Loaded in one batch at kernel entry, written back once at the end: the
    // dependency chain of a frame never waits on a state row again (it used to pay an L2 round trip per field).
    int16_t *r16;     // [kVadRegFields]
    int32_t *r32;     // [V32_WORDS]
    int16_t *minbuf;  // LDS copy of index_vector / low_value_vector (V16_AGE .. V16_MEAN_VALUE), element f at minbuf[f * 64]
    __device__ __forceinline__ int16_t &h(int f) const { return r16[f < V16_AGE ? f : f - (V16_MEAN_VALUE - V16_AGE)]; }
    __device__ __forceinline__ int32_t &w(int f) const { return r32[f]; }
    // the 2 x 96 order-statistics entries: WebRtcVad_FindMinimum walks and shifts them element by element
    // (vad_sp.c:59-177), a long chain of dependent accesses -- from LDS, not from HBM / L2
    __device__ __forceinline__ int16_t &hm(int f) const { return minbuf[(f - V16_AGE) * 64]; }
};

#ifdef WMX_VAD_PROF  // developer build only (make EXTRA=-DWMX_VAD_PROF): cycles per phase of the kernel, summed over waves
__device__ unsigned long long g_vad_prof[16];
__shared__ long long g_t_prev;
#define VAD_PROF(i)                                                                        \
    do {                                                                                   \
        if (threadIdx.x == 0) {                                                            \
            const long long t_now = clock64();                                             \
            atomicAdd(&g_vad_prof[i], (unsigned long long)(t_now - g_t_prev));             \
            g_t_prev = clock64();                                                          \
        }                                                                                  \
    } while (0)
#else
#define VAD_PROF(i)
#endif

// LDS int16 buffer private to one lane: element i lives at base[i * 64]
struct LaneBuf {
    int16_t *base;
    __device__ __forceinline__ int16_t &operator[](int i) const { return base[i * 64]; }
};

__constant__ int16_t kNoiseW[12] = {34, 62, 72, 66, 53, 25, 94, 66, 56, 62, 75, 103};
__constant__ int16_t kSpeechW[12] = {48, 82, 45, 87, 50, 47, 80, 46, 83, 41, 78, 81};
__constant__ int16_t kSpecW[6] = {6, 8, 10, 12, 14, 16};
__constant__ int16_t kMinDiff[6] = {544, 544, 576, 576, 576, 576};
__constant__ int16_t kMaxSpeech[6] = {11392, 11392, 11520, 11520, 11520, 11520};
__constant__ int16_t kMaxNoise[6] = {9216, 9088, 8960, 8832, 8704, 8576};

// vad_sp.c:27-54: one output sample of the 2:1 all-pass decimator
__device__ __forceinline__ int16_t ds2_step(int16_t a, int16_t b, int32_t &s1, int32_t &s2) {
    const int16_t t1 = (int16_t)((s1 >> 1) + ((5243 * a) >> 14));
    s1 = (int32_t)a - ((5243 * t1) >> 12);
    const int16_t t2 = (int16_t)((s2 >> 1) + ((1392 * b) >> 14));
    s2 = (int32_t)b - ((1392 * t2) >> 12);
    return (int16_t)(t1 + t2);
}

// vad_filterbank.c:83-118: one step of the first-order all-pass (state kept as the Q15 32-bit value)
__device__ __forceinline__ int16_t allpass_step(int16_t x, int16_t coef, int32_t &s32) {
    const int32_t t32 = wadd(s32, coef * x);
    const int16_t t16 = (int16_t)(t32 >> 16);
    s32 = (int32_t)x << 14;
    s32 = wsub(s32, coef * t16);
    s32 = wshl(s32, 1);
    return t16;
}

// vad_filterbank.c:121-145 for a source that is already in LDS
__device__ void split_lds(const LaneBuf in, int len, int16_t &up_state, int16_t &lo_state, LaneBuf hp, LaneBuf lp) {
    const int half = len >> 1;
    int32_t su = wshl(up_state, 16), sl = wshl(lo_state, 16);
    for (int i = 0; i < half; i++) {
        const int16_t h = allpass_step(in[2 * i], 20972, su);
        const int16_t l = allpass_step(in[2 * i + 1], 5571, sl);
        hp[i] = (int16_t)(h - l);
        lp[i] = (int16_t)(l + h);
    }
    up_state = (int16_t)(su >> 16);
    lo_state = (int16_t)(sl >> 16);
}

// vad_filterbank.c:155-243 (+ energy.c:20-39, get_scaling_square.c:20-47)
__device__ void log_energy(const LaneBuf v, int n, int16_t offset, int16_t &total, int16_t &out) {
    const int nbits = size_in_bits((uint32_t)n);
    int16_t smax = -1;
    for (int i = 0; i < n; i++) {
        const int16_t x = v[i];
        const int16_t sabs = (int16_t)(x > 0 ? x : -x);
        if (sabs > smax) smax = sabs;
    }
    const int t = norm_w32((int32_t)smax * smax);
    int rsh = (smax == 0) ? 0 : ((t > nbits) ? 0 : nbits - t);
    uint32_t energy = 0;
    for (int i = 0; i < n; i++) {
        const int32_t x = v[i];
        energy += (uint32_t)((x * x) >> rsh);
    }
    if (energy == 0) {
        out = offset;
        return;
    }
    const int norm = 17 - norm_u32(energy);
    int16_t log2e = 14336;
    rsh += norm;
    if (norm < 0)
        energy <<= -norm;
    else
        energy >>= norm;
    log2e = (int16_t)(log2e + (int16_t)((energy & 0x3FFF) >> 4));
    int16_t le = (int16_t)(((24660 * log2e) >> 19) + ((rsh * 24660) >> 9));
    if (le < 0) le = 0;
    out = (int16_t)(le + offset);
    if (total <= 10) {
        if (rsh >= 0)
            total = (int16_t)(total + 10 + 1);
        else
            total = (int16_t)(total + (int16_t)(energy >> -rsh));
    }
}

// vad_gmm.c:30-83
__device__ int32_t gauss_prob(int16_t input, int16_t mean, int16_t std, int16_t &delta) {
    int16_t exp_value = 0;
    int32_t t32 = (int32_t)131072 + (int32_t)(std >> 1);
    const int16_t inv_std = (int16_t)div_w32_w16(t32, std);
    int16_t t16 = (int16_t)(inv_std >> 2);
    const int16_t inv_std2 = (int16_t)((t16 * t16) >> 2);
    t16 = (int16_t)(input << 3);
    t16 = (int16_t)(t16 - mean);
    delta = (int16_t)((inv_std2 * t16) >> 10);
    t32 = (delta * t16) >> 9;
    if (t32 < 22005) {
        t16 = (int16_t)((5909 * t32) >> 12);
        t16 = (int16_t)-t16;
        exp_value = (int16_t)(0x0400 | (t16 & 0x03FF));
        t16 = (int16_t)(t16 ^ (int16_t)0xFFFF);
        t16 = (int16_t)(t16 >> 10);
        t16 = (int16_t)(t16 + 1);
        exp_value = (int16_t)(exp_value >> t16);
    }
    return inv_std * exp_value;
}

// vad_sp.c:59-177, operating directly on the field-major state rows
__device__ int16_t find_minimum(const VadRef &S, int16_t v, int ch, int32_t frame_counter) {
    const int a0 = V16_AGE + (ch << 4), l0 = V16_LOW + (ch << 4);
#ifdef WMX_VAD_NOFINDMIN  // timing experiment only (wrong results): what the order-statistics walk costs
    return (int16_t)(v + S.h(V16_MEAN_VALUE + ch));
#endif
    for (int i = 0; i < 16; i++) {
        const int16_t age = S.hm(a0 + i);
        if (age != 100) {
            S.hm(a0 + i) = (int16_t)(age + 1);
        } else {
            for (int j = i; j < 15; j++) {
                S.hm(l0 + j) = S.hm(l0 + j + 1);
                S.hm(a0 + j) = S.hm(a0 + j + 1);
            }
            S.hm(a0 + 15) = 101;
            S.hm(l0 + 15) = 10000;
        }
    }
    int pos = -1;
    if (v < S.hm(l0 + 15)) {  // sorted ascending: the reference's unrolled binary search == first larger element
        pos = 0;
        while (!(v < S.hm(l0 + pos))) pos++;
    }
    if (pos > -1) {
        for (int i = 15; i > pos; i--) {
            S.hm(l0 + i) = S.hm(l0 + i - 1);
            S.hm(a0 + i) = S.hm(a0 + i - 1);
        }
        S.hm(l0 + pos) = v;
        S.hm(a0 + pos) = 1;
    }
    int16_t median = 1600, alpha = 0;
    if (frame_counter > 2)
        median = S.hm(l0 + 2);
    else if (frame_counter > 0)
        median = S.hm(l0);
    const int16_t mean = S.h(V16_MEAN_VALUE + ch);
    if (frame_counter > 0) alpha = (median < mean) ? 6553 : 32439;
    int32_t t = (alpha + 1) * mean;
    t += (32767 - alpha) * median;
    t += 16384;
    const int16_t r = (int16_t)(t >> 15);
    S.h(V16_MEAN_VALUE + ch) = r;
    return r;
}

// vad_core.c:108-118 on rows base+c and base+c+6
__device__ __forceinline__ int32_t weighted_avg(const VadRef &S, int base, int c, int16_t offset, const int16_t *w) {
    int32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int16_t d = (int16_t)(S.h(base + c + 6 * k) + offset);
        S.h(base + c + 6 * k) = d;
        acc += d * w[c + 6 * k];
    }
    return acc;
}

// vad_core.c:124-479 with the mode-3 thresholds of vad_core.c:88-91
__device__ int16_t gmm_probability(const VadRef &S, const int16_t *feat, int16_t total_power, int idx) {
    const int16_t oh1 = idx == 0 ? 6 : (idx == 1 ? 3 : 2), oh2 = idx == 0 ? 9 : (idx == 1 ? 5 : 3);
    const int16_t loc = 94, glob = idx == 1 ? 1050 : 1100;
    int16_t vadflag = 0;
    if (total_power > 10) {
        int16_t dN[12], dS[12], ngpr[12], sgpr[12];
#pragma unroll
        for (int i = 0; i < 12; i++) ngpr[i] = sgpr[i] = 0;
        int32_t sum_llr = 0;
#pragma unroll
        for (int c = 0; c < 6; c++) {
            int32_t h0t = 0, h1t = 0, np0 = 0, sp0 = 0;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int g = c + k * 6;
                const int32_t npk = kNoiseW[g] * gauss_prob(feat[c], S.h(V16_NOISE_MEANS + g), S.h(V16_NOISE_STDS + g), dN[g]);
                h0t += npk;
                const int32_t spk = kSpeechW[g] * gauss_prob(feat[c], S.h(V16_SPEECH_MEANS + g), S.h(V16_SPEECH_STDS + g), dS[g]);
                h1t += spk;
                if (k == 0) {
                    np0 = npk;
                    sp0 = spk;
                }
            }
            int16_t sh0 = (int16_t)norm_w32(h0t), sh1 = (int16_t)norm_w32(h1t);
            if (h0t == 0) sh0 = 31;
            if (h1t == 0) sh1 = 31;
            const int16_t llr = (int16_t)(sh0 - sh1);
            sum_llr += (int32_t)(llr * kSpecW[c]);
            if ((llr * 4) > loc) vadflag = 1;
            const int16_t h0 = (int16_t)(h0t >> 12);
            if (h0 > 0) {
                ngpr[c] = (int16_t)div_w32_w16(wshl((int32_t)(np0 & 0xFFFFF000), 2), h0);
                ngpr[c + 6] = (int16_t)(16384 - ngpr[c]);
            } else {
                ngpr[c] = 16384;
            }
            const int16_t h1 = (int16_t)(h1t >> 12);
            if (h1 > 0) {
                sgpr[c] = (int16_t)div_w32_w16(wshl((int32_t)(sp0 & 0xFFFFF000), 2), h1);
                sgpr[c + 6] = (int16_t)(16384 - sgpr[c]);
            }
        }
        vadflag |= (sum_llr >= glob);
        VAD_PROF(4);  // gaussian probabilities
        const int32_t frame_counter = S.w(V32_FRAME_COUNTER);
        int16_t maxspe = 12800;
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const int16_t fmin = find_minimum(S, feat[c], c, frame_counter);
            VAD_PROF(5);  // find_minimum
            int32_t ngm = weighted_avg(S, V16_NOISE_MEANS, c, 0, kNoiseW);
            const int16_t t1 = (int16_t)(ngm >> 6);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int g = c + k * 6;
                const int16_t nmk = S.h(V16_NOISE_MEANS + g), smk = S.h(V16_SPEECH_MEANS + g);
                int16_t nsk = S.h(V16_NOISE_STDS + g), ssk = S.h(V16_SPEECH_STDS + g);
                int16_t nmk2 = nmk, t16;
                if (!vadflag) {
                    const int16_t delt = (int16_t)((ngpr[g] * dN[g]) >> 11);
                    nmk2 = (int16_t)(nmk + (int16_t)((delt * 655) >> 22));
                }
                const int16_t ndelt = (int16_t)((fmin << 4) - t1);
                int16_t nmk3 = (int16_t)(nmk2 + (int16_t)((ndelt * 154) >> 9));
                t16 = (int16_t)((k + 5) << 7);
                if (nmk3 < t16) nmk3 = t16;
                t16 = (int16_t)((72 + k - c) << 7);
                if (nmk3 > t16) nmk3 = t16;
                S.h(V16_NOISE_MEANS + g) = nmk3;
                if (vadflag) {
                    const int16_t delt = (int16_t)((sgpr[g] * dS[g]) >> 11);
                    t16 = (int16_t)((delt * 6554) >> 21);
                    int16_t smk2 = (int16_t)(smk + ((t16 + 1) >> 1));
                    const int16_t maxmu = (int16_t)(maxspe + 640);
                    const int16_t minmean = k == 0 ? 640 : 768;
                    if (smk2 < minmean) smk2 = minmean;
                    if (smk2 > maxmu) smk2 = maxmu;
                    S.h(V16_SPEECH_MEANS + g) = smk2;
                    t16 = (int16_t)((smk + 4) >> 3);
                    t16 = (int16_t)(feat[c] - t16);
                    int32_t a = (dS[g] * t16) >> 3;
                    int32_t b = a - 4096;
                    t16 = (int16_t)(sgpr[g] >> 2);
                    a = t16 * b;
                    b = a >> 4;
                    if (b > 0) {
                        t16 = (int16_t)div_w32_w16(b, (int16_t)(ssk * 10));
                    } else {
                        t16 = (int16_t)div_w32_w16(-b, (int16_t)(ssk * 10));
                        t16 = (int16_t)-t16;
                    }
                    t16 = (int16_t)(t16 + 128);
                    ssk = (int16_t)(ssk + (t16 >> 8));
                    if (ssk < 384) ssk = 384;
                    S.h(V16_SPEECH_STDS + g) = ssk;
                } else {
                    t16 = (int16_t)(feat[c] - (nmk >> 3));
                    int32_t a = (dN[g] * t16) >> 3;
                    a -= 4096;
                    t16 = (int16_t)((ngpr[g] + 2) >> 2);
                    const int32_t b = wmul(t16, a);  // wraps in the reference (vad_core.c:395)
                    a = b >> 14;
                    if (a > 0) {
                        t16 = (int16_t)div_w32_w16(a, nsk);
                    } else {
                        t16 = (int16_t)div_w32_w16(-a, nsk);
                        t16 = (int16_t)-t16;
                    }
                    t16 = (int16_t)(t16 + 32);
                    nsk = (int16_t)(nsk + (t16 >> 6));
                    if (nsk < 384) nsk = 384;
                    S.h(V16_NOISE_STDS + g) = nsk;
                }
            }
            ngm = weighted_avg(S, V16_NOISE_MEANS, c, 0, kNoiseW);
            int32_t sgm = weighted_avg(S, V16_SPEECH_MEANS, c, 0, kSpeechW);
            const int16_t diff = (int16_t)((int16_t)(sgm >> 9) - (int16_t)(ngm >> 9));
            if (diff < kMinDiff[c]) {
                const int16_t t16 = (int16_t)(kMinDiff[c] - diff);
                const int16_t u1 = (int16_t)((13 * t16) >> 2), u2 = (int16_t)((3 * t16) >> 2);
                sgm = weighted_avg(S, V16_SPEECH_MEANS, c, u1, kSpeechW);
                ngm = weighted_avg(S, V16_NOISE_MEANS, c, (int16_t)-u2, kNoiseW);
            }
            maxspe = kMaxSpeech[c];
            int16_t t2 = (int16_t)(sgm >> 7);
            if (t2 > maxspe) {
                t2 = (int16_t)(t2 - maxspe);
#pragma unroll
                for (int k = 0; k < 2; k++) S.h(V16_SPEECH_MEANS + c + 6 * k) = (int16_t)(S.h(V16_SPEECH_MEANS + c + 6 * k) - t2);
            }
            t2 = (int16_t)(ngm >> 7);
            if (t2 > kMaxNoise[c]) {
                t2 = (int16_t)(t2 - kMaxNoise[c]);
#pragma unroll
                for (int k = 0; k < 2; k++) S.h(V16_NOISE_MEANS + c + 6 * k) = (int16_t)(S.h(V16_NOISE_MEANS + c + 6 * k) - t2);
            }
            VAD_PROF(6);  // model update
        }
        S.w(V32_FRAME_COUNTER) = frame_counter + 1;
    }
    int16_t over_hang = S.h(V16_OVER_HANG), num_speech = S.h(V16_NUM_SPEECH);
    if (!vadflag) {
        if (over_hang > 0) {
            vadflag = (int16_t)(2 + over_hang);
            over_hang--;
        }
        num_speech = 0;
    } else {
        num_speech++;
        if (num_speech > 6) {
            num_speech = 6;
            over_hang = oh2;
        } else {
            over_hang = oh1;
        }
    }
    S.h(V16_OVER_HANG) = over_hang;
    S.h(V16_NUM_SPEECH) = num_speech;
    return vadflag;
}

// WebRtcVad_Process (webrtc_vad.c:71-104) on one packet of NB*RATIO samples at fs = 8000*RATIO.
// RATIO = 1, 2, 4 (vad_core.c:623-674: one or two chained 2:1 decimators in front of the 8 kHz core).
// sample sources of vad_packet: the packet in memory, or the packet already in registers (8 samples per uint4; the
// index is a compile-time constant there because the decimation loop is fully unrolled for it)
struct MemSrc {
    const int16_t *p;
    static constexpr bool kUnroll = false;
    __device__ __forceinline__ int16_t operator()(int i) const { return p[i]; }
};
template <int NV>
struct RegSrc {
    const uint4 *raw;
    static constexpr bool kUnroll = true;
    __device__ __forceinline__ int16_t operator()(int i) const {
        const uint4 v = raw[i >> 3];
        const unsigned w = ((i >> 1) & 3) == 0 ? v.x : (((i >> 1) & 3) == 1 ? v.y : (((i >> 1) & 3) == 2 ? v.z : v.w));
        return (int16_t)((i & 1) ? (w >> 16) : (w & 0xffffu));
    }
};

template <int NB>
__device__ __forceinline__ void vad_features_rest(const VadRef &S, LaneBuf hp120, LaneBuf lp120, LaneBuf hp60, LaneBuf lp60,
                                                  int16_t (&feat)[6], int16_t &total);

// WebRtcVad_CalculateFeatures on one packet: the six band log-energies and the total power indicator
template <int NB, int RATIO, class Src>
__device__ __forceinline__ void vad_features(const VadRef &S, const Src p, LaneBuf hp120, LaneBuf lp120, LaneBuf hp60, LaneBuf lp60,
                                             int16_t (&feat)[6], int16_t &total) {
    // ---- decimation to 8 kHz fused with the first band split (vad_filterbank.c:268-270)
    {
        int32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        if (RATIO >= 2) {
            d0 = S.w(V32_DS + 0);
            d1 = S.w(V32_DS + 1);
        }
        if (RATIO == 4) {
            d2 = S.w(V32_DS + 2);
            d3 = S.w(V32_DS + 3);
        }
        int16_t up = S.h(V16_UPPER + 0), lo = S.h(V16_LOWER + 0);
        int32_t su = wshl(up, 16), sl = wshl(lo, 16);
        constexpr int kUn = Src::kUnroll ? NB / 2 : 1;
#pragma unroll kUn
        for (int i = 0; i < NB / 2; i++) {
            int16_t nb[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int q = (2 * i + e) * RATIO;
                if (RATIO == 1) {
                    nb[e] = p(q);
                } else if (RATIO == 2) {
                    nb[e] = ds2_step(p(q), p(q + 1), d0, d1);
                } else {
                    const int16_t w0 = ds2_step(p(q), p(q + 1), d2, d3);
                    const int16_t w1 = ds2_step(p(q + 2), p(q + 3), d2, d3);
                    nb[e] = ds2_step(w0, w1, d0, d1);
                }
            }
            const int16_t h = allpass_step(nb[0], 20972, su);
            const int16_t l = allpass_step(nb[1], 5571, sl);
            hp120[i] = (int16_t)(h - l);
            lp120[i] = (int16_t)(l + h);
        }
        S.h(V16_UPPER + 0) = (int16_t)(su >> 16);
        S.h(V16_LOWER + 0) = (int16_t)(sl >> 16);
        if (RATIO >= 2) {
            S.w(V32_DS + 0) = d0;
            S.w(V32_DS + 1) = d1;
        }
        if (RATIO == 4) {
            S.w(V32_DS + 2) = d2;
            S.w(V32_DS + 3) = d3;
        }
    }
    VAD_PROF(2);  // decimation + first split
    vad_features_rest<NB>(S, hp120, lp120, hp60, lp60, feat, total);
}

// ---- rest of WebRtcVad_CalculateFeatures (vad_filterbank.c:272-332) behind the first band split
template <int NB>
__device__ __forceinline__ void vad_features_rest(const VadRef &S, LaneBuf hp120, LaneBuf lp120, LaneBuf hp60, LaneBuf lp60,
                                                  int16_t (&feat)[6], int16_t &total) {
    total = 0;
    auto do_split = [&](LaneBuf in, int len, int band, LaneBuf hp, LaneBuf lp) {
        int16_t up = S.h(V16_UPPER + band), lo = S.h(V16_LOWER + band);
        split_lds(in, len, up, lo, hp, lp);
        S.h(V16_UPPER + band) = up;
        S.h(V16_LOWER + band) = lo;
    };
    do_split(hp120, NB / 2, 1, hp60, lp60);
    log_energy(hp60, NB / 4, 176, total, feat[5]);
    log_energy(lp60, NB / 4, 176, total, feat[4]);
    do_split(lp120, NB / 2, 2, hp60, lp60);
    log_energy(hp60, NB / 4, 176, total, feat[3]);
    do_split(lp60, NB / 4, 3, hp120, lp120);
    log_energy(hp120, NB / 8, 272, total, feat[2]);
    do_split(lp120, NB / 8, 4, hp60, lp60);
    log_energy(hp60, NB / 16, 368, total, feat[1]);
    {
        // HighPassFilter vad_filterbank.c:41-80
        int16_t s0 = S.h(V16_HP + 0), s1 = S.h(V16_HP + 1), s2 = S.h(V16_HP + 2), s3 = S.h(V16_HP + 3);
        for (int i = 0; i < NB / 16; i++) {
            const int16_t x = lp60[i];
            int32_t t = 6631 * x;
            t += -13262 * s0;
            t += 6631 * s1;
            s1 = s0;
            s0 = x;
            t -= -7756 * s2;
            t -= 5620 * s3;
            s3 = s2;
            s2 = (int16_t)(t >> 14);
            hp120[i] = s2;
        }
        S.h(V16_HP + 0) = s0;
        S.h(V16_HP + 1) = s1;
        S.h(V16_HP + 2) = s2;
        S.h(V16_HP + 3) = s3;
    }
    log_energy(hp120, NB / 16, 368, total, feat[0]);
    VAD_PROF(3);  // remaining splits + log energies
}

template <int NB, int RATIO, class Src>
__device__ __forceinline__ int vad_packet(const VadRef &S, const Src p, LaneBuf hp120, LaneBuf lp120, LaneBuf hp60, LaneBuf lp60) {
    int16_t feat[6], total;
    vad_features<NB, RATIO>(S, p, hp120, lp120, hp60, lp60, feat, total);
    const int v = gmm_probability(S, feat, total, NB == 80 ? 0 : (NB == 160 ? 1 : 2));
    return v > 0 ? 1 : v;
}

// ================================================================== the 10 ms mono packet as a four-wave pipeline
// vad_kernel above runs a stream in one lane from end to end: 65 536 streams are 1 024 waves, one per SIMD, and the launch
// lasts as long as one wave's chain of ~9 000 dependent-issue instructions -- 60 % of it the six channels of the GMM
// (probabilities, order statistics, model update), which do not depend on each other.  Here a workgroup still owns 64
// streams (lane = stream), but as four waves: wave 3 runs the filter bank and the packet, waves 0..2 two GMM channels each
// (their state rows are requested while wave 3 is still filtering).  Features and the per-channel log-likelihood ratios
// cross through LDS at two workgroup barriers; every integer operation of a stream is the one vad_kernel performs.
struct VadChan {
    int16_t dN[2], dS[2], ngpr[2], sgpr[2];
};
// first channel loop of gmm_probability (vad_core.c:152-232) for channel C
template <int C>
__device__ __forceinline__ void gmm_prob_channel(const VadRef &S, int16_t feat_c, VadChan &R, int32_t &llr_w, int &flag) {
    R.ngpr[0] = R.ngpr[1] = R.sgpr[0] = R.sgpr[1] = 0;
    int32_t h0t = 0, h1t = 0, np0 = 0, sp0 = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int g = C + k * 6;
        const int32_t npk = kNoiseW[g] * gauss_prob(feat_c, S.h(V16_NOISE_MEANS + g), S.h(V16_NOISE_STDS + g), R.dN[k]);
        h0t += npk;
        const int32_t spk = kSpeechW[g] * gauss_prob(feat_c, S.h(V16_SPEECH_MEANS + g), S.h(V16_SPEECH_STDS + g), R.dS[k]);
        h1t += spk;
        if (k == 0) {
            np0 = npk;
            sp0 = spk;
        }
    }
    int16_t sh0 = (int16_t)norm_w32(h0t), sh1 = (int16_t)norm_w32(h1t);
    if (h0t == 0) sh0 = 31;
    if (h1t == 0) sh1 = 31;
    const int16_t llr = (int16_t)(sh0 - sh1);
    llr_w = (int32_t)(llr * kSpecW[C]);
    flag = (llr * 4) > 94;
    const int16_t h0 = (int16_t)(h0t >> 12);
    if (h0 > 0) {
        R.ngpr[0] = (int16_t)div_w32_w16(wshl((int32_t)(np0 & 0xFFFFF000), 2), h0);
        R.ngpr[1] = (int16_t)(16384 - R.ngpr[0]);
    } else {
        R.ngpr[0] = 16384;
    }
    const int16_t h1 = (int16_t)(h1t >> 12);
    if (h1 > 0) {
        R.sgpr[0] = (int16_t)div_w32_w16(wshl((int32_t)(sp0 & 0xFFFFF000), 2), h1);
        R.sgpr[1] = (int16_t)(16384 - R.sgpr[0]);
    }
}
// second channel loop of gmm_probability (vad_core.c:240-440) for channel C
template <int C>
__device__ __forceinline__ void gmm_update_channel(const VadRef &S, int16_t feat_c, int vadflag, int32_t frame_counter, const VadChan &R) {
    const int16_t maxspe_in = C == 0 ? (int16_t)12800 : kMaxSpeech[C == 0 ? 0 : C - 1];
    const int16_t fmin = find_minimum(S, feat_c, C, frame_counter);
    int32_t ngm = weighted_avg(S, V16_NOISE_MEANS, C, 0, kNoiseW);
    const int16_t t1 = (int16_t)(ngm >> 6);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int g = C + k * 6;
        const int16_t nmk = S.h(V16_NOISE_MEANS + g), smk = S.h(V16_SPEECH_MEANS + g);
        int16_t nsk = S.h(V16_NOISE_STDS + g), ssk = S.h(V16_SPEECH_STDS + g);
        int16_t nmk2 = nmk, t16;
        if (!vadflag) {
            const int16_t delt = (int16_t)((R.ngpr[k] * R.dN[k]) >> 11);
            nmk2 = (int16_t)(nmk + (int16_t)((delt * 655) >> 22));
        }
        const int16_t ndelt = (int16_t)((fmin << 4) - t1);
        int16_t nmk3 = (int16_t)(nmk2 + (int16_t)((ndelt * 154) >> 9));
        t16 = (int16_t)((k + 5) << 7);
        if (nmk3 < t16) nmk3 = t16;
        t16 = (int16_t)((72 + k - C) << 7);
        if (nmk3 > t16) nmk3 = t16;
        S.h(V16_NOISE_MEANS + g) = nmk3;
        if (vadflag) {
            const int16_t delt = (int16_t)((R.sgpr[k] * R.dS[k]) >> 11);
            t16 = (int16_t)((delt * 6554) >> 21);
            int16_t smk2 = (int16_t)(smk + ((t16 + 1) >> 1));
            const int16_t maxmu = (int16_t)(maxspe_in + 640);
            const int16_t minmean = k == 0 ? 640 : 768;
            if (smk2 < minmean) smk2 = minmean;
            if (smk2 > maxmu) smk2 = maxmu;
            S.h(V16_SPEECH_MEANS + g) = smk2;
            t16 = (int16_t)((smk + 4) >> 3);
            t16 = (int16_t)(feat_c - t16);
            int32_t a = (R.dS[k] * t16) >> 3;
            int32_t b = a - 4096;
            t16 = (int16_t)(R.sgpr[k] >> 2);
            a = t16 * b;
            b = a >> 4;
            if (b > 0) {
                t16 = (int16_t)div_w32_w16(b, (int16_t)(ssk * 10));
            } else {
                t16 = (int16_t)div_w32_w16(-b, (int16_t)(ssk * 10));
                t16 = (int16_t)-t16;
            }
            t16 = (int16_t)(t16 + 128);
            ssk = (int16_t)(ssk + (t16 >> 8));
            if (ssk < 384) ssk = 384;
            S.h(V16_SPEECH_STDS + g) = ssk;
        } else {
            t16 = (int16_t)(feat_c - (nmk >> 3));
            int32_t a = (R.dN[k] * t16) >> 3;
            a -= 4096;
            t16 = (int16_t)((R.ngpr[k] + 2) >> 2);
            const int32_t b = wmul(t16, a);  // wraps in the reference (vad_core.c:395)
            a = b >> 14;
            if (a > 0) {
                t16 = (int16_t)div_w32_w16(a, nsk);
            } else {
                t16 = (int16_t)div_w32_w16(-a, nsk);
                t16 = (int16_t)-t16;
            }
            t16 = (int16_t)(t16 + 32);
            nsk = (int16_t)(nsk + (t16 >> 6));
            if (nsk < 384) nsk = 384;
            S.h(V16_NOISE_STDS + g) = nsk;
        }
    }
    ngm = weighted_avg(S, V16_NOISE_MEANS, C, 0, kNoiseW);
    int32_t sgm = weighted_avg(S, V16_SPEECH_MEANS, C, 0, kSpeechW);
    const int16_t diff = (int16_t)((int16_t)(sgm >> 9) - (int16_t)(ngm >> 9));
    if (diff < kMinDiff[C]) {
        const int16_t t16 = (int16_t)(kMinDiff[C] - diff);
        const int16_t u1 = (int16_t)((13 * t16) >> 2), u2 = (int16_t)((3 * t16) >> 2);
        sgm = weighted_avg(S, V16_SPEECH_MEANS, C, u1, kSpeechW);
        ngm = weighted_avg(S, V16_NOISE_MEANS, C, (int16_t)-u2, kNoiseW);
    }
    const int16_t maxspe = kMaxSpeech[C];
    int16_t t2 = (int16_t)(sgm >> 7);
    if (t2 > maxspe) {
        t2 = (int16_t)(t2 - maxspe);
#pragma unroll
        for (int k = 0; k < 2; k++) S.h(V16_SPEECH_MEANS + C + 6 * k) = (int16_t)(S.h(V16_SPEECH_MEANS + C + 6 * k) - t2);
    }
    t2 = (int16_t)(ngm >> 7);
    if (t2 > kMaxNoise[C]) {
        t2 = (int16_t)(t2 - kMaxNoise[C]);
#pragma unroll
        for (int k = 0; k < 2; k++) S.h(V16_NOISE_MEANS + C + 6 * k) = (int16_t)(S.h(V16_NOISE_MEANS + C + 6 * k) - t2);
    }
}

// mono sample i of a packet that lies in registers as uint4's: CHN = 1 sample i itself (RegSrc); CHN = 2 the interleaved frame i
// = one 32-bit word (left | right << 16) averaged the way vad_process's in-place down-mix does it (src/webrtc.c:104-116:
// (L + R) / 2 in int32, truncated towards zero)
template <int CHN>
struct FrameSrc {
    const uint4 *raw;
    __device__ __forceinline__ int16_t operator()(int i) const {
        if constexpr (CHN == 1) {
            const uint4 v = raw[i >> 3];
            const unsigned w = ((i >> 1) & 3) == 0 ? v.x : (((i >> 1) & 3) == 1 ? v.y : (((i >> 1) & 3) == 2 ? v.z : v.w));
            return (int16_t)((i & 1) ? (w >> 16) : (w & 0xffffu));
        } else {
            const uint4 v = raw[i >> 2];
            const unsigned w = (i & 3) == 0 ? v.x : ((i & 3) == 1 ? v.y : ((i & 3) == 2 ? v.z : v.w));
            const int32_t acc = (int32_t)(int16_t)(w & 0xffffu) + (int32_t)(int16_t)(w >> 16);
            return (int16_t)(acc / 2);
        }
    }
};

// Decimation + first band split of vad_features for the pipeline's filter-bank wave, with the packet passing through the
// registers a few output samples at a time (the next block's uint4's are requested while the current one is filtered) instead
// of the whole packet held at once: the kernel's register count decides how many workgroups share a CU.  RATIO = 1, 2, 4: 8, 16,
// 32 kHz (one or two chained 2:1 decimators in front of the 8 kHz core, vad_core.c:623-674); CHN = 1, 2 interleaved channels.
template <int RATIO, int NB, int CHN>
__device__ __forceinline__ void vad_front_blocked(const VadRef &S, const uint4 *frame4, LaneBuf hp120, LaneBuf lp120) {
    static_assert(RATIO == 1 || RATIO == 2 || RATIO == 4, "8, 16 or 32 kHz");
    // IT pairs of 8 kHz samples per block = 2 IT RATIO CHN int16 of input: at most four uint4 per block in flight twice
    constexpr int IT = RATIO * CHN <= 2 ? 8 : 16 / (RATIO * CHN), VPB = IT * 2 * RATIO * CHN / 8, NBLK = NB / 2 / IT;
    static_assert(VPB >= 1 && VPB <= 4 && NBLK * IT * 2 == NB, "block geometry");
    int32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    if (RATIO >= 2) {
        d0 = S.w(V32_DS + 0);
        d1 = S.w(V32_DS + 1);
    }
    if (RATIO == 4) {
        d2 = S.w(V32_DS + 2);
        d3 = S.w(V32_DS + 3);
    }
    int32_t su = wshl(S.h(V16_UPPER + 0), 16), sl = wshl(S.h(V16_LOWER + 0), 16);
    uint4 cur[VPB], nxt[VPB];
#pragma unroll
    for (int j = 0; j < VPB; j++) cur[j] = frame4[j];
#pragma unroll 1
    for (int blk = 0; blk < NBLK; blk++) {
        const int nb = blk + 1 < NBLK ? blk + 1 : blk;  // the last block re-requests itself (no branch around the loads)
#pragma unroll
        for (int j = 0; j < VPB; j++) nxt[j] = frame4[nb * VPB + j];
        const FrameSrc<CHN> p{cur};
#pragma unroll
        for (int it = 0; it < IT; it++) {
            int16_t s8[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int q = (2 * it + e) * RATIO;
                if (RATIO == 1) {
                    s8[e] = p(q);
                } else if (RATIO == 2) {
                    s8[e] = ds2_step(p(q), p(q + 1), d0, d1);
                } else {
                    const int16_t w0 = ds2_step(p(q), p(q + 1), d2, d3);
                    const int16_t w1 = ds2_step(p(q + 2), p(q + 3), d2, d3);
                    s8[e] = ds2_step(w0, w1, d0, d1);
                }
            }
            const int16_t h = allpass_step(s8[0], 20972, su);
            const int16_t l = allpass_step(s8[1], 5571, sl);
            hp120[blk * IT + it] = (int16_t)(h - l);
            lp120[blk * IT + it] = (int16_t)(l + h);
        }
#pragma unroll
        for (int j = 0; j < VPB; j++) cur[j] = nxt[j];
    }
    S.h(V16_UPPER + 0) = (int16_t)(su >> 16);
    S.h(V16_LOWER + 0) = (int16_t)(sl >> 16);
    if (RATIO >= 2) {
        S.w(V32_DS + 0) = d0;
        S.w(V32_DS + 1) = d1;
    }
    if (RATIO == 4) {
        S.w(V32_DS + 2) = d2;
        S.w(V32_DS + 3) = d3;
    }
}

// exchange area (int32 [field][lane], aliases the band buffers, which are dead between the filter bank and the next packet)
enum : int { X_FEAT = 0, X_TOTAL = 6, X_LLR = 7, X_FLAG = 13, X_FIELDS = 19 };

// one pair of GMM channels (C0, C0 + 1) of the workgroup's 64 streams: state in, per packet probabilities -> barrier ->
// decision + update, state out
// GLOB: the mode-3 total threshold of the packet length (vad_core.c:88-91: 1100 for 10 ms frames, 1050 for 20 ms)
template <int C0, int GLOB>
__device__ __forceinline__ void vad_pipe_channels(const VadRef &S, int16_t *s16, int32_t *s32, int32_t *xch, int16_t *minlds, int lane,
                                                  int stream, bool live, int n_streams, int n_calls) {
    const int16_t *g16 = s16 + stream;
    // this wave's state rows, all requested before the first one is used
#pragma unroll
    for (int j = 0; j < 2; j++) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int g = C0 + j + 6 * k;
            S.h(V16_NOISE_MEANS + g) = g16[(size_t)(V16_NOISE_MEANS + g) * n_streams];
            S.h(V16_SPEECH_MEANS + g) = g16[(size_t)(V16_SPEECH_MEANS + g) * n_streams];
            S.h(V16_NOISE_STDS + g) = g16[(size_t)(V16_NOISE_STDS + g) * n_streams];
            S.h(V16_SPEECH_STDS + g) = g16[(size_t)(V16_SPEECH_STDS + g) * n_streams];
        }
        S.h(V16_MEAN_VALUE + C0 + j) = g16[(size_t)(V16_MEAN_VALUE + C0 + j) * n_streams];
    }
    S.w(V32_FRAME_COUNTER) = s32[(size_t)V32_FRAME_COUNTER * n_streams + stream];
    // index_vector / low_value_vector rows of the two channels (2 x 16 each)
#pragma unroll 8
    for (int i = 0; i < 32; i++) {
        const int fa = V16_AGE + 16 * C0 + i, fl = V16_LOW + 16 * C0 + i;
        minlds[(fa - V16_AGE) * 64 + lane] = g16[(size_t)fa * n_streams];
        minlds[(fl - V16_AGE) * 64 + lane] = g16[(size_t)fl * n_streams];
    }
    for (int call = 0; call < n_calls; call++) {
        __syncthreads();  // 1: the packet's features are in xch
        const int16_t total = (int16_t)xch[X_TOTAL * 64 + lane];
        const int16_t f0 = (int16_t)xch[(X_FEAT + C0) * 64 + lane], f1 = (int16_t)xch[(X_FEAT + C0 + 1) * 64 + lane];
        VadChan R0, R1;
        int32_t l0 = 0, l1 = 0;
        int g0 = 0, g1 = 0;
        if (total > 10) {
            gmm_prob_channel<C0>(S, f0, R0, l0, g0);
            gmm_prob_channel<C0 + 1>(S, f1, R1, l1, g1);
        }
        xch[(X_LLR + C0) * 64 + lane] = l0;
        xch[(X_LLR + C0 + 1) * 64 + lane] = l1;
        xch[(X_FLAG + C0) * 64 + lane] = g0;
        xch[(X_FLAG + C0 + 1) * 64 + lane] = g1;
        __syncthreads();  // 2: every channel's log-likelihood ratio is in xch
        if (total > 10) {
            int32_t sum_llr = 0;
            int vadflag = 0;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                sum_llr += xch[(X_LLR + c) * 64 + lane];
                vadflag |= xch[(X_FLAG + c) * 64 + lane];
            }
            vadflag |= (sum_llr >= GLOB);
            const int32_t frame_counter = S.w(V32_FRAME_COUNTER);
            gmm_update_channel<C0>(S, f0, vadflag, frame_counter, R0);
            gmm_update_channel<C0 + 1>(S, f1, vadflag, frame_counter, R1);
            S.w(V32_FRAME_COUNTER) = frame_counter + 1;
        }
        __syncthreads();  // 3: xch may be overwritten by the next packet's filter bank
    }
    if (!live) return;
    int16_t *o16 = s16 + stream;
#pragma unroll
    for (int j = 0; j < 2; j++) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int g = C0 + j + 6 * k;
            o16[(size_t)(V16_NOISE_MEANS + g) * n_streams] = S.h(V16_NOISE_MEANS + g);
            o16[(size_t)(V16_SPEECH_MEANS + g) * n_streams] = S.h(V16_SPEECH_MEANS + g);
            o16[(size_t)(V16_NOISE_STDS + g) * n_streams] = S.h(V16_NOISE_STDS + g);
            o16[(size_t)(V16_SPEECH_STDS + g) * n_streams] = S.h(V16_SPEECH_STDS + g);
        }
        o16[(size_t)(V16_MEAN_VALUE + C0 + j) * n_streams] = S.h(V16_MEAN_VALUE + C0 + j);
    }
    if (C0 == 0) s32[(size_t)V32_FRAME_COUNTER * n_streams + stream] = S.w(V32_FRAME_COUNTER);
#pragma unroll 8
    for (int i = 0; i < 32; i++) {
        const int fa = V16_AGE + 16 * C0 + i, fl = V16_LOW + 16 * C0 + i;
        o16[(size_t)fa * n_streams] = minlds[(fa - V16_AGE) * 64 + lane];
        o16[(size_t)fl * n_streams] = minlds[(fl - V16_AGE) * 64 + lane];
    }
}

// RATIO = 1 (8 kHz), 2 (16 kHz) or 4 (32 kHz); NB = 80: 10 ms packets, 160: 20 ms packets (what vad_init makes of the daemon's
// WMIX_INTERVAL_MS = 20 up to 16 kHz, src/webrtc.c:57-66, src/wmixConf.h:112); CHN = 1, or 2 interleaved channels -- every shape
// vad_init accepts for the platforms' formats (src/webrtc.c:40-82).  With CHN = 2 the call is vad_process's down-mix / analyse /
// attenuate / re-expand on ONE packet per call (src/webrtc.c:104-150): the analysed samples are (L + R) / 2 and BOTH channels of
// the packet come back as the attenuated mean.  16-byte aligned rows (what wmx_vad_process checks).
template <int RATIO, int NB, int CHN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NB == 80 ? 4 : 3, NB == 80 ? 4 : 3))) void vad_pipe_kernel(int16_t *s16, int32_t *s32, int16_t *pcm, int n_streams, int packets_per_call,
                                                       int n_calls, long stream_stride, long call_stride, const uint8_t *__restrict__ active) {
    constexpr int PKG = NB * RATIO, NV = PKG * CHN / 8;
    // mode-3 thresholds and hangover lengths of the frame length (vad_core.c:88-91; gmm_probability above)
    constexpr int GLOB = NB == 80 ? 1100 : 1050, OH1 = NB == 80 ? 6 : 3, OH2 = NB == 80 ? 9 : 5;
    // band buffers of the filter-bank wave: hp120 | lp120 | lp60, with hp60 IN PLACE over hp120 -- a split writes output i from inputs
    // 2 i and 2 i + 1, so its high-pass half can take the place of its own input, and by the time the later splits write into
    // these regions what they held has been turned into its log-energy (vad_features_rest's order).  1.25 NB instead of 1.5 NB
    // samples per stream: the 20 ms kernel fits three workgroups per CU instead of two (50.2 KB with the order statistics).
    __shared__ __attribute__((aligned(16))) int16_t lds[64 * (NB / 2 + NB / 2 + NB / 4)];
    __shared__ int16_t minlds[64 * kVadMinFields];
    int32_t *xch = reinterpret_cast<int32_t *>(lds);
    static_assert(X_FIELDS * 64 * 4 <= (int)sizeof(lds), "exchange area must fit in the band buffers");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int stream_raw = (int)blockIdx.x * 64 + lane;
    // a lane without a stream recomputes the last one, a lane whose stream is switched off its own, and neither stores anything
    const bool live = stream_active(active, stream_raw, n_streams);
    const int stream = stream_raw < n_streams ? stream_raw : n_streams - 1;
    int16_t r16[kVadRegFields];
    int32_t r32[V32_WORDS];
    const VadRef S{r16, r32, minlds + lane};
    if (wave == 0) {
        vad_pipe_channels<0, GLOB>(S, s16, s32, xch, minlds, lane, stream, live, n_streams, n_calls * packets_per_call);
        return;
    }
    if (wave == 1) {
        vad_pipe_channels<2, GLOB>(S, s16, s32, xch, minlds, lane, stream, live, n_streams, n_calls * packets_per_call);
        return;
    }
    if (wave == 2) {
        vad_pipe_channels<4, GLOB>(S, s16, s32, xch, minlds, lane, stream, live, n_streams, n_calls * packets_per_call);
        return;
    }
    // ---- wave 3: filter bank, hangover, attenuation
    const int16_t *g16 = s16 + stream;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        S.h(V16_UPPER + i) = g16[(size_t)(V16_UPPER + i) * n_streams];
        S.h(V16_LOWER + i) = g16[(size_t)(V16_LOWER + i) * n_streams];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) S.h(V16_HP + i) = g16[(size_t)(V16_HP + i) * n_streams];
    S.h(V16_REDUCE) = g16[(size_t)V16_REDUCE * n_streams];
    S.h(V16_OVER_HANG) = g16[(size_t)V16_OVER_HANG * n_streams];
    S.h(V16_NUM_SPEECH) = g16[(size_t)V16_NUM_SPEECH * n_streams];
#pragma unroll
    for (int i = 0; i < 4; i++) S.w(V32_DS + i) = s32[(size_t)(V32_DS + i) * n_streams + stream];
    const LaneBuf hp120{lds + lane}, lp120{lds + lane + 64 * (NB / 2)}, hp60{lds + lane}, lp60{lds + lane + 64 * NB};
    // A call of several packets analyses its FIRST packet once per packet (the wrapper never advances its pointer)
    // and attenuates that packet after the first analysis only: the later analyses see the attenuated samples.
    for (int pass = 0; pass < n_calls * packets_per_call; pass++) {
        const int call = pass / packets_per_call, it = pass - call * packets_per_call;
        uint4 *frame4 = reinterpret_cast<uint4 *>(pcm + (size_t)stream * stream_stride + (size_t)call * call_stride);
        int16_t feat[6], total;
        vad_front_blocked<RATIO, NB, CHN>(S, frame4, hp120, lp120);
        vad_features_rest<NB>(S, hp120, lp120, hp60, lp60, feat, total);
        // the band buffers are read for the last time above and xch lives in the same bytes: keep the compiler from moving
        // the stores below in front of those reads (the hardware executes a wave's LDS instructions in order)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int c = 0; c < 6; c++) xch[(X_FEAT + c) * 64 + lane] = feat[c];
        xch[X_TOTAL * 64 + lane] = total;
        __syncthreads();  // 1
        __syncthreads();  // 2
        int vadflag = 0;
        if (total > 10) {
            int32_t sum_llr = 0;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                sum_llr += xch[(X_LLR + c) * 64 + lane];
                vadflag |= xch[(X_FLAG + c) * 64 + lane];
            }
            vadflag |= (sum_llr >= GLOB);
        }
        // hangover smoothing (vad_core.c:443-468), mode-3 values of the frame length
        int16_t over_hang = S.h(V16_OVER_HANG), num_speech = S.h(V16_NUM_SPEECH);
        if (!vadflag) {
            if (over_hang > 0) {
                vadflag = 2 + over_hang;
                over_hang--;
            }
            num_speech = 0;
        } else {
            num_speech++;
            if (num_speech > 6) {
                num_speech = 6;
                over_hang = OH2;
            } else {
                over_hang = OH1;
            }
        }
        S.h(V16_OVER_HANG) = over_hang;
        S.h(V16_NUM_SPEECH) = num_speech;
        int reduce = S.h(V16_REDUCE);
        if (vadflag == 0) {
            if (reduce < 4) reduce += 1;
        } else {
            if (reduce > 0) reduce -= 1;
        }
        S.h(V16_REDUCE) = (int16_t)reduce;
        auto att = [&](unsigned w) {  // both int16 halves >> reduce (arithmetic)
            const int lo = (int)(int16_t)(w & 0xffffu) >> reduce, hi = (int)(int16_t)(w >> 16) >> reduce;
            return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16);
        };
        // mono: >> 0 leaves the packet as it is, and vad_process works in place: nothing to fetch or store.  Two channels: the packet
        // always comes back as the mean on both channels (src/webrtc.c:145-150), attenuated or not.
        if (live && it == 0 && (CHN == 2 || reduce != 0)) {
            // the packet is fetched a second time (L2) rather than held in registers across the two barriers: the kernel's
            // register count decides how many workgroups share a CU
            constexpr int NVC = NV <= 20 ? NV : 5;  // a packet of up to 20 uint4 in one piece (80 registers, nothing else is live here); longer ones five at a time
            static_assert(NV % NVC == 0, "whole chunks");
            auto dup = [&](unsigned w) {  // (L + R) / 2 >> reduce on both channels
                const int m = ((int)(int16_t)(w & 0xffffu) + (int)(int16_t)(w >> 16)) / 2;
                const unsigned v = (unsigned)((int)(int16_t)m >> reduce) & 0xffffu;
                return v | (v << 16);
            };
#pragma unroll 1
            for (int j0 = 0; j0 < NV; j0 += NVC) {
                uint4 again[NVC];
#pragma unroll
                for (int j = 0; j < NVC; j++) again[j] = frame4[j0 + j];
#pragma unroll
                for (int j = 0; j < NVC; j++)
                    frame4[j0 + j] = CHN == 2 ? make_uint4(dup(again[j].x), dup(again[j].y), dup(again[j].z), dup(again[j].w))
                                              : make_uint4(att(again[j].x), att(again[j].y), att(again[j].z), att(again[j].w));
            }
            // the next analysis of this call reads what this very lane has just stored: the stores must have left the wave, nothing more
            // (a workgroup-scope fence; the agent-scope __threadfence() that stood here is a cache write-back per packet and workgroup)
            if (packets_per_call > 1) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
        __syncthreads();  // 3
    }
    if (!live) return;
    int16_t *o16 = s16 + stream;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        o16[(size_t)(V16_UPPER + i) * n_streams] = S.h(V16_UPPER + i);
        o16[(size_t)(V16_LOWER + i) * n_streams] = S.h(V16_LOWER + i);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) o16[(size_t)(V16_HP + i) * n_streams] = S.h(V16_HP + i);
    o16[(size_t)V16_REDUCE * n_streams] = S.h(V16_REDUCE);
    o16[(size_t)V16_OVER_HANG * n_streams] = S.h(V16_OVER_HANG);
    o16[(size_t)V16_NUM_SPEECH * n_streams] = S.h(V16_NUM_SPEECH);
#pragma unroll
    for (int i = 0; i < 4; i++) s32[(size_t)(V32_DS + i) * n_streams + stream] = S.w(V32_DS + i);
}

template <int NB, int RATIO>
__global__ __launch_bounds__(64) void vad_kernel(int16_t *s16, int32_t *s32, int16_t *pcm, int n_streams, int packets_per_call,
                                                 int n_calls, long stream_stride, long call_stride, int chn, const uint8_t *__restrict__ active) {
    __shared__ int16_t lds[64 * (NB / 2 + NB / 2 + NB / 4 + NB / 4)];
    __shared__ int16_t minlds[64 * kVadMinFields];
    const int lane = threadIdx.x;
    const int stream = blockIdx.x * 64 + lane;
    if (!stream_active(active, stream, n_streams)) return;  // lanes are independent: no barriers anywhere in this kernel
    int16_t r16[kVadRegFields];
    int32_t r32[V32_WORDS];
    const VadRef S{r16, r32, minlds + lane};
    const int16_t *g16 = s16 + stream;
    const int32_t *g32 = s32 + stream;
#ifdef WMX_VAD_PROF
    if (threadIdx.x == 0) g_t_prev = clock64();
#endif
    // state in: every row requested before the first one is used (coalesced: field-major rows)
#pragma unroll
    for (int f = 0; f < V16_AGE; f++) r16[f] = g16[(size_t)f * n_streams];
#pragma unroll
    for (int f = V16_MEAN_VALUE; f < V16_WORDS; f++) r16[f - kVadMinFields] = g16[(size_t)f * n_streams];
#pragma unroll
    for (int f = 0; f < V32_WORDS; f++) r32[f] = g32[(size_t)f * n_streams];
#pragma unroll 8
    for (int f = 0; f < kVadMinFields; f++) minlds[f * 64 + lane] = g16[(size_t)(V16_AGE + f) * n_streams];
    // warm L2 with this wave's PCM lines (wmx_internal.h: touch_line); the register path below fetches them itself
    int sink = 0;
    if (!(NB * RATIO <= 160 && chn == 1 && packets_per_call == 1)) {
        const int16_t *row = pcm + (size_t)stream * stream_stride;
        const int n_i16 = packets_per_call * NB * RATIO * chn;
#pragma unroll 1
        for (int i = 0; i < n_i16; i += 32) touch_line(row + i, sink);
    }
    touch_done(sink);  // one HBM round trip for everything, instead of one per field along the chain
    VAD_PROF(0);  // state in
    const LaneBuf hp120{lds + lane}, lp120{lds + lane + 64 * (NB / 2)}, hp60{lds + lane + 64 * NB},
        lp60{lds + lane + 64 * (NB + NB / 4)};
    constexpr int PKG = NB * RATIO;  // frames (mono samples) per packet at the stream's rate
    // One mono packet per call with 16-byte aligned rows (the batched chain's case): the lane's packet is fetched as
    // uint4's in one batch, analysed from registers, attenuated there and written back as uint4's -- instead of ~2 x PKG
    // dependent two-byte accesses along the chain.
    constexpr bool kRegPath = PKG <= 160;  // 20 uint4 per lane; longer packets keep the memory path (register budget)
    constexpr int NV = kRegPath ? PKG / 8 : 1;
    const bool fast = kRegPath && chn == 1 && packets_per_call == 1 && (stream_stride % 8) == 0 && (call_stride % 8) == 0 &&
                      (reinterpret_cast<size_t>(pcm) % 16) == 0;
    bool done = false;
    if constexpr (kRegPath) {
        if (fast) {
        done = true;
        for (int call = 0; call < n_calls; call++) {
            uint4 *frame4 = reinterpret_cast<uint4 *>(pcm + (size_t)stream * stream_stride + (size_t)call * call_stride);
            uint4 raw[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) raw[j] = frame4[j];
            int reduce = S.h(V16_REDUCE);
            VAD_PROF(1);  // packet in (issue only: the loads are consumed inside the first loop)
            const int r = vad_packet<NB, RATIO>(S, RegSrc<NV>{raw}, hp120, lp120, hp60, lp60);
            if (r == 0) {
                if (reduce < 4) reduce += 1;
            } else {
                if (reduce > 0) reduce -= 1;
            }
            auto att = [&](unsigned w) {  // both int16 halves >> reduce (arithmetic)
                const int lo = (int)(int16_t)(w & 0xffffu) >> reduce, hi = (int)(int16_t)(w >> 16) >> reduce;
                return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16);
            };
#pragma unroll
            for (int j = 0; j < NV; j++) frame4[j] = make_uint4(att(raw[j].x), att(raw[j].y), att(raw[j].z), att(raw[j].w));
            S.h(V16_REDUCE) = (int16_t)reduce;
            VAD_PROF(7);  // hangover + attenuate + packet out
        }
        }
    }
    for (int call = 0; call < (done ? 0 : n_calls); call++) {
        int16_t *frame = pcm + (size_t)stream * stream_stride + (size_t)call * call_stride;
        const int n_mono = packets_per_call * PKG;
        if (chn > 1) {  // in-place mean downmix, src/webrtc.c:104-116
            for (int i = 0; i < n_mono; i++) {
                int32_t acc = 0;
                for (int c = 0; c < chn; c++) acc += frame[i * chn + c];
                frame[i] = (int16_t)(acc / chn);
            }
        }
        int reduce = S.h(V16_REDUCE);
        for (int it = 0; it < packets_per_call; it++) {
            const int r = vad_packet<NB, RATIO>(S, MemSrc{frame}, hp120, lp120, hp60, lp60);  // always packet 0 (quirk 1)
            if (r == 0) {
                if (reduce < 4) reduce += 1;
            } else {
                if (reduce > 0) reduce -= 1;
            }
            if (it == 0)  // for (cReduce = cLen; cReduce < pkgFrame; ...) only covers anything when cLen == 0
                for (int i = 0; i < PKG; i++) frame[i] = (int16_t)(frame[i] >> reduce);
        }
        S.h(V16_REDUCE) = (int16_t)reduce;
        if (chn > 1) {  // expand backwards, src/webrtc.c:145-150
            for (int i = n_mono - 1; i >= 0; i--) {
                const int16_t v = frame[i];
                for (int c = chn - 1; c >= 0; c--) frame[i * chn + c] = v;
            }
        }
    }
    {
        int16_t *o16 = s16 + stream;
        int32_t *o32 = s32 + stream;
#pragma unroll
        for (int f = 0; f < V16_AGE; f++) o16[(size_t)f * n_streams] = r16[f];
#pragma unroll
        for (int f = V16_MEAN_VALUE; f < V16_WORDS; f++) o16[(size_t)f * n_streams] = r16[f - kVadMinFields];
#pragma unroll
        for (int f = 0; f < V32_WORDS; f++) o32[(size_t)f * n_streams] = r32[f];
#pragma unroll 8
        for (int f = 0; f < kVadMinFields; f++) o16[(size_t)(V16_AGE + f) * n_streams] = minlds[f * 64 + lane];
    }
    VAD_PROF(8);  // state out (issue)
}
#ifdef WMX_VAD_PROF
extern "C" int wmx_debug_vad_prof(unsigned long long *out16, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_vad_prof), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_vad_prof), z, sizeof(z));
    }
    return 0;
}
#endif

__global__ void vad_fill_state(int16_t *s16, int32_t *s32, const int16_t *t16, int n_streams) {
    const size_t total = (size_t)V16_WORDS * n_streams;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        s16[i] = t16[i / n_streams];
    const size_t total32 = (size_t)V32_WORDS * n_streams;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total32; i += (size_t)gridDim.x * blockDim.x) s32[i] = 0;
}
// vad_release + vad_init for the listed streams
__global__ void vad_fill_idx(int16_t *s16, int32_t *s32, const int16_t *t16, int n_streams, const int32_t *idx, int n_idx) {
    for (int j = blockIdx.x; j < n_idx; j += gridDim.x) {
        const size_t i = (size_t)idx[j];
        for (int f = threadIdx.x; f < V16_WORDS; f += blockDim.x) s16[(size_t)f * n_streams + i] = t16[f];
        for (int f = threadIdx.x; f < V32_WORDS; f += blockDim.x) s32[(size_t)f * n_streams + i] = 0;
    }
}

}  // namespace
}  // namespace wmx

struct wmx_vad {
    int device;  // the HIP device the state lives on (current device at create); every entry point switches to it
    int n_streams, chn, freq, interval_ms, pkg;
    int16_t *d_s16;
    int32_t *d_s32;
    int16_t *d_tmpl;  // the 16-bit fields vad_init gives a stream (the 32-bit ones start at zero)
    bool one_lane;    // WMIX_AMD_VAD_ONE_LANE (developer A/B switch), read once at create
    wmx::StreamLife life;
};

extern "C" {

int wmx_vad_destroy(wmx_vad *h) {
    WMX_ON_DEVICE(h);
    if (!h) return 0;
    if (h->d_s16) (void)hipFree(h->d_s16);
    if (h->d_s32) (void)hipFree(h->d_s32);
    if (h->d_tmpl) (void)hipFree(h->d_tmpl);
    h->life.release();
    delete h;
    return 0;
}

// vad_release + vad_init for the listed streams (src/webrtc.c:40-82, 153-164): WebRtcVad_InitCore state, reduce = 4
int wmx_vad_reset_streams(wmx_vad *h, const int32_t *idx, int n, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h || n < 0 || (n > 0 && !idx)) return WMX_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = wmx::as_stream(stream);
    const int32_t *d_idx = nullptr;
    const int rc = h->life.upload(idx, n, h->n_streams, s, &d_idx);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(wmx::vad_fill_idx, dim3((unsigned)(n < 4096 ? n : 4096)), dim3(64), 0, s, h->d_s16, h->d_s32, (const int16_t *)h->d_tmpl,
                       h->n_streams, d_idx, n);
    WMX_LAUNCH_CHECK();
    return h->life.done(s);
}

int wmx_vad_set_active(wmx_vad *h, const uint8_t *host_mask, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    return h->life.set_active(h->n_streams, host_mask, wmx::as_stream(stream));
}

int wmx_vad_create(wmx_vad **out, int n_streams, int chn, int freq, int interval_ms) {
    using namespace wmx;
    if (!out) return WMX_EINVAL;
    *out = nullptr;
    // vad_init: freq <= 32000 and a multiple of 8000 (src/webrtc.c:43-44)
    if ((freq != 8000 && freq != 16000 && freq != 32000) || chn < 1 || n_streams < 1) {
        set_error("wmx_vad_create: unsupported n_streams=%d chn=%d freq=%d", n_streams, chn, freq);
        return WMX_EINVAL;
    }
    wmx_vad *h = new wmx_vad();
    if ((h->device = wmx::current_device()) < 0) {
        delete h;
        return WMX_ENODEV;
    }
    h->n_streams = n_streams;
    h->chn = chn;
    h->freq = freq;
    h->interval_ms = (freq <= 16000 && interval_ms % 20 == 0) ? 20 : 10;  // src/webrtc.c:57-66
    {
        const char *e = getenv("WMIX_AMD_VAD_ONE_LANE");  // unset, empty or "0": the pipelines
        h->one_lane = e && e[0] && e[0] != '0';
    }
    h->pkg = freq / 1000 * h->interval_ms;
    // WebRtcVad_InitCore vad_core.c:482-531 (start values vad_core.c:46-57) + reduce = 4 (src/webrtc.c:68)
    static const int16_t nm[12] = {6738, 4892, 7065, 6715, 6771, 3369, 7646, 3863, 7820, 7266, 5020, 4362};
    static const int16_t sm[12] = {8306, 10085, 10078, 11823, 11843, 6309, 9473, 9571, 10879, 7581, 8180, 7483};
    static const int16_t ns[12] = {378, 1064, 493, 582, 688, 593, 474, 697, 475, 688, 421, 455};
    static const int16_t ss[12] = {555, 505, 567, 524, 585, 1231, 509, 828, 492, 1540, 1079, 850};
    std::vector<int16_t> t(V16_WORDS, 0);
    for (int i = 0; i < 12; i++) {
        t[V16_NOISE_MEANS + i] = nm[i];
        t[V16_SPEECH_MEANS + i] = sm[i];
        t[V16_NOISE_STDS + i] = ns[i];
        t[V16_SPEECH_STDS + i] = ss[i];
    }
    for (int i = 0; i < 96; i++) t[V16_LOW + i] = 10000;
    for (int i = 0; i < 6; i++) t[V16_MEAN_VALUE + i] = 1600;
    t[V16_REDUCE] = 4;
    hipError_t e;
#define VAD_TRY(x)                                         \
    if ((e = (x)) != hipSuccess) {                         \
        int rc = hip_fail(e, #x, __FILE__, __LINE__);      \
        wmx_vad_destroy(h);                                \
        return rc;                                         \
    }
    VAD_TRY(hipMalloc(&h->d_s16, (size_t)V16_WORDS * n_streams * sizeof(int16_t)));
    VAD_TRY(hipMalloc(&h->d_s32, (size_t)V32_WORDS * n_streams * sizeof(int32_t)));
    VAD_TRY(hipMalloc(&h->d_tmpl, V16_WORDS * sizeof(int16_t)));
    VAD_TRY(hipMemcpy(h->d_tmpl, t.data(), V16_WORDS * sizeof(int16_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(vad_fill_state, dim3(512), dim3(256), 0, nullptr, h->d_s16, h->d_s32, h->d_tmpl, n_streams);
    VAD_TRY(hipGetLastError());
    VAD_TRY(hipDeviceSynchronize());
#undef VAD_TRY
    *out = h;
    return 0;
}

// stream migration: [header | V32_WORDS int32 fields | V16_WORDS int16 fields]
static constexpr uint32_t kVadBlobVersion = 1;  // bump when the meaning of a state word changes (wmx_internal.h: blob_layout)
int wmx_vad_stream_state_bytes(const wmx_vad *h) { return h ? (int)(sizeof(wmx::BlobHeader) + wmx::V32_WORDS * 4 + wmx::V16_WORDS * 2) : WMX_EINVAL; }

int wmx_vad_export_stream(wmx_vad *h, int stream_index, void *host_blob) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || !host_blob || stream_index < 0 || stream_index >= h->n_streams) return WMX_EINVAL;
    WMX_HIP(hipDeviceSynchronize());
    char *p = static_cast<char *>(host_blob);
    blob_begin(p, blob_tag("VAD "), blob_layout((uint32_t)(h->freq + h->interval_ms), kVadBlobVersion), V32_WORDS * 4 + V16_WORDS * 2);
    p += sizeof(BlobHeader);
    WMX_HIP(column_to_host(reinterpret_cast<int32_t *>(p), h->d_s32, V32_WORDS, h->n_streams, stream_index));
    WMX_HIP(column_to_host(reinterpret_cast<int16_t *>(p + V32_WORDS * 4), h->d_s16, V16_WORDS, h->n_streams, stream_index));
    return 0;
}

int wmx_vad_import_stream(wmx_vad *h, int stream_index, const void *host_blob) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || !host_blob || stream_index < 0 || stream_index >= h->n_streams) return WMX_EINVAL;
    const int rc = blob_check(host_blob, blob_tag("VAD "), blob_layout((uint32_t)(h->freq + h->interval_ms), kVadBlobVersion), V32_WORDS * 4 + V16_WORDS * 2);
    if (rc) return rc;
    WMX_HIP(hipDeviceSynchronize());
    const char *p = static_cast<const char *>(host_blob) + sizeof(BlobHeader);
    WMX_HIP(column_from_host(h->d_s32, reinterpret_cast<const int32_t *>(p), V32_WORDS, h->n_streams, stream_index));
    WMX_HIP(column_from_host(h->d_s16, reinterpret_cast<const int16_t *>(p + V32_WORDS * 4), V16_WORDS, h->n_streams, stream_index));
    return 0;
}

int wmx_vad_packet_samples(const wmx_vad *h) { return h ? h->pkg * h->chn : WMX_EINVAL; }

int wmx_vad_process(wmx_vad *h, int16_t *d_pcm, int packets_per_call, int n_calls, long stream_stride, long call_stride,
                    void *stream) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || packets_per_call < 1 || n_calls < 0) {
        set_error("wmx_vad_process: bad argument");
        return WMX_EINVAL;
    }
    if (n_calls == 0) return 0;  // nothing to do, whatever the pointer is
    if (!d_pcm) {
        set_error("wmx_vad_process: null buffer");
        return WMX_EINVAL;
    }
    // Only the FIRST packet of a call is ever read or written (the wrapper never advances its pointer),
    // so that packet is what must not overlap a neighbour -- a call's later packets may lie anywhere (packet-major batches).
    // That holds for ONE channel.  With interleaved channels the wrapper first averages the channels of the WHOLE call in place
    // and expands them again at the end (src/webrtc.c:104-116, 145-150): packets_per_call * pkg * chn contiguous int16 from the
    // stream's first packet are read and rewritten, so all of them must belong to the stream (round-3 ADVICE).
    const long per_pkt = (long)h->pkg * h->chn;
    const long touched = h->chn > 1 ? per_pkt * packets_per_call : per_pkt;
    if (call_stride < touched || (h->n_streams > 1 && stream_stride < touched)) {
        set_error("wmx_vad_process: strides (%ld, %ld) smaller than what a call touches (%ld samples: %d channel(s), %d packet(s) per call)",
                  stream_stride, call_stride, touched, h->chn, packets_per_call);
        return WMX_EINVAL;
    }
    const dim3 grid((h->n_streams + 63) / 64), block(64);
    hipStream_t s = as_stream(stream);
    const int nb = h->pkg / (h->freq / 8000);  // packet length at 8 kHz: 80 or 160
    const int ratio = h->freq / 8000;
    // one- and two-channel 10 ms and 20 ms packets at 8 / 16 / 32 kHz with 16-byte aligned rows -- every shape vad_init accepts for
    // the batched chains and the daemon's own cadence -- go through the four-wave pipeline (two channels: one packet per call, which is
    // what the heartbeat makes; several packets of interleaved channels per call keep the one-lane kernel's whole-call down-mix);
    // more than two channels and odd alignment through the one-lane-per-stream kernel
    const bool pipe = (h->chn == 1 || (h->chn == 2 && packets_per_call == 1)) && (nb == 80 || (nb == 160 && ratio <= 2)) &&
                      (stream_stride % 8) == 0 && (call_stride % 8) == 0 && (reinterpret_cast<size_t>(d_pcm) % 16) == 0 && !h->one_lane;
    if (pipe) {
#define VAD_PIPE(R, NB, CH)                                                                                                          \
    hipLaunchKernelGGL((vad_pipe_kernel<R, NB, CH>), grid, dim3(256), 0, s, h->d_s16, h->d_s32, d_pcm, h->n_streams, packets_per_call, \
                       n_calls, stream_stride, call_stride, h->life.d_active)
#define VAD_PIPE_CH(R, NB)     \
    do {                       \
        if (h->chn == 1)       \
            VAD_PIPE(R, NB, 1); \
        else                   \
            VAD_PIPE(R, NB, 2); \
    } while (0)
        if (ratio == 1 && nb == 80)
            VAD_PIPE_CH(1, 80);
        else if (ratio == 2 && nb == 80)
            VAD_PIPE_CH(2, 80);
        else if (ratio == 4)
            VAD_PIPE_CH(4, 80);
        else if (ratio == 1)
            VAD_PIPE_CH(1, 160);
        else
            VAD_PIPE_CH(2, 160);
#undef VAD_PIPE_CH
#undef VAD_PIPE
        WMX_LAUNCH_CHECK();
        return 0;
    }
#define VAD_LAUNCH(NB, R)                                                                                                 \
    hipLaunchKernelGGL((vad_kernel<NB, R>), grid, block, 0, s, h->d_s16, h->d_s32, d_pcm, h->n_streams, packets_per_call, \
                       n_calls, stream_stride, call_stride, h->chn, h->life.d_active)
    if (nb == 80 && ratio == 1)
        VAD_LAUNCH(80, 1);
    else if (nb == 80 && ratio == 2)
        VAD_LAUNCH(80, 2);
    else if (nb == 80 && ratio == 4)
        VAD_LAUNCH(80, 4);
    else if (nb == 160 && ratio == 1)
        VAD_LAUNCH(160, 1);
    else if (nb == 160 && ratio == 2)
        VAD_LAUNCH(160, 2);
    else {
        set_error("wmx_vad_process: unsupported packet %d samples at %d Hz", h->pkg, h->freq);
        return WMX_EINVAL;
    }
#undef VAD_LAUNCH
    WMX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
