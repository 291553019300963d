// agc.hip -- batched legacy (fixed-point) AGC for gfx950: one LANE per stream (agc_kernel), and for mono packets one lane
// per stream in four waves that split a packet's work (agc_pipe_kernel, further down).
//
// Replaces, for many independent streams per launch, wmix's agc_process() (src/webrtc.c:767-819)
// over WebRtcAgc_Process in adaptive-digital mode, target 0 dBFS, limiter off
// (W:modules/audio_processing/agc/legacy/analog_agc.c:1134-1229 -> digital_agc.c:294-604
// ProcessDigital, :633-771 ProcessVad; WebRtcSpl_DownsampleBy2 resample_by_2.c:70-124;
// WebRtcSpl_Sqrt spl_sqrt.c).  Everything is a per-stream integer recurrence (ten 1 ms
// envelope steps, a 4 kHz level detector, a per-sample gain ramp), so 64 streams share a
// wavefront, state is field-major ([field][stream], one coalesced line per access) and the
// 32-entry gain table sits in LDS: ONE table while every stream of the batch has the same compression gain (the common case,
// kernels with PS = false), else the table of each of the workgroup's 64 streams, [entry][lane] (PS = true: the compression
// gain is per handle in the reference, agc_init's `value` / agc_addition, src/webrtc.c:694-753, 824-839).  The channel count is
// a template parameter (1, 2; 0 = any, at run time): with a run-time `for (c < chn)` around every
// sample access the compiler could not batch the packet's loads and the kernel ran 3x longer.
// Bit-exact.
//
// WebRtcAgc_ProcessAnalog also runs in the reference (lowLevelSignal == 0) but with
// inMicLevel = 0 it only moves analog-side bookkeeping that wmix throws away and cannot fail
// (SURVEY.md section 8 row a15); it has no device counterpart.  The gain table itself is computed on
// the host by WebRtcAgc_CalculateGainTable's integer recipe (digital_agc.c:61-257).
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "wmx_internal.h"
#include "agc_gain_table.h"
#include "spl_dev.h"

namespace wmx {
namespace {

enum : int {  // DigitalAgc + AgcVad vadNearend (digital_agc.h:26-53)
    A32_CAP_SLOW = 0,
    A32_CAP_FAST = 1,
    A32_GAIN = 2,
    A32_DOWN = 3,  // downState[8]
    A32_VAR_LONG = 11,
    A32_VAR_SHORT = 12,
    A32_WORDS = 13,
    A16_GATE_PREV = 0,
    A16_HP = 1,
    A16_COUNTER = 2,
    A16_LOG_RATIO = 3,
    A16_MEAN_LONG = 4,
    A16_STD_LONG = 5,
    A16_MEAN_SHORT = 6,
    A16_STD_SHORT = 7,
    A16_WORDS = 8,
};

struct AgcRef {
    // the lane's copy of the stream's 21 state words, in registers for the length of the launch (every index is a
    // compile-time constant): one batch of loads at kernel entry, one batch of stores at the end
    // (pitch 1); the run-time channel-count fallback keeps its loops rolled and addresses the state rows in memory
    // (pitch = number of streams)
    int16_t *r16;  // [A16_WORDS]
    int32_t *r32;  // [A32_WORDS]
    size_t pitch;
    __device__ __forceinline__ int16_t &h(int f) const { return r16[(size_t)f * pitch]; }
    __device__ __forceinline__ int32_t &w(int f) const { return r32[(size_t)f * pitch]; }
};

// resample_by_2.c:70-124: two input samples -> one output sample
__device__ __forceinline__ int16_t down2_step(int16_t a, int16_t b, int32_t *st) {
    int32_t in32 = (int32_t)a << 10, diff, t1, t2;
    diff = wsub(in32, st[1]);
    t1 = spl_scalediff32(12199, diff, st[0]);
    st[0] = in32;
    diff = wsub(t1, st[2]);
    t2 = spl_scalediff32(37471, diff, st[1]);
    st[1] = t1;
    diff = wsub(t2, st[3]);
    st[3] = spl_scalediff32(60255, diff, st[2]);
    st[2] = t2;
    in32 = (int32_t)b << 10;
    diff = wsub(in32, st[5]);
    t1 = spl_scalediff32(3284, diff, st[4]);
    st[4] = in32;
    diff = wsub(t1, st[6]);
    t2 = spl_scalediff32(24441, diff, st[5]);
    st[5] = t1;
    diff = wsub(t2, st[7]);
    st[7] = spl_scalediff32(49528, diff, st[6]);
    st[6] = t2;
    return sat_w16(wadd(wadd(st[3], st[7]), 1024) >> 11);
}

// The gain table as the kernels see it: one table shared by the workgroup, or one per lane ([entry][lane]: conflict-free whatever
// entries the lanes ask for)
struct GainShared {
    const int32_t *t;
    __device__ __forceinline__ int32_t operator()(int e) const { return t[e]; }
};
struct GainPerLane {
    const int32_t *t;  // already offset by the lane
    __device__ __forceinline__ int32_t operator()(int e) const { return t[e * 64]; }
};
// stages the tables of a workgroup's 64 streams: entry e of lane l at [e * 64 + l]
__device__ __forceinline__ void stage_lane_tables(int32_t *lds, const int32_t *__restrict__ tables_g, const uint16_t *__restrict__ stream_table,
                                                  int n_streams) {
    for (int i = threadIdx.x; i < 32 * 64; i += blockDim.x) {
        const int e = i >> 6, l = i & 63;
        int sidx = (int)blockIdx.x * 64 + l;
        if (sidx >= n_streams) sidx = n_streams - 1;
        lds[i] = tables_g[(int)stream_table[sidx] * 32 + e];
    }
}

// Everything between the two passes over a packet: ProcessVad's statistics from the level detector's energy, the envelope
// followers over the ten per-millisecond peaks, the gain curve, the gate and the overflow limiter -> gains[0..10]
template <class GT>
__device__ __forceinline__ void agc_decide(const AgcRef &S, const GT gain_table, int32_t nrg, const int32_t (&env)[10],
                                           int32_t (&gains)[11]) {
    // ---- ProcessVad statistics (digital_agc.c:685-770)
    int16_t std_long, std_short, logratio;
    {
        const int16_t zeros = nrg == 0 ? 31 : (int16_t)__clz(nrg);
        const int16_t dB = (int16_t)((15 - zeros) << 11);
        int16_t counter = S.h(A16_COUNTER);
        if (counter < 250) counter++;
        S.h(A16_COUNTER) = counter;
        int32_t t32 = S.h(A16_MEAN_SHORT) * 15 + dB;
        const int16_t mean_short = (int16_t)(t32 >> 4);
        S.h(A16_MEAN_SHORT) = mean_short;
        t32 = (dB * dB) >> 12;
        t32 += S.w(A32_VAR_SHORT) * 15;
        const int32_t var_short = t32 / 16;
        S.w(A32_VAR_SHORT) = var_short;
        t32 = mean_short * mean_short;
        t32 = wsub(wshl(var_short, 12), t32);
        std_short = (int16_t)spl_sqrt(t32);
        S.h(A16_STD_SHORT) = std_short;
        t32 = S.h(A16_MEAN_LONG) * counter + dB;
        const int16_t mean_long = (int16_t)(t32 / sat_w16((int32_t)counter + 1));
        S.h(A16_MEAN_LONG) = mean_long;
        t32 = (dB * dB) >> 12;
        t32 += S.w(A32_VAR_LONG) * counter;
        const int32_t var_long = div_w32_w16(t32, sat_w16((int32_t)counter + 1));
        S.w(A32_VAR_LONG) = var_long;
        t32 = mean_long * mean_long;
        t32 = wsub(wshl(var_long, 12), t32);
        std_long = (int16_t)spl_sqrt(t32);
        S.h(A16_STD_LONG) = std_long;
        const int16_t t16 = 3 << 12;
        t32 = t16 * (int16_t)(dB - mean_long);
        t32 = div_w32_w16(t32, std_long);
        const int32_t t32b = (int32_t)S.h(A16_LOG_RATIO) * 53248;
        t32 += t32b >> 10;
        logratio = (int16_t)(t32 >> 6);
        if (logratio > 2048) logratio = 2048;
        if (logratio < -2048) logratio = -2048;
        S.h(A16_LOG_RATIO) = logratio;
    }
    // ---- decay, envelope followers, gain curve (digital_agc.c:336-460)
    int16_t decay;
    if (logratio > 1024)
        decay = -65;
    else if (logratio < 0)
        decay = 0;
    else
        decay = (int16_t)(((0 - logratio) * 65) >> 10);
    if (std_long < 4000)
        decay = 0;
    else if (std_long < 8096)
        decay = (int16_t)(((std_long - 4000) * decay) >> 12);
    int32_t cap_fast = S.w(A32_CAP_FAST), cap_slow = S.w(A32_CAP_SLOW);
    gains[0] = S.w(A32_GAIN);
    int16_t zeros = 0, frac = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        cap_fast = agc_scalediff32(-1000, cap_fast, cap_fast);
        if (env[k] > cap_fast) cap_fast = env[k];
        if (env[k] > cap_slow)
            cap_slow = agc_scalediff32(500, wsub(env[k], cap_slow), cap_slow);
        else
            cap_slow = agc_scalediff32(decay, cap_slow, cap_slow);
        const int32_t cur = cap_fast > cap_slow ? cap_fast : cap_slow;
        zeros = (int16_t)norm_u32((uint32_t)cur);
        if (cur == 0) zeros = 31;
        int32_t t32 = wshl(cur, zeros) & 0x7FFFFFFF;
        frac = (int16_t)(t32 >> 19);
        t32 = wmul(wsub(gain_table((zeros - 1) & 31), gain_table(zeros & 31)), frac);
        gains[k + 1] = wadd(gain_table(zeros & 31), t32 >> 12);
    }
    S.w(A32_CAP_FAST) = cap_fast;
    S.w(A32_CAP_SLOW) = cap_slow;
    // ---- gate (digital_agc.c:462-512)
    zeros = (int16_t)((zeros << 9) - (frac >> 3));
    int16_t zeros_fast = (int16_t)norm_u32((uint32_t)cap_fast);
    if (cap_fast == 0) zeros_fast = 31;
    {
        const int32_t t32 = wshl(cap_fast, zeros_fast) & 0x7FFFFFFF;
        zeros_fast = (int16_t)(zeros_fast << 9);
        zeros_fast = (int16_t)(zeros_fast - (int16_t)(t32 >> 22));
    }
    int16_t gate = (int16_t)(1000 + zeros_fast - zeros - std_short);
    if (gate < 0) {
        S.h(A16_GATE_PREV) = 0;
    } else {
        const int32_t t32 = S.h(A16_GATE_PREV) * 7;
        gate = (int16_t)((gate + t32) >> 3);
        S.h(A16_GATE_PREV) = gate;
    }
    const int32_t g0 = gain_table(0);
    if (gate > 0) {
        const int16_t gain_adj = gate < 2500 ? (int16_t)((2500 - gate) >> 5) : (int16_t)0;
#pragma unroll
        for (int k = 0; k < 10; k++) {
            int32_t t32;
            if (wsub(gains[k + 1], g0) > 8388608) {
                t32 = wsub(gains[k + 1], g0) >> 8;
                t32 = wmul(t32, 178 + gain_adj);
            } else {
                t32 = wmul(wsub(gains[k + 1], g0), 178 + gain_adj);
                t32 >>= 8;
            }
            gains[k + 1] = wadd(g0, t32);
        }
    }
    // ---- overflow limiter (digital_agc.c:514-541)
#pragma unroll
    for (int k = 0; k < 10; k++) {
        int z = 10;
        if (gains[k + 1] > 47453132) z = 16 - norm_w32(gains[k + 1]);
        int32_t gain32 = (gains[k + 1] >> z) + 1;
        gain32 = wmul(gain32, gain32);
        while (agc_mul32((env[k] >> 12) + 1, gain32) > shift_w32((int32_t)32767, 2 * (1 - z + 10))) {
            if (gains[k + 1] > 8388607)
                gains[k + 1] = (gains[k + 1] / 256) * 253;
            else
                gains[k + 1] = (gains[k + 1] * 253) / 256;
            gain32 = (gains[k + 1] >> z) + 1;
            gain32 = wmul(gain32, gain32);
        }
    }
#pragma unroll
    for (int k = 1; k < 10; k++)
        if (gains[k] > gains[k + 1]) gains[k] = gains[k + 1];
    S.w(A32_GAIN) = gains[10];
}

// one output sample of pass 2 (digital_agc.c:552-603); the first sub-frame saturates, the others wrap
__device__ __forceinline__ int16_t agc_apply(int16_t x, int32_t gain32, bool first_subframe) {
    if (first_subframe) {
        const int32_t o = wmul(x, wadd(gain32, 127) >> 7) >> 16;
        if (o > 4095) return 32767;
        if (o < -4096) return -32768;
    }
    return (int16_t)(wmul(x, gain32 >> 4) >> 16);
}

// One packet (10*L mono samples) of one stream.  in/out point at the packet's first frame;
// `chn` interleaved channels are averaged on input and duplicated on output (src/webrtc.c:789-815).
template <int L, int CHN, class GT>  // L samples per millisecond sub-frame: 8 (8 kHz) or 16 (16 / 32 kHz); CHN interleaved channels
__device__ void agc_packet(const AgcRef &S, const GT gain_table, const int16_t *in, int16_t *out, int chn_rt) {
    const int chn = CHN ? CHN : chn_rt;  // CHN = 1, 2: compile-time (the daemon's cases); 0: any count, at run time
    constexpr int L2 = (L == 8) ? 3 : 4;
    auto load = [&](int i) -> int16_t {
        if (chn == 1) return in[i];
        int32_t acc = 0;
        for (int c = 0; c < chn; c++) acc += in[i * chn + c];
        return (int16_t)(acc / chn);
    };
    // ---- pass 1 over the packet: level detector (ProcessVad) and per-millisecond peak energy
    int32_t env[10];
    int32_t nrg = 0;
    {
        int32_t ds[8];
#pragma unroll
        for (int i = 0; i < 8; i++) ds[i] = S.w(A32_DOWN + i);
        int16_t hp = S.h(A16_HP);
#pragma unroll
        for (int k = 0; k < 10; k++) {
            int32_t mx = 0;
            int16_t x[L];
#pragma unroll
            for (int n = 0; n < L; n++) {
                x[n] = load(k * L + n);
                const int32_t e = x[n] * x[n];
                if (e > mx) mx = e;
            }
            env[k] = mx;
            int16_t b2[4];
            if (L == 16) {
                int16_t b1[8];
#pragma unroll
                for (int j = 0; j < 8; j++) b1[j] = (int16_t)(((int32_t)x[2 * j] + (int32_t)x[2 * j + 1]) >> 1);
#pragma unroll
                for (int j = 0; j < 4; j++) b2[j] = down2_step(b1[2 * j], b1[2 * j + 1], ds);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) b2[j] = down2_step(x[2 * j], x[2 * j + 1], ds);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int32_t o = b2[j] + hp;
                const int32_t t = 600 * o;
                hp = (int16_t)((t >> 10) - b2[j]);
                nrg = wadd(nrg, wmul(o, o) >> 6);  // |o| reaches 65 534: the square wraps in the reference (digital_agc.c:633), negative included
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) S.w(A32_DOWN + i) = ds[i];
        S.h(A16_HP) = hp;
    }
    int32_t gains[11];
    agc_decide(S, gain_table, nrg, env, gains);
    // ---- pass 2: apply the ramped gain (digital_agc.c:552-603)
    auto store = [&](int i, int16_t v) {
        for (int c = 0; c < chn; c++) out[i * chn + c] = v;
    };
#pragma unroll
    for (int k = 0; k < 10; k++) {
        const int32_t delta = wshl(wsub(gains[k + 1], gains[k]), 4 - L2);
        int32_t gain32 = wshl(gains[k], 4);
        int16_t x[L];
#pragma unroll
        for (int n = 0; n < L; n++) x[n] = load(k * L + n);  // all loads of the sub-frame before its stores (in == out)
#pragma unroll
        for (int n = 0; n < L; n++) {
            int16_t y;
            if (k == 0) {
                const int32_t o = wmul(x[n], wadd(gain32, 127) >> 7) >> 16;
                if (o > 4095)
                    y = 32767;
                else if (o < -4096)
                    y = -32768;
                else
                    y = (int16_t)(wmul(x[n], gain32 >> 4) >> 16);
            } else {
                y = (int16_t)(wmul(x[n], gain32 >> 4) >> 16);
            }
            store(k * L + n, y);
            gain32 = wadd(gain32, delta);
        }
    }
}

template <int L, int CHN, bool PS>
__global__ __launch_bounds__(64) void agc_kernel(int16_t *s16, int32_t *s32, const int32_t *gain_table_g, const uint16_t *__restrict__ stream_table,
                                                 const int16_t *in, int16_t *out, int n_streams, int n_packets, long stream_stride,
                                                 long packet_stride, int chn_rt, const uint8_t *__restrict__ active) {
    __shared__ int32_t gain_lds[PS ? 32 * 64 : 32];
    if constexpr (PS)
        stage_lane_tables(gain_lds, gain_table_g, stream_table, n_streams);
    else if (threadIdx.x < 32)
        gain_lds[threadIdx.x] = gain_table_g[threadIdx.x];
    __syncthreads();
    using GT = typename std::conditional<PS, GainPerLane, GainShared>::type;
    const GT gain_table{PS ? gain_lds + threadIdx.x : gain_lds};
    const int stream = blockIdx.x * 64 + threadIdx.x;
    if (!stream_active(active, stream, n_streams)) return;
    if constexpr (CHN == 0) {
        const AgcRef S{s16 + stream, s32 + stream, (size_t)n_streams};
        for (int p = 0; p < n_packets; p++) {
            const size_t off = (size_t)stream * stream_stride + (size_t)p * packet_stride;
            agc_packet<L, CHN, GT>(S, gain_table, in + off, out + off, chn_rt);
        }
    } else {
        int16_t r16[A16_WORDS];
        int32_t r32[A32_WORDS];
#pragma unroll
        for (int f = 0; f < A32_WORDS; f++) r32[f] = s32[(size_t)f * n_streams + stream];
#pragma unroll
        for (int f = 0; f < A16_WORDS; f++) r16[f] = s16[(size_t)f * n_streams + stream];
        const AgcRef S{r16, r32, 1};
        for (int p = 0; p < n_packets; p++) {
            const size_t off = (size_t)stream * stream_stride + (size_t)p * packet_stride;
            agc_packet<L, CHN, GT>(S, gain_table, in + off, out + off, chn_rt);
        }
#pragma unroll
        for (int f = 0; f < A32_WORDS; f++) s32[(size_t)f * n_streams + stream] = r32[f];
#pragma unroll
        for (int f = 0; f < A16_WORDS; f++) s16[(size_t)f * n_streams + stream] = r16[f];
    }
}

// ================================================================== the mono packet as a four-wave pipeline
// agc_kernel runs a stream in one lane from end to end: 65 536 streams are 1 024 waves, one per SIMD, and the launch lasts as
// long as one wave's chain of ~3 900 instructions issued one every ~5 cycles at best (a wave alone on its SIMD, DESIGN_HISTORY.md section 5c).  Here a workgroup still owns 64 streams
// (lane = stream), but as four waves: waves 0..2 share the packet's ten 1 ms sub-frames ({0..3}, {4..6}, {7..9}), load them
// once (two 16-byte loads per sub-frame at 16 kHz), take their peak energies and hand the level detector's input (pair
// averages at 16 kHz) over through LDS; wave 3, which holds no samples, runs the serial part -- the detector's decimator,
// ProcessVad's statistics, the envelope followers, gain curve, gate and limiter (agc_decide) -- and publishes the eleven
// gains; waves 0..2 then apply the gain ramp to the samples they still hold and store them.  Same integer operations per
// stream as agc_kernel, which stays for more than two channels and unaligned rows.  CHN = 2: the interleaved pair is averaged
// on the way in and the result written to both channels (src/webrtc.c:789-815), like agc_packet's load / store.
template <int L, int CHN, bool PS>
__global__ __launch_bounds__(256, CHN == 1 ? 4 : 2) void agc_pipe_kernel(int16_t *s16, int32_t *s32, const int32_t *gain_table_g,
                                                       const uint16_t *__restrict__ stream_table, const int16_t *in,
                                                       int16_t *out, int n_streams, int n_packets, long stream_stride,
                                                       long packet_stride, const uint8_t *__restrict__ active) {
    constexpr int L2 = (L == 8) ? 3 : 4, VPS = L * CHN / 8;  // uint4 per sub-frame
    __shared__ int32_t gain_lds[PS ? 32 * 64 : 32];
    // Exchange areas, TWICE: the packets of a launch are pipelined -- while wave 3 runs packet p's serial part on one set, waves 0..2
    // apply packet p - 1's gains from the other gain area and put packet p + 1's peaks and detector input into the other input
    // set.  One workgroup barrier per packet instead of three, and a packet behind the first costs max(serial part, pass 2 + pass 1)
    // instead of their sum (round 4: the daemon's own heartbeat is two packets per call, configs[4]'s 10 ms are two 5 ms packets).
    __shared__ int16_t xdet[2][80 * 64];  // the detector's 80 input samples of every stream, [sample][lane]
    __shared__ int32_t xenv[2][10 * 64];  // per-millisecond peak energies
    __shared__ int32_t xgain[2][11 * 64];
    if constexpr (PS)
        stage_lane_tables(gain_lds, gain_table_g, stream_table, n_streams);
    else if (threadIdx.x < 32)
        gain_lds[threadIdx.x] = gain_table_g[threadIdx.x];
    const int lane = threadIdx.x & 63;
    using GT = typename std::conditional<PS, GainPerLane, GainShared>::type;
    const GT gain_table{PS ? gain_lds + lane : gain_lds};
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int stream_raw = (int)blockIdx.x * 64 + lane;
    // a lane without a stream recomputes the last one, a lane whose stream is switched off its own, and neither stores anything
    const bool live = stream_active(active, stream_raw, n_streams);
    const int stream = stream_raw < n_streams ? stream_raw : n_streams - 1;
    const int k0 = wave == 0 ? 0 : (wave == 1 ? 4 : 7), nk = wave == 0 ? 4 : (wave == 3 ? 0 : 3);  // this wave's sub-frames
    int16_t r16[A16_WORDS];
    int32_t r32[A32_WORDS];
    const AgcRef S{r16, r32, 1};
    if (wave == 3) {
#pragma unroll
        for (int f = 0; f < A32_WORDS; f++) r32[f] = s32[(size_t)f * n_streams + stream];
#pragma unroll
        for (int f = 0; f < A16_WORDS; f++) r16[f] = s16[(size_t)f * n_streams + stream];
    }
    auto sample = [](const uint4 (&raw)[VPS], int i) -> int16_t {  // mono sample i of a sub-frame (compile-time i)
        if constexpr (CHN == 1) {
            const uint4 v = raw[i >> 3];
            const unsigned w = ((i >> 1) & 3) == 0 ? v.x : (((i >> 1) & 3) == 1 ? v.y : (((i >> 1) & 3) == 2 ? v.z : v.w));
            return (int16_t)((i & 1) ? (w >> 16) : (w & 0xffffu));
        } else {  // frame i = one word: left | right << 16
            const uint4 v = raw[i >> 2];
            const unsigned w = (i & 3) == 0 ? v.x : ((i & 3) == 1 ? v.y : ((i & 3) == 2 ? v.z : v.w));
            const int32_t acc = (int32_t)(int16_t)(w & 0xffffu) + (int32_t)(int16_t)(w >> 16);
            return (int16_t)(acc / 2);
        }
    };
    // ---- pass 1 of packet p (waves 0..2, this wave's share): peak energy per sub-frame, the detector's input samples -> set p & 1.
    //      All of the wave's sub-frames are requested at once; pass 2 fetches them again from L2 -- held across the barriers, the
    //      samples cost the registers that decide how many workgroups share a CU.
    auto pass1 = [&](int p) {
        const uint4 *in4 = reinterpret_cast<const uint4 *>(in + (size_t)stream * stream_stride + (size_t)p * packet_stride);
        int16_t *det = xdet[p & 1];
        int32_t *env = xenv[p & 1];
        uint4 raw[4][VPS];
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
            if (kk < nk) {
#pragma unroll
                for (int j = 0; j < VPS; j++) raw[kk][j] = in4[(k0 + kk) * VPS + j];
            }
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
            if (kk < nk) {
                int32_t mx = 0;
#pragma unroll
                for (int n = 0; n < L; n++) {
                    const int32_t x = sample(raw[kk], n);
                    const int32_t e = x * x;
                    if (e > mx) mx = e;
                }
                env[(k0 + kk) * 64 + lane] = mx;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int16_t d = L == 16 ? (int16_t)(((int32_t)sample(raw[kk], 2 * j) + (int32_t)sample(raw[kk], 2 * j + 1)) >> 1) : sample(raw[kk], j);
                    det[((k0 + kk) * 8 + j) * 64 + lane] = d;
                }
            }
    };
    // ---- the serial part of packet p (wave 3): the detector's decimator, ProcessVad's statistics, the envelope followers, gain curve,
    //      gate and limiter -> the eleven gains of set p & 1
    auto serial = [&](int p) {
        const int16_t *det = xdet[p & 1];
        int32_t env[10], gains[11], nrg = 0;
        int32_t ds[8];
#pragma unroll
        for (int i = 0; i < 8; i++) ds[i] = S.w(A32_DOWN + i);
        int16_t hp = S.h(A16_HP);
#pragma unroll 2
        for (int k = 0; k < 10; k++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int16_t b2 = down2_step(det[(k * 8 + 2 * j) * 64 + lane], det[(k * 8 + 2 * j + 1) * 64 + lane], ds);
                const int32_t o = b2 + hp;
                const int32_t t = 600 * o;
                hp = (int16_t)((t >> 10) - b2);
                nrg = wadd(nrg, wmul(o, o) >> 6);  // |o| reaches 65 534: the square wraps in the reference (digital_agc.c:633), negative included
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) S.w(A32_DOWN + i) = ds[i];
        S.h(A16_HP) = hp;
#pragma unroll
        for (int k = 0; k < 10; k++) env[k] = xenv[p & 1][k * 64 + lane];
        agc_decide(S, gain_table, nrg, env, gains);
#pragma unroll
        for (int k = 0; k < 11; k++) xgain[p & 1][k * 64 + lane] = gains[k];
    };
    // ---- pass 2 of packet p (waves 0..2, this wave's share): the ramped gain applied, samples stored
    auto pass2 = [&](int p) {
        const size_t off = (size_t)stream * stream_stride + (size_t)p * packet_stride;
        const uint4 *in4 = reinterpret_cast<const uint4 *>(in + off);
        uint4 *out4 = reinterpret_cast<uint4 *>(out + off);
        const int32_t *gn = xgain[p & 1];
        uint4 raw[4][VPS];  // every load before the first store (in == out)
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
            if (kk < nk) {
#pragma unroll
                for (int j = 0; j < VPS; j++) raw[kk][j] = in4[(k0 + kk) * VPS + j];
            }
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
            if (kk < nk) {
                const int k = k0 + kk;
                const int32_t ga = gn[k * 64 + lane], gb = gn[(k + 1) * 64 + lane];
                const int32_t delta = wshl(wsub(gb, ga), 4 - L2);
                int32_t gain32 = wshl(ga, 4);
                unsigned yw[L * CHN / 2];
#pragma unroll
                for (int n = 0; n < L; n++) {
                    const int16_t y = agc_apply(sample(raw[kk], n), gain32, k == 0);
                    gain32 = wadd(gain32, delta);
                    if (CHN == 2)
                        yw[n] = (unsigned)(uint16_t)y | ((unsigned)(uint16_t)y << 16);
                    else if (n & 1)
                        yw[n >> 1] |= (unsigned)(uint16_t)y << 16;
                    else
                        yw[n >> 1] = (unsigned)(uint16_t)y;
                }
                if (live) {
#pragma unroll
                    for (int j = 0; j < VPS; j++) out4[k * VPS + j] = make_uint4(yw[4 * j], yw[4 * j + 1], yw[4 * j + 2], yw[4 * j + 3]);
                }
            }
    };
    if (nk) pass1(0);
    __syncthreads();  // packet 0's peaks and detector input are in LDS (and the gain table)
    for (int p = 0; p < n_packets; p++) {
        if (wave == 3) {
            serial(p);
        } else {
            if (p > 0) pass2(p - 1);             // its gains were published before the previous barrier
            if (p + 1 < n_packets) pass1(p + 1);  // into the set the serial wave is not reading
        }
        __syncthreads();  // packet p's gains and packet p + 1's inputs are in LDS; set p & 1 of the inputs may be rewritten
    }
    if (nk) pass2(n_packets - 1);
    if (wave == 3 && live) {
#pragma unroll
        for (int f = 0; f < A32_WORDS; f++) s32[(size_t)f * n_streams + stream] = r32[f];
#pragma unroll
        for (int f = 0; f < A16_WORDS; f++) s16[(size_t)f * n_streams + stream] = r16[f];
    }
}

// idx == nullptr: every stream; else the n_idx listed ones (wmx_agc_reset_streams)
__global__ void agc_fill_state(int16_t *s16, int32_t *s32, int n_streams, const int32_t *idx, int n_idx) {
    // WebRtcAgc_InitDigital digital_agc.c:259-282 + WebRtcAgc_InitVad :606-631
    const size_t count = idx ? (size_t)n_idx : (size_t)n_streams;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (size_t)gridDim.x * blockDim.x) {
        const size_t i = idx ? (size_t)idx[j] : j;
        for (int f = 0; f < A32_WORDS; f++) s32[(size_t)f * n_streams + i] = 0;
        for (int f = 0; f < A16_WORDS; f++) s16[(size_t)f * n_streams + i] = 0;
        s32[(size_t)A32_CAP_SLOW * n_streams + i] = 134217728;
        s32[(size_t)A32_GAIN * n_streams + i] = 65536;
        s32[(size_t)A32_VAR_LONG * n_streams + i] = 500 << 8;
        s32[(size_t)A32_VAR_SHORT * n_streams + i] = 500 << 8;
        s16[(size_t)A16_MEAN_LONG * n_streams + i] = 15 << 10;
        s16[(size_t)A16_MEAN_SHORT * n_streams + i] = 15 << 10;
        s16[(size_t)A16_COUNTER * n_streams + i] = 3;
    }
}

// stream_table[idx[j]] = t for the listed streams (idx == nullptr: every stream)
__global__ void agc_set_table(uint16_t *stream_table, int n_streams, const int32_t *idx, int n_idx, uint16_t t) {
    const size_t count = idx ? (size_t)n_idx : (size_t)n_streams;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (size_t)gridDim.x * blockDim.x)
        stream_table[idx ? (size_t)idx[j] : j] = t;
}

}  // namespace
}  // namespace wmx

struct wmx_agc {
    int device;  // the HIP device the state lives on (current device at create); every entry point switches to it
    int n_streams, chn, freq, pkg;
    int16_t *d_s16;
    int32_t *d_s32;
    // The compression gain is per handle in the reference (agc_init's `value`, agc_addition): one 32-entry table per DISTINCT value
    // in use, tables[0] being the batch's own (wmx_agc_create / wmx_agc_set_gain).  While every stream points at table 0 --
    // n_off_batch == 0 -- the launches are the PS = false kernels and never look at the map.
    static constexpr int kMaxTables = 256;  // valid gains are 0 .. 186 dB (host_gain_table): never reached
    int32_t *d_table;                       // [kMaxTables][32]
    uint16_t *d_stream_table;               // [n_streams] -> table index
    std::vector<uint16_t> stream_table;     // the host's copy (export / import, counting)
    std::vector<int> table_value;           // compression gain of table i
    std::vector<int32_t> tables;            // [n][32] host copies
    int n_off_batch;                        // streams whose table is not table 0
    bool one_lane;  // WMIX_AMD_AGC_ONE_LANE (developer A/B switch), read once at create
    wmx::StreamLife life;
};

namespace wmx {
namespace {
// index of the table for `value`, made (and uploaded, ordered on `s`) when new; < 0: WMX error (the reference's set_config fails)
int agc_table_for(wmx_agc *h, int value, hipStream_t s, const char *who) {
    for (size_t i = 1; i < h->table_value.size(); i++)
        if (h->table_value[i] == value) return (int)i;
    if (!h->table_value.empty() && h->table_value[0] == value) return 0;
    int32_t t[32];
    const int16_t comp = (int16_t)value;
    if (host_gain_table(t, comp, 0, false, analog_target_for(comp)) != 0) {
        set_error("%s: compression gain %d dB is outside the gain-table range", who, value);
        return WMX_EINVAL;
    }
    if ((int)h->table_value.size() >= wmx_agc::kMaxTables) {
        set_error("%s: more than %d distinct compression gains in one batch", who, wmx_agc::kMaxTables);
        return WMX_EINVAL;
    }
    const size_t i = h->table_value.size();
    h->table_value.push_back(value);
    h->tables.insert(h->tables.end(), t, t + 32);
    // the source is the vector's storage, which a later push_back may move: a blocking copy (a new gain value is a rare event)
    (void)s;
    hipError_t e = hipMemcpy(h->d_table + i * 32, t, sizeof(t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        h->table_value.pop_back();
        h->tables.resize(i * 32);
        return hip_fail(e, "hipMemcpy(gain table)", __FILE__, __LINE__);
    }
    return (int)i;
}
// the listed streams (already uploaded: d_idx) now use table t: host copy, count, device map (a scatter kernel ordered on `s`)
int agc_point_streams(wmx_agc *h, const int32_t *idx, const int32_t *d_idx, int n, int t, hipStream_t s) {
    for (int j = 0; j < n; j++) {
        uint16_t &cur = h->stream_table[(size_t)idx[j]];
        h->n_off_batch += (t != 0) - (cur != 0);
        cur = (uint16_t)t;
    }
    hipLaunchKernelGGL(agc_set_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h->d_stream_table, h->n_streams, d_idx, n, (uint16_t)t);
    WMX_LAUNCH_CHECK();
    return 0;
}
}  // namespace
}  // namespace wmx

extern "C" {

int wmx_agc_destroy(wmx_agc *h) {
    WMX_ON_DEVICE(h);
    if (!h) return 0;
    if (h->d_s16) (void)hipFree(h->d_s16);
    if (h->d_s32) (void)hipFree(h->d_s32);
    if (h->d_table) (void)hipFree(h->d_table);
    if (h->d_stream_table) (void)hipFree(h->d_stream_table);
    h->life.release();
    delete h;
    return 0;
}

// agc_release + agc_init for the listed streams (src/webrtc.c:694-753, 841-860).  wmx_agc_reset_streams: agc_init with the gain
// each stream has; wmx_agc_reset_streams_gain: agc_init(..., value, ...) -- the new handles' own compression gain.
static int agc_reset(wmx_agc *h, const int32_t *idx, int n, const int *value, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h || n < 0 || (n > 0 && !idx)) return WMX_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = wmx::as_stream(stream);
    const int32_t *d_idx = nullptr;
    int rc = h->life.upload(idx, n, h->n_streams, s, &d_idx);  // validates the list
    if (rc != 0) return rc;
    if (value) {
        const int t = wmx::agc_table_for(h, *value, s, "wmx_agc_reset_streams_gain");  // agc_init returns NULL: nothing is reset
        if (t < 0) return t;
        if ((rc = wmx::agc_point_streams(h, idx, d_idx, n, t, s)) != 0) return rc;
    }
    hipLaunchKernelGGL(wmx::agc_fill_state, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h->d_s16, h->d_s32, h->n_streams, d_idx, n);
    WMX_LAUNCH_CHECK();
    return h->life.done(s);
}
int wmx_agc_reset_streams(wmx_agc *h, const int32_t *idx, int n, void *stream) { return agc_reset(h, idx, n, nullptr, stream); }
int wmx_agc_reset_streams_gain(wmx_agc *h, const int32_t *idx, int n, int value, void *stream) { return agc_reset(h, idx, n, &value, stream); }

// agc_addition for the listed streams (src/webrtc.c:824-839: WebRtcAgc_set_config on a running handle -- the state stays, the
// table changes).  A value the reference's set_config refuses leaves every stream as it was (agc_addition prints and carries on).
int wmx_agc_set_gain_streams(wmx_agc *h, const int32_t *idx, int n, int value, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h || n < 0 || (n > 0 && !idx)) return WMX_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = wmx::as_stream(stream);
    const int32_t *d_idx = nullptr;
    int rc = h->life.upload(idx, n, h->n_streams, s, &d_idx);  // validates the list
    if (rc != 0) return rc;
    const int t = wmx::agc_table_for(h, value, s, "wmx_agc_set_gain_streams");
    if (t < 0) return t;
    if ((rc = wmx::agc_point_streams(h, idx, d_idx, n, t, s)) != 0) return rc;
    return h->life.done(s);
}

// the compression gain stream i runs with
int wmx_agc_stream_gain(const wmx_agc *h, int stream_index) {
    if (!h || stream_index < 0 || stream_index >= h->n_streams) return WMX_EINVAL;
    return h->table_value[h->stream_table[(size_t)stream_index]];
}

int wmx_agc_set_active(wmx_agc *h, const uint8_t *host_mask, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    return h->life.set_active(h->n_streams, host_mask, wmx::as_stream(stream));
}

// agc_addition for EVERY stream of the batch (src/webrtc.c:824-839): WebRtcAgc_set_config with a new compression gain -> the batch
// is back on one table (table 0) and on the kernels that know no other.  Returns WMX_EINVAL and changes nothing when the
// reference's set_config would fail.  Blocking (the device is drained: launches in flight still read the old tables).
int wmx_agc_set_gain(wmx_agc *h, int value) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h) return WMX_EINVAL;
    int32_t t[32];
    const int16_t comp = (int16_t)value;
    if (host_gain_table(t, comp, 0, false, analog_target_for(comp)) != 0) {
        set_error("wmx_agc_set_gain: compression gain %d dB is outside the gain-table range", value);
        return WMX_EINVAL;
    }
    WMX_HIP(hipDeviceSynchronize());
    WMX_HIP(hipMemcpy(h->d_table, t, sizeof(t), hipMemcpyHostToDevice));
    if (h->n_off_batch) WMX_HIP(hipMemset(h->d_stream_table, 0, (size_t)h->n_streams * sizeof(uint16_t)));
    h->table_value.assign(1, value);
    h->tables.assign(t, t + 32);
    std::fill(h->stream_table.begin(), h->stream_table.end(), (uint16_t)0);
    h->n_off_batch = 0;
    return 0;
}

int wmx_agc_create(wmx_agc **out, int n_streams, int chn, int freq, int interval_ms, int value) {
    using namespace wmx;
    (void)interval_ms;
    if (!out) return WMX_EINVAL;
    *out = nullptr;
    if ((freq != 8000 && freq != 16000 && freq != 32000) || chn < 1 || n_streams < 1) {  // src/webrtc.c:711-712
        set_error("wmx_agc_create: unsupported n_streams=%d chn=%d freq=%d", n_streams, chn, freq);
        return WMX_EINVAL;
    }
    wmx_agc *h = new wmx_agc();
    if ((h->device = wmx::current_device()) < 0) {
        delete h;
        return WMX_ENODEV;
    }
    h->n_streams = n_streams;
    {
        const char *e = getenv("WMIX_AMD_AGC_ONE_LANE");  // unset, empty or "0": the pipelines
        h->one_lane = e && e[0] && e[0] != '0';
    }
    h->chn = chn;
    h->freq = freq;
    h->pkg = freq / 1000 * (freq <= 16000 ? 10 : 5);  // 5 ms packets at 32 kHz, src/webrtc.c:724-728
    hipError_t e;
#define AGC_TRY(x)                                         \
    if ((e = (x)) != hipSuccess) {                         \
        int rc = hip_fail(e, #x, __FILE__, __LINE__);      \
        wmx_agc_destroy(h);                                \
        return rc;                                         \
    }
    AGC_TRY(hipMalloc(&h->d_s16, (size_t)A16_WORDS * n_streams * sizeof(int16_t)));
    AGC_TRY(hipMalloc(&h->d_s32, (size_t)A32_WORDS * n_streams * sizeof(int32_t)));
    AGC_TRY(hipMalloc(&h->d_table, (size_t)wmx_agc::kMaxTables * 32 * sizeof(int32_t)));
    AGC_TRY(hipMalloc(&h->d_stream_table, (size_t)n_streams * sizeof(uint16_t)));
    AGC_TRY(hipMemset(h->d_stream_table, 0, (size_t)n_streams * sizeof(uint16_t)));
    h->stream_table.assign((size_t)n_streams, (uint16_t)0);
    h->n_off_batch = 0;
    hipLaunchKernelGGL(agc_fill_state, dim3(512), dim3(256), 0, nullptr, h->d_s16, h->d_s32, n_streams, (const int32_t *)nullptr, 0);
    AGC_TRY(hipGetLastError());
    AGC_TRY(hipDeviceSynchronize());
#undef AGC_TRY
    const int rc = wmx_agc_set_gain(h, value);  // agc_init fails (NULL) when set_config fails
    if (rc != 0) {
        wmx_agc_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

// stream migration: [header | 13 int32 fields | 8 int16 fields | int32 compression gain] -- the gain is the handle's own
// (WebRtcAgc_set_config's value lives in the reference's handle), so it travels with the stream
static constexpr uint32_t kAgcBlobVersion = 1;  // bump when the meaning of a state word changes (wmx_internal.h: blob_layout)
int wmx_agc_stream_state_bytes(const wmx_agc *h) { return h ? (int)(sizeof(wmx::BlobHeader) + wmx::A32_WORDS * 4 + wmx::A16_WORDS * 2 + 4) : WMX_EINVAL; }

int wmx_agc_export_stream(wmx_agc *h, int stream_index, void *host_blob) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || !host_blob || stream_index < 0 || stream_index >= h->n_streams) return WMX_EINVAL;
    WMX_HIP(hipDeviceSynchronize());
    char *p = static_cast<char *>(host_blob);
    blob_begin(p, blob_tag("AGC "), blob_layout((uint32_t)h->freq, kAgcBlobVersion), A32_WORDS * 4 + A16_WORDS * 2 + 4);
    p += sizeof(BlobHeader);
    WMX_HIP(column_to_host(reinterpret_cast<int32_t *>(p), h->d_s32, A32_WORDS, h->n_streams, stream_index));
    WMX_HIP(column_to_host(reinterpret_cast<int16_t *>(p + A32_WORDS * 4), h->d_s16, A16_WORDS, h->n_streams, stream_index));
    const int32_t value = h->table_value[h->stream_table[(size_t)stream_index]];
    memcpy(p + A32_WORDS * 4 + A16_WORDS * 2, &value, 4);
    return 0;
}

int wmx_agc_import_stream(wmx_agc *h, int stream_index, const void *host_blob) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || !host_blob || stream_index < 0 || stream_index >= h->n_streams) return WMX_EINVAL;
    const int rc = blob_check(host_blob, blob_tag("AGC "), blob_layout((uint32_t)h->freq, kAgcBlobVersion), A32_WORDS * 4 + A16_WORDS * 2 + 4);
    if (rc) return rc;
    WMX_HIP(hipDeviceSynchronize());
    const char *p = static_cast<const char *>(host_blob) + sizeof(BlobHeader);
    int32_t value;
    memcpy(&value, p + A32_WORDS * 4 + A16_WORDS * 2, 4);
    const int t = agc_table_for(h, value, nullptr, "wmx_agc_import_stream");
    if (t < 0) return t;
    WMX_HIP(column_from_host(h->d_s32, reinterpret_cast<const int32_t *>(p), A32_WORDS, h->n_streams, stream_index));
    WMX_HIP(column_from_host(h->d_s16, reinterpret_cast<const int16_t *>(p + A32_WORDS * 4), A16_WORDS, h->n_streams, stream_index));
    uint16_t &cur = h->stream_table[(size_t)stream_index];
    h->n_off_batch += (t != 0) - (cur != 0);
    cur = (uint16_t)t;
    WMX_HIP(hipMemcpy(h->d_stream_table + stream_index, &cur, sizeof(uint16_t), hipMemcpyHostToDevice));
    return 0;
}

int wmx_agc_packet_samples(const wmx_agc *h) { return h ? h->pkg * h->chn : WMX_EINVAL; }

int wmx_agc_gain_table(const wmx_agc *h, int32_t *host_table32) {
    WMX_ON_DEVICE(h);
    if (!h || !host_table32) return WMX_EINVAL;
    memcpy(host_table32, h->tables.data(), 32 * sizeof(int32_t));  // table 0: the batch's own
    return 0;
}

int wmx_agc_process(wmx_agc *h, const int16_t *d_in, int16_t *d_out, int n_packets, long stream_stride, long packet_stride,
                    void *stream) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || n_packets < 0) {
        set_error("wmx_agc_process: bad argument");
        return WMX_EINVAL;
    }
    if (n_packets == 0) return 0;  // frameNum == 0: nothing to do, whatever the pointers are
    if (!d_in || !d_out) {
        set_error("wmx_agc_process: null buffer");
        return WMX_EINVAL;
    }
    const long per_pkt = (long)h->pkg * h->chn;
    if (packet_stride < per_pkt || (h->n_streams > 1 && stream_stride < per_pkt)) {
        set_error("wmx_agc_process: strides (%ld, %ld) smaller than a packet (%ld samples)", stream_stride, packet_stride, per_pkt);
        return WMX_EINVAL;
    }
    const dim3 grid((h->n_streams + 63) / 64), block(64);
    hipStream_t s = as_stream(stream);
    // one- and two-channel packets with 16-byte aligned rows -- the batched chains' cases -- go through the four-wave pipeline
    const bool pipe = h->chn <= 2 && (stream_stride % 8) == 0 && (packet_stride % 8) == 0 && (reinterpret_cast<size_t>(d_in) % 16) == 0 &&
                      (reinterpret_cast<size_t>(d_out) % 16) == 0 && !h->one_lane;
    const bool ps = h->n_off_batch != 0;  // some stream has a compression gain of its own: per-lane tables
    if (pipe) {
#define AGC_PIPE(LL, CC)                                                                                                             \
    do {                                                                                                                             \
        if (ps)                                                                                                                      \
            hipLaunchKernelGGL((agc_pipe_kernel<LL, CC, true>), grid, dim3(256), 0, s, h->d_s16, h->d_s32, h->d_table, h->d_stream_table, \
                               d_in, d_out, h->n_streams, n_packets, stream_stride, packet_stride, h->life.d_active);                \
        else                                                                                                                         \
            hipLaunchKernelGGL((agc_pipe_kernel<LL, CC, false>), grid, dim3(256), 0, s, h->d_s16, h->d_s32, h->d_table,              \
                               (const uint16_t *)nullptr, d_in, d_out, h->n_streams, n_packets, stream_stride, packet_stride,        \
                               h->life.d_active);                                                                                    \
    } while (0)
        if (h->freq == 8000) {
            if (h->chn == 1)
                AGC_PIPE(8, 1);
            else
                AGC_PIPE(8, 2);
        } else {
            if (h->chn == 1)
                AGC_PIPE(16, 1);
            else
                AGC_PIPE(16, 2);
        }
#undef AGC_PIPE
        WMX_LAUNCH_CHECK();
        return 0;
    }
#define AGC_LAUNCH(LL, CC)                                                                                                          \
    do {                                                                                                                            \
        if (ps)                                                                                                                     \
            hipLaunchKernelGGL((agc_kernel<LL, CC, true>), grid, block, 0, s, h->d_s16, h->d_s32, h->d_table, h->d_stream_table, d_in, \
                               d_out, h->n_streams, n_packets, stream_stride, packet_stride, h->chn, h->life.d_active);             \
        else                                                                                                                        \
            hipLaunchKernelGGL((agc_kernel<LL, CC, false>), grid, block, 0, s, h->d_s16, h->d_s32, h->d_table,                      \
                               (const uint16_t *)nullptr, d_in, d_out, h->n_streams, n_packets, stream_stride, packet_stride,       \
                               h->chn, h->life.d_active);                                                                           \
    } while (0)
    if (h->freq == 8000) {
        if (h->chn == 1)
            AGC_LAUNCH(8, 1);
        else if (h->chn == 2)
            AGC_LAUNCH(8, 2);
        else
            AGC_LAUNCH(8, 0);
    } else {
        if (h->chn == 1)
            AGC_LAUNCH(16, 1);
        else if (h->chn == 2)
            AGC_LAUNCH(16, 2);
        else
            AGC_LAUNCH(16, 0);
    }
#undef AGC_LAUNCH
    WMX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
