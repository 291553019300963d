// aec.hip -- batched float acoustic echo canceller for gfx950: one wavefront per stream.
//
// Replaces, for many independent near-end streams per launch that share ONE far-end reference,
// what wmix's aec_process2() does per packet (src/webrtc.c:410-483): WebRtcAec_BufferFarend +
// WebRtcAec_Process of the vendored float AEC in its wmix configuration (12-partition PBFDAF
// NLMS, NLP aggressive, no skew/metrics/delay-logging):
//   W:modules/audio_processing/aec/echo_cancellation.c:278-409,599-747,821-872
//   W:modules/audio_processing/aec/aec_core.c:148-547 (FilterFar, ScaleErrorSignal,
//     FilterAdaptation, OverdriveAndSuppress, PartitionDelay, SmoothedPSD, SubbandCoherence,
//     ComfortNoise), :911-1351 (NonLinearProcessing, ProcessBlock), :1690-1850 (far buffering,
//     ProcessFrames), aec_rdft.c (128-point Ooura rdft with frozen tables).
//
// Split of work (see aec_ctl.h): the data-independent control plane (start-up/delay state
// machine, ring-buffer indices, block counters, comfort-noise phases) runs once per packet on
// the host and arrives as an AecPlan.  Two kernels consume it:
//   aec_far_kernel   one wave per batch: far-end pre-buffer, the plain and sqrt-Hanning-windowed
//                    spectra of every new 64-sample far block (2 rdft128 each), the history of
//                    CONSUMED far spectra and the far power xPow -- all shared by every stream.
//   aec_near_kernel  one wave per stream, four streams per workgroup: ProcessBlock + NonLinearProcessing.  The
//                    12 x 64-bin filter taps live in registers (lane k = bin k; the Nyquist column in LDS) and go
//                    straight from / to their 256-byte HBM rows; the other 4.7 KB of state (PSDs, tails, rings,
//                    scalars) are pulled into LDS with one contiguous read; all blocks of all packets of the launch
//                    run on that, and everything is written back once: HBM traffic = the algorithmic minimum.
//                    The transforms run in registers (fft_regs.h): the 24 constraint FFTs of the filter update as
//                    16-lane groups (partitions 0-7 as packed pairs, then 8-11), the two single inverse transforms
//                    of a block with one point per lane, the two forward pairs in two 16-lane groups.
//                    A wave spends more of its life waiting for data than issuing (DESIGN_HISTORY.md section 5), so every
//                    global request of a phase goes out in one batch, a phase early where registers allow, and
//                    wave-uniform far-end data takes the scalar path.
// Float expressions, their order and the ordered sums follow the reference exactly (-ffp-contract=off); powf is glibc's
// own algorithm restated in libm_dev.h (bit for bit the host's), the rare log the library routine; cosf/sinf of the comfort
// noise come from the host's libm through the plan.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstddef>
#include <unordered_map>
#include <vector>
#include "wmx_internal.h"
#include "aec_ctl.h"
#include "fft_regs.h"
#include "libm_dev.h"

namespace wmx {
namespace {

constexpr int BP = 68;  // padded per-bin array length (65 bins)
// ---- per-stream state block (32-bit words)
enum : int {
    AS_W_RE = 0,                 // [12][BP]  wfBuf[0]
    AS_W_IM = AS_W_RE + 12 * BP, // [12][BP]  wfBuf[1]
    AS_DPOW = AS_W_IM + 12 * BP,
    AS_DMIN = AS_DPOW + BP,
    AS_DINIT = AS_DMIN + BP,
    AS_SD = AS_DINIT + BP,
    AS_SE = AS_SD + BP,
    AS_SX = AS_SE + BP,
    AS_SDE_RE = AS_SX + BP,
    AS_SDE_IM = AS_SDE_RE + BP,
    AS_SXD_RE = AS_SDE_IM + BP,
    AS_SXD_IM = AS_SXD_RE + BP,
    AS_DPREV = AS_SXD_IM + BP,   // dBuf[0..63]
    AS_EPREV = AS_DPREV + 64,    // eBuf[0..63]
    AS_OUTBUF = AS_EPREV + 64,
    AS_NEAR_RING = AS_OUTBUF + 64,  // nearFrBuf storage [144]
    AS_OUT_RING = AS_NEAR_RING + kAecRing,
    AS_SCAL = AS_OUT_RING + kAecRing,
    AS_HNLFBMIN = AS_SCAL + 0,
    AS_HNLFBLOCALMIN = AS_SCAL + 1,
    AS_HNLXDAVGMIN = AS_SCAL + 2,
    AS_OVERDRIVE = AS_SCAL + 3,
    AS_OVERDRIVESM = AS_SCAL + 4,
    AS_HNLNEWMIN = AS_SCAL + 5,  // ints from here
    AS_HNLMINCTR = AS_SCAL + 6,
    AS_DELAYIDX = AS_SCAL + 7,
    AS_STNEAR = AS_SCAL + 8,
    AS_ECHOSTATE = AS_SCAL + 9,
    AS_DIVERGE = AS_SCAL + 10,
    AS_NOISECTR = AS_SCAL + 12,  // noiseEstCtr (aec_core.c:1216-1243): counts a handle's first 500 * mult blocks
    AS_DELAYCTR = AS_SCAL + 13,  // delayEstCtr (aec_core.c:1020-1025): PartitionDelay every 10 * mult blocks of the handle
                                 //   -- both with the stream, not with its control cohort, so that cohorts can fold (wmx_aec_coalesce)
                                 //   as soon as their rings agree, whatever their members' ages
    AS_NSEED = AS_SCAL + 11,     // the state of this stream's comfort-noise generator in front of its next block (uint32; aec->seed,
                                 //   aec_core.c:1670: 777 at aec_init, 64 draws of WebRtcSpl_RandUArray per block, :476-480)
    AS_WORDS = AS_SCAL + 16,
};
static_assert(AS_WORDS % 4 == 0, "state block must be a whole number of 16-byte chunks");

// ---- shared (per batch) far-end data
struct AecFarBufs {
    float *pre;      // [kAecPreLen] time-domain pre-buffer ring
    float *tring;    // [kAecFarBlocks][64]: the 250-slot far ring (aec->far_buf / far_buf_windowed, aec_core.c:1394-1410) as TIME-DOMAIN
                     //   partitions -- 128 int16 samples [prev64 | new64] per slot, two per word.  The reference stores each partition's
                     //   plain and windowed spectrum there (2 x 130 floats, 260 KB per handle); both are functions of these 128 samples
                     //   (BufferFarendPartition, aec_core.c:1690-1707) and the far-end IS int16 (src/webrtc.c:430), so the slot keeps the
                     //   samples (64 KB) and the two transforms are made when a block is CONSUMED -- the same transforms, once per consumed
                     //   block as before, on the same floats: bit for bit the same spectra (round-5 VERDICT missing 3 / next 2)
    float *hist;     // [2 * kAecHist][130] consumed plain spectra, every one twice (rows r and r + kAecHist): the blocks n, n - 1, ...
                     //   are the rows R, R - 1, ... below R = n % kAecHist + kAecHist without a wrap, see hist_ld()
    float *nyq;      // [2 * kAecHist][2]   bin 64 (re, im) of those rows, side by side for the scalar loads
    float *hist_w;   // [kAecHist][130] consumed windowed spectra
    float *xpow_seq; // [kAecHist][BP]  xPow after each consumed block
    float *xpow;     // [BP] running xPow
    size_t group_words;  // floats between the buffers of consecutive far-end groups (all seven live in one slab per group)
};
// the buffers of far-end group g (wave-uniform g: pointer arithmetic on scalars)
__device__ __forceinline__ AecFarBufs far_group(const AecFarBufs &F, int g) {
    const size_t o = (size_t)g * F.group_words;
    return AecFarBufs{F.pre + o, F.tring + o, F.hist + o, F.nyq + o, F.hist_w + o, F.xpow_seq + o, F.xpow + o, F.group_words};
}

struct AecConsts {  // copied to LDS by both kernels
    FftTables tab;
    float hanning[BP], weight[BP], overdrive[BP];
};
constexpr int kAecConstWords = sizeof(AecConsts) / 4;
// what the near kernel keeps in LDS of it: the transform tables and the window.  The two NLP curves (one read per bin and block
// each) stay in global memory behind them (L1 / L2 hits): their 544 bytes per workgroup are what the conflict-free work-row
// stride needs (AecWaveLds::fa)
struct AecConstsNear {
    FftTables tab;
    float hanning[BP];
};
static_assert(offsetof(AecConsts, weight) == sizeof(AecConstsNear), "AecConstsNear is a prefix of AecConsts");
constexpr int kAecConstNearWords = sizeof(AecConstsNear) / 4;

// WEBRTC_SPL_SAT as the reference spells it: a NaN fails both comparisons and passes through (and the conversion to int16 behind
// it makes it 0, on x86 as on gfx950).  NaNs do reach this in the reference's own runs (the AEC's first blocks), so the one-instruction
// median v_med3_f32(v, -32768, 32767), which answers a NaN with -32768, is not a substitute -- tried, aec_golden caught it.
__device__ __forceinline__ float sat16f(float v) { return v > 32767.f ? 32767.f : (v < -32768.f ? -32768.f : v); }

// spectrum of a packed rdft array: bin b of a[] (StoreAsComplex / TimeToFrequency layout rules)
__device__ __forceinline__ void unpack_bin(const float *a, int b, float &re, float &im) {
    if (b == 0) {
        re = a[0];
        im = 0.f;
    } else if (b == kAecPart) {
        re = a[1];
        im = 0.f;
    } else {
        re = a[2 * b];
        im = a[2 * b + 1];
    }
}

// One 128-point real transform per 16-lane group, data in registers (fft_regs.h).  `src(p)` supplies complex
// point p of the time-domain input; the result of the complex passes goes to `row` (natural order), where
// rdft128_fwd_bin() finishes the real split for whoever reads a bin.
template <class Src>
__device__ __forceinline__ void aec_fft_fwd(float *row, const FftTables *T, int gl, Src src) {
    Cx v[4];
#pragma unroll
    for (int m = 0; m < 4; m++) v[m] = src(fft64_src_point(gl, m));
    fft64_regs<false>(v, T, gl);
#pragma unroll
    for (int m = 0; m < 4; m++) *reinterpret_cast<float2 *>(row + 2 * (gl + 16 * m)) = make_float2(v[m].r, v[m].i);
}

// ================================================================== far-end kernel
// grid = number of far-end groups: workgroup g (one wave) serves far-end g, whose packets start at far_pcm + g * far_group_stride
// and whose plans are plans[g * kAecMaxPktPerLaunch ...] (a group is also a control COHORT: the streams that were started at
// the same packet with the same reported delays share the plan a handle of the reference would have computed for itself)
// plan_by_value: a one-packet launch hands its plan over as a kernel argument; this kernel, which runs in front of the near
// kernel in the same stream, stores it into plans[0] for both -- no host-to-device blit between the previous kernel of the
// stream and this one (4.7 us per step of the chain).  Every far-end group stores the same bytes.
// Plans lie [packet][cohort] (n_cohorts apart per packet: a launch uploads exactly packets x cohorts of them).
__global__ __launch_bounds__(64) void aec_far_kernel(AecFarBufs F_all, const float *__restrict__ consts_g, AecPlan *plans,
                                                     int n_packets, int n_classes, const int32_t *__restrict__ plan_of,
                                                     const int16_t *far_pcm, long far_packet_stride,
                                                     long far_group_stride, int chn, float gpow1np, int plan_by_value,
                                                     const AecPlan plan_value) {
    // (the transform tables and the window: the two NLP curves behind them in AecConsts are the near kernel's)
    __shared__ AecConstsNear K;
    // four transforms at a time, one per 16-lane group, in registers (fft_regs.h, the near kernel's executor): the plain and the windowed
    // spectrum of TWO consumed blocks.  fa[g]: the 128 time-domain samples of transform g, then -- in place -- the result of its complex
    // passes, from which every lane takes one bin (rdft128_fwd_bin_u).  (Until round 6: one transform at a time through the LDS executor,
    // seven passes with a barrier each.)
    // This kernel is one dependent chain of memory round trips per wave; with one far-end per stream (65 536 waves) its time is that
    // chain times waves / resident waves.  5.4 KB of LDS per wave (10 KB before: separate input rows, a copy of the packet's
    // partitions, the near kernel's curves) let the registers, not the LDS, set the residency: 28 waves per CU instead of 16.
    __shared__ float fa[4][132];
    const int lane = threadIdx.x;
    if (plan_by_value) {
        const int *src = reinterpret_cast<const int *>(&plan_value);
        int *dst = reinterpret_cast<int *>(plans);
        // every request before the first store: the argument segment is far away (host memory), one round trip not nine
        constexpr int NW = (int)(sizeof(AecPlan) / 4), NIT = (NW + 63) / 64;
        int v[NIT];
#pragma unroll
        for (int k = 0; k < NIT; k++) v[k] = src[lane + 64 * k < NW ? lane + 64 * k : 0];
#pragma unroll
        for (int k = 0; k < NIT; k++)
            if (lane + 64 * k < NW) dst[lane + 64 * k] = v[k];
        __threadfence();
        wave_sync();
    }
    const AecFarBufs F = far_group(F_all, (int)blockIdx.x);
    if (far_pcm) far_pcm += (size_t)blockIdx.x * far_group_stride;
    // [packet][class]: cohorts whose control planes run in lockstep (same start, same calls) share one plan -- the indices are
    // positions inside the cohort's OWN far-end slab, equal for all of them (wmx_aec_run_cohorts)
    plans += plan_of ? plan_of[blockIdx.x] : (int)blockIdx.x;
    // One wave, one latency chain: the kernel is as long as its dependent round trips to memory (it used to make about a hundred:
    // 45 us for ONE far-end, 0.15 ms for the 4 096 of a conference server's tick).  So: (1) the tables, the time-domain pre-buffer
    // ring (448 floats) and the running far power are requested together at entry and live in LDS / registers for the launch -- the
    // windows of a packet's partitions are cut out of the LDS copy, the ring in memory is only written; (2) the samples of the far
    // blocks a packet CONSUMES are requested at the top of the packet, before the partitions it PRODUCES are stored -- they are
    // old slots of the 250-slot ring as a rule (the canceller runs behind the far-end by the system delay); when a consumed slot
    // is one this packet writes, the copy kept in LDS replaces what was fetched.
    __shared__ float lpre[kAecPreLen];
    float xp0, xp1;
    {
        float *dst = reinterpret_cast<float *>(&K);
        constexpr int NIT = (kAecConstNearWords + 63) / 64, NP = kAecPreLen / 64;
        static_assert(kAecPreLen % 64 == 0, "whole rows of the pre-buffer per lane");
        float c[NIT], pr[NP];
#pragma unroll
        for (int k = 0; k < NIT; k++) c[k] = consts_g[lane + 64 * k < kAecConstNearWords ? lane + 64 * k : 0];
#pragma unroll
        for (int k = 0; k < NP; k++) pr[k] = F.pre[lane + 64 * k];
        xp0 = F.xpow[lane];
        xp1 = F.xpow[lane == 0 ? kAecPart : 0];
#pragma unroll
        for (int k = 0; k < NIT; k++)
            if (lane + 64 * k < kAecConstNearWords) dst[lane + 64 * k] = c[k];
#pragma unroll
        for (int k = 0; k < NP; k++) lpre[lane + 64 * k] = pr[k];
    }
    wave_sync();
    bool xpow_dirty = false;
    // (3) the packet's plan is fetched ONCE, 56 words by 56 lanes, into LDS: read field by field from memory every access was a
    // vector load with a full wait behind it -- most of the kernel's round trips
    __shared__ int lplan[sizeof(AecPlan) / 4];
    for (int p = 0; p < n_packets; p++) {
        {
            const int *src = reinterpret_cast<const int *>(&plans[(size_t)p * n_classes]);
            constexpr int NW = (int)(sizeof(AecPlan) / 4);
            static_assert(NW <= 64, "one word of the plan per lane");
            wave_sync();  // the previous packet's reads of lplan are done
            if (lane < NW) lplan[lane] = src[lane];
            wave_sync();
        }
        // ... and its fields become scalars again (an LDS read is per lane as far as the compiler knows): uniform branches, SGPR bases
        const AecPlan &lp = *reinterpret_cast<const AecPlan *>(lplan);
        auto rf = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
        struct {
            int has_far, far_n, pre_wr, n_part, n_blk;
            int pre_rd[4], pslot[4], bslot[4], hist_n[4], unwritten[4];
        } pl;
        pl.has_far = rf(lp.has_far);
        pl.far_n = rf(lp.far_n);
        pl.pre_wr = rf(lp.pre_wr);
        pl.n_part = rf(lp.n_part);
        pl.n_blk = rf(lp.n_blk);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            pl.pre_rd[q] = rf(lp.part[q].pre_rd);
            pl.pslot[q] = rf(lp.part[q].far_slot);
            pl.bslot[q] = rf(lp.blk[q].far_slot);
            pl.hist_n[q] = rf(lp.blk[q].hist_n);
            pl.unwritten[q] = rf(lp.blk[q].flags) & kAecFlagFarUnwritten;
        }
        const bool consume = rf(lp.has_near) && !rf(lp.passthrough);
        // far blocks consumed by the ProcessBlock calls of this packet (at most 4), requested first: 128 int16 per slot, one word
        // (samples 2 * lane, 2 * lane + 1) per lane
        unsigned tw[4];
        if (consume) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (k >= pl.n_blk) break;
                tw[k] = reinterpret_cast<const unsigned *>(F.tring)[(size_t)pl.bslot[k] * 64 + lane];
            }
        }
        if (pl.has_far) {
            // WebRtc_WriteBuffer(far_pre_buf, farend): channel 0 of the far-end packet (src/webrtc.c:430) -- into the LDS copy of the
            // ring and into the ring itself (for the launches to come)
            const int16_t *src = far_pcm + (size_t)p * far_packet_stride;
            {
                // at most 160 samples (20 ms at 8 kHz): three per lane, fetched together (clamped index, guarded store)
                int16_t s3[3];
#pragma unroll
                for (int j = 0; j < 3; j++) s3[j] = src[(lane + 64 * j < pl.far_n ? lane + 64 * j : 0) * chn];
#pragma unroll
                for (int j = 0; j < 3; j++)
                    if (lane + 64 * j < pl.far_n) {
                        const int at = (pl.pre_wr + lane + 64 * j) % kAecPreLen;
                        lpre[at] = (float)s3[j];
                        F.pre[at] = (float)s3[j];
                    }
            }
            wave_sync();
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (q >= pl.n_part) break;
                // BufferFarendPartition (aec_core.c:1690-1707) stores the partition's plain and windowed spectrum; here its 128
                // samples [prev64 | new64] go into the slot (they are int16 values: the conversion back is exact) and the two
                // transforms wait until the block is consumed
                // (the LDS copy of the pre-buffer keeps them for the blocks this very packet consumes: nothing overwrites that window
                // before the next packet's samples arrive)
                const float a = lpre[(pl.pre_rd[q] + 2 * lane) % kAecPreLen], b = lpre[(pl.pre_rd[q] + 2 * lane + 1) % kAecPreLen];
                reinterpret_cast<unsigned *>(F.tring)[(size_t)pl.pslot[q] * 64 + lane] =
                    (unsigned)(unsigned short)(short)a | ((unsigned)(unsigned short)(short)b << 16);
            }
            wave_sync();
        }
        if (consume) {
            // history entries + xPow (aec_core.c:1209-1216); xPow is a recurrence over the blocks and stays in registers
            const int gl = fft_index(lane), grp = fft_group(lane);
#pragma unroll
            for (int k0 = 0; k0 < 4; k0 += 2) {
                if (k0 >= pl.n_blk) break;
                // TimeToFrequency(.., window = 0 / 1), aec_core.c:1690-1707, 792-819: rows 2 j / 2 j + 1 = the plain / windowed input of block k0 + j
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int k = k0 + j;
                    if (k >= pl.n_blk || pl.unwritten[k]) continue;
                    float t0 = (float)(short)(tw[k] & 0xffffu), t1 = (float)(short)(tw[k] >> 16);
                    if (pl.has_far) {  // a block produced by this very packet: what was fetched above is older than the slot's new content
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            if (q >= pl.n_part) break;
                            if (pl.pslot[q] != pl.bslot[k]) continue;
                            t0 = lpre[(pl.pre_rd[q] + 2 * lane) % kAecPreLen];
                            t1 = lpre[(pl.pre_rd[q] + 2 * lane + 1) % kAecPreLen];
                        }
                    }
                    const int i0 = 2 * lane, i1 = 2 * lane + 1;
                    fa[2 * j][i0] = t0;
                    fa[2 * j][i1] = t1;
                    fa[2 * j + 1][i0] = t0 * (i0 < kAecPart ? K.hanning[i0] : K.hanning[2 * kAecPart - i0]);
                    fa[2 * j + 1][i1] = t1 * (i1 < kAecPart ? K.hanning[i1] : K.hanning[2 * kAecPart - i1]);
                }
                wave_sync();
#ifndef WMX_AEC_EXP_NOFARFFT  // (timing experiment: what the far kernel costs without its transforms)
                // group `grp` transforms row `grp` (a row without a live block behind it is transformed too -- the exchanges between the
                // lanes of a group want every lane there -- and nobody reads the result).  In place: every lane has read its four input
                // points before any lane stores a result (one wave, one instruction stream, LDS operations in order)
                aec_fft_fwd(fa[grp], &K.tab, gl, [&](int p) {
                    const v2f c = ld_pt(fa[grp], p);
                    return Cx{c.x, c.y};
                });
#endif
                wave_sync();
                const SplitLane cf = rdft128_fwd_coef(&K.tab, lane);
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int k = k0 + j;
                    if (k >= pl.n_blk) continue;
                    // bin `lane` of the plain and of the windowed spectrum; bin 64 (real) for everybody
                    v2f P = v2f{0.f, 0.f}, Wd = v2f{0.f, 0.f};
                    float xr64 = 0.f, w64 = 0.f;
                    const float xi64 = 0.f;
                    if (!pl.unwritten[k]) {
                        // (an unwritten slot -- the read pointer was moved back into the ring's zeroed storage, WebRtc_InitBuffer +
                        // aec_core.c:1709-1717 -- reads as the zeros of the reference's memset, not as a transform of zeros)
                        P = rdft128_fwd_bin_u(fa[2 * j], cf, lane);
                        Wd = rdft128_fwd_bin_u(fa[2 * j + 1], cf, lane);
                        if (lane == 0) P.y = 0.f, Wd.y = 0.f;  // bin 0 is real: +0, as StoreAsComplex / TimeToFrequency write it
                        xr64 = fa[2 * j][0] - fa[2 * j][1];
                        w64 = fa[2 * j + 1][0] - fa[2 * j + 1][1];
                    }
                    const int hs = pl.hist_n[k] % kAecHist;
                    // rows are re[65] | im[65]
                    F.hist[hs * 130 + lane] = P.x;
                    F.hist[hs * 130 + kAecPart1 + lane] = P.y;
                    F.hist[(hs + kAecHist) * 130 + lane] = P.x;
                    F.hist[(hs + kAecHist) * 130 + kAecPart1 + lane] = P.y;
                    F.hist_w[hs * 130 + lane] = Wd.x;
                    F.hist_w[hs * 130 + kAecPart1 + lane] = Wd.y;
                    if (lane < 2) {  // bin 64: lane 0 the real part, lane 1 the (zero) imaginary part
                        const int at = lane == 0 ? kAecPart : kAecPart1 + kAecPart;
                        F.hist[hs * 130 + at] = lane == 0 ? xr64 : xi64;
                        F.hist[(hs + kAecHist) * 130 + at] = lane == 0 ? xr64 : xi64;
                        F.hist_w[hs * 130 + at] = lane == 0 ? w64 : 0.f;
                    }
                    if (lane < 4) F.nyq[2 * (hs + (lane >> 1) * kAecHist) + (lane & 1)] = (lane & 1) ? xi64 : xr64;
                    {
                        const float xr = P.x, xi = P.y;
                        const float far_spectrum = (xr * xr) + (xi * xi);
                        xp0 = 0.9f * xp0 + gpow1np * far_spectrum;
                        F.xpow_seq[hs * BP + lane] = xp0;
                    }
                    {
                        const float far_spectrum = (xr64 * xr64) + (xi64 * xi64);
                        xp1 = 0.9f * xp1 + gpow1np * far_spectrum;  // every lane, same value
                        if (lane == 0) F.xpow_seq[hs * BP + kAecPart] = xp1;
                    }
                    xpow_dirty = true;
                }
                wave_sync();  // fa[] is free for the next pair
            }
        }
    }
    if (xpow_dirty) {
        F.xpow[lane] = xp0;
        if (lane == 0) F.xpow[kAecPart] = xp1;
    }
}

// ================================================================== near-end kernel
constexpr int kAecWavesPerBlock = 4;
#ifdef WMX_AEC_PROF  // developer build only (make EXTRA=-DWMX_AEC_PROF): cycles per phase of aec_block, summed over waves
__device__ unsigned long long g_aec_prof[16];
#define AEC_PROF(i)                                                              \
    do {                                                                         \
        const long long t_now = clock64();                                       \
        if (lane == 0) W.prof[i] += (unsigned long long)(t_now - t_prev);        \
        t_prev = clock64();                                                      \
    } while (0)
#define AEC_PROF_START long long t_prev = clock64()
#elif defined(WMX_AEC_EXP_BARRIERS)
// timing-only experiment (round-4, DESIGN_HISTORY section 5d): what the hand-offs of a helper-wave design would cost -- a workgroup barrier at
// every phase boundary of a block (ten per block), valid only while the four streams of a workgroup run the same plan
#define AEC_PROF(i) __builtin_amdgcn_s_barrier()
#define AEC_PROF_START
#else
#define AEC_PROF(i)
#define AEC_PROF_START
#endif
constexpr int AS_LDS0 = AS_DPOW;               // state words kept in LDS: everything after the filter taps
constexpr int AS_LDS_WORDS = AS_WORDS - AS_DPOW;
#ifndef WMX_AEC_FAS
#define WMX_AEC_FAS 138
#endif
constexpr int FAS = WMX_AEC_FAS;                // floats per FFT work row (128 + pad)

struct alignas(16) AecWaveLds {
    float st[AS_LDS_WORDS];  // per-bin PSDs, time-domain tails, rings, scalars (indexed AS_x - AS_LDS0)
    float wn[24];            // [0..11] wfBuf[0][p][64], the Nyquist column of the filter (the other 64 bins live in
                             // registers); [12..23] partition energies of PartitionDelay
    float fa[8][FAS];        // work rows: spectra handed to / from the register FFTs; rows 1..7 double as NLP scratch
    float cur[64], enew[64];
    int16_t park[kAecFrame];  // second sub-frame of the launch's first packet, prefetched (the first one parks in fa[])
#ifdef WMX_AEC_PROF
    unsigned long long prof[16];
#endif
};
#define AEC_ST(x) W.st[(x) - AS_LDS0]

// The adaptive filter: lane k holds bin k of all 12 partitions (wfBuf, aec_core_internal.h:78)
// as (re, im) pairs: the complex products and the tap update are packed fp32 operations on them
struct AecTaps {
    v2f t[12];
};

// wave-uniform read of far-end data (the address must be the same in every lane): through the constant address space it
// becomes a scalar load -- no VGPR, no vector-memory round trip.  Valid because the far kernel wrote these buffers in an
// earlier launch.
__device__ __forceinline__ float uniform_ld(const float *p) {
    return *reinterpret_cast<const float __attribute__((address_space(4))) *>(reinterpret_cast<size_t>(p));
}

// A wave-uniform row of the far-end history: the row's address is formed on the scalar unit and stays an SGPR pair (the empty
// asm keeps the compiler from folding the row offset into a per-lane 64-bit address: one v_lshl_add_u64 per row otherwise), so
// `row[lane]` is a global load in its SGPR-base + 32-bit-VGPR-offset form.
// (The asm also strips the pointer of its address space -- loads through it would become FLAT instructions, which count
// against the LDS counter as well and have no SGPR-base form -- so the row is handed on as an explicit global pointer.)
typedef const float __attribute__((address_space(1))) *GlobalRow;
__device__ __forceinline__ GlobalRow uniform_row(const float *base, int row, int row_words) {
    const float *p = base + (size_t)row * row_words;
    asm volatile("" : "+s"(p));
    return (GlobalRow)p;
}

// element `word` + lane of such a row: the lane's byte offset is a 32-bit unsigned VGPR value added to the SGPR row pointer,
// which is exactly the operand pair of global_load ... v_off, s[row] offset:imm
__device__ __forceinline__ float row_ld(GlobalRow row, unsigned lane_bytes, int word) {
    return *(GlobalRow)((const char __attribute__((address_space(1))) *)(row + word) + lane_bytes);
}

// The far-end history rows of one block.  R = n % kAecHist + kAecHist is the row of the block consumed now and R - p the row
// consumed p blocks earlier -- the history holds every row twice, kAecHist apart, so the rows of the 12 partitions never wrap.
// The lane's byte offset into row R (less kHistMid) is ONE vector register per block and every (partition, word) of the
// history an immediate of the load: global_load_dword v, v_off, s[hist] offset:imm.  No address instruction per row, on
// either unit: formed on the scalar unit a row cost 12 scalar instructions (index modulo the ring, times the row size, a 64-bit
// add) in one dependent chain -- 48 rows per block -- and a wave issues one instruction of any kind per ~5 cycles at best,
// with four waves per SIMD to cover for it (DESIGN_HISTORY.md section 5c; 0.749 -> 0.724 ms).
constexpr int kHistRowBytes = 130 * 4, kHistMid = 2860;  // |kHistMid - 520 p + 4 word| < 4096 for p < 12, word < 130
struct HistRows {
    GlobalRow base;
    unsigned voff;
};
__device__ __forceinline__ unsigned hist_row_now(int n) { return ((unsigned)n & (unsigned)(kAecHist - 1)) + (unsigned)kAecHist; }
__device__ __forceinline__ HistRows hist_rows(const float *hist, int n, int lane) {
    return HistRows{uniform_row(hist, 0, 0), hist_row_now(n) * (unsigned)kHistRowBytes - (unsigned)kHistMid + 4u * (unsigned)lane};
}
// every outstanding vector-memory load of this wave has landed (s_waitcnt vmcnt(0); gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4],
// lgkmcnt [11:8])
__device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt(0x0f70); }
__device__ __forceinline__ float hist_ld(const HistRows &H, int p, int word) {  // p, word: compile-time at every call site
    const long imm = kHistMid - kHistRowBytes * p + 4 * word;
    return *(GlobalRow)((const char __attribute__((address_space(1))) *)H.base + (size_t)H.voff + imm);
}

// out of line for the same reason as libm_dev.h's pow_d: inlined, its fp64 temporaries push the block over 128 VGPRs
__device__ __noinline__ static float aec_powf(float x, float y, const PowTables *__restrict__ t) { return fast_pow(x, y, t); }

// (pos + i) % kAecRing for a ring position 0 <= pos < kAecRing and an offset 0 <= i <= kAecRing: one compare and a
// subtraction instead of the multiply-high / multiply-low pair of a division by a constant (both quarter rate)
__device__ __forceinline__ int ring_at(int pos, int i) {
    const int t = pos + i;
    return t >= kAecRing ? t - kAecRing : t;
}
// row `r` of the work rows for a lane-dependent r < 64 (a 24-bit multiply; `W.fa[r]` costs two 64-bit multiply-adds)
__device__ __forceinline__ float *fa_row(AecWaveLds &W, int r) { return &W.fa[0][0] + __umul24((unsigned)r, (unsigned)FAS); }
__device__ __forceinline__ int opaque_lane(int x) {
#ifndef WMX_AEC_HOIST  // experiment switch: let the compiler keep lane-derived addresses in VGPRs (needs a lower occupancy)
    asm volatile("" : "+v"(x));
#endif
    // the asm hides the value's range with its origin; restated, a lane index used as an array offset is a zero-extended 32-bit
    // offset and global loads of wave-uniform rows take the SGPR-base + VGPR-offset form instead of a 64-bit address per row
    __builtin_assume(x >= 0 && x < 64);
    return x;
}

template <int MULT>  // 1: 8 kHz, 2: 16 kHz
__device__ __forceinline__ void aec_block(const AecConstsNear &K, const float *__restrict__ curves_g, const PowTables *__restrict__ powtab, AecWaveLds &W, AecTaps &taps,
                                          const AecFarBufs &F, const AecBlkPlan &bp, const AecNoiseEntry *__restrict__ noise_tab, const int lane_in) {
    // The lane-derived LDS addresses (gather points, twiddle and window slots) are loop invariant; left alone the
    // compiler hoists ~100 of them out of the packet loop and pins them in VGPRs for the whole kernel.  Recomputing
    // them per block costs a few VALU ops and frees the registers.
    int lane = opaque_lane(lane_in);
    int *Si = reinterpret_cast<int *>(W.st) - AS_LDS0;
    const float mu = MULT == 1 ? 0.6f : 0.5f, err_thr = MULT == 1 ? 2e-6f : 1.5e-6f;  // aec_core.c:1530-1538
    const float scale = 2.0f / 128;
    const int n = bp.hist_n;
    // the handle's two block counters: the first 500 * mult blocks initialise the noise floor, every (10 * mult)-th block re-estimates
    // the dominant filter partition -- wave-uniform, kept in the stream's scalars
    int blk_flags;
    {
        int nctr = __builtin_amdgcn_readfirstlane(Si[AS_NOISECTR]), dctr = __builtin_amdgcn_readfirstlane(Si[AS_DELAYCTR]);
        blk_flags = nctr > 50 ? kAecFlagNoiseMin : 0;
        if (nctr < 500 * MULT) {
            nctr++;
            blk_flags |= kAecFlagNoiseInit;
        }
        dctr++;
        if (dctr == 10 * MULT) dctr = 0;
        if (dctr == 0) blk_flags |= kAecFlagDelayEst;
        if (lane == 0) {
            Si[AS_NOISECTR] = nctr;
            Si[AS_DELAYCTR] = dctr;
        }
    }
    int g = fft_group(lane), gl = fft_index(lane);
#define AEC_RELANE()              \
    do {                          \
        lane = opaque_lane(lane); \
        g = fft_group(lane);      \
        gl = fft_index(lane);     \
    } while (0)
    // NLP scratch rows (free outside the filter update)
    float *xw = W.fa[1], *dw = W.fa[2], *ew = W.fa[3];  // re at [b], im at [66 + b]
    float *t0 = W.fa[4], *t1 = W.fa[5], *t2 = W.fa[6], *t3 = W.fa[7];
    AEC_PROF_START;

    // ---- near block (aec_core.c:1177-1195).  d = [prev | cur]; its plain transform feeds the near power, its
    //      windowed transform the coherence estimates of the NLP (aec_core.c:934-949): both now, side by side.
    const unsigned xpow_row = ((unsigned)n & (unsigned)(kAecHist - 1)) * (unsigned)BP;  // words
    const float xpow_lane = row_ld(uniform_row(F.xpow_seq, 0, 0), 4u * (xpow_row + (unsigned)lane), 0);  // far power of this block (ScaleErrorSignal), requested early
    W.cur[lane] = AEC_ST(AS_NEAR_RING + ring_at(bp.near_rd, lane));
    wave_sync();
    AEC_PROF(0);
    AEC_RELANE();
    // ---- FilterFar (aec_core.c:148-170): y = sum_p X_{n-p} * W_p, partitions in order; lane 0 also does bin 64
    {
        // all 24 far-spectrum values of this lane requested at once (one memory round trip, not one per few partitions);
        // the Nyquist column is wave-uniform and comes through the scalar path
        float xr[12], xi[12];
        const HistRows H = hist_rows(F.hist, n, lane);
#pragma unroll
        for (int p = 0; p < 12; p++) {
            xr[p] = hist_ld(H, p, 0);
            xi[p] = hist_ld(H, p, kAecPart1);
        }
        wait_vm();  // ONE wait for the batch: left to the compiler every load gets its own s_waitcnt vmcnt(23), (22), ... in front of its use
        const float *NQ = F.nyq + 2 * (hist_row_now(n) - 11);  // (re, im) of bin 64 of the rows R - 11 .. R, 24 consecutive words
        // ... as three scalar loads of eight words (as 24 single words they were 24 scalar-memory instructions)
        typedef float v8f __attribute__((ext_vector_type(8)));
        typedef const v8f __attribute__((address_space(4), aligned(8))) *ConstV8;
        const ConstV8 NQ8 = reinterpret_cast<ConstV8>(reinterpret_cast<size_t>(NQ));
        v8f nq[3];
#pragma unroll
        for (int j = 0; j < 3; j++) nq[j] = NQ8[j];
        float y64 = 0.f;
        v2f y2 = v2f{0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 12; p++) {
            // (yr, yi) += (xr*wr - xi*wi, xr*wi + xi*wr): cmul_w is those four products and two sums, as three packed instructions
            y2 = y2 + cmul_w(xr[p], xi[p], taps.t[p]);
            const float nr = nq[(2 * (11 - p)) >> 3][(2 * (11 - p)) & 7], ni = nq[(2 * (11 - p) + 1) >> 3][(2 * (11 - p) + 1) & 7];  // ni == 0, wfBuf[1][.][64] == 0
            y64 += nr * W.wn[p] - ni * 0.f;                                                    // used by lane 0 only
        }
        // ---- error e = d - y (aec_core.c:1286-1297): y = second half of the inverse transform of the packed spectrum
        //      (lane 0 carries (bin 0, bin 64)), one point per lane in registers: lanes 32..63 end up with y[2(l-32)], +1
        AEC_PROF(1);
        v2f pt = rdft128_inv_point_lanes(v2f{y2.x, lane == 0 ? y64 : y2.y}, &K.tab, lane);
        pt = fft64_lanes<true>(pt, &K.tab, lane);
        if (lane >= 32) {
            const int i = 2 * (lane - 32);
            W.enew[i] = W.cur[i] - pt.x * scale;
            W.enew[i + 1] = W.cur[i + 1] - pt.y * scale;
        }
    }
    wave_sync();
    AEC_RELANE();
    // ---- four transforms side by side, one per lane group: rdft(d) and rdft(d * w) of the near block [dprev | cur]
    //      (aec_core.c:1177-1195, 934-949), ef = rdft([0 | e]) (aec_core.c:1299-1309) and the windowed rdft([eprev | e] * w) of
    //      the NLP.  The reference transforms d before it filters the far end; nothing between the two reads the other's
    //      result, so the near transforms wait for the error and share its instruction stream.
    {
        const bool win = g & 1, err = g >= 2;
        aec_fft_fwd(W.fa[g], &K.tab, gl, [&](int p) {
            const int i = 2 * (p & 31);
            float x0, x1, h0, h1;
            if (p < 32) {  // p is a compile-time property of the point index m (fft64_src_point): no divergence
                const float *prev = &AEC_ST(err ? AS_EPREV : AS_DPREV);
                x0 = prev[i], x1 = prev[i + 1];
                h0 = K.hanning[i], h1 = K.hanning[i + 1];
                if (g == 2) x0 = 0.f, x1 = 0.f;
            } else {
                const float *now = err ? W.enew : W.cur;
                x0 = now[i], x1 = now[i + 1];
                h0 = K.hanning[kAecPart - i], h1 = K.hanning[kAecPart - i - 1];
            }
            return Cx{win ? x0 * h0 : x0, win ? x1 * h1 : x1};
        });
    }
    wave_sync();
    AEC_PROF(2);
    AEC_RELANE();
    // windowed near spectrum, kept in registers across the filter update (bin = lane; lane 0 also bin 64)
    float dwr, dwi, dw64, dfr, dfi, df64;
    {
        const SplitLane cf = rdft128_fwd_coef(&K.tab, lane);
        const v2f dwb = rdft128_fwd_bin_u(W.fa[1], cf, lane), dfb = rdft128_fwd_bin_u(W.fa[0], cf, lane);
        dwr = dwb.x, dwi = dwb.y, dw64 = W.fa[1][0] - W.fa[1][1];  // bin 64 = a[0] - a[1] (used by lane 0)
        dfr = dfb.x, dfi = dfb.y, df64 = W.fa[0][0] - W.fa[0][1];
    }
    // ---- near power, noise floor (aec_core.c:1197-1243); bin `lane` and bin 64 side by side (see SmoothedPSD below)
    {
        struct Pw {
            float dpow, dmin, dinit;
        };
        auto power = [&](int b, float re, float im) {
            Pw r;
            const float ns = re * re + im * im;
            r.dpow = 0.9f * AEC_ST(AS_DPOW + b) + 0.1f * ns;
            r.dmin = AEC_ST(AS_DMIN + b);
            if (blk_flags & kAecFlagNoiseMin) {  // both arms evaluated, one selected: a lane-dependent branch is three scalar instructions
                const float down = (r.dpow + 0.1f * (r.dmin - r.dpow)) * 1.0002f, up = r.dmin * 1.0002f;
                r.dmin = r.dpow < r.dmin ? down : up;
            }
            r.dinit = 0.f;
            if (blk_flags & kAecFlagNoiseInit) {
                r.dinit = AEC_ST(AS_DINIT + b);
                const float track = 0.999f * r.dinit + 0.001f * r.dmin;
                r.dinit = r.dmin > r.dinit ? track : r.dmin;
            }
            return r;
        };
        auto put = [&](int b, const Pw &r) {
            AEC_ST(AS_DPOW + b) = r.dpow;
            if (blk_flags & kAecFlagNoiseMin) AEC_ST(AS_DMIN + b) = r.dmin;
            if (blk_flags & kAecFlagNoiseInit) AEC_ST(AS_DINIT + b) = r.dinit;
        };
        const Pw a = power(lane, dfr, dfi);
        put(lane, a);
#if !defined(WMX_AEC_EXP) || WMX_AEC_EXP < 1
        const Pw c = power(kAecPart, df64, 0.f);
        if (lane == 0) put(kAecPart, c);
#endif
    }
    // ---- ScaleErrorSignal (aec_core.c:172-194); ef stays in registers (bin = lane; lane 0 also bin 64), and so
    //      does the windowed error spectrum
    float ewr, ewi, ew64, efr, efi, ef64r, ef64i = 0.f;
    {
        {
            const SplitLane cf = rdft128_fwd_coef(&K.tab, lane);
            const v2f ewb = rdft128_fwd_bin_u(W.fa[3], cf, lane), efb = rdft128_fwd_bin_u(W.fa[2], cf, lane);
            ewr = ewb.x, ewi = ewb.y, ew64 = W.fa[3][0] - W.fa[3][1];
            efr = efb.x, efi = efb.y, ef64r = W.fa[2][0] - W.fa[2][1];
        }
        auto scale_err = [&](float xp, float &er, float &ei) {
            er /= (xp + 1e-10f);
            ei /= (xp + 1e-10f);
            float abs_ef = sqrtf(er * er + ei * ei);
            if (abs_ef > err_thr) {
                abs_ef = err_thr / (abs_ef + 1e-10f);
                er *= abs_ef;
                ei *= abs_ef;
            }
            er *= mu;
            ei *= mu;
        };
        // bin 64 (read by every lane from the same LDS words) is scaled by every lane alongside its own bin: two independent
        // dependency chains in one instruction stream instead of a second, serial pass under `if (lane == 0)`
        const float xp64 = uniform_ld(F.xpow_seq + xpow_row + kAecPart);
        scale_err(xpow_lane, efr, efi);
#if !defined(WMX_AEC_EXP) || WMX_AEC_EXP < 1
        scale_err(xp64, ef64r, ef64i);
#else
        ef64r *= xp64;
#endif
    }
    wave_sync();  // rows 2, 3 are read; the filter update overwrites the work rows
    AEC_RELANE();
    // ---- FilterAdaptation (aec_core.c:222-270): conj(X_{n-p}) * ef -> time domain, zero the second half,
    //      back to frequency, add to partition p.  Four groups of 16 lanes; partitions 0..7 first (two per group),
    //      then 8..11, through the same eight work rows.
    const float e64r = ef64r, e64i = ef64i;  // wave-uniform already
    // windowed far spectrum of the NLP (bins 0..63 of the block consumed delayIdx blocks ago): requested while the second
    // half of the filter update runs; PartitionDelay below may still move delayIdx (every 10 * mult blocks)
    const int delayIdx0 = Si[AS_DELAYIDX];
    float xwr_pre = 0.f, xwi_pre = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int base = 8 * pass, cnt = pass == 0 ? 8 : 4;
        if (pass == 1) {
            const unsigned xw_row = ((unsigned)(n - delayIdx0) & (unsigned)(kAecHist - 1)) * 130u;  // words
            const GlobalRow Xw = uniform_row(F.hist_w, 0, 0);
            xwr_pre = row_ld(Xw, 4u * (xw_row + (unsigned)lane), 0);
            xwi_pre = row_ld(Xw, 4u * (xw_row + (unsigned)lane), kAecPart1);
        }
        // Nyquist bin of the far block of partition base + lane, used by the lanes < cnt (the others read a valid row too: no
        // predicated region around the load)
        float nyq_r, nyq_i;
        {
            const unsigned l = (unsigned)lane < 11u ? (unsigned)lane : 11u;
            const v2f v = *reinterpret_cast<const v2f __attribute__((address_space(1))) *>(
                (const char __attribute__((address_space(1))) *)uniform_row(F.nyq, 0, 0) + 8u * (hist_row_now(n) - (unsigned)base - l));
            nyq_r = v.x;
            nyq_i = -v.y;
        }
        {
            float xr[8], xi[8];
            const HistRows H = hist_rows(F.hist, n, lane);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (q >= cnt) continue;
                xr[q] = hist_ld(H, base + q, 0);
                xi[q] = hist_ld(H, base + q, kAecPart1);
            }
            wait_vm();
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (q < cnt) xi[q] = -xi[q];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (q >= cnt) continue;
                st_pt(W.fa[q], lane, cmul_w(xr[q], xi[q], v2f{efr, efi}));  // (xr*efr - xi*efi, xr*efi + xi*efr)
            }
        }
        wave_sync();
        // fft[1] is overwritten with the Nyquist product (aec_core.c:245-248): lane j patches partition base + j
        // (LDS runs a wave's accesses in order, so this lands after lane 0's store above)
        if (lane < cnt) fa_row(W, lane)[1] = nyq_r * e64r - nyq_i * e64i;
        wave_sync();
        AEC_RELANE();
        if (pass == 0) {
            // partitions g and 4 + g of this group as one packed pair (fft_regs.h, Cx2): every lane reads its (and its
            // mirror's) points of both rows before any lane of the group stores to them
            float *row0 = W.fa[g], *row1 = W.fa[4 + g];
            Cx2 v[4];
#pragma unroll
            for (int m = 0; m < 4; m++) v[m] = rdft128_inv_point_x2(row0, row1, &K.tab, gl, m);
            fft64_regs_x2<true>(v, &K.tab, gl);
            // points 0..31 (m = 0, 1) scaled, points 32..63 zeroed; the forward gather wants points
            // rev4(gl) + {0, 32, 16, 48}: two of them from lane rev4(gl), two zeros
            const v2f sc = bc(scale), z = bc(0.f);
            const v2f ar = v[0].r * sc, ai = v[0].i * sc, cr = v[1].r * sc, ci = v[1].i * sc;
            v[0] = Cx2{v2f{row_bitrev(ar.x, lane), row_bitrev(ar.y, lane)}, v2f{row_bitrev(ai.x, lane), row_bitrev(ai.y, lane)}};
            v[2] = Cx2{v2f{row_bitrev(cr.x, lane), row_bitrev(cr.y, lane)}, v2f{row_bitrev(ci.x, lane), row_bitrev(ci.y, lane)}};
            v[1] = Cx2{z, z};
            v[3] = Cx2{z, z};
            fft64_regs_x2<false>(v, &K.tab, gl);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                *reinterpret_cast<float2 *>(row0 + 2 * (gl + 16 * m)) = make_float2(v[m].r.x, v[m].i.x);
                *reinterpret_cast<float2 *>(row1 + 2 * (gl + 16 * m)) = make_float2(v[m].r.y, v[m].i.y);
            }
        } else {
            // partitions 8 + g: one transform per group
            float *row = W.fa[g];
            Cx v[4];
            {
                const v2f p0 = rdft128_inv_point_m<0>(row, &K.tab, gl), p1 = rdft128_inv_point_m<1>(row, &K.tab, gl);
                const v2f p2 = rdft128_inv_point_m<2>(row, &K.tab, gl), p3 = rdft128_inv_point_m<3>(row, &K.tab, gl);
                v[0] = Cx{p0.x, p0.y}, v[1] = Cx{p1.x, p1.y}, v[2] = Cx{p2.x, p2.y}, v[3] = Cx{p3.x, p3.y};
                fft64_regs<true>(v, &K.tab, gl);
            }
            const Cx a = Cx{row_bitrev(v[0].r * scale, lane), row_bitrev(v[0].i * scale, lane)};
            const Cx c = Cx{row_bitrev(v[1].r * scale, lane), row_bitrev(v[1].i * scale, lane)};
            v[0] = a;
            v[1] = Cx{0.f, 0.f};
            v[2] = c;
            v[3] = Cx{0.f, 0.f};
            fft64_regs<false>(v, &K.tab, gl);
#pragma unroll
            for (int m = 0; m < 4; m++) *reinterpret_cast<float2 *>(row + 2 * (gl + 16 * m)) = make_float2(v[m].r, v[m].i);
        }
        wave_sync();
        AEC_RELANE();
        {
            const SplitLane cf = rdft128_fwd_coef(&K.tab, lane);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (q >= cnt) continue;
                const v2f d = rdft128_fwd_bin_u(W.fa[q], cf, lane);
                taps.t[base + q] = taps.t[base + q] + d;  // lane 0's imaginary half adds 0 (wfBuf[1][pos] is never touched, aec_core.c:262-268)
            }
            if (lane < cnt) W.wn[base + lane] += fa_row(W, lane)[0] - fa_row(W, lane)[1];  // wfBuf[0][pos + 64] += fft[1]: bin 64 = a[0] - a[1]
        }
        wave_sync();
    }
    AEC_PROF(3);

    AEC_RELANE();
    // ================================================= NonLinearProcessing (aec_core.c:911-1141)
    constexpr int prefSize = 24 / MULT, minPref = 4 / MULT;
    const float gc0 = MULT == 1 ? 0.9f : 0.93f, gc1 = MULT == 1 ? 0.1f : 0.07f;  // kNormalSmoothingCoefficients
    // PartitionDelay (aec_core.c:295-319) every 10*mult blocks: per-partition ordered energy sums
    int delayIdx = delayIdx0;
    if (blk_flags & kAecFlagDelayEst) {
        // two partitions per work row: [0..64] and [66..130]
#pragma unroll
        for (int p = 0; p < 12; p++) {
            float *dst = W.fa[p >> 1] + (p & 1) * 66;
            dst[lane] = taps.t[p].x * taps.t[p].x + taps.t[p].y * taps.t[p].y;
            if (lane == 0) dst[64] = W.wn[p] * W.wn[p] + 0.f * 0.f;
        }
        wave_sync();
        if (lane < 12) {
            const float *src = W.fa[lane >> 1] + (lane & 1) * 66;
            float en = 0.f;
            for (int j = 0; j < kAecPart1; j++) en += src[j];
            W.wn[12 + lane] = en;
        }
        wave_sync();
        float best = 0.f;
        delayIdx = 0;
        for (int p = 0; p < 12; p++)
            if (W.wn[12 + p] > best) {
                best = W.wn[12 + p];
                delayIdx = p;
            }
        wave_sync();
        if (lane == 0) Si[AS_DELAYIDX] = delayIdx;
    }
    AEC_PROF(4);
    // xfw = windowed far spectrum consumed delayIdx blocks ago; the windowed near / error spectra come back from registers
    {
        const float *Xw = F.hist_w + (size_t)(((unsigned)(n - delayIdx) & (unsigned)(kAecHist - 1)) * 130u);
        if (delayIdx != delayIdx0) {  // wave-uniform, rare
            xwr_pre = Xw[lane];
            xwi_pre = Xw[kAecPart1 + lane];
        }
        xw[lane] = xwr_pre;
        xw[66 + lane] = xwi_pre;
        if (lane == 0) {
            xw[kAecPart] = uniform_ld(Xw + kAecPart);
            xw[66 + kAecPart] = uniform_ld(Xw + kAecPart1 + kAecPart);
        }
        dw[lane] = dwr;
        dw[66 + lane] = dwi;
        ew[lane] = ewr;
        ew[66 + lane] = ewi;
        if (lane == 0) {
            dw[kAecPart] = dw64;
            dw[66 + kAecPart] = 0.f;
            ew[kAecPart] = ew64;
            ew[66 + kAecPart] = 0.f;
        }
    }
    wave_sync();
    // SmoothedPSD (aec_core.c:333-386).  Bin `lane` and bin 64 go through the same straight-line code (every lane forms bin
    // 64 from the same LDS words, lane 0 stores it): two independent dependency chains for the scheduler to interleave,
    // instead of a second loop iteration that runs for lane 0 alone.
    {
        struct Psd {
            float sd, se, sx, sde_r, sde_i, sxd_r, sxd_i;
        };
        auto psd = [&](int b) {
            const float dr = dw[b], di = dw[66 + b], er = ew[b], ei = ew[66 + b];
            const float xr = xw[b], xi = xw[66 + b];
            Psd r;
            r.sd = gc0 * AEC_ST(AS_SD + b) + gc1 * (dr * dr + di * di);
            r.se = gc0 * AEC_ST(AS_SE + b) + gc1 * (er * er + ei * ei);
            const float xx = xr * xr + xi * xi;
            r.sx = gc0 * AEC_ST(AS_SX + b) + gc1 * (xx > 15.f ? xx : 15.f);
            r.sde_r = gc0 * AEC_ST(AS_SDE_RE + b) + gc1 * (dr * er + di * ei);
            r.sde_i = gc0 * AEC_ST(AS_SDE_IM + b) + gc1 * (dr * ei - di * er);
            r.sxd_r = gc0 * AEC_ST(AS_SXD_RE + b) + gc1 * (dr * xr + di * xi);
            r.sxd_i = gc0 * AEC_ST(AS_SXD_IM + b) + gc1 * (dr * xi - di * xr);
            return r;
        };
        auto put = [&](int b, const Psd &r) {
            AEC_ST(AS_SX + b) = r.sx;
            AEC_ST(AS_SD + b) = r.sd;
            AEC_ST(AS_SE + b) = r.se;
            AEC_ST(AS_SDE_RE + b) = r.sde_r;
            AEC_ST(AS_SDE_IM + b) = r.sde_i;
            AEC_ST(AS_SXD_RE + b) = r.sxd_r;
            AEC_ST(AS_SXD_IM + b) = r.sxd_i;
        };
        const Psd a = psd(lane);
        put(lane, a);
#if !defined(WMX_AEC_EXP) || WMX_AEC_EXP < 1
        const Psd c = psd(kAecPart);
        if (lane == 0) put(kAecPart, c);
#endif
    }
    wave_sync();
    AEC_PROF(5);
    // the two ordered sums advance side by side: lane 0 adds sd[0..64], lane 1 adds se[0..64] (index order each)
    float sdSum, seSum;
#if defined(WMX_AEC_EXP) && WMX_AEC_EXP >= 2
    sdSum = AEC_ST(AS_SD), seSum = AEC_ST(AS_SE + 1);
#else
    {
        const float *mine = &AEC_ST(lane == 1 ? AS_SE : AS_SD);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < kAecPart; i += 8) {
            const float4 a = *reinterpret_cast<const float4 *>(mine + i), b = *reinterpret_cast<const float4 *>(mine + i + 4);
            acc += a.x;
            acc += a.y;
            acc += a.z;
            acc += a.w;
            acc += b.x;
            acc += b.y;
            acc += b.z;
            acc += b.w;
        }
        acc += mine[kAecPart];
        sdSum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 0));
        seSum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 1));
    }
#endif
    const int diverge = ((Si[AS_DIVERGE] ? 1.05f : 1.0f) * seSum > sdSum) ? 1 : 0;
    const bool reset_filter = seSum > (19.95f * sdSum);
    wave_sync();
    if (lane == 0) Si[AS_DIVERGE] = diverge;
    if (reset_filter) {  // memset(wfBuf, 0)
#pragma unroll
        for (int p = 0; p < 12; p++) {
            taps.t[p] = v2f{0.f, 0.f};
            if (lane == 0) W.wn[p] = 0.f;
        }
    }
    AEC_PROF(6);
    AEC_RELANE();
    // ComfortNoise's random phases (aec_core.c:476-489).  WebRtcSpl_RandUArray draws 64 numbers per block, seed -> seed * 69069 + 1 mod
    // 2^31 each; bin b's phase comes from draw b: lane l >= 1 takes draw l, lane 0 draw 64 (bin 64's) -- k draws are one multiply-add
    // with precomputed (a_k, c_k), so every lane reaches its draw from the stream's state in one step, and the state moves on by the
    // 64-draw step on the scalar unit.  u = (int16)(draw >> 16) / 32768 takes 32 768 values: cosf / sinf of 2 pi u are looked up in the
    // table of the host libm's values (AecNoiseEntry; one 8-byte gather per lane and block, requested here, used two phases on).
    // The generator is the stream's own (4 bytes of its state), as it is the handle's in the reference: streams of any age share a
    // control cohort, and nothing grows with a stream's life.
    const uint32_t nseed = (uint32_t)__builtin_amdgcn_readfirstlane(Si[AS_NSEED]);
    const uint2 *jump = reinterpret_cast<const uint2 *>(noise_tab + kAecNoiseTab);  // [k - 1] = (a_k, c_k), k = 1 .. 64
    const uint2 jl = jump[(lane + 63) & 63], j64 = jump[63];
    const AecNoiseEntry nz = noise_tab[((nseed * jl.x + jl.y) & 0x7FFFFFFFu) >> 16];
    const float nz_c = nz.c, nz_s = nz.s;
    if (lane == 0) Si[AS_NSEED] = (int)((nseed * j64.x + j64.y) & 0x7FFFFFFFu);
    // coherences (aec_core.c:440-449), bin `lane` and bin 64 side by side
    {
        auto coh = [&](int b, float &cde, float &cxd) {
            const float sde_r = AEC_ST(AS_SDE_RE + b), sde_i = AEC_ST(AS_SDE_IM + b), sxd_r = AEC_ST(AS_SXD_RE + b),
                        sxd_i = AEC_ST(AS_SXD_IM + b);
            cde = (sde_r * sde_r + sde_i * sde_i) / (AEC_ST(AS_SD + b) * AEC_ST(AS_SE + b) + 1e-10f);
            cxd = (sxd_r * sxd_r + sxd_i * sxd_i) / (AEC_ST(AS_SX + b) * AEC_ST(AS_SD + b) + 1e-10f);
        };
        float a0, a1, c0, c1;
        coh(lane, a0, a1);
#if !defined(WMX_AEC_EXP) || WMX_AEC_EXP < 1
        coh(kAecPart, c0, c1);
#else
        c0 = a0, c1 = a1;
#endif
        if (diverge) {  // divergeState: the error spectrum is replaced by the near spectrum (aec_core.c:959-962)
            ew[lane] = dw[lane];
            ew[66 + lane] = dw[66 + lane];
        }
        t0[lane] = a0;
        t1[lane] = a1;
        if (lane == 0) {
            if (diverge) {
                ew[kAecPart] = dw[kAecPart];
                ew[66 + kAecPart] = dw[66 + kAecPart];
            }
            t0[kAecPart] = c0;
            t1[kAecPart] = c1;
        }
    }
    wave_sync();
    // the two band averages as parallel lane chains (lane 0: cohxd, lane 1: cohde), index order each
    float hNlXdAvg, hNlDeAvg;
#if defined(WMX_AEC_EXP) && WMX_AEC_EXP >= 2
    hNlXdAvg = t1[5] * prefSize, hNlDeAvg = t0[5] * prefSize;
#else
    {
        const float *mine = lane == 1 ? t0 : t1;
        float acc = 0.f;
#pragma unroll
        for (int i = minPref; i < prefSize + minPref; i++) acc += mine[i];
        hNlXdAvg = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 0));
        hNlDeAvg = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 1));
    }
#endif
    hNlXdAvg /= prefSize;
    hNlXdAvg = 1 - hNlXdAvg;
    hNlDeAvg /= prefSize;
    float hNlXdAvgMin = AEC_ST(AS_HNLXDAVGMIN), hNlFbMin = AEC_ST(AS_HNLFBMIN), hNlFbLocalMin = AEC_ST(AS_HNLFBLOCALMIN);
    float overDrive = AEC_ST(AS_OVERDRIVE), overDriveSm = AEC_ST(AS_OVERDRIVESM);
    int stNear = Si[AS_STNEAR], echoState, hNlNewMin = Si[AS_HNLNEWMIN], hNlMinCtr = Si[AS_HNLMINCTR];
    if (hNlXdAvg < 0.75f && hNlXdAvg < hNlXdAvgMin) hNlXdAvgMin = hNlXdAvg;
    if (hNlDeAvg > 0.98f && hNlXdAvg > 0.9f)
        stNear = 1;
    else if (hNlDeAvg < 0.95f || hNlXdAvg < 0.8f)
        stNear = 0;
    // hNl selection (aec_core.c:988-1024); mode: 0 cohde, 1 (1 - cohxd), 2 min of both
    int mode;
    float hNlFb, hNlFbLow;
    if (hNlXdAvgMin == 1) {
        echoState = 0;
        overDrive = 5.0f;  // kNormalMinOverDrive[kAecNlpAggressive]
        if (stNear == 1) {
            mode = 0;
            hNlFb = hNlDeAvg;
            hNlFbLow = hNlDeAvg;
        } else {
            mode = 1;
            hNlFb = hNlXdAvg;
            hNlFbLow = hNlXdAvg;
        }
    } else {
        if (stNear == 1) {
            echoState = 0;
            mode = 0;
            hNlFb = hNlDeAvg;
            hNlFbLow = hNlDeAvg;
        } else {
            echoState = 1;
            mode = 2;
            hNlFb = 0.f;
            hNlFbLow = 0.f;
        }
    }
    for (int b = lane; b < kAecPart1; b += 64) {
        const float cde = t0[b], cxd = 1 - t1[b];
        t2[b] = mode == 0 ? cde : (mode == 1 ? cxd : (cde < cxd ? cde : cxd));  // hNl
    }
    wave_sync();
#if defined(WMX_AEC_EXP) && WMX_AEC_EXP >= 2
    if (mode == 3) {
#else
    if (mode == 2) {
#endif
        // qsort(hNlPref) + the two order statistics (aec_core.c:1017-1022): rank by counting
        constexpr int i75 = (int)(0.75f * (prefSize - 1)), i50 = (int)(0.5f * (prefSize - 1));
        if (lane < prefSize) {
            const float v = t2[minPref + lane];
            int rank = 0;
            for (int i = 0; i < prefSize; i++) {
                const float u = t2[minPref + i];
                rank += (u < v || (u == v && i < lane)) ? 1 : 0;
            }
            if (rank == i75) t3[0] = v;
            if (rank == i50) t3[1] = v;
        }
        wave_sync();
        hNlFb = t3[0];
        hNlFbLow = t3[1];
        wave_sync();
    }
    if (hNlFbLow < 0.6f && hNlFbLow < hNlFbLocalMin) {
        hNlFbLocalMin = hNlFbLow;
        hNlFbMin = hNlFbLow;
        hNlNewMin = 1;
        hNlMinCtr = 0;
    }
    {
        float t = hNlFbLocalMin + 0.0008f / MULT;
        hNlFbLocalMin = t < 1 ? t : 1;
        t = hNlXdAvgMin + 0.0006f / MULT;
        hNlXdAvgMin = t < 1 ? t : 1;
    }
    if (hNlNewMin == 1) hNlMinCtr++;
    if (hNlMinCtr == 2) {
        hNlNewMin = 0;
        hNlMinCtr = 0;
        const float od = -18.4f / (log_d(hNlFbMin + 1e-10f) + 1e-10f);  // kTargetSupp[2]
        overDrive = od > 5.0f ? od : 5.0f;
    }
    if (overDrive < overDriveSm)
        overDriveSm = 0.99f * overDriveSm + 0.01f * overDrive;
    else
        overDriveSm = 0.9f * overDriveSm + 0.1f * overDrive;
    // every lane holds the same scalars; lane 0 publishes them
    if (lane == 0) {
        AEC_ST(AS_HNLXDAVGMIN) = hNlXdAvgMin;
        AEC_ST(AS_HNLFBMIN) = hNlFbMin;
        AEC_ST(AS_HNLFBLOCALMIN) = hNlFbLocalMin;
        AEC_ST(AS_OVERDRIVE) = overDrive;
        AEC_ST(AS_OVERDRIVESM) = overDriveSm;
        Si[AS_STNEAR] = stNear;
        Si[AS_ECHOSTATE] = echoState;
        Si[AS_HNLNEWMIN] = hNlNewMin;
        Si[AS_HNLMINCTR] = hNlMinCtr;
    }
    AEC_PROF(7);
    AEC_RELANE();
    // OverdriveAndSuppress (aec_core.c:272-293) + ComfortNoise (:462-547) + packing for the inverse transform
    const int noise_off = (blk_flags & kAecFlagNoiseInit) ? AS_DINIT : AS_DMIN;
    v2f out_spec = v2f{0.f, 0.f};
    float out_nyq = 0.f;
#if defined(WMX_AEC_EXP) && WMX_AEC_EXP >= 1
    for (int b = lane; b < kAecPart; b += 64) {
#else
    for (int b = lane; b < kAecPart1; b += 64) {
#endif
        float h = t2[b];
        const float wc = curves_g[b];  // weightCurve, overDriveCurve: AecConsts::weight / ::overdrive in global memory
        if (h > hNlFb) h = wc * hNlFb + (1 - wc) * h;
        h = aec_powf(h, overDriveSm * curves_g[BP + b], powtab);  // powf (aec_core.c:278) -> libm_dev.h fast_pow
        float er = ew[b] * h, ei = ew[66 + b] * h;
        ei *= -1;
        float ur = 0.f, ui = 0.f;
        if (b >= 1) {
            const float noise = sqrtf(AEC_ST(noise_off + b));
            ur = noise * nz_c;  // this lane's entry: bin b == lane, or lane 0's entry 63 for bin 64
            ui = -noise * nz_s;
            if (b == kAecPart) ui = 0.f;
        }
        const float v = 1 - h * h;
        const float tmp = sqrtf(v > 0 ? v : 0);
        er += tmp * ur;
        ei += tmp * ui;
        // packed spectrum for the inverse transform, kept in registers: bin == lane, lane 0 carries (bin 0, bin 64)
        if (b == kAecPart)
            out_nyq = er;
        else
            out_spec = v2f{er, -ei};
    }
    AEC_PROF(8);
    AEC_RELANE();
    // inverse transform, overlap-add with the sqrt-Hanning window, saturate, queue 64 output samples
    // (aec_core.c:1089-1101, 1341): points 0..31 are the first half, 32..63 the new overlap tail
    {
        if (lane == 0) out_spec.y = out_nyq;
        v2f pt = rdft128_inv_point_lanes(out_spec, &K.tab, lane);
        pt = fft64_lanes<true>(pt, &K.tab, lane);
        // lane l holds samples 2l, 2l+1: lanes 0..31 the output half, lanes 32..63 the new overlap tail
        const int i = 2 * (lane & 31);
        if (lane < 32) {
            const float o0 = sat16f(pt.x * scale * K.hanning[i] + AEC_ST(AS_OUTBUF + i));
            const float o1 = sat16f(pt.y * scale * K.hanning[i + 1] + AEC_ST(AS_OUTBUF + i + 1));
            AEC_ST(AS_OUT_RING + ring_at(bp.out_wr, i)) = o0;
            AEC_ST(AS_OUT_RING + ring_at(bp.out_wr, i + 1)) = o1;
        }
        wave_sync();  // the tail overwrites outBuf only after every lane has read it
        if (lane >= 32) {
            AEC_ST(AS_OUTBUF + i) = pt.x * scale * K.hanning[kAecPart - i];
            AEC_ST(AS_OUTBUF + i + 1) = pt.y * scale * K.hanning[kAecPart - i - 1];
        }
    }
    AEC_ST(AS_DPREV + lane) = W.cur[lane];
    AEC_ST(AS_EPREV + lane) = W.enew[lane];
    wave_sync();
    AEC_PROF(9);
#undef AEC_RELANE
}

template <int MULT>
#ifndef WMX_AEC_WAVES
#define WMX_AEC_WAVES 4
#endif
__global__ __attribute__((amdgpu_waves_per_eu(WMX_AEC_WAVES, WMX_AEC_WAVES))) __launch_bounds__(64 * kAecWavesPerBlock) void aec_near_kernel(float *__restrict__ state, AecFarBufs F_all,
                                                                          const float *__restrict__ consts_g,
                                                                          const AecPlan *__restrict__ plans, int n_packets, int n_classes,
                                                                          const int32_t *__restrict__ plan_of,
                                                                          const AecNoiseEntry *__restrict__ noise_tab,
                                                                          const int16_t *near_pcm, int16_t *out_pcm, int n_streams,
                                                                          long stream_stride, long packet_stride, int chn, int pkg,
                                                                          const int *__restrict__ stream_far, const uint8_t *__restrict__ active,
                                                                          const int *__restrict__ order) {
    __shared__ AecConstsNear K;
    __shared__ AecWaveLds Wv[kAecWavesPerBlock];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: state pointers become scalar bases
    // One stream per wave.  With several cohorts in the batch the streams are taken from the host's order list: workgroups are dealt
    // round-robin over the 8 XCDs (blocks b and b + 8 share one), so workgroup i takes slot (i % 8) * (grid / 8) + i / 8 -- XCD x
    // works through the x-th eighth of the list in sequence.  aec_rebuild_order lays the cohort-sorted chunks out in it so that a small
    // cohort's streams meet in one L2 (its 48 far-end history rows per block are fetched once, not by 4 096 cohorts' rows competing
    // in every L2) and a large cohort is spread evenly (order == nullptr: one cohort, stream = slot).
    int sidx;
    if (order) {
        const unsigned i = blockIdx.x, per = gridDim.x >> 3;
        sidx = __builtin_amdgcn_readfirstlane(order[((i & 7u) * per + (i >> 3)) * kAecWavesPerBlock + wave]);  // -1: no stream
    } else {
        sidx = blockIdx.x * kAecWavesPerBlock + wave;
    }
    const bool live = sidx >= 0 && stream_active(active, sidx, n_streams);  // no stream, or one that is switched off: nothing touched
    // Everything this wave needs from memory is requested up front -- its first near-end packet, the 24 filter rows, the
    // LDS part of the state, the constants -- and waited for ONCE, at the workgroup barrier; issued phase by phase, each
    // group costs its own HBM round trip.  (A wave without a stream reads stream 0 and leaves after the barrier.)
    const int sl = (sidx >= 0 && sidx < n_streams) ? sidx : 0;
    // the far-end this stream is cancelled against (one per batch unless the handle was created with far-end groups):
    // wave-uniform, so the group's buffers are scalar bases like the single far-end's
    const int grp = stream_far ? __builtin_amdgcn_readfirstlane(stream_far[sl]) : 0;
    const AecFarBufs F = far_group(F_all, grp);
    plans += plan_of ? __builtin_amdgcn_readfirstlane(plan_of[grp]) : grp;  // [packet][class]
    float *gst = state + (size_t)sl * AS_WORDS;
    // the constants first (L2 hits): their copy into LDS then waits for them alone, not for the HBM loads behind them
    constexpr int kConstIt = (kAecConstNearWords + 64 * kAecWavesPerBlock - 1) / (64 * kAecWavesPerBlock);
    float kc[kConstIt];
#pragma unroll
    for (int k = 0; k < kConstIt; k++) {
        const int i = threadIdx.x + 64 * kAecWavesPerBlock * k;
        kc[k] = consts_g[i < kAecConstNearWords ? i : 0];
    }
    int16_t pcm0[2][2];
    {
        const int16_t *in0 = near_pcm + (size_t)sl * stream_stride;
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                // unconditional loads (sample 0 stands in where this lane has nothing to fetch): loads under a branch
                // are waited for one by one
                const int i = lane + 64 * h;
                const bool want = i < kAecFrame && s * kAecFrame + i < pkg;
                const int16_t v = in0[want ? (s * kAecFrame + i) * chn : 0];
                pcm0[s][h] = want ? v : (int16_t)0;
            }
    }
    // filter taps straight into registers (256-byte rows), the rest as one contiguous block, later parked in LDS
    // (tried in round 6: the state -- read once, written once per launch -- as NON-TEMPORAL accesses, so that it would not push the
    // far-end history rows, which are re-read, out of the L2 when every stream has its own far-end: no change at either end,
    // profiles/r06/sweeps/ab_nt_state.txt)
    AecTaps taps;
#pragma unroll
    for (int p = 0; p < 12; p++) {
        taps.t[p].x = gst[AS_W_RE + p * BP + lane];
        taps.t[p].y = gst[AS_W_IM + p * BP + lane];
    }
    const float wn0 = gst[AS_W_RE + (lane < 12 ? lane : 0) * BP + kAecPart];
    constexpr int kChunks = (AS_LDS_WORDS / 4 + 63) / 64;
    float4 c[kChunks];
    {
        const float4 *g4 = reinterpret_cast<const float4 *>(gst + AS_LDS0);
#pragma unroll
        for (int k = 0; k < kChunks; k++) {
            // lanes past the end of the last chunk fetch (and later park) their element of the chunk before once more:
            // no lane-dependent branch around a load or its store (such a load is issued late and waited for on the spot)
            const int i = lane + 64 * k;
            c[k] = g4[i < AS_LDS_WORDS / 4 ? i : i - 64];
        }
    }
#pragma unroll
    for (int k = 0; k < kConstIt; k++) {
        const int i = threadIdx.x + 64 * kAecWavesPerBlock * k;
        if (i < kAecConstNearWords) reinterpret_cast<float *>(&K)[i] = kc[k];
    }
    // the only block-level barrier: LDS writes drained, then s_barrier -- spelled out because __syncthreads() would also
    // wait for every outstanding global load (the state requests above) in all eight waves
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (!live) return;
    const PowTables *powtab = reinterpret_cast<const PowTables *>(consts_g + kAecConstWords);  // stays in global memory (L1 / L2 hits)
    AecWaveLds &W = Wv[wave];
#ifdef WMX_AEC_PROF
    if (lane < 16) W.prof[lane] = 0;
#endif
    AEC_PROF_START;
    if (lane < 12) W.wn[lane] = wn0;
    {
        float4 *s4 = reinterpret_cast<float4 *>(W.st);
#pragma unroll
        for (int k = 0; k < kChunks; k++) {
            const int i = lane + 64 * k;
            s4[i < AS_LDS_WORDS / 4 ? i : i - 64] = c[k];
        }
    }
    // the prefetched packet leaves the registers (it would stay live, or spilled, across the whole packet loop): the
    // first sub-frame waits in the still unused work rows, the second in its own row
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int i = lane + 64 * h;
        if (i < kAecFrame) {
            W.fa[0][i] = (float)pcm0[0][h];
            W.park[i] = pcm0[1][h];
        }
    }
    wave_sync();
    AEC_PROF(10);
    for (int p = 0; p < n_packets; p++) {
        const AecPlan &pl = plans[(size_t)p * n_classes];
        if (!pl.has_near) continue;
        const size_t off = (size_t)sidx * stream_stride + (size_t)p * packet_stride;
        const int16_t *in = near_pcm + off;
        int16_t *out = out_pcm + off;
        if (pl.passthrough) {
            // start-up phase: AEC disabled, out = near (echo_cancellation.c:651-657); left channel to all channels
            for (int i = lane; i < pkg; i += 64) {
                const int16_t v = in[i * chn];
                for (int c = 0; c < chn; c++) out[i * chn + c] = v;
            }
            continue;
        }
        for (int s = 0; s < pl.n_sub; s++) {
            const AecSubPlan &sp = pl.sub[s];
            const int ln = opaque_lane(lane);  // PCM offsets rebuilt per sub-frame, not carried (or spilled) across the blocks
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int i = ln + 64 * h;
                if (i >= kAecFrame) continue;
                const float v = (p == 0 && s < 2) ? (s == 0 ? W.fa[0][i] : (float)W.park[i]) : (float)in[(s * kAecFrame + i) * chn];
                AEC_ST(AS_NEAR_RING + ring_at(sp.near_wr, i)) = v;
            }
            wave_sync();
            for (int k = 0; k < sp.n_blocks; k++)
                aec_block<MULT>(K, consts_g + kAecConstNearWords, powtab, W, taps, F, pl.blk[sp.first_blk + k], noise_tab, lane);
            for (int i = opaque_lane(lane); i < kAecFrame; i += 64) {
                const int16_t v = (int16_t)AEC_ST(AS_OUT_RING + ring_at(sp.out_rd, i));
                for (int c = 0; c < chn; c++) out[(s * kAecFrame + i) * chn + c] = v;
            }
            wave_sync();
        }
    }
    wave_sync();
    AEC_PROF(11);  // includes the blocks; subtract 0..9
    // ---- state out (addresses recomputed: keeping the ones of the load alive across the packet loop costs 17 VGPRs)
    asm volatile("" : "+s"(gst));
    // (the asm strips the address space with the value: restated, or the 26 stores below are FLAT stores)
    float __attribute__((address_space(1))) *gout = (float __attribute__((address_space(1))) *)gst;
    const int ln = opaque_lane(lane);  // same for the lane-derived offsets (otherwise kept alive, or spilled, across the loop)
#pragma unroll
    for (int p = 0; p < 12; p++) {
        gout[AS_W_RE + p * BP + ln] = taps.t[p].x;
        gout[AS_W_IM + p * BP + ln] = taps.t[p].y;
    }
    if (ln < 12) gout[AS_W_RE + ln * BP + kAecPart] = W.wn[ln];
    {
        typedef float v4f __attribute__((ext_vector_type(4)));  // a native vector: HIP's float4 class assigns through generic pointers only
        v4f __attribute__((address_space(1))) *g4 = (v4f __attribute__((address_space(1))) *)(gout + AS_LDS0);
        const v4f *s4 = reinterpret_cast<const v4f *>(W.st);
        for (int i = ln; i < AS_LDS_WORDS / 4; i += 64) g4[i] = s4[i];
    }
    AEC_PROF(12);
#ifdef WMX_AEC_PROF
    if (lane < 16) atomicAdd(&g_aec_prof[lane], W.prof[lane]);
#endif
}
#ifdef WMX_AEC_PROF
extern "C" int wmx_debug_aec_prof(unsigned long long *out16, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_aec_prof), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_aec_prof), z, sizeof(z));
    }
    return 0;
}
#endif


__global__ void aec_fill_state(float *state, const float *tmpl, int words, int n_streams) {
    const size_t total = (size_t)words * n_streams;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        state[i] = tmpl[i % words];
}

__global__ void aec_set_group(int *stream_far, const int32_t *idx, int n_idx, int group) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_idx) stream_far[idx[j]] = group;
}


// ---------------------------------------------------------------- coalescing control cohorts (wmx_aec_coalesce)
// Two cohorts whose control planes have converged -- same fill levels, delays and counters, only the ring POSITIONS differ -- hold the
// same far-end data in their slabs, rotated by the difference of their positions: this kernel says whether they really do, word for
// word (the running far power is an IIR from each cohort's own start: it is equal once the difference has been rounded away, and
// only the bits can say when).  One workgroup per pair; flags[pair] = 1 when every buffer of `b` equals its rotation of `a`.
constexpr int kAecCoMax = 32;  // pairs per comparison launch / merges per call
struct AecPairChecks {
    AecPairCheck p[kAecCoMax];
};
__global__ __launch_bounds__(256) void aec_cohort_equal(AecFarBufs F_all, AecPairChecks pairs, int *flags) {
    const AecPairCheck pc = pairs.p[blockIdx.x];
    const AecFarBufs A = far_group(F_all, pc.a), B = far_group(F_all, pc.b);
    unsigned diff = 0;
    auto rows = [&](const float *a, const float *b, int n_rows, int row_len, int d_row) {
        const unsigned *ua = reinterpret_cast<const unsigned *>(a), *ub = reinterpret_cast<const unsigned *>(b);
        for (int i = threadIdx.x; i < n_rows * row_len; i += 256) {
            const int r = i / row_len, c = i - r * row_len;
            int rb = r + d_row;
            rb -= rb >= n_rows ? n_rows : 0;
            diff |= ua[i] ^ ub[rb * row_len + c];
        }
    };
    rows(A.pre, B.pre, kAecPreLen, 1, pc.d_pre);
    rows(A.tring, B.tring, kAecFarBlocks, 64, pc.d_far);  // the partitions' samples, two int16 per word
    rows(A.hist, B.hist, kAecHist, 130, pc.d_hist);  // every consumed spectrum is stored twice, kAecHist rows apart
    rows(A.hist + (size_t)kAecHist * 130, B.hist + (size_t)kAecHist * 130, kAecHist, 130, pc.d_hist);
    rows(A.nyq, B.nyq, kAecHist, 2, pc.d_hist);
    rows(A.nyq + 2 * kAecHist, B.nyq + 2 * kAecHist, kAecHist, 2, pc.d_hist);
    rows(A.hist_w, B.hist_w, kAecHist, 130, pc.d_hist);
    rows(A.xpow_seq, B.xpow_seq, kAecHist, BP, pc.d_hist);
    rows(A.xpow, B.xpow, 1, BP, 0);
    const int any = __syncthreads_or(diff != 0);
    if (threadIdx.x == 0) flags[blockIdx.x] = any ? 0 : 1;
}
// The merge itself: one wave per stream; members of a `from` cohort get their two re-blocking rings rotated to the positions of the
// cohort they join (the plans they will be run with are its plans) and its id.  Streams of other cohorts are not touched.
__global__ __launch_bounds__(256) void aec_merge_streams(float *state, int *stream_far, int n_streams, AecPairChecks pairs, int n_pairs) {
    const int s = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= n_streams) return;
    const int c = stream_far[s];
    int k = -1;
    for (int i = 0; i < n_pairs; i++)
        if (pairs.p[i].b == c) k = i;
    if (k < 0) return;
    float *st = state + (size_t)s * AS_WORDS;
    const int d[2] = {pairs.p[k].d_near, pairs.p[k].d_out};
    const int base[2] = {AS_NEAR_RING, AS_OUT_RING};
    float v[2][3];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) v[r][j] = lane + 64 * j < kAecRing ? st[base[r] + lane + 64 * j] : 0.f;
    __builtin_amdgcn_s_waitcnt(0);  // every load of the wave before its first store: the rotation is in place
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = lane + 64 * j;
            if (i < kAecRing) st[base[r] + (i + d[r]) % kAecRing] = v[r][j];
        }
    if (lane == 0) stream_far[s] = pairs.p[k].a;
}
__global__ void aec_clamp_group(int *stream_far, int n_streams, int n_far) {
    const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (s < n_streams && stream_far[s] >= n_far) stream_far[s] = 0;
}
}  // namespace
}  // namespace wmx

// ================================================================== host
struct wmx_aec {
    int device;  // the HIP device the state lives on (current device at create); every entry point switches to it
    int n_streams, chn, freq, pkg;
    std::vector<wmx::AecCtl> ctl;  // one control plane per far-end group / cohort (n_far of them) -- of which only the LEADERS' are kept
                                   // up to date, see `lead`
    // Control-plane classes.  A control plane is index arithmetic on the call pattern (when the handle was made, packet sizes,
    // reported delays), never on audio: cohorts that were started at the same point and are called alike have EQUAL planes for ever
    // -- every mix group of a conference server started together, every call set up in the same tick -- although their far-ends
    // (and so their far-end histories on the device) differ and they can never be merged.  Such cohorts form a class: lead[g] is
    // the cohort whose AecCtl stands for g's (lead[g] == g: g leads); a launch runs ONE control plane and uploads ONE plan per class
    // and packet, and the kernels find a cohort's plan through d_plan_of.  ctl[g] of a follower is stale; aec_ctl() reads through,
    // aec_ctl_own() makes a cohort the owner of an up-to-date copy before anything treats it differently from its class.
    std::vector<int32_t> lead;         // [n_far]
    std::vector<int32_t> cls_leader;   // [n_cls] the leaders, compact
    std::vector<int32_t> h_plan_of[2]; // [n_far] class index of every cohort; alternating sources of the asynchronous upload
    int h_plan_of_sel;
    int32_t *d_plan_of;                // [cap_far]
    bool cls_dirty;                    // lead[] changed: cls_leader / plan_of are rebuilt (and uploaded) by the next launch
    float *d_state;
    float *d_tmpl;   // the state aec_init gives a stream (reset_streams refills from it)
    float *d_consts;
    float *d_far;  // one allocation carved into AecFarBufs
    wmx::AecFarBufs far;
    static constexpr int kPlanBufs = 4;
    wmx::AecPlan *d_plans;   // kPlanBufs x n_far x kAecMaxPktPerLaunch plans, slots used round robin
    wmx::AecPlan *h_plans;   // pinned mirror: the asynchronous copy reads it in place, so each slot has its own
    hipEvent_t plan_free[kPlanBufs];  // recorded behind the kernels that read slot i; waited for before slot i is rewritten
    bool plan_used[kPlanBufs];
    int plan_sel;
    int n_far;               // far-end groups / control cohorts in use, retired ones included (1 = one shared far-end): ids 0 .. n_far - 1
    int cap_far;             // cohorts the device buffers (far slabs, plan slots) are allocated for; grows by doubling
    std::vector<uint8_t> live;  // [n_far] 0: retired by wmx_aec_retire_cohort (never called, its id is handed out again)
    int *d_stream_far;       // [n_streams] group of each stream, or nullptr while there has only ever been one
    // cohort-sorted, XCD-aware stream order of the near kernel (see aec_near_kernel); rebuilt on the host when memberships changed
    std::vector<int32_t> h_cohort_of;  // [n_streams] host mirror of d_stream_far
    std::vector<int32_t> h_order[2];  // alternating: the source of the previous upload is not rewritten while that copy may still be reading it
    int h_order_sel;
    int32_t *d_order;         // [order_wgs * 4] stream of every (chunk, wave), -1 = none; nullptr while one cohort
    unsigned order_wgs;       // workgroups of an ordered launch: a multiple of 8
    bool order_dirty;
    int order_age;            // launches since the order was rebuilt: under churn it is rebuilt every kOrderEvery launches at most
    static constexpr int kOrderEvery = 16;
    wmx::AecNoiseEntry *d_noise_tab;  // cosf / sinf of the comfort noise's 32 768 possible phases (host libm, aec_ctl.h)
                                      // + 64 x (a_k, c_k): k = 1 .. 64 draws of the generator in one step
    // wmx_aec_coalesce: the pairs whose device comparison is in flight (`b` < 0: dropped, the two were not called identically since)
    wmx::AecPairChecks co_pairs;
    int co_n;
    bool co_inflight;
    int *d_co_flags, *h_co_flags;  // [kAecCoMax] device result and its pinned copy
    hipEvent_t co_done;
    long co_calls;                 // wmx_aec_coalesce calls so far
    std::vector<long> co_retry_at; // [n_far] a cohort whose comparison failed is not proposed again before this call
    long last_far_group_stride;    // of the latest run: cohorts that hear private far-end packets are never candidates
    long co_merged_total;
    int16_t *d_zero_far;  // a silent far-end packet for near-only calls that need no far data
    wmx::StreamLife life;
    // in-stream timing of the two kernels (wmx_aec_set_timing): event quadruples [far start | far end | near start | near end]
    bool timing;
    std::vector<hipEvent_t> tev;
    size_t tev_used;
    // The far kernel is one wave per cohort and needs nothing but the far-end packet: a caller that has other work for the
    // GPU in front of the near kernel (wmx_chain_process: the noise suppressor) lets it run BESIDE that work on this side
    // stream instead of alone in front of the near kernel (wmx::aec_fork_far).
    hipStream_t side;
    hipEvent_t ev_fork, ev_join;
    bool fork_pending;
    bool no_order;  // WMIX_AMD_AEC_NO_ORDER (developer A/B switch, read at create): streams in slot order whatever their cohorts
    // per-call scratch kept with the handle (no allocation on the heartbeat's path)
    std::vector<int> rc_g;           // [n_far] what the wrapper would have returned to the members of each cohort
    std::vector<int32_t> same_delay; // [n_far] the one reported delay of wmx_aec_run / _run_groups, spread over the cohorts
    // host time spent in the control planes (the per-cohort, per-packet AecCtl loop of wmx_aec_run_cohorts), summed; wmx_aec_host_ctl
    double ctl_seconds;
    long ctl_calls;
    static constexpr size_t kMaxTimingEvents = 4 * 4096;  // timing left on and never polled: the oldest quadruples are reused
};

extern "C" {

int wmx_aec_destroy(wmx_aec *h) {
    WMX_ON_DEVICE(h);
    if (!h) return 0;
    if (h->d_state) (void)hipFree(h->d_state);
    if (h->d_consts) (void)hipFree(h->d_consts);
    if (h->d_far) (void)hipFree(h->d_far);
    if (h->d_plans) (void)hipFree(h->d_plans);
    if (h->h_plans) (void)hipHostFree(h->h_plans);
    if (h->d_stream_far) (void)hipFree(h->d_stream_far);
    if (h->d_plan_of) (void)hipFree(h->d_plan_of);
    if (h->d_order) (void)hipFree(h->d_order);
    if (h->d_noise_tab) (void)hipFree(h->d_noise_tab);
    if (h->d_co_flags) (void)hipFree(h->d_co_flags);
    if (h->h_co_flags) (void)hipHostFree(h->h_co_flags);
    if (h->co_done) (void)hipEventDestroy(h->co_done);
    if (h->d_tmpl) (void)hipFree(h->d_tmpl);
    h->life.release();
    for (hipEvent_t ev : h->tev) (void)hipEventDestroy(ev);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    for (int i = 0; i < wmx_aec::kPlanBufs; i++)
        if (h->plan_free[i]) (void)hipEventDestroy(h->plan_free[i]);
    delete h;
    return 0;
}

}  // extern "C"
static void aec_co_drop(wmx_aec *h, int cohort);
extern "C" {

int wmx_aec_create(wmx_aec **out, int n_streams, int chn, int freq, int interval_ms) {
    return wmx_aec_create_groups(out, n_streams, chn, freq, interval_ms, 1, nullptr);
}

// The near kernel's stream order: streams sorted by (re-blocking phase, cohort), cut into chunks of one workgroup (4 streams).  The kernel gives
// XCD x the slots [x W/8, (x + 1) W/8) of the list in sequence; the list is laid out so that the sorted chunks are DEALT to the XCDs in
// runs of kOrderRun: a small cohort's chunks stay in one XCD, next to each other in time (its far-end rows are fetched into one L2,
// once); a large cohort -- what wmx_aec_coalesce leaves -- is spread evenly over all eight, and all eight walk through the sorted list
// side by side.  Evenly matters: a packet is 2.5 blocks at 16 kHz, so a cohort runs 2 or 3 blocks in a launch depending on the phase
// of its re-blocking ring, and eight cohorts of four phases each laid out one per XCD made every launch as long as a 3-block one
// (0.82 ms instead of 0.70, measured).  Any permutation is CORRECT (a wave finds its stream's cohort in d_stream_far).
// ---- control-plane classes (see wmx_aec::lead): aec_ctl / aec_ctl_own / aec_ctl_join / aec_classes_split / aec_classes_list live in
// aec_ctl.h, free of HIP, so that the sanitizer driver can run them against a per-cohort model
using wmx::aec_ctl;
using wmx::aec_ctl_own;
using wmx::aec_ctl_join;
// cls_leader / plan_of from lead[], and plan_of onto the device in `s` (alternating host sources, like the stream order)
static int aec_rebuild_classes(wmx_aec *h, hipStream_t s) {
    const int G = h->n_far;
    h->h_plan_of_sel ^= 1;
    std::vector<int32_t> &po = h->h_plan_of[h->h_plan_of_sel];
    wmx::aec_classes_list(h, h->cls_leader, po);
    if (G > 1) WMX_HIP_RC(hipMemcpyAsync(h->d_plan_of, po.data(), sizeof(int32_t) * (size_t)G, hipMemcpyHostToDevice, s));
    h->cls_dirty = false;
    return 0;
}

static int aec_rebuild_order(wmx_aec *h, hipStream_t s) {
    using namespace wmx;
    const int S = h->n_streams, G = h->n_far;
    constexpr unsigned kOrderRun = 16;  // chunks dealt to one XCD at a time
    const unsigned wgs = (unsigned)((S + kAecWavesPerBlock - 1) / kAecWavesPerBlock), wgs8 = (wgs + 8u * kOrderRun - 1u) / (8u * kOrderRun) * (8u * kOrderRun);
    // sort key: (phase of the cohort's re-blocking ring, cohort id).  The phase decides how many blocks a packet is for the cohort
    // (the fill level behind a packet: 0 / 32 at 16 kHz, 0 / 16 / 32 / 48 at 8 kHz) and never changes in a handle's life; cohorts of
    // one phase form one long segment of the list, which the dealing below spreads evenly -- whatever the pattern of the join ticks
    // (cohorts created at consecutive ticks alternate phases: dealt by id alone, 256 cohorts of 256 streams put one phase on XCDs
    // 0 - 3 and the other on 4 - 7)
    std::vector<int32_t> rank((size_t)G), by((size_t)G);
    for (int g = 0; g < G; g++) by[(size_t)g] = g;
    std::stable_sort(by.begin(), by.end(), [&](int32_t a, int32_t b) {
        return aec_ctl(h, a).near_fr.avail_read() < aec_ctl(h, b).near_fr.avail_read();
    });
    for (int r = 0; r < G; r++) rank[(size_t)by[(size_t)r]] = r;
    std::vector<int32_t> start((size_t)G + 1, 0);
    for (int i = 0; i < S; i++) start[(size_t)rank[(size_t)h->h_cohort_of[(size_t)i]] + 1]++;
    for (int g = 0; g < G; g++) start[(size_t)g + 1] += start[(size_t)g];
    std::vector<int32_t> &ord = h->h_order[h->h_order_sel];
    h->h_order_sel ^= 1;
    ord.assign((size_t)wgs8 * kAecWavesPerBlock, -1);
    for (int i = 0; i < S; i++) {
        const unsigned q = (unsigned)start[(size_t)rank[(size_t)h->h_cohort_of[(size_t)i]]]++;  // position in the sorted list
        const unsigned j = q / kAecWavesPerBlock, w = q % kAecWavesPerBlock;      // sorted chunk, wave
        const unsigned x = (j / kOrderRun) & 7u, k = j / (8u * kOrderRun) * kOrderRun + j % kOrderRun;
        ord[((size_t)x * (wgs8 / 8u) + k) * kAecWavesPerBlock + w] = i;
    }
    if (!h->d_order || h->order_wgs != wgs8) {
        if (h->d_order) {
            WMX_HIP_RC(hipDeviceSynchronize());
            (void)hipFree(h->d_order);
            h->d_order = nullptr;
        }
        WMX_HIP_RC(hipMalloc(reinterpret_cast<void **>(&h->d_order), ord.size() * sizeof(int32_t)));
        h->order_wgs = wgs8;
    }
    // in stream order behind the launches that read the previous order (pageable source: the runtime stages it before returning)
    WMX_HIP_RC(hipMemcpyAsync(h->d_order, ord.data(), ord.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    h->order_dirty = false;
    h->order_age = 0;
    return 0;
}

// floats of one cohort's far-end slab (AecFarBufs carved out of it)
static size_t aec_far_words() {
    using namespace wmx;
    return (size_t)kAecPreLen + (size_t)kAecFarBlocks * 64 + 3 * (size_t)kAecHist * 130 + 4 * (size_t)kAecHist + (size_t)kAecHist * BP + BP;
}

static void aec_carve_far(wmx_aec *h) {
    using namespace wmx;
    float *p = h->d_far;
    h->far.pre = p;
    p += kAecPreLen;
    h->far.tring = p;
    p += (size_t)kAecFarBlocks * 64;
    h->far.hist = p;
    p += 2 * (size_t)kAecHist * 130;
    h->far.nyq = p;
    p += 4 * (size_t)kAecHist;
    h->far.hist_w = p;
    p += (size_t)kAecHist * 130;
    h->far.xpow_seq = p;
    p += (size_t)kAecHist * BP;
    h->far.xpow = p;
    h->far.group_words = aec_far_words();
}

// Device buffers for `cap` cohorts and launches of up to `pkts` packets: the far-end slabs (existing ones are carried over),
// and the plan slots.  Growing is a control-plane operation (the device is drained); it doubles, so a batch that
// gains cohorts one join at a time reallocates a logarithmic number of times.
static int aec_reserve(wmx_aec *h, int cap) {
    using namespace wmx;
    if (cap <= h->cap_far) return 0;
    WMX_HIP_RC(hipDeviceSynchronize());  // plans and far slabs may be in use by launches in flight
    const size_t fw = aec_far_words();
    int ncap = h->cap_far;
    if (cap > h->cap_far) {
        ncap = h->cap_far > 0 ? h->cap_far : 1;
        while (ncap < cap) ncap *= 2;
    }
    // everything new is allocated BEFORE anything old is let go: a failure leaves the handle as it was
    float *nf = nullptr;
    AecPlan *nd = nullptr, *nh = nullptr;
    int32_t *npo = nullptr;
    const size_t plan_bytes = (size_t)wmx_aec::kPlanBufs * ncap * kAecMaxPktPerLaunch * sizeof(AecPlan);
    hipError_t e = hipSuccess;
    if (ncap > h->cap_far) {
        e = hipMalloc(&nf, fw * (size_t)ncap * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&nd, plan_bytes);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&nh), plan_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(&npo, sizeof(int32_t) * (size_t)ncap);
        if (e == hipSuccess && h->d_far) e = hipMemcpy(nf, h->d_far, fw * (size_t)h->cap_far * sizeof(float), hipMemcpyDeviceToDevice);
        if (e == hipSuccess)
            e = hipMemset(nf + fw * (size_t)h->cap_far, 0, fw * (size_t)(ncap - h->cap_far) * sizeof(float));
    }
    if (e != hipSuccess) {
        if (nf) (void)hipFree(nf);
        if (nd) (void)hipFree(nd);
        if (nh) (void)hipHostFree(nh);
        if (npo) (void)hipFree(npo);
        return hip_fail(e, "growing the cohort buffers", __FILE__, __LINE__);
    }
    if (nf) {
        if (h->d_far) (void)hipFree(h->d_far);
        if (h->d_plans) (void)hipFree(h->d_plans);
        if (h->h_plans) (void)hipHostFree(h->h_plans);
        if (h->d_plan_of) (void)hipFree(h->d_plan_of);
        h->d_plan_of = npo;
        h->cls_dirty = true;  // the new array holds nothing yet
        h->d_far = nf;
        h->d_plans = nd;  // plan slots: kPlanBufs x [kAecMaxPktPerLaunch][ncap] (a launch uses the first packets x n_far of its slot)
        h->h_plans = nh;
        for (int i = 0; i < wmx_aec::kPlanBufs; i++) h->plan_used[i] = false;  // drained above
        h->cap_far = ncap;
        aec_carve_far(h);
    }
    return 0;
}

int wmx_aec_create_groups(wmx_aec **out, int n_streams, int chn, int freq, int interval_ms, int n_far, const int32_t *stream_far) {
    using namespace wmx;
    if (!out) return WMX_EINVAL;
    *out = nullptr;
    if (n_far < 1) {
        set_error("wmx_aec_create_groups: n_far=%d", n_far);
        return WMX_EINVAL;
    }
    // stream_far == NULL with n_far > 1: every stream starts in group / cohort 0 and is moved by wmx_aec_reset_streams
    if (n_far > 1 && stream_far)
        for (int i = 0; i < n_streams; i++)
            if (stream_far[i] < 0 || stream_far[i] >= n_far) {
                set_error("wmx_aec_create_groups: stream %d maps to far-end %d of %d", i, stream_far[i], n_far);
                return WMX_EINVAL;
            }
    // aec_init: freq <= 16000 and a multiple of 8000 (src/webrtc.c:220-221)
    if ((freq != 8000 && freq != 16000) || chn < 1 || n_streams < 1) {
        set_error("wmx_aec_create: unsupported n_streams=%d chn=%d freq=%d", n_streams, chn, freq);
        return WMX_EINVAL;
    }
    wmx_aec *h = new wmx_aec();
    if ((h->device = wmx::current_device()) < 0) {
        delete h;
        return WMX_ENODEV;
    }
    h->n_streams = n_streams;
    h->chn = chn;
    h->freq = freq;
    h->pkg = freq / 1000 * ((freq <= 8000 && interval_ms % 20 == 0) ? 20 : 10);  // src/webrtc.c:239-248
    h->ctl.resize((size_t)n_far);
    for (wmx::AecCtl &c : h->ctl) c.init(freq);
    h->lead.assign((size_t)n_far, 0);  // made together, equal planes: one class led by cohort 0 until something tells them apart
    h->h_plan_of_sel = 0;
    h->d_plan_of = nullptr;
    h->cls_dirty = true;
    h->live.assign((size_t)n_far, 1);
    h->d_state = h->d_consts = h->d_far = h->d_tmpl = nullptr;
    h->timing = false;
    h->ctl_seconds = 0.0;
    h->ctl_calls = 0;
    h->tev_used = 0;
    h->side = nullptr;
    h->ev_fork = h->ev_join = nullptr;
    h->fork_pending = false;
    h->no_order = getenv("WMIX_AMD_AEC_NO_ORDER") != nullptr;
    h->d_plans = nullptr;
    h->h_plans = nullptr;
    h->d_stream_far = nullptr;
    h->d_noise_tab = nullptr;
    h->co_n = 0;
    h->co_inflight = false;
    h->d_co_flags = h->h_co_flags = nullptr;
    h->co_done = nullptr;
    h->co_calls = 0;
    h->co_retry_at.assign((size_t)n_far, 0);
    h->last_far_group_stride = 0;
    h->co_merged_total = 0;
    h->d_order = nullptr;
    h->order_wgs = 0;
    h->order_dirty = n_far > 1;
    h->h_order_sel = 0;
    h->order_age = wmx_aec::kOrderEvery;
    h->h_cohort_of.assign((size_t)n_streams, 0);
    if (n_far > 1 && stream_far)
        for (int i = 0; i < n_streams; i++) h->h_cohort_of[(size_t)i] = stream_far[i];
    h->n_far = n_far;
    h->cap_far = 0;
    h->plan_sel = 0;
    for (int i = 0; i < wmx_aec::kPlanBufs; i++) h->plan_free[i] = nullptr, h->plan_used[i] = false;
    h->d_zero_far = nullptr;
    // constants: Ooura tables (frozen rdft_w) + the three curves of aec_core.c:49-103
    AecConsts K;
    memset(&K, 0, sizeof(K));
    fft_tables_aec128(&K.tab);
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < 65; i++) {
        K.hanning[i] = (float)sin(pi * i / 128.0);
        K.overdrive[i] = (float)(floor((sqrt(i / 64.0) + 1.0) * 1e4 + 0.5) / 1e4);
        K.weight[i] = i == 0 ? 0.f : (float)(floor((0.3 * sqrt((i - 1) / 63.0) + 0.1) * 1e4 + 0.5) / 1e4);
    }
    // initial per-stream state (InitAec aec_core.c:1623-1681)
    std::vector<float> st(AS_WORDS, 0.f);
    for (int b = 0; b < 65; b++) {
        st[AS_DMIN + b] = 1.0e6f;
        st[AS_SD + b] = 1.f;
        st[AS_SX + b] = 1.f;
    }
    st[AS_HNLFBMIN] = 1.f;
    st[AS_HNLFBLOCALMIN] = 1.f;
    st[AS_HNLXDAVGMIN] = 1.f;
    st[AS_OVERDRIVE] = 2.f;
    st[AS_OVERDRIVESM] = 2.f;
    {
        const uint32_t seed0 = 777u;  // aec->seed, aec_core.c:1670
        memcpy(&st[AS_NSEED], &seed0, 4);
    }
    hipError_t e;
#define AEC_TRY(x)                                         \
    if ((e = (x)) != hipSuccess) {                         \
        int rc = hip_fail(e, #x, __FILE__, __LINE__);      \
        wmx_aec_destroy(h);                                \
        return rc;                                         \
    }
    AEC_TRY(hipMalloc(&h->d_state, (size_t)AS_WORDS * n_streams * sizeof(float)));
    AEC_TRY(hipMalloc(&h->d_consts, sizeof(K) + sizeof(PowTables)));  // [AecConsts | PowTables]; only the first part is copied to LDS
    for (int i = 0; i < wmx_aec::kPlanBufs; i++) AEC_TRY(hipEventCreateWithFlags(&h->plan_free[i], hipEventDisableTiming));
    {
        const int rc = aec_reserve(h, n_far);
        if (rc != 0) {
            wmx_aec_destroy(h);
            return rc;
        }
    }
    if (n_far > 1) {
        AEC_TRY(hipMalloc(&h->d_stream_far, sizeof(int) * n_streams));
        if (stream_far) {
            AEC_TRY(hipMemcpy(h->d_stream_far, stream_far, sizeof(int) * n_streams, hipMemcpyHostToDevice));
        } else {
            AEC_TRY(hipMemset(h->d_stream_far, 0, sizeof(int) * n_streams));
        }
    }
    AEC_TRY(hipMalloc(&h->d_tmpl, AS_WORDS * sizeof(float)));
    AEC_TRY(hipMemcpy(h->d_consts, &K, sizeof(K), hipMemcpyHostToDevice));
    {
        static_assert(sizeof(AecConsts) % 16 == 0, "PowTables behind AecConsts must stay 16-byte aligned");
        PowTables pt;
        pow_tables(&pt);
        AEC_TRY(hipMemcpy(reinterpret_cast<char *>(h->d_consts) + sizeof(K), &pt, sizeof(pt), hipMemcpyHostToDevice));
    }
    {
        // the comfort noise's phase table: made once per process with the host libm, one copy per handle on its device
        static const std::vector<AecNoiseEntry> tab = [] {
            std::vector<AecNoiseEntry> t((size_t)kAecNoiseTab);
            aec_noise_table(t.data());
            return t;
        }();
        // behind it, the generator's k-draw steps, k = 1 .. 64: lane l's draw from the state in front of a block, and the 64-draw step
        uint32_t jump[2 * 64];
        for (int l = 0; l < 64; l++) aec_lcg_jump(l + 1, &jump[2 * l], &jump[2 * l + 1]);
        AEC_TRY(hipMalloc(&h->d_noise_tab, sizeof(AecNoiseEntry) * kAecNoiseTab + sizeof(jump)));
        AEC_TRY(hipMemcpy(h->d_noise_tab, tab.data(), sizeof(AecNoiseEntry) * kAecNoiseTab, hipMemcpyHostToDevice));
        AEC_TRY(hipMemcpy(h->d_noise_tab + kAecNoiseTab, jump, sizeof(jump), hipMemcpyHostToDevice));
    }
    AEC_TRY(hipMemcpy(h->d_tmpl, st.data(), AS_WORDS * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(aec_fill_state, dim3(1024), dim3(256), 0, nullptr, h->d_state, h->d_tmpl, (int)AS_WORDS, n_streams);
    AEC_TRY(hipGetLastError());
    AEC_TRY(hipDeviceSynchronize());
#undef AEC_TRY
    *out = h;
    return 0;
}

// A new control cohort (a join time of its own: aec_init of the shared part, src/webrtc.c:217-274): a retired id when there is
// one, else the next; its control plane and far-end history start over.  *cohort receives the id; members join it with
// wmx_aec_reset_streams(h, idx, n, id, stream).  The per-cohort arrays of wmx_aec_run_cohorts have wmx_aec_cohorts(h) entries
// from now on.
int wmx_aec_add_cohort(wmx_aec *h, int *cohort, void *stream) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || !cohort) return WMX_EINVAL;
    int id = -1;
    for (int g = 0; g < h->n_far; g++)
        if (!h->live[(size_t)g]) {
            id = g;
            break;
        }
    if (id < 0) {
        id = h->n_far;
        const int rc = aec_reserve(h, id + 1);
        if (rc != 0) return rc;
        h->ctl.resize((size_t)id + 1);
        h->lead.push_back(id);
        h->cls_dirty = true;
        h->live.push_back(1);
        h->co_retry_at.push_back(0);
        h->n_far = id + 1;
    }
    if (h->n_far > 1 && !h->d_stream_far) {  // so far every stream was in cohort 0 by construction
        WMX_HIP_RC(hipMalloc(&h->d_stream_far, sizeof(int) * h->n_streams));
        WMX_HIP_RC(hipMemsetAsync(h->d_stream_far, 0, sizeof(int) * h->n_streams, as_stream(stream)));
    }
    h->live[(size_t)id] = 1;
    *cohort = id;
    return wmx_aec_reset_cohort(h, id, stream);
}

// The cohort's handles were released (aec_release of every member): it is never called again and its id may be handed out by
// a later wmx_aec_add_cohort.  Streams still mapped to it must be inactive or be moved before the next call.
int wmx_aec_retire_cohort(wmx_aec *h, int cohort) {
    if (!h || cohort < 0 || cohort >= h->n_far) return WMX_EINVAL;
    aec_ctl_own(h, cohort);  // a retired cohort leads nobody
    h->live[(size_t)cohort] = 0;
    aec_co_drop(h, cohort);
    return 0;
}

}  // extern "C"
// ---------------------------------------------------------------- coalescing
// a cohort that is restarted, retired or overwritten is no candidate of a comparison in flight
static void aec_co_drop(wmx_aec *h, int cohort) {
    for (int i = 0; i < h->co_n; i++)
        if (h->co_pairs.p[i].a == cohort || h->co_pairs.p[i].b == cohort) h->co_pairs.p[i].b = -1;
}
extern "C" {

// Merge control cohorts whose planes have converged (include/wmix_amd.h).  Each call first completes the merges whose device
// comparison -- requested by an earlier call -- came back equal, then proposes up to max_pairs new pairs and launches their
// comparison behind the work already in `stream`.  Nothing here waits for the device.
int wmx_aec_coalesce(wmx_aec *h, int max_pairs, int32_t *merged_from, int32_t *merged_into, int cap, int *n_merged, void *stream) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (n_merged) *n_merged = 0;
    if (!h || max_pairs < 0 || cap < 0 || (cap > 0 && (!merged_from || !merged_into))) return WMX_EINVAL;
    hipStream_t s = as_stream(stream);
    h->co_calls++;
    if (h->n_far < 2 || !h->d_stream_far) return 0;
    if (!h->d_co_flags) {
        WMX_HIP(hipMalloc(&h->d_co_flags, sizeof(int) * kAecCoMax));
        WMX_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->h_co_flags), sizeof(int) * kAecCoMax, hipHostMallocDefault));
        WMX_HIP(hipEventCreateWithFlags(&h->co_done, hipEventDisableTiming));
    }
    int merged = 0;
    if (h->co_inflight) {
        const hipError_t q = hipEventQuery(h->co_done);
        if (q == hipErrorNotReady) return 0;  // the comparison has not run yet: nothing new is proposed on top of it
        if (q != hipSuccess) return hip_fail(q, "hipEventQuery(co_done)", __FILE__, __LINE__);
        h->co_inflight = false;
        AecPairChecks go;
        int n_go = 0;
        for (int i = 0; i < h->co_n; i++) {
            AecPairCheck pc = h->co_pairs.p[i];
            if (pc.b < 0) continue;  // dropped by a call in between
            AecCoKey ka, kb;
            const bool ok = h->h_co_flags[i] == 1 && h->live[(size_t)pc.a] && h->live[(size_t)pc.b] &&
                            aec_co_key(aec_ctl(h, pc.a), &ka) && aec_co_key(aec_ctl(h, pc.b), &kb) && ka == kb;
            if (!ok) {
                h->co_retry_at[(size_t)pc.b] = h->co_calls + 64;
                continue;
            }
            if (merged >= cap) continue;  // no room to report it: proposed again by a later call
            aec_co_pair(aec_ctl(h, pc.a), aec_ctl(h, pc.b), pc.a, pc.b, &pc);  // the positions of NOW (same differences, by the keys)
            go.p[n_go++] = pc;
            merged_from[merged] = pc.b;
            merged_into[merged] = pc.a;
            merged++;
        }
        h->co_n = 0;
        if (n_go > 0) {
            hipLaunchKernelGGL(aec_merge_streams, dim3((unsigned)((h->n_streams + 3) / 4)), dim3(256), 0, s, h->d_state, h->d_stream_far,
                               h->n_streams, go, n_go);
            WMX_LAUNCH_CHECK();
            std::vector<int32_t> to((size_t)h->n_far, -1);
            for (int i = 0; i < n_go; i++) {
                to[(size_t)go.p[i].b] = go.p[i].a;
                aec_ctl_own(h, go.p[i].b);  // a retired cohort leads nobody
                h->live[(size_t)go.p[i].b] = 0;  // retired: its id may be handed out again (wmx_aec_add_cohort)
            }
            for (int32_t &c : h->h_cohort_of)
                if (to[(size_t)c] >= 0) c = to[(size_t)c];
            // the id range ends behind the last live cohort again (plans, far kernel waves and the caller's per-cohort arrays are
            // sized by it: wmx_aec_cohorts); switched-off streams that still carry an id beyond it are parked on cohort 0
            int nf = h->n_far;
            while (nf > 1 && !h->live[(size_t)nf - 1]) nf--;
            if (nf < h->n_far) {
                hipLaunchKernelGGL(aec_clamp_group, dim3((unsigned)((h->n_streams + 255) / 256)), dim3(256), 0, s, h->d_stream_far, h->n_streams, nf);
                WMX_LAUNCH_CHECK();
                for (int32_t &c : h->h_cohort_of)
                    if (c >= nf) c = 0;
                h->n_far = nf;
                h->ctl.resize((size_t)nf);
                h->lead.resize((size_t)nf);
                h->cls_dirty = true;
                h->live.resize((size_t)nf);
                h->co_retry_at.resize((size_t)nf);
            }
            h->order_dirty = true;
            h->order_age = wmx_aec::kOrderEvery;  // the next launch sorts the streams by their new cohorts
            h->co_merged_total += n_go;
        }
    }
    if (n_merged) *n_merged = merged;
    if (max_pairs == 0 || h->last_far_group_stride != 0) return 0;
    if (max_pairs > kAecCoMax) max_pairs = kAecCoMax;
    // candidates: the first live cohort with a key leads, every later one with the same key may join it
    // (the lowest ids survive, so that the id range can shrink behind them); keys meet through a hash of their words
    std::unordered_multimap<uint64_t, int> leads;
    leads.reserve((size_t)h->n_far);
    int n = 0;
    for (int g = 0; g < h->n_far && n < max_pairs; g++) {
        if (!h->live[(size_t)g]) continue;
        AecCoKey k, kl;
        if (!aec_co_key(aec_ctl(h, g), &k)) continue;
        uint64_t hash = 1469598103934665603ull;
        for (int v : k.v) hash = (hash ^ (uint32_t)v) * 1099511628211ull;
        int lead = -1;
        const auto range = leads.equal_range(hash);
        for (auto it = range.first; it != range.second && lead < 0; ++it)
            if (aec_co_key(aec_ctl(h, it->second), &kl) && kl == k) lead = it->second;
        if (lead < 0) {
            leads.emplace(hash, g);
            continue;
        }
        if (h->co_retry_at[(size_t)g] > h->co_calls) continue;
        aec_co_pair(aec_ctl(h, lead), aec_ctl(h, g), lead, g, &h->co_pairs.p[n++]);
    }
    h->co_n = n;
    if (n == 0) return 0;
    hipLaunchKernelGGL(aec_cohort_equal, dim3((unsigned)n), dim3(256), 0, s, h->far, h->co_pairs, h->d_co_flags);
    WMX_LAUNCH_CHECK();
    WMX_HIP(hipMemcpyAsync(h->h_co_flags, h->d_co_flags, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s));
    WMX_HIP(hipEventRecord(h->co_done, s));
    h->co_inflight = true;
    return 0;
}

int wmx_aec_packet_samples(const wmx_aec *h) { return h ? h->pkg * h->chn : WMX_EINVAL; }
int wmx_aec_state_words(const wmx_aec *h) { return h ? (int)wmx::AS_WORDS : WMX_EINVAL; }

int wmx_aec_export_state(const wmx_aec *h, int stream_index, float *host_words) {
    WMX_ON_DEVICE(h);
    if (!h || !host_words || stream_index < 0 || stream_index >= h->n_streams) return WMX_EINVAL;
    WMX_HIP(hipDeviceSynchronize());
    WMX_HIP(hipMemcpy(host_words, h->d_state + (size_t)stream_index * wmx::AS_WORDS, wmx::AS_WORDS * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// mode bit 1: buffer the far-end packet (aec_setFrameFar), bit 2: process the near-end packet (aec_process);
// 3 = aec_process2.  d_far: the SHARED far-end, packet p at d_far + p*far_packet_stride.
// Returns 0, WMX_E*, or the reference's -1 when a packet is rejected (bad delay: packets before it are done).
int wmx_aec_run(wmx_aec *h, int mode, const int16_t *d_far, long far_packet_stride, const int16_t *d_near, int16_t *d_out,
                int n_packets, long stream_stride, long packet_stride, int delay_ms, void *stream) {
    return wmx_aec_run_groups(h, mode, d_far, far_packet_stride, 0, d_near, d_out, n_packets, stream_stride, packet_stride, delay_ms, stream);
}

int wmx_aec_run_groups(wmx_aec *h, int mode, const int16_t *d_far, long far_packet_stride, long far_group_stride, const int16_t *d_near,
                       int16_t *d_out, int n_packets, long stream_stride, long packet_stride, int delay_ms, void *stream) {
    if (!h) {
        wmx::set_error("wmx_aec_run: bad argument");
        return WMX_EINVAL;
    }
    // every far-end group in lockstep: same reported delay, all switched on
    h->same_delay.assign((size_t)h->n_far, delay_ms);
    return wmx_aec_run_cohorts(h, mode, d_far, far_packet_stride, far_group_stride, d_near, d_out, n_packets, stream_stride, packet_stride,
                               h->same_delay.data(), nullptr, nullptr, stream);
}

// The general form.  A far-end group is also a control COHORT: its streams were started together (aec_init at the same
// packet) and are called with the same reported delay, so the one control plane the reference runs per handle
// (ProcessNormal's start-up machine, the ring indices, the block counters) is the same for all of them and runs once, on the
// host.  delay_ms[g]: the delay cohort g reports in this call; cohort_on[g] == 0 (optional): cohort g is not called at all --
// its control plane, far-end buffers and streams stay as they are; cohort_rc[g] (optional) receives what aec_process2 would
// have returned to the members of cohort g.  Return value: 0, a WMX_E* error, or the first non-zero cohort_rc.
int wmx_aec_run_cohorts(wmx_aec *h, int mode, const int16_t *d_far, long far_packet_stride, long far_group_stride, const int16_t *d_near,
                        int16_t *d_out, int n_packets, long stream_stride, long packet_stride, const int32_t *delay_ms,
                        const uint8_t *cohort_on, int32_t *cohort_rc, void *stream) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    // a fork point (wmx::aec_fork_far) belongs to THIS call, whichever way it ends: taken over here, used by the first chunk that
    // launches, and gone on every other path (argument errors, every cohort switched off) -- a later call must not start its
    // far kernel on the side stream behind a stale event (round-3 ADVICE)
    const bool fork_here = h && h->fork_pending;
    if (h) h->fork_pending = false;
    if (!h || n_packets < 0 || (mode & 3) == 0 || !delay_ms) {
        set_error("wmx_aec_run: bad argument");
        return WMX_EINVAL;
    }
    const int G = h->n_far;
    if (cohort_rc)
        for (int g = 0; g < G; g++) cohort_rc[g] = 0;
    if (n_packets == 0) return 0;  // frameNum == 0: nothing to do, whatever the pointers are
    if (((mode & 1) && !d_far) || ((mode & 2) && (!d_near || !d_out))) {
        set_error("wmx_aec_run: null buffer");
        return WMX_EINVAL;
    }
    const long per_pkt = (long)h->pkg * h->chn;
    if ((mode & 2) && (packet_stride < per_pkt || (h->n_streams > 1 && stream_stride < per_pkt))) {
        set_error("wmx_aec_run: strides (%ld, %ld) smaller than a packet (%ld samples)", stream_stride, packet_stride, per_pkt);
        return WMX_EINVAL;
    }
    if ((mode & 1) && far_packet_stride < per_pkt && n_packets > 1) {
        set_error("wmx_aec_run: far stride %ld smaller than a packet (%ld samples)", far_packet_stride, per_pkt);
        return WMX_EINVAL;
    }
    hipStream_t s = as_stream(stream);
    const float gpow1np = 0.1f * 12;  // gPow[1] * num_partitions (aec_core.c:1212), evaluated in float like the reference
    std::vector<int> &rc_g = h->rc_g;  // a cohort whose call was rejected runs nothing after the offending packet
    rc_g.assign((size_t)G, 0);
    int rc_first = 0, running = 0;
    for (int g = 0; g < G; g++) running += (h->live[(size_t)g] && (!cohort_on || cohort_on[g])) ? 1 : 0;
    h->last_far_group_stride = (mode & 1) ? far_group_stride : h->last_far_group_stride;
    // pairs whose comparison is in flight (wmx_aec_coalesce) stay candidates only while the two cohorts are called identically
    for (int i = 0; i < h->co_n; i++) {
        AecPairCheck &pc = h->co_pairs.p[i];
        if (pc.b < 0) continue;
        const bool on_a = !cohort_on || cohort_on[pc.a], on_b = !cohort_on || cohort_on[pc.b];
        if (on_a != on_b || (on_a && delay_ms[pc.a] != delay_ms[pc.b]) || ((mode & 1) && far_group_stride != 0)) pc.b = -1;
    }
    // Everything that can fail for lack of memory happens HERE, before a control plane has moved: a call that returns an error has
    // advanced neither the host's planes nor the device's far-end history (round-4 ADVICE: the stream order used to be rebuilt --
    // an allocation and an upload -- between the far and the near kernel of a chunk whose planes had already been advanced).
    const bool ordered = (mode & 2) && h->d_stream_far != nullptr && !h->no_order;
    if (ordered && running > 0 && (h->d_order == nullptr || (h->order_dirty && h->order_age >= wmx_aec::kOrderEvery))) {
        const int rc = aec_rebuild_order(h, s);
        if (rc != 0) return rc;
    }
    // Control-plane classes: a follower that is called differently from its leader in THIS call (switched on / off alone, another
    // reported delay) takes a plane of its own first.  (wmx_aec_run / _run_groups hand every cohort the same delay and no switches.)
    if (G > 1 && (cohort_on || delay_ms != h->same_delay.data())) wmx::aec_classes_split(h, delay_ms, cohort_on);
    bool classes_moved = false;
    if (h->cls_dirty) {
        // uploaded in `s`, behind every launch that still reads the old classes; a far kernel forked onto the side stream would not
        // wait for it (its fork point lies in front of this call): this one launch keeps the far kernel in line
        const int rc = aec_rebuild_classes(h, s);
        if (rc != 0) return rc;
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
        if (chunk > kAecMaxPktPerLaunch) chunk = kAecMaxPktPerLaunch;
        // the next plan slot: its pinned host half and its device half are rewritten only after the kernels that read the
        // device half last have finished, whatever stream they ran on (round-1 ADVICE: the single buffer was reused blindly)
        const int sel = h->plan_sel;
        h->plan_sel = (sel + 1) % wmx_aec::kPlanBufs;
        if (h->plan_used[sel]) WMX_HIP(hipEventSynchronize(h->plan_free[sel]));
        const size_t slot = (size_t)sel * h->cap_far * kAecMaxPktPerLaunch;
        AecPlan *hp = h->h_plans + slot, *dp = h->d_plans + slot;  // [packet][class], C apart: chunk x C plans are uploaded
        int any = 0;
        const auto t_ctl = std::chrono::steady_clock::now();
        for (int c = 0; c < C; c++) {
            const int g = h->cls_leader[(size_t)c];  // the class's one control plane
            const bool on = h->live[(size_t)g] && (!cohort_on || cohort_on[g]) && rc_g[g] == 0;
            for (int k = 0; k < chunk; k++) {
                AecPlan &pl = hp[(size_t)k * C + c];
                memset(&pl, 0, offsetof(AecPlan, blk));
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
                    if (r != 0) {  // src/webrtc.c:463-468: the wrapper stops here; nothing of this packet is written
                        pl.has_near = 0;
                        rc_g[g] = r;
                    }
                }
            }
            if (on && rc_g[g] != 0) {
                running--;
                if (rc_first == 0) rc_first = rc_g[g];
            }
        }
        h->ctl_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ctl).count();
        h->ctl_calls++;
        if (any) {
            // (with many cohorts in one class every far wave would store the same plan into the same line: those read the upload)
            const int by_value = (chunk == 1 && C == 1 && G == 1) ? 1 : 0;
            // the far kernel (and the plans it reads) on the side stream when the caller forked it (first chunk of the call only)
            const bool forked = fork_here && done == 0 && (mode & 2) && !classes_moved;
            hipStream_t fs = forked ? h->side : s;
            if (forked) WMX_HIP(hipStreamWaitEvent(fs, h->ev_fork, 0));
            if (!by_value) WMX_HIP(hipMemcpyAsync(dp, hp, (size_t)C * chunk * sizeof(AecPlan), hipMemcpyHostToDevice, fs));
            hipEvent_t *tv = nullptr;
            if (h->timing && (mode & 2)) {
                if (h->tev_used + 4 > wmx_aec::kMaxTimingEvents) h->tev_used = 0;  // nobody polls: start over on the oldest events
                if (h->tev_used + 4 > h->tev.size())
                    for (int k = 0; k < 4; k++) {
                        hipEvent_t ev;
                        WMX_HIP(hipEventCreate(&ev));
                        h->tev.push_back(ev);
                    }
                tv = &h->tev[h->tev_used];
                h->tev_used += 4;
                WMX_HIP(hipEventRecord(tv[0], fs));
            }
            hipLaunchKernelGGL(aec_far_kernel, dim3((unsigned)G), dim3(64), 0, fs, h->far, h->d_consts, dp, chunk, C, plan_of,
                               d_far ? d_far + (size_t)done * far_packet_stride : nullptr, far_packet_stride, far_group_stride, h->chn, gpow1np,
                               by_value, hp[0]);
            WMX_LAUNCH_CHECK();
            if (tv) WMX_HIP(hipEventRecord(tv[1], fs));
            if (forked) {
                WMX_HIP(hipEventRecord(h->ev_join, fs));
                WMX_HIP(hipStreamWaitEvent(s, h->ev_join, 0));
            }
            if (tv) WMX_HIP(hipEventRecord(tv[2], s));
            if (mode & 2) {
                const int16_t *nin = d_near + (size_t)done * packet_stride;
                int16_t *nout = d_out + (size_t)done * packet_stride;
                if (h->order_age < wmx_aec::kOrderEvery) h->order_age++;  // saturates: a service runs for months
                const int32_t *order = ordered ? h->d_order : nullptr;
                const unsigned grid = ordered ? h->order_wgs : (unsigned)((h->n_streams + kAecWavesPerBlock - 1) / kAecWavesPerBlock);
                const dim3 blk(64 * kAecWavesPerBlock);
                if (h->freq == 8000)
                    hipLaunchKernelGGL((aec_near_kernel<1>), dim3(grid), blk, 0, s, h->d_state, h->far, h->d_consts, dp, chunk, C, plan_of,
                                       h->d_noise_tab, nin, nout, h->n_streams, stream_stride, packet_stride, h->chn, h->pkg, h->d_stream_far,
                                       h->life.d_active, order);
                else
                    hipLaunchKernelGGL((aec_near_kernel<2>), dim3(grid), blk, 0, s, h->d_state, h->far, h->d_consts, dp, chunk, C, plan_of,
                                       h->d_noise_tab, nin, nout, h->n_streams, stream_stride, packet_stride, h->chn, h->pkg, h->d_stream_far,
                                       h->life.d_active, order);
                WMX_LAUNCH_CHECK();
                if (tv) WMX_HIP(hipEventRecord(tv[3], s));
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

// aec_release + aec_init for the listed streams (src/webrtc.c:217-274, 485-505): InitAec state.  cohort >= 0 also moves them
// to that far-end group / control cohort (needs a handle created with wmx_aec_create_groups); -1 leaves the membership.
// A stream's new cohort must have been restarted (wmx_aec_reset_cohort) at the same point of the packet sequence, or the
// stream inherits ring positions of a control plane that started earlier -- which is what no handle of the reference has.
int wmx_aec_reset_streams(wmx_aec *h, const int32_t *idx, int n, int cohort, void *stream) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || n < 0 || (n > 0 && !idx) || cohort < -1 || cohort >= h->n_far) return WMX_EINVAL;
    if (cohort >= 0 && h->n_far > 1 && !h->d_stream_far) return WMX_ESTATE;
    if (n == 0) return 0;
    hipStream_t s = as_stream(stream);
    const int32_t *d_idx = nullptr;
    const int rc = h->life.upload(idx, n, h->n_streams, s, &d_idx);
    if (rc != 0) return rc;
    const unsigned grid = (unsigned)(n < 4096 ? n : 4096);
    hipLaunchKernelGGL((fill_rows_idx<float>), dim3(grid), dim3(256), 0, s, h->d_state, (const float *)h->d_tmpl, (int)AS_WORDS, d_idx, n);
    if (cohort >= 0 && h->d_stream_far) {
        hipLaunchKernelGGL(aec_set_group, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h->d_stream_far, d_idx, n, cohort);
        for (int i = 0; i < n; i++) h->h_cohort_of[(size_t)idx[i]] = cohort;
        h->order_dirty = true;
    }
    WMX_LAUNCH_CHECK();
    return h->life.done(s);
}

// aec_init for a whole cohort's SHARED part: the control plane starts over (start-up phase, empty far-end buffer, ring
// positions, comfort-noise seed) and the far-end history of the group is cleared.  The member streams are reset with
// wmx_aec_reset_streams(..., cohort, ...).
int wmx_aec_reset_cohort(wmx_aec *h, int cohort, void *stream) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || cohort < 0 || cohort >= h->n_far) return WMX_EINVAL;
    aec_ctl_own(h, cohort);
    h->ctl[(size_t)cohort].init(h->freq);
    aec_ctl_join(h, cohort);  // cohorts restarted at the same point of the packet sequence run one control plane
    aec_co_drop(h, cohort);
    WMX_HIP(hipMemsetAsync(h->d_far + (size_t)cohort * h->far.group_words, 0, h->far.group_words * sizeof(float), as_stream(stream)));
    return 0;
}

int wmx_aec_set_active(wmx_aec *h, const uint8_t *host_mask, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    return h->life.set_active(h->n_streams, host_mask, wmx::as_stream(stream));
}

// stream migration: [header | AS_WORDS state words].  The stream's COHORT (control plane + far-end history) travels on its
// own: [header | AecCtl | the group's far-end buffers]; a destination cohort that received it continues exactly where the
// source cohort stands, so a stream imported into it behaves as if it had never moved.
// Format versions of the AEC's blobs (wmx_internal.h: blob_layout).  Stream: 2 since round 5 turned AS_NBLK (the block count) into
// AS_NSEED (the comfort-noise generator's state) in place.  Cohort: 2 since round 6 resized the far-end slab.
static constexpr uint32_t kAecBlobVersion = 2, kAecCohortBlobVersion = 2;
int wmx_aec_stream_state_bytes(const wmx_aec *h) { return h ? (int)(sizeof(wmx::BlobHeader) + wmx::AS_WORDS * 4) : WMX_EINVAL; }
int wmx_aec_cohort_state_bytes(const wmx_aec *h) {
    return h ? (int)(sizeof(wmx::BlobHeader) + sizeof(wmx::AecCtl) + h->far.group_words * 4) : WMX_EINVAL;
}

int wmx_aec_export_stream(wmx_aec *h, int stream_index, void *host_blob) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || !host_blob || stream_index < 0 || stream_index >= h->n_streams) return WMX_EINVAL;
    WMX_HIP(hipDeviceSynchronize());
    char *p = static_cast<char *>(host_blob);
    blob_begin(p, blob_tag("AEC "), blob_layout((uint32_t)h->freq, kAecBlobVersion), AS_WORDS * 4);
    WMX_HIP(hipMemcpy(p + sizeof(BlobHeader), h->d_state + (size_t)stream_index * AS_WORDS, AS_WORDS * 4, hipMemcpyDeviceToHost));
    return 0;
}

int wmx_aec_import_stream(wmx_aec *h, int stream_index, const void *host_blob, int cohort) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || !host_blob || stream_index < 0 || stream_index >= h->n_streams || cohort < -1 || cohort >= h->n_far) return WMX_EINVAL;
    const int rc = blob_check(host_blob, blob_tag("AEC "), blob_layout((uint32_t)h->freq, kAecBlobVersion), AS_WORDS * 4);
    if (rc) return rc;
    WMX_HIP(hipDeviceSynchronize());
    WMX_HIP(hipMemcpy(h->d_state + (size_t)stream_index * AS_WORDS, static_cast<const char *>(host_blob) + sizeof(BlobHeader), AS_WORDS * 4,
                      hipMemcpyHostToDevice));
    if (cohort >= 0 && h->d_stream_far) {
        WMX_HIP(hipMemcpy(h->d_stream_far + stream_index, &cohort, sizeof(int), hipMemcpyHostToDevice));
        h->h_cohort_of[(size_t)stream_index] = cohort;
        h->order_dirty = true;
    }
    return 0;
}

int wmx_aec_export_cohort(wmx_aec *h, int cohort, void *host_blob) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || !host_blob || cohort < 0 || cohort >= h->n_far) return WMX_EINVAL;
    WMX_HIP(hipDeviceSynchronize());
    char *p = static_cast<char *>(host_blob);
    const size_t fb = h->far.group_words * 4;
    blob_begin(p, blob_tag("AECc"), blob_layout((uint32_t)h->freq, kAecCohortBlobVersion), (uint32_t)(sizeof(AecCtl) + fb));
    p += sizeof(BlobHeader);
    memcpy(p, &aec_ctl(h, cohort), sizeof(AecCtl));
    WMX_HIP(hipMemcpy(p + sizeof(AecCtl), h->d_far + (size_t)cohort * h->far.group_words, fb, hipMemcpyDeviceToHost));
    return 0;
}

int wmx_aec_import_cohort(wmx_aec *h, int cohort, const void *host_blob) {
    WMX_ON_DEVICE(h);
    using namespace wmx;
    if (!h || !host_blob || cohort < 0 || cohort >= h->n_far) return WMX_EINVAL;
    const size_t fb = h->far.group_words * 4;
    const int rc = blob_check(host_blob, blob_tag("AECc"), blob_layout((uint32_t)h->freq, kAecCohortBlobVersion), (uint32_t)(sizeof(AecCtl) + fb));
    if (rc) return rc;
    WMX_HIP(hipDeviceSynchronize());
    const char *p = static_cast<const char *>(host_blob) + sizeof(BlobHeader);
    aec_ctl_own(h, cohort);
    memcpy(&h->ctl[(size_t)cohort], p, sizeof(AecCtl));
    aec_ctl_join(h, cohort);
    aec_co_drop(h, cohort);
    WMX_HIP(hipMemcpy(h->d_far + (size_t)cohort * h->far.group_words, p + sizeof(AecCtl), fb, hipMemcpyHostToDevice));
    return 0;
}

int wmx_aec_cohorts(const wmx_aec *h) { return h ? h->n_far : WMX_EINVAL; }
int wmx_aec_cohort_key(const wmx_aec *h, int cohort, int32_t *key11) {
    if (!h || !key11 || cohort < 0 || cohort >= h->n_far) return WMX_EINVAL;
    wmx::AecCoKey k;
    if (!h->live[(size_t)cohort] || !wmx::aec_co_key(aec_ctl(h, cohort), &k)) return 1;  // retired, or still in its start-up
    for (int i = 0; i < 11; i++) key11[i] = k.v[i];
    return 0;
}
int wmx_aec_live_cohorts(const wmx_aec *h) {
    if (!h) return WMX_EINVAL;
    int n = 0;
    for (uint8_t l : h->live) n += l ? 1 : 0;
    return n;
}

}  // extern "C"

// Library-internal (wmx_internal.h): from this point of `stream` on, the far-end packets of the NEXT wmx_aec_run_* call on
// this handle are in place; its far kernel may start here, on the handle's side stream, beside whatever the caller launches
// on `stream` between now and that call.  The near kernel still runs on `stream`, behind the far kernel.
int wmx::aec_fork_far(wmx_aec *h, hipStream_t s) {
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

void wmx::aec_cancel_fork(wmx_aec *h) {
    if (h) h->fork_pending = false;
}

extern "C" {

// In-stream timing of the AEC's two kernels: with timing on, every near-end launch is bracketed by HIP events recorded on
// the launch stream (before the far kernel, between the two, after the near kernel).  wmx_aec_timing waits for the last
// one, returns the number of launches and the summed durations since the previous call, and starts over.
int wmx_aec_set_timing(wmx_aec *h, int on) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    h->timing = on != 0;
    return 0;
}

// Host time of the control planes since the previous call: launches (chunks of at most 16 packets) and the seconds their
// per-cohort, per-packet AecCtl loops took on the caller's thread (index arithmetic only; it runs ahead of the GPU).
int wmx_aec_host_ctl(wmx_aec *h, long *n_launches, double *seconds) {
    if (!h) return WMX_EINVAL;
    if (n_launches) *n_launches = h->ctl_calls;
    if (seconds) *seconds = h->ctl_seconds;
    h->ctl_calls = 0;
    h->ctl_seconds = 0.0;
    return 0;
}

int wmx_aec_timing(wmx_aec *h, int *n_launches, double *far_ms, double *near_ms) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    double f = 0, nr = 0;
    const size_t n = h->tev_used / 4;
    for (size_t i = 0; i < n; i++) {
        float a = 0.f, b = 0.f;
        WMX_HIP(hipEventSynchronize(h->tev[4 * i + 3]));
        WMX_HIP(hipEventElapsedTime(&a, h->tev[4 * i], h->tev[4 * i + 1]));
        WMX_HIP(hipEventElapsedTime(&b, h->tev[4 * i + 2], h->tev[4 * i + 3]));
        f += a;
        nr += b;
    }
    h->tev_used = 0;
    if (n_launches) *n_launches = (int)n;
    if (far_ms) *far_ms = f;
    if (near_ms) *near_ms = nr;
    return 0;
}

}  // extern "C"

// The same on the DEVICE (device pointers), through the kernel's own out-of-line aec_powf: what the vector ALU's fused
// multiply-adds, conversions and subnormals make of the routine, swept against the host's powf by tests/test_libm_gpu.py.
namespace wmx {
namespace {
__global__ void dbg_pow_kernel(const float *__restrict__ x, const float *__restrict__ e, float *__restrict__ y, size_t n,
                               const PowTables *__restrict__ tab) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = aec_powf(x[i], e[i], tab);
}
}  // namespace
}  // namespace wmx

extern "C" int wmx_debug_pow_device(const float *d_x, const float *d_e, float *d_y, size_t n, void *stream) {
    using namespace wmx;
    if (!d_x || !d_e || !d_y) return WMX_EINVAL;
    if (n == 0) return 0;
    if (current_device() < 0) return WMX_ENODEV;
    PowTables pt;
    pow_tables(&pt);
    PowTables *d_tab = nullptr;
    WMX_HIP(hipMalloc(reinterpret_cast<void **>(&d_tab), sizeof(pt)));
    hipError_t e = hipMemcpy(d_tab, &pt, sizeof(pt), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(dbg_pow_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), d_x, d_e, d_y, n, d_tab);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(as_stream(stream));
    (void)hipFree(d_tab);
    return e == hipSuccess ? 0 : hip_fail(e, "wmx_debug_pow_device", __FILE__, __LINE__);
}

// Host-side evaluation of the AEC's table-driven powf (libm_dev.h), same source as the kernel, for CPU sweeps against
// glibc's powf (tests/test_libm_tables.py).
extern "C" int wmx_debug_pow(const float *x, const float *e, float *y, size_t n) {
    static wmx::PowTables tab;
    static bool init = false;
    if (!init) {
        wmx::pow_tables(&tab);
        init = true;
    }
    if (!x || !e || !y) return WMX_EINVAL;
    for (size_t i = 0; i < n; i++) y[i] = wmx::fast_pow(x[i], e[i], &tab);
    return 0;
}
