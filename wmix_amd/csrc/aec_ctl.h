// aec_ctl.h -- host-side control plane of the batched AEC.
//
// Everything in the reference's AEC that decides WHERE data goes -- the start-up state machine
// and delay filter of ProcessNormal / EstBufDelayNormal (W:modules/audio_processing/aec/
// echo_cancellation.c:599-747, 821-872), the far/near/out ring buffers with their negative
// read-pointer moves (W:common_audio/ring_buffer.c:25-247; aec_core.c:1690-1717, 1719-1850), the
// block counters (noiseEstCtr, delayEstCtr, xfBufBlockPos) and the comfort-noise random
// generator (seed*69069+1, randomization_functions.c:87-112) -- depends only on the call
// pattern (packet sizes, reported delay), never on the audio.  All streams of a batch are
// driven in lockstep, so this logic runs ONCE per packet on the host, with indices only, and is
// handed to the kernels as a small plan; the GPU does the arithmetic (aec.hip).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>
#include <cstring>

namespace wmx {

constexpr int kAecPart = 64, kAecPart1 = 65, kAecFrame = 80;
constexpr int kAecFarBlocks = 250;   // kBufSizePartitions, aec_core.c:38
constexpr int kAecPreLen = 128 + 320;  // far_pre_buf: PART_LEN2 + kResamplerBufferSize, echo_cancellation.c:147
constexpr int kAecRing = kAecFrame + kAecPart;  // nearFrBuf / outFrBuf, aec_core.c:1361,1368
// Consumed-block history kept for the kernels.  A launch's far kernel writes the rows of ALL its blocks before the near kernel reads,
// for each block n, the rows n .. n - 11 (12 partitions; the delayed rows n - delayIdx lie inside): the rows n_first - 11 .. n_last
// must be distinct, blocks per launch + 11 <= kAecHist.  8 packets of 160 samples are 20 blocks: 31 rows.  (64 rows and 16 packets
// until round 5: the history is three quarters of what a far-end still costs, see aec.hip AecFarBufs.)
constexpr int kAecHist = 32;
constexpr int kAecMaxPktPerLaunch = 8;
static_assert((kAecHist & (kAecHist - 1)) == 0 && kAecMaxPktPerLaunch * 160 / 64 + 11 <= kAecHist, "history rows of one launch");

enum : int { kAecFlagNoiseMin = 1, kAecFlagNoiseInit = 2, kAecFlagDelayEst = 4,
             kAecFlagFarUnwritten = 8 };  // the far slot consumed has never been written: the reference reads its ring's zeroed storage

struct AecBlkPlan {
    int near_rd;   // near ring position of the block's 64 samples
    int out_wr;    // out ring position for the block's 64 output samples
    int hist_n;    // sequence number of the consumed far block (history slot = hist_n % kAecHist)
    int far_slot;  // far ring slot consumed (plain and windowed rings move together)
    int flags;
    int pad[3];
};
static_assert(sizeof(AecBlkPlan) == 32, "AecBlkPlan layout");

// The comfort-noise phases (aec_core.c:476-489): u = (int16)(seed >> 16) / 32768 -> cosf / sinf of 2 pi u.  The generator's state
// has 31 bits, so seed >> 16 takes 32 768 values: the table holds the host libm's cosf / sinf for every one of them, evaluated
// with the reference's own float expressions -- the values a handle of the reference computes per block, looked up instead.
constexpr int kAecNoiseTab = 32768;
struct AecNoiseEntry {
    float c, s;
};
inline void aec_noise_table(AecNoiseEntry *t) {
    const float pi2 = 6.28318530717959f;
    for (int i = 0; i < kAecNoiseTab; i++) {
        const float r = ((float)(int16_t)i) / 32768;
        const float tmp = pi2 * r;
        t[i].c = cosf(tmp);
        t[i].s = sinf(tmp);
    }
}
// Every handle's generator starts from the same state (aec->seed = 777, aec_core.c:1670) and ComfortNoise draws 64 numbers per
// block, unconditionally (aec_core.c:476-480).  The generator's state is part of the STREAM's state (AS_NSEED in aec.hip): the near
// kernel reaches lane l's draw with one multiply-add (aec_lcg_jump) and moves the state on by the 64-draw step.
// k draws in one step: seed_k = (seed * a_k + c_k) mod 2^31 with a_k = 69069^k, c_k = 1 + 69069 + ... + 69069^(k-1)
// (arithmetic mod 2^32, masked: the low 31 bits of a product depend on the low 31 bits of its factors only)
inline void aec_lcg_jump(int k, uint32_t *a_k, uint32_t *c_k) {
    uint32_t a = 1u, c = 0u;
    for (int i = 0; i < k; i++) {
        c = c * 69069u + 1u;
        a = a * 69069u;
    }
    *a_k = a;
    *c_k = c;
}
struct AecSubPlan {
    int near_wr;   // near ring position where the 80 new samples go
    int n_blocks;  // blocks processed in this 80-sample sub-frame
    int first_blk; // index into blk[]
    int out_rd;    // out ring position the 80 output samples are read from (after stuffing)
};
struct AecPartPlan {
    int pre_rd;    // far_pre ring position of the partition's 128 samples
    int far_slot;  // far ring slot written
};
struct AecPlan {
    int has_far;       // BufferFarend part present: write `far_n` samples at pre_wr, then n_part partitions
    int far_n, pre_wr, n_part;
    int has_near;      // Process part present
    int passthrough;   // start-up phase: out = near
    int n_sub, n_blk;
    AecSubPlan sub[2];
    AecPartPlan part[4];
    AecBlkPlan blk[4];
};

// index-only ring buffer: same arithmetic as W:common_audio/ring_buffer.c:112-247
struct RingIdx {
    int count = 0, rd = 0, wr = 0, diff_wrap = 0;
    void init(int n) {
        count = n;
        rd = wr = 0;
        diff_wrap = 0;
    }
    bool same_as(const RingIdx &o) const { return count == o.count && rd == o.rd && wr == o.wr && diff_wrap == o.diff_wrap; }
    int avail_read() const { return diff_wrap ? count - rd + wr : wr - rd; }
    int avail_write() const { return count - avail_read(); }
    int move_read(int n) {
        const int fr = avail_write(), readable = avail_read();
        int pos = rd;
        if (n > readable) n = readable;
        if (n < -fr) n = -fr;
        pos += n;
        if (pos > count) {
            pos -= count;
            diff_wrap = 0;
        }
        if (pos < 0) {
            pos += count;
            diff_wrap = 1;
        }
        rd = pos;
        return n;
    }
    // returns the number of elements written; *pos = first slot (mod count)
    int write(int n, int *pos) {
        const int fr = avail_write(), w = fr < n ? fr : n;
        int left = w;
        const int margin = count - wr;
        *pos = wr % count;
        if (w > margin) {
            wr = 0;
            left -= margin;
            diff_wrap = 1;
        }
        wr += left;
        return w;
    }
    int read(int n, int *pos) {
        const int readable = avail_read(), k = readable < n ? readable : n;
        *pos = rd % count;
        move_read(k);
        return k;
    }
};

struct AecCtl {
    int fs = 0, mult = 1, rate_factor = 1;
    RingIdx near_fr, out_fr, far_buf, far_pre;
    int system_delay = 0, core_known_delay = 0;
    uint32_t blocks = 0;  // blocks planned since init (a counter for the sanitizer driver; nothing on the device depends on it)
    int hist_n = 0;
    int far_writes = 0;   // partitions written into the far ring so far, saturating at its size: slots [far_writes, 250) still hold the
                          //   zeros of WebRtc_InitBuffer (the ring is written slot after slot from 0)
    // Aec wrapper (echo_cancellation_internal.h:17-65)
    int bufSizeStart = 0, knownDelay = 0, sum = 0, timeForDelayChange = 0, startup_phase = 1, checkBuffSize = 1;
    short counter = 0, firstVal = 0, checkBufSizeCtr = 0, msInSndCardBuf = 0, filtDelay = -1, lastDelayDiff = 0;

    // Every word that decides the plane's future equal: two such planes called alike (same packets, same reported delay) stay equal
    // for ever and hand out the same plans -- control planes are index arithmetic on the call pattern, never on audio.  (`blocks`
    // is a counter nothing reads.)
    bool same_as(const AecCtl &o) const {
        return fs == o.fs && mult == o.mult && rate_factor == o.rate_factor && near_fr.same_as(o.near_fr) && out_fr.same_as(o.out_fr) &&
               far_buf.same_as(o.far_buf) && far_pre.same_as(o.far_pre) && far_writes == o.far_writes && system_delay == o.system_delay &&
               core_known_delay == o.core_known_delay && hist_n == o.hist_n && bufSizeStart == o.bufSizeStart && knownDelay == o.knownDelay &&
               sum == o.sum && timeForDelayChange == o.timeForDelayChange && startup_phase == o.startup_phase &&
               checkBuffSize == o.checkBuffSize && counter == o.counter && firstVal == o.firstVal && checkBufSizeCtr == o.checkBufSizeCtr &&
               msInSndCardBuf == o.msInSndCardBuf && filtDelay == o.filtDelay && lastDelayDiff == o.lastDelayDiff;
    }

    void init(int freq) {  // WebRtcAec_Init echo_cancellation.c:179-275 + InitAec aec_core.c:1527-1688
        *this = AecCtl();
        fs = freq;
        mult = freq / 8000;
        rate_factor = freq / 8000;
        near_fr.init(kAecRing);
        out_fr.init(kAecRing);
        far_buf.init(kAecFarBlocks);
        far_pre.init(kAecPreLen);
        far_pre.move_read(-kAecPart);  // "Start overlap", echo_cancellation.c:227
    }

    int move_far_read(int n) {  // aec_core.c:1709-1717 (plain and windowed rings stay in lockstep)
        const int moved = far_buf.move_read(n);
        system_delay -= moved * kAecPart;
        return moved;
    }

    // WebRtcAec_BufferFarend, echo_cancellation.c:278-339
    int buffer_farend(int n, AecPlan *pl) {
        if (n != 80 && n != 160) return -1;
        pl->has_far = 1;
        pl->far_n = n;
        system_delay += n;
        far_pre.write(n, &pl->pre_wr);
        pl->n_part = 0;
        while (far_pre.avail_read() >= 2 * kAecPart) {
            AecPartPlan &pp = pl->part[pl->n_part++];
            far_pre.read(2 * kAecPart, &pp.pre_rd);
            if (far_buf.avail_write() < 1) move_far_read(1);  // BufferFarendPartition aec_core.c:1693-1695
            far_buf.write(1, &pp.far_slot);
            if (far_writes < kAecFarBlocks) far_writes++;
            far_pre.move_read(-kAecPart);
        }
        return 0;
    }

    void est_buf_delay() {  // echo_cancellation.c:821-872
        const int nSamp = msInSndCardBuf * 8 * rate_factor;
        int cur = nSamp - system_delay;
        cur += kAecFrame * rate_factor;
        if (cur < kAecPart) cur += move_far_read(1) * kAecPart;
        filtDelay = filtDelay < 0 ? (short)0 : filtDelay;
        {
            const short f = (short)(0.8 * filtDelay + 0.2 * cur);
            filtDelay = f > 0 ? f : (short)0;
        }
        const int diff = filtDelay - knownDelay;
        if (diff > 224) {
            if (lastDelayDiff < 96)
                timeForDelayChange = 0;
            else
                timeForDelayChange++;
        } else if (diff < 96 && knownDelay > 0) {
            if (lastDelayDiff > 224)
                timeForDelayChange = 0;
            else
                timeForDelayChange++;
        } else {
            timeForDelayChange = 0;
        }
        lastDelayDiff = (short)diff;
        if (timeForDelayChange > 25) {
            const int k = (int)filtDelay - 160;
            knownDelay = k > 0 ? k : 0;
        }
    }

    void plan_block(AecPlan *pl) {  // the index side of ProcessBlock + NonLinearProcessing (aec_core.c:1143-1351, 911-1141)
        AecBlkPlan &b = pl->blk[pl->n_blk++];
        near_fr.read(kAecPart, &b.near_rd);
        far_buf.read(1, &b.far_slot);
        b.hist_n = hist_n;
        hist_n = (hist_n + 1) & 0x3fffffff;  // only differences modulo kAecHist matter; stays non-negative for ever
        // (noiseEstCtr and delayEstCtr count the STREAM's blocks: aec.hip AS_NOISECTR / AS_DELAYCTR)
        b.flags = b.far_slot >= far_writes ? kAecFlagFarUnwritten : 0;
        blocks++;
        out_fr.write(kAecPart, &b.out_wr);
    }

    // WebRtcAec_Process -> ProcessNormal (echo_cancellation.c:341-409, 599-747) + ProcessFrames (aec_core.c:1719-1850)
    int process(int n, int ms_in_snd_card_buf, AecPlan *pl) {
        int ret = 0;
        if (n != 80 && n != 160) return -1;
        short ms = (short)ms_in_snd_card_buf;
        if (ms < 0) {
            ms = 0;
            ret = -1;
        } else if (ms > 500) {
            ret = -1;
        }
        ms = ms > 500 ? (short)500 : ms;
        ms = (short)(ms + 10);
        msInSndCardBuf = ms;
        pl->has_near = 1;
        pl->passthrough = 0;
        pl->n_sub = 0;
        pl->n_blk = 0;
        const short nBlocks10ms = (short)(n / (kAecFrame * rate_factor));
        if (startup_phase) {
            pl->passthrough = 1;
            if (checkBuffSize) {
                checkBufSizeCtr++;
                if (counter == 0) {
                    firstVal = msInSndCardBuf;
                    sum = 0;
                }
                const double lim = 0.2 * msInSndCardBuf;
                if (std::abs(firstVal - msInSndCardBuf) < (lim > 8 ? lim : 8)) {
                    sum += msInSndCardBuf;
                    counter++;
                } else {
                    counter = 0;
                }
                if (counter * nBlocks10ms >= 6) {
                    const int v = (3 * sum * rate_factor * 8) / (4 * counter * kAecPart);
                    bufSizeStart = v < 62 ? v : 62;
                    checkBuffSize = 0;
                }
                if (checkBufSizeCtr * nBlocks10ms > 50) {
                    const int v = (msInSndCardBuf * rate_factor * 3) / 40;
                    bufSizeStart = v < 62 ? v : 62;
                    checkBuffSize = 0;
                }
            }
            if (!checkBuffSize) {
                const int overhead = system_delay / kAecPart - bufSizeStart;
                if (overhead == 0) {
                    startup_phase = 0;
                } else if (overhead > 0) {
                    move_far_read(overhead);
                    startup_phase = 0;
                }
            }
            return ret;
        }
        est_buf_delay();
        for (int j = 0; j < n; j += kAecFrame) {
            AecSubPlan &sp = pl->sub[pl->n_sub++];
            near_fr.write(kAecFrame, &sp.near_wr);
            if (system_delay < kAecFrame) move_far_read(-(mult + 1));
            {
                const int move = (core_known_delay - knownDelay - 32) / kAecPart;
                const int moved = far_buf.move_read(move);
                core_known_delay -= moved * kAecPart;
            }
            sp.first_blk = pl->n_blk;
            sp.n_blocks = 0;
            while (near_fr.avail_read() >= kAecPart) {
                plan_block(pl);
                sp.n_blocks++;
            }
            system_delay -= kAecFrame;
            const int avail = out_fr.avail_read();
            if (avail < kAecFrame) out_fr.move_read(avail - kAecFrame);
            out_fr.read(kAecFrame, &sp.out_rd);
        }
        return ret;
    }
};

// ---- coalescing (wmx_aec_coalesce): when do two control planes make the same plans from here on?
// What decides a control plane's future, positions taken out: two planes with equal keys make the same plans up to a rotation of
// their rings, as long as they are called with the same delays.  (The start-up fields of AecCtl are dead once startup_phase is 0;
// the comfort-noise generator belongs to the streams once cohorts have been merged; hist_n and the ring positions are what the
// rotation absorbs.)  tools_dev/san/host_ctl_san.cpp drives pairs of planes for thousands of packets behind an equal key.
struct AecCoKey {
    int v[11];
    bool operator==(const AecCoKey &o) const {
        for (int i = 0; i < 11; i++)
            if (v[i] != o.v[i]) return false;
        return true;
    }
};
inline bool aec_co_key(const AecCtl &c, AecCoKey *k) {
    if (c.startup_phase) return false;
    const int v[11] = {c.near_fr.avail_read(), c.out_fr.avail_read(), c.far_buf.avail_read(), c.far_pre.avail_read(), c.system_delay,
                       c.core_known_delay, c.knownDelay, c.timeForDelayChange, (int)c.msInSndCardBuf, (int)c.filtDelay, (int)c.lastDelayDiff};
    for (int i = 0; i < 11; i++) k->v[i] = v[i];
    return true;
}
// the rotations between cohort a (`into`) and cohort b (`from`)
struct AecPairCheck {
    int a, b;
    int d_pre;       // b's far_pre positions = a's + d_pre   (mod kAecPreLen)
    int d_far;       // b's far ring slots    = a's + d_far   (mod kAecFarBlocks)
    int d_hist;      // b's history rows      = a's + d_hist  (mod kAecHist)
    int d_near, d_out;  // a's near / out ring positions = b's + d (mod kAecRing): the rotation a merge applies to b's member streams
    int pad;
};
inline int aec_mod(int x, int m) {
    x %= m;
    return x < 0 ? x + m : x;
}
inline void aec_co_pair(const AecCtl &a, const AecCtl &b, int ia, int ib, AecPairCheck *pc) {
    pc->a = ia;
    pc->b = ib;
    pc->d_pre = aec_mod(b.far_pre.rd - a.far_pre.rd, kAecPreLen);
    pc->d_far = aec_mod(b.far_buf.rd - a.far_buf.rd, kAecFarBlocks);
    pc->d_hist = aec_mod(b.hist_n - a.hist_n, kAecHist);
    pc->d_near = aec_mod(a.near_fr.rd - b.near_fr.rd, kAecRing);
    pc->d_out = aec_mod(a.out_fr.rd - b.out_fr.rd, kAecRing);
    pc->pad = 0;
}

// ---------------------------------------------------------------- control-plane classes (host bookkeeping, no HIP)
// A control plane is index arithmetic on the call pattern, never on audio: cohorts that were started at the same point and are
// called alike have EQUAL planes for ever, although their far-ends differ.  H is anything with
//     std::vector<Ctl> ctl;  std::vector<int32_t> lead;  std::vector<uint8_t> live;  bool cls_dirty;     (Ctl: AecCtl or AecmCtl --
//     anything with same_as(); lead.size() is the number of cohort ids in use)
// (wmx_aec in aec.hip; a plain struct in tools_dev/san/host_ctl_san.cpp, where these run under ASan / UBSan against a model that
// keeps one plane per cohort).  lead[g] = the cohort whose plane stands for g's; ctl[g] of a follower is stale.
template <class H>
inline auto aec_ctl(H *h, int g) -> decltype((h->ctl[0])) { return h->ctl[(size_t)h->lead[(size_t)g]]; }
// cohort g leaves its class with an up-to-date plane of its own (a leader hands the class over to its first follower)
template <class H>
inline void aec_ctl_own(H *h, int g) {
    const int l = h->lead[(size_t)g];
    if (l != g) {
        h->ctl[(size_t)g] = h->ctl[(size_t)l];
        h->lead[(size_t)g] = g;
        h->cls_dirty = true;
        return;
    }
    int heir = -1;
    for (int x = 0; x < (int)h->lead.size(); x++)
        if (x != g && h->lead[(size_t)x] == g) {
            if (heir < 0) {
                heir = x;
                h->ctl[(size_t)x] = h->ctl[(size_t)g];
            }
            h->lead[(size_t)x] = heir;
            h->cls_dirty = true;
        }
}
// cohort g (a leader of itself alone, its plane just rewritten: aec_init, an import) joins a class whose plane is equal, if one of
// the first few hundred leaders has it -- planes made at the same point of the packet sequence (a bounded search: a miss costs
// a control plane of its own, nothing else)
template <class H>
inline void aec_ctl_join(H *h, int g) {
    int seen = 0;
    for (int x = 0; x < (int)h->lead.size() && seen < 256; x++) {
        if (x == g || h->lead[(size_t)x] != x || !h->live[(size_t)x]) continue;
        seen++;
        if (h->ctl[(size_t)x].same_as(h->ctl[(size_t)g])) {
            h->lead[(size_t)g] = x;
            h->cls_dirty = true;
            return;
        }
    }
}
// in front of a launch: a follower that is called differently from its leader in THIS call (switched on / off alone, another
// reported delay) takes a plane of its own first.  cohort_on may be null (all on).
template <class H>
inline void aec_classes_split(H *h, const int32_t *delay_ms, const uint8_t *cohort_on) {
    for (int g = 0; g < (int)h->lead.size(); g++) {
        const int l = h->lead[(size_t)g];
        if (l == g || !h->live[(size_t)g]) continue;
        const bool on_g = !cohort_on || cohort_on[g], on_l = !cohort_on || cohort_on[l];
        if (on_g != on_l || (on_g && delay_ms[g] != delay_ms[l])) aec_ctl_own(h, g);
    }
}
// the leaders, compact, and every cohort's class index
template <class H>
inline void aec_classes_list(const H *h, std::vector<int32_t> &leaders, std::vector<int32_t> &plan_of) {
    const int G = (int)h->lead.size();
    plan_of.assign((size_t)G, 0);
    leaders.clear();
    for (int g = 0; g < G; g++)
        if (h->lead[(size_t)g] == g) {
            plan_of[(size_t)g] = (int32_t)leaders.size();
            leaders.push_back(g);
        }
    for (int g = 0; g < G; g++) plan_of[(size_t)g] = plan_of[(size_t)h->lead[(size_t)g]];
}

}  // namespace wmx
