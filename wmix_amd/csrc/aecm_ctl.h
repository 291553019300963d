// aecm_ctl.h -- host-side control plane of the batched AECM.
//
// As in the float AEC (aec_ctl.h), everything in the reference's AECM that decides WHERE data goes depends on the call
// pattern only, never on the audio: the start-up phase and far-end buffer bookkeeping of WebRtcAecm_BufferFarend /
// WebRtcAecm_Process with DelayComp and EstBufDelay (W:modules/audio_processing/aecm/echo_control_mobile.c:233-275,
// 277-482, 633-720), and the 80 -> 64 sample re-blocking rings of WebRtcAecm_ProcessFrame (aecm_core.c:569-664) with the
// output ring's read-pointer rewind.  All streams of a batch are driven in lockstep against one shared far-end, so this
// runs once per packet on the host with indices only (RingIdx = ring_buffer.c arithmetic) and is handed to the kernels as
// a plan.  What IS data dependent -- the binary-spectrum delay estimate -- runs per stream on the GPU (aecm.hip).
#pragma once
#include <cstdint>
#include <cstdlib>
#include "aec_ctl.h"  // RingIdx

namespace wmx {

constexpr int kAecmFrame = 80, kAecmPart = 64;
constexpr int kAecmFarRing = 50 * kAecmFrame;      // BUF_SIZE_FRAMES * FRAME_LEN, echo_control_mobile.c:24,31
constexpr int kAecmFrameRing = kAecmFrame + kAecmPart;  // far / near / out frame rings, aecm_core.c:218-245
constexpr int kAecmHist = 256;                     // device history of far spectra, >= MAX_DELAY (100) + blocks per launch
constexpr int kAecmMaxPktPerLaunch = 32;           // <= 96 blocks per launch

struct AecmFramePlan {
    int far_src;       // far ring position of the frame's 80 samples, or -1: take farendOld[old_slot]
    int old_slot;      // 0 / 1: the slot refreshed (far_src >= 0) or replayed (far_src < 0), echo_control_mobile.c:422-433
    int ring_w;        // position in the 144-sample far / near frame rings where the 80 new samples go
    int n_blocks;      // 64-sample blocks completed by this frame (0..2)
    int blk_r[2];      // frame-ring position of each block's samples (far and near rings move together)
    int blk_out_w[2];  // out ring position of each block's output
    int blk_t[2];      // absolute block number (history slot = t % kAecmHist)
    int out_r;         // out ring position the frame's 80 output samples are read from (after the rewind)
    int pad[5];
};
struct AecmPlan {
    int has_far, far_w, far_n;  // BufferFarend: far_n samples of the packet go to the far ring at far_w
    int has_near, passthrough, n_frames;
    int discard_out;  // the packet is processed but its output is not copied out (WebRtcAecm_Process returned -1)
    int pad[1];
    AecmFramePlan fr[2];
};

struct AecmCtl {
    int fs = 0, mult = 1;
    RingIdx farend, frame_ring, out_ring;
    int known_delay = 0, time_for_delay_change = 0, ec_startup = 1, check_buff_size = 1;
    short buf_size_start = 0, counter = 0, sum = 0, first_val = 0, check_buf_size_ctr = 0, ms_in_snd = 0, filt_delay = 0, last_delay_diff = 0;
    int block_t = 0;

    // every word that decides the plane's future equal (see AecCtl::same_as): such planes, called alike, stay equal for ever
    bool same_as(const AecmCtl &o) const {
        return fs == o.fs && mult == o.mult && farend.same_as(o.farend) && frame_ring.same_as(o.frame_ring) && out_ring.same_as(o.out_ring) &&
               known_delay == o.known_delay && time_for_delay_change == o.time_for_delay_change && ec_startup == o.ec_startup &&
               check_buff_size == o.check_buff_size && buf_size_start == o.buf_size_start && counter == o.counter && sum == o.sum &&
               first_val == o.first_val && check_buf_size_ctr == o.check_buf_size_ctr && ms_in_snd == o.ms_in_snd &&
               filt_delay == o.filt_delay && last_delay_diff == o.last_delay_diff && block_t == o.block_t;
    }

    void init(int freq) {  // WebRtcAecm_Init echo_control_mobile.c:177-231, WebRtcAecm_InitCore aecm_core.c:401-546
        *this = AecmCtl();
        fs = freq;
        mult = freq / 8000;
        farend.init(kAecmFarRing);
        frame_ring.init(kAecmFrameRing);
        out_ring.init(kAecmFrameRing);
    }

    void delay_comp() {  // echo_control_mobile.c:693-720
        const int n_far = farend.avail_read(), n_snd = ms_in_snd * 8 * mult, delay_new = n_snd - n_far;
        if (delay_new > 256 - kAecmFrame * mult) {
            int add = (n_snd >> 1) - n_far > kAecmFrame ? (n_snd >> 1) - n_far : kAecmFrame;
            add = add < 10 * kAecmFrame ? add : 10 * kAecmFrame;
            farend.move_read(-add);
        }
    }

    void est_buf_delay() {  // echo_control_mobile.c:633-691
        const short n_far = (short)farend.avail_read(), n_snd = (short)(ms_in_snd * 8 * mult);
        short delay_new = (short)(n_snd - n_far);
        if (delay_new < kAecmFrame) {
            farend.move_read(kAecmFrame);
            delay_new = (short)(delay_new + kAecmFrame);
        }
        const int f = (8 * filt_delay + 2 * delay_new) / 10;
        filt_delay = (short)(0 > f ? 0 : f);
        const short diff = (short)(filt_delay - known_delay);
        if (diff > 224) {
            time_for_delay_change = last_delay_diff < 96 ? 0 : time_for_delay_change + 1;
        } else if (diff < 96 && known_delay > 0) {
            time_for_delay_change = last_delay_diff > 224 ? 0 : time_for_delay_change + 1;
        } else {
            time_for_delay_change = 0;
        }
        last_delay_diff = diff;
        if (time_for_delay_change > 25) known_delay = (int)filt_delay - 160 > 0 ? (int)filt_delay - 160 : 0;
    }

    int buffer_farend(int n, AecmPlan *pl) {  // echo_control_mobile.c:233-275
        if (n != 80 && n != 160) return -1;
        if (!ec_startup) delay_comp();
        pl->has_far = 1;
        pl->far_n = farend.write(n, &pl->far_w);
        return 0;
    }

    // WebRtcAecm_ProcessFrame's ring bookkeeping, aecm_core.c:569-664
    void process_frame(AecmFramePlan *fp) {
        int pos;
        frame_ring.write(kAecmFrame, &fp->ring_w);
        fp->n_blocks = 0;
        while (frame_ring.avail_read() >= kAecmPart) {
            frame_ring.read(kAecmPart, &pos);
            fp->blk_r[fp->n_blocks] = pos;
            out_ring.write(kAecmPart, &fp->blk_out_w[fp->n_blocks]);
            fp->blk_t[fp->n_blocks] = block_t++;
            fp->n_blocks++;
        }
        const int size = out_ring.avail_read();
        if (size < kAecmFrame) out_ring.move_read(size - kAecmFrame);
        out_ring.read(kAecmFrame, &fp->out_r);
    }

    // returns what WebRtcAecm_Process returns: 0, or -1 for a delay outside [0, 500] ms -- AFTER the packet has been
    // processed; the wmix wrapper then drops the packet's output and stops (src/webrtc.c:382-387)
    int process(int n, int ms, AecmPlan *pl) {  // echo_control_mobile.c:277-482
        int ret = 0;
        if (n != 80 && n != 160) return -1;
        if (ms < 0)
            ms = 0, ret = -1;
        else if (ms > 500)
            ms = 500, ret = -1;
        ms += 10;
        ms_in_snd = (short)ms;
        const short n_frames = (short)(n / kAecmFrame), n_blocks = (short)(n_frames / mult);
        pl->has_near = 1;
        pl->n_frames = n_frames;
        if (ec_startup) {
            pl->passthrough = 1;
            const short filled = (short)((short)farend.avail_read() / kAecmFrame);
            if (check_buff_size) {
                check_buf_size_ctr++;
                if (counter == 0) {
                    first_val = ms_in_snd;
                    sum = 0;
                }
                const double lim = 0.2 * ms_in_snd > 8 ? 0.2 * ms_in_snd : 8;
                if (abs(first_val - ms_in_snd) < lim) {
                    sum = (short)(sum + ms_in_snd);
                    counter++;
                } else {
                    counter = 0;
                }
                if (counter * n_blocks >= 6) {
                    const int v = (3 * sum * mult) / (counter * 40);
                    buf_size_start = (short)(v < 50 ? v : 50);
                    check_buff_size = 0;
                }
                if (check_buf_size_ctr * n_blocks > 50) {
                    const int v = (3 * ms_in_snd * mult) / 40;
                    buf_size_start = (short)(v < 50 ? v : 50);
                    check_buff_size = 0;
                }
            }
            if (!check_buff_size) {
                if (filled == buf_size_start) {
                    ec_startup = 0;
                } else if (filled > buf_size_start) {
                    farend.move_read(farend.avail_read() - (int)buf_size_start * kAecmFrame);
                    ec_startup = 0;
                }
            }
        } else {
            pl->passthrough = 0;
            for (short i = 0; i < n_frames; i++) {
                AecmFramePlan *fp = &pl->fr[i];
                const short filled = (short)((short)farend.avail_read() / kAecmFrame);
                fp->old_slot = i;
                if (filled > 0)
                    farend.read(kAecmFrame, &fp->far_src);
                else
                    fp->far_src = -1;
                if ((i == 0 && fs == 8000) || (i == 1 && fs == 16000)) est_buf_delay();
                process_frame(fp);
            }
        }
        return ret;
    }
};

// ---- coalescing (wmx_aecm_coalesce), as for the float AEC (aec_ctl.h): what decides a control plane's future, positions taken out.
// The AECM's plane has no periodic counters: two planes past their start-up with the same fill levels and delay filter make the same
// plans up to a rotation of their rings and of the far-end history's block numbers.
struct AecmCoKey {
    int v[8];
    bool operator==(const AecmCoKey &o) const {
        for (int i = 0; i < 8; i++)
            if (v[i] != o.v[i]) return false;
        return true;
    }
};
inline bool aecm_co_key(const AecmCtl &c, AecmCoKey *k) {
    if (c.ec_startup) return false;
    const int v[8] = {c.farend.avail_read(), c.frame_ring.avail_read(), c.out_ring.avail_read(), c.known_delay, c.time_for_delay_change,
                      (int)c.ms_in_snd, (int)c.filt_delay, (int)c.last_delay_diff};
    for (int i = 0; i < 8; i++) k->v[i] = v[i];
    return true;
}
struct AecmPairCheck {
    int a, b;       // cohorts: `b` (from) would join `a` (into)
    int d_ring;     // b's far ring positions    = a's + d_ring   (mod kAecmFarRing)
    int d_frame;    // b's frame ring positions  = a's + d_frame  (mod kAecmFrameRing); a's = b's + (ring - d_frame): the members' near rings
    int d_out;      // b's out ring positions    = a's + d_out    (mod kAecmFrameRing)
    int d_hist;     // b's history slots         = a's + d_hist   (mod kAecmHist)
    int pad[2];
};
inline void aecm_co_pair(const AecmCtl &a, const AecmCtl &b, int ia, int ib, AecmPairCheck *pc) {
    pc->a = ia;
    pc->b = ib;
    pc->d_ring = aec_mod(b.farend.rd - a.farend.rd, kAecmFarRing);
    pc->d_frame = aec_mod(b.frame_ring.rd - a.frame_ring.rd, kAecmFrameRing);
    pc->d_out = aec_mod(b.out_ring.rd - a.out_ring.rd, kAecmFrameRing);
    pc->d_hist = aec_mod(b.block_t - a.block_t, kAecmHist);
    pc->pad[0] = pc->pad[1] = 0;
}

}  // namespace wmx
