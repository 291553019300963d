// chain.hip -- the daemon's record heartbeat as ONE C call per tick (host code only: it sequences launches).
//
// wmix_shmem_write_circle (src/wmix.c:613-709) runs, per WMIX_INTERVAL_MS of captured audio and on one buffer in place,
//     ns_process -> aec_process2(far = playPkgBuff_get(...), near = out = buffer, delayms) -> agc_process -> vad_process
// each behind its webrtcEnable[] switch, creating the handle on first use and releasing it when the switch drops or
// recording idles (src/wmix.c:565-600, 617-618, 635-636, 683-684, 702-703).  wmx_chain is that heartbeat for a batch of
// streams: the four batched handles, one wmx_chain_process per tick launching their kernels back to back on the caller's
// HIP stream (no host synchronisation, nothing copied), and the per-stream lifetime calls forwarded to every stage.
// A C host (examples/host_chain.c) needs nothing else of the library for the chain.
#include <cstdlib>
#include <vector>
#include "wmx_internal.h"

struct wmx_chain {
    int device;  // first member of every handle (wmx_handle_device)
    int n_streams, chn, freq, interval_ms;
    unsigned stages;
    wmx_ns *ns;      // the float noise suppressor ...
    wmx_nsx *nsx;    // ... or the reference's MAKE_WEBRTC_NSX build of the same stage (WMX_CHAIN_NSX): at most one is set
    wmx_aec *aec;    // the float echo canceller ...
    wmx_aecm *aecm;  // ... or its AECM build (WMX_CHAIN_AECM)
    wmx_agc *agc;
    wmx_vad *vad;
    int pkg10;                       // int16 elements of one 10 ms packet of one stream (freq / 100 * chn)
    int aec_pkg, agc_pkg, vad_pkg;   // the stages' own packets, in int16 elements
    int n_cohorts;
    bool no_fork;                    // WMIX_AMD_CHAIN_NO_FORK, read once at create (developer A/B switch)
    std::vector<int32_t> zero_delays;  // what the daemon reports (delayms = 0), one per cohort, for callers that pass NULL
    // what a stage that is switched on later is made with (wmx_chain_set_stages): the arguments of wmx_chain_create_groups
    int agc_value, n_cohorts_made;
    std::vector<int32_t> stream_cohort_made;
    hipEvent_t gate_after_ns = nullptr;  // recorded behind the noise suppressor's launch by the next process call (wmx::chain_gate_after_ns)
};

namespace wmx {
// For the packet pipeline (pipe.hip): an event the NEXT wmx_chain_process call records on its stream between the noise suppressor and
// the echo canceller, once.  The suppressor moves 24 KB of state per stream at 70 % of the HBM peak; the canceller's near kernel is
// bound by arithmetic -- a device-to-host copy (a blit kernel of posted PCIe writes) belongs beside the latter.
void chain_gate_after_ns(wmx_chain *h, hipEvent_t ev) {
    if (h) h->gate_after_ns = ev;
}
// For the daemon's tick (tick.hip): the cohorts an echo canceller is made with when wmx_chain_set_stages switches one on LATER -- a
// tick made without a canceller has a chain of one cohort, but its groups' far-ends are there from the start.
void chain_cohorts_when_made(wmx_chain *h, int n_cohorts, const int32_t *stream_cohort) {
    if (!h) return;
    h->n_cohorts_made = n_cohorts;
    if (stream_cohort)
        h->stream_cohort_made.assign(stream_cohort, stream_cohort + h->n_streams);
    else
        h->stream_cohort_made.clear();
}
}  // namespace wmx

extern "C" {

int wmx_chain_destroy(wmx_chain *h) {
    WMX_ON_DEVICE(h);
    if (!h) return 0;
    if (h->ns) wmx_ns_destroy(h->ns);
    if (h->nsx) wmx_nsx_destroy(h->nsx);
    if (h->aec) wmx_aec_destroy(h->aec);
    if (h->aecm) wmx_aecm_destroy(h->aecm);
    if (h->agc) wmx_agc_destroy(h->agc);
    if (h->vad) wmx_vad_destroy(h->vad);
    delete h;
    return 0;
}

}  // extern "C"

// Makes the stage handles that `stages` asks for and the chain does not have, in the heartbeat's order -- the same *_init calls; an
// unsupported format fails the way the reference's *_init returns NULL (the reference picks WebRtcNsx_* / WebRtcAecm_* by build
// switches, src/webrtc.c:512-521, 168-191: here two stage bits).  A handle that exists is left alone.
static int chain_make_stages(wmx_chain *h, unsigned stages) {
    const int n_streams = h->n_streams, chn = h->chn, freq = h->freq, interval_ms = h->interval_ms, n_cohorts = h->n_cohorts_made;
    const int32_t *stream_cohort = h->stream_cohort_made.empty() ? nullptr : h->stream_cohort_made.data();
    int rc = 0;
    if (rc == 0 && (stages & WMX_CHAIN_NS) && !h->ns && !h->nsx)
        rc = (stages & WMX_CHAIN_NSX) ? wmx_nsx_create(&h->nsx, n_streams, chn, freq) : wmx_ns_create(&h->ns, n_streams, chn, freq);
    if (rc == 0 && (stages & WMX_CHAIN_AEC) && !h->aec && !h->aecm) {
        rc = (stages & WMX_CHAIN_AECM) ? wmx_aecm_create_cohorts(&h->aecm, n_streams, chn, freq, interval_ms, n_cohorts)
                                       : wmx_aec_create_groups(&h->aec, n_streams, chn, freq, interval_ms, n_cohorts, stream_cohort);
        if (rc == 0 && h->aecm && stream_cohort && n_cohorts > 1) {
            // the fixed-point canceller takes memberships through its reset call: one call per cohort that has members
            std::vector<std::vector<int32_t>> members((size_t)n_cohorts);
            for (int s = 0; s < n_streams && rc == 0; s++) {
                if (stream_cohort[s] < 0 || stream_cohort[s] >= n_cohorts) {
                    wmx::set_error("wmx_chain_create_groups: stream %d in cohort %d of %d", s, stream_cohort[s], n_cohorts);
                    rc = WMX_EINVAL;
                } else {
                    members[(size_t)stream_cohort[s]].push_back(s);
                }
            }
            for (int c = 1; c < n_cohorts && rc == 0; c++)
                if (!members[(size_t)c].empty())
                    rc = wmx_aecm_reset_streams(h->aecm, members[(size_t)c].data(), (int)members[(size_t)c].size(), c, nullptr);
            if (rc == 0 && hipDeviceSynchronize() != hipSuccess) rc = WMX_ENODEV;
        }
        if (rc == 0) {
            h->n_cohorts = n_cohorts;
            h->zero_delays.assign((size_t)n_cohorts, 0);
        }
    }
    if (rc == 0 && (stages & WMX_CHAIN_AGC) && !h->agc) rc = wmx_agc_create(&h->agc, n_streams, chn, freq, interval_ms, h->agc_value);
    if (rc == 0 && (stages & WMX_CHAIN_VAD) && !h->vad) rc = wmx_vad_create(&h->vad, n_streams, chn, freq, interval_ms);
    if (rc != 0) return rc;
    h->stages = stages;
    h->aec_pkg = h->aec ? wmx_aec_packet_samples(h->aec) : (h->aecm ? wmx_aecm_packet_samples(h->aecm) : h->pkg10);
    h->agc_pkg = h->agc ? wmx_agc_packet_samples(h->agc) : h->pkg10;
    h->vad_pkg = h->vad ? wmx_vad_packet_samples(h->vad) : h->pkg10;
    return 0;
}

extern "C" {

// The heartbeat's switches at run time (webrtcEnable[], set by the daemon's message thread, src/wmix.c:1010-1050): a stage whose
// switch drops is RELEASED (src/wmix.c:783-813: *_release, the pointer zeroed) and one that comes on is made anew inside the next
// heartbeat (:617-618, 635-636, 683-684, 702-703: *_init on first use) -- fresh state for every stream, the cohorts the chain was
// created with, agc_value as agc_init's value (the daemon passes the volumeAgc of that moment, :684; < 0: the value the chain was
// made with).  Stages that stay on keep their state.  A control-plane call: the device is drained when a stage goes.
int wmx_chain_set_stages(wmx_chain *h, unsigned stages, int agc_value) {
    WMX_ON_DEVICE(h);
    if (!h || (stages & ~63u) != 0 || ((stages & WMX_CHAIN_NSX) && !(stages & WMX_CHAIN_NS)) || ((stages & WMX_CHAIN_AECM) && !(stages & WMX_CHAIN_AEC))) {
        wmx::set_error("wmx_chain_set_stages: stages=0x%x", stages);
        return WMX_EINVAL;
    }
    if (agc_value >= 0) h->agc_value = agc_value;
    // what goes: a stage switched off, or kept on in its OTHER build (float <-> fixed point)
    const bool ns_goes = h->ns && (!(stages & WMX_CHAIN_NS) || (stages & WMX_CHAIN_NSX));
    const bool nsx_goes = h->nsx && (!(stages & WMX_CHAIN_NS) || !(stages & WMX_CHAIN_NSX));
    const bool aec_goes = h->aec && (!(stages & WMX_CHAIN_AEC) || (stages & WMX_CHAIN_AECM));
    const bool aecm_goes = h->aecm && (!(stages & WMX_CHAIN_AEC) || !(stages & WMX_CHAIN_AECM));
    if (ns_goes) wmx_ns_destroy(h->ns), h->ns = nullptr;
    if (nsx_goes) wmx_nsx_destroy(h->nsx), h->nsx = nullptr;
    if (aec_goes) wmx_aec_destroy(h->aec), h->aec = nullptr;
    if (aecm_goes) wmx_aecm_destroy(h->aecm), h->aecm = nullptr;
    if (h->agc && !(stages & WMX_CHAIN_AGC)) wmx_agc_destroy(h->agc), h->agc = nullptr;
    if (h->vad && !(stages & WMX_CHAIN_VAD)) wmx_vad_destroy(h->vad), h->vad = nullptr;
    if (!h->aec && !h->aecm) {  // without a canceller a chain has one cohort
        h->n_cohorts = 1;
        h->zero_delays.assign(1, 0);
    }
    h->gate_after_ns = nullptr;
    return chain_make_stages(h, stages);
}
unsigned wmx_chain_stages(const wmx_chain *h) { return h ? h->stages : 0u; }

int wmx_chain_create(wmx_chain **out, int n_streams, int chn, int freq, int interval_ms, int agc_value, unsigned stages, int n_cohorts) {
    return wmx_chain_create_groups(out, n_streams, chn, freq, interval_ms, agc_value, stages, n_cohorts, nullptr);
}

// stream_cohort (HOST array of n_streams entries, or NULL = every stream in cohort 0): the cohort -- control plane and far-end --
// each stream belongs to from the start, e.g. the mix group whose playback a record stream hears (wmx_tick)
int wmx_chain_create_groups(wmx_chain **out, int n_streams, int chn, int freq, int interval_ms, int agc_value, unsigned stages, int n_cohorts,
                            const int32_t *stream_cohort) {
    if (!out) return WMX_EINVAL;
    *out = nullptr;
    // (no stage at all is a heartbeat with every webrtcEnable[] switch off: the package passes through untouched, src/wmix.c:613-709)
    if ((stages & ~63u) != 0 || n_cohorts < 1 || ((stages & WMX_CHAIN_NSX) && !(stages & WMX_CHAIN_NS)) ||
        ((stages & WMX_CHAIN_AECM) && !(stages & WMX_CHAIN_AEC))) {
        wmx::set_error("wmx_chain_create: stages=0x%x n_cohorts=%d", stages, n_cohorts);
        return WMX_EINVAL;
    }
    wmx_chain *h = new wmx_chain();
    if ((h->device = wmx::current_device()) < 0) {
        delete h;
        return WMX_ENODEV;
    }
    h->n_streams = n_streams;
    h->chn = chn;
    h->freq = freq;
    h->interval_ms = interval_ms;
    h->stages = stages;
    h->n_cohorts = n_cohorts;
    h->pkg10 = freq / 100 * chn;
    h->no_fork = getenv("WMIX_AMD_CHAIN_NO_FORK") != nullptr;
    h->zero_delays.assign((size_t)n_cohorts, 0);
    h->agc_value = agc_value;
    h->n_cohorts_made = n_cohorts;
    if (stream_cohort) h->stream_cohort_made.assign(stream_cohort, stream_cohort + n_streams);
    const int rc = chain_make_stages(h, stages);
    if (rc != 0) {
        wmx_chain_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

wmx_ns *wmx_chain_ns(wmx_chain *h) { return h ? h->ns : nullptr; }
wmx_aec *wmx_chain_aec(wmx_chain *h) { return h ? h->aec : nullptr; }
wmx_nsx *wmx_chain_nsx(wmx_chain *h) { return h ? h->nsx : nullptr; }
wmx_aecm *wmx_chain_aecm(wmx_chain *h) { return h ? h->aecm : nullptr; }
wmx_agc *wmx_chain_agc(wmx_chain *h) { return h ? h->agc : nullptr; }
wmx_vad *wmx_chain_vad(wmx_chain *h) { return h ? h->vad : nullptr; }

// One heartbeat.  n10: 10 ms packets per stream in this tick (the daemon's WMIX_FRAME_NUM is 2 of them); packet p of stream s
// at d_in / d_out + s * stream_stride + p * packet_stride.  The first enabled stage reads d_in and writes d_out, the others
// work on d_out in place (d_out == d_in is the daemon's case).  A stage whose own packet is longer than 10 ms (AEC at 8 kHz
// with a 20 ms interval, VAD with a 20 ms interval) needs the tick's packets contiguous (packet_stride == one packet);
// vad_process is ONE call per tick over the whole buffer, as in the heartbeat (so it analyses the tick's first packet only,
// SURVEY section 0 quirk 1).  delay_ms: one reported delay per cohort (NULL: 0, what the daemon passes).
int wmx_chain_process(wmx_chain *h, const int16_t *d_far, long far_packet_stride, const int16_t *d_in, int16_t *d_out, int n10,
                      long stream_stride, long packet_stride, const int32_t *delay_ms, const uint8_t *cohort_on, int32_t *cohort_rc,
                      void *stream) {
    return wmx_chain_process_groups(h, d_far, far_packet_stride, 0, d_in, d_out, n10, stream_stride, packet_stride, delay_ms, cohort_on, cohort_rc,
                                    stream);
}

// The same with a far-end PER COHORT: cohort c's 10 ms packet p at d_far + c * far_group_stride + p * far_packet_stride
// (far_group_stride 0: one far-end for all, wmx_chain_process).  aec_process2 takes the far-end per handle (src/webrtc.c:410); in the
// daemon it is playPkgBuff_get() of that daemon's own playback (src/wmix.c:651-657).
int wmx_chain_process_groups(wmx_chain *h, const int16_t *d_far, long far_packet_stride, long far_group_stride, const int16_t *d_in,
                             int16_t *d_out, int n10, long stream_stride, long packet_stride, const int32_t *delay_ms, const uint8_t *cohort_on,
                             int32_t *cohort_rc, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h || n10 < 0) {
        wmx::set_error("wmx_chain_process: bad argument");
        return WMX_EINVAL;
    }
    if (n10 == 0) return 0;
    if (!d_in || !d_out || ((h->stages & WMX_CHAIN_AEC) && !d_far)) {
        wmx::set_error("wmx_chain_process: null buffer");
        return WMX_EINVAL;
    }
    if ((h->stages & 15u) == 0) {  // every switch off: the heartbeat leaves the package as it is (src/wmix.c:613-709)
        if (d_in == d_out) return 0;
        wmx::set_error("wmx_chain_process: a chain with every stage off works in place (d_in == d_out)");
        return WMX_EINVAL;
    }
    const long total = (long)n10 * h->pkg10;  // int16 elements of the tick per stream
    const bool contiguous = packet_stride == h->pkg10;
    auto fits = [&](int pkg) { return total % pkg == 0 && (pkg <= h->pkg10 || contiguous); };
    const bool any_aec = h->aec || h->aecm;
    if ((any_aec && !fits(h->aec_pkg)) || (h->vad && !fits(h->vad_pkg)) || (any_aec && h->aec_pkg > h->pkg10 && far_packet_stride != h->pkg10)) {
        wmx::set_error("wmx_chain_process: %d x 10 ms with packet stride %ld does not hold whole %d / %d-sample packets", n10, packet_stride,
                       h->aec_pkg, h->vad_pkg);
        return WMX_EINVAL;
    }
    // vad_process averages interleaved channels over the WHOLE call in place and expands them again at the end
    // (src/webrtc.c:104-116, 145-150): with several VAD packets per tick the tick must lie in one piece per stream, or the
    // downmix would rewrite the neighbouring streams' rows (round-3 ADVICE).  Checked before anything is launched.
    if (h->vad && h->chn > 1 && total / h->vad_pkg > 1 && !(contiguous && (h->n_streams == 1 || stream_stride >= total))) {
        wmx::set_error("wmx_chain_process: the VAD of a %d-channel chain needs the tick's %d packets contiguous per stream "
                       "(packet stride %ld, stream stride %ld)", h->chn, n10, packet_stride, stream_stride);
        return WMX_EINVAL;
    }
    const int16_t *src = d_in;
    int rc = 0, rc_aec = 0;
    // the AEC's far kernel (one wave per cohort, 10-20 us with the GPU otherwise idle) needs only the far-end packet: it runs
    // on the AEC handle's side stream beside the noise suppressor instead of between it and the near kernel
    if ((h->ns || h->nsx) && h->aec && !h->no_fork && (rc = wmx::aec_fork_far(h->aec, wmx::as_stream(stream))) != 0) return rc;
    // the AECM's likewise while its cohorts are few (measured: one cohort -1.0 % of the fixed-point chain's step, two -1.1 %; with 256
    // cohorts' far waves the NSX's own workgroups queue behind them for LDS: +1.3 %, so those stay in line)
    if ((h->ns || h->nsx) && h->aecm && !h->no_fork && h->n_cohorts <= 16 && (rc = wmx::aecm_fork_far(h->aecm, wmx::as_stream(stream))) != 0)
        return rc;
    if (h->ns) {
        if ((rc = wmx_ns_process(h->ns, src, d_out, n10, stream_stride, packet_stride, stream)) != 0) {
            if (h->aec) wmx::aec_cancel_fork(h->aec);  // no AEC call follows: the next one must not start behind a stale fork point
            if (h->aecm) wmx::aecm_cancel_fork(h->aecm);
            return rc;
        }
        src = d_out;
    }
    if (h->nsx) {
        if ((rc = wmx_nsx_process(h->nsx, src, d_out, n10, stream_stride, packet_stride, stream)) != 0) {
            if (h->aecm) wmx::aecm_cancel_fork(h->aecm);
            if (h->aec) wmx::aec_cancel_fork(h->aec);
            return rc;
        }
        src = d_out;
    }
    if (h->gate_after_ns) {
        const hipEvent_t ev = h->gate_after_ns;
        h->gate_after_ns = nullptr;
        WMX_HIP(hipEventRecord(ev, wmx::as_stream(stream)));
    }
    if (h->aec) {
        const int per = h->aec_pkg / h->pkg10;  // 10 ms packets per AEC packet (1 or 2)
        if (!delay_ms) {
            // sized by the canceller's own count: a cohort added on the inner handle (wmx_chain_aec + wmx_aec_add_cohort) is covered too
            const size_t nc = (size_t)wmx_aec_cohorts(h->aec);
            if (h->zero_delays.size() < nc) h->zero_delays.resize(nc, 0);
            delay_ms = h->zero_delays.data();
        }
        rc_aec = wmx_aec_run_cohorts(h->aec, 3, d_far, far_packet_stride * per, far_group_stride, src, d_out, n10 / per, stream_stride, packet_stride * per,
                                     delay_ms, cohort_on, cohort_rc, stream);
        if (rc_aec != 0 && rc_aec != -1) return rc_aec;  // -1: a cohort's delay was rejected (its code is in cohort_rc); the others ran
        src = d_out;
    }
    if (h->aecm) {
        const int per = h->aec_pkg / h->pkg10;
        if (!delay_ms) {
            const size_t nc = (size_t)wmx_aecm_cohorts(h->aecm);
            if (h->zero_delays.size() < nc) h->zero_delays.resize(nc, 0);
            delay_ms = h->zero_delays.data();
        }
        rc_aec = wmx_aecm_run_cohorts(h->aecm, 3, d_far, far_packet_stride * per, far_group_stride, src, d_out, n10 / per, stream_stride, packet_stride * per,
                                      delay_ms, cohort_on, cohort_rc, stream);
        if (rc_aec != 0 && rc_aec != -1) return rc_aec;
        src = d_out;
    }
    if (h->agc) {
        if (h->agc_pkg == h->pkg10) {
            if ((rc = wmx_agc_process(h->agc, src, d_out, n10, stream_stride, packet_stride, stream)) != 0) return rc;
        } else if (contiguous) {  // 5 ms packets at 32 kHz, the tick in one piece
            if ((rc = wmx_agc_process(h->agc, src, d_out, (int)(total / h->agc_pkg), stream_stride, h->agc_pkg, stream)) != 0) return rc;
        } else {  // 5 ms packets inside 10 ms packets that lie apart: one call per 10 ms packet
            const int per = h->pkg10 / h->agc_pkg;
            for (int p = 0; p < n10; p++)
                if ((rc = wmx_agc_process(h->agc, src + (size_t)p * packet_stride, d_out + (size_t)p * packet_stride, per, stream_stride,
                                          h->agc_pkg, stream)) != 0)
                    return rc;
        }
        src = d_out;
    }
    if (h->vad) {
        if (src != d_out) {
            wmx::set_error("wmx_chain_process: a VAD-only chain works in place (d_in == d_out)");
            return WMX_EINVAL;
        }
        // one call over the tick's packets.  Mono: only its first packet is touched, wherever the others lie.  Interleaved
        // channels: the whole tick, in one piece per stream (checked at the top)
        const int vad_calls = (int)(total / h->vad_pkg);
        if ((rc = wmx_vad_process(h->vad, d_out, vad_calls, 1, stream_stride, h->chn > 1 ? total : h->vad_pkg, stream)) != 0) return rc;
    }
    return rc_aec;
}

// *_release + *_init of every stage for the listed streams; cohort >= 0 makes them members of that AEC cohort (which the
// caller restarts once per join time with wmx_chain_reset_cohort)
int wmx_chain_reset_streams(wmx_chain *h, const int32_t *idx, int n, int cohort, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    int rc = 0;
    if (rc == 0 && h->ns) rc = wmx_ns_reset_streams(h->ns, idx, n, stream);
    if (rc == 0 && h->nsx) rc = wmx_nsx_reset_streams(h->nsx, idx, n, stream);
    if (rc == 0 && h->aec) rc = wmx_aec_reset_streams(h->aec, idx, n, cohort, stream);
    if (rc == 0 && h->aecm) rc = wmx_aecm_reset_streams(h->aecm, idx, n, cohort, stream);
    if (rc == 0 && h->agc) rc = wmx_agc_reset_streams(h->agc, idx, n, stream);
    if (rc == 0 && h->vad) rc = wmx_vad_reset_streams(h->vad, idx, n, stream);
    return rc;
}

int wmx_chain_reset_streams_gain(wmx_chain *h, const int32_t *idx, int n, int cohort, int agc_value, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    int rc = 0;
    // the one stage that can refuse (agc_init returns NULL for a gain outside the table's range) goes first: nothing is reset then
    if (rc == 0 && h->agc) rc = wmx_agc_reset_streams_gain(h->agc, idx, n, agc_value, stream);
    if (rc == 0 && h->ns) rc = wmx_ns_reset_streams(h->ns, idx, n, stream);
    if (rc == 0 && h->nsx) rc = wmx_nsx_reset_streams(h->nsx, idx, n, stream);
    if (rc == 0 && h->aec) rc = wmx_aec_reset_streams(h->aec, idx, n, cohort, stream);
    if (rc == 0 && h->aecm) rc = wmx_aecm_reset_streams(h->aecm, idx, n, cohort, stream);
    if (rc == 0 && h->vad) rc = wmx_vad_reset_streams(h->vad, idx, n, stream);
    return rc;
}

int wmx_chain_set_agc_gain_streams(wmx_chain *h, const int32_t *idx, int n, int agc_value, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    return h->agc ? wmx_agc_set_gain_streams(h->agc, idx, n, agc_value, stream) : 0;
}

int wmx_chain_reset_cohort(wmx_chain *h, int cohort, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    return h->aec ? wmx_aec_reset_cohort(h->aec, cohort, stream) : (h->aecm ? wmx_aecm_reset_cohort(h->aecm, cohort, stream) : 0);
}

int wmx_chain_add_cohort(wmx_chain *h, int *cohort, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h || !cohort) return WMX_EINVAL;
    if (!h->aec && !h->aecm) {
        *cohort = 0;
        return 0;
    }
    const int rc = h->aec ? wmx_aec_add_cohort(h->aec, cohort, stream) : wmx_aecm_add_cohort(h->aecm, cohort, stream);
    if (rc != 0) return rc;
    h->n_cohorts = h->aec ? wmx_aec_cohorts(h->aec) : wmx_aecm_cohorts(h->aecm);
    h->zero_delays.assign((size_t)h->n_cohorts, 0);
    return 0;
}

int wmx_chain_retire_cohort(wmx_chain *h, int cohort) {
    if (!h) return WMX_EINVAL;
    return h->aec ? wmx_aec_retire_cohort(h->aec, cohort) : (h->aecm ? wmx_aecm_retire_cohort(h->aecm, cohort) : 0);
}

int wmx_chain_coalesce(wmx_chain *h, int max_pairs, int32_t *merged_from, int32_t *merged_into, int cap, int *n_merged, void *stream) {
    WMX_ON_DEVICE(h);
    if (n_merged) *n_merged = 0;
    if (!h) return WMX_EINVAL;
    if (!h->aec && !h->aecm) return 0;
    const int rc = h->aec ? wmx_aec_coalesce(h->aec, max_pairs, merged_from, merged_into, cap, n_merged, stream)
                          : wmx_aecm_coalesce(h->aecm, max_pairs, merged_from, merged_into, cap, n_merged, stream);
    const int n = h->aec ? wmx_aec_cohorts(h->aec) : wmx_aecm_cohorts(h->aecm);
    if (n != h->n_cohorts) {
        h->n_cohorts = n;
        h->zero_delays.assign((size_t)n, 0);
    }
    return rc;
}

int wmx_chain_cohorts(const wmx_chain *h) { return h ? h->n_cohorts : WMX_EINVAL; }

// A stream's state in every stage, concatenated in the heartbeat's order (each part is the stage's own blob); the AEC cohort
// travels separately through wmx_aec_export_cohort / wmx_aec_import_cohort on wmx_chain_aec(h).
int wmx_chain_stream_state_bytes(const wmx_chain *h) {
    if (!h) return WMX_EINVAL;
    return (h->ns ? wmx_ns_stream_state_bytes(h->ns) : 0) + (h->nsx ? wmx_nsx_stream_state_bytes(h->nsx) : 0) +
           (h->aec ? wmx_aec_stream_state_bytes(h->aec) : 0) + (h->aecm ? wmx_aecm_stream_state_bytes(h->aecm) : 0) +
           (h->agc ? wmx_agc_stream_state_bytes(h->agc) : 0) + (h->vad ? wmx_vad_stream_state_bytes(h->vad) : 0);
}

int wmx_chain_export_stream(wmx_chain *h, int stream_index, void *host_blob) {
    WMX_ON_DEVICE(h);
    if (!h || !host_blob) return WMX_EINVAL;
    char *p = static_cast<char *>(host_blob);
    int rc = 0;
    if (rc == 0 && h->ns) rc = wmx_ns_export_stream(h->ns, stream_index, p), p += wmx_ns_stream_state_bytes(h->ns);
    if (rc == 0 && h->nsx) rc = wmx_nsx_export_stream(h->nsx, stream_index, p), p += wmx_nsx_stream_state_bytes(h->nsx);
    if (rc == 0 && h->aec) rc = wmx_aec_export_stream(h->aec, stream_index, p), p += wmx_aec_stream_state_bytes(h->aec);
    if (rc == 0 && h->aecm) rc = wmx_aecm_export_stream(h->aecm, stream_index, p), p += wmx_aecm_stream_state_bytes(h->aecm);
    if (rc == 0 && h->agc) rc = wmx_agc_export_stream(h->agc, stream_index, p), p += wmx_agc_stream_state_bytes(h->agc);
    if (rc == 0 && h->vad) rc = wmx_vad_export_stream(h->vad, stream_index, p), p += wmx_vad_stream_state_bytes(h->vad);
    return rc;
}

int wmx_chain_import_stream(wmx_chain *h, int stream_index, const void *host_blob, int cohort) {
    WMX_ON_DEVICE(h);
    if (!h || !host_blob) return WMX_EINVAL;
    const char *p = static_cast<const char *>(host_blob);
    int rc = 0;
    if (rc == 0 && h->ns) rc = wmx_ns_import_stream(h->ns, stream_index, p), p += wmx_ns_stream_state_bytes(h->ns);
    if (rc == 0 && h->nsx) rc = wmx_nsx_import_stream(h->nsx, stream_index, p), p += wmx_nsx_stream_state_bytes(h->nsx);
    if (rc == 0 && h->aec) rc = wmx_aec_import_stream(h->aec, stream_index, p, cohort), p += wmx_aec_stream_state_bytes(h->aec);
    if (rc == 0 && h->aecm) rc = wmx_aecm_import_stream(h->aecm, stream_index, p, cohort), p += wmx_aecm_stream_state_bytes(h->aecm);
    if (rc == 0 && h->agc) rc = wmx_agc_import_stream(h->agc, stream_index, p), p += wmx_agc_stream_state_bytes(h->agc);
    if (rc == 0 && h->vad) rc = wmx_vad_import_stream(h->vad, stream_index, p), p += wmx_vad_stream_state_bytes(h->vad);
    return rc;
}

int wmx_chain_set_active(wmx_chain *h, const uint8_t *host_mask, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    int rc = 0;
    if (rc == 0 && h->ns) rc = wmx_ns_set_active(h->ns, host_mask, stream);
    if (rc == 0 && h->nsx) rc = wmx_nsx_set_active(h->nsx, host_mask, stream);
    if (rc == 0 && h->aec) rc = wmx_aec_set_active(h->aec, host_mask, stream);
    if (rc == 0 && h->aecm) rc = wmx_aecm_set_active(h->aecm, host_mask, stream);
    if (rc == 0 && h->agc) rc = wmx_agc_set_active(h->agc, host_mask, stream);
    if (rc == 0 && h->vad) rc = wmx_vad_set_active(h->vad, host_mask, stream);
    return rc;
}

}  // extern "C"
