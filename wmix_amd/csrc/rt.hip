// rt.hip -- the paced heartbeat over S concurrent streams in host memory (host code only: it sequences the sub-batch pipelines).
//
// The reference's record thread handles ONE stream per tick of WMIX_INTERVAL_MS and must be done 2 ms before the next package is due
// (src/wmix.c:536-538, 820; the play thread :1468-1474; src/wmixConf.h:112).  A wmx_rt is that tick for S streams: B = ceil(S / sub)
// sub-batches, each a wmx_pipe (pipe.hip) with its own chain state, pinned host rows and device twins, all on ONE upload stream and
// ONE download stream (two FIFOs: sub-batch 0's rows go up first and the DMA engines are never asked to interleave two uploads).
// wmx_rt_submit queues the B sub-batches back to back on the caller's compute stream; the download of sub-batch b is queued by the
// submit of b + 1 behind ITS noise suppressor (the blit kernel of a download starves a memory-bound kernel beside it, pipe.hip), the
// last one by wmx_rt_wait.  Tick latency = upload(sub-batch 0) + compute(all) + download(sub-batch B - 1).
#include "pipe_internal.h"

struct wmx_rt {
    int device;  // first member of every handle (wmx_handle_device)
    long n_streams;
    int sub, slots;
    bool pcm, far_rows;
    std::vector<wmx_pipe *> pipe;
    hipStream_t s_in, s_out;
    int next;  // the slot of the next tick (every sub-batch rotates in lockstep, also past a failed one)
    // compute streams of the library's own (wmx_rt_set_compute_streams): sub-batch b runs on cs[b % n_cs], forked from the caller's
    // stream at the start of the tick and joined to it at the end
    static constexpr int kMaxCs = 4;
    int n_cs;
    hipStream_t cs[kMaxCs];
    hipEvent_t ev_fork, ev_join[kMaxCs];
};

extern "C" {

int wmx_rt_destroy(wmx_rt *h) {
    WMX_ON_DEVICE(h);
    if (!h) return 0;
    for (wmx_pipe *p : h->pipe)
        if (p) wmx_pipe_destroy(p);
    if (h->s_in) (void)hipStreamDestroy(h->s_in);
    if (h->s_out) (void)hipStreamDestroy(h->s_out);
    for (int i = 0; i < wmx_rt::kMaxCs; i++) {
        if (h->cs[i]) (void)hipStreamDestroy(h->cs[i]);
        if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    delete h;
    return 0;
}

// Sub-batch b of a tick runs on compute stream b % n: with n = 1 (the default) that is the caller's stream and the sub-batches run
// strictly one behind the other -- every kernel boundary drains the device (the last waves of a launch run alone) before the next
// launch ramps up; with n >= 2 the library's own streams take turns, forked from the caller's stream when the tick starts and joined
// to it when it ends, and the tail of one sub-batch's kernel overlaps the head of the next one's.  Measured (profiles/r06/sweeps/
// paced_exploration.jsonl, 458 752 streams resident): 16.12 / 16.16 / 16.15 ms per tick for n = 1 / 2 / 3 -- the chain's kernels are bound
// by vector issue, a draining launch leaves nothing idle for the next one to use.  The default stays 1.
int wmx_rt_set_compute_streams(wmx_rt *h, int n) {
    WMX_ON_DEVICE(h);
    if (!h || n < 1 || n > wmx_rt::kMaxCs) {
        wmx::set_error("wmx_rt_set_compute_streams: n=%d (1 .. %d)", n, wmx_rt::kMaxCs);
        return WMX_EINVAL;
    }
    if (n > 1) {
        if (!h->ev_fork) WMX_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        for (int i = 0; i < n; i++) {
            if (!h->cs[i]) WMX_HIP(hipStreamCreateWithFlags(&h->cs[i], hipStreamNonBlocking));
            if (!h->ev_join[i]) WMX_HIP(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
        }
    }
    h->n_cs = n;
    return 0;
}

static int rt_make(wmx_rt **out, long n_streams, int sub_batch, int slots, bool pcm, int law, int chn, int freq, int interval_ms, int agc_value,
                   unsigned stages, bool far_rows = false) {
    if (!out) return WMX_EINVAL;
    *out = nullptr;
    if (n_streams < 1 || sub_batch < 1 || slots < 1 || slots > 16 || (n_streams + sub_batch - 1) / sub_batch > 4096) {
        wmx::set_error("wmx_rt_create: n_streams=%ld sub_batch=%d slots=%d", n_streams, sub_batch, slots);
        return WMX_EINVAL;
    }
    wmx_rt *h = new wmx_rt();
    if ((h->device = wmx::current_device()) < 0) {
        delete h;
        return WMX_ENODEV;
    }
    h->n_streams = n_streams;
    h->sub = sub_batch;
    h->slots = slots;
    h->pcm = pcm;
    h->far_rows = far_rows;
    h->next = 0;
    h->n_cs = 1;
    hipError_t e = hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking);
    int rc = e == hipSuccess ? 0 : wmx::hip_fail(e, "wmx_rt_create: copy streams", __FILE__, __LINE__);
    for (long lo = 0; rc == 0 && lo < n_streams; lo += sub_batch) {
        const int n = (int)(n_streams - lo < sub_batch ? n_streams - lo : sub_batch);
        wmx_pipe *p = nullptr;
        rc = wmx::pipe_make(&p, n, slots, pcm, law, chn, freq, interval_ms, agc_value, stages, h->s_in, h->s_out, far_rows);
        if (rc == 0) h->pipe.push_back(p);
    }
    if (rc != 0) {
        char why[512];
        snprintf(why, sizeof(why), "%s", wmx_last_error());
        wmx_rt_destroy(h);
        wmx::set_error("%s", why);
        return rc;
    }
    *out = h;
    return 0;
}

int wmx_rt_create_pcm(wmx_rt **out, long n_streams, int sub_batch, int slots, int chn, int freq, int interval_ms, int agc_value, unsigned stages) {
    if ((chn != 1 && chn != 2) || freq < 8000 || freq % 100 || interval_ms < 10 || interval_ms % 10 || interval_ms > 100) {
        if (out) *out = nullptr;
        wmx::set_error("wmx_rt_create_pcm: chn=%d freq=%d interval_ms=%d", chn, freq, interval_ms);
        return WMX_EINVAL;
    }
    return rt_make(out, n_streams, sub_batch, slots, true, 0, chn, freq, interval_ms, agc_value, stages);
}

// calls: every stream its own far-end (wmx_pipe_create_pcm_calls); the far rows of sub-batch b: wmx_pipe_far(wmx_rt_pipe(h, b), slot)
int wmx_rt_create_pcm_calls(wmx_rt **out, long n_streams, int sub_batch, int slots, int chn, int freq, int interval_ms, int agc_value, unsigned stages) {
    if ((chn != 1 && chn != 2) || freq < 8000 || freq % 100 || interval_ms < 10 || interval_ms % 10 || interval_ms > 100 || !(stages & WMX_CHAIN_AEC) ||
        (stages & WMX_CHAIN_AECM)) {
        if (out) *out = nullptr;
        wmx::set_error("wmx_rt_create_pcm_calls: chn=%d freq=%d interval_ms=%d stages=0x%x (the float canceller must be on)", chn, freq, interval_ms, stages);
        return WMX_EINVAL;
    }
    return rt_make(out, n_streams, sub_batch, slots, true, 0, chn, freq, interval_ms, agc_value, stages, true);
}

int wmx_rt_create_rtp(wmx_rt **out, long n_streams, int sub_batch, int slots, int law, int agc_value, unsigned stages) {
    if (law != WMX_LAW_A) {  // as wmx_pipe_create: the reference has an A-law receiver only (src/wmixTask.c:1282)
        if (out) *out = nullptr;
        wmx::set_error("wmx_rt_create_rtp: law=%d", law);
        return WMX_EINVAL;
    }
    return rt_make(out, n_streams, sub_batch, slots, false, law, 1, wmx::kPipeFreq, 20, agc_value, stages);
}

int wmx_rt_batches(const wmx_rt *h) { return h ? (int)h->pipe.size() : WMX_EINVAL; }
int wmx_rt_batch_streams(const wmx_rt *h, int b) { return (h && b >= 0 && b < (int)h->pipe.size()) ? h->pipe[(size_t)b]->n_streams : WMX_EINVAL; }
wmx_pipe *wmx_rt_pipe(wmx_rt *h, int b) { return (h && b >= 0 && b < (int)h->pipe.size()) ? h->pipe[(size_t)b] : nullptr; }
int16_t *wmx_rt_far(wmx_rt *h, int slot) { return h ? wmx_pipe_far(h->pipe[0], slot) : nullptr; }

// the tick's launches: sub-batch b on `main`, or on the library's stream b % n_cs between a fork from and a join to `main`
static int rt_launches(wmx_rt *h, hipStream_t main, int (*one)(wmx_rt *, size_t, void *, void *), void *ctx) {
    const bool own = h->n_cs > 1 && h->pipe.size() > 1;
    if (own) {
        WMX_HIP(hipEventRecord(h->ev_fork, main));
        for (int i = 0; i < h->n_cs; i++) WMX_HIP(hipStreamWaitEvent(h->cs[i], h->ev_fork, 0));
    }
    int first = 0;
    for (size_t b = 0; b < h->pipe.size(); b++) {
        const int rc = one(h, b, own ? (void *)h->cs[b % (size_t)h->n_cs] : (void *)main, ctx);
        if (rc != 0 && first == 0) first = rc;
    }
    if (own)
        for (int i = 0; i < h->n_cs; i++) {
            WMX_HIP(hipEventRecord(h->ev_join[i], h->cs[i]));
            WMX_HIP(hipStreamWaitEvent(main, h->ev_join[i], 0));
        }
    return first;
}

struct SubmitCtx {
    const int16_t *d_far;
    wmx_pipe *prev;
    int k;
    bool lost;
    long lo;  // first stream of the sub-batch at hand
};
static int submit_one(wmx_rt *h, size_t b, void *stream, void *vctx) {
    SubmitCtx *c = static_cast<SubmitCtx *>(vctx);
    if (c->lost) return 0;
    wmx_pipe *p = h->pipe[b];
    p->next = c->k;
    // sub-batch 0 uploads the tick's far-end (unless it is on the device already), the others use its copy; with a far-end per stream
    // every sub-batch has (and uploads) its own far rows, or finds them at its streams' offset of d_far
    const int16_t *far_b = c->d_far;
    if (h->far_rows && far_b) far_b += (size_t)c->lo * (size_t)p->far_samples;
    c->lo += p->n_streams;
    const int rc = wmx::pipe_submit(p, far_b, nullptr, stream, c->prev ? c->prev : p, (c->d_far || b == 0 || h->far_rows) ? nullptr : h->pipe[0]);
    if (rc != 0) {
        if (b == 0 && !c->d_far && !h->far_rows) c->lost = true;  // nobody has the far-end: the tick is lost as a whole
        return rc;  // this sub-batch has lost its step; `prev` keeps its pending download for the next one that runs
    }
    c->prev = p;
    return 0;
}

int wmx_rt_submit(wmx_rt *h, const int16_t *d_far, int *slot, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    const int k = h->next;
    h->next = (k + 1) % h->slots;  // a tick takes its slot whatever becomes of its sub-batches
    SubmitCtx c{d_far, nullptr, k, false, 0};
    const int rc = rt_launches(h, wmx::as_stream(stream), submit_one, &c);
    if (slot) *slot = k;
    return rc;
}

int wmx_rt_wait(wmx_rt *h) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    int first = 0;
    // queue every download still owed before blocking on the first
    for (wmx_pipe *p : h->pipe)
        if (p->pending >= 0) {
            const int s = p->pending;
            const int rc = wmx_pipe_wait(p, s);  // one sub-batch at most has a pending download after a complete submit: the last one
            if (rc != 0 && first == 0) first = rc;
        }
    for (wmx_pipe *p : h->pipe) {
        const int rc = wmx_pipe_wait(p, -1);
        if (rc != 0 && first == 0) first = rc;
    }
    return first;
}

// Non-blocking wmx_rt_wait: queues every download still owed and returns 1 when every row of every queued tick is in host memory, 0
// when something is still on its way (a host that releases groups of streams at staggered phases polls the older groups while it
// waits for the next release, examples/host_paced.c --phases).
int wmx_rt_poll(wmx_rt *h) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    int done = 1;
    for (wmx_pipe *p : h->pipe) {
        const int rc = wmx_pipe_poll(p, -1);
        if (rc < 0) return rc;
        done &= rc;
    }
    return done;
}

int wmx_rt_tick(wmx_rt *h, const int16_t *d_far, int *slot, void *stream) {
    const int rc = wmx_rt_submit(h, d_far, slot, stream);
    const int rw = wmx_rt_wait(h);
    return rc ? rc : rw;
}

struct ResidentCtx {
    uint8_t *d_rows, *d_out;
    long stride, out_stride;
    const int16_t *d_far;
    long lo;
};
static int resident_one(wmx_rt *h, size_t b, void *stream, void *vctx) {
    ResidentCtx *c = static_cast<ResidentCtx *>(vctx);
    wmx_pipe *p = h->pipe[b];
    uint8_t *in = c->d_rows + (size_t)c->lo * (size_t)c->stride;
    uint8_t *o = h->pcm ? in : c->d_out + (size_t)c->lo * (size_t)c->out_stride;
    const int16_t *far_b = h->far_rows ? c->d_far + (size_t)c->lo * (size_t)p->far_samples : c->d_far;
    c->lo += p->n_streams;
    return wmx_pipe_step_resident(p, in, c->stride, far_b, o, h->pcm ? c->stride : c->out_stride, stream);
}

int wmx_rt_step_resident(wmx_rt *h, uint8_t *d_rows, long stride, const int16_t *d_far, uint8_t *d_out, long out_stride, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h || !d_rows || !d_far || (!h->pcm && !d_out)) {
        wmx::set_error("wmx_rt_step_resident: bad argument");
        return WMX_EINVAL;
    }
    ResidentCtx c{d_rows, d_out, stride, out_stride, d_far, 0};
    return rt_launches(h, wmx::as_stream(stream), resident_one, &c);
}

}  // extern "C"
