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
    bool pcm;
    std::vector<wmx_pipe *> pipe;
    hipStream_t s_in, s_out;
    int next;  // the slot of the next tick (every sub-batch rotates in lockstep, also past a failed one)
};

extern "C" {

int wmx_rt_destroy(wmx_rt *h) {
    WMX_ON_DEVICE(h);
    if (!h) return 0;
    for (wmx_pipe *p : h->pipe)
        if (p) wmx_pipe_destroy(p);
    if (h->s_in) (void)hipStreamDestroy(h->s_in);
    if (h->s_out) (void)hipStreamDestroy(h->s_out);
    delete h;
    return 0;
}

static int rt_make(wmx_rt **out, long n_streams, int sub_batch, int slots, bool pcm, int law, int chn, int freq, int interval_ms, int agc_value,
                   unsigned stages) {
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
    h->next = 0;
    hipError_t e = hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking);
    int rc = e == hipSuccess ? 0 : wmx::hip_fail(e, "wmx_rt_create: copy streams", __FILE__, __LINE__);
    for (long lo = 0; rc == 0 && lo < n_streams; lo += sub_batch) {
        const int n = (int)(n_streams - lo < sub_batch ? n_streams - lo : sub_batch);
        wmx_pipe *p = nullptr;
        rc = wmx::pipe_make(&p, n, slots, pcm, law, chn, freq, interval_ms, agc_value, stages, h->s_in, h->s_out);
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

int wmx_rt_submit(wmx_rt *h, const int16_t *d_far, int *slot, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h) return WMX_EINVAL;
    const int k = h->next;
    h->next = (k + 1) % h->slots;  // a tick takes its slot whatever becomes of its sub-batches
    int first = 0;
    wmx_pipe *prev = nullptr;
    for (size_t b = 0; b < h->pipe.size(); b++) {
        wmx_pipe *p = h->pipe[b];
        p->next = k;
        // sub-batch 0 uploads the tick's far-end (unless it is on the device already), the others use its copy
        const int rc = wmx::pipe_submit(p, d_far, nullptr, stream, prev ? prev : p, (d_far || b == 0) ? nullptr : h->pipe[0]);
        if (rc != 0) {
            if (first == 0) first = rc;
            if (b == 0 && !d_far) break;  // nobody has the far-end: the tick is lost as a whole
            continue;  // this sub-batch has lost its step; `prev` keeps its pending download for the next one that runs
        }
        prev = p;
    }
    if (slot) *slot = k;
    return first;
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

int wmx_rt_tick(wmx_rt *h, const int16_t *d_far, int *slot, void *stream) {
    const int rc = wmx_rt_submit(h, d_far, slot, stream);
    const int rw = wmx_rt_wait(h);
    return rc ? rc : rw;
}

int wmx_rt_step_resident(wmx_rt *h, uint8_t *d_rows, long stride, const int16_t *d_far, uint8_t *d_out, long out_stride, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h || !d_rows || !d_far || (!h->pcm && !d_out)) {
        wmx::set_error("wmx_rt_step_resident: bad argument");
        return WMX_EINVAL;
    }
    long lo = 0;
    for (wmx_pipe *p : h->pipe) {
        uint8_t *in = d_rows + (size_t)lo * (size_t)stride;
        uint8_t *o = h->pcm ? in : d_out + (size_t)lo * (size_t)out_stride;
        const int rc = wmx_pipe_step_resident(p, in, stride, d_far, o, h->pcm ? stride : out_stride, stream);
        if (rc != 0) return rc;
        lo += p->n_streams;
    }
    return 0;
}

}  // extern "C"
