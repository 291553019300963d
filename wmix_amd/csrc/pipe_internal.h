// pipe_internal.h -- the packet pipeline's handle, shared by pipe.hip (one batch) and rt.hip (the real-time tick over several batches).
#pragma once
#include <vector>
#include "wmx_internal.h"

namespace wmx {
constexpr int kDatagram = 172;  // 12-byte RTP header + 160 G.711 codes (20 ms at 8 kHz), src/rtp.h:33, src/rtp.c:86-95
constexpr int kPipeFreq = 8000, kPkt10 = 80;
}  // namespace wmx

struct wmx_pipe {
    int device;  // first member of every handle (wmx_handle_device)
    int n_streams, slots;
    bool pcm;          // rows are PCM packages (wmx_pipe_create_pcm), not RTP datagrams
    bool far_rows;     // every stream hears a far-end of its own (wmx_pipe_create_pcm_calls): far rows like near rows, one cohort per stream
    int row_bytes;     // bytes of one stream's row in a slot: 172, or WMIX_PKG_SIZE
    int far_samples;   // int16 elements of the far-end of one step
    int pkt10, ppc;    // int16 elements of one 10 ms packet, packets per step
    wmx_chain *chain;
    wmx_rtp *snd;
    int16_t *d_pcm;        // [n][160] the 20 ms of every stream between ingest and egress
    uint32_t *d_nbytes;    // [n] what rtp_recv + G711a2PCM delivered (320 or 0)
    uint16_t *d_seq;       // [n] header sequence numbers as the reference leaves them
    struct Slot {
        uint8_t *h_in, *h_out;   // pinned [n][172]
        int16_t *h_far;          // pinned [160]: the shared far-end of these 20 ms, for hosts that have it in host memory
        uint8_t *d_in, *d_out;   // [n][172]
        int16_t *d_far;          // [160]
        hipEvent_t ev_in, ev_done, ev_gate, ev_out;
        bool in_flight;
    };
    std::vector<Slot> slot;
    hipStream_t s_in, s_out;
    bool own_copy_streams;  // false: s_in / s_out belong to the wmx_rt this pipe is a sub-batch of (one FIFO of uploads, one of downloads)
    int next;
    int pending;  // the slot whose D2H is not queued yet (see wmx_pipe_submit), or -1
    long failed_steps;  // submits that failed after their first launch: the step is lost (see wmx_pipe_submit)
};

namespace wmx {
// rt.hip -> pipe.hip.  pipe_make with `s_in` / `s_out` given makes a sub-batch of a wmx_rt on the rt's copy streams.
int pipe_make(wmx_pipe **out, int n_streams, int slots, bool pcm, int law, int chn, int freq, int interval_ms, int agc_value, unsigned stages,
              hipStream_t s_in, hipStream_t s_out, bool far_rows = false);
// wmx_pipe_submit with two extras for a tick made of several sub-batches: `flush_for` (may be NULL) = the pipe whose pending download
// this submit queues behind ITS noise suppressor (the previous sub-batch of the same tick; h's own pending one otherwise);
// `d_far_shared` as d_far, but when `upload_far` is false and d_far is NULL nothing is uploaded and far_of's slot copy is used
// (the tick's far-end went up once, with the first sub-batch).
int pipe_submit(wmx_pipe *h, const int16_t *d_far, int *slot, void *stream, wmx_pipe *flush_for, const wmx_pipe *far_of);
}  // namespace wmx

