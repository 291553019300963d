/* host_rtp_pipe.c -- a C host of the packet edge: plain C99, the library's C ABI, nothing else (not even the HIP
 * runtime API: the library owns the pinned buffers, the copy streams and the events -- wmx_pipe_*).
 *
 * What wmix_thread_rtp_recv_pcma + the record heartbeat + wmix_thread_rtp_send_pcma do for one stream per 20 ms
 * (src/wmixTask.c:1278-1316, src/wmix.c:613-709, src/wmixTask.c:1124-1143), for n_streams per step:
 *
 *     for every 20 ms:  fill the next slot's host rows with the datagrams that arrived (and the far-end's 160 samples)
 *                       wmx_pipe_submit          -- H2D, ingest + chain + egress, D2H: queued, returns at once
 *                       wmx_pipe_wait(oldest)    -- the datagrams of `slots` steps ago are ready to be sent
 *
 *   host_rtp_pipe far.i16 in.rtp out.rtp n_streams n_steps [slots]
 *
 * far.i16  int16 [n_steps][160]            the shared far-end (20 ms at 8 kHz per step)
 * in.rtp   uint8 [n_steps][n_streams][172] RTP/PCMA datagrams, step-major (what arrives per 20 ms)
 * out.rtp  same shape                      what the senders put on the wire
 * Prints one JSON line with the wall time per step.
 *
 *   host_rtp_pipe far.i16 in.pcm out.pcm n_streams n_steps slots --pcm chn freq interval_ms
 *
 * the same loop over wmx_pipe_create_pcm: the heartbeat's own boundary, a package of chn x freq x interval_ms per stream in host
 * memory, worked on in place (src/wmix.c:609-709).  in.pcm / out.pcm: int16 [n_steps][n_streams][package], far.i16: int16
 * [n_steps][package] (a far-end package of the same format per step).  Row sizes come from the library (wmx_pipe_datagram_bytes).
 *
 * Build (what __graft_entry__.build() runs):
 *   gcc -std=c99 -O2 -Iinclude examples/host_rtp_pipe.c -o examples/host_rtp_pipe -Lwmix_amd -lwmix_amd -Wl,-rpath,'$ORIGIN/../wmix_amd'
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "wmix_amd.h"

static void *read_file(const char *path, size_t bytes) {
    FILE *f = fopen(path, "rb");
    void *p = malloc(bytes);
    if (!f || !p || fread(p, 1, bytes, f) != bytes) {
        fprintf(stderr, "host_rtp_pipe: cannot read %zu bytes of %s\n", bytes, path);
        exit(2);
    }
    fclose(f);
    return p;
}

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

int main(int argc, char **argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s far.i16 in.rtp out.rtp n_streams n_steps [slots]\n", argv[0]);
        return 2;
    }
    const int n = atoi(argv[4]), steps = atoi(argv[5]), slots = argc > 6 ? atoi(argv[6]) : 3;
    if (n < 1 || steps < 1) return 2;
    const int pcm = argc == 11 && !strcmp(argv[7], "--pcm");
    const unsigned stages = WMX_CHAIN_NS | WMX_CHAIN_AEC | WMX_CHAIN_AGC | WMX_CHAIN_VAD;
    wmx_pipe *p = NULL;
    const int rcc = pcm ? wmx_pipe_create_pcm(&p, n, slots, atoi(argv[8]), atoi(argv[9]), atoi(argv[10]), 5, stages)
                        : wmx_pipe_create(&p, n, slots, WMX_LAW_A, 5, stages);
    if (rcc != 0) {
        fprintf(stderr, "host_rtp_pipe: wmx_pipe_create%s: %s\n", pcm ? "_pcm" : "", wmx_last_error());
        return 3;
    }
    const size_t row = (size_t)wmx_pipe_datagram_bytes(p), step_bytes = (size_t)n * row;
    const size_t far_n = pcm ? row / 2 : 160; /* int16 elements of the far-end of one step */
    int16_t *far = read_file(argv[1], (size_t)steps * far_n * 2);
    uint8_t *in = read_file(argv[2], (size_t)steps * step_bytes);
    uint8_t *out = calloc((size_t)steps, step_bytes);
    int rc = 0;
    const double t0 = now_ms();
    for (int k = 0; k < steps + slots && rc == 0; k++) {
        if (k >= slots) { /* the step that used this slot last: its datagrams go out before the slot is refilled */
            const int slot = (k - slots) % slots;
            rc = wmx_pipe_wait(p, slot);
            memcpy(out + (size_t)(k - slots) * step_bytes, wmx_pipe_out(p, slot), step_bytes);
        }
        if (k < steps && rc == 0) {
            const int slot = k % slots;
            int got = -1;
            memcpy(wmx_pipe_in(p, slot), in + (size_t)k * step_bytes, step_bytes);
            memcpy(wmx_pipe_far(p, slot), far + (size_t)k * far_n, far_n * 2);
            rc = wmx_pipe_submit(p, NULL, &got, NULL);
            if (rc == 0 && got != slot) rc = -1;
        }
    }
    const double wall = now_ms() - t0;
    if (rc != 0) fprintf(stderr, "host_rtp_pipe: failed (rc %d): %s\n", rc, wmx_last_error());
    wmx_pipe_destroy(p);
    if (rc == 0) {
        FILE *f = fopen(argv[3], "wb");
        if (!f || fwrite(out, 1, (size_t)steps * step_bytes, f) != (size_t)steps * step_bytes) rc = 4;
        if (f) fclose(f);
    }
    printf("{\"streams\": %d, \"steps\": %d, \"slots\": %d, \"row_bytes\": %zu, \"wall_ms\": %.3f, \"ms_per_step\": %.4f, \"rc\": %d}\n", n, steps,
           slots, row, wall, wall / steps, rc);
    return rc ? 1 : 0;
}
