/* host_paced.c -- the reference's PACED heartbeat for S concurrent streams, in plain C99 over the library's C ABI (wmx_rt_*).
 *
 * wmix's record thread handles one package of WMIX_INTERVAL_MS = 20 ms per tick (src/wmixConf.h:112) and paces itself so that the
 * tick's work and its sleep add up to WMIX_INTERVAL_MS * 1000 - 2000 us (src/wmix.c:536-538, 820: DELAY_US(intervalUs); the play
 * thread :1468-1474): a tick has to be over 2 ms before the next package is due.  This host does the same for S streams whose
 * packages lie in pinned host memory:
 *
 *     every tick_ms, on an ABSOLUTE schedule (clock_nanosleep(TIMER_ABSTIME), the last 200 us spun):
 *         wmx_rt_tick  -- every sub-batch: H2D -> NS -> AEC -> AGC -> VAD -> D2H; returns when the LAST row is back in host memory
 *         latency = that moment - the SCHEDULED release of the tick (a tick that starts late carries its predecessor's overrun)
 *
 * and reports p50 / p99 / p99.9 / max, the ticks that missed the reference's budget (tick_ms - 2 ms) and those that overran the
 * period itself, as one JSON line.  The rows of a tick are whatever the slot holds: the caller's pattern file fills the `slots` sets
 * of rows once (stream s gets pattern row s % n_pattern), so tick t works on pattern slot t % slots -- what a NIC or a capture
 * thread would have written there is not this example's business.  The outputs of the listed sample streams are kept for the last
 * `keep` ticks and written to --dump for the parity check (tests/test_paced_host_gpu.py replays them through the oracle).
 *
 *   host_paced --streams S [--sub 32768] [--slots 4] [--tick-ms 20] [--ticks 1500] [--prime 150] [--kind pcm|rtp] [--freq 16000]
 *              [--interval-ms 20] [--phases 1] [--calls 0 [--n-far 16]] [--spin 0] [--rt-prio 0] [--pattern file --n-pattern 256] [--dump file --keep 32 --sample a,b,c] [--lat file]
 *
 * --phases P > 1: the streams of a server do not all deliver their package at the same instant.  P groups of S / P streams (one
 * wmx_rt each), group g released at t0 + (k * P + g) * tick_ms / P with the whole tick as its period and tick_ms - 2 ms as its budget:
 * wmx_rt_submit at the release, wmx_rt_poll between releases (a completion is seen within microseconds).  The device then works in P
 * short bursts per period and never idles long enough for its power management to clock it down (profiles/r06/README_paced.md).
 *
 * pattern file: int16 far [slots][n_far][far_samples] (n_far = 1 without --calls), then rows [slots][n_pattern][row_bytes] (row_bytes from the library).
 * dump file:    rows [keep][n_sample][row_bytes] of the last `keep` ticks.     lat / lag file (--lag): double latency_ms / release_lag_ms [ticks * phases].
 *
 * Build (what __graft_entry__.build() runs):
 *   gcc -std=c99 -O2 -Iinclude examples/host_paced.c -o examples/host_paced -Lwmix_amd -lwmix_amd -Wl,-rpath,'$ORIGIN/../wmix_amd' -lm
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include "wmix_amd.h"

static int64_t now_ns(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (int64_t)t.tv_sec * 1000000000LL + t.tv_nsec;
}

static void sleep_until(int64_t due_ns, int64_t spin_ns) {
    const int64_t coarse = due_ns - spin_ns;
    if (now_ns() < coarse) {
        struct timespec t = {(time_t)(coarse / 1000000000LL), (long)(coarse % 1000000000LL)};
        while (clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &t, NULL) != 0) {
        }
    }
    while (now_ns() < due_ns) {
    }
}

static int cmp_double(const void *a, const void *b) {
    const double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : x > y;
}

static double quantile(const double *sorted, int n, double q) { /* numpy's default (linear interpolation) */
    const double pos = q * (n - 1);
    const int lo = (int)floor(pos);
    const int hi = lo + 1 < n ? lo + 1 : lo;
    return sorted[lo] + (sorted[hi] - sorted[lo]) * (pos - lo);
}

static const char *arg_of(int argc, char **argv, const char *name, const char *dflt) {
    for (int i = 1; i + 1 < argc; i++)
        if (!strcmp(argv[i], name)) return argv[i + 1];
    return dflt;
}

#define MAX_PHASES 16

int main(int argc, char **argv) {
    const long S = atol(arg_of(argc, argv, "--streams", "0"));
    const int sub = atoi(arg_of(argc, argv, "--sub", "32768")), slots = atoi(arg_of(argc, argv, "--slots", "4"));
    const double tick_ms = atof(arg_of(argc, argv, "--tick-ms", "20"));
    const int ticks = atoi(arg_of(argc, argv, "--ticks", "1500")), prime = atoi(arg_of(argc, argv, "--prime", "150"));
    const char *kind = arg_of(argc, argv, "--kind", "pcm");
    const int freq = atoi(arg_of(argc, argv, "--freq", "16000")), interval_ms = atoi(arg_of(argc, argv, "--interval-ms", "20"));
    const int P = atoi(arg_of(argc, argv, "--phases", "1"));
    /* --calls 1: every stream hears a far-end of its own (wmx_rt_create_pcm_calls: aec_process2's far-end is per handle); the pattern
     * file then carries n_far far-end signals per slot and stream s hears number (s % n_pattern) % n_far */
    const int calls = atoi(arg_of(argc, argv, "--calls", "0")), n_far = calls ? atoi(arg_of(argc, argv, "--n-far", "16")) : 1;
    const char *pattern = arg_of(argc, argv, "--pattern", NULL), *dump = arg_of(argc, argv, "--dump", NULL), *latf = arg_of(argc, argv, "--lat", NULL), *lagf = arg_of(argc, argv, "--lag", NULL);
    const int n_pattern = atoi(arg_of(argc, argv, "--n-pattern", "256"));
    int keep = atoi(arg_of(argc, argv, "--keep", "32"));
    const char *sample_s = arg_of(argc, argv, "--sample", "0");
    /* --spin 1: never sleep between ticks, spin on the clock (a core burnt for the sake of never being woken up late) */
    const int64_t spin_ns = atoi(arg_of(argc, argv, "--spin", "0")) ? (int64_t)1 << 60 : 200000;
    if (S < 1 || ticks < 1 || tick_ms <= 2.0 || slots < 1 || n_pattern < 1 || P < 1 || P > MAX_PHASES || S < P || n_far < 1) {
        fprintf(stderr, "usage: %s --streams S [--sub N] [--slots N] [--tick-ms T] [--ticks N] [--prime N] [--kind pcm|rtp] [--phases P] ...\n", argv[0]);
        return 2;
    }
    if (keep > ticks) keep = ticks;
    /* --rt-prio N: what a real-time host does about late wake-ups -- SCHED_FIFO at priority N and its pages locked.  Needs
     * CAP_SYS_NICE / an rtprio limit; where the box refuses, the run goes on in the ordinary class and the line says so */
    const int rt_prio = atoi(arg_of(argc, argv, "--rt-prio", "0"));
    int rt_granted = 0;
    if (rt_prio > 0) {
        struct sched_param sp;
        memset(&sp, 0, sizeof(sp));
        sp.sched_priority = rt_prio;
        rt_granted = sched_setscheduler(0, SCHED_FIFO, &sp) == 0;
        if (rt_granted) (void)mlockall(MCL_CURRENT | MCL_FUTURE);
        else perror("host_paced: sched_setscheduler(SCHED_FIFO)");
    }
    const unsigned stages = WMX_CHAIN_NS | WMX_CHAIN_AEC | WMX_CHAIN_AGC | WMX_CHAIN_VAD;
    const int rtp = !strcmp(kind, "rtp");
    wmx_rt *rt[MAX_PHASES] = {0};
    long lo_of[MAX_PHASES + 1];
    int rc = 0, n_sub = 0;
    for (int g = 0; g <= P; g++) lo_of[g] = S * g / P; /* group g = streams [lo_of[g], lo_of[g + 1]) */
    for (int g = 0; g < P && rc == 0; g++) {
        const long n = lo_of[g + 1] - lo_of[g];
        rc = rtp ? wmx_rt_create_rtp(&rt[g], n, sub, slots, WMX_LAW_A, 5, stages)
                 : (calls ? wmx_rt_create_pcm_calls(&rt[g], n, sub, slots, 1, freq, interval_ms, 5, stages)
                          : wmx_rt_create_pcm(&rt[g], n, sub, slots, 1, freq, interval_ms, 5, stages));
        if (rc == 0) n_sub += wmx_rt_batches(rt[g]);
    }
    if (rc != 0) {
        fprintf(stderr, "host_paced: wmx_rt_create: %s\n", wmx_last_error());
        return 3;
    }
    const size_t row = (size_t)wmx_pipe_datagram_bytes(wmx_rt_pipe(rt[0], 0));
    const size_t far_n = rtp ? 160 : row / 2;
    /* the slots' rows: the pattern, tiled (stream s of the whole server gets pattern row s % n_pattern) */
    if (pattern) {
        FILE *f = fopen(pattern, "rb");
        const size_t far_bytes = (size_t)slots * (size_t)n_far * far_n * 2, rows_bytes = (size_t)slots * (size_t)n_pattern * row;
        uint8_t *buf = malloc(far_bytes + rows_bytes);
        if (!f || !buf || fread(buf, 1, far_bytes + rows_bytes, f) != far_bytes + rows_bytes) {
            fprintf(stderr, "host_paced: cannot read %zu bytes of %s\n", far_bytes + rows_bytes, pattern);
            return 2;
        }
        fclose(f);
        for (int g = 0; g < P; g++)
            for (int k = 0; k < slots; k++) {
                if (!calls) memcpy(wmx_rt_far(rt[g], k), buf + (size_t)k * far_n * 2, far_n * 2);
                long s = lo_of[g];
                for (int b = 0; b < wmx_rt_batches(rt[g]); b++) {
                    uint8_t *dst = wmx_pipe_in(wmx_rt_pipe(rt[g], b), k);
                    uint8_t *fdst = (uint8_t *)wmx_pipe_far(wmx_rt_pipe(rt[g], b), k);
                    const int nb = wmx_rt_batch_streams(rt[g], b);
                    for (int r = 0; r < nb; r++, s++) {
                        memcpy(dst + (size_t)r * row, buf + far_bytes + ((size_t)k * (size_t)n_pattern + (size_t)(s % n_pattern)) * row, row);
                        if (calls)
                            memcpy(fdst + (size_t)r * far_n * 2, buf + ((size_t)k * (size_t)n_far + (size_t)((s % n_pattern) % n_far)) * far_n * 2, far_n * 2);
                    }
                }
            }
        free(buf);
    }
    /* the sample streams whose rows are kept */
    long sample[64];
    int n_sample = 0;
    {
        char *tmp = strdup(sample_s), *save = NULL;
        for (char *t = strtok_r(tmp, ",", &save); t && n_sample < 64; t = strtok_r(NULL, ",", &save)) {
            const long v = atol(t);
            if (v >= 0 && v < S) sample[n_sample++] = v;
        }
        free(tmp);
    }
    uint8_t *kept = dump ? calloc((size_t)keep * (size_t)n_sample, row) : NULL;
    const int total = ticks * P; /* group-ticks */
    double *lat = malloc(sizeof(double) * (size_t)total), *lag = malloc(sizeof(double) * (size_t)total);
    /* past the start-up phases of every stage, back to back (tick t of a group's life works on slot t % slots) */
    for (int k = 0; k < prime && rc == 0; k++)
        for (int g = 0; g < P && rc == 0; g++) rc = wmx_rt_tick(rt[g], NULL, NULL, NULL);
    const int64_t period = (int64_t)(tick_ms * 1e6), step = period / P;
    const int64_t t0 = now_ns() + period;
    int flying_j[MAX_PHASES], flying_slot[MAX_PHASES];
    int64_t flying_due[MAX_PHASES];
    for (int g = 0; g < P; g++) flying_j[g] = -1;
    /* a group whose rows are back: its latency, and (behind the clock) the sampled rows of the last ticks */
#define REAP(g)                                                                                                                          \
    do {                                                                                                                                 \
        const int j_ = flying_j[g], k_ = j_ / P;                                                                                         \
        lat[j_] = (double)(now_ns() - flying_due[g]) * 1e-6;                                                                             \
        flying_j[g] = -1;                                                                                                                \
        if (kept && k_ >= ticks - keep)                                                                                                  \
            for (int i_ = 0; i_ < n_sample; i_++) {                                                                                      \
                long s_ = sample[i_] - lo_of[g];                                                                                         \
                if (sample[i_] < lo_of[g] || sample[i_] >= lo_of[g + 1]) continue;                                                       \
                int b_ = 0;                                                                                                              \
                while (s_ >= wmx_rt_batch_streams(rt[g], b_)) s_ -= wmx_rt_batch_streams(rt[g], b_++);                                   \
                memcpy(kept + ((size_t)(k_ - (ticks - keep)) * (size_t)n_sample + (size_t)i_) * row,                                     \
                       wmx_pipe_out(wmx_rt_pipe(rt[g], b_), flying_slot[g]) + (size_t)s_ * row, row);                                    \
            }                                                                                                                            \
    } while (0)
    int64_t next_note = t0 + 30000000000LL;
    for (int j = 0; j < total && rc == 0; j++) {
        const int64_t due = t0 + (int64_t)j * step;
        const int g = j % P;
        if (due >= next_note) { /* a long run says it is alive twice a minute (behind a completed tick, in front of the sleep) */
            fprintf(stderr, "host_paced: tick %d of %d\n", j / P, ticks);
            next_note += 30000000000LL;
        }
        if (P == 1) {
            sleep_until(due, spin_ns);
        } else {
            for (;;) { /* until the release: see the groups in flight come back */
                int any = 0;
                for (int q = 0; q < P; q++)
                    if (flying_j[q] >= 0) {
                        const int d = wmx_rt_poll(rt[q]);
                        if (d < 0) rc = d;
                        if (d == 1) REAP(q);
                        else any = 1;
                    }
                if (now_ns() >= due || rc != 0) break;
                if (!any) {
                    sleep_until(due, spin_ns);
                    break;
                }
            }
            if (flying_j[g] >= 0 && rc == 0) { /* its previous tick is not back yet: the release waits for it (and the wait counts) */
                rc = wmx_rt_wait(rt[g]);
                REAP(g);
            }
        }
        if (rc != 0) break;
        const int64_t start = now_ns();
        lag[j] = (double)(start - due) * 1e-6;
        flying_due[g] = due;
        flying_j[g] = j;
        if (P == 1) {
            rc = wmx_rt_tick(rt[g], NULL, &flying_slot[g], NULL);
            REAP(g);
        } else {
            rc = wmx_rt_submit(rt[g], NULL, &flying_slot[g], NULL);
        }
    }
    for (int g = 0; g < P; g++)
        if (flying_j[g] >= 0) {
            const int rw = wmx_rt_wait(rt[g]);
            if (rc == 0) rc = rw;
            REAP(g);
        }
    if (rc != 0) fprintf(stderr, "host_paced: failed (rc %d): %s\n", rc, wmx_last_error());
    long failed = 0;
    for (int g = 0; g < P; g++) {
        for (int b = 0; b < wmx_rt_batches(rt[g]); b++) failed += wmx_pipe_failed_steps(wmx_rt_pipe(rt[g], b));
        wmx_rt_destroy(rt[g]);
    }
    if (rc == 0 && dump) {
        FILE *f = fopen(dump, "wb");
        if (!f || fwrite(kept, row, (size_t)keep * (size_t)n_sample, f) != (size_t)keep * (size_t)n_sample) rc = 4;
        if (f) fclose(f);
    }
    if (rc == 0 && latf) {
        FILE *f = fopen(latf, "wb");
        if (!f || fwrite(lat, sizeof(double), (size_t)total, f) != (size_t)total) rc = 4;
        if (f) fclose(f);
    }
    if (rc == 0 && lagf) { /* release lag per group-tick: how late the host thread was (its own oversleep, or a predecessor that overran) */
        FILE *f = fopen(lagf, "wb");
        if (!f || fwrite(lag, sizeof(double), (size_t)total, f) != (size_t)total) rc = 4;
        if (f) fclose(f);
    }
    int worst = 0, misses = 0, overruns = 0;
    double lag_max = 0;
    const double budget = tick_ms - 2.0;
    for (int k = 0; k < total; k++) {
        if (lat[k] > lat[worst]) worst = k;
        if (lag[k] > lag_max) lag_max = lag[k];
        misses += lat[k] > budget;
        overruns += lat[k] > tick_ms;
    }
    const double worst_ms = lat[worst];
    qsort(lat, (size_t)total, sizeof(double), cmp_double);
    qsort(lag, (size_t)total, sizeof(double), cmp_double);
    printf("{\"host\": \"examples/host_paced.c\", \"kind\": \"%s\", \"far_end_per_stream\": %s, \"streams\": %ld, \"phases\": %d, \"sub_batch\": %d, \"sub_batches\": %d, \"slots\": %d, "
           "\"row_bytes\": %zu, \"tick_ms\": %.3f, \"budget_ms\": %.3f, \"ticks\": %d, \"group_ticks\": %d, \"primed_ticks\": %d, \"p50_ms\": %.4f, \"p99_ms\": %.4f, "
           "\"p99_9_ms\": %.4f, \"max_ms\": %.4f, \"misses\": %d, \"overruns_of_the_period\": %d, \"release_lag_p50_ms\": %.4f, \"release_lag_max_ms\": %.4f, "
           "\"worst_tick\": %d, \"failed_steps\": %ld, \"kept_ticks\": %d, \"sched_fifo\": %s, \"rc\": %d}\n",
           kind, calls ? "true" : "false", S, P, sub, n_sub, slots, row, tick_ms, budget, ticks, total, prime, quantile(lat, total, 0.5), quantile(lat, total, 0.99),
           quantile(lat, total, 0.999), worst_ms, misses, overruns, quantile(lag, total, 0.5), lag_max, worst / P, failed, dump ? keep : 0,
           rt_prio > 0 ? (rt_granted ? "true" : "\"refused\"") : "false", rc);
    return rc ? 1 : 0;
}
