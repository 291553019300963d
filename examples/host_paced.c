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
 *              [--interval-ms 20] [--pattern file --n-pattern 256] [--dump file --keep 32 --sample a,b,c] [--lat file]
 *
 * pattern file: int16 far [slots][far_samples], then rows [slots][n_pattern][row_bytes] (row_bytes from the library).
 * dump file:    rows [keep][n_sample][row_bytes] of the last `keep` ticks.     lat file: double latency_ms[ticks].
 *
 * Build (what __graft_entry__.build() runs):
 *   gcc -std=c99 -O2 -Iinclude examples/host_paced.c -o examples/host_paced -Lwmix_amd -lwmix_amd -Wl,-rpath,'$ORIGIN/../wmix_amd' -lm
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

int main(int argc, char **argv) {
    const long S = atol(arg_of(argc, argv, "--streams", "0"));
    const int sub = atoi(arg_of(argc, argv, "--sub", "32768")), slots = atoi(arg_of(argc, argv, "--slots", "4"));
    const double tick_ms = atof(arg_of(argc, argv, "--tick-ms", "20"));
    const int ticks = atoi(arg_of(argc, argv, "--ticks", "1500")), prime = atoi(arg_of(argc, argv, "--prime", "150"));
    const char *kind = arg_of(argc, argv, "--kind", "pcm");
    const int freq = atoi(arg_of(argc, argv, "--freq", "16000")), interval_ms = atoi(arg_of(argc, argv, "--interval-ms", "20"));
    const char *pattern = arg_of(argc, argv, "--pattern", NULL), *dump = arg_of(argc, argv, "--dump", NULL), *latf = arg_of(argc, argv, "--lat", NULL);
    const int n_pattern = atoi(arg_of(argc, argv, "--n-pattern", "256"));
    int keep = atoi(arg_of(argc, argv, "--keep", "32"));
    const char *sample_s = arg_of(argc, argv, "--sample", "0");
    if (S < 1 || ticks < 1 || tick_ms <= 2.0 || slots < 1 || n_pattern < 1) {
        fprintf(stderr, "usage: %s --streams S [--sub N] [--slots N] [--tick-ms T] [--ticks N] [--prime N] [--kind pcm|rtp] ...\n", argv[0]);
        return 2;
    }
    if (keep > ticks) keep = ticks;
    const unsigned stages = WMX_CHAIN_NS | WMX_CHAIN_AEC | WMX_CHAIN_AGC | WMX_CHAIN_VAD;
    const int rtp = !strcmp(kind, "rtp");
    wmx_rt *rt = NULL;
    int rc = rtp ? wmx_rt_create_rtp(&rt, S, sub, slots, WMX_LAW_A, 5, stages) : wmx_rt_create_pcm(&rt, S, sub, slots, 1, freq, interval_ms, 5, stages);
    if (rc != 0) {
        fprintf(stderr, "host_paced: wmx_rt_create: %s\n", wmx_last_error());
        return 3;
    }
    const int B = wmx_rt_batches(rt);
    const size_t row = (size_t)wmx_pipe_datagram_bytes(wmx_rt_pipe(rt, 0));
    const size_t far_n = rtp ? 160 : row / 2;
    /* the slots' rows: the pattern, tiled */
    if (pattern) {
        FILE *f = fopen(pattern, "rb");
        const size_t far_bytes = (size_t)slots * far_n * 2, rows_bytes = (size_t)slots * (size_t)n_pattern * row;
        uint8_t *buf = malloc(far_bytes + rows_bytes);
        if (!f || !buf || fread(buf, 1, far_bytes + rows_bytes, f) != far_bytes + rows_bytes) {
            fprintf(stderr, "host_paced: cannot read %zu bytes of %s\n", far_bytes + rows_bytes, pattern);
            return 2;
        }
        fclose(f);
        for (int k = 0; k < slots; k++) {
            memcpy(wmx_rt_far(rt, k), buf + (size_t)k * far_n * 2, far_n * 2);
            long s = 0;
            for (int b = 0; b < B; b++) {
                uint8_t *dst = wmx_pipe_in(wmx_rt_pipe(rt, b), k);
                const int nb = wmx_rt_batch_streams(rt, b);
                for (int r = 0; r < nb; r++, s++)
                    memcpy(dst + (size_t)r * row, buf + far_bytes + ((size_t)k * (size_t)n_pattern + (size_t)(s % n_pattern)) * row, row);
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
    double *lat = malloc(sizeof(double) * (size_t)ticks), *lag = malloc(sizeof(double) * (size_t)ticks);
    /* past the start-up phases of every stage, back to back (tick t of the whole run works on slot t % slots) */
    for (int k = 0; k < prime && rc == 0; k++) rc = wmx_rt_tick(rt, NULL, NULL, NULL);
    const int64_t period = (int64_t)(tick_ms * 1e6);
    const int64_t t0 = now_ns() + period;
    int misses = 0, overruns = 0;
    const double budget = tick_ms - 2.0;
    for (int k = 0; k < ticks && rc == 0; k++) {
        const int64_t due = t0 + (int64_t)k * period;
        sleep_until(due, 200000);
        const int64_t start = now_ns();
        int slot = -1;
        rc = wmx_rt_tick(rt, NULL, &slot, NULL);
        const int64_t end = now_ns();
        lag[k] = (double)(start - due) * 1e-6;
        lat[k] = (double)(end - due) * 1e-6;
        misses += lat[k] > budget;
        overruns += lat[k] > tick_ms;
        if (kept && k >= ticks - keep) { /* behind the clock: the copy of a few rows is not part of the tick */
            for (int j = 0; j < n_sample; j++) {
                long s = sample[j];
                int b = 0;
                while (s >= wmx_rt_batch_streams(rt, b)) s -= wmx_rt_batch_streams(rt, b++);
                memcpy(kept + ((size_t)(k - (ticks - keep)) * (size_t)n_sample + (size_t)j) * row, wmx_pipe_out(wmx_rt_pipe(rt, b), slot) + (size_t)s * row, row);
            }
        }
    }
    if (rc != 0) fprintf(stderr, "host_paced: failed (rc %d): %s\n", rc, wmx_last_error());
    long failed = 0;
    for (int b = 0; b < B; b++) failed += wmx_pipe_failed_steps(wmx_rt_pipe(rt, b));
    wmx_rt_destroy(rt);
    if (rc == 0 && dump) {
        FILE *f = fopen(dump, "wb");
        if (!f || fwrite(kept, row, (size_t)keep * (size_t)n_sample, f) != (size_t)keep * (size_t)n_sample) rc = 4;
        if (f) fclose(f);
    }
    if (rc == 0 && latf) {
        FILE *f = fopen(latf, "wb");
        if (!f || fwrite(lat, sizeof(double), (size_t)ticks, f) != (size_t)ticks) rc = 4;
        if (f) fclose(f);
    }
    int worst = 0;
    double lag_max = 0;
    for (int k = 0; k < ticks; k++) {
        if (lat[k] > lat[worst]) worst = k;
        if (lag[k] > lag_max) lag_max = lag[k];
    }
    const double worst_ms = lat[worst];
    qsort(lat, (size_t)ticks, sizeof(double), cmp_double);
    qsort(lag, (size_t)ticks, sizeof(double), cmp_double);
    printf("{\"host\": \"examples/host_paced.c\", \"kind\": \"%s\", \"streams\": %ld, \"sub_batch\": %d, \"sub_batches\": %d, \"slots\": %d, \"row_bytes\": %zu, "
           "\"tick_ms\": %.3f, \"budget_ms\": %.3f, \"ticks\": %d, \"primed_ticks\": %d, \"p50_ms\": %.4f, \"p99_ms\": %.4f, \"p99_9_ms\": %.4f, \"max_ms\": %.4f, "
           "\"misses\": %d, \"overruns_of_the_period\": %d, \"release_lag_p50_ms\": %.4f, \"release_lag_max_ms\": %.4f, \"worst_tick\": %d, "
           "\"failed_steps\": %ld, \"kept_ticks\": %d, \"rc\": %d}\n",
           kind, S, sub, B, slots, row, tick_ms, budget, ticks, prime, quantile(lat, ticks, 0.5), quantile(lat, ticks, 0.99), quantile(lat, ticks, 0.999),
           worst_ms, misses, overruns, quantile(lag, ticks, 0.5), lag_max, worst, failed, dump ? keep : 0, rc);
    return rc ? 1 : 0;
}
