/* host_legacy_threads.c -- the reference's THREADING over the legacy signatures (include/wmix_compat.h), in plain C99 + pthreads.
 *
 * The daemon calls its per-handle functions from many threads at once (SURVEY.md 8b "Threading"): six task threads feed the play ring
 * with wmix_load_data (src/wmixTask.c:85, 973, 1311, 1484, 1704, 1927), the message thread turns the AGC with agc_addition
 * (src/wmix.c:1070) while the record thread runs the four-call heartbeat ns_process -> aec_process2 -> agc_process -> vad_process
 * (src/wmix.c:613-709).  This host does exactly that against libwmix_amd.so, with a BATCH of `batch_streams` streams running
 * wmx_chain_process back to back on a BLOCKING stream of the same process beside it (a gateway that serves legacy callers and a
 * batch at once), and reports what the heartbeat costs alone and in that company:
 *
 *   phase 1  beats [0, n / 2):   the heartbeat thread alone
 *   phase 2  beats [n / 2, n):   + six loader threads, the agc_addition thread, the batch thread
 *
 * The heartbeat's handles live through both phases, so its output is ONE run of n beats and is checked as such; every loader owns a
 * 1 x 8000 ring (WMix_Struct_Head) and loads one 20 ms chunk of 2 x 16000 per beat with its cursor carried along; the agc_addition
 * thread owns an AGC handle, turns it (value 3 + call % 5) in front of every agc_process, and also hammers agc_addition(5) on the
 * HEARTBEAT's handle (its own value: no audible change, but the lock between agc_addition and agc_process is exercised).
 *
 *   host_legacy_threads <dir> <n_beats> <batch_streams>
 *
 * reads  <dir>/hb_far.i16, hb_near.i16 [n_beats][320] (16 kHz mono, 20 ms); src_<k>.i16 [n_beats][1280] for k = 0..5; agc_in.i16 [n_beats][320]
 * writes <dir>/hb_out.i16; ring_<k>.i16 [8000] + ring_<k>.meta (uint32 head offset, tick); agc_out.i16
 * prints one JSON line: the heartbeat's microseconds per beat (p50 / p99 / max) in both phases, the batch's steps during phase 2.
 *
 * Build (what __graft_entry__.build() runs):
 *   gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/host_legacy_threads.c -o examples/host_legacy_threads \
 *       -Lwmix_amd -lwmix_amd -L/opt/rocm/lib -lamdhip64 -lpthread -lm
 */
#define _POSIX_C_SOURCE 200809L
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "wmix_amd.h"
#include "wmix_compat.h"

#define BEAT 320   /* 20 ms of 16 kHz mono */
#define CHUNK 1280 /* 20 ms of 2 x 16000, int16 elements */
#define N_LOAD 6

static const char *g_dir;
static int g_beats;
static int g_go = 0, g_stop = 0; /* read and written with the __atomic builtins (GO / STOP below) */
#define GO() __atomic_load_n(&g_go, __ATOMIC_ACQUIRE)
#define STOP() __atomic_load_n(&g_stop, __ATOMIC_ACQUIRE)

static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

static void *read_file(const char *name, size_t bytes) {
    char path[512];
    snprintf(path, sizeof(path), "%s/%s", g_dir, name);
    FILE *f = fopen(path, "rb");
    void *p = malloc(bytes);
    if (!f || !p || fread(p, 1, bytes, f) != bytes) {
        fprintf(stderr, "host_legacy_threads: cannot read %zu bytes of %s\n", bytes, path);
        exit(2);
    }
    fclose(f);
    return p;
}

static void write_file(const char *name, const void *p, size_t bytes) {
    char path[512];
    snprintf(path, sizeof(path), "%s/%s", g_dir, name);
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) {
        fprintf(stderr, "host_legacy_threads: cannot write %s\n", path);
        exit(2);
    }
    fclose(f);
}

static int cmp_double(const void *a, const void *b) {
    const double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : x > y;
}
static double quantile(double *v, int n, double q) {
    qsort(v, (size_t)n, sizeof(double), cmp_double);
    const double pos = q * (n - 1);
    const int lo = (int)floor(pos), hi = lo + 1 < n ? lo + 1 : lo;
    return v[lo] + (v[hi] - v[lo]) * (pos - lo);
}

/* ---- the task threads: wmix_load_data, one 20 ms chunk per beat, cursor carried along */
typedef struct {
    int k;
} LoadArg;
static void *loader(void *vp) {
    const int k = ((LoadArg *)vp)->k;
    char name[64];
    snprintf(name, sizeof(name), "src_%d.i16", k);
    int16_t *src = read_file(name, (size_t)g_beats * CHUNK * 2);
    int16_t *ring = calloc(8000 + 8, 2);
    WMix_Struct_Head w;
    memset(&w, 0, sizeof(w));
    w.start.S16 = ring;
    w.end.U8 = w.start.U8 + 16000;
    w.head.S16 = ring;
    w.run = true;
    w.reduceMode = 1;
    WMix_Point head = {.U8 = NULL};
    uint32_t tick = 0;
    while (!GO()) {
    }
    for (int b = g_beats / 2; b < g_beats; b++) { /* phase 2 only */
        WMix_Point sp = {.S16 = src + (size_t)b * CHUNK};
        head = wmix_load_data(&w, sp, CHUNK * 2, 16000, 2, 16, head, 1, &tick);
        w.tick += 0; /* (nobody plays: the ring's head stands still, the sources pile up behind it) */
    }
    snprintf(name, sizeof(name), "ring_%d.i16", k);
    write_file(name, ring, 16000);
    uint32_t meta[2] = {head.U8 ? (uint32_t)(head.U8 - w.start.U8) : 0xffffffffu, tick};
    snprintf(name, sizeof(name), "ring_%d.meta", k);
    write_file(name, meta, sizeof(meta));
    free(src);
    free(ring);
    return NULL;
}

/* ---- the message thread: agc_addition, on its own handle (checked) and on the heartbeat's (its own value: the lock is exercised) */
static void *g_hb_agc;
static void *turner(void *unused) {
    (void)unused;
    int16_t *x = read_file("agc_in.i16", (size_t)g_beats * BEAT * 2);
    void *agc = agc_init(1, 16000, 20, 5, NULL);
    if (!agc) exit(3);
    while (!GO()) {
    }
    for (int b = g_beats / 2; b < g_beats; b++) {
        agc_addition(agc, (uint8_t)(3 + b % 5));
        agc_process(agc, x + (size_t)b * BEAT, x + (size_t)b * BEAT, BEAT);
        agc_addition(g_hb_agc, 5);
    }
    agc_release(agc);
    write_file("agc_out.i16", x, (size_t)g_beats * BEAT * 2);
    free(x);
    return NULL;
}

/* ---- the batch: wmx_chain_process back to back on a BLOCKING stream of the same process */
static long g_batch_steps = 0; /* written by the batch thread, read after it was joined */
static int g_batch_streams;
static void *batch(void *unused) {
    (void)unused;
    if (g_batch_streams < 1) return NULL;
    wmx_chain *ch = NULL;
    hipStream_t s = NULL;
    int16_t *d_pcm = NULL, *d_far = NULL;
    const unsigned stages = WMX_CHAIN_NS | WMX_CHAIN_AEC | WMX_CHAIN_AGC | WMX_CHAIN_VAD;
    if (wmx_chain_create(&ch, g_batch_streams, 1, 16000, 20, 5, stages, 1) != 0 || hipStreamCreate(&s) != hipSuccess /* a BLOCKING stream */ ||
        hipMalloc((void **)&d_pcm, (size_t)g_batch_streams * BEAT * 2) != hipSuccess || hipMalloc((void **)&d_far, BEAT * 2) != hipSuccess ||
        hipMemset(d_pcm, 0, (size_t)g_batch_streams * BEAT * 2) != hipSuccess || hipMemset(d_far, 0, BEAT * 2) != hipSuccess) {
        fprintf(stderr, "host_legacy_threads: batch setup: %s\n", wmx_last_error());
        exit(3);
    }
    while (!GO()) {
    }
    while (!STOP()) { /* a few steps queued ahead, like any streaming host */
        for (int i = 0; i < 4; i++)
            if (wmx_chain_process(ch, d_far, 160, d_pcm, d_pcm, 2, BEAT, 160, NULL, NULL, NULL, s) != 0) {
                fprintf(stderr, "host_legacy_threads: batch step: %s\n", wmx_last_error());
                exit(3);
            }
        hipStreamSynchronize(s);
        g_batch_steps += 4;
    }
    wmx_chain_destroy(ch);
    hipFree(d_pcm);
    hipFree(d_far);
    hipStreamDestroy(s);
    return NULL;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <dir> <n_beats> <batch_streams>\n", argv[0]);
        return 2;
    }
    g_dir = argv[1];
    g_beats = atoi(argv[2]);
    g_batch_streams = atoi(argv[3]);
    if (g_beats < 4 || g_beats % 2) return 2;
    int16_t *far = read_file("hb_far.i16", (size_t)g_beats * BEAT * 2), *buf = read_file("hb_near.i16", (size_t)g_beats * BEAT * 2);
    void *ns = ns_init(1, 16000, NULL), *aec = aec_init(1, 16000, 20, NULL), *agc = agc_init(1, 16000, 20, 5, NULL), *vad = vad_init(1, 16000, 20, NULL);
    if (!ns || !aec || !agc || !vad) {
        fprintf(stderr, "host_legacy_threads: *_init: %s\n", wmx_last_error());
        return 3;
    }
    g_hb_agc = agc;
    pthread_t tl[N_LOAD], tt, tb;
    LoadArg la[N_LOAD];
    for (int k = 0; k < N_LOAD; k++) {
        la[k].k = k;
        pthread_create(&tl[k], NULL, loader, &la[k]);
    }
    pthread_create(&tt, NULL, turner, NULL);
    pthread_create(&tb, NULL, batch, NULL);
    double *us = malloc(sizeof(double) * (size_t)g_beats);
    int rc = 0;
    for (int b = 0; b < g_beats && rc == 0; b++) {
        if (b == g_beats / 2) {
            struct timespec nap = {1, 0}; /* the company's handles and device buffers are made by now */
            nanosleep(&nap, NULL);
            __atomic_store_n(&g_go, 1, __ATOMIC_RELEASE);
            nap.tv_sec = 0, nap.tv_nsec = 50000000; /* ... and at work */
            nanosleep(&nap, NULL);
        }
        int16_t *p = buf + (size_t)b * BEAT;
        const double t0 = now_us();
        ns_process(ns, p, p, BEAT);
        rc = aec_process2(aec, far + (size_t)b * BEAT, p, p, BEAT, 0);
        if (rc == 0) rc = agc_process(agc, p, p, BEAT);
        vad_process(vad, p, BEAT);
        us[b] = now_us() - t0;
        struct timespec gap = {0, 300000}; /* the daemon's heartbeat is paced; a third of a millisecond keeps the test short */
        nanosleep(&gap, NULL);
    }
    __atomic_store_n(&g_stop, 1, __ATOMIC_RELEASE);
    for (int k = 0; k < N_LOAD; k++) pthread_join(tl[k], NULL);
    pthread_join(tt, NULL);
    pthread_join(tb, NULL);
    ns_release(ns), aec_release(aec), agc_release(agc), vad_release(vad);
    write_file("hb_out.i16", buf, (size_t)g_beats * BEAT * 2);
    const int h = g_beats / 2, warm = h / 5;
    double a50 = quantile(us + warm, h - warm, 0.5), a99 = quantile(us + warm, h - warm, 0.99), amax = us[h - 1];
    double b50 = quantile(us + h, h, 0.5), b99 = quantile(us + h, h, 0.99), bmax = us[g_beats - 1]; /* (sorted in place: the last is the largest) */
    printf("{\"beats\": %d, \"batch_streams\": %d, \"batch_steps_in_phase_2\": %ld, \"alone_us\": {\"p50\": %.1f, \"p99\": %.1f, \"max\": %.1f}, "
           "\"in_company_us\": {\"p50\": %.1f, \"p99\": %.1f, \"max\": %.1f}, \"p99_ratio\": %.3f, \"rc\": %d}\n",
           g_beats, g_batch_streams, g_batch_steps, a50, a99, amax, b50, b99, bmax, b99 / a99, rc);
    return rc ? 1 : 0;
}
