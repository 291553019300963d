/* host_chain.c -- a C host for the record chain on one or more MI355X: plain C99, the library's C ABI and the HIP
 * runtime API, nothing else.
 *
 * What the reference does in one thread for one sound card -- the heartbeat of wmix_shmem_write_circle,
 * src/wmix.c:613-709: ns_process -> aec_process2 -> agc_process -> vad_process on the captured packet, in place -- done for
 * n_streams streams: one worker thread per GPU (SURVEY 8b "Threading"), streams sharded by contiguous ranges, state resident
 * on its GPU for the whole run, and per 10 ms tick each worker receives the SHARED far-end packet with a hipMemcpyAsync
 * (the single exchange of the path; a host that produces the far-end on a GPU uses one ncclBroadcast instead),
 * uploads its shard's captured packets, makes ONE library call (wmx_chain_process) and downloads the result.
 *
 *   host_chain far.i16 near.i16 out.i16 n_streams n_ticks [n_workers] [freq] [--devices k] [--interval-ms 20] [--far-chunk K]
 *
 * far.i16  int16 [n_ticks][pkt]             the shared far-end          (pkt = freq / 100 samples, mono)
 * near.i16 int16 [n_streams][n_ticks][pkt]  captured audio, stream-major
 * out.i16  same shape as near.i16           what the chain leaves in the daemon's buffer
 * n_workers defaults to the device count; worker w runs on device w % device_count, so a 1-GPU box can still drive
 * several shards (tests/test_host_chain_gpu.py does, and compares out.i16 with the oracle).  --devices k: use the first k
 * devices of the node (default: all of them), so that one binary covers 1 ... 8 GPUs.  --interval-ms 20: the daemon's own cadence --
 * handles made with WMIX_INTERVAL_MS = 20 (src/wmixConf.h:112) and a heartbeat of 20 ms = two 10 ms packets per tick (n_ticks then
 * counts 20 ms heartbeats and every array holds 2 pkt samples per tick).  --far-chunk K: the far-end of K ticks travels at once -- one
 * upload (one ncclBroadcast in the RCCL build) per K ticks instead of one per tick ("one ncclBroadcast per batch of K
 * frames"; the daemon's far-end is 400 ms old when the canceller gets it, src/wmix.c:651-657: it is known long before it is needed).
 *
 * Built with -DWMX_EXAMPLE_RCCL (examples/host_chain_rccl, links librccl) the far-end travels the way north_star puts it:
 * worker 0 alone uploads the packet, and ONE ncclBroadcast per tick (every worker calls it on its own communicator and stream)
 * delivers it from GPU 0 to the others over xGMI.  That needs one device per worker (RCCL does not put two ranks on one GPU).
 *
 * Build (what __graft_entry__.build() runs):
 *   gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/host_chain.c -o examples/host_chain \
 *       -Lwmix_amd -lwmix_amd -L/opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,'$ORIGIN/../wmix_amd' -Wl,-rpath,/opt/rocm/lib
 */
#define _POSIX_C_SOURCE 200809L
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "wmix_amd.h"
#ifdef WMX_EXAMPLE_RCCL
#include <rccl/rccl.h>
#endif

typedef struct {
    int worker, dev, lo, n;      /* this shard: streams [lo, lo + n) on HIP device dev */
    int n_ticks, pkt, freq;      /* pkt: int16 samples of one TICK of one stream (n10 packets of 10 ms) */
    int interval_ms, n10;
    int far_chunk;               /* ticks of far-end per upload / broadcast */
    const int16_t *far_host;     /* [n_ticks][pkt], shared */
    const int16_t *near_host;    /* [n_streams][n_ticks][pkt] */
    int16_t *out_host;
    pthread_barrier_t *tick;
    volatile int *any_failed;    /* set by a worker that failed: from the next collective on, nobody calls it (a worker that
                                  * skipped its ncclBroadcast would leave the others waiting in theirs for ever) */
    int rc;
    double busy_ms;
#ifdef WMX_EXAMPLE_RCCL
    ncclComm_t comm;             /* this worker's rank in the far-end broadcast group (rank == worker, root 0) */
#endif
} Shard;

#define HIP_OK(x)                                                                             \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "worker %d: %s -> %s\n", s->worker, #x, hipGetErrorString(e_));   \
            s->rc = 1;                                                                        \
            goto done;                                                                        \
        }                                                                                     \
    } while (0)
#define WMX_OK(x)                                                                             \
    do {                                                                                      \
        int r_ = (x);                                                                         \
        if (r_ != 0) {                                                                        \
            fprintf(stderr, "worker %d: %s -> %d (%s)\n", s->worker, #x, r_, wmx_last_error()); \
            s->rc = 1;                                                                        \
            goto done;                                                                        \
        }                                                                                     \
    } while (0)

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static void *gpu_worker(void *arg) {
    Shard *s = (Shard *)arg;
    hipStream_t st = NULL;
    wmx_chain *chain = NULL;
    int16_t *d_near = NULL, *d_far = NULL;
    const size_t row = (size_t)s->pkt * sizeof(int16_t);          /* one packet */
    const size_t pitch = (size_t)s->n_ticks * row;                /* host rows are n_ticks packets apart */
    int created = 0;
    if (hipSetDevice(s->dev) == hipSuccess && hipStreamCreate(&st) == hipSuccess &&
        wmx_chain_create(&chain, s->n, 1, s->freq, s->interval_ms, 5, WMX_CHAIN_NS | WMX_CHAIN_AEC | WMX_CHAIN_AGC | WMX_CHAIN_VAD, 1) == 0 &&
        hipMalloc((void **)&d_near, (size_t)s->n * row) == hipSuccess && hipMalloc((void **)&d_far, row * (size_t)s->far_chunk) == hipSuccess)
        created = 1;
    else
        fprintf(stderr, "worker %d: set-up failed (%s)\n", s->worker, wmx_last_error());
    /* the handle remembers its device: from here on nothing depends on the thread's current device */
    for (int t = 0; t < s->n_ticks; t++) {
        pthread_barrier_wait(s->tick); /* the 10 ms heartbeat: every worker starts tick t together */
#ifdef WMX_EXAMPLE_RCCL
        if (!created || s->rc) { /* keep the barrier count even after a failure, and tell the others before their collective */
            *s->any_failed = 1;
            pthread_barrier_wait(s->tick);
            continue;
        }
#else
        if (!created || s->rc) continue; /* keep the barrier count even after a failure */
#endif
        const double t0 = now_ms();
#ifdef WMX_EXAMPLE_RCCL
        /* the far-end reaches GPU 0 from the host and every other GPU from GPU 0 (the path's one exchange) */
        const int fj = t % s->far_chunk;                                        /* tick t's place in its chunk */
        const int fn = s->n_ticks - t < s->far_chunk ? s->n_ticks - t : s->far_chunk; /* ticks in the chunk that starts at t */
        if (fj == 0 && s->worker == 0 &&
            hipMemcpyAsync(d_far, s->far_host + (size_t)t * s->pkt, row * (size_t)fn, hipMemcpyHostToDevice, st) != hipSuccess) {
            fprintf(stderr, "worker 0: far-end upload failed\n");
            s->rc = 1;
        }
        if (s->rc) *s->any_failed = 1;
        pthread_barrier_wait(s->tick); /* everybody knows by now whether everybody will call the collective */
        if (*s->any_failed) {
            s->rc = 1;
            goto done;
        }
        if (fj == 0) { /* ONE collective for the chunk's fn ticks */
            ncclResult_t nr = ncclBroadcast(d_far, d_far, row * (size_t)fn, ncclInt8, 0, s->comm, st);
            if (nr != ncclSuccess) {
                fprintf(stderr, "worker %d: ncclBroadcast -> %s\n", s->worker, ncclGetErrorString(nr));
                s->rc = 1;
                goto done;
            }
        }
#else
        const int fj = t % s->far_chunk;
        const int fn = s->n_ticks - t < s->far_chunk ? s->n_ticks - t : s->far_chunk;
        if (fj == 0) HIP_OK(hipMemcpyAsync(d_far, s->far_host + (size_t)t * s->pkt, row * (size_t)fn, hipMemcpyHostToDevice, st));
#endif
        HIP_OK(hipMemcpy2DAsync(d_near, row, s->near_host + ((size_t)s->lo * s->n_ticks + t) * s->pkt, pitch, row, (size_t)s->n,
                                hipMemcpyHostToDevice, st));
        /* a stream's tick in one piece: n10 packets of pkt / n10 samples, streams pkt apart */
        WMX_OK(wmx_chain_process(chain, d_far + (size_t)fj * s->pkt, s->pkt / s->n10, d_near, d_near, s->n10, s->pkt, s->pkt / s->n10, NULL, NULL, NULL, st));
        HIP_OK(hipMemcpy2DAsync(s->out_host + ((size_t)s->lo * s->n_ticks + t) * s->pkt, pitch, d_near, row, row, (size_t)s->n,
                                hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        s->busy_ms += now_ms() - t0;
    done:;
    }
    if (!created) s->rc = 1;
    if (chain) wmx_chain_destroy(chain);
    if (d_near) (void)hipFree(d_near);
    if (d_far) (void)hipFree(d_far);
    if (st) (void)hipStreamDestroy(st);
    return NULL;
}

static void *read_file(const char *path, size_t bytes) {
    FILE *f = fopen(path, "rb");
    void *p = malloc(bytes ? bytes : 1);
    if (!f || !p || fread(p, 1, bytes, f) != bytes) {
        fprintf(stderr, "host_chain: cannot read %zu bytes from %s\n", bytes, path);
        exit(2);
    }
    fclose(f);
    return p;
}

int main(int argc, char **argv) {
    int want_dev = 0, interval_ms = 10, far_chunk = 1;
    for (int i = 1; i + 1 < argc;) /* --devices k / --interval-ms m / --far-chunk K, wherever they stand: taken out of the positional arguments */
        if (strcmp(argv[i], "--devices") == 0 || strcmp(argv[i], "--interval-ms") == 0 || strcmp(argv[i], "--far-chunk") == 0) {
            if (argv[i][2] == 'd')
                want_dev = atoi(argv[i + 1]);
            else if (argv[i][2] == 'f')
                far_chunk = atoi(argv[i + 1]);
            else
                interval_ms = atoi(argv[i + 1]);
            for (int j = i; j + 2 < argc; j++) argv[j] = argv[j + 2];
            argc -= 2;
        } else {
            i++;
        }
    if (interval_ms != 10 && interval_ms != 20) {
        fprintf(stderr, "host_chain: --interval-ms 10 or 20\n");
        return 2;
    }
    if (far_chunk < 1 || far_chunk > 4096) {
        fprintf(stderr, "host_chain: --far-chunk 1 .. 4096\n");
        return 2;
    }
    if (argc < 6) {
        fprintf(stderr, "usage: %s far.i16 near.i16 out.i16 n_streams n_ticks [n_workers] [freq] [--devices k] [--interval-ms 20] [--far-chunk K]\n", argv[0]);
        return 2;
    }
    const int n_streams = atoi(argv[4]), n_ticks = atoi(argv[5]);
    int n_dev = wmx_device_count();
    if (n_dev < 1) {
        fprintf(stderr, "host_chain: no HIP device (%s)\n", wmx_last_error());
        return 3;
    }
    if (want_dev > n_dev || want_dev < 0) {
        fprintf(stderr, "host_chain: --devices %d, %d present\n", want_dev, n_dev);
        return 2;
    }
    if (want_dev > 0) n_dev = want_dev;
    int n_workers = argc > 6 ? atoi(argv[6]) : n_dev;
    const int freq = argc > 7 ? atoi(argv[7]) : 16000;
    const int n10 = interval_ms / 10, pkt = freq / 100 * n10; /* samples of one tick of one stream */
    if (n_workers < 1 || n_workers > n_streams || n_ticks < 1) return 2;
#ifdef WMX_EXAMPLE_RCCL
    if (n_workers > n_dev) {
        fprintf(stderr, "host_chain_rccl: %d workers need %d devices, %d present\n", n_workers, n_workers, n_dev);
        return 2;
    }
    ncclComm_t *comms = calloc((size_t)n_workers, sizeof(ncclComm_t));
    {
        int *devs = calloc((size_t)n_workers, sizeof(int));
        for (int w = 0; w < n_workers; w++) devs[w] = w;
        ncclResult_t nr = ncclCommInitAll(comms, n_workers, devs);
        free(devs);
        if (nr != ncclSuccess) {
            fprintf(stderr, "host_chain_rccl: ncclCommInitAll -> %s\n", ncclGetErrorString(nr));
            return 3;
        }
    }
#endif
    int16_t *far = read_file(argv[1], (size_t)n_ticks * pkt * 2);
    int16_t *near = read_file(argv[2], (size_t)n_streams * n_ticks * pkt * 2);
    int16_t *out = calloc((size_t)n_streams * n_ticks * pkt, 2);
    Shard *sh = calloc((size_t)n_workers, sizeof(Shard));
    pthread_t *th = calloc((size_t)n_workers, sizeof(pthread_t));
    pthread_barrier_t tick;
    static volatile int any_failed = 0;
    pthread_barrier_init(&tick, NULL, (unsigned)n_workers);
    /* contiguous stream ranges, the remainder spread over the first workers (wmix_amd/shard.py: stream_range) */
    const int base = n_streams / n_workers, rem = n_streams % n_workers;
    const double t0 = now_ms();
    for (int w = 0; w < n_workers; w++) {
        Shard *s = &sh[w];
        s->worker = w;
        s->dev = w % n_dev;
        s->lo = w * base + (w < rem ? w : rem);
        s->n = base + (w < rem ? 1 : 0);
        s->n_ticks = n_ticks;
        s->pkt = pkt;
        s->freq = freq;
        s->interval_ms = interval_ms;
        s->far_chunk = far_chunk;
        s->n10 = n10;
        s->far_host = far;
        s->near_host = near;
        s->out_host = out;
        s->tick = &tick;
        s->any_failed = &any_failed;
#ifdef WMX_EXAMPLE_RCCL
        s->comm = comms[w];
#endif
        pthread_create(&th[w], NULL, gpu_worker, s);
    }
    int rc = 0;
    for (int w = 0; w < n_workers; w++) {
        pthread_join(th[w], NULL);
        rc |= sh[w].rc;
    }
    const double wall = now_ms() - t0;
#ifdef WMX_EXAMPLE_RCCL
    for (int w = 0; w < n_workers; w++) (void)ncclCommDestroy(comms[w]);
    free(comms);
#endif
    if (rc == 0) {
        FILE *f = fopen(argv[3], "wb");
        if (!f || fwrite(out, 2, (size_t)n_streams * n_ticks * pkt, f) != (size_t)n_streams * n_ticks * pkt) rc = 4;
        if (f) fclose(f);
    }
#ifdef WMX_EXAMPLE_RCCL
    printf("{\"far_end\": \"ncclBroadcast from GPU 0\", ");
#else
    printf("{\"far_end\": \"hipMemcpyAsync per worker\", ");
#endif
    printf("\"workers\": %d, \"devices\": %d, \"streams\": %d, \"ticks\": %d, \"wall_ms\": %.3f, \"busy_ms_per_tick\": [", n_workers, n_dev,
           n_streams, n_ticks, wall);
    for (int w = 0; w < n_workers; w++) printf("%s%.4f", w ? ", " : "", sh[w].busy_ms / n_ticks);
    printf("], \"rc\": %d}\n", rc);
    return rc;
}
