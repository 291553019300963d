/* host_tick.c -- a C host of the daemon's whole tick for many mixers side by side: plain C99, the library's C ABI (wmx_tick_*) and the
 * HIP runtime API for the buffers, nothing else.
 *
 * What ONE wmix daemon does per WMIX_INTERVAL_MS = 20 ms (src/wmix.c:1347-1440 with wmix_shmem_write_circle, :528-780, inside) --
 *   task threads: wmix_load_data of every source into the play ring          wmx_tick_load
 *   play thread:  drain one package -> playPkgBuff_add -> sound card,        wmx_tick_play   (the group's far-end package is left on
 *                 playPkgBuff_get(AEC_INTERVALMS)                                             the device: wmx_tick_far)
 *   the room:     what the microphones pick up (this harness: local + the far-end delayed by 40 samples, halved -- the model of
 *                 tests/test_tick_gpu.py and oracle/loader.py tick_room), computed HERE on the host
 *   heartbeat:    ns -> aec_process2(far) -> agc -> vad in place, [rwTest: load the recording back], zoom to 1 x 8000
 *                                                                            wmx_tick_record
 * -- for n_groups daemons with n_src sources and n_rec record streams each, in the 1 x 8000 Hz format all three platform
 * directories of the reference ship.
 *
 *   host_tick src.i16 local.i16 out.i16 n_groups n_src n_rec n_ticks src_freq src_chn [--platform alsa|hi3516|t31] [--rwtest]
 *
 * src.i16    int16 [n_ticks][n_groups][n_src][20 ms of (src_freq, src_chn)]   what the task threads play
 * local.i16  int16 [n_ticks][n_groups * n_rec][160]                          the rooms without their loudspeakers
 * out.i16    int16 [n_ticks][ n_groups play | n_groups far | n_groups * n_rec record ][160]
 * --platform: PLAT_AEC_INTERVALMS / PLAT_PLAY_CORRECT of platform/<name>/plat.h (400 ms / 3200 B, 700 / 0, 0 / 0; default alsa)
 * --rwtest:   wmix->rwTest (src/wmix.c:714-732)
 * Prints one JSON line.  tests/test_host_chain_gpu.py compares out.i16 with one oracle daemon per group.
 *
 * Build (what __graft_entry__.build() runs):
 *   gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/host_tick.c -o examples/host_tick \
 *       -Lwmix_amd -lwmix_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../wmix_amd' -Wl,-rpath,/opt/rocm/lib
 */
#define _POSIX_C_SOURCE 200809L
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "wmix_amd.h"

#define PKG 160        /* int16 of one 20 ms package at 1 x 8000 Hz */
#define ECHO_DELAY 40  /* samples between loudspeaker and microphone in the harness' room */

static void *read_file(const char *path, size_t bytes) {
    FILE *f = fopen(path, "rb");
    void *p = malloc(bytes ? bytes : 1);
    if (!f || !p || fread(p, 1, bytes, f) != bytes) {
        fprintf(stderr, "host_tick: cannot read %zu bytes of %s\n", bytes, path);
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

#define HIP_OK(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "host_tick: %s: %s\n", #call, hipGetErrorString(e_));            \
            return 3;                                                                        \
        }                                                                                    \
    } while (0)
#define WMX_OK(call)                                                                         \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_ != 0) {                                                                      \
            fprintf(stderr, "host_tick: %s = %d: %s\n", #call, rc_, wmx_last_error());       \
            return 4;                                                                        \
        }                                                                                    \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 10) {
        fprintf(stderr, "usage: %s src.i16 local.i16 out.i16 n_groups n_src n_rec n_ticks src_freq src_chn [--platform name] [--rwtest]\n", argv[0]);
        return 2;
    }
    const int G = atoi(argv[4]), n_src = atoi(argv[5]), R = atoi(argv[6]), T = atoi(argv[7]), sfreq = atoi(argv[8]), schn = atoi(argv[9]);
    int aec_ms = 400, rwtest = 0;
    long correct = -1; /* -1: the library's default = platform/alsa */
    const char *platform = "alsa";
    for (int i = 10; i < argc; i++) {
        if (!strcmp(argv[i], "--rwtest")) {
            rwtest = 1;
        } else if (!strcmp(argv[i], "--platform") && i + 1 < argc) {
            platform = argv[++i];
            if (!strcmp(platform, "alsa")) {
                aec_ms = 400, correct = 3200;
            } else if (!strcmp(platform, "hi3516")) {
                aec_ms = 700, correct = 0;
            } else if (!strcmp(platform, "t31")) {
                aec_ms = 0, correct = 0;
            } else {
                fprintf(stderr, "host_tick: no platform directory '%s' in the reference\n", platform);
                return 2;
            }
        } else {
            fprintf(stderr, "host_tick: what is '%s'?\n", argv[i]);
            return 2;
        }
    }
    if (G < 1 || n_src < 1 || R < 1 || T < 1 || sfreq < 1000 || (schn != 1 && schn != 2)) return 2;
    const size_t per = (size_t)sfreq / 1000 * 20 * schn;  /* int16 of one source's 20 ms */
    const size_t srow = per + 2 * (size_t)schn;            /* + the frame the up-sampling fill looks ahead to (src/wmix.c:1857) */
    const size_t S = (size_t)G * R;
    int16_t *src = read_file(argv[1], (size_t)T * G * n_src * per * 2);
    int16_t *local = read_file(argv[2], (size_t)T * S * PKG * 2);
    const size_t out_row = ((size_t)2 * G + S) * PKG;
    int16_t *out = calloc((size_t)T * out_row, 2);
    int16_t *farline = calloc((size_t)2 * G * PKG, 2); /* per group: the previous far-end package, then this one */
    int16_t *near = malloc(S * PKG * 2), *pad = calloc((size_t)G * n_src * srow, 2);
    if (!out || !farline || !near || !pad) return 2;

    wmx_tick *h = NULL;
    WMX_OK(wmx_tick_create(&h, G, R, 1, 8000, 20, aec_ms, 5, WMX_CHAIN_NS | WMX_CHAIN_AEC | WMX_CHAIN_AGC | WMX_CHAIN_VAD));
    if (correct >= 0) WMX_OK(wmx_tick_set_play_correct(h, (uint32_t)correct));
    if (rwtest) WMX_OK(wmx_tick_rw_test(h, 1));
    if (wmx_tick_package_samples(h) != PKG) return 5;
    int16_t *d_src = NULL, *d_play = NULL, *d_rec = NULL, *d_zoom = NULL;
    HIP_OK(hipMalloc((void **)&d_src, (size_t)G * n_src * srow * 2));
    HIP_OK(hipMalloc((void **)&d_play, (size_t)G * PKG * 2));
    HIP_OK(hipMalloc((void **)&d_rec, S * PKG * 2));
    HIP_OK(hipMalloc((void **)&d_zoom, S * PKG * 2));
    uint32_t head = UINT32_MAX, tick = 0; /* the sources' common cursor: NULL head = first call, like a task thread that just started */
    const double t0 = now_ms();
    for (int t = 0; t < T; t++) {
        int16_t *o = out + (size_t)t * out_row;
        /* the task threads */
        for (size_t r = 0; r < (size_t)G * n_src; r++) memcpy(pad + r * srow, src + ((size_t)t * G * n_src + r) * per, per * 2);
        HIP_OK(hipMemcpy(d_src, pad, (size_t)G * n_src * srow * 2, hipMemcpyHostToDevice));
        WMX_OK(wmx_tick_load(h, d_src, (uint32_t)(per * 2), sfreq, schn, 16, n_src, (long)(n_src * srow), (long)srow, 1, &head, &tick, NULL));
        /* the play thread */
        WMX_OK(wmx_tick_play(h, d_play, PKG, NULL));
        HIP_OK(hipMemcpy(o, d_play, (size_t)G * PKG * 2, hipMemcpyDeviceToHost));
        for (int g = 0; g < G; g++) memcpy(farline + (size_t)g * 2 * PKG, farline + (size_t)g * 2 * PKG + PKG, PKG * 2);
        HIP_OK(hipMemcpy2D(farline + PKG, 2 * PKG * 2, wmx_tick_far(h), PKG * 2, PKG * 2, (size_t)G, hipMemcpyDeviceToHost));
        for (int g = 0; g < G; g++) memcpy(o + ((size_t)G + g) * PKG, farline + (size_t)g * 2 * PKG + PKG, PKG * 2);
        /* the rooms */
        for (size_t s = 0; s < S; s++) {
            const int16_t *line = farline + (s / R) * 2 * PKG, *loc = local + ((size_t)t * S + s) * PKG;
            for (int i = 0; i < PKG; i++) {
                int v = loc[i] + (line[PKG + i - ECHO_DELAY] >> 1);
                near[s * PKG + i] = (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
            }
        }
        /* the heartbeats */
        HIP_OK(hipMemcpy(d_rec, near, S * PKG * 2, hipMemcpyHostToDevice));
        uint32_t zoomed = 0;
        WMX_OK(wmx_tick_record(h, d_rec, PKG, d_zoom, PKG, PKG * 2, &zoomed, NULL));
        if (zoomed != PKG * 2) return 6;
        HIP_OK(hipMemcpy(o + (size_t)2 * G * PKG, d_rec, S * PKG * 2, hipMemcpyDeviceToHost));
    }
    const double wall = now_ms() - t0;
    wmx_tick_destroy(h);
    FILE *f = fopen(argv[3], "wb");
    int rc = (!f || fwrite(out, 2, (size_t)T * out_row, f) != (size_t)T * out_row) ? 7 : 0;
    if (f) fclose(f);
    printf("{\"groups\": %d, \"sources\": %d, \"record_streams\": %d, \"ticks\": %d, \"platform\": \"%s\", \"aec_delay_ms\": %d, \"rw_test\": %d, "
           "\"wall_ms\": %.3f, \"ms_per_tick\": %.4f, \"rc\": %d}\n", G, n_src, R, T, platform, aec_ms, rwtest, wall, wall / T, rc);
    return rc;
}
