/* oracle/ref_mix_driver.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Our own command-line driver around the REAL reference mixer arithmetic
 * (src/wmix.c:49-222 wmix_len_of_out/in + wmix_pcm_zoom, src/wmix.c:1639-1957
 * wmix_load_data) as compiled into oracle/_ref/libwmixref_mix.so.  That library
 * keeps the daemon's HAL / task-thread references undefined, so it only loads
 * with lazy binding -- hence an executable instead of ctypes (which forces
 * RTLD_NOW).  It is used to validate oracle/orc_mix.c and to generate
 * tests/golden/mix_*.bin.
 *
 *   ref_mix_driver lenout inChn inFreq inLen outChn outFreq          -> prints n
 *   ref_mix_driver lenin  inChn inFreq outChn outFreq outLen         -> prints n
 *   ref_mix_driver zoom   inChn inFreq outChn outFreq  <in.pcm >out.pcm
 *   ref_mix_driver load   freq chn reduceMode reduceArg nSrc srcBytes startOff <srcs.pcm >ring.bin
 *        reads nSrc sources of srcBytes each, loads them one after the other
 *        with head=NULL,tick=0 into a fresh ring whose play head sits startOff
 *        bytes from ring start; writes the whole ring (WMIX_BUFF_SIZE bytes) then
 *        per source: uint32 tick, uint32 head offset.
 *   ref_mix_driver consts  -> prints WMIX_CHN WMIX_FREQ WMIX_BUFF_SIZE VIEW_PLAY_CORRECT
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "wmix.h"

static unsigned char *slurp(size_t *n)
{
    size_t cap = 1 << 20, len = 0, r;
    unsigned char *b = malloc(cap);
    while ((r = fread(b + len, 1, cap - len, stdin)) > 0) {
        len += r;
        if (len == cap) b = realloc(b, cap *= 2);
    }
    *n = len;
    return b;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    if (!strcmp(argv[1], "consts")) {
        printf("%d %d %d %d\n", WMIX_CHN, WMIX_FREQ, WMIX_BUFF_SIZE, VIEW_PLAY_CORRECT);
        return 0;
    }
    if (!strcmp(argv[1], "lenout") && argc == 7) {
        printf("%u\n", wmix_len_of_out(atoi(argv[2]), atoi(argv[3]), strtoul(argv[4], 0, 10), atoi(argv[5]), atoi(argv[6])));
        return 0;
    }
    if (!strcmp(argv[1], "lenin") && argc == 7) {
        printf("%u\n", wmix_len_of_in(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), strtoul(argv[6], 0, 10)));
        return 0;
    }
    if (!strcmp(argv[1], "zoom") && argc == 6) {
        size_t n;
        unsigned char *in = slurp(&n);
        unsigned char *out = calloc(n * 8 + 64, 1);
        uint32_t m = wmix_pcm_zoom(atoi(argv[2]), atoi(argv[3]), in, (uint32_t)n, atoi(argv[4]), atoi(argv[5]), out);
        fwrite(out, 1, m, stdout);
        return 0;
    }
    if (!strcmp(argv[1], "load") && argc == 9) {
        int freq = atoi(argv[2]), chn = atoi(argv[3]), rmode = atoi(argv[4]), rarg = atoi(argv[5]);
        int nsrc = atoi(argv[6]);
        uint32_t sbytes = strtoul(argv[7], 0, 10), start = strtoul(argv[8], 0, 10);
        size_t n;
        unsigned char *src = slurp(&n);
        if (n < (size_t)nsrc * sbytes) return 3;
        WMix_Struct *w = calloc(1, sizeof(WMix_Struct));
        w->buff = calloc(WMIX_BUFF_SIZE + 64, 1);
        w->start.U8 = w->buff;
        w->end.U8 = w->buff + WMIX_BUFF_SIZE;
        w->head.U8 = w->tail.U8 = w->buff + start;
        w->run = true;
        w->reduceMode = rmode;
        uint32_t *meta = calloc(2 * nsrc, sizeof(uint32_t));
        for (int i = 0; i < nsrc; i++) {
            WMix_Point s = {.U8 = src + (size_t)i * sbytes}, h = {.U8 = NULL};
            uint32_t tick = 0;
            h = wmix_load_data(w, s, sbytes, freq, chn, 16, h, rarg, &tick);
            meta[2 * i] = tick;
            meta[2 * i + 1] = (uint32_t)(h.U8 - w->buff);
        }
        fwrite(w->buff, 1, WMIX_BUFF_SIZE, stdout);
        fwrite(meta, sizeof(uint32_t), 2 * nsrc, stdout);
        return 0;
    }
    return 2;
}
