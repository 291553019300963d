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
 *   ref_mix_driver rtpsend chn freq <ring.pcm >packets.bin
 *        the loop body of wmix_thread_rtp_send_pcma (src/wmixTask.c:1124-1143) over stdin cut into chunks of
 *        wmix_len_of_in(..20 ms..) bytes: wmix_pcm_zoom -> PCM2G711a -> timestamp += n/chn -> rtp_send over UDP
 *        loopback; what arrives on the wire is written as [uint32 len][bytes] per packet.
 *   ref_mix_driver pkgfifo delayms <ops.bin >out.bin
 *        stdin = a sequence of WMIX_PKG_SIZE-byte packets; each is pushed with playPkgBuff_add and followed by a
 *        playPkgBuff_get(buff, delayms) (src/wmix.c:487-526); the WMIX_PKG_SIZE bytes of every get are written out.
 *   ref_mix_driver tick nSrc srcFreq srcChn nRec nTicks stages agcValue <in.bin >out.bin
 *        ONE daemon's tick composed from the reference's own functions, in the order the play thread runs them with
 *        WMIX_RECORD_PLAY_SYNC (src/wmix.c:1347-1440 with wmix_shmem_write_circle inside, :528-780).  Per tick, stdin holds nSrc
 *        sources of one WMIX_INTERVAL_MS each in (srcFreq, srcChn) and then nRec captured packages ("local": what the microphone
 *        would pick up WITHOUT the loudspeaker).  The task threads' wmix_load_data calls (every source keeps its head / tick like a
 *        task thread does), the play thread's drain of one package (its loop body restated here: copy a sample out, zero it, head
 *        and tick move on, wrap -- :1347-1366), playPkgBuff_add, playPkgBuff_get(AEC_INTERVALMS) = the far-end, the room (near =
 *        sat(local + far delayed by 40 samples >> 1): the harness' own input model), then per record handle set ns_process ->
 *        aec_process2(far, near, near, .., 0) -> agc_process -> vad_process (:613-709, `stages` bits 1 2 4 8 = the webrtcEnable
 *        switches; bit 16 = WR_NS_PA: ns_process over the played package in front of playPkgBuff_add, :1370-1386; bit 32 = wmix->rwTest:
 *        record handle set 0's output goes back into the play ring through wmix_load_data with a cursor of its own, :714-732) and wmix_pcm_zoom to 1 x 8000 (:730).  Written per tick: played package, far-end package, then per record handle set
 *        its chain output and the zoomed copy (2 * 8000 * WMIX_INTERVAL_MS / 1000 bytes).
 *   ref_mix_driver rtprecv <packets.bin >pcm.bin
 *        every [uint32 len][bytes] record is sent to a socket opened with rtp_socket(bind) and taken through
 *        rtp_recv + G711a2PCM (src/wmixTask.c:1278-1282); writes [uint32 pcm_bytes][pcm][uint16 header seq as stored].
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "wmix.h"
#include "rtp.h"
#include "g711codec.h"
#include "webrtc.h"
#include <arpa/inet.h>
#include <sys/socket.h>
#include <unistd.h>

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
    if (!strcmp(argv[1], "rtpsend") && argc == 4) {
        const int chn = atoi(argv[2]), freq = atoi(argv[3]);
        size_t n;
        unsigned char *in = slurp(&n);
        /* a plain receiving socket on an ephemeral loopback port */
        int rx = socket(AF_INET, SOCK_DGRAM, 0);
        struct sockaddr_in a;
        socklen_t al = sizeof(a);
        memset(&a, 0, sizeof(a));
        a.sin_family = AF_INET;
        a.sin_addr.s_addr = inet_addr("127.0.0.1");
        if (bind(rx, (struct sockaddr *)&a, sizeof(a)) < 0 || getsockname(rx, (struct sockaddr *)&a, &al) < 0) return 3;
        SocketStruct *ss = rtp_socket("127.0.0.1", ntohs(a.sin_port), false);
        if (!ss) return 4;
        RtpPacket pkt;
        rtp_header(&pkt, 0, 0, 0, RTP_VESION, RTP_PAYLOAD_TYPE_PCMA, 1, 0, 0, 0);
        const uint32_t distSize = WMIX_INTERVAL_MS * freq / 1000 * chn * 16 / 8;
        const uint32_t srcSize = wmix_len_of_in(WMIX_CHN, WMIX_FREQ, chn, freq, distSize);
        unsigned char *dist = calloc(distSize * 2 + 64, 1);
        for (size_t off = 0; off + srcSize <= n; off += srcSize) {
            int ret = wmix_pcm_zoom(WMIX_CHN, WMIX_FREQ, in + off, srcSize, chn, freq, dist);
            ret = PCM2G711a((char *)dist, (char *)pkt.payload, ret, 0);
            pkt.rtpHeader.timestamp += ret / chn;
            ret = rtp_send(ss, &pkt, ret);
            if (ret < 0) return 5;
            unsigned char wire[5000];
            int got = -1;
            for (int tries = 0; tries < 1000 && got < 0; tries++) {
                got = recv(rx, wire, sizeof(wire), MSG_DONTWAIT);
                if (got < 0) usleep(1000);
            }
            if (got < 0) return 6;
            uint32_t g = (uint32_t)got;
            fwrite(&g, 4, 1, stdout);
            fwrite(wire, 1, g, stdout);
        }
        return 0;
    }
    if (!strcmp(argv[1], "pkgfifo") && argc == 3) {
        extern void playPkgBuff_add(uint8_t *pkgBuff);
        extern uint8_t *playPkgBuff_get(uint8_t *buff, int delayms);
        size_t n;
        unsigned char *in = slurp(&n);
        unsigned char buff[WMIX_PKG_SIZE];
        for (size_t off = 0; off + WMIX_PKG_SIZE <= n; off += WMIX_PKG_SIZE) {
            playPkgBuff_add(in + off);
            playPkgBuff_get(buff, atoi(argv[2]));
            fwrite(buff, 1, WMIX_PKG_SIZE, stdout);
        }
        return 0;
    }
    if (!strcmp(argv[1], "rtprecv") && argc == 2) {
        size_t n;
        unsigned char *in = slurp(&n);
        const int port = 20000 + (getpid() % 20000);
        SocketStruct *ss = rtp_socket("127.0.0.1", port, true);
        if (!ss) return 4;
        int tx = socket(AF_INET, SOCK_DGRAM, 0);
        struct sockaddr_in a;
        memset(&a, 0, sizeof(a));
        a.sin_family = AF_INET;
        a.sin_port = htons(port);
        a.sin_addr.s_addr = inet_addr("127.0.0.1");
        for (size_t off = 0; off + 4 <= n;) {
            uint32_t len;
            memcpy(&len, in + off, 4);
            off += 4;
            if (sendto(tx, in + off, len, 0, (struct sockaddr *)&a, sizeof(a)) < 0) return 5;
            off += len;
            RtpPacket pkt;
            uint32_t retSize = 0;
            int ret = -1;
            for (int tries = 0; tries < 1000 && ret <= 0; tries++) {
                ret = rtp_recv(ss, &pkt, &retSize);
                if (ret <= 0) usleep(1000);
            }
            if (ret <= 0) return 6;
            unsigned char buff[1024];
            memset(buff, 0, sizeof(buff));
            uint32_t pcm = (uint32_t)G711a2PCM((char *)pkt.payload, (char *)buff, retSize, 0);
            uint16_t seq = pkt.rtpHeader.seq;
            fwrite(&pcm, 4, 1, stdout);
            fwrite(buff, 1, pcm > sizeof(buff) ? sizeof(buff) : pcm, stdout);
            fwrite(&seq, 2, 1, stdout);
        }
        return 0;
    }
    if (!strcmp(argv[1], "tick") && argc == 9) {
        extern void playPkgBuff_add(uint8_t *pkgBuff);
        extern uint8_t *playPkgBuff_get(uint8_t *buff, int delayms);
        extern int ref_pin_generic_c(void); /* oracle/ref_shim.c in libwmixref.so: the generic-C AEC kernels */
        const int nsrc = atoi(argv[2]), sfreq = atoi(argv[3]), schn = atoi(argv[4]), nrec = atoi(argv[5]), nticks = atoi(argv[6]);
        const unsigned stages = (unsigned)atoi(argv[7]);
        const int agc_value = atoi(argv[8]);
        const uint32_t sbytes = (uint32_t)(sfreq / 1000 * WMIX_INTERVAL_MS * schn * 2);
        const int N = WMIX_FRAME_NUM * WMIX_CHN; /* int16 per package */
        if (ref_pin_generic_c() != 1) return 7;
        size_t n;
        unsigned char *in = slurp(&n);
        const size_t per_tick = (size_t)nsrc * sbytes + (size_t)nrec * WMIX_PKG_SIZE;
        if (n < per_tick * nticks) return 3;
        WMix_Struct *w = calloc(1, sizeof(WMix_Struct));
        w->buff = calloc(WMIX_BUFF_SIZE + 64, 1);
        w->start.U8 = w->head.U8 = w->tail.U8 = w->buff;
        w->end.U8 = w->buff + WMIX_BUFF_SIZE;
        w->run = true;
        w->reduceMode = 1;
        WMix_Point *heads = calloc(nsrc, sizeof(WMix_Point));
        uint32_t *ticks = calloc(nsrc, sizeof(uint32_t));
        void *ns_pa = (stages & 16) ? ns_init(WMIX_CHN, WMIX_FREQ, NULL) : NULL; /* webrtcEnable[WR_NS_PA], src/wmix.c:1370-1386 */
        if ((stages & 16) && !ns_pa) return 8;
        void **ns = calloc(nrec, sizeof(void *)), **aec = calloc(nrec, sizeof(void *)), **agc = calloc(nrec, sizeof(void *)),
             **vad = calloc(nrec, sizeof(void *));
        for (int r = 0; r < nrec; r++) {
            if (stages & 1) ns[r] = ns_init(WMIX_CHN, WMIX_FREQ, NULL);
            if (stages & 2) aec[r] = aec_init(WMIX_CHN, WMIX_FREQ, WMIX_INTERVAL_MS, NULL);
            if (stages & 4) agc[r] = agc_init(WMIX_CHN, WMIX_FREQ, WMIX_INTERVAL_MS, agc_value, NULL);
            if (stages & 8) vad[r] = vad_init(WMIX_CHN, WMIX_FREQ, WMIX_INTERVAL_MS, NULL);
            if (((stages & 1) && !ns[r]) || ((stages & 2) && !aec[r]) || ((stages & 4) && !agc[r]) || ((stages & 8) && !vad[r])) return 8;
        }
        unsigned char *srcpad = calloc(sbytes + 64, 1); /* the up-sampling fill reads one frame past the source (src/wmix.c:1857) */
        int16_t *farline = calloc((size_t)(nticks + 1) * N, sizeof(int16_t)); /* package t at (t + 1) * N: one package of silence in front */
        uint8_t playBuff[WMIX_PKG_SIZE], farBuff[WMIX_PKG_SIZE], buffSrc[WMIX_PKG_SIZE];
        uint8_t buffDist[2 * 8000 * WMIX_INTERVAL_MS / 1000 + 64];
        const int echo_delay = 40;
        WMix_Point rwTestSrc = {.U8 = 0}, rwTestHead = {.U8 = 0}; /* wmix->rwTest, src/wmix.c:531-532, 714-732 */
        uint32_t rwTick = 0;
        for (int t = 0; t < nticks; t++) {
            const unsigned char *p = in + (size_t)t * per_tick;
            for (int i = 0; i < nsrc; i++) { /* the task threads */
                memcpy(srcpad, p + (size_t)i * sbytes, sbytes);
                WMix_Point s = {.U8 = srcpad};
                heads[i] = wmix_load_data(w, s, sbytes, sfreq, schn, 16, heads[i], 1, &ticks[i]);
            }
            /* the play thread: one package */
            if (w->head.U8 >= w->end.U8) w->head.U8 = w->start.U8;
            WMix_Point dist = {.U8 = playBuff};
            for (uint32_t count = 0; count < WMIX_PKG_SIZE;) {
#if (WMIX_CHN == 1)
                *dist.U16++ = *w->head.U16;
                *w->head.U16++ = 0;
                w->tick += 2;
                count += 2;
#else
                *dist.U32++ = *w->head.U32;
                *w->head.U32++ = 0;
                w->tick += 4;
                count += 4;
#endif
                if (w->head.U8 >= w->end.U8) w->head.U8 = w->start.U8;
            }
            if (ns_pa) ns_process(ns_pa, (int16_t *)playBuff, (int16_t *)playBuff, WMIX_FRAME_NUM);
            playPkgBuff_add(playBuff);
            fwrite(playBuff, 1, WMIX_PKG_SIZE, stdout);
            playPkgBuff_get(farBuff, AEC_INTERVALMS);
            fwrite(farBuff, 1, WMIX_PKG_SIZE, stdout);
            memcpy(farline + (size_t)(t + 1) * N, farBuff, WMIX_PKG_SIZE);
            for (int r = 0; r < nrec; r++) { /* the record heartbeat of handle set r */
                const int16_t *local = (const int16_t *)(p + (size_t)nsrc * sbytes + (size_t)r * WMIX_PKG_SIZE);
                int16_t *near = (int16_t *)buffSrc;
                for (int i = 0; i < N; i++) {
                    int v = local[i] + (farline[(size_t)(t + 1) * N + i - echo_delay] >> 1);
                    near[i] = (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
                }
                if (ns[r]) ns_process(ns[r], near, near, WMIX_FRAME_NUM);
                if (aec[r] && aec_process2(aec[r], (int16_t *)farBuff, near, near, WMIX_FRAME_NUM, 0) != 0) return 9;
                if (agc[r] && agc_process(agc[r], near, near, WMIX_FRAME_NUM) != 0) return 10;
                if (vad[r]) vad_process(vad[r], near, WMIX_FRAME_NUM);
                if ((stages & 32) && r == 0) { /* the self send-receive test: what was recorded goes back into the play ring */
                    rwTestSrc.U8 = buffSrc;
                    rwTestHead = wmix_load_data(w, rwTestSrc, WMIX_PKG_SIZE, WMIX_FREQ, WMIX_CHN, WMIX_SAMPLE, rwTestHead, 1, &rwTick);
                }
                fwrite(buffSrc, 1, WMIX_PKG_SIZE, stdout);
                memset(buffDist, 0, sizeof(buffDist));
                wmix_pcm_zoom(WMIX_CHN, WMIX_FREQ, buffSrc, WMIX_PKG_SIZE, 1, 8000, buffDist);
                fwrite(buffDist, 1, 2 * 8000 * WMIX_INTERVAL_MS / 1000, stdout);
            }
        }
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
        unsigned char *out = calloc(n * 48 + 64, 1); /* 1 x 5000 -> 2 x 48000 writes 19.2 bytes per input byte */
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
