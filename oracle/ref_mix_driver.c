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
