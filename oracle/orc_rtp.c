/* oracle/orc_rtp.c -- TEST INFRASTRUCTURE ONLY (CPU checker; never on the product path).
 *
 * Restatement of the reference's RTP / G.711 packet edge:
 *   egress  = the loop body of wmix_thread_rtp_send_pcma, src/wmixTask.c:1124-1143:
 *             wmix_pcm_zoom -> PCM2G711a -> header.timestamp += codes / chn -> rtp_send -> header.seq++
 *             with the header as rtp_header() fills it (src/rtp.c:20-33, src/wmixTask.c:1058: v = 2, m = 1,
 *             pt = 8, seq = timestamp = ssrc = 0) and rtp_send's byte order (src/rtp.c:42-44); wire layout
 *             src/rtp.h:37-75.
 *   ingest  = rtp_recv's payload-length rule (src/rtp.c:86-95: 160 bytes for PCMA / PCMU, 0 for anything that is
 *             not AAC) followed by G711a2PCM (src/wmixTask.c:1278-1282).  The receive path does not byte-swap the
 *             header back (rtp_recv has no ntohs), so `seq` is reported as stored.
 * Pinned against the real functions (oracle/_ref/ref_mix_driver rtpsend / rtprecv over UDP loopback) in
 * tests/test_rtp_oracle.py.  mu-law egress (pt 0, PCM2G711u) is our extension of the same loop.
 */
#include <stdint.h>
#include <string.h>
#include "orc_mix.h"

int orc_PCM2G711a(char *in, char *out, int len, int reserve);
int orc_PCM2G711u(char *in, char *out, int len, int reserve);
int orc_G711a2PCM(char *in, char *out, int len, int reserve);

typedef struct {
    uint16_t seq;
    uint32_t timestamp, ssrc;
    uint8_t pt;
} orc_rtp_sender;

void orc_rtp_sender_init(orc_rtp_sender *s, int law)
{
    memset(s, 0, sizeof(*s));
    s->pt = law == 0 ? 8 : 0;
}

/* returns the datagram size; `packet` needs 12 + in_bytes bytes */
int orc_rtp_egress(orc_rtp_sender *s, int in_chn, int in_freq, const uint8_t *pcm, uint32_t in_bytes, int out_chn, int out_freq,
                   uint8_t *packet)
{
    uint8_t dist[16384];
    int n = (int)orc_pcm_zoom((uint8_t)in_chn, (uint16_t)in_freq, pcm, in_bytes, (uint8_t)out_chn, (uint16_t)out_freq, dist);
    n = s->pt == 8 ? orc_PCM2G711a((char *)dist, (char *)packet + 12, n, 0) : orc_PCM2G711u((char *)dist, (char *)packet + 12, n, 0);
    s->timestamp += (uint32_t)(n / out_chn);
    packet[0] = 2u << 6;                  /* v = 2, p = x = 0, cc = 0 */
    packet[1] = (uint8_t)(0x80 | s->pt);  /* m = 1 */
    packet[2] = (uint8_t)(s->seq >> 8);
    packet[3] = (uint8_t)s->seq;
    for (int i = 0; i < 4; i++) {
        packet[4 + i] = (uint8_t)(s->timestamp >> (24 - 8 * i));
        packet[8 + i] = (uint8_t)(s->ssrc >> (24 - 8 * i));
    }
    s->seq++;
    return n + 12;
}

/* returns the number of PCM bytes written (320 or 0); *seq_raw = header bytes 2..3 read as a native uint16 */
int orc_rtp_ingest(const uint8_t *packet, uint8_t *pcm, uint16_t *seq_raw)
{
    const int pt = packet[1] & 0x7F;
    int size = 0;
    if (pt == 97) /* AAC carries its size in the payload; not a G.711 packet -- nothing to decode here */
        size = 0;
    else if (pt == 8 || pt == 0)
        size = 160;
    if (seq_raw) memcpy(seq_raw, packet + 2, 2);
    return orc_G711a2PCM((char *)packet + 12, (char *)pcm, size, 0);
}
