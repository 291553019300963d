/* oracle/orc_g711.c -- TEST INFRASTRUCTURE ONLY (CPU checker; never shipped,
 * never on the product path).
 *
 * Restatement of the reference's G.711 companders.  Pinned against
 * oracle/_ref/libwmixref.so (the real src/g711codec.c) exhaustively over all
 * 65 536 int16 inputs / 256 codes in tests/test_g711.py, and against the
 * FNV-1a hashes of audio/1x8000.wav recorded in SURVEY.md section 8c
 * (tests/golden/g711_golden.json).
 *
 * The reference is the 16-bit-domain Sun g711.c variant: segment ends
 * 0xFF,0x1FF..0x7FFF applied to the raw 16-bit magnitude (src/g711codec.c:9-10).
 */
#include <stdint.h>

/* src/g711codec.c:12-22 `search`: index of the first segment end >= val, 8 if none. */
static int orc_segment(int val)
{
    int seg = 0, end = 0xFF;
    while (seg < 8 && val > end) {
        end = (end << 1) | 1;
        seg++;
    }
    return seg;
}

/* src/g711codec.c:82-114 */
unsigned char orc_linear2alaw(int pcm)
{
    int sign_mask = 0xD5;
    if (pcm < 0) {
        sign_mask = 0x55;
        pcm = -pcm - 8; /* may stay negative for -1..-7: reference behaviour (a(-1) = 0x5A) */
    }
    int seg = orc_segment(pcm);
    if (seg >= 8)
        return (unsigned char)(0x7F ^ sign_mask);
    int mant = (seg < 2) ? (pcm >> 4) : (pcm >> (seg + 3));
    return (unsigned char)(((seg << 4) | (mant & 0x0F)) ^ sign_mask);
}

/* src/g711codec.c:120-152 */
unsigned char orc_linear2ulaw(int pcm)
{
    int sign_mask;
    if (pcm < 0) {
        pcm = 0x84 - pcm;
        sign_mask = 0x7F;
    } else {
        pcm = pcm + 0x84;
        sign_mask = 0xFF;
    }
    int seg = orc_segment(pcm);
    if (seg >= 8)
        return (unsigned char)(0x7F ^ sign_mask);
    return (unsigned char)(((seg << 4) | ((pcm >> (seg + 3)) & 0x0F)) ^ sign_mask);
}

/* src/g711codec.c:28-51 */
int orc_alaw2linear(unsigned char code)
{
    unsigned a = code ^ 0x55u;
    int mag = (int)(a & 0x0F) << 4;
    int seg = (int)((a >> 4) & 7);
    if (seg == 0)
        mag += 8;
    else if (seg == 1)
        mag += 0x108;
    else
        mag = (mag + 0x108) << (seg - 1);
    return (a & 0x80) ? mag : -mag;
}

/* src/g711codec.c:62-76 */
int orc_ulaw2linear(unsigned char code)
{
    unsigned u = (unsigned char)~code;
    int mag = (((int)(u & 0x0F) << 3) + 0x84) << ((u >> 4) & 7);
    return (u & 0x80) ? (0x84 - mag) : (mag - 0x84);
}

/* src/g711codec.c:194-216 */
int orc_g711a_encode(unsigned char *out, const short *amp, int len)
{
    for (int i = 0; i < len; i++) out[i] = orc_linear2alaw(amp[i]);
    return len;
}
int orc_g711u_encode(unsigned char *out, const short *amp, int len)
{
    for (int i = 0; i < len; i++) out[i] = orc_linear2ulaw(amp[i]);
    return len;
}
/* src/g711codec.c:154-192: returns BYTES written */
int orc_g711a_decode(short *amp, const unsigned char *data, int bytes)
{
    int n = 0;
    for (int i = 0; i < bytes; i++) amp[n++] = (short)orc_alaw2linear(data[i]);
    return n * 2;
}
int orc_g711u_decode(short *amp, const unsigned char *data, int bytes)
{
    int n = 0;
    for (int i = 0; i < bytes; i++) amp[n++] = (short)orc_ulaw2linear(data[i]);
    return n * 2;
}

/* src/g711codec.c:227-308: -1 only when in, out and len are ALL null/0 */
#define ORC_ALL_NULL(i, o, n) (!(i) && !(o) && (n) == 0)
int orc_PCM2G711a(char *in, char *out, int len, int reserve)
{
    (void)reserve;
    if (ORC_ALL_NULL(in, out, len)) return -1;
    return orc_g711a_encode((unsigned char *)out, (const short *)in, len / 2);
}
int orc_PCM2G711u(char *in, char *out, int len, int reserve)
{
    (void)reserve;
    if (ORC_ALL_NULL(in, out, len)) return -1;
    return orc_g711u_encode((unsigned char *)out, (const short *)in, len / 2);
}
int orc_G711a2PCM(char *in, char *out, int len, int reserve)
{
    (void)reserve;
    if (ORC_ALL_NULL(in, out, len)) return -1;
    return orc_g711a_decode((short *)out, (const unsigned char *)in, len);
}
int orc_G711u2PCM(char *in, char *out, int len, int reserve)
{
    (void)reserve;
    if (ORC_ALL_NULL(in, out, len)) return -1;
    return orc_g711u_decode((short *)out, (const unsigned char *)in, len);
}
