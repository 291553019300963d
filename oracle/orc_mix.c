/* oracle/orc_mix.c -- TEST INFRASTRUCTURE ONLY (CPU checker; never on the product path).
 *
 * Restatement of wmix's resample + mix arithmetic:
 *   wmix_len_of_out / wmix_len_of_in / wmix_pcm_zoom      src/wmix.c:49-222
 *   volumeAdd, wmix_load_data                              src/wmix.c:1617-1957
 * The reference fixes the ring format at compile time (WMIX_CHN / WMIX_FREQ from platform/alsa/plat.h:48-50,
 * default 1 x 8000 Hz x 16 bit, 1 s ring); here it is a run-time field of orc_mix_ring.  Pinned against the
 * real functions (oracle/_ref/ref_mix_driver over libwmixref_mix.so, ring 1 x 8000) in tests/test_mix_oracle.py.
 * The float32 phase accumulators are kept exactly (`-= 1.0` is a double subtraction rounded back to float).
 */
#include <stdint.h>
#include <string.h>
#include "orc_mix.h"

/* src/wmix.c:49-91 */
uint32_t orc_len_of_out(uint8_t inChn, uint16_t inFreq, uint32_t inLen, uint8_t outChn, uint16_t outFreq)
{
    uint32_t inCount = 0, outCount = 0;
    float div, divStep = 0;
    if (inFreq == outFreq && inChn == outChn) return inLen;
    if (inFreq < outFreq) {
        div = (float)inFreq / outFreq;
        while (inCount < inLen) {
            outCount += outChn;
            divStep += div;
            if ((int)divStep > 0) {
                inCount += inChn;
                divStep -= 1.0;
            }
        }
    } else {
        div = (float)outFreq / inFreq;
        while (inCount < inLen) {
            divStep += div;
            if ((int)divStep > 0) {
                outCount += outChn;
                divStep -= 1.0;
            }
            inCount += inChn;
        }
    }
    return outCount;
}

/* src/wmix.c:94-136 */
uint32_t orc_len_of_in(uint8_t inChn, uint16_t inFreq, uint8_t outChn, uint16_t outFreq, uint32_t outLen)
{
    uint32_t inCount = 0, outCount = 0;
    float div, divStep = 0;
    if (inFreq == outFreq && inChn == outChn) return outLen;
    if (inFreq < outFreq) {
        div = (float)inFreq / outFreq;
        while (outCount < outLen) {
            outCount += outChn;
            divStep += div;
            if ((int)divStep > 0) {
                inCount += inChn;
                divStep -= 1.0;
            }
        }
    } else {
        div = (float)outFreq / inFreq;
        while (outCount < outLen) {
            divStep += div;
            if ((int)divStep > 0) {
                outCount += outChn;
                divStep -= 1.0;
            }
            inCount += inChn;
        }
    }
    return inCount;
}

/* src/wmix.c:139-222.  The 2ch->2ch branch tests chnMode == 0x12 a second time and is dead, so a
 * 2ch->2ch rate change writes nothing. */
uint32_t orc_pcm_zoom(uint8_t inChn, uint16_t inFreq, const uint8_t *in, uint32_t inLen, uint8_t outChn, uint16_t outFreq,
                      uint8_t *out)
{
    const int16_t *ip = (const int16_t *)in, *iend = (const int16_t *)(in + inLen);
    int16_t *op = (int16_t *)out;
    float div, divStep = 0;
    uint8_t mode = (uint8_t)((inChn << 4) | (outChn & 0x0F));
    if (inFreq == outFreq && inChn == outChn) {
        memcpy(out, in, inLen);
        return inLen;
    }
    if (inFreq < outFreq) {
        div = (float)inFreq / outFreq;
        while (ip < iend) {
            if (mode == 0x11 || mode == 0x21) {
                *op++ = *ip;
            } else if (mode == 0x12) {
                *op++ = *ip;
                *op++ = *ip;
            }
            divStep += div;
            if ((int)divStep > 0) {
                ip += inChn;
                divStep -= 1.0;
            }
        }
    } else {
        div = (float)outFreq / inFreq;
        while (ip < iend) {
            divStep += div;
            if ((int)divStep > 0) {
                if (mode == 0x11 || mode == 0x21) {
                    *op++ = *ip;
                } else if (mode == 0x12) {
                    *op++ = *ip;
                    *op++ = *ip;
                }
                divStep -= 1.0;
            }
            ip += inChn;
        }
    }
    return (uint32_t)((uint8_t *)op - out);
}

/* src/wmix.c:1617-1636 */
static int16_t volume_add(int16_t a, int16_t b)
{
    if (a == 0) return b;
    if (b == 0) return a;
    int32_t s = (int32_t)a + b;
    return (int16_t)(s < -32768 ? -32768 : (s > 32767 ? 32767 : s));
}

void orc_mix_ring_init(orc_mix_ring *r, uint8_t *storage, int chn, int freq)
{
    memset(r, 0, sizeof(*r));
    r->chn = chn;
    r->freq = freq;
    r->size = (uint32_t)(chn * 2) * (uint32_t)freq; /* WMIX_BUFF_SIZE: 1 s */
    r->buff = storage;
    memset(storage, 0, r->size);
    r->reduce_mode = 1;
    r->play_correct = (uint32_t)(chn * freq * 16 / 8 / 5); /* PLAT_PLAY_CORRECT, plat.h:54 */
}

/* src/wmix.c:1639-1957.  Offsets are bytes into the ring; head_off == UINT32_MAX plays the role of the
 * reference's NULL head.  Returns the new head offset; *tick is updated. */
uint32_t orc_load_data(orc_mix_ring *r, const int16_t *src, uint32_t srcU8Len, uint16_t freq, uint8_t channels, uint8_t sample,
                       uint32_t head_off, uint8_t reduce, uint32_t *tick)
{
    if (!r || !src || srcU8Len < 1) return head_off;
    uint32_t tickAdd = 0, count;
    int rdce = (reduce == r->reduce_mode) ? 1 : r->reduce_mode;
    int32_t freqErr = r->freq - freq;
    float divCount, divPow;
    int16_t repair[64];
    int repairCount = 0;
    const int16_t *ps = src;
    if (head_off == UINT32_MAX || *tick < r->tick) {
        head_off = r->head_off + r->play_correct;
        *tick = r->tick + r->play_correct;
        if (head_off >= r->size) head_off = 0;
    }
    int16_t *ph = (int16_t *)(r->buff + head_off), *pend = (int16_t *)(r->buff + r->size), *pstart = (int16_t *)r->buff;
#define PUT(v)                                     \
    do {                                           \
        *ph = volume_add(*ph, (int16_t)((v) / rdce)); \
        ph++;                                      \
        tickAdd += 2;                              \
    } while (0)
    if (freq == r->freq && channels == r->chn && sample == 16) {
        for (count = 0; count < srcU8Len;) {
            PUT(*ps);
            ps++;
            count += 2;
            if (r->chn != 1) {
                PUT(*ps);
                ps++;
                count += 2;
            }
            if (ph >= pend) ph = pstart;
        }
    } else if (sample == 16 && (channels == 1 || channels == 2)) {
        const int step = channels; /* int16 per source frame */
        if (freqErr < 0) {
            divPow = (float)(-freqErr) / r->freq;
            for (count = 0, divCount = 0; count < srcU8Len;) {
                if (divCount >= 1.0) {
                    ps += step;
                    divCount -= 1.0;
                    count += 2 * step;
                } else {
                    PUT(ps[0]);
                    if (r->chn != 1) PUT(channels == 2 ? ps[1] : ps[0]);
                    ps += step;
                    divCount += divPow;
                    count += 2 * step;
                }
                if (ph >= pend) ph = pstart;
            }
        } else {
            divPow = (float)freqErr / freq;
            for (count = 0, divCount = 0; count < srcU8Len;) {
                if (divCount >= 1.0) {
                    PUT(repair[repairCount]);
                    if (r->chn != 1) PUT(repair[repairCount]);
                    divCount -= 1.0;
                    repairCount += 1;
                } else {
                    PUT(ps[0]);
                    if (r->chn != 1) PUT(channels == 2 ? ps[1] : ps[0]);
                    ps += step;
                    divCount += divPow;
                    count += 2 * step;
                    if (divCount >= 1.0) {
                        int n2 = (int)divCount + 1;
                        int16_t prev = *(ps - step);
                        float st = (float)((*ps) - prev) / n2, sum = st;
                        for (repairCount = 0; repairCount < n2;) {
                            repair[repairCount] = (int16_t)(prev + sum);
                            repairCount += 1;
                            sum += st;
                        }
                        repairCount = 0;
                    }
                }
                if (ph >= pend) ph = pstart;
            }
        }
    }
#undef PUT
    uint32_t new_head = (uint32_t)((uint8_t *)ph - r->buff);
    if (*tick < r->tick) {
        new_head = r->head_off + tickAdd;
        tickAdd += r->tick;
        if (new_head >= r->size) new_head -= r->size;
    } else {
        tickAdd += *tick;
    }
    *tick = tickAdd;
    return new_head;
}
