/* oracle/orc_mix.h -- TEST INFRASTRUCTURE ONLY. See orc_mix.c. */
#ifndef ORC_MIX_H
#define ORC_MIX_H
#include <stdint.h>

typedef struct { /* the fields of WMix_Struct (src/wmixConf.h:176-232) that wmix_load_data touches */
    int chn, freq;        /* WMIX_CHN, WMIX_FREQ */
    uint32_t size;        /* WMIX_BUFF_SIZE */
    uint8_t *buff;        /* start .. start+size */
    uint32_t head_off;    /* wmix->head - wmix->start */
    uint32_t tick;        /* wmix->tick */
    uint8_t reduce_mode;  /* wmix->reduceMode */
    uint32_t play_correct;/* VIEW_PLAY_CORRECT */
} orc_mix_ring;

uint32_t orc_len_of_out(uint8_t inChn, uint16_t inFreq, uint32_t inLen, uint8_t outChn, uint16_t outFreq);
uint32_t orc_len_of_in(uint8_t inChn, uint16_t inFreq, uint8_t outChn, uint16_t outFreq, uint32_t outLen);
uint32_t orc_pcm_zoom(uint8_t inChn, uint16_t inFreq, const uint8_t *in, uint32_t inLen, uint8_t outChn, uint16_t outFreq,
                      uint8_t *out);
void orc_mix_ring_init(orc_mix_ring *r, uint8_t *storage, int chn, int freq);
uint32_t orc_load_data(orc_mix_ring *r, const int16_t *src, uint32_t srcU8Len, uint16_t freq, uint8_t channels, uint8_t sample,
                       uint32_t head_off, uint8_t reduce, uint32_t *tick);
#endif
