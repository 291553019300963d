/* wmix_compat.h -- the reference's own C signatures, exported unchanged by
 * libwmix_amd.so (drop-in boundary).  A maintainer of wmix keeps
 * including src/webrtc.h, src/g711codec.h and src/wmix.h; this header exists so
 * that OUR tests and tools can bind the same symbols without the reference tree.
 *
 * HOST pointers, reference semantics (in==out aliasing allowed, int16
 * interleaved, frameNum counted in frames of `chn` samples).
 */
#ifndef WMIX_COMPAT_H
#define WMIX_COMPAT_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- src/g711codec.h:24-34 (+ linear2alaw/linear2ulaw, exported by src/g711codec.c:82,120) */
int PCM2G711a(char *InAudioData, char *OutAudioData, int DataLen, int reserve);
int PCM2G711u(char *InAudioData, char *OutAudioData, int DataLen, int reserve);
int G711a2PCM(char *InAudioData, char *OutAudioData, int DataLen, int reserve);
int G711u2PCM(char *InAudioData, char *OutAudioData, int DataLen, int reserve);
int g711a_decode(short amp[], const unsigned char g711a_data[], int g711a_bytes);
int g711u_decode(short amp[], const unsigned char g711u_data[], int g711u_bytes);
int g711a_encode(unsigned char g711_data[], const short amp[], int len);
int g711u_encode(unsigned char g711_data[], const short amp[], int len);
unsigned char linear2alaw(int pcm_val);
unsigned char linear2ulaw(int pcm_val);

/* ---- src/webrtc.h:40-45 (AEC) */
void *aec_init(int chn, int freq, int intervalMs, bool *debug);
int aec_setFrameFar(void *fp, int16_t *frameFar, int frameNum);
int aec_process(void *fp, int16_t *frameNear, int16_t *frameOut, int frameNum, int delayms);
int aec_process2(void *fp, int16_t *frameFar, int16_t *frameNear, int16_t *frameOut, int frameNum, int delayms);
void aec_release(void *fp);

/* ---- src/webrtc.h:47-51 (NS) */
void *ns_init(int chn, int freq, bool *debug);
void ns_process(void *fp, int16_t *frame, int16_t *frameOut, int frameNum);
void ns_release(void *fp);

/* ---- src/webrtc.h:32-36 (VAD) */
void *vad_init(int chn, int freq, int intervalMs, bool *debug);
void vad_process(void *fp, int16_t *frame, int frameNum);
void vad_release(void *fp);

/* ---- src/webrtc.h:55-60 (AGC) */
void *agc_init(int chn, int freq, int intervalMs, int value, bool *debug);
int agc_process(void *fp, int16_t *frame, int16_t *frameOut, int frameNum);
void agc_addition(void *fp, uint8_t value);
void agc_release(void *fp);

#ifdef __cplusplus
}
#endif
#endif /* WMIX_COMPAT_H */
