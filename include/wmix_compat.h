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

/* ---- src/wmix.h:40-49, 113-127 (resample + mix).  WMix_Point is src/wmixConf.h:149-156; WMix_Struct_Head
 * mirrors, field for field, the leading part of WMix_Struct (src/wmixConf.h:176-203) up to `reduceMode`, the
 * last field wmix_load_data reads -- a daemon passes its own WMix_Struct*. */
typedef union {
    int8_t *S8;
    uint8_t *U8;
    int16_t *S16;
    uint16_t *U16;
    int32_t *S32;
    uint32_t *U32;
} WMix_Point;
typedef struct {
    void *objAo, *objAi;
    uint8_t *buff;
    WMix_Point start, end;
    WMix_Point head, tail;
    bool run;
    uint8_t loopWord, loopWordRecord, loopWordFifo, loopWordRtp;
    uint32_t tick;
    uint32_t thread_sys, thread_record, thread_play;
    bool playRun, recordRun;
    int shmemRun;
    int msg_key; /* key_t in the reference (src/wmixConf.h:196): int on every Linux ABI; spelled int so that the header stands on
                  * its own under strict -std=c99, where <sys/types.h> hides key_t */
    int msg_fd;
    uint8_t reduceMode;
} WMix_Struct_Head;
WMix_Point wmix_load_data(WMix_Struct_Head *wmix, WMix_Point src, uint32_t srcU8Len, uint16_t freq, uint8_t channels,
                          uint8_t sample, WMix_Point head, uint8_t reduce, uint32_t *tick);
uint32_t wmix_len_of_out(uint8_t inChn, uint16_t inFreq, uint32_t inLen, uint8_t outChn, uint16_t outFreq);
uint32_t wmix_len_of_in(uint8_t inChn, uint16_t inFreq, uint8_t outChn, uint16_t outFreq, uint32_t outLen);
uint32_t wmix_pcm_zoom(uint8_t inChn, uint16_t inFreq, uint8_t *in, uint32_t inLen, uint8_t outChn, uint16_t outFreq,
                       uint8_t *out);

/* The same four functions under names that cannot collide with the daemon's own definitions.  The reference defines the
 * wmix.h group inside src/wmix.c, the translation unit of main(); a maintainer who links the unchanged daemon against
 * libwmix_amd.so weakens those four symbols in wmix.o (`objcopy --weaken-symbol`) and adds the object built from
 * wmix_amd/csrc/daemon_shim.c, whose strong definitions forward here (proven by
 * tools_dev/link_daemon.sh). */
WMix_Point wmx_compat_load_data(WMix_Struct_Head *wmix, WMix_Point src, uint32_t srcU8Len, uint16_t freq, uint8_t channels,
                                uint8_t sample, WMix_Point head, uint8_t reduce, uint32_t *tick);
uint32_t wmx_compat_len_of_out(uint8_t inChn, uint16_t inFreq, uint32_t inLen, uint8_t outChn, uint16_t outFreq);
uint32_t wmx_compat_len_of_in(uint8_t inChn, uint16_t inFreq, uint8_t outChn, uint16_t outFreq, uint32_t outLen);
uint32_t wmx_compat_pcm_zoom(uint8_t inChn, uint16_t inFreq, uint8_t *in, uint32_t inLen, uint8_t outChn, uint16_t outFreq,
                             uint8_t *out);

/* ------------------------------------------------------------------ math/fft.h:19-51 (stand-alone FFT helpers;
 * no callers inside the daemon, kept for link compatibility with tools that use them).  Host arrays. */
void FFT(float inReal[], float inImag[], float outReal[], float outImag[], float outAF[], float outPF[], unsigned int N);
void FFTR(float inReal[], float inImag[], float outReal[], float outImag[], float outAF[], float outPF[], unsigned int N);
void IFFT(float inReal[], float inImag[], float outReal[], float outImag[], unsigned int N);
void IFFTR(float inReal[], float inImag[], float outReal[], float outImag[], unsigned int N);
void fft_stream(float in[], unsigned int inLen, float stream[], unsigned int stLen, float outAF[], float outPF[]);

#ifdef __cplusplus
}
#endif
#endif /* WMIX_COMPAT_H */
