/* oracle/orc_vad.h -- TEST INFRASTRUCTURE ONLY. See orc_vad.c. */
#ifndef ORC_VAD_H
#define ORC_VAD_H
#include <stdint.h>

typedef struct { /* VadInstT, vad_core.h:27-56 (mode-3 thresholds are constants in orc_vad.c) */
    int32_t ds_state[4];
    int16_t noise_means[12], speech_means[12], noise_stds[12], speech_stds[12];
    int32_t frame_counter;
    int16_t over_hang, num_of_speech;
    int16_t index_vector[96], low_value_vector[96];
    int16_t mean_value[6], upper_state[5], lower_state[5], hp_filter_state[4];
} orc_vad_core;

typedef struct { /* Vad_Struct, src/webrtc.c:18-27 */
    orc_vad_core core;
    int chn, freq, interval_ms, pkg, reduce;
} orc_vad;

int orc_norm_w32(int32_t a);
int orc_norm_u32(uint32_t a);
int32_t orc_div_w32_w16(int32_t num, int16_t den);
void orc_vad_downsample(const int16_t *in, int16_t *out, int32_t *st, int in_len);
void orc_vad_core_init(orc_vad_core *s);
int orc_vad_core_process(orc_vad_core *s, int fs, const int16_t *frame, int frame_len);
int16_t orc_vad_features(orc_vad_core *s, const int16_t *in, int len, int16_t *f);
int32_t orc_vad_gauss(int16_t input, int16_t mean, int16_t std, int16_t *delta);
int16_t orc_vad_find_min(orc_vad_core *s, int16_t v, int ch);
orc_vad *orc_vad_init(int chn, int freq, int interval_ms);
void orc_vad_run(orc_vad *h, int16_t *frame, int frame_num);
void orc_vad_release(orc_vad *h);
int orc_run_vad(int chn, int freq, int interval_ms, const int16_t *in, int16_t *out, int frames_per_call, int n_calls);
#endif
