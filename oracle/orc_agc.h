/* oracle/orc_agc.h -- TEST INFRASTRUCTURE ONLY. See orc_agc.c. */
#ifndef ORC_AGC_H
#define ORC_AGC_H
#include <stdint.h>

typedef struct { /* AgcVad, digital_agc.h:26-37 */
    int32_t down_state[8];
    int16_t hp_state, counter, log_ratio, mean_long;
    int32_t var_long;
    int16_t std_long, mean_short;
    int32_t var_short;
    int16_t std_short;
} orc_agc_vad;

typedef struct { /* DigitalAgc, digital_agc.h:39-53 (+ fs from LegacyAgc) */
    int fs;
    int32_t capacitor_slow, capacitor_fast, gain, gain_table[32];
    int16_t gate_prev;
    orc_agc_vad vad_near;
} orc_agc_core;

typedef struct { /* Agc_Struct, src/webrtc.c:667-677 */
    orc_agc_core core;
    int chn, freq, pkg;
} orc_agc;

int32_t orc_spl_sqrt(int32_t value);
int orc_agc_gain_table(int32_t *table, int16_t comp_gain_db, int16_t target_dbfs, int limiter, int16_t analog_target);
int orc_agc_core_init(orc_agc_core *s, int fs, int16_t comp_gain_db);
int orc_agc_core_set_gain(orc_agc_core *s, int16_t comp_gain_db);
int orc_agc_core_process(orc_agc_core *s, const int16_t *in, int16_t *out);
orc_agc *orc_agc_init(int chn, int freq, int interval_ms, int value);
int orc_agc_run(orc_agc *h, const int16_t *frame, int16_t *frame_out, int frame_num);
void orc_agc_addition(orc_agc *h, uint8_t value);
void orc_agc_release(orc_agc *h);
int orc_run_agc(int chn, int freq, int value, const int16_t *in, int16_t *out, int frames_per_call, int n_calls);
#endif
