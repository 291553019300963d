/* oracle/orc_aecm.h -- TEST INFRASTRUCTURE ONLY.  State of the AECM restatement (orc_aecm.c): the live parts of AecMobile
 * (W:modules/audio_processing/aecm/echo_control_mobile.c:37-79), AecmCore (aecm_core.h:31-133) and the binary delay
 * estimator (W:modules/audio_processing/utility/delay_estimator.h:24-73, delay_estimator_internal.h:17-47). */
#ifndef ORC_AECM_H
#define ORC_AECM_H
#include <stdint.h>

#define ORC_AECM_MAX_DELAY 100

typedef struct {
    int16_t *data;
    int count, rd, wr, diff_wrap;
} orc_r16;

typedef struct {
    /* ---- AecMobile */
    int fs, known_delay, time_for_delay_change, ec_startup, check_buff_size, delay_change;
    short buf_size_start, counter, sum, first_val, check_buf_size_ctr, ms_in_snd, filt_delay, last_delay_diff;
    int16_t farend_old[2][80];
    orc_r16 farend;
    int16_t farend_store[50 * 80];
    /* ---- AecmCore */
    int first_vad, far_history_pos, far_q_domains[ORC_AECM_MAX_DELAY], current_vad;
    orc_r16 far_fr, near_fr, out_fr;
    int16_t far_fr_store[144], near_fr_store[144], out_fr_store[144];
    int16_t mult, nlp_flag, fixed_delay, cng_mode;
    uint32_t seed, tot_count;
    uint16_t far_history[65 * ORC_AECM_MAX_DELAY];
    int16_t dfa_clean_q, dfa_clean_q_old, dfa_noisy_q, dfa_noisy_q_old;
    int16_t near_log[64], far_log, echo_adapt_log[64], echo_stored_log[64];
    int16_t ch_stored[65], ch_adapt16[65];
    int32_t ch_adapt32[65];
    int16_t x_buf[128], d_buf[128], out_buf[64];
    int32_t echo_filt[65];
    int16_t near_filt[65];
    int32_t noise_est[65];
    int noise_low_ctr[65], noise_high_ctr[65];
    int16_t noise_est_ctr;
    int32_t mse_adapt_old, mse_stored_old, mse_threshold;
    int16_t far_energy_min, far_energy_max, far_energy_maxmin, far_energy_vad, far_energy_mse, vad_update_count, startup_state,
        mse_channel_count, sup_gain, sup_gain_old, sup_a, sup_d, sup_diff_ab, sup_diff_bd;
    /* ---- delay estimator, far and near halves */
    int32_t mean_far[65], mean_near[65], mean_bit_counts[ORC_AECM_MAX_DELAY + 1];
    int far_initialized, near_initialized, far_bit_counts[ORC_AECM_MAX_DELAY];
    uint32_t bin_far_hist[ORC_AECM_MAX_DELAY];
    int32_t minimum_probability;
    int last_delay_probability, last_delay;
    /* ---- wmix wrapper */
    int chn, pkg;
} orc_aecm;

orc_aecm *orc_aecm_init(int chn, int freq, int interval_ms);
int orc_aecm_buffer_farend(orc_aecm *a, const int16_t *far, int n);
int orc_aecm_process(orc_aecm *a, const int16_t *nearp, int16_t *out, int n, int ms);
int orc_aecm_run(orc_aecm *a, int mode, const int16_t *far, const int16_t *nearp, int16_t *out, int frame_num, int delay_ms);
void orc_aecm_release(orc_aecm *a);
int orc_run_aecm(int chn, int freq, int interval_ms, const int16_t *far, const int16_t *nearp, int16_t *out, int frames_per_call,
                 int n_calls, int delay_ms, int split);
#endif
