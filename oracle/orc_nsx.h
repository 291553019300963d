/* oracle/orc_nsx.h -- TEST INFRASTRUCTURE ONLY.  State of the fixed-point noise suppressor restatement (orc_nsx.c);
 * field by field the live part of NoiseSuppressionFixedC, W:modules/audio_processing/ns/nsx_core.h:23-123. */
#ifndef ORC_NSX_H
#define ORC_NSX_H
#include <stdint.h>

#define ORC_NSX_ANA 256
#define ORC_NSX_BINS 129
#define ORC_NSX_HIST 1000

typedef struct {
    int fs, block, ana, ana2, nbins, stages;
    const int16_t *window, *factor2;
    int16_t ana_buf[ORC_NSX_ANA], syn_buf[ORC_NSX_ANA];
    uint16_t filt[ORC_NSX_BINS];            /* noiseSupFilter, Q14 */
    uint16_t overdrive, denoise_bound;
    int16_t lq[3 * ORC_NSX_BINS], dens[3 * ORC_NSX_BINS], counter[3], quant[ORC_NSX_BINS];
    int gain_map;
    int32_t max_lrt, min_lrt, lrt_avg[ORC_NSX_BINS], feat_lrt, thr_lrt;
    int16_t w_lrt, w_diff, w_flat;
    uint32_t feat_diff, thr_diff, feat_flat, thr_flat;
    int32_t pause[ORC_NSX_BINS];            /* avgMagnPause */
    uint32_t magn_energy, sum_magn, cur_avg_energy, time_avg_energy, time_avg_energy_tmp;
    uint32_t white, init_magn[ORC_NSX_BINS];
    int32_t pink_num, pink_exp;
    int min_norm, zero_input;
    uint32_t prev_noise[ORC_NSX_BINS];
    uint16_t prev_magn[ORC_NSX_BINS];
    int16_t prior_nonspeech;
    int block_index, model_update, cnt_thr;
    int16_t hist_lrt[ORC_NSX_HIST], hist_flat[ORC_NSX_HIST], hist_diff[ORC_NSX_HIST];
    int16_t hb[2][ORC_NSX_ANA];
    int q_noise, prev_q_noise, prev_q_magn;
    int16_t re[ORC_NSX_ANA], im[ORC_NSX_ANA]; /* frame-local in effect: written by the analysis, read by the synthesis */
    int32_t energy_in;
    int scale_energy_in, norm_data;
} orc_nsx_core;

typedef struct {
    orc_nsx_core core;
    int chn, freq, pkg;
    int16_t in[2][320], out[2][320];
} orc_nsx;

int orc_nsx_core_init(orc_nsx_core *s, int fs, int mode);
void orc_nsx_core_process(orc_nsx_core *s, const int16_t *const *in, int num_bands, int16_t *const *out);
orc_nsx *orc_nsx_init(int chn, int freq);
void orc_nsx_run(orc_nsx *h, const int16_t *frame, int16_t *frame_out, int frame_num);
void orc_nsx_release(orc_nsx *h);
int orc_run_nsx(int chn, int freq, const int16_t *in, int16_t *out, int frames_per_call, int n_calls);
void orc_spl_real_fft(int order, const int16_t *in, int16_t *out);
int orc_spl_real_ifft(int order, const int16_t *in, int16_t *out);
#endif
