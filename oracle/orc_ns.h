/* oracle/orc_ns.h -- TEST INFRASTRUCTURE ONLY. See orc_ns.c. */
#ifndef ORC_NS_H
#define ORC_NS_H
#include <stdint.h>
#include "orc_fft.h"

#define ORC_NS_MAXLEN 256
#define ORC_NS_MAXBINS 129

typedef struct {
    int fs, block_len, ana_len, magn_len;
    float window[ORC_NS_MAXLEN];
    orc_fft_t fft;
    /* sliding buffers (ns_core.h:60-62,112) */
    float analyze_buf[ORC_NS_MAXLEN], data_buf[ORC_NS_MAXLEN], synt_buf[ORC_NS_MAXLEN];
    float data_buf_hb[2][ORC_NS_MAXLEN];
    /* quantile noise estimator (ns_core.h:66-70) */
    float density[3 * ORC_NS_MAXBINS], lquantile[3 * ORC_NS_MAXBINS], quantile[ORC_NS_MAXBINS];
    int counter[3], updates;
    /* per-bin state (ns_core.h:72,86-93,96,101,104,110) */
    float smooth[ORC_NS_MAXBINS], noise[ORC_NS_MAXBINS], noise_prev[ORC_NS_MAXBINS];
    float magn_prev_analyze[ORC_NS_MAXBINS], magn_prev_process[ORC_NS_MAXBINS];
    float log_lrt_avg[ORC_NS_MAXBINS], magn_avg_pause[ORC_NS_MAXBINS], init_magn_est[ORC_NS_MAXBINS];
    float parametric_noise[ORC_NS_MAXBINS], speech_prob[ORC_NS_MAXBINS];
    /* scalars */
    float overdrive, denoise_bound;
    int gainmap, block_ind;
    int update_flag;      /* modelUpdatePars[0] */
    int window_countdown; /* modelUpdatePars[3] */
    float thr_lrt, thr_flat, thr_diff, w_lrt, w_flat, w_diff; /* priorModelPars[0,1,3,4,5,6] */
    float prior_speech_prob;
    float feat_flatness, feat_lrt, feat_diff, feat_energy_norm, feat_energy_acc; /* featureData[0,3,4,5,6] */
    float signal_energy, sum_magn, white_level, pink_num, pink_exp;
    int hist_lrt[1000], hist_flat[1000], hist_diff[1000];
} orc_ns_core;

typedef struct {
    orc_ns_core core;
    int chn, freq, pkg;
    float in[2][320], out[2][320];
} orc_ns;

void orc_ns_window(int ana_len, float *w);
void orc_ns_core_init(orc_ns_core *s, int fs);
void orc_ns_analyze(orc_ns_core *s, const float *frame);
void orc_ns_process(orc_ns_core *s, const float *const *in, int num_bands, float *const *out);
orc_ns *orc_ns_init(int chn, int freq);
void orc_ns_run(orc_ns *h, const int16_t *frame, int16_t *frame_out, int frame_num);
void orc_ns_release(orc_ns *h);
int orc_run_ns(int chn, int freq, const int16_t *in, int16_t *out, int frames_per_call, int n_calls);
#endif
