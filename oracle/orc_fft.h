/* oracle/orc_fft.h -- TEST INFRASTRUCTURE ONLY. See orc_fft.c. */
#ifndef ORC_FFT_H
#define ORC_FFT_H
typedef struct {
    int n, nw, nc;
    float w[128]; /* w[0..nw) twiddles, w[nw..nw+nc) split cosines; n <= 256 */
    float w2;     /* w[2]: the block-1 special twiddle */
    float W1[32][2], W2[32][2], W3[32][2]; /* per radix-4 block index b >= 2 */
} orc_fft_t;
void orc_fft_init_aec128(orc_fft_t *f);
const orc_fft_t *orc_fft_aec128(void);
void orc_aec_rdft(int isgn, float *a);
void orc_fft_init(orc_fft_t *f, int n);
void orc_rdft_forward(const orc_fft_t *f, float *a);
void orc_rdft_inverse(const orc_fft_t *f, float *a);
void orc_rdft(int n, int isgn, float *a);
const float *orc_fft_tables(int n);
#endif
