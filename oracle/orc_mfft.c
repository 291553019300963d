/* oracle/orc_mfft.c -- TEST INFRASTRUCTURE ONLY (CPU checker; never on the product path).
 *
 * Restatement of the reference's stand-alone radix-2 FFT helpers, math/fft.c:
 *   FFT    math/fft.c:121-153     complex DIT FFT, + amplitude / phase curves
 *   FFTR   math/fft.c:156-253     real FFT through a half-size complex FFT
 *   IFFT   math/fft.c:299-316     complex inverse (every stage halves)
 *   IFFTR  math/fft.c:319-398     the "real inverse" exactly as the reference defines it
 *   fft_stream math/fft.c:413-424 FIFO of samples + FFT
 * (bit reversal :37-78, forward stages :81-118, inverse stages :256-296).
 *
 * Numeric contract that matters for parity: data are float; every twiddle is
 * cos/sin(2.0 * FFT_PI * p / N) evaluated in DOUBLE at each butterfly with FFT_PI = 3.1415926535897
 * (math/fft.c:21), the two products and their sum are double and are rounded to float once; the butterfly
 * add/sub are float; the inverse divides by 2 after every add/sub.  AF = sqrt(re^2 + im^2) / (N/2) with the
 * sum of squares in float and sqrt/divide in double; PF = atan2 in double.
 * Pinned against the real math/fft.c (oracle/_ref/libwmixref.so) in tests/test_mfft_oracle.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_FFT_PI 3.1415926535897

static unsigned ilog2u(unsigned n)
{
    unsigned m = 0;
    while ((1u << (m + 1)) <= n) m++;
    return m;
}

/* math/fft.c:37-78: swap element I with its M-bit mirror image J when I < J */
static void bit_reverse(float *re, float *im, unsigned n, unsigned m)
{
    for (unsigned i = 0; i < n; i++) {
        unsigned j = 0;
        for (unsigned b = 0; b < m; b++)
            if (i & (1u << b)) j |= 1u << (m - 1 - b);
        if (i < j) {
            float t = re[i];
            re[i] = re[j];
            re[j] = t;
            t = im[i];
            im[i] = im[j];
            im[j] = t;
        }
    }
}

/* math/fft.c:81-118 (inverse = 0) and :256-296 (inverse = 1) */
static void stages(float *re, float *im, unsigned n, unsigned m, int inverse)
{
    for (unsigned l = 1; l <= m; l++) {
        const int half = 1 << (l - 1), step = 1 << (m - l);
        for (int j = 0; j < half; j++) {
            const int p = j * step;
            const double c = cos(2.0 * ORC_FFT_PI * p / n), s = sin(2.0 * ORC_FFT_PI * p / n);
            for (int i = 0; i < step; i++) {
                const int r = j + 2 * half * i, q = r + half;
                float tr, ti;
                if (!inverse) {
                    tr = re[q] * c + im[q] * s;
                    ti = im[q] * c - re[q] * s;
                    re[q] = re[r] - tr;
                    im[q] = im[r] - ti;
                    re[r] = re[r] + tr;
                    im[r] = im[r] + ti;
                } else {
                    tr = re[q] * c - im[q] * s;
                    ti = im[q] * c + re[q] * s;
                    re[q] = (re[r] - tr) / 2;
                    im[q] = (im[r] - ti) / 2;
                    re[r] = (re[r] + tr) / 2;
                    im[r] = (im[r] + ti) / 2;
                }
            }
        }
    }
}

static void curves(const float *re, const float *im, float *af, float *pf, unsigned n)
{
    if (af)
        for (unsigned i = 0; i < n; i++) af[i] = sqrt(re[i] * re[i] + im[i] * im[i]) / (n / 2);
    if (pf)
        for (unsigned i = 0; i < n; i++) pf[i] = atan2(im[i], re[i]);
}

/* kind: 0 FFT, 1 FFTR, 2 IFFT, 3 IFFTR.  NULL in_re / in_im read as zeros; NULL outputs are skipped.
 * n must be a power of two >= 2.  IFFT/IFFTR ignore out_af / out_pf (the reference has no such outputs). */
int orc_mfft(int kind, const float *in_re, const float *in_im, float *out_re, float *out_im, float *out_af, float *out_pf,
             unsigned n)
{
    if (n < 2 || (n & (n - 1))) return -1;
    const unsigned m = ilog2u(n), h = n / 2;
    float *re = calloc(n, sizeof(float)), *im = calloc(n, sizeof(float));
    if (in_re) memcpy(re, in_re, n * sizeof(float));
    if (in_im) memcpy(im, in_im, n * sizeof(float));
    if (kind == 0 || kind == 2) {
        bit_reverse(re, im, n, m);
        stages(re, im, n, m, kind == 2);
    } else {
        /* the real variants transform y[i] = in[2i] + j in[2i+1] at half size (the imaginary input is unused) */
        const int inverse = kind == 3;
        float *yr = malloc(h * sizeof(float)), *yi = malloc(h * sizeof(float));
        float *x1r = malloc(h * sizeof(float)), *x1i = malloc(h * sizeof(float));
        float *x2r = malloc(h * sizeof(float)), *x2i = malloc(h * sizeof(float));
        for (unsigned i = 0; i < h; i++) {
            yr[i] = re[2 * i];
            yi[i] = re[2 * i + 1];
        }
        bit_reverse(yr, yi, h, m - 1);
        stages(yr, yi, h, m - 1, inverse);
        x1r[0] = yr[0];
        x1i[0] = yi[0];
        x2r[0] = yi[0];
        x2i[0] = -yr[0];
        for (unsigned k = 1; k < h; k++) {
            x1r[k] = (yr[k] + yr[h - k]) / 2;
            x1i[k] = (yi[k] - yi[h - k]) / 2;
            x2r[k] = (yi[k] + yi[h - k]) / 2;
            x2i[k] = (yr[h - k] - yr[k]) / 2;
        }
        for (unsigned j = 0; j < h; j++) {
            const double c = cos(2.0 * ORC_FFT_PI * (int)j / n), s = sin(2.0 * ORC_FFT_PI * (int)j / n);
            float tr, ti;
            if (!inverse) {
                tr = x2r[j] * c + x2i[j] * s;
                ti = x2i[j] * c - x2r[j] * s;
                re[j] = x1r[j] + tr;
                im[j] = x1i[j] + ti;
            } else {
                tr = x2r[j] * c - x2i[j] * s;
                ti = x2i[j] * c + x2r[j] * s;
                re[j] = (x1r[j] + tr) / 2;
                im[j] = (x1i[j] + ti) / 2;
            }
            if (j == 0) {
                re[h] = x1r[0] - x2r[0];
                im[h] = x1i[0] - x2i[0];
                if (inverse) {
                    re[h] = re[h] / 2;
                    im[h] = im[h] / 2;
                }
            } else {
                re[n - j] = re[j];
                im[n - j] = -im[j];
            }
        }
        free(yr), free(yi), free(x1r), free(x1i), free(x2r), free(x2i);
    }
    if (out_re) memcpy(out_re, re, n * sizeof(float));
    if (out_im) memcpy(out_im, im, n * sizeof(float));
    if (kind < 2) curves(re, im, out_af, out_pf, n);
    free(re), free(im);
    return 0;
}

/* math/fft.c:413-424 */
int orc_mfft_stream(const float *in, unsigned in_len, float *stream, unsigned st_len, float *out_af, float *out_pf)
{
    unsigned i, j;
    for (i = 0, j = in_len; i < in_len; i++, j++) stream[i] = stream[j];
    for (j = 0; i < st_len && j < in_len; i++, j++) stream[i] = in[j];
    return orc_mfft(0, stream, NULL, NULL, NULL, out_af, out_pf, st_len);
}
