/* oracle/orc_fft.c -- TEST INFRASTRUCTURE ONLY (CPU checker).
 *
 * Restatement of the split-radix-4 real FFT the reference's float paths use:
 *   - NS:  WebRtc_rdft(n = 128 | 256) from the vendored common_audio/fft4g.c
 *          (W:common_audio/fft4g.c:324-361 rdft, :642-690 makewt/makect, :693-790
 *          bitrv2, :902-999 cftfsub/cftbsub, :1002-1104 cft1st, :1107-1231 cftmdl,
 *          :1234-1284 rftfsub/rftbsub);
 *   - AEC: aec_rdft_forward_128 / aec_rdft_inverse_128 (W:modules/audio_processing/
 *          aec/aec_rdft.c:126-563), the same algorithm specialised to n = 128 with
 *          the twiddles frozen in tables (aec_rdft.c:32-122).
 *
 * It is written over COMPLEX indices with one butterfly routine per twiddle class
 * instead of the reference's unrolled float-index loops, but every output is
 * produced by the same sequence of float operations, so results are bit-identical
 * (tests/test_fft_oracle.py checks that against oracle/_ref for both sizes, both
 * directions).  Compile with -ffp-contract=off.
 *
 * Data layout = the reference's: a[2q], a[2q+1] = re, im of complex point q;
 * after the forward transform a[0] = DC, a[1] = Nyquist (both real).
 */
#include <math.h>
#include <string.h>
#include "orc_fft.h"

static int bit_reverse(int q, int bits)
{
    int r = 0;
    for (int b = 0; b < bits; b++)
        if (q & (1 << b)) r |= 1 << (bits - 1 - b);
    return r;
}

/* fft4g.c:693-790: in-place bit-reversal permutation of n/2 complex points. */
static void permute_bitrev(int n, float *a)
{
    int nc = n >> 1, bits = 0;
    while ((1 << bits) < nc) bits++;
    for (int q = 0; q < nc; q++) {
        int r = bit_reverse(q, bits);
        if (r > q) {
            float tr = a[2 * q], ti = a[2 * q + 1];
            a[2 * q] = a[2 * r];
            a[2 * q + 1] = a[2 * r + 1];
            a[2 * r] = tr;
            a[2 * r + 1] = ti;
        }
    }
}

/* Expand the packed twiddle table w[] into one (W1, W2, W3) triple per radix-4 block
 * index b >= 2, exactly as cft1st/cftmdl derive them (fft4g.c:1050-1056,1079-1082 /
 * aec_rdft.c:247-254,277-280; aec's rdft_wk3ri_first/second tables hold the same
 * float expressions evaluated on rdft_w, checked in tests/test_fft_oracle.py). */
static void expand_block_twiddles(orc_fft_t *f)
{
    const float *w = f->w;
    int nblk = f->n >> 3; /* complex points / 4 = blocks of the first pass */
    f->w2 = w[2];
    for (int b = 2; b < nblk; b++) {
        int k1 = 2 * (b >> 1), k2 = 2 * k1;
        float wk2r = w[k1], wk2i = w[k1 + 1], wk1r, wk1i;
        if ((b & 1) == 0) {
            wk1r = w[k2];
            wk1i = w[k2 + 1];
            f->W3[b][0] = wk1r - 2 * wk2i * wk1i;
            f->W3[b][1] = 2 * wk2i * wk1r - wk1i;
            f->W2[b][0] = wk2r;
            f->W2[b][1] = wk2i;
        } else {
            wk1r = w[k2 + 2];
            wk1i = w[k2 + 3];
            f->W3[b][0] = wk1r - 2 * wk2r * wk1i;
            f->W3[b][1] = 2 * wk2r * wk1r - wk1i;
            f->W2[b][0] = -wk2i;
            f->W2[b][1] = wk2r;
        }
        f->W1[b][0] = wk1r;
        f->W1[b][1] = wk1i;
    }
}

/* fft4g.c:642-690 (makewt, makect): w[0..nw) twiddles (bit-reversed pairs), then the
 * nc split cosines/sines. */
void orc_fft_init(orc_fft_t *f, int n)
{
    memset(f, 0, sizeof(*f));
    f->n = n;
    int nw = n >> 2, nc = n >> 2;
    f->nw = nw;
    f->nc = nc;
    float *w = f->w, *c = f->w + nw;
    {
        int nwh = nw >> 1;
        float delta = (float)atan(1.0f) / nwh;
        w[0] = 1;
        w[1] = 0;
        w[nwh] = (float)cos(delta * nwh);
        w[nwh + 1] = w[nwh];
        for (int j = 2; j < nwh; j += 2) {
            float x = (float)cos(delta * j), y = (float)sin(delta * j);
            w[j] = x;
            w[j + 1] = y;
            w[nw - j] = y;
            w[nw - j + 1] = x;
        }
        if (nwh > 2) permute_bitrev(nw, w);
    }
    {
        int nch = nc >> 1;
        float delta = (float)atan(1.0f) / nch;
        c[0] = (float)cos(delta * nch);
        c[nch] = 0.5f * c[0];
        for (int j = 1; j < nch; j++) {
            c[j] = 0.5f * (float)cos(delta * j);
            c[nc - j] = 0.5f * (float)sin(delta * j);
        }
    }
    expand_block_twiddles(f);
}

/* aec_rdft.c:32-49: the AEC's n = 128 transform uses a FROZEN table rdft_w[64] whose
 * entries differ from what makewt/makect(128) produce with today's libm by 1 ulp in
 * 8 places (indices 4,7,20,27,40,41,42,47), so it is carried as data (IEEE-754 bit
 * patterns of the 64 floats). */
static const unsigned kAecRdftW[64] = {
    0x3f800000, 0x00000000, 0x3f3504f3, 0x3f3504f3, 0x3f6c835f, 0x3ec3ef16, 0x3ec3ef16, 0x3f6c835f,
    0x3f7b14be, 0x3e47c5c2, 0x3f0e39da, 0x3f54db31, 0x3f54db31, 0x3f0e39da, 0x3e47c5c2, 0x3f7b14be,
    0x3f7ec46d, 0x3dc8bd36, 0x3f22679a, 0x3f45e403, 0x3f61c598, 0x3ef15aea, 0x3e94a031, 0x3f74fa0b,
    0x3f74fa0b, 0x3e94a031, 0x3ef15aea, 0x3f61c598, 0x3f45e403, 0x3f22679a, 0x3dc8bd36, 0x3f7ec46d,
    0x3f3504f3, 0x3effb10f, 0x3efec46d, 0x3efd3aac, 0x3efb14be, 0x3ef853f8, 0x3ef4fa0b, 0x3ef10908,
    0x3eec835f, 0x3ee76bd7, 0x3ee1c598, 0x3edb941a, 0x3ed4db31, 0x3ecd9f02, 0x3ec5e403, 0x3ebdaefa,
    0x3eb504f3, 0x3eabeb4a, 0x3ea2679a, 0x3e987fc0, 0x3e8e39da, 0x3e839c3d, 0x3e715aea, 0x3e5ae880,
    0x3e43ef16, 0x3e2c7cd4, 0x3e14a031, 0x3df8cfcd, 0x3dc7c5c2, 0x3d964083, 0x3d48bd36, 0x3cc8fb30,
};

void orc_fft_init_aec128(orc_fft_t *f)
{
    memset(f, 0, sizeof(*f));
    f->n = 128;
    f->nw = 32;
    f->nc = 32;
    memcpy(f->w, kAecRdftW, sizeof(kAecRdftW));
    expand_block_twiddles(f);
}

typedef struct { float r, i; } cpx;

/* One radix-4 butterfly on complex points p0..p3 (stride hc apart), twiddle class by
 * block index b (fft4g.c:1002-1231): 0 none, 1 the w[2] special, >=2 general. */
static void bfly4(float *a, int p0, int hc, int b, const orc_fft_t *f)
{
    int p1 = p0 + hc, p2 = p1 + hc, p3 = p2 + hc;
    cpx A = {a[2 * p0], a[2 * p0 + 1]}, B = {a[2 * p1], a[2 * p1 + 1]};
    cpx C = {a[2 * p2], a[2 * p2 + 1]}, D = {a[2 * p3], a[2 * p3 + 1]};
    float x0r = A.r + B.r, x0i = A.i + B.i, x1r = A.r - B.r, x1i = A.i - B.i;
    float x2r = C.r + D.r, x2i = C.i + D.i, x3r = C.r - D.r, x3i = C.i - D.i;
    float tr, ti;
    a[2 * p0] = x0r + x2r;
    a[2 * p0 + 1] = x0i + x2i;
    if (b == 0) {
        a[2 * p2] = x0r - x2r;
        a[2 * p2 + 1] = x0i - x2i;
        a[2 * p1] = x1r - x3i;
        a[2 * p1 + 1] = x1i + x3r;
        a[2 * p3] = x1r + x3i;
        a[2 * p3 + 1] = x1i - x3r;
    } else if (b == 1) {
        float wk1r = f->w2;
        a[2 * p2] = x2i - x0i;
        a[2 * p2 + 1] = x0r - x2r;
        tr = x1r - x3i;
        ti = x1i + x3r;
        a[2 * p1] = wk1r * (tr - ti);
        a[2 * p1 + 1] = wk1r * (tr + ti);
        tr = x3i + x1r;
        ti = x3r - x1i;
        a[2 * p3] = wk1r * (ti - tr);
        a[2 * p3 + 1] = wk1r * (ti + tr);
    } else {
        const float *W1 = f->W1[b], *W2 = f->W2[b], *W3 = f->W3[b];
        tr = x0r - x2r;
        ti = x0i - x2i;
        a[2 * p2] = W2[0] * tr - W2[1] * ti;
        a[2 * p2 + 1] = W2[0] * ti + W2[1] * tr;
        tr = x1r - x3i;
        ti = x1i + x3r;
        a[2 * p1] = W1[0] * tr - W1[1] * ti;
        a[2 * p1 + 1] = W1[0] * ti + W1[1] * tr;
        tr = x1r + x3i;
        ti = x1i - x3r;
        a[2 * p3] = W3[0] * tr - W3[1] * ti;
        a[2 * p3 + 1] = W3[0] * ti + W3[1] * tr;
    }
}

/* fft4g.c:902-999: twiddled radix-4 passes, then an untwiddled radix-4 or radix-2
 * closing pass; `inverse` selects cftbsub's conjugating closing pass. */
static void complex_passes(const orc_fft_t *f, float *a, int inverse)
{
    int nc = f->n >> 1; /* complex points */
    int hc = 1;         /* butterfly stride in complex points */
    while (hc * 4 < nc) {
        int blocks = nc / (4 * hc);
        for (int b = 0; b < blocks; b++)
            for (int j = 0; j < hc; j++) bfly4(a, b * 4 * hc + j, hc, b, f);
        hc *= 4;
    }
    if (hc * 4 == nc) {
        for (int j = 0; j < hc; j++) {
            int p0 = j, p1 = p0 + hc, p2 = p1 + hc, p3 = p2 + hc;
            float Ar = a[2 * p0], Ai = a[2 * p0 + 1], Br = a[2 * p1], Bi = a[2 * p1 + 1];
            float Cr = a[2 * p2], Ci = a[2 * p2 + 1], Dr = a[2 * p3], Di = a[2 * p3 + 1];
            float x0r = Ar + Br, x1r = Ar - Br, x2r = Cr + Dr, x2i = Ci + Di, x3r = Cr - Dr, x3i = Ci - Di;
            if (!inverse) {
                float x0i = Ai + Bi, x1i = Ai - Bi;
                a[2 * p0] = x0r + x2r;
                a[2 * p0 + 1] = x0i + x2i;
                a[2 * p2] = x0r - x2r;
                a[2 * p2 + 1] = x0i - x2i;
                a[2 * p1] = x1r - x3i;
                a[2 * p1 + 1] = x1i + x3r;
                a[2 * p3] = x1r + x3i;
                a[2 * p3 + 1] = x1i - x3r;
            } else {
                float x0i = -Ai - Bi, x1i = -Ai + Bi;
                a[2 * p0] = x0r + x2r;
                a[2 * p0 + 1] = x0i - x2i;
                a[2 * p2] = x0r - x2r;
                a[2 * p2 + 1] = x0i + x2i;
                a[2 * p1] = x1r - x3i;
                a[2 * p1 + 1] = x1i - x3r;
                a[2 * p3] = x1r + x3i;
                a[2 * p3 + 1] = x1i + x3r;
            }
        }
    } else { /* hc * 2 == nc */
        for (int j = 0; j < hc; j++) {
            int p0 = j, p1 = j + hc;
            float Ar = a[2 * p0], Ai = a[2 * p0 + 1], Br = a[2 * p1], Bi = a[2 * p1 + 1];
            if (!inverse) {
                a[2 * p0] = Ar + Br;
                a[2 * p0 + 1] = Ai + Bi;
                a[2 * p1] = Ar - Br;
                a[2 * p1 + 1] = Ai - Bi;
            } else {
                a[2 * p0] = Ar + Br;
                a[2 * p0 + 1] = -Ai - Bi;
                a[2 * p1] = Ar - Br;
                a[2 * p1 + 1] = -Ai + Bi;
            }
        }
    }
}

/* fft4g.c:1234-1284: real<->complex split.  Pair q with nc-q, q = 1..nc/2-1. */
static void real_split(const orc_fft_t *f, float *a, int inverse)
{
    int n = f->n, half = n >> 2; /* n/2 complex points, pairs up to half-1 */
    const float *c = f->w + f->nw;
    int ncq = f->nc, ks = 2 * ncq / (n >> 1);
    if (inverse) a[1] = -a[1];
    for (int q = 1; q < half; q++) {
        int j = 2 * q, k = n - j, kk = ks * q;
        float wkr = 0.5f - c[ncq - kk], wki = c[kk];
        float xr = a[j] - a[k], xi = a[j + 1] + a[k + 1];
        if (!inverse) {
            float yr = wkr * xr - wki * xi, yi = wkr * xi + wki * xr;
            a[j] -= yr;
            a[j + 1] -= yi;
            a[k] += yr;
            a[k + 1] -= yi;
        } else {
            float yr = wkr * xr + wki * xi, yi = wkr * xi - wki * xr;
            a[j] -= yr;
            a[j + 1] = yi - a[j + 1];
            a[k] += yr;
            a[k + 1] = yi - a[k + 1];
        }
    }
    if (inverse) a[(n >> 1) + 1] = -a[(n >> 1) + 1];
}

/* fft4g.c:324-361 with isgn = +1 */
void orc_rdft_forward(const orc_fft_t *f, float *a)
{
    permute_bitrev(f->n, a);
    complex_passes(f, a, 0);
    real_split(f, a, 0);
    float xi = a[0] - a[1];
    a[0] += a[1];
    a[1] = xi;
}

/* fft4g.c:324-361 with isgn = -1 (unnormalised: caller scales by 2/n) */
void orc_rdft_inverse(const orc_fft_t *f, float *a)
{
    a[1] = 0.5f * (a[0] - a[1]);
    a[0] -= a[1];
    real_split(f, a, 1);
    permute_bitrev(f->n, a);
    complex_passes(f, a, 1);
}

/* convenience entry points for ctypes */
static orc_fft_t g_plan[3];
static const orc_fft_t *plan_for(int n)
{
    orc_fft_t *p = &g_plan[n == 256];
    if (p->n != n) orc_fft_init(p, n);
    return p;
}
const orc_fft_t *orc_fft_aec128(void)
{
    if (g_plan[2].n != 128) orc_fft_init_aec128(&g_plan[2]);
    return &g_plan[2];
}
void orc_aec_rdft(int isgn, float *a)
{
    if (isgn >= 0)
        orc_rdft_forward(orc_fft_aec128(), a);
    else
        orc_rdft_inverse(orc_fft_aec128(), a);
}
void orc_rdft(int n, int isgn, float *a)
{
    if (isgn >= 0)
        orc_rdft_forward(plan_for(n), a);
    else
        orc_rdft_inverse(plan_for(n), a);
}
const float *orc_fft_tables(int n) { return plan_for(n)->w; }
