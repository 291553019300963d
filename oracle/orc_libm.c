/* oracle/orc_libm.c -- TEST INFRASTRUCTURE ONLY.
 * The reference's way of calling libm in the float NS / AEC paths: promote to double, call glibc, round to float
 * (ns_core.c:228 `(float)log((double)magn)`, :748 `exp`, :700 `tanh`, aec_core.c:278 powf).  Bulk versions, used by
 * tests/test_libm_tables.py to sweep the product's table-driven replacements (wmix_amd/csrc/libm_dev.h). */
#include <math.h>
#include <stddef.h>

void orc_libm_log(const float *x, float *y, size_t n)
{
    for (size_t i = 0; i < n; i++) y[i] = (float)log((double)x[i]);
}
void orc_libm_exp(const float *x, float *y, size_t n)
{
    for (size_t i = 0; i < n; i++) y[i] = (float)exp((double)x[i]);
}
void orc_libm_tanh(const float *x, float *y, size_t n)
{
    for (size_t i = 0; i < n; i++) y[i] = (float)tanh((double)x[i]);
}
/* aec_core.c:278 calls the float routine powf directly */
void orc_libm_powf(const float *x, const float *e, float *y, size_t n)
{
    for (size_t i = 0; i < n; i++) y[i] = powf(x[i], e[i]);
}
/* the same power evaluated in double and rounded once: what a correctly rounded powf would return in all but ~2^-29 of cases */
void orc_libm_pow_d(const float *x, const float *e, float *y, size_t n)
{
    for (size_t i = 0; i < n; i++) y[i] = (float)pow((double)x[i], (double)e[i]);
}
