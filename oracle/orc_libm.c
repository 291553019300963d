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
