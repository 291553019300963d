// libm_dev.h -- the reference's double-precision libm calls, rounded to float.
//
// The float NS / AEC call log / exp / pow / tanh on a float promoted to double and round the result back
// (ns_core.c:228,233,...; aec_core.c:278,1049).  The hot call sites use the table-driven routines below (DESIGN.md
// section 3.1 "libm"); the rare ones stay on ocml's fp64 routines, kept out of line: inlined, the compiler hoists
// their ~40 registers of polynomial coefficients out of the kernels' packet loops and keeps them live for the whole
// kernel, which costs a wave per SIMD.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

namespace wmx {

__device__ __noinline__ static float log_d(float x) { return (float)log((double)x); }
// tanh of a promoted float, rounded back (ns_core.c:700,1321); out of line so that its fp64 coefficients are not hoisted
// out of the callers' packet loops and held (or spilled) there
__device__ __noinline__ static float tanh_d(float x) { return (float)tanh((double)x); }
// the start-up pink-noise model (ns_core.c:1111-1153, first 50 blocks only): exp and x / pow(b, e) in double, rounded back
__device__ __noinline__ static float exp_d(float x) { return (float)exp((double)x); }
__device__ __noinline__ static float div_pow_d(float num, float base, float e) { return (float)((double)num / pow((double)base, (double)e)); }

// the library routines behind the table-driven ones' guard branches (arguments the NS never produces): out of line on the
// device for the reason given at the top -- inlined three times each, their coefficients had 32 VGPRs pinned in ns_kernel
__host__ __device__ __forceinline__ float slow_log(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return log_d(x);
#else
    return (float)log((double)x);
#endif
}
__host__ __device__ __forceinline__ float slow_exp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return exp_d(x);
#else
    return (float)exp((double)x);
#endif
}

// p * r + c for a literal coefficient c.  On the device the coefficient sits in an SGPR pair and the operation is one
// three-address v_fma_f64; written as plain fma() the compiler keeps every coefficient of a polynomial in a VGPR pair for
// the whole kernel and spells each Horner step as v_mov_b64 (copy the coefficient) + v_fmac_f64 (accumulate into the
// copy): twice the VALU work and two dozen registers.  Same single rounding either way.
__host__ __device__ __forceinline__ double fma_c(double p, double r, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(r), "s"(c));
    return d;
#else
    return fma(p, r, c);
#endif
}

// a / b, correctly rounded, for ORDINARY operands: a == 0 or 2^-96 <= |a| <= 2^96, the same box for b, exponents less than
// 96 apart, a quotient that is a normal number.  (A numerator below 2^-100 is NOT ordinary: the residuals f2, f4 underflow and
// one quotient in four comes out an ulp off -- measured; that is what v_div_scale_f32 is there for.)  What the compiler emits for `a / b` is this same sequence -- reciprocal
// estimate, one Newton step on it, quotient, two residual corrections (W:llvm AMDGPU LowerFDIV32; Markstein's theorem makes
// the last one exact) -- wrapped in two v_div_scale_f32 in front and v_div_fmas_f32 / v_div_fixup_f32 behind, which rescale
// operands outside that box and patch zeros, infinities and NaNs: for operands inside it they pass their inputs through and
// the result is fma(f4, f1, f3) below, bit for bit.  43.6 cycles of SIMD time become 31.2 (profiles/r03/issue_costs.json).
// Callers state their operand ranges; tests/test_div_ordinary.py sweeps 10^8 pairs on the host and on the GPU.
__host__ __device__ __forceinline__ float div_ordinary(float a, float b) {
#if defined(WMX_DIV_IEEE)  // developer switch (make EXTRA=-DWMX_DIV_IEEE): the compiler's full sequence, for A/B timing
    return a / b;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(b);
#else
    const float r0 = 1.0f / b;  // any estimate within an ulp ends in the same quotient
#endif
    const float nb = -b;
    const float f0 = __builtin_fmaf(nb, r0, 1.0f);
    const float f1 = __builtin_fmaf(f0, r0, r0);
    const float m = a * f1;
    const float f2 = __builtin_fmaf(nb, m, a);
    const float f3 = __builtin_fmaf(f2, f1, m);
    const float f4 = __builtin_fmaf(nb, f3, a);
    return __builtin_fmaf(f4, f1, f3);
}

// ---------------------------------------------------------------------------------------------------------------
// Table-driven log / exp for the NS's per-bin calls.  The reference computes float(log((double)x)) and
// float(exp((double)x)) with glibc (error < 1 ulp of double).  Any double result within ~1 ulp of the true value rounds
// to the same float except when the true value lies within that ulp of a float rounding boundary (probability ~2^-28
// per call) -- the same exposure the ocml routines have.  These keep that accuracy class (a few 2^-53 relative) with
// ~25 fp64 operations instead of ocml's ~100-400 (it carries double-double arithmetic for a 0.5-ulp bound that the
// final rounding to float throws away):
//   log(x), x >= 1:  x = 2^e * m, m in [1,2); c = 1 + idx/128 (idx = top 7 mantissa bits), r = m * RN(1/c) - 1 in
//                    [0, 2^-7); log x = e*ln2 + log c + log1p(r), log1p by its Taylor polynomial to r^9 (r^10/10 <
//                    2^-73).  idx = 0 has c = 1, log c = 0, so log(1.0f) = 0 exactly and values next to 1 keep full
//                    relative accuracy.
//   exp(x):          k = rint(x * 64/ln2), r = x - k*ln2/64 (two-part constant), |r| <= 0.0055;
//                    e^x = 2^(k>>6) * T[k&63] * (1 + expm1(r)), expm1 by Taylor to r^6 (r^7/5040 < 2^-63).
// Tables (NsLibmTables, built by the host with the host libm) live in LDS with the other per-block constants.
//   tanh(x):         odd, so for a = |x|:  tanh a = E / (E + 2) with E = expm1(2a) = 2^q * T[j] * (1 + p) - 1 from the
//                    exp reduction of 2a (k = 64 q + j).  For q = 0 the subtraction would cancel against the rounding
//                    error of T[j], so E = Tm1[j] + T[j] * p with Tm1[j] = RN(expm1(j ln2/64)) from the host's libm
//                    (j = 0: E = p, full relative accuracy next to 0); for q >= 1, E >= 1 and the subtraction is
//                    harmless.  One correctly rounded fp64 division; 2a >= 40 returns 1 (tanh differs from 1 by
//                    < 2^-56 there).  The NS calls it once per frame (three indicator arguments, ns_core.c:700-731).
struct NsLibmTables {
    double2 logtab[128];  // (RN(1/c), RN(log c)), c = 1 + i/128
    double exptab[64];    // 2^(j/64)
    double expm1tab[64];  // 2^(j/64) - 1, rounded from the exact value
};
constexpr int kNsLibmWords = sizeof(NsLibmTables) / 4;

inline void ns_libm_tables(NsLibmTables *t) {
    for (int i = 0; i < 128; i++) {
        const double c = 1.0 + i / 128.0;
        t->logtab[i].x = 1.0 / c;
        t->logtab[i].y = log(c);
    }
    for (int j = 0; j < 64; j++) {
        t->exptab[j] = exp2(j / 64.0);
        t->expm1tab[j] = expm1(j * 0.010830424696249145);  // j * ln2 / 64 (the product is exact enough: 2^-60 relative)
    }
}

__host__ __device__ __forceinline__ float fast_log_ge1(float x, const NsLibmTables &M) {
    if (!(x >= 1.0f && x < 3.0e38f)) return slow_log(x);  // never taken on the NS's arguments (|X| + 1, 1 + 2 snr)
    unsigned u;
    __builtin_memcpy(&u, &x, 4);
    const int e = (int)(u >> 23) - 127;
    const unsigned mu = (u & 0x007FFFFFu) | 0x3F800000u;
    float mf;
    __builtin_memcpy(&mf, &mu, 4);
    const double m = (double)mf;
    const double2 t = M.logtab[(u >> 16) & 0x7F];
    const double r = fma(m, t.x, -1.0);
    double p = -1.0 / 9.0 * r + 1.0 / 8.0;  // alternating series, highest term r^9/9
    p = fma_c(p, r, -1.0 / 7.0);
    p = fma_c(p, r, 1.0 / 6.0);
    p = fma_c(p, r, -1.0 / 5.0);
    p = fma_c(p, r, 1.0 / 4.0);
    p = fma_c(p, r, -1.0 / 3.0);
    p = fma(p, r, 1.0 / 2.0);
    p = fma(-p, r, 1.0);
    p = p * r;  // log1p(r) = r - r^2/2 + ... + r^9/9
    const double ed = (double)e;
    constexpr double kLn2Hi = 0x1.62e42fefa38p-1, kLn2Lo = 0x1.ef35793c7673p-45;  // ln2 split: e * hi is exact for |e| < 2^11
    return (float)(fma(ed, kLn2Hi, t.y) + fma(ed, kLn2Lo, p));
}

__host__ __device__ __forceinline__ float fast_exp(float xf, const NsLibmTables &M) {
    const double x = (double)xf;
    if (!(x > -700.0 && x < 700.0)) return slow_exp(xf);  // overflow / deep underflow / NaN: the library routine
    constexpr double kInv = 0x1.71547652b82fep+6;                                   // 64 / ln2
    constexpr double kHi = 0x1.62e42fefa0000p-7, kLo = 0x1.cf79abc9e3b3ap-46;       // ln2 / 64, kHi * k exact for |k| < 2^21
    const double kd = rint(x * kInv);
    const int k = (int)kd;
    double r = fma(-kd, kHi, x);
    r = fma(-kd, kLo, r);
    double p = fma_c(r, 1.0 / 720.0, 1.0 / 120.0);
    p = fma_c(p, r, 1.0 / 24.0);
    p = fma_c(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = p * r;  // expm1(r)
    const double t = M.exptab[k & 63];
    const double y = fma(t, p, t);
    // scale by 2^(k >> 6): y in [1, 2), |k >> 6| < 1100 / 64, the result is a normal double
    long long bits;
    __builtin_memcpy(&bits, &y, 8);
    bits += (long long)(k >> 6) << 52;
    double z;
    __builtin_memcpy(&z, &bits, 8);
    return (float)z;
}

__host__ __device__ __forceinline__ float fast_tanh(float xf, const NsLibmTables &M) {
    const double a2 = 2.0 * fabs((double)xf);
    if (!(a2 < 40.0)) {
        if (a2 != a2) return xf + xf;  // NaN in, (quiet) NaN out, like the library routine
        return xf < 0.f ? -1.0f : 1.0f;
    }
    constexpr double kInv = 0x1.71547652b82fep+6;                              // 64 / ln2
    constexpr double kHi = 0x1.62e42fefa0000p-7, kLo = 0x1.cf79abc9e3b3ap-46;  // ln2 / 64 in two parts
    const double kd = rint(a2 * kInv);
    const int k = (int)kd;  // 0 .. 3694
    double r = fma(-kd, kHi, a2);
    r = fma(-kd, kLo, r);
    double p = fma_c(r, 1.0 / 720.0, 1.0 / 120.0);
    p = fma_c(p, r, 1.0 / 24.0);
    p = fma_c(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = p * r;  // expm1(r)
    const double t = M.exptab[k & 63];
    double E;
    if ((k >> 6) == 0) {
        E = fma(t, p, M.expm1tab[k & 63]);
    } else {
        const double y = fma(t, p, t);
        long long bits;
        __builtin_memcpy(&bits, &y, 8);
        bits += (long long)(k >> 6) << 52;
        double z;
        __builtin_memcpy(&z, &bits, 8);
        E = z - 1.0;
    }
    const double th = E / (E + 2.0);
    return copysignf((float)th, xf);
}

// ---------------------------------------------------------------------------------------------------------------
// powf for the AEC's OverdriveAndSuppress (aec_core.c:278: hNl[i] = powf(hNl[i], overDriveSm * curve[i])), same
// scheme: x^y = exp(y * log x) with log and exp as above but kept in double end to end (no intermediate rounding), for
// 0 < x < inf.  The log is the general-argument version: mantissas above sqrt(2) are folded down (c/2 with e + 1) so
// that arguments next to 1 -- no suppression, the common case -- see no cancellation between e*ln2 and log c.
// Error ~ y * 2^-53 * |log x| relative, i.e. the result rounds to glibc's float except within ~2^-48 of a rounding
// boundary; glibc's powf itself is not correctly rounded, and the parity criterion for the AEC is <= 1 LSB.
struct PowTables {
    double2 logtab[128];  // (RN(1/c), RN(log c)) for c < sqrt 2, (RN(1/c), RN(log(c/2))) above; c = 1 + i/128
    double exptab[64];    // 2^(j/64)
};
constexpr int kPowFold = 53;  // first i with 1 + i/128 > sqrt(2)

inline void pow_tables(PowTables *t) {
    for (int i = 0; i < 128; i++) {
        const double c = 1.0 + i / 128.0;
        t->logtab[i].x = 1.0 / c;
        t->logtab[i].y = i < kPowFold ? log(c) : log(c / 2.0);
    }
    for (int j = 0; j < 64; j++) t->exptab[j] = exp2(j / 64.0);
}

__host__ __device__ __forceinline__ float fast_pow(float x, float y, const PowTables *__restrict__ M) {
    unsigned u;
    __builtin_memcpy(&u, &x, 4);
    if (!(u >= 0x00800000u && u < 0x7F800000u)) return (float)pow((double)x, (double)y);  // zero, denormal, negative, inf, NaN
    const int idx = (u >> 16) & 0x7F;
    const int e = (int)(u >> 23) - 127 + (idx >= kPowFold ? 1 : 0);
    const unsigned mu = (u & 0x007FFFFFu) | 0x3F800000u;
    float mf;
    __builtin_memcpy(&mf, &mu, 4);
    const double2 t = M->logtab[idx];
    const double r = fma((double)mf, t.x, -1.0);
    double p = -1.0 / 9.0 * r + 1.0 / 8.0;
    p = fma_c(p, r, -1.0 / 7.0);
    p = fma_c(p, r, 1.0 / 6.0);
    p = fma_c(p, r, -1.0 / 5.0);
    p = fma_c(p, r, 1.0 / 4.0);
    p = fma_c(p, r, -1.0 / 3.0);
    p = fma(p, r, 1.0 / 2.0);
    p = fma(-p, r, 1.0);
    p = p * r;
    const double ed = (double)e;
    constexpr double kLn2Hi = 0x1.62e42fefa38p-1, kLn2Lo = 0x1.ef35793c7673p-45;
    const double lg = fma(ed, kLn2Hi, t.y) + fma(ed, kLn2Lo, p);
    const double z = (double)y * lg;
    if (!(z > -700.0 && z < 700.0)) return (float)exp(z);
    constexpr double kInv = 0x1.71547652b82fep+6, kHi = 0x1.62e42fefa0000p-7, kLo = 0x1.cf79abc9e3b3ap-46;
    const double kd = rint(z * kInv);
    const int k = (int)kd;
    double rr = fma(-kd, kHi, z);
    rr = fma(-kd, kLo, rr);
    double q = fma_c(rr, 1.0 / 720.0, 1.0 / 120.0);
    q = fma_c(q, rr, 1.0 / 24.0);
    q = fma_c(q, rr, 1.0 / 6.0);
    q = fma(q, rr, 0.5);
    q = fma(q, rr, 1.0);
    q = q * rr;
    const double tt = M->exptab[k & 63];
    const double yv = fma(tt, q, tt);
    long long bits;
    __builtin_memcpy(&bits, &yv, 8);
    bits += (long long)(k >> 6) << 52;
    double out;
    __builtin_memcpy(&out, &bits, 8);
    return (float)out;
}

}  // namespace wmx
