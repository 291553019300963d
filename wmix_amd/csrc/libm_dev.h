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
// powf for the AEC's OverdriveAndSuppress (aec_core.c:278: hNl[i] = powf(hNl[i], overDriveSm * curve[i])) -- glibc's own
// algorithm, bit for bit (round 5).  Rounds 1-4 evaluated x^y = exp(y log x) in double end to end: nearly correctly rounded, which
// glibc's powf is NOT, so 0.09 % of arguments came out one float ulp apart and a handful of samples per 10^8 one LSB (behind the AGC
// two) off the reference.  The reference links the host's libm -- here glibc 2.35 (Ubuntu 2.35-0ubuntu3.11), not part of
// /root/reference.  glibc 2.28+ computes powf as exp2(y * log2 x) in double: log2 by a 16-entry (1/c, log2 c) table and a
// degree-4 polynomial in r = z/c - 1, exp2 by a 32-entry 2^(i/32) table and a degree-3 polynomial, one rounding to float at the end
// (sysdeps/ieee754/flt-32/e_powf.c, from Arm's optimized routines; on x86-64 the ifunc picks the build with fused multiply-adds on
// any CPU that has them -- the reference's hosts here do).  The restatement below follows it operation for operation, every a * b + c
// of that build a fused multiply-add; the 16 + 5 + 3 constants are the published ones (each log2 c re-derives from its 1/c; the
// exp2 table is 2^(i/32) with i << 47 taken off the bits), and tests/test_libm_tables.py sweeps it against the host's powf:
// 6 M arguments of the AEC's domain and the special cases, bit for bit.
struct PowTables {
    double2 lt[16];             // log2: c near the centre of [0x1.66p-1 * 2^(i/16) ...), .x = invc = RN(1/c), .y = logc = RN(log2 c)
    double A[5];                // log2(1 + r) / r - polynomial, degree 4
    unsigned long long E[32];   // bits(2^(i/32)) - (i << 47)
    double C[3];                // 2^r - 1 polynomial, degree 3
};

inline void pow_tables(PowTables *t) {
    static const double invc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
                                    0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
                                    0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                                    0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
    static const double logc[16] = {-0x1.efec65b963019p-2, -0x1.b0b6832d4fca4p-2, -0x1.7418b0a1fb77bp-2, -0x1.39de91a6dcf7bp-2,
                                    -0x1.01d9bf3f2b631p-2, -0x1.97c1d1b3b7afp-3,  -0x1.2f9e393af3c9fp-3, -0x1.960cbbf788d5cp-4,
                                    -0x1.a6f9db6475fcep-5, 0x0p+0,                0x1.338ca9f24f53dp-4,  0x1.476a9543891bap-3,
                                    0x1.e840b4ac4e4d2p-3,  0x1.40645f0c6651cp-2,  0x1.88e9c2c1b9ff8p-2,  0x1.ce0a44eb17bccp-2};
    static const unsigned long long E[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
        0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
        0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
        0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
        0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
        0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
    for (int i = 0; i < 16; i++) t->lt[i].x = invc[i], t->lt[i].y = logc[i];
    t->A[0] = 0x1.27616c9496e0bp-2, t->A[1] = -0x1.71969a075c67ap-2, t->A[2] = 0x1.ec70a6ca7baddp-2, t->A[3] = -0x1.7154748bef6c8p-1,
    t->A[4] = 0x1.71547652ab82bp0;
    for (int i = 0; i < 32; i++) t->E[i] = E[i];
    t->C[0] = 0x1.c6af84b912394p-5, t->C[1] = 0x1.ebfce50fac4f3p-3, t->C[2] = 0x1.62e42ff0c52d6p-1;
}

__host__ __device__ __forceinline__ float fast_pow(float x, float y, const PowTables *__restrict__ M) {
    unsigned ix, iy;
    __builtin_memcpy(&ix, &x, 4);
    __builtin_memcpy(&iy, &y, 4);
    unsigned long long sign_bias = 0;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || 2u * iy - 1u >= 2u * 0x7f800000u - 1u) {
        // x zero, subnormal, negative, inf or NaN; y zero, inf or NaN.  What glibc decides by RULE (zeros, infinities, NaNs) the double
        // pow decides the same way; negative finite x and subnormal x continue into the arithmetic like there (e_powf.c:150-190)
        if (2u * iy - 1u >= 2u * 0x7f800000u - 1u || 2u * ix - 1u >= 2u * 0x7f800000u - 1u) return (float)pow((double)x, (double)y);
        if (ix > 0x7f800000u) {  // x < 0: NaN unless y is an integer; an odd one flips the sign
            const int e = (int)(iy >> 23 & 0xff);
            int yint;
            if (e < 0x7f)
                yint = 0;
            else if (e > 0x7f + 23)
                yint = 2;
            else if (iy & ((1u << (0x7f + 23 - e)) - 1u))
                yint = 0;
            else
                yint = (iy & (1u << (0x7f + 23 - e))) ? 1 : 2;
            if (yint == 0) return (float)pow((double)x, (double)y);  // invalid: NaN
            if (yint == 1) sign_bias = 1ull << 16;
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {  // subnormal x: normalise
            const float xs = x * 0x1p23f;
            __builtin_memcpy(&ix, &xs, 4);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    // log2_inline
    const unsigned tmp = ix - 0x3f330000u;
    const unsigned i = (tmp >> (23 - 4)) & 15u;
    const unsigned top = tmp & 0xff800000u, iz = ix - top;
    const int k = (int)top >> 23;  // arithmetic shift
    float zf;
    __builtin_memcpy(&zf, &iz, 4);
    const double z = (double)zf;
    const double2 ic = M->lt[i];  // one 16-byte load
    const double r = fma(z, ic.x, -1.0);
    const double y0 = ic.y + (double)k;
    const double r2 = r * r;
    double yy = fma(M->A[0], r, M->A[1]);
    const double p = fma(M->A[2], r, M->A[3]);
    const double r4 = r2 * r2;
    double q = fma(M->A[4], r, y0);
    q = fma(p, r2, q);
    yy = fma(yy, r4, q);
    const double ylogx = (double)y * yy;
    unsigned long long yb;
    __builtin_memcpy(&yb, &ylogx, 8);
    if (((unsigned)(yb >> 32) >> 15 & 0xffffu) >= (0x405f8000u >> 15)) {  // |y log2 x| >= 126 (bits 47 .. 62 of the double, as glibc tests them)
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_huge_valf() : __builtin_huge_valf();
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
    }
    // exp2_inline
    constexpr double kShift = 0x1.8p+52 / 32;
    double kd = ylogx + kShift;
    unsigned long long ki;
    __builtin_memcpy(&ki, &kd, 8);
    kd -= kShift;
    const double rr = ylogx - kd;
    // t = E[ki % 32] + ((ki + sign_bias) << 47): only the low 17 bits of the sum reach the double, all of them in its high word
    unsigned long long t = M->E[(unsigned)ki & 31u];
    t += (unsigned long long)(((unsigned)ki + (unsigned)sign_bias) << 15) << 32;
    double sc;
    __builtin_memcpy(&sc, &t, 8);
    const double zz = fma(M->C[0], rr, M->C[1]);
    const double rr2 = rr * rr;
    double out = fma(M->C[2], rr, 1.0);
    out = fma(zz, rr2, out);
    out = out * sc;
    return (float)out;
}

}  // namespace wmx
