// fft_ooura.h -- the reference's split-radix real FFT, re-expressed for one wavefront.
//
// The float paths of the reference transform with Ooura's rdft (NS: n = 128 | 256,
// W:common_audio/fft4g.c:324-361; AEC: the n = 128 specialisation with frozen tables,
// W:modules/audio_processing/aec/aec_rdft.c:32-563).  NS and AEC feed the spectra back
// into threshold decisions, so we keep the reference's exact dataflow graph: every
// output element is produced by the same float operations in the same order, hence
// bit-identical results (compiled with -ffp-contract=off).  What changes is WHO computes
// what: the n/2 complex points live in LDS and each lane of the wave evaluates one
// radix-4 butterfly (or one conjugate pair of the real split) per pass.
//
//   pass 1      : fft4g.c:1002-1104 cft1st   (stride 1), bit reversal (fft4g.c:693-790)
//                 fused into its gather
//   passes 2..  : fft4g.c:1107-1231 cftmdl   (stride 4, 16)
//   closing pass: fft4g.c:913-948 / 963-998  (radix-4 when 4*stride == n/2, else radix-2;
//                 the inverse uses cftbsub's conjugating form)
//   real split  : fft4g.c:1234-1284 rftfsub / rftbsub
//
// Host side: FftTables is filled by fft_tables_ooura() (makewt/makect, fft4g.c:642-690,
// same libm calls as the reference) or fft_tables_aec128() (frozen rdft_w).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>

namespace wmx {

struct FftTables {
    float w2;           // w[2]: twiddle of the "block 1" special butterfly
    float pad[3];
    float W1[32][2];    // per radix-4 block index b >= 2 (b < n/8)
    float W2[32][2];
    float W3[32][2];
    float c[64];        // real-split table, n/4 entries (+ c[n/4] unused)
};
static_assert(sizeof(FftTables) == (4 + 64 * 3 + 64) * 4, "FftTables layout");
constexpr int kFftTableWords = sizeof(FftTables) / 4;

// ---------------------------------------------------------------- host: table builders
inline int host_bitrev(int q, int bits) {
    int r = 0;
    for (int b = 0; b < bits; b++)
        if (q & (1 << b)) r |= 1 << (bits - 1 - b);
    return r;
}

inline void host_expand_twiddles(const float *w, int n, FftTables *t) {
    volatile float tmp;  // force float rounding of every product (no host FMA contraction)
    t->w2 = w[2];
    // blocks 0 and 1 (the reference's twiddle-free / w2-only special butterflies) as table entries for the one
    // straight-line butterfly of bfly4_v (see there): b = 0 multiplies by 1, b = 1 by i, w2 and -w2
    t->W1[0][0] = t->W2[0][0] = t->W3[0][0] = 1.f;
    t->W1[1][0] = w[2];
    t->W2[1][1] = 1.f;
    t->W3[1][0] = -w[2];
    for (int b = 2; b < n / 8; b++) {
        int k1 = 2 * (b >> 1), k2 = 2 * k1;
        float wk2r = w[k1], wk2i = w[k1 + 1], wk1r, wk1i, two;
        if ((b & 1) == 0) {
            wk1r = w[k2];
            wk1i = w[k2 + 1];
            two = 2 * wk2i;
            t->W2[b][0] = wk2r;
            t->W2[b][1] = wk2i;
        } else {
            wk1r = w[k2 + 2];
            wk1i = w[k2 + 3];
            two = 2 * wk2r;
            t->W2[b][0] = -wk2i;
            t->W2[b][1] = wk2r;
        }
        tmp = two * wk1i;
        t->W3[b][0] = wk1r - tmp;
        tmp = two * wk1r;
        t->W3[b][1] = tmp - wk1i;
        t->W1[b][0] = wk1r;
        t->W1[b][1] = wk1i;
    }
}

// fft4g.c:642-690 makewt(n/4) + makect(n/4)
inline void fft_tables_ooura(int n, FftTables *t) {
    std::memset(t, 0, sizeof(*t));
    float w[64];
    const int nw = n >> 2, nwh = nw >> 1;
    {
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
        int bits = 0;
        while ((1 << bits) < nwh) bits++;
        for (int q = 0; q < nwh; q++) {  // bitrv2(nw, w): permute the nw/2 complex entries
            int r = host_bitrev(q, bits);
            if (r > q) {
                float a = w[2 * q], b = w[2 * q + 1];
                w[2 * q] = w[2 * r];
                w[2 * q + 1] = w[2 * r + 1];
                w[2 * r] = a;
                w[2 * r + 1] = b;
            }
        }
    }
    {
        const int nc = n >> 2, nch = nc >> 1;
        float delta = (float)atan(1.0f) / nch;
        t->c[0] = (float)cos(delta * nch);
        t->c[nch] = 0.5f * t->c[0];
        for (int j = 1; j < nch; j++) {
            t->c[j] = 0.5f * (float)cos(delta * j);
            t->c[nc - j] = 0.5f * (float)sin(delta * j);
        }
    }
    host_expand_twiddles(w, n, t);
}

// aec_rdft.c:32-49: frozen rdft_w[64] (IEEE-754 bit patterns; 8 entries differ by 1 ulp
// from what makewt/makect(128) give with today's libm, so the table is data).
inline void fft_tables_aec128(FftTables *t) {
    static const unsigned kW[64] = {
        0x3f800000, 0x00000000, 0x3f3504f3, 0x3f3504f3, 0x3f6c835f, 0x3ec3ef16, 0x3ec3ef16, 0x3f6c835f,
        0x3f7b14be, 0x3e47c5c2, 0x3f0e39da, 0x3f54db31, 0x3f54db31, 0x3f0e39da, 0x3e47c5c2, 0x3f7b14be,
        0x3f7ec46d, 0x3dc8bd36, 0x3f22679a, 0x3f45e403, 0x3f61c598, 0x3ef15aea, 0x3e94a031, 0x3f74fa0b,
        0x3f74fa0b, 0x3e94a031, 0x3ef15aea, 0x3f61c598, 0x3f45e403, 0x3f22679a, 0x3dc8bd36, 0x3f7ec46d,
        0x3f3504f3, 0x3effb10f, 0x3efec46d, 0x3efd3aac, 0x3efb14be, 0x3ef853f8, 0x3ef4fa0b, 0x3ef10908,
        0x3eec835f, 0x3ee76bd7, 0x3ee1c598, 0x3edb941a, 0x3ed4db31, 0x3ecd9f02, 0x3ec5e403, 0x3ebdaefa,
        0x3eb504f3, 0x3eabeb4a, 0x3ea2679a, 0x3e987fc0, 0x3e8e39da, 0x3e839c3d, 0x3e715aea, 0x3e5ae880,
        0x3e43ef16, 0x3e2c7cd4, 0x3e14a031, 0x3df8cfcd, 0x3dc7c5c2, 0x3d964083, 0x3d48bd36, 0x3cc8fb30,
    };
    std::memset(t, 0, sizeof(*t));
    float w[64];
    std::memcpy(w, kW, sizeof(w));
    std::memcpy(t->c, w + 32, 32 * sizeof(float));
    host_expand_twiddles(w, 128, t);
}

// ---------------------------------------------------------------- device: one wave, data in LDS
// `a` points to n floats in LDS owned by this wave (packed complex, reference layout);
// `T` points to the FftTables copy in LDS.  Every lane of the wave must call.  The callers
// bracket these with wave_sync() (one wave per workgroup).

// Hand-off of LDS data between lanes of ONE wavefront.  The kernels that use this header run one wave per
// workgroup; a wave's LDS instructions are executed in issue order by the hardware, so a ds_read issued after
// a ds_write of another lane of the same wave already sees the data.  What is needed is only that the COMPILER
// keeps the order: a wavefront-scope fence (no s_waitcnt vmcnt(0), no s_barrier -- unlike __syncthreads(),
// which would also drain every outstanding global load/store at each of the ~40 LDS phases of a block).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

__device__ __forceinline__ int dev_bitrev(int q, int bits) { return (int)(__brev((unsigned)q) >> (32 - bits)); }

// One full radix-4 butterfly on complex points p0 + {0,1,2,3}*hc with inputs already in
// registers.  b = block index (twiddle class).  INV_CLOSE selects cftbsub's closing form.
struct Cx { float r, i; };

// Complex values as 2-element vectors: gfx950 has packed fp32 add / mul (v_pk_add_f32, v_pk_mul_f32: both halves in
// one VALU issue, each half an ordinary IEEE single operation), which is exactly the (re, im) pairing of a butterfly.
// Every expression below is the reference's, operand for operand; only the instruction count changes.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f cx(Cx a) { return v2f{a.r, a.i}; }
__device__ __forceinline__ v2f swap(v2f a) { return __builtin_shufflevector(a, a, 1, 0); }
// (a.x - b.y, a.y + b.x) = a + i*b      and      (a.x + b.y, a.y - b.x) = a - i*b
// One v_pk_add_f32 each: its operand-select bits swap the halves of b, its negation bits apply per half.  (Written as
// vector code the compiler builds the swapped / negated operand first: an xor and a copy in front of every add.)
#define WMX_PK_ADD(name, mods)                                                      \
    __device__ __forceinline__ v2f name(v2f a, v2f b) {                             \
        v2f d;                                                                      \
        asm("v_pk_add_f32 %0, %1, %2 " mods : "=v"(d) : "v"(a), "v"(b));           \
        return d;                                                                   \
    }
WMX_PK_ADD(add_i, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")                  // (a.x - b.y, a.y + b.x)
WMX_PK_ADD(sub_i, "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")                  // (a.x + b.y, a.y - b.x)
WMX_PK_ADD(add_swap, "op_sel:[0,1] op_sel_hi:[1,0]")                            // (a.x + b.y, a.y + b.x)
WMX_PK_ADD(sub_swap, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")  // (a.x - b.y, a.y - b.x)
WMX_PK_ADD(add_conj, "neg_hi:[0,1]")                                            // (a.x + b.x, a.y - b.y)
WMX_PK_ADD(sub_conj, "neg_lo:[0,1]")                                            // (a.x - b.x, a.y + b.y)
WMX_PK_ADD(conj_sub, "neg_hi:[1,1]")                                            // (a.x + b.x, -a.y - b.y)
WMX_PK_ADD(conj_add, "neg_lo:[0,1] neg_hi:[1,0]")                               // (a.x - b.x, -a.y + b.y)
#undef WMX_PK_ADD
// (wr*t.x - wi*t.y, wr*t.y + wi*t.x): the reference's twiddle product, two packed multiplies and one packed add
__device__ __forceinline__ v2f cmul_w(float wr, float wi, v2f t) { return v2f{wr, wr} * t + v2f{-wi, wi} * swap(t); }

// in place on v[0..3] = A, B, C, D -> o0, o1, o2, o3.
// The reference has three butterfly forms: twiddle-free (block 0, fft4g.c:1013-1032 / 1118-1141), the w2-only form of
// block 1 (fft4g.c:1034-1056 / 1143-1170) and the general one.  Lanes of one wave hold butterflies of all three kinds,
// so branching on b makes every wave execute all three; instead every lane runs the general form with table entries
// that reproduce the special forms operand for operand:
//   b = 0:  W1 = W2 = W3 = 1                        1*t + 0*t' == t
//   b = 1:  o2 = i*(x0 - x2)                        W2 = i:  (0*tr - 1*ti, 0*ti + 1*tr) == (-ti, tr)
//           o1 = w2*(tr - ti, tr + ti)              t rotated first (m = 1), then W1 = (w2, 0)
//           o3 = w2*(-ti - tr, tr - ti)             == -w2*(tr + ti, ti - tr): rotated the other way, W3 = (-w2, 0)
// with m = 0 outside block 1 (tr - 0*ti == tr).  Products with 0 and 1 and sign changes are exact, so each output is
// the reference's value (a zero result may carry the other sign; nothing downstream can see that).
__device__ __forceinline__ void bfly4_v(int b, const FftTables *T, v2f v[4]) {
    const v2f x0 = v[0] + v[1], x1 = v[0] - v[1], x2 = v[2] + v[3], x3 = v[2] - v[3];
    v[0] = x0 + x2;
    const float m = b == 1 ? 1.f : 0.f;
    const float w1r = T->W1[b][0], w1i = T->W1[b][1], w2r = T->W2[b][0], w2i = T->W2[b][1];
    const float w3r = T->W3[b][0], w3i = T->W3[b][1];
    v2f t1 = add_i(x1, x3);  // (x1r - x3i, x1i + x3r)
    v2f t3 = sub_i(x1, x3);  // (x1r + x3i, x1i - x3r)
    // m is 0 or 1: the product is exact, so the fused form rounds once to the value the product + sum pair rounds to (a zero
    // result takes the same sign either way) -- one v_pk_fma_f32 instead of v_pk_mul_f32 + v_pk_add_f32
    t1 = __builtin_elementwise_fma(v2f{-m, m}, swap(t1), t1);
    t3 = __builtin_elementwise_fma(v2f{m, -m}, swap(t3), t3);
    v[2] = cmul_w(w2r, w2i, x0 - x2);
    v[1] = cmul_w(w1r, w1i, t1);
    v[3] = cmul_w(w3r, w3i, t3);
}

__device__ __forceinline__ void bfly4(int b, const FftTables *T, Cx A, Cx B, Cx C, Cx D, float2 &o0, float2 &o1, float2 &o2,
                                      float2 &o3) {
    v2f v[4] = {cx(A), cx(B), cx(C), cx(D)};
    bfly4_v(b, T, v);
    o0 = make_float2(v[0].x, v[0].y);
    o1 = make_float2(v[1].x, v[1].y);
    o2 = make_float2(v[2].x, v[2].y);
    o3 = make_float2(v[3].x, v[3].y);
}

__device__ __forceinline__ void bfly4_store(float *a, int p0, int hc, int b, const FftTables *T, Cx A, Cx B, Cx C, Cx D) {
    const int p1 = p0 + hc, p2 = p1 + hc, p3 = p2 + hc;
    float2 o0, o1, o2, o3;
    bfly4(b, T, A, B, C, D, o0, o1, o2, o3);
    *reinterpret_cast<float2 *>(a + 2 * p0) = o0;
    *reinterpret_cast<float2 *>(a + 2 * p1) = o1;
    *reinterpret_cast<float2 *>(a + 2 * p2) = o2;
    *reinterpret_cast<float2 *>(a + 2 * p3) = o3;
}

// closing radix-4 without twiddles (fft4g.c:913-934 forward / 963-984 conjugating inverse), in place on v[0..3]
template <bool INVERSE>
__device__ __forceinline__ void bfly4_close_v(v2f v[4]) {
    const v2f x2 = v[2] + v[3], x3 = v[2] - v[3];
    if constexpr (!INVERSE) {
        const v2f x0 = v[0] + v[1], x1 = v[0] - v[1];
        v[0] = x0 + x2;
        v[2] = x0 - x2;
        v[1] = add_i(x1, x3);  // (x1r - x3i, x1i + x3r)
        v[3] = sub_i(x1, x3);  // (x1r + x3i, x1i - x3r)
    } else {
        // x0r = A.r + B.r, x0i = -A.i - B.i; x1r = A.r - B.r, x1i = -A.i + B.i
        const v2f x0 = conj_sub(v[0], v[1]), x1 = conj_add(v[0], v[1]);
        v[0] = add_conj(x0, x2);  // (x0r + x2r, x0i - x2i)
        v[2] = sub_conj(x0, x2);  // (x0r - x2r, x0i + x2i)
        v[1] = sub_swap(x1, x3);  // (x1r - x3i, x1i - x3r)
        v[3] = add_swap(x1, x3);  // (x1r + x3i, x1i + x3r)
    }
}
template <bool INVERSE>
__device__ __forceinline__ void bfly4_close(Cx A, Cx B, Cx C, Cx D, float2 &o0, float2 &o1, float2 &o2, float2 &o3) {
    v2f v[4] = {cx(A), cx(B), cx(C), cx(D)};
    bfly4_close_v<INVERSE>(v);
    o0 = make_float2(v[0].x, v[0].y);
    o1 = make_float2(v[1].x, v[1].y);
    o2 = make_float2(v[2].x, v[2].y);
    o3 = make_float2(v[3].x, v[3].y);
}

__device__ __forceinline__ Cx ld_cx(const float *a, int p) {
    float2 v = *reinterpret_cast<const float2 *>(a + 2 * p);
    return Cx{v.x, v.y};
}

// The complex passes (bitrv2 + cftfsub / cftbsub) for NC = n/2 complex points in LDS, one transform per wave:
// every pass has NC/4 butterflies, lane k takes butterfly k.
template <int NC, bool INVERSE>
__device__ __forceinline__ void fft_complex_passes(float *a, const FftTables *T, int lane) {
    static_assert(NC / 4 <= 64, "one butterfly per lane");
    constexpr int BITS = (NC == 128) ? 7 : 6;
    constexpr int NB = NC / 4;  // butterflies per radix-4 pass
    // pass 1: stride 1, gathering from bit-reversed positions.  rev(4g + j) over BITS bits =
    // rev(g) over BITS-2 bits + {0, NC/2, NC/4, 3NC/4}[j].
    {
        Cx A{0.f, 0.f}, B = A, C = A, D = A;
        const bool act = lane < NB;
        if (act) {
            const int r0 = dev_bitrev(lane, BITS - 2);
            A = ld_cx(a, r0);
            B = ld_cx(a, r0 + NC / 2);
            C = ld_cx(a, r0 + NC / 4);
            D = ld_cx(a, r0 + 3 * NC / 4);
        }
        wave_sync();
        if (act) bfly4_store(a, 4 * lane, 1, lane, T, A, B, C, D);
        wave_sync();
    }
    // twiddled passes with stride 4, 16 while 4*hc < NC
#pragma unroll
    for (int hc = 4; hc * 4 < NC; hc *= 4) {
        if (lane < NB) {
            const int b = lane / hc, j = lane % hc, p0 = b * 4 * hc + j;
            bfly4_store(a, p0, hc, b, T, ld_cx(a, p0), ld_cx(a, p0 + hc), ld_cx(a, p0 + 2 * hc), ld_cx(a, p0 + 3 * hc));
        }
        wave_sync();
    }
    constexpr int HC = (NC == 128) ? 64 : 16;  // stride of the closing pass
    if constexpr (HC * 4 == NC) {
        if (lane < HC) {
            const int p0 = lane, p1 = p0 + HC, p2 = p1 + HC, p3 = p2 + HC;
            const Cx A = ld_cx(a, p0), B = ld_cx(a, p1), C = ld_cx(a, p2), D = ld_cx(a, p3);
            float2 o0, o1, o2, o3;
            bfly4_close<INVERSE>(A, B, C, D, o0, o1, o2, o3);
            *reinterpret_cast<float2 *>(a + 2 * p0) = o0;
            *reinterpret_cast<float2 *>(a + 2 * p1) = o1;
            *reinterpret_cast<float2 *>(a + 2 * p2) = o2;
            *reinterpret_cast<float2 *>(a + 2 * p3) = o3;
        }
    } else {
        // closing radix-2 (fft4g.c:936-947 / 986-997), HC == NC/2
        for (int p0 = lane; p0 < HC; p0 += 64) {
            const int p1 = p0 + HC;
            const Cx A = ld_cx(a, p0), B = ld_cx(a, p1);
            float2 o0, o1;
            if constexpr (!INVERSE) {
                o0 = make_float2(A.r + B.r, A.i + B.i);
                o1 = make_float2(A.r - B.r, A.i - B.i);
            } else {
                o0 = make_float2(A.r + B.r, -A.i - B.i);
                o1 = make_float2(A.r - B.r, -A.i + B.i);
            }
            *reinterpret_cast<float2 *>(a + 2 * p0) = o0;
            *reinterpret_cast<float2 *>(a + 2 * p1) = o1;
        }
    }
    wave_sync();
}

// fft4g.c:1234-1284 rftfsub / rftbsub: conjugate pairs (q, NC - q), q = 1 .. NC/2 - 1.
template <int NC, bool INVERSE>
__device__ __forceinline__ void fft_real_split(float *a, const FftTables *T, int lane) {
    constexpr int NQ = NC / 2;  // table length n/4
    for (int q = lane; q < NQ; q += 64) {
        if (q == 0) continue;
        const int j = 2 * q, k = 2 * NC - j;
        const float wkr = 0.5f - T->c[NQ - q], wki = T->c[q];
        const float aj = a[j], aj1 = a[j + 1], ak = a[k], ak1 = a[k + 1];
        const float xr = aj - ak, xi = aj1 + ak1;
        if constexpr (!INVERSE) {
            const float yr = wkr * xr - wki * xi, yi = wkr * xi + wki * xr;
            a[j] = aj - yr;
            a[j + 1] = aj1 - yi;
            a[k] = ak + yr;
            a[k + 1] = ak1 - yi;
        } else {
            const float yr = wkr * xr + wki * xi, yi = wkr * xi - wki * xr;
            a[j] = aj - yr;
            a[j + 1] = yi - aj1;
            a[k] = ak + yr;
            a[k + 1] = yi - ak1;
        }
    }
    if constexpr (INVERSE) {
        if (lane == 0) {
            a[1] = -a[1];
            a[NC + 1] = -a[NC + 1];
        }
    }
    wave_sync();
}

// WebRtc_rdft(n, +1, a) / aec_rdft_forward_128(a).  n = 2*NC.
template <int NC>
__device__ __forceinline__ void rdft_forward(float *a, const FftTables *T, int lane) {
    fft_complex_passes<NC, false>(a, T, lane);
    fft_real_split<NC, false>(a, T, lane);
    if (lane == 0) {
        const float a0 = a[0], a1 = a[1];
        a[0] = a0 + a1;
        a[1] = a0 - a1;
    }
    wave_sync();
}

// WebRtc_rdft(n, -1, a) / aec_rdft_inverse_128(a); unnormalised like the reference.
template <int NC>
__device__ __forceinline__ void rdft_inverse(float *a, const FftTables *T, int lane) {
    if (lane == 0) {
        const float a0 = a[0], a1 = a[1];
        const float h = 0.5f * (a0 - a1);
        a[1] = h;
        a[0] = a0 - h;
    }
    wave_sync();
    fft_real_split<NC, true>(a, T, lane);
    fft_complex_passes<NC, true>(a, T, lane);
}

}  // namespace wmx
