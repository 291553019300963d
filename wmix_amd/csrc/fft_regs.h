// fft_regs.h -- the n = 128 real FFT of fft_ooura.h with the data kept in registers.
//
// Same dataflow graph, same float operations per output element (bfly4 / bfly4_close / the rftfsub-rftbsub
// pair formulas are shared with fft_ooura.h), hence the same bits; what changes is where the 64 complex
// points live between the three radix-4 passes.  A group of 16 lanes -- one DPP row -- owns a transform,
// 4 points per lane:
//
//     gather      lane l takes points rev4(l) + {0, 32, 16, 48}        (bit reversal fused, fft4g.c:693-790)
//     pass 1      butterfly b = l           -> points 4l + e            (cft1st,  fft4g.c:1002-1104)
//     transpose   lane bits 0-1 <-> e       (DPP quad_perm)            -> points 16(l/4) + l%4 + 4m
//     pass 2      butterfly b = l/4                                     (cftmdl,  fft4g.c:1107-1231)
//     transpose   lane bits 2-3 <-> e       (DPP row_ror)              -> points l + 16m
//     pass 3      closing radix-4, no twiddles                          (fft4g.c:913-934 / 963-984)
//
// so a transform costs no LDS round trip between passes; four transforms run side by side in a wave, and a
// lane can carry several (the callers unroll over REP).  The real-data pre/post processing (rftbsub before
// the inverse passes, rftfsub after the forward ones) is folded into the gather of the consumer:
// rdft128_inv_point() and rdft128_fwd_bin() evaluate one point / one bin from the packed LDS row.
#pragma once
#include "fft_ooura.h"

namespace wmx {

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// value held by lane (l ^ X) of the same 16-lane row.  quad_perm for X = 1, 2; row_ror (lane l receives from
// l - n mod 16) for X = 8 and, selected by the lane's own bit, for X = 4.
template <int X>
__device__ __forceinline__ float row_xor(float v, bool bit) {
    if constexpr (X == 1) {
        return dpp_mov<0xB1>(v);  // quad_perm:[1,0,3,2]
    } else if constexpr (X == 2) {
        return dpp_mov<0x4E>(v);  // quad_perm:[2,3,0,1]
    } else if constexpr (X == 8) {
        return dpp_mov<0x128>(v);  // row_ror:8
    } else {
        static_assert(X == 4, "row_xor: 1, 2, 4, 8");
        const float from_minus4 = dpp_mov<0x124>(v), from_plus4 = dpp_mov<0x12C>(v);  // row_ror:4, row_ror:12
        return bit ? from_minus4 : from_plus4;
    }
}

// swap lane bit X with one element-index bit: the lane whose bit is clear sends `hi` and receives into `hi`,
// its partner sends `lo` and receives into `lo`
template <int X>
__device__ __forceinline__ void xstage(Cx &lo, Cx &hi, bool bit) {
    const float sr = bit ? lo.r : hi.r, si = bit ? lo.i : hi.i;
    const float xr = row_xor<X>(sr, bit), xi = row_xor<X>(si, bit);
    lo.r = bit ? xr : lo.r;
    lo.i = bit ? xi : lo.i;
    hi.r = bit ? hi.r : xr;
    hi.i = bit ? hi.i : xi;
}

// 4x4 transpose between the lane bits (XA, XB) and the element index of v[0..3]
template <int XA, int XB>
__device__ __forceinline__ void transpose4(Cx v[4], int gl) {
    const bool ba = (gl & XA) != 0, bb = (gl & XB) != 0;
    xstage<XA>(v[0], v[1], ba);
    xstage<XA>(v[2], v[3], ba);
    xstage<XB>(v[0], v[2], bb);
    xstage<XB>(v[1], v[3], bb);
}

// point that lane gl (0..15) must supply as v[m] to fft64_regs
__device__ __forceinline__ int fft64_src_point(int gl, int m) {
    return dev_bitrev(gl, 4) + ((m & 1) << 5) + ((m & 2) << 3);  // + {0, 32, 16, 48}[m]
}

// bitrv2 + cftfsub (forward) / cftbsub (inverse) on 64 complex points.  in: v[m] = point fft64_src_point(gl, m);
// out: v[m] = point gl + 16 m.
template <bool INVERSE>
__device__ __forceinline__ void fft64_regs(Cx v[4], const FftTables *T, int gl) {
    float2 o0, o1, o2, o3;
    bfly4(gl, T, v[0], v[1], v[2], v[3], o0, o1, o2, o3);
    v[0] = Cx{o0.x, o0.y};
    v[1] = Cx{o1.x, o1.y};
    v[2] = Cx{o2.x, o2.y};
    v[3] = Cx{o3.x, o3.y};
    transpose4<1, 2>(v, gl);
    bfly4(gl >> 2, T, v[0], v[1], v[2], v[3], o0, o1, o2, o3);
    v[0] = Cx{o0.x, o0.y};
    v[1] = Cx{o1.x, o1.y};
    v[2] = Cx{o2.x, o2.y};
    v[3] = Cx{o3.x, o3.y};
    transpose4<4, 8>(v, gl);
    bfly4_close<INVERSE>(v[0], v[1], v[2], v[3], o0, o1, o2, o3);
    v[0] = Cx{o0.x, o0.y};
    v[1] = Cx{o1.x, o1.y};
    v[2] = Cx{o2.x, o2.y};
    v[3] = Cx{o3.x, o3.y};
}

// Input side of rdft(128, -1, a): point p (0..63) of the complex array the inverse passes start from, computed
// from the packed spectrum row a[128] (a[1] fix-up fft4g.c:349-351, rftbsub fft4g.c:1260-1284 incl. its sign
// flips).
__device__ __forceinline__ Cx rdft128_inv_point(const float *a, const FftTables *T, int p) {
    // one straight-line path for every point: the general pair formula is evaluated with q clamped into the
    // table (p = 0 reads a[128..129], inside the 132-float row, and discards the result), then the two special
    // points are selected in.  No divergent branches inside a wave.
    const int q = p < 32 ? p : 64 - p;
    const int j = 2 * q, k = 128 - j;
    const float wkr = 0.5f - T->c[32 - q], wki = T->c[q];
    const float aj = a[j], aj1 = a[j + 1], ak = a[k], ak1 = a[k + 1];
    const float xr = aj - ak, xi = aj1 + ak1;
    const float yr = wkr * xr + wki * xi, yi = wkr * xi - wki * xr;
    float re = p < 32 ? aj - yr : ak + yr;
    float im = p < 32 ? yi - aj1 : yi - ak1;
    const float h = 0.5f * (aj - aj1);  // p == 0: aj = a[0], aj1 = a[1]
    re = p == 0 ? aj - h : (p == 32 ? aj : re);
    im = p == 0 ? -h : (p == 32 ? -aj1 : im);
    return Cx{re, im};
}

// Output side of rdft(128, +1, a): bin b (0..64) of the spectrum, from the row holding the result of the
// forward complex passes (rftfsub fft4g.c:1234-1257 + the a[0]/a[1] fix-up fft4g.c:340-342), in the
// StoreAsComplex convention (bin 0 and bin 64 real).
__device__ __forceinline__ void rdft128_fwd_bin(const float *a, const FftTables *T, int b, float &re, float &im) {
    if (b == 0 || b == 64) {
        const float a0 = a[0], a1 = a[1];
        re = b == 0 ? a0 + a1 : a0 - a1;
        im = 0.f;
        return;
    }
    if (b == 32) {
        re = a[64];
        im = a[65];
        return;
    }
    const int q = b < 32 ? b : 64 - b;
    const int j = 2 * q, k = 128 - j;
    const float wkr = 0.5f - T->c[32 - q], wki = T->c[q];
    const float aj = a[j], aj1 = a[j + 1], ak = a[k], ak1 = a[k + 1];
    const float xr = aj - ak, xi = aj1 + ak1;
    const float yr = wkr * xr - wki * xi, yi = wkr * xi + wki * xr;
    if (b < 32) {
        re = aj - yr;
        im = aj1 - yi;
    } else {
        re = ak + yr;
        im = ak1 - yi;
    }
}

// rdft128_fwd_bin for bin == lane (0..63), straight-line; `nyq` is bin 64 (real), valid in lane 0 only.
__device__ __forceinline__ void rdft128_fwd_bin_lane(const float *a, const FftTables *T, int lane, float &re, float &im, float &nyq) {
    const int q = lane < 32 ? lane : 64 - lane;  // lane 0: q = 0 (reads a[128..129], discarded); lane 32: j == k == 64
    const int j = 2 * q, k = 128 - j;
    const float wkr = 0.5f - T->c[32 - q], wki = T->c[q];
    const float aj = a[j], aj1 = a[j + 1], ak = a[k], ak1 = a[k + 1];
    const float xr = aj - ak, xi = aj1 + ak1;
    const float yr = wkr * xr - wki * xi, yi = wkr * xi + wki * xr;
    const float gre = lane < 32 ? aj - yr : ak + yr;
    const float gim = lane < 32 ? aj1 - yi : ak1 - yi;
    re = lane == 0 ? aj + aj1 : (lane == 32 ? aj : gre);
    im = lane == 0 ? 0.f : (lane == 32 ? aj1 : gim);
    nyq = aj - aj1;
}

// value held by lane rev4(gl) of the same row (ds_bpermute: no LDS memory, no barrier)
__device__ __forceinline__ float row_bitrev(float v, int lane) {
    const int src = (lane & ~15) | dev_bitrev(lane & 15, 4);
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v)));
}

}  // namespace wmx
