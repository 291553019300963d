// fft_regs.h -- the n = 128 real FFT of fft_ooura.h with the data kept in registers.
//
// Same dataflow graph, same float operations per output element (bfly4 / bfly4_close / the rftfsub-rftbsub
// pair formulas are shared with fft_ooura.h), hence the same bits; what changes is where the 64 complex
// points live between the three radix-4 passes.  A group of 16 lanes (the lanes with the same lane % 4; l = lane / 4
// below) owns a transform, 4 points per lane:
//
//     gather      lane l takes points rev4(l) + {0, 32, 16, 48}        (bit reversal fused, fft4g.c:693-790)
//     pass 1      butterfly b = l           -> points 4l + e            (cft1st,  fft4g.c:1002-1104)
//     transpose   bits 0-1 of l <-> e       (DPP row_ror)              -> points 16(l/4) + l%4 + 4m
//     pass 2      butterfly b = l/4                                     (cftmdl,  fft4g.c:1107-1231)
//     transpose   bits 2-3 of l <-> e       (v_permlane16/32_swap)     -> points l + 16m
//     pass 3      closing radix-4, no twiddles                          (fft4g.c:913-934 / 963-984)
//
// so a transform costs no LDS round trip between passes; four transforms run side by side in a wave, and a
// lane can carry several (the callers unroll over REP).  The real-data pre/post processing (rftbsub before
// the inverse passes, rftfsub after the forward ones) is folded into the gather of the consumer:
// rdft128_inv_point_m() and rdft128_fwd_bin_u() evaluate one point / one bin from the packed LDS row.
#pragma once
#include <cstddef>
#include "fft_ooura.h"

namespace wmx {

// Which lanes form a transform: the 16 lanes with the same lane % 4 (group g = lane & 3, index inside the transform
// gl = lane >> 2), not a contiguous DPP row.  Bit k of gl is then lane bit k + 2, and every one of the four lane-bit <->
// element-bit exchanges of the two transposes has a cheap form on gfx950:
//     gl bit 0 / 1 = lane bit 2 / 3: the partner is 4 / 8 lanes away in the same DPP row: v_mov_b32_dpp row_ror under a
//                    bank mask (a DPP bank is 4 lanes, so "bit 2 / bit 3 of the lane" is a set of banks; only the
//                    receiving lanes are written, no select)
//     gl bit 2 = lane bit 4: v_permlane16_swap_b32 (odd rows of one register <-> even rows of the other)
//     gl bit 3 = lane bit 5: v_permlane32_swap_b32 (upper half of one register <-> lower half of the other)
// -- the two swaps are exactly the exchange a transpose step needs, one instruction for both registers, in place.
// Measured (tools_dev/ubench/issue_cost.hip, cycles of SIMD time per wave64 instruction at 4 waves/SIMD, profiles/r03/issue_costs.json):
// v_mov_b32 2.0, v_mov_b32_dpp 4.0, v_permlane16_swap / v_permlane32_swap 7.6, v_cndmask_b32 4.0 behind its v_cmp (4.0), ds_bpermute 24 of
// the LDS pipe.  Per register pair an exchange costs 10 (copy + two DPP moves) or 7.6 (swap); the contiguous-row mapping used before
// needed a quad_perm fetch + select per register for lane bits 0 / 1 (16 per pair).
__device__ __forceinline__ int fft_group(int lane) { return lane & 3; }
__device__ __forceinline__ int fft_index(int lane) { return lane >> 2; }
template <int CTRL, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, BANK_MASK, false));
}
// Swap bit X of gl with one element-index bit: lanes whose bit is clear keep `lo` and receive the partner's `lo` into
// `hi`; lanes whose bit is set keep `hi` and receive the partner's `hi` into `lo`.
template <int X>
__device__ __forceinline__ void xstage1(float &lo, float &hi) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    if constexpr (X == 1) {
        const float nlo = dpp_mov<0x124, 0xA>(lo, hi);  // row_ror:4  -> lanes 4-7, 12-15 take hi of lane - 4
        const float nhi = dpp_mov<0x12C, 0x5>(hi, lo);  // row_ror:12 -> lanes 0-3, 8-11 take lo of lane + 4
        lo = nlo, hi = nhi;
    } else if constexpr (X == 2) {
        const float nlo = dpp_mov<0x128, 0xC>(lo, hi);  // row_ror:8 -> lanes 8-15 take hi of lane - 8
        const float nhi = dpp_mov<0x128, 0x3>(hi, lo);  //           -> lanes 0-7 take lo of lane + 8
        lo = nlo, hi = nhi;
    } else if constexpr (X == 4) {
        const v2u r = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
        lo = __uint_as_float(r.x), hi = __uint_as_float(r.y);
    } else {
        const v2u r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
        lo = __uint_as_float(r.x), hi = __uint_as_float(r.y);
    }
}
template <int X>
__device__ __forceinline__ void xstage(v2f &lo, v2f &hi) {
    float lx = lo.x, hx = hi.x, ly = lo.y, hy = hi.y;
    xstage1<X>(lx, hx);
    xstage1<X>(ly, hy);
    lo = v2f{lx, ly};
    hi = v2f{hx, hy};
}

// 4x4 transpose between the bits (XA, XB) of gl and the element index of v[0..3]
template <int XA, int XB>
__device__ __forceinline__ void transpose4(v2f v[4]) {
    xstage<XA>(v[0], v[1]);
    xstage<XA>(v[2], v[3]);
    xstage<XB>(v[0], v[2]);
    xstage<XB>(v[1], v[3]);
}

// point that lane gl (0..15) must supply as v[m] to fft64_regs
__device__ __forceinline__ int fft64_src_point(int gl, int m) {
    return dev_bitrev(gl, 4) + ((m & 1) << 5) + ((m & 2) << 3);  // + {0, 32, 16, 48}[m]
}

// bitrv2 + cftfsub (forward) / cftbsub (inverse) on 64 complex points.  in: v[m] = point fft64_src_point(gl, m);
// out: v[m] = point gl + 16 m.
template <bool INVERSE>
__device__ __forceinline__ void fft64_regs(v2f v[4], const FftTables *T, int gl) {
    bfly4_v(gl, T, v);
    transpose4<1, 2>(v);
    bfly4_v(gl >> 2, T, v);
    transpose4<4, 8>(v);
    bfly4_close_v<INVERSE>(v);
}
template <bool INVERSE>
__device__ __forceinline__ void fft64_regs(Cx c[4], const FftTables *T, int gl) {
    v2f v[4] = {cx(c[0]), cx(c[1]), cx(c[2]), cx(c[3])};
    fft64_regs<INVERSE>(v, T, gl);
#pragma unroll
    for (int m = 0; m < 4; m++) c[m] = Cx{v[m].x, v[m].y};
}

// ---------------------------------------------------------------------------------------------------------------
// TWO transforms per lane, packed across the transforms: every value is a v2f whose halves belong to transform 0 and
// transform 1 of the same 16-lane group (real and imaginary parts in separate vectors).  The butterflies then are
// purely element-wise -- v_pk_add_f32 / v_pk_mul_f32 with the (shared) twiddle broadcast, no swizzles, no sign
// shuffles -- and cost what one transform costs; only the DPP transposes move each 32-bit half on its own.  Formulas:
// the scalar ones of the reference (fft4g.c:1002-1231, 913-984), operand for operand.
struct Cx2 {
    v2f r, i;
};
__device__ __forceinline__ v2f bc(float x) { return v2f{x, x}; }

__device__ __forceinline__ void bfly4_x2(int b, const FftTables *T, Cx2 v[4]) {
    // the straight-line butterfly of bfly4_v (fft_ooura.h), element-wise on the two packed transforms
    const v2f x0r = v[0].r + v[1].r, x0i = v[0].i + v[1].i, x1r = v[0].r - v[1].r, x1i = v[0].i - v[1].i;
    const v2f x2r = v[2].r + v[3].r, x2i = v[2].i + v[3].i, x3r = v[2].r - v[3].r, x3i = v[2].i - v[3].i;
    v[0].r = x0r + x2r;
    v[0].i = x0i + x2i;
    const v2f m = bc(b == 1 ? 1.f : 0.f);
    const v2f w1r = bc(T->W1[b][0]), w1i = bc(T->W1[b][1]), w2r = bc(T->W2[b][0]), w2i = bc(T->W2[b][1]);
    const v2f w3r = bc(T->W3[b][0]), w3i = bc(T->W3[b][1]);
    v2f tr = x0r - x2r, ti = x0i - x2i;
    v[2].r = w2r * tr - w2i * ti;
    v[2].i = w2r * ti + w2i * tr;
    tr = x1r - x3i;
    ti = x1i + x3r;
    {
        // m is 0 or 1 (exact product): fused, one rounding, the value of the product + sum pair (see bfly4_v)
        const v2f a = __builtin_elementwise_fma(-m, ti, tr), c = __builtin_elementwise_fma(m, tr, ti);
        tr = a;
        ti = c;
    }
    v[1].r = w1r * tr - w1i * ti;
    v[1].i = w1r * ti + w1i * tr;
    tr = x1r + x3i;
    ti = x1i - x3r;
    {
        const v2f a = __builtin_elementwise_fma(m, ti, tr), c = __builtin_elementwise_fma(-m, tr, ti);
        tr = a;
        ti = c;
    }
    v[3].r = w3r * tr - w3i * ti;
    v[3].i = w3r * ti + w3i * tr;
}

template <bool INVERSE>
__device__ __forceinline__ void bfly4_close_x2(Cx2 v[4]) {
    const v2f x0r = v[0].r + v[1].r, x1r = v[0].r - v[1].r, x2r = v[2].r + v[3].r, x2i = v[2].i + v[3].i, x3r = v[2].r - v[3].r,
              x3i = v[2].i - v[3].i;
    if constexpr (!INVERSE) {
        const v2f x0i = v[0].i + v[1].i, x1i = v[0].i - v[1].i;
        v[0] = Cx2{x0r + x2r, x0i + x2i};
        v[2] = Cx2{x0r - x2r, x0i - x2i};
        v[1] = Cx2{x1r - x3i, x1i + x3r};
        v[3] = Cx2{x1r + x3i, x1i - x3r};
    } else {
        const v2f x0i = -v[0].i - v[1].i, x1i = -v[0].i + v[1].i;
        v[0] = Cx2{x0r + x2r, x0i - x2i};
        v[2] = Cx2{x0r - x2r, x0i + x2i};
        v[1] = Cx2{x1r - x3i, x1i - x3r};
        v[3] = Cx2{x1r + x3i, x1i + x3r};
    }
}

template <int XA, int XB>
__device__ __forceinline__ void transpose4_x2(Cx2 v[4]) {
    xstage<XA>(v[0].r, v[1].r);
    xstage<XA>(v[0].i, v[1].i);
    xstage<XA>(v[2].r, v[3].r);
    xstage<XA>(v[2].i, v[3].i);
    xstage<XB>(v[0].r, v[2].r);
    xstage<XB>(v[0].i, v[2].i);
    xstage<XB>(v[1].r, v[3].r);
    xstage<XB>(v[1].i, v[3].i);
}

// in: v[m] = point fft64_src_point(gl, m) of both transforms; out: v[m] = point gl + 16 m
template <bool INVERSE>
__device__ __forceinline__ void fft64_regs_x2(Cx2 v[4], const FftTables *T, int gl) {
    bfly4_x2(gl, T, v);
    transpose4_x2<1, 2>(v);
    bfly4_x2(gl >> 2, T, v);
    transpose4_x2<4, 8>(v);
    bfly4_close_x2<INVERSE>(v);
}

// ---------------------------------------------------------------------------------------------------------------
// (64 - p) & 63, kept from the optimiser: scaled to a byte offset it would become (p * -8) & 0x1f8 with a full 32-bit
// multiply (quarter rate) where a subtraction and a shift do
__device__ __forceinline__ int mirror64(int p) {
    int q = (64 - p) & 63;
    asm("" : "+v"(q));
    return q;
}

// Real-split pre/post processing without selects.
__device__ __forceinline__ v2f ld_pt(const float *row, int p) {
    const float2 v = *reinterpret_cast<const float2 *>(row + 2 * p);
    return v2f{v.x, v.y};
}
__device__ __forceinline__ void st_pt(float *row, int p, v2f c) { *reinterpret_cast<float2 *>(row + 2 * p) = make_float2(c.x, c.y); }

// rdft128_inv_point without selects: point p = rev4(b) + {0, 32, 16, 48}[M] of the complex array the inverse passes
// start from.  With own = point p, other = point 64 - p of the packed spectrum, xr = own.r - other.r and
// xi = own.i + other.i, the reference's two cases (p < 32: own is a[j]; p > 32: own is a[k], fft4g.c:1260-1284) are
//     re = own.r - (wkr*xr +- wki*xi),    im = (wkr*xi -+ wki*xr) - own.i          (upper signs: p < 32)
// -- for p > 32 the reference forms xr with the other sign and adds; negating a product or a difference is exact, so
// these are the same floats.  M fixes the case at compile time (M odd <=> p >= 32); p == 32 (own == other) runs with
// zero coefficients, p == 0 (the a[1] fix-up, fft4g.c:349-351) is selected in.
template <int M>
__device__ __forceinline__ v2f rdft128_inv_point_m(const float *a, const FftTables *T, int b) {
    constexpr int OFF = M == 0 ? 0 : (M == 1 ? 32 : (M == 2 ? 16 : 48));
    const int p = dev_bitrev(b, 4) + OFF;
    const v2f own = ld_pt(a, p), other = ld_pt(a, mirror64(p));
    const int q = (M & 1) ? 64 - p : p;
    float wr = 0.5f - T->c[32 - q], wi = T->c[q];
    if constexpr (M == 1) {
        wr = b == 0 ? 0.f : wr;  // p == 32: re = a[64], im = -a[65]
        wi = b == 0 ? 0.f : wi;
    }
    const v2f x = own + v2f{-other.x, other.y};
    const v2f y = (M & 1) ? cmul_w(wr, wi, x) : cmul_w(wr, -wi, x);  // (wr*xr -+ wi*xi, wr*xi +- wi*xr)
    v2f out = conj_add(own, y);  // {own.x - y.x, -own.y + y.y}
    if constexpr (M == 0) {
        const float h = 0.5f * (own.x - own.y);
        out = b == 0 ? v2f{own.x - h, -h} : out;
    }
    return out;
}

// the same point of two rows, as a packed pair for fft64_regs_x2
__device__ __forceinline__ Cx2 rdft128_inv_point_x2(const float *a0, const float *a1, const FftTables *T, int b, int m) {
    v2f u, w;
    switch (m) {  // m is a compile-time constant at every call site (unrolled loops)
        case 0: u = rdft128_inv_point_m<0>(a0, T, b), w = rdft128_inv_point_m<0>(a1, T, b); break;
        case 1: u = rdft128_inv_point_m<1>(a0, T, b), w = rdft128_inv_point_m<1>(a1, T, b); break;
        case 2: u = rdft128_inv_point_m<2>(a0, T, b), w = rdft128_inv_point_m<2>(a1, T, b); break;
        default: u = rdft128_inv_point_m<3>(a0, T, b), w = rdft128_inv_point_m<3>(a1, T, b); break;
    }
    return Cx2{v2f{u.x, w.x}, v2f{u.y, w.y}};
}

// Output side without selects (rftfsub + the a[0]/a[1] fix-up, fft4g.c:340-342,1234-1257): bin == lane of the packed
// row `a` holding the result of the forward complex passes.  With own = point lane, other = point 64 - lane:
//     (re, im) = own - (wr*xr - wi*xi, wr*xi + wi*xr),    wr = wkr, wi = +-wki (- for lane > 32)
// lane 32 runs with wr = wi = 0 (re = a[64], im = a[65]); lane 0 reads itself as `other` and runs with wr = 0,
// wi = 0.5: re = a[0] - (0 - 0.5*(a[1] + a[1])) = a[0] + a[1], and z = 0 clears its imaginary part (bin 0 is real;
// a zero of either sign).  Bin 64 = a[0] - a[1] is left to the caller.
struct SplitLane {
    float wr, wi, z;
};
__device__ __forceinline__ SplitLane rdft128_fwd_coef(const FftTables *T, int lane) {
    const int q = lane < 32 ? lane : 64 - lane;
    float wr = 0.5f - T->c[32 - q], wi = T->c[q];
    wi = lane < 32 ? wi : -wi;
    wr = (lane & 31) == 0 ? 0.f : wr;
    wi = lane == 0 ? 0.5f : (lane == 32 ? 0.f : wi);
    return SplitLane{wr, wi, lane == 0 ? 0.f : 1.f};
}
__device__ __forceinline__ v2f rdft128_fwd_bin_u(const float *a, SplitLane s, int lane) {
    const v2f own = ld_pt(a, lane), other = ld_pt(a, mirror64(lane));
    const v2f x = own + v2f{-other.x, other.y};
    return (own - cmul_w(s.wr, s.wi, x)) * v2f{1.f, s.z};
}

// ---------------------------------------------------------------------------------------------------------------
// The same transform with ONE point per lane (lane l holds point l before and after every pass) for the places
// where a single transform is on the critical path and the 16-lane form would leave three quarters of the wave
// idle.  Each lane fetches the four inputs of its butterfly with ds_bpermute (LDS crossbar, no VALU, no memory) and
// evaluates only its own output j of it:
//     xa = A (+/-) B, xb = C (+/-) D            (- for odd j)
//     t  = xa (+/-) xb      for even j,   t = xa (+/-) i*xb   for odd j      (- for j >= 2)
//     out = t, or W_j[b] * t in a twiddled block, or the reference's special forms in block b == 1
// which is operand for operand what bfly4 / bfly4_close compute for output j (x +/- y is written fma(+/-1, y, x):
// the product is exact, one rounding, the same value as the add / subtract).  The conjugating inverse closing pass
// (fft4g.c:963-984) conjugates A and B on the way in and xb on the way out.
__device__ __forceinline__ v2f lane_fetch(v2f v, int src_lane) {
    return v2f{__int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v.x))),
               __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v.y)))};
}

// KIND 0: twiddled pass (block class b), 1: closing forward, 2: closing inverse
template <int KIND>
__device__ __forceinline__ v2f bfly4_lane(int j, int b, const FftTables *T, v2f A, v2f B, v2f C, v2f D) {
    const float s1 = (j & 1) ? -1.f : 1.f, s2 = (j & 2) ? -1.f : 1.f;
    if constexpr (KIND == 2) {
        A.y = -A.y;
        B.y = -B.y;
    }
    const v2f xa = v2f{fmaf(s1, B.x, A.x), fmaf(s1, B.y, A.y)};
    const v2f xb = v2f{fmaf(s1, D.x, C.x), fmaf(s1, D.y, C.y)};
    v2f r;  // what is added to / subtracted from xa
    if constexpr (KIND == 2)
        r = (j & 1) ? v2f{-xb.y, -xb.x} : v2f{xb.x, -xb.y};
    else
        r = (j & 1) ? v2f{-xb.y, xb.x} : xb;
    v2f t = v2f{fmaf(s2, r.x, xa.x), fmaf(s2, r.y, xa.y)};
    if constexpr (KIND != 0) return t;
    // twiddle of output j in block b: W_0 = 1; blocks 0 and 1 come out of the table like any other (bfly4_v,
    // fft_ooura.h), block 1 with its outputs 1 and 3 rotated first
    const float ms = (b == 1 && (j & 1)) ? (j == 1 ? 1.f : -1.f) : 0.f;
    t = __builtin_elementwise_fma(v2f{-ms, ms}, swap(t), t);  // ms is 0 or +-1: exact product, one rounding (see bfly4_v)
    // W1, W2, W3 lie behind one another (FftTables): entry (j - 1) * 32 + b of the three as one array, entry 0 (= W1[0]) for
    // j == 0 -- an index computation; as a select among four pointers it compiles to nested predicated regions, a dozen scalar
    // instructions per butterfly
    static_assert(offsetof(FftTables, W2) == offsetof(FftTables, W1) + 32 * 2 * sizeof(float) &&
                      offsetof(FftTables, W3) == offsetof(FftTables, W1) + 64 * 2 * sizeof(float),
                  "W1 | W2 | W3 contiguous");
    const float *wp = &T->W1[0][0] + 2 * (j == 0 ? 0 : (j - 1) * 32 + b);
    return cmul_w(wp[0], wp[1], t);
}

template <bool INVERSE>
__device__ __forceinline__ v2f fft64_lanes(v2f pt, const FftTables *T, int lane) {
    {  // pass 1: butterfly lane>>2 gathers the bit-reversed points rev4(bt) + {0, 32, 16, 48}
        const int bt = lane >> 2, r0 = dev_bitrev(bt, 4);
        const v2f A = lane_fetch(pt, r0), B = lane_fetch(pt, r0 + 32), C = lane_fetch(pt, r0 + 16), D = lane_fetch(pt, r0 + 48);
        pt = bfly4_lane<0>(lane & 3, bt, T, A, B, C, D);
    }
    {  // pass 2: stride 4
        const int base = lane & 0x33;
        const v2f A = lane_fetch(pt, base), B = lane_fetch(pt, base + 4), C = lane_fetch(pt, base + 8), D = lane_fetch(pt, base + 12);
        pt = bfly4_lane<0>((lane >> 2) & 3, lane >> 4, T, A, B, C, D);
    }
    {  // closing pass: stride 16, no twiddles
        const int base = lane & 15;
        const v2f A = lane_fetch(pt, base), B = lane_fetch(pt, base + 16), C = lane_fetch(pt, base + 32), D = lane_fetch(pt, base + 48);
        pt = bfly4_lane<INVERSE ? 2 : 1>(lane >> 4, 0, T, A, B, C, D);
    }
    return pt;
}

// rdft128_inv_point for point == lane with the packed spectrum in registers: `own` = bin lane (lane 0: (a[0], a[1]) =
// (bin 0, bin 64), both real), partner = bin 64 - lane fetched from its lane.
__device__ __forceinline__ v2f rdft128_inv_point_lanes(v2f own, const FftTables *T, int lane) {
    const v2f other = lane_fetch(own, mirror64(lane));
    const int q = lane < 32 ? lane : 64 - lane;
    const float wkr = 0.5f - T->c[32 - q], wki = T->c[q];
    const v2f aj = lane < 32 ? own : other, ak = lane < 32 ? other : own;  // pair (j = 2q, k = 128 - 2q)
    const float xr = aj.x - ak.x, xi = aj.y + ak.y;
    const float yr = wkr * xr + wki * xi, yi = wkr * xi - wki * xr;
    float re = lane < 32 ? aj.x - yr : ak.x + yr;
    float im = lane < 32 ? yi - aj.y : yi - ak.y;
    const float h = 0.5f * (own.x - own.y);  // lane 0: a[0] = own.x, a[1] = own.y
    re = lane == 0 ? own.x - h : (lane == 32 ? own.x : re);
    im = lane == 0 ? -h : (lane == 32 ? -own.y : im);
    return v2f{re, im};
}

// value held by the lane with index rev4(gl) of the same transform (ds_bpermute: no LDS memory, no barrier)
__device__ __forceinline__ float row_bitrev(float v, int lane) {
    const int src = (dev_bitrev(lane >> 2, 4) << 2) | (lane & 3);
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v)));
}

}  // namespace wmx
