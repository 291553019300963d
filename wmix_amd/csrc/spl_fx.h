// spl_fx.h -- device-side pieces shared by the fixed-point kernels (nsx.hip, aecm.hip): wave reductions, the SPL
// primitives they both use, and the SPL radix-2 fixed-point complex FFT run across one wavefront.
//   W:common_audio/signal_processing/complex_fft.c:30-296 (mode 1), complex_bit_reverse.c, real_fft.c:46-100,
//   spl_sqrt_floor.c:48-75, include/spl_inl.h:144-164 (NormW16), include/signal_processing_library.h:73-75
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
#include "spl_dev.h"
#include "fft_ooura.h"  // wave_sync()

namespace wmx {

__device__ __forceinline__ int norm_w16(int16_t a) {
    if (a == 0) return 0;
    const int v = a < 0 ? (int16_t)~a : a;
    return v ? __clz(v) - 17 : 15;
}
__device__ __forceinline__ int32_t mul_rsft_round(int16_t a, int16_t b, int c) { return ((int32_t)a * b + ((int32_t)1 << (c - 1))) >> c; }
__device__ __forceinline__ uint32_t div_u32_u16(uint32_t num, uint16_t den) { return den ? num / den : 0xFFFFFFFFu; }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Wave reductions on the DPP path: four in-row steps (lane ^ 1, lane ^ 2, the two mirrors -- after each step the lanes
// that are combined next already agree, so mirrors serve as the xor 4 / xor 8 exchanges) leave every 16-lane row holding
// its row's result; the four rows meet in scalar registers.  No LDS-crossbar traffic (`__shfl_xor` = ds_bpermute_b32, six
// dependent round trips per reduction), and the result is wave-uniform by construction.  Integer sums wrap, max / min are
// exact: the order of combination does not matter.
// For wave-uniform control flow only (every lane on): a lane whose source lane is switched off would read 0.  `old` = 0 is
// what lets the compiler fold the move into the consuming instruction (v_add_u32_dpp / v_max_i32_dpp ... bound_ctrl:1); with
// `old` = v it stays a separate v_mov_b32_dpp and the two kernels lose 1-5 %.
template <int CTRL>
__device__ __forceinline__ int dpp_rows(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
// (the exchanged value goes into a temporary first: written as `dpp(v) > v ? dpp(v) : v` the compiler evaluates the
// second DPP move under the comparison's lane mask, where a disabled source lane leaves the destination untouched)
#define WMX_ROW_REDUCE(v, OP)                                                          \
    do {                                                                               \
        decltype(v) o_;                                                                \
        o_ = (decltype(v))dpp_rows<0xB1>((int)v), v = OP(v, o_);  /* quad_perm [1,0,3,2] */ \
        o_ = (decltype(v))dpp_rows<0x4E>((int)v), v = OP(v, o_);  /* quad_perm [2,3,0,1] */ \
        o_ = (decltype(v))dpp_rows<0x141>((int)v), v = OP(v, o_); /* row_half_mirror */     \
        o_ = (decltype(v))dpp_rows<0x140>((int)v), v = OP(v, o_); /* row_mirror */          \
    } while (0)
#define WMX_OP_ADD(a, b) ((a) + (b))
#define WMX_OP_MAX(a, b) ((b) > (a) ? (b) : (a))
#define WMX_OP_MIN(a, b) ((b) < (a) ? (b) : (a))
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    WMX_ROW_REDUCE(v, WMX_OP_ADD);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return (r0 + r1) + (r2 + r3);
}
// true in every lane if the predicate holds in any lane (all 64 lanes on)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ int32_t wave_max(int32_t v) {
    WMX_ROW_REDUCE(v, WMX_OP_MAX);
    const int32_t r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int32_t r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    const int32_t a = r1 > r0 ? r1 : r0, b = r3 > r2 ? r3 : r2;
    return b > a ? b : a;
}
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) {
    WMX_ROW_REDUCE(v, WMX_OP_MAX);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t a = r1 > r0 ? r1 : r0, b = r3 > r2 ? r3 : r2;
    return b > a ? b : a;
}
__device__ __forceinline__ uint32_t wave_umin(uint32_t v) {
    WMX_ROW_REDUCE(v, WMX_OP_MIN);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t a = r1 < r0 ? r1 : r0, b = r3 < r2 ? r3 : r2;
    return b < a ? b : a;
}
__device__ __forceinline__ int32_t wave_min(int32_t v) {
    WMX_ROW_REDUCE(v, WMX_OP_MIN);
    const int32_t r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int32_t r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    const int32_t a = r1 < r0 ? r1 : r0, b = r3 < r2 ? r3 : r2;
    return b < a ? b : a;
}
__device__ __forceinline__ int wave_any(int p) { return __builtin_amdgcn_ballot_w64(p != 0) != 0; }

// WebRtcSpl_SqrtFloor (spl_sqrt_floor.c:48-75) = floor(sqrt(value)) for value > 0 and 0 otherwise: the reference's 16
// successive-approximation steps are replaced by the float square root and one correction each way.  float(value) and
// sqrtf are each good to 2^-24 relative, so the estimate is within 0.01 of the true root (< 46 341): after truncation it
// is the floor or one off, and the two exact integer comparisons settle it.
__device__ __forceinline__ int32_t sqrt_floor(int32_t value) {
    if (value <= 0) return 0;
    const uint32_t v = (uint32_t)value;
    uint32_t r = (uint32_t)__fsqrt_rn((float)v);
    r -= (r * r > v) ? 1u : 0u;
    r += ((r + 1) * (r + 1) <= v) ? 1u : 0u;
    return (int32_t)r;
}
__device__ __forceinline__ int16_t lo16(int32_t w) { return (int16_t)(w & 0xffff); }
__device__ __forceinline__ int16_t hi16(int32_t w) { return (int16_t)(w >> 16); }
__device__ __forceinline__ int32_t pack16(int16_t lo, int16_t hi) { return (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16)); }

// ---------------------------------------------------------------- a stream's scalar state, resident in LDS
// The scalars of a stream live in its LDS copy of the state block and are fetched where they are used (one broadcast LDS
// read + readfirstlane).  Held in registers for the whole kernel, a few dozen of them plus the kernel's pointers exceed the
// SGPR file and every use turns into v_readlane / v_writelane spill traffic (half of the AECM kernel's instructions when
// measured).  Every lane stores the same value; reads are wave-uniform.
struct LdsScalRef {
    int32_t *p;
    __device__ __forceinline__ operator int32_t() const { return uni(*p); }
    __device__ __forceinline__ LdsScalRef &operator=(int32_t v) {
        *p = v;
        return *this;
    }
    __device__ __forceinline__ LdsScalRef &operator=(const LdsScalRef &o) { return *this = (int32_t)o; }
    __device__ __forceinline__ void operator++(int) { *p = uni(*p) + 1; }
    __device__ __forceinline__ void operator--(int) { *p = uni(*p) - 1; }
    __device__ __forceinline__ void operator+=(int32_t v) { *p = uni(*p) + v; }
    __device__ __forceinline__ void operator-=(int32_t v) { *p = uni(*p) - v; }
};
struct LdsScal {
    int32_t *base;
    __device__ __forceinline__ LdsScalRef operator[](int k) const { return LdsScalRef{base + k}; }
};

// ---------------------------------------------------------------- SPL complex FFT across the wave (complex_fft.c mode 1)
// Twiddles as packed int16 pairs, so that a butterfly's two rotated components are one dot-product instruction each on
// the packed point (re | im << 16):  tr = wr*re - wi*im + 1 = dot2(x, A) + 1,  ti = wr*im + wi*re + 1 = dot2(x, B) + 1
// with A = (wr, -wi), B = (wi, wr); wr = kSinTable1024[j + 256], wi = -/+ kSinTable1024[j] (forward / inverse).  Entry q
// is table position j = 4 q (a 256-point transform uses every entry, a 128-point one every second).  Only the forward
// pair is stored: with s = kSinTable1024[j] it is A = (wr, s), B = (-s, wr), and the inverse pair A' = (wr, -s),
// B' = (s, wr) is B and A with their halves exchanged (one v_alignbit each).
// One word of padding follows every 32 entries: stage s reads entries m << (7 - s), a stride that is a multiple of the
// bank count in the early stages -- with the padding the distinct entries of a stage fall into distinct banks.
struct SplTwiddles {
    static constexpr int kEntries = 128 + 128 / 32;
    int32_t a[kEntries], b[kEntries];
    __host__ __device__ static constexpr int slot(int q) { return q + (q >> 5); }
};
inline void spl_twiddles(const int16_t *sin1024, SplTwiddles *t) {
    for (int i = 0; i < SplTwiddles::kEntries; i++) t->a[i] = t->b[i] = 0;
    for (int q = 0; q < 128; q++) {
        const int16_t wr = sin1024[4 * q + 256], sn = sin1024[4 * q];
        t->a[SplTwiddles::slot(q)] = (int32_t)((uint32_t)(uint16_t)wr | ((uint32_t)(uint16_t)sn << 16));
        t->b[SplTwiddles::slot(q)] = (int32_t)((uint32_t)(uint16_t)(int16_t)-sn | ((uint32_t)(uint16_t)wr << 16));
    }
}

typedef short spl_v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int32_t dot2_i16(int32_t x, int32_t w, int32_t c) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(spl_v2s, x), __builtin_bit_cast(spl_v2s, w), c, false);
}
// packed |x| of both int16 halves with -32768 kept as 0x8000, which an UNSIGNED comparison ranks above every other
// magnitude -- what WebRtcSpl_MaxAbsValueW16's abs() + cap at 32767 needs for the two thresholds it is compared with
__device__ __forceinline__ uint32_t pk_abs16(int32_t x) {
    uint32_t n, r;
    asm("v_pk_sub_i16 %0, 0, %1" : "=v"(n) : "v"(x));
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(n));
    return r;
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// cx holds N = 1 << STAGES packed complex points in bit-reversed order on entry; one radix-2 pass per stage, every output
// rounded to int16 exactly as the reference does (the stages cannot be merged).  Each lane owns butterfly `lane` (and
// lane + 64 when N = 256) of every stage.  INVERSE: the stage's extra shift (0..2 bits) comes from the largest |value| of
// the whole array (complex_fft.c:170-186): found once by a scan before the first stage, afterwards carried along from
// the values each lane has just produced.  Returns the number of one-bit shifts applied (WebRtcSpl_ComplexIFFT's result).
template <int STAGES, bool INVERSE, bool UNROLL = false>
__device__ int spl_cfft(int32_t *cx, const SplTwiddles &T, int lane) {
    constexpr int N = 1 << STAGES, PER = N / 128 > 0 ? N / 128 : 1;  // butterflies per lane and stage
    int scale = 0;
    uint32_t mag = 0;  // packed running maximum of |re|, |im| of this lane's values (INVERSE only)
    if (INVERSE) {
        for (int i = lane; i < N; i += 64) mag = pk_max_u16(mag, pk_abs16(cx[i]));
    }
    // UNROLL: the stages as straight-line code (shift counts and strides become immediates).  Worth 6 % to the AECM block
    // (three 128-point transforms, +6 KB of code) once its kernel had shrunk to 26 KB; at 70 KB the same unrolling pushed it
    // further past the 64 KB instruction cache and cost 20 %; the 256-point NSX transforms (two butterflies per lane) spill.
    constexpr int kUnroll = UNROLL ? STAGES : 1;
#pragma unroll kUnroll
    for (int s = 0; s < STAGES; s++) {
        const int l = 1 << s;
        int shift = INVERSE ? 0 : 1;
        int32_t round2 = INVERSE ? 8192 : 16384;
        if (INVERSE) {
            // the two thresholds on the largest |value| of the array (complex_fft.c:170-186; its cap at 32767 lies above both):
            // "any lane above" is two compares into lane masks -- the wave-wide maximum itself (a dependent chain of four DPP
            // steps, four v_readlane and scalar maxima, once per stage) is never needed
            const uint32_t m16 = (mag & 0xffffu) > (mag >> 16) ? (mag & 0xffffu) : (mag >> 16);
            if (wave_any(m16 > 13573u)) shift++, scale++, round2 <<= 1;
            if (wave_any(m16 > 27146u)) shift++, scale++, round2 <<= 1;
            mag = 0;
        }
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const int b = lane + 64 * r;
            if (N >= 128 || b < N / 2) {
                const int m = b & (l - 1), i = 2 * b - m, j = i + l;  // i = (b >> s << (s + 1)) + m
                const int q = m << (7 - s);                            // table position j0 / 4, j0 = m << (9 - s)
                const int32_t xi = cx[i], xj = cx[j];
                const int32_t fa = T.a[SplTwiddles::slot(q)], fb = T.b[SplTwiddles::slot(q)];
                const int32_t wa = INVERSE ? (int32_t)__builtin_amdgcn_alignbit((uint32_t)fb, (uint32_t)fb, 16) : fa;
                const int32_t wb = INVERSE ? (int32_t)__builtin_amdgcn_alignbit((uint32_t)fa, (uint32_t)fa, 16) : fb;
                const int32_t tr = dot2_i16(xj, wa, 1) >> 1, ti = dot2_i16(xj, wb, 1) >> 1;
                const int32_t qr = (int32_t)((uint32_t)xi << 16) >> 2, qi = (int32_t)((uint32_t)xi & 0xffff0000u) >> 2;  // re << 14, im << 14
                const int sh = shift + 14;
                const int32_t o_j = (int32_t)__builtin_amdgcn_perm((uint32_t)((qi - ti + round2) >> sh), (uint32_t)((qr - tr + round2) >> sh), 0x05040100u);
                const int32_t o_i = (int32_t)__builtin_amdgcn_perm((uint32_t)((qi + ti + round2) >> sh), (uint32_t)((qr + tr + round2) >> sh), 0x05040100u);
                cx[j] = o_j;
                cx[i] = o_i;
                if (INVERSE && s + 1 < STAGES) mag = pk_max_u16(mag, pk_max_u16(pk_abs16(o_i), pk_abs16(o_j)));
            }
        }
        wave_sync();
    }
    return scale;
}
// ---------------------------------------------------------------- the 128-point transform without LDS between stages
// Lane L holds the two points of ITS butterfly of the current stage, (u, v) = points (i, i + 2^s) with i = L with a zero
// bit inserted at position s.  After the butterfly, the pairing of stage s + 1 is a 2 x 2 transpose between the lanes that
// differ in bit s: the lane with the bit clear keeps its u and takes the partner's u as its new v, the lane with the bit
// set keeps its v and takes the partner's v as its new u -- one v_permlane32_swap / v_permlane16_swap for lane bits 5 / 4,
// two bank-masked row rotations for bits 3 / 2, two quad permutes + two selects for bits 1 / 0.  LDS is touched once on the
// way in (one 8-byte read per lane) and once on the way out; the arithmetic is that of spl_cfft, operation for operation.
template <int BIT>
__device__ __forceinline__ void spl_xchg(int32_t &u, int32_t &v, int lane) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    if constexpr (BIT == 5) {
        const v2u r = __builtin_amdgcn_permlane32_swap((unsigned)u, (unsigned)v, false, false);
        u = (int32_t)r.x, v = (int32_t)r.y;
    } else if constexpr (BIT == 4) {
        const v2u r = __builtin_amdgcn_permlane16_swap((unsigned)u, (unsigned)v, false, false);
        u = (int32_t)r.x, v = (int32_t)r.y;
    } else if constexpr (BIT == 3) {
        const int32_t nu = __builtin_amdgcn_update_dpp(u, v, 0x128, 0xf, 0xC, false);  // row_ror:8, lanes 8-15: v of lane - 8
        const int32_t nv = __builtin_amdgcn_update_dpp(v, u, 0x128, 0xf, 0x3, false);  //            lanes 0-7:  u of lane + 8
        u = nu, v = nv;
    } else if constexpr (BIT == 2) {
        const int32_t nu = __builtin_amdgcn_update_dpp(u, v, 0x124, 0xf, 0xA, false);  // row_ror:4,  lanes 4-7, 12-15: v of lane - 4
        const int32_t nv = __builtin_amdgcn_update_dpp(v, u, 0x12C, 0xf, 0x5, false);  // row_ror:12, lanes 0-3, 8-11:  u of lane + 4
        u = nu, v = nv;
    } else {
        constexpr int kCtrl = BIT == 1 ? 0x4E : 0xB1;  // quad_perm [2,3,0,1] / [1,0,3,2]
        const int32_t pu = dpp_rows<kCtrl>(u), pv = dpp_rows<kCtrl>(v);
        const bool set = (lane >> BIT) & 1;
        u = set ? pv : u;
        v = set ? v : pu;
    }
}

template <bool INVERSE>
__device__ int spl_cfft128(int32_t *cx, const SplTwiddles &T, int lane) {
    int scale = 0;
    int32_t u, v;
    {
        const int2 uv = *reinterpret_cast<const int2 *>(cx + 2 * lane);  // bit-reversed order on entry: stage 0 pairs (2 L, 2 L + 1)
        u = uv.x, v = uv.y;
    }
    uint32_t mag = INVERSE ? pk_max_u16(pk_abs16(u), pk_abs16(v)) : 0u;
    auto stage = [&](auto S_) {
        constexpr int s = decltype(S_)::value, l = 1 << s;
        int shift = INVERSE ? 0 : 1;
        int32_t round2 = INVERSE ? 8192 : 16384;
        if (INVERSE) {
            // the two thresholds on the largest |value| of the array (complex_fft.c:170-186; its cap at 32767 lies above both):
            // "any lane above" is two compares into lane masks -- the wave-wide maximum itself (a dependent chain of four DPP
            // steps, four v_readlane and scalar maxima, once per stage) is never needed
            const uint32_t m16 = (mag & 0xffffu) > (mag >> 16) ? (mag & 0xffffu) : (mag >> 16);
            if (wave_any(m16 > 13573u)) shift++, scale++, round2 <<= 1;
            if (wave_any(m16 > 27146u)) shift++, scale++, round2 <<= 1;
        }
        const int q = (lane & (l - 1)) << (7 - s);
        const int32_t fa = T.a[SplTwiddles::slot(q)], fb = T.b[SplTwiddles::slot(q)];
        const int32_t wa = INVERSE ? (int32_t)__builtin_amdgcn_alignbit((uint32_t)fb, (uint32_t)fb, 16) : fa;
        const int32_t wb = INVERSE ? (int32_t)__builtin_amdgcn_alignbit((uint32_t)fa, (uint32_t)fa, 16) : fb;
        const int32_t tr = dot2_i16(v, wa, 1) >> 1, ti = dot2_i16(v, wb, 1) >> 1;
        const int32_t qr = (int32_t)((uint32_t)u << 16) >> 2, qi = (int32_t)((uint32_t)u & 0xffff0000u) >> 2;  // re << 14, im << 14
        const int sh = shift + 14;
        v = (int32_t)__builtin_amdgcn_perm((uint32_t)((qi - ti + round2) >> sh), (uint32_t)((qr - tr + round2) >> sh), 0x05040100u);
        u = (int32_t)__builtin_amdgcn_perm((uint32_t)((qi + ti + round2) >> sh), (uint32_t)((qr + tr + round2) >> sh), 0x05040100u);
        if (INVERSE && s < 6) mag = pk_max_u16(pk_abs16(u), pk_abs16(v));
        if constexpr (s < 6) spl_xchg<s>(u, v, lane);
    };
    stage(std::integral_constant<int, 0>{});
    stage(std::integral_constant<int, 1>{});
    stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 3>{});
    stage(std::integral_constant<int, 4>{});
    stage(std::integral_constant<int, 5>{});
    stage(std::integral_constant<int, 6>{});
    wave_sync();  // every lane's read of the input precedes the stores below (also in the compiler's eyes)
    cx[lane] = u;  // after the last stage lane L holds points L and L + 64
    cx[lane + 64] = v;
    wave_sync();
    return scale;
}

// ---------------------------------------------------------------- the 256-point transform without LDS between stages
// Four points per lane, x[k] = point (index bits) with two of the eight index bits in the register number k and six in the
// lane number.  A stage whose bit is a register bit pairs two registers of one lane: stages 2 t and 2 t + 1 run on register
// bits 0 and 1, then both register bits change places with lane bits 2 t and 2 t + 1 (spl_xchg on the register pairs that
// differ in the bit: the same 2 x 2 transposes as above, four per round, three rounds).  Entry: lane L holds points
// 4 L .. 4 L + 3 of the bit-reversed input (one 16-byte read); exit: register k of lane L is point L + 64 k.  The arithmetic
// is that of spl_cfft, operation for operation; the in-LDS version spent a third of the NSX kernel's wave time in its sixteen
// LDS round trips per frame (tools_dev/nsx_prof.py).
template <bool INVERSE>
__device__ int spl_cfft256(int32_t *cx, const SplTwiddles &T, int lane) {
    int scale = 0;
    int32_t x[4];
    {
        const int4 q = *reinterpret_cast<const int4 *>(cx + 4 * lane);
        x[0] = q.x, x[1] = q.y, x[2] = q.z, x[3] = q.w;
    }
    uint32_t mag = 0;
    if (INVERSE) mag = pk_max_u16(pk_max_u16(pk_abs16(x[0]), pk_abs16(x[1])), pk_max_u16(pk_abs16(x[2]), pk_abs16(x[3])));
    // low: the index bits below the stage's bit that live in the lane number (the first 2 t lane bits); the register bit below
    // the stage's bit joins them in the odd stages
    auto stage = [&](auto S_) {
        constexpr int s = decltype(S_)::value, rb = s & 1;  // rb: the register bit this stage pairs
        int shift = INVERSE ? 0 : 1;
        int32_t round2 = INVERSE ? 8192 : 16384;
        if (INVERSE) {
            // the two thresholds on the largest |value| of the array (complex_fft.c:170-186; its cap at 32767 lies above both):
            // "any lane above" is two compares into lane masks -- the wave-wide maximum itself (a dependent chain of four DPP
            // steps, four v_readlane and scalar maxima, once per stage) is never needed
            const uint32_t m16 = (mag & 0xffffu) > (mag >> 16) ? (mag & 0xffffu) : (mag >> 16);
            if (wave_any(m16 > 13573u)) shift++, scale++, round2 <<= 1;
            if (wave_any(m16 > 27146u)) shift++, scale++, round2 <<= 1;
            mag = 0;
        }
        const int low = lane & ((1 << (s - rb)) - 1);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            // butterfly c: registers (u, v) = (c, c + 1) for register bit 0 [c = 0, 2], (c, c + 2) for register bit 1 [c = 0, 1]
            const int ku = rb == 0 ? 2 * c : c, kv = rb == 0 ? 2 * c + 1 : c + 2;
            const int m = rb == 0 ? low : (low | (c << (s >= 1 ? s - 1 : 0)));  // index bits below s: lane part, plus register bit 0 in odd stages
            const int q = m << (7 - s);
            const int32_t fa = T.a[SplTwiddles::slot(q)], fb = T.b[SplTwiddles::slot(q)];
            const int32_t wa = INVERSE ? (int32_t)__builtin_amdgcn_alignbit((uint32_t)fb, (uint32_t)fb, 16) : fa;
            const int32_t wb = INVERSE ? (int32_t)__builtin_amdgcn_alignbit((uint32_t)fa, (uint32_t)fa, 16) : fb;
            const int32_t u = x[ku], v = x[kv];
            const int32_t tr = dot2_i16(v, wa, 1) >> 1, ti = dot2_i16(v, wb, 1) >> 1;
            const int32_t qr = (int32_t)((uint32_t)u << 16) >> 2, qi = (int32_t)((uint32_t)u & 0xffff0000u) >> 2;  // re << 14, im << 14
            const int sh = shift + 14;
            x[kv] = (int32_t)__builtin_amdgcn_perm((uint32_t)((qi - ti + round2) >> sh), (uint32_t)((qr - tr + round2) >> sh), 0x05040100u);
            x[ku] = (int32_t)__builtin_amdgcn_perm((uint32_t)((qi + ti + round2) >> sh), (uint32_t)((qr + tr + round2) >> sh), 0x05040100u);
        }
        if (INVERSE && s < 7) mag = pk_max_u16(pk_max_u16(pk_abs16(x[0]), pk_abs16(x[1])), pk_max_u16(pk_abs16(x[2]), pk_abs16(x[3])));
        if constexpr (rb == 1 && s < 7) {
            // register bit 0 <-> lane bit s - 1, register bit 1 <-> lane bit s
            spl_xchg<s - 1>(x[0], x[1], lane);
            spl_xchg<s - 1>(x[2], x[3], lane);
            spl_xchg<s>(x[0], x[2], lane);
            spl_xchg<s>(x[1], x[3], lane);
        }
    };
    stage(std::integral_constant<int, 0>{});
    stage(std::integral_constant<int, 1>{});
    stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 3>{});
    stage(std::integral_constant<int, 4>{});
    stage(std::integral_constant<int, 5>{});
    stage(std::integral_constant<int, 6>{});
    stage(std::integral_constant<int, 7>{});
    wave_sync();  // every lane's read of the input precedes the stores below (also in the compiler's eyes)
#pragma unroll
    for (int k = 0; k < 4; k++) cx[lane + 64 * k] = x[k];
    wave_sync();
    return scale;
}

template <int STAGES>
__device__ __forceinline__ int bitrev(int i) { return (int)(__brev((unsigned)i) >> (32 - STAGES)); }


}  // namespace wmx
