// spl_fx.h -- device-side pieces shared by the fixed-point kernels (nsx.hip, aecm.hip): wave reductions, the SPL
// primitives they both use, and the SPL radix-2 fixed-point complex FFT run across one wavefront.
//   W:common_audio/signal_processing/complex_fft.c:30-296 (mode 1), complex_bit_reverse.c, real_fft.c:46-100,
//   spl_sqrt_floor.c:48-75, include/spl_inl.h:144-164 (NormW16), include/signal_processing_library.h:73-75
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
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
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return (uint32_t)uni((int)v);
}
__device__ __forceinline__ int32_t wave_max(int32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return uni(v);
}
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t w = (uint32_t)__shfl_xor((int)v, o, 64);
        v = w > v ? w : v;
    }
    return (uint32_t)uni((int)v);
}
__device__ __forceinline__ int32_t wave_min(int32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return uni(v);
}
__device__ __forceinline__ int wave_any(int p) { return __builtin_amdgcn_ballot_w64(p != 0) != 0; }

// spl_sqrt_floor.c:48-75: floor(sqrt(value)) for value >= 0, 0 for a negative argument
__device__ __forceinline__ int32_t sqrt_floor(int32_t value) {
    int32_t root = 0;
#pragma unroll
    for (int n = 15; n >= 0; n--) {
        const int32_t t = wshl(root + (1 << n), n);
        if (value >= t) {
            value -= t;
            root |= 2 << n;
        }
    }
    return root >> 1;
}
__device__ __forceinline__ int16_t lo16(int32_t w) { return (int16_t)(w & 0xffff); }
__device__ __forceinline__ int16_t hi16(int32_t w) { return (int16_t)(w >> 16); }
__device__ __forceinline__ int32_t pack16(int16_t lo, int16_t hi) { return (int32_t)((uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16)); }

// ---------------------------------------------------------------- SPL complex FFT across the wave (complex_fft.c mode 1)
// cx holds N = 1 << STAGES packed complex points in bit-reversed order on entry.  INVERSE: returns the number of
// one-bit shifts the data-dependent scaling applied (WebRtcSpl_ComplexIFFT's return value).
template <int STAGES, bool INVERSE>
__device__ int spl_cfft(int32_t *cx, const int16_t *sin1024, int lane) {
    constexpr int N = 1 << STAGES, PER = N / 128 > 0 ? N / 128 : 1;  // butterflies per lane and stage
    int scale = 0;
#pragma unroll 1
    for (int s = 0; s < STAGES; s++) {
        const int l = 1 << s, k = 9 - s;
        int shift = INVERSE ? 0 : 1;
        int32_t round2 = INVERSE ? 8192 : 16384;
        if (INVERSE) {
            int32_t mx = 0;
            for (int i = lane; i < N; i += 64) {
                const int32_t w = cx[i];
                int a = lo16(w), b = hi16(w);
                a = a < 0 ? -a : a;
                b = b < 0 ? -b : b;
                mx = a > mx ? a : mx;
                mx = b > mx ? b : mx;
            }
            mx = wave_max(mx);
            if (mx > 32767) mx = 32767;
            if (mx > 13573) shift++, scale++, round2 <<= 1;
            if (mx > 27146) shift++, scale++, round2 <<= 1;
        }
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const int b = lane + 64 * r;
            if (N >= 128 || b < N / 2) {
                const int m = b & (l - 1), i = ((b >> s) << (s + 1)) + m, j = i + l, j0 = m << k;
                const int16_t wr = sin1024[j0 + 256], wi = (int16_t)(INVERSE ? sin1024[j0] : -sin1024[j0]);
                const int32_t xi = cx[i], xj = cx[j];
                const int32_t jr = lo16(xj), ji = hi16(xj);
                const int32_t tr = (wr * jr - wi * ji + 1) >> 1, ti = (wr * ji + wi * jr + 1) >> 1;
                const int32_t qr = (int32_t)lo16(xi) << 14, qi = (int32_t)hi16(xi) << 14;
                cx[j] = pack16((int16_t)((qr - tr + round2) >> (shift + 14)), (int16_t)((qi - ti + round2) >> (shift + 14)));
                cx[i] = pack16((int16_t)((qr + tr + round2) >> (shift + 14)), (int16_t)((qi + ti + round2) >> (shift + 14)));
            }
        }
        wave_sync();
    }
    return scale;
}
template <int STAGES>
__device__ __forceinline__ int bitrev(int i) { return (int)(__brev((unsigned)i) >> (32 - STAGES)); }


}  // namespace wmx
