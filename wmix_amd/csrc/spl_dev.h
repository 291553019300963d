// spl_dev.h -- device-side integer helpers with the exact semantics of the WebRTC
// signal_processing primitives the VAD / AGC paths use (two's-complement wrap made explicit).
//   NormW32 / NormU32 / GetSizeInBits   W:common_audio/signal_processing/include/spl_inl.h:88-141
//   DivW32W16                            W:.../division_operations.c:38-47
//   WebRtcSpl_Sqrt / SqrtLocal           W:.../spl_sqrt.c
//   WEBRTC_SPL_SCALEDIFF32               W:.../include/signal_processing_library.h:78
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace wmx {

__device__ __forceinline__ int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
// -a with INT32_MIN staying INT32_MIN (what the reference's `-x` does on the hardware it runs on; as `-a` it is undefined, and the
// compiler may use that)
__device__ __forceinline__ int32_t wneg(int32_t a) { return (int32_t)(0u - (uint32_t)a); }
__device__ __forceinline__ int32_t wshl(int32_t a, int s) { return (int32_t)((uint32_t)a << s); }
__device__ __forceinline__ int32_t shift_w32(int32_t x, int c) { return c >= 0 ? wshl(x, c) : (x >> (-c)); }
__device__ __forceinline__ int16_t sat_w16(int32_t v) { return (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v)); }

__device__ __forceinline__ int norm_w32(int32_t a) {
    if (a == 0) return 0;
    if (a < 0) a = ~a;
    return a == 0 ? 31 : __clz(a) - 1;
}
__device__ __forceinline__ int norm_u32(uint32_t a) { return a == 0 ? 0 : __clz((int)a); }
__device__ __forceinline__ int size_in_bits(uint32_t n) { return n == 0 ? 0 : 32 - __clz((int)n); }
__device__ __forceinline__ int32_t div_w32_w16(int32_t num, int16_t den) {
    return den != 0 ? (int32_t)(num / (int32_t)den) : (int32_t)0x7FFFFFFF;
}

// digital_agc.h:22,24
__device__ __forceinline__ int32_t agc_scalediff32(int32_t A, int32_t B, int32_t C) {
    return wadd(wadd(C, wmul(B >> 16, A)), wmul(0x0000FFFF & B, A) >> 16);
}
__device__ __forceinline__ int32_t agc_mul32(int32_t A, int32_t B) { return wadd(wmul(B >> 13, A), wmul(0x00001FFF & B, A) >> 13); }
// signal_processing_library.h:78 (low half multiplied as uint32)
__device__ __forceinline__ int32_t spl_scalediff32(int32_t A, int32_t B, int32_t C) {
    return (int32_t)((uint32_t)C + (uint32_t)wmul(B >> 16, A) + (((uint32_t)(0x0000FFFF & B) * (uint32_t)A) >> 16));
}

__device__ inline int32_t spl_sqrt_local(int32_t in) {
    int32_t B = in / 2, A, x2;
    B = wsub(B, 0x40000000);
    const int16_t x_half = (int16_t)(B >> 16);
    int16_t t16;
    B = wadd(B, 0x40000000);
    B = wadd(B, 0x40000000);
    x2 = wmul(wmul(x_half, x_half), 2);
    A = wneg(x2);  // x_half == -32768: x2 wraps to INT32_MIN and stays there
    B = wadd(B, A >> 1);
    A >>= 16;
    A = wmul(wmul(A, A), 2);
    t16 = (int16_t)(A >> 16);
    B = wadd(B, wmul(-20480 * t16, 2));
    A = wmul(x_half * t16, 2);
    t16 = (int16_t)(A >> 16);
    B = wadd(B, wmul(28672 * t16, 2));
    t16 = (int16_t)(x2 >> 16);
    A = wmul(x_half * t16, 2);
    B = wadd(B, A >> 1);
    B = wadd(B, 32768);
    return B;
}

__device__ inline int32_t spl_sqrt(int32_t value) {
    int32_t A = value;
    if (A == 0) return 0;
    const int16_t sh = (int16_t)norm_w32(A);
    A = wshl(A, sh);
    if (A < (0x7FFFFFFF - 32767))
        A = A + 32768;
    else
        A = 0x7FFFFFFF;
    const int16_t x_norm = (int16_t)(A >> 16), nshift = (int16_t)(sh / 2);
    A = wshl((int32_t)x_norm, 16);
    A = A >= 0 ? A : wneg(A);
    A = spl_sqrt_local(A);
    if (2 * nshift == sh) {
        const int16_t t16 = (int16_t)(A >> 16);
        A = wmul(23170 * t16, 2);
        A = wadd(A, 32768);
        A = A & 0x7fff0000;
        A >>= 15;
    } else {
        A >>= 16;
    }
    A = A & 0x0000ffff;
    A >>= nshift;
    return A;
}

}  // namespace wmx
