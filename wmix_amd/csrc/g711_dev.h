// g711_dev.h -- the G.711 companding arithmetic as device functions (shared by g711.hip and rtp.hip).
// The reference's 16-bit-domain Sun variant, src/g711codec.c:12-152 (see g711.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace wmx {

__device__ __forceinline__ int seg_of(int v) {
    // first i with v <= (0x100<<i)-1, i in 0..7, else 8 (src/g711codec.c:12-22)
    if (v <= 0xFF) return 0;
    int s = 24 - __clz(v);  // bit_length(v) - 8
    return s > 8 ? 8 : s;
}

__device__ __forceinline__ unsigned enc_alaw(int pcm) {
    // src/g711codec.c:82-114
    unsigned mask = 0xD5;
    if (pcm < 0) {
        mask = 0x55;
        pcm = -pcm - 8;
    }
    int seg = seg_of(pcm);
    if (seg >= 8) return 0x7F ^ mask;
    unsigned a = (unsigned)seg << 4;
    a |= (unsigned)(pcm >> (seg < 2 ? 4 : seg + 3)) & 0xF;
    return (a ^ mask) & 0xFF;
}

__device__ __forceinline__ unsigned enc_ulaw(int pcm) {
    // src/g711codec.c:120-152
    unsigned mask;
    if (pcm < 0) {
        pcm = 0x84 - pcm;
        mask = 0x7F;
    } else {
        pcm += 0x84;
        mask = 0xFF;
    }
    int seg = seg_of(pcm);
    if (seg >= 8) return 0x7F ^ mask;
    unsigned u = ((unsigned)seg << 4) | ((unsigned)(pcm >> (seg + 3)) & 0xF);
    return (u ^ mask) & 0xFF;
}

__device__ __forceinline__ int dec_alaw(unsigned a) {
    // src/g711codec.c:28-51
    a ^= 0x55;
    int t = (int)(a & 0xF) << 4;
    int seg = (int)(a & 0x70) >> 4;
    if (seg == 0)
        t += 8;
    else
        t = (t + 0x108) << (seg - 1);
    return (a & 0x80) ? t : -t;
}

__device__ __forceinline__ int dec_ulaw(unsigned u) {
    // src/g711codec.c:62-76
    u = ~u & 0xFF;
    int t = ((int)(u & 0xF) << 3) + 0x84;
    t <<= (u & 0x70) >> 4;
    return (u & 0x80) ? (0x84 - t) : (t - 0x84);
}

}  // namespace wmx
