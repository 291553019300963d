// g711.hip -- G.711 A-law / mu-law companding on gfx950.
//
// Replaces linear2alaw / linear2ulaw / alaw2linear / ulaw2linear and the
// g711x_encode / g711x_decode / PCM2G711x / G711x2PCM loops of the reference
// (src/g711codec.c:28-308).  The arithmetic is the reference's 16-bit-domain Sun
// variant (segment ends 0xFF..0x7FFF, no 13/14-bit prescale; negative A-law input
// is mapped with -pcm-8 and may stay negative: SURVEY.md section 0 quirk 6); the segment
// search is a count-leading-zeros instead of the table walk.
//
// Roofline: pure HBM streaming, 3 B per sample each way (2 B PCM + 1 B code).
// One lane handles 8 samples: one 16-byte PCM access and one 8-byte code access
// per lane, so every wave instruction moves 1 KiB / 512 B fully coalesced.
#include "wmx_internal.h"
#include "g711_dev.h"

namespace {
using namespace wmx;

template <int LAW>
__device__ __forceinline__ unsigned enc1(int pcm) {
    return LAW == WMX_LAW_A ? enc_alaw(pcm) : enc_ulaw(pcm);
}
template <int LAW>
__device__ __forceinline__ int dec1(unsigned c) {
    return LAW == WMX_LAW_A ? dec_alaw(c) : dec_ulaw(c);
}

// Encoding through a table in LDS.  Both laws quantise a magnitude v whose low bits never reach the code:
//   mu-law: v = |pcm| + 0x84 in [0x84, 32900], code bits = f(v >> 3);   A-law: v = pcm or -pcm - 8 (which may stay
//   negative) in [-7, 32767], code bits = f(v >> 4)
// so a 4 KB (mu) / 2 KB (A) byte table indexed by v >> 3 / (v >> 4) + 1 holds every case; each workgroup fills its copy
// with enc_ulaw / enc_alaw themselves (g711_dev.h, the reference's arithmetic) on one representative per bucket, and
// the stream loop is sign, magnitude, one LDS byte read, sign mask: about half the VALU work per sample of the
// segment search, which is what the kernel was short of.
template <int LAW>
struct EncTable {
    static constexpr int kSize = LAW == WMX_LAW_U ? (32900 >> 3) + 1 : (32767 >> 4) + 2;
    static constexpr unsigned kPosMask = LAW == WMX_LAW_U ? 0xFFu : 0xD5u;
    __device__ static void fill(uint8_t *t) {
        for (int i = threadIdx.x; i < kSize; i += blockDim.x) {
            if (LAW == WMX_LAW_U) {
                const int v = (i << 3) < 0x84 ? 0x84 : (i << 3);  // buckets below 0x84 >> 3 are never indexed
                t[i] = (uint8_t)(enc_ulaw(v - 0x84) ^ kPosMask);
            } else {
                t[i] = i == 0 ? (uint8_t)(enc_alaw(-7) ^ 0x55u) : (uint8_t)(enc_alaw((i - 1) << 4) ^ kPosMask);
            }
        }
    }
    __device__ __forceinline__ static unsigned encode(const uint8_t *t, int pcm) {
        const int sign = pcm >> 31;            // 0 or -1
        const int mag = (pcm ^ sign) - sign;   // |pcm| (32768 for -32768)
        int idx;
        if (LAW == WMX_LAW_U)
            idx = (mag + 0x84) >> 3;
        else
            idx = ((mag + (sign & -8)) >> 4) + 1;  // -pcm - 8 for negative input, arithmetic shift
        return (unsigned)t[idx] ^ (kPosMask ^ ((unsigned)sign & 0x80u));
    }
};

// n8 = number of full 8-sample groups; `aligned` says both pointers allow the
// wide accesses (16 B PCM / 8 B code).  The tail (< 8 samples) and the unaligned
// case run through the scalar loop at the end.
template <int LAW>
__global__ __launch_bounds__(256) void g711_encode_kernel(const int16_t *__restrict__ pcm,
                                                          uint8_t *__restrict__ code, size_t n, int aligned) {
    __shared__ uint8_t tab[(EncTable<LAW>::kSize + 15) & ~15];
    EncTable<LAW>::fill(tab);
    __syncthreads();
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    size_t done = 0;
    if (aligned) {
        const size_t n8 = n / 8;
        const uint4 *src = reinterpret_cast<const uint4 *>(pcm);
        uint2 *dst = reinterpret_cast<uint2 *>(code);
        for (size_t i = tid; i < n8; i += nthreads) {
            uint4 v = src[i];
            unsigned w[4] = {v.x, v.y, v.z, v.w};
            unsigned c[8];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                c[2 * k] = EncTable<LAW>::encode(tab, (int)(int16_t)(w[k] & 0xFFFF));
                c[2 * k + 1] = EncTable<LAW>::encode(tab, (int)(int16_t)(w[k] >> 16));
            }
            uint2 o;
            o.x = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
            o.y = c[4] | (c[5] << 8) | (c[6] << 16) | (c[7] << 24);
            dst[i] = o;
        }
        done = n8 * 8;
    }
    for (size_t i = done + tid; i < n; i += nthreads) code[i] = (uint8_t)EncTable<LAW>::encode(tab, (int)pcm[i]);
}

template <int LAW>
__global__ __launch_bounds__(256) void g711_decode_kernel(const uint8_t *__restrict__ code,
                                                          int16_t *__restrict__ pcm, size_t n, int aligned) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    size_t done = 0;
    if (aligned) {
        const size_t n8 = n / 8;
        const uint2 *src = reinterpret_cast<const uint2 *>(code);
        uint4 *dst = reinterpret_cast<uint4 *>(pcm);
        for (size_t i = tid; i < n8; i += nthreads) {
            uint2 v = src[i];
            unsigned w[2] = {v.x, v.y};
            unsigned p[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                unsigned lo = (w[k / 2] >> (16 * (k & 1))) & 0xFF;
                unsigned hi = (w[k / 2] >> (16 * (k & 1) + 8)) & 0xFF;
                p[k] = ((unsigned)dec1<LAW>(lo) & 0xFFFF) | ((unsigned)dec1<LAW>(hi) << 16);
            }
            dst[i] = make_uint4(p[0], p[1], p[2], p[3]);
        }
        done = n8 * 8;
    }
    for (size_t i = done + tid; i < n; i += nthreads) pcm[i] = (int16_t)dec1<LAW>(code[i]);
}

inline bool ok16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool ok8(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

}  // namespace

extern "C" {

int wmx_g711_encode(int law, const int16_t *d_pcm, uint8_t *d_code, size_t n, void *stream) {
    if (law != WMX_LAW_A && law != WMX_LAW_U) {
        wmx::set_error("wmx_g711_encode: law must be 0 (A) or 1 (mu), got %d", law);
        return WMX_EINVAL;
    }
    if (n == 0) return 0;
    if (!d_pcm || !d_code) {
        wmx::set_error("wmx_g711_encode: null buffer");
        return WMX_EINVAL;
    }
    const int aligned = ok16(d_pcm) && ok8(d_code);
    const unsigned block = 256, grid = wmx::stream_grid((n + 7) / 8, block);
    if (law == WMX_LAW_A)
        hipLaunchKernelGGL(g711_encode_kernel<WMX_LAW_A>, dim3(grid), dim3(block), 0, wmx::as_stream(stream), d_pcm, d_code, n, aligned);
    else
        hipLaunchKernelGGL(g711_encode_kernel<WMX_LAW_U>, dim3(grid), dim3(block), 0, wmx::as_stream(stream), d_pcm, d_code, n, aligned);
    WMX_LAUNCH_CHECK();
    return 0;
}

int wmx_g711_decode(int law, const uint8_t *d_code, int16_t *d_pcm, size_t n, void *stream) {
    if (law != WMX_LAW_A && law != WMX_LAW_U) {
        wmx::set_error("wmx_g711_decode: law must be 0 (A) or 1 (mu), got %d", law);
        return WMX_EINVAL;
    }
    if (n == 0) return 0;
    if (!d_pcm || !d_code) {
        wmx::set_error("wmx_g711_decode: null buffer");
        return WMX_EINVAL;
    }
    const int aligned = ok16(d_pcm) && ok8(d_code);
    const unsigned block = 256, grid = wmx::stream_grid((n + 7) / 8, block);
    if (law == WMX_LAW_A)
        hipLaunchKernelGGL(g711_decode_kernel<WMX_LAW_A>, dim3(grid), dim3(block), 0, wmx::as_stream(stream), d_code, d_pcm, n, aligned);
    else
        hipLaunchKernelGGL(g711_decode_kernel<WMX_LAW_U>, dim3(grid), dim3(block), 0, wmx::as_stream(stream), d_code, d_pcm, n, aligned);
    WMX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
