// mfft.hip -- the reference's stand-alone radix-2 FFT helpers (math/fft.c) batched for gfx950.
//
// Replaces FFT / FFTR / IFFT / IFFTR / fft_stream (math/fft.h:19-51, math/fft.c:121-424) for many independent
// transforms per launch.  The reference is a textbook decimation-in-time radix-2 FFT on float arrays whose
// twiddles are cos/sin(2.0 * FFT_PI * p / N) evaluated in double at every butterfly (FFT_PI = 3.1415926535897,
// math/fft.c:21); the products and their sum are double, rounded to float once, the butterfly add/sub are float.
//
// Mapping: one wavefront per transform, data in LDS (8 KB for N = 1024), log2(N) passes of N/2 butterflies spread
// over the 64 lanes, wave-scope fences between passes.  The twiddle table for a size is built once on the host
// with the reference's own expression and libm (so every entry has the reference's bits) and cached on the device
// as double2[N/2]; kernels keep the reference's double multiply-add and single rounding.  Bit-exact for re / im /
// amplitude; the phase curve goes through the device's double atan2 (<= 1 float ulp from glibc's).
#include <cmath>
#include <map>
#include <mutex>
#include <utility>
#include <vector>
#include <cstring>
#include "wmx_internal.h"
#include "fft_ooura.h"  // wave_sync
#include "../../include/wmix_compat.h"

namespace wmx {
namespace {

constexpr double kMfftPi = 3.1415926535897;  // math/fft.c:21
constexpr unsigned kMfftMaxN = 4096;

// ---------------------------------------------------------------- host: twiddle tables, cached per size
std::mutex g_tw_mutex;
// (current device, size n) -> device double2[n/2] = (cos, sin)(2.0 * PI * p / n).  The entry points take raw device
// pointers and run on the caller's current device (one worker thread per GPU in a C host): a table
// made on one device must not be handed to a launch on another.
typedef std::pair<int, unsigned> TwKey;
std::map<TwKey, double2 *> g_tw;

int twiddles_for(unsigned n, const double2 **out) {
    int dev = 0;
    WMX_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_tw_mutex);
    auto it = g_tw.find(TwKey(dev, n));
    if (it != g_tw.end()) {
        *out = it->second;
        return 0;
    }
    const unsigned h = n / 2 ? n / 2 : 1;
    std::vector<double2> t(h);
    for (unsigned p = 0; p < h; p++) {
        const int pi = (int)p;
        t[p].x = cos(2.0 * kMfftPi * pi / n);  // same expression, same libm as math/fft.c:108
        t[p].y = sin(2.0 * kMfftPi * pi / n);
    }
    double2 *d = nullptr;
    WMX_HIP(hipMalloc(&d, h * sizeof(double2)));
    WMX_HIP(hipMemcpy(d, t.data(), h * sizeof(double2), hipMemcpyHostToDevice));
    g_tw[TwKey(dev, n)] = d;
    *out = d;
    return 0;
}

// The same values stage by stage: entry (2^b - 1) + j is the twiddle of butterfly j of stage b + 1 (what dit_passes reads
// as tw[j << (m - 1 - b)]), so the lanes of a wave read consecutive entries (mfft_regs_kernel).  double2[nc], nc - 1 used.
std::map<TwKey, double2 *> g_tw_staged;

int staged_twiddles_for(unsigned nc, const double2 **out) {
    int dev = 0;
    WMX_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_tw_mutex);
    auto it = g_tw_staged.find(TwKey(dev, nc));
    if (it != g_tw_staged.end()) {
        *out = it->second;
        return 0;
    }
    unsigned m = 0;
    while ((1u << (m + 1)) <= nc) m++;
    std::vector<double2> t(nc);
    t[nc - 1] = double2{0.0, 0.0};
    for (unsigned b = 0; b < m; b++)
        for (unsigned j = 0; j < (1u << b); j++) {
            const int pi = (int)(j << (m - 1 - b));
            t[(1u << b) - 1 + j].x = cos(2.0 * kMfftPi * pi / nc);  // the expression of twiddles_for
            t[(1u << b) - 1 + j].y = sin(2.0 * kMfftPi * pi / nc);
        }
    double2 *d = nullptr;
    WMX_HIP(hipMalloc(&d, nc * sizeof(double2)));
    WMX_HIP(hipMemcpy(d, t.data(), nc * sizeof(double2), hipMemcpyHostToDevice));
    g_tw_staged[TwKey(dev, nc)] = d;
    *out = d;
    return 0;
}

// ---------------------------------------------------------------- device
// one radix-2 DIT butterfly of the reference (math/fft.c:106-113 / 281-292): t = w * x[q] in double, rounded once;
// x[q] = x[r] - t, x[r] = x[r] + t (the inverse halves both)
template <bool INVERSE>
__device__ __forceinline__ void dit_bfly(float &ar, float &ai, float &xr, float &xi, double2 w) {
    float tr, ti;
    if constexpr (!INVERSE) {
        tr = (float)((double)xr * w.x + (double)xi * w.y);
        ti = (float)((double)xi * w.x - (double)xr * w.y);
        xr = ar - tr;
        xi = ai - ti;
        ar = ar + tr;
        ai = ai + ti;
    } else {
        tr = (float)((double)xr * w.x - (double)xi * w.y);
        ti = (float)((double)xi * w.x + (double)xr * w.y);
        xr = (ar - tr) / 2;
        xi = (ai - ti) / 2;
        ar = (ar + tr) / 2;
        ai = (ai + ti) / 2;
    }
}

// the same butterfly on (re, im) register pairs
template <bool INVERSE>
__device__ __forceinline__ void dit_bfly2(v2f &a, v2f &x, double2 w) {
    v2f t;
    if constexpr (!INVERSE) {
        t.x = (float)((double)x.x * w.x + (double)x.y * w.y);
        t.y = (float)((double)x.y * w.x - (double)x.x * w.y);
        x = a - t;
        a = a + t;
    } else {
        t.x = (float)((double)x.x * w.x - (double)x.y * w.y);
        t.y = (float)((double)x.y * w.x + (double)x.x * w.y);
        x = (a - t) / 2;
        a = (a + t) / 2;
    }
}

// log2(n) radix-2 DIT passes on n complex points in LDS (math/fft.c:81-118 / 256-296).  Two consecutive stages are
// evaluated on four points held in registers (the same butterflies in the same order, one LDS round trip instead of
// two); an odd last stage runs alone.  `tw` is the size-n twiddle table in LDS.
template <bool INVERSE>
__device__ __forceinline__ void dit_passes(float *re, float *im, unsigned n, unsigned m, const double2 *tw, int lane) {
    unsigned l = 1;
    for (; l + 1 <= m; l += 2) {
        const unsigned half = 1u << (l - 1), s1 = m - l, s2 = m - l - 1;  // twiddle steps 2^(m-l), 2^(m-l-1)
        for (unsigned t = lane; t < n / 4; t += 64) {
            const unsigned i = t >> (l - 1), j = t & (half - 1);
            const unsigned r0 = j + 4 * half * i, r1 = r0 + half, r2 = r1 + half, r3 = r2 + half;
            float ar = re[r0], ai = im[r0], br = re[r1], bi = im[r1], cr = re[r2], ci = im[r2], dr = re[r3], di = im[r3];
            const double2 wa = tw[j << s1];
            dit_bfly<INVERSE>(ar, ai, br, bi, wa);  // stage l:     (r0, r1) and (r2, r3), same twiddle
            dit_bfly<INVERSE>(cr, ci, dr, di, wa);
            dit_bfly<INVERSE>(ar, ai, cr, ci, tw[j << s2]);           // stage l + 1: (r0, r2)
            dit_bfly<INVERSE>(br, bi, dr, di, tw[(j + half) << s2]);  //              (r1, r3)
            re[r0] = ar, im[r0] = ai, re[r1] = br, im[r1] = bi, re[r2] = cr, im[r2] = ci, re[r3] = dr, im[r3] = di;
        }
        wave_sync();
    }
    if (l == m) {
        const unsigned half = 1u << (l - 1);
        for (unsigned t = lane; t < n / 2; t += 64) {
            const unsigned i = t >> (l - 1), j = t & (half - 1);
            const unsigned r = j + 2 * half * i, q = r + half;
            float ar = re[r], ai = im[r], xr = re[q], xi = im[q];
            dit_bfly<INVERSE>(ar, ai, xr, xi, tw[j]);
            re[r] = ar, im[r] = ai, re[q] = xr, im[q] = xi;
        }
        wave_sync();
    }
}

__device__ __forceinline__ unsigned rev_bits(unsigned i, unsigned m) { return m ? (__brev(i) >> (32 - m)) : 0u; }

__device__ __forceinline__ void emit(unsigned idx, float r, float i, unsigned n, float *o_re, float *o_im, float *o_af, float *o_pf) {
    if (o_re) o_re[idx] = r;
    if (o_im) o_im[idx] = i;
    // math/fft.c:143-146: sqrt(double(float sum)) / (N/2), rounded to float once.  N/2 is a power of two, so the quotient
    // is an exact scaling, and rounding a double square root of a float to float equals the correctly rounded float square
    // root (53 >= 2*24 + 2 bits): sqrtf(x) * 2^-k is the same float, without the fp64 sqrt and divide.
    if (o_af) o_af[idx] = sqrtf(r * r + i * i) * (1.0f / (float)(n / 2));
    if (o_pf) o_pf[idx] = (float)atan2((double)i, (double)r);                        // math/fft.c:149-152
}

// KIND 0 FFT, 1 FFTR, 2 IFFT, 3 IFFTR; STREAM: fft_stream's FIFO update in front of a KIND 0 transform
template <int KIND, bool STREAM>
__global__ void mfft_kernel(int n_batch, unsigned n, unsigned m, const double2 *__restrict__ tw_inner, const double2 *__restrict__ tw_n,
                            const float *in_re, const float *in_im, float *out_re, float *out_im, float *out_af, float *out_pf,
                            unsigned in_len) {
    extern __shared__ double2 lds_tw[];  // [nc / 2] twiddles of the complex transform, then the waves' data
    constexpr bool REAL = (KIND == 1 || KIND == 3), INV = (KIND >= 2);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int idx = blockIdx.x * (blockDim.x >> 6) + wave;  // one transform per wave
    const unsigned nc = REAL ? n / 2 : n, mc = REAL ? m - 1 : m;  // points of the complex transform
    const unsigned ntw = nc / 2 ? nc / 2 : 1;
    for (unsigned i = threadIdx.x; i < ntw; i += blockDim.x) lds_tw[i] = tw_inner[i];
    __syncthreads();  // the only block-level barrier
    if (idx >= n_batch) return;
    float *lds = reinterpret_cast<float *>(lds_tw + ntw);
    float *re = lds + (size_t)wave * 2 * n, *im = re + n;
    const size_t base = (size_t)idx * n;
    float *o_re = out_re ? out_re + base : nullptr, *o_im = out_im ? out_im + base : nullptr;
    float *o_af = out_af ? out_af + base : nullptr, *o_pf = out_pf ? out_pf + base : nullptr;

    if constexpr (STREAM) {
        // fft_stream (math/fft.c:413-424): stream[0..L) = stream[L..2L), stream[L..2L) = in[0..L), then FFT(stream).
        // `out_re` is the stream pool [n_batch][n] (updated in place), `in_re` the new samples [n_batch][in_len].
        float *pool = out_re + base;
        const float *fresh = in_re + (size_t)idx * in_len;
        float *stage = lds + (size_t)(blockDim.x >> 6) * 2 * n + (size_t)wave * 2 * in_len;  // the pool's new head
        for (unsigned i = lane; i < 2 * in_len; i += 64) stage[i] = i < in_len ? pool[i + in_len] : fresh[i - in_len];
        wave_sync();  // every read of pool[in_len..2 in_len) is done before any lane overwrites it
        for (unsigned i = lane; i < n; i += 64) {
            float v;
            if (i < 2 * in_len) {
                v = stage[i];
                pool[i] = v;
            } else {
                v = pool[i];
            }
            re[rev_bits(i, mc)] = v;
            im[i] = 0.f;
        }
        o_re = nullptr;  // the pool is not a spectrum output
    } else if constexpr (!REAL) {
        for (unsigned i = lane; i < n; i += 64) {
            const unsigned j = rev_bits(i, mc);
            re[j] = in_re ? in_re[base + i] : 0.f;
            im[j] = in_im ? in_im[base + i] : 0.f;
        }
    } else {
        // y[i] = in[2i] + j in[2i+1]; the imaginary input array is not used by the real variants
        for (unsigned i = lane; i < nc; i += 64) {
            const unsigned j = rev_bits(i, mc);
            re[j] = in_re ? in_re[base + 2 * i] : 0.f;
            im[j] = in_re ? in_re[base + 2 * i + 1] : 0.f;
        }
    }
    wave_sync();
    dit_passes<INV>(re, im, nc, mc, lds_tw, lane);

    if constexpr (!REAL) {
        for (unsigned i = lane; i < n; i += 64) emit(i, re[i], im[i], n, o_re, o_im, INV ? nullptr : o_af, INV ? nullptr : o_pf);
    } else {
        // split into the spectra of the even / odd samples and the last butterfly stage (math/fft.c:182-232 / 345-392)
        const unsigned h = nc;
        for (unsigned j = lane; j < h; j += 64) {
            float x1r, x1i, x2r, x2i;
            if (j == 0) {
                x1r = re[0];
                x1i = im[0];
                x2r = im[0];
                x2i = -re[0];
            } else {
                const float yr = re[j], yi = im[j], zr = re[h - j], zi = im[h - j];
                x1r = (yr + zr) / 2;
                x1i = (yi - zi) / 2;
                x2r = (yi + zi) / 2;
                x2i = (zr - yr) / 2;
            }
            const double2 w = tw_n[j];
            float xr, xi;
            if constexpr (!INV) {
                const float tr = (float)((double)x2r * w.x + (double)x2i * w.y);
                const float ti = (float)((double)x2i * w.x - (double)x2r * w.y);
                xr = x1r + tr;
                xi = x1i + ti;
            } else {
                const float tr = (float)((double)x2r * w.x - (double)x2i * w.y);
                const float ti = (float)((double)x2i * w.x + (double)x2r * w.y);
                xr = (x1r + tr) / 2;
                xi = (x1i + ti) / 2;
            }
            emit(j, xr, xi, n, o_re, o_im, INV ? nullptr : o_af, INV ? nullptr : o_pf);
            if (j == 0) {
                float mr = x1r - x2r, mi = x1i - x2i;
                if constexpr (INV) {
                    mr = mr / 2;
                    mi = mi / 2;
                }
                emit(h, mr, mi, n, o_re, o_im, INV ? nullptr : o_af, INV ? nullptr : o_pf);
            } else {
                emit(n - j, xr, -xi, n, o_re, o_im, INV ? nullptr : o_af, INV ? nullptr : o_pf);
            }
        }
    }
}

// ---------------------------------------------------------------- the register-resident kernel (32 .. 4096 complex points)
// A transform stays in registers, P = 2^LP points per lane as (re, im) pairs, on T = 2^LB lanes: 4 / 8 / 16 lanes for 32 / 64 /
// 128 points (16 / 8 / 4 transforms per wave), one wave for 256 / 512 / 1024, two / four waves for 2048 / 4096.  The log2(nc) stages
// run in passes of LP stages whose butterflies pair registers of one lane; between two passes the points change places through
// the transform's LDS buffer (one float2 write and one read per point, padded by one element per P so that both sides are
// spread over the banks; wave-scope fences up to one wave per transform, workgroup barriers above).  The same butterflies with
// the same operands as dit_passes -- the order of independent butterflies is all that differs.
//   * the bit-reversed load needs no LDS: register k of lane L takes sample (rev(k) << LB) | rev(L), so one load instruction
//     covers contiguous T-point runs of the input, lanes permuted inside them;
//   * pass 0's twiddles are the same for every lane (scalar loads, kept in SGPRs across the transforms of a wave); later
//     passes read the stage-by-stage table at consecutive entries -- from LDS up to 1024 points, from global memory (L2)
//     above, where the table is as large as the LDS a transform's buffer leaves;
//   * a workgroup works through transforms g, g + grid, ...: the table is staged once per workgroup, and the next input is
//     fetched while the current transform is computed.
// Point index of register k in a pass whose register bits are [S, S + LP):
//   r = (lane >> S) << (S + LP) | k << S | lane & (2^S - 1).
template <int LP>
__device__ constexpr unsigned rev_small(unsigned k) {
    unsigned r = 0;
    for (int b = 0; b < LP; b++) r |= ((k >> b) & 1u) << (LP - 1 - b);
    return r;
}

constexpr int regs_lane_bits(int mc) { return mc <= 7 ? mc - 3 : (mc <= 10 ? 6 : mc - 4); }  // LB: P = 8, 8, 4, 8, 16, 16, 16

template <int MC>
struct RegFft {
    static constexpr int NC = 1 << MC, LB = regs_lane_bits(MC), T = 1 << LB, LP = MC - LB, P = 1 << LP, NPASS = (MC + LP - 1) / LP;
    static constexpr int TPB = 256 >> LB;               // transforms per 256-thread workgroup
    static constexpr bool TW_IN_LDS = MC <= 10;         // later passes' twiddles: LDS copy of the staged table, or the table itself
    static constexpr int kBufElems = NC + (NC >> LP);   // float2 elements of a transform's exchange buffer
    static constexpr int shift_of(int q) { return (q + 1) * LP <= MC ? q * LP : MC - LP; }
    __device__ static __forceinline__ unsigned pad(unsigned r) { return r + (r >> LP); }
    template <int S>
    __device__ static __forceinline__ unsigned point(int k, unsigned lane) {
        return ((lane >> S) << (S + LP)) | ((unsigned)k << S) | (lane & ((1u << S) - 1u));
    }
    // the lanes of one transform have exchanged data through LDS
    __device__ static __forceinline__ void group_sync() {
        if constexpr (LB <= 6)
            wave_sync();
        else
            __syncthreads();
    }

    template <bool INV, int Q>
    __device__ static __forceinline__ void pass(v2f (&v)[P], v2f *buf, const double2 *tw_lds,
                                                const double2 *__restrict__ tw_glb, unsigned lane) {
        if constexpr (Q < NPASS) {
            constexpr int S = shift_of(Q);
            if constexpr (Q > 0) {
                constexpr int S0 = shift_of(Q - 1);
#pragma unroll
                for (int k = 0; k < P; k++) buf[pad(point<S0>(k, lane))] = v[k];
                group_sync();
#pragma unroll
                for (int k = 0; k < P; k++) v[k] = buf[pad(point<S>(k, lane))];
                group_sync();
            }
            constexpr int B0 = Q * LP, B1 = (Q + 1) * LP <= MC ? (Q + 1) * LP : MC;  // stages B0 + 1 .. B1 pair index bit b
            const unsigned low = lane & ((1u << S) - 1u);
#pragma unroll
            for (int b = B0; b < B1; b++) {
                const int kb = b - S;  // the register bit of this stage
#pragma unroll
                for (int c = 0; c < (1 << kb); c++) {
                    // butterfly j = low b bits of r = low | c << S
                    const double2 w = S == 0 ? tw_glb[(1 << b) - 1 + c]
                                             : (TW_IN_LDS ? tw_lds : tw_glb)[(1u << b) - 1u + low + ((unsigned)c << S)];
#pragma unroll
                    for (int hi = 0; hi < (P >> (kb + 1)); hi++) {
                        const int k0 = (hi << (kb + 1)) | c, k1 = k0 | (1 << kb);
                        dit_bfly2<INV>(v[k0], v[k1], w);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // a stage's twiddles are fetched for that stage
            }
            pass<INV, Q + 1>(v, buf, tw_lds, tw_glb, lane);
        }
    }
};

// waves per SIMD the kernel is compiled for (its register budget), by points per lane; the real forward kind needs more
#ifndef WMX_MFFT_W9
#define WMX_MFFT_W9 4
#endif
constexpr int regs_waves_per_simd(int kind, int mc) {
    const int lp = mc - regs_lane_bits(mc);
    if (mc <= 7) return 4;  // several transforms per wave: per-lane base addresses on top of the P = 8 points
    return lp <= 2 ? (kind == 1 ? 5 : 6) : (lp == 3 ? (kind == 1 ? WMX_MFFT_W9 : 5) : (kind == 1 ? 2 : 3));
}

template <int KIND, bool STREAM, int MC>
__global__ __launch_bounds__(256, regs_waves_per_simd(KIND, MC)) void mfft_regs_kernel(int n_batch, const double2 *__restrict__ tw_staged,
                                                                                     const double2 *__restrict__ tw_n, const float *in_re,
                                                                                     const float *in_im, float *out_re, float *out_im,
                                                                                     float *out_af, float *out_pf, unsigned in_len) {
    using R = RegFft<MC>;
    constexpr bool REAL = (KIND == 1 || KIND == 3), INV = (KIND >= 2);
    constexpr int NC = R::NC, P = R::P, LP = R::LP, LB = R::LB, TPB = R::TPB;
    constexpr unsigned N = REAL ? 2u * NC : (unsigned)NC, T = R::T;
    extern __shared__ double2 lds_tw[];  // [NC] stage-by-stage twiddles (up to 1024 points), then one exchange buffer per transform
    constexpr int NTW = R::TW_IN_LDS ? NC : 0;
    if constexpr (R::TW_IN_LDS) {
        for (unsigned i = threadIdx.x; i < (unsigned)NC; i += blockDim.x) lds_tw[i] = tw_staged[i];
        __syncthreads();
    }
    const unsigned lane0 = threadIdx.x & (T - 1u);  // the lane's index inside its transform
    // the transform inside the workgroup: wave-uniform from one wave per transform up (addresses stay on the scalar unit)
    const int sub = LB >= 6 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> LB)) : (int)(threadIdx.x >> LB);
    v2f *buf = reinterpret_cast<v2f *>(lds_tw + NTW) + (size_t)sub * R::kBufElems;
    // The input of the next transform is fetched while the current one is computed (not fft_stream's, whose load also
    // stores).  load_input: register k of lane L takes sample (rev(k) << LB) | rev(L).
    auto load_input = [&](int idx, unsigned rl, v2f(&dst)[P]) {
        const size_t base = (size_t)idx * N;
#pragma unroll
        for (int k = 0; k < P; k++) dst[k] = v2f{0.f, 0.f};
        if constexpr (!REAL) {
            if (in_re) {
#pragma unroll
                for (int k = 0; k < P; k++) dst[k].x = in_re[base + ((rev_small<LP>(k) << LB) | rl)];
            }
            if (in_im) {
#pragma unroll
                for (int k = 0; k < P; k++) dst[k].y = in_im[base + ((rev_small<LP>(k) << LB) | rl)];
            }
        } else if (in_re) {
            // y[i] = in[2i] + j in[2i+1] (8-byte loads: the host checks the alignment)
            const v2f *src = reinterpret_cast<const v2f *>(in_re + base);
#pragma unroll
            for (int k = 0; k < P; k++) dst[k] = src[(rev_small<LP>(k) << LB) | rl];
        }
    };
    // A lane whose transform lies beyond the batch (the ragged end of the last group) computes the last transform again and
    // stores nothing: the loop and its barriers stay uniform over the workgroup.
    const int last = n_batch - 1;
    v2f nxt[P];
    if constexpr (!STREAM) {
        const int idx0 = (int)blockIdx.x * TPB + sub;
        load_input(idx0 < n_batch ? idx0 : last, __brev(lane0) >> (32 - LB), nxt);
    }

    for (int g = blockIdx.x; g * TPB < n_batch; g += gridDim.x) {
        // Every address of the body is a function of the lane alone; hoisted out of this loop they would be a hundred
        // live registers (and were: scratch spills).  The lane index is made opaque per iteration so they are formed
        // where they are used.
        unsigned lane = lane0;
        asm volatile("" : "+v"(lane));
        const unsigned rl = __brev(lane) >> (32 - LB);  // the lane's place inside a T-point run of the input
        const int idx_raw = g * TPB + sub;
        const bool valid = idx_raw < n_batch;
        const int idx = valid ? idx_raw : last;
        const size_t base = (size_t)idx * N;
        float *o_re = out_re && valid ? out_re + base : nullptr, *o_im = out_im && valid ? out_im + base : nullptr;
        float *o_af = out_af && valid ? out_af + base : nullptr, *o_pf = out_pf && valid ? out_pf + base : nullptr;
        v2f v[P];

        if constexpr (STREAM) {
            // fft_stream (math/fft.c:413-424), as in mfft_kernel: the pool's new head is stored back, every read of the
            // old pool is complete before the first store
            float *pool = out_re + base;
            const float *fresh = in_re + (size_t)idx * in_len;
#pragma unroll
            for (int k = 0; k < P; k++) {
                const unsigned i = (rev_small<LP>(k) << LB) | rl;
                v[k].x = i < in_len ? pool[i + in_len] : (i < 2 * in_len ? fresh[i - in_len] : pool[i]);
                v[k].y = 0.f;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (LB > 6) __syncthreads();  // the other waves of the transform have read the old pool too
#pragma unroll
            for (int k = 0; k < P; k++) {
                const unsigned i = (rev_small<LP>(k) << LB) | rl;
                if (valid && i < 2 * in_len) pool[i] = v[k].x;
            }
            o_re = nullptr;  // the pool is not a spectrum output
        } else {
#pragma unroll
            for (int k = 0; k < P; k++) v[k] = nxt[k];
            const int idx_n = (g + (int)gridDim.x) * TPB + sub;
            if ((g + (int)gridDim.x) * TPB < n_batch) load_input(idx_n < n_batch ? idx_n : last, rl, nxt);
        }

        R::template pass<INV, 0>(v, buf, lds_tw, tw_staged, lane);
        // now register k of lane L holds point k * T + L

        // The phase curve (double atan2, math/fft.c:149-152) is the one output that is expensive per point and rarely asked
        // for: it runs as a rolled loop over the buffer behind everything else, so the transform's registers are not
        // priced for 2 P inlined copies of it.  (The branches on the output arrays test the kernel arguments: uniform.)
        float *const phase = INV ? nullptr : o_pf, *const ampl = INV ? nullptr : o_af;
        const bool want_phase = !INV && out_pf, want_ampl = !INV && out_af, want_re = out_re && !STREAM, want_im = out_im != nullptr;
        constexpr float kAmpScale = 1.0f / (float)(N / 2);  // emit(): sqrtf(r*r + i*i) * 2^-k
        if constexpr (!REAL) {
            // Loads and stores share one in-order counter and the branches below hide their number from the compiler: the
            // wait for the prefetched input would land behind this transform's stores.  A use in front of them puts it here,
            // where the loads are long done.
            if constexpr (!STREAM) {
#pragma unroll
                for (int k = 0; k < P; k++) asm volatile("" : "+v"(nxt[k]));
            }
            // one uniform branch per output array, its P stores together
            if (want_re) {
#pragma unroll
                for (int k = 0; k < P; k++)
                    if (valid) o_re[k * T + lane] = v[k].x;
            }
            if (want_im) {
#pragma unroll
                for (int k = 0; k < P; k++)
                    if (valid) o_im[k * T + lane] = v[k].y;
            }
            if (want_ampl) {
#pragma unroll
                for (int k = 0; k < P; k++) {
                    const float a = sqrtf(v[k].x * v[k].x + v[k].y * v[k].y) * kAmpScale;
                    if (valid) ampl[k * T + lane] = a;
                }
            }
            if (want_phase) {
                R::group_sync();
#pragma unroll
                for (int k = 0; k < P; k++) buf[k * T + lane] = v[k];
                R::group_sync();
#pragma nounroll
                for (int k = 0; k < P; k++) {
                    const v2f t = buf[k * T + lane];
                    const float ph = (float)atan2((double)t.y, (double)t.x);
                    if (valid) phase[k * T + lane] = ph;
                }
                R::group_sync();
            }
        } else {
            // The split into the spectra of the even / odd samples and the last stage (math/fft.c:182-232 / 345-392).  Points
            // j and NC - j are each other's partners, so one lane takes both and writes both results back in place: P / 2
            // steps over the buffer (unpadded: both sides run along the banks) with nothing but arithmetic and the two table
            // entries of the next step in flight.  Unrolled on the registers with the stores inside, this stage was the larger
            // half of the kernel's code and twice the registers of everything else, and every step waited for the stores
            // (one counter for loads and stores).  Lane 0's first step has j = 0, which has no partner, and takes the
            // self-partnered point NC / 2 beside it.
            const double2 *const twl = tw_n + lane;
            double2 wa = twl[0], wb = twl[lane == 0 ? NC / 2 : NC - 2 * lane];  // entries j and NC - j of step 0
            R::group_sync();
#pragma unroll
            for (int k = 0; k < P; k++) buf[k * T + lane] = v[k];
            R::group_sync();
            constexpr unsigned h = NC;
            // point h belongs to j = 0: x1 - x2 of (re[0], im[0]) (every lane computes it, lane 0 stores it)
            const v2f y0 = buf[0];
            float mr = y0.x - y0.y, mi = y0.y - (-y0.x);
            if constexpr (INV) {
                mr = mr / 2;
                mi = mi / 2;
            }
            R::group_sync();
#pragma unroll
            for (int t = 0; t < P / 2; t++) {
                const unsigned j = t * T + lane;
                const bool first = j == 0;
                const unsigned jb = first ? h / 2 : h - j;
                const unsigned jn = j + (t + 1 < P / 2 ? T : 0u);  // the next step's entries (the last step's again)
                const double2 wa_next = tw_n[jn], wb_next = tw_n[jn == 0 ? h / 2 : h - jn];
                const v2f y = buf[j], z = buf[jb];
                // point j (partner z; none for j = 0) and point jb (partner y; itself for NC / 2)
                const v2f pa = first ? y : z, pb = first ? z : y;
                float a1r = (y.x + pa.x) / 2, a1i = (y.y - pa.y) / 2, a2r = (y.y + pa.y) / 2, a2i = (pa.x - y.x) / 2;
                if (first) a1r = y.x, a1i = y.y, a2r = y.y, a2i = -y.x;  // math/fft.c:189-194
                const float b1r = (z.x + pb.x) / 2, b1i = (z.y - pb.y) / 2, b2r = (z.y + pb.y) / 2, b2i = (pb.x - z.x) / 2;
                v2f xa, xb;
                if constexpr (!INV) {
                    xa.x = a1r + (float)((double)a2r * wa.x + (double)a2i * wa.y);
                    xa.y = a1i + (float)((double)a2i * wa.x - (double)a2r * wa.y);
                    xb.x = b1r + (float)((double)b2r * wb.x + (double)b2i * wb.y);
                    xb.y = b1i + (float)((double)b2i * wb.x - (double)b2r * wb.y);
                } else {
                    xa.x = (a1r + (float)((double)a2r * wa.x - (double)a2i * wa.y)) / 2;
                    xa.y = (a1i + (float)((double)a2i * wa.x + (double)a2r * wa.y)) / 2;
                    xb.x = (b1r + (float)((double)b2r * wb.x - (double)b2i * wb.y)) / 2;
                    xb.y = (b1i + (float)((double)b2i * wb.x + (double)b2r * wb.y)) / 2;
                }
                buf[j] = xa;
                buf[jb] = xb;
                wa = wa_next, wb = wb_next;
            }
            R::group_sync();
#pragma unroll
            for (int k = 0; k < P; k++) v[k] = buf[k * T + lane];
#pragma unroll
            for (int k = 0; k < P; k++) asm volatile("" : "+v"(nxt[k]));  // the wait for the next input: here, not behind the stores
            // outputs j and N - j (h beside j = 0), one uniform branch per array
            if (want_re) {
#pragma unroll
                for (int k = 0; k < P; k++) {
                    const unsigned j = k * T + lane;
                    if (valid) {
                        o_re[j] = v[k].x;
                        o_re[k == 0 && j == 0 ? h : N - j] = k == 0 && j == 0 ? mr : v[k].x;
                    }
                }
            }
            if (want_im) {
#pragma unroll
                for (int k = 0; k < P; k++) {
                    const unsigned j = k * T + lane;
                    if (valid) {
                        o_im[j] = v[k].y;
                        o_im[k == 0 && j == 0 ? h : N - j] = k == 0 && j == 0 ? mi : -v[k].y;
                    }
                }
            }
            if (want_ampl) {
#pragma unroll
                for (int k = 0; k < P; k++) {
                    const unsigned j = k * T + lane;
                    const float a = sqrtf(v[k].x * v[k].x + v[k].y * v[k].y) * kAmpScale;  // (-xi)^2 == xi^2: point N - j has it too
                    float a2 = a;
                    if (k == 0) {
                        const float ah = sqrtf(mr * mr + mi * mi) * kAmpScale;
                        a2 = j == 0 ? ah : a;
                    }
                    if (valid) {
                        ampl[j] = a;
                        ampl[k == 0 && j == 0 ? h : N - j] = a2;
                    }
                }
            }
            if (want_phase) {
#pragma nounroll
                for (int k = 0; k < P; k++) {
                    const unsigned j = k * T + lane;
                    const v2f t = buf[j];
                    const float p1 = (float)atan2((double)t.y, (double)t.x);
                    const float p2 = j == 0 ? (float)atan2((double)mi, (double)mr) : (float)atan2((double)-t.y, (double)t.x);
                    if (valid) {
                        phase[j] = p1;
                        phase[j == 0 ? h : N - j] = p2;
                    }
                }
            }
            R::group_sync();  // the buffer is free for the next transform
        }
    }
}

int check_size(unsigned n, unsigned *m) {
    if (n < 2 || n > kMfftMaxN || (n & (n - 1))) {
        set_error("math/fft: N = %u (must be a power of two in [2, %u])", n, kMfftMaxN);
        return WMX_EINVAL;
    }
    unsigned b = 0;
    while ((1u << (b + 1)) <= n) b++;
    *m = b;
    return 0;
}

unsigned waves_per_block(unsigned n) { return n <= 1024 ? 4u : (n <= 2048 ? 2u : 1u); }

// mfft_regs_kernel: complex sizes 2^5 .. 2^12 (every kind from 64 samples up); a grid of as many workgroups as the device holds at once (256 CUs, 160 KB of
// LDS each), every workgroup loops over its share of the batch
bool regs_path(unsigned mc) { return mc >= 5 && mc <= 12; }
struct RegsLaunch {
    dim3 grid, block;
    size_t lds;
};
RegsLaunch regs_launch(int kind, unsigned mc, int n_batch) {
    const int lb = regs_lane_bits((int)mc), lp = (int)mc - lb;
    const size_t nc = (size_t)1 << mc, buf = nc + (nc >> lp), tpb = (size_t)256 >> lb;
    RegsLaunch L;
    L.lds = (mc <= 10 ? nc * sizeof(double2) : 0) + tpb * buf * sizeof(float2);
    size_t resident = (160u * 1024u) / L.lds;
    const size_t by_registers = (size_t)regs_waves_per_simd(kind, (int)mc);
    if (resident > by_registers) resident = by_registers;
    if (resident < 1) resident = 1;
    const size_t cap = 256u * resident, need = ((size_t)n_batch + tpb - 1) / tpb;
    L.grid = dim3((unsigned)(need < cap ? need : cap));
    L.block = dim3(256);
    return L;
}

}  // namespace
}  // namespace wmx

using namespace wmx;

extern "C" int wmx_mfft(int kind, int n_batch, unsigned n, const float *d_in_re, const float *d_in_im, float *d_out_re,
                        float *d_out_im, float *d_out_af, float *d_out_pf, void *stream) {
    unsigned m;
    if (int rc = check_size(n, &m)) return rc;
    if (kind < 0 || kind > 3 || n_batch < 0) {
        set_error("wmx_mfft: kind %d / n_batch %d", kind, n_batch);
        return WMX_EINVAL;
    }
    if (n_batch == 0) return 0;
    const bool real = kind == 1 || kind == 3;
    const double2 *tw_inner = nullptr, *tw_n = nullptr;
    if (int rc = twiddles_for(n, &tw_n)) return rc;
    if (real) {
        if (int rc = twiddles_for(n / 2, &tw_inner)) return rc;
    } else {
        tw_inner = tw_n;
    }
    hipStream_t s = as_stream(stream);
    const unsigned mc = real ? m - 1 : m;
    if (regs_path(mc) && (!real || (reinterpret_cast<uintptr_t>(d_in_re) & 7) == 0)) {
        const double2 *tw_staged = nullptr;
        if (int rc = staged_twiddles_for(1u << mc, &tw_staged)) return rc;
        const RegsLaunch L = regs_launch(kind, mc, n_batch);
#define WMX_MFFT_REGS(K, MC) \
    hipLaunchKernelGGL((mfft_regs_kernel<K, false, MC>), L.grid, L.block, L.lds, s, n_batch, tw_staged, tw_n, d_in_re, d_in_im, d_out_re, d_out_im, d_out_af, d_out_pf, 0u)
#define WMX_MFFT_REGS_K(K) \
    switch (mc) { \
        case 5: WMX_MFFT_REGS(K, 5); break; \
        case 6: WMX_MFFT_REGS(K, 6); break; \
        case 7: WMX_MFFT_REGS(K, 7); break; \
        case 8: WMX_MFFT_REGS(K, 8); break; \
        case 9: WMX_MFFT_REGS(K, 9); break; \
        case 10: WMX_MFFT_REGS(K, 10); break; \
        case 11: WMX_MFFT_REGS(K, 11); break; \
        default: WMX_MFFT_REGS(K, 12); break; \
    }
        switch (kind) {
            case 0: WMX_MFFT_REGS_K(0); break;
            case 1: WMX_MFFT_REGS_K(1); break;
            case 2: WMX_MFFT_REGS_K(2); break;
            default: WMX_MFFT_REGS_K(3); break;
        }
#undef WMX_MFFT_REGS_K
#undef WMX_MFFT_REGS
        WMX_LAUNCH_CHECK();
        return 0;
    }
    const unsigned wpb = waves_per_block(n);
    const dim3 grid((unsigned)((n_batch + wpb - 1) / wpb)), block(64 * wpb);
    const size_t lds = (size_t)wpb * 2 * n * sizeof(float) + (size_t)((real ? n / 4 : n / 2) ? (real ? n / 4 : n / 2) : 1) * sizeof(double2);
#define WMX_MFFT_LAUNCH(K) \
    hipLaunchKernelGGL((mfft_kernel<K, false>), grid, block, lds, s, n_batch, n, m, tw_inner, tw_n, d_in_re, d_in_im, d_out_re, d_out_im, d_out_af, d_out_pf, 0u)
    switch (kind) {
        case 0: WMX_MFFT_LAUNCH(0); break;
        case 1: WMX_MFFT_LAUNCH(1); break;
        case 2: WMX_MFFT_LAUNCH(2); break;
        default: WMX_MFFT_LAUNCH(3); break;
    }
#undef WMX_MFFT_LAUNCH
    WMX_LAUNCH_CHECK();
    return 0;
}

extern "C" int wmx_mfft_stream(int n_streams, const float *d_in, unsigned in_len, float *d_pool, unsigned st_len, float *d_out_af,
                               float *d_out_pf, void *stream) {
    unsigned m;
    if (int rc = check_size(st_len, &m)) return rc;
    // the reference moves stream[inLen..2*inLen) down and appends inLen new samples behind it (math/fft.c:417-421):
    // 2*inLen must fit the pool or it reads past it
    if (n_streams < 0 || !d_in || !d_pool || in_len == 0 || 2 * in_len > st_len) {
        set_error("wmx_mfft_stream: n_streams %d, in_len %u, st_len %u", n_streams, in_len, st_len);
        return WMX_EINVAL;
    }
    if (n_streams == 0) return 0;
    const double2 *tw = nullptr;
    if (int rc = twiddles_for(st_len, &tw)) return rc;
    if (regs_path(m)) {
        const double2 *tw_staged = nullptr;
        if (int rc = staged_twiddles_for(st_len, &tw_staged)) return rc;
        const RegsLaunch L = regs_launch(0, m, n_streams);
#define WMX_MFFT_REGS(MC) \
    hipLaunchKernelGGL((mfft_regs_kernel<0, true, MC>), L.grid, L.block, L.lds, as_stream(stream), n_streams, tw_staged, tw, d_in, (const float *)nullptr, d_pool, (float *)nullptr, d_out_af, d_out_pf, in_len)
        switch (m) {
            case 5: WMX_MFFT_REGS(5); break;
            case 6: WMX_MFFT_REGS(6); break;
            case 7: WMX_MFFT_REGS(7); break;
            case 8: WMX_MFFT_REGS(8); break;
            case 9: WMX_MFFT_REGS(9); break;
            case 10: WMX_MFFT_REGS(10); break;
            case 11: WMX_MFFT_REGS(11); break;
            default: WMX_MFFT_REGS(12); break;
        }
#undef WMX_MFFT_REGS
        WMX_LAUNCH_CHECK();
        return 0;
    }
    const unsigned wpb = waves_per_block(st_len);
    const dim3 grid((unsigned)((n_streams + wpb - 1) / wpb)), block(64 * wpb);
    const size_t lds = (size_t)wpb * (2 * st_len + 2 * in_len) * sizeof(float) + (size_t)(st_len / 2) * sizeof(double2);
    hipLaunchKernelGGL((mfft_kernel<0, true>), grid, block, lds, as_stream(stream), n_streams, st_len, m, tw, tw, d_in, (const float *)nullptr,
                       d_pool, (float *)nullptr, d_out_af, d_out_pf, in_len);
    WMX_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ legacy signatures (math/fft.h:19-51): one transform,
// host arrays, NULLs as in the reference
namespace {
constexpr size_t kLegacyMappedMax = 1u << 20;  // bytes of staging up to which a legacy call goes through a mapped pinned buffer

int legacy(int kind, float *in_re, float *in_im, float *out_re, float *out_im, float *out_af, float *out_pf, unsigned n) {
    unsigned m;
    if (check_size(n, &m)) return -1;
    float *d = nullptr;
    const size_t bytes = (size_t)n * sizeof(float);
    // One transform of a few KB (the daemon's spectrum display: N = 1024, src/wmix.c:1124-1137): a device allocation and six copies
    // around the launch cost many times the work.  Pinned host memory mapped into the device instead, owned by the calling thread:
    // memcpy in, one launch, one synchronisation, memcpy out.
    static thread_local MapVec mv;
    hipStream_t ts = wmx::thread_stream();  // the calling thread's own non-blocking stream (wmx_internal.h)
    if (6 * bytes <= kLegacyMappedMax && mv.ensure(6 * bytes) == 0) {
        float *hst = reinterpret_cast<float *>(mv.host), *dv = reinterpret_cast<float *>(mv.dev);
        if (in_re) memcpy(hst, in_re, bytes);
        if (in_im) memcpy(hst + n, in_im, bytes);
        if (wmx_mfft(kind, 1, n, in_re ? dv : nullptr, in_im ? dv + n : nullptr, out_re ? dv + 2 * n : nullptr, out_im ? dv + 3 * n : nullptr,
                     out_af ? dv + 4 * n : nullptr, out_pf ? dv + 5 * n : nullptr, ts) != 0 ||
            hipStreamSynchronize(ts) != hipSuccess)
            return -1;
        float *outs[4] = {out_re, out_im, out_af, out_pf};
        for (int k = 0; k < 4; k++)
            if (outs[k]) memcpy(outs[k], hst + (size_t)(2 + k) * n, bytes);
        return 0;
    }
    if (hipMalloc(&d, 6 * bytes) != hipSuccess) return -1;
    float *d_ir = in_re ? d : nullptr, *d_ii = in_im ? d + n : nullptr;
    float *d_or = out_re ? d + 2 * n : nullptr, *d_oi = out_im ? d + 3 * n : nullptr;
    float *d_af = out_af ? d + 4 * n : nullptr, *d_pf = out_pf ? d + 5 * n : nullptr;
    int rc = 0;
    if (in_re && hipMemcpyAsync(d_ir, in_re, bytes, hipMemcpyHostToDevice, ts) != hipSuccess) rc = -1;
    if (in_im && hipMemcpyAsync(d_ii, in_im, bytes, hipMemcpyHostToDevice, ts) != hipSuccess) rc = -1;
    if (!rc) rc = wmx_mfft(kind, 1, n, d_ir, d_ii, d_or, d_oi, d_af, d_pf, ts);
    float *outs[4] = {out_re, out_im, out_af, out_pf}, *devs[4] = {d_or, d_oi, d_af, d_pf};
    for (int k = 0; k < 4 && !rc; k++)
        if (outs[k] && hipMemcpyAsync(outs[k], devs[k], bytes, hipMemcpyDeviceToHost, ts) != hipSuccess) rc = -1;
    if (hipStreamSynchronize(ts) != hipSuccess) rc = -1;  // also on failure: nothing of this call may still touch `d`
    (void)hipFree(d);
    return rc;
}
}  // namespace

extern "C" void FFT(float inReal[], float inImag[], float outReal[], float outImag[], float outAF[], float outPF[], unsigned int N) {
    legacy(0, inReal, inImag, outReal, outImag, outAF, outPF, N);
}
extern "C" void FFTR(float inReal[], float inImag[], float outReal[], float outImag[], float outAF[], float outPF[], unsigned int N) {
    legacy(1, inReal, inImag, outReal, outImag, outAF, outPF, N);
}
extern "C" void IFFT(float inReal[], float inImag[], float outReal[], float outImag[], unsigned int N) {
    legacy(2, inReal, inImag, outReal, outImag, nullptr, nullptr, N);
}
extern "C" void IFFTR(float inReal[], float inImag[], float outReal[], float outImag[], unsigned int N) {
    legacy(3, inReal, inImag, outReal, outImag, nullptr, nullptr, N);
}
extern "C" void fft_stream(float in[], unsigned int inLen, float stream[], unsigned int stLen, float outAF[], float outPF[]) {
    unsigned m;
    if (!in || !stream || inLen == 0 || 2 * inLen > stLen || check_size(stLen, &m)) return;
    float *d = nullptr;
    const size_t sb = (size_t)stLen * sizeof(float), ib = (size_t)inLen * sizeof(float);
    static thread_local MapVec mv;
    hipStream_t ts = wmx::thread_stream();
    if (3 * sb + ib <= kLegacyMappedMax && mv.ensure(3 * sb + ib) == 0) {
        float *hst = reinterpret_cast<float *>(mv.host), *dv = reinterpret_cast<float *>(mv.dev);
        memcpy(hst, stream, sb);
        memcpy(hst + 3 * (size_t)stLen, in, ib);
        if (wmx_mfft_stream(1, dv + 3 * (size_t)stLen, inLen, dv, stLen, outAF ? dv + stLen : nullptr, outPF ? dv + 2 * (size_t)stLen : nullptr,
                            ts) != 0 ||
            hipStreamSynchronize(ts) != hipSuccess)
            return;
        memcpy(stream, hst, sb);
        if (outAF) memcpy(outAF, hst + stLen, sb);
        if (outPF) memcpy(outPF, hst + 2 * (size_t)stLen, sb);
        return;
    }
    if (hipMalloc(&d, 3 * sb + ib) != hipSuccess) return;
    float *d_pool = d, *d_af = d + stLen, *d_pf = d + 2 * stLen, *d_in = d + 3 * stLen;
    bool ok = hipMemcpyAsync(d_pool, stream, sb, hipMemcpyHostToDevice, ts) == hipSuccess && hipMemcpyAsync(d_in, in, ib, hipMemcpyHostToDevice, ts) == hipSuccess;
    ok = ok && wmx_mfft_stream(1, d_in, inLen, d_pool, stLen, outAF ? d_af : nullptr, outPF ? d_pf : nullptr, ts) == 0;
    ok = ok && hipMemcpyAsync(stream, d_pool, sb, hipMemcpyDeviceToHost, ts) == hipSuccess;
    if (ok && outAF) ok = hipMemcpyAsync(outAF, d_af, sb, hipMemcpyDeviceToHost, ts) == hipSuccess;
    if (ok && outPF) ok = hipMemcpyAsync(outPF, d_pf, sb, hipMemcpyDeviceToHost, ts) == hipSuccess;
    (void)hipStreamSynchronize(ts);
    (void)hipFree(d);
}
