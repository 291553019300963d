// fft_debug.hip -- test hooks: the cross-lane FFT executors of the product kernels run stand-alone, one transform per
// wavefront, so that `-m gpu` tests can pin them directly against the reference's known answers (round-1 VERDICT: a
// regression in a5 / a13 used to surface only as an NS / AEC mismatch).  Same headers, same code paths as ns.hip / aec.hip /
// nsx.hip; nothing here is on the product path.
//   kind 0 / 1   WebRtc_rdft(128, +1 / -1)   fft_ooura.h LDS executor, makewt / makect tables      (NS at 8 kHz)
//   kind 2 / 3   WebRtc_rdft(256, +1 / -1)                                                        (NS at 16 / 32 kHz)
//   kind 4 / 5   aec_rdft_forward_128 / inverse_128: LDS executor with the frozen rdft_w tables   (AEC far kernel)
//   kind 6       aec_rdft_forward_128 through the register executor (fft64_regs in 16-lane groups + rdft128_fwd_bin_u)
//   kind 7       aec_rdft_inverse_128 through the one-point-per-lane executor (rdft128_inv_point_lanes + fft64_lanes)
//   kind 8 / 9   WebRtcSpl_RealForwardFFT / RealInverseFFT, order 7 (int16 data; kind 9 returns the scale in d_aux)
//   kind 10 / 11 the same, order 8                                                                 (NSX, AECM)
#include <vector>
#include "wmx_internal.h"
#include "fft_regs.h"
#include "spl_fx.h"
#include "fx_tables.h"

namespace wmx {
namespace {

template <int N, bool INV>
__global__ __launch_bounds__(64) void dbg_ooura_lds(const FftTables *__restrict__ tab, float *data, int n_batch) {
    __shared__ FftTables T;
    __shared__ float a[N + 4];
    const int lane = threadIdx.x;
    for (int i = lane; i < kFftTableWords; i += 64) reinterpret_cast<float *>(&T)[i] = reinterpret_cast<const float *>(tab)[i];
    float *x = data + (size_t)blockIdx.x * N;
    for (int i = lane; i < N; i += 64) a[i] = x[i];
    wave_sync();
    if (INV)
        rdft_inverse<N / 2>(a, &T, lane);
    else
        rdft_forward<N / 2>(a, &T, lane);
    wave_sync();
    for (int i = lane; i < N; i += 64) x[i] = a[i];
}

// four transforms per wave, exactly as aec.hip's aec_fft_fwd does them
__global__ __launch_bounds__(64) void dbg_aec_regs_fwd(const FftTables *__restrict__ tab, float *data, int n_batch) {
    __shared__ FftTables T;
    __shared__ float rows[4][132];
    const int lane = threadIdx.x, g = fft_group(lane), gl = fft_index(lane);
    for (int i = lane; i < kFftTableWords; i += 64) reinterpret_cast<float *>(&T)[i] = reinterpret_cast<const float *>(tab)[i];
    wave_sync();
    const int t = blockIdx.x * 4 + g;
    const bool live = t < n_batch;
    float *x = data + (size_t)(live ? t : 0) * 128;
    Cx v[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const int p = fft64_src_point(gl, m);
        v[m] = Cx{x[2 * p], x[2 * p + 1]};
    }
    fft64_regs<false>(v, &T, gl);
#pragma unroll
    for (int m = 0; m < 4; m++) *reinterpret_cast<float2 *>(&rows[g][2 * (gl + 16 * m)]) = make_float2(v[m].r, v[m].i);
    wave_sync();
    // every lane finishes bin `lane` of each of the four rows; output in the reference's packed layout
    const SplitLane cf = rdft128_fwd_coef(&T, lane);
    for (int r = 0; r < 4; r++) {
        const int tt = blockIdx.x * 4 + r;
        if (tt >= n_batch) break;
        const v2f b = rdft128_fwd_bin_u(rows[r], cf, lane);
        float *o = data + (size_t)tt * 128;
        const float nyq = rows[r][0] - rows[r][1];
        wave_sync();
        if (lane == 0) {
            o[0] = b.x;
            o[1] = nyq;
        } else {
            o[2 * lane] = b.x;
            o[2 * lane + 1] = b.y;
        }
    }
}

__global__ __launch_bounds__(64) void dbg_aec_lanes_inv(const FftTables *__restrict__ tab, float *data, int n_batch) {
    __shared__ FftTables T;
    const int lane = threadIdx.x;
    for (int i = lane; i < kFftTableWords; i += 64) reinterpret_cast<float *>(&T)[i] = reinterpret_cast<const float *>(tab)[i];
    wave_sync();
    float *x = data + (size_t)blockIdx.x * 128;
    v2f pt = rdft128_inv_point_lanes(v2f{x[2 * lane], x[2 * lane + 1]}, &T, lane);  // lane 0 carries (a[0], a[1]) = (bin 0, bin 64)
    pt = fft64_lanes<true>(pt, &T, lane);
    x[2 * lane] = pt.x;
    x[2 * lane + 1] = pt.y;
}

template <int ORDER, bool INV>
__global__ __launch_bounds__(64) void dbg_spl_fft(const SplTwiddles *__restrict__ tw, int16_t *data, int32_t *aux, int n_batch) {
    constexpr int N = 1 << ORDER;
    __shared__ SplTwiddles S;
    __shared__ int32_t cx[N];
    const int lane = threadIdx.x;
    for (int i = lane; i < (int)(sizeof(SplTwiddles) / 4); i += 64) reinterpret_cast<int32_t *>(&S)[i] = reinterpret_cast<const int32_t *>(tw)[i];
    int16_t *x = data + (size_t)blockIdx.x * (N + 2);
    if (!INV) {  // real_fft.c:46-70
        for (int i = lane; i < N; i += 64) cx[bitrev<ORDER>(i)] = (int32_t)(uint16_t)x[i];
        wave_sync();
        if constexpr (ORDER == 7)
            spl_cfft128<false>(cx, S, lane);
        else
            spl_cfft<ORDER, false>(cx, S, lane);
        for (int i = lane; i <= N / 2; i += 64) {
            x[2 * i] = lo16(cx[i]);
            x[2 * i + 1] = hi16(cx[i]);
        }
    } else {  // real_fft.c:72-100
        for (int b = lane; b <= N / 2; b += 64) {
            const int16_t re = x[2 * b], im = x[2 * b + 1];
            cx[bitrev<ORDER>(b)] = pack16(re, im);
            if (b > 0 && b < N / 2) cx[bitrev<ORDER>(N - b)] = pack16(re, (int16_t)-im);
        }
        wave_sync();
        int sc;
        if constexpr (ORDER == 7)
            sc = spl_cfft128<true>(cx, S, lane);
        else
            sc = spl_cfft<ORDER, true>(cx, S, lane);
        for (int i = lane; i < N; i += 64) x[i] = lo16(cx[i]);
        if (lane == 0) aux[blockIdx.x] = sc;
    }
}

}  // namespace
}  // namespace wmx

// d_data: [n_batch][n] float for kinds 0..7 (n = 128 or 256), [n_batch][n + 2] int16 for kinds 8..11; transformed in place.
// d_aux: [n_batch] int32, written by kinds 9 and 11 (the inverse transform's scale count); may be NULL otherwise.
extern "C" int wmx_debug_fft(int kind, int n_batch, void *d_data, int32_t *d_aux, void *stream) {
    using namespace wmx;
    if (kind < 0 || kind > 11 || n_batch < 1 || !d_data) return WMX_EINVAL;
    hipStream_t s = as_stream(stream);
    if (kind <= 7) {
        FftTables h;
        if (kind >= 4)
            fft_tables_aec128(&h);
        else
            fft_tables_ooura(kind < 2 ? 128 : 256, &h);
        FftTables *d = nullptr;
        WMX_HIP(hipMalloc(&d, sizeof(h)));
        WMX_HIP(hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice));
        float *x = static_cast<float *>(d_data);
        const dim3 grid((unsigned)n_batch), blk(64);
        switch (kind) {
            case 0: hipLaunchKernelGGL((dbg_ooura_lds<128, false>), grid, blk, 0, s, d, x, n_batch); break;
            case 1: hipLaunchKernelGGL((dbg_ooura_lds<128, true>), grid, blk, 0, s, d, x, n_batch); break;
            case 2: hipLaunchKernelGGL((dbg_ooura_lds<256, false>), grid, blk, 0, s, d, x, n_batch); break;
            case 3: hipLaunchKernelGGL((dbg_ooura_lds<256, true>), grid, blk, 0, s, d, x, n_batch); break;
            case 4: hipLaunchKernelGGL((dbg_ooura_lds<128, false>), grid, blk, 0, s, d, x, n_batch); break;
            case 5: hipLaunchKernelGGL((dbg_ooura_lds<128, true>), grid, blk, 0, s, d, x, n_batch); break;
            case 6: hipLaunchKernelGGL(dbg_aec_regs_fwd, dim3((unsigned)((n_batch + 3) / 4)), blk, 0, s, d, x, n_batch); break;
            default: hipLaunchKernelGGL(dbg_aec_lanes_inv, grid, blk, 0, s, d, x, n_batch); break;
        }
        const hipError_t e = hipGetLastError();
        (void)hipStreamSynchronize(s);
        (void)hipFree(d);
        if (e != hipSuccess) return hip_fail(e, "debug fft launch", __FILE__, __LINE__);
        return 0;
    }
    if ((kind == 9 || kind == 11) && !d_aux) return WMX_EINVAL;
    SplTwiddles *d_sin = nullptr;
    {
        SplTwiddles h;
        spl_twiddles(fx_spl_sin1024, &h);
        WMX_HIP(hipMalloc(&d_sin, sizeof(h)));
        WMX_HIP(hipMemcpy(d_sin, &h, sizeof(h), hipMemcpyHostToDevice));
    }
    int16_t *x = static_cast<int16_t *>(d_data);
    const dim3 grid((unsigned)n_batch), blk(64);
    switch (kind) {
        case 8: hipLaunchKernelGGL((dbg_spl_fft<7, false>), grid, blk, 0, s, d_sin, x, d_aux, n_batch); break;
        case 9: hipLaunchKernelGGL((dbg_spl_fft<7, true>), grid, blk, 0, s, d_sin, x, d_aux, n_batch); break;
        case 10: hipLaunchKernelGGL((dbg_spl_fft<8, false>), grid, blk, 0, s, d_sin, x, d_aux, n_batch); break;
        default: hipLaunchKernelGGL((dbg_spl_fft<8, true>), grid, blk, 0, s, d_sin, x, d_aux, n_batch); break;
    }
    const hipError_t e = hipGetLastError();
    (void)hipStreamSynchronize(s);
    (void)hipFree(d_sin);
    if (e != hipSuccess) return hip_fail(e, "debug fft launch", __FILE__, __LINE__);
    return 0;
}
