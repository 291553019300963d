// issue_spread.hip -- companion of issue_cost.hip: is the wall-clock price of a vector instruction (HIP events around a grid
// that fills every SIMD with W waves) the price every SIMD pays, or an artefact of how the grid is dealt to the CUs?
// Every workgroup stamps s_memrealtime (100 MHz) when its first wave starts and ends its loop and s_memtime (shader clock)
// over the loop; the host prints, per class: the grid's span, the workgroups' own durations (min / median / max), how many
// started late (after the first one ended) and the cycles per instruction per SIMD from the workgroups' own clocks.
//   hipcc --offload-arch=gfx950 -O2 -o issue_spread issue_spread.hip && ./issue_spread [waves_per_simd]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int C>
__global__ void k(float *out, unsigned long long *stamp, int iters) {
    extern __shared__ float dyn_lds[];
    if (iters < 0) dyn_lds[threadIdx.x] = 0.f;
    float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float c = out[0] + 1.0000001f;
    const v2f cc = {c, c};
    int s0 = 0, s1 = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        if constexpr (C == 0)
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if constexpr (C == 1)
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));
        else if constexpr (C == 2)  // a vector and a scalar instruction alternating: does the scalar one cost the SIMD anything?
            asm volatile("v_add_f32 %0, %0, %8\n s_add_u32 %9, %9, %10\n v_add_f32 %1, %1, %8\n s_add_u32 %10, %10, %9\n v_add_f32 %2, %2, %8\n s_add_u32 %9, %9, %10\n v_add_f32 %3, %3, %8\n s_add_u32 %10, %10, %9\n"
                         "v_add_f32 %4, %4, %8\n s_add_u32 %9, %9, %10\n v_add_f32 %5, %5, %8\n s_add_u32 %10, %10, %9\n v_add_f32 %6, %6, %8\n s_add_u32 %9, %9, %10\n v_add_f32 %7, %7, %8\n s_add_u32 %10, %10, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "s"(s0), "s"(s1) : "scc");
        else if constexpr (C == 3)  // two vector instructions per scalar one (the near kernel's ratio)
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n s_add_u32 %9, %9, %10\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n s_add_u32 %10, %10, %9\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n s_add_u32 %9, %9, %10\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n s_add_u32 %10, %10, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "s"(s0), "s"(s1) : "scc");
        else if constexpr (C == 4)  // v_pk with a scalar between
            asm volatile("v_pk_add_f32 %0, %0, %4\n s_add_u32 %5, %5, %6\n v_pk_add_f32 %1, %1, %4\n s_add_u32 %6, %6, %5\n v_pk_add_f32 %2, %2, %4\n s_add_u32 %5, %5, %6\n v_pk_add_f32 %3, %3, %4\n s_add_u32 %6, %6, %5\n"
                         "v_pk_add_f32 %0, %0, %4\n s_add_u32 %5, %5, %6\n v_pk_add_f32 %1, %1, %4\n s_add_u32 %6, %6, %5\n v_pk_add_f32 %2, %2, %4\n s_add_u32 %5, %5, %6\n v_pk_add_f32 %3, %3, %4\n s_add_u32 %6, %6, %5"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc), "s"(s0), "s"(s1) : "scc");
        else if constexpr (C == 5)  // plain and packed alternating
            asm volatile("v_add_f32 %0, %0, %8\n v_pk_add_f32 %4, %4, %9\n v_add_f32 %1, %1, %8\n v_pk_add_f32 %5, %5, %9\n v_add_f32 %2, %2, %8\n v_pk_add_f32 %6, %6, %9\n v_add_f32 %3, %3, %8\n v_pk_add_f32 %7, %7, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c), "v"(cc));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(s0 + s1);
    if ((threadIdx.x & 63) == 0) {
        const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamp += 4 * wv - 4 * blockIdx.x;  // (the stores below index by blockIdx.x)
        stamp[4 * blockIdx.x + 0] = r0;
        stamp[4 * blockIdx.x + 1] = r1;
        stamp[4 * blockIdx.x + 2] = t1 - t0;
        unsigned xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned hwid = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        stamp[4 * blockIdx.x + 3] = ((unsigned long long)xcc << 32) | hwid;
    }
}

template <int C>
void run(const char *name, int per_cu, int vec_per_iter) {
    float *out;
    unsigned long long *stamp;
    const int wgs = 256 * per_cu, blocks = 4 * wgs, iters = 20000;  // `blocks` counts waves below: every wave stamps
    (void)hipMalloc(&out, 4 * (1 + 2048 * 1024));
    (void)hipMalloc(&stamp, 32 * blocks);
    (void)hipMemset(out, 0, 4);
    const size_t lds = ((160 * 1024) / per_cu) & ~(size_t)255;
    (void)hipFuncSetAttribute((const void *)k<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<C>, dim3(wgs), dim3(256), lds, 0, out, stamp, 2000);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<C>, dim3(wgs), dim3(256), lds, 0, out, stamp, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st(4 * blocks);
    (void)hipMemcpy(st.data(), stamp, 32 * blocks, hipMemcpyDeviceToHost);
    unsigned long long first = ~0ull, last = 0, first_end = ~0ull;
    std::vector<double> dur, cyc;
    for (int b = 0; b < blocks; b++) {
        first = std::min(first, st[4 * b]);
        last = std::max(last, st[4 * b + 1]);
        first_end = std::min(first_end, st[4 * b + 1]);
        dur.push_back((double)(st[4 * b + 1] - st[4 * b]) * 10.0);  // ns
        cyc.push_back((double)st[4 * b + 2]);
    }
    int late = 0;
    std::vector<int> per_xcc(8, 0);
    // waves per SIMD: HW_ID (gfx9 layout) simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0]
    std::vector<int> simd_load(8 * 8 * 2 * 16 * 4, 0);
    for (int b = 0; b < blocks; b++) {
        late += st[4 * b] > first_end;
        const unsigned xcc = (unsigned)(st[4 * b + 3] >> 32) & 7, hw = (unsigned)st[4 * b + 3];
        per_xcc[xcc]++;
        simd_load[(((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15)) * 4 + ((hw >> 4) & 3)]++;
    }
    std::vector<int> load_hist(17, 0);
    for (int v : simd_load)
        if (v) load_hist[std::min(v, 16)]++;
    // per XCC: median wave duration and its clock (do the eight dies run alike?)
    char xcc_txt[512];
    int xo = 0;
    for (int x = 0; x < 8; x++) {
        std::vector<std::pair<double, double>> v;
        for (int b = 0; b < blocks; b++)
            if (((st[4 * b + 3] >> 32) & 7) == (unsigned)x) v.push_back({dur[b], cyc[b]});
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        const auto m = v[v.size() / 2];
        xo += snprintf(xcc_txt + xo, sizeof xcc_txt - xo, "%s[%.4f, %.4f, %.4f, %.3f]", x ? ", " : "", v.front().first * 1e-6, m.first * 1e-6, v.back().first * 1e-6, m.second / m.first);
    }
    std::sort(dur.begin(), dur.end());
    std::sort(cyc.begin(), cyc.end());
    const double n = (double)iters * vec_per_iter;
    printf("{\"class\": \"%s\", \"waves_per_simd\": %d, \"event_ms\": %.4f, \"grid_span_ms\": %.4f, \"wave_ms\": [%.4f, %.4f, %.4f], \"late_waves\": %d, "
           "\"ns_per_vector_instruction_per_simd\": {\"events\": %.3f, \"median_wave\": %.3f}, \"cycles_per_vector_instruction_per_simd_median\": %.3f, "
           "\"clock_ghz_median\": %.3f, \"waves_per_xcc\": [%d, %d, %d, %d, %d, %d, %d, %d], \"simds_holding_1_to_10_waves\": [%d, %d, %d, %d, %d, %d, %d, %d, %d, %d], \"per_xcc_wave_ms_min_median_max_and_clock_ghz\": [%s]}\n",
           name, per_cu, ms, (double)(last - first) * 1e-5, dur.front() * 1e-6, dur[dur.size() / 2] * 1e-6, dur.back() * 1e-6, late, ms * 1e6 / n / per_cu,
           dur[dur.size() / 2] / n / per_cu, cyc[cyc.size() / 2] / n / per_cu, cyc[cyc.size() / 2] / dur[dur.size() / 2], per_xcc[0], per_xcc[1], per_xcc[2],
           per_xcc[3], per_xcc[4], per_xcc[5], per_xcc[6], per_xcc[7], load_hist[1], load_hist[2], load_hist[3], load_hist[4], load_hist[5], load_hist[6],
           load_hist[7], load_hist[8], load_hist[9], load_hist[10], xcc_txt);
    (void)hipFree(out);
    (void)hipFree(stamp);
}

int main(int argc, char **argv) {
    const int w = argc > 1 ? atoi(argv[1]) : 4;
    run<0>("v_add_f32", w, 8);
    run<1>("v_pk_add_f32", w, 8);
    run<2>("v_add_f32 + s_add_u32 alternating", w, 8);
    run<3>("2 v_add_f32 per s_add_u32", w, 8);
    run<4>("v_pk_add_f32 + s_add_u32 alternating", w, 8);
    run<5>("v_add_f32 + v_pk_add_f32 alternating", w, 8);
    return 0;
}
