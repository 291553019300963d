// exec_half.hip -- does a wave64 vector instruction cost less when the upper 32 lanes (or all but one lane) are switched off?
// Time-boxed like issue_cost.hip: 4 waves per SIMD, every CU busy, v_add_f32 / v_fma_f32 / v_pk_add_f32 streams under
// EXEC = all lanes, the lower 32, lane 0 alone.
//   hipcc --offload-arch=gfx950 -O2 -o exec_half exec_half.hip && ./exec_half
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void k(float *out, unsigned long long *stamp, int ticks, unsigned long long mask) {
    extern __shared__ float lds[];
    if (ticks < 0) lds[threadIdx.x] = 0.f;
    float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float c = out[0] + 1.0000001f;
    const v2f cc = {c, c};
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), r_end = r0 + (unsigned long long)ticks;
    unsigned long long groups = 0;
    const unsigned long long full = __builtin_amdgcn_read_exec();
    do {
        asm volatile("s_mov_b64 exec, %0" : : "s"(mask));
#pragma unroll
        for (int rep = 0; rep < 32; rep++) {
            if constexpr (OP == 0)
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            else if constexpr (OP == 1)
                asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            else
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));
        }
        asm volatile("s_mov_b64 exec, %0" : : "s"(full));
        groups += 32;
    } while (__builtin_amdgcn_s_memrealtime() < r_end);
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *st = stamp + 2 * (blockIdx.x * 4 + (threadIdx.x >> 6));
        st[0] = groups;
        st[1] = r1 - r0;
    }
}
template <int OP>
double run(unsigned long long mask) {
    float *out;
    unsigned long long *stamp;
    const int blocks = 1024, waves = 4096;
    (void)hipMalloc(&out, 4 * (1 + 2048 * 1024));
    (void)hipMalloc(&stamp, 16 * waves);
    (void)hipMemset(out, 0, 4);
    const size_t lds = 40960 - 1024;
    (void)hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, out, stamp, 4000, mask);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, out, stamp, 40000, mask);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> st(2 * waves);
    (void)hipMemcpy(st.data(), stamp, 16 * waves, hipMemcpyDeviceToHost);
    double rate = 0;
    for (int w = 0; w < waves; w++) rate += (double)st[2 * w] * 8.0 / ((double)st[2 * w + 1] * 10.0);
    (void)hipFree(out);
    (void)hipFree(stamp);
    return 1024.0 / rate;  // ns of SIMD time per instruction
}
int main() {
    const unsigned long long masks[3] = {~0ull, 0xffffffffull, 1ull};
    const char *mn[3] = {"all 64 lanes", "lower 32 lanes", "lane 0 alone"};
    printf("{\n");
    for (int m = 0; m < 3; m++)
        printf(" \"%s\": {\"v_add_f32\": %.3f, \"v_fma_f32\": %.3f, \"v_pk_add_f32\": %.3f}%s\n", mn[m], run<0>(masks[m]), run<1>(masks[m]), run<2>(masks[m]),
               m < 2 ? "," : "");
    printf("}\n");
    return 0;
}
