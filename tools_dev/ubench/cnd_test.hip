// cnd_test.hip -- when is v_cndmask_b32 slow on gfx950?  (issue_cost.hip: 8 back-to-back VOP2 selects on vcc run at ~20 cycles each,
// the VOP3 form with the same mask at ~4.)  Sequences of selects behind ONE compare, in both encodings, with and without other
// instructions in between.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int M>
__global__ void k(float *out, int iters) {
    extern __shared__ float dyn_lds[];
    if (iters < 0) dyn_lds[threadIdx.x] = 0.f;
    float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float c = out[0] + 1.0000001f, c2 = out[0] + 2.f;
    const int msk = (threadIdx.x & 1) ? -1 : 0;
    for (int i = 0; i < iters; i++) {
        if constexpr (M == 0)  // 1 cmp + 8 e32 selects
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n"
                         "v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "vcc");
        else if constexpr (M == 1)  // 1 cmp + 8 e64 selects on vcc
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e64 %0, %8, %9, vcc\n v_cndmask_b32_e64 %1, %8, %9, vcc\n v_cndmask_b32_e64 %2, %8, %9, vcc\n v_cndmask_b32_e64 %3, %8, %9, vcc\n"
                         "v_cndmask_b32_e64 %4, %8, %9, vcc\n v_cndmask_b32_e64 %5, %8, %9, vcc\n v_cndmask_b32_e64 %6, %8, %9, vcc\n v_cndmask_b32_e64 %7, %8, %9, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "vcc");
        else if constexpr (M == 2)  // 4 x (cmp, select e32, select e32)
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "vcc");
        else if constexpr (M == 3)  // cmp, then e32 selects separated by plain adds (4 selects, 4 adds)
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_add_f32 %1, %1, %8\n v_cndmask_b32 %0, %8, %9, vcc\n v_add_f32 %3, %3, %8\n v_cndmask_b32 %2, %8, %9, vcc\n"
                         "v_add_f32 %5, %5, %8\n v_cndmask_b32 %4, %8, %9, vcc\n v_add_f32 %7, %7, %8\n v_cndmask_b32 %6, %8, %9, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "vcc");
        else if constexpr (M == 4)  // 8 e32 selects, vcc written by SALU before the group (s_mov_b64 vcc)
            asm volatile("s_mov_b64 vcc, 0x5555\n s_nop 4\n v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n"
                         "v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "vcc");
        else if constexpr (M == 5)  // 8 x (cmp, select e32) alternating: the fast case of pk_rate
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %8, %9, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %8, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %8, %9, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %8, %9, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "vcc");
        else if constexpr (M == 6)  // cmp into an SGPR pair, 8 e64 selects on it
            asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %8\n v_cndmask_b32_e64 %0, %8, %9, s[20:21]\n v_cndmask_b32_e64 %1, %8, %9, s[20:21]\n v_cndmask_b32_e64 %2, %8, %9, s[20:21]\n v_cndmask_b32_e64 %3, %8, %9, s[20:21]\n"
                         "v_cndmask_b32_e64 %4, %8, %9, s[20:21]\n v_cndmask_b32_e64 %5, %8, %9, s[20:21]\n v_cndmask_b32_e64 %6, %8, %9, s[20:21]\n v_cndmask_b32_e64 %7, %8, %9, s[20:21]"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "s20", "s21");
        else if constexpr (M == 7)  // selects whose two sources are the SAME kind as in the kernels: registers that other selects also read
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %1, %2, vcc\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_cndmask_b32 %5, %6, %7, vcc\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "vcc");
        else if constexpr (M == 8)  // cmp, 4 x (select e32, s_nop 0, select e32): does a scalar no-op between two selects help?
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %8, %9, vcc\n s_nop 0\n v_cndmask_b32 %1, %8, %9, vcc\n s_nop 0\n v_cndmask_b32 %2, %8, %9, vcc\n s_nop 0\n v_cndmask_b32 %3, %8, %9, vcc\n"
                         "s_nop 0\n v_cndmask_b32 %4, %8, %9, vcc\n s_nop 0\n v_cndmask_b32 %5, %8, %9, vcc\n s_nop 0\n v_cndmask_b32 %6, %8, %9, vcc\n s_nop 0\n v_cndmask_b32 %7, %8, %9, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "vcc");
        else if constexpr (M == 9)  // cmp, 4 x (select e32, select e64) alternating encodings
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32_e64 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32_e64 %3, %8, %9, vcc\n"
                         "v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32_e64 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32_e64 %7, %8, %9, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "vcc");
        else if constexpr (M == 10)  // 8 v_bfi_b32 with a lane mask in a VGPR: the select-free select
            asm volatile("v_bfi_b32 %0, %10, %8, %9\n v_bfi_b32 %1, %10, %8, %9\n v_bfi_b32 %2, %10, %8, %9\n v_bfi_b32 %3, %10, %8, %9\n"
                         "v_bfi_b32 %4, %10, %8, %9\n v_bfi_b32 %5, %10, %8, %9\n v_bfi_b32 %6, %10, %8, %9\n v_bfi_b32 %7, %10, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2), "v"(msk));
        else if constexpr (M == 11)  // cmp, 4 x (select e32, s_mov_b32, select e32)
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %8, %9, vcc\n s_mov_b32 s20, 0\n v_cndmask_b32 %1, %8, %9, vcc\n s_mov_b32 s20, 0\n v_cndmask_b32 %2, %8, %9, vcc\n s_mov_b32 s20, 0\n v_cndmask_b32 %3, %8, %9, vcc\n"
                         "s_mov_b32 s20, 0\n v_cndmask_b32 %4, %8, %9, vcc\n s_mov_b32 s20, 0\n v_cndmask_b32 %5, %8, %9, vcc\n s_mov_b32 s20, 0\n v_cndmask_b32 %6, %8, %9, vcc\n s_mov_b32 s20, 0\n v_cndmask_b32 %7, %8, %9, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "vcc", "s20");
    }
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int M>
void run(const char *name, int n_instr_per_iter) {
    float *out;
    (void)hipMalloc(&out, 4 * (1 + 512 * 1024));
    (void)hipMemset(out, 0, 4);
    const int iters = 20000, threads = 1024;
    const size_t lds = 96 * 1024;
    (void)hipFuncSetAttribute((const void *)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<M>, dim3(256), dim3(threads), lds, 0, out, 2000);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<M>, dim3(256), dim3(threads), lds, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %2d instr/iter: %6.2f ns of SIMD time per iteration = %5.2f ns per instruction (4 waves/SIMD)\n", name, n_instr_per_iter,
           ms * 1e6 / iters / 4, ms * 1e6 / iters / 4 / n_instr_per_iter);
    (void)hipFree(out);
}
int main() {
    run<0>("1 cmp + 8 select e32 (vcc)", 9);
    run<1>("1 cmp + 8 select e64 (vcc)", 9);
    run<2>("4 x (cmp, select e32, select e32)", 12);
    run<3>("cmp, 4 x (add, select e32)", 9);
    run<4>("s_mov vcc, 8 select e32", 10);
    run<5>("4 x (cmp, select e32)", 8);
    run<6>("cmp -> sgpr pair, 8 select e64 (sgpr)", 9);
    run<7>("cmp, select e32, add, add, select e32, add, add", 7);
    run<8>("cmp, 8 x select e32 with s_nop 0 between", 16);
    run<9>("cmp, 4 x (select e32, select e64)", 9);
    run<10>("8 v_bfi_b32", 8);
    run<11>("cmp, 8 x select e32 with s_mov_b32 between", 16);
    return 0;
}
