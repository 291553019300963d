// developer micro-benchmark: issue cost of v_pk_add_f32 / v_pk_mul_f32 / v_cndmask_b32_dpp vs plain fp32 VALU on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float *out, long long *cyc, int iters) {
    extern __shared__ float dyn_lds[];
    if (iters < 0) dyn_lds[threadIdx.x] = 0.f;
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float c = out[0];
    const v2f cc = {c, c};
    const unsigned long long smask = 0x5555555555555555ull;
    const int vmask = (threadIdx.x & 1) ? -1 : 0;
    const float c2 = out[0] + 1.f;
    const int idx = ((threadIdx.x ^ 1) & 63) << 2;
    const int waddr = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 2048, raddr = ((threadIdx.x ^ 5) & 63) * 8 + (threadIdx.x >> 6) * 2048;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 8 scalar adds
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (MODE == 1) {  // 4 packed adds (same flops)
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));
        } else if (MODE == 2) {  // 8 packed adds
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                         "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));
        } else if (MODE == 3) {  // 8 packed muls
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));
        } else if (MODE == 4) {  // 8 cndmask_dpp
            asm volatile("v_cndmask_b32_dpp %0, %1, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %1, %2, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_cndmask_b32_dpp %2, %3, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %3, %4, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_cndmask_b32_dpp %4, %5, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %5, %6, %7, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_cndmask_b32_dpp %6, %7, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %7, %0, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
        } else if (MODE == 5) {  // 8 plain v_mov_dpp + 8 cndmask (the old form)
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (MODE == 6) {  // 8 s_nop 0
            asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
        } else if (MODE == 7) {  // 8 s_mov_b64
            asm volatile("s_mov_b64 vcc, exec\n s_mov_b64 vcc, exec\n s_mov_b64 vcc, exec\n s_mov_b64 vcc, exec\n s_mov_b64 vcc, exec\n s_mov_b64 vcc, exec\n s_mov_b64 vcc, exec\n s_mov_b64 vcc, exec" ::: "vcc");
        } else if (MODE == 9) {  // v_mov_b32_dpp quad_perm, full masks
            asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 10) {  // v_mov_b32_dpp row_ror with bank mask
            asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %1, %2 row_ror:4 row_mask:0xf bank_mask:0xa\n"
                         "v_mov_b32_dpp %2, %3 row_ror:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %3, %4 row_ror:4 row_mask:0xf bank_mask:0xa\n"
                         "v_mov_b32_dpp %4, %5 row_ror:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %5, %6 row_ror:4 row_mask:0xf bank_mask:0xa\n"
                         "v_mov_b32_dpp %6, %7 row_ror:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %7, %0 row_ror:4 row_mask:0xf bank_mask:0xa"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 11) {  // v_add_f32_dpp
            asm volatile("s_nop 1\n v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %4, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %6, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 12) {  // plain v_cndmask_b32 (vcc)
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %2, %3, %4, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n"
                         "v_cndmask_b32 %4, %5, %6, vcc\n v_cndmask_b32 %5, %6, %7, vcc\n v_cndmask_b32 %6, %7, %0, vcc\n v_cndmask_b32 %7, %0, %1, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
        } else if (MODE == 13) {  // ds_bpermute_b32
            asm volatile("ds_bpermute_b32 %0, %8, %1\n ds_bpermute_b32 %1, %8, %2\n ds_bpermute_b32 %2, %8, %3\n ds_bpermute_b32 %3, %8, %4\n"
                         "ds_bpermute_b32 %4, %8, %5\n ds_bpermute_b32 %5, %8, %6\n ds_bpermute_b32 %6, %8, %7\n ds_bpermute_b32 %7, %8, %0\n s_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(idx));
        } else if (MODE == 14) {  // ds_swizzle_b32
            asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %1, %2 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %2, %3 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %3, %4 offset:swizzle(SWAP,1)\n"
                         "ds_swizzle_b32 %4, %5 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %5, %6 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %6, %7 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %7, %0 offset:swizzle(SWAP,1)\n s_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 15) {  // ds_write_b64 + ds_read_b64 round trips (4 + 4)
            asm volatile("ds_write_b64 %4, %0\n ds_write_b64 %4, %1 offset:512\n ds_write_b64 %4, %2 offset:1024\n ds_write_b64 %4, %3 offset:1536\n"
                         "ds_read_b64 %0, %5\n ds_read_b64 %1, %5 offset:512\n ds_read_b64 %2, %5 offset:1024\n ds_read_b64 %3, %5 offset:1536\n s_waitcnt lgkmcnt(0)"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(waddr), "v"(raddr) : "memory");
        } else if (MODE == 16) {  // v_mov_b32 (plain)
            asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 17) {  // v_permlane16_swap (gfx950)
            asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                         "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 18) {  // v_cndmask_b32_e64 with an SGPR-pair mask
            asm volatile("v_cndmask_b32_e64 %0, %1, %2, %8\n v_cndmask_b32_e64 %1, %2, %3, %8\n v_cndmask_b32_e64 %2, %3, %4, %8\n v_cndmask_b32_e64 %3, %4, %5, %8\n"
                         "v_cndmask_b32_e64 %4, %5, %6, %8\n v_cndmask_b32_e64 %5, %6, %7, %8\n v_cndmask_b32_e64 %6, %7, %0, %8\n v_cndmask_b32_e64 %7, %0, %1, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(smask));
        } else if (MODE == 19) {  // v_bfi_b32 with a VGPR lane mask
            asm volatile("v_bfi_b32 %0, %8, %1, %2\n v_bfi_b32 %1, %8, %2, %3\n v_bfi_b32 %2, %8, %3, %4\n v_bfi_b32 %3, %8, %4, %5\n"
                         "v_bfi_b32 %4, %8, %5, %6\n v_bfi_b32 %5, %8, %6, %7\n v_bfi_b32 %6, %8, %7, %0\n v_bfi_b32 %7, %8, %0, %1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(vmask));
        } else if (MODE == 20) {  // v_cmp_lt_f32 (writes vcc)
            asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %4\n"
                         "v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cmp_lt_f32 vcc, %7, %0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
        } else if (MODE == 21) {  // v_cndmask with distinct finite operands (values: small floats)
            asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n"
                         "v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "vcc");
        } else if (MODE == 22) {  // v_max_f32
            asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                         "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (MODE == 23) {  // 4 x (v_cmp -> vcc, v_cndmask <- vcc): what the compiler emits for a select
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "vcc");
        } else if (MODE == 24) {  // the same through an SGPR pair
            asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %8\n v_cndmask_b32_e64 %1, %1, %9, s[20:21]\n v_cmp_lt_f32_e64 s[22:23], %2, %8\n v_cndmask_b32_e64 %3, %3, %9, s[22:23]\n"
                         "v_cmp_lt_f32_e64 s[20:21], %4, %8\n v_cndmask_b32_e64 %5, %5, %9, s[20:21]\n v_cmp_lt_f32_e64 s[22:23], %6, %8\n v_cndmask_b32_e64 %7, %7, %9, s[22:23]"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "s20", "s21", "s22", "s23");
        } else if (MODE == 25) {  // v_cndmask_b32_e64 reading vcc explicitly (VOP3 encoding, mask operand = vcc)
            asm volatile("v_cndmask_b32_e64 %0, %1, %2, vcc\n v_cndmask_b32_e64 %1, %2, %3, vcc\n v_cndmask_b32_e64 %2, %3, %4, vcc\n v_cndmask_b32_e64 %3, %4, %5, vcc\n"
                         "v_cndmask_b32_e64 %4, %5, %6, vcc\n v_cndmask_b32_e64 %5, %6, %7, vcc\n v_cndmask_b32_e64 %6, %7, %0, vcc\n v_cndmask_b32_e64 %7, %0, %1, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
        } else if (MODE == 26) {  // v_div_fmas_f32 (reads vcc implicitly)
            asm volatile("v_div_fmas_f32 %0, %0, %8, %9\n v_div_fmas_f32 %1, %1, %8, %9\n v_div_fmas_f32 %2, %2, %8, %9\n v_div_fmas_f32 %3, %3, %8, %9\n"
                         "v_div_fmas_f32 %4, %4, %8, %9\n v_div_fmas_f32 %5, %5, %8, %9\n v_div_fmas_f32 %6, %6, %8, %9\n v_div_fmas_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "vcc");
        } else if (MODE == 27) {  // v_fma_f32 for comparison
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2));
        } else if (MODE == 28) {  // v_permlane32_swap (gfx950)
            asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                         "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 29 || MODE == 30 || MODE == 31) {  // 8 v_add_f32 with 1 / 16 / 32 active lanes: does a mostly empty EXEC cost less?
            asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, %9\n"
                         "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         "s_mov_b64 exec, s[20:21]"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(c), "s"(MODE == 29 ? 1ull : (MODE == 30 ? 0xffffull : 0xffffffffull)) : "s20", "s21");
        } else if (MODE == 8) {  // 8 v_pk_fma
            asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                         "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));
        }
    }
    long long t1 = clock64();
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void run(const char *name, int waves_per_simd) {
    // occupancy is pinned through LDS: one workgroup per CU (96 KB of dynamic LDS) for 1..4 waves/SIMD, two (64 KB) for 8
    float *out; long long *cyc;
    (void)hipMalloc(&out, 4 * (1 + 512 * 1024)); (void)hipMalloc(&cyc, 8);
    (void)hipMemset(out, 0, 4);
    const int iters = 40000;
    const int threads = waves_per_simd >= 4 ? 1024 : 256 * waves_per_simd;
    const int per_cu = waves_per_simd == 8 ? 2 : 1;
    const size_t lds = per_cu == 2 ? 64 * 1024 : 96 * 1024;
    (void)hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * per_cu), dim3(threads), lds, 0, out, cyc, 100);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * per_cu), dim3(threads), lds, 0, out, cyc, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n_instr = (double)iters * (MODE == 1 ? 4 : (MODE >= 29 && MODE <= 31 ? 34 : 8));
    printf("%-24s %d waves/SIMD: %6.2f ns per instr per wave, %6.3f ns per instr per SIMD (ticks/instr/wave %5.2f)\n", name, waves_per_simd,
           ms * 1e6 / n_instr, ms * 1e6 / n_instr / waves_per_simd, (double)c / n_instr);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int w = 1; w <= 4; w *= 4) {
        run<0>("v_add_f32", w); run<2>("v_pk_add_f32", w);
        run<4>("v_cndmask_b32_dpp", w); run<18>("v_cndmask_b32_e64 sgpr", w); run<19>("v_bfi_b32", w); run<20>("v_cmp_lt_f32 vcc", w); run<21>("v_cndmask indep", w); run<22>("v_max_f32", w);
        run<9>("v_mov_b32_dpp quad_perm", w); run<10>("v_mov_b32_dpp ror+bank", w); run<11>("v_add_f32_dpp", w); run<12>("v_cndmask_b32 vcc", w);
        run<13>("ds_bpermute_b32", w); run<14>("ds_swizzle_b32", w); run<15>("ds_write+read_b64 (8)", w); run<16>("v_mov_b32", w); run<17>("v_permlane16_swap", w); run<28>("v_permlane32_swap", w);
        run<23>("cmp+cndmask via vcc (8)", w); run<24>("cmp+cndmask via sgpr (8)", w); run<25>("v_cndmask_e64 vcc", w); run<26>("v_div_fmas_f32", w); run<27>("v_fma_f32", w);
        run<29>("v_add_f32 exec=1 lane", w); run<30>("v_add_f32 exec=16 lanes", w); run<31>("v_add_f32 exec=32 lanes", w);
    }
    return 0;
}
