// issue_cost.hip -- what one wave64 instruction of each CLASS costs its SIMD on gfx950, measured.
//
// The per-stream DSP kernels of this library are bound by vector-instruction issue, not by HBM (DESIGN_HISTORY.md section 5).  Round 2
// priced every VALU instruction at 4 cycles and called the product a floor; the guide (MI355X_MICROARCH.md, "Per-instruction
// cycle constants") says a wave64 v_fma_f32 occupies a SIMD-32 for 2 cycles when another wave can issue beside it and 4 when a
// wave is alone, and that packed / DPP / transcendental / fp64 forms cost more.  This program measures those prices in the
// regime the kernels run in -- W resident waves per SIMD, every CU busy, long dependent-free streams of one instruction -- and
// prints them as JSON: ns and shader cycles of SIMD time per instruction.  tools_dev/issue_model.py multiplies them with a
// kernel's ISA class histogram.
//
// How (second version).  Every wave runs its instruction stream for a FIXED TIME (s_memrealtime, 100 MHz) and counts what it
// got through; a SIMD's throughput is the sum over its W waves, the price of an instruction its inverse.  The first version
// gave every wave a fixed number of instructions and divided the grid's wall time by it: the SIMD's arbiter is not fair (the
// oldest wave runs at its own limit of one instruction per ~8 cycles, the youngest gets what is left), so the waves of a SIMD
// finish up to 1.5 x apart and the last one runs the end of the grid alone -- the wall time priced a plain v_add_f32 at 3
// cycles where the SIMD sustains one per 2.1-2.3 (tools_dev/ubench/issue_spread.hip shows the spread).  Mixed streams
// (vector + scalar, vector + LDS) are classes of their own: the price of a scalar instruction BESIDE vector work is what the
// kernels pay.
//
//   hipcc --offload-arch=gfx950 -O2 -o issue_cost issue_cost.hip && ./issue_cost [waves_per_simd ...] > issue_costs.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

enum Cls {
    V_ADD_F32, V_MUL_F32, V_FMA_F32, V_MOV_B32, V_CNDMASK, V_CNDMASK_PAIR, V_CMP_F32, V_ADD_U32, V_LSHL_ADD, V_MUL_LO_U32, V_MAD_U32_U24, V_MUL_U32_U24,
    V_PK_ADD_F32, V_PK_MUL_F32, V_PK_FMA_F32, V_MOV_DPP, V_ADD_DPP, V_PERMLANE16_SWAP, V_PERMLANE32_SWAP, V_READLANE, V_READFIRSTLANE,
    V_RCP_F32, V_SQRT_F32, V_RSQ_F32, V_EXP_F32, V_LOG_F32, V_ADD_F64, V_MUL_F64, V_FMA_F64, V_RCP_F64, V_CVT_F32_I32, V_CVT_F64_F32,
    V_DIV_SCALE_F32, V_DIV_FMAS_F32, V_DIV_FIXUP_F32, V_LDEXP_F32, V_MAX_F32, V_BFE_I32, V_AND_B32, V_LSHLREV_B64, S_NOP0, S_MOV_B32,
    S_ADD_U32, MIX_V_S11, MIX_V_S21, MIX_V_S41, MIX_V_NOP21, MIX_V_WAIT21, MIX_PK_S21, MIX_V_PK11, MIX_V_DS41, DS_READ_B32, DS_READ_B64, DS_READ_B128, DS_WRITE_B32, DS_WRITE_B64, DS_BPERMUTE, N_CLS
};
static const char *kName[N_CLS] = {
    "v_add_f32", "v_mul_f32", "v_fma_f32", "v_mov_b32", "v_cndmask_b32", "v_cndmask_b32_vop2_behind_vop2", "v_cmp_f32", "v_add_u32", "v_lshl_add_u32", "v_mul_lo_u32",
    "v_mad_u32_u24", "v_mul_u32_u24", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_mov_b32_dpp", "v_add_f32_dpp", "v_permlane16_swap",
    "v_permlane32_swap", "v_readlane_b32", "v_readfirstlane_b32", "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32",
    "v_add_f64", "v_mul_f64", "v_fma_f64", "v_rcp_f64", "v_cvt_f32_i32", "v_cvt_f64_f32", "v_div_scale_f32", "v_div_fmas_f32",
    "v_div_fixup_f32", "v_ldexp_f32", "v_max_f32", "v_bfe_i32", "v_and_b32", "v_lshlrev_b64", "s_nop", "s_mov_b32", "s_add_u32", "mix: 8 v_add_f32 + 8 s_add_u32 (per 8)", "mix: 8 v_add_f32 + 4 s_add_u32 (per 8)", "mix: 8 v_add_f32 + 2 s_add_u32 (per 8)",
    "mix: 8 v_add_f32 + 4 s_nop (per 8)", "mix: 8 v_add_f32 + 4 s_waitcnt (per 8)", "mix: 8 v_pk_add_f32 + 4 s_add_u32 (per 8)", "mix: 4 v_add_f32 + 4 v_pk_add_f32 (per 8)",
    "mix: 8 v_add_f32 + 2 ds_read_b64 (per 8)", "ds_read_b32",
    "ds_read_b64", "ds_read_b128", "ds_write_b32", "ds_write_b64", "ds_bpermute_b32"};

constexpr int kRep = 32;  // asm groups (8 instructions each) between two looks at the clock
template <int C>
__global__ void k(float *out, unsigned long long *stamp, int ticks) {
    extern __shared__ float dyn_lds[];
    if (ticks < 0) dyn_lds[threadIdx.x] = 0.f;
    float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    const float c = out[0] + 1.0000001f, c2 = out[0] + 2.f;
    const v2f cc = {c, c};
    const double dc = c;
    int s0 = 0, s1 = 0;
    const int laddr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 2048;
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f q = {a0, a1, a2, a3};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long r_end = r0 + (unsigned long long)ticks;
    unsigned long long groups = 0;
    do {
#pragma unroll
      for (int rep = 0; rep < kRep; rep++) {
        if constexpr (C == V_ADD_F32)
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if constexpr (C == V_MUL_F32)
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if constexpr (C == V_FMA_F32)
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if constexpr (C == V_MOV_B32)
            asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if constexpr (C == V_CNDMASK)
            // a select as the kernels mostly have it: the VOP3 encoding, or a VOP2 one behind some other vector instruction (same price)
            asm volatile("v_cndmask_b32_e64 %0, %8, %9, vcc\n v_cndmask_b32_e64 %1, %8, %9, vcc\n v_cndmask_b32_e64 %2, %8, %9, vcc\n v_cndmask_b32_e64 %3, %8, %9, vcc\n v_cndmask_b32_e64 %4, %8, %9, vcc\n v_cndmask_b32_e64 %5, %8, %9, vcc\n v_cndmask_b32_e64 %6, %8, %9, vcc\n v_cndmask_b32_e64 %7, %8, %9, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "vcc");
        else if constexpr (C == V_CNDMASK_PAIR)
            // a VOP2 select (implicit vcc) directly behind another VOP2 select: ~20 cycles of SIMD time whatever the occupancy
            // (tools_dev/ubench/cnd_test.hip has the variants: a scalar instruction in between does not help, a vector one does)
            asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "vcc");
        else if constexpr (C == V_CMP_F32)
            asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %4\n v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cmp_lt_f32 vcc, %7, %0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
        else if constexpr (C == V_ADD_U32)
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if constexpr (C == V_LSHL_ADD)
            asm volatile("v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n v_lshl_add_u32 %4, %4, 2, %8\n v_lshl_add_u32 %5, %5, 2, %8\n v_lshl_add_u32 %6, %6, 2, %8\n v_lshl_add_u32 %7, %7, 2, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if constexpr (C == V_MUL_LO_U32)
            asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if constexpr (C == V_MAD_U32_U24)
            asm volatile("v_mad_u32_u24 %0, %0, %8, %8\n v_mad_u32_u24 %1, %1, %8, %8\n v_mad_u32_u24 %2, %2, %8, %8\n v_mad_u32_u24 %3, %3, %8, %8\n v_mad_u32_u24 %4, %4, %8, %8\n v_mad_u32_u24 %5, %5, %8, %8\n v_mad_u32_u24 %6, %6, %8, %8\n v_mad_u32_u24 %7, %7, %8, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if constexpr (C == V_MUL_U32_U24)
            asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if constexpr (C == V_PK_ADD_F32)
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));
        else if constexpr (C == V_PK_MUL_F32)
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));
        else if constexpr (C == V_PK_FMA_F32)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));
        else if constexpr (C == V_MOV_DPP)
            asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %1, %2 row_ror:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %2, %3 row_ror:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %3, %4 row_ror:4 row_mask:0xf bank_mask:0xa\n"
                         "v_mov_b32_dpp %4, %5 row_ror:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %5, %6 row_ror:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %6, %7 row_ror:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %7, %0 row_ror:4 row_mask:0xf bank_mask:0xa"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if constexpr (C == V_ADD_DPP)
            asm volatile("s_nop 1\n v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %4, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if constexpr (C == V_PERMLANE16_SWAP)
            asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if constexpr (C == V_PERMLANE32_SWAP)
            asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if constexpr (C == V_READLANE)
            asm volatile("v_readlane_b32 %0, %2, 3\n v_readlane_b32 %1, %3, 5\n v_readlane_b32 %0, %4, 7\n v_readlane_b32 %1, %5, 9\n v_readlane_b32 %0, %6, 11\n v_readlane_b32 %1, %7, 13\n v_readlane_b32 %0, %8, 15\n v_readlane_b32 %1, %9, 17"
                         : "+s"(s0), "+s"(s1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        else if constexpr (C == V_READFIRSTLANE)
            asm volatile("v_readfirstlane_b32 %0, %2\n v_readfirstlane_b32 %1, %3\n v_readfirstlane_b32 %0, %4\n v_readfirstlane_b32 %1, %5\n v_readfirstlane_b32 %0, %6\n v_readfirstlane_b32 %1, %7\n v_readfirstlane_b32 %0, %8\n v_readfirstlane_b32 %1, %9"
                         : "+s"(s0), "+s"(s1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
#define TRANS(op)                                                                                                                           \
    asm volatile(op " %0, %0\n " op " %1, %1\n " op " %2, %2\n " op " %3, %3\n " op " %4, %4\n " op " %5, %5\n " op " %6, %6\n " op " %7, %7" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
        else if constexpr (C == V_RCP_F32) TRANS("v_rcp_f32");
        else if constexpr (C == V_SQRT_F32) TRANS("v_sqrt_f32");
        else if constexpr (C == V_RSQ_F32) TRANS("v_rsq_f32");
        else if constexpr (C == V_EXP_F32) TRANS("v_exp_f32");
        else if constexpr (C == V_LOG_F32) TRANS("v_log_f32");
        else if constexpr (C == V_CVT_F32_I32) TRANS("v_cvt_f32_i32");
        else if constexpr (C == V_ADD_F64)
            asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dc));
        else if constexpr (C == V_MUL_F64)
            asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dc));
        else if constexpr (C == V_FMA_F64)
            asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dc));
        else if constexpr (C == V_RCP_F64)
            asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        else if constexpr (C == V_CVT_F64_F32)
            asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n v_cvt_f64_f32 %0, %5\n v_cvt_f64_f32 %1, %6\n v_cvt_f64_f32 %2, %7\n v_cvt_f64_f32 %3, %4"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        else if constexpr (C == V_DIV_SCALE_F32)
            asm volatile("v_div_scale_f32 %0, vcc, %0, %8, %0\n v_div_scale_f32 %1, vcc, %1, %8, %1\n v_div_scale_f32 %2, vcc, %2, %8, %2\n v_div_scale_f32 %3, vcc, %3, %8, %3\n v_div_scale_f32 %4, vcc, %4, %8, %4\n v_div_scale_f32 %5, vcc, %5, %8, %5\n v_div_scale_f32 %6, vcc, %6, %8, %6\n v_div_scale_f32 %7, vcc, %7, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
        else if constexpr (C == V_DIV_FMAS_F32)
            asm volatile("v_div_fmas_f32 %0, %0, %8, %8\n v_div_fmas_f32 %1, %1, %8, %8\n v_div_fmas_f32 %2, %2, %8, %8\n v_div_fmas_f32 %3, %3, %8, %8\n v_div_fmas_f32 %4, %4, %8, %8\n v_div_fmas_f32 %5, %5, %8, %8\n v_div_fmas_f32 %6, %6, %8, %8\n v_div_fmas_f32 %7, %7, %8, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
        else if constexpr (C == V_DIV_FIXUP_F32)
            asm volatile("v_div_fixup_f32 %0, %0, %8, %8\n v_div_fixup_f32 %1, %1, %8, %8\n v_div_fixup_f32 %2, %2, %8, %8\n v_div_fixup_f32 %3, %3, %8, %8\n v_div_fixup_f32 %4, %4, %8, %8\n v_div_fixup_f32 %5, %5, %8, %8\n v_div_fixup_f32 %6, %6, %8, %8\n v_div_fixup_f32 %7, %7, %8, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if constexpr (C == V_LDEXP_F32)
            asm volatile("v_ldexp_f32 %0, %0, 1\n v_ldexp_f32 %1, %1, 1\n v_ldexp_f32 %2, %2, 1\n v_ldexp_f32 %3, %3, 1\n v_ldexp_f32 %4, %4, 1\n v_ldexp_f32 %5, %5, 1\n v_ldexp_f32 %6, %6, 1\n v_ldexp_f32 %7, %7, 1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if constexpr (C == V_MAX_F32)
            asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if constexpr (C == V_BFE_I32)
            asm volatile("v_bfe_i32 %0, %0, 1, 16\n v_bfe_i32 %1, %1, 1, 16\n v_bfe_i32 %2, %2, 1, 16\n v_bfe_i32 %3, %3, 1, 16\n v_bfe_i32 %4, %4, 1, 16\n v_bfe_i32 %5, %5, 1, 16\n v_bfe_i32 %6, %6, 1, 16\n v_bfe_i32 %7, %7, 1, 16"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if constexpr (C == V_AND_B32)
            asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if constexpr (C == V_LSHLREV_B64)
            asm volatile("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3\n v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        else if constexpr (C == S_NOP0)
            asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
        else if constexpr (C == S_MOV_B32)
            asm volatile("s_mov_b32 %0, %1\n s_mov_b32 %1, %0\n s_mov_b32 %0, %1\n s_mov_b32 %1, %0\n s_mov_b32 %0, %1\n s_mov_b32 %1, %0\n s_mov_b32 %0, %1\n s_mov_b32 %1, %0" : "+s"(s0), "+s"(s1));
        else if constexpr (C == S_ADD_U32)
            asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %0" : "+s"(s0), "+s"(s1) : : "scc");
        else if constexpr (C == MIX_V_S11)
            asm volatile("v_add_f32 %0, %0, %10\n s_add_u32 %8, %8, %9\n v_add_f32 %1, %1, %10\n s_add_u32 %9, %9, %8\n v_add_f32 %2, %2, %10\n s_add_u32 %8, %8, %9\n v_add_f32 %3, %3, %10\n s_add_u32 %9, %9, %8\n"
                         "v_add_f32 %4, %4, %10\n s_add_u32 %8, %8, %9\n v_add_f32 %5, %5, %10\n s_add_u32 %9, %9, %8\n v_add_f32 %6, %6, %10\n s_add_u32 %8, %8, %9\n v_add_f32 %7, %7, %10\n s_add_u32 %9, %9, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : "v"(c) : "scc");
        else if constexpr (C == MIX_V_S21)
            asm volatile("v_add_f32 %0, %0, %10\n v_add_f32 %1, %1, %10\n s_add_u32 %8, %8, %9\n v_add_f32 %2, %2, %10\n v_add_f32 %3, %3, %10\n s_add_u32 %9, %9, %8\n"
                         "v_add_f32 %4, %4, %10\n v_add_f32 %5, %5, %10\n s_add_u32 %8, %8, %9\n v_add_f32 %6, %6, %10\n v_add_f32 %7, %7, %10\n s_add_u32 %9, %9, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : "v"(c) : "scc");
        else if constexpr (C == MIX_V_S41)
            asm volatile("v_add_f32 %0, %0, %10\n v_add_f32 %1, %1, %10\n v_add_f32 %2, %2, %10\n v_add_f32 %3, %3, %10\n s_add_u32 %8, %8, %9\n"
                         "v_add_f32 %4, %4, %10\n v_add_f32 %5, %5, %10\n v_add_f32 %6, %6, %10\n v_add_f32 %7, %7, %10\n s_add_u32 %9, %9, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : "v"(c) : "scc");
        else if constexpr (C == MIX_V_NOP21)
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n s_nop 0\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n s_nop 0\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n s_nop 0\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n s_nop 0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if constexpr (C == MIX_V_WAIT21)
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n s_waitcnt lgkmcnt(0)\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n s_waitcnt vmcnt(0)\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n s_waitcnt lgkmcnt(0)\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n s_waitcnt vmcnt(0)"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if constexpr (C == MIX_PK_S21)
            asm volatile("v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %1, %1, %6\n s_add_u32 %4, %4, %5\n v_pk_add_f32 %2, %2, %6\n v_pk_add_f32 %3, %3, %6\n s_add_u32 %5, %5, %4\n"
                         "v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %1, %1, %6\n s_add_u32 %4, %4, %5\n v_pk_add_f32 %2, %2, %6\n v_pk_add_f32 %3, %3, %6\n s_add_u32 %5, %5, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+s"(s0), "+s"(s1) : "v"(cc) : "scc");
        else if constexpr (C == MIX_V_PK11)
            asm volatile("v_add_f32 %0, %0, %8\n v_pk_add_f32 %4, %4, %9\n v_add_f32 %1, %1, %8\n v_pk_add_f32 %5, %5, %9\n v_add_f32 %2, %2, %8\n v_pk_add_f32 %6, %6, %9\n v_add_f32 %3, %3, %8\n v_pk_add_f32 %7, %7, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c), "v"(cc));
        else if constexpr (C == MIX_V_DS41)
            asm volatile("v_add_f32 %0, %0, %10\n v_add_f32 %1, %1, %10\n v_add_f32 %2, %2, %10\n v_add_f32 %3, %3, %10\n ds_read_b64 %8, %11\n"
                         "v_add_f32 %4, %4, %10\n v_add_f32 %5, %5, %10\n v_add_f32 %6, %6, %10\n v_add_f32 %7, %7, %10\n ds_read_b64 %9, %11 offset:1024\n s_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0), "+v"(p1) : "v"(c), "v"(laddr) : "memory");
        else if constexpr (C == DS_READ_B32)
            asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4\n ds_read_b32 %2, %8 offset:8\n ds_read_b32 %3, %8 offset:12\n ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1028\n ds_read_b32 %6, %8 offset:1032\n ds_read_b32 %7, %8 offset:1036\n s_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(laddr) : "memory");
        else if constexpr (C == DS_READ_B64)
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1032\n ds_read_b64 %0, %4 offset:2048\n ds_read_b64 %1, %4 offset:2056\n ds_read_b64 %2, %4 offset:3072\n ds_read_b64 %3, %4 offset:3080\n s_waitcnt lgkmcnt(0)"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(laddr) : "memory");
        else if constexpr (C == DS_READ_B128)
            asm volatile("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:1024\n ds_read_b128 %0, %1 offset:2048\n ds_read_b128 %0, %1 offset:3072\n ds_read_b128 %0, %1 offset:4096\n ds_read_b128 %0, %1 offset:5120\n ds_read_b128 %0, %1 offset:6144\n ds_read_b128 %0, %1 offset:7168\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(q) : "v"(laddr) : "memory");
        else if constexpr (C == DS_WRITE_B32)
            asm volatile("ds_write_b32 %8, %0\n ds_write_b32 %8, %1 offset:4\n ds_write_b32 %8, %2 offset:8\n ds_write_b32 %8, %3 offset:12\n ds_write_b32 %8, %4 offset:1024\n ds_write_b32 %8, %5 offset:1028\n ds_write_b32 %8, %6 offset:1032\n ds_write_b32 %8, %7 offset:1036\n s_waitcnt lgkmcnt(0)"
                         : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(laddr) : "memory");
        else if constexpr (C == DS_WRITE_B64)
            asm volatile("ds_write_b64 %4, %0\n ds_write_b64 %4, %1 offset:8\n ds_write_b64 %4, %2 offset:1024\n ds_write_b64 %4, %3 offset:1032\n ds_write_b64 %4, %0 offset:2048\n ds_write_b64 %4, %1 offset:2056\n ds_write_b64 %4, %2 offset:3072\n ds_write_b64 %4, %3 offset:3080\n s_waitcnt lgkmcnt(0)"
                         : : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(laddr) : "memory");
        else if constexpr (C == DS_BPERMUTE)
            asm volatile("ds_bpermute_b32 %0, %8, %1\n ds_bpermute_b32 %1, %8, %2\n ds_bpermute_b32 %2, %8, %3\n ds_bpermute_b32 %3, %8, %4\n ds_bpermute_b32 %4, %8, %5\n ds_bpermute_b32 %5, %8, %6\n ds_bpermute_b32 %6, %8, %7\n ds_bpermute_b32 %7, %8, %0\n s_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(laddr));
      }
      groups += kRep;
    } while (__builtin_amdgcn_s_memrealtime() < r_end);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                                     (float)(d0 + d1 + d2 + d3) + (float)(s0 + s1) + q.x + q.y + q.z + q.w;
    if ((threadIdx.x & 63) == 0) {  // every wave: instruction groups done, its own duration (100 MHz ticks) and shader cycles
        unsigned long long *st = stamp + 4 * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
        st[0] = groups;
        st[1] = r1 - r0;
        st[2] = t1 - t0;
        st[3] = r0;
    }
}

struct Res {
    double ns_simd, cyc_simd, ghz, resident;  // resident: waves per SIMD that were running together (must equal W)
};

template <int C>
Res run(int waves_per_simd) {
    // occupancy is pinned through dynamic LDS, in the shape the library's kernels have: four-wave workgroups (one wave per SIMD),
    // W of them per CU
    float *out;
    unsigned long long *stamp;
    const int per_cu = waves_per_simd, threads = 256, blocks = 256 * per_cu, waves = blocks * 4;
    (void)hipMalloc(&out, 4 * (1 + 2048 * 1024));
    (void)hipMalloc(&stamp, 32 * (size_t)waves);
    (void)hipMemset(out, 0, 4);
    const int ticks = 40000;  // 0.4 ms per wave
    // W workgroups fit a CU's 160 KB, W + 1 do not; a little under the even share, in case the allocator rounds a request up
    const size_t lds = (((160 * 1024) / per_cu) & ~(size_t)255) - 1024;
    (void)hipFuncSetAttribute((const void *)k<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k<C>, dim3(blocks), dim3(threads), lds, 0, out, stamp, 4000);  // clock up
    hipLaunchKernelGGL(k<C>, dim3(blocks), dim3(threads), lds, 0, out, stamp, ticks);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> st(4 * (size_t)waves);
    (void)hipMemcpy(st.data(), stamp, 32 * (size_t)waves, hipMemcpyDeviceToHost);
    // a wave's rate = its instructions / its own duration; a SIMD's throughput = the sum over its waves; all SIMDs alike
    double rate_ns = 0, rate_cyc = 0, cyc = 0, tk = 0;
    unsigned long long first = ~0ull, last = 0;
    for (int w = 0; w < waves; w++) {
        const double n = (double)st[4 * w] * 8.0;
        rate_ns += n / ((double)st[4 * w + 1] * 10.0);
        rate_cyc += n / (double)st[4 * w + 2];
        cyc += (double)st[4 * w + 2];
        tk += (double)st[4 * w + 1];
        first = first < st[4 * w + 3] ? first : st[4 * w + 3];
        last = last > st[4 * w + 3] + st[4 * w + 1] ? last : st[4 * w + 3] + st[4 * w + 1];
    }
    Res r;
    r.ns_simd = 1024.0 / rate_ns;   // SIMD time per wave64 instruction
    r.cyc_simd = 1024.0 / rate_cyc;
    r.ghz = cyc / (tk * 10.0);
    r.resident = tk / ((double)(last - first) * 1024.0);
    (void)hipFree(out);
    (void)hipFree(stamp);
    return r;
}

template <int C>
void all(std::vector<std::vector<Res>> &tab, const std::vector<int> &ws) {
    if constexpr (C < N_CLS) {
        std::vector<Res> row;
        for (int w : ws) row.push_back(run<C>(w));
        tab.push_back(row);
        all<C + 1>(tab, ws);
    }
}

int main(int argc, char **argv) {
    std::vector<int> ws;
    for (int i = 1; i < argc; i++) ws.push_back(atoi(argv[i]));
    if (ws.empty()) ws = {1, 2, 4, 5, 8};
    std::vector<std::vector<Res>> tab;
    all<0>(tab, ws);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    printf("{\n \"device\": \"%s\", \"arch\": \"%s\", \"unit\": \"SIMD time per wave64 instruction = 1 / (sum over the SIMD's waves of instructions per ns, every wave running for a fixed 0.4 ms): ns and shader cycles (s_memtime)\", \"method\": \"time-boxed waves, v2\",\n", p.name, p.gcnArchName);
    printf(" \"waves_per_simd\": [");
    for (size_t i = 0; i < ws.size(); i++) printf("%s%d", i ? ", " : "", ws[i]);
    printf("],\n \"classes\": {\n");
    for (int c = 0; c < N_CLS; c++) {
        printf("  \"%s\": {\"ns\": [", kName[c]);
        for (size_t i = 0; i < ws.size(); i++) printf("%s%.3f", i ? ", " : "", tab[c][i].ns_simd);
        printf("], \"cycles\": [");
        for (size_t i = 0; i < ws.size(); i++) printf("%s%.2f", i ? ", " : "", tab[c][i].cyc_simd);
        printf("], \"clock_ghz\": [");
        for (size_t i = 0; i < ws.size(); i++) printf("%s%.3f", i ? ", " : "", tab[c][i].ghz);
        printf("], \"resident_waves_per_simd\": [");
        for (size_t i = 0; i < ws.size(); i++) printf("%s%.2f", i ? ", " : "", tab[c][i].resident);
        printf("]}%s\n", c + 1 < N_CLS ? "," : "");
    }
    printf(" }\n}\n");
    return 0;
}
