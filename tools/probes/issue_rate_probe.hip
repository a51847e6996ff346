// Developer probe (not part of the library): issue cost of single gfx950 instructions, measured as kernel time / instructions
// per wave, for 1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O2 -o gpurun_variants/issue_rate_probe tools/probes/issue_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// each body: 64 copies of one instruction (or a small group), independent destinations round-robin over 8 registers
#define KERNEL(NAME, DECL, BODY, SINK)                                                       \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float seed) {         \
        DECL;                                                                                \
        for (int it = 0; it < iters; ++it) {                                                 \
            BODY;                                                                            \
        }                                                                                    \
        SINK;                                                                                \
    }

#define DECL_F8 float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.5f, c = 0.25f
#define SINK_F8 out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7

#define ASM8(OP)                                                  \
    asm volatile(OP " %0, %8, %9\n" OP " %1, %8, %9\n" OP " %2, %8, %9\n" OP " %3, %8, %9\n" \
                 OP " %4, %8, %9\n" OP " %5, %8, %9\n" OP " %6, %8, %9\n" OP " %7, %8, %9\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#define ASM8_3(OP)                                                  \
    asm volatile(OP " %0, %8, %9, %0\n" OP " %1, %8, %9, %1\n" OP " %2, %8, %9, %2\n" OP " %3, %8, %9, %3\n" \
                 OP " %4, %8, %9, %4\n" OP " %5, %8, %9, %5\n" OP " %6, %8, %9, %6\n" OP " %7, %8, %9, %7\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));

KERNEL(k_fmac, DECL_F8, REP8(ASM8("v_fmac_f32_e32")), SINK_F8)
KERNEL(k_mul, DECL_F8, REP8(ASM8("v_mul_f32_e32")), SINK_F8)
KERNEL(k_fma3, DECL_F8, REP8(ASM8_3("v_fma_f32")), SINK_F8)
KERNEL(k_med3, DECL_F8, REP8(ASM8_3("v_med3_f32")), SINK_F8)
KERNEL(k_and, DECL_F8, REP8(ASM8("v_and_b32_e32")), SINK_F8)
KERNEL(k_lshl, DECL_F8, REP8(ASM8("v_lshlrev_b32_e32")), SINK_F8)
KERNEL(k_cvtpk, DECL_F8, REP8(ASM8("v_cvt_pk_bf16_f32")), SINK_F8)
KERNEL(k_dot2c, DECL_F8, REP8(ASM8("v_dot2c_f32_bf16_e32")), SINK_F8)
KERNEL(k_max, DECL_F8, REP8(ASM8("v_max_f32_e32")), SINK_F8)
KERNEL(k_perm, DECL_F8, REP8(ASM8_3("v_perm_b32")), SINK_F8)
KERNEL(k_cndmask, DECL_F8, REP8(ASM8("v_cndmask_b32_e32")), SINK_F8)


KERNEL(k_sub, DECL_F8, REP8(ASM8("v_sub_f32_e32")), SINK_F8)
KERNEL(k_add, DECL_F8, REP8(ASM8("v_add_f32_e32")), SINK_F8)
KERNEL(k_addu, DECL_F8, REP8(ASM8("v_add_u32_e32")), SINK_F8)
KERNEL(k_or, DECL_F8, REP8(ASM8("v_or_b32_e32")), SINK_F8)
KERNEL(k_lshr, DECL_F8, REP8(ASM8("v_lshrrev_b32_e32")), SINK_F8)
KERNEL(k_min, DECL_F8, REP8(ASM8("v_min_f32_e32")), SINK_F8)
KERNEL(k_bfi, DECL_F8, REP8(ASM8_3("v_bfi_b32")), SINK_F8)
KERNEL(k_andor, DECL_F8, REP8(ASM8_3("v_and_or_b32")), SINK_F8)
KERNEL(k_lshlor, DECL_F8, REP8(ASM8_3("v_lshl_or_b32")), SINK_F8)
KERNEL(k_add3, DECL_F8, REP8(ASM8_3("v_add3_u32")), SINK_F8)
KERNEL(k_mullo, DECL_F8, REP8(ASM8("v_mul_lo_u32")), SINK_F8)
KERNEL(k_mad24, DECL_F8, REP8(ASM8_3("v_mad_u32_u24")), SINK_F8)
#define ASM8_1(OP)                                                  \
    asm volatile(OP " %0, %8\n" OP " %1, %8\n" OP " %2, %8\n" OP " %3, %8\n" \
                 OP " %4, %8\n" OP " %5, %8\n" OP " %6, %8\n" OP " %7, %8\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
KERNEL(k_mov, DECL_F8, REP8(ASM8_1("v_mov_b32_e32")), SINK_F8)
KERNEL(k_exp, DECL_F8, REP8(ASM8_1("v_exp_f32_e32")), SINK_F8)
KERNEL(k_rcp, DECL_F8, REP8(ASM8_1("v_rcp_f32_e32")), SINK_F8)
KERNEL(k_cvtfi, DECL_F8, REP8(ASM8_1("v_cvt_f32_i32_e32")), SINK_F8)
#define ASM8_MULABS                                                  \
    asm volatile("v_mul_f32_e64 %0, %8, |%9|\n v_mul_f32_e64 %1, %8, |%9|\n v_mul_f32_e64 %2, %8, |%9|\n v_mul_f32_e64 %3, %8, |%9|\n" \
                 "v_mul_f32_e64 %4, %8, |%9|\n v_mul_f32_e64 %5, %8, |%9|\n v_mul_f32_e64 %6, %8, |%9|\n v_mul_f32_e64 %7, %8, |%9|\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
KERNEL(k_mulabs, DECL_F8, REP8(ASM8_MULABS), SINK_F8)
#define ASM8_DPP(OP, CTRL)                                                  \
    asm volatile(OP " %0, %8, %9 " CTRL "\n" OP " %1, %8, %9 " CTRL "\n" OP " %2, %8, %9 " CTRL "\n" OP " %3, %8, %9 " CTRL "\n" \
                 OP " %4, %8, %9 " CTRL "\n" OP " %5, %8, %9 " CTRL "\n" OP " %6, %8, %9 " CTRL "\n" OP " %7, %8, %9 " CTRL "\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
KERNEL(k_add_dpp_quad, DECL_F8, REP8(ASM8_DPP("v_add_f32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")), SINK_F8)
KERNEL(k_add_dpp_shr, DECL_F8, REP8(ASM8_DPP("v_add_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")), SINK_F8)
KERNEL(k_add_dpp_mirror, DECL_F8, REP8(ASM8_DPP("v_add_f32_dpp", "row_mirror row_mask:0xf bank_mask:0xf")), SINK_F8)
KERNEL(k_fmac_sdwa, DECL_F8, REP8(ASM8_DPP("v_mul_f32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")), SINK_F8)
// v_cndmask in its forms: VCC written by a compare first; an SGPR-pair mask (e64)
#define ASM8_CND_E64                                                  \
    asm volatile("v_cndmask_b32_e64 %0, %8, %9, %10\n v_cndmask_b32_e64 %1, %8, %9, %10\n v_cndmask_b32_e64 %2, %8, %9, %10\n v_cndmask_b32_e64 %3, %8, %9, %10\n" \
                 "v_cndmask_b32_e64 %4, %8, %9, %10\n v_cndmask_b32_e64 %5, %8, %9, %10\n v_cndmask_b32_e64 %6, %8, %9, %10\n v_cndmask_b32_e64 %7, %8, %9, %10\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(msk));
#define DECL_F8M DECL_F8; unsigned long msk = __ballot(threadIdx.x & 1)
KERNEL(k_cnd_e64, DECL_F8M, REP8(ASM8_CND_E64), SINK_F8)
#define ASM8_CND_VCC                                                  \
    asm volatile("v_cmp_lt_f32_e32 vcc, %8, %9\n v_cndmask_b32_e32 %0, %8, %9, vcc\n v_cndmask_b32_e32 %1, %8, %9, vcc\n v_cndmask_b32_e32 %2, %8, %9, vcc\n v_cndmask_b32_e32 %3, %8, %9, vcc\n" \
                 "v_cndmask_b32_e32 %4, %8, %9, vcc\n v_cndmask_b32_e32 %5, %8, %9, vcc\n v_cndmask_b32_e32 %6, %8, %9, vcc\n v_cndmask_b32_e32 %7, %8, %9, vcc\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
KERNEL(k_cnd_vcc, DECL_F8, REP8(ASM8_CND_VCC), SINK_F8)
// a compare + select pair per element, the way a conditional zeroing compiles
#define ASM8_CMPCND                                                  \
    asm volatile("v_cmp_lt_f32_e32 vcc, %8, %0\n v_cndmask_b32_e32 %0, %8, %9, vcc\n v_cmp_lt_f32_e32 vcc, %8, %1\n v_cndmask_b32_e32 %1, %8, %9, vcc\n" \
                 "v_cmp_lt_f32_e32 vcc, %8, %2\n v_cndmask_b32_e32 %2, %8, %9, vcc\n v_cmp_lt_f32_e32 vcc, %8, %3\n v_cndmask_b32_e32 %3, %8, %9, vcc\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
KERNEL(k_cmpcnd, DECL_F8, REP8(ASM8_CMPCND), SINK_F8)
KERNEL(k_cmp, DECL_F8, REP8(asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cmp_lt_f32_e32 vcc, %0, %1\n v_cmp_lt_f32_e32 vcc, %0, %1\n v_cmp_lt_f32_e32 vcc, %0, %1\n v_cmp_lt_f32_e32 vcc, %0, %1\n v_cmp_lt_f32_e32 vcc, %0, %1\n v_cmp_lt_f32_e32 vcc, %0, %1\n v_cmp_lt_f32_e32 vcc, %0, %1\n" :: "v"(b), "v"(c) : "vcc");), SINK_F8)
// LDS reads: 8 x ds_read_b128 of different addresses, waited for at the end of each group
__global__ __launch_bounds__(256) void k_ldsr128(float* out, int iters, float seed) {
    __shared__ f32x4 buf[1024];
    buf[threadIdx.x] = f32x4{seed, seed, seed, seed};
    buf[threadIdx.x + 256] = buf[threadIdx.x + 512] = buf[threadIdx.x + 768] = f32x4{seed, 1, 2, 3};
    __syncthreads();
    f32x4 acc = {0, 0, 0, 0};
    const f32x4* p = buf + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            f32x4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)p), "n"((j & 15) * 1024));
            asm volatile("" : "+v"(v));
            if ((j & 7) == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc += v; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

// packed fp32: 64-bit register pairs
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define DECL_P8 f32x2 a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f, b = a0 * 0.5f, c = {0.25f, 0.5f}
#define SINK_P8 out[blockIdx.x * blockDim.x + threadIdx.x] = (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)[0] + (a0 + a1 + a2 + a3)[1]
KERNEL(k_pkfma, DECL_P8, REP8(ASM8_3("v_pk_fma_f32")), SINK_P8)
#define ASM8P(OP)                                                  \
    asm volatile(OP " %0, %8, %9\n" OP " %1, %8, %9\n" OP " %2, %8, %9\n" OP " %3, %8, %9\n" \
                 OP " %4, %8, %9\n" OP " %5, %8, %9\n" OP " %6, %8, %9\n" OP " %7, %8, %9\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
KERNEL(k_pkmul, DECL_P8, REP8(ASM8P("v_pk_mul_f32")), SINK_P8)
KERNEL(k_pkadd, DECL_P8, REP8(ASM8P("v_pk_add_f32")), SINK_P8)

// MFMA bf16 16x16x32: 8 independent accumulators, or one dependent chain
#define DECL_M f32x4 c0 = {seed, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0; bf16x8 A = {1, 2, 3, 4, 5, 6, 7, 8}, B = {8, 7, 6, 5, 4, 3, 2, 1}
#define SINK_M out[blockIdx.x * blockDim.x + threadIdx.x] = (c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7)[0]
#define MF(C) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0);
KERNEL(k_mfma_indep, DECL_M, REP8(MF(c0) MF(c1) MF(c2) MF(c3) MF(c4) MF(c5) MF(c6) MF(c7)), SINK_M)
KERNEL(k_mfma_chain, DECL_M, REP64(MF(c0)), SINK_M)
KERNEL(k_mfma_chain2, DECL_M, REP8(MF(c0) MF(c1) MF(c0) MF(c1) MF(c0) MF(c1) MF(c0) MF(c1)), SINK_M)
// 6-chains over two accumulators the way mfma_split runs them (r = 0, 1 interleaved by the compiler or not)
KERNEL(k_mfma_6x2, DECL_M, REP8(MF(c0) MF(c0) MF(c0) MF(c0) MF(c0) MF(c0) MF(c1) MF(c1)) , SINK_M)
typedef float f32x4b __attribute__((ext_vector_type(4)));
#define MF32(C) C = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 0.5f, C, 0, 0, 0);
KERNEL(k_mfma_f32_indep, DECL_M, REP8(MF32(c0) MF32(c1) MF32(c2) MF32(c3) MF32(c4) MF32(c5) MF32(c6) MF32(c7)), SINK_M)

// one MFMA followed by N independent VALU ops: does the vector ALU run beside the matrix pipe?
#define DECL_MV DECL_M; float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.5f, c = 0.25f
#define SINK_MV out[blockIdx.x * blockDim.x + threadIdx.x] = (c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7)[0] + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7
#define ASM4(OP) asm volatile(OP " %0, %4, %5\n" OP " %1, %4, %5\n" OP " %2, %4, %5\n" OP " %3, %4, %5\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
#define MFA(C) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(C) : "v"(A), "v"(B));
KERNEL(k_mfma_valu4, DECL_MV, REP8(MFA(c0) ASM4("v_fmac_f32_e32") MFA(c1) ASM4("v_fmac_f32_e32") MFA(c2) ASM4("v_fmac_f32_e32") MFA(c3) ASM4("v_fmac_f32_e32") MFA(c4) ASM4("v_fmac_f32_e32") MFA(c5) ASM4("v_fmac_f32_e32") MFA(c6) ASM4("v_fmac_f32_e32") MFA(c7) ASM4("v_fmac_f32_e32")), SINK_MV)
KERNEL(k_mfma_valu8, DECL_MV, REP8(MFA(c0) ASM8("v_fmac_f32_e32") MFA(c1) ASM8("v_fmac_f32_e32") MFA(c2) ASM8("v_fmac_f32_e32") MFA(c3) ASM8("v_fmac_f32_e32") MFA(c4) ASM8("v_fmac_f32_e32") MFA(c5) ASM8("v_fmac_f32_e32") MFA(c6) ASM8("v_fmac_f32_e32") MFA(c7) ASM8("v_fmac_f32_e32")), SINK_MV)
KERNEL(k_mfma_only_asm, DECL_MV, REP8(MFA(c0) MFA(c1) MFA(c2) MFA(c3) MFA(c4) MFA(c5) MFA(c6) MFA(c7)), SINK_MV)

struct Case { const char* name; void (*fn)(float*, int, float); int per_iter; };

int main() {
    float* out;
    hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<Case> cases = {
        {"v_fmac_f32", k_fmac, 64}, {"v_mul_f32", k_mul, 64}, {"v_fma_f32 (vop3)", k_fma3, 64}, {"v_med3_f32", k_med3, 64},
        {"v_and_b32", k_and, 64}, {"v_lshlrev_b32", k_lshl, 64}, {"v_cvt_pk_bf16_f32", k_cvtpk, 64}, {"v_dot2c_f32_bf16", k_dot2c, 64},
        {"v_max_f32", k_max, 64}, {"v_perm_b32", k_perm, 64}, {"v_cndmask_b32", k_cndmask, 64},
        {"v_sub_f32", k_sub, 64}, {"v_add_f32", k_add, 64}, {"v_add_u32", k_addu, 64}, {"v_or_b32", k_or, 64}, {"v_lshrrev_b32", k_lshr, 64},
        {"v_min_f32", k_min, 64}, {"v_bfi_b32", k_bfi, 64}, {"v_and_or_b32", k_andor, 64}, {"v_lshl_or_b32", k_lshlor, 64}, {"v_add3_u32", k_add3, 64},
        {"v_mul_lo_u32", k_mullo, 64}, {"v_mad_u32_u24", k_mad24, 64}, {"v_mov_b32", k_mov, 64}, {"v_exp_f32", k_exp, 64}, {"v_rcp_f32", k_rcp, 64},
        {"v_cvt_f32_i32", k_cvtfi, 64}, {"v_mul_f32_e64 with |.|", k_mulabs, 64},
        {"v_add_f32_dpp quad_perm", k_add_dpp_quad, 64}, {"v_add_f32_dpp row_shr:1", k_add_dpp_shr, 64}, {"v_add_f32_dpp row_mirror", k_add_dpp_mirror, 64},
        {"v_mul_f32_sdwa src0 WORD_1", k_fmac_sdwa, 64},
        {"v_cndmask_b32_e64 (sgpr mask)", k_cnd_e64, 64}, {"v_cndmask_b32_e32 (vcc by v_cmp) [64 of 72]", k_cnd_vcc, 64},
        {"v_cmp + v_cndmask pairs [per pair]", k_cmpcnd, 32}, {"v_cmp_lt_f32 -> vcc", k_cmp, 64},
        {"ds_read_b128 (8 in flight)", k_ldsr128, 64},
        {"v_pk_fma_f32", k_pkfma, 64}, {"v_pk_mul_f32", k_pkmul, 64}, {"v_pk_add_f32", k_pkadd, 64},
        {"mfma bf16 16x16x32, 8 independent", k_mfma_indep, 64}, {"mfma bf16, one chain", k_mfma_chain, 64},
        {"mfma bf16, two chains alternating", k_mfma_chain2, 64}, {"mfma bf16, 6 + 2 pattern", k_mfma_6x2, 64},
        {"mfma f32 16x16x4, 8 independent", k_mfma_f32_indep, 64},
        {"mfma bf16 (asm) alone [per mfma]", k_mfma_only_asm, 64},
        {"mfma bf16 + 4 v_fmac [per mfma]", k_mfma_valu4, 64}, {"mfma bf16 + 8 v_fmac [per mfma]", k_mfma_valu8, 64},
    };
    const int iters = 4000;
    int dev_clock_khz = 0;
    hipDeviceGetAttribute(&dev_clock_khz, hipDeviceAttributeClockRate, 0);
    printf("device clock attribute: %d kHz\n", dev_clock_khz);
    printf("%-40s %12s %12s %12s\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    for (auto& cs : cases) {
        printf("%-40s", cs.name);
        for (int wg_per_cu : {1, 2, 4}) {
            const int grid = 256 * wg_per_cu;          // 256 CUs, 256 threads = 1 wave per SIMD per workgroup
            cs.fn<<<grid, 256>>>(out, 10, 1.0f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            cs.fn<<<grid, 256>>>(out, iters, 1.0f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            // ns per instruction and SIMD (all resident waves of a SIMD together issue wg_per_cu * per_iter * iters instructions)
            const double ns = (double)ms * 1e6 / ((double)iters * cs.per_iter * wg_per_cu);
            printf(" %9.3f ns", ns);
        }
        printf("\n");
    }
    printf("(ns per instruction per SIMD; at 2.4 GHz a 4-cycle issue is 1.667 ns, a 16-cycle MFMA 6.67 ns)\n");
    return 0;
}
