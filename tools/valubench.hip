// valubench.hip -- per-instruction VALU issue rate on gfx950 (diagnostic tool, not product).
// Each kernel runs ITER x 64 independent copies of one instruction on 8 register sets per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 2000
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DEFK(NAME, ASM)                                                                         \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned seed)                   \
    {                                                                                           \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,  \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = a0 ^ 0x5a5a, c = a0 + 77;        \
        for (int i = 0; i < ITER; ++i) {                                                        \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                     \
                asm volatile(ASM(0) "\n\t" ASM(1) "\n\t" ASM(2) "\n\t" ASM(3) "\n\t" ASM(4) "\n\t" ASM(5) \
                             "\n\t" ASM(6) "\n\t" ASM(7)                                        \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                             : "v"(b), "v"(c), "s"(seed));                                      \
            }                                                                                   \
        }                                                                                       \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;            \
    }
#define A_PKADD(n) "v_pk_add_u16 %" #n ", %" #n ", %8"
#define A_PKSUB(n) "v_pk_sub_i16 %" #n ", %" #n ", %8"
#define A_PKASHR(n) "v_pk_ashrrev_i16 %" #n ", 1, %" #n " op_sel_hi:[0,1]"
#define A_DOT2(n) "v_dot2_i32_i16 %" #n ", %" #n ", %8, 0"
#define A_DOT2C(n) "v_dot2c_i32_i16 %" #n ", %8, %9"
#define A_BFE(n) "v_bfe_i32 %" #n ", %" #n ", 15, 16"
#define A_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %10"
#define A_ADD(n) "v_add_u32 %" #n ", %" #n ", %8"
#define A_LSHL(n) "v_lshlrev_b32 %" #n ", 1, %" #n
#define A_AND(n) "v_and_b32 %" #n ", %" #n ", %8"
#define A_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9"
#define A_MAD24(n) "v_mad_i32_i24 %" #n ", %" #n ", %8, %9"
#define A_ALIGN(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 16"
#define A_MOV(n) "v_mov_b32 %" #n ", %8"
#define A_PKMAD(n) "v_pk_mad_i16 %" #n ", %" #n ", %8, %9"
#define A_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 16, %8"
#define A_PKLSHL(n) "v_pk_lshlrev_b16 %" #n ", 1, %" #n " op_sel_hi:[0,1]"
#define A_ADD16(n) "v_add_u16 %" #n ", %" #n ", %8"
#define A_PKFMA32(n) "v_add_u32 %" #n ", %" #n ", %8"
DEFK(k_pkadd, A_PKADD) DEFK(k_pksub, A_PKSUB) DEFK(k_pkashr, A_PKASHR) DEFK(k_dot2, A_DOT2) DEFK(k_dot2c, A_DOT2C)
DEFK(k_bfe, A_BFE) DEFK(k_perm, A_PERM) DEFK(k_add, A_ADD) DEFK(k_lshl, A_LSHL) DEFK(k_and, A_AND) DEFK(k_fma, A_FMA)
DEFK(k_mad24, A_MAD24) DEFK(k_align, A_ALIGN) DEFK(k_mov, A_MOV) DEFK(k_pkmad, A_PKMAD) DEFK(k_lshladd, A_LSHLADD)
DEFK(k_pklshl, A_PKLSHL) DEFK(k_add16, A_ADD16)
#define A_MIX1(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n\tv_and_b32 %9, %9, %8"
#define A_MIX2(n) "v_dot2_i32_i16 %" #n ", %" #n ", %8, 0\n\tv_add_u32 %9, %9, %8"
#define A_MIX3(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n\tv_dot2_i32_i16 %9, %9, %8, 0"
#define A_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8"
#define A_SUB(n) "v_sub_u32 %" #n ", %" #n ", %8"
#define A_ASHR32(n) "v_ashrrev_i32 %" #n ", 1, %" #n
#define A_LSHR32(n) "v_lshrrev_b32 %" #n ", 1, %" #n
#define A_ASHR16(n) "v_ashrrev_i16 %" #n ", 1, %" #n
#define A_MULLO16(n) "v_mul_lo_u16 %" #n ", %" #n ", %8"
#define A_BFI(n) "v_bfi_b32 %" #n ", %8, %" #n ", %9"
#define A_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9"
#define A_MAX(n) "v_max_i32 %" #n ", %" #n ", %8"
#define A_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc"
#define A_PKFMAF16(n) "v_pk_fma_f16 %" #n ", %" #n ", %8, %9"
#define A_PKADDF16(n) "v_pk_add_f16 %" #n ", %" #n ", %8"
#define A_PKMAX(n) "v_pk_max_i16 %" #n ", %" #n ", %8"
#define A_MADU16(n) "v_mad_u16 %" #n ", %" #n ", %8, %9"
#define A_SDWA(n) "v_add_u16_sdwa %" #n ", %" #n ", %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1"
#define A_MULI24(n) "v_mul_i32_i24 %" #n ", %" #n ", %8"
#define A_PKFMA32(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9"
DEFK(k_mix1, A_MIX1) DEFK(k_mix2, A_MIX2) DEFK(k_mix3, A_MIX3) DEFK(k_xor, A_XOR) DEFK(k_sub, A_SUB) DEFK(k_ashr32, A_ASHR32)
DEFK(k_lshr32, A_LSHR32) DEFK(k_ashr16, A_ASHR16) DEFK(k_mullo16, A_MULLO16) DEFK(k_bfi, A_BFI) DEFK(k_andor, A_ANDOR)
DEFK(k_max, A_MAX) DEFK(k_cnd, A_CNDMASK) DEFK(k_pkfmaf16, A_PKFMAF16) DEFK(k_pkaddf16, A_PKADDF16) DEFK(k_pkmax, A_PKMAX)
DEFK(k_madu16, A_MADU16) DEFK(k_sdwa, A_SDWA) DEFK(k_muli24, A_MULI24)
#define A_MULLO32(n) "v_mul_lo_u32 %" #n ", %" #n ", %8"
#define A_MULHI32(n) "v_mul_hi_i32 %" #n ", %" #n ", %8"
#define A_MULHI24(n) "v_mul_hi_i32_i24 %" #n ", %" #n ", %8"
#define A_MADI16(n) "v_mad_i32_i16 %" #n ", %" #n ", %8, %9"
#define A_DOT4(n) "v_dot4_i32_i8 %" #n ", %" #n ", %8, %9"
#define A_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9"
#define A_ASHR64(n) "v_add_lshl_u32 %" #n ", %" #n ", %8, 1"
DEFK(k_mullo32, A_MULLO32) DEFK(k_mulhi32, A_MULHI32) DEFK(k_mulhi24, A_MULHI24) DEFK(k_madi16, A_MADI16) DEFK(k_dot4, A_DOT4)
DEFK(k_add3, A_ADD3) DEFK(k_addlshl, A_ASHR64)
__global__ __launch_bounds__(256) void k_mad64(unsigned* out, unsigned seed)
{
    long long a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    int b = (int)a0 ^ 0x5a5a, c = (int)a0 + 77;
    for (int i = 0; i < ITER; ++i) {
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {
            asm volatile("v_mad_i64_i32 %0, vcc, %8, %9, %0\n\tv_mad_i64_i32 %1, vcc, %8, %9, %1\n\tv_mad_i64_i32 %2, vcc, %8, %9, %2\n\t"
                         "v_mad_i64_i32 %3, vcc, %8, %9, %3\n\tv_mad_i64_i32 %4, vcc, %8, %9, %4\n\tv_mad_i64_i32 %5, vcc, %8, %9, %5\n\t"
                         "v_mad_i64_i32 %6, vcc, %8, %9, %6\n\tv_mad_i64_i32 %7, vcc, %8, %9, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}

template <typename K> void run(const char* name, K k, unsigned* out, int blocks)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1u);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 2u);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double insts = (double)blocks * 4 * ITER * 64;            // wave-instructions
    double per_simd_per_clk = insts / (256.0 * 4) / (ms * 1e-3 * 2.4e9);
    printf("%-26s %.3f ms  %.2f wave-instr/clk/SIMD @2.4GHz  (%.2f clk per instr)\n", name, ms, per_simd_per_clk, 1.0 / per_simd_per_clk);
}
int main(int argc, char** argv)
{
    int wps = argc > 1 ? atoi(argv[1]) : 4; // waves per SIMD
    int blocks = 256 * wps;
    unsigned* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    for (int r = 0; r < 2; ++r) {
    run("pk_add", k_pkadd, out, blocks); run("pk_sub", k_pksub, out, blocks); run("pk_ashr", k_pkashr, out, blocks);
    run("dot2", k_dot2, out, blocks); run("dot2c", k_dot2c, out, blocks); run("bfe", k_bfe, out, blocks);
    run("perm", k_perm, out, blocks); run("add_u32", k_add, out, blocks); run("lshl", k_lshl, out, blocks);
    run("and", k_and, out, blocks); run("fma_f32", k_fma, out, blocks); run("mad_i24", k_mad24, out, blocks);
    run("alignbit", k_align, out, blocks); run("mov", k_mov, out, blocks); run("pk_mad16", k_pkmad, out, blocks);
    run("lshl_add", k_lshladd, out, blocks); run("pk_lshl16", k_pklshl, out, blocks); run("add_u16", k_add16, out, blocks);
    run("MIX pkadd+and (2 instr)", k_mix1, out, blocks); run("MIX dot2+add (2 instr)", k_mix2, out, blocks); run("MIX pkadd+dot2 (2)", k_mix3, out, blocks);
    run("xor", k_xor, out, blocks); run("sub_u32", k_sub, out, blocks); run("ashr_i32", k_ashr32, out, blocks); run("lshr_b32", k_lshr32, out, blocks);
    run("ashr_i16", k_ashr16, out, blocks); run("mul_lo_u16", k_mullo16, out, blocks); run("bfi", k_bfi, out, blocks); run("and_or", k_andor, out, blocks);
    run("max_i32", k_max, out, blocks); run("cndmask", k_cnd, out, blocks); run("pk_fma_f16", k_pkfmaf16, out, blocks); run("pk_add_f16", k_pkaddf16, out, blocks);
    run("pk_max_i16", k_pkmax, out, blocks); run("mad_u16", k_madu16, out, blocks); run("add_u16_sdwa", k_sdwa, out, blocks); run("mul_i32_i24", k_muli24, out, blocks);
    run("mul_lo_u32", k_mullo32, out, blocks); run("mul_hi_i32", k_mulhi32, out, blocks); run("mul_hi_i32_i24", k_mulhi24, out, blocks); run("mad_i32_i16", k_madi16, out, blocks);
    run("dot4_i32_i8", k_dot4, out, blocks); run("add3_u32", k_add3, out, blocks); run("add_lshl_u32", k_addlshl, out, blocks); run("mad_i64_i32", k_mad64, out, blocks);
    }
    return 0;
}
