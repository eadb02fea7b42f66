// Throughput (not latency) of single vector instructions: every wave runs eight independent chains of one
// instruction, 8 waves per SIMD (two 1024-thread workgroups per compute unit), so the SIMD's vector unit is
// the only limit.  Prints nanoseconds of SIMD time per wave64 instruction and the ratio to v_fma_f32.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_throughput valu_throughput.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define KERNEL(...) KERNEL_(__VA_ARGS__)
#define KERNEL_(name, A0, A1, A2, A3, A4, A5, A6, A7)                                                \
__global__ __launch_bounds__(1024) void name(int iters, float* sink)                               \
{                                                                                                  \
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7; \
    float b = 1.0000001f, c = 0.25f;                                                               \
    uint32_t u0 = threadIdx.x * 2654435761u, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7; \
    uint32_t s = (threadIdx.x & 1) << 4;                                                           \
    for (int i = 0; i < iters; i++) {                                                               \
        asm volatile(A0 "\n\t" A1 "\n\t" A2 "\n\t" A3 "\n\t" A4 "\n\t" A5 "\n\t" A6 "\n\t" A7 "\n\t" \
                     A0 "\n\t" A1 "\n\t" A2 "\n\t" A3 "\n\t" A4 "\n\t" A5 "\n\t" A6 "\n\t" A7 "\n\t" \
                     A0 "\n\t" A1 "\n\t" A2 "\n\t" A3 "\n\t" A4 "\n\t" A5 "\n\t" A6 "\n\t" A7 "\n\t" \
                     A0 "\n\t" A1 "\n\t" A2 "\n\t" A3 "\n\t" A4 "\n\t" A5 "\n\t" A6 "\n\t" A7          \
                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), \
                       "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) \
                     : "v"(b), "v"(c), "v"(s) : "vcc", "s20", "s21");                                \
    }                                                                                              \
    if (f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 == 12345.0f) sink[0] = u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7; \
}
// operand numbering: %0..%7 = f0..f7, %8..%15 = u0..u7, %16 = b, %17 = c, %18 = s
#define F8(op, rest) op " %0, %0, " rest, op " %1, %1, " rest, op " %2, %2, " rest, op " %3, %3, " rest, op " %4, %4, " rest, op " %5, %5, " rest, op " %6, %6, " rest, op " %7, %7, " rest
#define U8(op, rest) op " %8, %8, " rest, op " %9, %9, " rest, op " %10, %10, " rest, op " %11, %11, " rest, op " %12, %12, " rest, op " %13, %13, " rest, op " %14, %14, " rest, op " %15, %15, " rest

KERNEL(k_fma, F8("v_fma_f32", "%16, %17"))
KERNEL(k_mul, F8("v_mul_f32_e32", "%16"))
KERNEL(k_add, F8("v_add_f32_e32", "%16"))
KERNEL(k_sub, F8("v_sub_f32_e32", "%16"))
KERNEL(k_max, F8("v_max_f32_e32", "%16"))
KERNEL(k_max3, F8("v_max3_f32", "%16, %17"))
KERNEL(k_min3, F8("v_min3_f32", "%16, %17"))
KERNEL(k_med3, F8("v_med3_f32", "%16, %17"))
KERNEL(k_fmac, F8("v_fmac_f32_e32", "%16"))
KERNEL(k_alignbit, U8("v_alignbit_b32", "%8, %18"))
KERNEL(k_perm, U8("v_perm_b32", "%8, %18"))
KERNEL(k_and_or, U8("v_and_or_b32", "%18, %9"))
KERNEL(k_and, U8("v_and_b32_e32", "%18"))
KERNEL(k_or, U8("v_or_b32_e32", "%18"))
KERNEL(k_minu, U8("v_min_u32_e32", "%18"))
KERNEL(k_maxu, U8("v_max_u32_e32", "%18"))
KERNEL(k_addu, U8("v_add_u32_e32", "%18"))
KERNEL(k_add3, U8("v_add3_u32", "%18, %9"))
KERNEL(k_lshladd, U8("v_lshl_add_u32", "1, %18"))
KERNEL(k_lshl, U8("v_lshlrev_b32_e32", "%18"))   /* v_lshlrev_b32 dst, dst(shift), s(value) */
KERNEL(k_bfe, U8("v_bfe_u32", "3, 16"))
KERNEL(k_madu24, U8("v_mad_u32_u24", "%18, %9"))
KERNEL(k_cnd32, "v_cndmask_b32_e32 %0, %0, %16, vcc", "v_cndmask_b32_e32 %1, %1, %16, vcc", "v_cndmask_b32_e32 %2, %2, %16, vcc", "v_cndmask_b32_e32 %3, %3, %16, vcc", "v_cndmask_b32_e32 %4, %4, %16, vcc", "v_cndmask_b32_e32 %5, %5, %16, vcc", "v_cndmask_b32_e32 %6, %6, %16, vcc", "v_cndmask_b32_e32 %7, %7, %16, vcc")
KERNEL(k_cnd64, "v_cndmask_b32_e64 %0, %0, %16, s[20:21]", "v_cndmask_b32_e64 %1, %1, %16, s[20:21]", "v_cndmask_b32_e64 %2, %2, %16, s[20:21]", "v_cndmask_b32_e64 %3, %3, %16, s[20:21]", "v_cndmask_b32_e64 %4, %4, %16, s[20:21]", "v_cndmask_b32_e64 %5, %5, %16, s[20:21]", "v_cndmask_b32_e64 %6, %6, %16, s[20:21]", "v_cndmask_b32_e64 %7, %7, %16, s[20:21]")
KERNEL(k_cmp32, "v_cmp_le_f32_e32 vcc, %0, %16", "v_cmp_le_f32_e32 vcc, %1, %16", "v_cmp_le_f32_e32 vcc, %2, %16", "v_cmp_le_f32_e32 vcc, %3, %16", "v_cmp_le_f32_e32 vcc, %4, %16", "v_cmp_le_f32_e32 vcc, %5, %16", "v_cmp_le_f32_e32 vcc, %6, %16", "v_cmp_le_f32_e32 vcc, %7, %16")
KERNEL(k_cmp64, "v_cmp_le_f32_e64 s[20:21], %0, %16", "v_cmp_le_f32_e64 s[20:21], %1, %16", "v_cmp_le_f32_e64 s[20:21], %2, %16", "v_cmp_le_f32_e64 s[20:21], %3, %16", "v_cmp_le_f32_e64 s[20:21], %4, %16", "v_cmp_le_f32_e64 s[20:21], %5, %16", "v_cmp_le_f32_e64 s[20:21], %6, %16", "v_cmp_le_f32_e64 s[20:21], %7, %16")
KERNEL(k_cmpu32, "v_cmp_le_u32_e32 vcc, %8, %18", "v_cmp_le_u32_e32 vcc, %9, %18", "v_cmp_le_u32_e32 vcc, %10, %18", "v_cmp_le_u32_e32 vcc, %11, %18", "v_cmp_le_u32_e32 vcc, %12, %18", "v_cmp_le_u32_e32 vcc, %13, %18", "v_cmp_le_u32_e32 vcc, %14, %18", "v_cmp_le_u32_e32 vcc, %15, %18")
KERNEL(k_cvt, "v_cvt_f32_u32_e32 %0, %8", "v_cvt_f32_u32_e32 %1, %9", "v_cvt_f32_u32_e32 %2, %10", "v_cvt_f32_u32_e32 %3, %11", "v_cvt_f32_u32_e32 %4, %12", "v_cvt_f32_u32_e32 %5, %13", "v_cvt_f32_u32_e32 %6, %14", "v_cvt_f32_u32_e32 %7, %15")
#define SDWA " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
KERNEL(k_cvtsdwa, "v_cvt_f32_u32_sdwa %0, %8" SDWA, "v_cvt_f32_u32_sdwa %1, %9" SDWA, "v_cvt_f32_u32_sdwa %2, %10" SDWA, "v_cvt_f32_u32_sdwa %3, %11" SDWA, "v_cvt_f32_u32_sdwa %4, %12" SDWA, "v_cvt_f32_u32_sdwa %5, %13" SDWA, "v_cvt_f32_u32_sdwa %6, %14" SDWA, "v_cvt_f32_u32_sdwa %7, %15" SDWA)
KERNEL(k_cvtub, "v_cvt_f32_ubyte1_e32 %0, %8", "v_cvt_f32_ubyte1_e32 %1, %9", "v_cvt_f32_ubyte1_e32 %2, %10", "v_cvt_f32_ubyte1_e32 %3, %11", "v_cvt_f32_ubyte1_e32 %4, %12", "v_cvt_f32_ubyte1_e32 %5, %13", "v_cvt_f32_ubyte1_e32 %6, %14", "v_cvt_f32_ubyte1_e32 %7, %15")
KERNEL(k_fmamix, "v_fma_mix_f32 %0, %8, %16, %0 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %1, %9, %16, %1 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %2, %10, %16, %2 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %3, %11, %16, %3 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %4, %12, %16, %4 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %5, %13, %16, %5 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %6, %14, %16, %6 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %7, %15, %16, %7 op_sel_hi:[1,0,0]")
KERNEL(k_mov, "v_mov_b32_e32 %0, %16", "v_mov_b32_e32 %1, %16", "v_mov_b32_e32 %2, %16", "v_mov_b32_e32 %3, %16", "v_mov_b32_e32 %4, %16", "v_mov_b32_e32 %5, %16", "v_mov_b32_e32 %6, %16", "v_mov_b32_e32 %7, %16")
KERNEL(k_rcp, "v_rcp_f32_e32 %0, %0", "v_rcp_f32_e32 %1, %1", "v_rcp_f32_e32 %2, %2", "v_rcp_f32_e32 %3, %3", "v_rcp_f32_e32 %4, %4", "v_rcp_f32_e32 %5, %5", "v_rcp_f32_e32 %6, %6", "v_rcp_f32_e32 %7, %7")
KERNEL(k_addc, "v_addc_co_u32_e64 %8, s[20:21], 0, %8, s[20:21]", "v_addc_co_u32_e64 %9, s[20:21], 0, %9, s[20:21]", "v_addc_co_u32_e64 %10, s[20:21], 0, %10, s[20:21]", "v_addc_co_u32_e64 %11, s[20:21], 0, %11, s[20:21]", "v_addc_co_u32_e64 %12, s[20:21], 0, %12, s[20:21]", "v_addc_co_u32_e64 %13, s[20:21], 0, %13, s[20:21]", "v_addc_co_u32_e64 %14, s[20:21], 0, %14, s[20:21]", "v_addc_co_u32_e64 %15, s[20:21], 0, %15, s[20:21]")


KERNEL(k_xor, U8("v_xor_b32_e32", "%18"))
KERNEL(k_subu, U8("v_sub_u32_e32", "%18"))
KERNEL(k_addco, "v_add_co_u32_e32 %8, vcc, %8, %18", "v_add_co_u32_e32 %9, vcc, %9, %18", "v_add_co_u32_e32 %10, vcc, %10, %18", "v_add_co_u32_e32 %11, vcc, %11, %18", "v_add_co_u32_e32 %12, vcc, %12, %18", "v_add_co_u32_e32 %13, vcc, %13, %18", "v_add_co_u32_e32 %14, vcc, %14, %18", "v_add_co_u32_e32 %15, vcc, %15, %18")
KERNEL(k_swap, "v_swap_b32 %8, %9", "v_swap_b32 %10, %11", "v_swap_b32 %12, %13", "v_swap_b32 %14, %15", "v_swap_b32 %0, %1", "v_swap_b32 %2, %3", "v_swap_b32 %4, %5", "v_swap_b32 %6, %7")
KERNEL(k_bfi, U8("v_bfi_b32", "%18, %9"))
KERNEL(k_lshlor, U8("v_lshl_or_b32", "3, %18"))
KERNEL(k_mulu24, U8("v_mul_u32_u24_e32", "%18"))
KERNEL(k_pkminu16, U8("v_pk_min_u16", "%18"))
KERNEL(k_pkaddu16, U8("v_pk_add_u16", "%18"))
KERNEL(k_bcnt, U8("v_bcnt_u32_b32", "%18"))
KERNEL(k_not, "v_not_b32_e32 %8, %8", "v_not_b32_e32 %9, %9", "v_not_b32_e32 %10, %10", "v_not_b32_e32 %11, %11", "v_not_b32_e32 %12, %12", "v_not_b32_e32 %13, %13", "v_not_b32_e32 %14, %14", "v_not_b32_e32 %15, %15")
KERNEL(k_cvtf16, "v_cvt_f32_f16_e32 %0, %8", "v_cvt_f32_f16_e32 %1, %9", "v_cvt_f32_f16_e32 %2, %10", "v_cvt_f32_f16_e32 %3, %11", "v_cvt_f32_f16_e32 %4, %12", "v_cvt_f32_f16_e32 %5, %13", "v_cvt_f32_f16_e32 %6, %14", "v_cvt_f32_f16_e32 %7, %15")
KERNEL(k_movdpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_minf, F8("v_min_f32_e32", "%16"))
KERNEL(k_cmpcnd, "v_cmp_le_f32_e32 vcc, %0, %16\n\tv_cndmask_b32_e32 %0, %0, %17, vcc", "v_cmp_le_f32_e32 vcc, %1, %16\n\tv_cndmask_b32_e32 %1, %1, %17, vcc", "v_cmp_le_f32_e32 vcc, %2, %16\n\tv_cndmask_b32_e32 %2, %2, %17, vcc", "v_cmp_le_f32_e32 vcc, %3, %16\n\tv_cndmask_b32_e32 %3, %3, %17, vcc", "v_cmp_le_f32_e32 vcc, %4, %16\n\tv_cndmask_b32_e32 %4, %4, %17, vcc", "v_cmp_le_f32_e32 vcc, %5, %16\n\tv_cndmask_b32_e32 %5, %5, %17, vcc", "v_cmp_le_f32_e32 vcc, %6, %16\n\tv_cndmask_b32_e32 %6, %6, %17, vcc", "v_cmp_le_f32_e32 vcc, %7, %16\n\tv_cndmask_b32_e32 %7, %7, %17, vcc")
KERNEL(k_cmpcnd64, "v_cmp_le_f32_e64 s[20:21], %0, %16\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %17, s[20:21]", "v_cmp_le_f32_e64 s[20:21], %1, %16\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, %1, %17, s[20:21]", "v_cmp_le_f32_e64 s[20:21], %2, %16\n\ts_nop 1\n\tv_cndmask_b32_e64 %2, %2, %17, s[20:21]", "v_cmp_le_f32_e64 s[20:21], %3, %16\n\ts_nop 1\n\tv_cndmask_b32_e64 %3, %3, %17, s[20:21]", "v_cmp_le_f32_e64 s[20:21], %4, %16\n\ts_nop 1\n\tv_cndmask_b32_e64 %4, %4, %17, s[20:21]", "v_cmp_le_f32_e64 s[20:21], %5, %16\n\ts_nop 1\n\tv_cndmask_b32_e64 %5, %5, %17, s[20:21]", "v_cmp_le_f32_e64 s[20:21], %6, %16\n\ts_nop 1\n\tv_cndmask_b32_e64 %6, %6, %17, s[20:21]", "v_cmp_le_f32_e64 s[20:21], %7, %16\n\ts_nop 1\n\tv_cndmask_b32_e64 %7, %7, %17, s[20:21]")

static double g_fma = 0;
template <typename K> void run(const char* name, K k, int perIter = 32)
{
    float* sink; (void)hipMalloc(&sink, 4);
    const int iters = 20000;
    hipLaunchKernelGGL(k, dim3(512), dim3(1024), 0, 0, 100, sink);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(512), dim3(1024), 0, 0, iters, sink); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // 512 workgroups x 16 waves over 1024 SIMDs = 8 waves per SIMD
    double ns = ms * 1e6 / ((double)iters * perIter * 8.0);
    if (g_fma == 0) g_fma = ns;
    printf("%-10s %7.3f ns of SIMD time per wave64 instruction, %.2f x v_fma_f32 (%.2f cycles if v_fma_f32 takes 2)\n", name, ns, ns / g_fma, 2 * ns / g_fma);
    (void)hipFree(sink);
}

int main()
{
    run("fma", k_fma); run("mul", k_mul); run("add", k_add); run("sub", k_sub); run("fmac", k_fmac); run("mov", k_mov); run("max", k_max); run("max3", k_max3); run("min3", k_min3); run("med3", k_med3);
    run("alignbit", k_alignbit); run("perm", k_perm); run("and_or", k_and_or); run("and", k_and); run("or", k_or); run("min_u32", k_minu); run("max_u32", k_maxu);
    run("add_u32", k_addu); run("add3_u32", k_add3); run("lshl_add", k_lshladd); run("lshlrev", k_lshl); run("bfe", k_bfe); run("mad_u24", k_madu24);
    run("cnd_e32", k_cnd32); run("cnd_e64", k_cnd64); run("cmp_e32", k_cmp32); run("cmp_e64", k_cmp64); run("cmpu_e32", k_cmpu32); run("addc", k_addc);
    run("cvt", k_cvt); run("cvt_sdwa", k_cvtsdwa); run("cvt_ubyte", k_cvtub); run("fma_mix", k_fmamix); run("rcp", k_rcp);
    run("xor", k_xor); run("sub_u32", k_subu); run("add_co", k_addco); run("swap", k_swap); run("bfi", k_bfi); run("lshl_or", k_lshlor); run("mul_u24", k_mulu24);
    run("pk_min_u16", k_pkminu16); run("pk_add_u16", k_pkaddu16); run("bcnt", k_bcnt); run("not", k_not); run("cvt_f16", k_cvtf16); run("mov_dpp", k_movdpp); run("min_f32", k_minf);
    run("cmp+cnd32", k_cmpcnd, 64); run("cmp+cnd64", k_cmpcnd64, 64);
    return 0;
}
