// Issue cost of single vector instructions on one SIMD: every wave runs ITER x 32 independent
// copies of one instruction; 1 or 4 waves per SIMD.  Prints SIMD cycles per wave64 instruction.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))

#define KERNEL(name, ASM)                                                                          \
__global__ __launch_bounds__(1024) void name(int iters, uint64_t* out, float* sink)               \
{                                                                                                  \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.5f, c = 0.25f;             \
    float p0 = a0, p1 = a1, p2 = a2, p3 = a3;                                                      \
    uint32_t u = threadIdx.x * 2654435761u, s = (threadIdx.x & 1) << 4;                            \
    __syncthreads();                                                                               \
    uint64_t t0 = __builtin_readcyclecounter();                                                    \
    for (int i = 0; i < iters; i++) { REP32(ASM) }                                                  \
    uint64_t t1 = __builtin_readcyclecounter();                                                    \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                               \
    if (a0 + a1 + a2 + a3 + p0 + p1 + p2 + p3 == 12345.0f) sink[0] = a0 + u;                       \
}

KERNEL(k_fma,      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));)
KERNEL(k_pkfma,    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&p0) : "v"(*(double*)&p2));)
KERNEL(k_max,      asm volatile("v_max_f32 %0, %0, %1" : "+v"(a0) : "v"(b));)
KERNEL(k_max3,     asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));)
KERNEL(k_alignbit, asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(u) : "v"(s));)
KERNEL(k_perm,     asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(u) : "v"(s));)
KERNEL(k_cvtsdwa,  asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a0) : "v"(u));)
KERNEL(k_cvt,      asm volatile("v_cvt_f32_u32_e32 %0, %1" : "=v"(a0) : "v"(u));)
KERNEL(k_cndmask,  asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b) : );)
KERNEL(k_cnd_nodep, asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a0) : "v"(b), "v"(c));)
KERNEL(k_cnd_e64,  asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(a0) : "v"(b), "v"(c));)
KERNEL(k_cmpcnd,   asm volatile("v_cmp_le_f32_e32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a0) : "v"(b), "v"(c) : "vcc");)
KERNEL(k_cmpcnd64, asm volatile("v_cmp_le_f32_e64 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(a0) : "v"(b), "v"(c) : "s20", "s21");)
KERNEL(k_cmp2cnd2, asm volatile("v_cmp_le_f32_e32 vcc, %1, %2\n\tv_cmp_le_f32_e64 s[20:21], %2, %1\n\tv_cndmask_b32_e32 %0, %1, %2, vcc\n\tv_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(a0) : "v"(b), "v"(c) : "vcc", "s20", "s21");)
KERNEL(k_addc,     asm volatile("v_addc_co_u32_e64 %0, s[20:21], 0, %0, s[20:21]" : "+v"(u) : : "s20", "s21");)
KERNEL(k_min_u32,  asm volatile("v_min_u32_e32 %0, %0, %1" : "+v"(u) : "v"(s));)
KERNEL(k_med3,     asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));)
KERNEL(k_bfi,      asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(u) : "v"(s));)
KERNEL(k_lshladd,  asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u) : "v"(s));)
KERNEL(k_bfe,      asm volatile("v_bfe_u32 %0, %0, 3, 16" : "+v"(u));)
KERNEL(k_fmamix,   asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a0) : "v"(u), "v"(b));)
KERNEL(k_rcp,      asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a0));)
KERNEL(k_cmp,      asm volatile("v_cmp_le_f32_e32 vcc, %0, %1" : : "v"(a0), "v"(b) : "vcc");)
KERNEL(k_mul,      asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(a0) : "v"(b));)
KERNEL(k_mov,      asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a0) : "v"(b));)

template <typename K> void run(const char* name, K k)
{
    uint64_t* d; float* sink; hipMalloc(&d, 8 * 16); hipMalloc(&sink, 4);
    const int iters = 2000;
    for (int threads : { 256, 1024 }) {
        uint64_t h[1];
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, iters, d, sink);
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, iters, d, sink);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, iters * 10, d, sink); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double nsPerInstr = ms * 1e6 / (iters * 10 * 32.0 * (threads / 256.0));
        double wavesPerSimd = threads / 256.0;
        // s_memtime ticks at 100 MHz; convert with the measured ratio printed by the mov row
        printf("%-10s waves/SIMD %.0f: %8.3f counter ticks per wave instruction\n", name, wavesPerSimd, (double)h[0] / (iters * 32.0 * wavesPerSimd));
        printf("%-10s waves/SIMD %.0f: %8.3f ns of SIMD time per wave instruction (256 workgroups, events) = %.2f cycles at 2.4 GHz\n", name, wavesPerSimd, nsPerInstr, nsPerInstr * 2.4);
    }
    hipFree(d); hipFree(sink);
}

int main()
{
    run("mov", k_mov); run("fma", k_fma); run("mul", k_mul); run("pk_fma", k_pkfma); run("max", k_max); run("max3", k_max3);
    run("alignbit", k_alignbit); run("perm", k_perm); run("cvt_sdwa", k_cvtsdwa); run("cvt", k_cvt); run("cndmask", k_cndmask);
    run("cnd_nodep", k_cnd_nodep); run("cnd_e64", k_cnd_e64); run("cmp+cnd", k_cmpcnd); run("cmp+cnd64", k_cmpcnd64); run("2cmp+2cnd", k_cmp2cnd2); run("addc", k_addc); run("min_u32", k_min_u32); run("med3", k_med3); run("bfi", k_bfi); run("lshl_add", k_lshladd); run("bfe", k_bfe); run("fma_mix", k_fmamix); run("rcp", k_rcp); run("cmp", k_cmp);
    return 0;
}
