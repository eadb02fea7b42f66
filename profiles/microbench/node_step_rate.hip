// Throughput of the arithmetic of one node step (four box tests on 16-bit grid words, entries, sort,
// choice of the next node) with no memory access at all: WAVES waves per SIMD run ITERS steps each on
// register data.  Prints SIMD cycles per step and per vector instruction (STEP_VALU, the static count of
// the loop body, is read off the disassembly).
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -o node_step_rate node_step_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

struct BoxRay { float ix, iy, iz, nx, ny, nz; uint32_t sx, sy, sz; };

__device__ __forceinline__ bool box_test(const BoxRay& r, uint32_t wx, uint32_t wy, uint32_t wz, float tcull, float& tnear, float& tfar)
{
    uint32_t px = __builtin_amdgcn_alignbit(wx, wx, r.sx), py = __builtin_amdgcn_alignbit(wy, wy, r.sy), pz = __builtin_amdgcn_alignbit(wz, wz, r.sz);
    float x0 = __builtin_fmaf((float)(px & 0xffffu), r.ix, r.nx), x1 = __builtin_fmaf((float)(px >> 16), r.ix, r.nx);
    float y0 = __builtin_fmaf((float)(py & 0xffffu), r.iy, r.ny), y1 = __builtin_fmaf((float)(py >> 16), r.iy, r.ny);
    float z0 = __builtin_fmaf((float)(pz & 0xffffu), r.iz, r.nz), z1 = __builtin_fmaf((float)(pz >> 16), r.iz, r.nz);
    float tmin = fmaxf(fmaxf(x0, y0), fmaxf(z0, 0.0f));
    float tmax = fminf(fminf(x1, y1), fminf(z1, tcull));
    tnear = tmin; tfar = tmax;
    return tmin <= tmax;
}
__device__ __forceinline__ void order_pair(uint32_t& a, uint32_t& b) { uint32_t lo = a < b ? a : b, hi = a < b ? b : a; a = hi; b = lo; }

__global__ __launch_bounds__(1024) void k_steps(int iters, uint64_t* out, uint32_t* sink, uint32_t seed)
{
    BoxRay r;
    float f = (float)(threadIdx.x + 1);
    r.ix = 0.001f * f; r.iy = -0.002f * f; r.iz = 0.0015f * f; r.nx = -3.0f; r.ny = 70.0f; r.nz = -20.0f;
    r.sx = 0; r.sy = 16; r.sz = 0;
    float tcull = 1e30f;
    uint32_t w[16];
    for (int k = 0; k < 16; k++) w[k] = (seed * (k + 3) + threadIdx.x * 2654435761u) | 0x00010000u;
    uint32_t cur = threadIdx.x, sp = 0, cullEntry = 0x7fffffffu;
    const uint32_t keyMask = 0x7ff80000u, slotMask = 0x7ffffu;
    __syncthreads();
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        uint32_t e[4];
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            float a, b;
            bool h = box_test(r, w[3 * k] ^ cur, w[3 * k + 1] + cur, w[3 * k + 2] ^ (cur << 3), tcull, a, b);
            uint32_t en = (w[12 + k] & slotMask) | (__float_as_uint(a) & keyMask);
            e[k] = h ? en : 0u;
        }
        order_pair(e[0], e[1]); order_pair(e[2], e[3]); order_pair(e[0], e[2]); order_pair(e[1], e[3]); order_pair(e[1], e[2]);
        uint32_t nearest = e[0]; int more = 0;
        #pragma unroll
        for (int k = 1; k < 4; k++) { bool h = e[k] != 0; nearest = h ? e[k] : nearest; more += h; }
        uint32_t upper = e[0] + 5, lower = e[1] + 7;
        bool ok = upper <= cullEntry;
        uint32_t old = ok ? upper : lower; int rows = ok ? 1 : 2;
        bool any = e[0] != 0;
        sp += any ? more : -rows;
        cur = (any ? nearest : old) & slotMask;
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
    if (cur + sp == 0x12345678u) sink[0] = cur;
}

int main(int argc, char** argv)
{
    const int stepValu = argc > 1 ? atoi(argv[1]) : 118;
    uint64_t* d; uint32_t* sink; hipMalloc(&d, 8 * 16 * 1024); hipMalloc(&sink, 4);
    const int iters = 20000;
    for (int wavesPerSimd : { 1, 2, 4, 6, 8 }) {
        int threads = wavesPerSimd * 256;          // one workgroup per compute unit, 4 SIMDs
        if (threads > 1024) threads = 1024;
        int groupsPerCu = wavesPerSimd * 256 / threads;
        hipLaunchKernelGGL(k_steps, dim3(256 * groupsPerCu), dim3(threads), 0, 0, iters, d, sink, 12345u);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k_steps, dim3(256 * groupsPerCu), dim3(threads), 0, 0, iters, d, sink, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        uint64_t h[16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double cyc = (double)h[0] / iters;        // cycles one wave needs per step while its SIMD holds wavesPerSimd waves
        printf("waves/SIMD %d: %.0f shader cycles per step per wave -> %.1f SIMD cycles per step, %.2f per vector instruction (%d per step); wall %.3f ms -> %.2f GHz-cycles per step per SIMD at 2.4 GHz\n",
               wavesPerSimd, cyc, cyc / wavesPerSimd, cyc / wavesPerSimd / stepValu, stepValu, ms, ms * 1e-3 * 2.4e9 / iters / wavesPerSimd);
    }
    return 0;
}
