// Per-lane pointer chase: every lane reads a 32-byte record (two 16-byte loads), does K dependent
// fused multiply-adds on it and derives the index of its next record from the data.  This is the
// memory/compute skeleton of a one-ray-per-lane BVH walk.  Prints record visits per CU cycle for
// several working-set sizes, amounts of arithmetic and numbers of waves per CU.
// Build: hipcc -O3 --offload-arch=gfx950 -o chase chase.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int K, int STRIDE, int LOADS>
__global__ __launch_bounds__(64) void k(const uint4* __restrict__ recs, uint32_t mask, int iters, uint32_t* out)
{
    extern __shared__ uint32_t pad[];
    uint32_t idx = (blockIdx.x * 64 + threadIdx.x) * 2654435761u & mask;
    float acc = 0.0f;
    for (int i = 0; i < iters; i++) {
        uint4 a = recs[STRIDE * idx], b = recs[STRIDE * idx + 1];
        if (LOADS >= 3) { uint4 c = recs[STRIDE * idx + 2]; b.w += c.x; }
        if (LOADS >= 4) { uint4 c = recs[STRIDE * idx + 3]; b.w += c.y; }
        float x = __uint_as_float((a.y & 0x007fffffu) | 0x3f800000u);
        #pragma unroll
        for (int j = 0; j < K; j++) x = __builtin_fmaf(x, 0.999f, 0.001f);
        acc += x;
        idx = (a.x + b.w + (uint32_t)i) & mask;          // next record comes from the data
    }
    if (acc == 12345.0f) out[0] = idx + pad[0];
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int iters = 4000;
    for (uint32_t nrec : { 512u, 2048u, 8192u, 32768u, 131072u, 524288u }) {
        std::vector<uint32_t> h((size_t)nrec * 8);
        uint32_t s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 4; }
        uint4* d; uint32_t* o;
        hipMalloc(&d, (size_t)nrec * 32); hipMalloc(&o, 4);
        hipMemcpy(d, h.data(), (size_t)nrec * 32, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        // the four-load variant (kk == 2) is kept in the source but not run: its index chain collapses into a short cycle
        for (int kk = 0; kk < 2; kk++) {
            for (int wavesPerCU : { 24 }) {
                size_t lds = (160 * 1024) / wavesPerCU - 64;     // LDS sized so that exactly this many waves fit
                if (lds > 64 * 1024) lds = 64 * 1024 - 64;
                int blocks = cus * wavesPerCU;
                float best = 1e30f;
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(e0);
                    if (kk == 0) hipLaunchKernelGGL((k<32, 2, 2>), dim3(blocks), dim3(64), lds, 0, d, nrec - 1, iters, o);
                    if (kk == 1) hipLaunchKernelGGL((k<32, 4, 3>), dim3(blocks), dim3(64), lds, 0, d, nrec / 2 - 1, iters, o);
                    if (kk == 2) hipLaunchKernelGGL((k<32, 4, 4>), dim3(blocks), dim3(64), lds, 0, d, nrec / 2 - 1, iters, o);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                double visits = (double)blocks * 64 * iters;
                printf("set %7u x 32 B (%6.0f KB) %s waves/CU %2d: %7.1f Gvisits/s, %.3f lane visits per CU cycle (2.4 GHz), %.0f ns per step\n",
                       nrec, nrec * 32 / 1024.0, kk == 0 ? "32-B records, 2 loads" : kk == 1 ? "64-B stride, 3 loads " : "64-B records, 4 loads", wavesPerCU, visits / best * 1e-6, visits / (best * 1e-3) / (cus * 2.4e9), best * 1e6 / iters);
            }
        }
        hipFree(d); hipFree(o);
    }
    return 0;
}
