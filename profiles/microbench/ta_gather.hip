// Cost of gathering 32-byte records by per-lane index: every lane issues two 16-byte loads for
// its own record (variant 0), or lanes pair up so that each 16-byte load instruction covers 32
// records with two neighbouring lanes reading the two halves of one record (variant 1), or four
// lanes read one 64-byte record (variant 2, one instruction per 16 records).
// Build: hipcc -O3 --offload-arch=gfx950 -o ta_gather ta_gather.hip ; run: ./ta_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int V>
__global__ __launch_bounds__(64) void k(const uint4* __restrict__ recs, uint32_t mask, int iters, uint32_t* out)
{
    uint32_t lane = threadIdx.x, acc = 0;
    uint32_t h = mix(blockIdx.x * 64 + lane + 1);
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        uint32_t idx = (h >> 8) & mask;            // record wanted by this lane
        if (V == 0) {
            uint4 a = recs[2 * idx], b = recs[2 * idx + 1];
            acc += a.x + b.w;
        } else if (V == 1) {
            uint32_t i0 = __shfl(idx, lane >> 1), i1 = __shfl(idx, 32 + (lane >> 1));
            uint4 a = recs[2 * i0 + (lane & 1)], b = recs[2 * i1 + (lane & 1)];
            acc += a.x + b.w;
        } else if (V == 2) {
            // 64-byte records: four lanes per record, four instructions for 64 records
            #pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t iq = __shfl(idx, q * 16 + (lane >> 2)) & (mask >> 1);
                uint4 a = recs[4 * iq + (lane & 3)];
                acc += a.x;
            }
        } else {
            // 64-byte records, every lane reads its whole record with four loads
            uint32_t iq = idx & (mask >> 1);
            uint4 a = recs[4 * iq], b = recs[4 * iq + 1], c = recs[4 * iq + 2], d = recs[4 * iq + 3];
            acc += a.x + b.y + c.z + d.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    const int iters = 2000, blocks = 256 * 24;
    for (uint32_t nrec : { 256u, 8192u, 65536u, 262144u }) {
        uint4* d; uint32_t* o;
        hipMalloc(&d, (size_t)nrec * 32); hipMalloc(&o, 4);
        hipMemset(d, 1, (size_t)nrec * 32);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int v = 0; v < 4; v++) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (v == 0) k<0><<<blocks, 64>>>(d, nrec - 1, iters, o);
                if (v == 1) k<1><<<blocks, 64>>>(d, nrec - 1, iters, o);
                if (v == 2) k<2><<<blocks, 64>>>(d, nrec - 1, iters, o);
                if (v == 3) k<3><<<blocks, 64>>>(d, nrec - 1, iters, o);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            double recsPerS = (double)blocks * 64 * iters / (best * 1e-3);
            double cuCyclesPerRec = 256.0 * 2.4e9 / recsPerS;
            printf("records %7u (%6.0f KB) variant %d: %8.3f ms  %7.1f Grec/s  %5.2f CU-cycles per record\n", nrec, nrec * 32 / 1024.0, v, best, recsPerS * 1e-9, cuCyclesPerRec);
        }
        hipFree(d); hipFree(o);
    }
    return 0;
}
