// What a per-lane gather costs in the vector memory pipeline, by request width and count: every
// lane picks a random record of a table and reads the first B bytes of it with loads of 4, 8 or 16
// bytes (variants "n x w"), or groups of G neighbouring lanes read one record together, 16 bytes
// per lane ("coop G").  The same from LDS (64-byte records, four ds_read_b128 per lane).
// The table size decides where the records come from (L1, L2, Infinity Cache).
// Build: hipcc -O3 --offload-arch=gfx950 -o ta_width ta_width.hip ; run: ./ta_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// records are 128 bytes apart (one cache line each); V selects what is read of a record
template <int V>
__global__ __launch_bounds__(64) void k(const uint4* __restrict__ recs, uint32_t mask, int iters, uint32_t* out)
{
    uint32_t lane = threadIdx.x, acc = 0;
    uint32_t h = mix(blockIdx.x * 64 + lane + 1);
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        uint32_t idx = (h >> 8) & mask;
        const uint4* r = recs + (size_t)idx * 8;
        if (V == 0) { acc += *(const uint32_t*)r; }
        else if (V == 1) { uint2 a = *(const uint2*)r; acc += a.x + a.y; }
        else if (V == 2) { uint4 a = r[0]; acc += a.x + a.w; }
        else if (V == 3) { uint4 a = r[0], b = r[1]; acc += a.x + b.w; }
        else if (V == 4) { uint4 a = r[0], b = r[1], c = r[2]; acc += a.x + b.w + c.y; }
        else if (V == 5) { uint4 a = r[0], b = r[1], c = r[2], d = r[3]; acc += a.x + b.w + c.y + d.z; }
        else if (V == 6) { uint4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5], g = r[6], hh = r[7]; acc += a.x + b.w + c.y + d.z + e.x + f.y + g.z + hh.w; }
        else if (V == 7) {          // coop 8: eight lanes read one 128-byte record, one instruction covers 8 records
            uint32_t ig = __shfl(idx, lane & ~7u);
            uint4 a = recs[(size_t)ig * 8 + (lane & 7)];
            acc += a.x + a.w;
        } else if (V == 8) {        // coop 4: four lanes read 64 bytes of a record, one instruction covers 16 records
            uint32_t ig = __shfl(idx, lane & ~3u);
            uint4 a = recs[(size_t)ig * 8 + (lane & 3)];
            acc += a.x + a.w;
        } else if (V == 9) {        // 4 x dword: four separate 4-byte loads of one record
            const uint32_t* p = (const uint32_t*)r; acc += p[0] + p[5] + p[10] + p[15];
        } else if (V == 10) {       // 2 x dwordx2
            const uint2* p = (const uint2*)r; uint2 a = p[0], b = p[1]; acc += a.x + b.y;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(64) void k_lds(uint32_t mask, int iters, uint32_t* out)
{
    extern __shared__ uint4 tab[];        // (mask + 1) records of 64 bytes
    uint32_t lane = threadIdx.x, acc = 0;
    for (uint32_t i = lane; i < (mask + 1) * 4; i += 64) tab[i] = make_uint4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    uint32_t h = mix(blockIdx.x * 64 + lane + 1);
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        uint32_t idx = (h >> 8) & mask;
        const uint4* r = tab + idx * 4;
        uint4 a = r[0], b = r[1], c = r[2], d = r[3];
        acc += a.x + b.w + c.y + d.z;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int V> static float run(const uint4* d, uint32_t mask, int iters, int blocks, uint32_t* o)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k<V><<<blocks, 64>>>(d, mask, iters, o);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best;
}

int main()
{
    const int iters = 1000, blocks = 256 * 24;
    const char* names[] = { "1 x dword (4 B)", "1 x dwordx2 (8 B)", "1 x dwordx4 (16 B)", "2 x dwordx4 (32 B)", "3 x dwordx4 (48 B)", "4 x dwordx4 (64 B)",
                            "8 x dwordx4 (128 B)", "coop 8 lanes x 16 B (128 B)", "coop 4 lanes x 16 B (64 B)", "4 x dword (16 B)", "2 x dwordx2 (16 B)" };
    for (uint32_t nrec : { 64u, 16384u, 262144u }) {       // 8 KB (L1), 2 MB (L2), 32 MB (Infinity Cache)
        uint4* d; uint32_t* o;
        hipMalloc(&d, (size_t)nrec * 128); hipMalloc(&o, 4);
        hipMemset(d, 1, (size_t)nrec * 128);
        printf("table %u records x 128 B = %.0f KB\n", nrec, nrec * 128 / 1024.0);
        for (int v = 0; v < 11; v++) {
            float ms = 0;
            switch (v) {
            case 0: ms = run<0>(d, nrec - 1, iters, blocks, o); break; case 1: ms = run<1>(d, nrec - 1, iters, blocks, o); break;
            case 2: ms = run<2>(d, nrec - 1, iters, blocks, o); break; case 3: ms = run<3>(d, nrec - 1, iters, blocks, o); break;
            case 4: ms = run<4>(d, nrec - 1, iters, blocks, o); break; case 5: ms = run<5>(d, nrec - 1, iters, blocks, o); break;
            case 6: ms = run<6>(d, nrec - 1, iters, blocks, o); break; case 7: ms = run<7>(d, nrec - 1, iters, blocks, o); break;
            case 8: ms = run<8>(d, nrec - 1, iters, blocks, o); break; case 9: ms = run<9>(d, nrec - 1, iters, blocks, o); break;
            case 10: ms = run<10>(d, nrec - 1, iters, blocks, o); break;
            }
            // lane-iterations per second; for the coop variants one lane-iteration is 1/G of a record
            double lanePerS = (double)blocks * 64 * iters / (ms * 1e-3);
            double cyc = 256.0 * 2.4e9 / lanePerS;
            int loads = (v == 3) ? 2 : (v == 4) ? 3 : (v == 5 || v == 9) ? 4 : (v == 6) ? 8 : (v == 10) ? 2 : 1;
            printf("  %-30s %8.3f ms  %6.2f CU-cycles per lane-iteration, %5.2f per load instruction lane\n", names[v], ms, cyc, cyc / loads);
        }
        hipFree(d); hipFree(o);
    }
    {
        uint32_t* o; hipMalloc(&o, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (uint32_t nrec : { 64u, 256u }) {      // 4 KB and 16 KB of LDS per one-wave workgroup
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                k_lds<<<blocks, 64, nrec * 64>>>(nrec - 1, iters, o);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            double lanePerS = (double)blocks * 64 * iters / (best * 1e-3);
            printf("LDS table %u x 64 B per wave: %8.3f ms  %6.2f CU-cycles per 64-byte record (4 x ds_read_b128)\n", nrec, best, 256.0 * 2.4e9 / lanePerS);
        }
    }
    return 0;
}
