// Cost of gathering one shading record per lane by a random triangle index, for record strides
// of 80 bytes (five 16-byte loads, two 64-byte lines), 128 bytes (the same five loads, one
// aligned 128-byte line) and 64 bytes (four loads, one line), with and without a 4-byte gather
// from a 100 MB texel table behind it.
// Build: hipcc -O3 --offload-arch=gfx950 -o shade_gather shade_gather.hip ; run: ./shade_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int STRIDE16, int LOADS, bool TEX>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ recs, uint32_t nrec, const uint32_t* __restrict__ tex, uint32_t ntex, int iters, uint32_t* out)
{
    uint32_t acc = 0;
    uint32_t h = mix(blockIdx.x * 256 + threadIdx.x + 1);
    for (int i = 0; i < iters; i++) {
        h = mix(h + i);
        uint32_t idx = h % nrec;
        const uint4* r = recs + (size_t)idx * STRIDE16;
        uint4 a = r[0], b = r[1], c = r[2], d = r[3];
        uint32_t s = a.x + b.y + c.z + d.w;
        if (LOADS == 5) { uint4 e = r[4]; s += e.x; }
        if (TEX) s += tex[(s ^ h) % ntex];
        acc += s;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    const int iters = 200, blocks = 256 * 8;
    const uint32_t nrec = 262144, ntex = 25u << 20;
    uint4* d; uint32_t* o; uint32_t* t;
    hipMalloc(&d, (size_t)nrec * 128); hipMalloc(&o, 4); hipMalloc(&t, (size_t)ntex * 4);
    hipMemset(d, 1, (size_t)nrec * 128); hipMemset(t, 1, (size_t)ntex * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 6; v++) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            switch (v) {
            case 0: k<5, 5, false><<<blocks, 256>>>(d, nrec, t, ntex, iters, o); break;
            case 1: k<8, 5, false><<<blocks, 256>>>(d, nrec, t, ntex, iters, o); break;
            case 2: k<4, 4, false><<<blocks, 256>>>(d, nrec, t, ntex, iters, o); break;
            case 3: k<5, 5, true><<<blocks, 256>>>(d, nrec, t, ntex, iters, o); break;
            case 4: k<8, 5, true><<<blocks, 256>>>(d, nrec, t, ntex, iters, o); break;
            case 5: k<4, 4, true><<<blocks, 256>>>(d, nrec, t, ntex, iters, o); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const char* names[] = { "80-B stride", "128-B stride", "64-B stride", "80-B stride + texel", "128-B stride + texel", "64-B stride + texel" };
        double recsPerS = (double)blocks * 256 * iters / (best * 1e-3);
        printf("%-22s %8.3f ms  %7.1f Grec/s  %5.2f CU-cycles per record\n", names[v], best, recsPerS * 1e-9, 256.0 * 2.4e9 / recsPerS);
    }
    return 0;
}
