// The sun view: a second search structure, for shadow rays only.
//
// Every shadow ray of a frame leaves its hit point P towards the one light position S
// (CPUPathTracer.cpp:373-375, 402: origin P + 0.01 N, direction normalise(S - P), any hit with
// t >= 0 counts, however far).  Seen from S, all points of such a ray lie in (nearly) one direction.
// So the triangles are binned once per sun position into the cells of a perspective grid centred at
// S, and a shadow ray tests, with the exact ray/triangle test of the traversal, only the triangles
// of the one cell its direction falls into that begin no farther from the sun than the ray does.
// The answer is the same boolean the BVH search gives (an OR over exact tests of a set of triangles
// that contains every triangle the ray can touch), at a fraction of the work: no boxes, no stack.
//
// Geometry of the grid.  The scene's bounding box B and S are separated along one coordinate axis k
// (the one with the largest gap G between S and B; without such an axis, i.e. with the sun inside
// the box, the view is not built and the BVH answers the shadow rays).  depth(X) = |S_k - X_k| >= G
// for every X in B, and the cell of X is the one that holds
//     p(X) = ((X_i - S_i) / depth(X), (X_j - S_j) / depth(X)),   i, j the two other axes.
// p maps a triangle to a triangle (all depths are positive), so a triangle lies inside the
// bounding rectangle of its projected corners.
//
// Why one cell is enough.  With L = |S - P| and sd the unit direction, a point of the ray is
// X(t) - S = -(L - t) sd + 0.01 N.  Inside B, depth(X(t)) >= G gives L - t >= G - 0.01, so with
// eps = 0.01 / (G - 0.01), Tm the largest |p| over the corners of B and W = sqrt(1 + 2 Tm^2) >= 1 / |sd_k|
//     |p(X(t)) - p0| <= eps (1 + Tm) W / (1 - eps W),      p0 = p(P) = -sd_i / |sd_k|, -sd_j / |sd_k|.
// Every triangle is entered into all cells that the rectangle of its corners, widened by that distance
// (and a little more for rounding), touches, so the cell of p0 lists every triangle that any point of the ray inside B
// can belong to; outside B there is nothing to hit (the ray moves away from B along axis k once it
// has left it).  depth decreases along the ray, so its largest depth is that of its origin: a
// triangle whose smallest corner depth (less a margin for rounding) exceeds it cannot be met and
// ends the walk through the cell's list, which is sorted by that depth.
#include "pt_device.h"
#include <hipcub/hipcub.hpp>
#include <vector>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = MIPT_E_DEVICE_; fprintf(stderr, "sun view: %s: %s\n", #x, hipGetErrorString(e_)); goto done; } } while (0)
#define MIPT_E_DEVICE_ (-3)
#ifndef WAVE
#define WAVE 64
#endif
#define SV_SMALL_RECT 64        // rectangles of up to this many cells are entered by the thread that owns the triangle

struct SvGrid {
    float sun[3];
    int32_t axis, ax1, ax2;
    float sgn;
    float u0, v0, invCellU, invCellV;
    int32_t res;
    float spread;          // a triangle is entered into every cell its rectangle, widened by this much on all sides, touches
    float depthMargin;
};

struct SvRect { int iu0, iu1, iv0, iv1; unsigned long long key; bool ok; };

// rectangle of cells and list entry of the triangle in leaf slot i; ok = false for a triangle with a
// corner that is not finite or not on the far side of the sun plane (the view is then given up)
__device__ __forceinline__ SvRect sv_rect(const float4* __restrict__ tris, int64_t i, const SvGrid& g)
{
    SvRect r; r.ok = false; r.iu0 = r.iv0 = 0; r.iu1 = r.iv1 = -1; r.key = 0;
    float4 a4 = tris[i * MIPT_TRI_F4 + 0], b4 = tris[i * MIPT_TRI_F4 + 1], c4 = tris[i * MIPT_TRI_F4 + 2];
    float v[3][3] = { { a4.x, a4.y, a4.z }, { a4.x + b4.x, a4.y + b4.y, a4.z + b4.z }, { a4.x + c4.x, a4.y + c4.y, a4.z + c4.z } };
    float dmin = INFINITY, umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY;
    for (int j = 0; j < 3; j++) {
        float d = g.sgn * (g.sun[g.axis] - v[j][g.axis]);
        if (!(d > 0.0f) || !(d < INFINITY)) return r;
        float pu = (v[j][g.ax1] - g.sun[g.ax1]) / d, pv = (v[j][g.ax2] - g.sun[g.ax2]) / d;
        if (!(fabsf(pu) < INFINITY) || !(fabsf(pv) < INFINITY)) return r;
        dmin = fminf(dmin, d); umin = fminf(umin, pu); umax = fmaxf(umax, pu); vmin = fminf(vmin, pv); vmax = fmaxf(vmax, pv);
    }
    const float top = (float)(g.res - 1);
    auto cell = [&](float p, float p0, float inv) { float c = floorf((p - p0) * inv); return c < -4096.0f ? -4096.0f : (c > top + 4096.0f ? top + 4096.0f : c); };
    int iu0 = (int)cell(umin - g.spread, g.u0, g.invCellU), iu1 = (int)cell(umax + g.spread, g.u0, g.invCellU);
    int iv0 = (int)cell(vmin - g.spread, g.v0, g.invCellV), iv1 = (int)cell(vmax + g.spread, g.v0, g.invCellV);
    // a ray whose p0 lies outside the grid is looked up in the nearest border cell: rectangles are clamped the same way
    r.iu0 = iu0 < 0 ? 0 : (iu0 > g.res - 1 ? g.res - 1 : iu0); r.iu1 = iu1 < 0 ? 0 : (iu1 > g.res - 1 ? g.res - 1 : iu1);
    r.iv0 = iv0 < 0 ? 0 : (iv0 > g.res - 1 ? g.res - 1 : iv0); r.iv1 = iv1 < 0 ? 0 : (iv1 > g.res - 1 ? g.res - 1 : iv1);
    float lower = dmin - g.depthMargin;
    if (!(lower > 0.0f)) lower = 0.0f;
    r.key = ((unsigned long long)__float_as_uint(lower) << 32) | (unsigned long long)(uint32_t)i;      // non-negative floats order like their bit patterns
    r.ok = true;
    return r;
}

// FILL = false: counts[cell]++ for every cell of every rectangle; FILL = true: the entry is written to
// the cell's range, the count of the cell (left by the counting pass) serving as the cursor.
// Triangles with a large rectangle are set aside in bigList for k_sv_enter_big.
template <bool FILL>
__global__ void __launch_bounds__(256) k_sv_enter(const float4* __restrict__ tris, int64_t n, SvGrid g, uint32_t* counts,
                                                 const uint32_t* __restrict__ cellStart, unsigned long long* entries,
                                                 uint32_t* bigList, uint32_t* bigCount, uint32_t* invalid)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SvRect r = sv_rect(tris, i, g);
    if (!r.ok) { atomicOr(invalid, 1u); return; }
    int w = r.iu1 - r.iu0 + 1, h = r.iv1 - r.iv0 + 1;
    if ((int64_t)w * h > SV_SMALL_RECT) { bigList[atomicAdd(bigCount, 1u)] = (uint32_t)i; return; }
    for (int y = r.iv0; y <= r.iv1; y++)
        for (int x = r.iu0; x <= r.iu1; x++) {
            uint32_t c = (uint32_t)y * (uint32_t)g.res + (uint32_t)x;
            if (!FILL) atomicAdd(&counts[c], 1u);
            else entries[cellStart[c] + (atomicSub(&counts[c], 1u) - 1u)] = r.key;
        }
}

template <bool FILL>
__global__ void __launch_bounds__(256) k_sv_enter_big(const float4* __restrict__ tris, SvGrid g, uint32_t* counts,
                                                     const uint32_t* __restrict__ cellStart, unsigned long long* entries,
                                                     const uint32_t* __restrict__ bigList, const uint32_t* __restrict__ bigCount)
{
    const uint32_t nBig = *bigCount;
    for (uint32_t b = blockIdx.x; b < nBig; b += gridDim.x) {
        SvRect r = sv_rect(tris, (int64_t)bigList[b], g);
        int w = r.iu1 - r.iu0 + 1, h = r.iv1 - r.iv0 + 1;
        for (int64_t k = threadIdx.x; k < (int64_t)w * h; k += blockDim.x) {
            uint32_t c = (uint32_t)(r.iv0 + (int)(k / w)) * (uint32_t)g.res + (uint32_t)(r.iu0 + (int)(k % w));
            if (!FILL) atomicAdd(&counts[c], 1u);
            else entries[cellStart[c] + (atomicSub(&counts[c], 1u) - 1u)] = r.key;
        }
    }
}

__global__ void k_sv_offsets_end(const uint32_t* __restrict__ counts, uint32_t* cellStart, int64_t cells)
{
    // cellStart[cells] = total: start of the last cell + its count
    if (blockIdx.x == 0 && threadIdx.x == 0) cellStart[cells] = cellStart[cells - 1] + counts[cells - 1];
}

void mipt_free_sun_view(SunView* v)
{
    if (!v) return;
    void* ptrs[] = { v->cellStart, v->entries, v->cellOf, v->rankOf, v->order, v->rayCount, v->rayStart, v->scanTemp };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    *v = SunView();
}

// Builds the view for the sun position `sun` over the triangles of the scene's leaf slots.  On return
// out->dev.axis < 0 means that no view exists for this sun (it lies inside or too close to the
// scene's box, or a triangle is not finite): the caller then sends the shadow rays through the BVH.
int mipt_build_sun_view(const DevScene& S, int64_t numLeafSlots, const float sun[3], hipStream_t stream, SunView* out)
{
    int rc = 0;
    mipt_free_sun_view(out);
    out->dev.axis = -1;
    for (int a = 0; a < 3; a++) out->sun[a] = sun[a];
    out->built = true;
    if (numLeafSlots <= 0 || getenv("MIPT_NO_SUN_VIEW")) return 0;
    if (!(std::isfinite(sun[0]) && std::isfinite(sun[1]) && std::isfinite(sun[2]))) return 0;

    // the box of the node grid holds every triangle
    double lo[3], hi[3], ext = 0;
    for (int a = 0; a < 3; a++) { lo[a] = S.qmin[a]; hi[a] = (double)S.qmin[a] + 65535.0 * (double)S.qcell[a]; ext = std::max(ext, hi[a] - lo[a]); }
    int axis = -1; double gap = 0, sgn = 1;
    for (int a = 0; a < 3; a++) {
        if (sun[a] - hi[a] > gap) { gap = sun[a] - hi[a]; axis = a; sgn = 1; }
        if (lo[a] - sun[a] > gap) { gap = lo[a] - sun[a]; axis = a; sgn = -1; }
    }
    if (axis < 0 || gap < 0.5) return 0;                    // the bound above needs 0.01 << gap
    const int ax1 = (axis + 1) % 3, ax2 = (axis + 2) % 3;
    double u0 = 1e300, u1 = -1e300, v0 = 1e300, v1 = -1e300;
    for (int c = 0; c < 8; c++) {
        double X[3] = { (c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2] };
        double d = sgn * (sun[axis] - X[axis]);
        double pu = (X[ax1] - sun[ax1]) / d, pv = (X[ax2] - sun[ax2]) / d;
        u0 = std::min(u0, pu); u1 = std::max(u1, pu); v0 = std::min(v0, pv); v1 = std::max(v1, pv);
    }
    const double Tm = std::max(std::max(std::fabs(u0), std::fabs(u1)), std::max(std::fabs(v0), std::fabs(v1)));
    if (!(Tm < 16.0)) return 0;                              // sun close to a face of the box: the grid would be mostly empty
    const double eps = 0.01 / (gap - 0.01), W = std::sqrt(1.0 + 2.0 * Tm * Tm);
    if (eps * W > 0.05) return 0;
    const double rho = 1.25 * eps * (1.0 + Tm) * W / (1.0 - eps * W);
    const double padRange = 1e-4 * std::max(u1 - u0, v1 - v0) + 1e-30;
    u0 -= padRange; u1 += padRange; v0 -= padRange; v1 += padRange;
    int res = (int)(2.0 * std::sqrt((double)numLeafSlots));
    if (const char* e = getenv("MIPT_SUN_VIEW_RES")) res = atoi(e);
    res = (res + 15) / 16 * 16;
    if (res < 64) res = 64;
    if (res > 4096) res = 4096;
    // spread of a ray's points around p0, plus the rounding of p on both sides (a few ulp of Tm) and a
    // thousandth of a cell for the rounding of the cell index
    double spread = rho + 8e-7 * (1.0 + Tm);
    while (res > 64 && spread > 0.5 * std::min(u1 - u0, v1 - v0) / res) res /= 2;     // cells finer than the spread buy nothing
    spread += 1e-3 * std::max(u1 - u0, v1 - v0) / res;
    if (const char* e = getenv("MIPT_SUN_VIEW_SPREAD")) spread *= atof(e);

    SvGrid g;
    for (int a = 0; a < 3; a++) g.sun[a] = sun[a];
    g.axis = axis; g.ax1 = ax1; g.ax2 = ax2; g.sgn = (float)sgn;
    g.u0 = (float)u0; g.v0 = (float)v0; g.invCellU = (float)(res / (u1 - u0)); g.invCellV = (float)(res / (v1 - v0));
    g.res = res; g.spread = (float)spread;
    {
        double dist = 0;
        for (int a = 0; a < 3; a++) dist = std::max(dist, std::max(std::fabs(sun[a] - lo[a]), std::fabs(sun[a] - hi[a])));
        g.depthMargin = (float)(1e-4 * ext + 1e-5 * dist);
        if (const char* e = getenv("MIPT_SUN_VIEW_MARGIN")) g.depthMargin = (float)atof(e);
    }

    const int64_t cells = (int64_t)res * res;
    uint32_t* counts = nullptr; uint32_t* bigList = nullptr; uint32_t* small = nullptr;      // small: [0] big count, [1] invalid flag
    unsigned long long* unsorted = nullptr; void* temp = nullptr; size_t tempBytes = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    uint32_t h[2] = { 0, 0 }, total = 0;
    const unsigned triBlocks = (unsigned)((numLeafSlots + 255) / 256);
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, stream));
    HIPCHK(hipMalloc(&counts, 4 * (size_t)cells));
    HIPCHK(hipMalloc(&out->cellStart, 4 * (size_t)(cells + 1)));
    HIPCHK(hipMalloc(&bigList, 4 * (size_t)numLeafSlots));
    HIPCHK(hipMalloc(&small, 8));
    HIPCHK(hipMemsetAsync(counts, 0, 4 * (size_t)cells, stream));
    HIPCHK(hipMemsetAsync(small, 0, 8, stream));
    hipLaunchKernelGGL((k_sv_enter<false>), dim3(triBlocks), dim3(256), 0, stream, S.tris, numLeafSlots, g, counts, (const uint32_t*)nullptr,
                       (unsigned long long*)nullptr, bigList, small, small + 1);
    hipLaunchKernelGGL((k_sv_enter_big<false>), dim3(1024), dim3(256), 0, stream, S.tris, g, counts, (const uint32_t*)nullptr,
                       (unsigned long long*)nullptr, bigList, small);
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tempBytes, counts, out->cellStart, (int)cells, stream));
    HIPCHK(hipMalloc(&temp, tempBytes ? tempBytes : 16));
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(temp, tempBytes, counts, out->cellStart, (int)cells, stream));
    hipLaunchKernelGGL(k_sv_offsets_end, dim3(1), dim3(64), 0, stream, counts, out->cellStart, cells);
    HIPCHK(hipMemcpyAsync(h, small, 8, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(&total, out->cellStart + cells, 4, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (h[1] != 0 || total == 0 || (uint64_t)total >= (1ull << 31)) goto done;          // no view: axis stays -1
    HIPCHK(hipMalloc(&unsorted, 8 * (size_t)total));
    HIPCHK(hipMalloc(&out->entries, 8 * (size_t)total));
    HIPCHK(hipMemsetAsync(small, 0, 8, stream));
    hipLaunchKernelGGL((k_sv_enter<true>), dim3(triBlocks), dim3(256), 0, stream, S.tris, numLeafSlots, g, counts, (const uint32_t*)out->cellStart,
                       unsorted, bigList, small, small + 1);
    hipLaunchKernelGGL((k_sv_enter_big<true>), dim3(1024), dim3(256), 0, stream, S.tris, g, counts, (const uint32_t*)out->cellStart,
                       unsorted, bigList, small);
    // every cell's list in ascending order of the depth bound (the upper 32 bits of an entry)
    (void)hipFree(temp); temp = nullptr; tempBytes = 0;
    HIPCHK(hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, tempBytes, unsorted, out->entries, (int)total, (int)cells,
                                                      out->cellStart, out->cellStart + 1, 0, 64, stream));
    HIPCHK(hipMalloc(&temp, tempBytes ? tempBytes : 16));
    HIPCHK(hipcub::DeviceSegmentedRadixSort::SortKeys(temp, tempBytes, unsorted, out->entries, (int)total, (int)cells,
                                                      out->cellStart, out->cellStart + 1, 0, 64, stream));
    HIPCHK(hipEventRecord(e1, stream));
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipGetLastError());
    {
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        out->buildMs = ms;
    }
    out->numEntries = total; out->res = res;
    out->dev.cellStart = out->cellStart; out->dev.entries = out->entries;
    for (int a = 0; a < 3; a++) out->dev.sun[a] = sun[a];
    out->dev.ax1 = ax1; out->dev.ax2 = ax2; out->dev.sgn = g.sgn;
    out->dev.u0 = g.u0; out->dev.v0 = g.v0; out->dev.invCellU = g.invCellU; out->dev.invCellV = g.invCellV; out->dev.res = res;
    out->dev.axis = axis;
    if (getenv("MIPT_BUILD_VERBOSE"))
        fprintf(stderr, "sun view: axis %d gap %.3g, %d x %d cells, spread %.3g cells, %u entries (%.2f per triangle, %.2f per cell), %u triangles with large rectangles, %.2f ms\n",
                axis, gap, res, res, spread * res / std::min(u1 - u0, v1 - v0), total, (double)total / (double)numLeafSlots, (double)total / (double)cells, h[0], out->buildMs);
done:
    if (counts) (void)hipFree(counts);
    if (bigList) (void)hipFree(bigList);
    if (small) (void)hipFree(small);
    if (unsorted) (void)hipFree(unsorted);
    if (temp) (void)hipFree(temp);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (rc != 0 || out->dev.axis < 0) {
        if (out->cellStart) (void)hipFree(out->cellStart);
        if (out->entries) (void)hipFree(out->entries);
        out->cellStart = nullptr; out->entries = nullptr; out->dev.axis = -1;
    }
    return rc;
}

// ---------------------------------------------------------------------------------------
// Shadow rays answered from the sun view: q[i] for i < *d_count (a queue in blocks: only the first
// blockCount[b] entries of block b).  The rays are first ordered by the cell they look up (a counting
// sort of their indices: one atomic per ray hands out its rank within the cell), so that the rays
// of a wave walk the same list, or neighbouring ones: the lists and the triangles they name are then
// read once per wave instead of once per ray, from L1, and the lanes of a wave stay together.
// An occluded ray clears the direct term of the slot it belongs to, exactly as the any-hit
// traversal does (MODE 1 of k_trace).
__device__ __forceinline__ uint32_t sv_cell_of(const DevSunView& V, float4 d4)
{
    const float dd[3] = { d4.x, d4.y, d4.z };
    // cell of the direction from the sun to the hit point: p0 = -sd / |sd_k| on the two other axes
    float dk = V.sgn * dd[V.axis];                   // > 0: the ray runs towards the sun's side of the box
    float inv = 1.0f / dk;
    float pu = -dd[V.ax1] * inv, pv = -dd[V.ax2] * inv;
    float cu = floorf((pu - V.u0) * V.invCellU), cv = floorf((pv - V.v0) * V.invCellV);
    const float top = (float)(V.res - 1);
    // (a direction that does not point at the sun's side, or is not finite, cannot come from the shading kernels;
    // it would land in a border cell and be answered from that cell's list, like a p0 outside the grid)
    int iu = (int)(cu > 0.0f ? (cu < top ? cu : top) : 0.0f), iv = (int)(cv > 0.0f ? (cv < top ? cv : top) : 0.0f);
    return (uint32_t)iv * (uint32_t)V.res + (uint32_t)iu;
}

#define SV_NO_CELL 0xFFFFFFFFu

template <bool BLOCKS>
__global__ void __launch_bounds__(256) k_sv_bin(DevSunView V, RayQueue q, const int32_t* __restrict__ d_count, const int32_t* __restrict__ blockCount,
                                               uint32_t* rayCount, uint32_t* cellOf, uint32_t* rankOf)
{
    const int n = *d_count;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (BLOCKS && (int)(i & ((1 << MIPT_QBLOCK_SHIFT) - 1)) >= blockCount[i >> MIPT_QBLOCK_SHIFT]) { cellOf[i] = SV_NO_CELL; continue; }
        uint32_t c = sv_cell_of(V, q.d[i]);
        cellOf[i] = c;
        rankOf[i] = atomicAdd(&rayCount[c], 1u);
    }
}

__global__ void __launch_bounds__(256) k_sv_order(const int32_t* __restrict__ d_count, const uint32_t* __restrict__ cellOf, const uint32_t* __restrict__ rankOf,
                                                 const uint32_t* __restrict__ rayStart, uint32_t* order)
{
    const int n = *d_count;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t c = cellOf[i];
        if (c != SV_NO_CELL) order[rayStart[c] + rankOf[i]] = (uint32_t)i;
    }
}

// rayStart[cells] (the number of rays that were binned) = start of the last cell + its count
__global__ void k_sv_total(const uint32_t* __restrict__ rayCount, uint32_t* rayStart, int64_t cells)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) rayStart[cells] = rayStart[cells - 1] + rayCount[cells - 1];
}

__global__ void __launch_bounds__(256) k_sun_shadow(DevScene S, DevSunView V, RayQueue q, const uint32_t* __restrict__ order, const uint32_t* __restrict__ cellOf,
                                                   const uint32_t* __restrict__ d_total, float4* pixState0, float4* pathState,
                                                   int64_t slotsPerLevel, float eps, unsigned long long* stats)
{
    const uint32_t n = *d_total;
    uint32_t rays = 0, tests = 0;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t i = order[j];
        float4 o4 = q.o[i], d4 = q.d[i];
        F3 o = f3(o4.x, o4.y, o4.z), d = f3(d4.x, d4.y, d4.z);
        const float oo[3] = { o4.x, o4.y, o4.z };
        rays++;
#ifdef MIPT_SV_HIST
        const uint32_t testsBefore = tests;
#endif
        const uint32_t cell = cellOf[i];
        uint32_t e = V.cellStart[cell], eEnd = V.cellStart[cell + 1];
        float zs = V.sgn * (V.sun[V.axis] - oo[V.axis]);  // the largest depth on the ray
        bool occluded = false;
        for (; e < eEnd; e++) {
            unsigned long long key = V.entries[e];
            if (__uint_as_float((uint32_t)(key >> 32)) > zs) break;       // this and all later triangles of the list begin behind the ray's origin
            const float4* tr = S.tris + (size_t)(uint32_t)key * MIPT_TRI_F4;
            float4 a = tr[0], b = tr[1], c = tr[2];
            tests++;
            float t, u, v;
            if (ray_triangle(o, d, f3(a.x, a.y, a.z), f3(b.x, b.y, b.z), f3(c.x, c.y, c.z), eps, t, u, v) && t < FLT_MAX) { occluded = true; break; }
        }
        if (occluded) {       // no direct light at this hit (PathTracer.cu:421-428)
            int tag0 = __float_as_int(o4.w), tag1 = __float_as_int(d4.w);
            if (tag1 == 0) pixState0[(size_t)tag0 * 4].x = 0.0f;
            else pathState[(size_t)(tag1 - 1) * slotsPerLevel + tag0].x = 0.0f;
        }
#ifdef MIPT_SV_HIST
        {
            uint32_t len = eEnd - V.cellStart[cell], done = tests - testsBefore;
            int b = len <= 4 ? 0 : len <= 8 ? 1 : len <= 16 ? 2 : len <= 64 ? 3 : 4;
            atomicAdd(&stats[ST_PROF + b], 1ull); (void)done;
            int b2 = done <= 4 ? 0 : done <= 16 ? 1 : 2;
            atomicAdd(&stats[ST_PROF + 5 + b2], (unsigned long long)done);
        }
#endif
    }
    unsigned long long rs = wave_sum_u32(rays), ts = wave_sum_u32(tests);
    if ((threadIdx.x & (WAVE - 1)) == 0 && rs) {
        atomicAdd(&stats[ST_SHADOW_RAYS], rs); atomicAdd(&stats[ST_SUNVIEW_RAYS], rs);
        atomicAdd(&stats[ST_SHADOW_TRIS], ts);
    }
}

// Working memory of the ray ordering: three words per queue entry, two per cell, and the scan's own.
int mipt_sun_view_reserve(SunView* v, int64_t maxRays)
{
    if (v->dev.axis < 0) return 0;
    const int64_t cells = (int64_t)v->res * v->res;
    if (v->rayCap >= maxRays) return 0;
    void* old[] = { v->cellOf, v->rankOf, v->order, v->rayCount, v->rayStart, v->scanTemp };
    for (void* p : old) if (p) (void)hipFree(p);
    v->cellOf = v->rankOf = v->order = v->rayCount = v->rayStart = nullptr; v->scanTemp = nullptr; v->rayCap = 0;
    size_t tb = 0;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, tb, v->rayCount, v->rayStart, (int)cells, (hipStream_t)0) != hipSuccess) return MIPT_E_DEVICE_;
    v->scanTempBytes = tb ? tb : 16;
    if (hipMalloc(&v->cellOf, 4 * (size_t)maxRays) != hipSuccess || hipMalloc(&v->rankOf, 4 * (size_t)maxRays) != hipSuccess ||
        hipMalloc(&v->order, 4 * (size_t)maxRays) != hipSuccess || hipMalloc(&v->rayCount, 4 * (size_t)cells) != hipSuccess ||
        hipMalloc(&v->rayStart, 4 * (size_t)(cells + 1)) != hipSuccess || hipMalloc(&v->scanTemp, v->scanTempBytes) != hipSuccess)
        return MIPT_E_DEVICE_;
    v->rayCap = maxRays;
    return 0;
}

void launch_sun_shadow(const DevScene& S, SunView& v, RayQueue q, const int32_t* d_count, const int32_t* d_blockCount,
                       int32_t maxCount, float4* pixState0, float4* pathState, int64_t slotsPerLevel, float eps,
                       unsigned long long* d_stats, int numCUs, hipStream_t s)
{
    if (maxCount <= 0 || maxCount > v.rayCap) return;
    const int64_t cells = (int64_t)v.res * v.res;
    long long blocks = ((long long)maxCount + 255) / 256;
    long long cap = (long long)numCUs * 8;          // the threads stride over the queue: few, long-lived workgroups, one atomic pair per wave
    if (blocks > cap) blocks = cap;
    (void)hipMemsetAsync(v.rayCount, 0, 4 * (size_t)cells, s);
    if (d_blockCount) hipLaunchKernelGGL((k_sv_bin<true>), dim3((unsigned)blocks), dim3(256), 0, s, v.dev, q, d_count, d_blockCount, v.rayCount, v.cellOf, v.rankOf);
    else hipLaunchKernelGGL((k_sv_bin<false>), dim3((unsigned)blocks), dim3(256), 0, s, v.dev, q, d_count, d_blockCount, v.rayCount, v.cellOf, v.rankOf);
    (void)hipcub::DeviceScan::ExclusiveSum(v.scanTemp, v.scanTempBytes, v.rayCount, v.rayStart, (int)cells, s);
    hipLaunchKernelGGL(k_sv_total, dim3(1), dim3(64), 0, s, v.rayCount, v.rayStart, cells);
    hipLaunchKernelGGL(k_sv_order, dim3((unsigned)blocks), dim3(256), 0, s, d_count, v.cellOf, v.rankOf, v.rayStart, v.order);
    hipLaunchKernelGGL(k_sun_shadow, dim3((unsigned)blocks), dim3(256), 0, s, S, v.dev, q, v.order, v.cellOf, v.rayStart + cells, pixState0, pathState, slotsPerLevel, eps, d_stats);
}
