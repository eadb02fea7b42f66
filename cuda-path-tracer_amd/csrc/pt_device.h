// Device-side arithmetic of the path tracer: float3 helpers, the ray/triangle test, the
// BVH box test, sampling, the counter-based random stream and wave64 queue primitives.
//
// Everything that decides a result (which triangle is hit, at what t, with what radiance)
// is plain IEEE fp32 with one rounding per operation, in the operand order of the
// reference's CPU path, so that results can be compared bit for bit with a CPU
// restatement.  The translation unit is compiled with -ffp-contract=off; the only fused
// operations are the explicit __builtin_fmaf calls of the box test, which is conservative
// by construction (boxes are padded) and never decides a result.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include "pt_internal.h"

struct F3 { float x, y, z; };

__device__ __forceinline__ F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ F3 f3sub(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 f3add(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 f3scale(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
// XMVector3Dot: (x*x' + y*y') + z*z'
__device__ __forceinline__ float f3dot(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// XMVector3Cross
__device__ __forceinline__ F3 f3cross(F3 a, F3 b)
{
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// XMVector3Normalize: v / sqrt(dot(v,v)), 0 for a zero vector, NaN for an infinite one
__device__ __forceinline__ F3 f3normalize(F3 v)
{
    float lsq = f3dot(v, v);
    float len = sqrtf(lsq);
    if (len == 0.0f) return f3(0.0f, 0.0f, 0.0f);
    if (lsq == INFINITY) return f3(NAN, NAN, NAN);
    return f3(v.x / len, v.y / len, v.z / len);
}

// ---------------------------------------------------------------------------------------
// Random stream of the harness: uniform in [0,1) keyed by (seed, pixel, sample, depth, dim).
// Replaces cuRAND XORWOW of PathTracer.cu:516-518,436-437 and mt19937 of
// CPUPathTracer.cpp:345-346,418-419.
__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float rng_u01(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t dim)
{
    uint32_t h = mix32(seed + 0x9e3779b9U);
    h = mix32(h ^ pixel);
    h = mix32(h ^ sample);
    h = mix32(h ^ (depth * 2U + dim + 0x632be5abU));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

#define PT_PI 3.14159265f      /* M_PI of PathTracer.cu:15 / CPUPathTracer.cpp:12 */
#define PT_BIAS 0.01f          /* PathTracer.cu:400 */

// uniformSampleHemisphere, CPUPathTracer.cpp:448-456 / PathTracer.cu:362-371
__device__ __forceinline__ F3 sample_hemisphere(float r1, float r2)
{
    float sinTheta = sqrtf(1 - r1 * r1);
    float phi = 2 * PT_PI * r2;
    float x = sinTheta * cosf(phi);
    float z = sinTheta * sinf(phi);
    return f3(x, r1, z);
}

// CreateCoordinateSystem, CPUPathTracer.cpp:458-472: the divisor is hypot() evaluated in
// double and rounded to float
__device__ __forceinline__ void coordinate_system(F3 N, F3& Nt, F3& Nb)
{
    if (fabsf(N.x) > fabsf(N.y)) {
        float h = (float)sqrt((double)N.x * (double)N.x + (double)N.z * (double)N.z);
        Nt = f3(N.z / h, 0.0f / h, -N.x / h);
    } else {
        float h = (float)sqrt((double)N.y * (double)N.y + (double)N.z * (double)N.z);
        Nt = f3(0.0f / h, -N.z / h, N.y / h);
    }
    Nb = f3cross(N, Nt);
}

// world = x*Nb + y*N + z*Nt, CPUPathTracer.cpp:421-425
__device__ __forceinline__ F3 to_world(F3 s, F3 N, F3 Nt, F3 Nb)
{
    return f3(s.x * Nb.x + s.y * N.x + s.z * Nt.x,
              s.x * Nb.y + s.y * N.y + s.z * Nt.y,
              s.x * Nb.z + s.y * N.z + s.z * Nt.z);
}

// ---------------------------------------------------------------------------------------
// XNA::IntersectRayTriangle, xnacollision.cpp:1068-1168 (epsilon as a parameter; 1e-5
// gives IntersectRayTriangle of PathTracer.cu:149-191).  e1 = v1 - v0 and e2 = v2 - v0 are
// the same fp32 differences the reference forms per test, taken once at build time.
// Returns true on acceptance with t, u, v divided by det.
__device__ __forceinline__ bool ray_triangle(F3 o, F3 d, F3 v0, F3 e1, F3 e2, float eps, float& t, float& u, float& v)
{
    F3 p = f3cross(d, e2);
    float det = f3dot(e1, p);
    F3 s = f3sub(o, v0);
    float uu = f3dot(s, p);
    F3 q = f3cross(s, e1);
    float vv = f3dot(d, q);
    float tt = f3dot(e2, q);
    bool pos = det >= eps;
    bool neg = det <= -eps;
    // the reference collects "no intersection" conditions, so a NaN comparison does not reject
    bool rejP = (uu < 0.0f) | (uu > det) | (vv < 0.0f) | ((uu + vv) > det) | (tt < 0.0f);
    bool rejN = (uu > 0.0f) | (uu < det) | (vv > 0.0f) | ((uu + vv) < det) | (tt > 0.0f);
    bool accept = pos ? !rejP : (neg & !rejN);
    if (accept) {
        float inv = 1.0f / det;
        t = tt * inv; u = uu * inv; v = vv * inv;
    }
    return accept;
}

// The same test with the division done on every lane, accepted or not (straight-line code for the
// traversal; t, u, v of a rejected triangle are not used).  Same operations in the same order.
__device__ __forceinline__ bool ray_triangle_all(F3 o, F3 d, F3 v0, F3 e1, F3 e2, float eps, float& t, float& u, float& v)
{
    F3 p = f3cross(d, e2);
    float det = f3dot(e1, p);
    F3 s = f3sub(o, v0);
    float uu = f3dot(s, p);
    F3 q = f3cross(s, e1);
    float vv = f3dot(d, q);
    float tt = f3dot(e2, q);
    bool pos = det >= eps;
    bool neg = det <= -eps;
    bool rejP = (uu < 0.0f) | (uu > det) | (vv < 0.0f) | ((uu + vv) > det) | (tt < 0.0f);
    bool rejN = (uu > 0.0f) | (uu < det) | (vv > 0.0f) | ((uu + vv) < det) | (tt > 0.0f);
    float inv = 1.0f / det;
    t = tt * inv; u = uu * inv; v = vv * inv;
    return (pos & !rejP) | (!pos & neg & !rejN);
}

// ---------------------------------------------------------------------------------------
// Ray prepared for the box test on the 16-bit grid: with box coordinate = qmin + q * qcell the
// slab distance (coordinate - origin) / direction becomes q * (qcell / d) + (qmin - o) / d, one
// fused multiply-add per plane: a plane at grid coordinate q is crossed at t = q * i + n.  A node word holds
// (min | max << 16) of one child on one axis.  A direction component below 1e-20 is replaced by 1e-20 with its
// sign: along such an axis the ray does not leave a box's margin within 10^15 scene sizes, and the products stay
// finite for scenes up to 10^16 units.
//
// THE MARGIN.  The box test must never reject a box that holds a triangle the exact test accepts, although both
// are computed in rounded fp32 arithmetic.  The node boxes are rounded outwards to the grid and carry next to no
// margin of their own (bvh_build.hip); the margin is applied here, per ray and axis, as two constant terms: the
// plane a ray meets first on an axis uses cn = c - m / |d|, the plane it meets last cf = c + m / |d|, which is
// the slab test of the box grown by m on that axis.  m = MIPT_BOX_MARGIN_ULPS units in the last place of the
// ray's reach (the largest distance from its origin to a face of the grid, plus the grid's extent: the magnitude of
// every operand of both tests, whose rounding errors are a few of those units each), plus, for the permute decode,
// 0.51 grid cells of the axis.  Costs nothing per node step (a plane's multiply-add takes cn or cf where it took c),
// grows with the distance of a far-away origin as the errors do, and is smaller than the 1.5 * 2^-15 of the largest
// coordinate that the boxes would have to carry instead: about two thirds of it on the scene's longest axis, far less
// on a short one, whose cells are small (10 M triangles in 38 instances side by side: a quarter fewer triangle tests).
#ifndef MIPT_BOX_MARGIN_ULPS
#define MIPT_BOX_MARGIN_ULPS 32.0f
#endif
#ifndef MIPT_BOX_MARGIN_CELLS
#define MIPT_BOX_MARGIN_CELLS 0.51f      /* permute decode: grid cells by which a plane's constant term may be off, and a little */
#endif
#ifndef MIPT_DIR_FLOOR
#define MIPT_DIR_FLOOR 1e-20f      /* smallest direction component the box test divides by */
#endif
#ifndef MIPT_ROTATE_PLANES
// A plane's 16 bits q are placed under the exponent of 128.0f with one v_perm_b32 (value 128 + q / 65536, exact), so
// that no integer has to be converted: t = value * I + C with I = 65536 * i and C = n - 128 * I.  C is rounded at the
// magnitude of 128 * I, which moves a plane by up to half a grid cell (2^-17 of the grid's extent): that half cell is
// part of the margin.  sn / sf are the permute selectors that pick the half of a node word the ray meets first / last.
// (-DMIPT_ROTATE_PLANES builds the decode this one replaced: s is 16 on an axis where the direction is negative and 0
// elsewhere, rotating a word by s puts the plane the ray meets first into the low half, and both halves are converted
// to float; 12 more vector instructions per node step, closest-hit kernel +1.7 %, any-hit kernel +2.5 %.)
struct BoxRay { float ix, iy, iz, cnx, cny, cnz, cfx, cfy, cfz; uint32_t snx, sny, snz, sfx, sfy, sfz; };
#define PERM_SEL_LO 0x070c0100u     /* bytes 0, 1 of the word, 0x00, byte 3 of the constant */
#define PERM_SEL_HI 0x070c0302u     /* bytes 2, 3 of the word, 0x00, byte 3 of the constant */
#define PERM_EXP    0x43000000u     /* 128.0f */
#else
struct BoxRay { float ix, iy, iz, cnx, cny, cnz, cfx, cfy, cfz; uint32_t sx, sy, sz; };
#endif

__device__ __forceinline__ BoxRay make_box_ray(const DevScene& S, F3 o, F3 d)
{
    BoxRay r;
    const float floor_ = MIPT_DIR_FLOOR;
    float dx = fabsf(d.x) < floor_ ? copysignf(floor_, d.x) : d.x;
    float dy = fabsf(d.y) < floor_ ? copysignf(floor_, d.y) : d.y;
    float dz = fabsf(d.z) < floor_ ? copysignf(floor_, d.z) : d.z;
    // one-ulp reciprocals (the margin counts them in)
    float ix = __builtin_amdgcn_rcpf(dx), iy = __builtin_amdgcn_rcpf(dy), iz = __builtin_amdgcn_rcpf(dz);
    const float ox = S.qmin[0] - o.x, oy = S.qmin[1] - o.y, oz = S.qmin[2] - o.z;
    float nx = ox * ix, ny = oy * iy, nz = oz * iz;
    r.ix = ix * S.qcell[0]; r.iy = iy * S.qcell[1]; r.iz = iz * S.qcell[2];
    // the ray's reach: no operand of the box test or of the triangle test is larger
    const float ex = 65535.0f * S.qcell[0], ey = 65535.0f * S.qcell[1], ez = 65535.0f * S.qcell[2];
    const float reach = fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)) + fmaxf(fmaxf(ex, ey), ez);
    float mx = reach * (MIPT_BOX_MARGIN_ULPS * 1.1920929e-7f), my = mx, mz = mx;
#ifndef MIPT_ROTATE_PLANES
    r.ix *= 65536.0f; r.iy *= 65536.0f; r.iz *= 65536.0f;
    nx = __builtin_fmaf(-128.0f, r.ix, nx); ny = __builtin_fmaf(-128.0f, r.iy, ny); nz = __builtin_fmaf(-128.0f, r.iz, nz);
    mx = __builtin_fmaf(MIPT_BOX_MARGIN_CELLS, S.qcell[0], mx); my = __builtin_fmaf(MIPT_BOX_MARGIN_CELLS, S.qcell[1], my); mz = __builtin_fmaf(MIPT_BOX_MARGIN_CELLS, S.qcell[2], mz);
    const bool gx = (int32_t)__float_as_uint(dx) < 0, gy = (int32_t)__float_as_uint(dy) < 0, gz = (int32_t)__float_as_uint(dz) < 0;
    r.snx = gx ? PERM_SEL_HI : PERM_SEL_LO; r.sfx = gx ? PERM_SEL_LO : PERM_SEL_HI;
    r.sny = gy ? PERM_SEL_HI : PERM_SEL_LO; r.sfy = gy ? PERM_SEL_LO : PERM_SEL_HI;
    r.snz = gz ? PERM_SEL_HI : PERM_SEL_LO; r.sfz = gz ? PERM_SEL_LO : PERM_SEL_HI;
#else
    r.sx = (__float_as_uint(dx) >> 31) << 4; r.sy = (__float_as_uint(dy) >> 31) << 4; r.sz = (__float_as_uint(dz) >> 31) << 4;
#endif
    const float tx = mx * fabsf(ix), ty = my * fabsf(iy), tz = mz * fabsf(iz);      // the margin as a distance along the ray
    r.cnx = nx - tx; r.cny = ny - ty; r.cnz = nz - tz;
    r.cfx = nx + tx; r.cfy = ny + ty; r.cfz = nz + tz;
    return r;
}

// Slab test against a box given as three node words; tnear, tfar = entry and exit distance.  Conservative by
// the ray's margin, never exact.
// CULL: the exit distance is also cut at tcull, the distance beyond which a box cannot matter (closest hit).
template <bool CULL>
__device__ __forceinline__ bool box_test(const BoxRay& r, uint32_t wx, uint32_t wy, uint32_t wz, float tcull, float& tnear, float& tfar)
{
#ifndef MIPT_ROTATE_PLANES
    auto plane = [](uint32_t w, uint32_t sel) { return __uint_as_float(__builtin_amdgcn_perm(PERM_EXP, w, sel)); };
    float x0 = __builtin_fmaf(plane(wx, r.snx), r.ix, r.cnx), x1 = __builtin_fmaf(plane(wx, r.sfx), r.ix, r.cfx);
    float y0 = __builtin_fmaf(plane(wy, r.sny), r.iy, r.cny), y1 = __builtin_fmaf(plane(wy, r.sfy), r.iy, r.cfy);
    float z0 = __builtin_fmaf(plane(wz, r.snz), r.iz, r.cnz), z1 = __builtin_fmaf(plane(wz, r.sfz), r.iz, r.cfz);
#else
    uint32_t px = __builtin_amdgcn_alignbit(wx, wx, r.sx), py = __builtin_amdgcn_alignbit(wy, wy, r.sy), pz = __builtin_amdgcn_alignbit(wz, wz, r.sz);
    float x0 = __builtin_fmaf((float)(px & 0xffffu), r.ix, r.cnx), x1 = __builtin_fmaf((float)(px >> 16), r.ix, r.cfx);
    float y0 = __builtin_fmaf((float)(py & 0xffffu), r.iy, r.cny), y1 = __builtin_fmaf((float)(py >> 16), r.iy, r.cfy);
    float z0 = __builtin_fmaf((float)(pz & 0xffffu), r.iz, r.cnz), z1 = __builtin_fmaf((float)(pz >> 16), r.iz, r.cfz);
#endif
    float tmin = fmaxf(fmaxf(x0, y0), fmaxf(z0, 0.0f));
    float zc = z1;
    // the minimum with tcull as the bare instruction: tcull is never a NaN, which the compiler cannot know (it
    // would first bring the value into canonical form, one more instruction per node step)
    if (CULL) asm("v_min_f32 %0, %1, %2" : "=v"(zc) : "v"(z1), "v"(tcull));
    float tmax = fminf(fminf(x1, y1), zc);
    tnear = tmin; tfar = tmax;
    return tmin <= tmax;
}

struct Hit { float t, u, v; int32_t tri; };

// ---------------------------------------------------------------------------------------
// wave64 primitives
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Appends one entry per lane with pred set to a queue whose tail is *counter: one ballot,
// one prefix popcount and a single atomic per wave.  Must be called by all lanes of the
// wave that are still running.  Returns the entry index (valid where pred is set).
__device__ __forceinline__ int wave_append(bool pred, int32_t* counter)
{
    unsigned long long mask = __ballot(pred);
    int total = __popcll(mask);
    int prefix = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    int base = 0;
    if (total) {
        int leader = __ffsll((long long)mask) - 1;
        if (lane_id() == leader) base = atomicAdd(counter, total);
        base = __shfl(base, leader, 64);
    }
    return base + prefix;
}

__device__ __forceinline__ unsigned long long wave_sum_u32(uint32_t v)
{
    unsigned long long s = v;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    return s;
}
