// Internal declarations shared by the HIP translation units of libmipt.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mipt.h"

// ---- device-side scene --------------------------------------------------------------
//
// BVH node slot (64 B = 4 x uint4, fetched by one lane with four 16-byte loads from one half of a
// 128-byte line): the boxes of up to MIPT_NODE_WIDTH children live in the parent, so that one
// fetch decides all descents.  Box coordinates are 16-bit cells of a uniform grid over the scene
// (coordinate = qmin + q * qcell, minima rounded down, maxima rounded up after padding): the
// box test only has to be generous, never tight.  One word per child and axis, min | max << 16:
//   words 3k .. 3k+2 = x, y, z of child k;  words 12 .. 15 = child codes
// An unused child has min > max (word 0x0000FFFF) and code MIPT_NO_NODE.
// Child codes are indices into one space that holds the node slots first and the leaf slots after
// them: a code below DevScene::leafBase (= the number of node slots) is an inner node, a code c at
// or above it is a leaf whose first triangle is leaf slot c - leafBase.  A leaf slot is one
// record of MIPT_TRI_F4 float4 in BVH order (triangles of one leaf are consecutive), so that the
// three loads of a lane fall into one cache line.
//   tris[s * MIPT_TRI_F4 + 0] = (v0.xyz, bits global triangle id)
//   tris[s * MIPT_TRI_F4 + 1] = (v1 - v0, bits 1 if last triangle of its leaf else 0)
//   tris[s * MIPT_TRI_F4 + 2] = (v2 - v0, 0)
#ifndef MIPT_TRI_F4
#define MIPT_TRI_F4 4      // 64-byte records: a record never straddles a 128-byte line
#endif
#ifndef MIPT_NODE_WIDTH
#define MIPT_NODE_WIDTH 4      // children per node slot.  The builder can also collapse to 8 (128-byte slots: words 3k .. 3k+2 = box of child k, words 24 .. 31 = child codes); the traversal kernel is written for 4 only
#endif
#define MIPT_NODE_U4 MIPT_NODE_WIDTH   // uint4 per node slot
#define MIPT_STACK_LDS_MAX 16   // stack rows (4 bytes per lane) kept in LDS (4 KiB per wave), the two bottom rows that mark the end of every lane's stack included; deeper rows, which one node step in two hundred needs, go to the spill slice
#define MIPT_SHADE_F4 8
// Node slots of the top of the array that every traversal workgroup copies into LDS (32 KiB).  The builder puts the
// nodes with the largest boxes, which the most rays visit, into these slots.
#define MIPT_TRACE_CACHE_SLOTS 512
// A queue in blocks: entries [b << MIPT_QBLOCK_SHIFT, (b << MIPT_QBLOCK_SHIFT) + blockCount[b]) of every block b are valid
#define MIPT_QBLOCK_SHIFT 10
#define MIPT_LEAF_BIT 0x80000000u   // marks leaves in the binary tree the builder works on; the finished tree uses leafBase
#define MIPT_NO_NODE  0xFFFFFFFFu

#define MIPT_TEXFMT_NONE 0
#define MIPT_TEXFMT_U8   1   // uint32 BGRA8 texels in DevScene::texels
#define MIPT_TEXFMT_F32  2   // 3 floats per texel in DevScene::texelsF

struct DevMaterial {
    uint64_t texOffset;      // first texel (U8: uint32 index, F32: float index)
    uint64_t nmlOffset;
    int32_t  texW, texH, texFmt;
    int32_t  nmlW, nmlH, nmlFmt;
    float    diffuse[3];
    int32_t  pad;
};

struct DevScene {
    const uint4*       nodes;
    const float4*      tris;       // MIPT_TRI_F4 float4 per leaf slot
    const float4*      shade;      // MIPT_SHADE_F4 float4 per global triangle, see k_make_shading
    const DevMaterial* mats;
    const uint32_t*    texels;
    const float*       texelsF;
    float              qmin[3];    // origin and cell size of the 16-bit grid the node boxes live on
    float              qcell[3];
    uint32_t           rootCode;   // MIPT_NO_NODE for an empty scene
    uint32_t           leafBase;   // child codes from here on are leaves (= number of node slots)
    uint32_t           slotMask;   // 2^b - 1 with the smallest b that holds every child code and the code after the last one (leafBase + numTris, "finished"); the bits above it carry the sort key in a stack entry
    int32_t            stackDepth; // rows of the per-lane traversal stack the tree can need, the two bottom rows that hold "finished" included
    int32_t            stackLds;   // of these, rows kept in LDS (at least 3); the rest live in stackSpill
    uint32_t*          stackSpill; // (stackDepth - stackLds) * 64 entries per resident wave, or null
    int32_t            numTris;
    int32_t            numMats;
    int64_t            numNodeSlots;
};

// render constants handed to the shading kernels by value
struct DevRender {
    float    toLocal[16];
    float    p0, p1;
    int32_t  clientW, clientH, div, W, H;
    float    sun[3];
    float    sunIntensity;
    int32_t  gi, useTextures, useNormals, normalizeNormalMap;
    float    eps;
    uint32_t seed;
    int32_t  numSamples;
    int32_t  shardRank, shardCount, tileSize, tilesX;
    int32_t  ownPixelsOnly;   // only the pixels of this shard's tiles are rendered
    int32_t  numOwn;          // with ownPixelsOnly: number of those pixels
    const int32_t* ownPix;    // with ownPixelsOnly: their indices in ascending order (primary ray i belongs to pixel ownPix[i]); else null: ray i = pixel i
    int32_t  countBoxTests;   // the closest-hit launches also count the child boxes of the nodes they visit
};

// ---- statistics slots (uint64 each) ---------------------------------------------------
enum {
    ST_CLOSEST_RAYS = 0, ST_SHADOW_RAYS, ST_CLOSEST_NODES, ST_CLOSEST_TRIS, ST_SHADOW_NODES, ST_SHADOW_TRIS,
    ST_SHADOW_CULLED, // shadow rays that were not traced because the direct term they guard is zero
    ST_CLOSEST_BOXES, // child boxes present in the nodes the closest-hit queries visited (counted on request only)
    ST_SHADOW_BOXES,  // the same for the any-hit queries
    ST_SHADE_HITS,    // hits shaded by k_shade_bounce
    ST_SHADE_NEXT,    // rays k_shade_bounce wrote to the queue of the next depth
    ST_PROF = 11,      // with -DMIPT_TRACE_PROFILE: wave-level step counts of the closest-hit kernel
    ST_COUNT = 32
};

// ---- BVH build result -------------------------------------------------------------------
struct BvhResult {
    uint4*   nodes = nullptr;     // numNodeSlots * 4
    float    qmin[3] = { 0, 0, 0 }, qcell[3] = { 1, 1, 1 };
    float4*  tris = nullptr;      // numLeafSlots * MIPT_TRI_F4
    int64_t  numNodeSlots = 0;
    int64_t  numLeafSlots = 0;
    int64_t  numInner = 0;
    int64_t  numLeaves = 0;
    int32_t  maxDepth = 0;
    uint32_t rootCode = MIPT_NO_NODE;
    double   buildMs = 0;
    double   sahCost = 0;
};

// Builds the BVH over numTris triangles whose corners are in d_attr (6 float4 per triangle).
// Returns MIPT_OK or an error code; on error nothing is left allocated.
int mipt_build_bvh(const float4* d_attr, int64_t numTris, const mipt_build_options* opt, hipStream_t stream,
                   BvhResult* out);
void mipt_free_bvh(BvhResult* r);

// ---- kernel launchers (pt_kernels.hip) --------------------------------------------------
struct RayQueue {
    float4* o;        // origin.xyz, bits slot
    float4* d;        // direction.xyz, bits level (shadow rays) / 0
};

void launch_raygen(const DevRender& R, RayQueue q, hipStream_t s);
// closest-hit over q[0 .. *d_count): hit[i] = (t, ua, va, bits triangle id or -1)
void launch_make_shading(const float4* attr, const int32_t* triMat, const DevMaterial* mats, int64_t n, float4* shade, hipStream_t s);
void launch_closest(const DevScene& S, RayQueue q, const int32_t* d_count, int32_t maxCount, int32_t* d_head,
                    float4* hit, float eps, unsigned long long* d_stats, int numCUs, hipStream_t s, bool tagInHit = false, bool countBoxes = false);
// any-hit over q[0 .. *d_count): clears the direct term of the slot the ray belongs to when occluded
void launch_shadow(const DevScene& S, RayQueue q, const int32_t* d_count, const int32_t* d_blockCount, int32_t maxCount, int32_t* d_head,
                   float4* pixState0, float4* pathState, int64_t slotsPerLevel, float eps,
                   unsigned long long* d_stats, int numCUs, hipStream_t s, bool countBoxes = false);
// any-hit for caller-supplied rays: out[i] = 1 if occluded
void launch_anyhit_flags(const DevScene& S, RayQueue q, const int32_t* d_count, int32_t maxCount, int32_t* d_head,
                         uint8_t* out, float eps, unsigned long long* d_stats, int numCUs, hipStream_t s);
// shading of the primary hits (level 0)
void launch_shade_primary(const DevScene& S, const DevRender& R, const float4* hit, float4* pixState,
                          int32_t* triIdImage, RayQueue shadowQ, int32_t* d_shadowCount, int32_t* actPix,
                          int32_t* d_actCount, unsigned long long* d_stats, hipStream_t s);
// first-bounce rays of samples [n0, n0+S) of every active pixel
void launch_gen_first_bounce(const DevRender& R, const float4* pixState, const int32_t* actPix, int32_t nAct,
                             int32_t n0, int32_t S, RayQueue q, hipStream_t s);
// shading of bounce `level` (1..3): path state, shadow ray, next ray
void launch_shade_bounce(const DevScene& S, const DevRender& R, RayQueue q, const float4* hit,
                         const int32_t* d_count, int32_t maxCount, int32_t level, const int32_t* actPix,
                         int32_t nAct, int32_t n0, float4* pathState, int64_t slotsPerLevel, RayQueue shadowQ,
                         int32_t* d_shadowBlockCount, RayQueue nextQ, int32_t* d_nextCount,
                         unsigned long long* d_stats, hipStream_t s);
// backward resolve of one pass and accumulation into the per-active-pixel sums
void launch_resolve(const DevRender& R, const float4* pathState, int64_t slotsPerLevel, const int32_t* actPix,
                    int32_t nAct, int32_t n0, int32_t S, float4* pixAccum, hipStream_t s);
void launch_scatter_accum(const float4* pixAccum, const int32_t* actPix, int32_t nAct, float* d_accum, hipStream_t s);
void launch_finalize(const DevRender& R, const float4* pixState, const float* d_accum, float* d_out, hipStream_t s);
void launch_set_i32(int32_t* d_ptr, int32_t value, hipStream_t s);
void launch_pack_rays(const float* d_rays6, int64_t n, RayQueue q, hipStream_t s);
void launch_unpack_hits(const float4* hit, int64_t n, int32_t* d_outTri, float* d_outTuv, hipStream_t s);
