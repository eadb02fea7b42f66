// GPU construction of the bounding volume hierarchy that replaces the reference's octree
// (CMesh::ComputeSubmeshes, Mesh.cpp:88-166).
//
// Top-down binned surface-area-heuristic build, one tree level per round, every round made of
// data-parallel kernels over all triangle positions:
//   k_bin        each triangle of a node being split adds its box to one of `bins` bins per axis
//                (integer atomics on order-preserving float keys)
//   k_split      one thread per node sweeps the bins, picks the cheapest (axis, plane), decides
//                leaf or split and allocates the two children
//   k_partition  each triangle moves to its side of the plane; the children's centroid bounds
//                are gathered on the way
// The finished tree is written in the traversal layout described in pt_internal.h.
#include "pt_internal.h"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <utility>
#include <vector>

#define MAX_BINS 32
#define MAX_BUILD_DEPTH 60

enum { NODE_LEAF = 0, NODE_ACTIVE = 1, NODE_INNER = 2 };

struct BNode {
    int32_t bminI[3], bmaxI[3];   // box of the node, order-preserving integer keys
    int32_t cminI[3], cmaxI[3];   // bounds of the triangle centroids
    int32_t start, count;         // triangle positions [start, start + count)
    int32_t left;                 // children are left and left + 1
    int32_t depth;
    int32_t state;
    int32_t axis;                 // split axis, -1 = split by position (no plane separates the centroids)
    int32_t splitBin;
    int32_t nLeft;
    int32_t fillL, fillR;         // slots handed out during partition
    int32_t activeSlot;
    int32_t pad;
};

struct Bin { int32_t minI[3], maxI[3]; int32_t count; int32_t pad; };

struct BuildCounters {
    int32_t nodeCount;
    int32_t nextActive;
    int32_t leafCount;
    int32_t maxDepth;
    double  sah;                  // sum of area * cost, divided by the root area on the host
};

// float <-> integer key whose signed order equals the float order
__device__ __forceinline__ int32_t f2key(float f) { int32_t i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float key2f(int32_t i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__device__ __forceinline__ float box_area(const float* mn, const float* mx)
{
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}

// ---------------------------------------------------------------------------------------
// Atomics to one address are served at a fixed, low rate (tens per microsecond), so every kernel
// below first combines what the 64 lanes of a wave (or the 256 threads of a block) have for the
// same target and sends one atomic for the group.
__device__ __forceinline__ int32_t wave_min_i32(int32_t v)
{
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int32_t wave_max_i32(int32_t v)
{
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}
// true if every active lane of the wave holds the same value
__device__ __forceinline__ bool wave_uniform_i32(int32_t v)
{
    return __ballot(v != __builtin_amdgcn_readfirstlane(v)) == 0;
}

__global__ void __launch_bounds__(256) k_prim_bounds(const float4* __restrict__ attr, int32_t n, float4* primMin,
                                                    float4* primMax, int32_t* primIdx, int32_t* posNode, BNode* root)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int32_t kmin[3] = { INT_MAX, INT_MAX, INT_MAX }, kmax[3] = { INT_MIN, INT_MIN, INT_MIN };
    int32_t cmin[3] = { INT_MAX, INT_MAX, INT_MAX }, cmax[3] = { INT_MIN, INT_MIN, INT_MIN };
    if (i < n) {
        const float4* a = attr + (size_t)i * 6;
        float4 p0 = a[0], p1 = a[2], p2 = a[4];
        float mnx = fminf(p0.x, fminf(p1.x, p2.x)), mny = fminf(p0.y, fminf(p1.y, p2.y)), mnz = fminf(p0.z, fminf(p1.z, p2.z));
        float mxx = fmaxf(p0.x, fmaxf(p1.x, p2.x)), mxy = fmaxf(p0.y, fmaxf(p1.y, p2.y)), mxz = fmaxf(p0.z, fmaxf(p1.z, p2.z));
        primMin[i] = make_float4(mnx, mny, mnz, 0);
        primMax[i] = make_float4(mxx, mxy, mxz, 0);
        primIdx[i] = i;
        posNode[i] = 0;
        float cx = 0.5f * (mnx + mxx), cy = 0.5f * (mny + mxy), cz = 0.5f * (mnz + mxz);
        kmin[0] = f2key(mnx); kmin[1] = f2key(mny); kmin[2] = f2key(mnz);
        kmax[0] = f2key(mxx); kmax[1] = f2key(mxy); kmax[2] = f2key(mxz);
        cmin[0] = cmax[0] = f2key(cx); cmin[1] = cmax[1] = f2key(cy); cmin[2] = cmax[2] = f2key(cz);
    }
    for (int a = 0; a < 3; a++) {
        kmin[a] = wave_min_i32(kmin[a]); kmax[a] = wave_max_i32(kmax[a]);
        cmin[a] = wave_min_i32(cmin[a]); cmax[a] = wave_max_i32(cmax[a]);
    }
    if ((threadIdx.x & 63) == 0 && kmin[0] != INT_MAX) {
        for (int a = 0; a < 3; a++) {
            atomicMin(&root->bminI[a], kmin[a]); atomicMax(&root->bmaxI[a], kmax[a]);
            atomicMin(&root->cminI[a], cmin[a]); atomicMax(&root->cmaxI[a], cmax[a]);
        }
    }
}

__global__ void k_init_root(BNode* nodes, BuildCounters* ctr, int32_t* active, int32_t n)
{
    BNode r = {};
    for (int a = 0; a < 3; a++) { r.bminI[a] = INT_MAX; r.bmaxI[a] = INT_MIN; r.cminI[a] = INT_MAX; r.cmaxI[a] = INT_MIN; }
    r.start = 0; r.count = n; r.left = -1; r.depth = 0;
    r.state = (n > 1) ? NODE_ACTIVE : NODE_LEAF;
    r.axis = -1; r.activeSlot = 0;
    nodes[0] = r;
    ctr->nodeCount = 1; ctr->nextActive = 0; ctr->leafCount = (n > 1) ? 0 : 1; ctr->maxDepth = 0; ctr->sah = 0.0;
    active[0] = 0;
}

__global__ void __launch_bounds__(256) k_clear_bins(Bin* bins, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Bin b;
    for (int a = 0; a < 3; a++) { b.minI[a] = INT_MAX; b.maxI[a] = INT_MIN; }
    b.count = 0; b.pad = 0;
    bins[i] = b;
}

__device__ __forceinline__ int bin_of(float c, float cmin, float cmax, int nb)
{
    float ext = cmax - cmin;
    if (!(ext > 0.0f)) return 0;
    int b = (int)((c - cmin) * ((float)nb / ext));
    return b < 0 ? 0 : (b >= nb ? nb - 1 : b);
}

// Positions of one node are contiguous, so a block that walks BIN_CHUNK consecutive positions sees
// few nodes.  Triangles of a large node are binned into a copy of that node's bins in LDS, which
// is added to the global bins when the block moves on to another node; triangles of small nodes
// (little contention) go to the global bins directly.
#define BIN_CHUNK 2048
#define BIN_LDS_MIN_COUNT 512
__global__ void __launch_bounds__(256) k_bin(const int32_t* __restrict__ primIdx, const int32_t* __restrict__ posNode,
                                            const BNode* __restrict__ nodes, const float4* __restrict__ primMin,
                                            const float4* __restrict__ primMax, Bin* bins, int32_t n, int nb)
{
    __shared__ int32_t lds[3 * MAX_BINS * 7];     // per bin: min xyz, max xyz, count
    __shared__ int32_t ldsNode;                   // node whose bins are in LDS, -1 = none
    const int nEntries = 3 * nb;
    const int t = threadIdx.x;
    if (t == 0) ldsNode = -1;
    auto clear_lds = [&]() {
        for (int e = t; e < nEntries; e += 256) {
            int32_t* L = lds + e * 7;
            L[0] = L[1] = L[2] = INT_MAX; L[3] = L[4] = L[5] = INT_MIN; L[6] = 0;
        }
    };
    auto flush_lds = [&](int node) {
        Bin* base = bins + (size_t)nodes[node].activeSlot * 3 * nb;
        for (int e = t; e < nEntries; e += 256) {
            const int32_t* L = lds + e * 7;
            if (L[6] == 0) continue;
            Bin* B = base + e;
            atomicMin(&B->minI[0], L[0]); atomicMin(&B->minI[1], L[1]); atomicMin(&B->minI[2], L[2]);
            atomicMax(&B->maxI[0], L[3]); atomicMax(&B->maxI[1], L[4]); atomicMax(&B->maxI[2], L[5]);
            atomicAdd(&B->count, L[6]);
        }
    };
    __syncthreads();
    const int64_t first = (int64_t)blockIdx.x * BIN_CHUNK;
    for (int step = 0; step < BIN_CHUNK / 256; step++) {
        const int64_t p0 = first + (int64_t)step * 256;
        if (p0 >= n) break;
        // the node of the step's first position decides which node may use LDS in this step
        int want = posNode[p0];
        const BNode& wn = nodes[want];
        if (!(wn.state == NODE_ACTIVE && wn.count >= BIN_LDS_MIN_COUNT)) want = -1;
        int have = ldsNode;
        if (have != want) {                       // uniform over the block
            if (have >= 0) flush_lds(have);
            __syncthreads();
            if (want >= 0) clear_lds();
            if (t == 0) ldsNode = want;
            __syncthreads();
        }
        const int64_t p = p0 + t;
        if (p < n) {
            const int id = posNode[p];
            const BNode& nd = nodes[id];
            if (nd.state == NODE_ACTIVE) {
                int prim = primIdx[p];
                float4 mn = primMin[prim], mx = primMax[prim];
                const float c[3] = { 0.5f * (mn.x + mx.x), 0.5f * (mn.y + mx.y), 0.5f * (mn.z + mx.z) };
                int kmin[3] = { f2key(mn.x), f2key(mn.y), f2key(mn.z) }, kmax[3] = { f2key(mx.x), f2key(mx.y), f2key(mx.z) };
                if (id == want) {
                    for (int a = 0; a < 3; a++) {
                        int b = bin_of(c[a], key2f(nd.cminI[a]), key2f(nd.cmaxI[a]), nb);
                        int32_t* L = lds + (a * nb + b) * 7;
                        atomicMin(&L[0], kmin[0]); atomicMin(&L[1], kmin[1]); atomicMin(&L[2], kmin[2]);
                        atomicMax(&L[3], kmax[0]); atomicMax(&L[4], kmax[1]); atomicMax(&L[5], kmax[2]);
                        atomicAdd(&L[6], 1);
                    }
                } else {
                    Bin* base = bins + (size_t)nd.activeSlot * 3 * nb;
                    for (int a = 0; a < 3; a++) {
                        int b = bin_of(c[a], key2f(nd.cminI[a]), key2f(nd.cmaxI[a]), nb);
                        Bin* B = base + a * nb + b;
                        atomicMin(&B->minI[0], kmin[0]); atomicMin(&B->minI[1], kmin[1]); atomicMin(&B->minI[2], kmin[2]);
                        atomicMax(&B->maxI[0], kmax[0]); atomicMax(&B->maxI[1], kmax[1]); atomicMax(&B->maxI[2], kmax[2]);
                        atomicAdd(&B->count, 1);
                    }
                }
            }
        }
        __syncthreads();
    }
    int have = ldsNode;
    if (have >= 0) flush_lds(have);
}

__device__ __forceinline__ void make_child(BNode* nodes, int id, const BNode& parent, int start, int count,
                                           const int32_t* bminI, const int32_t* bmaxI, int maxLeaf, BuildCounters* ctr,
                                           int32_t* nextActive)
{
    BNode c = {};
    for (int a = 0; a < 3; a++) { c.bminI[a] = bminI[a]; c.bmaxI[a] = bmaxI[a]; c.cminI[a] = INT_MAX; c.cmaxI[a] = INT_MIN; }
    c.start = start; c.count = count; c.left = -1; c.depth = parent.depth + 1; c.axis = -1;
    bool leaf = (count <= 1) || (c.depth >= MAX_BUILD_DEPTH);
    c.state = leaf ? NODE_LEAF : NODE_ACTIVE;
    if (leaf) {
        atomicAdd(&ctr->leafCount, 1);
        atomicMax(&ctr->maxDepth, c.depth);
    } else {
        int slot = atomicAdd(&ctr->nextActive, 1);
        c.activeSlot = slot;
        nextActive[slot] = id;
    }
    nodes[id] = c;
}

__global__ void __launch_bounds__(64) k_split(BNode* nodes, const int32_t* __restrict__ active, int32_t numActive,
                                             const Bin* __restrict__ bins, int nb, int maxLeaf, float travCost,
                                             BuildCounters* ctr, int32_t* nextActive)
{
    int ai = blockIdx.x * blockDim.x + threadIdx.x;
    if (ai >= numActive) return;
    int id = active[ai];
    BNode nd = nodes[id];
    const Bin* base = bins + (size_t)ai * 3 * nb;
    float pmn[3] = { key2f(nd.bminI[0]), key2f(nd.bminI[1]), key2f(nd.bminI[2]) };
    float pmx[3] = { key2f(nd.bmaxI[0]), key2f(nd.bmaxI[1]), key2f(nd.bmaxI[2]) };
    float parentArea = box_area(pmn, pmx);

    float bestCost = FLT_MAX; int bestAxis = -1, bestPlane = -1, bestNL = 0;
    for (int a = 0; a < 3; a++) {
        // right-to-left sweep: area and count of bins [i, nb)
        float rightArea[MAX_BINS]; int rightCount[MAX_BINS];
        float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        int cnt = 0;
        for (int i = nb - 1; i >= 1; i--) {
            const Bin& B = base[a * nb + i];
            if (B.count > 0) {
                for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], key2f(B.minI[k])); mx[k] = fmaxf(mx[k], key2f(B.maxI[k])); }
                cnt += B.count;
            }
            rightArea[i] = cnt ? box_area(mn, mx) : 0.0f;
            rightCount[i] = cnt;
        }
        for (int k = 0; k < 3; k++) { mn[k] = FLT_MAX; mx[k] = -FLT_MAX; }
        cnt = 0;
        for (int i = 1; i < nb; i++) {          // plane i separates bins [0, i) from [i, nb)
            const Bin& B = base[a * nb + i - 1];
            if (B.count > 0) {
                for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], key2f(B.minI[k])); mx[k] = fmaxf(mx[k], key2f(B.maxI[k])); }
                cnt += B.count;
            }
            if (cnt == 0 || rightCount[i] == 0) continue;
            float cost = box_area(mn, mx) * (float)cnt + rightArea[i] * (float)rightCount[i];
            if (cost < bestCost) { bestCost = cost; bestAxis = a; bestPlane = i; bestNL = cnt; }
        }
    }
    float leafCost = (float)nd.count;
    float splitCost = (bestAxis >= 0 && parentArea > 0.0f) ? travCost + bestCost / parentArea : FLT_MAX;
    bool makeLeaf;
    if (bestAxis < 0) makeLeaf = nd.count <= maxLeaf;              // no separating plane: split by position if too big
    else makeLeaf = (nd.count <= maxLeaf) && (leafCost <= splitCost);
    if (makeLeaf) {
        nd.state = NODE_LEAF;
        nodes[id] = nd;
        atomicAdd(&ctr->leafCount, 1);
        atomicMax(&ctr->maxDepth, nd.depth);
        return;
    }
    int left = atomicAdd(&ctr->nodeCount, 2);
    nd.state = NODE_INNER; nd.left = left; nd.fillL = 0; nd.fillR = 0;
    int32_t lmin[3] = { INT_MAX, INT_MAX, INT_MAX }, lmax[3] = { INT_MIN, INT_MIN, INT_MIN };
    int32_t rmin[3] = { INT_MAX, INT_MAX, INT_MAX }, rmax[3] = { INT_MIN, INT_MIN, INT_MIN };
    if (bestAxis >= 0) {
        nd.axis = bestAxis; nd.splitBin = bestPlane; nd.nLeft = bestNL;
        for (int i = 0; i < nb; i++) {
            const Bin& B = base[bestAxis * nb + i];
            if (B.count == 0) continue;
            int32_t* tmn = (i < bestPlane) ? lmin : rmin; int32_t* tmx = (i < bestPlane) ? lmax : rmax;
            for (int k = 0; k < 3; k++) { tmn[k] = min(tmn[k], B.minI[k]); tmx[k] = max(tmx[k], B.maxI[k]); }
        }
    } else {
        nd.axis = -1; nd.splitBin = 0; nd.nLeft = nd.count / 2;   // children's boxes are gathered by k_partition
    }
    nodes[id] = nd;
    make_child(nodes, left, nd, nd.start, nd.nLeft, lmin, lmax, maxLeaf, ctr, nextActive);
    make_child(nodes, left + 1, nd, nd.start + nd.nLeft, nd.count - nd.nLeft, rmin, rmax, maxLeaf, ctr, nextActive);
}

__global__ void __launch_bounds__(256) k_partition(const int32_t* __restrict__ primIdx, const int32_t* __restrict__ posNode,
                                                  int32_t* primIdxOut, int32_t* posNodeOut, BNode* nodes,
                                                  const float4* __restrict__ primMin, const float4* __restrict__ primMax,
                                                  int32_t n, int nb)
{
    // every lane stays to the end: the wave-wide reductions below need all 64 lanes running
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = p < n;
    int id = valid ? posNode[p] : 0;
    int prim = valid ? primIdx[p] : 0;
    bool moves = valid && nodes[id].state == NODE_INNER;
    if (valid && !moves) {               // finished leaf: stays where it is
        primIdxOut[p] = prim; posNodeOut[p] = id;
    }
    const unsigned long long movers = __ballot(moves);
    if (movers == 0) return;             // uniform
    BNode& nd = nodes[moves ? id : 0];
    bool goLeft = false;
    int32_t ck[3] = { 0, 0, 0 }, bmn[3] = { 0, 0, 0 }, bmx[3] = { 0, 0, 0 };
    if (moves) {
        float4 mn = primMin[prim], mx = primMax[prim];
        const float c[3] = { 0.5f * (mn.x + mx.x), 0.5f * (mn.y + mx.y), 0.5f * (mn.z + mx.z) };
        if (nd.axis >= 0) goLeft = bin_of(c[nd.axis], key2f(nd.cminI[nd.axis]), key2f(nd.cmaxI[nd.axis]), nb) < nd.splitBin;
        else goLeft = (p - nd.start) < nd.nLeft;
        ck[0] = f2key(c[0]); ck[1] = f2key(c[1]); ck[2] = f2key(c[2]);
        bmn[0] = f2key(mn.x); bmn[1] = f2key(mn.y); bmn[2] = f2key(mn.z);
        bmx[0] = f2key(mx.x); bmx[1] = f2key(mx.y); bmx[2] = f2key(mx.z);
    }
    // When all moving lanes share ONE node (the usual case near the top of the tree, where a node
    // spans many waves) the wave takes its places on either side with one atomic per side and sends
    // one set of bounds per child.
    const int leader = __ffsll((long long)movers) - 1;
    const int leaderId = __shfl(id, leader);
    const bool sameNode = __ballot(moves && id != leaderId) == 0;
    if (sameNode) {
        const unsigned long long lmask = __ballot(moves && goLeft), rmask = movers & ~lmask;
        int baseL = 0, baseR = 0;
        if (lane == leader) {
            if (lmask) baseL = atomicAdd(&nd.fillL, __popcll(lmask));
            if (rmask) baseR = atomicAdd(&nd.fillR, __popcll(rmask));
        }
        baseL = __shfl(baseL, leader); baseR = __shfl(baseR, leader);
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        if (moves) {
            int dst = goLeft ? nd.start + baseL + __popcll(lmask & below) : nd.start + nd.nLeft + baseR + __popcll(rmask & below);
            primIdxOut[dst] = prim;
            posNodeOut[dst] = nd.left + (goLeft ? 0 : 1);
        }
        const int axisL = __shfl(moves ? nd.axis : 0, leader), leftL = __shfl(moves ? nd.left : 0, leader);
        const bool gather = axisL < 0;
        for (int side = 0; side < 2; side++) {
            if (((side == 0) ? lmask : rmask) == 0) continue;          // uniform
            const bool mine = moves && ((side == 0) == goLeft);
            int32_t vmin[3], vmax[3], wmin[3] = { 0, 0, 0 }, wmax[3] = { 0, 0, 0 };
            for (int a = 0; a < 3; a++) {
                vmin[a] = wave_min_i32(mine ? ck[a] : INT_MAX); vmax[a] = wave_max_i32(mine ? ck[a] : INT_MIN);
                if (gather) { wmin[a] = wave_min_i32(mine ? bmn[a] : INT_MAX); wmax[a] = wave_max_i32(mine ? bmx[a] : INT_MIN); }
            }
            if (lane == leader) {
                BNode& ch = nodes[leftL + side];
                for (int a = 0; a < 3; a++) {
                    atomicMin(&ch.cminI[a], vmin[a]); atomicMax(&ch.cmaxI[a], vmax[a]);
                    if (gather) { atomicMin(&ch.bminI[a], wmin[a]); atomicMax(&ch.bmaxI[a], wmax[a]); }
                }
            }
        }
        return;
    }
    if (!moves) return;
    int child = nd.left + (goLeft ? 0 : 1);
    int dst = goLeft ? nd.start + atomicAdd(&nd.fillL, 1) : nd.start + nd.nLeft + atomicAdd(&nd.fillR, 1);
    primIdxOut[dst] = prim;
    posNodeOut[dst] = child;
    BNode& ch = nodes[child];
    atomicMin(&ch.cminI[0], ck[0]); atomicMin(&ch.cminI[1], ck[1]); atomicMin(&ch.cminI[2], ck[2]);
    atomicMax(&ch.cmaxI[0], ck[0]); atomicMax(&ch.cmaxI[1], ck[1]); atomicMax(&ch.cmaxI[2], ck[2]);
    if (nd.axis < 0) {
        atomicMin(&ch.bminI[0], bmn[0]); atomicMin(&ch.bminI[1], bmn[1]); atomicMin(&ch.bminI[2], bmn[2]);
        atomicMax(&ch.bmaxI[0], bmx[0]); atomicMax(&ch.bmaxI[1], bmx[1]); atomicMax(&ch.bmaxI[2], bmx[2]);
    }
}

// ---------------------------------------------------------------------------------------
// traversal layout
__device__ __forceinline__ uint32_t child_code(const BNode& c, int id)
{
    return c.state == NODE_LEAF ? (MIPT_LEAF_BIT | (uint32_t)c.start) : (uint32_t)id;
}

struct QuantGrid { float qmin[3], invCell[3]; };

__device__ __forceinline__ uint32_t quant_lo(float v, float qmin, float invCell)
{
    float q = floorf((v - qmin) * invCell);
    return (uint32_t)fminf(fmaxf(q, 0.0f), 65535.0f);
}
__device__ __forceinline__ uint32_t quant_hi(float v, float qmin, float invCell)
{
    float q = ceilf((v - qmin) * invCell);
    return (uint32_t)fminf(fmaxf(q, 0.0f), 65535.0f);
}

__global__ void __launch_bounds__(256) k_emit_nodes(const BNode* __restrict__ nodes, int32_t numNodes, float pad, QuantGrid G,
                                                   uint4* out, BuildCounters* ctr)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numNodes) return;
    const BNode& nd = nodes[i];
    float mn[3] = { key2f(nd.bminI[0]), key2f(nd.bminI[1]), key2f(nd.bminI[2]) };
    float mx[3] = { key2f(nd.bmaxI[0]), key2f(nd.bmaxI[1]), key2f(nd.bmaxI[2]) };
    double area = (double)box_area(mn, mx);
    uint4* o = out + (size_t)i * 2;
    if (nd.state != NODE_INNER) {
        atomicAdd(&ctr->sah, area * (double)nd.count);
        o[0] = o[1] = make_uint4(0, 0, 0, 0);
        return;
    }
    atomicAdd(&ctr->sah, area);
    const BNode& l = nodes[nd.left];
    const BNode& r = nodes[nd.left + 1];
    // boxes grow by `pad` on every side and are then rounded outwards to the 16-bit grid: the box
    // test is allowed to be generous, never tight
    uint32_t q[12];
    for (int a = 0; a < 3; a++) {
        q[a] = quant_lo(key2f(l.bminI[a]) - pad, G.qmin[a], G.invCell[a]);
        q[3 + a] = quant_hi(key2f(l.bmaxI[a]) + pad, G.qmin[a], G.invCell[a]);
        q[6 + a] = quant_lo(key2f(r.bminI[a]) - pad, G.qmin[a], G.invCell[a]);
        q[9 + a] = quant_hi(key2f(r.bmaxI[a]) + pad, G.qmin[a], G.invCell[a]);
    }
    // one word per child and axis: min | max << 16
    o[0] = make_uint4(q[0] | (q[3] << 16), q[1] | (q[4] << 16), q[2] | (q[5] << 16), q[6] | (q[9] << 16));
    o[1] = make_uint4(q[7] | (q[10] << 16), q[8] | (q[11] << 16), child_code(l, nd.left), child_code(r, nd.left + 1));
}

__global__ void __launch_bounds__(256) k_emit_tris(const float4* __restrict__ attr, const int32_t* __restrict__ primIdx,
                                                  const int32_t* __restrict__ posNode, const BNode* __restrict__ nodes, int32_t n,
                                                  float4* tris)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int prim = primIdx[p];
    const BNode& nd = nodes[posNode[p]];
    const float4* a = attr + (size_t)prim * 6;
    float4 p0 = a[0], p1 = a[2], p2 = a[4];
    uint32_t last = (p == nd.start + nd.count - 1) ? 1u : 0u;
    float4* t = tris + (size_t)p * MIPT_TRI_F4;
    t[0] = make_float4(p0.x, p0.y, p0.z, __uint_as_float((uint32_t)prim));
    t[1] = make_float4(p1.x - p0.x, p1.y - p0.y, p1.z - p0.z, __uint_as_float(last));
    t[2] = make_float4(p2.x - p0.x, p2.y - p0.y, p2.z - p0.z, 0.0f);
    for (int k = 3; k < MIPT_TRI_F4; k++) t[k] = make_float4(0, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "bvh build: %s failed: %s\n", #x, hipGetErrorString(e_)); rc = MIPT_E_DEVICE; goto done; } } while (0)

// ---------------------------------------------------------------------------------------
// Wide nodes.  The binary tree is collapsed into nodes with up to four children (64 bytes: twelve
// box words and four child codes, see pt_internal.h): starting from the two children of a binary
// node, the inner child with the largest surface is replaced by its own two children until four
// are reached.  A traversal step then costs one cache line per lane for two binary levels.  Two
// wide nodes share a 128-byte line: a node and its largest inner child.
//
// One launch per level of the binary tree, top down (the node ids of a level are contiguous, the
// host knows the ranges from the build rounds): a binary node that has been given a slot becomes a
// wide node, absorbs up to two descendants and hands slots to its inner children, which lie on
// deeper levels and are reached by later launches.
struct WChild { uint32_t code; uint32_t w[3]; };     // w[a] = min | max << 16 on axis a
struct CollapseCounters { int32_t lines; int32_t maxDepth; int32_t numWide; int32_t pad; };

__device__ __forceinline__ float wchild_area(const WChild& c, float cx, float cy, float cz)
{
    float ex = ((float)(c.w[0] >> 16) - (float)(c.w[0] & 0xffffu)) * cx;
    float ey = ((float)(c.w[1] >> 16) - (float)(c.w[1] & 0xffffu)) * cy;
    float ez = ((float)(c.w[2] >> 16) - (float)(c.w[2] & 0xffffu)) * cz;
    return ex * ey + ey * ez + ez * ex;
}

__global__ void __launch_bounds__(256) k_collapse_level(const uint4* __restrict__ bin2, int32_t first, int32_t last,
                                                       uint32_t* slotOf, int32_t* depthOf, float cx, float cy, float cz,
                                                       uint4* wide, CollapseCounters* cc)
{
    const int x = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= last) return;
    const uint32_t slot = slotOf[x];
    if (slot == MIPT_NO_NODE) return;                 // a leaf, or absorbed by its parent
    WChild k[MIPT_NODE_WIDTH];
    int n = 2;
    auto load2 = [&](int node, WChild& c0, WChild& c1) {
        uint4 u0 = bin2[(size_t)node * 2], u1 = bin2[(size_t)node * 2 + 1];
        c0.w[0] = u0.x; c0.w[1] = u0.y; c0.w[2] = u0.z; c0.code = u1.z;
        c1.w[0] = u0.w; c1.w[1] = u1.x; c1.w[2] = u1.y; c1.code = u1.w;
    };
    load2(x, k[0], k[1]);
    while (n < MIPT_NODE_WIDTH) {
        int best = -1; float bestArea = -1.0f;
        for (int i = 0; i < n; i++) {
            if (k[i].code & MIPT_LEAF_BIT) continue;
            float ar = wchild_area(k[i], cx, cy, cz);
            if (ar > bestArea) { bestArea = ar; best = i; }
        }
        if (best < 0) break;
        WChild c0, c1;
        load2((int)k[best].code, c0, c1);
        // constant indices only, so that k[] stays in registers
        for (int i = 0; i < MIPT_NODE_WIDTH; i++) if (i == best) k[i] = c0;
        for (int i = 0; i < MIPT_NODE_WIDTH; i++) if (i == n) k[i] = c1;
        n++;
    }
    const int depth = depthOf[x];
    int mate = -1;
#if MIPT_NODE_WIDTH == 4
    if ((slot & 1u) == 0) {                            // first slot of a line: the largest inner child shares the line
        float mateArea = -1.0f;
        for (int i = 0; i < n; i++) {
            if (k[i].code & MIPT_LEAF_BIT) continue;
            float ar = wchild_area(k[i], cx, cy, cz);
            if (ar > mateArea) { mateArea = ar; mate = i; }
        }
    }
#endif
    uint32_t words[4 * MIPT_NODE_WIDTH];
    for (int i = 0; i < MIPT_NODE_WIDTH; i++) {
        if (i < n) {
            words[3 * i] = k[i].w[0]; words[3 * i + 1] = k[i].w[1]; words[3 * i + 2] = k[i].w[2];
            uint32_t code = k[i].code;
            if (!(code & MIPT_LEAF_BIT)) {
#if MIPT_NODE_WIDTH == 4
                uint32_t cs = (i == mate) ? slot + 1 : 2u * (uint32_t)atomicAdd(&cc->lines, 1);
#else
                uint32_t cs = (uint32_t)atomicAdd(&cc->lines, 1);        // a 128-byte slot is a line of its own
#endif
                slotOf[code] = cs;
                depthOf[code] = depth + 1;
                code = cs;
            }
            words[3 * MIPT_NODE_WIDTH + i] = code;
        } else {
            words[3 * i] = words[3 * i + 1] = words[3 * i + 2] = 0x0000FFFFu;    // min above max: no ray enters an unused child
            words[3 * MIPT_NODE_WIDTH + i] = MIPT_NO_NODE;
        }
    }
    uint4* o = wide + (size_t)slot * MIPT_NODE_U4;
    for (int i = 0; i < MIPT_NODE_U4; i++) o[i] = make_uint4(words[4 * i], words[4 * i + 1], words[4 * i + 2], words[4 * i + 3]);
    atomicMax(&cc->maxDepth, depth + 1);
    atomicAdd(&cc->numWide, 1);
}

__global__ void k_collapse_init(uint32_t* slotOf, int32_t* depthOf, CollapseCounters* cc)
{
    slotOf[0] = 0; depthOf[0] = 0;
    cc->lines = 1; cc->maxDepth = 0; cc->numWide = 0; cc->pad = 0;
}

// levelEnd[d] = number of binary nodes of depth <= d
static int collapse_to_wide(BvhResult* out, const std::vector<int32_t>& levelEnd, hipStream_t stream)
{
    const int64_t oldSlots = out->numNodeSlots;
    if (oldSlots <= 0 || (out->rootCode & MIPT_LEAF_BIT)) { out->numInner = 0; out->maxDepth = 0; return MIPT_OK; }
    const int64_t inner2 = (oldSlots - 1) / 2;
#if MIPT_NODE_WIDTH == 4
    const int64_t slotsPerLine = 2;
#else
    const int64_t slotsPerLine = 1;
#endif
    const int64_t capSlots = slotsPerLine * (inner2 + 1);          // every wide node alone in its line: cannot be exceeded
    uint32_t* slotOf = nullptr; int32_t* depthOf = nullptr; uint4* wide = nullptr; CollapseCounters* cc = nullptr;
    CollapseCounters h = {};
    int rc = MIPT_OK;
    hipError_t e = hipSuccess;
    auto tryq = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    tryq(hipMalloc(&slotOf, 4 * (size_t)oldSlots)); tryq(hipMalloc(&depthOf, 4 * (size_t)oldSlots));
    tryq(hipMalloc(&wide, 16 * MIPT_NODE_U4 * (size_t)capSlots)); tryq(hipMalloc(&cc, sizeof(CollapseCounters)));
    if (e == hipSuccess) {
        tryq(hipMemsetAsync(slotOf, 0xFF, 4 * (size_t)oldSlots, stream));
        tryq(hipMemsetAsync(wide, 0, 16 * MIPT_NODE_U4 * (size_t)capSlots, stream));
        hipLaunchKernelGGL(k_collapse_init, dim3(1), dim3(1), 0, stream, slotOf, depthOf, cc);
        int32_t first = 0;
        for (size_t d = 0; d < levelEnd.size(); d++) {
            int32_t last = levelEnd[d];
            if (last > first)
                hipLaunchKernelGGL(k_collapse_level, dim3((unsigned)((last - first + 255) / 256)), dim3(256), 0, stream, out->nodes, first, last,
                                   slotOf, depthOf, out->qcell[0], out->qcell[1], out->qcell[2], wide, cc);
            first = last;
        }
        tryq(hipMemcpyAsync(&h, cc, sizeof(h), hipMemcpyDeviceToHost, stream));
        tryq(hipStreamSynchronize(stream));
        tryq(hipGetLastError());
    }
    if (e == hipSuccess) {
        const int64_t used = slotsPerLine * (int64_t)h.lines;
        uint4* fit = nullptr;
        if (used < capSlots && hipMalloc(&fit, 16 * MIPT_NODE_U4 * (size_t)used) == hipSuccess) {      // give the unused tail back
            tryq(hipMemcpy(fit, wide, 16 * MIPT_NODE_U4 * (size_t)used, hipMemcpyDeviceToDevice));
            (void)hipFree(wide);
            wide = fit;
        }
    }
    if (e == hipSuccess) {
        (void)hipFree(out->nodes);
        out->nodes = wide; wide = nullptr;
        out->numNodeSlots = slotsPerLine * (int64_t)h.lines;
        out->numInner = h.numWide;
        out->maxDepth = h.maxDepth;
    } else {
        fprintf(stderr, "bvh build: collapse failed: %s\n", hipGetErrorString(e));
        rc = MIPT_E_DEVICE;
    }
    if (slotOf) (void)hipFree(slotOf);
    if (depthOf) (void)hipFree(depthOf);
    if (wide) (void)hipFree(wide);
    if (cc) (void)hipFree(cc);
    return rc;
}

// Child codes of the finished tree: one index space, node slots first, leaf slots behind them.
__global__ void __launch_bounds__(256) k_unify_codes(uint4* wide, int64_t numSlots, uint32_t leafBase)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numSlots) return;
    auto fix = [&](uint32_t x) { return (x != MIPT_NO_NODE && (x & MIPT_LEAF_BIT)) ? leafBase + (x & ~MIPT_LEAF_BIT) : x; };
    for (int k = 3 * MIPT_NODE_WIDTH / 4; k < MIPT_NODE_U4; k++) {      // the uint4 that hold the child codes
        uint4 c = wide[i * MIPT_NODE_U4 + k];
        wide[i * MIPT_NODE_U4 + k] = make_uint4(fix(c.x), fix(c.y), fix(c.z), fix(c.w));
    }
}

// ---------------------------------------------------------------------------------------
// The slots that the traversal keeps in LDS.  The collapse hands out slots level by level, so the first
// MIPT_TRACE_CACHE_SLOTS slots are the top levels of the tree and whatever part of the next level came first.  A ray
// visits a node with a probability that goes with the surface of its box, so the nodes with the largest boxes are
// exchanged into those slots (on the headline scene they receive 0.535 of the visits of random rays against 0.508).
__global__ void __launch_bounds__(256) k_slot_area(const uint4* __restrict__ wide, int64_t numSlots, float cx, float cy, float cz, float* area)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numSlots) return;
    uint32_t w[4 * MIPT_NODE_U4];
    for (int k = 0; k < MIPT_NODE_U4; k++) { uint4 q = wide[i * MIPT_NODE_U4 + k]; w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w; }
    int lo[3] = { 1 << 20, 1 << 20, 1 << 20 }, hi[3] = { -1, -1, -1 };
    bool any = false;
    for (int c = 0; c < MIPT_NODE_WIDTH; c++) {
        if (w[3 * MIPT_NODE_WIDTH + c] == MIPT_NO_NODE) continue;
        // a slot that holds no node is all zeros: its "children" have code 0 and empty extent, and count for nothing
        any = true;
        for (int a = 0; a < 3; a++) { int l = (int)(w[3 * c + a] & 0xFFFFu), h = (int)(w[3 * c + a] >> 16); lo[a] = l < lo[a] ? l : lo[a]; hi[a] = h > hi[a] ? h : hi[a]; }
    }
    float dx = any ? fmaxf((float)(hi[0] - lo[0]), 0.0f) * cx : 0.0f, dy = any ? fmaxf((float)(hi[1] - lo[1]), 0.0f) * cy : 0.0f, dz = any ? fmaxf((float)(hi[2] - lo[2]), 0.0f) * cz : 0.0f;
    area[i] = dx * dy + dy * dz + dz * dx;
}
__global__ void __launch_bounds__(256) k_remap_identity(uint32_t* remap, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) remap[i] = (uint32_t)i;
}
// pairs[i] = (a, b): the nodes in slots a and b change places
__global__ void __launch_bounds__(256) k_swap_slots(uint4* wide, const uint2* __restrict__ pairs, int n, uint32_t* remap)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * MIPT_NODE_U4) return;
    const uint2 p = pairs[i / MIPT_NODE_U4];
    const int k = i % MIPT_NODE_U4;
    uint4 x = wide[(size_t)p.x * MIPT_NODE_U4 + k], y = wide[(size_t)p.y * MIPT_NODE_U4 + k];
    wide[(size_t)p.x * MIPT_NODE_U4 + k] = y; wide[(size_t)p.y * MIPT_NODE_U4 + k] = x;
    if (k == 0) { remap[p.x] = p.y; remap[p.y] = p.x; }
}
__global__ void __launch_bounds__(256) k_remap_codes(uint4* wide, int64_t numSlots, const uint32_t* __restrict__ remap)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numSlots) return;
    auto fix = [&](uint32_t x) { return (x != MIPT_NO_NODE && x < (uint32_t)numSlots) ? remap[x] : x; };      // codes from numSlots on are leaves
    for (int k = 3 * MIPT_NODE_WIDTH / 4; k < MIPT_NODE_U4; k++) {
        uint4 c = wide[i * MIPT_NODE_U4 + k];
        wide[i * MIPT_NODE_U4 + k] = make_uint4(fix(c.x), fix(c.y), fix(c.z), fix(c.w));
    }
}

// Called with the finished tree (unified child codes).  Returns MIPT_OK or an error code; the tree stays valid either way
// up to the point of failure being a HIP error, after which the caller frees it.
static int promote_large_nodes(BvhResult* out, int64_t topSlots, hipStream_t stream)
{
    const int64_t n = out->numNodeSlots;
    if (n <= topSlots || topSlots <= 0 || getenv("MIPT_BUILD_NO_PROMOTE")) return MIPT_OK;
    float* dArea = nullptr; uint32_t* dRemap = nullptr; uint2* dPairs = nullptr;
    std::vector<float> area((size_t)n);
    std::vector<uint2> pairs;
    hipError_t e = hipSuccess;
    auto tryq = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    const unsigned grid = (unsigned)((n + 255) / 256);
    tryq(hipMalloc(&dArea, sizeof(float) * (size_t)n));
    tryq(hipMalloc(&dRemap, sizeof(uint32_t) * (size_t)n));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_slot_area, dim3(grid), dim3(256), 0, stream, out->nodes, n, out->qcell[0], out->qcell[1], out->qcell[2], dArea);
        hipLaunchKernelGGL(k_remap_identity, dim3(grid), dim3(256), 0, stream, dRemap, n);
        tryq(hipMemcpyAsync(area.data(), dArea, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, stream));
        tryq(hipStreamSynchronize(stream));
    }
    if (e == hipSuccess) {
        // the topSlots slots of largest area, equal areas by lower slot number: those of them that lie beyond the top
        // change places with the slots of the top that are not among them
        std::vector<uint32_t> order((size_t)n);
        for (int64_t i = 0; i < n; i++) order[(size_t)i] = (uint32_t)i;
        auto larger = [&](uint32_t a, uint32_t b) { return area[a] > area[b] || (area[a] == area[b] && a < b); };
        std::nth_element(order.begin(), order.begin() + topSlots, order.end(), larger);
        std::vector<uint8_t> chosen((size_t)topSlots, 0);
        std::vector<uint32_t> in;
        for (int64_t i = 0; i < topSlots; i++) { uint32_t s = order[(size_t)i]; if ((int64_t)s < topSlots) chosen[s] = 1; else in.push_back(s); }
        std::sort(in.begin(), in.end());
        size_t k = 0;
        for (int64_t s = 0; s < topSlots && k < in.size(); s++) if (!chosen[(size_t)s]) pairs.push_back(make_uint2(in[k++], (uint32_t)s));
    }
    if (e == hipSuccess && !pairs.empty()) {
        tryq(hipMalloc(&dPairs, sizeof(uint2) * pairs.size()));
        tryq(hipMemcpyAsync(dPairs, pairs.data(), sizeof(uint2) * pairs.size(), hipMemcpyHostToDevice, stream));
        if (e == hipSuccess) {
            const int np = (int)pairs.size();
            hipLaunchKernelGGL(k_swap_slots, dim3((unsigned)((np * MIPT_NODE_U4 + 255) / 256)), dim3(256), 0, stream, out->nodes, dPairs, np, dRemap);
            hipLaunchKernelGGL(k_remap_codes, dim3(grid), dim3(256), 0, stream, out->nodes, n, dRemap);
            uint32_t root = out->rootCode;
            if (root != MIPT_NO_NODE && root < (uint32_t)n) tryq(hipMemcpyAsync(&root, dRemap + root, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            tryq(hipStreamSynchronize(stream));
            tryq(hipGetLastError());
            out->rootCode = root;
        }
    }
    if (dArea) (void)hipFree(dArea);
    if (dRemap) (void)hipFree(dRemap);
    if (dPairs) (void)hipFree(dPairs);
    if (e != hipSuccess) { fprintf(stderr, "bvh build: promoting the largest nodes failed: %s\n", hipGetErrorString(e)); return MIPT_E_DEVICE; }
    return MIPT_OK;
}

void mipt_free_bvh(BvhResult* r)
{
    if (!r) return;
    if (r->nodes) (void)hipFree(r->nodes);
    if (r->tris) (void)hipFree(r->tris);
    *r = BvhResult();
}

int mipt_build_bvh(const float4* d_attr, int64_t numTris, const mipt_build_options* opt, hipStream_t stream, BvhResult* out)
{
    *out = BvhResult();
    if (numTris <= 0) return MIPT_OK;
    if (numTris >= (1ll << 30)) return MIPT_E_LIMIT;
    int rc = MIPT_OK;
    const int32_t n = (int32_t)numTris;
    int nb = (opt && opt->sah_bins > 0) ? opt->sah_bins : 32;
    if (nb < 2) nb = 2;
    if (nb > MAX_BINS) nb = MAX_BINS;
    int maxLeaf = (opt && opt->max_leaf_triangles > 0) ? opt->max_leaf_triangles : 4;
    float travCost = (opt && opt->traversal_cost > 0) ? opt->traversal_cost : 1.0f;

    float4 *primMin = nullptr, *primMax = nullptr;
    int32_t *primIdx[2] = { nullptr, nullptr }, *posNode[2] = { nullptr, nullptr }, *active[2] = { nullptr, nullptr };
    BNode* nodes = nullptr; Bin* bins = nullptr; BuildCounters* ctr = nullptr;
    int64_t binCap = 0;
    const int64_t maxNodes = 2 * (int64_t)n + 2;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    BuildCounters h = {};
    std::vector<int32_t> levelEnd{ 1 };            // levelEnd[d] = number of nodes of depth <= d
    BNode rootH;
    int cur = 0, numActive = 0;
    const unsigned gridN = (unsigned)((n + 255) / 256);
    float sceneAbs = 0, pad = 0, ms = 0;
    QuantGrid grid = {};

    HIPCHK(hipMalloc(&primMin, sizeof(float4) * (size_t)n));
    HIPCHK(hipMalloc(&primMax, sizeof(float4) * (size_t)n));
    for (int k = 0; k < 2; k++) {
        HIPCHK(hipMalloc(&primIdx[k], sizeof(int32_t) * (size_t)n));
        HIPCHK(hipMalloc(&posNode[k], sizeof(int32_t) * (size_t)n));
        HIPCHK(hipMalloc(&active[k], sizeof(int32_t) * (size_t)(n + 2)));
    }
    HIPCHK(hipMalloc(&nodes, sizeof(BNode) * (size_t)maxNodes));
    HIPCHK(hipMalloc(&ctr, sizeof(BuildCounters)));
    HIPCHK(hipEventCreate(&ev0)); HIPCHK(hipEventCreate(&ev1));
    HIPCHK(hipEventRecord(ev0, stream));

    hipLaunchKernelGGL(k_init_root, dim3(1), dim3(1), 0, stream, nodes, ctr, active[0], n);
    hipLaunchKernelGGL(k_prim_bounds, dim3(gridN), dim3(256), 0, stream, d_attr, n, primMin, primMax, primIdx[0], posNode[0], nodes);
    numActive = (n > 1) ? 1 : 0;
    for (int level = 0; numActive > 0 && level <= MAX_BUILD_DEPTH + 1; level++) {
        int64_t needBins = (int64_t)numActive * 3 * nb;
        if (needBins > binCap) {
            if (bins) { HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipFree(bins)); bins = nullptr; }
            binCap = needBins + needBins / 2 + 3 * nb * 64;
            HIPCHK(hipMalloc(&bins, sizeof(Bin) * (size_t)binCap));
        }
        hipLaunchKernelGGL(k_clear_bins, dim3((unsigned)((needBins + 255) / 256)), dim3(256), 0, stream, bins, needBins);
        hipLaunchKernelGGL(k_bin, dim3((unsigned)((n + BIN_CHUNK - 1) / BIN_CHUNK)), dim3(256), 0, stream, primIdx[cur], posNode[cur], nodes, primMin, primMax, bins, n, nb);
        hipLaunchKernelGGL(k_split, dim3((unsigned)((numActive + 63) / 64)), dim3(64), 0, stream, nodes, active[cur], numActive, bins, nb,
                           maxLeaf, travCost, ctr, active[cur ^ 1]);
        hipLaunchKernelGGL(k_partition, dim3(gridN), dim3(256), 0, stream, primIdx[cur], posNode[cur], primIdx[cur ^ 1], posNode[cur ^ 1],
                           nodes, primMin, primMax, n, nb);
        HIPCHK(hipMemcpyAsync(&h, ctr, sizeof(h), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        numActive = h.nextActive;
        if (h.nodeCount > maxNodes) { rc = MIPT_E_LIMIT; goto done; }
        levelEnd.push_back(h.nodeCount);
        HIPCHK(hipMemsetAsync(&ctr->nextActive, 0, sizeof(int32_t), stream));
        cur ^= 1;
    }
    HIPCHK(hipMemcpyAsync(&h, ctr, sizeof(h), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(&rootH, nodes, sizeof(BNode), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    {
        // Every box grows by 2^-20 of the largest coordinate magnitude of the scene before it is rounded outwards to the
        // grid, which covers the rounding of the quantisation itself (a 32nd of a cell of the longest axis against the
        // 2^-8 cells by which (v - qmin) / cell can be off).  The margin against the rounding of the traversal's box
        // test and of the triangle test is not in the boxes: the traversal applies it per ray (pt_device.h: BoxRay).
#ifndef MIPT_PAD_SCALE
#define MIPT_PAD_SCALE (1.0f / 32.0f)
#endif
        auto k2f = [](int32_t i) { int32_t j = i >= 0 ? i : i ^ 0x7fffffff; float f; memcpy(&f, &j, 4); return f; };
        for (int a = 0; a < 3; a++) {
            float lo = fabsf(k2f(rootH.bminI[a])), hi = fabsf(k2f(rootH.bmaxI[a]));
            if (lo > sceneAbs) sceneAbs = lo;
            if (hi > sceneAbs) sceneAbs = hi;
        }
        pad = sceneAbs * (MIPT_PAD_SCALE / 32768.0f);
        if (!(pad > 0.0f)) pad = 1e-30f;
        // the 16-bit grid spans the padded scene box; a flat axis still gets a positive cell
        for (int a = 0; a < 3; a++) {
            float lo = k2f(rootH.bminI[a]) - 2.0f * pad, hi = k2f(rootH.bmaxI[a]) + 2.0f * pad;
            float cell = (hi - lo) / 65535.0f;
            if (!(cell > 0.0f)) cell = pad;
            out->qmin[a] = lo; out->qcell[a] = cell;
            grid.qmin[a] = lo; grid.invCell[a] = 1.0f / cell;
        }
    }
    out->numNodeSlots = h.nodeCount;
    out->numLeafSlots = n;
    HIPCHK(hipMalloc(&out->nodes, sizeof(uint4) * 2 * (size_t)out->numNodeSlots));
    HIPCHK(hipMalloc(&out->tris, sizeof(float4) * MIPT_TRI_F4 * (size_t)n));
    HIPCHK(hipMemsetAsync(&ctr->sah, 0, sizeof(double), stream));
    hipLaunchKernelGGL(k_emit_nodes, dim3((unsigned)((h.nodeCount + 255) / 256)), dim3(256), 0, stream, nodes, h.nodeCount, pad, grid, out->nodes, ctr);
    hipLaunchKernelGGL(k_emit_tris, dim3(gridN), dim3(256), 0, stream, d_attr, primIdx[cur], posNode[cur], nodes, n, out->tris);
    HIPCHK(hipEventRecord(ev1, stream));
    HIPCHK(hipMemcpyAsync(&h, ctr, sizeof(h), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
    out->buildMs = ms;
    out->numLeaves = h.leafCount;
    out->numInner = (h.nodeCount - 1) / 2;
    out->maxDepth = h.maxDepth;
    out->rootCode = (rootH.state == NODE_LEAF || n == 1) ? (MIPT_LEAF_BIT | 0u) : 0u;
    {
        auto k2f = [](int32_t i) { int32_t j = i >= 0 ? i : i ^ 0x7fffffff; float f; memcpy(&f, &j, 4); return f; };
        float dx = k2f(rootH.bmaxI[0]) - k2f(rootH.bminI[0]), dy = k2f(rootH.bmaxI[1]) - k2f(rootH.bminI[1]), dz = k2f(rootH.bmaxI[2]) - k2f(rootH.bminI[2]);
        double ra = 2.0 * ((double)dx * dy + (double)dy * dz + (double)dz * dx);
        out->sahCost = ra > 0 ? h.sah / ra : 0.0;
    }
    {
        auto t0 = std::chrono::steady_clock::now();
        rc = collapse_to_wide(out, levelEnd, stream);
        if (rc) goto done;
        float hostMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (getenv("MIPT_BUILD_VERBOSE")) fprintf(stderr, "bvh build: %d triangles, device %.2f ms, collapse to wide nodes %.2f ms\n", n, (float)out->buildMs, hostMs);
        out->buildMs += hostMs;
    }
    if (out->rootCode & MIPT_LEAF_BIT) {
        // the whole scene is one leaf: no node slots at all, the root is leaf slot 0
        (void)hipFree(out->nodes); out->nodes = nullptr;
        out->numNodeSlots = 0;
        out->rootCode = 0;
    } else {
        hipLaunchKernelGGL(k_unify_codes, dim3((unsigned)((out->numNodeSlots + 255) / 256)), dim3(256), 0, stream, out->nodes, out->numNodeSlots, (uint32_t)out->numNodeSlots);
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipGetLastError());
        {
            auto t0 = std::chrono::steady_clock::now();
            rc = promote_large_nodes(out, MIPT_TRACE_CACHE_SLOTS, stream);
            if (rc) goto done;
            out->buildMs += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
    }
done:
    if (primMin) (void)hipFree(primMin);
    if (primMax) (void)hipFree(primMax);
    for (int k = 0; k < 2; k++) {
        if (primIdx[k]) (void)hipFree(primIdx[k]);
        if (posNode[k]) (void)hipFree(posNode[k]);
        if (active[k]) (void)hipFree(active[k]);
    }
    if (nodes) (void)hipFree(nodes);
    if (bins) (void)hipFree(bins);
    if (ctr) (void)hipFree(ctr);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (rc != MIPT_OK) mipt_free_bvh(out);
    return rc;
}
