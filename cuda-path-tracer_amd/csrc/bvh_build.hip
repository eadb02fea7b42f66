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
#include <chrono>
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
__global__ void __launch_bounds__(256) k_prim_bounds(const float4* __restrict__ attr, int32_t n, float4* primMin,
                                                    float4* primMax, int32_t* primIdx, int32_t* posNode, BNode* root)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* a = attr + (size_t)i * 6;
    float4 p0 = a[0], p1 = a[2], p2 = a[4];
    float mnx = fminf(p0.x, fminf(p1.x, p2.x)), mny = fminf(p0.y, fminf(p1.y, p2.y)), mnz = fminf(p0.z, fminf(p1.z, p2.z));
    float mxx = fmaxf(p0.x, fmaxf(p1.x, p2.x)), mxy = fmaxf(p0.y, fmaxf(p1.y, p2.y)), mxz = fmaxf(p0.z, fmaxf(p1.z, p2.z));
    primMin[i] = make_float4(mnx, mny, mnz, 0);
    primMax[i] = make_float4(mxx, mxy, mxz, 0);
    primIdx[i] = i;
    posNode[i] = 0;
    float cx = 0.5f * (mnx + mxx), cy = 0.5f * (mny + mxy), cz = 0.5f * (mnz + mxz);
    atomicMin(&root->bminI[0], f2key(mnx)); atomicMin(&root->bminI[1], f2key(mny)); atomicMin(&root->bminI[2], f2key(mnz));
    atomicMax(&root->bmaxI[0], f2key(mxx)); atomicMax(&root->bmaxI[1], f2key(mxy)); atomicMax(&root->bmaxI[2], f2key(mxz));
    atomicMin(&root->cminI[0], f2key(cx)); atomicMin(&root->cminI[1], f2key(cy)); atomicMin(&root->cminI[2], f2key(cz));
    atomicMax(&root->cmaxI[0], f2key(cx)); atomicMax(&root->cmaxI[1], f2key(cy)); atomicMax(&root->cmaxI[2], f2key(cz));
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

__global__ void __launch_bounds__(256) k_bin(const int32_t* __restrict__ primIdx, const int32_t* __restrict__ posNode,
                                            const BNode* __restrict__ nodes, const float4* __restrict__ primMin,
                                            const float4* __restrict__ primMax, Bin* bins, int32_t n, int nb)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const BNode& nd = nodes[posNode[p]];
    if (nd.state != NODE_ACTIVE) return;
    int prim = primIdx[p];
    float4 mn = primMin[prim], mx = primMax[prim];
    const float c[3] = { 0.5f * (mn.x + mx.x), 0.5f * (mn.y + mx.y), 0.5f * (mn.z + mx.z) };
    int kmin[3] = { f2key(mn.x), f2key(mn.y), f2key(mn.z) }, kmax[3] = { f2key(mx.x), f2key(mx.y), f2key(mx.z) };
    Bin* base = bins + (size_t)nd.activeSlot * 3 * nb;
    for (int a = 0; a < 3; a++) {
        int b = bin_of(c[a], key2f(nd.cminI[a]), key2f(nd.cmaxI[a]), nb);
        Bin* B = base + a * nb + b;
        atomicMin(&B->minI[0], kmin[0]); atomicMin(&B->minI[1], kmin[1]); atomicMin(&B->minI[2], kmin[2]);
        atomicMax(&B->maxI[0], kmax[0]); atomicMax(&B->maxI[1], kmax[1]); atomicMax(&B->maxI[2], kmax[2]);
        atomicAdd(&B->count, 1);
    }
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
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int id = posNode[p];
    int prim = primIdx[p];
    BNode& nd = nodes[id];
    if (nd.state != NODE_INNER) {        // finished leaf: stays where it is
        primIdxOut[p] = prim; posNodeOut[p] = id;
        return;
    }
    float4 mn = primMin[prim], mx = primMax[prim];
    const float c[3] = { 0.5f * (mn.x + mx.x), 0.5f * (mn.y + mx.y), 0.5f * (mn.z + mx.z) };
    bool goLeft;
    if (nd.axis >= 0) goLeft = bin_of(c[nd.axis], key2f(nd.cminI[nd.axis]), key2f(nd.cmaxI[nd.axis]), nb) < nd.splitBin;
    else goLeft = (p - nd.start) < nd.nLeft;
    int child = nd.left + (goLeft ? 0 : 1);
    int dst = goLeft ? nd.start + atomicAdd(&nd.fillL, 1) : nd.start + nd.nLeft + atomicAdd(&nd.fillR, 1);
    primIdxOut[dst] = prim;
    posNodeOut[dst] = child;
    BNode& ch = nodes[child];
    atomicMin(&ch.cminI[0], f2key(c[0])); atomicMin(&ch.cminI[1], f2key(c[1])); atomicMin(&ch.cminI[2], f2key(c[2]));
    atomicMax(&ch.cmaxI[0], f2key(c[0])); atomicMax(&ch.cmaxI[1], f2key(c[1])); atomicMax(&ch.cmaxI[2], f2key(c[2]));
    if (nd.axis < 0) {
        atomicMin(&ch.bminI[0], f2key(mn.x)); atomicMin(&ch.bminI[1], f2key(mn.y)); atomicMin(&ch.bminI[2], f2key(mn.z));
        atomicMax(&ch.bmaxI[0], f2key(mx.x)); atomicMax(&ch.bmaxI[1], f2key(mx.y)); atomicMax(&ch.bmaxI[2], f2key(mx.z));
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
// Wide nodes.  The builder's binary tree is collapsed into nodes with up to four children
// (64 bytes: twelve box words and four child codes, see pt_internal.h): starting from the two
// children of a binary node, the inner child with the largest surface is replaced by its own
// two children until four are reached.  A traversal step then costs one cache line per lane for
// two binary levels.  Two wide nodes share a 128-byte line: a node and its largest inner child.
// Runs on the host over the downloaded binary nodes (a few milliseconds per 100 k triangles).
struct WChild { uint32_t code; uint32_t w[3]; };     // w[a] = min | max << 16 on axis a

static inline double wchild_area(const WChild& c, const float* qcell)
{
    double e[3];
    for (int a = 0; a < 3; a++) e[a] = ((double)(c.w[a] >> 16) - (double)(c.w[a] & 0xffffu)) * qcell[a];
    return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
}

static int collapse_to_wide(BvhResult* out, hipStream_t stream)
{
    const int64_t oldSlots = out->numNodeSlots;
    if (oldSlots <= 0 || (out->rootCode & MIPT_LEAF_BIT)) { out->numInner = 0; out->maxDepth = 0; return MIPT_OK; }
    std::vector<uint32_t> src((size_t)oldSlots * 8);
    if (hipMemcpyAsync(src.data(), out->nodes, src.size() * 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return MIPT_E_DEVICE;
    if (hipStreamSynchronize(stream) != hipSuccess) return MIPT_E_DEVICE;

    struct Wide { WChild k[MIPT_NODE_WIDTH]; int n; };
    auto make_wide = [&](uint32_t x) {
        Wide W; W.n = 2;
        for (int c = 0; c < 2; c++) {
            W.k[c].code = src[(size_t)x * 8 + 6 + c];
            for (int a = 0; a < 3; a++) W.k[c].w[a] = src[(size_t)x * 8 + 3 * c + a];
        }
        while (W.n < MIPT_NODE_WIDTH) {
            int best = -1; double bestArea = -1.0;
            for (int i = 0; i < W.n; i++) {
                if (W.k[i].code & MIPT_LEAF_BIT) continue;
                double ar = wchild_area(W.k[i], out->qcell);
                if (ar > bestArea) { bestArea = ar; best = i; }
            }
            if (best < 0) break;
            uint32_t y = W.k[best].code;
            for (int c = 0; c < 2; c++) {
                WChild& d = (c == 0) ? W.k[best] : W.k[W.n];
                d.code = src[(size_t)y * 8 + 6 + c];
                for (int a = 0; a < 3; a++) d.w[a] = src[(size_t)y * 8 + 3 * c + a];
            }
            W.n++;
        }
        return W;
    };

    std::vector<uint32_t> slotOf((size_t)oldSlots, MIPT_NO_NODE);
    std::vector<Wide> wide;                       // by slot; n = 0 marks an unused slot
    struct Pending { uint32_t node; int32_t depth; };
    std::vector<Pending> roots{ { 0u, 0 } };
    int32_t maxDepth = 0; int64_t numWide = 0;
    for (size_t head = 0; head < roots.size(); head++) {
        const Pending r = roots[head];
        slotOf[r.node] = (uint32_t)wide.size();
        wide.push_back(make_wide(r.node)); numWide++;
        const Wide W = wide.back();
        if (r.depth + 1 > maxDepth) maxDepth = r.depth + 1;
        int mate = -1; double mateArea = -1.0;
        for (int i = 0; i < W.n; i++) {
            if (W.k[i].code & MIPT_LEAF_BIT) continue;
            double ar = wchild_area(W.k[i], out->qcell);
            if (ar > mateArea) { mateArea = ar; mate = i; }
        }
        for (int i = 0; i < W.n; i++)
            if (!(W.k[i].code & MIPT_LEAF_BIT) && i != mate) roots.push_back({ W.k[i].code, r.depth + 1 });
        if (mate >= 0) {
            uint32_t y = W.k[mate].code;
            slotOf[y] = (uint32_t)wide.size();
            wide.push_back(make_wide(y)); numWide++;
            const Wide M = wide.back();
            if (r.depth + 2 > maxDepth) maxDepth = r.depth + 2;
            for (int i = 0; i < M.n; i++)
                if (!(M.k[i].code & MIPT_LEAF_BIT)) roots.push_back({ M.k[i].code, r.depth + 2 });
        } else {
            Wide E; E.n = 0; wide.push_back(E);
        }
    }
    std::vector<uint32_t> dst(wide.size() * 16, 0u);
    for (size_t i = 0; i < wide.size(); i++) {
        const Wide& W = wide[i];
        if (W.n == 0) continue;
        for (int k = 0; k < MIPT_NODE_WIDTH; k++) {
            uint32_t* boxw = &dst[i * 16 + 3 * k];
            if (k < W.n) {
                boxw[0] = W.k[k].w[0]; boxw[1] = W.k[k].w[1]; boxw[2] = W.k[k].w[2];
                uint32_t code = W.k[k].code;
                dst[i * 16 + 12 + k] = (code & MIPT_LEAF_BIT) ? code : slotOf[code];
            } else {
                boxw[0] = boxw[1] = boxw[2] = 0x0000FFFFu;     // min above max: no ray enters an unused child
                dst[i * 16 + 12 + k] = MIPT_NO_NODE;
            }
        }
    }
    uint4* nn = nullptr;
    if (hipMalloc(&nn, dst.size() * 4) != hipSuccess) return MIPT_E_DEVICE;
    if (hipMemcpy(nn, dst.data(), dst.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(nn); return MIPT_E_DEVICE; }
    (void)hipFree(out->nodes);
    out->nodes = nn;
    out->numNodeSlots = (int64_t)wide.size();
    out->numInner = numWide;
    out->maxDepth = maxDepth;
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
        hipLaunchKernelGGL(k_bin, dim3(gridN), dim3(256), 0, stream, primIdx[cur], posNode[cur], nodes, primMin, primMax, bins, n, nb);
        hipLaunchKernelGGL(k_split, dim3((unsigned)((numActive + 63) / 64)), dim3(64), 0, stream, nodes, active[cur], numActive, bins, nb,
                           maxLeaf, travCost, ctr, active[cur ^ 1]);
        hipLaunchKernelGGL(k_partition, dim3(gridN), dim3(256), 0, stream, primIdx[cur], posNode[cur], primIdx[cur ^ 1], posNode[cur ^ 1],
                           nodes, primMin, primMax, n, nb);
        HIPCHK(hipMemcpyAsync(&h, ctr, sizeof(h), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        numActive = h.nextActive;
        if (h.nodeCount > maxNodes) { rc = MIPT_E_LIMIT; goto done; }
        HIPCHK(hipMemsetAsync(&ctr->nextActive, 0, sizeof(int32_t), stream));
        cur ^= 1;
    }
    HIPCHK(hipMemcpyAsync(&h, ctr, sizeof(h), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(&rootH, nodes, sizeof(BNode), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    {
        // every box grows by 2^-15 of the largest coordinate magnitude of the scene
        auto k2f = [](int32_t i) { int32_t j = i >= 0 ? i : i ^ 0x7fffffff; float f; memcpy(&f, &j, 4); return f; };
        for (int a = 0; a < 3; a++) {
            float lo = fabsf(k2f(rootH.bminI[a])), hi = fabsf(k2f(rootH.bmaxI[a]));
            if (lo > sceneAbs) sceneAbs = lo;
            if (hi > sceneAbs) sceneAbs = hi;
        }
        pad = sceneAbs * (1.0f / 32768.0f);
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
        rc = collapse_to_wide(out, stream);
        if (rc) goto done;
        out->buildMs += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
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
