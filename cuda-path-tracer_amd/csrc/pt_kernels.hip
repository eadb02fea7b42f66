// Wavefront kernels of the path tracer.
//
// The reference evaluates one recursive Radiance() per pixel inside a single kernel
// (PTKernel, PathTracer.cu:501-554).  Here the same estimator is cut into passes over
// queues of rays:
//   k_raygen          primary rays                           PathTracer.cu:512-539
//   k_trace<0>        closest-hit search                     TraceRays, PathTracer.cu:244-288
//   k_shade_primary   surface + light at the first hit       Radiance, PathTracer.cu:389-428
//   k_gen_first       the N hemisphere samples of a pixel    PathTracer.cu:435-445
//   k_shade_bounce    surface + light + next sample          Radiance at depth 1..3
//   k_trace<1>        any-hit towards the light              TraceRays(bHitOnly), :421
//   k_resolve         per-level clamp, backwards             PathTracer.cu:449-462
//   k_finalize        first-level sum, clamp, *255           PathTracer.cu:453-462, 550-552
// A path slot is (first-bounce sample s of this pass, active pixel a): slot = s * nAct + a.
#include "pt_device.h"
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <stdio.h>

#define WAVE 64
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// with -DMIPT_ISA_MARKERS the traversal steps are bracketed by comments in the assembly (profiles/isa_loops.py cuts them out)
#ifdef MIPT_ISA_MARKERS
#define ISA_MARK(text) asm volatile("; MIPT " text)
#else
#define ISA_MARK(text) do { } while (0)
#endif
// queue entries a wave of a shading kernel handles per reservation in its output queues
#define SHADE_ITEMS 16

// Streaming store: the ray queues and path states are written once and read once by a later
// kernel; marking the stores non-temporal keeps them from displacing the shading records and
// texels, which are gathered again and again, from the caches (bounce shading -5 %; the same
// mark on the loads of queue entries made the traversal slower).
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_stream(float4* p, float4 v)
{
    v4f_t x = { v.x, v.y, v.z, v.w };
    __builtin_nontemporal_store(x, (v4f_t*)p);
}

// ---------------------------------------------------------------------------------------
// k_raygen: CPUPathTracer.cpp:489-497 (the CPU flavour transforms (vx,vy,1) first and
// normalises once; XMVector3TransformCoord/Normal evaluate ((z*r2 [+ r3]) + y*r1) + x*r0)
__device__ __forceinline__ bool pixel_owned(const DevRender& R, int p);

// Ray i of the primary queue belongs to pixel i, or, when only this shard's pixels are rendered, to pixel ownPix[i].
__device__ __forceinline__ int primary_count(const DevRender& R) { return R.ownPix ? R.numOwn : R.W * R.H; }

__global__ void __launch_bounds__(256) k_raygen(DevRender R, RayQueue q)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= primary_count(R)) return;
    const int p = R.ownPix ? R.ownPix[i] : i;
    int j = p % R.W, h = p / R.W;
    const float* m = R.toLocal;
    float vx = (+2.0f * (float)R.div * j / (float)R.clientW - 1.0f) / R.p0;
    float vy = (-2.0f * (float)R.div * h / (float)R.clientH + 1.0f) / R.p1;
    float r0 = ((0.0f * m[8] + m[12]) + 0.0f * m[4]) + 0.0f * m[0];
    float r1 = ((0.0f * m[9] + m[13]) + 0.0f * m[5]) + 0.0f * m[1];
    float r2 = ((0.0f * m[10] + m[14]) + 0.0f * m[6]) + 0.0f * m[2];
    float r3 = ((0.0f * m[11] + m[15]) + 0.0f * m[7]) + 0.0f * m[3];
    F3 dir;
    dir.x = (1.0f * m[8] + vy * m[4]) + vx * m[0];
    dir.y = (1.0f * m[9] + vy * m[5]) + vx * m[1];
    dir.z = (1.0f * m[10] + vy * m[6]) + vx * m[2];
    dir = f3normalize(dir);
    q.o[i] = make_float4(r0 / r3, r1 / r3, r2 / r3, __int_as_float(p));
    q.d[i] = make_float4(dir.x, dir.y, dir.z, 0.0f);
}

void launch_raygen(const DevRender& R, RayQueue q, hipStream_t s)
{
    int n = R.ownPix ? R.numOwn : R.W * R.H;
    hipLaunchKernelGGL(k_raygen, dim3((n + 255) / 256), dim3(256), 0, s, R, q);
}

__global__ void k_set_i32(int32_t* p, int32_t v) { *p = v; }
void launch_set_i32(int32_t* d_ptr, int32_t value, hipStream_t s)
{
    hipLaunchKernelGGL(k_set_i32, dim3(1), dim3(1), 0, s, d_ptr, value);
}

__global__ void __launch_bounds__(256) k_pack_rays(const float* rays, int64_t n, RayQueue q)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    q.o[i] = make_float4(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2], __int_as_float((int)i));
    q.d[i] = make_float4(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5], 0.0f);
}
void launch_pack_rays(const float* d_rays6, int64_t n, RayQueue q, hipStream_t s)
{
    hipLaunchKernelGGL(k_pack_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_rays6, n, q);
}

// hit records of caller-supplied rays -> the arrays of mipt_trace_rays: id or -1, (t, ua, va) or zeros
__global__ void __launch_bounds__(256) k_unpack_hits(const float4* __restrict__ hit, int64_t n, int32_t* outTri, float* outTuv)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 h = hit[i];
    int32_t tri = __float_as_int(h.w);
    if (outTri) outTri[i] = tri;
    if (outTuv) {
        outTuv[3 * i] = tri >= 0 ? h.x : 0.0f; outTuv[3 * i + 1] = tri >= 0 ? h.y : 0.0f; outTuv[3 * i + 2] = tri >= 0 ? h.z : 0.0f;
    }
}
void launch_unpack_hits(const float4* hit, int64_t n, int32_t* d_outTri, float* d_outTuv, hipStream_t s)
{
    hipLaunchKernelGGL(k_unpack_hits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, hit, n, d_outTri, d_outTuv);
}

// ---------------------------------------------------------------------------------------
// Traversal kernels: TraceRays / Intersect of PathTracer.cu:193-288 over a queue of rays.
//
// One ray per lane, per-lane stack in LDS (row k of lane l at stack[k * 64 + l]: every lane of a
// half-wave hits its own bank; rows beyond the LDS part, which only very deep trees need, live in
// a per-wave slice of global memory).  TRACE_WAVES waves form a workgroup that shares a copy of the
// first TRACE_CACHE_SLOTS node slots (the top of the tree) in LDS and nothing else.  Waves are
// persistent: a wave reserves TRACE_CHUNK consecutive rays of the queue with one atomic (its first
// reservation is the chunk with the wave's own number and needs none) and hands them to its lanes as
// soon as they fall idle.  The queue is either dense (entries 0 .. count-1) or
// in blocks of 1 << MIPT_QBLOCK_SHIFT entries of which the first blockCount[b] are valid (the
// shadow rays of bounce shading, which writes them without any atomic); a reservation never
// straddles a block.
//
// Every lane is at an inner node, at a triangle of a leaf, finished, or idle.  One round of the
// loop hands rays to idle lanes, then runs a phase and writes the results of finished rays.  The
// node phase is a tight loop of node steps that ends when `nodeDrop` of the lanes that entered it
// have left for a leaf or finished; the triangle phase is a short run of triangle steps for the
// lanes that wait at a triangle, entered when `triVote` of them do or no lane is at an inner node
// and repeated while more than `triStay` eighths of them are still at a leaf.  Lanes of the other
// kind sit a phase out.  A round runs the node phase and then the triangle phase.
// The thresholds were chosen by measurement (profiles/sweep_thresholds.sh).
//
// Both steps are straight-line code: no loop and no branch that depends on a lane's data, because
// on this kernel a scalar or branch instruction costs as much time as a vector instruction and
// the exec-mask bookkeeping of a per-lane loop was a third of all instructions of a step.
//   * A node step fetches one 64-byte node (one cache line per lane, or from the copy in LDS) and,
//     with it, the two stack rows below the lane's stack pointer.  It tests the four child boxes,
//     sorts the four stack entries (closest hit: by entry distance; any hit: longest way through
//     the box first; see box_entry), stores the three farthest unconditionally and goes on with the
//     nearest child that was hit, which never passes through the stack.  Only when no child was
//     hit the next node is one of the two rows fetched ahead.
//   * A triangle step is run by all lanes of the wave.  A lane at a leaf ("owner") fetches one 64-byte
//     triangle record and the same two stack rows and runs the exact ray/triangle test including its
//     division; the k-th lane that is not at a leaf tests the NEXT triangle of the k-th owner's leaf with
//     the owner's ray (handed over through ds_bpermute) and hands the result back.  The owner merges
//     both (closest hit: smaller t, then lower id), moves on by two triangles and, past the last
//     triangle of its leaf, goes on with one of the two rows.
// Taking an entry off the stack looks at two rows and never more: the upper one unless its box
// lies beyond the best hit found since it was pushed (then it is dropped), else the lower one,
// whether that one is beyond the best hit or not - visiting a box beyond the best hit is wasted
// work, never a wrong result.  Measured on the headline frame, the upper row is taken in 95 % of
// the cases in node steps and 89 % in triangle steps, and 7 % of the lower rows are such
// needless visits.  The two rows at the bottom of every lane's stack hold the entry "finished",
// which is never dropped, so an empty stack needs no test.
//
// Closest hit (MODE 0): the accepted triangle with the smallest t below FLT_MAX, equal t ->
// lowest global triangle id; hit[i] = (t, ua, va, id or -1).
// Any hit (MODE 1, 2): true as soon as one triangle is accepted with t < FLT_MAX (the bHitOnly
// result, PathTracer.cu:291).  MODE 1 clears the direct term of the slot the shadow ray belongs
// to; MODE 2 writes flags[i].
#define TRACE_CHUNK 256
#define TRACE_CHUNK_SHADOW 512
#define TRACE_WAVES 12            // waves per workgroup
#define TRACE_CACHE_SLOTS MIPT_TRACE_CACHE_SLOTS     // node slots copied to LDS per workgroup (32 KiB)
#define TRACE_TRI_VOTE 12
#define TRACE_NODE_DROP 24
#define TRACE_REFILL_MIN 1     // idle lanes a wave waits for before it hands out new rays
#define TRACE_TRI_VOTE_SHADOW 12
#define TRACE_NODE_DROP_SHADOW 24
#define TRACE_TRI_STAY_SHADOW 3
#define TRACE_TRI_STAY 3      // triangle steps repeat while more than this many eighths of the entering lanes stay at a leaf (0 = one step)
static_assert(MIPT_NODE_WIDTH == 4, "the node step is written for four-wide nodes");

__device__ __forceinline__ void stack_store(uint32_t* lds, uint32_t* spill, int ldsCap, int idx, uint32_t v)
{
    if (idx < ldsCap) lds[idx * WAVE] = v;
    else if (spill) spill[(size_t)(idx - ldsCap) * WAVE] = v;       // no slice: the tree cannot need this row
}
__device__ __forceinline__ uint32_t stack_load(const uint32_t* lds, const uint32_t* spill, int ldsCap, int idx, uint32_t none)
{
    if (idx < ldsCap) return lds[idx * WAVE];
    return spill ? spill[(size_t)(idx - ldsCap) * WAVE] : none;
}
// A child that the ray has to visit travels as ONE word, in the sort and on the stack: the child
// code in the low bits (DevScene::slotMask) and, above them, the high bits of its sort key.  Closest
// hit descends nearest first, the key is the entry distance of the box; any hit descends first into
// the child the ray spends the longest way in (measured: it meets an occluder sooner than
// nearest-first, any-hit kernel -10 %), the key is the largest finite float minus that length.
// Keys are bit patterns of non-negative floats (the sign bit is cut off with the code bits, so a
// -0 from the box test cannot sort as the farthest entry): they order like unsigned integers, so
// the four entries of a node are sorted with integer min/max alone, and cutting the low bits of a
// key off rounds it down.  That keeps the second use of the key safe: when a closest-hit entry is
// taken off the stack and its (rounded down) entry distance lies beyond the best hit found
// meanwhile, the child is dropped without being fetched.  A child that was missed is ENTRY_MISS,
// below every entry of a hit: the four entries are sorted in DESCENDING order (farthest child
// first, nearest last, misses behind it), so the nearest child that was hit is the last entry that
// is not ENTRY_MISS and the others stand in the order in which the stack wants them.
#define ENTRY_MISS 0u
#define CULL_NONE  0x7fffffffu     // no hit yet: every entry passes
template <int MODE>
__device__ __forceinline__ uint32_t box_entry(const BoxRay& r, uint32_t wx, uint32_t wy, uint32_t wz, float tcull, uint32_t code, uint32_t keyMask, unsigned long long& hitLanes)
{
    float t0, t1;
    bool h = box_test<MODE == 0>(r, wx, wy, wz, tcull, t0, t1);
    hitLanes = __ballot(h);          // the lanes of the wave that hit this child: the compare's own result, no further instruction
    uint32_t key = (MODE != 0) ? 0x7f7fffffu - __float_as_uint(t1 - t0) : __float_as_uint(t0);
    uint32_t e = code | (key & keyMask);      // codes of children that exist are below slotMask; a child that does not exist is always missed
    return h ? e : ENTRY_MISS;
}
// compare-exchange: afterwards a >= b
__device__ __forceinline__ void order_pair(uint32_t& a, uint32_t& b)
{
    uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    a = hi; b = lo;
}

template <int MODE, bool BLOCKS, bool COUNT_BOXES = false>
__global__ void __launch_bounds__(1024) k_trace(DevScene S, RayQueue q, const int32_t* __restrict__ d_count,
                                                const int32_t* __restrict__ blockCount, int32_t* head, float4* __restrict__ hit, float4* pixState0,
                                                float4* pathState, int64_t slotsPerLevel, uint8_t* flags,
                                                float eps, unsigned long long* stats, int triVote, int chunk, int nodeDrop, int triStay, int tagAndRefill, int cacheSlots)
{
    const int tagInHit = tagAndRefill & 1, refillMin = tagAndRefill >> 8;      // lanes are handed new rays once this many of them are idle

    // LDS of the workgroup: the first cacheSlots node slots of the tree (its top levels, where more than
    // half of all node visits go), then one stack area per wave.  The waves of a workgroup share the
    // node copy and nothing else: after the copy is made they never meet again.
    extern __shared__ uint4 smem4[];
    // (the wave's number is the same in all its lanes: said explicitly, so that what is derived from it stays in scalar registers)
    const int lane = threadIdx.x & (WAVE - 1), waveInGroup = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const int ldsCap = S.stackLds;
    const uint4* __restrict__ nodes = S.nodes;
    const uint32_t leafBase = S.leafBase, slotMask = S.slotMask, numLeaf = (uint32_t)S.numTris;
    const uint32_t keyMask = ~slotMask & 0x7fffffffu;
    // a lane is at an inner node (cur < leafBase), at a triangle of a leaf (leafBase <= cur < LANE_FINISHED), finished or idle
    const uint32_t LANE_FINISHED = leafBase + numLeaf;      // ray done, result not yet written; also the entry in the two bottom rows of the stack
    const uint32_t LANE_IDLE = LANE_FINISHED + 1;           // no ray
    for (int i = threadIdx.x; i < cacheSlots * MIPT_NODE_U4; i += blockDim.x) smem4[i] = nodes[i];
    // The node copy starts at LDS address 0 (the kernel has no other LDS object), so a node's byte offset in the tree
    // is also its LDS address in the copy.  Checked, not assumed: a build that places it elsewhere stops here.
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem4 != 0u) __builtin_trap();
    uint32_t* stack = (uint32_t*)(smem4 + (size_t)cacheSlots * MIPT_NODE_U4) + (size_t)waveInGroup * ldsCap * WAVE + lane;
    stack[0] = LANE_FINISHED; stack[WAVE] = LANE_FINISHED;          // never overwritten: a ray's own rows start at 2
    const uint32_t* stackBelow = stack - 2 * WAVE;                  // stackBelow + sp * WAVE = the lower of the two rows below the stack pointer
    __syncthreads();
    const int n = *d_count;
    // per-wave slice of the spill area
    const size_t waveId = (size_t)blockIdx.x * (blockDim.x / WAVE) + waveInGroup;
    uint32_t* spill = S.stackSpill ? S.stackSpill + waveId * (size_t)(S.stackDepth - ldsCap) * WAVE + lane : nullptr;
    auto at_node = [&](uint32_t c) { return c < leafBase; };
    auto at_leaf = [&](uint32_t c) { return c - leafBase < numLeaf; };
    uint32_t nodeCount = 0, triCount = 0, rayCount = 0, boxCount = 0;
#ifdef MIPT_ABLATE_LOAD
    uint32_t ablAcc = 0;
#endif
#ifdef MIPT_TRACE_PROFILE
    uint32_t pOuter = 0, pNodeIters = 0, pTriIters = 0, pNodePhases = 0, pRefills = 0, pFinish = 0, pFinishLanes = 0, pSlow = 0;
#endif
#ifdef MIPT_TRACE_CLOCK
    // shader-clock cycles this wave spends in node phases, in triangle phases and in the rest of a round
    unsigned long long cNode = 0, cTri = 0, cAll = __builtin_readcyclecounter();
#endif

#ifdef MIPT_TRACE_TAIL
    // constant-rate clock (100 MHz) at the wave's start and at the moment the queue had nothing more for it
    const unsigned long long wStart = wall_clock64();
    unsigned long long wDry = 0;
#endif
    int rangeNext = 0, rangeEnd = 0;        // rays reserved by this wave and not yet handed out
    bool queueDry = false, firstReservation = true;
    // per-lane traversal state
    uint32_t cur = LANE_IDLE;
    int sp = 2, rayIdx = -1, tag0 = 0, tag1 = 0;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 0);
    BoxRay br = {};
    float tcull = FLT_MAX;
    uint32_t cullEntry = CULL_NONE;        // entries above this one lie beyond the best hit: bits of tcull with the code bits all set
    Hit best; best.t = FLT_MAX; best.u = 0; best.v = 0; best.tri = -1;
    // result of a finished ray
    auto write_result = [&]() {
        bool found = best.tri >= 0;
        if (MODE == 0) {
            // the shading kernels do not use the distance: with tagInHit the ray's slot id travels in its place
            hit[rayIdx] = make_float4(tagInHit ? __int_as_float(tag0) : best.t, best.u, best.v, __int_as_float(found ? best.tri : -1));
        } else if (MODE == 1) {
            if (found) {       // occluded: no direct light at this hit (PathTracer.cu:421-428)
                if (tag1 == 0) pixState0[(size_t)tag0 * 4].x = 0.0f;
                else pathState[(size_t)(tag1 - 1) * slotsPerLevel + tag0].x = 0.0f;
            }
        } else {
            flags[rayIdx] = found ? 1 : 0;
        }
    };
    // The entry a lane goes on with when its node gave it nothing or its leaf is done: `upper` is the
    // row below the stack pointer, `lower` the one below that (both fetched ahead of the step).
    auto take = [&](uint32_t upper, uint32_t lower, uint32_t& entry, int& rows) {
        bool ok = (MODE != 0) || upper <= cullEntry;
        entry = ok ? upper : lower;
        rows = ok ? 1 : 2;
    };
    // The two rows below the stack pointer.  `lds`: they are in the LDS part for every lane of the wave (the
    // caller has checked), the common case.
    auto rows_below = [&](bool lds, uint32_t& upper, uint32_t& lower) {
        if (__builtin_expect(lds, 1)) { const uint32_t* st = stackBelow + sp * WAVE; lower = st[0]; upper = st[WAVE]; }
        else {
            upper = stack_load(stack, spill, ldsCap, sp - 1, LANE_FINISHED); lower = stack_load(stack, spill, ldsCap, sp - 2, LANE_FINISHED);
            asm volatile("" : "+v"(upper), "+v"(lower));      // keeps these loads apart from the LDS reads of the common case (merged, both would become flat loads)
        }
    };
    auto node_step = [&](bool lds) {
        ISA_MARK("node step begin");
        // The node slot as words: 3k .. 3k+2 = box of child k, 12 .. 15 = child codes.  The top of the tree is
        // read from the workgroup's copy in LDS, the rest from memory, into the same registers: the lanes of
        // either kind issue their four 16-byte loads under their own exec mask and the wave then waits once
        // for both kinds.  (Written as two branches, each kind would wait for its own loads before the other
        // kind issues; written as one load whose address decides, all lanes would go through the vector
        // memory pipeline.)
        u32x4 q0, q1, q2, q3;
#ifdef MIPT_ABLATE_LOAD
        // what three more 8-byte loads of the node's own cache line cost (lanes that read the node from memory only)
        unsigned long long ablX0 = 0, ablX1 = 0, ablX2 = 0;
#endif
        {
            // byte offset of the node slot: the same number in the tree (32 bits: the builder keeps the node array below 4 GiB)
            // and in the copy, which starts at the workgroup's LDS base
            const uint32_t nodeOff = cur * (uint32_t)(MIPT_NODE_U4 * sizeof(uint4));
            const uint32_t ldsAddr = nodeOff;
            unsigned long long saved;
            asm volatile(
                "v_cmp_gt_u32_e32 vcc, %[cs], %[cur]\n\t"
                "s_and_saveexec_b64 %[sv], vcc\n\t"
                "ds_read_b128 %[q0], %[la]\n\t"
                "ds_read_b128 %[q1], %[la] offset:16\n\t"
                "ds_read_b128 %[q2], %[la] offset:32\n\t"
                "ds_read_b128 %[q3], %[la] offset:48\n\t"
                "s_andn2_b64 exec, %[sv], vcc\n\t"
                "global_load_dwordx4 %[q0], %[go], %[gb]\n\t"
                "global_load_dwordx4 %[q1], %[go], %[gb] offset:16\n\t"
                "global_load_dwordx4 %[q2], %[go], %[gb] offset:32\n\t"
                "global_load_dwordx4 %[q3], %[go], %[gb] offset:48\n\t"
#ifdef MIPT_ABLATE_LOAD
                "global_load_dwordx2 %[x0], %[go], %[gb] offset:8\n\t"
                "global_load_dwordx2 %[x1], %[go], %[gb] offset:24\n\t"
                "global_load_dwordx2 %[x2], %[go], %[gb] offset:40\n\t"
#endif
                "s_mov_b64 exec, %[sv]\n\t"
                "s_waitcnt vmcnt(0) lgkmcnt(0)"
                : [q0] "=&v"(q0), [q1] "=&v"(q1), [q2] "=&v"(q2), [q3] "=&v"(q3), [sv] "=&s"(saved)
#ifdef MIPT_ABLATE_LOAD
                  , [x0] "=&v"(ablX0), [x1] "=&v"(ablX1), [x2] "=&v"(ablX2)
#endif
                : [cs] "s"((uint32_t)cacheSlots), [cur] "v"(cur), [la] "v"(ldsAddr), [go] "v"(nodeOff), [gb] "s"(nodes)
                : "vcc", "memory");
        }
#ifdef MIPT_ABLATE_LOAD
        ablAcc ^= (uint32_t)(ablX0 ^ ablX1 ^ ablX2);
#endif
        // What a group of the node step's own instructions costs: the same number of them once more, independent
        // of each other, on registers nothing reads.  (Taking them out would change the result; adding them does not.)
#ifdef MIPT_ABLATE_ALIGNBIT      // the 12 near/far rotations of box_test (of its -DMIPT_ROTATE_PLANES form)
        {
            uint32_t x0, x1, x2, x3;
            asm volatile("v_alignbit_b32 %0, %4, %4, 16\n\tv_alignbit_b32 %1, %5, %5, 16\n\tv_alignbit_b32 %2, %6, %6, 16\n\tv_alignbit_b32 %3, %7, %7, 16\n\t"
                         "v_alignbit_b32 %0, %5, %5, 15\n\tv_alignbit_b32 %1, %6, %6, 15\n\tv_alignbit_b32 %2, %7, %7, 15\n\tv_alignbit_b32 %3, %4, %4, 15\n\t"
                         "v_alignbit_b32 %0, %6, %6, 14\n\tv_alignbit_b32 %1, %7, %7, 14\n\tv_alignbit_b32 %2, %4, %4, 14\n\tv_alignbit_b32 %3, %5, %5, 14"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(q0.x), "v"(q0.y), "v"(q0.z), "v"(q0.w));
        }
#endif
#ifdef MIPT_ABLATE_CVT           // the 24 conversions of 16-bit planes to float (of the -DMIPT_ROTATE_PLANES form; the byte permutes that replaced them cost the same each)
        {
            float x0, x1, x2, x3;
            #pragma unroll
            for (int r = 0; r < 3; r++)
                asm volatile("v_cvt_f32_u32_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\tv_cvt_f32_u32_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
                             "v_cvt_f32_u32_sdwa %2, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\tv_cvt_f32_u32_sdwa %3, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
                             "v_cvt_f32_u32_sdwa %0, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\tv_cvt_f32_u32_sdwa %1, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
                             "v_cvt_f32_u32_sdwa %2, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\tv_cvt_f32_u32_sdwa %3, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
                             : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(q1.x), "v"(q1.y), "v"(q1.z), "v"(q1.w));
        }
#endif
#ifdef MIPT_ABLATE_FMA           // 24 multiply-adds (the plane distances)
        {
            float x0, x1, x2, x3;
            #pragma unroll
            for (int r = 0; r < 6; r++)
                asm volatile("v_fma_f32 %0, %4, %5, %6\n\tv_fma_f32 %1, %5, %6, %7\n\tv_fma_f32 %2, %6, %7, %4\n\tv_fma_f32 %3, %7, %4, %5"
                             : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(q2.x), "v"(q2.y), "v"(q2.z), "v"(q2.w));
        }
#endif
        const u32x4 q4[MIPT_NODE_U4] = { q0, q1, q2, q3 };
        auto word = [&](int i) -> uint32_t { const u32x4& v = q4[i >> 2]; return (i & 3) == 0 ? v.x : (i & 3) == 1 ? v.y : (i & 3) == 2 ? v.z : v.w; };
        nodeCount++;
        if (COUNT_BOXES) {
            #pragma unroll
            for (int k = 0; k < MIPT_NODE_WIDTH; k++) boxCount += word(3 * MIPT_NODE_WIDTH + k) != MIPT_NO_NODE;
        }
        // one entry per child, ENTRY_MISS where the ray misses the child's box
        uint32_t e[MIPT_NODE_WIDTH];
        unsigned long long hitLanes[MIPT_NODE_WIDTH];
        #pragma unroll
        for (int k = 0; k < MIPT_NODE_WIDTH; k++)
            e[k] = box_entry<MODE>(br, word(3 * k), word(3 * k + 1), word(3 * k + 2), tcull, word(3 * MIPT_NODE_WIDTH + k), keyMask, hitLanes[k]);
        // a network of compare-exchanges sorts the entries, largest (farthest) first
        order_pair(e[0], e[1]); order_pair(e[2], e[3]);
        order_pair(e[0], e[2]); order_pair(e[1], e[3]);
        order_pair(e[1], e[2]);
        // all four go on the stack as they stand: the children that were hit end up below the new stack pointer with
        // the nearest one on top, the misses above it
        if (__builtin_expect(lds, 1)) { uint32_t* st = stack + sp * WAVE; st[0] = e[0]; st[WAVE] = e[1]; st[2 * WAVE] = e[2]; st[3 * WAVE] = e[3]; }
        else { stack_store(stack, spill, ldsCap, sp, e[0]); stack_store(stack, spill, ldsCap, sp + 1, e[1]); stack_store(stack, spill, ldsCap, sp + 2, e[2]); stack_store(stack, spill, ldsCap, sp + 3, e[3]); }
        // the stack pointer rises by the number of children hit: four additions of 0 with the lane's bit of a hit mask as carry
        asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1\n\tv_addc_co_u32_e64 %0, vcc, 0, %0, %2\n\t"
            "v_addc_co_u32_e64 %0, vcc, 0, %0, %3\n\tv_addc_co_u32_e64 %0, vcc, 0, %0, %4"
            : "+v"(sp) : "s"(hitLanes[0]), "s"(hitLanes[1]), "s"(hitLanes[2]), "s"(hitLanes[3]) : "vcc");
        // and the lane goes on with one of the two rows below the new stack pointer: the nearest child if one was hit
        // (it cannot lie beyond the best hit: its box has just been tested against it), else an older entry
        uint32_t upper, lower, next; int rows;
        rows_below(lds, upper, lower);
        take(upper, lower, next, rows);
        sp -= rows;
        cur = next & slotMask;
        ISA_MARK("node step end");
    };
    // the exact triangle test on one triangle of the lane's leaf
    auto tri_step = [&](bool lds) {
        uint32_t slot = cur - leafBase;
        const float4* tr = S.tris + (size_t)slot * MIPT_TRI_F4;
        float4 a = tr[0], b = tr[1], c = tr[2];
        uint32_t upper, lower;
        rows_below(lds, upper, lower);
        triCount++;
        float t, u, v;
        bool accept = ray_triangle_all(o, d, f3(a.x, a.y, a.z), f3(b.x, b.y, b.z), f3(c.x, c.y, c.z), eps, t, u, v);
        int32_t id = (int32_t)__float_as_uint(a.w);
        bool occluded = false;
        if (MODE != 0) {
            occluded = accept & (t < FLT_MAX);
            best.tri = occluded ? id : best.tri;
        } else {
            bool better = accept & ((t < best.t) | ((t == best.t) & (id < best.tri)));
            best.t = better ? t : best.t; best.u = better ? u : best.u; best.v = better ? v : best.v; best.tri = better ? id : best.tri;
            tcull = better ? t * 1.0000153f : tcull;       // boxes are culled a little beyond the best hit
            cullEntry = __float_as_uint(tcull) | slotMask;
        }
        bool last = (__float_as_uint(b.w) != 0) | (slot + 1 >= numLeaf);
        uint32_t old; int rows;
        take(upper, lower, old, rows);
        sp -= last ? rows : 0;
        uint32_t next = last ? (old & slotMask) : cur + 1;
        cur = occluded ? LANE_FINISHED : next;
    };

#ifndef MIPT_NO_COOP_LEAF
    // A triangle step in which the lanes that have no leaf help the lanes that have one.  The k-th lane
    // without a leaf takes the SECOND triangle of the leaf of the k-th lane with one (the ray travels through
    // ds_bpermute), tests it at the same time as the owner tests the first, and the owner takes the result over if
    // its own triangle was not the last of the leaf.  Run by ALL lanes of the wave (the lane exchange needs every lane).
    auto shfl_u = [&](int srcLane, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute(srcLane << 2, (int)v); };
    auto shfl_f = [&](int srcLane, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(srcLane << 2, __float_as_int(v))); };
    auto tri_step_coop = [&](bool lds, unsigned long long om) {      // om: the lanes at a leaf
        ISA_MARK("triangle step begin");
        const bool owner = __builtin_amdgcn_inverse_ballot_w64(om);
        const int T = __popcll(om), H = WAVE - T;
        const int orank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0u));
        const int hrank = lane - orank;                       // lanes without a leaf before this one
        // tab: the lanes in the order "owners first": lane p of tab holds the id of the p-th owner, lane T + h the h-th helper
        const int pos = owner ? orank : T + hrank;
        const int tab = __builtin_amdgcn_ds_permute(pos << 2, lane);
        const unsigned long long helperLanes = ~om & __ballot(hrank < T), helpedLanes = om & __ballot(orank < H);
        const bool helper = __builtin_amdgcn_inverse_ballot_w64(helperLanes);
        const int partner = __builtin_amdgcn_ds_bpermute((owner ? T + orank : hrank) << 2, tab);
        const int src = helper ? partner : lane;
        F3 po = f3(shfl_f(src, o.x), shfl_f(src, o.y), shfl_f(src, o.z)), pd = f3(shfl_f(src, d.x), shfl_f(src, d.y), shfl_f(src, d.z));
        const uint32_t pcur = shfl_u(src, cur);
        uint32_t slot = pcur - leafBase + (helper ? 1u : 0u);
        slot = min(slot, numLeaf - 1);                           // a helper's triangle may lie beyond the owner's leaf, even beyond the array
        // the record and the stack rows of a lane that takes no part are never looked at: their registers stay as they are
        float4 a, b, c;
        uint32_t upper, lower;
        asm("; not set" : "=v"(a.x), "=v"(a.y), "=v"(a.z), "=v"(a.w), "=v"(b.x), "=v"(b.y), "=v"(b.z), "=v"(b.w));
        asm("; not set" : "=v"(c.x), "=v"(c.y), "=v"(c.z), "=v"(c.w), "=v"(upper), "=v"(lower));
        if (__builtin_amdgcn_inverse_ballot_w64(om | helperLanes)) {
            const float4* tr = (const float4*)((const char*)S.tris + slot * (uint32_t)(MIPT_TRI_F4 * sizeof(float4)));      // the builder keeps the array below 4 GiB
            a = tr[0]; b = tr[1]; c = tr[2];
        }
        if (owner) rows_below(lds, upper, lower);
        float t, u, v;
        bool accept = ray_triangle_all(po, pd, f3(a.x, a.y, a.z), f3(b.x, b.y, b.z), f3(c.x, c.y, c.z), eps, t, u, v);
        if (MODE != 0) accept = accept & (t < FLT_MAX);
        const int32_t id = (int32_t)__float_as_uint(a.w);
        const uint32_t lastBits = __float_as_uint(b.w);         // 1 at the last triangle of a leaf (the last triangle of the array is one)
        // what the owner learns from its helper: (t, u, v) with t = infinity for a triangle that was not accepted, and the id
        // with the end-of-leaf flag above it
        const uint32_t packed = (uint32_t)id | (lastBits << 30);
        const float ht = shfl_f(partner, accept ? t : __builtin_inff()), hu = shfl_f(partner, u), hv = shfl_f(partner, v);
        const uint32_t hp = shfl_u(partner, packed);
        const unsigned long long lastLanes = __ballot(lastBits != 0);
        const unsigned long long use2Lanes = helpedLanes & ~lastLanes;       // owners whose own triangle was not the last and who have a helper
        const bool use2 = __builtin_amdgcn_inverse_ballot_w64(use2Lanes);
        const bool last2 = hp >= 0x40000000u;
        const int32_t id2 = (int32_t)(hp & 0x3fffffffu);
        asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1\n\tv_addc_co_u32_e64 %0, vcc, 0, %0, %2" : "+v"(triCount) : "s"(om), "s"(use2Lanes) : "vcc");
        bool occluded = false;
        if (MODE != 0) {
            const bool acc2 = use2 & (ht < __builtin_inff());
            occluded = owner & (accept | acc2);
            best.tri = (owner & accept) ? id : ((owner & acc2) ? id2 : best.tri);
        } else {
            bool better = owner & accept & ((t < best.t) | ((t == best.t) & (id < best.tri)));
            best.t = better ? t : best.t; best.u = better ? u : best.u; best.v = better ? v : best.v; best.tri = better ? id : best.tri;
            bool better2 = use2 & ((ht < best.t) | ((ht == best.t) & (id2 < best.tri)));
            best.t = better2 ? ht : best.t; best.u = better2 ? hu : best.u; best.v = better2 ? hv : best.v; best.tri = better2 ? id2 : best.tri;
            tcull = (better | better2) ? best.t * 1.0000153f : tcull;
            cullEntry = __float_as_uint(tcull) | slotMask;
        }
        const bool last = (lastBits != 0) | (use2 & last2);
        uint32_t old; int rows;
        take(upper, lower, old, rows);
        if (owner) {
            sp -= last ? rows : 0;
            uint32_t next = last ? (old & slotMask) : cur + (use2 ? 2u : 1u);
            cur = occluded ? LANE_FINISHED : next;
        }
        ISA_MARK("triangle step end");
    };
#endif

    for (;;) {
#ifdef MIPT_TRACE_PROFILE
        pOuter++;
#endif
        // ---- hand rays to idle lanes
        unsigned long long idleMask = __ballot(cur == LANE_IDLE);
        if (idleMask && !(queueDry && rangeNext >= rangeEnd) && (__popcll(idleMask) >= refillMin || __ballot(cur < LANE_FINISHED) == 0)) {
            // A wave's first reservation is the chunk with the wave's own number, taken without an atomic (thousands of
            // waves start at the same moment, and atomics on one address complete at about 90 per microsecond); the
            // shared head counts the chunks behind those.
            auto reserve = [&]() {
                if (firstReservation) { firstReservation = false; return (int)waveId * chunk; }
                int b = 0;
                if (lane == 0) b = atomicAdd(head, chunk);
                return __builtin_amdgcn_readfirstlane(b) + (int)(gridDim.x * (blockDim.x / WAVE)) * chunk;
            };
            if (!BLOCKS) {
                if (rangeNext >= rangeEnd) {
                    int b = reserve();
                    if ((uint32_t)b >= (uint32_t)n) queueDry = true;      // the head runs past n by up to one chunk per wave and may wrap: compared unsigned
                    else { rangeNext = b; rangeEnd = (b + chunk < n) ? b + chunk : n; }
                }
            } else {
                // a reservation in a queue in blocks can be without a valid entry: a few more are tried at once
                for (int tries = 0; tries < 4 && rangeNext >= rangeEnd && !queueDry; tries++) {
                    int b = reserve();
                    if ((uint32_t)b >= (uint32_t)n) { queueDry = true; break; }
                    int end = ((b >> MIPT_QBLOCK_SHIFT) << MIPT_QBLOCK_SHIFT) + blockCount[b >> MIPT_QBLOCK_SHIFT];
                    rangeNext = b; rangeEnd = (b + chunk < end) ? b + chunk : end;
                }
            }
            int avail = rangeEnd - rangeNext;
            if (avail > 0) {
#ifdef MIPT_TRACE_PROFILE
                pRefills++;
#endif
                int prefix = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idleMask, 0u));
                if (cur == LANE_IDLE && prefix < avail) {
                    rayIdx = rangeNext + prefix;
                    float4 o4 = q.o[rayIdx], d4 = q.d[rayIdx];
                    o = f3(o4.x, o4.y, o4.z); d = f3(d4.x, d4.y, d4.z);
                    tag0 = __float_as_int(o4.w); tag1 = __float_as_int(d4.w);
                    br = make_box_ray(S, o, d);
                    tcull = FLT_MAX; cullEntry = CULL_NONE;
                    best.t = FLT_MAX; best.u = 0; best.v = 0; best.tri = -1;
                    sp = 2;
                    rayCount++;
                    cur = (S.rootCode == MIPT_NO_NODE) ? LANE_FINISHED : S.rootCode;   // empty scene: every ray misses
                }
                int want = __popcll(idleMask);
                rangeNext += (want < avail) ? want : avail;
            }
        }
#ifdef MIPT_TRACE_TAIL
        if (wDry == 0 && queueDry && rangeNext >= rangeEnd) wDry = wall_clock64();
#endif
        // where the lanes are: one subtraction, two compares (codes are below 2^30: a node's code minus leafBase is negative)
        auto lane_masks = [&](unsigned long long& leafLanes, unsigned long long& nodeLanes) {
            const uint32_t rel = cur - leafBase;
            leafLanes = __ballot(rel < numLeaf); nodeLanes = __ballot((int32_t)rel < 0);
        };
        unsigned long long triMask, nodeMask;
        lane_masks(triMask, nodeMask);
        // ---- node steps, in a tight loop, until `nodeDrop` of the lanes that entered have left
        auto nodePhase = [&]() {
            int stay = __popcll(nodeMask) - nodeDrop;
            if (stay < 0) stay = 0;
#ifdef MIPT_TRACE_PROFILE
            pNodePhases++;
#endif
            // the common case first: a loop of steps whose stack rows all lie in LDS, left for the general loop as soon
            // as a lane that is about to step could push beyond the LDS part (one step in 300 on the headline frame)
            // (the two ballots are combined as masks: a ballot of the combined condition would cost three more instructions)
            unsigned long long atNode = __ballot(at_node(cur));
            auto deep_lanes = [&]() { return atNode & __ballot(sp + 3 >= ldsCap); };
            bool stepped = false;       // a phase is at least one step
            if (__builtin_expect(deep_lanes() == 0, 1)) {
                bool again;
                stepped = true;
                do {
#ifdef MIPT_TRACE_PROFILE
                    pNodeIters++;
#endif
                    if (__builtin_amdgcn_inverse_ballot_w64(atNode)) node_step(true);
                    atNode = __ballot(at_node(cur));
                    // lanes at a node, or 0 if one of them is about to leave the LDS part: three scalar instructions
                    // (written as C++ the condition is rebuilt from two 64-bit truth values and costs five more)
                    int nodeLanes; unsigned long long deepNow;
                    asm("s_bcnt1_i32_b64 %0, %2\n\ts_and_b64 %1, %2, %3\n\ts_cselect_b32 %0, 0, %0"
                        : "=&s"(nodeLanes), "=&s"(deepNow) : "s"(atNode), "s"(__ballot(sp + 3 >= ldsCap)) : "scc");
                    again = nodeLanes > stay;
                } while (again);
            }
            const int left = stepped ? __popcll(atNode) : WAVE + 1;      // lanes still at a node; "all of them" if no step was taken yet
            if (__builtin_expect(left > stay, 0)) {
                do {
#ifdef MIPT_TRACE_PROFILE
                    pNodeIters++;
#endif
                    if (at_node(cur)) {
                        const bool lds = __ballot(sp + 3 >= ldsCap) == 0;
#ifdef MIPT_TRACE_PROFILE
                        pSlow += !lds;
#endif
                        node_step(lds);
                    }
                } while (__popcll(__ballot(at_node(cur))) > stay);
            }
        };
        // ---- triangle steps
        auto triPhase = [&]() {
            // further triangle steps follow at once while enough of the lanes that entered are still at a leaf
            const int triKeep = triStay > 0 ? (__popcll(triMask) * triStay) >> 3 : WAVE;      // triStay 0: one step
#ifndef MIPT_NO_COOP_LEAF
            // as in the node phase: a loop for the common case (the rows the owners read lie in LDS), left for the general one
            unsigned long long atLeaf = triMask;
            auto deep_lanes = [&]() { return atLeaf & __ballot(sp > ldsCap); };
            bool first = true;
            if (__builtin_expect(deep_lanes() == 0, 1)) {
                bool again;
                do {
#ifdef MIPT_TRACE_PROFILE
                    pTriIters++;
#endif
                    tri_step_coop(true, atLeaf);
                    first = false;
                    atLeaf = __ballot(at_leaf(cur));
                    int leafLanes; unsigned long long deepNow;
                    asm("s_bcnt1_i32_b64 %0, %2\n\ts_and_b64 %1, %2, %3\n\ts_cselect_b32 %0, 0, %0"
                        : "=&s"(leafLanes), "=&s"(deepNow) : "s"(atLeaf), "s"(__ballot(sp > ldsCap)) : "scc");
                    again = leafLanes > triKeep;
                } while (again);
            }
            const int left = first ? WAVE + 1 : __popcll(atLeaf);
            if (__builtin_expect(left > triKeep, 0)) {
                do {
#ifdef MIPT_TRACE_PROFILE
                    pTriIters++;
#endif
                    atLeaf = __ballot(at_leaf(cur));
                    tri_step_coop((atLeaf & __ballot(sp > ldsCap)) == 0, atLeaf);
                } while (__popcll(__ballot(at_leaf(cur))) > triKeep);
            }
#else
            do {
#ifdef MIPT_TRACE_PROFILE
                pTriIters++;
#endif
                if (at_leaf(cur)) {
                    tri_step(__ballot(sp > ldsCap) == 0);
                }
            } while (__popcll(__ballot(at_leaf(cur))) > triKeep);
#endif
        };
        // the triangle phase follows the node phase in the same round
        if (nodeMask != 0 && __popcll(triMask) < triVote) {
#ifdef MIPT_TRACE_CLOCK
            unsigned long long c0 = __builtin_readcyclecounter();
#endif
            nodePhase();
#ifdef MIPT_TRACE_CLOCK
            cNode += __builtin_readcyclecounter() - c0;
#endif
            lane_masks(triMask, nodeMask);
        }
        if (triMask != 0 && (nodeMask == 0 || __popcll(triMask) >= triVote)) {
#ifdef MIPT_TRACE_CLOCK
            unsigned long long c0 = __builtin_readcyclecounter();
#endif
            triPhase();
#ifdef MIPT_TRACE_CLOCK
            cTri += __builtin_readcyclecounter() - c0;
#endif
        }
#ifdef MIPT_TRACE_PROFILE
        { unsigned long long fm = __ballot(cur == LANE_FINISHED); if (fm) { pFinish++; pFinishLanes += __popcll(fm); } }
#endif
        if (cur == LANE_FINISHED) {
            write_result();
            cur = LANE_IDLE;
        }
        // the wave is done when the queue has nothing more for it and every lane is idle
        if (queueDry && rangeNext >= rangeEnd && __ballot(cur != LANE_IDLE) == 0) break;
    }
    unsigned long long rs = wave_sum_u32(rayCount), ns = wave_sum_u32(nodeCount), ts = wave_sum_u32(triCount);
#ifdef MIPT_TRACE_PROFILE
    if (lane == 0 && MODE == 0) {
        atomicAdd(&stats[ST_PROF + 0], (unsigned long long)pOuter); atomicAdd(&stats[ST_PROF + 1], (unsigned long long)pNodeIters);
        atomicAdd(&stats[ST_PROF + 2], (unsigned long long)pTriIters); atomicAdd(&stats[ST_PROF + 3], (unsigned long long)pNodePhases);
        atomicAdd(&stats[ST_PROF + 4], (unsigned long long)pRefills); atomicAdd(&stats[ST_PROF + 5], (unsigned long long)pFinish);
        atomicAdd(&stats[ST_PROF + 6], (unsigned long long)pFinishLanes); atomicAdd(&stats[ST_PROF + 7], (unsigned long long)pSlow);
    }
#endif
#ifdef MIPT_TRACE_CLOCK
    if (lane == 0 && MODE == 0) {
        atomicAdd(&stats[ST_PROF + 8], cNode); atomicAdd(&stats[ST_PROF + 9], cTri); atomicAdd(&stats[ST_PROF + 10], __builtin_readcyclecounter() - cAll);
    }
#endif
#ifdef MIPT_TRACE_TAIL
    if (lane == 0 && MODE == 0 && BLOCKS == false && tagInHit) {      // the closest-hit launches of the bounce levels
        const unsigned long long wExit = wall_clock64();
        atomicAdd(&stats[ST_PROF + 12], wExit - wStart); atomicAdd(&stats[ST_PROF + 13], wExit - (wDry ? wDry : wExit)); atomicAdd(&stats[ST_PROF + 14], 1ull);
    }
#endif
#ifdef MIPT_ABLATE_LOAD
    if (ablAcc == 0x12345u) stats[ST_PROF] = 1;      // keeps the extra loads alive
#endif
    if (COUNT_BOXES) { unsigned long long bs = wave_sum_u32(boxCount); if (lane == 0 && bs) atomicAdd(&stats[MODE == 0 ? ST_CLOSEST_BOXES : ST_SHADOW_BOXES], bs); }
    if (lane == 0 && rs) {
        atomicAdd(&stats[MODE == 0 ? ST_CLOSEST_RAYS : ST_SHADOW_RAYS], rs);
        atomicAdd(&stats[MODE == 0 ? ST_CLOSEST_NODES : ST_SHADOW_NODES], ns);
        atomicAdd(&stats[MODE == 0 ? ST_CLOSEST_TRIS : ST_SHADOW_TRIS], ts);
    }
}


// Phase thresholds, per kind of query (the any-hit kernel leaves a ray at its first accepted triangle and spends a
// larger share of its steps on triangles): [0] closest hit, [1] any hit.  Environment switches for sweeps:
// MIPT_TRACE_TRI_VOTE, _NODE_DROP, _TRI_STAY set both kinds, the same names with _SHADOW appended the any-hit kind alone.
struct TraceThresholds { int triVote, nodeDrop, triStay; };
static const TraceThresholds& trace_thresholds(bool anyHit)
{
    static TraceThresholds t[2];
    static std::once_flag once;
    std::call_once(once, [] {
        auto get = [](const char* name, bool shadow, int dflt) {
            char buf[64];
            snprintf(buf, sizeof(buf), "%s%s", name, shadow ? "_SHADOW" : "");
            const char* e = getenv(buf);
            if (!e && shadow) e = getenv(name);
            return e ? atoi(e) : dflt;
        };
        for (int k = 0; k < 2; k++) {
            t[k].triVote = get("MIPT_TRACE_TRI_VOTE", k == 1, k ? TRACE_TRI_VOTE_SHADOW : TRACE_TRI_VOTE);
            t[k].nodeDrop = get("MIPT_TRACE_NODE_DROP", k == 1, k ? TRACE_NODE_DROP_SHADOW : TRACE_NODE_DROP);
            int s = get("MIPT_TRACE_TRI_STAY", k == 1, k ? TRACE_TRI_STAY_SHADOW : TRACE_TRI_STAY);
            t[k].triStay = s < 0 ? 0 : (s > 8 ? 8 : s);
        }
    });
    return t[anyHit ? 1 : 0];
}
static int trace_tri_vote(bool anyHit = false) { return trace_thresholds(anyHit).triVote; }
static int trace_node_drop(bool anyHit = false) { return trace_thresholds(anyHit).nodeDrop; }
static int trace_tri_stay(bool anyHit = false) { return trace_thresholds(anyHit).triStay; }

// rays an any-hit wave reserves at a time: a reservation in the shadow queue of a bounce level ends at its
// block's last valid entry, so larger reservations lose fewer lanes to short tails (measured: 512 is 4 % faster
// than 256 for the any-hit kernel and no different for the closest-hit kernel)
static int trace_chunk_shadow()
{
    static const int v = [] { const char* e = getenv("MIPT_TRACE_CHUNK_SHADOW"); int x = e ? atoi(e) : TRACE_CHUNK_SHADOW; return x < 64 ? 64 : x; }();
    return v;
}

static int trace_refill_min()
{
    static const int v = [] { const char* e = getenv("MIPT_TRACE_REFILL_MIN"); int x = e ? atoi(e) : TRACE_REFILL_MIN; return x < 1 ? 1 : (x > 64 ? 64 : x); }();
    return v;
}

static int trace_chunk()
{
    static const int v = [] { const char* e = getenv("MIPT_TRACE_CHUNK"); int x = e ? atoi(e) : TRACE_CHUNK; return x < 64 ? 64 : x; }();
    return v;
}

// Shape of a traversal launch: waves per workgroup, node slots copied to LDS, LDS bytes per workgroup,
// workgroups in the grid.  The grid is sized to the machine (the waves are persistent): as many
// workgroups per compute unit as its 160 KiB of LDS and its 32 wave slots admit.
struct TraceShape { int waves; int cacheSlots; size_t smem; int grid; };

static int env_int(const char* name, int dflt)
{
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

static TraceShape trace_shape(const DevScene& S, bool closest, int32_t maxCount, int numCUs, int64_t numNodeSlots)
{
    static int wavesEnv = TRACE_WAVES, cacheEnv = TRACE_CACHE_SLOTS, extra = 0;
    static std::once_flag once;
    std::call_once(once, [] {
        wavesEnv = env_int("MIPT_TRACE_WAVES", TRACE_WAVES); if (wavesEnv < 1) wavesEnv = 1; if (wavesEnv > 16) wavesEnv = 16;
        cacheEnv = env_int("MIPT_TRACE_CACHE", TRACE_CACHE_SLOTS); if (cacheEnv < 0) cacheEnv = 0;
        extra = env_int("MIPT_TRACE_LDS_EXTRA", 0);      // additional bytes per workgroup, to study occupancy
    });
    TraceShape t;
    t.waves = wavesEnv;
    t.cacheSlots = (int)(numNodeSlots < cacheEnv ? numNodeSlots : cacheEnv);
    const size_t perWave = (size_t)S.stackLds * WAVE * sizeof(uint32_t);
    const size_t slotBytes = 16 * MIPT_NODE_U4;
    t.smem = (size_t)t.cacheSlots * slotBytes + perWave * (size_t)t.waves + (size_t)extra;
    while (t.smem > 160 * 1024 && t.cacheSlots > 0) { t.cacheSlots /= 2; t.smem = (size_t)t.cacheSlots * slotBytes + perWave * (size_t)t.waves + (size_t)extra; }
    int perCU = t.smem ? (int)((160 * 1024) / t.smem) : 32;
    if (perCU * t.waves > 32) perCU = 32 / t.waves;
    if (perCU < 1) perCU = 1;
    long long resident = (long long)numCUs * perCU;
    const int chunk = closest ? trace_chunk() : trace_chunk_shadow();
    long long neededWaves = ((long long)maxCount + chunk - 1) / chunk;
    long long needed = (neededWaves + t.waves - 1) / t.waves;
    long long g = needed < resident ? needed : resident;
    t.grid = (int)(g < 1 ? 1 : g);
    return t;
}

// The workgroups ask for more than 64 KiB of LDS: said once per kernel and device, before the first launch
// (two threads that launch their first traversal at the same time both say it, which is harmless).
template <typename K> static void allow_large_lds(K kernel)
{
    static std::atomic<uint32_t> done{ 0 };
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return;
    if (done.load(std::memory_order_acquire) & (1u << dev)) return;
    // the kernel takes a node's byte offset in the tree for its LDS address in the copy: the copy must be the kernel's
    // only LDS object (the kernel checks the address itself and stops if it is not 0; here the cause is named)
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel)) == hipSuccess && fa.sharedSizeBytes != 0)
        fprintf(stderr, "mipt: the traversal kernel was built with %zu bytes of static LDS in front of its node copy; its launches will stop\n", (size_t)fa.sharedSizeBytes);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess)      // the launch that follows fails with a launch error whose cause would otherwise be invisible
        fprintf(stderr, "mipt: the traversal kernel was refused its LDS size (hipFuncSetAttribute: %s); its launches will fail\n", hipGetErrorString(e));
    done.fetch_or(1u << dev, std::memory_order_release);
}

void launch_closest(const DevScene& S, RayQueue q, const int32_t* d_count, int32_t maxCount, int32_t* d_head,
                    float4* hit, float eps, unsigned long long* d_stats, int numCUs, hipStream_t s, bool tagInHit, bool countBoxes)
{
    TraceShape t = trace_shape(S, true, maxCount, numCUs, S.numNodeSlots);
    allow_large_lds(k_trace<0, false, true>); allow_large_lds(k_trace<0, false, false>);
    if (countBoxes)     // the same kernel with one more counter: child boxes that exist in the nodes it visits
        hipLaunchKernelGGL((k_trace<0, false, true>), dim3(t.grid), dim3(t.waves * WAVE), t.smem, s, S, q, d_count, (const int32_t*)nullptr, d_head, hit,
                           (float4*)nullptr, (float4*)nullptr, (int64_t)0, (uint8_t*)nullptr, eps, d_stats, trace_tri_vote(), trace_chunk(), trace_node_drop(), trace_tri_stay(), (tagInHit ? 1 : 0) | (trace_refill_min() << 8), t.cacheSlots);
    else
        hipLaunchKernelGGL((k_trace<0, false>), dim3(t.grid), dim3(t.waves * WAVE), t.smem, s, S, q, d_count, (const int32_t*)nullptr, d_head, hit,
                           (float4*)nullptr, (float4*)nullptr, (int64_t)0, (uint8_t*)nullptr, eps, d_stats, trace_tri_vote(), trace_chunk(), trace_node_drop(), trace_tri_stay(), (tagInHit ? 1 : 0) | (trace_refill_min() << 8), t.cacheSlots);
}
void launch_shadow(const DevScene& S, RayQueue q, const int32_t* d_count, const int32_t* d_blockCount, int32_t maxCount, int32_t* d_head,
                   float4* pixState0, float4* pathState, int64_t slotsPerLevel, float eps,
                   unsigned long long* d_stats, int numCUs, hipStream_t s, bool countBoxes)
{
    TraceShape t = trace_shape(S, false, maxCount, numCUs, S.numNodeSlots);
    int chunk = trace_chunk_shadow();
    if (d_blockCount && (chunk > (1 << MIPT_QBLOCK_SHIFT) || ((1 << MIPT_QBLOCK_SHIFT) % chunk) != 0)) chunk = TRACE_CHUNK_SHADOW;   // must divide a block
    allow_large_lds(k_trace<1, true, false>); allow_large_lds(k_trace<1, false, false>);
    allow_large_lds(k_trace<1, true, true>); allow_large_lds(k_trace<1, false, true>);
    const int tagArg = trace_refill_min() << 8;
#define SHADOW_LAUNCH(BLOCKS, COUNT) hipLaunchKernelGGL((k_trace<1, BLOCKS, COUNT>), dim3(t.grid), dim3(t.waves * WAVE), t.smem, s, S, q, d_count, d_blockCount, d_head, \
                           (float4*)nullptr, pixState0, pathState, slotsPerLevel, (uint8_t*)nullptr, eps, d_stats, trace_tri_vote(true), chunk, trace_node_drop(true), trace_tri_stay(true), tagArg, t.cacheSlots)
    if (d_blockCount) { if (countBoxes) SHADOW_LAUNCH(true, true); else SHADOW_LAUNCH(true, false); }
    else { if (countBoxes) SHADOW_LAUNCH(false, true); else SHADOW_LAUNCH(false, false); }
#undef SHADOW_LAUNCH
}
void launch_anyhit_flags(const DevScene& S, RayQueue q, const int32_t* d_count, int32_t maxCount, int32_t* d_head,
                         uint8_t* out, float eps, unsigned long long* d_stats, int numCUs, hipStream_t s)
{
    TraceShape t = trace_shape(S, false, maxCount, numCUs, S.numNodeSlots);
    allow_large_lds(k_trace<2, false, false>);
    hipLaunchKernelGGL((k_trace<2, false>), dim3(t.grid), dim3(t.waves * WAVE), t.smem, s, S, q, d_count, (const int32_t*)nullptr, d_head,
                       (float4*)nullptr, (float4*)nullptr, (float4*)nullptr, (int64_t)0, out, eps, d_stats, trace_tri_vote(true), trace_chunk_shadow(), trace_node_drop(true), trace_tri_stay(true), trace_refill_min() << 8, t.cacheSlots);
}

// ---------------------------------------------------------------------------------------
// Shading record of a triangle: MIPT_SHADE_F4 float4 = one aligned 128-byte line, so that a hit
// gathers a single line and needs no second, dependent fetch of its material: the three positions,
// the flat normal of CMesh::Intersect (CPUPathTracer.cpp:236-243: normalised mean of the corner
// normals, the same for every hit of the triangle), the corner texture coordinates and a copy of
// the material's texture descriptors.
//   [0] = (p0, bits material)  [1] = (p1, N.x)  [2] = (p2, N.y)  [3] = (t0.x, t0.y, t1.x, t1.y)  [4] = (t2.x, t2.y, N.z, 0)
//   [5] = bits (first texel low, high, width | format << 30, height) of the colour texture
//   [6] = (diffuse colour, 0)
//   [7] = the same four words as [5] for the normal map
// attr holds the corners as the caller gave them: 2 float4 each = (pos3, normal3, tex2).
__global__ void __launch_bounds__(256) k_make_shading(const float4* __restrict__ attr, const int32_t* __restrict__ triMat,
                                                     const DevMaterial* __restrict__ mats, int64_t n, float4* shade)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* a = attr + (size_t)i * 6;
    float4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5];
    // corner k: pos = (a[2k].x, a[2k].y, a[2k].z), normal = (a[2k].w, a[2k+1].x, a[2k+1].y), tex = (a[2k+1].z, a[2k+1].w)
    F3 nn;
    nn.x = (a0.w + a2.w + a4.w) / 3;
    nn.y = (a1.x + a3.x + a5.x) / 3;
    nn.z = (a1.y + a3.y + a5.y) / 3;
    F3 N = f3normalize(nn);
    int mi = triMat[i];
    DevMaterial M = {};
    if (mi >= 0) M = mats[mi];
    float4* o = shade + (size_t)i * MIPT_SHADE_F4;
    o[0] = make_float4(a0.x, a0.y, a0.z, __int_as_float(mi));
    o[1] = make_float4(a2.x, a2.y, a2.z, N.x);
    o[2] = make_float4(a4.x, a4.y, a4.z, N.y);
    o[3] = make_float4(a1.z, a1.w, a3.z, a3.w);
    o[4] = make_float4(a5.z, a5.w, N.z, 0.0f);
    o[5] = make_float4(__uint_as_float((uint32_t)M.texOffset), __uint_as_float((uint32_t)(M.texOffset >> 32)),
                       __uint_as_float((uint32_t)M.texW | ((uint32_t)M.texFmt << 30)), __int_as_float(M.texH));
    o[6] = make_float4(M.diffuse[0], M.diffuse[1], M.diffuse[2], 0.0f);
    o[7] = make_float4(__uint_as_float((uint32_t)M.nmlOffset), __uint_as_float((uint32_t)(M.nmlOffset >> 32)),
                       __uint_as_float((uint32_t)M.nmlW | ((uint32_t)M.nmlFmt << 30)), __int_as_float(M.nmlH));
}
void launch_make_shading(const float4* attr, const int32_t* triMat, const DevMaterial* mats, int64_t n, float4* shade, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_make_shading, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, attr, triMat, mats, n, shade);
}

// ---------------------------------------------------------------------------------------
// Surface at a hit: attributes of CMesh::Intersect (CPUPathTracer.cpp:229-245), texture and
// normal-map lookup of Trace (:286-341), shading frame and normal map of Radiance (:389-400).
struct Surface {
    F3 hp, N, Nt, Nb, albedo;
};

__device__ __forceinline__ float wrap01(float x)
{
    // the +-1 loops of CPUPathTracer.cpp:292-295; values they cannot bring into range are set to 0
    if (!(fabsf(x) < 16777216.0f)) x = 0;
    while (x > 1) { x = x - 1; }
    while (x < 0) { x = x + 1; }
    return x;
}

// The three channel values of a texel as floats, .x = blue.  A byte texel gives (float)byte, which is
// what both flavours of the reference compute with: the CPU path converts the bytes of a DWORD
// (CPUPathTracer.cpp:307-309, 331-334), the GPU path reads byte values stored as floats
// (PathTracer.cu:311-313, 344-346).  A texture of the float pool holds other values than bytes
// and is used as it is, as PathTracer.cu does.
__device__ __forceinline__ F3 fetch_texel_bgr(const DevScene& S, int fmt, uint64_t off, long idx)
{
    if (fmt == MIPT_TEXFMT_U8) {
        uint32_t dw = S.texels[off + (uint64_t)idx];
        return f3((float)(dw & 255u), (float)((dw >> 8) & 255u), (float)((dw >> 16) & 255u));
    }
    const float* f = S.texelsF + off + (uint64_t)idx * 3;
    return f3(f[0], f[1], f[2]);
}

// the part of a shading record every hit needs, fetched ahead of its use by bounce shading
struct ShadeRec { float4 r0, r1, r2, r3, r4, tex; const float4* a; };

__device__ __forceinline__ ShadeRec load_shade_rec(const DevScene& S, const DevRender& R, int tri)
{
    ShadeRec c;
    c.a = S.shade + (size_t)tri * MIPT_SHADE_F4;
    c.r0 = c.a[0]; c.r1 = c.a[1]; c.r2 = c.a[2]; c.r3 = c.a[3]; c.r4 = c.a[4];
    c.tex = R.useTextures ? c.a[5] : make_float4(0, 0, 0, 0);
    return c;
}

__device__ __forceinline__ Surface surface_at(const DevScene& S, const DevRender& R, const ShadeRec& c, float ua, float va)
{
    Surface sf;
    const float4* a = c.a;
    const float4 r0 = c.r0, r1 = c.r1, r2 = c.r2, r3 = c.r3, r4 = c.r4;
    float w = 1 - ua - va;
    float u = w * r3.y + ua * r3.w + va * r4.y;          // :231 uses tex.y
    float v = w * r3.x + ua * r3.z + va * r4.x;          // :232 uses tex.x
    sf.hp.x = w * r0.x + ua * r1.x + va * r2.x;
    sf.hp.y = w * r0.y + ua * r1.y + va * r2.y;
    sf.hp.z = w * r0.z + ua * r1.z + va * r2.z;
    sf.N = f3(r1.w, r2.w, r4.z);                        // normalised mean of the corner normals, :236-243, made once per triangle

    F3 nmlMap = f3(0, 1, 0);
    long texW = 0, texH = 0, nmlW = 0, nmlH = 0;
    int texFmt = MIPT_TEXFMT_NONE, nmlFmt = MIPT_TEXFMT_NONE;
    uint64_t texOff = 0, nmlOff = 0;
    if (R.useTextures) {
        float4 t = c.tex;
        texOff = (uint64_t)__float_as_uint(t.x) | ((uint64_t)__float_as_uint(t.y) << 32);
        texW = (long)(__float_as_uint(t.z) & 0x3fffffffu); texFmt = (int)(__float_as_uint(t.z) >> 30); texH = __float_as_int(t.w);
    }
    if (R.useNormals) {
        float4 t = a[7];
        nmlOff = (uint64_t)__float_as_uint(t.x) | ((uint64_t)__float_as_uint(t.y) << 32);
        nmlW = (long)(__float_as_uint(t.z) & 0x3fffffffu); nmlFmt = (int)(__float_as_uint(t.z) >> 30); nmlH = __float_as_int(t.w);
    }
    u = wrap01(u); v = wrap01(v);
    if (nmlFmt != MIPT_TEXFMT_NONE && (R.useNormals)) {      // the map only matters when it is applied
        long jn = (long)(nmlW * v), hn = (long)(nmlH * u);    // :303-306
        jn = jn < nmlW - 1 ? jn : nmlW - 1;
        hn = hn < nmlH - 1 ? hn : nmlH - 1;
        F3 tx = fetch_texel_bgr(S, nmlFmt, nmlOff, hn * nmlW + jn);
        float x = tx.x, y = tx.y, z = tx.z;
        nmlMap.x = 2 * z / 255.0f - 1;                        // :315-317
        nmlMap.y = (2 * x / 255.0f - 1);
        nmlMap.z = 2 * y / 255.0f - 1;
        if (R.normalizeNormalMap) nmlMap = f3normalize(nmlMap);
    }
    if (!R.useTextures) {
        sf.albedo = f3(0.5f, 0.5f, 0.5f);                     // :320-326
    } else if (texFmt != MIPT_TEXFMT_NONE && texW > 0) {
        long j = (long)(texW * v), h = (long)(texH * u);      // :296-299
        j = j < texW - 1 ? j : texW - 1;
        h = h < texH - 1 ? h : texH - 1;
        F3 tx = fetch_texel_bgr(S, texFmt, texOff, h * texW + j);
        sf.albedo = f3(tx.x / 255.0f, tx.y / 255.0f, tx.z / 255.0f);
    } else {
        float4 dc = a[6];
        sf.albedo = f3(dc.x, dc.y, dc.z);                     // :338
    }
    coordinate_system(sf.N, sf.Nt, sf.Nb);                    // :389
    if (R.useNormals) {                                       // :390-400
        F3 nn = to_world(nmlMap, sf.N, sf.Nt, sf.Nb);
        sf.N = f3normalize(nn);
        coordinate_system(sf.N, sf.Nt, sf.Nb);
    }
    return sf;
}

// direct term before the shadow test and the shadow ray, CPUPathTracer.cpp:373-375,402-409
__device__ __forceinline__ float direct_term(const DevRender& R, const Surface& sf, F3& so, F3& sd)
{
    F3 sun = f3(R.sun[0], R.sun[1], R.sun[2]);
    sd = f3normalize(f3sub(sun, sf.hp));
    so = f3add(sf.hp, f3scale(sf.N, PT_BIAS));
    float dp = f3dot(sf.N, sd);
    return R.sunIntensity * ((0.0f > dp) ? 0.0f : dp);
}

// The shadow ray only decides whether the direct term D is kept or replaced by +0
// (CPUPathTracer.cpp:402-409).  Where D is +0 already (the surface faces away from the sun) the
// answer changes nothing and the ray is not traced.
__device__ __forceinline__ bool shadow_matters(float D) { return __float_as_uint(D) != 0u; }

__device__ __forceinline__ bool pixel_owned(const DevRender& R, int p)
{
    if (R.shardCount <= 1) return true;
    int x = p % R.W, y = p / R.W;
    int tile = (y / R.tileSize) * R.tilesX + (x / R.tileSize);
    return (tile % R.shardCount) == R.shardRank;
}

// pixState layout (4 float4 per pixel):
//   [0] = (direct term D0, albedo.xyz)   [1] = (hitpoint.xyz, 1 if hit)
//   [2] = (N.xyz, Nt.x)                  [3] = (Nt.y, Nt.z, 0, 0)
__global__ void __launch_bounds__(256) k_shade_primary(DevScene S, DevRender R, const float4* __restrict__ hit,
                                                      float4* pixState, int32_t* triIdImage, RayQueue shadowQ,
                                                      int32_t* shadowCount, int32_t* actPix, int32_t* actCount,
                                                      unsigned long long* stats)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool inRange = i < primary_count(R);
    const int p = (inRange && R.ownPix) ? R.ownPix[i] : i;      // the pixel of primary ray i
    float4 h = inRange ? hit[i] : make_float4(0, 0, 0, __int_as_float(-1));
    int tri = __float_as_int(h.w);
    bool isHit = inRange && tri >= 0;
    F3 so = f3(0, 0, 0), sd = f3(0, 0, 0);
    bool wantShadow = false;
    if (inRange) triIdImage[p] = tri;
    if (isHit) {
        Surface sf = surface_at(S, R, load_shade_rec(S, R, tri), h.y, h.z);
        float D = direct_term(R, sf, so, sd);
        wantShadow = shadow_matters(D);
        pixState[(size_t)p * 4 + 0] = make_float4(D, sf.albedo.x, sf.albedo.y, sf.albedo.z);
        pixState[(size_t)p * 4 + 1] = make_float4(sf.hp.x, sf.hp.y, sf.hp.z, 1.0f);
        pixState[(size_t)p * 4 + 2] = make_float4(sf.N.x, sf.N.y, sf.N.z, sf.Nt.x);
        pixState[(size_t)p * 4 + 3] = make_float4(sf.Nt.y, sf.Nt.z, 0.0f, 0.0f);
    } else if (inRange) {
        pixState[(size_t)p * 4 + 0] = make_float4(0, 0, 0, 0);
        pixState[(size_t)p * 4 + 1] = make_float4(0, 0, 0, 0);
    }
    int si = wave_append(wantShadow, shadowCount);
    if (wantShadow) {
        shadowQ.o[si] = make_float4(so.x, so.y, so.z, __int_as_float(p));
        shadowQ.d[si] = make_float4(sd.x, sd.y, sd.z, __int_as_float(0));
    }
    unsigned long long culled = __ballot(isHit && !wantShadow);
    if (culled && lane_id() == 0) atomicAdd(&stats[ST_SHADOW_CULLED], (unsigned long long)__popcll(culled));
    bool active = isHit && R.gi && pixel_owned(R, p);
    int ai = wave_append(active, actCount);
    if (active) actPix[ai] = p;
}

void launch_shade_primary(const DevScene& S, const DevRender& R, const float4* hit, float4* pixState,
                          int32_t* triIdImage, RayQueue shadowQ, int32_t* d_shadowCount, int32_t* actPix,
                          int32_t* d_actCount, unsigned long long* d_stats, hipStream_t s)
{
    int n = R.ownPix ? R.numOwn : R.W * R.H;
    hipLaunchKernelGGL(k_shade_primary, dim3((n + 255) / 256), dim3(256), 0, s, S, R, hit, pixState, triIdImage, shadowQ,
                       d_shadowCount, actPix, d_actCount, d_stats);
}

// ---------------------------------------------------------------------------------------
// first-bounce rays, CPUPathTracer.cpp:417-428 at depth 0: slot = s * nAct + a.  A thread makes
// GEN_SAMPLES consecutive samples of one active pixel: the pixel's state is read once for all of
// them, and the 64 lanes of a wave write 64 consecutive slots of every sample.
#define GEN_SAMPLES 8
__global__ void __launch_bounds__(256) k_gen_first(DevRender R, const float4* __restrict__ pixState,
                                                  const int32_t* __restrict__ actPix, int32_t nAct, int32_t n0,
                                                  int32_t Sn, RayQueue q)
{
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nAct) return;
    int p = actPix[a];
    float4 s1 = pixState[(size_t)p * 4 + 1], s2 = pixState[(size_t)p * 4 + 2], s3 = pixState[(size_t)p * 4 + 3];
    F3 hp = f3(s1.x, s1.y, s1.z), N = f3(s2.x, s2.y, s2.z), Nt = f3(s2.w, s3.x, s3.y);
    F3 Nb = f3cross(N, Nt);
    const int sEnd = min((int)(blockIdx.y + 1) * GEN_SAMPLES, Sn);
    for (int s = blockIdx.y * GEN_SAMPLES; s < sEnd; s++) {
        int64_t slot = (int64_t)s * nAct + a;
        uint32_t n = (uint32_t)(n0 + s);
        float r1 = rng_u01(R.seed, (uint32_t)p, n, 0, 0), r2 = rng_u01(R.seed, (uint32_t)p, n, 0, 1);
        F3 sw = to_world(sample_hemisphere(r1, r2), N, Nt, Nb);
        F3 o = f3add(hp, f3scale(sw, PT_BIAS));
        store_stream(&q.o[slot], make_float4(o.x, o.y, o.z, __int_as_float((int)slot)));
        store_stream(&q.d[slot], make_float4(sw.x, sw.y, sw.z, 0.0f));
    }
}

void launch_gen_first_bounce(const DevRender& R, const float4* pixState, const int32_t* actPix, int32_t nAct,
                             int32_t n0, int32_t Sn, RayQueue q, hipStream_t s)
{
    if (nAct <= 0 || Sn <= 0) return;
    hipLaunchKernelGGL(k_gen_first, dim3((unsigned)((nAct + 255) / 256), (unsigned)((Sn + GEN_SAMPLES - 1) / GEN_SAMPLES)), dim3(256), 0, s,
                       R, pixState, actPix, nAct, n0, Sn, q);
}

// ---------------------------------------------------------------------------------------
// Bounce shading (Radiance at depth `level`, 1..3).  pathState[(level-1)*slots + slot] =
// (direct term, albedo.xyz), zero where the ray of this level missed; levels beyond the end of a path keep
// whatever an earlier pass left there (finite numbers), which k_resolve multiplies by that zero.
__global__ void __launch_bounds__(256) k_shade_bounce(DevScene S, DevRender R, RayQueue q, const float4* __restrict__ hit,
                                                     const int32_t* __restrict__ d_count, int32_t level,
                                                     const int32_t* __restrict__ actPix, int32_t nAct, int32_t n0,
                                                     float4* pathState, int64_t slotsPerLevel, RayQueue shadowQ,
                                                     int32_t* shadowBlockCount, RayQueue nextQ, int32_t* nextCount,
                                                     unsigned long long* stats)
{
    const int n = *d_count;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wavesPerBlock = blockDim.x / WAVE;
    const int64_t wave = (int64_t)blockIdx.x * wavesPerBlock + threadIdx.x / WAVE;
    const int64_t numWaves = (int64_t)gridDim.x * wavesPerBlock;
    const bool spawnLevel = level < 3;          // a sample drawn at depth 3 is not traced (PathTracer.cu:380)
    // a wave owns one block of SHADE_ITEMS * 64 consecutive queue entries at a time.  It first counts
    // its hits (one ballot per 64 entries) and reserves their places in the queue of the next level
    // with one atomic, then shades and writes.  Its shadow rays go to the same block of the shadow
    // queue, packed from the start of the block, and their number to shadowBlockCount: no atomic,
    // and every block of the input writes its count.
    uint32_t culledCount = 0, hitCount = 0;
    for (int64_t first = wave * (SHADE_ITEMS * WAVE); first < n; first += numWaves * (SHADE_ITEMS * WAVE)) {
        int total = 0;
        #pragma unroll
        for (int k = 0; k < SHADE_ITEMS; k++) {
            int64_t i = first + k * WAVE + lane;
            bool h = (i < n) && (__float_as_int(hit[i].w) >= 0);
            total += __popcll(__ballot(h));
        }
        int nbase = 0;
        if (total != 0 && spawnLevel) {
            if (lane == 0) nbase = atomicAdd(nextCount, total);
            nbase = __builtin_amdgcn_readfirstlane(nbase);
        }
        int running = 0, shadowRun = 0;
        // The chain hit record -> shading record -> texel is three dependent fetches, the last two of
        // them gathers from beyond L2, whose traffic bounds the kernel.  The hit record is fetched two
        // items ahead and the shading record one item ahead of the item that is being shaded, which
        // takes their latency out of the chain (-2 %).
        const float4 noHit = make_float4(0, 0, 0, __int_as_float(-1));
        auto load_hit = [&](int k) { int64_t i = first + k * WAVE + lane; return (k < SHADE_ITEMS && i < n) ? hit[i] : noHit; };
        float4 hNext = load_hit(0), hNext2 = load_hit(1);
        ShadeRec recNext = {};
        if (__float_as_int(hNext.w) >= 0) recNext = load_shade_rec(S, R, __float_as_int(hNext.w));
        for (int k = 0; k < SHADE_ITEMS; k++) {
            int64_t i = first + k * WAVE + lane;
            bool inRange = i < n;
            const float4 h = hNext;
            const ShadeRec rec = recNext;
            hNext = hNext2;
            hNext2 = load_hit(k + 2);
            if (__float_as_int(hNext.w) >= 0) recNext = load_shade_rec(S, R, __float_as_int(hNext.w));
            int tri = __float_as_int(h.w);
            bool isHit = inRange && tri >= 0;
            // a ray that left the scene ends its path: the state of this level is zero.  Every slot that
            // reaches a level is written here, hit or miss, so the buffer needs no clearing between passes
            // (levels a path did not reach are multiplied by the zero albedo of the level where it ended)
            if (inRange && !isHit) store_stream(&pathState[(size_t)(level - 1) * slotsPerLevel + __float_as_int(h.x)], make_float4(0, 0, 0, 0));
            unsigned long long mask = __ballot(isHit);
            if (mask == 0) continue;
            int prefix = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            int idx = running + prefix;
            running += __popcll(mask);
            bool wantShadow = false;
            int slot = 0;
            F3 so = f3(0, 0, 0), sd = f3(0, 0, 0);
            if (isHit) {
                slot = __float_as_int(h.x);               // the traversal put the ray's slot id where the distance would be
                Surface sf = surface_at(S, R, rec, h.y, h.z);
                float D = direct_term(R, sf, so, sd);
                wantShadow = shadow_matters(D);
                store_stream(&pathState[(size_t)(level - 1) * slotsPerLevel + slot], make_float4(D, sf.albedo.x, sf.albedo.y, sf.albedo.z));
                if (spawnLevel) {
                    int a = slot % nAct, s = slot / nAct;
                    uint32_t p = (uint32_t)actPix[a], smp = (uint32_t)(n0 + s);
                    float r1 = rng_u01(R.seed, p, smp, (uint32_t)level, 0), r2 = rng_u01(R.seed, p, smp, (uint32_t)level, 1);
                    F3 nd = to_world(sample_hemisphere(r1, r2), sf.N, sf.Nt, sf.Nb);
                    F3 no = f3add(sf.hp, f3scale(nd, PT_BIAS));
                    store_stream(&nextQ.o[nbase + idx], make_float4(no.x, no.y, no.z, __int_as_float(slot)));
                    store_stream(&nextQ.d[nbase + idx], make_float4(nd.x, nd.y, nd.z, 0.0f));
                }
            }
            unsigned long long smask = __ballot(wantShadow);
            if (wantShadow) {
                int64_t si = first + shadowRun + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(smask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)smask, 0u));
                store_stream(&shadowQ.o[si], make_float4(so.x, so.y, so.z, __int_as_float(slot)));
                store_stream(&shadowQ.d[si], make_float4(sd.x, sd.y, sd.z, __int_as_float(level)));
            }
            shadowRun += __popcll(smask);
        }
        if (lane == 0) shadowBlockCount[first >> MIPT_QBLOCK_SHIFT] = shadowRun;
        culledCount += (uint32_t)(total - shadowRun);
        hitCount += (uint32_t)total;
    }
    if (lane == 0 && culledCount) atomicAdd(&stats[ST_SHADOW_CULLED], (unsigned long long)culledCount);
    if (lane == 0 && hitCount) {
        atomicAdd(&stats[ST_SHADE_HITS], (unsigned long long)hitCount);
        if (spawnLevel) atomicAdd(&stats[ST_SHADE_NEXT], (unsigned long long)hitCount);
    }
}

void launch_shade_bounce(const DevScene& S, const DevRender& R, RayQueue q, const float4* hit,
                         const int32_t* d_count, int32_t maxCount, int32_t level, const int32_t* actPix,
                         int32_t nAct, int32_t n0, float4* pathState, int64_t slotsPerLevel, RayQueue shadowQ,
                         int32_t* d_shadowBlockCount, RayQueue nextQ, int32_t* d_nextCount,
                         unsigned long long* d_stats, hipStream_t s)
{
    static_assert(SHADE_ITEMS * WAVE == (1 << MIPT_QBLOCK_SHIFT), "a wave shades one queue block at a time");
    if (maxCount <= 0) return;
    const int perBlock = (256 / WAVE) * SHADE_ITEMS * WAVE;
    long long blocks = ((long long)maxCount + perBlock - 1) / perBlock;
    if (blocks > 256 * 32) blocks = 256 * 32;      // waves stride over the queue beyond this
    hipLaunchKernelGGL(k_shade_bounce, dim3((unsigned)blocks), dim3(256), 0, s, S, R, q, hit, d_count, level, actPix, nAct, n0,
                       pathState, slotsPerLevel, shadowQ, d_shadowBlockCount, nextQ, d_nextCount, d_stats);
}

// ---------------------------------------------------------------------------------------
// Backward resolve of one pass.  For sample n of pixel p, with c_k the first random number
// drawn at depth k and (D_k, A_k) the stored state:
//   L_4 = 0;  L_k = min(1, ((D_k + I * ((c_k * L_{k+1}) / pdf)) * A_k) / pi)   k = 3, 2, 1
// (PathTracer.cu:449-462 with N = 1), and the pixel's running sum gains (c_0 * L_1) / pdf
// in sample order (PathTracer.cu:449-451 at depth 0).
__device__ __forceinline__ float min1(float x) { return (x < 1.0f) ? x : 1.0f; }

__global__ void __launch_bounds__(256) k_resolve(DevRender R, const float4* __restrict__ pathState, int64_t slotsPerLevel,
                                                const int32_t* __restrict__ actPix, int32_t nAct, int32_t n0, int32_t Sn,
                                                float4* pixAccum)
{
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nAct) return;
    uint32_t p = (uint32_t)actPix[a];
    const float pdf = 1 / (2 * PT_PI);
    const float I = R.sunIntensity;
    float4 acc = pixAccum[a];
    for (int s = 0; s < Sn; s++) {
        uint32_t smp = (uint32_t)(n0 + s);
        size_t slot = (size_t)s * nAct + a;
        F3 L = f3(0, 0, 0);
        #pragma unroll
        for (int level = 3; level >= 1; level--) {
            float4 st = pathState[(size_t)(level - 1) * slotsPerLevel + slot];
            float c = rng_u01(R.seed, p, smp, (uint32_t)level, 0);
            F3 ind = f3(c * L.x / pdf, c * L.y / pdf, c * L.z / pdf);
            L.x = min1((st.x + I * ind.x) * st.y / PT_PI);
            L.y = min1((st.x + I * ind.y) * st.z / PT_PI);
            L.z = min1((st.x + I * ind.z) * st.w / PT_PI);
        }
        float c0 = rng_u01(R.seed, p, smp, 0, 0);
        acc.x = acc.x + c0 * L.x / pdf;
        acc.y = acc.y + c0 * L.y / pdf;
        acc.z = acc.z + c0 * L.z / pdf;
    }
    pixAccum[a] = acc;
}

void launch_resolve(const DevRender& R, const float4* pathState, int64_t slotsPerLevel, const int32_t* actPix,
                    int32_t nAct, int32_t n0, int32_t Sn, float4* pixAccum, hipStream_t s)
{
    if (nAct <= 0) return;
    hipLaunchKernelGGL(k_resolve, dim3((nAct + 255) / 256), dim3(256), 0, s, R, pathState, slotsPerLevel, actPix, nAct, n0, Sn, pixAccum);
}

__global__ void __launch_bounds__(256) k_scatter_accum(const float4* __restrict__ pixAccum, const int32_t* __restrict__ actPix,
                                                      int32_t nAct, float* accum)
{
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nAct) return;
    size_t p = (size_t)actPix[a];
    float4 v = pixAccum[a];
    accum[3 * p] = v.x; accum[3 * p + 1] = v.y; accum[3 * p + 2] = v.z;
}
void launch_scatter_accum(const float4* pixAccum, const int32_t* actPix, int32_t nAct, float* d_accum, hipStream_t s)
{
    if (nAct <= 0) return;
    hipLaunchKernelGGL(k_scatter_accum, dim3((nAct + 255) / 256), dim3(256), 0, s, pixAccum, actPix, nAct, d_accum);
}

// PathTracer.cu:453-462 and 550-552 at depth 0
__global__ void __launch_bounds__(256) k_finalize(DevRender R, const float4* __restrict__ pixState, const float* __restrict__ accum,
                                                 float* out)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= R.W * R.H) return;
    float4 s0 = pixState[(size_t)p * 4 + 0], s1 = pixState[(size_t)p * 4 + 1];
    F3 rgb = f3(0, 0, 0);
    if (s1.w != 0.0f) {
        F3 ind = f3(0, 0, 0);
        if (R.gi) {
            uint32_t N = (uint32_t)R.numSamples;
            ind = f3(accum[3 * (size_t)p] / N, accum[3 * (size_t)p + 1] / N, accum[3 * (size_t)p + 2] / N);
        }
        rgb.x = (s0.x + R.sunIntensity * ind.x) * s0.y / PT_PI;
        rgb.y = (s0.x + R.sunIntensity * ind.y) * s0.z / PT_PI;
        rgb.z = (s0.x + R.sunIntensity * ind.z) * s0.w / PT_PI;
    }
    out[3 * (size_t)p] = 255 * min1(rgb.x);
    out[3 * (size_t)p + 1] = 255 * min1(rgb.y);
    out[3 * (size_t)p + 2] = 255 * min1(rgb.z);
}
void launch_finalize(const DevRender& R, const float4* pixState, const float* d_accum, float* d_out, hipStream_t s)
{
    int n = R.W * R.H;
    hipLaunchKernelGGL(k_finalize, dim3((n + 255) / 256), dim3(256), 0, s, R, pixState, d_accum, d_out);
}
