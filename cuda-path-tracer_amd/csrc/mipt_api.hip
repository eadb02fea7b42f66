// C ABI of libmipt.so (include/mipt.h): scene upload + BVH build, work buffers, the
// wavefront render loop.  Replaces the host side of CCUDAPathTracer::CalcRays,
// PathTracer.cu:556-771.
#include "pt_internal.h"

#include <chrono>
#include <math.h>
#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <atomic>
#include <memory>
#include <thread>
#include <vector>

static thread_local std::string g_err;
static int g_device = -1;
static int g_numCUs = 256;

// Work memory of mipt_trace_rays_device, the entry point that allocates nothing: ray queue, hit
// records, queue counters and statistics of one call.  The array is part of the code object, so
// every GPU the library runs on holds its own copy, at its own address, until the process ends.
// In host code the name is only a handle for hipGetSymbolAddress: g_scratchPtr[device] is the
// address kernels and memsets are given.
#define SCRATCH_RAYS   MIPT_TRACE_DEVICE_MAX_RAYS
#define SCRATCH_Q_O    0
#define SCRATCH_Q_D    (SCRATCH_Q_O + 16 * (size_t)SCRATCH_RAYS)
#define SCRATCH_HIT    (SCRATCH_Q_D + 16 * (size_t)SCRATCH_RAYS)
#define SCRATCH_CTR    (SCRATCH_HIT + 16 * (size_t)SCRATCH_RAYS)      /* 4 x int32: count, closest head, any-hit head, unused */
#define SCRATCH_STATS  (SCRATCH_CTR + 256)
#define SCRATCH_BYTES  (SCRATCH_STATS + 256)
static_assert(8 * ST_COUNT <= 256, "statistics slots must fit their part of the scratch");
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];
#define MIPT_MAX_DEVICES 64
static unsigned char* g_scratchPtr[MIPT_MAX_DEVICES];

static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIPTRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(MIPT_E_DEVICE, "%s: %s", #x, hipGetErrorString(e_)); } while (0)

struct mipt_scene {
    DevScene dev = {};
    BvhResult bvh;
    float4* attr = nullptr;
    int32_t* triMat = nullptr;
    float4* shade = nullptr;
    DevMaterial* mats = nullptr;
    uint32_t* texels = nullptr;
    float* texelsF = nullptr;
    int64_t numTris = 0;
    int32_t numMats = 0;
    int64_t geometryBytes = 0, textureBytes = 0;
    double uploadMs = 0;
    uint32_t* stackSpill = nullptr;
    hipStream_t stream = nullptr;
};

struct mipt_frame {
    mipt_scene* scene = nullptr;
    mipt_camera cam = {};
    mipt_params prm = {};
    DevRender R = {};
    int32_t W = 0, H = 0, P = 0;
    int32_t chunk = 1;            // samples per pass
    int64_t slotCap = 0;          // path slots per pass
    hipStream_t stream = nullptr;
    // per pixel
    RayQueue primQ = {}; float4* primHit = nullptr; float4* pixState = nullptr; int32_t* triId = nullptr;
    int32_t* actPix = nullptr; float4* pixAccum = nullptr; RayQueue primShadowQ = {};
    // per path slot
    RayQueue q[2] = {}; RayQueue shadowQ = {}; int32_t* shadowBlocks = nullptr; float4* hit = nullptr; float4* pathState = nullptr;
    // counters
    int32_t* counters = nullptr; int32_t numCounters = 0;
    unsigned long long* stats = nullptr;
    std::vector<hipEvent_t> events;
    // with own_pixels_only: the pixels of this shard's tiles, ascending (made when the shard fields change)
    int32_t* ownPix = nullptr; int32_t numOwn = 0;
    int32_t ownRank = -1, ownCount = -1, ownTile = -1;
    int32_t lastActive = 0;
    int32_t samplesDone = 0;                  // first-bounce samples summed into d_accum by the last accumulate call
    std::atomic<int> cancel{ 0 };
};

extern "C" {

const char* mipt_last_error(void) { return g_err.c_str(); }

int mipt_init(int32_t device)
{
    int count = 0;
    HIPTRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(MIPT_E_INVALID, "device %d out of range (%d devices)", device, count);
    HIPTRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPTRY(hipGetDeviceProperties(&prop, device));
    g_numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (device < MIPT_MAX_DEVICES && !g_scratchPtr[device]) {
        // this device's copy of the scratch array (the device is current here)
        void* p = nullptr;
        HIPTRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        g_scratchPtr[device] = (unsigned char*)p;
    }
    g_device = device;
    return MIPT_OK;
}

int mipt_device_name(char* buf, int32_t buflen)
{
    if (!buf || buflen <= 0) return fail(MIPT_E_INVALID, "null buffer");
    if (g_device < 0) { int rc = mipt_init(0); if (rc) return rc; }
    hipDeviceProp_t prop;
    HIPTRY(hipGetDeviceProperties(&prop, g_device));
    snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return MIPT_OK;
}

static int ensure_init()
{
    if (g_device < 0) return mipt_init(0);
    HIPTRY(hipSetDevice(g_device));
    return MIPT_OK;
}

void mipt_scene_destroy(mipt_scene* s)
{
    if (!s) return;
    mipt_free_bvh(&s->bvh);
    if (s->attr) (void)hipFree(s->attr);
    if (s->triMat) (void)hipFree(s->triMat);
    if (s->shade) (void)hipFree(s->shade);
    if (s->mats) (void)hipFree(s->mats);
    if (s->texels) (void)hipFree(s->texels);
    if (s->texelsF) (void)hipFree(s->texelsF);
    if (s->stackSpill) (void)hipFree(s->stackSpill);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

// Packs a run of float3 texels into BGRA8 words; false if a value is not an integer in 0..255
// (then 4 bytes per texel would lose something and the texture stays float).
static bool pack_byte_texels(const float* f, size_t count, uint32_t* dst)
{
    for (size_t i = 0; i < count; i++) {
        float b = f[3 * i], g = f[3 * i + 1], r = f[3 * i + 2];
        // (-0.0f is no byte value either: -0/255 and +0/255 differ in their sign bit)
        if (!(b >= 0.0f && b <= 255.0f) || b != (float)(int)b || !(g >= 0.0f && g <= 255.0f) || g != (float)(int)g ||
            !(r >= 0.0f && r <= 255.0f) || r != (float)(int)r || signbit(b) || signbit(g) || signbit(r)) return false;
        dst[i] = (uint32_t)b | ((uint32_t)g << 8) | ((uint32_t)r << 16) | (255u << 24);
    }
    return true;
}

// One texture of the scene descriptor on its way into the texel pools.
struct TexJob {
    const void* data; int32_t fmt; size_t count; uint64_t off8;
    std::atomic<int> notBytes{ 0 };
    TexJob(const void* d, int32_t f, size_t c, uint64_t o) : data(d), fmt(f), count(c), off8(o) {}
};

// Fills pool8 from all jobs with a few host threads (the Crytek texture set is 67 M texels, 800 MB
// as float3: one thread takes a third of a second over it).
static void run_tex_jobs(std::vector<std::unique_ptr<TexJob>>& jobs, uint32_t* pool8)
{
    const size_t PIECE = 1u << 18;
    struct Piece { TexJob* job; size_t first, count; };
    std::vector<Piece> pieces;
    for (auto& j : jobs)
        for (size_t f = 0; f < j->count; f += PIECE) pieces.push_back({ j.get(), f, j->count - f < PIECE ? j->count - f : PIECE });
    std::atomic<size_t> next{ 0 };
    auto work = [&]() {
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= pieces.size()) return;
            const Piece& p = pieces[i];
            uint32_t* dst = pool8 + p.job->off8 + p.first;
            if (p.job->fmt == MIPT_TEX_BGRA8) memcpy(dst, (const uint32_t*)p.job->data + p.first, 4 * p.count);
            else if (!p.job->notBytes.load(std::memory_order_relaxed) && !pack_byte_texels((const float*)p.job->data + 3 * p.first, p.count, dst))
                p.job->notBytes.store(1, std::memory_order_relaxed);
        }
    };
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = hw ? hw : 4;
    if (nt > 16) nt = 16;
    if (nt > pieces.size()) nt = pieces.size();
    std::vector<std::thread> pool;
    for (size_t t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
}

int mipt_scene_create(const mipt_scene_desc* desc, const mipt_build_options* opt, mipt_scene** out)
{
    if (!desc || !out) return fail(MIPT_E_INVALID, "null argument");
    *out = nullptr;
    if (desc->num_triangles < 0 || desc->num_meshes < 0 || desc->num_materials < 0)
        return fail(MIPT_E_INVALID, "negative count in scene descriptor");
    if ((desc->num_triangles > 0 && !desc->vertices) || (desc->num_meshes > 0 && !desc->meshes) ||
        (desc->num_materials > 0 && !desc->materials))
        return fail(MIPT_E_INVALID, "null array in scene descriptor");
    int rc = ensure_init();
    if (rc) return rc;
    auto t0 = std::chrono::steady_clock::now();

    // global triangle order: non-light meshes, in mesh order
    int64_t n = 0;
    for (int32_t i = 0; i < desc->num_meshes; i++) {
        const mipt_mesh& m = desc->meshes[i];
        if (m.num_triangles < 0 || m.first_triangle < 0 || m.first_triangle + m.num_triangles > desc->num_triangles)
            return fail(MIPT_E_INVALID, "mesh %d addresses triangles outside the vertex array", i);
        if (m.material >= desc->num_materials) return fail(MIPT_E_INVALID, "mesh %d names material %d of %d", i, m.material, desc->num_materials);
        if (!m.is_light) n += m.num_triangles;
    }
    if (n >= (1ll << 30)) return fail(MIPT_E_LIMIT, "%lld triangles exceed the supported 2^30", (long long)n);

    std::vector<mipt_vertex> attr; attr.reserve((size_t)n * 3);
    std::vector<int32_t> triMat; triMat.reserve((size_t)n);
    for (int32_t i = 0; i < desc->num_meshes; i++) {
        const mipt_mesh& m = desc->meshes[i];
        if (m.is_light) continue;
        const mipt_vertex* v = desc->vertices + 3 * m.first_triangle;
        attr.insert(attr.end(), v, v + 3 * m.num_triangles);
        triMat.insert(triMat.end(), (size_t)m.num_triangles, m.material < 0 ? -1 : m.material);
    }

    // materials: byte-valued float texels are repacked to 4 bytes per texel (the reference
    // uploads 12, PathTracer.cu:599-618); anything else stays float
    std::vector<DevMaterial> mats((size_t)desc->num_materials);
    std::vector<float> poolF;
    std::vector<std::unique_ptr<TexJob>> jobs;
    struct Slot { uint64_t* off; int32_t* w; int32_t* h; int32_t* fmt; int job; };
    std::vector<Slot> slots;
    uint64_t pool8Count = 0;
    auto addTex = [&](const void* data, int32_t w, int32_t h, int32_t fmt, uint64_t& off, int32_t& dw, int32_t& dh, int32_t& dfmt) -> int {
        off = 0; dw = dh = 0; dfmt = MIPT_TEXFMT_NONE;
        if (!data || fmt == MIPT_TEX_NONE || w <= 0 || h <= 0) return MIPT_OK;
        if (fmt != MIPT_TEX_BGRA8 && fmt != MIPT_TEX_F32X3) return MIPT_E_INVALID;
        if (w >= (1 << 30)) return MIPT_E_INVALID;       // the shading records keep the format in the two top bits of the width
        size_t cnt = (size_t)w * (size_t)h;
        slots.push_back({ &off, &dw, &dh, &dfmt, (int)jobs.size() });
        jobs.emplace_back(new TexJob(data, fmt, cnt, pool8Count));
        off = pool8Count; dw = w; dh = h; dfmt = MIPT_TEXFMT_U8;
        pool8Count += cnt;
        return MIPT_OK;
    };
    for (int32_t i = 0; i < desc->num_materials; i++) {
        const mipt_material& m = desc->materials[i];
        DevMaterial& d = mats[(size_t)i];
        memset(&d, 0, sizeof(d));
        if (addTex(m.tex_data, m.tex_width, m.tex_height, m.tex_format, d.texOffset, d.texW, d.texH, d.texFmt) ||
            addTex(m.nml_data, m.nml_width, m.nml_height, m.nml_format, d.nmlOffset, d.nmlW, d.nmlH, d.nmlFmt))
            return fail(MIPT_E_INVALID, "material %d has an unknown texel format or a texture wider than 2^30", i);
        d.diffuse[0] = m.diffuse[0]; d.diffuse[1] = m.diffuse[1]; d.diffuse[2] = m.diffuse[2];
    }
    std::unique_ptr<uint32_t[]> pool8(pool8Count ? new uint32_t[pool8Count] : nullptr);
    if (pool8Count) run_tex_jobs(jobs, pool8.get());
    for (const Slot& sl : slots) {          // a texture with other values than bytes stays float (its place in the byte pool is left unused)
        TexJob& j = *jobs[(size_t)sl.job];
        if (!j.notBytes.load()) continue;
        *sl.off = poolF.size();
        const float* f = (const float*)j.data;
        poolF.insert(poolF.end(), f, f + j.count * 3);
        *sl.fmt = MIPT_TEXFMT_F32;
    }

    mipt_scene* s = new mipt_scene();
    s->numTris = n; s->numMats = desc->num_materials;
#define SCTRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { mipt_scene_destroy(s); return fail(MIPT_E_DEVICE, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
    SCTRY(hipStreamCreate(&s->stream));
    if (n > 0) {
        SCTRY(hipMalloc(&s->attr, sizeof(mipt_vertex) * 3 * (size_t)n));
        SCTRY(hipMemcpy(s->attr, attr.data(), sizeof(mipt_vertex) * 3 * (size_t)n, hipMemcpyHostToDevice));
        SCTRY(hipMalloc(&s->triMat, sizeof(int32_t) * (size_t)n));
        SCTRY(hipMemcpy(s->triMat, triMat.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
    }
    if (!mats.empty()) {
        SCTRY(hipMalloc(&s->mats, sizeof(DevMaterial) * mats.size()));
        SCTRY(hipMemcpy(s->mats, mats.data(), sizeof(DevMaterial) * mats.size(), hipMemcpyHostToDevice));
    }
    if (pool8Count) {
        SCTRY(hipMalloc(&s->texels, 4 * (size_t)pool8Count));
        SCTRY(hipMemcpy(s->texels, pool8.get(), 4 * (size_t)pool8Count, hipMemcpyHostToDevice));
    }
    if (!poolF.empty()) {
        SCTRY(hipMalloc(&s->texelsF, 4 * poolF.size()));
        SCTRY(hipMemcpy(s->texelsF, poolF.data(), 4 * poolF.size(), hipMemcpyHostToDevice));
    }
    s->textureBytes = (int64_t)(4 * (size_t)pool8Count + 4 * poolF.size());
    s->uploadMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

    rc = mipt_build_bvh(s->attr, n, opt, s->stream, &s->bvh);
    if (rc) { mipt_scene_destroy(s); return fail(rc, "BVH build failed"); }
    s->geometryBytes = (int64_t)(16 * MIPT_SHADE_F4 * (size_t)n + 16 * MIPT_NODE_WIDTH * (size_t)s->bvh.numNodeSlots + 16 * MIPT_TRI_F4 * (size_t)s->bvh.numLeafSlots);

    DevScene& D = s->dev;
    D.nodes = s->bvh.nodes; D.tris = s->bvh.tris;
    if (n > 0) {
        // per-triangle shading records; the corner arrays are only needed to make them and the BVH
        SCTRY(hipMalloc(&s->shade, sizeof(float4) * MIPT_SHADE_F4 * (size_t)n));
        launch_make_shading(s->attr, s->triMat, s->mats, n, s->shade, s->stream);
        SCTRY(hipStreamSynchronize(s->stream));
        SCTRY(hipGetLastError());
        (void)hipFree(s->attr); s->attr = nullptr;
        (void)hipFree(s->triMat); s->triMat = nullptr;
    }
    D.shade = s->shade; D.mats = s->mats; D.texels = s->texels; D.texelsF = s->texelsF;
    for (int a = 0; a < 3; a++) { D.qmin[a] = s->bvh.qmin[a]; D.qcell[a] = s->bvh.qcell[a]; }
    D.rootCode = s->bvh.rootCode;
    // depth-first walk of a tree whose nodes push up to (width - 1) siblings per level
    // (two bottom rows that mark the end of the stack, and spare rows a node step may scribble on)
    D.stackDepth = 2 + (MIPT_NODE_WIDTH - 1) * s->bvh.maxDepth + 2 + MIPT_NODE_WIDTH;
    D.stackLds = D.stackDepth < MIPT_STACK_LDS_MAX ? D.stackDepth : MIPT_STACK_LDS_MAX;
    if (const char* e = getenv("MIPT_STACK_LDS")) {          // test hook: a small LDS part (this many rows above the two bottom rows) sends the stack through its global-memory slice
        int v = atoi(e);
        if (v >= 1 && v + 2 < D.stackLds) D.stackLds = v + 2;
    }
    D.numTris = (int32_t)n; D.numMats = s->numMats; D.numNodeSlots = s->bvh.numNodeSlots;
    D.leafBase = (uint32_t)s->bvh.numNodeSlots;
    {
        uint64_t codes = (uint64_t)s->bvh.numNodeSlots + (uint64_t)s->bvh.numLeafSlots + 1;      // child codes are 0 .. codes - 2, codes - 1 marks the end of a stack
        if (codes >= (1ull << 30)) { mipt_scene_destroy(s); return fail(MIPT_E_LIMIT, "BVH too large"); }
        // the traversal addresses node slots and triangle records by 32-bit byte offsets
        const uint64_t slotLimit = (1ull << 32) / (16 * (MIPT_NODE_WIDTH > MIPT_TRI_F4 ? MIPT_NODE_WIDTH : MIPT_TRI_F4));
        if ((uint64_t)s->bvh.numNodeSlots >= slotLimit || (uint64_t)s->bvh.numLeafSlots >= slotLimit) {
            mipt_scene_destroy(s);
            return fail(MIPT_E_LIMIT, "BVH too large: %lld node slots, %lld triangle records, limit %llu each", (long long)s->bvh.numNodeSlots, (long long)s->bvh.numLeafSlots, (unsigned long long)slotLimit - 1);
        }
        uint32_t m = 1;
        while (m < codes) m <<= 1;
        D.slotMask = m - 1;
    }
    if (D.stackDepth > D.stackLds) {
        // one slice per wave that can be resident: 32 per compute unit
        size_t bytes = (size_t)(D.stackDepth - D.stackLds) * 64 * 4 * (size_t)g_numCUs * 32;
        if (bytes > ((size_t)4 << 30)) { mipt_scene_destroy(s); return fail(MIPT_E_LIMIT, "BVH too deep for the traversal stack"); }
        SCTRY(hipMalloc(&s->stackSpill, bytes));
        D.stackSpill = s->stackSpill;
    }
    *out = s;
    return MIPT_OK;
}

int mipt_scene_get_info(const mipt_scene* s, mipt_scene_info* o)
{
    if (!s || !o) return fail(MIPT_E_INVALID, "null argument");
    memset(o, 0, sizeof(*o));
    o->num_triangles = s->numTris; o->num_nodes = s->bvh.numInner; o->num_leaves = s->bvh.numLeaves;
    o->max_depth = s->bvh.maxDepth; o->num_materials = s->numMats;
    o->geometry_bytes = s->geometryBytes; o->texture_bytes = s->textureBytes;
    o->build_ms = s->bvh.buildMs; o->upload_ms = s->uploadMs; o->sah_cost = s->bvh.sahCost;
    o->node_width = MIPT_NODE_WIDTH;
    return MIPT_OK;
}

int mipt_scene_bvh_sizes(const mipt_scene* s, int64_t* nodeSlots, int64_t* leafSlots, uint32_t* rootCode)
{
    if (!s) return fail(MIPT_E_INVALID, "null scene");
    if (nodeSlots) *nodeSlots = s->bvh.numNodeSlots;
    if (leafSlots) *leafSlots = s->bvh.numLeafSlots;
    if (rootCode) *rootCode = s->bvh.rootCode;
    return MIPT_OK;
}

int mipt_scene_bvh_grid(const mipt_scene* s, float* qmin, float* qcell)
{
    if (!s || !qmin || !qcell) return fail(MIPT_E_INVALID, "null argument");
    for (int a = 0; a < 3; a++) { qmin[a] = s->bvh.qmin[a]; qcell[a] = s->bvh.qcell[a]; }
    return MIPT_OK;
}

int mipt_scene_download_bvh(const mipt_scene* s, uint32_t* nodes, float* v0, float* e1, float* e2)
{
    if (!s) return fail(MIPT_E_INVALID, "null scene");
    if (nodes && s->bvh.numNodeSlots) HIPTRY(hipMemcpy(nodes, s->bvh.nodes, 16 * MIPT_NODE_WIDTH * (size_t)s->bvh.numNodeSlots, hipMemcpyDeviceToHost));
    if ((v0 || e1 || e2) && s->bvh.numLeafSlots) {
        std::vector<float> rec((size_t)s->bvh.numLeafSlots * 4 * MIPT_TRI_F4);
        HIPTRY(hipMemcpy(rec.data(), s->bvh.tris, rec.size() * 4, hipMemcpyDeviceToHost));
        float* dst[3] = { v0, e1, e2 };
        for (int k = 0; k < 3; k++)
            if (dst[k])
                for (int64_t i = 0; i < s->bvh.numLeafSlots; i++) memcpy(dst[k] + 4 * i, &rec[((size_t)i * MIPT_TRI_F4 + k) * 4], 16);
    }
    return MIPT_OK;
}

// ---------------------------------------------------------------------------------------
// The two queries of mipt_trace_rays over rays in device memory, with every buffer given: the ray
// queue q and the hit records (numRays entries each), ctr and stats (256 bytes each: 4 x int32 and ST_COUNT x uint64 at their start).
// Results go to device arrays.  Enqueues on the scene's stream and waits for it.
static hipError_t trace_device(mipt_scene* s, const float* dRays, int64_t n, float eps, RayQueue q, float4* hit, int32_t* ctr,
                               unsigned long long* stats, int32_t* dTri, float* dTuv, uint8_t* dAny)
{
    hipError_t e = hipSuccess;
    auto tryq = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    tryq(hipMemsetAsync(ctr, 0, 256, s->stream));
    tryq(hipMemsetAsync(stats, 0, 256, s->stream));
    if (e != hipSuccess) return e;
    launch_set_i32(ctr, (int32_t)n, s->stream);
    launch_pack_rays(dRays, n, q, s->stream);
    if (dTri || dTuv) {
        launch_closest(s->dev, q, ctr, (int32_t)n, ctr + 1, hit, eps, stats, g_numCUs, s->stream);
        launch_unpack_hits(hit, n, dTri, dTuv, s->stream);
    }
    if (dAny) launch_anyhit_flags(s->dev, q, ctr, (int32_t)n, ctr + 2, dAny, eps, stats, g_numCUs, s->stream);
    tryq(hipStreamSynchronize(s->stream));
    tryq(hipGetLastError());
    return e;
}

int mipt_trace_rays(mipt_scene* s, const float* rays, int64_t n, float eps, int32_t* outTri, float* outTuv, uint8_t* outAny)
{
    if (!s || (n > 0 && !rays)) return fail(MIPT_E_INVALID, "null argument");
    if (n <= 0) return MIPT_OK;
    if (n >= (1ll << 31) - 64) return fail(MIPT_E_LIMIT, "too many rays in one call");
    int rc = ensure_init();
    if (rc) return rc;
    float* dRays = nullptr; RayQueue q = {}; float4* hit = nullptr; uint8_t* flags = nullptr; int32_t* ctr = nullptr;
    int32_t* dTri = nullptr; float* dTuv = nullptr;
    unsigned long long* stats = nullptr;
    hipError_t e = hipSuccess;
    auto tryq = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    tryq(hipMalloc(&dRays, 24 * (size_t)n)); tryq(hipMalloc(&q.o, 16 * (size_t)n)); tryq(hipMalloc(&q.d, 16 * (size_t)n));
    tryq(hipMalloc(&hit, 16 * (size_t)n)); tryq(hipMalloc(&ctr, 256)); tryq(hipMalloc(&stats, 256));
    if (outAny) tryq(hipMalloc(&flags, (size_t)n));
    if (outTri) tryq(hipMalloc(&dTri, 4 * (size_t)n));
    if (outTuv) tryq(hipMalloc(&dTuv, 12 * (size_t)n));
    if (e == hipSuccess) {
        tryq(hipMemcpy(dRays, rays, 24 * (size_t)n, hipMemcpyHostToDevice));
        if (e == hipSuccess) tryq(trace_device(s, dRays, n, eps, q, hit, ctr, stats, dTri, dTuv, flags));
        if (e == hipSuccess && outTri) tryq(hipMemcpy(outTri, dTri, 4 * (size_t)n, hipMemcpyDeviceToHost));
        if (e == hipSuccess && outTuv) tryq(hipMemcpy(outTuv, dTuv, 12 * (size_t)n, hipMemcpyDeviceToHost));
        if (e == hipSuccess && outAny) tryq(hipMemcpy(outAny, flags, (size_t)n, hipMemcpyDeviceToHost));
    }
    (void)hipFree(dRays); (void)hipFree(q.o); (void)hipFree(q.d); (void)hipFree(hit); (void)hipFree(flags); (void)hipFree(ctr); (void)hipFree(stats);
    (void)hipFree(dTri); (void)hipFree(dTuv);
    if (e != hipSuccess) return fail(MIPT_E_DEVICE, "mipt_trace_rays: %s", hipGetErrorString(e));
    return MIPT_OK;
}

// The same queries without a single allocation: rays and results are device arrays of the caller,
// everything in between lives in this GPU's copy of `scratch`.  Every call on a GPU uses that one
// array (calls that overlap on one GPU race); what must start at zero - the queue heads and the
// statistics - is cleared by trace_device, the queue and the hit records are written before they are read.
int mipt_trace_rays_device(mipt_scene* s, const float* dRays, int64_t n, float eps, int32_t* dTri, float* dTuv, uint8_t* dAny)
{
    if (!s || (n > 0 && !dRays)) return fail(MIPT_E_INVALID, "null argument");
    if (n <= 0) return MIPT_OK;
    if (n > SCRATCH_RAYS) return fail(MIPT_E_LIMIT, "%lld rays exceed the %d that mipt_trace_rays_device takes per call", (long long)n, (int)SCRATCH_RAYS);
    int rc = ensure_init();
    if (rc) return rc;
    unsigned char* base = g_device < MIPT_MAX_DEVICES ? g_scratchPtr[g_device] : nullptr;
    if (!base) return fail(MIPT_E_DEVICE, "no scratch memory on device %d", g_device);
    RayQueue q = { (float4*)(base + SCRATCH_Q_O), (float4*)(base + SCRATCH_Q_D) };
    hipError_t e = trace_device(s, dRays, n, eps, q, (float4*)(base + SCRATCH_HIT), (int32_t*)(base + SCRATCH_CTR),
                                (unsigned long long*)(base + SCRATCH_STATS), dTri, dTuv, dAny);
    if (e != hipSuccess) return fail(MIPT_E_DEVICE, "mipt_trace_rays_device: %s", hipGetErrorString(e));
    return MIPT_OK;
}

// ---------------------------------------------------------------------------------------
static int fill_render(mipt_frame* f, const mipt_camera* cam, const mipt_params* p)
{
    if (cam->div <= 0 || cam->client_width <= 0 || cam->client_height <= 0) return fail(MIPT_E_INVALID, "bad camera size");
    int32_t W = cam->client_width / cam->div, H = cam->client_height / cam->div;   // PathTracer.cu:559-560
    if (W <= 0 || H <= 0) return fail(MIPT_E_INVALID, "empty image");
    if ((int64_t)W * H >= (1ll << 30)) return fail(MIPT_E_LIMIT, "image too large");
    DevRender& R = f->R;
    memcpy(R.toLocal, cam->to_local, sizeof(R.toLocal));
    R.p0 = cam->p0; R.p1 = cam->p1; R.clientW = cam->client_width; R.clientH = cam->client_height; R.div = cam->div;
    R.W = W; R.H = H;
    R.sun[0] = p->sun_pos[0]; R.sun[1] = p->sun_pos[1]; R.sun[2] = p->sun_pos[2];
    R.sunIntensity = p->sun_intensity;
    R.gi = p->global_illumination != 0; R.useTextures = p->use_textures != 0; R.useNormals = p->use_normals != 0;
    R.normalizeNormalMap = p->normalize_normal_map != 0;
    R.eps = p->epsilon; R.seed = p->seed;
    int32_t ns = p->num_samples;
    if (ns < 1) ns = 1;
    if (ns > 10000) ns = 10000;                                                       // RayTracerView.cpp:785-786
    R.numSamples = ns;
    R.shardCount = p->shard_count > 1 ? p->shard_count : 1;
    R.shardRank = p->shard_rank;
    R.ownPixelsOnly = (p->own_pixels_only != 0 && R.shardCount > 1) ? 1 : 0;
    R.countBoxTests = p->count_box_tests != 0;
    if (R.shardRank < 0 || R.shardRank >= R.shardCount) return fail(MIPT_E_INVALID, "shard_rank %d outside 0..%d", p->shard_rank, R.shardCount - 1);
    R.tileSize = p->tile_size > 0 ? p->tile_size : 32;
    R.tilesX = (W + R.tileSize - 1) / R.tileSize;
    if (p->sample_begin < 0 || p->sample_end < 0 || p->sample_end > ns || p->sample_begin > p->sample_end)
        return fail(MIPT_E_INVALID, "sample range [%d,%d) outside [0,%d)", p->sample_begin, p->sample_end, ns);
    f->cam = *cam; f->prm = *p; f->W = W; f->H = H; f->P = W * H;
    return MIPT_OK;
}

// With own_pixels_only the primary rays, their hits and their shading run over the pixels of this shard's tiles
// alone.  The list of those pixels is made when the shard fields change; the pixels of the other shards are set
// to "no hit" then, once, since nothing writes them afterwards.
static int update_own_pixels(mipt_frame* f)
{
    DevRender& R = f->R;
    R.ownPix = nullptr; R.numOwn = 0;
    if (!R.ownPixelsOnly) { f->ownRank = f->ownCount = f->ownTile = -1; return MIPT_OK; }
    if (f->ownRank != R.shardRank || f->ownCount != R.shardCount || f->ownTile != R.tileSize || !f->ownPix) {
        std::vector<int32_t> own;
        own.reserve((size_t)f->P / (size_t)R.shardCount + (size_t)R.tileSize * R.tileSize);
        for (int32_t y = 0; y < R.H; y++) {
            const int32_t tileRow = (y / R.tileSize) * R.tilesX;
            for (int32_t tx = 0; tx < R.tilesX; tx++) {
                if ((tileRow + tx) % R.shardCount != R.shardRank) continue;
                const int32_t x0 = tx * R.tileSize, x1 = x0 + R.tileSize < R.W ? x0 + R.tileSize : R.W;
                for (int32_t x = x0; x < x1; x++) own.push_back(y * R.W + x);
            }
        }
        if (!f->ownPix) HIPTRY(hipMalloc(&f->ownPix, 4 * (size_t)f->P));
        if (!own.empty()) HIPTRY(hipMemcpy(f->ownPix, own.data(), 4 * own.size(), hipMemcpyHostToDevice));
        HIPTRY(hipMemset(f->pixState, 0, 64 * (size_t)f->P));
        HIPTRY(hipMemset(f->triId, 0xff, 4 * (size_t)f->P));
        f->numOwn = (int32_t)own.size();
        f->ownRank = R.shardRank; f->ownCount = R.shardCount; f->ownTile = R.tileSize;
    }
    R.ownPix = f->ownPix; R.numOwn = f->numOwn;
    return MIPT_OK;
}

void mipt_frame_destroy(mipt_frame* f)
{
    if (!f) return;
    void* ptrs[] = { f->ownPix, f->primQ.o, f->primQ.d, f->primHit, f->pixState, f->triId, f->actPix, f->pixAccum, f->primShadowQ.o, f->primShadowQ.d,
                     f->q[0].o, f->q[0].d, f->q[1].o, f->q[1].d, f->shadowQ.o, f->shadowQ.d, f->shadowBlocks, f->hit, f->pathState, f->counters, f->stats };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (hipEvent_t e : f->events) (void)hipEventDestroy(e);
    delete f;
}

#define MAX_PASSES 10000
// counters: [0] primary count, [1] primary head, [2] primary shadow count, [3] primary shadow head,
// [4] active count, then per pass 12: for level 1..3: count, closest head, shadow count, shadow head
#define CTR_FIXED 8
#define CTR_PER_PASS 12

int mipt_frame_create(mipt_scene* s, const mipt_camera* cam, const mipt_params* p, mipt_frame** out)
{
    if (!s || !cam || !p || !out) return fail(MIPT_E_INVALID, "null argument");
    *out = nullptr;
    int rc = ensure_init();
    if (rc) return rc;
    mipt_frame* f = new mipt_frame();
    f->scene = s; f->stream = s->stream;
    rc = fill_render(f, cam, p);
    if (rc) { delete f; return rc; }
    const int64_t P = f->P;
    const int32_t ns = f->R.numSamples;
    // samples per pass: up to 128M path slots (160 bytes each) unless the caller fixes it; long
    // passes keep the machine full, every launch pays a tail while its last waves drain
    int32_t chunk = p->chunk_samples > 0 ? p->chunk_samples : (int32_t)((128ll << 20) / (P > 0 ? P : 1));
    if (chunk < 1) chunk = 1;
    if (chunk > ns) chunk = ns;
    if (p->chunk_samples <= 0 && f->R.gi) {
        // the automatic pass also has to fit the memory that is free now (a smaller device, ranks that share
        // one): 164 bytes per path slot and 180 per pixel, with a tenth of the free memory left alone
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
            const int64_t perSlot = 16 * 2 * 3 + 16 + 48 + 4, perPixel = 16 * 5 + 64 + 4 + 4 + 16 + 12;
            int64_t room = (int64_t)(freeB - freeB / 10) - perPixel * P;
            int64_t fit = room > 0 ? room / (perSlot * P) : 0;
            if (fit < 1) fit = 1;
            if (chunk > fit) chunk = (int32_t)fit;
        }
    }
    if ((int64_t)chunk * P >= (1ll << 31) - 256) chunk = (int32_t)(((1ll << 31) - 256) / P);
    if (chunk < 1) { delete f; return fail(MIPT_E_LIMIT, "image too large for one pass"); }
    f->chunk = chunk;
    f->slotCap = f->R.gi ? (int64_t)chunk * P : 0;
#define FRTRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { mipt_frame_destroy(f); return fail(e_ == hipErrorOutOfMemory ? MIPT_E_NOMEM : MIPT_E_DEVICE, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
    FRTRY(hipMalloc(&f->primQ.o, 16 * (size_t)P)); FRTRY(hipMalloc(&f->primQ.d, 16 * (size_t)P));
    FRTRY(hipMalloc(&f->primShadowQ.o, 16 * (size_t)P)); FRTRY(hipMalloc(&f->primShadowQ.d, 16 * (size_t)P));
    FRTRY(hipMalloc(&f->primHit, 16 * (size_t)P));
    FRTRY(hipMalloc(&f->pixState, 64 * (size_t)P));
    FRTRY(hipMalloc(&f->triId, 4 * (size_t)P));
    FRTRY(hipMalloc(&f->actPix, 4 * (size_t)P));
    FRTRY(hipMalloc(&f->pixAccum, 16 * (size_t)P));
    if (f->slotCap > 0) {
        size_t sc = (size_t)f->slotCap;
        for (int k = 0; k < 2; k++) { FRTRY(hipMalloc(&f->q[k].o, 16 * sc)); FRTRY(hipMalloc(&f->q[k].d, 16 * sc)); }
        FRTRY(hipMalloc(&f->shadowQ.o, 16 * sc)); FRTRY(hipMalloc(&f->shadowQ.d, 16 * sc));
        FRTRY(hipMalloc(&f->shadowBlocks, 4 * ((sc >> MIPT_QBLOCK_SHIFT) + 1)));
        FRTRY(hipMalloc(&f->hit, 16 * sc));
        FRTRY(hipMalloc(&f->pathState, 16 * 3 * sc));
        FRTRY(hipMemset(f->pathState, 0, 16 * 3 * sc));      // once: afterwards every pass overwrites what it reads (see k_shade_bounce)
    }
    int32_t passes = (ns + chunk - 1) / chunk;
    f->numCounters = CTR_FIXED + CTR_PER_PASS * passes;
    FRTRY(hipMalloc(&f->counters, 4 * (size_t)f->numCounters));
    FRTRY(hipMalloc(&f->stats, 8 * ST_COUNT));
    f->events.resize(2 + 2 * (size_t)(2 + passes * 9));
    for (auto& e : f->events) { e = nullptr; FRTRY(hipEventCreate(&e)); }
    rc = update_own_pixels(f);
    if (rc) { mipt_frame_destroy(f); return rc; }
    *out = f;
    return MIPT_OK;
}

int mipt_frame_set_params(mipt_frame* f, const mipt_camera* cam, const mipt_params* p)
{
    if (!f || !cam || !p) return fail(MIPT_E_INVALID, "null argument");
    if (cam->client_width / (cam->div > 0 ? cam->div : 1) != f->W || cam->client_height / (cam->div > 0 ? cam->div : 1) != f->H)
        return fail(MIPT_E_INVALID, "image size differs from the frame's");
    int32_t ns = p->num_samples < 1 ? 1 : (p->num_samples > 10000 ? 10000 : p->num_samples);
    int32_t passes = (ns + f->chunk - 1) / f->chunk;
    if (CTR_FIXED + CTR_PER_PASS * passes > f->numCounters) return fail(MIPT_E_INVALID, "more samples than the frame was created for");
    if ((p->global_illumination != 0) && f->slotCap == 0) return fail(MIPT_E_INVALID, "frame was created without global illumination");
    // values are checked on a copy: a rejected call leaves the frame as it was
    const DevRender keepR = f->R;
    int rc = fill_render(f, cam, p);
    if (rc) { f->R = keepR; return rc; }
    return update_own_pixels(f);
}

int64_t mipt_frame_num_pixels(const mipt_frame* f) { return f ? f->P : 0; }

int mipt_frame_accumulate(mipt_frame* f, float* d_accum, mipt_stats* stats, mipt_progress_fn progress, void* user)
{
    if (!f || !d_accum) return fail(MIPT_E_INVALID, "null argument");
    int rc = ensure_init();
    if (rc) return rc;
    mipt_scene* sc = f->scene;
    hipStream_t st = f->stream;
    const DevRender& R = f->R;
    const DevScene& S = sc->dev;
    const int32_t P = f->P;
    const int64_t total = P;
    size_t ev = 0;
    std::vector<std::pair<size_t, int>> spans;    // (first event index, kind 0 closest / 1 shadow / 2 bounce shading)
    auto spanBegin = [&](int kind) { spans.push_back({ ev, kind }); (void)hipEventRecord(f->events[ev++], st); };
    auto spanEnd = [&]() { (void)hipEventRecord(f->events[ev++], st); };

    if (progress) progress(0, total, user);                                             // PathTracer.cu:721
    HIPTRY(hipEventRecord(f->events[ev++], st));                                   // events[0] = start of the call
    size_t evEndIdx = ev++;                                                        // events[1] = end of the call
    HIPTRY(hipMemsetAsync(f->counters, 0, 4 * (size_t)f->numCounters, st));
    HIPTRY(hipMemsetAsync(f->stats, 0, 8 * ST_COUNT, st));
    HIPTRY(hipMemsetAsync(d_accum, 0, 12 * (size_t)P, st));
    const int32_t nPrimary = R.ownPix ? R.numOwn : P;        // primary rays: one per pixel, or one per pixel of this shard's tiles
    launch_set_i32(f->counters + 0, nPrimary, st);

    // depth 0: primary rays, their surfaces and shadow rays
    launch_raygen(R, f->primQ, st);
    spanBegin(0);
    launch_closest(S, f->primQ, f->counters + 0, nPrimary, f->counters + 1, f->primHit, R.eps, f->stats, g_numCUs, st, false, R.countBoxTests != 0);
    spanEnd();
    launch_shade_primary(S, R, f->primHit, f->pixState, f->triId, f->primShadowQ, f->counters + 2, f->actPix, f->counters + 4, f->stats, st);
    spanBegin(1);
    launch_shadow(S, f->primShadowQ, f->counters + 2, nullptr, P, f->counters + 3, f->pixState, f->pathState, f->slotCap, R.eps, f->stats, g_numCUs, st, R.countBoxTests != 0);
    spanEnd();

    int32_t nAct = 0;
    int32_t usedChunk = 0;
    bool cancelled = false;
    f->samplesDone = 0;
    if (R.gi) {
        HIPTRY(hipMemcpyAsync(&nAct, f->counters + 4, 4, hipMemcpyDeviceToHost, st));
        HIPTRY(hipStreamSynchronize(st));
        f->lastActive = nAct;
        int32_t sb = f->prm.sample_begin, se = f->prm.sample_end;
        if (sb == 0 && se == 0) se = R.numSamples;
        if (nAct > 0 && se > sb) {
            HIPTRY(hipMemsetAsync(f->pixAccum, 0, 16 * (size_t)nAct, st));
            // the pass size was fixed for the whole image; with fewer active pixels more samples fit
            int64_t fit = f->slotCap / nAct;
            int32_t chunk = (f->prm.chunk_samples > 0) ? f->chunk : (int32_t)(fit > (se - sb) ? (se - sb) : fit);
            if (chunk < 1) chunk = 1;
            int32_t maxPasses = (f->numCounters - CTR_FIXED) / CTR_PER_PASS;
            if ((se - sb + chunk - 1) / chunk > maxPasses) chunk = (se - sb + maxPasses - 1) / maxPasses;
            usedChunk = chunk;
            int pass = 0;
            for (int32_t n0 = sb; n0 < se; n0 += chunk, pass++) {
                int32_t Sn = (se - n0 < chunk) ? se - n0 : chunk;
                int64_t slots = (int64_t)nAct * Sn;
                int32_t* C = f->counters + CTR_FIXED + CTR_PER_PASS * pass;
                launch_set_i32(C + 0, (int32_t)slots, st);
                launch_gen_first_bounce(R, f->pixState, f->actPix, nAct, n0, Sn, f->q[0], st);
                for (int level = 1; level <= 3; level++) {
                    // per level: closest head, hit count (= rays of the next level), shadow head;
                    // the rays of level 1 are counted in C[0]
                    int32_t* Lc = C + 1 + 3 * (level - 1);
                    const int32_t* inCount = (level == 1) ? C : Lc - 2;
                    RayQueue cur = f->q[(level - 1) & 1], nxt = f->q[level & 1];
                    spanBegin(0);
                    launch_closest(S, cur, inCount, (int32_t)slots, Lc + 0, f->hit, R.eps, f->stats, g_numCUs, st, true, R.countBoxTests != 0);
                    spanEnd();
                    spanBegin(2);
                    launch_shade_bounce(S, R, cur, f->hit, inCount, (int32_t)slots, level, f->actPix, nAct, n0, f->pathState, slots,
                                        f->shadowQ, f->shadowBlocks, nxt, Lc + 1, f->stats, st);
                    spanEnd();
                    spanBegin(1);
                    launch_shadow(S, f->shadowQ, inCount, f->shadowBlocks, (int32_t)slots, Lc + 2, f->pixState, f->pathState, slots, R.eps, f->stats, g_numCUs, st, R.countBoxTests != 0);
                    spanEnd();
                    if (progress && level < 3) {
                        // the reference reports every integer percent (PathTracer.cu:726-731): between the passes the
                        // work done is reported after every depth.  The sums handed over so far are those of the
                        // passes that are complete (samples_done is unchanged until this pass ends).
                        HIPTRY(hipStreamSynchronize(st));
                        progress(total * ((int64_t)(n0 - sb) * 3 + (int64_t)Sn * level) / ((int64_t)(se - sb) * 3), total, user);
                    }
                }
                launch_resolve(R, f->pathState, slots, f->actPix, nAct, n0, Sn, f->pixAccum, st);
                f->samplesDone = n0 + Sn - sb;
                if (progress) {
                    // the sums so far are handed over before every call-back, so that the caller can
                    // look at (or finalize) a partial image from inside it
                    launch_scatter_accum(f->pixAccum, f->actPix, nAct, d_accum, st);
                    HIPTRY(hipStreamSynchronize(st));
                    progress(total * (n0 + Sn - sb) / (se - sb), total, user);
                }
                if (f->cancel.load(std::memory_order_relaxed)) { cancelled = true; break; }
            }
            launch_scatter_accum(f->pixAccum, f->actPix, nAct, d_accum, st);
        }
    }
    HIPTRY(hipEventRecord(f->events[evEndIdx], st));
    HIPTRY(hipStreamSynchronize(st));
    HIPTRY(hipGetLastError());
    if (progress && !(R.gi && nAct > 0)) progress(total, total, user);
    if (!cancelled && f->cancel.load(std::memory_order_relaxed)) cancelled = true;     // asked for during the last pass or before the call
    f->cancel.store(0, std::memory_order_relaxed);
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        unsigned long long h[ST_COUNT];
        HIPTRY(hipMemcpy(h, f->stats, sizeof(h), hipMemcpyDeviceToHost));
        stats->closest_rays = h[ST_CLOSEST_RAYS]; stats->shadow_rays = h[ST_SHADOW_RAYS]; stats->shadow_rays_culled = h[ST_SHADOW_CULLED]; stats->closest_box_tests = h[ST_CLOSEST_BOXES];
        stats->shadow_box_tests = h[ST_SHADOW_BOXES]; stats->shaded_hits = h[ST_SHADE_HITS]; stats->bounce_rays_written = h[ST_SHADE_NEXT];
#ifdef MIPT_TRACE_PROFILE
        fprintf(stderr, "trace profile (closest): rays %llu nodeVisits %llu triTests %llu | wave: outer %llu nodeIters %llu triIters %llu nodePhases %llu refills %llu finishEvents %llu finishLanes %llu slowNodeSteps %llu | lanes per node step %.1f per triangle step %.1f\n",
                h[ST_CLOSEST_RAYS], h[ST_CLOSEST_NODES], h[ST_CLOSEST_TRIS], h[ST_PROF], h[ST_PROF + 1], h[ST_PROF + 2], h[ST_PROF + 3], h[ST_PROF + 4], h[ST_PROF + 5], h[ST_PROF + 6], h[ST_PROF + 7],
                h[ST_PROF + 1] ? (double)h[ST_CLOSEST_NODES] / (double)h[ST_PROF + 1] : 0.0, h[ST_PROF + 2] ? (double)h[ST_CLOSEST_TRIS] / (double)h[ST_PROF + 2] : 0.0);
#endif
#ifdef MIPT_TRACE_TAIL
        fprintf(stderr, "trace tail (closest-hit launches of the bounce levels): %llu waves, alive %.4g ms each on average, of that %.4g ms after the queue ran dry\n",
                (unsigned long long)h[ST_PROF + 14], h[ST_PROF + 14] ? 1e-5 * (double)h[ST_PROF + 12] / (double)h[ST_PROF + 14] : 0.0, h[ST_PROF + 14] ? 1e-5 * (double)h[ST_PROF + 13] / (double)h[ST_PROF + 14] : 0.0);
#endif
#ifdef MIPT_TRACE_CLOCK
        fprintf(stderr, "trace clock (closest, summed over waves): node phases %.4g cycles (%.3f), triangle phases %.4g (%.3f), all %.4g\n",
                (double)h[ST_PROF + 8], (double)h[ST_PROF + 8] / (double)h[ST_PROF + 10], (double)h[ST_PROF + 9], (double)h[ST_PROF + 9] / (double)h[ST_PROF + 10], (double)h[ST_PROF + 10]);
#endif
        stats->node_visits = h[ST_CLOSEST_NODES] + h[ST_SHADOW_NODES];
        stats->triangle_tests = h[ST_CLOSEST_TRIS] + h[ST_SHADOW_TRIS];
        stats->closest_node_visits = h[ST_CLOSEST_NODES]; stats->closest_triangle_tests = h[ST_CLOSEST_TRIS];
        stats->active_pixels = (uint64_t)nAct;
        for (auto& sp : spans) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, f->events[sp.first], f->events[sp.first + 1]) == hipSuccess) {
                if (sp.second == 0) { stats->closest_ms += ms; stats->closest_launches++; }
                else if (sp.second == 1) { stats->shadow_ms += ms; stats->shadow_launches++; }
                else { stats->shade_bounce_ms += ms; stats->shade_bounce_launches++; }
            }
        }
        float tot = 0;
        if (hipEventElapsedTime(&tot, f->events[0], f->events[evEndIdx]) == hipSuccess) stats->total_ms = tot;
        stats->shade_ms = stats->total_ms - stats->closest_ms - stats->shadow_ms;
        stats->chunk_samples = usedChunk;
        stats->samples_done = f->samplesDone;
    }
    if (cancelled) return fail(MIPT_E_CANCELLED, "cancelled after %d samples", f->samplesDone);
    return MIPT_OK;
}

int mipt_frame_finalize_partial(mipt_frame* f, const float* d_accum, int32_t samples, float* d_out)
{
    if (!f || !d_accum || !d_out) return fail(MIPT_E_INVALID, "null argument");
    if (samples < 0 || samples > f->R.numSamples) return fail(MIPT_E_INVALID, "sample count %d outside 0..%d", samples, f->R.numSamples);
    int rc = ensure_init();
    if (rc) return rc;
    DevRender R = f->R;
    R.numSamples = samples > 0 ? samples : 1;        // no samples: the sums are zero, direct light only
    launch_finalize(R, f->pixState, d_accum, d_out, f->stream);
    HIPTRY(hipStreamSynchronize(f->stream));
    HIPTRY(hipGetLastError());
    return MIPT_OK;
}

int mipt_frame_finalize(mipt_frame* f, const float* d_accum, float* d_out)
{
    if (!f) return fail(MIPT_E_INVALID, "null argument");
    return mipt_frame_finalize_partial(f, d_accum, f->R.numSamples, d_out);
}

int mipt_frame_cancel(mipt_frame* f)
{
    if (!f) return fail(MIPT_E_INVALID, "null argument");
    f->cancel.store(1, std::memory_order_relaxed);
    return MIPT_OK;
}

int32_t mipt_frame_samples_done(const mipt_frame* f) { return f ? f->samplesDone : 0; }

int mipt_frame_primary_ids(mipt_frame* f, int32_t* out)
{
    if (!f || !out) return fail(MIPT_E_INVALID, "null argument");
    HIPTRY(hipMemcpy(out, f->triId, 4 * (size_t)f->P, hipMemcpyDeviceToHost));
    return MIPT_OK;
}

int mipt_render(mipt_scene* s, const mipt_camera* cam, const mipt_params* p, float* out, int32_t* outTri, mipt_stats* stats,
                mipt_progress_fn progress, void* user)
{
    if (!out) return fail(MIPT_E_INVALID, "null output image");
    mipt_frame* f = nullptr;
    int rc = mipt_frame_create(s, cam, p, &f);
    if (rc) return rc;
    float *dAccum = nullptr, *dOut = nullptr;
    size_t bytes = 12 * (size_t)f->P;
    hipError_t e = hipMalloc(&dAccum, bytes);
    if (e == hipSuccess) e = hipMalloc(&dOut, bytes);
    if (e != hipSuccess) { (void)hipFree(dAccum); mipt_frame_destroy(f); return fail(MIPT_E_NOMEM, "hipMalloc: %s", hipGetErrorString(e)); }
    rc = mipt_frame_accumulate(f, dAccum, stats, progress, user);
    if (!rc) rc = mipt_frame_finalize(f, dAccum, dOut);
    if (!rc && hipMemcpy(out, dOut, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MIPT_E_DEVICE, "copy of the image failed");
    if (!rc && outTri) rc = mipt_frame_primary_ids(f, outTri);
    (void)hipFree(dAccum); (void)hipFree(dOut);
    mipt_frame_destroy(f);
    return rc;
}

int mipt_device_alloc(void** out, int64_t bytes)
{
    if (!out || bytes <= 0) return fail(MIPT_E_INVALID, "bad allocation request");
    int rc = ensure_init();
    if (rc) return rc;
    HIPTRY(hipMalloc(out, (size_t)bytes));
    return MIPT_OK;
}
int mipt_device_free(void* p) { if (p) HIPTRY(hipFree(p)); return MIPT_OK; }
int mipt_copy_to_host(void* dst, const void* src, int64_t bytes) { HIPTRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost)); return MIPT_OK; }
int mipt_copy_to_device(void* dst, const void* src, int64_t bytes) { HIPTRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice)); return MIPT_OK; }
int mipt_device_synchronize(void) { HIPTRY(hipDeviceSynchronize()); return MIPT_OK; }

} // extern "C"
