/*
 * mipt.h - C ABI of the MI355X path-tracing core.
 *
 * This is the drop-in boundary for ONE path of JohnLeber/CUDA-Path-Tracer: the
 * render call that the reference exposes as CCUDAPathTracer::CalcRays
 * (PathTracer.h:48-49, defined PathTracer.cu:556-771) and its CPU twin
 * CCPUPathTracer::CalcRays (CPUPathTracer.h:17-18).  Everything here is plain
 * C: fixed-width integers, plain pointers and sizes, no C++ or torch types.
 *
 * Conventions shared with the reference
 *  - colour triples are BGR (.x = blue): MainFrm.cpp:247, RayTracerView.cpp:824-826
 *  - matrices are row-major, row-vector convention (v' = v * M): PathTracer.cu:17-24
 *  - the output image is (client_width/div) x (client_height/div), row-major,
 *    3 floats per pixel holding 255*rgb: PathTracer.cu:550-552
 *  - "number of samples" is the number of hemisphere samples at the FIRST hit;
 *    deeper bounces use one sample: PathTracer.cu:447
 *
 * A global triangle id is the position of a triangle in the concatenation, in
 * mesh order, of the triangle lists of all meshes with is_light == 0.
 * Ties in the closest-hit search are broken towards the lowest id.
 *
 * Every function returns 0 on success and a negative MIPT_E_* code on failure;
 * mipt_last_error() then returns a message for the calling thread.
 */
#ifndef MIPT_H
#define MIPT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIPT_OK            0
#define MIPT_E_INVALID    -1   /* bad argument */
#define MIPT_E_DEVICE     -2   /* HIP runtime error */
#define MIPT_E_NOMEM      -3   /* host or device allocation failed */
#define MIPT_E_LIMIT      -4   /* a size exceeds what the library supports */
#define MIPT_E_CANCELLED  -5   /* mipt_frame_cancel ended the call early; partial sums are valid */

/* texel formats of mipt_material::tex_format / nml_format */
#define MIPT_TEX_NONE   0
#define MIPT_TEX_F32X3  1   /* 3 floats per texel, byte values 0..255 stored as float,
                               .x = blue.  Layout of CUDAMaterial::pTexData,
                               PathTracer.h:21, filled at RayTracerView.cpp:819-842 */
#define MIPT_TEX_BGRA8  2   /* one uint32 per texel, byte0 = blue, byte1 = green,
                               byte2 = red.  Layout of CTexture::m_pTexData,
                               framework.h:78, filled at MainFrm.cpp:262-269 */

/* One vertex of a triangle.  Replaces CCUDAVertex (PathTracer.h:12-17) and
 * VBVertex (framework.h:66-70); same 32-byte layout. */
typedef struct mipt_vertex {
    float pos[3];
    float normal[3];
    float tex[2];
} mipt_vertex;

/* Replaces CUDAMaterial (PathTracer.h:19-28) / the texture members of CMesh
 * (framework.h:130-136).  Pointers are HOST pointers owned by the caller. */
typedef struct mipt_material {
    const void* tex_data;
    int32_t     tex_width;
    int32_t     tex_height;
    int32_t     tex_format;
    int32_t     nml_format;
    const void* nml_data;
    int32_t     nml_width;
    int32_t     nml_height;
    float       diffuse[3];      /* BGR; used when the material has no texture */
    int32_t     reserved;
} mipt_material;

/* Replaces the root CUDAMesh (PathTracer.h:30-39): a run of triangles that
 * share one material.  Octree children of the reference's mesh tree are not
 * part of the interface: the root always carries the full triangle list
 * (RayTracerView.cpp:751-766) and the library builds its own BVH. */
typedef struct mipt_mesh {
    int64_t first_triangle;      /* index into mipt_scene_desc::vertices / 3 */
    int64_t num_triangles;
    int32_t material;            /* index into materials, -1 = none (pMaterial == 0) */
    int32_t is_light;            /* CUDAMesh::bLight; such meshes are skipped, PathTracer.cu:261 */
} mipt_mesh;

typedef struct mipt_scene_desc {
    const mipt_vertex*   vertices;       /* 3 * num_triangles entries */
    int64_t              num_triangles;
    const mipt_mesh*     meshes;
    int32_t              num_meshes;
    int32_t              num_materials;
    const mipt_material* materials;
} mipt_scene_desc;

/* Camera scalars exactly as CalcRays receives them (PathTracer.cu:556):
 * nClientWidth, nClientHeight, nDiv, P0 = Proj(0,0), P1 = Proj(1,1), ToLocal. */
typedef struct mipt_camera {
    int32_t client_width;
    int32_t client_height;
    int32_t div;
    float   p0;
    float   p1;
    float   to_local[16];
} mipt_camera;

typedef struct mipt_params {
    int32_t  num_samples;            /* nNumSamples, 1..10000 (RayTracerView.cpp:785-786) */
    float    sun_pos[3];             /* nSunPos */
    float    sun_intensity;          /* nSunIntensity, 3 in the reference (framework.h:99) */
    int32_t  global_illumination;    /* bGlobalIllumination */
    int32_t  use_textures;           /* bUseTextures */
    int32_t  use_normals;            /* bUseNormals */
    float    epsilon;                /* |det| threshold of the ray/triangle test: 1e-20 follows
                                        xnacollision.cpp:1073-1076, 1e-5 follows PathTracer.cu:153 */
    int32_t  normalize_normal_map;   /* 0 follows CPUPathTracer.cpp:315-317, 1 follows PathTracer.cu:314 */
    uint32_t seed;                   /* key of the counter-based sample stream */
    /* partition of the image and of the first-bounce sample loop over GPUs */
    int32_t  shard_rank;             /* this process's index, 0..shard_count-1 */
    int32_t  shard_count;            /* 0 or 1 = unsharded */
    int32_t  tile_size;              /* edge of the square pixel tiles dealt round-robin; 0 = 32 */
    int32_t  sample_begin;           /* first-bounce samples [sample_begin, sample_end) are */
    int32_t  sample_end;             /* traced by this call; 0,0 = all num_samples */
    int32_t  chunk_samples;          /* samples per wavefront pass, 0 = automatic */
    int32_t  own_pixels_only;        /* 1: with tile sharding, the primary rays, their hits and their shading run over
                                        the pixels of this shard's tiles alone;
                                        pixels of other shards finalize to black (and -1 in the id image), so
                                        the FINAL images of the shards add up to the frame.  0: every shard
                                        traces depth 0 for the whole image (needed when the sums are reduced
                                        and finalized on one rank, as sample-range sharding requires) */
    int32_t  count_box_tests;        /* 1: mipt_stats.closest_box_tests and shadow_box_tests are filled (the traversal kernels then
                                        count the child boxes that exist in every node they visit; a few percent slower) */
} mipt_params;

typedef struct mipt_build_options {
    int32_t sah_bins;                /* 0 = 32 (the maximum) */
    int32_t max_leaf_triangles;      /* 0 = 4 */
    float   traversal_cost;          /* 0 = 1.0 (in units of one triangle test) */
    int32_t reserved;
} mipt_build_options;

typedef struct mipt_scene_info {
    int64_t num_triangles;           /* non-light triangles in the BVH */
    int64_t num_nodes;               /* inner (wide) nodes */
    int64_t num_leaves;
    int32_t max_depth;               /* deepest leaf; root = 0 */
    int32_t num_materials;
    int64_t geometry_bytes;          /* device bytes: nodes + triangle arrays + attributes */
    int64_t texture_bytes;
    double  build_ms;                /* device time of the BVH build */
    double  upload_ms;               /* host flattening + H2D copies */
    double  sah_cost;                /* SAH cost of the binary tree the wide nodes are made from */
    int32_t node_width;              /* children per node slot */
    int32_t reserved;
} mipt_scene_info;

/* Device-side counters and timings of one mipt_frame_accumulate call. */
typedef struct mipt_stats {
    uint64_t closest_rays;           /* closest-hit queries issued */
    uint64_t shadow_rays;            /* any-hit queries traced */
    uint64_t node_visits;            /* BVH nodes fetched (both kinds of query) */
    uint64_t triangle_tests;         /* ray/triangle tests executed */
    uint64_t closest_node_visits;
    uint64_t closest_triangle_tests;
    uint64_t active_pixels;          /* pixels of this shard whose primary ray hit */
    uint64_t shadow_rays_culled;     /* any-hit queries of the reference algorithm (CPUPathTracer.cpp:402) that were not traced:
                                        the direct term they guard is +0 whatever the answer */
    double   closest_ms;             /* summed HIP-event time of the closest-hit kernel launches */
    double   shadow_ms;              /* ... of the any-hit kernel launches */
    double   shade_ms;               /* ... of ray generation, shading and resolve kernels */
    double   total_ms;               /* HIP-event time of the whole call */
    int32_t  closest_launches;
    int32_t  shadow_launches;
    int32_t  chunk_samples;          /* samples per pass actually used */
    int32_t  samples_done;           /* first-bounce samples summed into d_accum (all of them unless cancelled) */
    uint64_t closest_box_tests;      /* ray/box tests of the closest-hit queries against child boxes that exist (a node
                                        visit tests 2..4 of them); 0 unless mipt_params.count_box_tests */
    uint64_t shadow_box_tests;       /* the same for the any-hit queries */
    uint64_t shaded_hits;            /* hits shaded at depth 1..3 (the bounce-shading kernel) */
    uint64_t bounce_rays_written;    /* rays that kernel put into the queue of the next depth */
    double   shade_bounce_ms;        /* summed HIP-event time of the bounce-shading kernel launches (part of shade_ms) */
    int32_t  shade_bounce_launches;
    int32_t  reserved;
} mipt_stats;

/* Replaces CPTCallback::UpdateProgress (PTProgress.h:3-7).  Called on the calling
 * thread: first (0,total), then with a rising `done`, as PathTracer.cu:719-733. */
typedef void (*mipt_progress_fn)(int64_t done, int64_t total, void* user);

typedef struct mipt_scene mipt_scene;
typedef struct mipt_frame mipt_frame;

/* Process-wide state, as in the reference (whose render call is not re-entrant either: its device
 * flags and stack limit are process-global, PathTracer.cu:577,650): the library renders on ONE device
 * per process, the one of the last mipt_init call (one process per GPU is the intended use), and that
 * choice is shared by all threads.  A scene owns the HIP stream its frames run on and the spill part
 * of the traversal stack, which all frames of the scene share: calls on one scene, or on frames of
 * one scene, must not overlap in time, and a scene must outlive its frames (mipt_scene_destroy with a
 * frame still alive leaves that frame unusable).  Different scenes may be used from different threads.
 *
 * Selects the HIP device for this process (PathTracer.cu:577 hard-codes 0). */
int mipt_init(int32_t device);
const char* mipt_last_error(void);
int mipt_device_name(char* buf, int32_t buflen);

/* Flattens the scene, uploads it once and builds the SAH BVH on the GPU.
 * Replaces the per-call deep copy of PathTracer.cu:593-677 (CopyMesh :466-499)
 * and the octree of Mesh.cpp:88-166.
 * MIPT_E_LIMIT when the tree would hold 2^26 or more node slots or triangle records (the traversal
 * addresses both by 32-bit byte offsets; a scene of 50 million triangles is within the limit). */
int mipt_scene_create(const mipt_scene_desc* desc, const mipt_build_options* opt, mipt_scene** out);
void mipt_scene_destroy(mipt_scene* scene);
int mipt_scene_get_info(const mipt_scene* scene, mipt_scene_info* out);

/* Copies the finished BVH to host memory for validation.  nodes: 4 * node_width uint32 per node
 * slot (num_node_slots of them): 3 * node_width words of 16-bit box coordinates on the grid
 * returned by mipt_scene_bvh_grid (coordinate = qmin + q * qcell; words 3k .. 3k+2 = x, y, z of
 * child k, each word min | max << 16; an unused child has min > max), then node_width child
 * codes.  Child codes index ONE space that holds the node slots first and the leaf slots behind
 * them: a code c < num_node_slots is the inner node in slot c, a code c >= num_node_slots is a
 * leaf whose first triangle is leaf slot c - num_node_slots (its triangles are consecutive leaf
 * slots, the last one carries 1 in the fourth float of tri_e1); 0xFFFFFFFF = unused child.
 * root_code follows the same rule: 0xFFFFFFFF for an empty scene; 0 with num_node_slots == 0 for
 * a scene that is a single leaf (leaf slot 0); else 0, the root node's slot.  Slots that hold
 * no node are zero.  tri_v0/e1/e2: 4 floats per leaf slot = (v0, triangle id), (v1 - v0, last
 * flag), (v2 - v0, 0). */
int mipt_scene_bvh_sizes(const mipt_scene* scene, int64_t* num_node_slots, int64_t* num_leaf_slots,
                         uint32_t* root_code);
int mipt_scene_bvh_grid(const mipt_scene* scene, float qmin[3], float qcell[3]);
int mipt_scene_download_bvh(const mipt_scene* scene, uint32_t* nodes, float* tri_v0, float* tri_e1,
                            float* tri_e2);

/* Closest-hit / any-hit queries for caller-supplied rays (host arrays, 6 floats
 * per ray: origin, direction).  This is the search of TraceRays, PathTracer.cu:
 * 244-288 / CCPUPathTracer::Trace, CPUPathTracer.cpp:251-285, without shading.
 * out_tri[i] = global triangle id or -1; out_tuv = 3 floats per ray (t, ua, va);
 * out_any[i] = 1 if any triangle is hit with t >= 0 (the bHitOnly result).
 * The results are those of testing every triangle whose box the ray reaches before its best hit (a
 * triangle almost in the ray's plane can be accepted by the test with a distance that is rounding noise
 * and lies before its own box: one of 4 000 000 test rays), as long as the ray starts
 * within a few hundred scene sizes of the scene: the boxes are tested with a margin of 32 units in
 * the last place of the ray's reach, and from some thousand scene sizes away the rounding of the
 * triangle test itself (barycentric coordinates off by 0.1) accepts triangles beyond that margin. */
int mipt_trace_rays(mipt_scene* scene, const float* rays, int64_t num_rays, float epsilon,
                    int32_t* out_tri, float* out_tuv, uint8_t* out_any);

/* The same two queries for callers that cannot allocate device memory (tight serving loops,
 * harnesses that forbid hipMalloc).  This entry point has no counterpart in the reference.  d_rays (6 floats per ray) and the three outputs are DEVICE arrays
 * of the caller, each output may be null; the call allocates and frees nothing: its ray queue, hit
 * records and counters live in a fixed array inside the library's code object (48 bytes per ray,
 * 12.6 MB, one copy per GPU for the life of the process).  At most MIPT_TRACE_DEVICE_MAX_RAYS rays
 * per call, MIPT_E_LIMIT beyond.  The call runs on the scene's own stream and waits for it before it
 * returns (so it is no call for a thread that is capturing a HIP graph: nothing it launches would
 * land in that graph, and the wait is not allowed there).  All calls on one GPU share the array:
 * calls that overlap in time race. */
#define MIPT_TRACE_DEVICE_MAX_RAYS (1 << 18)
int mipt_trace_rays_device(mipt_scene* scene, const float* d_rays, int64_t num_rays, float epsilon,
                           int32_t* d_out_tri, float* d_out_tuv, uint8_t* d_out_any);

/* Work buffers for one image size; reused across calls so that a render call
 * allocates nothing. */
int mipt_frame_create(mipt_scene* scene, const mipt_camera* cam, const mipt_params* params,
                      mipt_frame** out);
void mipt_frame_destroy(mipt_frame* frame);
int mipt_frame_set_params(mipt_frame* frame, const mipt_camera* cam, const mipt_params* params);

/* Traces this shard's part of the image.  d_accum is a DEVICE buffer of
 * 3 floats per pixel that receives the unclamped first-bounce sum
 * sum_n r1_n * L1_n / pdf of PathTracer.cu:449-451 (zero for pixels of other
 * shards), ready for a sum-reduce across GPUs. */
int mipt_frame_accumulate(mipt_frame* frame, float* d_accum, mipt_stats* stats,
                          mipt_progress_fn progress, void* user);
/* Applies PathTracer.cu:453-462 and :550-552 to the reduced sums: d_out (DEVICE,
 * 3 floats per pixel) = 255 * min(1, (direct + I * sum / N) * albedo / pi). */
int mipt_frame_finalize(mipt_frame* frame, const float* d_accum, float* d_out_bgr255);
/* Cancel and partial results.  The reference has neither (README.md:45: a render cannot be
 * stopped; its poll loop, PathTracer.cu:719-733, only reports progress).  mipt_frame_cancel may be
 * called from the progress call-back or from another thread; mipt_frame_accumulate then stops
 * after the pass of samples it is in, leaves the sums of the samples done so far in d_accum and
 * returns MIPT_E_CANCELLED.  A cancel asked for while no call runs ends the next call after its
 * first pass.  Whenever the call-back runs, d_accum already holds the sums of
 * mipt_frame_samples_done() samples, and mipt_frame_finalize_partial turns them into an image
 * (the final formula with N = samples): a caller can show the picture as it converges. */
int mipt_frame_cancel(mipt_frame* frame);
int32_t mipt_frame_samples_done(const mipt_frame* frame);
int mipt_frame_finalize_partial(mipt_frame* frame, const float* d_accum, int32_t samples, float* d_out_bgr255);
/* Primary-hit triangle ids of the last accumulate call (HOST buffer, one per pixel). */
int mipt_frame_primary_ids(mipt_frame* frame, int32_t* out_tri);
int64_t mipt_frame_num_pixels(const mipt_frame* frame);

/* Whole render with host buffers: the body of CCUDAPathTracer::CalcRays
 * (PathTracer.cu:556-771) for a prepared scene.  out_bgr255: 3 floats per pixel. */
int mipt_render(mipt_scene* scene, const mipt_camera* cam, const mipt_params* params,
                float* out_bgr255, int32_t* out_primary_tri, mipt_stats* stats,
                mipt_progress_fn progress, void* user);

/* Device memory helpers for hosts that have no allocator of their own. */
int mipt_device_alloc(void** out, int64_t bytes);
int mipt_device_free(void* p);
int mipt_copy_to_host(void* dst, const void* d_src, int64_t bytes);
int mipt_copy_to_device(void* d_dst, const void* src, int64_t bytes);
int mipt_device_synchronize(void);

#ifdef __cplusplus
}
#endif
#endif /* MIPT_H */
