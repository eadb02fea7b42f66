/*
 * pt_oracle.h - CPU restatement of the reference's CPU path tracer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (the HIP library, the
 * headless harness, the Python binding) may include, link or call this; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED: the reference holds no tests, golden vectors or known-answer
 * fixtures for this path, and its own sources cannot be compiled here (MFC,
 * DirectXMath, Windows SDK, nvcc are absent).  The restatement is checked
 * against analytic known answers and against itself (octree vs brute force).
 *
 * Each function cites the reference lines it follows.  Arithmetic is plain
 * IEEE fp32, one rounding per operation, in the operand order of the cited
 * code (build with -ffp-contract=off, never -ffast-math).
 *
 * Input structures are the data types of include/mipt.h so that a test hands
 * the same bytes to the oracle and to the library under test.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>
#include "../include/mipt.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PTO_ACCEL_OCTREE 0  /* the reference's search: per-mesh octree, XNA slab test,
                               first-found wins on equal t (CPUPathTracer.cpp:195-285) */
#define PTO_ACCEL_BRUTE  1  /* definition of the result: every triangle tested, no box
                               test, lowest global triangle id wins on equal t */

typedef struct pto_scene pto_scene;

typedef struct pto_hit {
    int32_t tri;        /* global triangle id, -1 = miss */
    float   t, ua, va;  /* distance and barycentrics of the winning triangle */
    float   u, v;       /* interpolated texture coordinates as CMesh::Intersect stores them */
    float   hitpoint[3];
    float   normal[3];
    int32_t mesh;       /* index of the winning mesh */
} pto_hit;

typedef struct pto_stats {
    uint64_t closest_rays;
    uint64_t shadow_rays;
    uint64_t triangle_tests;
    uint64_t box_tests;
    double   seconds;
    int32_t  threads;
    int32_t  reserved;
} pto_stats;

/* xnacollision.cpp:1068-1168 with the epsilon as a parameter */
int pto_intersect_ray_triangle(const float o[3], const float d[3], const float v0[3],
                               const float v1[3], const float v2[3], float eps,
                               float* t, float* u, float* v);
/* xnacollision.cpp:1228-1297 */
int pto_intersect_ray_aabb(const float o[3], const float d[3], const float center[3],
                           const float extents[3], float* dist);
/* xnacollision.cpp:964-972 */
int pto_point_in_aabb(const float p[3], const float center[3], const float extents[3]);
/* CPUPathTracer.cpp:448-456 */
void pto_uniform_sample_hemisphere(float r1, float r2, float out[3]);
/* CPUPathTracer.cpp:458-472 */
void pto_create_coordinate_system(const float n[3], float nt[3], float nb[3]);
/* the harness's counter-based uniform in [0,1): key = (seed, pixel, sample, depth, dim) */
float pto_rng_u01(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t dim);
/* CPUPathTracer.cpp:489-497: primary ray of image pixel (j,h) */
void pto_primary_ray(const mipt_camera* cam, int32_t j, int32_t h, float o[3], float d[3]);

/* Copies the scene and builds the per-mesh octrees of Mesh.cpp:58-186. */
pto_scene* pto_scene_create(const mipt_scene_desc* desc);
void pto_scene_destroy(pto_scene* s);
int64_t pto_scene_num_triangles(const pto_scene* s);
/* number of octree nodes and maximum subdivision depth over all meshes (Mesh.cpp:44-56) */
void pto_scene_octree_info(const pto_scene* s, int64_t* nodes, int32_t* max_depth, int64_t* leaf_triangles);
/* How much of a mesh its octree can still find.  CMesh::ComputeSubmeshes gives a triangle to an
 * octant only if one of its corners lies inside the octant's box (CMesh::AddTriangle,
 * Mesh.cpp:168-186), with the boxes computed from centre +- extent in fp32 (Mesh.cpp:95-141).  After
 * a child box has been re-fitted (RecomputeBB, Mesh.cpp:153-156) the faces of its own octants can
 * land one ulp inside the re-fitted box, and a triangle whose three corners lie on that face of the
 * box is then in no octant: the search (CMesh::Intersect, CPUPathTracer.cpp:195-249) only tests
 * leaf triangles, so it can never hit it.  mesh = -1 sums over all non-light meshes.
 * *total = triangles of the mesh(es); *reachable = distinct triangles present in at least one leaf;
 * mask (optional, one byte per global triangle id) = 1 where the triangle is in a leaf. */
void pto_scene_octree_reachable(const pto_scene* s, int32_t mesh, int64_t* total, int64_t* reachable, uint8_t* mask);

/* CCPUPathTracer::Trace search part, CPUPathTracer.cpp:251-285 */
int pto_trace(const pto_scene* s, const float o[3], const float d[3], float eps, int accel,
              pto_hit* hit);
/* batch form for tests: rays = 6 floats each; out_any = bHitOnly result */
void pto_trace_rays(const pto_scene* s, const float* rays, int64_t n, float eps, int accel,
                    int32_t* out_tri, float* out_tuv, uint8_t* out_any, int threads);

/*
 * CCPUPathTracer::CalcRays, CPUPathTracer.cpp:474-531, with its Radiance :348-446.
 * out_bgr255: 3 floats per pixel = 255 * rgb (the value the reference truncates to a
 * byte at :509); out_dword: the packed DWORD of :509; out_tri: primary-hit ids.
 * Any output may be NULL.  Only image rows h with h % row_step == row_phase are
 * computed (others are left untouched) so that a bounded sample can be timed.
 * threads <= 0 uses all OpenMP threads; the reference itself runs on one.
 */
int pto_render(const pto_scene* s, const mipt_camera* cam, const mipt_params* p, int accel,
               float* out_bgr255, uint32_t* out_dword, int32_t* out_tri,
               int32_t row_step, int32_t row_phase, int threads, pto_stats* stats);
/* The same for the columns [col_begin, col_end) of those rows only (col_end < 0: to the end of the row). */
int pto_render_window(const pto_scene* s, const mipt_camera* cam, const mipt_params* p, int accel,
                      float* out_bgr255, uint32_t* out_dword, int32_t* out_tri,
                      int32_t row_step, int32_t row_phase, int32_t col_begin, int32_t col_end, int threads, pto_stats* stats);

/* The estimator cut the way a multi-GPU run cuts it.  pto_accumulate: for the pixels of shard
 * (p->shard_rank of p->shard_count, square tiles dealt round-robin) and the first-bounce samples
 * [p->sample_begin, p->sample_end), the unclamped sum of r1 * L1 / pdf (CPUPathTracer.cpp:
 * 429-431), zero elsewhere; sums of different shards add.  pto_finalize applies
 * CPUPathTracer.cpp:433-444 and the *255 of :509 to the total. */
int pto_accumulate(const pto_scene* s, const mipt_camera* cam, const mipt_params* p, int accel,
                   float* out_accum, int threads);
int pto_finalize(const pto_scene* s, const mipt_camera* cam, const mipt_params* p, int accel,
                 const float* accum, float* out_bgr255, int threads);

#ifdef __cplusplus
}
#endif
#endif
