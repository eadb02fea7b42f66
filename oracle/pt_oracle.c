/*
 * pt_oracle.c - CPU restatement of the reference's CPU path tracer (see pt_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY - PARITY UNPINNED (no reference fixtures exist for this path).
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -fPIC -shared
 *
 * Vector helpers restate the DirectXMath SSE2 code paths that the reference's CPU
 * tracer goes through (DirectXMath is a Windows-SDK header, not part of the
 * reference tree): XMVector3Dot = (x*x' + y*y') + z*z'; XMVector3Cross =
 * (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x); XMVector3Normalize =
 * v / sqrt(dot(v,v)) with 0 for a zero vector and NaN for an infinite one;
 * XMVectorReciprocal = 1/x by IEEE division; XMVectorMin/Max = the SSE min/max
 * ((a<b)?a:b, (a>b)?a:b).
 */
#include "pt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PTO_MAX_BOUNCES 3          /* framework.h:60 */
static const float PTO_PI = 3.14159265f; /* CPUPathTracer.cpp:12 */
#define PTO_INFINITY FLT_MAX       /* MathHelper::Infinity, MathHelper.cpp */

typedef struct { float x, y, z; } f3;

static inline f3 f3_make(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f3 f3_ld(const float* p) { f3 r = { p[0], p[1], p[2] }; return r; }
static inline void f3_st(float* p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
static inline f3 f3_sub(f3 a, f3 b) { return f3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 f3_add(f3 a, f3 b) { return f3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 f3_scale(f3 a, float s) { return f3_make(a.x * s, a.y * s, a.z * s); }
static inline float f3_dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline f3 f3_cross(f3 a, f3 b)
{
    return f3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline f3 f3_normalize(f3 v)
{
    float lsq = f3_dot(v, v);
    float len = sqrtf(lsq);
    if (len == 0.0f) return f3_make(0.0f, 0.0f, 0.0f);
    if (lsq == INFINITY) return f3_make(NAN, NAN, NAN);
    return f3_make(v.x / len, v.y / len, v.z / len);
}
static inline float sse_min(float a, float b) { return (a < b) ? a : b; }
static inline float sse_max(float a, float b) { return (a > b) ? a : b; }

/* ------------------------------------------------------------------ */
/* xnacollision.cpp:1068-1168 */
int pto_intersect_ray_triangle(const float o[3], const float d[3], const float pv0[3],
                               const float pv1[3], const float pv2[3], float eps,
                               float* pt, float* pu, float* pv)
{
    f3 O = f3_ld(o), D = f3_ld(d), V0 = f3_ld(pv0), V1 = f3_ld(pv1), V2 = f3_ld(pv2);
    f3 e1 = f3_sub(V1, V0);                     /* :1080 */
    f3 e2 = f3_sub(V2, V0);                     /* :1081 */
    f3 p = f3_cross(D, e2);                     /* :1084 */
    float det = f3_dot(e1, p);                  /* :1087 */
    float u, v, t;
    if (det >= eps) {                           /* :1091 */
        f3 s = f3_sub(O, V0);                   /* :1094 */
        u = f3_dot(s, p);                       /* :1097 */
        int no = (u < 0.0f);                    /* :1099 */
        no |= (u > det);                        /* :1100 */
        f3 q = f3_cross(s, e1);                 /* :1103 */
        v = f3_dot(D, q);                       /* :1106 */
        no |= (v < 0.0f);                       /* :1108 */
        no |= ((u + v) > det);                  /* :1109 */
        t = f3_dot(e2, q);                      /* :1112 */
        no |= (t < 0.0f);                       /* :1114 */
        if (no) return 0;                       /* :1116 */
    } else if (det <= -eps) {                   /* :1119 */
        f3 s = f3_sub(O, V0);
        u = f3_dot(s, p);
        int no = (u > 0.0f);                    /* :1127 */
        no |= (u < det);                        /* :1128 */
        f3 q = f3_cross(s, e1);
        v = f3_dot(D, q);
        no |= (v > 0.0f);                       /* :1136 */
        no |= ((u + v) < det);                  /* :1137 */
        t = f3_dot(e2, q);
        no |= (t > 0.0f);                       /* :1142 */
        if (no) return 0;
    } else {
        return 0;                               /* :1147-1151 parallel ray */
    }
    float inv_det = 1.0f / det;                 /* :1153 */
    *pt = t * inv_det;                          /* :1155 */
    *pu = u * inv_det;                          /* :1157 */
    *pv = v * inv_det;                          /* :1158 */
    return 1;
}

/* xnacollision.cpp:1228-1297 */
int pto_intersect_ray_aabb(const float o[3], const float d[3], const float center[3],
                           const float extents[3], float* dist)
{
    const float eps = 1e-20f;
    float tmin[3], tmax[3];
    int no_parallel = 0;
    for (int i = 0; i < 3; i++) {
        float torigin = center[i] - o[i];                     /* :1252 */
        int par = (fabsf(d[i]) <= eps);                       /* :1260 */
        float inv = 1.0f / d[i];                              /* :1263 */
        float t1 = (torigin - extents[i]) * inv;              /* :1264 */
        float t2 = (torigin + extents[i]) * inv;              /* :1265 */
        tmin[i] = par ? -FLT_MAX : sse_min(t1, t2);           /* :1269 */
        tmax[i] = par ? FLT_MAX : sse_max(t1, t2);            /* :1270 */
        int overlap = (torigin <= extents[i]) && (-extents[i] <= torigin); /* :1286 */
        if (par && !overlap) no_parallel = 1;                 /* :1287 */
    }
    float t_min = sse_max(sse_max(tmin[0], tmin[1]), tmin[2]); /* :1274-1275 */
    float t_max = sse_min(sse_min(tmax[0], tmax[1]), tmax[2]); /* :1276-1277 */
    int no = (t_min > t_max);                                 /* :1280 */
    no |= (t_max < 0.0f);                                     /* :1283 */
    no |= no_parallel;
    if (!no) {
        if (dist) *dist = t_min;                              /* :1292 */
        return 1;
    }
    return 0;
}

/* xnacollision.cpp:964-972, XMVector3InBounds(P - C, E) */
int pto_point_in_aabb(const float p[3], const float center[3], const float extents[3])
{
    for (int i = 0; i < 3; i++) {
        float v = p[i] - center[i];
        if (!((v <= extents[i]) && (-extents[i] <= v))) return 0;
    }
    return 1;
}

/* CPUPathTracer.cpp:448-456 */
void pto_uniform_sample_hemisphere(float r1, float r2, float out[3])
{
    float sinTheta = sqrtf(1 - r1 * r1);
    float phi = 2 * PTO_PI * r2;
    float x = sinTheta * cosf(phi);
    float z = sinTheta * sinf(phi);
    out[0] = x; out[1] = r1; out[2] = z;
}

/* CPUPathTracer.cpp:458-472; _hypot works in double and the quotient divides by
 * its value converted to float */
void pto_create_coordinate_system(const float n[3], float nt[3], float nb[3])
{
    f3 N = f3_ld(n);
    f3 Nt;
    if (fabsf(N.x) > fabsf(N.y)) {
        float h = (float)hypot((double)N.x, (double)N.z);
        Nt = f3_make(N.z / h, 0.0f / h, -N.x / h);
    } else {
        float h = (float)hypot((double)N.y, (double)N.z);
        Nt = f3_make(0.0f / h, -N.z / h, N.y / h);
    }
    f3_st(nt, Nt);
    f3_st(nb, f3_cross(N, Nt));
}

/* Counter-based uniform in [0,1).  The reference draws from one global mt19937
 * (CPUPathTracer.cpp:345-346) and cuRAND XORWOW (PathTracer.cu:516-518); the harness
 * replaces both with this keyed hash so that any partition of the work sees the
 * same numbers. */
static inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
float pto_rng_u01(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t dim)
{
    uint32_t h = mix32(seed + 0x9e3779b9U);
    h = mix32(h ^ pixel);
    h = mix32(h ^ sample);
    h = mix32(h ^ (depth * 2U + dim + 0x632be5abU));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

/* CPUPathTracer.cpp:489-497.  XMVector3TransformCoord / TransformNormal evaluate
 * ((z*r2 [+ r3]) + y*r1) + x*r0; the w divide of TransformCoord is kept. */
void pto_primary_ray(const mipt_camera* cam, int32_t j, int32_t h, float o[3], float d[3])
{
    const float* m = cam->to_local;
    float vx = (+2.0f * (float)cam->div * j / (float)cam->client_width - 1.0f) / cam->p0;
    float vy = (-2.0f * (float)cam->div * h / (float)cam->client_height + 1.0f) / cam->p1;
    /* origin (0,0,0,1) */
    float r[4];
    for (int c = 0; c < 4; c++) r[c] = ((0.0f * m[8 + c] + m[12 + c]) + 0.0f * m[4 + c]) + 0.0f * m[0 + c];
    o[0] = r[0] / r[3]; o[1] = r[1] / r[3]; o[2] = r[2] / r[3];
    f3 dir;
    dir.x = (1.0f * m[8 + 0] + vy * m[4 + 0]) + vx * m[0 + 0];
    dir.y = (1.0f * m[8 + 1] + vy * m[4 + 1]) + vx * m[0 + 1];
    dir.z = (1.0f * m[8 + 2] + vy * m[4 + 2]) + vx * m[0 + 2];
    dir = f3_normalize(dir);
    f3_st(d, dir);
}

/* ------------------------------------------------------------------ */
/* Scene: CMesh restated (framework.h:102-158) */
typedef struct pto_node {
    f3 bbmin, bbmax;          /* CMesh::bbmin/bbmax */
    float center[3], extents[3]; /* CMesh::boundingbox */
    int32_t ntris;            /* nNumTriangles */
    int32_t cap;
    mipt_vertex* tris;        /* vTriangles, 3 per triangle */
    int32_t* ids;             /* global triangle id of each entry */
    struct pto_node* child;   /* m_pMesh: 8 children or NULL */
} pto_node;

typedef struct pto_mesh {
    pto_node root;
    int32_t material;
    int32_t is_light;
} pto_mesh;

typedef struct pto_mat {
    /* A texture is held in the layout it was given in: DWORD texels, byte0 = blue (CMesh::m_pTexData,
       framework.h:130-132, what the CPU path reads) or three floats per texel, .x = blue
       (CUDAMaterial::pTexData, PathTracer.h:21, what the GPU path reads).  Exactly one of the two
       pointers of a texture is set. */
    uint32_t* tex; float* texf; int32_t w, h;
    uint32_t* nml; float* nmlf; int32_t nw, nh;
    float diffuse[3];
} pto_mat;

struct pto_scene {
    pto_mesh* meshes; int32_t nmeshes;
    pto_mat* mats; int32_t nmats;
    int64_t ntris;
};

static void node_push(pto_node* n, const mipt_vertex* v3, int32_t id)
{
    if (n->ntris == n->cap) {
        n->cap = n->cap ? n->cap * 2 : 16;
        n->tris = (mipt_vertex*)realloc(n->tris, sizeof(mipt_vertex) * 3 * (size_t)n->cap);
        n->ids = (int32_t*)realloc(n->ids, sizeof(int32_t) * (size_t)n->cap);
    }
    memcpy(&n->tris[3 * (size_t)n->ntris], v3, sizeof(mipt_vertex) * 3);
    n->ids[n->ntris] = id;
    n->ntris++;
}

static void set_box_from_minmax(pto_node* n)
{
    n->center[0] = (n->bbmin.x + n->bbmax.x) / 2; n->center[1] = (n->bbmin.y + n->bbmax.y) / 2;
    n->center[2] = (n->bbmin.z + n->bbmax.z) / 2;
    n->extents[0] = (n->bbmax.x - n->bbmin.x) / 2; n->extents[1] = (n->bbmax.y - n->bbmin.y) / 2;
    n->extents[2] = (n->bbmax.z - n->bbmin.z) / 2;
}

#define GROW(P) do { \
    if ((P).x < bbmin->x) bbmin->x = (P).x; else if ((P).x > bbmax->x) bbmax->x = (P).x; \
    if ((P).y < bbmin->y) bbmin->y = (P).y; else if ((P).y > bbmax->y) bbmax->y = (P).y; \
    if ((P).z < bbmin->z) bbmin->z = (P).z; else if ((P).z > bbmax->z) bbmax->z = (P).z; } while (0)

/* Mesh.cpp:58-86 (also the bounding-box loop of MainFrm.cpp:369-416) */
static void recompute_bb(pto_node* n)
{
    f3* bbmin = &n->bbmin; f3* bbmax = &n->bbmax;
    for (int32_t h = 0; h < n->ntris; h++) {
        f3 p0 = f3_ld(n->tris[3 * h + 0].pos), p1 = f3_ld(n->tris[3 * h + 1].pos),
           p2 = f3_ld(n->tris[3 * h + 2].pos);
        if (h == 0) { *bbmin = p0; *bbmax = p0; }
        GROW(p0); GROW(p1); GROW(p2);
    }
    set_box_from_minmax(n);
}

/* Mesh.cpp:88-166 */
static void compute_submeshes(pto_node* n, long depth)
{
    n->child = (pto_node*)calloc(8, sizeof(pto_node));
    float ex = n->extents[0] / 2, ey = n->extents[1] / 2, ez = n->extents[2] / 2;
    for (int h = 0; h < 8; h++) {
        pto_node* c = &n->child[h];
        c->center[0] = (h & 1) ? n->center[0] + ex : n->center[0] - ex;   /* :95-125 */
        c->center[1] = (h & 2) ? n->center[1] + ey : n->center[1] - ey;
        c->center[2] = (h & 4) ? n->center[2] + ez : n->center[2] - ez;
        c->extents[0] = ex; c->extents[1] = ey; c->extents[2] = ez;       /* :127-132 */
        c->bbmax = f3_make(c->center[0] + ex, c->center[1] + ey, c->center[2] + ez); /* :136-141 */
        c->bbmin = f3_make(c->center[0] - ex, c->center[1] - ey, c->center[2] - ez);
    }
    for (int h = 0; h < 8; h++) {                                         /* :144-152 */
        pto_node* c = &n->child[h];
        for (int32_t j = 0; j < n->ntris; j++) {
            const mipt_vertex* t = &n->tris[3 * (size_t)j];
            /* CMesh::AddTriangle, Mesh.cpp:168-186 */
            if (pto_point_in_aabb(t[0].pos, c->center, c->extents) ||
                pto_point_in_aabb(t[1].pos, c->center, c->extents) ||
                pto_point_in_aabb(t[2].pos, c->center, c->extents))
                node_push(c, t, n->ids[j]);
        }
    }
    for (int h = 0; h < 8; h++) recompute_bb(&n->child[h]);               /* :153-156 */
    if (depth > 4) return;                                                /* :157 */
    for (int h = 0; h < 8; h++)
        if (n->child[h].ntris >= 200) compute_submeshes(&n->child[h], depth + 1); /* :159-165 */
}

static void node_free(pto_node* n)
{
    if (n->child) { for (int h = 0; h < 8; h++) node_free(&n->child[h]); free(n->child); }
    free(n->tris); free(n->ids);
}

static void copy_texels(const void* data, int32_t w, int32_t h, int32_t fmt, uint32_t** dw, float** fl)
{
    *dw = NULL; *fl = NULL;
    if (!data || w <= 0 || h <= 0 || fmt == MIPT_TEX_NONE) return;
    size_t n = (size_t)w * (size_t)h;
    if (fmt == MIPT_TEX_BGRA8) { *dw = (uint32_t*)malloc(n * 4); memcpy(*dw, data, n * 4); }
    else { *fl = (float*)malloc(n * 12); memcpy(*fl, data, n * 12); }   /* RayTracerView.cpp:819-842 */
}

/* The three channel values of a texel as floats, .x = blue.  From a DWORD texel they are the bytes
   (CPUPathTracer.cpp:307-309, 331-334); a float texel is taken as it is (PathTracer.cu:311-313, 344-346). */
static inline f3 texel_bgr(const uint32_t* dw, const float* fl, long idx)
{
    if (dw) { uint32_t t = dw[idx]; return f3_make((float)(t & 255u), (float)((t >> 8) & 255u), (float)((t >> 16) & 255u)); }
    return f3_make(fl[3 * idx], fl[3 * idx + 1], fl[3 * idx + 2]);
}

pto_scene* pto_scene_create(const mipt_scene_desc* desc)
{
    pto_scene* s = (pto_scene*)calloc(1, sizeof(pto_scene));
    s->nmats = desc->num_materials;
    s->mats = (pto_mat*)calloc((size_t)(s->nmats > 0 ? s->nmats : 1), sizeof(pto_mat));
    for (int32_t i = 0; i < s->nmats; i++) {
        const mipt_material* m = &desc->materials[i];
        pto_mat* o = &s->mats[i];
        copy_texels(m->tex_data, m->tex_width, m->tex_height, m->tex_format, &o->tex, &o->texf);
        if (o->tex || o->texf) { o->w = m->tex_width; o->h = m->tex_height; }
        copy_texels(m->nml_data, m->nml_width, m->nml_height, m->nml_format, &o->nml, &o->nmlf);
        if (o->nml || o->nmlf) { o->nw = m->nml_width; o->nh = m->nml_height; }
        memcpy(o->diffuse, m->diffuse, sizeof(o->diffuse));
    }
    s->nmeshes = desc->num_meshes;
    s->meshes = (pto_mesh*)calloc((size_t)(s->nmeshes > 0 ? s->nmeshes : 1), sizeof(pto_mesh));
    int32_t next_id = 0;
    for (int32_t i = 0; i < s->nmeshes; i++) {
        const mipt_mesh* m = &desc->meshes[i];
        pto_mesh* o = &s->meshes[i];
        o->material = m->material; o->is_light = m->is_light;
        pto_node* r = &o->root;
        for (int64_t k = 0; k < m->num_triangles; k++) {
            int32_t id = m->is_light ? -1 : next_id++;
            node_push(r, &desc->vertices[3 * (m->first_triangle + k)], id);
        }
        recompute_bb(r);                                    /* MainFrm.cpp:369-416, 463-470 */
        if (r->ntris >= 200) compute_submeshes(r, 0);       /* MainFrm.cpp:471-475 */
    }
    s->ntris = next_id;
    return s;
}

void pto_scene_destroy(pto_scene* s)
{
    if (!s) return;
    for (int32_t i = 0; i < s->nmeshes; i++) node_free(&s->meshes[i].root);
    for (int32_t i = 0; i < s->nmats; i++) { free(s->mats[i].tex); free(s->mats[i].nml); free(s->mats[i].texf); free(s->mats[i].nmlf); }
    free(s->meshes); free(s->mats); free(s);
}

int64_t pto_scene_num_triangles(const pto_scene* s) { return s->ntris; }

static void node_info(const pto_node* n, int32_t depth, int64_t* nodes, int32_t* maxd, int64_t* leaf)
{
    (*nodes)++;
    if (depth > *maxd) *maxd = depth;
    if (n->child) for (int h = 0; h < 8; h++) node_info(&n->child[h], depth + 1, nodes, maxd, leaf);
    else *leaf += n->ntris;
}
void pto_scene_octree_info(const pto_scene* s, int64_t* nodes, int32_t* max_depth, int64_t* leaf_triangles)
{
    int64_t nn = 0, lf = 0; int32_t md = 0;
    for (int32_t i = 0; i < s->nmeshes; i++) node_info(&s->meshes[i].root, 0, &nn, &md, &lf);
    if (nodes) *nodes = nn;
    if (max_depth) *max_depth = md;
    if (leaf_triangles) *leaf_triangles = lf;
}

static void node_mark_leaves(const pto_node* n, uint8_t* seen)
{
    if (n->child) { for (int h = 0; h < 8; h++) node_mark_leaves(&n->child[h], seen); return; }
    for (int32_t k = 0; k < n->ntris; k++) if (n->ids[k] >= 0) seen[n->ids[k]] = 1;
}
void pto_scene_octree_reachable(const pto_scene* s, int32_t mesh, int64_t* total, int64_t* reachable, uint8_t* mask)
{
    uint8_t* seen = (uint8_t*)calloc((size_t)(s->ntris > 0 ? s->ntris : 1), 1);
    int64_t tot = 0, got = 0;
    for (int32_t i = 0; i < s->nmeshes; i++) {
        if (s->meshes[i].is_light || (mesh >= 0 && i != mesh)) continue;
        tot += s->meshes[i].root.ntris;
        node_mark_leaves(&s->meshes[i].root, seen);
    }
    for (int64_t k = 0; k < s->ntris; k++) got += seen[k];
    if (mask) memcpy(mask, seen, (size_t)s->ntris);
    free(seen);
    if (total) *total = tot;
    if (reachable) *reachable = got;
}

/* ------------------------------------------------------------------ */
typedef struct {
    float dist; float u, v; f3 hitpoint, nml; int32_t tri; float ua, va;
    uint64_t tri_tests, box_tests;
    float eps; int brute;
} search;

/* the update block of CMesh::Intersect, CPUPathTracer.cpp:229-245 */
static inline void record_hit(search* S, const mipt_vertex* T, int32_t id, float t, float ua, float va)
{
    S->dist = t;
    float w = 1 - ua - va;
    S->u = w * T[0].tex[1] + ua * T[1].tex[1] + va * T[2].tex[1];       /* :231 (tex.y) */
    S->v = w * T[0].tex[0] + ua * T[1].tex[0] + va * T[2].tex[0];       /* :232 (tex.x) */
    S->hitpoint.x = w * T[0].pos[0] + ua * T[1].pos[0] + va * T[2].pos[0];
    S->hitpoint.y = w * T[0].pos[1] + ua * T[1].pos[1] + va * T[2].pos[1];
    S->hitpoint.z = w * T[0].pos[2] + ua * T[1].pos[2] + va * T[2].pos[2];
    f3 n;
    n.x = (T[0].normal[0] + T[1].normal[0] + T[2].normal[0]) / 3;       /* :239-241 */
    n.y = (T[0].normal[1] + T[1].normal[1] + T[2].normal[1]) / 3;
    n.z = (T[0].normal[2] + T[1].normal[2] + T[2].normal[2]) / 3;
    S->nml = f3_normalize(n);                                           /* :243 */
    S->tri = id; S->ua = ua; S->va = va;
}

/* CMesh::Intersect, CPUPathTracer.cpp:195-249 */
static int node_intersect(const pto_node* n, const float o[3], const float d[3], search* S)
{
    float dist2 = 0;
    S->box_tests++;
    if (!pto_intersect_ray_aabb(o, d, n->center, n->extents, &dist2)) return 0;   /* :200 */
    if (n->child) {                                                               /* :204-214 */
        int hit = 0;
        for (int h = 0; h < 8; h++)
            if (node_intersect(&n->child[h], o, d, S)) hit = 1;
        return hit;
    }
    int hit = 0;
    for (int32_t k = 0; k < n->ntris; k++) {                                      /* :217-247 */
        float t = 0.0f, ua = 0, va = 0;
        const mipt_vertex* T = &n->tris[3 * (size_t)k];
        S->tri_tests++;
        if (pto_intersect_ray_triangle(o, d, T[0].pos, T[1].pos, T[2].pos, S->eps, &t, &ua, &va)) {
            if (t < S->dist) { record_hit(S, T, n->ids[k], t, ua, va); hit = 1; }   /* :229 */
        }
    }
    return hit;
}

/* every triangle, no box test, lowest id wins on equal t */
static int node_brute(const pto_node* n, const float o[3], const float d[3], search* S)
{
    int hit = 0;
    for (int32_t k = 0; k < n->ntris; k++) {
        float t = 0.0f, ua = 0, va = 0;
        const mipt_vertex* T = &n->tris[3 * (size_t)k];
        S->tri_tests++;
        if (pto_intersect_ray_triangle(o, d, T[0].pos, T[1].pos, T[2].pos, S->eps, &t, &ua, &va)) {
            if (t < S->dist || (t == S->dist && S->tri >= 0 && n->ids[k] < S->tri)) {
                record_hit(S, T, n->ids[k], t, ua, va); hit = 1;
            }
        }
    }
    return hit;
}

/* search part of CCPUPathTracer::Trace, CPUPathTracer.cpp:251-285 */
static int trace_search(const pto_scene* s, const float o[3], const float d[3], search* S, int32_t* mesh)
{
    int hit = 0;
    S->dist = PTO_INFINITY; S->tri = -1; S->u = S->v = 0;
    S->hitpoint = f3_make(0, 0, 0); S->nml = f3_make(0, 0, 0);
    *mesh = -1;
    for (int32_t i = 0; i < s->nmeshes; i++) {
        const pto_mesh* m = &s->meshes[i];
        if (m->is_light) continue;                                   /* :268 */
        int h = S->brute ? node_brute(&m->root, o, d, S) : node_intersect(&m->root, o, d, S);
        if (!h) continue;                                            /* :271-273 */
        hit = 1; *mesh = i;
    }
    return hit;
}

int pto_trace(const pto_scene* s, const float o[3], const float d[3], float eps, int accel, pto_hit* hit)
{
    search S; memset(&S, 0, sizeof(S));
    S.eps = eps; S.brute = (accel == PTO_ACCEL_BRUTE);
    int32_t mesh;
    int h = trace_search(s, o, d, &S, &mesh);
    if (hit) {
        hit->tri = h ? S.tri : -1; hit->t = S.dist; hit->ua = S.ua; hit->va = S.va;
        hit->u = S.u; hit->v = S.v; f3_st(hit->hitpoint, S.hitpoint); f3_st(hit->normal, S.nml);
        hit->mesh = mesh;
    }
    return h;
}

void pto_trace_rays(const pto_scene* s, const float* rays, int64_t n, float eps, int accel,
                    int32_t* out_tri, float* out_tuv, uint8_t* out_any, int threads)
{
#ifdef _OPENMP
    int nt = threads > 0 ? threads : omp_get_max_threads();
#else
    int nt = 1; (void)threads;
#endif
    #pragma omp parallel for schedule(dynamic, 16) num_threads(nt)
    for (int64_t i = 0; i < n; i++) {
        pto_hit h;
        int hit = pto_trace(s, &rays[6 * i], &rays[6 * i + 3], eps, accel, &h);
        if (out_tri) out_tri[i] = hit ? h.tri : -1;
        if (out_tuv) { out_tuv[3 * i] = hit ? h.t : 0; out_tuv[3 * i + 1] = hit ? h.ua : 0; out_tuv[3 * i + 2] = hit ? h.va : 0; }
        if (out_any) out_any[i] = (uint8_t)hit;      /* bHitOnly, CPUPathTracer.cpp:288 */
    }
}

/* ------------------------------------------------------------------ */
typedef struct {
    const pto_scene* s; const mipt_params* p; int accel;
    uint64_t closest, shadow, tri_tests, box_tests;
} rctx;

/* CCPUPathTracer::Trace, CPUPathTracer.cpp:251-344 */
static int trace_full(rctx* R, f3 o, f3 d, int hit_only, f3* hitpoint, f3* nmlMesh, f3* nmlMap, f3* rgb,
                      int32_t* tri)
{
    search S; memset(&S, 0, sizeof(S));
    S.eps = R->p->epsilon; S.brute = (R->accel == PTO_ACCEL_BRUTE);
    float oo[3], dd[3]; f3_st(oo, o); f3_st(dd, d);
    int32_t mesh;
    int hit = trace_search(R->s, oo, dd, &S, &mesh);
    R->tri_tests += S.tri_tests; R->box_tests += S.box_tests;
    if (hit_only) R->shadow++; else R->closest++;
    if (tri) *tri = hit ? S.tri : -1;
    if (!hit) return 0;
    if (hit_only) return 1;                                          /* :288 */
    *hitpoint = S.hitpoint; *nmlMesh = S.nml;
    float u = S.u, v = S.v;
    const pto_mesh* m = &R->s->meshes[mesh];
    const pto_mat* mat = (m->material >= 0 && m->material < R->s->nmats) ? &R->s->mats[m->material] : NULL;
    long nTexWidth = mat ? mat->w : 0, nTexHeight = mat ? mat->h : 0;
    long nNmlWidth = mat ? mat->nw : 0, nNmlHeight = mat ? mat->nh : 0;
    const int hasTex = mat && (mat->tex || mat->texf), hasNml = mat && (mat->nml || mat->nmlf);
    /* wrap loops :292-295; values the loop cannot bring into range (NaN, |x| >= 2^24) are
       set to 0 so that the loops terminate */
    if (!(fabsf(u) < 16777216.0f)) u = 0;
    if (!(fabsf(v) < 16777216.0f)) v = 0;
    while (u > 1) { u = u - 1; }
    while (v > 1) { v = v - 1; }
    while (u < 0) { u = u + 1; }
    while (v < 0) { v = v + 1; }
    long j = (long)(nTexWidth * v);                                  /* :296-299 */
    long h = (long)(nTexHeight * u);
    j = (j < nTexWidth - 1) ? j : nTexWidth - 1;
    h = (h < nTexHeight - 1) ? h : nTexHeight - 1;
    if (hasNml) {                                                    /* :301-318 */
        long jn = (long)(nNmlWidth * v);
        long hn = (long)(nNmlHeight * u);
        jn = (jn < nNmlWidth - 1) ? jn : nNmlWidth - 1;
        hn = (hn < nNmlHeight - 1) ? hn : nNmlHeight - 1;
        f3 t = texel_bgr(mat->nml, mat->nmlf, hn * nNmlWidth + jn);
        float x = t.x, y = t.y, z = t.z;
        nmlMap->x = 2 * z / 255.0f - 1;
        nmlMap->y = (2 * x / 255.0f - 1);
        nmlMap->z = 2 * y / 255.0f - 1;
        if (R->p->normalize_normal_map) *nmlMap = f3_normalize(*nmlMap); /* PathTracer.cu:314 flavour */
    }
    if (!R->p->use_textures) {                                       /* :320-326 */
        *rgb = f3_make(0.5f, 0.5f, 0.5f);
    } else if (hasTex && nTexWidth > 0) {                            /* :330-335 */
        f3 t = texel_bgr(mat->tex, mat->texf, h * nTexWidth + j);
        rgb->x = t.x / 255.0f;
        rgb->y = t.y / 255.0f;
        rgb->z = t.z / 255.0f;
    } else {
        *rgb = mat ? f3_ld(mat->diffuse) : f3_make(0, 0, 0);         /* :338 */
    }
    return 1;
}

/* CCPUPathTracer::Radiance, CPUPathTracer.cpp:348-446.  (pixel, sample) key the
 * random stream: sample is the index of the first-bounce loop iteration the call
 * descends from. */
static f3 radiance(rctx* R, long nNumSamples, f3 rayOrigin, f3 rayDir, int depth, uint32_t pixel,
                   uint32_t sample, int32_t* primary_tri)
{
    const mipt_params* P = R->p;
    f3 rgb = { 0, 0, 0 }, nmlMesh = { 0, 0, 0 }, nmlMap = { 0, 1, 0 }, hitpoint = { 0, 0, 0 };
    if (depth > PTO_MAX_BOUNCES) return rgb;                         /* :357 */
    f3 direct = { 0, 0, 0 }, indirect = { 0, 0, 0 };
    int hit = trace_full(R, rayOrigin, rayDir, 0, &hitpoint, &nmlMesh, &nmlMap, &rgb, primary_tri); /* :365 */
    if (hit) {
        f3 sunPos = f3_ld(P->sun_pos);
        f3 sunDir = f3_normalize(f3_sub(sunPos, hitpoint));          /* :374-375 */
        float bias = 0.01f;
        f3 hitNormalMesh = nmlMesh;
        float n3[3], nt3[3], nb3[3];
        f3_st(n3, hitNormalMesh);
        pto_create_coordinate_system(n3, nt3, nb3);                  /* :389 */
        f3 Nt = f3_ld(nt3), Nb = f3_ld(nb3);
        if (P->use_normals) {                                        /* :390-400 */
            f3 m = nmlMap;
            f3 nn;
            nn.x = m.x * Nb.x + m.y * hitNormalMesh.x + m.z * Nt.x;
            nn.y = m.x * Nb.y + m.y * hitNormalMesh.y + m.z * Nt.y;
            nn.z = m.x * Nb.z + m.y * hitNormalMesh.z + m.z * Nt.z;
            hitNormalMesh = f3_normalize(nn);
            f3_st(n3, hitNormalMesh);
            pto_create_coordinate_system(n3, nt3, nb3);
            Nt = f3_ld(nt3); Nb = f3_ld(nb3);
        }
        f3 so = f3_add(hitpoint, f3_scale(hitNormalMesh, bias));     /* :402 */
        f3 ig1, ig2, ig3, ig4;
        int shadowed = trace_full(R, so, sunDir, 1, &ig1, &ig2, &ig3, &ig4, NULL);
        if (!shadowed) {                                             /* :403-410 */
            float dp = f3_dot(hitNormalMesh, sunDir);
            float l = P->sun_intensity * ((0.0f > dp) ? 0.0f : dp);
            direct = f3_make(l, l, l);
        }
        if (P->global_illumination) {                                /* :412-436 */
            uint32_t N = (uint32_t)nNumSamples;
            float pdf = 1 / (2 * PTO_PI);
            for (uint32_t n = 0; n < N; ++n) {
                uint32_t key = (depth == 0) ? n : sample;
                float r1 = pto_rng_u01(P->seed, pixel, key, (uint32_t)depth, 0);
                float r2 = pto_rng_u01(P->seed, pixel, key, (uint32_t)depth, 1);
                float smp[3];
                pto_uniform_sample_hemisphere(r1, r2, smp);
                f3 sw;
                sw.x = smp[0] * Nb.x + smp[1] * hitNormalMesh.x + smp[2] * Nt.x;
                sw.y = smp[0] * Nb.y + smp[1] * hitNormalMesh.y + smp[2] * Nt.y;
                sw.z = smp[0] * Nb.z + smp[1] * hitNormalMesh.z + smp[2] * Nt.z;
                f3 rd = radiance(R, 1, f3_add(hitpoint, f3_scale(sw, bias)), sw, depth + 1, pixel, key, NULL);
                indirect.x = indirect.x + r1 * rd.x / pdf;
                indirect.y = indirect.y + r1 * rd.y / pdf;
                indirect.z = indirect.z + r1 * rd.z / pdf;
            }
            indirect.x = indirect.x / N; indirect.y = indirect.y / N; indirect.z = indirect.z / N;
        }
        rgb.x = (direct.x + P->sun_intensity * indirect.x) * rgb.x / PTO_PI;   /* :438-440 */
        rgb.y = (direct.y + P->sun_intensity * indirect.y) * rgb.y / PTO_PI;
        rgb.z = (direct.z + P->sun_intensity * indirect.z) * rgb.z / PTO_PI;
    }
    rgb.x = (rgb.x < 1.0f) ? rgb.x : 1.0f;                           /* :442-444 */
    rgb.y = (rgb.y < 1.0f) ? rgb.y : 1.0f;
    rgb.z = (rgb.z < 1.0f) ? rgb.z : 1.0f;
    return rgb;
}

static double now_s(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* CCPUPathTracer::CalcRays, CPUPathTracer.cpp:474-531.  The reference's loop nest is
 * x-outer / y-inner and serial (:482-486); pixels are independent once the random
 * stream is keyed per pixel, so rows are spread over OpenMP threads here. */
int pto_render(const pto_scene* s, const mipt_camera* cam, const mipt_params* p, int accel,
               float* out_bgr255, uint32_t* out_dword, int32_t* out_tri,
               int32_t row_step, int32_t row_phase, int threads, pto_stats* stats)
{
    return pto_render_window(s, cam, p, accel, out_bgr255, out_dword, out_tri, row_step, row_phase, 0, -1, threads, stats);
}

/* The same, for the columns [col_begin, col_end) of the sampled rows only (col_end < 0: to the end of the
 * row): a short run of pixels at full sample count is what a brute-force search can afford. */
int pto_render_window(const pto_scene* s, const mipt_camera* cam, const mipt_params* p, int accel,
                      float* out_bgr255, uint32_t* out_dword, int32_t* out_tri,
                      int32_t row_step, int32_t row_phase, int32_t col_begin, int32_t col_end, int threads, pto_stats* stats)
{
    if (!s || !cam || !p || cam->div <= 0) return -1;
    long W = cam->client_width / cam->div, H = cam->client_height / cam->div;   /* :478-479 */
    if (row_step <= 0) row_step = 1;
    if (col_end < 0 || col_end > W) col_end = (int32_t)W;
    if (col_begin < 0) col_begin = 0;
    if (col_begin > col_end) col_begin = col_end;
    const long C0 = col_begin, CW = col_end - col_begin;
    long nNumSamples = p->num_samples;
    if (nNumSamples < 1) nNumSamples = 1;                        /* RayTracerView.cpp:950-951 */
    if (nNumSamples > 10000) nNumSamples = 10000;
    uint64_t closest = 0, shadow = 0, tt = 0, bt = 0;
#ifdef _OPENMP
    int nt = threads > 0 ? threads : omp_get_max_threads();
#else
    int nt = 1; (void)threads;
#endif
    double t0 = now_s();
    /* work items are 32-pixel pieces of the sampled rows, so that a sample of a few rows still
     * keeps every thread busy; pixels are independent (the random stream is keyed by the pixel) */
    const long PIECE = CW >= 1024 ? 32 : (CW >= 128 ? 4 : 1);
    long nRows = 0;
    for (long h = 0; h < H; ++h) if ((h % row_step) == row_phase) nRows++;
    const long piecesPerRow = (CW + PIECE - 1) / PIECE;
    #pragma omp parallel for schedule(dynamic, 1) num_threads(nt) reduction(+:closest,shadow,tt,bt)
    for (long item = 0; item < nRows * piecesPerRow; ++item) {
        long h = row_phase + (item / piecesPerRow) * row_step;
        long j0 = C0 + (item % piecesPerRow) * PIECE, j1 = j0 + PIECE < C0 + CW ? j0 + PIECE : C0 + CW;
        rctx R; R.s = s; R.p = p; R.accel = accel; R.closest = R.shadow = R.tri_tests = R.box_tests = 0;
        for (long j = j0; j < j1; ++j) {
            float o[3], d[3];
            pto_primary_ray(cam, (int32_t)j, (int32_t)h, o, d);
            int32_t tri = -1;
            uint32_t pixel = (uint32_t)(h * W + j);
            f3 rgb = radiance(&R, nNumSamples, f3_ld(o), f3_ld(d), 0, pixel, 0, &tri);
            size_t idx = (size_t)(h * W + j);
            if (out_bgr255) { out_bgr255[3 * idx] = 255 * rgb.x; out_bgr255[3 * idx + 1] = 255 * rgb.y; out_bgr255[3 * idx + 2] = 255 * rgb.z; }
            if (out_dword) {                                     /* :509 MAKELONG(MAKEWORD(b,g),MAKEWORD(r,255)) */
                uint32_t b = (uint32_t)(uint64_t)(rgb.x * 255) & 255u, g = (uint32_t)(uint64_t)(rgb.y * 255) & 255u,
                         r = (uint32_t)(uint64_t)(rgb.z * 255) & 255u;
                out_dword[idx] = b | (g << 8) | (r << 16) | (255u << 24);
            }
            if (out_tri) out_tri[idx] = tri;
        }
        closest += R.closest; shadow += R.shadow; tt += R.tri_tests; bt += R.box_tests;
    }
    if (stats) {
        stats->closest_rays = closest; stats->shadow_rays = shadow; stats->triangle_tests = tt;
        stats->box_tests = bt; stats->seconds = now_s() - t0; stats->threads = nt;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* The same estimator cut the way a multi-GPU run cuts it: pto_accumulate produces, for the
 * pixels of one shard and the first-bounce samples [sample_begin, sample_end), the
 * unclamped sum of r1 * L1 / pdf (the running value of indirectLighting at
 * CPUPathTracer.cpp:429-431 before the division by N at :433-435); sums of different
 * shards add; pto_finalize applies :433-444 and the *255 of :509 to the total. */
static int pixel_owned(const mipt_params* p, long W, long j, long h)
{
    int count = p->shard_count > 1 ? p->shard_count : 1;
    if (count == 1) return 1;
    int ts = p->tile_size > 0 ? p->tile_size : 32;
    long tilesX = (W + ts - 1) / ts;
    long tile = (h / ts) * tilesX + (j / ts);
    return (tile % count) == p->shard_rank;
}

/* depth-0 part of Radiance up to the sample loop: returns 0 on a miss */
static int primary_surface(rctx* R, const mipt_camera* cam, long j, long h, f3* hitpoint, f3* N, f3* Nt, f3* Nb,
                           f3* albedo, f3* direct)
{
    const mipt_params* P = R->p;
    float o[3], d[3];
    pto_primary_ray(cam, (int32_t)j, (int32_t)h, o, d);
    f3 nmlMesh = { 0, 0, 0 }, nmlMap = { 0, 1, 0 };
    *direct = f3_make(0, 0, 0);
    if (!trace_full(R, f3_ld(o), f3_ld(d), 0, hitpoint, &nmlMesh, &nmlMap, albedo, NULL)) return 0;
    f3 sunDir = f3_normalize(f3_sub(f3_ld(P->sun_pos), *hitpoint));
    float n3[3], nt3[3], nb3[3];
    f3 hn = nmlMesh;
    f3_st(n3, hn); pto_create_coordinate_system(n3, nt3, nb3);
    *Nt = f3_ld(nt3); *Nb = f3_ld(nb3);
    if (P->use_normals) {
        f3 m = nmlMap, nn;
        nn.x = m.x * Nb->x + m.y * hn.x + m.z * Nt->x;
        nn.y = m.x * Nb->y + m.y * hn.y + m.z * Nt->y;
        nn.z = m.x * Nb->z + m.y * hn.z + m.z * Nt->z;
        hn = f3_normalize(nn);
        f3_st(n3, hn); pto_create_coordinate_system(n3, nt3, nb3);
        *Nt = f3_ld(nt3); *Nb = f3_ld(nb3);
    }
    *N = hn;
    f3 ig1, ig2, ig3, ig4;
    if (!trace_full(R, f3_add(*hitpoint, f3_scale(hn, 0.01f)), sunDir, 1, &ig1, &ig2, &ig3, &ig4, NULL)) {
        float dp = f3_dot(hn, sunDir);
        float l = P->sun_intensity * ((0.0f > dp) ? 0.0f : dp);
        *direct = f3_make(l, l, l);
    }
    return 1;
}

int pto_accumulate(const pto_scene* s, const mipt_camera* cam, const mipt_params* p, int accel, float* out_accum, int threads)
{
    if (!s || !cam || !p || !out_accum || cam->div <= 0) return -1;
    long W = cam->client_width / cam->div, H = cam->client_height / cam->div;
    long ns = p->num_samples < 1 ? 1 : (p->num_samples > 10000 ? 10000 : p->num_samples);
    long sb = p->sample_begin, se = p->sample_end;
    if (sb == 0 && se == 0) se = ns;
#ifdef _OPENMP
    int nt = threads > 0 ? threads : omp_get_max_threads();
#else
    int nt = 1; (void)threads;
#endif
    #pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
    for (long h = 0; h < H; ++h) {
        rctx R; R.s = s; R.p = p; R.accel = accel; R.closest = R.shadow = R.tri_tests = R.box_tests = 0;
        for (long j = 0; j < W; ++j) {
            size_t idx = (size_t)(h * W + j);
            f3 sum = { 0, 0, 0 };
            f3 hp, N, Nt, Nb, albedo, direct;
            if (p->global_illumination && pixel_owned(p, W, j, h) && primary_surface(&R, cam, j, h, &hp, &N, &Nt, &Nb, &albedo, &direct)) {
                float pdf = 1 / (2 * PTO_PI);
                for (long n = sb; n < se; ++n) {
                    float r1 = pto_rng_u01(p->seed, (uint32_t)idx, (uint32_t)n, 0, 0);
                    float r2 = pto_rng_u01(p->seed, (uint32_t)idx, (uint32_t)n, 0, 1);
                    float smp[3];
                    pto_uniform_sample_hemisphere(r1, r2, smp);
                    f3 sw;
                    sw.x = smp[0] * Nb.x + smp[1] * N.x + smp[2] * Nt.x;
                    sw.y = smp[0] * Nb.y + smp[1] * N.y + smp[2] * Nt.y;
                    sw.z = smp[0] * Nb.z + smp[1] * N.z + smp[2] * Nt.z;
                    f3 rd = radiance(&R, 1, f3_add(hp, f3_scale(sw, 0.01f)), sw, 1, (uint32_t)idx, (uint32_t)n, NULL);
                    sum.x = sum.x + r1 * rd.x / pdf;
                    sum.y = sum.y + r1 * rd.y / pdf;
                    sum.z = sum.z + r1 * rd.z / pdf;
                }
            }
            out_accum[3 * idx] = sum.x; out_accum[3 * idx + 1] = sum.y; out_accum[3 * idx + 2] = sum.z;
        }
    }
    return 0;
}

int pto_finalize(const pto_scene* s, const mipt_camera* cam, const mipt_params* p, int accel, const float* accum,
                 float* out_bgr255, int threads)
{
    if (!s || !cam || !p || !accum || !out_bgr255 || cam->div <= 0) return -1;
    long W = cam->client_width / cam->div, H = cam->client_height / cam->div;
    uint32_t N = (uint32_t)(p->num_samples < 1 ? 1 : (p->num_samples > 10000 ? 10000 : p->num_samples));
#ifdef _OPENMP
    int nt = threads > 0 ? threads : omp_get_max_threads();
#else
    int nt = 1; (void)threads;
#endif
    #pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
    for (long h = 0; h < H; ++h) {
        rctx R; R.s = s; R.p = p; R.accel = accel; R.closest = R.shadow = R.tri_tests = R.box_tests = 0;
        for (long j = 0; j < W; ++j) {
            size_t idx = (size_t)(h * W + j);
            f3 rgb = { 0, 0, 0 };
            f3 hp, Nn, Nt, Nb, albedo, direct;
            if (primary_surface(&R, cam, j, h, &hp, &Nn, &Nt, &Nb, &albedo, &direct)) {
                f3 ind = { 0, 0, 0 };
                if (p->global_illumination) {
                    ind.x = accum[3 * idx] / N; ind.y = accum[3 * idx + 1] / N; ind.z = accum[3 * idx + 2] / N;
                }
                rgb.x = (direct.x + p->sun_intensity * ind.x) * albedo.x / PTO_PI;
                rgb.y = (direct.y + p->sun_intensity * ind.y) * albedo.y / PTO_PI;
                rgb.z = (direct.z + p->sun_intensity * ind.z) * albedo.z / PTO_PI;
            }
            rgb.x = (rgb.x < 1.0f) ? rgb.x : 1.0f; rgb.y = (rgb.y < 1.0f) ? rgb.y : 1.0f; rgb.z = (rgb.z < 1.0f) ? rgb.z : 1.0f;
            out_bgr255[3 * idx] = 255 * rgb.x; out_bgr255[3 * idx + 1] = 255 * rgb.y; out_bgr255[3 * idx + 2] = 255 * rgb.z;
        }
    }
    return 0;
}
