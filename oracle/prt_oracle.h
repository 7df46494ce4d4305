/*
 * prt_oracle.h -- CPU restatement of the reference's per-pixel path-tracing loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker.  The product path is prt_amd/ (HIP kernels behind include/prt_hip.h).
 *
 * Each function cites the reference file:line (under /root/reference/src) that it follows.
 * Parity status: pinned against the compiled reference (oracle/_ref, built from the reference's
 * own sources by oracle/Makefile) for RNG, camera packets, BVH build/flatten, the four
 * traversals, triangle and box tests, the bounce loop and the infinite-area light (create + sample);
 * the functions the reference keeps in
 * mesh.cpp / material.cpp / texture.cpp (surface fetch, bump/diffuse/alpha sampling) cannot be
 * compiled here (they include un-vendored third-party headers) and are pinned by restatement
 * only -- see DESIGN.md "Oracle".
 */
#ifndef PRT_ORACLE_H
#define PRT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } orc_v3;
typedef struct { float x, y; } orc_v2;

/* material.h:24-44 (POD subset the path reads) */
typedef struct {
    float diffuse[3];
    float emissive[3];
    uint32_t reflectionType; /* 0 diffuse, 1 specular, 2 refraction (material.h:24-28) */
    uint32_t alphaTest;
    int32_t diffuseMap; /* index into the scene's texture table, -1 = none */
    int32_t bumpMap;
} orc_material;

/* ray.h:182-198; miss <=> t == -1 */
typedef struct {
    float t, i, j, k;
    uint32_t primId, meshId;
} orc_hit;

/* bvh.h:49-60 */
typedef struct {
    float lower[3];
    float upper[3];
    uint32_t primOrSecondNodeIndex;
    uint32_t triVectorIndex; /* :26 */
    uint32_t primCount;      /* :4, 0xf = internal */
    uint32_t splitAxis;      /* :2 */
} orc_node;

/* stats.h:10-16 plus the algorithmic-traffic counters of SURVEY.md 8(d) */
typedef struct {
    uint64_t raysTraced;     /* path_tracer.cpp:62,219,242,276 */
    uint64_t occludedTraced; /* path_tracer.cpp:220,243 */
    uint64_t nBox;           /* node records whose box is tested, per ray */
    uint64_t nTri;           /* triangles tested, per ray */
    uint64_t nHit;           /* surface fetches (mesh.cpp:311) */
    uint64_t nTap;           /* bilinear texture taps */
    uint64_t nPx;            /* pixels written */
    uint64_t rngDraws;
} orc_stats;

typedef struct orc_mesh orc_mesh;
typedef struct orc_bvh orc_bvh;
typedef struct orc_scene orc_scene;

typedef struct {
    float pos[3], dir[3], up[3], right[3];
    uint32_t width, height;
    float invWidth, invHeight;
} orc_camera;

/* ---- mesh (mesh.cpp:90-149, 302-309) ---- */
orc_mesh* orc_mesh_create(uint32_t primCount, uint32_t vertexCount, uint32_t materialCount,
                          const uint32_t* indices, const float* positions, const float* normals /*nullable*/,
                          const float* texcoords /*nullable => hasTexcoord=false*/, const uint32_t* primMaterial,
                          const orc_material* materials);
void orc_mesh_destroy(orc_mesh*);
void orc_mesh_calculate_vertex_normals(orc_mesh*);
void orc_mesh_calculate_bounds(orc_mesh*);
const float* orc_mesh_normals(const orc_mesh*);
const float* orc_mesh_bbox(const orc_mesh*); /* lower[3], upper[3] */

/* ---- BVH (bvh.cpp:21-299) ---- */
orc_bvh* orc_bvh_build(orc_mesh* mesh); /* takes ownership of mesh */
void orc_bvh_destroy(orc_bvh*);
uint32_t orc_bvh_node_count(const orc_bvh*);
uint32_t orc_bvh_leaf_count(const orc_bvh*);
const orc_node* orc_bvh_nodes(const orc_bvh*);
const uint32_t* orc_bvh_prim_remap(const orc_bvh*);
uint32_t orc_bvh_prim_count(const orc_bvh*);

/* ---- scene (scene.cpp) ---- */
orc_scene* orc_scene_create(void);
void orc_scene_destroy(orc_scene*); /* destroys added bvhs */
void orc_scene_add(orc_scene*, orc_bvh*);
void orc_scene_set_directional_light(orc_scene*, const float dir[3], const float intensity[3]);
int32_t orc_scene_add_texture(orc_scene*, int32_t width, int32_t height, int32_t component, const uint8_t* texels);
/* InfiniteAreaLight::create (light.cpp:30-84) from float RGBA texels; sample (light.cpp:86-128) for n (u.x, u.y) pairs */
void orc_scene_set_env_light(orc_scene*, int32_t width, int32_t height, const float* rgba);
const float* orc_scene_env_vertical(const orc_scene*);
const float* orc_scene_env_horizontal(const orc_scene*);
void orc_x_env_sample(const orc_scene*, uint32_t n, const float* u, float* dirOut, float* colorOut);
void orc_x_tex_sample3f(int32_t w, int32_t h, int32_t comp, const float* texels, float u, float v, float out[3]);
float orc_scene_radius(const orc_scene*);
const float* orc_scene_bbox(const orc_scene*);

/* ---- camera (camera.h:17-36) ---- */
void orc_camera_create(orc_camera* cam, const float pos[3], const float dir[3], uint32_t width, uint32_t height);

/* ---- RNG (random.h) ---- */
uint32_t orc_pixel_seed(uint32_t x, uint32_t y, uint32_t width, uint32_t seed); /* build contract, SURVEY 8(a3) */
uint32_t orc_rng_next(uint32_t* state);
float orc_rng_float(uint32_t* state);

/* ---- leaf math ---- */
/* triangle.cpp:90-166, one lane; returns t (-1 on miss) and barycentrics */
float orc_intersect_triangle(const float org[3], const float dir[3], int swapXZ, int swapYZ, const float p0[3],
                             const float p1[3], const float p2[3], float ijk[3]);
/* triangle.cpp:8-88 (scalar version; used only by the reference's test) */
float orc_intersect_triangle_scalar(const float org[3], const float dir[3], const float p0[3], const float p1[3],
                                    const float p2[3], float ijk[3]);
float orc_bbox_intersect_t(const float lower[3], const float upper[3], const float org[3], const float invDir[3]);      /* vecmath.h:1402 */
int orc_bbox_intersect_bool(const float lower[3], const float upper[3], const float org[3], const float invDir[3], float maxT); /* :1449 */
int orc_bbox_intersect_soa(const float lower[3], const float upper[3], const float org[3], const float invDir[3], float maxT);  /* :1504 */
void orc_ray_prepare_single(const float dir[3], float invDir[3], int* swapXZ, int* swapYZ); /* ray.h:26-40 */
void orc_ray_prepare_soa(const float dir[3], float invDir[3], int* swapXZ, int* swapYZ);    /* ray.h:58-71 */

/* ---- traversal entry points (scene.cpp:47-94 over bvh.cpp:429-654) ---- */
void orc_intersect_single(const orc_scene*, const float org[3], const float dir[3], float maxT, orc_hit* hit, orc_stats* st);
void orc_intersect_packet(const orc_scene*, const float org[8][3], const float dir[8][3], const float avgDir[3],
                          float maxT, orc_hit hits[8], orc_stats* st);
int orc_occluded_single(const orc_scene*, const float org[3], const float dir[3], float maxT, orc_stats* st);
/* returns the occluded mask (inactive lanes reported occluded), bits 0..7 */
uint32_t orc_occluded_packet(const orc_scene*, uint32_t activeMask, const float org[8][3], const float dir[8][3],
                             float maxT, orc_stats* st);

/* camera.cpp:35-73: fills org/dir for the 8 lanes and avgDir, advancing *rng by 16 draws */
void orc_camera_packet(const orc_camera*, uint32_t* rng, uint32_t x, uint32_t y, float org[8][3], float dir[8][3],
                       float avgDir[3]);

/* path_tracer.cpp:17-33,57-75,77-308.  rgb is width*height*3 floats; pixel (x,y) -> rgb[(x+y*width)*3].
 * Pixels x0..x1, y0..y1 inclusive.  maxDepth = 14 and rrDepth = 4 reproduce the reference. */
void orc_trace_block(const orc_scene*, const orc_camera*, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                     uint32_t samples, uint32_t maxDepth, uint32_t seed, float exposure, float* rgb, orc_stats* st);
/* whole image over `threads` OpenMP threads (0 = all); same results as orc_trace_block */
void orc_render(const orc_scene*, const orc_camera*, uint32_t samples, uint32_t maxDepth, uint32_t seed,
                float exposure, int threads, float* rgb, orc_stats* st);
/* the same over an inclusive pixel rectangle (used for the bounded CPU-baseline sample of bench.py) */
void orc_render_rect(const orc_scene*, const orc_camera*, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t samples,
                     uint32_t maxDepth, uint32_t seed, float exposure, int threads, float* rgb, orc_stats* st);
/* GbufferVisualizer::TraceBlock (gbuffer_visualizer.cpp:17-51); type 0 = kDiffuse, 1 = kMeshNormal, 2 = kNormal */
void orc_gbuffer_block(const orc_scene*, const orc_camera*, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t type,
                       uint32_t seed, float exposure, float* rgb);
int orc_max_threads(void);
void orc_set_rr_depth(uint32_t rrDepth); /* default 4 = the reference (path_tracer.cpp:258) */

/* ---- explicit-argument leaf functions: mesh.cpp:311-364, texture.cpp:142-156, material.cpp:87-114 ---- */
void orc_x_get_surface(const orc_mesh* m, uint32_t primId, float i, float j, float k, float normal[3], uint32_t* matIndex,
                       float uv[2], float duv01[2], float duv02[2], float dp01[3], float dp02[3]);
int orc_x_tex_test_alpha(int32_t w, int32_t h, int32_t comp, const uint8_t* texels, float u, float v, int soa);
void orc_x_sample_diffuse(const float diffuse[3], int32_t w, int32_t h, int32_t comp, const uint8_t* texels /*nullable*/,
                          float u, float v, float out[3]);
void orc_x_sample_bump(const float normal[3], int32_t w, int32_t h, int32_t comp, const uint8_t* texels /*nullable*/,
                       const float uv[2], const float duv01[2], const float duv02[2], const float dp01[3],
                       const float dp02[3], float out[3]);

/* libm exactly as the reference calls it (checker for the kernels' own sincos / powf) */
void orc_libm_sincos(uint32_t n, const float* theta, float* s, float* c);
void orc_libm_powf22(uint32_t n, const float* x, float* y);

/* counters of occlusion queries: 0 = what the reference's traversal does (default), 1 = what prt_amd's near-first
 * per-ray visit does (see prt_oracle.c, "accounting of the any-hit queries").  Never changes a result. */
void orc_set_anyhit_accounting(int mode);

#ifdef __cplusplus
}
#endif
#endif
