/*
 * prt_oracle.c -- CPU restatement of the reference's per-pixel path-tracing loop (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see prt_oracle.h).  Build: oracle/Makefile (gcc -O2, no FMA
 * contraction, no fast-math -- the reference's object code contains no FMA either).
 *
 * The reference computes in 8-wide AVX vectors; every SIMD operation it uses is lane-wise, so
 * the restatement loops over lanes with scalar IEEE binary32 arithmetic.  Where the reference
 * uses SSE/AVX min/max (which return the SECOND operand when either is NaN) the helpers
 * sse_min/sse_max below keep that rule; where it uses ordered SIMD compares, C's < > == are the
 * same except '!=' (see neq_oq).
 */
#include "prt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LANES 8          /* SoaConstants::kLaneCount with R_AVX8L, vecmath.h:50-54 */
#define STACK_SIZE 64    /* bvh.cpp:432, 579 */
#define INTERNAL_NODE 0xf /* bvh.h:51 */
static const float kTriEpsilon = 0.0001f; /* triangle.cpp:6, bvh.h:109 */
static const float kPi = 3.14159265358979323846f; /* vecmath.h:162 */

/* ------------------------------------------------------------------ vector helpers */
typedef orc_v3 v3;
typedef orc_v2 v2;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3s(float f) { return V3(f, f, f); }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }         /* vecmath.h:255 */
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }         /* :251 */
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }         /* :259 */
static inline v3 div3(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }         /* :263 */
static inline v3 scale3(float f, v3 v) { return V3(f * v.x, f * v.y, f * v.z); }          /* :267,271 */
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { v3 v = mul3(a, b); return v.x + v.y + v.z; }       /* :1181 */
static inline v3 cross3(v3 a, v3 b)                                                       /* :1189 */
{
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float length3(v3 v) { return sqrtf(dot3(v, v)); }                           /* :1195 */
static inline v3 normalize3(v3 v) { float invlen = 1.0f / length3(v); return scale3(invlen, v); } /* :1200 */
static inline v3 ld3(const float* p) { return V3(p[0], p[1], p[2]); }
static inline void st3(float* p, v3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

static inline v2 V2(float x, float y) { v2 r = {x, y}; return r; }
static inline v2 add2(v2 a, v2 b) { return V2(a.x + b.x, a.y + b.y); }
static inline v2 sub2(v2 a, v2 b) { return V2(a.x - b.x, a.y - b.y); }
static inline v2 scale2(float f, v2 v) { return V2(f * v.x, f * v.y); }
static inline v2 safe_normalize2(v2 v)                                                    /* :1145 */
{
    v2 m = V2(v.x * v.x, v.y * v.y);
    float len = sqrtf(m.x + m.y);
    if (len < 0.00001f) return V2(0.0f, 0.0f);
    float invlen = 1.0f / len;
    return scale2(invlen, v);
}

/* _mm_min_ps / _mm_max_ps semantics: second operand when unordered or equal */
static inline float sse_min(float a, float b) { return a < b ? a : b; }
static inline float sse_max(float a, float b) { return a > b ? a : b; }
/* _CMP_NEQ_OQ: false when unordered */
static inline int neq_oq(float a, float b) { return a < b || a > b; }
/* std::max(a,b) = (a < b) ? b : a */
static inline float std_max(float a, float b) { return (a < b) ? b : a; }
/* cvttss2si: out-of-range and NaN give INT_MIN */
static inline int32_t cvtt(float f)
{
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT32_MIN;
    return (int32_t)f;
}

/* ------------------------------------------------------------------ data */
typedef struct {
    int32_t width, height, component;
    uint8_t* texels;
} texture_t;

struct orc_mesh {
    uint32_t primCount, vertexCount, materialCount;
    uint32_t* indices;
    v3* positions;
    v3* normals; /* NULL <=> !hasVertexNormal */
    v2* texcoords;
    int hasTexcoord;
    uint32_t* primMaterial;
    orc_material* materials;
    v3 lower, upper;
    uint32_t id;
};

/* bvh.h:71-86 -- one leaf's triangles, 8-wide SoA, zero padded */
typedef struct {
    float p[3][3][LANES]; /* [vertex][xyz][lane] */
    float uv[3][2][LANES];
    int32_t alphaTest[LANES];
} trivec_t;

struct orc_bvh {
    orc_mesh* mesh;
    uint32_t* primRemapping;
    orc_node* nodes;
    trivec_t* triVectors;
    uint32_t nodeCount, leafCount;
};

struct orc_scene {
    orc_bvh** bvh;
    uint32_t bvhCount;
    int hasDirectional;
    v3 lightDir, lightIntensity;
    v3 lower, upper;
    float radius;
    texture_t* textures;
    int32_t textureCount;
    /* InfiniteAreaLight (light.h:28-50): float RGBA image + the two CDF tables */
    int hasEnv;
    int32_t envWidth, envHeight;
    float* envTexels;      /* 4 floats per texel (Texture::loadExr: component = 4, texture.cpp:304) */
    float* envVerticalP;   /* [height] */
    float* envHorizontalP; /* [width*height] */
};

static void bbox_init(v3* lo, v3* hi) { *lo = v3s(FLT_MAX); *hi = v3s(-FLT_MAX); } /* vecmath.cpp:46 */
static void bbox_merge_p(v3* lo, v3* hi, v3 p)                                      /* vecmath.cpp:69, vecmath.h:1156 */
{
    lo->x = fminf(lo->x, p.x); lo->y = fminf(lo->y, p.y); lo->z = fminf(lo->z, p.z);
    hi->x = fmaxf(hi->x, p.x); hi->y = fmaxf(hi->y, p.y); hi->z = fmaxf(hi->z, p.z);
}
static void bbox_merge_b(v3* lo, v3* hi, v3 blo, v3 bhi)                            /* vecmath.cpp:63 */
{
    lo->x = fminf(lo->x, blo.x); lo->y = fminf(lo->y, blo.y); lo->z = fminf(lo->z, blo.z);
    hi->x = fmaxf(hi->x, bhi.x); hi->y = fmaxf(hi->y, bhi.y); hi->z = fmaxf(hi->z, bhi.z);
}
static float bbox_area(v3 lo, v3 hi)                                                /* vecmath.cpp:75 */
{
    v3 e = sub3(hi, lo);
    return 2.0f * (e.x * e.y + e.y * e.z + e.z * e.x);
}

/* ------------------------------------------------------------------ mesh */
orc_mesh* orc_mesh_create(uint32_t primCount, uint32_t vertexCount, uint32_t materialCount, const uint32_t* indices,
                          const float* positions, const float* normals, const float* texcoords,
                          const uint32_t* primMaterial, const orc_material* materials)
{
    orc_mesh* m = (orc_mesh*)calloc(1, sizeof(*m));
    m->primCount = primCount;
    m->vertexCount = vertexCount;
    m->materialCount = materialCount;
    m->indices = (uint32_t*)malloc(sizeof(uint32_t) * 3 * (size_t)primCount + 4);
    memcpy(m->indices, indices, sizeof(uint32_t) * 3 * (size_t)primCount);
    m->positions = (v3*)malloc(sizeof(v3) * (size_t)vertexCount + 4);
    memcpy(m->positions, positions, sizeof(v3) * (size_t)vertexCount);
    if (normals) {
        m->normals = (v3*)malloc(sizeof(v3) * (size_t)vertexCount + 4);
        memcpy(m->normals, normals, sizeof(v3) * (size_t)vertexCount);
    }
    m->texcoords = (v2*)calloc((size_t)vertexCount + 1, sizeof(v2));
    if (texcoords) {
        memcpy(m->texcoords, texcoords, sizeof(v2) * (size_t)vertexCount);
        m->hasTexcoord = 1;
    }
    m->primMaterial = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)primCount + 4);
    memcpy(m->primMaterial, primMaterial, sizeof(uint32_t) * (size_t)primCount);
    m->materials = (orc_material*)malloc(sizeof(orc_material) * (size_t)materialCount + 4);
    memcpy(m->materials, materials, sizeof(orc_material) * (size_t)materialCount);
    orc_mesh_calculate_bounds(m);
    return m;
}

void orc_mesh_destroy(orc_mesh* m)
{
    if (!m) return;
    free(m->indices); free(m->positions); free(m->normals); free(m->texcoords);
    free(m->primMaterial); free(m->materials); free(m);
}

/* mesh.cpp:302-309 */
void orc_mesh_calculate_bounds(orc_mesh* m)
{
    v3 lo, hi;
    bbox_init(&lo, &hi);
    for (uint32_t i = 0; i < m->vertexCount; i++) bbox_merge_p(&lo, &hi, m->positions[i]);
    m->lower = lo;
    m->upper = hi;
}

/* mesh.cpp:108-149: reverse prim order, accumulated normal never allowed to become zero */
void orc_mesh_calculate_vertex_normals(orc_mesh* m)
{
    free(m->normals);
    m->normals = (v3*)malloc(sizeof(v3) * (size_t)m->vertexCount + 4);
    for (uint32_t i = 0; i < m->vertexCount; i++) m->normals[i] = v3s(0.0f);
    for (int32_t i = (int32_t)m->primCount - 1; i >= 0; i--) {
        uint32_t v[3];
        v3 p[3];
        for (int j = 0; j < 3; j++) {
            v[j] = m->indices[3 * i + j];
            p[j] = m->positions[v[j]];
        }
        v3 normal = normalize3(cross3(sub3(p[1], p[0]), sub3(p[2], p[0])));
        if (isnan(normal.x) || isnan(normal.y) || isnan(normal.z)) normal = v3s(0.0f);
        for (int j = 0; j < 3; j++) {
            v3 n = add3(m->normals[v[j]], normal);
            if (length3(n) > 0.0f) m->normals[v[j]] = n;
        }
    }
    for (uint32_t i = 0; i < m->vertexCount; i++) m->normals[i] = normalize3(m->normals[i]);
}

const float* orc_mesh_normals(const orc_mesh* m) { return (const float*)m->normals; }
const float* orc_mesh_bbox(const orc_mesh* m) { return &m->lower.x; }

/* ------------------------------------------------------------------ BVH build, bvh.cpp:21-299 */
typedef struct {
    orc_bvh* bvh;
    uint32_t nodeCap, leafCap;
} build_ctx;

static float comp(v3 v, int d) { return d == 0 ? v.x : (d == 1 ? v.y : v.z); }

static v3 prim_centroid(const orc_mesh* m, uint32_t prim) /* bvh.cpp:77-85, 127-132 */
{
    v3 temp = v3s(0.0f);
    for (int j = 0; j < 3; j++) temp = add3(temp, m->positions[m->indices[3 * prim + j]]);
    return scale3(1.0f / 3.0f, temp);
}

static uint32_t emit_node(build_ctx* c)
{
    orc_bvh* b = c->bvh;
    if (b->nodeCount == c->nodeCap) {
        c->nodeCap = c->nodeCap ? c->nodeCap * 2 : 1024;
        b->nodes = (orc_node*)realloc(b->nodes, sizeof(orc_node) * c->nodeCap);
    }
    memset(&b->nodes[b->nodeCount], 0, sizeof(orc_node));
    return b->nodeCount++;
}

/* bvh.cpp:245-296 */
static void emit_leaf(build_ctx* c, orc_node* n, int32_t start, int32_t primCount)
{
    orc_bvh* b = c->bvh;
    const orc_mesh* m = b->mesh;
    if (b->leafCount == c->leafCap) {
        c->leafCap = c->leafCap ? c->leafCap * 2 : 512;
        b->triVectors = (trivec_t*)realloc(b->triVectors, sizeof(trivec_t) * c->leafCap);
    }
    trivec_t* tv = &b->triVectors[b->leafCount];
    memset(tv, 0, sizeof(*tv));
    for (int32_t i = 0; i < primCount; i++) {
        uint32_t prim = b->primRemapping[start + i];
        tv->alphaTest[i] = (int32_t)m->materials[m->primMaterial[prim]].alphaTest;
        for (int j = 0; j < 3; j++) {
            uint32_t v = m->indices[3 * prim + j];
            v3 p = m->positions[v];
            v2 t = m->texcoords[v];
            tv->p[j][0][i] = p.x; tv->p[j][1][i] = p.y; tv->p[j][2][i] = p.z;
            tv->uv[j][0][i] = t.x; tv->uv[j][1][i] = t.y;
        }
    }
    n->primOrSecondNodeIndex = (uint32_t)start;
    n->primCount = (uint32_t)primCount;
    n->triVectorIndex = b->leafCount;
    b->leafCount++;
}

/* bvh.cpp:31-171 with the DFS linearisation of bvh.cpp:230-243 folded in: the build recursion
 * visits nodes in the same pre-order that buildLinearBvhNodes numbers them. */
static void build_node(build_ctx* c, int32_t start, int32_t end)
{
    orc_bvh* b = c->bvh;
    const orc_mesh* m = b->mesh;
    uint32_t* remap = b->primRemapping;
    uint32_t self = emit_node(c);

    v3 lo, hi;
    bbox_init(&lo, &hi);
    for (int32_t i = start; i <= end; i++)
        for (int j = 0; j < 3; j++) bbox_merge_p(&lo, &hi, m->positions[m->indices[3 * remap[i] + j]]);
    st3(b->nodes[self].lower, lo);
    st3(b->nodes[self].upper, hi);

    int32_t primCount = end - start + 1;
    if (primCount <= LANES) {
        orc_node n = b->nodes[self];
        emit_leaf(c, &n, start, primCount);
        b->nodes[self] = n;
        return;
    }

    v3 extent = sub3(hi, lo);
    const uint32_t kBucketCount = 32;
    float lowestCost = FLT_MAX;
    uint32_t lowestDim = 0;
    int32_t lowestCostSplit = -1;

    for (int dim = 0; dim < 3; dim++) {
        const float splitExtent = comp(extent, dim) == 0.0f ? 0.0001f : comp(extent, dim);
        const float lowerPos = comp(lo, dim);
        uint32_t count[32];
        v3 blo[32], bhi[32];
        for (uint32_t k = 0; k < kBucketCount; k++) { count[k] = 0; bbox_init(&blo[k], &bhi[k]); }

        for (int32_t i = start; i <= end; i++) {
            v3 temp = v3s(0.0f), plo, phi;
            bbox_init(&plo, &phi);
            for (int j = 0; j < 3; j++) {
                v3 v = m->positions[m->indices[3 * remap[i] + j]];
                temp = add3(temp, v);
                bbox_merge_p(&plo, &phi, v);
            }
            temp = scale3(1.0f / 3.0f, temp);
            int32_t bk = cvtt((float)kBucketCount * (comp(temp, dim) - lowerPos) / splitExtent);
            /* bvh.cpp:88: int32 compared with a uint32 constant => negative values land in the last bucket */
            if ((uint32_t)bk >= kBucketCount) bk = (int32_t)kBucketCount - 1;
            count[bk]++;
            bbox_merge_b(&blo[bk], &bhi[bk], plo, phi);
        }

        for (uint32_t i = 0; i < kBucketCount - 1; i++) {
            uint32_t countLeft = 0, countRight = 0;
            v3 llo, lhi, rlo, rhi;
            bbox_init(&llo, &lhi);
            bbox_init(&rlo, &rhi);
            for (uint32_t j = 0; j <= i; j++) { countLeft += count[j]; bbox_merge_b(&llo, &lhi, blo[j], bhi[j]); }
            for (uint32_t j = i + 1; j < kBucketCount; j++) { countRight += count[j]; bbox_merge_b(&rlo, &rhi, blo[j], bhi[j]); }
            float cost = 0.125f + ((float)countLeft * bbox_area(llo, lhi) + (float)countRight * bbox_area(rlo, rhi));
            if (lowestCost > cost) {
                lowestDim = (uint32_t)dim;
                lowestCost = cost;
                lowestCostSplit = (int32_t)i;
            }
        }
    }

    const int dim = (int)lowestDim;
    const float splitExtent = comp(extent, dim) == 0.0f ? 0.0001f : comp(extent, dim);
    const float lowerPos = comp(lo, dim);
    float splitPos = lowerPos + (float)(lowestCostSplit + 1) * splitExtent / (float)kBucketCount;

#define SPLIT_LEFT(i) (comp(prim_centroid(m, remap[i]), dim) < splitPos)
    int32_t cursor;
    for (cursor = start; cursor <= end; cursor++)
        if (!SPLIT_LEFT(cursor)) break;
    for (int32_t i = cursor + 1; i <= end; i++) {
        if (SPLIT_LEFT(i)) {
            uint32_t t = remap[i]; remap[i] = remap[cursor]; remap[cursor] = t;
            cursor++;
        }
    }
#undef SPLIT_LEFT
    int32_t mid = cursor - 1;
    if (mid <= start || end <= mid) mid = (start + end) / 2;

    b->nodes[self].splitAxis = (uint32_t)dim;
    b->nodes[self].primCount = INTERNAL_NODE;
    build_node(c, start, mid);
    b->nodes[self].primOrSecondNodeIndex = b->nodeCount;
    build_node(c, mid + 1, end);
}

orc_bvh* orc_bvh_build(orc_mesh* mesh)
{
    orc_bvh* b = (orc_bvh*)calloc(1, sizeof(*b));
    b->mesh = mesh;
    b->primRemapping = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)mesh->primCount + 4);
    for (uint32_t i = 0; i < mesh->primCount; i++) b->primRemapping[i] = i;
    build_ctx c = {b, 0, 0};
    build_node(&c, 0, (int32_t)mesh->primCount - 1);
    return b;
}

void orc_bvh_destroy(orc_bvh* b)
{
    if (!b) return;
    orc_mesh_destroy(b->mesh);
    free(b->primRemapping); free(b->nodes); free(b->triVectors); free(b);
}
uint32_t orc_bvh_node_count(const orc_bvh* b) { return b->nodeCount; }
uint32_t orc_bvh_leaf_count(const orc_bvh* b) { return b->leafCount; }
const orc_node* orc_bvh_nodes(const orc_bvh* b) { return b->nodes; }
const uint32_t* orc_bvh_prim_remap(const orc_bvh* b) { return b->primRemapping; }
uint32_t orc_bvh_prim_count(const orc_bvh* b) { return b->mesh->primCount; }

/* ------------------------------------------------------------------ scene, scene.cpp:9-27 */
orc_scene* orc_scene_create(void)
{
    orc_scene* s = (orc_scene*)calloc(1, sizeof(*s));
    bbox_init(&s->lower, &s->upper);
    s->radius = FLT_MAX;
    return s;
}

void orc_scene_destroy(orc_scene* s)
{
    if (!s) return;
    for (uint32_t i = 0; i < s->bvhCount; i++) orc_bvh_destroy(s->bvh[i]);
    for (int32_t i = 0; i < s->textureCount; i++) free(s->textures[i].texels);
    free(s->envTexels); free(s->envVerticalP); free(s->envHorizontalP);
    free(s->textures); free(s->bvh); free(s);
}

void orc_scene_add(orc_scene* s, orc_bvh* b)
{
    b->mesh->id = s->bvhCount;
    s->bvh = (orc_bvh**)realloc(s->bvh, sizeof(orc_bvh*) * (s->bvhCount + 1));
    s->bvh[s->bvhCount++] = b;
    bbox_merge_b(&s->lower, &s->upper, b->mesh->lower, b->mesh->upper);
    v3 center = scale3(0.5f, add3(s->upper, s->lower)); /* vecmath.h:1109 */
    s->radius = length3(sub3(s->upper, center));
}

void orc_scene_set_directional_light(orc_scene* s, const float dir[3], const float intensity[3])
{
    s->hasDirectional = 1;
    s->lightDir = ld3(dir);
    s->lightIntensity = ld3(intensity);
}

/* InfiniteAreaLight::create, light.cpp:30-84, from float RGBA texels instead of an .exr file (scene.h:42-45 sets the
 * kInfiniteArea bit, which ComputeRadiance tests BEFORE the directional light, path_tracer.cpp:164-173) */
void orc_scene_set_env_light(orc_scene* s, int32_t width, int32_t height, const float* rgba)
{
    free(s->envTexels); free(s->envVerticalP); free(s->envHorizontalP);
    size_t n = (size_t)width * (size_t)height;
    s->envTexels = (float*)malloc(n * 4 * sizeof(float));
    memcpy(s->envTexels, rgba, n * 4 * sizeof(float));
    s->envWidth = width; s->envHeight = height;
    s->envVerticalP = (float*)malloc((size_t)height * sizeof(float));
    s->envHorizontalP = (float*)malloc(n * sizeof(float));
    float* vert = s->envVerticalP;
    float* hori = s->envHorizontalP;
    const float* p = s->envTexels;
    float vsum = 0.0f;
    for (uint32_t y = 0; y < (uint32_t)height; y++) {
        float hsum = 0.0f;
        for (uint32_t x = 0; x < (uint32_t)width; x++) {
            uint32_t indexBase = x + y * (uint32_t)width;
            float l = length3(V3(p[4 * indexBase + 0], p[4 * indexBase + 1], p[4 * indexBase + 2]));
            hori[indexBase] = l;
            hsum += l;
        }
        float sinPhi = sinf(kPi * ((float)y + 0.5f) / (float)height); /* :58 */
        vert[y] = hsum * sinPhi;
        vsum += hsum * sinPhi;
        float invH = 1.0f / hsum;
        float accumH = 0.0f;
        for (uint32_t x = 0; x < (uint32_t)width; x++) {
            uint32_t indexBase = x + y * (uint32_t)width;
            float ph = accumH + invH * hori[indexBase];
            hori[indexBase] = ph;
            accumH = ph;
        }
    }
    float invV = 1.0f / vsum;
    float accumV = 0.0f;
    for (uint32_t y = 0; y < (uint32_t)height; y++) {
        float pv = accumV + invV * vert[y];
        vert[y] = pv;
        accumV = pv;
    }
    s->hasEnv = 1;
}

const float* orc_scene_env_vertical(const orc_scene* s) { return s->envVerticalP; }
const float* orc_scene_env_horizontal(const orc_scene* s) { return s->envHorizontalP; }

int32_t orc_scene_add_texture(orc_scene* s, int32_t width, int32_t height, int32_t component, const uint8_t* texels)
{
    s->textures = (texture_t*)realloc(s->textures, sizeof(texture_t) * (size_t)(s->textureCount + 1));
    texture_t* t = &s->textures[s->textureCount];
    t->width = width; t->height = height; t->component = component;
    size_t sz = (size_t)width * (size_t)height * (size_t)component;
    t->texels = (uint8_t*)malloc(sz + 16);
    memcpy(t->texels, texels, sz);
    memset(t->texels + sz, 0, 16);
    return s->textureCount++;
}

float orc_scene_radius(const orc_scene* s) { return s->radius; }
const float* orc_scene_bbox(const orc_scene* s) { return &s->lower.x; }

/* ------------------------------------------------------------------ camera, camera.h:17-36 */
void orc_camera_create(orc_camera* cam, const float pos[3], const float dirIn[3], uint32_t width, uint32_t height)
{
    v3 dir = ld3(dirIn);
    st3(cam->pos, ld3(pos));
    st3(cam->dir, normalize3(dir));
    cam->width = width;
    cam->height = height;
    cam->invWidth = 1.0f / (float)width;
    cam->invHeight = 1.0f / (float)height;
    v3 up = V3(0, 1.0f, 0);
    v3 right = cross3(dir, up);
    if (length3(right) < 0.00001f) right = cross3(dir, V3(1, 0, 0));
    right = normalize3(right);
    up = normalize3(cross3(right, dir));
    st3(cam->up, up);
    st3(cam->right, right);
}

/* ------------------------------------------------------------------ RNG, random.h:23-48 */
uint32_t orc_pixel_seed(uint32_t x, uint32_t y, uint32_t width, uint32_t seed)
{
    /* Build contract: lowbias32(x + y*W + seed) | 1.  The reference seeds each
     * tile's generator from libc rand() (random.h:15-17); a fixed per-pixel state replaces it. */
    uint32_t h = x + y * width + seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h | 1u;
}

uint32_t orc_rng_next(uint32_t* state)
{
    uint32_t x = *state;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    return *state = x;
}

float orc_rng_float(uint32_t* state)
{
    uint32_t u = orc_rng_next(state);
    union { uint32_t u; float f; } ui;
    ui.u = (u & 0x007fffffu) | 0x3f800000u;
    return ui.f - 1.0f;
}

static inline float rng_pm1(uint32_t* s, orc_stats* st) { if (st) st->rngDraws++; return 2.0f * orc_rng_float(s) - 1.0f; }
static inline float rng_f(uint32_t* s, orc_stats* st) { if (st) st->rngDraws++; return orc_rng_float(s); }

/* ------------------------------------------------------------------ ray preparation */
/* ray.h:26-40 + vecmath.h:216-233: the swap axis is chosen on SIGNED components */
void orc_ray_prepare_single(const float d[3], float invDir[3], int* swapXZ, int* swapYZ)
{
    invDir[0] = 1.0f / d[0]; invDir[1] = 1.0f / d[1]; invDir[2] = 1.0f / d[2];
    uint32_t e;
    if (d[0] > d[1]) e = (d[0] > d[2]) ? 0 : 2;
    else e = (d[1] > d[2]) ? 1 : 2;
    *swapXZ = (e == 0);
    *swapYZ = (e == 1);
}

/* ray.h:58-71: largest |component|; x wins ties, then y */
void orc_ray_prepare_soa(const float d[3], float invDir[3], int* swapXZ, int* swapYZ)
{
    invDir[0] = 1.0f / d[0]; invDir[1] = 1.0f / d[1]; invDir[2] = 1.0f / d[2];
    float ax = fabsf(d[0]), ay = fabsf(d[1]), az = fabsf(d[2]);
    float max_e = sse_max(ax, sse_max(ay, az)); /* vecmath.h:1032-1038 */
    int mask_x = (max_e == ax);
    int mask_y = (max_e == ay) && !mask_x;
    *swapXZ = mask_x;
    *swapYZ = mask_y;
}

/* ------------------------------------------------------------------ box tests */
/* vecmath.h:1402-1424: xyz + a 4th lane (-inf/+inf); reduce as op(op(a0,a2),op(a1,a3)) (:1325-1345) */
float orc_bbox_intersect_t(const float lower[3], const float upper[3], const float org[3], const float invDir[3])
{
    float t0[4], t1[4];
    for (int k = 0; k < 3; k++) {
        float a = (lower[k] - org[k]) * invDir[k];
        float b = (upper[k] - org[k]) * invDir[k];
        t0[k] = sse_min(a, b);
        t1[k] = sse_max(a, b);
    }
    {
        float a = (-1.0f - 0.0f) * INFINITY;
        float b = (1.0f - 0.0f) * INFINITY;
        t0[3] = sse_min(a, b);
        t1[3] = sse_max(a, b);
    }
    float max_t0 = sse_max(sse_max(t0[0], t0[2]), sse_max(t0[1], t0[3]));
    float min_t1 = sse_min(sse_min(t1[0], t1[2]), sse_min(t1[1], t1[3]));
    if (min_t1 < max_t0) return INFINITY;
    return max_t0;
}

/* vecmath.h:1449-1466: 4th lane carries (-FLT_MAX, maxT) */
int orc_bbox_intersect_bool(const float lower[3], const float upper[3], const float org[3], const float invDir[3], float maxT)
{
    float t0[4], t1[4];
    for (int k = 0; k < 3; k++) {
        float a = (lower[k] - org[k]) * invDir[k];
        float b = (upper[k] - org[k]) * invDir[k];
        t0[k] = sse_min(a, b);
        t1[k] = sse_max(a, b);
    }
    {
        float a = (-FLT_MAX - 0.0f) * 1.0f;
        float b = (maxT - 0.0f) * 1.0f;
        t0[3] = sse_min(a, b);
        t1[3] = sse_max(a, b);
    }
    float max_t0 = sse_max(sse_max(t0[0], t0[2]), sse_max(t0[1], t0[3]));
    float min_t1 = sse_min(sse_min(t1[0], t1[2]), sse_min(t1[1], t1[3]));
    return min_t1 > max_t0;
}

/* vecmath.h:1504-1518, one lane */
int orc_bbox_intersect_soa(const float lower[3], const float upper[3], const float org[3], const float invDir[3], float maxT)
{
    float t0[3], t1[3];
    for (int k = 0; k < 3; k++) {
        float a = (lower[k] - org[k]) * invDir[k];
        float b = (upper[k] - org[k]) * invDir[k];
        t0[k] = sse_min(a, b);
        t1[k] = sse_max(a, b);
    }
    float max_t0 = sse_max(t0[0], sse_max(t0[1], t0[2]));
    float min_t1 = sse_min(t1[0], sse_min(t1[1], t1[2]));
    int mask = min_t1 >= max_t0;
    return (max_t0 < maxT) && mask;
}

/* ------------------------------------------------------------------ triangle */
/* triangle.cpp:90-166, one lane.  The early return at :136-140 only skips work for lanes that
 * already failed the edge test, so the per-lane result is unchanged. */
float orc_intersect_triangle(const float org[3], const float dir[3], int swapXZ, int swapYZ, const float p0[3],
                             const float p1[3], const float p2[3], float ijk[3])
{
    float d[3] = {dir[0], dir[1], dir[2]};
    float v0[3], v1[3], v2[3];
    for (int k = 0; k < 3; k++) { v0[k] = p0[k] - org[k]; v1[k] = p1[k] - org[k]; v2[k] = p2[k] - org[k]; }
    float t;
#define SWAP(a, i, j) { t = a[i]; a[i] = a[j]; a[j] = t; }
    if (swapXZ) { SWAP(d, 0, 2); SWAP(v0, 0, 2); SWAP(v1, 0, 2); SWAP(v2, 0, 2); }
    if (swapYZ) { SWAP(d, 1, 2); SWAP(v0, 1, 2); SWAP(v1, 1, 2); SWAP(v2, 1, 2); }
#undef SWAP
    float v0z = v0[2], v1z = v1[2], v2z = v2[2];
    float inv_dz = 1.0f / d[2];
    float idx = d[0] * inv_dz, idy = d[1] * inv_dz; /* invDzD = inv_dz*d, :119 */
    float v0x = v0[0] - idx * v0z, v0y = v0[1] - idy * v0z;
    float v1x = v1[0] - idx * v1z, v1y = v1[1] - idy * v1z;
    float v2x = v2[0] - idx * v2z, v2y = v2[1] - idy * v2z;

    float e0 = v1x * v2y - v1y * v2x;
    float e1 = v2x * v0y - v2y * v0x;
    float e2 = v0x * v1y - v0y * v1x;

    ijk[0] = ijk[1] = ijk[2] = 0.0f;
    int mask_eb = (e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f);
    if (mask_eb) return -1.0f;
    float det = e0 + e1 + e2;
    int mask = neq_oq(det, 0.0f);
    float inv_det = 1.0f / det;
    float t_scaled = (e0 * v0z + e1 * v1z + e2 * v2z) * inv_dz;
    float tt = t_scaled * inv_det;
    mask = mask && (tt > kTriEpsilon);
    ijk[0] = e0 * inv_det;
    ijk[1] = e1 * inv_det;
    ijk[2] = e2 * inv_det;
    return mask ? tt : -1.0f;
}

/* triangle.cpp:8-88 */
float orc_intersect_triangle_scalar(const float org[3], const float dir[3], const float p0[3], const float p1[3],
                                    const float p2[3], float ijk[3])
{
    v3 o = ld3(org), d = ld3(dir);
    v3 v0r = sub3(ld3(p0), o), v1r = sub3(ld3(p1), o), v2r = sub3(ld3(p2), o);
    v3 a = V3(fabsf(d.x), fabsf(d.y), fabsf(d.z));
    if (a.x > a.y) {
        if (a.x > a.z) {
            v0r = V3(v0r.z, v0r.y, v0r.x); v1r = V3(v1r.z, v1r.y, v1r.x); v2r = V3(v2r.z, v2r.y, v2r.x);
            d = V3(d.z, d.y, d.x);
        }
    } else {
        if (a.y > a.z) {
            v0r = V3(v0r.x, v0r.z, v0r.y); v1r = V3(v1r.x, v1r.z, v1r.y); v2r = V3(v2r.x, v2r.z, v2r.y);
            d = V3(d.x, d.z, d.y);
        }
    }
    float v0z = v0r.z, v1z = v1r.z, v2z = v2r.z;
    float inv_dz = 1.0f / d.z;
    v0r = sub3(v0r, scale3(v0r.z * inv_dz, d)); /* :51 v0r.z*inv_dz*d */
    v1r = sub3(v1r, scale3(v1r.z * inv_dz, d));
    v2r = sub3(v2r, scale3(v2r.z * inv_dz, d));
    float e0 = v1r.x * v2r.y - v1r.y * v2r.x;
    float e1 = v2r.x * v0r.y - v2r.y * v0r.x;
    float e2 = v0r.x * v1r.y - v0r.y * v1r.x;
    ijk[0] = ijk[1] = ijk[2] = 0.0f;
    if ((e0 < 0.0 || e1 < 0.0 || e2 < 0.0) && (e0 > 0.0 || e1 > 0.0 || e2 > 0.0)) return -1.0f;
    float det = e0 + e1 + e2;
    if (det == 0.0) return -1.0f;
    float inv_det = 1.0f / det;
    float t_scaled = (e0 * v0z + e1 * v1z + e2 * v2z) * inv_dz;
    float t = t_scaled * inv_det;
    if (t > kTriEpsilon) {
        ijk[0] = e0 * inv_det; ijk[1] = e1 * inv_det; ijk[2] = e2 * inv_det;
        return t;
    }
    return -1.0f;
}

/* ------------------------------------------------------------------ textures, texture.cpp:31-156 */
static void bilinear(float k[4], int32_t idx[4], int32_t component, v2 uv, int32_t width, int32_t height, int soa)
{
    float s = uv.x - floorf(uv.x);
    float t = uv.y - floorf(uv.y);
    float xt, yt;
    if (soa) { /* :64-65 _mm256_max_ps(a, 0) */
        xt = sse_max(s * (float)width - 0.5f, 0.0f);
        yt = sse_max(t * (float)height - 0.5f, 0.0f);
    } else { /* :36-37 std::max(a, 0.0f) */
        xt = std_max(s * (float)width - 0.5f, 0.0f);
        yt = std_max(t * (float)height - 0.5f, 0.0f);
    }
    int32_t x0 = (int32_t)floorf(xt), y0 = (int32_t)floorf(yt);
    int32_t x1 = (x0 + 1 < width - 1) ? x0 + 1 : width - 1;
    int32_t y1 = (y0 + 1 < height - 1) ? y0 + 1 : height - 1;
    idx[0] = component * (x0 + y0 * width);
    idx[1] = component * (x1 + y0 * width);
    idx[2] = component * (x0 + y1 * width);
    idx[3] = component * (x1 + y1 * width);
    s = xt - (float)x0;
    t = yt - (float)y0;
    k[0] = (1.0f - s) * (1.0f - t);
    k[1] = s * (1.0f - t);
    k[2] = (1.0f - s) * t;
    k[3] = s * t;
}

static int tex_test_alpha(const texture_t* tex, v2 uv, int soa, orc_stats* st) /* :142-183 */
{
    float k[4];
    int32_t idx[4];
    bilinear(k, idx, tex->component, uv, tex->width, tex->height, soa);
    if (st) st->nTap++;
    float alpha = 0.0f;
    for (int i = 0; i < 4; i++) alpha = alpha + k[i] * (float)tex->texels[idx[i] + 3];
    return alpha > 127.0f;
}

static v3 tex_sample3(const texture_t* tex, v2 uv, orc_stats* st) /* :106-139, T=Vector3f, C=uint8_t */
{
    float k[4];
    int32_t idx[4];
    bilinear(k, idx, tex->component, uv, tex->width, tex->height, 0);
    if (st) st->nTap++;
    v3 c = v3s(0.0f);
    for (int i = 0; i < 4; i++) {
        const uint8_t* p = tex->texels + idx[i];
        c = add3(c, scale3(k[i], V3((float)p[0], (float)p[1], (float)p[2])));
    }
    return scale3(1.0f / 255.0f, c);
}

static float tex_sample1(const texture_t* tex, v2 uv, orc_stats* st) /* T=float */
{
    float k[4];
    int32_t idx[4];
    bilinear(k, idx, tex->component, uv, tex->width, tex->height, 0);
    if (st) st->nTap++;
    float c = 0.0f;
    for (int i = 0; i < 4; i++) c = c + k[i] * (float)tex->texels[idx[i]];
    return (1.0f / 255.0f) * c;
}

/* ------------------------------------------------------------------ surface + material */
typedef struct { /* mesh.h:17-27 */
    v3 normal;
    const orc_material* material;
    v2 uv, duv01, duv02;
    v3 dp01, dp02;
} surf_t;

/* mesh.cpp:311-364 */
static void get_surface(const orc_scene* s, surf_t* prop, const orc_hit* hit, orc_stats* st)
{
    const orc_mesh* m = s->bvh[hit->meshId]->mesh;
    if (st) st->nHit++;
    uint32_t base = hit->primId * 3;
    uint32_t v0 = m->indices[base], v1 = m->indices[base + 1], v2i = m->indices[base + 2];
    v3 p0 = m->positions[v0], p1 = m->positions[v1], p2 = m->positions[v2i];
    v3 normal;
    if (m->normals) {
        v3 n = add3(add3(scale3(hit->i, m->normals[v0]), scale3(hit->j, m->normals[v1])), scale3(hit->k, m->normals[v2i]));
        normal = normalize3(n);
    } else {
        normal = normalize3(cross3(sub3(p1, p0), sub3(p2, p0)));
    }
    v2 t0, t1, t2;
    if (m->hasTexcoord) {
        t0 = m->texcoords[v0]; t1 = m->texcoords[v1]; t2 = m->texcoords[v2i]; /* texcoord index == vertex index, mesh.cpp:179,280 */
    } else {
        t0 = V2(0.0f, 0.0f); t1 = V2(1.0f, 0.0f); t2 = V2(0.0f, 1.0f);
    }
    prop->uv = add2(add2(scale2(hit->i, t0), scale2(hit->j, t1)), scale2(hit->k, t2));
    prop->normal = normal;
    prop->material = &m->materials[m->primMaterial[hit->primId]];
    prop->dp01 = normalize3(sub3(p1, p0));
    prop->dp02 = normalize3(sub3(p2, p0));
    prop->duv01 = safe_normalize2(sub2(t1, t0));
    prop->duv02 = safe_normalize2(sub2(t2, t0));
}

/* material.cpp:87-96; degamma :24-28 */
static v3 sample_diffuse(const orc_scene* s, const orc_material* mat, v2 uv, orc_stats* st)
{
    v3 color = ld3(mat->diffuse);
    if (mat->diffuseMap >= 0) {
        v3 c = tex_sample3(&s->textures[mat->diffuseMap], uv, st);
        color = mul3(color, V3(powf(c.x, 2.2f), powf(c.y, 2.2f), powf(c.z, 2.2f)));
    }
    return color;
}

/* material.cpp:98-114 */
static v3 sample_bump(const orc_scene* s, const orc_material* mat, const surf_t* prop, orc_stats* st)
{
    v3 normal = prop->normal;
    if (mat->bumpMap >= 0) {
        const texture_t* tex = &s->textures[mat->bumpMap];
        float onePixel = 0.5f / (float)tex->width + 0.5f / (float)tex->height; /* texture.h:26 */
        float b = tex_sample1(tex, prop->uv, st);
        float b01 = tex_sample1(tex, add2(prop->uv, scale2(onePixel, prop->duv01)), st) - b;
        float b02 = tex_sample1(tex, add2(prop->uv, scale2(onePixel, prop->duv02)), st) - b;
        float nk = 4.0f;
        normal = normalize3(add3(add3(normal, scale3(nk * b01, prop->dp01)), scale3(nk * b02, prop->dp02)));
    }
    return normal;
}

/* ---- explicit-argument entry points (used by oracle/ref_glue.cpp, which lends these restated
 * leaf functions to the compiled reference where its own mesh.cpp/material.cpp/texture.cpp
 * cannot be built) ---- */
void orc_x_get_surface(const orc_mesh* m, uint32_t primId, float i, float j, float k, float normal[3], uint32_t* matIndex,
                       float uv[2], float duv01[2], float duv02[2], float dp01[3], float dp02[3])
{
    orc_scene tmp;
    memset(&tmp, 0, sizeof(tmp));
    orc_bvh b;
    memset(&b, 0, sizeof(b));
    b.mesh = (orc_mesh*)m;
    orc_bvh* bl[1] = {&b};
    tmp.bvh = bl;
    tmp.bvhCount = 1;
    orc_hit h = {0.0f, i, j, k, primId, 0};
    surf_t p;
    get_surface(&tmp, &p, &h, NULL);
    st3(normal, p.normal);
    *matIndex = (uint32_t)(p.material - m->materials);
    uv[0] = p.uv.x; uv[1] = p.uv.y;
    duv01[0] = p.duv01.x; duv01[1] = p.duv01.y;
    duv02[0] = p.duv02.x; duv02[1] = p.duv02.y;
    st3(dp01, p.dp01);
    st3(dp02, p.dp02);
}

int orc_x_tex_test_alpha(int32_t w, int32_t h, int32_t comp, const uint8_t* texels, float u, float v, int soa)
{
    texture_t t = {w, h, comp, (uint8_t*)texels};
    return tex_test_alpha(&t, V2(u, v), soa, NULL);
}

void orc_x_sample_diffuse(const float diffuse[3], int32_t w, int32_t h, int32_t comp, const uint8_t* texels, float u,
                          float v, float out[3])
{
    orc_scene tmp;
    memset(&tmp, 0, sizeof(tmp));
    texture_t t = {w, h, comp, (uint8_t*)texels};
    tmp.textures = &t;
    tmp.textureCount = 1;
    orc_material mat;
    memset(&mat, 0, sizeof(mat));
    memcpy(mat.diffuse, diffuse, 12);
    mat.diffuseMap = texels ? 0 : -1;
    mat.bumpMap = -1;
    st3(out, sample_diffuse(&tmp, &mat, V2(u, v), NULL));
}

void orc_x_sample_bump(const float normal[3], int32_t w, int32_t h, int32_t comp, const uint8_t* texels,
                       const float uv[2], const float duv01[2], const float duv02[2], const float dp01[3],
                       const float dp02[3], float out[3])
{
    orc_scene tmp;
    memset(&tmp, 0, sizeof(tmp));
    texture_t t = {w, h, comp, (uint8_t*)texels};
    tmp.textures = &t;
    tmp.textureCount = 1;
    orc_material mat;
    memset(&mat, 0, sizeof(mat));
    mat.diffuseMap = -1;
    mat.bumpMap = texels ? 0 : -1;
    surf_t p;
    memset(&p, 0, sizeof(p));
    p.normal = ld3(normal);
    p.uv = V2(uv[0], uv[1]);
    p.duv01 = V2(duv01[0], duv01[1]);
    p.duv02 = V2(duv02[0], duv02[1]);
    p.dp01 = ld3(dp01);
    p.dp02 = ld3(dp02);
    st3(out, sample_bump(&tmp, &mat, &p, NULL));
}

/* ------------------------------------------------------------------ leaf intersection */
typedef struct {
    float org[3], dir[3], invDir[3];
    int swapXZ, swapYZ;
} ray1_t;

typedef struct {
    float org[LANES][3], dir[LANES][3], invDir[LANES][3];
    int swapXZ[LANES], swapYZ[LANES];
    float maxT[LANES];
    float avgDir[3];
} ray8_t;

static void tri_fetch(const trivec_t* tv, int lane, float p0[3], float p1[3], float p2[3])
{
    for (int k = 0; k < 3; k++) { p0[k] = tv->p[0][k][lane]; p1[k] = tv->p[1][k][lane]; p2[k] = tv->p[2][k][lane]; }
}

static v2 tri_uv(const trivec_t* tv, int lane, const float ijk[3]) /* bvh.cpp:332-336, 403-407 */
{
    v2 uv0 = V2(tv->uv[0][0][lane], tv->uv[0][1][lane]);
    v2 uv1 = V2(tv->uv[1][0][lane], tv->uv[1][1][lane]);
    v2 uv2 = V2(tv->uv[2][0][lane], tv->uv[2][1][lane]);
    return add2(add2(scale2(ijk[0], uv0), scale2(ijk[1], uv1)), scale2(ijk[2], uv2));
}

/* bvh.cpp:302-368.  mode 0 = nearest, 1 = occlude.  Returns 1 when occluded. */
static int intersect_single_leaf(const orc_scene* s, const orc_bvh* b, const orc_node* node, const ray1_t* ray,
                                 orc_hit* hit, float maxT, int mode, orc_stats* st)
{
    const trivec_t* tv = &b->triVectors[node->triVectorIndex];
    const uint32_t* primIndices = &b->primRemapping[node->primOrSecondNodeIndex];
    const orc_mesh* m = b->mesh;
    float nearestT = FLT_MAX;
    int nearestLane = -1;
    float nijk[3] = {0, 0, 0};
    if (st) st->nTri += node->primCount;
    for (uint32_t lane = 0; lane < node->primCount; lane++) {
        float p0[3], p1[3], p2[3], ijk[3];
        tri_fetch(tv, (int)lane, p0, p1, p2);
        float t = orc_intersect_triangle(ray->org, ray->dir, ray->swapXZ, ray->swapYZ, p0, p1, p2, ijk);
        if (!(t >= kTriEpsilon && t < maxT)) continue;
        if (tv->alphaTest[lane]) {
            const orc_material* mat = &m->materials[m->primMaterial[primIndices[lane]]];
            /* nTap counts NECESSARY taps: a candidate that is not nearer than the best of this leaf so far cannot
             * change the result, whatever its alpha (the reference still samples it, bvh.cpp:328-340) */
            orc_stats* cst = (mode == 1 || nearestT > t) ? st : NULL;
            if (!tex_test_alpha(&s->textures[mat->diffuseMap], tri_uv(tv, (int)lane, ijk), 0, cst)) continue;
        }
        if (mode == 0) {
            if (nearestT > t) {
                nearestT = t;
                nearestLane = (int)lane;
                nijk[0] = ijk[0]; nijk[1] = ijk[1]; nijk[2] = ijk[2];
            }
        } else {
            return 1;
        }
    }
    if (mode == 0 && nearestLane >= 0) {
        hit->t = nearestT;
        hit->i = nijk[0]; hit->j = nijk[1]; hit->k = nijk[2];
        hit->primId = primIndices[nearestLane];
        hit->meshId = m->id;
    }
    return 0;
}

/* bvh.cpp:370-427.  mask = lanes for which the node's box test passed.  Returns occluded bits (mode 1). */
static uint32_t intersect_packet_leaf(const orc_scene* s, const orc_bvh* b, const orc_node* node, uint32_t mask,
                                      const ray8_t* pk, orc_hit hits[LANES], int mode, orc_stats* st)
{
    const trivec_t* tv = &b->triVectors[node->triVectorIndex];
    const uint32_t* primIndices = &b->primRemapping[node->primOrSecondNodeIndex];
    const orc_mesh* m = b->mesh;
    uint32_t res = 0;
    if (st) st->nTri += (uint64_t)node->primCount * (uint64_t)__builtin_popcount(mask);
    for (uint32_t i = 0; i < node->primCount; i++) {
        float p0[3], p1[3], p2[3];
        tri_fetch(tv, (int)i, p0, p1, p2);
        for (int l = 0; l < LANES; l++) {
            if (!(mask & (1u << l))) continue; /* masked lanes return t = -1 (triangle.cpp:130-134,160) */
            float ijk[3];
            float t = orc_intersect_triangle(pk->org[l], pk->dir[l], pk->swapXZ[l], pk->swapYZ[l], p0, p1, p2, ijk);
            int maskHit;
            if (mode == 0) maskHit = (t >= kTriEpsilon) && (t < hits[l].t);
            else maskHit = (t >= kTriEpsilon) && (t < pk->maxT[l]);
            if (!maskHit) continue;
            uint32_t primIndex = primIndices[i];
            if (tv->alphaTest[i]) {
                const orc_material* mat = &m->materials[m->primMaterial[primIndex]];
                /* necessary taps only (see intersect_single_leaf): a lane already occluded by an earlier triangle of
                 * this leaf needs no further alpha test, although the reference keeps testing it (bvh.cpp:376-424) */
                orc_stats* cst = (mode == 1 && (res & (1u << l))) ? NULL : st;
                if (!tex_test_alpha(&s->textures[mat->diffuseMap], tri_uv(tv, (int)i, ijk), 1, cst)) continue;
            }
            if (mode == 0) {
                hits[l].t = t; hits[l].i = ijk[0]; hits[l].j = ijk[1]; hits[l].k = ijk[2];
                hits[l].primId = primIndex;
                hits[l].meshId = m->id;
            } else {
                res |= 1u << l;
            }
        }
    }
    return res;
}

/* ------------------------------------------------------------------ traversal */
/* bvh.cpp:429-570, single-ray branch (SORT_CHILDREN) */
static void bvh_intersect_single(const orc_scene* s, const orc_bvh* b, const ray1_t* ray, orc_hit* hit, orc_stats* st)
{
    const orc_node* nodes[STACK_SIZE];
    int32_t current = 0;
    nodes[0] = &b->nodes[0];
    if (st) st->nBox++;
    if (!orc_bbox_intersect_bool(b->nodes[0].lower, b->nodes[0].upper, ray->org, ray->invDir, hit->t)) return;
    do {
        const orc_node* node = nodes[current];
        if (node->primCount == INTERNAL_NODE) {
            const orc_node* node0 = node + 1;
            const orc_node* node1 = &b->nodes[node->primOrSecondNodeIndex];
            if (st) st->nBox += 2;
            float t0 = orc_bbox_intersect_t(node0->lower, node0->upper, ray->org, ray->invDir);
            float t1 = orc_bbox_intersect_t(node1->lower, node1->upper, ray->org, ray->invDir);
            int hit0 = t0 < hit->t;
            int hit1 = t1 < hit->t;
            if (!hit0) {
                if (hit1) { nodes[current] = node1; current++; }
            } else if (!hit1) {
                nodes[current] = node0; current++;
            } else if (t0 < t1) {
                nodes[current] = node1; nodes[current + 1] = node0; current += 2;
            } else {
                nodes[current] = node0; nodes[current + 1] = node1; current += 2;
            }
            if (current >= STACK_SIZE) abort(); /* bvh.cpp:552 */
        } else {
            intersect_single_leaf(s, b, node, ray, hit, hit->t, 0, st);
        }
        current--;
    } while (current >= 0);
}

/* bvh.cpp:429-570, packet branch */
static void bvh_intersect_packet(const orc_scene* s, const orc_bvh* b, const ray8_t* pk, orc_hit hits[LANES], orc_stats* st)
{
    const orc_node* nodes[STACK_SIZE];
    uint32_t masks[STACK_SIZE];
    int32_t current = 0;
    nodes[0] = &b->nodes[0];
    masks[0] = 0xff;
    int reverse[3];
    for (int i = 0; i < 3; i++) reverse[i] = pk->avgDir[i] < 0.0f;
    do {
        const orc_node* node = nodes[current];
        uint32_t mask = masks[current];
        if (st) st->nBox += (uint64_t)__builtin_popcount(mask);
        for (int l = 0; l < LANES; l++) {
            if (!(mask & (1u << l))) continue;
            if (!orc_bbox_intersect_soa(node->lower, node->upper, pk->org[l], pk->invDir[l], hits[l].t)) mask &= ~(1u << l);
        }
        if (!mask) {
        } else if (node->primCount == INTERNAL_NODE) {
            if (reverse[node->splitAxis]) {
                nodes[current] = node + 1;
                nodes[current + 1] = &b->nodes[node->primOrSecondNodeIndex];
            } else {
                nodes[current] = &b->nodes[node->primOrSecondNodeIndex];
                nodes[current + 1] = node + 1;
            }
            masks[current] = mask;
            masks[current + 1] = mask;
            current += 2;
            if (current >= STACK_SIZE) abort();
        } else {
            intersect_packet_leaf(s, b, node, mask, pk, hits, 0, st);
        }
        current--;
    } while (current >= 0);
}

/* bvh.cpp:576-654, single */
static int bvh_occluded_single(const orc_scene* s, const orc_bvh* b, const ray1_t* ray, float maxT, orc_stats* st)
{
    const orc_node* nodes[STACK_SIZE];
    int32_t current = 0;
    nodes[0] = &b->nodes[0];
    do {
        const orc_node* node = nodes[current];
        if (st) st->nBox++;
        int hit = orc_bbox_intersect_bool(node->lower, node->upper, ray->org, ray->invDir, maxT);
        if (!hit) {
        } else if (node->primCount == INTERNAL_NODE) {
            nodes[current] = &b->nodes[node->primOrSecondNodeIndex];
            nodes[current + 1] = node + 1;
            current += 2;
            if (current >= STACK_SIZE) abort();
        } else {
            if (intersect_single_leaf(s, b, node, ray, NULL, maxT, 1, st)) return 1;
        }
        current--;
    } while (current >= 0);
    return 0;
}

/* bvh.cpp:576-654, packet.  active = lanes still to be resolved; returns the occluded mask with
 * inactive lanes set (occludeMask = ~_mask, :590). */
static uint32_t bvh_occluded_packet(const orc_scene* s, const orc_bvh* b, uint32_t active, const ray8_t* pk, orc_stats* st)
{
    const orc_node* nodes[STACK_SIZE];
    uint32_t masks[STACK_SIZE];
    int32_t current = 0;
    nodes[0] = &b->nodes[0];
    masks[0] = active & 0xff;
    uint32_t occludeMask = (~active) & 0xff;
    do {
        const orc_node* node = nodes[current];
        uint32_t mask = masks[current] & ~occludeMask & 0xff;
        if (st) st->nBox += (uint64_t)__builtin_popcount(mask);
        for (int l = 0; l < LANES; l++) {
            if (!(mask & (1u << l))) continue;
            if (!orc_bbox_intersect_soa(node->lower, node->upper, pk->org[l], pk->invDir[l], pk->maxT[l])) mask &= ~(1u << l);
        }
        if (!mask) {
        } else if (node->primCount == INTERNAL_NODE) {
            nodes[current] = &b->nodes[node->primOrSecondNodeIndex];
            nodes[current + 1] = node + 1;
            masks[current] = mask;
            masks[current + 1] = mask;
            current += 2;
            if (current >= STACK_SIZE) abort();
        } else {
            uint32_t res = intersect_packet_leaf(s, b, node, mask, pk, NULL, 1, st);
            occludeMask |= res;
            if (occludeMask == 0xff) return occludeMask;
        }
        current--;
    } while (current >= 0);
    return occludeMask;
}

/* ------------------------------------------------------------------ accounting of the any-hit queries
 * Test infrastructure for the GPU path's event counters; RESULTS ARE NEVER TAKEN FROM HERE.  An occlusion query asks
 * whether any triangle below boxes the ray passes is accepted; with maxT constant every box and triangle test depends
 * on (ray, box or triangle) alone, so the answer does not depend on the visiting order.  prt_amd visits the child whose
 * box the ray enters first (ties: child 0) instead of the reference's fixed order (bvh.cpp:617-620).  With
 * orc_set_anyhit_accounting(1) nBox/nTri/nTap of occlusion queries count THAT visit, ray by ray (1 per root test, 2 per
 * internal node, every triangle tested up to the first accepted one), and the wrappers below abort() if its answer
 * ever differs from the reference traversal's -- the order-independence is checked on every ray the oracle traces. */
static int g_anyhit_accounting = 0;
void orc_set_anyhit_accounting(int mode) { g_anyhit_accounting = mode; }

static float any_hit_key(const orc_node* n, const float org[3], const float invDir[3]) /* prt_device.h slab_entry_select */
{
    float a, c, m, v;
    a = (n->lower[0] - org[0]) * invDir[0]; c = (n->upper[0] - org[0]) * invDir[0]; m = (a < c) ? a : c;
    a = (n->lower[1] - org[1]) * invDir[1]; c = (n->upper[1] - org[1]) * invDir[1]; v = (a < c) ? a : c; m = (v > m) ? v : m;
    a = (n->lower[2] - org[2]) * invDir[2]; c = (n->upper[2] - org[2]) * invDir[2]; v = (a < c) ? a : c; m = (v > m) ? v : m;
    return m;
}

static int any_hit_box(const orc_node* n, const float org[3], const float invDir[3], float maxT, int soa)
{
    return soa ? orc_bbox_intersect_soa(n->lower, n->upper, org, invDir, maxT) : orc_bbox_intersect_bool(n->lower, n->upper, org, invDir, maxT);
}

static int count_any_hit(const orc_scene* s, const orc_bvh* b, const float org[3], const float dir[3], const float invDir[3],
                         int swapXZ, int swapYZ, float maxT, int soa, orc_stats* st)
{
    const orc_node* nodes[STACK_SIZE];
    int32_t sp = 0;
    const orc_mesh* m = b->mesh;
    st->nBox++;
    if (!any_hit_box(&b->nodes[0], org, invDir, maxT, soa)) return 0;
    const orc_node* node = &b->nodes[0];
    for (;;) {
        if (node->primCount == INTERNAL_NODE) {
            const orc_node* c0 = node + 1;
            const orc_node* c1 = &b->nodes[node->primOrSecondNodeIndex];
            st->nBox += 2;
            int h0 = any_hit_box(c0, org, invDir, maxT, soa), h1 = any_hit_box(c1, org, invDir, maxT, soa);
            if (h0 && h1) {
                int near0 = any_hit_key(c0, org, invDir) <= any_hit_key(c1, org, invDir);
                if (sp >= STACK_SIZE) abort();
                nodes[sp++] = near0 ? c1 : c0;
                node = near0 ? c0 : c1;
                continue;
            }
            if (h0) { node = c0; continue; }
            if (h1) { node = c1; continue; }
        } else {
            const trivec_t* tv = &b->triVectors[node->triVectorIndex];
            const uint32_t* primIndices = &b->primRemapping[node->primOrSecondNodeIndex];
            for (uint32_t i = 0; i < node->primCount; i++) {
                float p0[3], p1[3], p2[3], ijk[3];
                tri_fetch(tv, (int)i, p0, p1, p2);
                st->nTri++;
                float t = orc_intersect_triangle(org, dir, swapXZ, swapYZ, p0, p1, p2, ijk);
                if (!(t >= kTriEpsilon && t < maxT)) continue;
                if (tv->alphaTest[i]) {
                    const orc_material* mat = &m->materials[m->primMaterial[primIndices[i]]];
                    if (!tex_test_alpha(&s->textures[mat->diffuseMap], tri_uv(tv, (int)i, ijk), soa, st)) continue;
                }
                return 1;
            }
        }
        if (sp == 0) return 0;
        node = nodes[--sp];
    }
}

static void make_ray1(ray1_t* r, const float org[3], const float dir[3])
{
    memcpy(r->org, org, 12);
    memcpy(r->dir, dir, 12);
    orc_ray_prepare_single(dir, r->invDir, &r->swapXZ, &r->swapYZ);
}

static void make_ray8(ray8_t* pk, const float org[LANES][3], const float dir[LANES][3], const float avgDir[3], float maxT)
{
    for (int l = 0; l < LANES; l++) {
        memcpy(pk->org[l], org[l], 12);
        memcpy(pk->dir[l], dir[l], 12);
        orc_ray_prepare_soa(dir[l], pk->invDir[l], &pk->swapXZ[l], &pk->swapYZ[l]);
        pk->maxT[l] = maxT;
    }
    if (avgDir) memcpy(pk->avgDir, avgDir, 12);
    else pk->avgDir[0] = pk->avgDir[1] = pk->avgDir[2] = 0.0f;
}

/* scene.cpp:47-63 */
void orc_intersect_single(const orc_scene* s, const float org[3], const float dir[3], float maxT, orc_hit* hit, orc_stats* st)
{
    ray1_t ray;
    make_ray1(&ray, org, dir);
    memset(hit, 0, sizeof(*hit));
    hit->t = maxT;
    for (uint32_t i = 0; i < s->bvhCount; i++) bvh_intersect_single(s, s->bvh[i], &ray, hit, st);
    if (hit->t == maxT) hit->t = -1.0f;
}

void orc_intersect_packet(const orc_scene* s, const float org[8][3], const float dir[8][3], const float avgDir[3],
                          float maxT, orc_hit hits[8], orc_stats* st)
{
    ray8_t pk;
    make_ray8(&pk, org, dir, avgDir, maxT);
    memset(hits, 0, sizeof(orc_hit) * LANES);
    for (int l = 0; l < LANES; l++) hits[l].t = maxT;
    for (uint32_t i = 0; i < s->bvhCount; i++) bvh_intersect_packet(s, s->bvh[i], &pk, hits, st);
    for (int l = 0; l < LANES; l++)
        if (hits[l].t == maxT) hits[l].t = -1.0f;
}

/* scene.cpp:69-94 */
int orc_occluded_single(const orc_scene* s, const float org[3], const float dir[3], float maxT, orc_stats* st)
{
    ray1_t ray;
    make_ray1(&ray, org, dir);
    const int recount = g_anyhit_accounting && st;
    int counted = 0;
    if (recount) {
        for (uint32_t i = 0; i < s->bvhCount && !counted; i++)
            counted = count_any_hit(s, s->bvh[i], ray.org, ray.dir, ray.invDir, ray.swapXZ, ray.swapYZ, maxT, 0, st);
        st = NULL; /* the reference traversal below gives the answer and counts nothing */
    }
    int res = 0;
    for (uint32_t i = 0; i < s->bvhCount && !res; i++) res = bvh_occluded_single(s, s->bvh[i], &ray, maxT, st);
    if (recount && res != counted) abort(); /* the visiting order changed an any-hit answer: impossible */
    return res;
}

uint32_t orc_occluded_packet(const orc_scene* s, uint32_t activeMask, const float org[8][3], const float dir[8][3],
                             float maxT, orc_stats* st)
{
    ray8_t pk;
    make_ray8(&pk, org, dir, NULL, maxT);
    const int recount = g_anyhit_accounting && st;
    uint32_t counted = (~activeMask) & 0xff;
    if (recount) {
        for (int l = 0; l < LANES; l++) {
            if (!(activeMask & (1u << l))) continue;
            int c = 0;
            for (uint32_t i = 0; i < s->bvhCount && !c; i++)
                c = count_any_hit(s, s->bvh[i], pk.org[l], pk.dir[l], pk.invDir[l], pk.swapXZ[l], pk.swapYZ[l], pk.maxT[l], 1, st);
            if (c) counted |= 1u << l;
        }
        st = NULL;
    }
    uint32_t occludeMask = (~activeMask) & 0xff;
    for (uint32_t i = 0; i < s->bvhCount; i++) {
        uint32_t res = bvh_occluded_packet(s, s->bvh[i], (~occludeMask) & 0xff, &pk, st);
        occludeMask |= res;
        if (occludeMask == 0xff) break;
    }
    if (recount && occludeMask != counted) abort();
    return occludeMask;
}

/* Texture::sample<Vector3f, float>, texture.cpp:102-139: scalar bilinear helper, c = c + k[i]*rgb, returned unscaled */
static v3 tex_sample3f(const float* texels, int32_t width, int32_t height, int32_t component, v2 uv, orc_stats* st)
{
    float k[4];
    int32_t idx[4];
    bilinear(k, idx, component, uv, width, height, 0);
    if (st) st->nTap++;
    v3 c = v3s(0.0f);
    for (int i = 0; i < 4; i++) {
        const float* p = texels + idx[i];
        c = add3(c, scale3(k[i], V3(p[0], p[1], p[2])));
    }
    return c;
}

void orc_x_tex_sample3f(int32_t w, int32_t h, int32_t comp, const float* texels, float u, float v, float out[3])
{
    st3(out, tex_sample3f(texels, w, h, comp, V2(u, v), NULL));
}

/* ------------------------------------------------------------------ InfiniteAreaLight::sample, light.cpp:86-128
 * Texture::sample<Vector3f> on k32Float texels (texture.cpp:88-139): the scalar bilinear helper, c += k[i]*rgb, no scale.
 * Defined where the reference is undefined: when no vertical CDF entry exceeds u.y the reference leaves y == m_height and
 * reads m_horizontalP one row past its end (:110-112); here that row is treated as "no entry exceeds u.x" (xf = 0,
 * pdfH = 1), which is what finite heap garbage below u.x would give. */
static void env_sample(const orc_scene* s, float ux, float uy, v3* dir, v3* color, orc_stats* st)
{
    const int32_t W = s->envWidth, H = s->envHeight;
    const float* vp = s->envVerticalP;
    const float* hp = s->envHorizontalP;
    int32_t y;
    float pdfV = 1.0f, yf = 0.0f;
    for (y = 1; y < H; y++) {
        if (vp[y] > uy) {
            float prev = vp[y - 1];
            float pdf = vp[y] - prev;
            if (pdf == 0.0f) continue;
            pdfV = pdf;
            yf = (float)y + (uy - prev) / pdfV - 1.0f;
            break;
        }
    }
    float pdfH = 1.0f, xf = 0.0f;
    if (y < H) {
        for (int32_t x = 1; x < W; x++) {
            if (hp[x + y * W] > ux) {
                float prev = hp[x - 1 + y * W];
                float pdf = hp[x + y * W] - prev;
                if (pdf == 0.0f) continue;
                pdfH = pdf;
                xf = (float)x + (ux - prev) / pdfH - 1.0f;
                break;
            }
        }
    }
    v2 uv = V2(xf / (float)W, yf / (float)H);
    v3 c = tex_sample3f(s->envTexels, W, H, 4, uv, st);
    c = div3(div3(c, v3s(pdfH * pdfV)), v3s((float)(W * H))); /* :118: Vector3f / Vector3f(float), vecmath.h:263 */
    float theta = 2.0f * kPi * (uv.x + 0.5f);
    float phi = kPi * uv.y;
    float sinPhi = sinf(phi);
    *dir = normalize3(V3(cosf(theta) * sinPhi, cosf(phi), sinf(theta) * sinPhi));
    *color = c;
}

void orc_x_env_sample(const orc_scene* s, uint32_t n, const float* u, float* dirOut, float* colorOut)
{
    for (uint32_t i = 0; i < n; i++) {
        v3 d, c;
        env_sample(s, u[2 * i], u[2 * i + 1], &d, &c, NULL);
        st3(dirOut + 3 * i, d);
        st3(colorOut + 3 * i, c);
    }
}

/* ------------------------------------------------------------------ camera.cpp:35-73 */
static void camera_packet(const orc_camera* cam, uint32_t* rng, uint32_t x, uint32_t y, float org[LANES][3],
                          float dir[LANES][3], float avgDirOut[3], orc_stats* st)
{
    v3 avgDir = v3s(0.0f);
    const float kScreenScale = 0.6f;
    const float kAspect = (float)cam->width / (float)cam->height;
    const float kScaleX = 0.5f * cam->invWidth;
    const float kScaleY = 0.5f * cam->invHeight;
    v3 right = ld3(cam->right), up = ld3(cam->up), fwd = ld3(cam->dir);
    for (int i = 0; i < LANES; i++) {
        float dx = rng_pm1(rng, st) * kScaleX;
        float dy = rng_pm1(rng, st) * kScaleY;
        float nx = 2.0f * ((float)x * cam->invWidth - 0.5f + dx) * kScreenScale * kAspect;
        float ny = -2.0f * ((float)y * cam->invHeight - 0.5f + dy) * kScreenScale;
        v3 d = normalize3(add3(add3(scale3(nx, right), scale3(ny, up)), fwd));
        st3(dir[i], d);
        memcpy(org[i], cam->pos, 12);
        avgDir = add3(avgDir, d);
    }
    st3(avgDirOut, div3(avgDir, v3s(8.0f))); /* avgDir/RayPacket::kSize -> Vector3f(8.0f), vecmath.h:263 */
}

void orc_camera_packet(const orc_camera* cam, uint32_t* rng, uint32_t x, uint32_t y, float org[8][3], float dir[8][3],
                       float avgDir[3])
{
    camera_packet(cam, rng, x, y, org, dir, avgDir, NULL);
}

/* ------------------------------------------------------------------ GbufferVisualizer, gbuffer_visualizer.cpp:17-51
 * One jittered single ray per pixel (Camera::GenerateJitteredRay, camera.cpp:12-33: two generateMinus1to1 draws), nearest
 * hit by the single-ray traversal, then the surface's diffuse colour (type 0) or its bump-mapped normal * 0.5 + 0.5 (types 1
 * and 2 -- the reference gives kMeshNormal and kNormal the same expression).  Per-pixel generator state as for the path tracer. */
void orc_gbuffer_block(const orc_scene* scene, const orc_camera* cam, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t type,
                       uint32_t seed, float exposure, float* rgb)
{
    const float kScreenScale = 0.6f;
    const float kAspect = (float)cam->width / (float)cam->height;
    const float kScaleX = 0.5f * cam->invWidth;
    const float kScaleY = 0.5f * cam->invHeight;
    v3 right = ld3(cam->right), up = ld3(cam->up), fwd = ld3(cam->dir);
    for (uint32_t y = y0; y <= y1; y++) {
        for (uint32_t x = x0; x <= x1; x++) {
            uint32_t rng = orc_pixel_seed(x, y, cam->width, seed);
            float dx = rng_pm1(&rng, NULL) * kScaleX;
            float dy = rng_pm1(&rng, NULL) * kScaleY;
            float nx = 2.0f * ((float)x * cam->invWidth - 0.5f + dx) * kScreenScale * kAspect;
            float ny = -2.0f * ((float)y * cam->invHeight - 0.5f + dy) * kScreenScale;
            v3 d = normalize3(add3(add3(scale3(nx, right), scale3(ny, up)), fwd));
            float dir[3];
            st3(dir, d);
            orc_hit hit;
            orc_intersect_single(scene, cam->pos, dir, 100000.0f, &hit, NULL);
            v3 color = v3s(0.0f);
            if (hit.t != -1.0f) {
                surf_t prop;
                get_surface(scene, &prop, &hit, NULL);
                if (type == 0) color = sample_diffuse(scene, prop.material, prop.uv, NULL);
                else color = add3(scale3(0.5f, sample_bump(scene, prop.material, &prop, NULL)), v3s(0.5f));
            }
            st3(rgb + ((size_t)x + (size_t)y * cam->width) * 3, scale3(exposure, color)); /* image.cpp:45 */
        }
    }
}

/* ------------------------------------------------------------------ path_tracer.cpp:77-308 */
static v3 diffuse_dir(v3 normal, float r2, float r1) /* :143-153 / :176-184 */
{
    float r2sq = sqrtf(r2);
    v3 u = (fabsf(normal.x) > 0.1f) ? V3(0, 1.0f, 0.0f) : V3(1.0f, 0, 0);
    v3 tangent = normalize3(cross3(normal, u));
    v3 binormal = normalize3(cross3(tangent, normal));
    float theta = 2.0f * kPi * r1;
    return add3(add3(scale3(r2sq * cosf(theta), binormal), scale3(r2sq * sinf(theta), tangent)), scale3(1 - r2, normal));
}

/* Russian roulette starts when depth > this; 4 is the reference's literal (path_tracer.cpp:258).  A parameter of the
 * product (prt_render_params.rrDepth), so the checker has it too. */
static uint32_t g_rr_depth = 4;
void orc_set_rr_depth(uint32_t d) { g_rr_depth = d; }

static v3 compute_radiance(const orc_scene* scene, uint32_t* rng, const orc_hit hitPacket[LANES], const float org[LANES][3],
                           const float dirs[LANES][3], uint32_t maxDepth, uint32_t rrDepth, orc_stats* st)
{
    v3 result[LANES], beta[LANES];
    for (int i = 0; i < LANES; i++) { beta[i] = v3s(1.0f); result[i] = v3s(0.0f); }
    v3 pos[LANES], rayDir[LANES], normals[LANES];
    surf_t props[LANES];
    const orc_material* materials[LANES];
    memset(props, 0, sizeof(props));
    memset(pos, 0, sizeof(pos)); memset(rayDir, 0, sizeof(rayDir)); memset(normals, 0, sizeof(normals));
    memset(materials, 0, sizeof(materials));

    uint32_t alivePaths = 0;
    for (int lane = 0; lane < LANES; lane++) {
        const orc_hit* sr = &hitPacket[lane];
        if (sr->t != -1.0f) {
            get_surface(scene, &props[alivePaths], sr, st);
            materials[alivePaths] = props[alivePaths].material;
            normals[alivePaths] = sample_bump(scene, materials[alivePaths], &props[alivePaths], st);
            v3 d = ld3(dirs[lane]);
            pos[alivePaths] = add3(scale3(sr->t, d), ld3(org[lane])); /* hit.t*ray.dir + ray.org, :100 */
            rayDir[alivePaths] = d;
            alivePaths++;
        }
    }

    uint32_t depth = 0;
    v3 lightDir[LANES], lightIntensity[LANES], nextRayDir[LANES];
    memset(lightDir, 0, sizeof(lightDir)); memset(lightIntensity, 0, sizeof(lightIntensity));
    memset(nextRayDir, 0, sizeof(nextRayDir));

    while (depth < maxDepth) {
        int directLighting = 0;
        for (uint32_t path = 0; path < alivePaths; path++) {
            const orc_material* material = materials[path];
            v3 normal = normals[path];
            surf_t prop = props[path];
            if (material->emissive[0] != 0) result[path] = add3(result[path], mul3(beta[path], ld3(material->emissive)));
            if (material->reflectionType == 0) {
                float r2 = rng_f(rng, st);
                float r1 = rng_f(rng, st);
                nextRayDir[path] = diffuse_dir(normal, r2, r1);
                beta[path] = mul3(beta[path], sample_diffuse(scene, material, prop.uv, st));
                if (scene->hasEnv) { /* path_tracer.cpp:164-167; argument evaluation left to right (clang) */
                    float ux = rng_f(rng, st);
                    float uy = rng_f(rng, st);
                    env_sample(scene, ux, uy, &lightDir[path], &lightIntensity[path], st);
                    directLighting = 1;
                } else if (scene->hasDirectional) {
                    lightDir[path] = scene->lightDir;
                    lightIntensity[path] = scene->lightIntensity;
                    directLighting = 1;
                }
            } else if (material->reflectionType == 1) {
                float r2 = rng_f(rng, st);
                float r1 = rng_f(rng, st);
                v3 dd = diffuse_dir(normal, r2, r1);
                v3 rdir = rayDir[path];
                v3 reflectDir = sub3(rdir, scale3(dot3(normal, rdir), scale3(2.0f, normal))); /* normal*2*dot, :186 */
                nextRayDir[path] = add3(scale3(0.9f, reflectDir), scale3(0.1f, dd));
            }
        }

        const float kFar = 2.0f * scene->radius;
        const float kEpsilon = 0.0008f;

        if (directLighting) {
            int grouping = (alivePaths & 0xf) > 2;
            if (grouping) {
                float so[LANES][3], sd[LANES][3];
                for (int l = 0; l < LANES; l++) {
                    /* lanes >= alivePaths read stale stack contents in the reference (:207-208); they are
                     * masked out, so any finite value does. */
                    v3 ld = (l < (int)alivePaths) ? lightDir[l] : V3(0, 1, 0);
                    v3 p = (l < (int)alivePaths) ? pos[l] : v3s(0.0f);
                    st3(so[l], add3(p, scale3(kFar, ld)));
                    st3(sd[l], scale3(-1.0f, ld));
                }
                uint32_t maskBits = (1u << alivePaths) - 1;
                if (st) { st->raysTraced += alivePaths; st->occludedTraced += alivePaths; }
                uint32_t omask = orc_occluded_packet(scene, maskBits, so, sd, kFar - kEpsilon, st);
                for (uint32_t bits = (~omask) & 0xff; bits; bits &= bits - 1) {
                    int path = __builtin_ctz(bits);
                    v3 lr = div3(scale3(std_max(dot3(lightDir[path], normals[path]), 0.0f), lightIntensity[path]), v3s(kPi));
                    result[path] = add3(result[path], mul3(beta[path], lr));
                }
            } else {
                for (uint32_t i = 0; i < alivePaths; i++) {
                    float so[3], sd[3];
                    st3(so, add3(pos[i], scale3(kFar, lightDir[i])));
                    st3(sd, neg3(lightDir[i]));
                    if (st) { st->raysTraced++; st->occludedTraced++; }
                    if (!orc_occluded_single(scene, so, sd, kFar - kEpsilon, st)) {
                        v3 lr = div3(scale3(std_max(dot3(lightDir[i], normals[i]), 0.0f), lightIntensity[i]), v3s(kPi));
                        result[i] = add3(result[i], mul3(beta[i], lr));
                    }
                }
            }
        }

        uint32_t ci = 0;
        for (uint32_t i = 0; i < alivePaths; i++) {
            if (depth > rrDepth) {
                float q = std_max(0.05f, 1.0f - length3(beta[i]));
                if (rng_f(rng, st) < q) continue;
                beta[ci] = div3(beta[i], v3s(1.0f - q));
            }
            v3 dir = normalize3(nextRayDir[i]);
            float o[3], d[3];
            st3(o, pos[i]);
            st3(d, dir);
            if (st) st->raysTraced++;
            orc_hit hit;
            orc_intersect_single(scene, o, d, kFar, &hit, st);
            if (hit.t != -1.0f) {
                surf_t old = props[i]; /* props[i] is read AFTER props[ci] is overwritten; differs only when ci == i */
                get_surface(scene, &props[ci], &hit, st);
                if (ci == i) old = props[ci];
                pos[ci] = add3(scale3(hit.t, dir), pos[i]);
                normals[ci] = old.normal;
                materials[ci] = old.material;
                normals[ci] = sample_bump(scene, materials[ci], &props[ci], st);
                rayDir[ci] = dir;
                ci++;
            }
        }
        if (ci == 0) break;
        alivePaths = ci;
        depth++;
    }

    v3 res = v3s(0.0f);
    for (int i = 0; i < LANES; i++) res = add3(res, result[i]);
    return res;
}

/* path_tracer.cpp:17-33 + 57-75 */
static void trace_pixel(const orc_scene* scene, const orc_camera* cam, uint32_t x, uint32_t y, uint32_t samples,
                        uint32_t maxDepth, uint32_t seed, float exposure, float* rgb, orc_stats* st)
{
    uint32_t rng = orc_pixel_seed(x, y, cam->width, seed);
    v3 color = v3s(0.0f);
    if (st) st->raysTraced += samples;
    for (uint32_t i = 0; i < samples / LANES; i++) {
        float org[LANES][3], dir[LANES][3], avgDir[3];
        camera_packet(cam, &rng, x, y, org, dir, avgDir, st);
        orc_hit hits[LANES];
        orc_intersect_packet(scene, org, dir, avgDir, 100000.0f, hits, st);
        color = add3(color, compute_radiance(scene, &rng, hits, org, dir, maxDepth, g_rr_depth, st));
    }
    color = div3(color, v3s((float)samples));
    v3 c = scale3(exposure, color); /* image.cpp:45 */
    float* p = rgb + ((size_t)x + (size_t)y * cam->width) * 3;
    p[0] = c.x; p[1] = c.y; p[2] = c.z;
    if (st) st->nPx++;
}

static void stats_add(orc_stats* a, const orc_stats* b)
{
    a->raysTraced += b->raysTraced; a->occludedTraced += b->occludedTraced; a->nBox += b->nBox; a->nTri += b->nTri;
    a->nHit += b->nHit; a->nTap += b->nTap; a->nPx += b->nPx; a->rngDraws += b->rngDraws;
}

void orc_trace_block(const orc_scene* scene, const orc_camera* cam, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                     uint32_t samples, uint32_t maxDepth, uint32_t seed, float exposure, float* rgb, orc_stats* st)
{
    for (uint32_t y = y0; y <= y1; y++)
        for (uint32_t x = x0; x <= x1; x++) trace_pixel(scene, cam, x, y, samples, maxDepth, seed, exposure, rgb, st);
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_render(const orc_scene* scene, const orc_camera* cam, uint32_t samples, uint32_t maxDepth, uint32_t seed,
                float exposure, int threads, float* rgb, orc_stats* st)
{
    orc_render_rect(scene, cam, 0, 0, cam->width - 1, cam->height - 1, samples, maxDepth, seed, exposure, threads, rgb, st);
}

void orc_render_rect(const orc_scene* scene, const orc_camera* cam, uint32_t rx0, uint32_t ry0, uint32_t rx1, uint32_t ry1,
                     uint32_t samples, uint32_t maxDepth, uint32_t seed, float exposure, int threads, float* rgb, orc_stats* st)
{
    const uint32_t W = rx1 - rx0 + 1, H = ry1 - ry0 + 1;
    const uint32_t tilesX = (W + 15) / 16, tilesY = (H + 15) / 16; /* main.cpp:123-124 tile size */
    orc_stats total;
    memset(&total, 0, sizeof(total));
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
#endif
    {
        orc_stats local;
        memset(&local, 0, sizeof(local));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int32_t tile = 0; tile < (int32_t)(tilesX * tilesY); tile++) {
            uint32_t tx = (uint32_t)tile % tilesX, ty = (uint32_t)tile / tilesX;
            uint32_t x0 = rx0 + tx * 16, y0 = ry0 + ty * 16;
            uint32_t x1 = x0 + 15 < rx1 ? x0 + 15 : rx1;
            uint32_t y1 = y0 + 15 < ry1 ? y0 + 15 : ry1;
            orc_trace_block(scene, cam, x0, y0, x1, y1, samples, maxDepth, seed, exposure, rgb, st ? &local : NULL);
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        stats_add(&total, &local);
    }
    if (st) *st = total;
}

/* ------------------------------------------------------------------ libm as the reference calls it
 * (path_tracer.cpp:153,184 cosf/sinf; material.cpp:27 powf(x, 2.2f)) -- checker for the kernels' own versions */
void orc_libm_sincos(uint32_t n, const float* theta, float* s, float* c)
{
    for (uint32_t i = 0; i < n; i++) { s[i] = sinf(theta[i]); c[i] = cosf(theta[i]); }
}

void orc_libm_powf22(uint32_t n, const float* x, float* y)
{
    for (uint32_t i = 0; i < n; i++) y[i] = powf(x[i], 2.2f);
}
