// prt_device.h -- per-ray device functions of the path: box tests, watertight triangle test,
// the four BVH traversals, surface fetch and material sampling.  gfx950 only.
//
// Arithmetic contract: every expression below is the reference's expression in the reference's
// order, evaluated in IEEE binary32 with no contraction (the file is compiled -ffp-contract=off,
// no fast-math, correctly-rounded / and sqrt).  SSE min/max (second operand on NaN) are written
// as selects, never fminf/fmaxf.  Reference citations are file:line under /root/reference/src.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "prt_devmath.h"

#define PRT_MAX_BVH 8
#ifndef PRT_STACK_LDS
#define PRT_STACK_LDS 12     // stack entries per lane kept in LDS (4 B each): 48 KB of a 1024-thread workgroup's 78 KB (2 workgroups share a CU's 160 KB)
#endif
#ifndef PRT_STACK_LDS_PACKET
#define PRT_STACK_LDS_PACKET (PRT_STACK_LDS / 2) // the same for the packet traversal (8 B each: reference + entry distance)
#endif
#define PRT_STACK_MAX 64     // bvh.cpp:432,579
#ifndef PRT_BLOCK
#define PRT_BLOCK 1024 // threads per workgroup: 16 waves share the frame kernel's pool, queues and hot records (2 workgroups per CU)
#endif
#define PRT_MAT_STRIDE 5 // float4 per material record
#ifndef PRT_TRI2
#define PRT_TRI2 1 // a leaf step tests two triangles (one: 569 ms against 547 on C3)
#endif
#ifndef PRT_IDLE_BREAK
#define PRT_IDLE_BREAK 24 // finished rays that send the wave back to its refill point (a refill turn costs about four rounds: 8 -> 463, 16 -> 424, 24-32 -> 412-422 ms on C3)
#endif
// ---------------------------------------------------------------------------- device scene
// wnodes: 4 x float4 (64 B) per INTERNAL node, (lo, hi) pairs per axis so that the slab arithmetic runs on packed
//         f32 pairs: {lo0.x hi0.x lo0.y hi0.y} {lo0.z hi0.z lo1.z hi1.z} {lo1.x hi1.x lo1.y hi1.y} {ref0 ref1 splitAxis 0}
//         child 0 is the reference's node i+1, child 1 its m_nodes[primOrSecondNodeIndex]; refs: see PRT_REF_LEAF
// roots:  per BVH the root's reference and box (rootRef, rootBox)
// tris:   9 floats (36 B) per triangle SLOT: primRemapping (LEAF) order, all BVHs behind one another, a leaf's triangles contiguous;
//         a leaf may start a few unused slots late so that it touches one 128-byte line less (prt_kernels.hip leaf_slots): p0 p1 p2.  A traversal reads
//         nothing else of a triangle unless it is a candidate: trees far larger than the caches are bound by the rate of L2 misses
//         (profiles/r03_frame_c4_counters.json), so every byte a leaf visit does not need is kept out of its lines
// triAlpha: per triangle slot: 0, or 1 + index of its alpha record (read for candidates of leaves whose reference says
//         that they hold an alpha-tested triangle)
// triPrim:  per triangle slot: its primId in its mesh (hits carry the LEAF-ORDER index on the device; only the row-level
//         test entry points, which report the reference's primId, read this)
// shade:  4 x float4 per triangle slot {n0 mat} {n1 uv0.x} {n2 uv0.y} {uv1 uv2}
//         (n0 = precomputed face normal when the mesh has no vertex normals)
// bump:   3 x float4 per triangle slot {dp01 duv01.x} {dp02 duv01.y} {duv02.xy 0 0}
// mats:   5 x float4 per material {kd reflType} {ke alphaTest} {diffuseTex bumpTex 0 0} {descriptor of the diffuse map} {... of the bump map}
//         (descriptor = {byte offset, width, height, component}, as in texDesc)
// alpha:  3 x float4 per alpha-tested leaf triangle {uv0 uv1} {uv2 tex classWord} {descriptor of the texture: byte offset, width, height, component}
//         (classWord = first word of the texture's cell classes in alphaClass: the record, not a table, says where everything is)
// alphaClass: 2 bits per bilinear CELL (x0, y0) of every alpha-tested texture, 16 cells per word: 1 = the cell's four texels all have
//         alpha >= 128 (the blend of Texture::testAlpha exceeds 127 whatever the weights), 2 = all four <= 126 (it cannot), 0 = mixed:
//         the cooperative leaf rounds decide most alpha tests from one word that stays in the L2 instead of four texel fetches
struct DevScene {
    const float4* wnodes;
    const float4* hotNodes; // PRT_HOT_NODES records (4 x float4 each): copies of the records that PRT_REF_HOT references name
    const float* tris;
    const uint32_t* triAlpha;
    const uint32_t* triPrim;
    const float4* shade;
    const float4* bump;
    const float4* mats;
    const float4* alpha;
    const uint32_t* alphaClass;
    const uint4* texDesc; // {byte offset, width, height, component}
    const uint8_t* texels;
    uint32_t bvhCount;
    uint32_t rootRef[PRT_MAX_BVH];
    float rootBox[PRT_MAX_BVH][6];
    uint32_t primBase[PRT_MAX_BVH];
    uint32_t hasNormals[PRT_MAX_BVH];
    uint32_t hasLight;
    float lightDir[3];
    float lightIntensity[3];
    float radius;
    // InfiniteAreaLight (light.h:28-50): float RGBA texels and the two CDF tables of light.cpp:30-84; envFirstX[y] /
    // envFirstY = the first index >= 1 whose CDF entry differs from its predecessor (width / height when none does)
    uint32_t hasEnv;
    int32_t envW, envH, envFirstY;
    const float4* envTexels;
    const float* envV;
    const float* envHor;
    const int32_t* envFirstX;
};

struct DevCamera {
    float pos[3], dir[3], up[3], right[3];
    uint32_t width, height;
    float invWidth, invHeight;
};

struct Vec3 {
    float x, y, z;
};
struct Vec2 {
    float x, y;
};

__device__ __forceinline__ Vec3 mk3(float x, float y, float z) { return Vec3{x, y, z}; }
__device__ __forceinline__ Vec3 add3(Vec3 a, Vec3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ Vec3 sub3(Vec3 a, Vec3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ Vec3 mul3(Vec3 a, Vec3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ Vec3 div3s(Vec3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ Vec3 scale3(float f, Vec3 v) { return mk3(f * v.x, f * v.y, f * v.z); }
__device__ __forceinline__ float dot3(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // vecmath.h:1181
__device__ __forceinline__ Vec3 cross3(Vec3 a, Vec3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float length3(Vec3 v) { return sqrtf(dot3(v, v)); }
__device__ __forceinline__ Vec3 normalize3(Vec3 v) // vecmath.h:1200
{
    float invlen = 1.0f / length3(v);
    return scale3(invlen, v);
}
__device__ __forceinline__ float sse_min(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float sse_max(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float asf(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t asu(float f) { return __float_as_uint(f); }

// Scene and state arrays live in global memory (hipMalloc).  Where a pointer VALUE reaches the code through memory (an
// argument block, a struct passed by reference) the compiler can no longer see that and emits flat loads, which take the
// slower path through the address-space check; these helpers state the address space at the access.
#define PRT_AS1 __attribute__((address_space(1)))
typedef float prt_f4 __attribute__((ext_vector_type(4)));
typedef float prt_f3 __attribute__((ext_vector_type(3)));
typedef prt_f3 prt_f3u __attribute__((aligned(4))); // a 3-vector at a 4-byte-aligned address (sizeof is still 16: index through float*)
typedef uint32_t prt_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t prt_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 gld4(const float4* p)
{
    prt_f4 v = *(const PRT_AS1 prt_f4*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 gld4u(const uint4* p)
{
    prt_u4 v = *(const PRT_AS1 prt_u4*)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 gld2u(const uint2* p)
{
    prt_u2 v = *(const PRT_AS1 prt_u2*)p;
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint32_t gld(const uint32_t* p) { return *(const PRT_AS1 uint32_t*)p; }
__device__ __forceinline__ int32_t gld(const int32_t* p) { return *(const PRT_AS1 int32_t*)p; }
__device__ __forceinline__ float gld(const float* p) { return *(const PRT_AS1 float*)p; }
__device__ __forceinline__ uint32_t gld(const uint8_t* p) { return (uint32_t) * (const PRT_AS1 uint8_t*)p; }
__device__ __forceinline__ void gst(uint32_t* p, uint32_t v) { *(PRT_AS1 uint32_t*)p = v; }
__device__ __forceinline__ void gst4(float4* p, float4 v)
{
    prt_f4 w = {v.x, v.y, v.z, v.w};
    *(PRT_AS1 prt_f4*)p = w;
}
__device__ __forceinline__ void gst2u(uint2* p, uint2 v)
{
    prt_u2 w = {v.x, v.y};
    *(PRT_AS1 prt_u2*)p = w;
}

struct DevRay {
    Vec3 org, dir, inv;
    bool swapXZ, swapYZ;
    bool fast; // all three reciprocal direction components finite: no slab product can be NaN
    // per-ray constants of the watertight test (triangle.cpp:118-119): 1/d.z and (d.x, d.y)/d.z AFTER the axis swap
    float invDz, shearX, shearY;
};

__device__ __forceinline__ void prepare_shear(DevRay& r)
{
    Vec3 d = r.dir;
    if (r.swapXZ) { float t = d.x; d.x = d.z; d.z = t; }
    if (r.swapYZ) { float t = d.y; d.y = d.z; d.z = t; }
    r.invDz = 1.0f / d.z;
    r.shearX = d.x * r.invDz;
    r.shearY = d.y * r.invDz;
    const float big = 3.402823466e+38f;
    r.fast = (fabsf(r.inv.x) <= big) && (fabsf(r.inv.y) <= big) && (fabsf(r.inv.z) <= big) && (fabsf(r.org.x) <= big) &&
             (fabsf(r.org.y) <= big) && (fabsf(r.org.z) <= big);
}

struct DevHit {
    float t, i, j, k;
    uint32_t primId, meshId;
};

struct Traffic {
    unsigned long long nBox, nTri; // a persistent trace lane can exceed 2^32
    uint32_t nHit, nTap;
#ifdef PRT_PROFILE
    // tallies of the profile build: rounds of either kind and the lanes that took part in them (wave-uniform); stack pops
    // and those that came from the spill area in HBM (per lane)
    unsigned long long pNodeRounds, pNodeLanes, pLeafRounds, pLeafLanes, pTri2Lanes, pPops, pDeepPops;
    unsigned long long pNodeWaitLeaf, pNodeDone, pNodeNoRay, pLeafWaitNode, pLeafDone, pLeafNoRay, pLeafUpdates; // lanes that sit a round out, by reason (wave-uniform)
    unsigned long long pDirectInt, pPopInt; // (per lane) steps that leave the lane on an internal record outside the hot set: reached by descending from its parent / by a pop
    unsigned long long pAlphaRounds, pAlphaCycles; // (wave-uniform) cooperative leaf rounds in which a candidate looked its alpha record up, and the cycles from the candidate test to the end of the alpha tests
    unsigned long long pNodeDistinct; // distinct records the lanes of the node rounds asked for (wave-uniform): the coherence of a wave's rays
#endif
};

// ray.h:26-40: swap axis from the SIGNED components (Vector3f::GetLongestElement, vecmath.h:216)
__device__ __forceinline__ void prepare_single(DevRay& r)
{
    r.inv = mk3(1.0f / r.dir.x, 1.0f / r.dir.y, 1.0f / r.dir.z);
    uint32_t e;
    if (r.dir.x > r.dir.y) e = (r.dir.x > r.dir.z) ? 0u : 2u;
    else e = (r.dir.y > r.dir.z) ? 1u : 2u;
    r.swapXZ = (e == 0u);
    r.swapYZ = (e == 1u);
    prepare_shear(r);
}

// ray.h:58-71: largest |component|, x wins ties, then y
__device__ __forceinline__ void prepare_soa(DevRay& r)
{
    r.inv = mk3(1.0f / r.dir.x, 1.0f / r.dir.y, 1.0f / r.dir.z);
    float ax = fabsf(r.dir.x), ay = fabsf(r.dir.y), az = fabsf(r.dir.z);
    float max_e = sse_max(ax, sse_max(ay, az));
    bool mx = (max_e == ax);
    r.swapXZ = mx;
    r.swapYZ = (max_e == ay) && !mx;
    prepare_shear(r);
}

// ---------------------------------------------------------------------------- box tests
struct Box {
    Vec3 lo, hi;
};

// SSE min/max return their SECOND operand when either is NaN; a slab product is NaN only when a box plane passes
// through the ray origin AND that direction component is exactly zero (0 * inf).  A ray whose three reciprocal
// components are finite (DevRay::fast, the overwhelmingly common case) can never produce one, and for NaN-free inputs
// v_min_f32/v_max_f32 give the same values as the SSE selects up to the sign of a zero, which only ever feeds
// comparisons.  FAST = false keeps the exact select form for the remaining rays.
template <bool FAST>
__device__ __forceinline__ float bmin(float a, float b) { return FAST ? __builtin_fminf(a, b) : (a < b ? a : b); }
template <bool FAST>
__device__ __forceinline__ float bmax(float a, float b) { return FAST ? __builtin_fmaxf(a, b) : (a > b ? a : b); }

template <bool FAST>
__device__ __forceinline__ void slabs(const Box& b, const DevRay& r, float t0[3], float t1[3])
{
    float a, c;
    a = (b.lo.x - r.org.x) * r.inv.x; c = (b.hi.x - r.org.x) * r.inv.x; t0[0] = bmin<FAST>(a, c); t1[0] = bmax<FAST>(a, c);
    a = (b.lo.y - r.org.y) * r.inv.y; c = (b.hi.y - r.org.y) * r.inv.y; t0[1] = bmin<FAST>(a, c); t1[1] = bmax<FAST>(a, c);
    a = (b.lo.z - r.org.z) * r.inv.z; c = (b.hi.z - r.org.z) * r.inv.z; t0[2] = bmin<FAST>(a, c); t1[2] = bmax<FAST>(a, c);
}

// vecmath.h:1402-1424.  4th SSE lane: t0 = -inf, t1 = +inf; reductions op(op(a0,a2),op(a1,a3)) (:1325-1345)
template <bool FAST>
__device__ __forceinline__ float box_t_impl(const Box& b, const DevRay& r)
{
    float t0[3], t1[3];
    slabs<FAST>(b, r, t0, t1);
    const float inf = __builtin_inff();
    float max_t0 = bmax<FAST>(bmax<FAST>(t0[0], t0[2]), bmax<FAST>(t0[1], -inf));
    float min_t1 = bmin<FAST>(bmin<FAST>(t1[0], t1[2]), bmin<FAST>(t1[1], inf));
    return (min_t1 < max_t0) ? inf : max_t0;
}
__device__ __forceinline__ float box_t(const Box& b, const DevRay& r) { return r.fast ? box_t_impl<true>(b, r) : box_t_impl<false>(b, r); }

// vecmath.h:1449-1466.  4th lane: t0 = min(-FLT_MAX, maxT), t1 = max(-FLT_MAX, maxT)
template <bool FAST>
__device__ __forceinline__ bool box_bool_impl(const Box& b, const DevRay& r, float maxT)
{
    float t0[3], t1[3];
    slabs<FAST>(b, r, t0, t1);
    const float lowest = -3.402823466e+38f;
    float w0 = sse_min(lowest, maxT), w1 = sse_max(lowest, maxT);
    float max_t0 = bmax<FAST>(bmax<FAST>(t0[0], t0[2]), bmax<FAST>(t0[1], w0));
    float min_t1 = bmin<FAST>(bmin<FAST>(t1[0], t1[2]), bmin<FAST>(t1[1], w1));
    return min_t1 > max_t0;
}
__device__ __forceinline__ bool box_bool(const Box& b, const DevRay& r, float maxT)
{
    return r.fast ? box_bool_impl<true>(b, r, maxT) : box_bool_impl<false>(b, r, maxT);
}

// vecmath.h:1504-1518, one lane: returns max_t0 when min_t1 >= max_t0, else +inf, so that `entry < limit` is the whole
// test (max_t0 < maxT && min_t1 >= max_t0)
template <bool FAST>
__device__ __forceinline__ float box_soa_entry_impl(const Box& b, const DevRay& r)
{
    float t0[3], t1[3];
    slabs<FAST>(b, r, t0, t1);
    float max_t0 = bmax<FAST>(t0[0], bmax<FAST>(t0[1], t0[2]));
    float min_t1 = bmin<FAST>(t1[0], bmin<FAST>(t1[1], t1[2]));
    return (min_t1 >= max_t0) ? max_t0 : __builtin_inff();
}
__device__ __forceinline__ float box_soa_entry(const Box& b, const DevRay& r)
{
    return r.fast ? box_soa_entry_impl<true>(b, r) : box_soa_entry_impl<false>(b, r);
}
__device__ __forceinline__ bool box_soa(const Box& b, const DevRay& r, float maxT) { return box_soa_entry(b, r) < maxT; }

// Ordering key of the any-hit traversal for rays off the NaN-free path: the largest slab entry, written with selects so
// that the test suite's CPU accounting of this visit evaluates the same expression.  For NaN-free rays it
// equals the max_t0 of the packed slab arithmetic.
__device__ __forceinline__ float slab_entry_select(const Box& b, const DevRay& r)
{
    float a, c, m, v;
    a = (b.lo.x - r.org.x) * r.inv.x; c = (b.hi.x - r.org.x) * r.inv.x; m = (a < c) ? a : c;
    a = (b.lo.y - r.org.y) * r.inv.y; c = (b.hi.y - r.org.y) * r.inv.y; v = (a < c) ? a : c; m = (v > m) ? v : m;
    a = (b.lo.z - r.org.z) * r.inv.z; c = (b.hi.z - r.org.z) * r.inv.z; v = (a < c) ? a : c; m = (v > m) ? v : m;
    return m;
}

// ---------------------------------------------------------------------------- triangle
// triangle.cpp:90-166, one lane.  Returns t, or -1 (kNoIntersection).
__device__ __forceinline__ float tri_intersect(const DevRay& r, Vec3 p0, Vec3 p1, Vec3 p2, float& bi, float& bj, float& bk)
{
    Vec3 v0 = sub3(p0, r.org), v1 = sub3(p1, r.org), v2 = sub3(p2, r.org);
    if (r.swapXZ) {
        float t;
        t = v0.x; v0.x = v0.z; v0.z = t;
        t = v1.x; v1.x = v1.z; v1.z = t;
        t = v2.x; v2.x = v2.z; v2.z = t;
    }
    if (r.swapYZ) {
        float t;
        t = v0.y; v0.y = v0.z; v0.z = t;
        t = v1.y; v1.y = v1.z; v1.z = t;
        t = v2.y; v2.y = v2.z; v2.z = t;
    }
    float v0z = v0.z, v1z = v1.z, v2z = v2.z;
    const float inv_dz = r.invDz, idx = r.shearX, idy = r.shearY; // invDzD = inv_dz*d, :119
    float v0x = v0.x - idx * v0z, v0y = v0.y - idy * v0z;
    float v1x = v1.x - idx * v1z, v1y = v1.y - idy * v1z;
    float v2x = v2.x - idx * v2z, v2y = v2.y - idy * v2z;
    float e0 = v1x * v2y - v1y * v2x;
    float e1 = v2x * v0y - v2y * v0x;
    float e2 = v0x * v1y - v0y * v1x;
    bool mask_eb = (e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f);
    bi = bj = bk = 0.0f;
    if (mask_eb) return -1.0f;
    float det = e0 + e1 + e2;
    bool mask = (det < 0.0f) || (det > 0.0f); // _CMP_NEQ_OQ
    float inv_det = 1.0f / det;
    float t_scaled = (e0 * v0z + e1 * v1z + e2 * v2z) * inv_dz;
    float t = t_scaled * inv_det;
    mask = mask && (t > 0.0001f);
    bi = e0 * inv_det;
    bj = e1 * inv_det;
    bk = e2 * inv_det;
    return mask ? t : -1.0f;
}

// ---------------------------------------------------------------------------- textures (texture.cpp:31-183)
// The cell of the bilinear tap at uv: (x0, y0) as bilinear() below computes them and the in-cell coordinates xt, yt.  False when
// uv is not finite (then the blend's weights are NaN in one of the reference's two flavours and the caller takes the full path).
__device__ __forceinline__ bool bilinear_cell(Vec2 uv, int32_t width, int32_t height, bool soa, int32_t& x0, int32_t& y0, float& xt, float& yt)
{
    float s = uv.x - floorf(uv.x);
    float t = uv.y - floorf(uv.y);
    if (soa) {
        xt = sse_max(s * (float)width - 0.5f, 0.0f);
        yt = sse_max(t * (float)height - 0.5f, 0.0f);
    } else {
        xt = std_max(s * (float)width - 0.5f, 0.0f);
        yt = std_max(t * (float)height - 0.5f, 0.0f);
    }
    x0 = (int32_t)floorf(xt);
    y0 = (int32_t)floorf(yt);
    return s == s && t == t;
}

__device__ __forceinline__ void bilinear(float k[4], int32_t idx[4], int32_t component, Vec2 uv, int32_t width, int32_t height, bool soa)
{
    float xt, yt;
    int32_t x0, y0;
    (void)bilinear_cell(uv, width, height, soa, x0, y0, xt, yt);
    int32_t x1 = (x0 + 1 < width - 1) ? x0 + 1 : width - 1;
    int32_t y1 = (y0 + 1 < height - 1) ? y0 + 1 : height - 1;
    idx[0] = component * (x0 + y0 * width);
    idx[1] = component * (x1 + y0 * width);
    idx[2] = component * (x0 + y1 * width);
    idx[3] = component * (x1 + y1 * width);
    const float s = xt - (float)x0;
    const float t = yt - (float)y0;
    k[0] = (1.0f - s) * (1.0f - t);
    k[1] = s * (1.0f - t);
    k[2] = (1.0f - s) * t;
    k[3] = s * t;
}

template <bool COUNT>
__device__ __forceinline__ bool tex_test_alpha(const DevScene& sc, uint4 d, Vec2 uv, bool soa, Traffic& tr)
{
    const uint8_t* p = sc.texels + d.x;
    float k[4];
    int32_t idx[4];
    bilinear(k, idx, (int32_t)d.w, uv, (int32_t)d.y, (int32_t)d.z, soa);
    if (COUNT) tr.nTap++;
    float alpha = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) alpha = alpha + k[i] * (float)gld(p + idx[i] + 3);
    return alpha > 127.0f;
}

template <bool COUNT>
__device__ __forceinline__ Vec3 tex_sample3(const DevScene& sc, uint4 d, Vec2 uv, Traffic& tr)
{
    const uint8_t* p = sc.texels + d.x;
    float k[4];
    int32_t idx[4];
    bilinear(k, idx, (int32_t)d.w, uv, (int32_t)d.y, (int32_t)d.z, false);
    if (COUNT) tr.nTap++;
    Vec3 c = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint8_t* q = p + idx[i];
        c = add3(c, scale3(k[i], mk3((float)gld(q), (float)gld(q + 1), (float)gld(q + 2))));
    }
    return scale3(1.0f / 255.0f, c);
}

template <bool COUNT>
__device__ __forceinline__ float tex_sample1(const DevScene& sc, uint4 d, Vec2 uv, Traffic& tr)
{
    const uint8_t* p = sc.texels + d.x;
    float k[4];
    int32_t idx[4];
    bilinear(k, idx, (int32_t)d.w, uv, (int32_t)d.y, (int32_t)d.z, false);
    if (COUNT) tr.nTap++;
    float c = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) c = c + k[i] * (float)gld(p + idx[i]);
    return (1.0f / 255.0f) * c;
}

// ---------------------------------------------------------------------------- traversal
// Child references.  A traversal never loads a node to learn what it is: the PARENT's record holds both children's
// boxes and references (one 64-byte fetch per internal node visited, nothing fetched at a pop).
//   internal: index of the child's own wide record (< 2^30)
//   leaf:     PRT_REF_LEAF | firstTriangle << 4 | hasAlpha << 3 | primCount - 1      (primCount 1..8, firstTriangle < 2^26;
//             hasAlpha: some triangle of the leaf is alpha-tested)
#define PRT_REF_LEAF 0x80000000u
#define PRT_LEAF_ALPHA 8u
__device__ __forceinline__ uint32_t leaf_count(uint32_t ref) { return (ref & 7u) + 1u; }
__device__ __forceinline__ uint32_t leaf_first(uint32_t ref) { return (ref >> 4) & 0x7ffffffu; }
// the three corners of leaf-order triangle `tri` (three 12-byte loads)
__device__ __forceinline__ void load_tri(const DevScene& sc, uint32_t tri, Vec3& p0, Vec3& p1, Vec3& p2)
{
    const float* tp = sc.tris + 9 * (size_t)tri;
    const prt_f3 a = *(const PRT_AS1 prt_f3u*)tp, b = *(const PRT_AS1 prt_f3u*)(tp + 3), c = *(const PRT_AS1 prt_f3u*)(tp + 6);
    p0 = mk3(a.x, a.y, a.z);
    p1 = mk3(b.x, b.y, b.z);
    p2 = mk3(c.x, c.y, c.z);
}
// The records nearest to the roots (breadth-first, PRT_HOT_NODES of them) are visited by every ray.  References to them
// carry PRT_REF_HOT | slot: a kernel that keeps a copy of those records in LDS (the frame kernel) reads them there with
// ds_read_b128 instead of sending four more gathers down the texture-address path, the busiest unit of these kernels
// (profiles/: TA busy 76 % of the frame); other kernels read the same copy from DevScene::hotNodes.
#define PRT_REF_HOT 0x40000000u
#ifndef PRT_HOT_NODES
#define PRT_HOT_NODES 256 // a power of two; 16 KB of LDS per workgroup, shared by its 16 waves (64: 381-386 ms, 128: 378-381, 256: 368-371 on C3)
#endif

// Per-lane stack: entries 0..PRT_STACK_LDS-1 in LDS ([entry][thread]: conflict-free whatever the depths), deeper
// ones in a per-thread global spill area.  `t` (the box entry distance) is only stored by the packet traversal.
// LDS-typed pointers: through a generic pointer the pop `lds[e] or spill[e]` compiles to one flat load, which goes down the
// texture path the node and triangle fetches already saturate
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) prt_f4 lds_f4;
template <int NLDS> // stack entries kept in LDS; deeper ones spill
struct StackT {
    static constexpr int kLds = NLDS;
    lds_u32* ldsRef; // &refs[threadIdx.x]
    lds_f32* ldsT;   // &ts[threadIdx.x]
    // the spill area: [entry][thread of the launch], two words per entry.  Only its (uniform) base is kept; the thread's column
    // is recomputed where a deep entry is touched -- a per-thread 64-bit pointer held across the step loop was the one value the
    // register allocator kept in scratch memory there (reloaded on this cold path, profiles/r03_step_loop_isa.txt)
    uint32_t* spillBase;
    uint32_t spillStride;
    const lds_f4* hot; // the block's LDS copy of DevScene::hotNodes, or NULL
    lds_u32* coop;     // the WAVE's 64-word pair table of the cooperative leaf rounds (tracer_leaf_coop)
    __device__ __forceinline__ uint32_t* deep(int e, int word) const
    {
        // the thread's index, from the LDS address of its stack column -- a value the loop keeps in a register anyway.  The asm is
        // opaque to the optimiser, which otherwise hoists "block * 256 + thread" out of the step loop and parks it in scratch
        uint32_t thread;
        asm volatile("v_lshrrev_b32 %0, 2, %1" : "=v"(thread) : "v"((uint32_t)(uintptr_t)ldsRef - (uint32_t)(uintptr_t)(ldsRef - threadIdx.x)));
        return spillBase + ((size_t)(2 * (e - NLDS) + word) * spillStride + (blockIdx.x * PRT_BLOCK + thread));
    }
    __device__ __forceinline__ void put(int e, uint32_t ref) const
    {
        if (e < NLDS) ldsRef[e * PRT_BLOCK] = ref;
        else gst(deep(e, 0), ref);
    }
    __device__ __forceinline__ uint32_t get(int e) const
    {
        uint32_t v = ldsRef[(e < NLDS ? e : 0) * PRT_BLOCK];
        if (e >= NLDS) v = gld(deep(e, 0));
        return v;
    }
    __device__ __forceinline__ void putT(int e, uint32_t ref, float t) const
    {
        if (e < NLDS) {
            ldsRef[e * PRT_BLOCK] = ref;
            ldsT[e * PRT_BLOCK] = t;
        } else {
            gst(deep(e, 0), ref);
            gst(deep(e, 1), asu(t));
        }
    }
    __device__ __forceinline__ float getT(int e) const
    {
        float v = ldsT[(e < NLDS ? e : 0) * PRT_BLOCK];
        if (e >= NLDS) v = asf(gld(deep(e, 1)));
        return v;
    }
};

typedef float f2_t __attribute__((ext_vector_type(2)));

struct WideNode {
    float4 w0, w1, w2; // the three (lo, hi) pair records as stored
    uint32_t ref0, ref1, axis;
    __device__ __forceinline__ Box box0() const { return Box{mk3(w0.x, w0.z, w1.x), mk3(w0.y, w0.w, w1.y)}; }
    __device__ __forceinline__ Box box1() const { return Box{mk3(w2.x, w2.z, w1.z), mk3(w2.y, w2.w, w1.w)}; }
};

template <class STK>
__device__ __forceinline__ void load_wide(const DevScene& sc, const STK& st, uint32_t idx, WideNode& w)
{
    float4 w3;
    if (st.hot != nullptr && (idx & PRT_REF_HOT)) {
        const lds_f4* h = st.hot + 4u * (idx & (PRT_HOT_NODES - 1u));
        const prt_f4 a = h[0], b = h[1], c = h[2], d = h[3];
        w.w0 = make_float4(a.x, a.y, a.z, a.w);
        w.w1 = make_float4(b.x, b.y, b.z, b.w);
        w.w2 = make_float4(c.x, c.y, c.z, c.w);
        w3 = make_float4(d.x, d.y, d.z, d.w);
    } else {
        const float4* p = (idx & PRT_REF_HOT) ? sc.hotNodes + 4u * (idx & (PRT_HOT_NODES - 1u)) : sc.wnodes + 4 * (size_t)idx;
        w.w0 = gld4(p);
        w.w1 = gld4(p + 1);
        w.w2 = gld4(p + 2);
        w3 = gld4(p + 3);
    }
    w.ref0 = asu(w3.x);
    w.ref1 = asu(w3.y);
    w.axis = asu(w3.z);
}

__device__ __forceinline__ Box root_box(const DevScene& sc, uint32_t m)
{
    return Box{mk3(sc.rootBox[m][0], sc.rootBox[m][1], sc.rootBox[m][2]), mk3(sc.rootBox[m][3], sc.rootBox[m][4], sc.rootBox[m][5])};
}

// One triangle of a leaf against one ray (the loop bodies of bvh.cpp:302-368 and :370-427 for one lane): true when the
// candidate is accepted -- t in [kTriEpsilon, limit) and the alpha test, if the triangle has one, passed.  Nearest-hit
// callers pass limit = hit.t and get hit updated; occlusion callers pass limit = maxT.
template <bool OCCLUDE, bool PACKET, bool COUNT>
__device__ __forceinline__ bool tri_candidate(const DevScene& sc, uint32_t tri, bool leafHasAlpha, uint32_t meshId, const DevRay& r, float limit, DevHit& hit,
                                              Traffic& tr)
{
    Vec3 p0, p1, p2;
    load_tri(sc, tri, p0, p1, p2);
    float bi, bj, bk;
    float t = tri_intersect(r, p0, p1, p2, bi, bj, bk);
    if (!(t >= 0.0001f && t < limit)) return false;
    const uint32_t alphaRef = leafHasAlpha ? gld(sc.triAlpha + tri) : 0u;
    if (alphaRef) {
        const float4* ap = sc.alpha + 3 * (size_t)(alphaRef - 1);
        float4 u0 = gld4(ap), u1 = gld4(ap + 1), u2 = gld4(ap + 2);
        // i*uv0 + j*uv1 + k*uv2 (bvh.cpp:336, 407)
        Vec2 uv = Vec2{bi * u0.x + bj * u0.z + bk * u1.x, bi * u0.y + bj * u0.w + bk * u1.y};
        if (!tex_test_alpha<COUNT>(sc, make_uint4(asu(u2.x), asu(u2.y), asu(u2.z), asu(u2.w)), uv, PACKET, tr)) return false;
    }
    if (!OCCLUDE) {
        hit.t = t;
        hit.i = bi;
        hit.j = bj;
        hit.k = bk;
        hit.primId = tri; // the LEAF-ORDER index: shade and bump records are stored in that order
        hit.meshId = meshId;
    }
    return true;
}

// ---------------------------------------------------------------------------- the four traversals as ONE stepper
// MODE 0  Scene::intersect<RayHitPacket,RayPacket> for one lane of a packet (scene.cpp:47-63, bvh.cpp:429-570 packet branch)
// MODE 1  Scene::intersect<SingleRayHitPacket,SingleRayPacket>               (bvh.cpp:429-570 single branch, SORT_CHILDREN)
// MODE 2  Scene::occluded<RayPacketMask,RayPacket> for one lane              (scene.cpp:69-94, bvh.cpp:576-654)
// MODE 3  Scene::occluded<bool,SingleRayPacket>
//
// What the reference's semantics fix, per mode:
//  0: a node's box is tested when it is POPPED, against the hit.t of that moment (max_t0 < hit.t && min_t1 >= max_t0,
//     vecmath.h:1504-1518); child order from sign(avgDir[splitAxis]) (bvh.cpp:523-529).  The reference walks 8 rays
//     with a lane mask; every mask/hit update is lane-wise and the order is shared, so a lane walking alone over the
//     nodes where its own mask bit is set sees the same nodes in the same order with the same hit.t.  Here the slab
//     distances are computed when the PARENT is visited (its record holds the child boxes), max_t0 travels on the
//     stack and the hit.t part of the test is redone at the pop -- the same decision.
//  1: root bool test (vecmath.h:1449); both children tested per internal node against the current hit.t
//     (vecmath.h:1402), near child first (t0 < t1, ties -> second child first), popped nodes NOT re-tested (bvh.cpp:474).
//  2/3: any-hit.  maxT is constant, so every box and triangle test is a function of (ray, box or triangle) alone and the
//     answer -- is ANY triangle below boxes the ray passes accepted -- does not depend on the order of the visit.  The
//     reference pops the first child first (bvh.cpp:617-620); here the child whose box the ray enters first goes first
//     (ties: child 0), which finds an occluder after about half the nodes and triangles on the C3 scene.  The counters
//     count the tests this traversal performs (2 per internal node visited, 1 per root, every triangle tested).
#define PRT_MODE_PACKET 0
#define PRT_MODE_SINGLE 1
#define PRT_MODE_OCC_PACKET 2
#define PRT_MODE_OCC_SINGLE 3
#define PRT_REF_NONE 0xfffffffeu // no current node: start the next BVH or finish (never a valid leaf reference: count <= 8)

struct Tracer {
    DevRay r;
    float maxT;
    DevHit hit;
    uint32_t ref;
    int sp;
    uint32_t m;   // current BVH; 0xffffffff before the first
    uint32_t rev; // packet mode: sign bits of avgDir
    bool occ;
};

template <int MODE>
__device__ __forceinline__ void tracer_begin(Tracer& T, Vec3 org, Vec3 dir, float maxT, uint32_t rev)
{
    T.r.org = org;
    T.r.dir = dir;
    if (MODE == PRT_MODE_PACKET || MODE == PRT_MODE_OCC_PACKET) prepare_soa(T.r);
    else prepare_single(T.r);
    T.maxT = maxT;
    T.hit.t = maxT; // Scene::intersect: memset + setMaxT (scene.cpp:50-53)
    T.hit.i = T.hit.j = T.hit.k = 0.0f;
    T.hit.primId = 0;
    T.hit.meshId = 0;
    T.ref = PRT_REF_NONE;
    T.sp = 0;
    T.m = 0xffffffffu;
    T.rev = rev;
    T.occ = false;
}

template <int MODE, bool COUNT, class STK>
__device__ __forceinline__ uint32_t tracer_pop(Tracer& T, const STK& st, Traffic& tr)
{
    if (MODE == PRT_MODE_PACKET) {
        while (T.sp > 0) {
            --T.sp;
            if (st.getT(T.sp) < T.hit.t) return st.get(T.sp); // the pop-time test of the reference
        }
        return PRT_REF_NONE;
    } else {
        if (T.sp == 0) return PRT_REF_NONE;
#ifdef PRT_PROFILE
        tr.pPops++;
        if (T.sp - 1 >= STK::kLds) tr.pDeepPops++;
#endif
        return st.get(--T.sp);
    }
}

// Advance to the next BVH whose root box the ray passes; false when there is none left (the ray is finished).
template <int MODE, bool COUNT>
__device__ __forceinline__ bool tracer_next_bvh(const DevScene& sc, Tracer& T, Traffic& tr)
{
    for (;;) {
        T.m++;
        if (T.m >= sc.bvhCount) return false;
        if (COUNT) tr.nBox++;
        // The lanes that come here together nearly always stand before the same BVH (new rays: the first): then the root's box and
        // reference are uniform and arrive by scalar loads, not by two rounds of per-lane gathers from the argument block.
        const uint32_t mu = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.m);
        Box rb;
        uint32_t rootRef;
        if (__ballot(T.m != mu) == 0ull) {
            rb = root_box(sc, mu);
            rootRef = sc.rootRef[mu];
        } else {
            rb = root_box(sc, T.m);
            rootRef = sc.rootRef[T.m];
        }
        bool pass;
        if (MODE == PRT_MODE_PACKET) pass = box_soa(rb, T.r, T.hit.t);
        else if (MODE == PRT_MODE_SINGLE) pass = box_bool(rb, T.r, T.hit.t);
        else if (MODE == PRT_MODE_OCC_PACKET) pass = box_soa(rb, T.r, T.maxT);
        else pass = box_bool(rb, T.r, T.maxT);
        if (pass) {
            T.ref = rootRef;
            T.sp = 0;
            return true;
        }
    }
}

// Any-hit modes: both children passed -> the nearer first, the other stacked.
template <int MODE, bool COUNT, class STK>
__device__ __forceinline__ void tracer_any_hit_order(Tracer& T, const STK& st, Traffic& tr, const WideNode& w, bool h0, bool h1, bool near0)
{
    if (h0 && h1) {
        st.put(T.sp++, near0 ? w.ref1 : w.ref0);
        T.ref = near0 ? w.ref0 : w.ref1;
    } else if (h0) {
        T.ref = w.ref0;
    } else if (h1) {
        T.ref = w.ref1;
    } else {
        T.ref = tracer_pop<MODE, COUNT>(T, st, tr);
    }
}

// The per-mode decision at an internal node for a NaN-free ray, from the slab extents of both children (max_t0, min_t1):
// every box test of the reference is a function of those two numbers when no NaN is involved.
struct NodeExt {
    float mx0, mn0, mx1, mn1;
};

template <int MODE, bool COUNT, class STK>
__device__ __forceinline__ void tracer_decide(const DevScene& sc, Tracer& T, const STK& st, Traffic& tr, const WideNode& w, const NodeExt& e)
{
    const float inf = __builtin_inff();
    if (MODE == PRT_MODE_SINGLE) {
        if (COUNT) tr.nBox += 2;
        float t0 = (e.mn0 < e.mx0) ? inf : e.mx0, t1 = (e.mn1 < e.mx1) ? inf : e.mx1; // vecmath.h:1420-1424
        bool h0 = t0 < T.hit.t, h1 = t1 < T.hit.t;
        if (h0 && h1) {
            bool near0 = t0 < t1;
            st.put(T.sp++, near0 ? w.ref1 : w.ref0);
            T.ref = near0 ? w.ref0 : w.ref1;
        } else if (h0) {
            T.ref = w.ref0;
        } else if (h1) {
            T.ref = w.ref1;
        } else {
            T.ref = tracer_pop<MODE, COUNT>(T, st, tr);
        }
    } else if (MODE == PRT_MODE_PACKET) {
        if (COUNT) tr.nBox += 2;
        float e0 = (e.mn0 >= e.mx0) ? e.mx0 : inf, e1 = (e.mn1 >= e.mx1) ? e.mx1 : inf; // vecmath.h:1515-1517
        bool rev = (T.rev >> (w.axis & 3u)) & 1u;
        uint32_t firstRef = rev ? w.ref1 : w.ref0, laterRef = rev ? w.ref0 : w.ref1;
        float firstE = rev ? e1 : e0, laterE = rev ? e0 : e1;
        if (laterE < T.hit.t) st.putT(T.sp++, laterRef, laterE);
        if (firstE < T.hit.t) T.ref = firstRef;
        else T.ref = tracer_pop<MODE, COUNT>(T, st, tr);
    } else {
        bool h0, h1;
        if (MODE == PRT_MODE_OCC_PACKET) {
            h0 = ((e.mn0 >= e.mx0) ? e.mx0 : inf) < T.maxT;
            h1 = ((e.mn1 >= e.mx1) ? e.mx1 : inf) < T.maxT;
        } else { // vecmath.h:1449-1466: the 4th SSE lane carries (-FLT_MAX, maxT)
            const float lowest = -3.402823466e+38f;
            float w0 = sse_min(lowest, T.maxT), w1 = sse_max(lowest, T.maxT);
            h0 = __builtin_fminf(e.mn0, w1) > __builtin_fmaxf(e.mx0, w0);
            h1 = __builtin_fminf(e.mn1, w1) > __builtin_fmaxf(e.mx1, w0);
        }
        if (COUNT) tr.nBox += 2;
        tracer_any_hit_order<MODE, COUNT>(T, st, tr, w, h0, h1, e.mx0 <= e.mx1);
    }
}

// One internal node: fetch its 64-byte record, test both children, choose where to go.
template <int MODE, bool COUNT, class STK>
__device__ __forceinline__ void tracer_node(const DevScene& sc, Tracer& T, const STK& st, Traffic& tr, uint32_t& overflow)
{
    WideNode w;
    load_wide(sc, st, T.ref, w);
    if (T.sp + 2 >= PRT_STACK_MAX) { // bvh.cpp:552, 627
        overflow = 1;
        T.ref = PRT_REF_NONE;
        T.sp = 0;
        T.m = sc.bvhCount;
        return;
    }
    if (T.r.fast) { // one straight-line block: six packed (lo, hi) slab pairs, two min3/max3 pairs
        const DevRay& r = T.r;
        const f2_t c0x = (f2_t{w.w0.x, w.w0.y} - r.org.x) * r.inv.x, c0y = (f2_t{w.w0.z, w.w0.w} - r.org.y) * r.inv.y;
        const f2_t c0z = (f2_t{w.w1.x, w.w1.y} - r.org.z) * r.inv.z, c1z = (f2_t{w.w1.z, w.w1.w} - r.org.z) * r.inv.z;
        const f2_t c1x = (f2_t{w.w2.x, w.w2.y} - r.org.x) * r.inv.x, c1y = (f2_t{w.w2.z, w.w2.w} - r.org.y) * r.inv.y;
        NodeExt e;
        e.mx0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(c0x.x, c0x.y), __builtin_fminf(c0y.x, c0y.y)), __builtin_fminf(c0z.x, c0z.y));
        e.mn0 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(c0x.x, c0x.y), __builtin_fmaxf(c0y.x, c0y.y)), __builtin_fmaxf(c0z.x, c0z.y));
        e.mx1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(c1x.x, c1x.y), __builtin_fminf(c1y.x, c1y.y)), __builtin_fminf(c1z.x, c1z.y));
        e.mn1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(c1x.x, c1x.y), __builtin_fmaxf(c1y.x, c1y.y)), __builtin_fmaxf(c1z.x, c1z.y));
        tracer_decide<MODE, COUNT>(sc, T, st, tr, w, e);
        return;
    }
    const Box wb0 = w.box0(), wb1 = w.box1();
    if (MODE == PRT_MODE_SINGLE) {
        if (COUNT) tr.nBox += 2;
        float t0 = box_t(wb0, T.r), t1 = box_t(wb1, T.r);
        bool h0 = t0 < T.hit.t, h1 = t1 < T.hit.t;
        if (h0 && h1) {
            bool near0 = t0 < t1;
            st.put(T.sp++, near0 ? w.ref1 : w.ref0);
            T.ref = near0 ? w.ref0 : w.ref1;
        } else if (h0) {
            T.ref = w.ref0;
        } else if (h1) {
            T.ref = w.ref1;
        } else {
            T.ref = tracer_pop<MODE, COUNT>(T, st, tr);
        }
    } else if (MODE == PRT_MODE_PACKET) {
        if (COUNT) tr.nBox += 2; // the reference pops and tests both children
        float e0 = box_soa_entry(wb0, T.r), e1 = box_soa_entry(wb1, T.r);
        bool rev = (T.rev >> (w.axis & 3u)) & 1u; // popped first = the second child when reverse
        uint32_t firstRef = rev ? w.ref1 : w.ref0, laterRef = rev ? w.ref0 : w.ref1;
        float firstE = rev ? e1 : e0, laterE = rev ? e0 : e1;
        if (laterE < T.hit.t) st.putT(T.sp++, laterRef, laterE); // hit.t only shrinks: failing now means failing at the pop
        if (firstE < T.hit.t) T.ref = firstRef;
        else T.ref = tracer_pop<MODE, COUNT>(T, st, tr);
    } else {
        bool h0 = (MODE == PRT_MODE_OCC_PACKET) ? box_soa(wb0, T.r, T.maxT) : box_bool(wb0, T.r, T.maxT);
        bool h1 = (MODE == PRT_MODE_OCC_PACKET) ? box_soa(wb1, T.r, T.maxT) : box_bool(wb1, T.r, T.maxT);
        if (COUNT) tr.nBox += 2;
        tracer_any_hit_order<MODE, COUNT>(T, st, tr, w, h0, h1, slab_entry_select(wb0, T.r) <= slab_entry_select(wb1, T.r));
    }
}

// One triangle of the leaf the lane stands on.  A leaf reference is its own cursor: (first triangle, triangles left), so
// the step leaves the lane on the rest of the leaf or pops.  Stepping triangle by triangle (not leaf by leaf) lets the
// wave loop regroup lanes after every triangle: a leaf of 8 no longer holds up lanes whose leaf had 1.
template <int MODE, bool COUNT, class STK>
__device__ __forceinline__ void tracer_tri(const DevScene& sc, Tracer& T, const STK& st, Traffic& tr)
{
    constexpr bool OCC = (MODE == PRT_MODE_OCC_PACKET || MODE == PRT_MODE_OCC_SINGLE);
    constexpr bool PACKET = (MODE == PRT_MODE_PACKET || MODE == PRT_MODE_OCC_PACKET);
    const uint32_t left = leaf_count(T.ref), tri = leaf_first(T.ref);
    const bool hasAlpha = (T.ref & PRT_LEAF_ALPHA) != 0u;
    {
        if (COUNT) tr.nTri++;
        if (tri_candidate<OCC, PACKET, COUNT>(sc, tri, hasAlpha, T.m, T.r, OCC ? T.maxT : T.hit.t, T.hit, tr) && OCC) {
            T.occ = true;
            T.ref = PRT_REF_NONE;
            T.sp = 0;
            T.m = sc.bvhCount; // finished
            return;
        }
    }
#if PRT_TRI2
    if (left > 1u) {
        if (COUNT) tr.nTri++;
        if (tri_candidate<OCC, PACKET, COUNT>(sc, tri + 1u, hasAlpha, T.m, T.r, OCC ? T.maxT : T.hit.t, T.hit, tr) && OCC) {
            T.occ = true;
            T.ref = PRT_REF_NONE;
            T.sp = 0;
            T.m = sc.bvhCount;
            return;
        }
    }
    if (left > 2u) T.ref += 30u; // first triangle + 2 (bits 4..), triangles left - 2 (bits 0..2; no borrow: left - 1 >= 2)
    else T.ref = tracer_pop<MODE, COUNT>(T, st, tr);
#else
    if (left > 1u) T.ref += 15u; // first triangle + 1 (bit 4), triangles left - 1
    else T.ref = tracer_pop<MODE, COUNT>(T, st, tr);
#endif
}

// ---------------------------------------------------------------------------- cooperative leaf rounds
// The reference's leaf shape for one ray is "1 ray x 8 triangles" (intersectSingleRay, bvh.cpp:302-368): all triangles of the
// leaf against the hit.t of the leaf's ENTRY, then the accepted candidates in ascending triangle order with a strict
// `nearestT > t`.  A wave does the same across its lanes: every (ray, triangle) pair of the lanes that stand on a leaf gets a
// lane of its own -- pairs are numbered by an exclusive prefix sum of the leaves' triangle counts (four ballots: counts are
// 1..8), a 64-word table in LDS tells pair p its triangle and its owner lane, the owner's ray constants arrive by
// ds_bpermute -- so a leaf costs its ray ONE round instead of one round per two triangles, and that round runs at full width.
// The owners then read the wave's ballot of accepted pairs: bits [prefix, prefix + count) are theirs, in triangle order.
//   nearest hit: the set bits are consumed in ascending order with `t < hit.t` -- candidates were admitted against the entry
//                hit.t, so this is exactly the reference's loop (and the packet branch's per-triangle select, bvh.cpp:376-424,
//                which admits against the running hit.t: the same set of updates);
//   any hit:     a set bit means occluded.
// The alpha test of a candidate is a function of (triangle, barycentrics) alone; it is evaluated by the pair's lane.
// Leaves that do not fit into the 64 pairs of this round keep their lanes waiting for the next one.
#ifndef PRT_COOP_LEAF
#define PRT_COOP_LEAF 1
#endif
#ifndef PRT_COOP_MIN
#define PRT_COOP_MIN 9u // leaf lanes that make a round a leaf round whatever the node lanes (9 leaves of 6 = 54 pairs)
#endif
#ifndef PRT_COOP_WEIGHT
#define PRT_COOP_WEIGHT 4u // ... or when leaf lanes x this many >= node lanes
#endif
#ifndef PRT_TRI_NT
#define PRT_TRI_NT 0 // triangle fetches with the non-temporal hint (an experiment for trees far larger than the caches)
#endif
#ifndef PRT_COOP_PREPOP
#define PRT_COOP_PREPOP 1
#endif
#define PRT_COOP_TRI_BITS 26 // a table word = triangle index | owner lane << 26

__device__ __forceinline__ uint32_t mbcnt64(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

#define PRT_COOP_STRIDE 72u // words of LDS per wave: 64 table words + a dump word (lanes that write nothing this round)
#define PRT_COOP_TRI_MASK ((1u << PRT_COOP_TRI_BITS) - 1u)
__device__ __forceinline__ float bperm(int byteAddr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byteAddr, __float_as_int(v))); }
__device__ __forceinline__ uint32_t bperm(int byteAddr, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute(byteAddr, (int)v); }

template <int MODE, class STK>
__device__ __forceinline__ void tracer_leaf_coop(const DevScene& sc, Tracer& T, bool onLeaf, const STK& st, Traffic& tr)
{
    constexpr bool OCC = (MODE == PRT_MODE_OCC_PACKET || MODE == PRT_MODE_OCC_SINGLE);
    constexpr bool PACKET = (MODE == PRT_MODE_PACKET || MODE == PRT_MODE_OCC_PACKET);
    const uint32_t lane = threadIdx.x & 63u;
    // ---- pairs: exclusive prefix sum of the triangle counts over the leaf lanes
    const uint32_t n = onLeaf ? leaf_count(T.ref) : 0u;
    const unsigned long long b0 = __ballot((n & 1u) != 0u), b1 = __ballot((n & 2u) != 0u), b2 = __ballot((n & 4u) != 0u), b3 = __ballot((n & 8u) != 0u);
    const uint32_t prefix = mbcnt64(b0) + 2u * mbcnt64(b1) + 4u * mbcnt64(b2) + 8u * mbcnt64(b3);
    const bool take = onLeaf && prefix + n <= 64u; // the prefix sum is monotone: the takers' pairs are 0 .. pairs-1
    const unsigned long long takeMask = __ballot(take);
    const uint32_t last = 63u - (uint32_t)__builtin_clzll(takeMask); // (the first leaf lane always takes: n <= 8)
    const uint32_t pairs = (uint32_t)__builtin_amdgcn_readlane((int)(prefix + n), (int)last);
#ifdef PRT_PROFILE
    tr.pTri2Lanes += pairs;
#endif
    // ---- table: pair p -> (first triangle - prefix) | owner lane << 26, the same word for every pair of a leaf, so that the
    // eight stores need no predicate: store i goes to entry min(i, n - 1), lanes that take no part store to the dump word
    {
        const uint32_t word = (((T.ref >> 4) - prefix) & PRT_COOP_TRI_MASK) | (lane << PRT_COOP_TRI_BITS);
        lds_u32* q = st.coop + (take ? prefix : 64u);
        const uint32_t nm1 = take ? n - 1u : 0u;
        q[0] = word;
#pragma unroll
        for (uint32_t i = 1; i < 8u; i++) q[i < nm1 ? i : nm1] = word;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // (no instruction: LDS operations of one wave complete in order)
    const bool pair = lane < pairs;
    const uint32_t word = st.coop[lane];
#if PRT_COOP_PREPOP
    // the entry a taker pops when its leaf is done is already known: read it beside the table, not after the round
    const int popAt = T.sp > 0 ? T.sp - 1 : 0;
    uint32_t popped = st.ldsRef[(popAt < STK::kLds ? popAt : 0) * PRT_BLOCK];
#endif
    const int oaddr = (int)((pair ? (word >> PRT_COOP_TRI_BITS) : lane) << 2);
    const uint32_t tri = (word + lane) & PRT_COOP_TRI_MASK;
    __builtin_amdgcn_sched_barrier(0); // (addresses first: a register of the fetch must not be reused for them while it is in flight)
    prt_f3 a, b, c;
    if (pair) {
        const float* tp = sc.tris + 9 * (size_t)tri;
#if PRT_TRI_NT
        a = __builtin_nontemporal_load((const PRT_AS1 prt_f3u*)tp);
        b = __builtin_nontemporal_load((const PRT_AS1 prt_f3u*)(tp + 3));
        c = __builtin_nontemporal_load((const PRT_AS1 prt_f3u*)(tp + 6));
#else
        a = *(const PRT_AS1 prt_f3u*)tp;
        b = *(const PRT_AS1 prt_f3u*)(tp + 3);
        c = *(const PRT_AS1 prt_f3u*)(tp + 6);
#endif
    }
    __builtin_amdgcn_sched_barrier(0); // the fetch is out before the lane exchanges, which need not wait for it
    // ---- the owner's ray constants (triangle.cpp:118-119 are per ray: prepare_shear)
    DevRay r;
    r.org = mk3(bperm(oaddr, T.r.org.x), bperm(oaddr, T.r.org.y), bperm(oaddr, T.r.org.z));
    r.shearX = bperm(oaddr, T.r.shearX);
    r.shearY = bperm(oaddr, T.r.shearY);
    r.invDz = bperm(oaddr, T.r.invDz);
    const uint32_t flags = bperm(oaddr, (T.r.swapXZ ? 1u : 0u) | (T.r.swapYZ ? 2u : 0u) | ((T.ref & PRT_LEAF_ALPHA) ? 4u : 0u));
    const float limit = bperm(oaddr, OCC ? T.maxT : T.hit.t);
    __builtin_amdgcn_sched_barrier(0);
    r.swapXZ = (flags & 1u) != 0u;
    r.swapYZ = (flags & 2u) != 0u;
    float t = -1.0f, bi = 0.0f, bj = 0.0f, bk = 0.0f;
    uint32_t primId = 0u;
    bool cand = false;
#ifdef PRT_PROFILE
    unsigned long long pAlpha0 = 0;
#endif
    if (pair) {
        t = tri_intersect(r, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), bi, bj, bk);
        primId = tri; // the LEAF-ORDER index (tri_candidate)
        cand = t >= 0.0001f && t < limit;
#ifdef PRT_PROFILE
        pAlpha0 = __builtin_amdgcn_s_memtime();
#endif
        // The alpha test of a candidate (bvh.cpp:336-338, 407-409) is a chain of dependent fetches that one or two lanes of the round walk
        // while the others wait (profile build: a quarter of C3's leaf rounds, a whole round's time each): the record carries its
        // texture's descriptor, and most tests end at the class of the tap's cell -- one word of a map that stays in the L2 -- instead
        // of four texel fetches.
        const uint32_t alphaRef = (cand && (flags & 4u)) ? gld(sc.triAlpha + tri) : 0u;
        if (alphaRef) {
            const float4* ap = sc.alpha + 3 * (size_t)(alphaRef - 1);
            const float4 u0 = gld4(ap), u1 = gld4(ap + 1), u2 = gld4(ap + 2);
            const Vec2 uv = Vec2{bi * u0.x + bj * u0.z + bk * u1.x, bi * u0.y + bj * u0.w + bk * u1.y}; // bvh.cpp:336, 407
            const uint4 d = make_uint4(asu(u2.x), asu(u2.y), asu(u2.z), asu(u2.w));
            int32_t x0, y0;
            float xt, yt;
            uint32_t cls = 0u;
            if (bilinear_cell(uv, (int32_t)d.y, (int32_t)d.z, PACKET, x0, y0, xt, yt)) {
                const uint32_t cell = (uint32_t)x0 + (uint32_t)y0 * d.y;
                cls = (gld(sc.alphaClass + asu(u1.w) + (cell >> 4)) >> ((cell & 15u) * 2u)) & 3u;
            }
            if (cls == 0u) cand = tex_test_alpha<false>(sc, d, uv, PACKET, tr); // a cell with texels on both sides of the threshold, or a non-finite uv
            else cand = cls == 1u;
        }
#ifdef PRT_PROFILE
        if (__ballot(cand && (flags & 4u)) != 0ull) { // (ballot of the lanes inside `if (pair)`: some candidate lies in a leaf with alpha-tested triangles)
            tr.pAlphaRounds++;
            tr.pAlphaCycles += __builtin_amdgcn_s_memtime() - pAlpha0;
        }
#endif
    }
    const unsigned long long acc = __ballot(cand);
    // ---- owners
    uint32_t seg = take ? ((uint32_t)(acc >> prefix) & ((1u << n) - 1u)) : 0u;
    if (OCC) {
        if (take) {
            if (seg) {
                T.occ = true;
                T.ref = PRT_REF_NONE;
                T.sp = 0;
                T.m = sc.bvhCount; // finished
            } else {
#if PRT_COOP_PREPOP
                if (T.sp == 0) T.ref = PRT_REF_NONE;
                else {
                    if (popAt >= STK::kLds) popped = st.get(popAt);
                    T.sp = popAt;
                    T.ref = popped;
                }
#else
                T.ref = tracer_pop<MODE, false>(T, st, tr);
#endif
            }
        }
    } else {
        if (acc != 0ull) {
            // the owners walk their accepted pairs in ascending order: only t travels per turn, the winner's barycentrics and index
            // once at the end
            uint32_t win = 0xffffffffu;
            for (;;) {
#ifdef PRT_PROFILE
                tr.pLeafUpdates++;
#endif
                const bool more = seg != 0u;
                if (!__any(more)) break;
                const uint32_t src = more ? prefix + (uint32_t)__builtin_ctz(seg) : lane;
                const float t2 = bperm((int)(src << 2), t);
                if (more) {
                    if (t2 < T.hit.t) {
                        T.hit.t = t2;
                        win = src;
                    }
                    seg &= seg - 1u;
                }
            }
            if (__any(win != 0xffffffffu)) {
                const int waddr = (int)((win != 0xffffffffu ? win : lane) << 2);
                const float i2 = bperm(waddr, bi), j2 = bperm(waddr, bj), k2 = bperm(waddr, bk);
                const uint32_t p2 = bperm(waddr, primId);
                if (win != 0xffffffffu) {
                    T.hit.i = i2;
                    T.hit.j = j2;
                    T.hit.k = k2;
                    T.hit.primId = p2;
                    T.hit.meshId = T.m;
                }
            }
        }
#if PRT_COOP_PREPOP
        if (take) {
            if (MODE == PRT_MODE_PACKET) T.ref = tracer_pop<MODE, false>(T, st, tr); // (its pop tests the entry distance against the new hit.t)
            else if (T.sp == 0) T.ref = PRT_REF_NONE;
            else {
                if (popAt >= STK::kLds) popped = st.get(popAt);
                T.sp = popAt;
                T.ref = popped;
            }
        }
#else
        if (take) T.ref = tracer_pop<MODE, false>(T, st, tr);
#endif
    }
}

__device__ __forceinline__ bool ref_is_internal(uint32_t ref) { return !(ref & PRT_REF_LEAF); }
__device__ __forceinline__ bool ref_is_leaf(uint32_t ref) { return (ref & PRT_REF_LEAF) && ref != PRT_REF_NONE; }

// Step phase of the wave loops: a step is one internal node or one triangle.  Each round the larger of the two groups of
// lanes steps (the other waits: SIMD lanes that cannot share an instruction stream), until PRT_IDLE_BREAK lanes have nothing
// to step on -- their ray is finished or moves to the next BVH -- and the wave goes back to its refill point to serve them.
// `queued()` (wave-uniform) tells how many rays wait in the queue the wave refills from: a wave that runs with lanes WITHOUT a
// ray -- the queue was short when it last refilled -- looks every fourth round and goes back to its refill point as soon as
// PRT_EMPTY_GAIN rays have arrived, instead of walking its few rays to their end at a fraction of its width.
#ifndef PRT_EMPTY_BREAK
#define PRT_EMPTY_BREAK 0u // lanes without a ray (plus finished ones) from which the queue is polled; 0 = never (measured: 16 / 16 costs 2 %, 8 / 8 7 %, 24 / 24 nothing: the refill turn it buys costs what the fuller rounds save)
#endif
#ifndef PRT_EMPTY_GAIN
#define PRT_EMPTY_GAIN 16u
#endif
struct NoQueue {
    __device__ __forceinline__ uint32_t operator()() const { return 0u; }
};
template <int MODE, bool COUNT, class STK, class Queued = NoQueue>
__device__ __forceinline__ void trace_step_phase(const DevScene& sc, Tracer& T, bool active, const STK& st, Traffic& tr, uint32_t& overflow,
                                                 const Queued& queued = Queued())
{
    const uint32_t empty = 64u - (uint32_t)__popcll(__ballot(active)); // (lanes take rays at the refill point only)
    uint32_t round = 0;
    for (;;) {
        const bool onNode = active && ref_is_internal(T.ref);
        const bool onLeaf = active && ref_is_leaf(T.ref);
        const uint32_t nNode = (uint32_t)__popcll(__ballot(onNode)), nLeaf = (uint32_t)__popcll(__ballot(onLeaf));
        if (nNode + nLeaf == 0u) break;
        // Which kind steps.  Serial leaf steps (counting build): the larger group.  Cooperative leaf rounds: a leaf lane stands for
        // about six pair lanes, so leaves wait until a round's worth of pairs has gathered or few lanes are left on nodes.
        const bool doNode = (PRT_COOP_LEAF && !COUNT) ? (nLeaf == 0u || (nLeaf < PRT_COOP_MIN && nLeaf * PRT_COOP_WEIGHT < nNode)) : nNode >= nLeaf;
#ifdef PRT_PAD_VALU
        { // sensitivity probe of tools/build_variants.py: PRT_PAD_VALU extra vector instructions per round (never in the product)
            float pad = T.maxT;
#pragma unroll
            for (int q = 0; q < PRT_PAD_VALU; q++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(pad));
            asm volatile("" ::"v"(pad));
        }
#endif
#ifdef PRT_PROFILE
        if (doNode) {
            { // distinct records of this round: one per group of lanes that stand on the same node
                unsigned long long m = __ballot(onNode);
                while (m) {
                    const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)T.ref, (int)__builtin_ctzll(m));
                    m &= ~__ballot(onNode && T.ref == r);
                    tr.pNodeDistinct++;
                }
            }
            tr.pNodeRounds++;
            tr.pNodeLanes += nNode;
            tr.pNodeWaitLeaf += nLeaf;
            tr.pNodeDone += (unsigned long long)__popcll(__ballot(active && T.ref == PRT_REF_NONE));
            tr.pNodeNoRay += (unsigned long long)__popcll(__ballot(!active));
        } else {
            tr.pLeafRounds++;
            tr.pLeafLanes += nLeaf;
            tr.pLeafWaitNode += nNode;
            tr.pLeafDone += (unsigned long long)__popcll(__ballot(active && T.ref == PRT_REF_NONE));
            tr.pLeafNoRay += (unsigned long long)__popcll(__ballot(!active));
            if (!(PRT_COOP_LEAF && !COUNT)) tr.pTri2Lanes += (unsigned long long)__popcll(__ballot(onLeaf && (T.ref & 15u) > 1u));
        }
#endif
        if (doNode) {
#ifdef PRT_PROFILE
            const int sp0 = T.sp;
#endif
            if (onNode) tracer_node<MODE, COUNT>(sc, T, st, tr, overflow);
#ifdef PRT_PROFILE
            if (onNode && ref_is_internal(T.ref) && !(T.ref & PRT_REF_HOT)) {
                if (T.sp < sp0) tr.pPopInt++;
                else tr.pDirectInt++;
            }
#endif
        } else if (PRT_COOP_LEAF && !COUNT) {
            tracer_leaf_coop<MODE>(sc, T, onLeaf, st, tr);
        } else {
            if (onLeaf) tracer_tri<MODE, COUNT>(sc, T, st, tr);
        }
        const uint32_t done = (uint32_t)__popcll(__ballot(active && T.ref == PRT_REF_NONE));
        if (done >= PRT_IDLE_BREAK) break;
        if (PRT_EMPTY_BREAK && empty + done >= PRT_EMPTY_BREAK && (++round & 3u) == 0u) {
            const uint32_t gain = empty + done < PRT_EMPTY_GAIN ? empty + done : PRT_EMPTY_GAIN;
            if (queued() >= gain) break;
        }
    }
}

// The wave-level loop: every lane owns one ray at a time and takes a new one from `src` as soon as its own is
// finished (persistent lanes, wave-aggregated claim: ballot + popcount + one atomic per wave).  "while-while": all
// lanes that stand on an internal node step together until every lane stands on a leaf (or has nothing), then the
// leaves are intersected together.
//   Src: uint32_t count(); uint32_t* cursor(); void load(uint32_t i, Vec3& org, Vec3& dir, float& maxT, uint32_t& rev);
//        void store_hit(uint32_t i, const DevHit&); void store_occ(uint32_t i, bool);
template <int MODE, bool COUNT, class Src, class STK>
__device__ __forceinline__ void trace_loop(const DevScene& sc, Src& src, const STK& st, Traffic& tr, uint32_t& overflow)
{
    const uint32_t n = src.count();
    const uint32_t lane = threadIdx.x & 63u;
    Tracer T{}; // every field defined in every lane (see trace_queue, prt_frame.h)
    T.ref = PRT_REF_NONE;
    T.sp = 0;
    T.m = 0;
    bool active = false, exhausted = false;
    uint32_t item = 0;
    // The wave reserves ranges of the queue CHUNK rays at a time (one returning atomic per range, not per refill) and
    // hands them to its lanes as they become free; both bounds are wave-uniform.
    const uint32_t totalWaves = gridDim.x * (PRT_BLOCK / 64);
    uint32_t chunk = n / (totalWaves * 4u);
    chunk = chunk < 64u ? 64u : (chunk > 1024u ? 1024u : chunk);
    if (n == 0u) return;
    // every wave starts on its own static range (no atomic); the shared cursor hands out what lies beyond them
    const uint32_t waveId = blockIdx.x * (PRT_BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t staticEnd = totalWaves * chunk;
    uint32_t rangeNext = waveId * chunk, rangeEnd = rangeNext + chunk;
    for (;;) {
        unsigned long long need = __ballot(!active && !exhausted);
        if (need) {
            const uint32_t k = (uint32_t)__popcll(need), avail = rangeEnd - rangeNext;
            uint32_t newBase = 0;
            if (avail < k) { // wave-uniform
                if (lane == 0) newBase = staticEnd + atomicAdd(src.cursor(), chunk);
                newBase = (uint32_t)__shfl((int)newBase, 0, 64);
            }
            if (!active && !exhausted) {
                uint32_t r = mbcnt64(need);
                item = (r < avail) ? rangeNext + r : newBase + (r - avail);
                if (item < n) {
                    Vec3 org, dir;
                    float maxT;
                    uint32_t rev;
                    src.load(item, org, dir, maxT, rev);
                    tracer_begin<MODE>(T, org, dir, maxT, rev);
                    // A ray with a NaN in its origin or direction (a degenerate normal upstream) misses every triangle
                    // -- some edge function, or the determinant, is NaN (triangle.cpp:126-150) -- yet under the
                    // reference's min/max-with-NaN semantics it passes EVERY box test, i.e. it walks the whole tree:
                    // milliseconds on a CPU, seconds for one GPU lane on a 5 M triangle scene.  Its answer is "miss" /
                    // "not occluded" either way, so the timed build skips the walk; the counting build keeps it and stays
                    // equal to the oracle's event counts.
                    if (!COUNT && !(org.x == org.x && org.y == org.y && org.z == org.z && dir.x == dir.x && dir.y == dir.y && dir.z == dir.z))
                        T.m = sc.bvhCount - 1u;
                    active = true;
                } else {
                    exhausted = true;
                }
            }
            if (avail < k) {
                rangeNext = newBase + (k - avail);
                rangeEnd = newBase + chunk;
            } else {
                rangeNext += k;
            }
        }
        if (!__any(active)) break;
        if (active && T.ref == PRT_REF_NONE) {
            if (!tracer_next_bvh<MODE, COUNT>(sc, T, tr)) {
                if (MODE == PRT_MODE_PACKET || MODE == PRT_MODE_SINGLE) {
                    if (T.hit.t == T.maxT) T.hit.t = -1.0f; // setMissForMaxT, scene.cpp:62
                    src.store_hit(item, T.hit);
                } else {
                    src.store_occ(item, T.occ);
                }
                active = false;
            }
        }
        trace_step_phase<MODE, COUNT>(sc, T, active, st, tr, overflow);
    }
}

// ---------------------------------------------------------------------------- surface + material
struct Surface { // mesh.h:17-27; dp/duv are re-fetched from the bump record by prim when needed
    Vec3 normal;
    Vec2 uv;
    uint32_t mat;  // global material index
    uint32_t prim; // global triangle index (mesh order)
};

// mesh.cpp:311-364
template <bool COUNT>
__device__ __forceinline__ void get_surface(const DevScene& sc, const DevHit& h, Surface& s, Traffic& tr)
{
    if (COUNT) tr.nHit++;
    uint32_t gp = h.primId; // leaf-order index over all BVHs
    const float4* sp = sc.shade + 4 * (size_t)gp;
    float4 s0 = gld4(sp), s1 = gld4(sp + 1), s2 = gld4(sp + 2), s3 = gld4(sp + 3);
    if (sc.hasNormals[h.meshId]) {
        Vec3 n = add3(add3(scale3(h.i, mk3(s0.x, s0.y, s0.z)), scale3(h.j, mk3(s1.x, s1.y, s1.z))), scale3(h.k, mk3(s2.x, s2.y, s2.z)));
        s.normal = normalize3(n);
    } else {
        s.normal = mk3(s0.x, s0.y, s0.z); // normalize(cross(p1-p0,p2-p0)) precomputed with the same operations
    }
    s.uv = Vec2{h.i * s1.w + h.j * s3.x + h.k * s3.z, h.i * s2.w + h.j * s3.y + h.k * s3.w};
    s.mat = asu(s0.w);
    s.prim = gp;
}

// ---------------------------------------------------------------------------- InfiniteAreaLight::sample, light.cpp:86-128
// The reference scans a CDF from index 1 for the first entry that exceeds u and differs from its predecessor (:93-104,
// :108-119).  The tables are non-decreasing (running sums of non-negative terms; upload checks it), so that index is the
// upper bound of u found by bisection -- whose predecessor is <= u < entry, hence differs -- except when already
// cdf[0] > u: then every entry exceeds u and the scan stops at the first one that differs from its predecessor, a
// constant of the table (`first`).  A row of NaNs (an all-black row, :63-70) has no entry that exceeds anything,
// under either search.  Returns `n` when the scan would run off the end.
__device__ __forceinline__ int32_t cdf_find(const float* cdf, int32_t n, int32_t first, float u)
{
    int32_t lo = 1, hi = n;
    while (lo < hi) {
        int32_t mid = (lo + hi) >> 1;
        if (gld(cdf + mid) > u) hi = mid;
        else lo = mid + 1;
    }
    if (lo == 1 && lo < n && gld(cdf) > u) lo = first;
    return lo;
}

template <bool COUNT>
__device__ __forceinline__ void env_sample(const DevScene& sc, float ux, float uy, Vec3& dir, Vec3& color, Traffic& tr)
{
    const int32_t W = sc.envW, H = sc.envH;
    float pdfV = 1.0f, yf = 0.0f, pdfH = 1.0f, xf = 0.0f;
    const int32_t y = cdf_find(sc.envV, H, sc.envFirstY, uy);
    if (y < H) {
        float prev = gld(sc.envV + (y - 1));
        pdfV = gld(sc.envV + y) - prev;
        yf = (float)y + (uy - prev) / pdfV - 1.0f;
        // the row of the horizontal table is the CDF index y, not the texel row y-1 (:109)
        const float* row = sc.envHor + (size_t)y * (size_t)W;
        const int32_t x = cdf_find(row, W, gld(sc.envFirstX + y), ux);
        if (x < W) {
            float prevx = gld(row + (x - 1));
            pdfH = gld(row + x) - prevx;
            xf = (float)x + (ux - prevx) / pdfH - 1.0f;
        }
    }
    // y == H: the reference reads one row past the table (:110); defined here as "nothing exceeds u.x" (xf = 0, pdfH = 1)
    Vec2 uv = Vec2{xf / (float)W, yf / (float)H};
    float k[4];
    int32_t idx[4];
    bilinear(k, idx, 1, uv, W, H, false); // texel indices (Texture::sample<Vector3f, float>, texture.cpp:102-139)
    if (COUNT) tr.nTap++;
    Vec3 c = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float4 t = gld4(sc.envTexels + idx[i]);
        c = add3(c, scale3(k[i], mk3(t.x, t.y, t.z)));
    }
    c = div3s(div3s(c, pdfH * pdfV), (float)(W * H)); // :118
    float theta = 2.0f * 3.14159265358979323846f * (uv.x + 0.5f); // :121
    float phi = 3.14159265358979323846f * uv.y;
    float st, ct, sp, cp;
    prt_sincosf(theta, &st, &ct);
    prt_sincosf(phi, &sp, &cp);
    dir = normalize3(mk3(ct * sp, cp, st * sp)); // :125
    color = c;
}

// material.cpp:87-96 (degamma :24-28 needs powf: see prt_powf in prt_devmath.h)
template <bool COUNT>
__device__ __forceinline__ Vec3 sample_diffuse(const DevScene& sc, uint32_t mat, Vec2 uv, Traffic& tr)
{
    // the material record carries the descriptors of its two maps: material and descriptor arrive in one round trip
    const float4* mp = sc.mats + PRT_MAT_STRIDE * (size_t)mat;
    float4 m0 = gld4(mp), m2 = gld4(mp + 2), m3 = gld4(mp + 3);
    Vec3 color = mk3(m0.x, m0.y, m0.z);
    int32_t tex = (int32_t)asu(m2.x);
    if (tex >= 0) {
        Vec3 c = tex_sample3<COUNT>(sc, make_uint4(asu(m3.x), asu(m3.y), asu(m3.z), asu(m3.w)), uv, tr);
        color = mul3(color, mk3(prt_powf_2p2(c.x), prt_powf_2p2(c.y), prt_powf_2p2(c.z)));
    }
    return color;
}

// material.cpp:98-114
template <bool COUNT>
__device__ __forceinline__ Vec3 sample_bump(const DevScene& sc, uint32_t mat, const Surface& s, Traffic& tr)
{
    Vec3 normal = s.normal;
    const float4* mp = sc.mats + PRT_MAT_STRIDE * (size_t)mat;
    float4 m2 = gld4(mp + 2), m4 = gld4(mp + 4);
    int32_t tex = (int32_t)asu(m2.y);
    if (tex >= 0) {
        const float4* bp = sc.bump + 3 * (size_t)s.prim;
        float4 b0 = gld4(bp), b1 = gld4(bp + 1), b2 = gld4(bp + 2);
        Vec3 dp01 = mk3(b0.x, b0.y, b0.z), dp02 = mk3(b1.x, b1.y, b1.z);
        Vec2 duv01 = Vec2{b0.w, b1.w}, duv02 = Vec2{b2.x, b2.y};
        const uint4 d = make_uint4(asu(m4.x), asu(m4.y), asu(m4.z), asu(m4.w));
        float onePixel = 0.5f / (float)(int32_t)d.y + 0.5f / (float)(int32_t)d.z; // texture.h:26
        float b = tex_sample1<COUNT>(sc, d, s.uv, tr);
        float b01 = tex_sample1<COUNT>(sc, d, Vec2{s.uv.x + onePixel * duv01.x, s.uv.y + onePixel * duv01.y}, tr) - b;
        float b02 = tex_sample1<COUNT>(sc, d, Vec2{s.uv.x + onePixel * duv02.x, s.uv.y + onePixel * duv02.y}, tr) - b;
        float nk = 4.0f;
        normal = normalize3(add3(add3(normal, scale3(nk * b01, dp01)), scale3(nk * b02, dp02)));
    }
    return normal;
}

// ---------------------------------------------------------------------------- RNG (random.h:23-48)
__device__ __forceinline__ uint32_t xorshift32(uint32_t x)
{
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return x;
}
__device__ __forceinline__ float rng_to_float(uint32_t u) { return asf((u & 0x007fffffu) | 0x3f800000u) - 1.0f; }
__device__ __forceinline__ uint32_t pixel_seed(uint32_t x, uint32_t y, uint32_t width, uint32_t seed)
{
    uint32_t h = x + y * width + seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h | 1u;
}
