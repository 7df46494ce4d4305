// prt_kernels.hip -- libprt_hip.so: HIP kernels of PRT's per-pixel path-tracing loop for MI355X
// (gfx950, wave64) and the C-ABI of include/prt_hip.h.
//
// Execution mapping (DESIGN.md "Kernel"): the reference traces a pixel as samples/8 packets of 8
// paths that share one xorshift32 stream (path_tracer.cpp:57-75).  Here one pixel is owned by 8
// consecutive lanes (one lane = one of the 8 path slots), so a wave64 holds 8 pixels; groups pull
// pixels from an atomic counter in tile-major order.  Every random-number phase of the reference
// draws a statically known number of values per alive slot in slot order, so a lane obtains its
// values by stepping the shared state "number of draws owed by lower slots" times (wave ballot +
// popcount); the ordered compaction of alive paths (path_tracer.cpp:255-293) is a ballot/prefix
// rank followed by an 8-lane gather.  BVH traversal is one lane = one ray with a per-lane stack in
// LDS.  No MFMA: there is no dense contraction on this path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/prt_hip.h"
#include "prt_device.h"
#include "prt_internal.h"

// ============================================================================ device: group helpers
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// ballot restricted to the caller's 8-lane group (bits 0..7)
__device__ __forceinline__ uint32_t group_ballot(bool p, uint32_t gbase)
{
    unsigned long long b = __ballot(p);
    return (uint32_t)(b >> gbase) & 0xffu;
}

__device__ __forceinline__ uint32_t nth_set(uint32_t m, uint32_t n)
{
    for (uint32_t k = 0; k < n; k++) m &= m - 1u;
    return m ? (uint32_t)__builtin_ctz(m) : 0u;
}

// sum over the wave's active lanes (all 64 lanes must call it); result valid in every lane
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}

__device__ __forceinline__ float shf(float v, uint32_t srcLane) { return __shfl(v, (int)srcLane, 64); }
__device__ __forceinline__ uint32_t shu(uint32_t v, uint32_t srcLane) { return (uint32_t)__shfl((int)v, (int)srcLane, 64); }
__device__ __forceinline__ Vec3 sh3(Vec3 v, uint32_t srcLane) { return mk3(shf(v.x, srcLane), shf(v.y, srcLane), shf(v.z, srcLane)); }

// path_tracer.cpp:143-153 (and :176-184): cosine-weighted direction about `normal`
__device__ __forceinline__ Vec3 diffuse_dir(Vec3 normal, float r2, float r1)
{
    const float kPi = 3.14159265358979323846f;
    float r2sq = sqrtf(r2);
    Vec3 u = (fabsf(normal.x) > 0.1f) ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
    Vec3 tangent = normalize3(cross3(normal, u));
    Vec3 binormal = normalize3(cross3(tangent, normal));
    float theta = 2.0f * kPi * r1;
    float sn, cs;
    prt_sincosf(theta, &sn, &cs);
    return add3(add3(scale3(r2sq * cs, binormal), scale3(r2sq * sn, tangent)), scale3(1.0f - r2, normal));
}

// camera.cpp:46-56 for one lane: consumes the two draws dxBits, dyBits
__device__ __forceinline__ Vec3 camera_dir(const DevCamera& cam, uint32_t x, uint32_t y, uint32_t dxBits, uint32_t dyBits)
{
    const float kScreenScale = 0.6f;
    const float kAspect = (float)cam.width / (float)cam.height;
    const float kScaleX = 0.5f * cam.invWidth;
    const float kScaleY = 0.5f * cam.invHeight;
    float dx = (2.0f * rng_to_float(dxBits) - 1.0f) * kScaleX;
    float dy = (2.0f * rng_to_float(dyBits) - 1.0f) * kScaleY;
    float nx = 2.0f * ((float)x * cam.invWidth - 0.5f + dx) * kScreenScale * kAspect;
    float ny = -2.0f * ((float)y * cam.invHeight - 0.5f + dy) * kScreenScale;
    Vec3 right = mk3(cam.right[0], cam.right[1], cam.right[2]);
    Vec3 up = mk3(cam.up[0], cam.up[1], cam.up[2]);
    Vec3 fwd = mk3(cam.dir[0], cam.dir[1], cam.dir[2]);
    return normalize3(add3(add3(scale3(nx, right), scale3(ny, up)), fwd));
}

// Camera::GenerateJitteredRayPacket (camera.cpp:35-73) across the 8 lanes of a group: lane s uses draws
// 2s and 2s+1 of the 16 the packet consumes; avgDir is the lane-ordered sum / 8.
__device__ __forceinline__ void camera_packet(const DevCamera& cam, uint32_t& rng, uint32_t x, uint32_t y, uint32_t slot, uint32_t gbase,
                                              DevRay& ray, Vec3& avgDir)
{
    uint32_t s = rng, dxb = 0, dyb = 0;
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) {
        s = xorshift32(s);
        if (j == 2 * slot) dxb = s;
        if (j == 2 * slot + 1) dyb = s;
    }
    rng = s;
    ray.org = mk3(cam.pos[0], cam.pos[1], cam.pos[2]);
    ray.dir = camera_dir(cam, x, y, dxb, dyb);
    Vec3 avg = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (uint32_t l = 0; l < 8; l++) avg = add3(avg, sh3(ray.dir, gbase + l));
    avgDir = div3s(avg, 8.0f);
    prepare_soa(ray);
}

struct Surf5 { // what moves between slots at a compaction
    Vec3 normal;
    Vec2 uv;
    uint32_t mat, prim;
};

// ============================================================================ the per-pixel loop
// The loop of the reference is run as a state machine over pixel groups (a group = the 8 path slots of one pixel, one lane
// each); one round of a group = a shade pass (consume the hits of its last rays, run the bounce of path_tracer.cpp:124-293,
// emit the next rays) followed by the traversal of those rays.  A packet needs 1 + maxDepth rounds, a pixel
// (samples/8)*(1+maxDepth)+1.  Scheduling, queues and kernels: prt_frame.h.
enum { Q_PRIMARY = PRT_MODE_PACKET, Q_SCATTER = PRT_MODE_SINGLE, Q_OCC_PACKET = PRT_MODE_OCC_PACKET, Q_OCC_SINGLE = PRT_MODE_OCC_SINGLE, Q_COUNT = 4 };
enum { PH_START = 0, PH_WAIT_PRIMARY = 1, PH_WAIT_BOUNCE = 2, PH_DONE = 3 };
#define SLOT_HAS_SHADOW 1u
#define SLOT_SURVIVE 2u
#define SLOT_LIGHT_SET 4u


// streaming (touch-once-per-iteration) state goes around the caches' retention so that the BVH stays resident
typedef float f4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p)
{
#ifdef PRT_STATE_PLAIN
    return gld4(p);
#else
    f4_t v = __builtin_nontemporal_load((const PRT_AS1 f4_t*)p);
    return make_float4(v.x, v.y, v.z, v.w);
#endif
}
__device__ __forceinline__ void nt_store4(float4* p, float4 v)
{
#ifdef PRT_STATE_PLAIN
    gst4(p, v);
#else
    f4_t w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, (PRT_AS1 f4_t*)p);
#endif
}
__device__ __forceinline__ uint32_t nt_load(const uint32_t* p) { return __builtin_nontemporal_load((const PRT_AS1 uint32_t*)p); }
__device__ __forceinline__ void nt_store(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, (PRT_AS1 uint32_t*)p); }

#include "prt_frame.h"

// ============================================================================ G-buffer visualiser
// GbufferVisualizer::TraceBlock (gbuffer_visualizer.cpp:17-51): per pixel ONE jittered single ray (Camera::GenerateJitteredRay,
// camera.cpp:12-33: two generateMinus1to1 draws from the pixel's generator), the single-ray nearest traversal, then the
// surface's diffuse colour (type 0) or bump-mapped normal * 0.5 + 0.5 (types 1, 2).  Same persistent wave loop as the trace
// kernels; the ray source makes the camera ray, the hit sink shades and writes the pixel (Image::writePixel, image.cpp:44-50).
struct GbufArgs {
    DevScene sc;
    DevCamera cam;
    uint32_t x0, y0, rw, rh, type, seed;
    float exposure;
    float* rgb;
    uint32_t* cursor;
    uint32_t* spill;
    uint32_t spillStride;
    unsigned long long* counters;
};

struct GbufSrc {
    const GbufArgs* A;
    __device__ __forceinline__ uint32_t count() const { return A->rw * A->rh; }
    __device__ __forceinline__ uint32_t* cursor() const { return A->cursor; }
    __device__ __forceinline__ void load(uint32_t i, Vec3& org, Vec3& dir, float& maxT, uint32_t& rev) const
    {
        const uint32_t x = A->x0 + i % A->rw, y = A->y0 + i / A->rw;
        const uint32_t s1 = xorshift32(pixel_seed(x, y, A->cam.width, A->seed)), s2 = xorshift32(s1);
        org = mk3(A->cam.pos[0], A->cam.pos[1], A->cam.pos[2]);
        dir = camera_dir(A->cam, x, y, s1, s2);
        maxT = 100000.0f; // camera.cpp:26
        rev = 0;
    }
    __device__ __forceinline__ void store_hit(uint32_t i, const DevHit& h) const
    {
        const uint32_t x = A->x0 + i % A->rw, y = A->y0 + i / A->rw;
        Vec3 color = mk3(0.0f, 0.0f, 0.0f);
        if (h.t != -1.0f) {
            Traffic tr{};
            Surface s;
            get_surface<false>(A->sc, h, s, tr);
            if (A->type == 0u) color = sample_diffuse<false>(A->sc, s.mat, s.uv, tr);
            else color = add3(scale3(0.5f, sample_bump<false>(A->sc, s.mat, s, tr)), mk3(0.5f, 0.5f, 0.5f));
        }
        float* px = A->rgb + ((size_t)x + (size_t)y * A->cam.width) * 3;
        px[0] = A->exposure * color.x;
        px[1] = A->exposure * color.y;
        px[2] = A->exposure * color.z;
    }
    __device__ __forceinline__ void store_occ(uint32_t, bool) const {}
};

__global__ __launch_bounds__(PRT_BLOCK) void gbuffer_kernel(GbufArgs A)
{
    __shared__ uint32_t ldsRef[PRT_STACK_LDS * PRT_BLOCK];
    __shared__ float ldsT[PRT_BLOCK];
    const uint32_t tid = threadIdx.x;
    __shared__ uint32_t coopTbl[(PRT_BLOCK / 64) * PRT_COOP_STRIDE];
    const StackT<PRT_STACK_LDS> st{(lds_u32*)&ldsRef[tid], (lds_f32*)&ldsT[tid], A.spill, A.spillStride, nullptr,
                                   (lds_u32*)&coopTbl[(tid >> 6) * PRT_COOP_STRIDE]};
    GbufSrc src{&A};
    Traffic tr{};
    uint32_t overflow = 0;
    trace_loop<PRT_MODE_SINGLE, false>(A.sc, src, st, tr, overflow);
    if (overflow) atomicAdd(&A.counters[7], 1ull);
}

// ============================================================================ row-level test kernels
// Only in the test build of the library (-DPRT_TEST_ENTRY_POINTS -> libprt_hip_test.so, include/prt_hip_test.h): the
// product library exports neither these kernels nor their entry points.
#ifdef PRT_TEST_ENTRY_POINTS
struct RaysArgs {
    DevScene sc;
    uint32_t n;
    const float* org;
    const float* dir;
    float maxT;
    prt_hit* hits;
    uint32_t* cursor;
    uint32_t* spill;
    uint32_t spillStride;
    unsigned long long* counters;
};

struct ArraySrc {
    const RaysArgs* A;
    int mode;
    __device__ __forceinline__ uint32_t count() const { return A->n; }
    __device__ __forceinline__ uint32_t* cursor() const { return A->cursor; }
    __device__ __forceinline__ void load(uint32_t i, Vec3& org, Vec3& dir, float& maxT, uint32_t& rev) const
    {
        org = mk3(A->org[3 * i], A->org[3 * i + 1], A->org[3 * i + 2]);
        dir = mk3(A->dir[3 * i], A->dir[3 * i + 1], A->dir[3 * i + 2]);
        maxT = A->maxT;
        rev = 0;
        if (mode == PRT_MODE_PACKET) { // avgDir = lane-ordered sum of the packet's 8 directions / 8 (camera.cpp:56,70)
            uint32_t g = i & ~7u;
            Vec3 avg = mk3(0, 0, 0);
            for (uint32_t l = 0; l < 8; l++) avg = add3(avg, mk3(A->dir[3 * (g + l)], A->dir[3 * (g + l) + 1], A->dir[3 * (g + l) + 2]));
            avg = div3s(avg, 8.0f);
            rev = (avg.x < 0.0f ? 1u : 0u) | (avg.y < 0.0f ? 2u : 0u) | (avg.z < 0.0f ? 4u : 0u);
        }
    }
    __device__ __forceinline__ void store_hit(uint32_t i, const DevHit& h) const
    {
        prt_hit o;
        // the device names a hit triangle by its leaf-order index; the reference's primId is that triangle's index in its mesh
        // (a NaN limit keeps hit.t NaN: no triangle was recorded and primId is not an index)
        o.t = h.t; o.i = h.i; o.j = h.j; o.k = h.k;
        o.primId = (h.t != -1.0f && h.t == h.t) ? gld(A->sc.triPrim + h.primId) : h.primId;
        o.meshId = h.meshId;
        A->hits[i] = o;
    }
    __device__ __forceinline__ void store_occ(uint32_t i, bool occ) const
    {
        prt_hit o;
        o.t = occ ? 1.0f : 0.0f; o.i = o.j = o.k = 0.0f; o.primId = 0; o.meshId = 0;
        A->hits[i] = o;
    }
};

template <int MODE>
__global__ __launch_bounds__(PRT_BLOCK) void rays_kernel(RaysArgs A)
{
    constexpr int NLDS = (MODE == PRT_MODE_PACKET) ? PRT_STACK_LDS_PACKET : PRT_STACK_LDS;
    __shared__ uint32_t ldsRef[NLDS * PRT_BLOCK];
    __shared__ float ldsT[(MODE == PRT_MODE_PACKET ? NLDS : 1) * PRT_BLOCK];
    const uint32_t tid = threadIdx.x;
    __shared__ uint32_t coopTbl[(PRT_BLOCK / 64) * PRT_COOP_STRIDE];
    const StackT<NLDS> st{(lds_u32*)&ldsRef[tid], (lds_f32*)&ldsT[tid], A.spill, A.spillStride, nullptr,
                          (lds_u32*)&coopTbl[(tid >> 6) * PRT_COOP_STRIDE]};
    ArraySrc src{&A, MODE};
    Traffic tr{};
    uint32_t overflow = 0;
    trace_loop<MODE, false>(A.sc, src, st, tr, overflow);
    if (overflow) atomicAdd(&A.counters[7], 1ull);
}

__global__ void leaf_kernel(uint32_t n, const float* rec, float* out)
{
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* p = rec + 22 * (size_t)r;
    float* q = out + 24 * (size_t)r;
    for (int k = 0; k < 24; k++) q[k] = 0.0f;
    DevRay rs, r1;
    rs.org = r1.org = mk3(p[0], p[1], p[2]);
    rs.dir = r1.dir = mk3(p[3], p[4], p[5]);
    Vec3 p0 = mk3(p[6], p[7], p[8]), p1 = mk3(p[9], p[10], p[11]), p2 = mk3(p[12], p[13], p[14]);
    Box b{mk3(p[15], p[16], p[17]), mk3(p[18], p[19], p[20])};
    float maxT = p[21];
    prepare_soa(rs);
    prepare_single(r1);
    float bi, bj, bk;
    float t = tri_intersect(rs, p0, p1, p2, bi, bj, bk);
    q[0] = t;
    if (t != -1.0f) { q[1] = bi; q[2] = bj; q[3] = bk; }
    t = tri_intersect(r1, p0, p1, p2, bi, bj, bk);
    q[4] = t;
    if (t != -1.0f) { q[5] = bi; q[6] = bj; q[7] = bk; }
    q[12] = box_t(b, r1);
    q[13] = box_bool(b, r1, maxT) ? 1.0f : 0.0f;
    q[14] = box_soa(b, rs, maxT) ? 1.0f : 0.0f;
    q[16] = rs.inv.x; q[17] = rs.inv.y; q[18] = rs.inv.z;
    q[19] = rs.swapXZ ? 1.0f : 0.0f;
    q[20] = rs.swapYZ ? 1.0f : 0.0f;
    q[21] = r1.swapXZ ? 1.0f : 0.0f;
    q[22] = r1.swapYZ ? 1.0f : 0.0f;
}

__global__ void sincos_kernel(uint32_t n, const float* theta, float* s, float* c)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float sn, cs;
    prt_sincosf(theta[i], &sn, &cs);
    s[i] = sn;
    c[i] = cs;
}

__global__ void powf_kernel(uint32_t n, const float* x, float* y)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = prt_powf_2p2(x[i]);
}

__global__ void camera_kernel(DevCamera cam, uint32_t x, uint32_t y, uint32_t state, float* out)
{
    uint32_t lane = threadIdx.x & 63u, slot = lane & 7u, gbase = lane & ~7u;
    uint32_t rng = state;
    DevRay r;
    Vec3 avg;
    camera_packet(cam, rng, x, y, slot, gbase, r, avg);
    if (lane < 8) {
        float* q = out + 11 * slot;
        q[0] = r.org.x; q[1] = r.org.y; q[2] = r.org.z;
        q[3] = r.dir.x; q[4] = r.dir.y; q[5] = r.dir.z;
        q[6] = r.inv.x; q[7] = r.inv.y; q[8] = r.inv.z;
        q[9] = r.swapXZ ? 1.0f : 0.0f;
        q[10] = r.swapYZ ? 1.0f : 0.0f;
    }
    if (lane == 0) {
        out[88] = avg.x; out[89] = avg.y; out[90] = avg.z;
        out[91] = asf(rng);
    }
}

#endif // PRT_TEST_ENTRY_POINTS

// ============================================================================ host side of the C-ABI
namespace {
thread_local std::string g_err;
} // namespace
int prt_fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}
const std::string& prt_last_error_string() { return g_err; }

namespace {

int fail(int code, const std::string& msg) { return prt_fail(code, msg); }

struct HVec3 { float x, y, z; };
inline HVec3 hsub(HVec3 a, HVec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline HVec3 hcross(HVec3 a, HVec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline HVec3 hnormalize(HVec3 v) // vecmath.h:1200 -- same operations as the device's normalize3
{
    float d = v.x * v.x + v.y * v.y + v.z * v.z;
    float invlen = 1.0f / sqrtf(d);
    return {invlen * v.x, invlen * v.y, invlen * v.z};
}
inline void hsafe_normalize2(float x, float y, float* ox, float* oy) // vecmath.h:1145
{
    float len = sqrtf(x * x + y * y);
    if (len < 0.00001f) { *ox = 0.0f; *oy = 0.0f; return; }
    float invlen = 1.0f / len;
    *ox = invlen * x;
    *oy = invlen * y;
}
inline float ubits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

} // namespace

template <typename T>
static int upload_vec(prt_hip_ctx* c, const std::vector<T>& v, const T** out)
{
    void* d = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 64);
    HIP_TRY(hipMalloc(&d, bytes));
    c->sceneAllocs.push_back(d);
    if (!v.empty()) HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)d;
    return PRT_HIP_OK;
}

extern "C" {

const char* prt_hip_last_error(void) { return g_err.c_str(); }

#ifndef PRT_SOURCE_SHA16
#define PRT_SOURCE_SHA16 "unstamped"
#endif
const char* prt_hip_source_sha16(void) { return PRT_SOURCE_SHA16; }

int prt_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int create_resources(prt_hip_ctx* c)
{
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    c->computeUnits = prop.multiProcessorCount;
    c->name = prop.name[0] ? prop.name : prop.gcnArchName;
    HIP_TRY(hipStreamCreate(&c->stream));
    HIP_TRY(hipEventCreateWithFlags(&c->evIn, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->evOut, hipEventDisableTiming));
    HIP_TRY(hipMalloc(&c->work, (PRT_WORK_WORDS + PRT_STICKY_WORDS) * sizeof(uint32_t)));
    HIP_TRY(hipMemset(c->work, 0, (PRT_WORK_WORDS + PRT_STICKY_WORDS) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&c->counters, PRT_STAT_SHARDS * PRT_STAT_STRIDE * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->counters, 0, PRT_STAT_SHARDS * PRT_STAT_STRIDE * sizeof(unsigned long long)));
    return PRT_HIP_OK;
}

// The library touches neither the process environment nor the HIP runtime's configuration: a render is one kernel on one
// stream of the context.
int prt_hip_create(int device, prt_hip_ctx** out)
{
    if (!out) return fail(PRT_HIP_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return fail(PRT_HIP_ENODEVICE, "no HIP device: libprt_hip has no CPU path (the GPU kernels are the product)");
    if (device < 0 || device >= n) return fail(PRT_HIP_EINVAL, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    prt_hip_ctx* c = new prt_hip_ctx();
    c->device = device;
    int rc = create_resources(c);
    if (rc != PRT_HIP_OK) {
        std::string why = prt_last_error_string(); // prt_hip_destroy makes HIP calls of its own
        prt_hip_destroy(c);      // frees whatever was created before the failure
        return fail(rc, why);
    }
    *out = c;
    return PRT_HIP_OK;
}

static void free_scene(prt_hip_ctx* c)
{
    for (void* p : c->sceneAllocs) (void)hipFree(p);
    c->sceneAllocs.clear();
    c->haveScene = false;
}

void prt_hip_destroy(prt_hip_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_scene(c);
    if (c->fb) (void)hipFree(c->fb);
    if (c->work) (void)hipFree(c->work);
    if (c->counters) (void)hipFree(c->counters);
    if (c->spill) (void)hipFree(c->spill);
    if (c->wfBuffer) (void)hipFree(c->wfBuffer);
    if (c->frameArgs) (void)hipFree(c->frameArgs);
    prt_gather_release(c);
    for (int k = 0; k < PRT_TIMING_RING; k++) {
        if (c->evT0[k]) (void)hipEventDestroy(c->evT0[k]);
        if (c->evT1[k]) (void)hipEventDestroy(c->evT1[k]);
    }
    if (c->evIn) (void)hipEventDestroy(c->evIn);
    if (c->evOut) (void)hipEventDestroy(c->evOut);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int prt_hip_device_info(prt_hip_ctx* c, char* name, size_t cap, int* computeUnits)
{
    if (!c) return fail(PRT_HIP_EINVAL, "ctx is NULL");
    if (name && cap) {
        strncpy(name, c->name.c_str(), cap - 1);
        name[cap - 1] = 0;
    }
    if (computeUnits) *computeUnits = c->computeUnits;
    return PRT_HIP_OK;
}

// Where the triangles of a mesh's leaves lie in the device arrays.  The reference keeps them in primRemapping order (leaf after leaf,
// depth first); a leaf of n triangles is n * 36 contiguous bytes here, fetched by the n pair lanes of a cooperative leaf round at
// once.  Beyond the caches such a fetch costs one DRAM row activation per 128-byte LINE it touches, whatever part of the line it
// reads (profiles/r03_rec_gather.txt), and a leaf that starts at an arbitrary multiple of 36 bytes touches more lines than
// ceil(36 n / 128) (C4's tree: 2.64 per leaf where 2.15 would do).  A leaf reference carries its first slot, so a leaf may start
// up to PRT_LEAF_PADS unused slots later when that saves it a line (two pads: 2.22 lines per leaf for 12 % more slots).  Results
// are unchanged: a hit's slot index is internal, and the shade, bump, alpha and primId records follow the slots.
// slotOf[k] = slot (within the mesh) of leaf-order index k; returns the number of slots (primCount when nothing is padded).
#ifndef PRT_LEAF_PADS
#define PRT_LEAF_PADS 2 // (C5 share 1480 -> 1408 ms, C4 834 -> 816, C3 unchanged; 0: 2.64, 1: 2.39, 2: 2.22, 3: 2.20 lines per leaf on C4's tree)
#endif
static uint32_t leaf_slots(const prt_mesh_desc& md, uint32_t triBase, std::vector<uint32_t>& slotOf)
{
    slotOf.resize(md.primCount);
    for (uint32_t k = 0; k < md.primCount; k++) slotOf[k] = k;
    if (PRT_LEAF_PADS <= 0) return md.primCount;
    std::vector<uint8_t> covered(md.primCount, 0);
    for (uint32_t i = 0; i < md.nodeCount; i++) {
        const prt_bvh_node& n = md.nodes[i];
        if (n.primCount == 0xf) continue;
        for (uint32_t t = 0; t < n.primCount; t++) {
            if (covered[n.primOrSecondNodeIndex + t]) return md.primCount; // two leaves share a triangle: keep the reference's layout
            covered[n.primOrSecondNodeIndex + t] = 1;
        }
    }
    for (uint32_t k = 0; k < md.primCount; k++)
        if (!covered[k]) return md.primCount;
    auto excess = [&](uint32_t slot, uint32_t n) { // lines touched from this slot beyond the fewest a leaf of n triangles can touch
        const uint32_t o = (uint32_t)(((uint64_t)(triBase + slot) * 36u) & 127u);
        return (o + 36u * n + 127u) / 128u - (36u * n + 127u) / 128u;
    };
    // leaves in the order of their first triangle (= depth-first order for the reference's builder)
    std::vector<std::pair<uint32_t, uint32_t>> leaves;
    for (uint32_t i = 0; i < md.nodeCount; i++)
        if (md.nodes[i].primCount != 0xf) leaves.push_back({md.nodes[i].primOrSecondNodeIndex, md.nodes[i].primCount});
    std::sort(leaves.begin(), leaves.end());
    uint32_t cursor = 0;
    for (const auto& L : leaves) {
        uint32_t best = 0, bestEx = excess(cursor, L.second);
        for (uint32_t p = 1; p <= (uint32_t)PRT_LEAF_PADS && bestEx != 0u; p++) {
            const uint32_t e = excess(cursor + p, L.second);
            if (e < bestEx) {
                best = p;
                bestEx = e;
            }
        }
        cursor += best;
        for (uint32_t t = 0; t < L.second; t++) slotOf[L.first + t] = cursor + t;
        cursor += L.second;
    }
    return cursor;
}

// Flattens Scene -> Bvh -> Mesh (scene.h:61-71, bvh.h:113-119, mesh.h:87-104) into the arrays of DevScene.
int prt_hip_upload_scene(prt_hip_ctx* c, const prt_scene_desc* s)
{
    if (!c || !s) return fail(PRT_HIP_EINVAL, "NULL argument");
    if (s->meshCount == 0 || s->meshCount > PRT_MAX_BVH) return fail(PRT_HIP_EINVAL, "meshCount must be 1..8");
    HIP_TRY(hipSetDevice(c->device));
    free_scene(c);

    std::vector<float4> wnodes, shade, bump, mats, alpha;
    std::vector<float> tris;                 // 9 floats per triangle, leaf order
    std::vector<uint32_t> triAlpha, triPrim; // per triangle, leaf order
    std::vector<uint32_t> alphaClass;        // 2 bits per bilinear cell of every alpha-tested texture (prt_device.h DevScene::alphaClass)
    std::vector<uint32_t> classWordOf;       // per texture: first word of its cell classes in alphaClass, 0xffffffff = not built yet
    std::vector<uint4> texDesc;
    std::vector<uint8_t> texels;
    DevScene sc{};
    sc.bvhCount = s->meshCount;

    for (uint32_t t = 0; t < s->textureCount; t++) {
        const prt_texture_desc& td = s->textures[t];
        if (td.width <= 0 || td.height <= 0 || td.component <= 0 || !td.texels) return fail(PRT_HIP_EINVAL, "bad texture");
        size_t off = (texels.size() + 15) & ~(size_t)15;
        size_t sz = (size_t)td.width * td.height * td.component;
        texels.resize(off + sz + 16, 0);
        memcpy(&texels[off], td.texels, sz);
        texDesc.push_back(make_uint4((uint32_t)off, (uint32_t)td.width, (uint32_t)td.height, (uint32_t)td.component));
    }

    classWordOf.assign(s->textureCount, 0xffffffffu);
    // Cell classes of texture t (built when the first alpha-tested material names it).  The tap of Texture::testAlpha at a uv in cell
    // (x0, y0) blends the alpha bytes of (x0, y0), (x1, y0), (x0, y1), (x1, y1), x1 = min(x0 + 1, w - 1), with weights >= 0 that sum to 1
    // up to rounding (texture.cpp:31-100): four bytes >= 128 give more than 127 whatever the weights, four bytes <= 126 give less
    // (the blend is off 128 x sum(k) by < 1e-4); a cell with a byte of 127, or with bytes on both sides, is left to the blend itself.
    auto classWord = [&](uint32_t t) -> uint32_t {
        if (classWordOf[t] != 0xffffffffu) return classWordOf[t];
        const uint4 d = texDesc[t];
        const int32_t w = (int32_t)d.y, h = (int32_t)d.z, comp = (int32_t)d.w;
        const uint8_t* px = texels.data() + d.x;
        const uint32_t first = (uint32_t)alphaClass.size();
        alphaClass.resize(first + ((size_t)w * h + 15) / 16, 0u);
        for (int32_t y0 = 0; y0 < h; y0++) {
            const int32_t y1 = (y0 + 1 < h - 1) ? y0 + 1 : h - 1;
            for (int32_t x0 = 0; x0 < w; x0++) {
                const int32_t x1 = (x0 + 1 < w - 1) ? x0 + 1 : w - 1;
                const uint32_t a[4] = {px[comp * (x0 + y0 * w) + 3], px[comp * (x1 + y0 * w) + 3], px[comp * (x0 + y1 * w) + 3], px[comp * (x1 + y1 * w) + 3]};
                const uint32_t lo = std::min(std::min(a[0], a[1]), std::min(a[2], a[3])), hi = std::max(std::max(a[0], a[1]), std::max(a[2], a[3]));
                const uint32_t cls = lo >= 128u ? 1u : (hi <= 126u ? 2u : 0u);
                const uint32_t cell = (uint32_t)x0 + (uint32_t)y0 * (uint32_t)w;
                alphaClass[first + (cell >> 4)] |= cls << ((cell & 15u) * 2u);
            }
        }
        return classWordOf[t] = first;
    };

    bool anyBump = false;
    for (uint32_t m = 0; m < s->meshCount; m++)
        for (uint32_t k = 0; k < s->meshes[m].materialCount; k++)
            if (s->meshes[m].materials[k].bumpMap >= 0) anyBump = true;

    for (uint32_t m = 0; m < s->meshCount; m++) {
        const prt_mesh_desc& md = s->meshes[m];
        if (!md.nodes || !md.primRemapping || !md.indices || !md.positions || !md.primMaterial || !md.materials || md.nodeCount == 0)
            return fail(PRT_HIP_EINVAL, "mesh descriptor has NULL arrays");
        const uint32_t triBase = (uint32_t)(tris.size() / 9);
        const uint32_t primBase = (uint32_t)(shade.size() / 4), matBase = (uint32_t)(mats.size() / PRT_MAT_STRIDE);
        sc.primBase[m] = primBase;
        sc.hasNormals[m] = md.normals ? 1u : 0u;
        auto P = [&](uint32_t v) { return HVec3{md.positions[3 * v], md.positions[3 * v + 1], md.positions[3 * v + 2]}; };
        for (uint32_t k = 0; k < md.materialCount; k++) {
            const prt_material& mt = md.materials[k];
            if (mt.diffuseMap >= (int32_t)s->textureCount || mt.bumpMap >= (int32_t)s->textureCount)
                return fail(PRT_HIP_EINVAL, "material texture index out of range");
            if (mt.alphaTest && mt.diffuseMap < 0) return fail(PRT_HIP_EINVAL, "alphaTest material without a diffuse map");
            // ONE place writes a material record and its size is tied to the stride the kernels index with (sample_diffuse,
            // sample_bump, shade_kernel: sc.mats + PRT_MAT_STRIDE * material).  A record count and an index stride that
            // disagree read another material's fields as texture descriptors -- a wild texel address on the device.
            const uint4 dd = mt.diffuseMap >= 0 ? texDesc[mt.diffuseMap] : make_uint4(0, 0, 0, 0);
            const uint4 bd = mt.bumpMap >= 0 ? texDesc[mt.bumpMap] : make_uint4(0, 0, 0, 0);
            const float4 record[] = {
                make_float4(mt.diffuse[0], mt.diffuse[1], mt.diffuse[2], ubits(mt.reflectionType)),
                make_float4(mt.emissive[0], mt.emissive[1], mt.emissive[2], ubits(mt.alphaTest)),
                make_float4(ubits((uint32_t)mt.diffuseMap), ubits((uint32_t)mt.bumpMap), 0.0f, 0.0f),
                make_float4(ubits(dd.x), ubits(dd.y), ubits(dd.z), ubits(dd.w)),
                make_float4(ubits(bd.x), ubits(bd.y), ubits(bd.z), ubits(bd.w)),
            };
            static_assert(sizeof(record) / sizeof(record[0]) == PRT_MAT_STRIDE, "material record size and PRT_MAT_STRIDE must agree");
            mats.insert(mats.end(), record, record + PRT_MAT_STRIDE);
        }
        std::vector<uint32_t> slotOf, kOfSlot; // leaf-order index <-> slot in the device arrays (leaf_slots)
        uint32_t slotCount = md.primCount;
        // Wide records: one per internal node, in the reference's DFS order.  wideIndex[i] = record of node i.
        {
            std::vector<uint32_t> wideIndex(md.nodeCount, 0);
            uint32_t nextWide = (uint32_t)(wnodes.size() / 4);
            for (uint32_t i = 0; i < md.nodeCount; i++) {
                const prt_bvh_node& n = md.nodes[i];
                if (n.primCount == 0xf) {
                    if (n.primOrSecondNodeIndex >= md.nodeCount || n.primOrSecondNodeIndex <= i || i + 1 >= md.nodeCount)
                        return fail(PRT_HIP_EINVAL, "bad child index");
                    wideIndex[i] = nextWide++;
                } else if (n.primCount == 0 || n.primCount > 8 || n.primOrSecondNodeIndex + n.primCount > md.primCount) {
                    return fail(PRT_HIP_EINVAL, "bad leaf range");
                }
            }
            if ((size_t)triBase + md.primCount >= (1u << PRT_COOP_TRI_BITS) || nextWide >= (1u << 30)) return fail(PRT_HIP_EINVAL, "scene too large for 32-bit child references"); // (a pair-table word holds a triangle index in 26 bits)
            slotCount = leaf_slots(md, triBase, slotOf);
            if ((size_t)triBase + slotCount >= (1u << PRT_COOP_TRI_BITS)) return fail(PRT_HIP_EINVAL, "scene too large for 32-bit child references");
            kOfSlot.assign(slotCount, 0xffffffffu); // (an unused slot between two leaves: no leaf reference reaches it)
            for (uint32_t k = 0; k < md.primCount; k++) kOfSlot[slotOf[k]] = k;
            auto refOf = [&](uint32_t i) -> uint32_t {
                const prt_bvh_node& n = md.nodes[i];
                if (n.primCount == 0xf) return wideIndex[i];
                bool anyAlpha = false; // some triangle of the leaf is alpha-tested: its candidates look their alpha record up
                for (uint32_t k = 0; k < n.primCount; k++) {
                    const uint32_t prim = md.primRemapping[n.primOrSecondNodeIndex + k];
                    if (prim < md.primCount && md.primMaterial[prim] < md.materialCount && md.materials[md.primMaterial[prim]].alphaTest) anyAlpha = true;
                }
                return PRT_REF_LEAF | ((triBase + slotOf[n.primOrSecondNodeIndex]) << 4) | (anyAlpha ? PRT_LEAF_ALPHA : 0u) | (n.primCount - 1u);
            };
            for (uint32_t i = 0; i < md.nodeCount; i++) {
                const prt_bvh_node& n = md.nodes[i];
                if (n.primCount != 0xf) continue;
                const prt_bvh_node& c0 = md.nodes[i + 1];
                const prt_bvh_node& c1 = md.nodes[n.primOrSecondNodeIndex];
                wnodes.push_back(make_float4(c0.lower[0], c0.upper[0], c0.lower[1], c0.upper[1])); // x and y of child 0
                wnodes.push_back(make_float4(c0.lower[2], c0.upper[2], c1.lower[2], c1.upper[2])); // z of both children
                wnodes.push_back(make_float4(c1.lower[0], c1.upper[0], c1.lower[1], c1.upper[1])); // x and y of child 1
                wnodes.push_back(make_float4(ubits(refOf(i + 1)), ubits(refOf(n.primOrSecondNodeIndex)), ubits(n.splitAxis & 3u), 0.0f));
            }
            sc.rootRef[m] = refOf(0);
            memcpy(&sc.rootBox[m][0], md.nodes[0].lower, 12);
            memcpy(&sc.rootBox[m][3], md.nodes[0].upper, 12);
        }
        // leaf triangles in primRemapping order (TriangleVector, bvh.cpp:245-296), leaf blocks placed by leaf_slots
        for (uint32_t slot = 0; slot < slotCount; slot++) {
            if (kOfSlot[slot] == 0xffffffffu) {
                tris.insert(tris.end(), 9, 0.0f);
                triAlpha.push_back(0u);
                triPrim.push_back(0u);
                continue;
            }
            uint32_t prim = md.primRemapping[kOfSlot[slot]];
            if (prim >= md.primCount) return fail(PRT_HIP_EINVAL, "bad primRemapping");
            uint32_t v0 = md.indices[3 * prim], v1 = md.indices[3 * prim + 1], v2 = md.indices[3 * prim + 2];
            if (v0 >= md.vertexCount || v1 >= md.vertexCount || v2 >= md.vertexCount) return fail(PRT_HIP_EINVAL, "bad vertex index");
            if (md.primMaterial[prim] >= md.materialCount) return fail(PRT_HIP_EINVAL, "bad material index");
            const prt_material& mt = md.materials[md.primMaterial[prim]];
            uint32_t alphaRef = 0;
            if (mt.alphaTest) {
                // leaf uv are the mesh texcoord buffer by vertex index (bvh.cpp:266-269), zero if there is none
                float u[6] = {0, 0, 0, 0, 0, 0};
                if (md.texcoords) {
                    u[0] = md.texcoords[2 * v0]; u[1] = md.texcoords[2 * v0 + 1];
                    u[2] = md.texcoords[2 * v1]; u[3] = md.texcoords[2 * v1 + 1];
                    u[4] = md.texcoords[2 * v2]; u[5] = md.texcoords[2 * v2 + 1];
                }
                const uint4 ad = texDesc[mt.diffuseMap];
                alpha.push_back(make_float4(u[0], u[1], u[2], u[3]));
                alpha.push_back(make_float4(u[4], u[5], ubits((uint32_t)mt.diffuseMap), ubits(classWord((uint32_t)mt.diffuseMap))));
                alpha.push_back(make_float4(ubits(ad.x), ubits(ad.y), ubits(ad.z), ubits(ad.w)));
                alphaRef = (uint32_t)(alpha.size() / 3);
            }
            HVec3 p0 = P(v0), p1 = P(v1), p2 = P(v2);
            const float corners[9] = {p0.x, p0.y, p0.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z};
            tris.insert(tris.end(), corners, corners + 9);
            triAlpha.push_back(alphaRef);
            triPrim.push_back(prim);
        }
        // shading records (Mesh::getSurfaceProperties, mesh.cpp:311-364) in LEAF order, like the triangles: a hit names its
        // triangle by that index
        for (uint32_t slot = 0; slot < slotCount; slot++) {
            if (kOfSlot[slot] == 0xffffffffu) {
                shade.insert(shade.end(), 4, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
                if (anyBump) bump.insert(bump.end(), 3, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
                continue;
            }
            const uint32_t prim = md.primRemapping[kOfSlot[slot]];
            uint32_t v0 = md.indices[3 * prim], v1 = md.indices[3 * prim + 1], v2 = md.indices[3 * prim + 2];
            HVec3 p0 = P(v0), p1 = P(v1), p2 = P(v2);
            HVec3 n0, n1{0, 0, 0}, n2{0, 0, 0};
            if (md.normals) {
                n0 = {md.normals[3 * v0], md.normals[3 * v0 + 1], md.normals[3 * v0 + 2]};
                n1 = {md.normals[3 * v1], md.normals[3 * v1 + 1], md.normals[3 * v1 + 2]};
                n2 = {md.normals[3 * v2], md.normals[3 * v2 + 1], md.normals[3 * v2 + 2]};
            } else {
                n0 = hnormalize(hcross(hsub(p1, p0), hsub(p2, p0))); // mesh.cpp:335
            }
            float t[6] = {0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 1.0f}; // mesh.cpp:351-353
            if (md.texcoords) {
                t[0] = md.texcoords[2 * v0]; t[1] = md.texcoords[2 * v0 + 1];
                t[2] = md.texcoords[2 * v1]; t[3] = md.texcoords[2 * v1 + 1];
                t[4] = md.texcoords[2 * v2]; t[5] = md.texcoords[2 * v2 + 1];
            }
            shade.push_back(make_float4(n0.x, n0.y, n0.z, ubits(matBase + md.primMaterial[prim])));
            shade.push_back(make_float4(n1.x, n1.y, n1.z, t[0]));
            shade.push_back(make_float4(n2.x, n2.y, n2.z, t[1]));
            shade.push_back(make_float4(t[2], t[3], t[4], t[5]));
            if (anyBump) {
                HVec3 dp01 = hnormalize(hsub(p1, p0)), dp02 = hnormalize(hsub(p2, p0)); // mesh.cpp:360-361
                float d01x, d01y, d02x, d02y;
                hsafe_normalize2(t[2] - t[0], t[3] - t[1], &d01x, &d01y);
                hsafe_normalize2(t[4] - t[0], t[5] - t[1], &d02x, &d02y);
                bump.push_back(make_float4(dp01.x, dp01.y, dp01.z, d01x));
                bump.push_back(make_float4(dp02.x, dp02.y, dp02.z, d01y));
                bump.push_back(make_float4(d02x, d02y, 0.0f, 0.0f));
            }
        }
    }
    // The PRT_HOT_NODES records nearest to the roots, breadth first over the BVHs in order; references to them become
    // PRT_REF_HOT | slot everywhere (parents' records, root references), and `hot` holds copies of the rewritten records.
    std::vector<float4> hot((size_t)PRT_HOT_NODES * 4, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    {
        auto bitsOf = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
        std::vector<uint32_t> order; // record indices, breadth first
        for (uint32_t m = 0; m < sc.bvhCount; m++)
            if (!(sc.rootRef[m] & PRT_REF_LEAF)) order.push_back(sc.rootRef[m]);
        for (size_t head = 0; head < order.size() && order.size() < PRT_HOT_NODES; head++) {
            const float4& refs = wnodes[(size_t)order[head] * 4 + 3];
            for (uint32_t r : {bitsOf(refs.x), bitsOf(refs.y)})
                if (!(r & PRT_REF_LEAF) && order.size() < PRT_HOT_NODES) order.push_back(r);
        }
        std::vector<uint32_t> slotOf(wnodes.size() / 4, 0xffffffffu);
        for (size_t k = 0; k < order.size(); k++) slotOf[order[k]] = (uint32_t)k;
        auto hotRef = [&](uint32_t r) { return (!(r & PRT_REF_LEAF) && slotOf[r] != 0xffffffffu) ? (PRT_REF_HOT | slotOf[r]) : r; };
        for (size_t rec = 0; rec < wnodes.size() / 4; rec++) {
            float4& refs = wnodes[rec * 4 + 3];
            refs.x = ubits(hotRef(bitsOf(refs.x)));
            refs.y = ubits(hotRef(bitsOf(refs.y)));
        }
        for (uint32_t m = 0; m < sc.bvhCount; m++) sc.rootRef[m] = hotRef(sc.rootRef[m]);
        for (size_t k = 0; k < order.size(); k++) memcpy(&hot[k * 4], &wnodes[(size_t)order[k] * 4], 4 * sizeof(float4));
    }
    sc.hasLight = s->hasDirectionalLight ? 1u : 0u;
    memcpy(sc.lightDir, s->lightDir, 12);
    memcpy(sc.lightIntensity, s->lightIntensity, 12);
    sc.radius = s->radius;

    int rc;
    // InfiniteAreaLight: texels + CDF tables as they are, plus the "first index that differs from its predecessor" constants
    // of the bisection (prt_device.h cdf_find).  The bisection needs non-decreasing tables: running sums of non-negative terms
    // are, unless the image holds negative, infinite or NaN radiance -- refuse those.
    sc.hasEnv = 0;
    if (s->hasInfiniteAreaLight) {
        const int32_t W = s->envWidth, H = s->envHeight;
        if (W <= 0 || H <= 0 || !s->envTexels || !s->envVerticalP || !s->envHorizontalP) return fail(PRT_HIP_EINVAL, "incomplete environment light");
        if ((int64_t)W * H > (1 << 28)) return fail(PRT_HIP_EINVAL, "environment map too large");
        auto firstStep = [](const float* cdf, int32_t n) {
            for (int32_t i = 1; i < n; i++) {
                float pdf = cdf[i] - cdf[i - 1];
                if (!(pdf == 0.0f)) return i; // light.cpp:96-98, 112-114
            }
            return n;
        };
        auto monotone = [](const float* cdf, int32_t n) {
            for (int32_t i = 1; i < n; i++)
                if (!(cdf[i] >= cdf[i - 1])) return false;
            return true;
        };
        if (!monotone(s->envVerticalP, H)) return fail(PRT_HIP_EINVAL, "environment light: vertical CDF is not non-decreasing (negative or non-finite radiance?)");
        std::vector<int32_t> firstX((size_t)H);
        for (int32_t y = 0; y < H; y++) {
            const float* row = s->envHorizontalP + (size_t)y * W;
            const bool allNaN = row[0] != row[0]; // an all-black row: 0 * inf (light.cpp:63-70); never selected, never exceeds u
            if (allNaN) {
                for (int32_t x = 0; x < W; x++)
                    if (row[x] == row[x]) return fail(PRT_HIP_EINVAL, "environment light: partly NaN CDF row");
            } else if (!monotone(row, W)) {
                return fail(PRT_HIP_EINVAL, "environment light: horizontal CDF is not non-decreasing (negative or non-finite radiance?)");
            }
            firstX[y] = firstStep(row, W);
        }
        std::vector<float4> tex((size_t)W * H);
        memcpy(tex.data(), s->envTexels, tex.size() * sizeof(float4));
        std::vector<float> vp(s->envVerticalP, s->envVerticalP + H), hp(s->envHorizontalP, s->envHorizontalP + (size_t)W * H);
        if ((rc = upload_vec(c, tex, &sc.envTexels))) return rc;
        if ((rc = upload_vec(c, vp, &sc.envV))) return rc;
        if ((rc = upload_vec(c, hp, &sc.envHor))) return rc;
        if ((rc = upload_vec(c, firstX, &sc.envFirstX))) return rc;
        sc.envW = W;
        sc.envH = H;
        sc.envFirstY = firstStep(s->envVerticalP, H);
        sc.hasEnv = 1;
    }
    // every texture descriptor a kernel can reach lies inside the texel array (checked here, on the host, once per upload)
    for (const uint4& d : texDesc)
        if ((size_t)d.x + (size_t)d.y * d.z * d.w > texels.size()) return fail(PRT_HIP_EINVAL, "internal: texture descriptor outside the texel array");
    for (size_t k = 0; k + PRT_MAT_STRIDE <= mats.size(); k += PRT_MAT_STRIDE)
        for (int j = 3; j <= 4; j++) {
            const float4& f = mats[k + j];
            uint32_t off, w, h, comp;
            memcpy(&off, &f.x, 4); memcpy(&w, &f.y, 4); memcpy(&h, &f.z, 4); memcpy(&comp, &f.w, 4);
            if ((size_t)off + (size_t)w * h * comp > texels.size()) return fail(PRT_HIP_EINVAL, "internal: material map descriptor outside the texel array");
        }
    if (mats.size() % PRT_MAT_STRIDE != 0) return fail(PRT_HIP_EINVAL, "internal: material table is not a whole number of records");
    if ((rc = upload_vec(c, wnodes, &sc.wnodes))) return rc;
    if ((rc = upload_vec(c, hot, &sc.hotNodes))) return rc;
    if ((rc = upload_vec(c, tris, &sc.tris))) return rc;
    if ((rc = upload_vec(c, triAlpha, &sc.triAlpha))) return rc;
    if ((rc = upload_vec(c, triPrim, &sc.triPrim))) return rc;
    if ((rc = upload_vec(c, shade, &sc.shade))) return rc;
    if ((rc = upload_vec(c, bump, &sc.bump))) return rc;
    if ((rc = upload_vec(c, mats, &sc.mats))) return rc;
    if ((rc = upload_vec(c, alpha, &sc.alpha))) return rc;
    if ((rc = upload_vec(c, alphaClass, &sc.alphaClass))) return rc;
    if ((rc = upload_vec(c, texDesc, &sc.texDesc))) return rc;
    if ((rc = upload_vec(c, texels, &sc.texels))) return rc;
    c->sc = sc;
    c->haveScene = true;
    return PRT_HIP_OK;
}

int prt_hip_set_camera(prt_hip_ctx* c, const prt_camera_desc* cam)
{
    if (!c || !cam) return fail(PRT_HIP_EINVAL, "NULL argument");
    if (cam->width == 0 || cam->height == 0) return fail(PRT_HIP_EINVAL, "empty image");
    static_assert(sizeof(DevCamera) == sizeof(prt_camera_desc), "camera layouts must match");
    memcpy(&c->cam, cam, sizeof(DevCamera));
    c->haveCamera = true;
    return PRT_HIP_OK;
}

// Per-thread spill columns of the traversal stacks (entries beyond those kept in LDS): [entry][thread], two words per entry.
static int ensure_launch_resources(prt_hip_ctx* c, uint32_t blocks)
{
    uint32_t threads = blocks * PRT_BLOCK;
    if (threads > c->spillThreads) {
        if (c->spill) (void)hipFree(c->spill);
        c->spill = nullptr;
        HIP_TRY(hipMalloc(&c->spill, (size_t)threads * 2 * (PRT_STACK_MAX - PRT_STACK_LDS_PACKET) * sizeof(uint32_t)));
        c->spillThreads = threads;
    }
    return PRT_HIP_OK;
}

// ---- frame kernel (prt_frame.h): resident blocks, pool state carved out of one allocation, one launch per render
static int frame_blocks(prt_hip_ctx* c)
{
    if (c->frameBlocksPerCU == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, frame_kernel<false, false>, PRT_BLOCK, 0) != hipSuccess || nb <= 0) nb = 4;
        // No block waits for another one, so a block the hardware admits later than the query says only starts later.
        c->frameBlocksPerCU = std::min(nb, 8);
#ifdef PRT_TUNING_ENV // tuning builds only (tools/build_variants.py): the product reads no tuning variable
        if (const char* e = getenv("PRT_FRAME_BPC")) c->frameBlocksPerCU = std::max(1, std::min(8, atoi(e)));
#endif
    }
    return c->computeUnits * c->frameBlocksPerCU;
}

static int frame_layout(prt_hip_ctx* c, uint32_t blocks, bool env, FrameArgs& A)
{
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t groups = (size_t)blocks * PRT_POOL_GROUPS, slots = groups * 8;
    size_t need = 3 * al(groups * sizeof(uint32_t)) + al(groups * sizeof(float4));
    need += (env ? 7 : 5) * al(slots * sizeof(float4));
    need += al(slots * sizeof(float4)) + al(slots * sizeof(uint2)) + al(slots * sizeof(uint32_t));
    need += al((size_t)blocks * Q_COUNT * PRT_POOL_SLOTS * sizeof(uint32_t));
    if (need > c->wfBytes) {
        if (c->wfBuffer) (void)hipFree(c->wfBuffer);
        c->wfBuffer = nullptr;
        c->wfBytes = 0;
        HIP_TRY(hipMalloc(&c->wfBuffer, need));
        c->wfBytes = need;
    }
    char* p = (char*)c->wfBuffer;
    auto take = [&](size_t bytes) { char* r = p; p += al(bytes); return r; };
    A.gRng = (uint32_t*)take(groups * sizeof(uint32_t));
    A.gInfo = (uint32_t*)take(groups * sizeof(uint32_t));
    A.gPixel = (uint32_t*)take(groups * sizeof(uint32_t));
    A.gColor = (float4*)take(groups * sizeof(float4));
    A.S0 = (float4*)take(slots * sizeof(float4));
    A.S1 = (float4*)take(slots * sizeof(float4));
    A.S2 = (float4*)take(slots * sizeof(float4));
    A.S3 = (float4*)take(slots * sizeof(float4));
    A.S4 = (float4*)take(slots * sizeof(float4));
    A.S5 = env ? (float4*)take(slots * sizeof(float4)) : nullptr;
    A.S6 = env ? (float4*)take(slots * sizeof(float4)) : nullptr;
    A.hitA = (float4*)take(slots * sizeof(float4));
    A.hitB = (uint2*)take(slots * sizeof(uint2));
    A.occl = (uint32_t*)take(slots * sizeof(uint32_t));
    A.qE = (uint32_t*)take((size_t)blocks * Q_COUNT * PRT_POOL_SLOTS * sizeof(uint32_t));
    return PRT_HIP_OK;
}

static int render_frame_kernel(prt_hip_ctx* c, FrameArgs& A, uint64_t totalWork, hipStream_t s)
{
    const prt_render_params* p = &A.p;
    A.totalWork = (uint32_t)totalWork;
    A.totalChunks = (uint32_t)((totalWork + PRT_CHUNK - 1) / PRT_CHUNK);
    A.counters = c->counters;
    A.ctrl = c->work;
    const uint32_t resident = (uint32_t)frame_blocks(c);
    const uint32_t blocks = std::max<uint32_t>(1, std::min<uint32_t>(resident, A.totalChunks));
    // a block holds at most rowsPerBlock rows at a time and takes one row per visit to the shade role: all of a small launch's
    // rows are in flight at once (a row that starts late costs a whole chain of rounds), spread evenly over the blocks
    A.rowsPerBlock = std::max<uint32_t>(1, std::min<uint32_t>(PRT_POOL_CHUNKS, (A.totalChunks + blocks - 1) / blocks));
    A.spreadRows = (A.totalChunks < (uint64_t)blocks * PRT_POOL_CHUNKS * 2 && totalWork % PRT_CHUNK == 0) ? 1u : 0u;
#ifdef PRT_TUNING_ENV
    if (const char* e = getenv("PRT_SPREAD")) A.spreadRows = atoi(e) && totalWork % PRT_CHUNK == 0;
    if (const char* e = getenv("PRT_ROWS")) A.rowsPerBlock = std::max<uint32_t>(1, std::min<uint32_t>(PRT_POOL_CHUNKS, (uint32_t)atoi(e)));
#endif
    int rc = ensure_launch_resources(c, resident);
    if (rc) return rc;
    A.spill = c->spill;
    A.spillStride = c->spillThreads;
    if ((rc = frame_layout(c, resident, c->sc.hasEnv != 0, A))) return rc;
    HIP_TRY(hipMemsetAsync(c->work, 0, PRT_WORK_WORDS * sizeof(uint32_t), s));
    if (A.totalChunks == 0) return PRT_HIP_OK;
    const FrameArgs& dA = A;
    const bool env = c->sc.hasEnv != 0;
    if (p->countTraffic) {
        if (env) hipLaunchKernelGGL((frame_kernel<true, true>), dim3(blocks), dim3(PRT_BLOCK), 0, s, dA);
        else hipLaunchKernelGGL((frame_kernel<true, false>), dim3(blocks), dim3(PRT_BLOCK), 0, s, dA);
    } else {
        if (env) hipLaunchKernelGGL((frame_kernel<false, true>), dim3(blocks), dim3(PRT_BLOCK), 0, s, dA);
        else hipLaunchKernelGGL((frame_kernel<false, false>), dim3(blocks), dim3(PRT_BLOCK), 0, s, dA);
    }
    return PRT_HIP_OK;
}

// Folds the recorded launches of the timing ring into the context's totals (waits for them: they were queued long ago).
static void fold_timing(prt_hip_ctx* c)
{
    for (uint32_t i = 0; i < c->ringUsed; i++) {
        float ms = 0.0f;
        if (hipEventSynchronize(c->evT1[i]) == hipSuccess && hipEventElapsedTime(&ms, c->evT0[i], c->evT1[i]) == hipSuccess) {
            c->lastMs = ms;
            c->accMs += ms;
            c->accLaunches++;
        }
    }
    c->ringUsed = 0;
}

int prt_hip_render(prt_hip_ctx* c, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, const prt_render_params* p, float* d_rgb,
                   void* stream)
{
    if (!c || !p) return fail(PRT_HIP_EINVAL, "NULL argument");
    if (!c->haveScene || !c->haveCamera) return fail(PRT_HIP_ESTATE, "upload a scene and set a camera first");
    const uint32_t W = c->cam.width, H = c->cam.height;
    if (x1 < x0 || y1 < y0 || x1 >= W || y1 >= H) return fail(PRT_HIP_EINVAL, "pixel rectangle outside the image");
    if (p->samples == 0 || p->tileSize == 0 || p->nranks == 0 || p->rank >= p->nranks) return fail(PRT_HIP_EINVAL, "bad render params");
    if (W > 65535 || H > 65535 || p->samples / 8 > 255 || p->maxDepth > 255) return fail(PRT_HIP_EINVAL, "image, sample count or depth too large");
    HIP_TRY(hipSetDevice(c->device));
    // The frame kernel runs on the context's own stream; a caller's stream is ordered around it with two events: work
    // queued on it before this call is finished before the kernel starts, and whatever the caller queues next waits for it.
    hipStream_t s = c->stream;
    hipStream_t caller = (stream && (hipStream_t)stream != c->stream) ? (hipStream_t)stream : nullptr;
    if (caller) {
        HIP_TRY(hipEventRecord(c->evIn, caller));
        HIP_TRY(hipStreamWaitEvent(s, c->evIn, 0));
    }
    if (!d_rgb) {
        if (c->fbPixels != (size_t)W * H) {
            if (c->fb) (void)hipFree(c->fb);
            c->fb = nullptr;
            HIP_TRY(hipMalloc(&c->fb, (size_t)W * H * 3 * sizeof(float)));
            HIP_TRY(hipMemsetAsync(c->fb, 0, (size_t)W * H * 3 * sizeof(float), s));
            c->fbPixels = (size_t)W * H;
        }
        d_rgb = c->fb;
    }
    FrameArgs A{};
    A.sc = c->sc;
    A.cam = c->cam;
    A.p = *p;
    A.x0 = x0; A.y0 = y0; A.x1 = x1; A.y1 = y1;
    const uint32_t T = p->tileSize;
    if ((uint64_t)T * T > (1u << 20)) return fail(PRT_HIP_EINVAL, "tile too large");
    A.tilesXImage = (W + T - 1) / T;
    A.rtx0 = x0 / T; A.rty0 = y0 / T;
    A.rtnx = x1 / T - A.rtx0 + 1;
    A.rtny = y1 / T - A.rty0 + 1;
    A.fullWidth = (x0 == 0 && x1 == W - 1) ? 1u : 0u;
    const uint32_t tilesInRect = A.rtnx * A.rtny;
    uint64_t totalWork;
    if (A.fullWidth) {
        // tile ids rty0*TX .. (rty0+rtny)*TX - 1 are contiguous; this rank owns ids == rank (mod nranks)
        uint32_t lo = A.rty0 * A.tilesXImage, hi = lo + tilesInRect;
        uint32_t first = lo + ((p->rank + p->nranks - lo % p->nranks) % p->nranks);
        A.firstOwned = first;
        uint32_t owned = first < hi ? (hi - first + p->nranks - 1) / p->nranks : 0;
        totalWork = (uint64_t)owned * T * T;
    } else {
        totalWork = (uint64_t)tilesInRect * T * T;
    }
    if (totalWork > 0xffffffffull) return fail(PRT_HIP_EINVAL, "rectangle too large");
    A.rgb = d_rgb;
    HIP_TRY(hipMemsetAsync(c->counters, 0, PRT_STAT_SHARDS * PRT_STAT_STRIDE * sizeof(unsigned long long), s));
    if (c->ringUsed == PRT_TIMING_RING) fold_timing(c);
    if (!c->evT0[c->ringUsed]) HIP_TRY(hipEventCreate(&c->evT0[c->ringUsed]));
    if (!c->evT1[c->ringUsed]) HIP_TRY(hipEventCreate(&c->evT1[c->ringUsed]));
    hipEvent_t ev0 = c->evT0[c->ringUsed], ev1 = c->evT1[c->ringUsed];
    c->ringUsed++;
    HIP_TRY(hipEventRecord(ev0, s));
    int rc = render_frame_kernel(c, A, totalWork, s);
    if (rc) return rc;
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(PRT_HIP_ELAUNCH, std::string("frame_kernel launch: ") + hipGetErrorString(le));
    HIP_TRY(hipEventRecord(ev1, s));
    c->lastRank = p->rank;
    c->lastNranks = p->nranks;
    c->lastTile = p->tileSize;
    c->lastTarget = d_rgb;
    if (caller) {
        HIP_TRY(hipEventRecord(c->evOut, s));
        HIP_TRY(hipStreamWaitEvent(caller, c->evOut, 0));
    }
    c->timed = true;
    c->frameLaunched = true;
    return PRT_HIP_OK;
}

int prt_hip_render_gbuffer(prt_hip_ctx* c, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t type, uint32_t seed, float exposure,
                           float* d_rgb, void* stream)
{
    if (!c) return fail(PRT_HIP_EINVAL, "NULL argument");
    if (!c->haveScene || !c->haveCamera) return fail(PRT_HIP_ESTATE, "upload a scene and set a camera first");
    const uint32_t W = c->cam.width, H = c->cam.height;
    if (x1 < x0 || y1 < y0 || x1 >= W || y1 >= H) return fail(PRT_HIP_EINVAL, "pixel rectangle outside the image");
    if (type > 2) return fail(PRT_HIP_EINVAL, "type must be 0 (diffuse), 1 (mesh normal) or 2 (normal)");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    hipStream_t caller = (stream && (hipStream_t)stream != c->stream) ? (hipStream_t)stream : nullptr;
    if (caller) {
        HIP_TRY(hipEventRecord(c->evIn, caller));
        HIP_TRY(hipStreamWaitEvent(s, c->evIn, 0));
    }
    if (!d_rgb) {
        if (c->fbPixels != (size_t)W * H) {
            if (c->fb) (void)hipFree(c->fb);
            c->fb = nullptr;
            HIP_TRY(hipMalloc(&c->fb, (size_t)W * H * 3 * sizeof(float)));
            HIP_TRY(hipMemsetAsync(c->fb, 0, (size_t)W * H * 3 * sizeof(float), s));
            c->fbPixels = (size_t)W * H;
        }
        d_rgb = c->fb;
    }
    const uint32_t rw = x1 - x0 + 1, rh = y1 - y0 + 1;
    if ((uint64_t)rw * rh > 0xffffffffull) return fail(PRT_HIP_EINVAL, "rectangle too large");
    const uint32_t want = (uint32_t)(((uint64_t)rw * rh + PRT_BLOCK - 1) / PRT_BLOCK);
    const uint32_t resident = (uint32_t)(c->computeUnits * 4); // persistent lanes: 4 blocks of 256 per CU
    const uint32_t blocks = std::min<uint32_t>(want, resident);
    int rc = ensure_launch_resources(c, resident);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->counters, 0, PRT_STAT_SHARDS * PRT_STAT_STRIDE * sizeof(unsigned long long), s));
    HIP_TRY(hipMemsetAsync(c->work, 0, PRT_WORK_WORDS * sizeof(uint32_t), s));
    GbufArgs A{c->sc, c->cam, x0, y0, rw, rh, type, seed, exposure, d_rgb, c->work, c->spill, c->spillThreads, c->counters};
    hipLaunchKernelGGL(gbuffer_kernel, dim3(blocks), dim3(PRT_BLOCK), 0, s, A);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(PRT_HIP_ELAUNCH, std::string("gbuffer_kernel launch: ") + hipGetErrorString(le));
    if (caller) {
        HIP_TRY(hipEventRecord(c->evOut, s));
        HIP_TRY(hipStreamWaitEvent(caller, c->evOut, 0));
    }
    c->timed = false;
    return PRT_HIP_OK;
}

float* prt_hip_framebuffer(prt_hip_ctx* c) { return c ? c->fb : nullptr; }

} // extern "C"

// Errors of launches since the last prt_hip_get_stats.  Every render clears its own control words and counters, so a watchdog
// abort or a stack overflow of an EARLIER frame of an asynchronous sequence (bench steps, render + gather loops) would be gone by
// the time anybody looks; the frame kernel therefore also ORs them into words no render clears.
int prt_sticky_error(prt_hip_ctx* c, bool clear)
{
    uint32_t S[PRT_STICKY_WORDS] = {0};
    if (!c->work) return PRT_HIP_OK;
    HIP_TRY(hipMemcpy(S, c->work + PRT_WORK_WORDS, sizeof(S), hipMemcpyDeviceToHost));
    if (S[0] == 0) return PRT_HIP_OK;
    if (clear) HIP_TRY(hipMemset(c->work + PRT_WORK_WORDS, 0, sizeof(S)));
    if (S[0] & 1u) {
        std::string msg = "frame kernel: scheduler watchdog fired in a launch since the last prt_hip_get_stats (a workgroup waited for work that never came; its image is incomplete);";
        for (uint32_t k = 0; k < std::min<uint32_t>(S[2], 8u); k++) {
            const uint32_t* D = S + 8 + 16 * k;
            char line[256];
            snprintf(line, sizeof(line), " [block %u wave %u: ready %u live %u exhausted %u lock %u, %u groups wait for %u rays, tails %u %u %u %u heads %u %u %u %u]",
                     D[0], D[1], D[2], D[3], D[4], D[5], D[6], D[7], D[8], D[9], D[10], D[11], D[12], D[13], D[14], D[15]);
            msg += line;
        }
        return fail(PRT_HIP_ELAUNCH, msg);
    }
    return fail(PRT_HIP_ESTACK, "BVH traversal needed more than 64 stack entries in a launch since the last prt_hip_get_stats (the reference asserts here, bvh.cpp:552)");
}

extern "C" {

int prt_hip_download(prt_hip_ctx* c, float* rgb_host, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1)
{
    if (!c || !rgb_host) return fail(PRT_HIP_EINVAL, "NULL argument");
    if (!c->fb) return fail(PRT_HIP_ESTATE, "nothing rendered into the context framebuffer");
    const uint32_t W = c->cam.width, H = c->cam.height;
    if (x1 < x0 || y1 < y0 || x1 >= W || y1 >= H) return fail(PRT_HIP_EINVAL, "pixel rectangle outside the image");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    size_t rowBytes = (size_t)(x1 - x0 + 1) * 3 * sizeof(float);
    size_t off = ((size_t)y0 * W + x0) * 3;
    HIP_TRY(hipMemcpy2D(rgb_host + off, (size_t)W * 3 * sizeof(float), c->fb + off, (size_t)W * 3 * sizeof(float), rowBytes,
                        y1 - y0 + 1, hipMemcpyDeviceToHost));
    return prt_sticky_error(c, false); // the pixels are delivered, but a caller must learn that a launch behind them was cut short
}

int prt_hip_get_stats(prt_hip_ctx* c, prt_hip_stats* st)
{
    if (!c || !st) return fail(PRT_HIP_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long h[PRT_STAT_STRIDE] = {0};
    {
        std::vector<unsigned long long> all((size_t)PRT_STAT_SHARDS * PRT_STAT_STRIDE);
        HIP_TRY(hipMemcpy(all.data(), c->counters, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int sh = 0; sh < PRT_STAT_SHARDS; sh++)
            for (int k = 0; k < PRT_STAT_STRIDE; k++) h[k] += all[(size_t)sh * PRT_STAT_STRIDE + k];
    }
    // errors of ANY launch since the last call (the last one included): see prt_sticky_error
    const int sticky = prt_sticky_error(c, true);
#ifdef PRT_PROFILE
    if (h[14])
        fprintf(stderr, "frame profile: waves %llu, per wave: shade %.1f%% (%.0f calls) trace %.1f%% (%.0f calls) idle/decide %.1f%% of %.2f Mcycles\n", h[14],
                100.0 * h[8] / h[13], (double)h[11] / h[14], 100.0 * h[9] / h[13], (double)h[12] / h[14], 100.0 * h[10] / h[13], h[13] / 1e6 / h[14]);
    if (h[14]) {
        for (int m = 0; m < 4; m++)
            fprintf(stderr, "  trace mode %d: %.1f M loop turns, %.1f lanes with a ray per turn, %.1f Gcycles in the loops => %.0f cycles per turn\n", m, h[16 + 3 * m] / 1e6,
                    (double)h[17 + 3 * m] / (double)(h[16 + 3 * m] ? h[16 + 3 * m] : 1), h[18 + 3 * m] * 1024.0 / 1e9,
                    h[18 + 3 * m] * 1024.0 / (double)(h[16 + 3 * m] ? h[16 + 3 * m] : 1));
        for (int m = 0; m < 4; m++) {
            const unsigned long long* q = h + 32 + 8 * m;
            fprintf(stderr, "  mode %d rounds: node %.1f M with %.1f lanes, leaf %.1f M with %.1f lanes on a leaf (%.1f pair lanes working; serial build: lanes on a second triangle); refills %.1f M with %.1f lanes\n", m,
                    q[0] / 1e6, (double)q[1] / (double)(q[0] ? q[0] : 1), q[2] / 1e6, (double)q[3] / (double)(q[2] ? q[2] : 1),
                    (double)q[4] / (double)(q[2] ? q[2] : 1), q[6] / 1e6, (double)q[5] / (double)(q[6] ? q[6] : 1));
        }
        for (int m = 0; m < 4; m++) {
            const unsigned long long* q = h + 32 + 8 * m;
            const unsigned long long* w = h + 64 + 8 * m;
            const double nr = (double)(q[0] ? q[0] : 1), lr = (double)(q[2] ? q[2] : 1);
            fprintf(stderr, "  mode %d lanes sitting out: node rounds %.1f on a leaf, %.1f finished, %.1f without a ray; leaf rounds %.1f on a node, %.1f finished, %.1f without a ray; %.2f hit-update turns per leaf round; %.1f distinct records per node round\n",
                    m, w[0] / nr, w[1] / nr, w[2] / nr, w[3] / lr, w[4] / lr, w[5] / lr, w[6] / lr, w[7] / nr);
        }
        for (int m = 0; m < 4; m++) {
            const unsigned long long* w = h + 96 + 4 * m;
            const double turns = (double)(h[16 + 3 * m] ? h[16 + 3 * m] : 1);
            fprintf(stderr, "  mode %d cycles per loop turn: refill + result hand-off %.0f, entering / leaving BVHs %.0f, step phase %.0f\n", m, w[0] * 1024.0 / turns,
                    w[1] * 1024.0 / turns, w[2] * 1024.0 / turns);
        }
        if (h[112])
            fprintf(stderr, "  refills that left lanes empty: %.1f M, %.1f lanes each; at that moment %.1f rays in the block's OTHER queues, %.1f groups ready for shading, shade role taken %.2f of the time\n",
                    h[112] / 1e6, (double)h[113] / h[112], (double)h[114] / h[112], (double)h[115] / h[112], (double)h[116] / h[112]);
        if (h[118])
            fprintf(stderr, "  shade passes with a bounce: %.1f M, %.1f of 64 lanes carry a path through it; the shade role is held %.2f of a block's time (sum of its waves' shade shares)\n",
                    h[118] / 1e6, (double)h[119] / h[118], 4.0 * h[8] / h[13]);
        fprintf(stderr, "  node steps that leave a lane on an internal record outside the hot set: %.2f G by descending from the parent, %.2f G by a pop (%.2f of them direct)\n",
                h[120] / 1e9, h[121] / 1e9, (double)h[120] / (double)(h[120] + h[121] ? h[120] + h[121] : 1));
        fprintf(stderr, "  cooperative leaf rounds in which a candidate of a leaf with alpha-tested triangles came up: %.1f M, %.0f cycles each from the candidate test to the end of the alpha tests = %.1f Gcycles of wave time\n",
                h[122] / 1e6, h[123] * 1024.0 / (double)(h[122] ? h[122] : 1), h[123] * 1024.0 / 1e9);
        fprintf(stderr, "  stack pops of modes 1-3: %.1f G, of them from the spill area in HBM: %.2f G\n", (h[32 + 15] + h[32 + 23] + h[32 + 31]) / 1e9, h[39] / 1e9);
        fprintf(stderr, "  claims %.1f M, empty %.1f M; shade passes %.1f M with %.2f groups each\n", h[28] / 1e6, h[29] / 1e6, h[30] / 1e6, (double)h[31] / (double)(h[30] ? h[30] : 1));
    }
#endif
    st->raysTraced = h[0];
    st->occludedTraced = h[1];
    st->nBox = h[2];
    st->nTri = h[3];
    st->nHit = h[4];
    st->nTap = h[5];
    st->nPx = h[6];
    for (int m = 0; m < 4; m++) {
#ifdef PRT_PROFILE
        st->modeBox[m] = st->modeTri[m] = st->modeTap[m] = 0; // the words carry the profile build's loop statistics
#else
        st->modeBox[m] = h[16 + 3 * m];
        st->modeTri[m] = h[17 + 3 * m];
        st->modeTap[m] = h[18 + 3 * m];
#endif
    }
    st->stackOverflow = h[7];
    fold_timing(c);
    st->kernelMs = c->accLaunches ? c->lastMs : 0.0;
    st->kernelMsSum = c->accMs;
    st->kernelLaunches = c->accLaunches;
    c->accMs = c->lastMs = 0.0;
    c->accLaunches = 0;
    if (sticky) return sticky;
    if (h[7]) return fail(PRT_HIP_ESTACK, "BVH traversal needed more than 64 stack entries (the reference asserts here, bvh.cpp:552)");
    return PRT_HIP_OK;
}

#ifdef PRT_TEST_ENTRY_POINTS
} // extern "C"
namespace {
// device scratch of one test call: freed on every return path
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <typename T> T* as() const { return (T*)p; }
};
} // namespace
extern "C" {

int prt_hip_trace_rays(prt_hip_ctx* c, int mode, uint32_t n, const float* org, const float* dir, float maxT, prt_hit* hits)
{
    if (!c || !org || !dir || !hits) return fail(PRT_HIP_EINVAL, "NULL argument");
    if (!c->haveScene) return fail(PRT_HIP_ESTATE, "upload a scene first");
    if (n == 0 || (n & 7u) || mode < 0 || mode > 3) return fail(PRT_HIP_EINVAL, "n must be a positive multiple of 8, mode 0..3");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t blocks = (n + PRT_BLOCK - 1) / PRT_BLOCK;
    int rc = ensure_launch_resources(c, blocks);
    if (rc) return rc;
    DevBuf dorg, ddir, dh;
    HIP_TRY(dorg.alloc((size_t)n * 12));
    HIP_TRY(ddir.alloc((size_t)n * 12));
    HIP_TRY(dh.alloc((size_t)n * sizeof(prt_hit)));
    HIP_TRY(hipMemcpy(dorg.p, org, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ddir.p, dir, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(c->counters, 0, PRT_STAT_SHARDS * PRT_STAT_STRIDE * sizeof(unsigned long long), c->stream));
    HIP_TRY(hipMemsetAsync(c->work, 0, PRT_WORK_WORDS * sizeof(uint32_t), c->stream));
    RaysArgs A{c->sc, n, dorg.as<float>(), ddir.as<float>(), maxT, dh.as<prt_hit>(), c->work, c->spill, c->spillThreads, c->counters};
    blocks = std::min<uint32_t>(blocks, (uint32_t)c->spillThreads / PRT_BLOCK);
    // timed like a render (prt_hip_get_stats reports it as kernelMs): tools/ray_order_experiment.py compares orders of one batch
    if (c->ringUsed == PRT_TIMING_RING) fold_timing(c);
    if (!c->evT0[c->ringUsed]) HIP_TRY(hipEventCreate(&c->evT0[c->ringUsed]));
    if (!c->evT1[c->ringUsed]) HIP_TRY(hipEventCreate(&c->evT1[c->ringUsed]));
    hipEvent_t ev0 = c->evT0[c->ringUsed], ev1 = c->evT1[c->ringUsed];
    c->ringUsed++;
    HIP_TRY(hipEventRecord(ev0, c->stream));
    if (mode == 0) hipLaunchKernelGGL(rays_kernel<PRT_MODE_SINGLE>, dim3(blocks), dim3(PRT_BLOCK), 0, c->stream, A);
    else if (mode == 1) hipLaunchKernelGGL(rays_kernel<PRT_MODE_PACKET>, dim3(blocks), dim3(PRT_BLOCK), 0, c->stream, A);
    else if (mode == 2) hipLaunchKernelGGL(rays_kernel<PRT_MODE_OCC_SINGLE>, dim3(blocks), dim3(PRT_BLOCK), 0, c->stream, A);
    else hipLaunchKernelGGL(rays_kernel<PRT_MODE_OCC_PACKET>, dim3(blocks), dim3(PRT_BLOCK), 0, c->stream, A);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(PRT_HIP_ELAUNCH, std::string("rays_kernel launch: ") + hipGetErrorString(le));
    HIP_TRY(hipEventRecord(ev1, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(hits, dh.p, (size_t)n * sizeof(prt_hit), hipMemcpyDeviceToHost));
    c->timed = false;
    return PRT_HIP_OK;
}

int prt_hip_test_leaf(prt_hip_ctx* c, uint32_t n, const float* records, float* out)
{
    if (!c || !records || !out || n == 0) return fail(PRT_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf din, dout;
    HIP_TRY(din.alloc((size_t)n * 22 * 4));
    HIP_TRY(dout.alloc((size_t)n * 24 * 4));
    HIP_TRY(hipMemcpy(din.p, records, (size_t)n * 22 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(leaf_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, din.as<float>(), dout.as<float>());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * 24 * 4, hipMemcpyDeviceToHost));
    return PRT_HIP_OK;
}

int prt_hip_test_sincos(prt_hip_ctx* c, uint32_t n, const float* theta, float* s, float* cs)
{
    if (!c || !theta || !s || !cs || n == 0) return fail(PRT_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf dt, ds, dc;
    HIP_TRY(dt.alloc((size_t)n * 4));
    HIP_TRY(ds.alloc((size_t)n * 4));
    HIP_TRY(dc.alloc((size_t)n * 4));
    HIP_TRY(hipMemcpy(dt.p, theta, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sincos_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, dt.as<float>(), ds.as<float>(), dc.as<float>());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(s, ds.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(cs, dc.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return PRT_HIP_OK;
}

int prt_hip_test_powf(prt_hip_ctx* c, uint32_t n, const float* x, float* y)
{
    if (!c || !x || !y || n == 0) return fail(PRT_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf dx, dy;
    HIP_TRY(dx.alloc((size_t)n * 4));
    HIP_TRY(dy.alloc((size_t)n * 4));
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(powf_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, dx.as<float>(), dy.as<float>());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(y, dy.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return PRT_HIP_OK;
}

int prt_hip_test_camera(prt_hip_ctx* c, uint32_t x, uint32_t y, uint32_t state, float* out92)
{
    if (!c || !out92) return fail(PRT_HIP_EINVAL, "bad argument");
    if (!c->haveCamera) return fail(PRT_HIP_ESTATE, "set a camera first");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf d;
    HIP_TRY(d.alloc(92 * 4));
    hipLaunchKernelGGL(camera_kernel, dim3(1), dim3(64), 0, c->stream, c->cam, x, y, state, d.as<float>());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out92, d.p, 92 * 4, hipMemcpyDeviceToHost));
    return PRT_HIP_OK;
}
#endif // PRT_TEST_ENTRY_POINTS

} // extern "C"
