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

// ============================================================================ wavefront pipeline
// The per-pixel loop of the reference is run as a state machine over ALL pixel groups at once (DESIGN.md "Kernels"):
//
//   shade_kernel   8 lanes per pixel group (one lane per path slot).  Consumes the hits of the previous iteration
//                  (gathering alive paths into slots 0..n-1 by ballot/prefix rank), runs the bounce phase of
//                  path_tracer.cpp:124-293 (RNG draws by prefix-counted stepping of the shared xorshift state, BSDF
//                  direction, Russian roulette) and EMITS the rays of the next iteration into per-mode queues, packed
//                  by wave ballot + prefix sum + one atomic per wave.
//   trace_kernel   one lane per queued ray, one instantiation per traversal mode (primary packet rays, scatter
//                  rays, packet / single occlusion rays), results written back to the owner slot.
//
// One iteration = shade + 4 traces; a packet needs 1 + maxDepth iterations, so an image needs
// (samples/8)*(1+maxDepth)+1 iterations, enqueued back to back without host synchronisation.
enum { Q_PRIMARY = PRT_MODE_PACKET, Q_SCATTER = PRT_MODE_SINGLE, Q_OCC_PACKET = PRT_MODE_OCC_PACKET, Q_OCC_SINGLE = PRT_MODE_OCC_SINGLE, Q_COUNT = 4 };
enum { PH_START = 0, PH_WAIT_PRIMARY = 1, PH_WAIT_BOUNCE = 2, PH_DONE = 3 };
#define PRT_QSHARDS 16
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

struct WfArgs {
    DevScene sc;
    DevCamera cam;
    prt_render_params p;
    uint32_t x0, y0, x1, y1;
    uint32_t tilesXImage;
    uint32_t rtx0, rty0, rtnx, rtny;
    uint32_t fullWidth, firstOwned;
    uint32_t workBase;   // first work item (tile-major pixel index) of this pass, a multiple of the tile area
    uint32_t partIndex, partCount; // the pass is dealt to partCount independent pipelines tile by tile; this is pipeline partIndex
    uint32_t groupCount; // pixel groups in this pass
    float* rgb;
    unsigned long long* counters; // rays, occl, nBox, nTri, nHit, nTap, nPx, overflow
    // per group
    uint32_t* gRng;
    uint32_t* gInfo;  // packet | depth << 8 | alive << 16 | phase << 20 | alive at depth 0 << 24
    uint32_t* gPixel; // x | y << 16, 0xffffffff = no pixel (outside the rectangle / not this rank's tile)
    float4* gColor;
    // per slot (8 per group): what a path carries from one iteration to the next (68 B)
    float4* S0; // pos.xyz, bits(material of the surface the path stands on)
    float4* S1; // shading normal xyz, bits(slot flags)
    float4* S2; // direction of the ray in flight (primary or scatter) xyz
    float4* S3; // beta.xyz
    float4* S4; // result.xyz -- belongs to the SLOT, not the path: it stays behind when the path dies or is compacted away
    float4* S5; // environment light only: lightDir[slot].xyz      (path_tracer.cpp:125; like result, it belongs to the slot and
    float4* S6; //                         lightIntensity[slot].xyz  keeps its last value while the packet lives)
    float4* hitA;   // t i j k
    uint2* hitB;    // primId meshId
    uint32_t* occl; // 1 = occluded
    // ray queues
    uint32_t* qE[Q_COUNT]; // queued ray = owner slot (26 bits) | reverseBits << 26 | lightSet << 29; the trace kernels rebuild
                           // the ray from the owner's state
    // Queue q is split into PRT_QSHARDS regions of shardCap entries; blocks append to region blockIdx % PRT_QSHARDS, so that
    // the returning atomics that reserve space are spread over 16 addresses per queue (one address takes ~88 of them
    // per microsecond, MI355X_MICROARCH.md "dequeue").  qWork: [0..3] claim cursors of the trace kernels,
    // [4 + q*PRT_QSHARDS + shard] entries in that region.
    uint32_t* qWork;
    uint32_t shardCap;
    uint32_t* spill;
    uint32_t spillStride;
};

// Diagnostic build (-DPRT_STAMP): per-segment wave time of the shade kernel, summed into counters[8..15].
#ifdef PRT_STAMP
#define STAMP(k)                                                                                    \
    do {                                                                                            \
        unsigned long long t_ = __builtin_amdgcn_s_memtime();                                       \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                         \
        stampAcc[k] += t_ - stampLast;                                                              \
        stampLast = t_;                                                                             \
    } while (0)
#else
#define STAMP(k)
#endif

template <bool COUNT, bool ENV>
__global__ __launch_bounds__(PRT_BLOCK, PRT_SHADE_WAVES) void shade_kernel(WfArgs A)
{
#ifdef PRT_STAMP
    unsigned long long stampAcc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stampLast = __builtin_amdgcn_s_memtime();
#endif
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, slot = lane & 7u, gbase = lane & ~7u;
    const uint32_t g = (blockIdx.x * PRT_BLOCK + tid) >> 3;
    const bool inRange = g < A.groupCount; // whole groups are in or out together
    const uint32_t gs = (inRange ? g : 0u) * 8u + slot;
    const DevScene& sc = A.sc;
    const DevCamera& cam = A.cam;
    const uint32_t samples = A.p.samples, maxDepth = A.p.maxDepth, rrDepth = A.p.rrDepth, packets = samples / 8u;
    const float kPi = 3.14159265358979323846f;
    const float kFar = 2.0f * sc.radius; // path_tracer.cpp:192
    const float kEpsilon = 0.0008f;      // :193
    const uint32_t lowerMask = (1u << slot) - 1u;

    Traffic tr{0, 0, 0, 0};
    unsigned long long nRays = 0, nOccl = 0, nPx = 0;

    uint32_t info = inRange ? A.gInfo[g] : ((uint32_t)PH_DONE << 20);
    uint32_t phase = (info >> 20) & 0xfu, pk = info & 0xffu, depth = (info >> 8) & 0xffu, alive = (info >> 16) & 0xfu;
    uint32_t alive0 = (info >> 24) & 0xfu; // slots that have held a path in this packet: the others' result is still 0
    const uint32_t aliveAtEntry = (phase == PH_WAIT_BOUNCE) ? alive : 0u;
    uint32_t rng = 0, pixel = 0xffffffffu;
    Vec3 color = mk3(0, 0, 0);
    if (inRange) { // with gInfo, not after it: one round trip (a finished group reads three words it will not use)
        rng = A.gRng[g];
        pixel = A.gPixel[g];
        float4 c = A.gColor[g];
        color = mk3(c.x, c.y, c.z);
    }
    const uint32_t x = pixel & 0xffffu, y = pixel >> 16;
    STAMP(0);

    // slot state
    Vec3 pos = mk3(0, 0, 0), rayDir = mk3(0, 0, 0), normal = mk3(0, 0, 0), beta = mk3(1, 1, 1), result = mk3(0, 0, 0), ndir = mk3(0, 0, 0);
    Surface props{mk3(0, 0, 0), Vec2{0, 0}, 0, 0};
    uint32_t material = 0, sflags = 0, lightSet = 0;
    Vec3 envL = mk3(0, 0, 0), envI = mk3(0, 0, 0); // ENV: this bounce's sampled light, when the slot samples one
    bool envSampled = false;

    bool needBounce = false, needEnd = false, needCamera = false;
    bool emitPrimary = false, emitShadow = false, emitScatter = false, shadowPacket = false;
    uint32_t reverseBits = 0;

    if (phase == PH_START) {
        if (pixel == 0xffffffffu || packets == 0u) {
            phase = PH_DONE;
            if (pixel != 0xffffffffu && slot == 0) { // samples < 8: the reference still writes 0/samples
                float* px = A.rgb + ((size_t)x + (size_t)y * cam.width) * 3;
                px[0] = px[1] = px[2] = A.p.exposure * (0.0f / (float)samples);
                nPx++;
                nRays += samples;
            }
        } else {
            needCamera = true;
            if (slot == 0) nRays += samples; // path_tracer.cpp:62
        }
    } else if (phase == PH_WAIT_PRIMARY) {
        // ---- ComputeRadiance set-up (path_tracer.cpp:81-120): hits gathered into slots 0..alive-1 in lane order
        float4 ha = nt_load4(&A.hitA[gs]);
        uint2 hb = A.hitB[gs];
        float4 s2 = nt_load4(&A.S2[gs]); // the primary ray's direction
        Vec3 pdir = mk3(s2.x, s2.y, s2.z), porg = mk3(cam.pos[0], cam.pos[1], cam.pos[2]);
        DevHit h{ha.x, ha.y, ha.z, ha.w, hb.x, hb.y};
        bool isHit = h.t != -1.0f;
        Surf5 sv{mk3(0, 0, 0), Vec2{0, 0}, 0, 0};
        Vec3 snormal = mk3(0, 0, 0), spos = mk3(0, 0, 0);
        if (isHit) {
            Surface s;
            get_surface<COUNT>(sc, h, s, tr);
            sv = Surf5{s.normal, s.uv, s.mat, s.prim};
            snormal = sample_bump<COUNT>(sc, s.mat, s, tr);
            spos = add3(scale3(h.t, pdir), porg);
        }
        uint32_t hm = group_ballot(isHit, gbase);
        alive = __popc(hm);
        uint32_t src = gbase + ((slot < alive) ? nth_set(hm, slot) : slot);
        props.normal = sh3(sv.normal, src);
        props.uv = Vec2{shf(sv.uv.x, src), shf(sv.uv.y, src)};
        props.mat = shu(sv.mat, src);
        props.prim = shu(sv.prim, src);
        material = props.mat;
        normal = sh3(snormal, src);
        pos = sh3(spos, src);
        rayDir = sh3(pdir, src);
        beta = mk3(1.0f, 1.0f, 1.0f);
        result = mk3(0.0f, 0.0f, 0.0f);
        lightSet = 0;
        depth = 0;
        alive0 = alive;
        if (alive != 0u && depth < maxDepth) needBounce = true;
        else needEnd = true;
    } else if (phase == PH_WAIT_BOUNCE) {
        uint32_t pmat = 0;
        if (slot < alive) { // dead slots carry nothing
            float4 s0 = nt_load4(&A.S0[gs]), s1 = nt_load4(&A.S1[gs]), s2 = nt_load4(&A.S2[gs]), s3 = nt_load4(&A.S3[gs]);
            pos = mk3(s0.x, s0.y, s0.z);
            pmat = asu(s0.w);
            normal = mk3(s1.x, s1.y, s1.z);
            sflags = asu(s1.w);
            ndir = mk3(s2.x, s2.y, s2.z);
            beta = mk3(s3.x, s3.y, s3.z);
            float4 s4 = nt_load4(&A.S4[gs]);
            result = mk3(s4.x, s4.y, s4.z);
        }
        props.mat = pmat;
        lightSet = (sflags & SLOT_LIGHT_SET) ? 1u : 0u;
#ifdef PRT_STAMP
        asm volatile("" ::"v"(pos.x), "v"(beta.x), "v"(result.x));
        __builtin_amdgcn_s_waitcnt(0x0070); // vmcnt(0): the state loads have landed
#endif
        STAMP(6);
        // ---- light contribution of the previous bounce (path_tracer.cpp:226-231, 246-249)
        if (sflags & SLOT_HAS_SHADOW) {
            if (A.occl[gs] == 0u) {
                Vec3 lightDir = mk3(0, 0, 0), lightInt = mk3(0, 0, 0);
                if (lightSet) {
                    if (ENV) {
                        float4 l5 = nt_load4(&A.S5[gs]), l6 = nt_load4(&A.S6[gs]);
                        lightDir = mk3(l5.x, l5.y, l5.z);
                        lightInt = mk3(l6.x, l6.y, l6.z);
                    } else {
                        lightDir = mk3(sc.lightDir[0], sc.lightDir[1], sc.lightDir[2]);
                        lightInt = mk3(sc.lightIntensity[0], sc.lightIntensity[1], sc.lightIntensity[2]);
                    }
                }
                Vec3 lr = div3s(scale3(std_max(dot3(lightDir, normal), 0.0f), lightInt), kPi);
                result = add3(result, mul3(beta, lr));
            }
        }
        // ---- scatter hits, ordered compaction into slot ci (path_tracer.cpp:281-293)
        bool hitNext = false;
        Surface ns{mk3(0, 0, 0), Vec2{0, 0}, 0, 0};
        Vec3 npos = mk3(0, 0, 0);
        if (sflags & SLOT_SURVIVE) {
            float4 ha = nt_load4(&A.hitA[gs]);
            uint2 hb = A.hitB[gs];
            DevHit nh{ha.x, ha.y, ha.z, ha.w, hb.x, hb.y};
            if (nh.t != -1.0f) {
                hitNext = true;
                get_surface<COUNT>(sc, nh, ns, tr);
                npos = add3(scale3(nh.t, ndir), pos);
            }
        }
#ifdef PRT_STAMP
        asm volatile("" ::"v"(npos.x), "v"(ns.normal.x));
        __builtin_amdgcn_s_waitcnt(0x0070);
#endif
        STAMP(7);
        uint32_t nm = group_ballot(hitNext, gbase);
        uint32_t nAlive = __popc(nm);
        if (nAlive == 0u) {
            needEnd = true; // path_tracer.cpp:295
        } else {
            uint32_t ci = __popc(nm & lowerMask);
            uint32_t smat = 0;
            Vec3 snorm = mk3(0, 0, 0);
            Surf5 nv{mk3(0, 0, 0), Vec2{0, 0}, 0, 0};
            if (hitNext) {
                // materials[ci] = props[i].material reads the slot's PREVIOUS surface unless ci == i (:286-288)
                smat = (ci == slot) ? ns.mat : props.mat;
                snorm = sample_bump<COUNT>(sc, smat, ns, tr);
                nv = Surf5{ns.normal, ns.uv, ns.mat, ns.prim};
            }
            // beta[ci] = beta[i]/(1-q) is only written under Russian roulette (:263); q is a function of beta
            Vec3 betaNew = beta;
            if (depth > rrDepth) {
                float q = std_max(0.05f, 1.0f - length3(beta));
                betaNew = div3s(beta, 1.0f - q);
            }
#ifdef PRT_STAMP
            asm volatile("" ::"v"(snorm.x));
            __builtin_amdgcn_s_waitcnt(0x0070);
#endif
            STAMP(8);
            uint32_t src2 = gbase + ((slot < nAlive) ? nth_set(nm, slot) : slot);
            props.normal = sh3(nv.normal, src2);
            props.uv = Vec2{shf(nv.uv.x, src2), shf(nv.uv.y, src2)};
            props.mat = shu(nv.mat, src2);
            props.prim = shu(nv.prim, src2);
            material = shu(smat, src2);
            normal = sh3(snorm, src2);
            pos = sh3(npos, src2);
            rayDir = sh3(ndir, src2);
            Vec3 bmoved = sh3(betaNew, src2);
            if (depth > rrDepth && slot < nAlive) beta = bmoved;
            alive = nAlive;
            depth++;
            if (depth < maxDepth) needBounce = true;
            else needEnd = true;
        }
    }

    STAMP(1);
    if (needBounce) {
        // ---- one bounce (path_tracer.cpp:131-190 and the Russian roulette of :258-265)
        const bool active = slot < alive;
        uint32_t rtype = 2u;
        if (active) {
            const float4* mp = sc.mats + PRT_MAT_STRIDE * (size_t)material;
            float4 m0 = mp[0], m1 = mp[1];
            rtype = asu(m0.w);
            if (m1.x != 0.0f) result = add3(result, mul3(beta, mk3(m1.x, m1.y, m1.z))); // :137-139
        }
        const bool draws = active && (rtype == 0u || rtype == 1u);
        uint32_t dm = group_ballot(draws, gbase);
        uint32_t pre = 2u * __popc(dm & lowerMask), tot = 2u * __popc(dm);
        if (ENV) { // a diffuse slot draws two more for InfiniteAreaLight::sample (:164-167), after its r2, r1
            uint32_t em = group_ballot(active && rtype == 0u, gbase);
            pre += 2u * __popc(em & lowerMask);
            tot += 2u * __popc(em);
        }
        uint32_t s = rng, r2b = 0, r1b = 0, uxb = 0, uyb = 0;
        for (uint32_t j = 0; j < tot; j++) {
            s = xorshift32(s);
            if (j == pre) r2b = s;
            if (j == pre + 1u) r1b = s;
            if (ENV && j == pre + 2u) uxb = s;
            if (ENV && j == pre + 3u) uyb = s;
        }
        rng = s;
        STAMP(9);
        Vec3 nextDir = mk3(0, 0, 0);
        bool wantLight = false;
        if (draws) {
            Vec3 dd = diffuse_dir(normal, rng_to_float(r2b), rng_to_float(r1b));
            if (rtype == 0u) {
                nextDir = dd;
                beta = mul3(beta, sample_diffuse<COUNT>(sc, material, props.uv, tr)); // :162
                if (ENV) { // :164-167
                    env_sample<COUNT>(sc, rng_to_float(uxb), rng_to_float(uyb), envL, envI, tr);
                    envSampled = true;
                    lightSet = 1u;
                    wantLight = true;
                } else if (sc.hasLight) { // :168-172
                    lightSet = 1u;
                    wantLight = true;
                }
            } else {
                Vec3 reflectDir = sub3(rayDir, scale3(dot3(normal, rayDir), scale3(2.0f, normal))); // :186
                nextDir = add3(scale3(0.9f, reflectDir), scale3(0.1f, dd));
            }
        }
#ifdef PRT_STAMP
        asm volatile("" ::"v"(nextDir.x), "v"(beta.x));
        __builtin_amdgcn_s_waitcnt(0x0070);
#endif
        STAMP(10);
        const bool directLighting = group_ballot(wantLight, gbase) != 0u;
        sflags = 0;
        if (directLighting && active) { // :196-252: every alive path gets an occlusion ray
            emitShadow = true; // the occlusion kernels build org = pos + kFar*L, dir = -L from the slot's state
            shadowPacket = (alive & 0xfu) > 2u; // :198
            sflags |= SLOT_HAS_SHADOW;
            nRays++;
            nOccl++;
        }
        bool survive = active;
        if (depth > rrDepth) { // one draw per alive slot, in slot order
            uint32_t s2 = rng, ub = 0;
            for (uint32_t j = 0; j < alive; j++) {
                s2 = xorshift32(s2);
                if (j == slot) ub = s2;
            }
            rng = s2;
            if (active) {
                float q = std_max(0.05f, 1.0f - length3(beta));
                if (rng_to_float(ub) < q) survive = false;
            }
        }
        if (survive) {
            ndir = normalize3(nextDir); // :267
            emitScatter = true;
            sflags |= SLOT_SURVIVE;
            nRays++;
        }
        phase = PH_WAIT_BOUNCE;
    }

    STAMP(2);
    if (needEnd) {
        // ---- Σ result[0..7] in slot order (path_tracer.cpp:303-307), color += (:71).  Slots whose path ended in an earlier
        // iteration left their result in memory.
        if (slot >= aliveAtEntry && slot < alive0 && phase == PH_WAIT_BOUNCE) {
            float4 s4 = nt_load4(&A.S4[gs]);
            result = mk3(s4.x, s4.y, s4.z);
        }
        Vec3 res = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
        for (uint32_t l = 0; l < 8; l++) res = add3(res, sh3(result, gbase + l));
        color = add3(color, res);
        pk++;
        if (pk < packets) {
            needCamera = true;
        } else {
            Vec3 c = scale3(A.p.exposure, div3s(color, (float)samples)); // path_tracer.cpp:28, image.cpp:45
            if (slot == 0) {
                float* px = A.rgb + ((size_t)x + (size_t)y * cam.width) * 3;
                px[0] = c.x;
                px[1] = c.y;
                px[2] = c.z;
                nPx++;
            }
            phase = PH_DONE;
        }
    }

    if (needCamera) {
        DevRay pr;
        Vec3 avgDir;
        camera_packet(cam, rng, x, y, slot, gbase, pr, avgDir);
        reverseBits = (avgDir.x < 0.0f ? 1u : 0u) | (avgDir.y < 0.0f ? 2u : 0u) | (avgDir.z < 0.0f ? 4u : 0u);
        ndir = pr.dir;
        emitPrimary = true;
        phase = PH_WAIT_PRIMARY;
    }

    STAMP(3);
    // ---- converged part: pack the rays of the next iteration into the queues.  Lane rank by wave ballot + popcount,
    // wave offset by an LDS atomic, one global (returning) atomic per block and queue on the block's shard.
    {
        __shared__ uint32_t blkCount[Q_COUNT], blkBase[Q_COUNT];
        if (tid < Q_COUNT) blkCount[tid] = 0;
        __syncthreads();
        const bool want[Q_COUNT] = {emitPrimary, emitScatter, emitShadow && shadowPacket, emitShadow && !shadowPacket};
        uint32_t waveOff[Q_COUNT], rank[Q_COUNT];
#pragma unroll
        for (int q = 0; q < Q_COUNT; q++) {
            unsigned long long mask = __ballot(want[q]);
            rank[q] = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            uint32_t off = 0;
            if (mask != 0ull && lane == 0) off = atomicAdd(&blkCount[q], (uint32_t)__popcll(mask));
            waveOff[q] = shu(off, 0);
        }
        __syncthreads();
        STAMP(11);
        const uint32_t shard = blockIdx.x % PRT_QSHARDS;
        if (tid < Q_COUNT && blkCount[tid] != 0u) blkBase[tid] = atomicAdd(&A.qWork[4 + tid * PRT_QSHARDS + shard], blkCount[tid]);
        __syncthreads();
        STAMP(12);
        const uint32_t owner = gs;
#pragma unroll
        for (int q = 0; q < Q_COUNT; q++) {
            if (!want[q]) continue;
            uint32_t idx = shard * A.shardCap + blkBase[q] + waveOff[q] + rank[q];
            uint32_t bits = owner | (q == Q_PRIMARY ? (reverseBits << 26) : 0u) | (q >= Q_OCC_PACKET ? (lightSet << 29) : 0u);
            __builtin_nontemporal_store(bits, &A.qE[q][idx]);
        }
    }

    STAMP(4);
    // ---- store state
    if (inRange && ((info >> 20) & 0xfu) != PH_DONE) {
        if (phase == PH_WAIT_BOUNCE && slot < alive) {
            nt_store4(&A.S0[gs], make_float4(pos.x, pos.y, pos.z, asf(props.mat)));
            nt_store4(&A.S1[gs], make_float4(normal.x, normal.y, normal.z, asf(sflags | (lightSet ? SLOT_LIGHT_SET : 0u))));
            nt_store4(&A.S2[gs], make_float4(ndir.x, ndir.y, ndir.z, 0.0f));
            nt_store4(&A.S3[gs], make_float4(beta.x, beta.y, beta.z, 0.0f));
            if (ENV && envSampled) {
                nt_store4(&A.S5[gs], make_float4(envL.x, envL.y, envL.z, 0.0f));
                nt_store4(&A.S6[gs], make_float4(envI.x, envI.y, envI.z, 0.0f));
            }
        }
        // a slot's result is stored while the slot is alive and once more in the iteration its path ends
        if (phase == PH_WAIT_BOUNCE && slot < (aliveAtEntry > alive ? aliveAtEntry : alive))
            nt_store4(&A.S4[gs], make_float4(result.x, result.y, result.z, 0.0f));
        if (phase == PH_WAIT_PRIMARY) nt_store4(&A.S2[gs], make_float4(ndir.x, ndir.y, ndir.z, 0.0f));
        if (slot == 0) {
            A.gInfo[g] = (pk & 0xffu) | ((depth & 0xffu) << 8) | ((alive & 0xfu) << 16) | (phase << 20) | ((alive0 & 0xfu) << 24);
            A.gRng[g] = rng;
            A.gColor[g] = make_float4(color.x, color.y, color.z, 0.0f);
        }
    }
    STAMP(5);
#ifdef PRT_STAMP
    if (lane == 0)
        for (int k = 0; k < 13; k++) atomicAdd(&A.counters[8 + k], stampAcc[k]);
    if (lane == 0) atomicAdd(&A.counters[23], 1ull);
#endif
    // ---- statistics: block-level sums in LDS, then one atomic per block and counter on one of PRT_STAT_SHARDS copies
    // (per-wave atomics on ONE address would be ~0.5 M same-address atomics per launch; they serialise at the memory side)
    {
        __shared__ uint32_t blkStat[5];
        if (tid < 5) blkStat[tid] = 0;
        __syncthreads();
        uint32_t r = wave_sum((uint32_t)nRays), o = wave_sum((uint32_t)nOccl), px = wave_sum((uint32_t)nPx);
        uint32_t nh = COUNT ? wave_sum(tr.nHit) : 0u, nt = COUNT ? wave_sum(tr.nTap) : 0u;
        if (lane == 0) {
            if (r) atomicAdd(&blkStat[0], r);
            if (o) atomicAdd(&blkStat[1], o);
            if (nh) atomicAdd(&blkStat[2], nh);
            if (nt) atomicAdd(&blkStat[3], nt);
            if (px) atomicAdd(&blkStat[4], px);
        }
        __syncthreads();
        if (tid < 5 && blkStat[tid] != 0u) {
            const int slotOf[5] = {0, 1, 4, 5, 6};
            atomicAdd(&A.counters[(blockIdx.x % PRT_STAT_SHARDS) * PRT_STAT_STRIDE + slotOf[tid]], (unsigned long long)blkStat[tid]);
        }
    }
}

// Assigns pixels to the groups of a pass (tile-major order, row-major inside a 16x16 tile, main.cpp:132-138) and
// seeds their generators (the build's per-pixel state, replacing random.h:15-17).
__global__ void init_groups_kernel(WfArgs A)
{
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= A.groupCount) return;
    const uint32_t tile = A.p.tileSize, tile2 = tile * tile;
    const uint32_t tl = g / tile2, pix = g - tl * tile2; // tl-th tile of this pipeline
    const uint32_t tq = A.workBase / tile2 + tl * A.partCount + A.partIndex;
    uint32_t gt;
    bool ok = true;
    if (A.fullWidth) {
        gt = A.firstOwned + tq * A.p.nranks;
    } else {
        uint32_t qx = tq % A.rtnx, qy = tq / A.rtnx;
        gt = (A.rty0 + qy) * A.tilesXImage + (A.rtx0 + qx);
        ok = (gt % A.p.nranks) == A.p.rank;
    }
    uint32_t tx = gt % A.tilesXImage, ty = gt / A.tilesXImage;
    uint32_t x = tx * tile + pix % tile, y = ty * tile + pix / tile;
    if (x < A.x0 || x > A.x1 || y < A.y0 || y > A.y1) ok = false;
    A.gPixel[g] = ok ? (x | (y << 16)) : 0xffffffffu;
    A.gRng[g] = ok ? pixel_seed(x, y, A.cam.width, A.p.seed) : 0u;
    A.gInfo[g] = (uint32_t)PH_START << 20;
    A.gColor[g] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// Rays of one (sharded) queue.  An entry names the owner slot; the ray itself is rebuilt from the slot's state exactly as
// the reference builds it (camera.cpp:64-66; path_tracer.cpp:209-211 / 236-238; :269-273).  Results go to the owner.
template <int MODE>
struct QueueSrc {
    const uint32_t* qe;
    uint32_t shardCount[PRT_QSHARDS];
    uint32_t shardCap;
    uint32_t n;
    uint32_t* cur;
    const float4* S0;
    const float4* S2;
    const float4* S5; // per-slot light direction (environment light) or NULL (the scene's directional light)
    float4* hitA;
    uint2* hitB;
    uint32_t* occl;
    Vec3 camPos, lightDir;
    float kFar;
    uint32_t owner; // of the lane's current ray
    __device__ __forceinline__ uint32_t count() const { return n; }
    __device__ __forceinline__ uint32_t* cursor() const { return cur; }
    __device__ __forceinline__ void load(uint32_t i, Vec3& org, Vec3& dir, float& maxT, uint32_t& rev)
    {
        // virtual index -> (shard, offset)
        uint32_t v = i, base = 0;
#pragma unroll
        for (int k = 0; k < PRT_QSHARDS - 1; k++) {
            bool next = v >= shardCount[k];
            v -= next ? shardCount[k] : 0u;
            base += next ? shardCap : 0u;
            if (!next) break;
        }
        const uint32_t bits = __builtin_nontemporal_load(&qe[base + v]);
        owner = bits & 0x3ffffffu;
        rev = (bits >> 26) & 7u;
        if (MODE == PRT_MODE_PACKET) {
            float4 s2 = nt_load4(&S2[owner]);
            org = camPos;
            dir = mk3(s2.x, s2.y, s2.z);
            maxT = 100000.0f; // camera.cpp:64
        } else if (MODE == PRT_MODE_SINGLE) {
            float4 s0 = nt_load4(&S0[owner]), s2 = nt_load4(&S2[owner]);
            org = mk3(s0.x, s0.y, s0.z);
            dir = mk3(s2.x, s2.y, s2.z);
            maxT = kFar; // path_tracer.cpp:270
        } else {
            float4 s0 = nt_load4(&S0[owner]);
            Vec3 L = mk3(0.0f, 0.0f, 0.0f);
            if ((bits >> 29) & 1u) {
                if (S5) {
                    float4 l5 = nt_load4(&S5[owner]);
                    L = mk3(l5.x, l5.y, l5.z);
                } else {
                    L = lightDir;
                }
            }
            org = add3(mk3(s0.x, s0.y, s0.z), scale3(kFar, L)); // path_tracer.cpp:210, 237
            dir = mk3(-L.x, -L.y, -L.z);
            maxT = kFar - 0.0008f; // :209, 236
        }
    }
    __device__ __forceinline__ void store_hit(uint32_t, const DevHit& h) const
    {
        nt_store4(&hitA[owner], make_float4(h.t, h.i, h.j, h.k));
        hitB[owner] = make_uint2(h.primId, h.meshId);
    }
    __device__ __forceinline__ void store_occ(uint32_t, bool occ) const { occl[owner] = occ ? 1u : 0u; }
};

// Persistent lanes over one ray queue.  MODE = the queue id = the traversal mode.
// Waves per SIMD the trace kernels are compiled for.  Without a bound the compiler takes 112 SGPRs, which caps a SIMD at 6 waves
// (800 SGPRs per SIMD, MI355X_MICROARCH.md "Occupancy API"); 7 gives 94 SGPRs / 72 VGPRs with no spills and the best frame time
// (C3: unbounded 533 ms, 8: 521 with 9 SGPR spills, 7: 514, 6 and 5: 535).
#ifndef PRT_TRACE_WAVES
#define PRT_TRACE_WAVES 7
#endif
template <int MODE, bool COUNT>
#if PRT_TRACE_WAVES
__global__ __launch_bounds__(PRT_BLOCK, PRT_TRACE_WAVES) void trace_kernel(WfArgs A)
#else
__global__ __launch_bounds__(PRT_BLOCK) void trace_kernel(WfArgs A)
#endif
{
    // the packet traversal stacks (reference, entry distance) pairs: half as many entries in the same 16 KB, so that its
    // blocks do not crowd the other kernels' out of the CU's LDS
    constexpr int NLDS = (MODE == PRT_MODE_PACKET) ? PRT_STACK_LDS_PACKET : PRT_STACK_LDS;
    __shared__ uint32_t ldsRef[NLDS * PRT_BLOCK];
    __shared__ float ldsT[(MODE == PRT_MODE_PACKET ? NLDS : 1) * PRT_BLOCK];
    const uint32_t tid = threadIdx.x;
    const StackT<NLDS> st{(lds_u32*)&ldsRef[tid], (lds_f32*)&ldsT[tid], A.spill + ((size_t)blockIdx.x * PRT_BLOCK + tid), A.spillStride};
    QueueSrc<MODE> src;
    src.qe = A.qE[MODE];
    src.n = 0;
#pragma unroll
    for (int k = 0; k < PRT_QSHARDS; k++) {
        src.shardCount[k] = A.qWork[4 + MODE * PRT_QSHARDS + k];
        src.n += src.shardCount[k];
    }
    src.shardCap = A.shardCap;
    src.cur = &A.qWork[MODE];
    src.S0 = A.S0;
    src.S2 = A.S2;
    src.S5 = A.sc.hasEnv ? A.S5 : nullptr;
    src.camPos = mk3(A.cam.pos[0], A.cam.pos[1], A.cam.pos[2]);
    src.lightDir = mk3(A.sc.lightDir[0], A.sc.lightDir[1], A.sc.lightDir[2]);
    src.kFar = 2.0f * A.sc.radius; // path_tracer.cpp:192
    src.hitA = A.hitA;
    src.hitB = A.hitB;
    src.occl = A.occl;
    src.owner = 0;
    Traffic tr{0, 0, 0, 0};
    uint32_t overflow = 0;
    trace_loop<MODE, COUNT>(A.sc, src, st, tr, overflow);
    unsigned long long* C = A.counters;
#ifdef PRT_PROFILE
    if ((tid & 63u) == 0) {
        atomicAdd(&C[16 + MODE * 3], tr.nBox);
        atomicAdd(&C[17 + MODE * 3], tr.nTri);
        atomicAdd(&C[18 + MODE * 3], (unsigned long long)tr.nTap);
    }
#endif
    if (COUNT) {
        // 64-bit per-wave sums (a persistent lane can count more than 2^32 box tests)
        unsigned long long b = tr.nBox, t = tr.nTri, p = tr.nTap;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            b += (unsigned long long)__shfl_xor((long long)b, o, 64);
            t += (unsigned long long)__shfl_xor((long long)t, o, 64);
            p += (unsigned long long)__shfl_xor((long long)p, o, 64);
        }
        if ((tid & 63u) == 0) {
            if (b) atomicAdd(&C[2], b);
            if (t) atomicAdd(&C[3], t);
            if (p) atomicAdd(&C[5], p);
        }
    }
    if (overflow) atomicAdd(&C[7], 1ull);
}

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
            Traffic tr{0, 0, 0, 0};
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
    const StackT<PRT_STACK_LDS> st{(lds_u32*)&ldsRef[tid], (lds_f32*)&ldsT[tid], A.spill + ((size_t)blockIdx.x * PRT_BLOCK + tid), A.spillStride};
    GbufSrc src{&A};
    Traffic tr{0, 0, 0, 0};
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
        o.t = h.t; o.i = h.i; o.j = h.j; o.k = h.k; o.primId = h.primId; o.meshId = h.meshId;
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
    const StackT<NLDS> st{(lds_u32*)&ldsRef[tid], (lds_f32*)&ldsT[tid], A.spill + ((size_t)blockIdx.x * PRT_BLOCK + tid), A.spillStride};
    ArraySrc src{&A, MODE};
    Traffic tr{0, 0, 0, 0};
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

// One pipeline of a pass: its groups, queues and streams.
struct PartRun {
    WfArgs A;
    uint32_t shadeBlocks = 0;
    hipStream_t main = nullptr, side[PRT_SIDE_STREAMS] = {};
    hipEvent_t fork = nullptr, join[PRT_SIDE_STREAMS] = {};
    int index = 0;
};

// One iteration of one pipeline.  The four trace kernels of an iteration are independent of each other: the scatter kernel
// (the longest) runs on the main stream, the other three on the side stream beside it, so that the tail of one persistent
// grid is filled by the next (each kernel has its own stack-spill area); the next shade kernel waits for all four.  The
// other pipeline's kernels fill what this one leaves idle -- above all the time its shade kernel would hold the GPU alone.
template <bool COUNT>
static void wf_iteration(const PartRun& P, uint32_t traceBlocks)
{
    const WfArgs& A = P.A;
    (void)hipMemsetAsync(A.qWork, 0, (4 + Q_COUNT * PRT_QSHARDS) * sizeof(uint32_t), P.main); // claim cursors + shard counts
    if (A.sc.hasEnv) hipLaunchKernelGGL((shade_kernel<COUNT, true>), dim3(P.shadeBlocks), dim3(PRT_BLOCK), 0, P.main, A);
    else hipLaunchKernelGGL((shade_kernel<COUNT, false>), dim3(P.shadeBlocks), dim3(PRT_BLOCK), 0, P.main, A);
    (void)hipEventRecord(P.fork, P.main);
    const size_t spillWords = (size_t)A.spillStride * 2 * (PRT_STACK_MAX - PRT_STACK_LDS_PACKET);
    WfArgs B = A;
    for (int k = 0; k < PRT_SIDE_STREAMS; k++) (void)hipStreamWaitEvent(P.side[k], P.fork, 0);
    // the three smaller kernels go round the side streams (1: one after the other, 3: each on its own)
    B.spill = A.spill + 1 * spillWords;
    hipLaunchKernelGGL((trace_kernel<Q_OCC_PACKET, COUNT>), dim3(traceBlocks), dim3(PRT_BLOCK), 0, P.side[0 % PRT_SIDE_STREAMS], B);
    B.spill = A.spill + 2 * spillWords;
    hipLaunchKernelGGL((trace_kernel<Q_PRIMARY, COUNT>), dim3(traceBlocks), dim3(PRT_BLOCK), 0, P.side[1 % PRT_SIDE_STREAMS], B);
    B.spill = A.spill + 3 * spillWords;
    hipLaunchKernelGGL((trace_kernel<Q_OCC_SINGLE, COUNT>), dim3(traceBlocks), dim3(PRT_BLOCK), 0, P.side[2 % PRT_SIDE_STREAMS], B);
    for (int k = 0; k < PRT_SIDE_STREAMS; k++) (void)hipEventRecord(P.join[k], P.side[k]);
    hipLaunchKernelGGL((trace_kernel<Q_SCATTER, COUNT>), dim3(traceBlocks), dim3(PRT_BLOCK), 0, P.main, A);
    for (int k = 0; k < PRT_SIDE_STREAMS; k++) (void)hipStreamWaitEvent(P.main, P.join[k], 0);
}

extern "C" {

const char* prt_hip_last_error(void) { return g_err.c_str(); }

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
    for (int k = 0; k < PRT_PARTS; k++) HIP_TRY(hipEventCreateWithFlags(&c->evFork[k], hipEventDisableTiming));
    for (int k = 0; k < PRT_PARTS * PRT_SIDE_STREAMS; k++) HIP_TRY(hipEventCreateWithFlags(&c->evJoin[k], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->evStart, hipEventDisableTiming));
    for (int k = 0; k < PRT_PARTS; k++) HIP_TRY(hipEventCreateWithFlags(&c->evDone[k], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->evIn, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->evOut, hipEventDisableTiming));
    for (int k = 0; k < (1 + PRT_SIDE_STREAMS) * PRT_PARTS - 1; k++) HIP_TRY(hipStreamCreate(&c->aux[k]));
    HIP_TRY(hipMalloc(&c->work, PRT_PARTS * PRT_WORK_WORDS * sizeof(uint32_t)));
    HIP_TRY(hipMemset(c->work, 0, PRT_PARTS * PRT_WORK_WORDS * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&c->counters, PRT_STAT_SHARDS * PRT_STAT_STRIDE * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->counters, 0, PRT_STAT_SHARDS * PRT_STAT_STRIDE * sizeof(unsigned long long)));
    return PRT_HIP_OK;
}

// The library runs four streams side by side.  A host process that owns more streams of its own (RCCL, a framework) should
// raise the HIP runtime's default of four hardware queues BEFORE its first HIP call (GPU_MAX_HW_QUEUES); the
// library does not touch the process environment.
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
    for (int k = 0; k < (1 + PRT_SIDE_STREAMS) * PRT_PARTS - 1; k++)
        if (c->aux[k]) (void)hipStreamDestroy(c->aux[k]);
    for (int k = 0; k < PRT_PARTS; k++)
        if (c->evFork[k]) (void)hipEventDestroy(c->evFork[k]);
    for (int k = 0; k < PRT_PARTS * PRT_SIDE_STREAMS; k++)
        if (c->evJoin[k]) (void)hipEventDestroy(c->evJoin[k]);
    if (c->evStart) (void)hipEventDestroy(c->evStart);
    for (int k = 0; k < PRT_PARTS; k++)
        if (c->evDone[k]) (void)hipEventDestroy(c->evDone[k]);
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

// Flattens Scene -> Bvh -> Mesh (scene.h:61-71, bvh.h:113-119, mesh.h:87-104) into the arrays of DevScene.
int prt_hip_upload_scene(prt_hip_ctx* c, const prt_scene_desc* s)
{
    if (!c || !s) return fail(PRT_HIP_EINVAL, "NULL argument");
    if (s->meshCount == 0 || s->meshCount > PRT_MAX_BVH) return fail(PRT_HIP_EINVAL, "meshCount must be 1..8");
    HIP_TRY(hipSetDevice(c->device));
    free_scene(c);

    std::vector<float4> wnodes, tris, shade, bump, mats, alpha;
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

    bool anyBump = false;
    for (uint32_t m = 0; m < s->meshCount; m++)
        for (uint32_t k = 0; k < s->meshes[m].materialCount; k++)
            if (s->meshes[m].materials[k].bumpMap >= 0) anyBump = true;

    for (uint32_t m = 0; m < s->meshCount; m++) {
        const prt_mesh_desc& md = s->meshes[m];
        if (!md.nodes || !md.primRemapping || !md.indices || !md.positions || !md.primMaterial || !md.materials || md.nodeCount == 0)
            return fail(PRT_HIP_EINVAL, "mesh descriptor has NULL arrays");
        const uint32_t triBase = (uint32_t)(tris.size() / 3);
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
            if ((size_t)triBase + md.primCount >= (1u << 27) || nextWide >= (1u << 30)) return fail(PRT_HIP_EINVAL, "scene too large for 32-bit child references");
            auto refOf = [&](uint32_t i) -> uint32_t {
                const prt_bvh_node& n = md.nodes[i];
                if (n.primCount == 0xf) return wideIndex[i];
                return PRT_REF_LEAF | ((triBase + n.primOrSecondNodeIndex) << 4) | n.primCount;
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
        // leaf triangles in primRemapping order (TriangleVector, bvh.cpp:245-296)
        for (uint32_t k = 0; k < md.primCount; k++) {
            uint32_t prim = md.primRemapping[k];
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
                alpha.push_back(make_float4(u[0], u[1], u[2], u[3]));
                alpha.push_back(make_float4(u[4], u[5], ubits((uint32_t)mt.diffuseMap), 0.0f));
                alphaRef = (uint32_t)(alpha.size() / 2);
            }
            HVec3 p0 = P(v0), p1 = P(v1), p2 = P(v2);
            tris.push_back(make_float4(p0.x, p0.y, p0.z, ubits(prim)));
            tris.push_back(make_float4(p1.x, p1.y, p1.z, ubits(alphaRef)));
            tris.push_back(make_float4(p2.x, p2.y, p2.z, 0.0f));
        }
        // shading records in mesh order (Mesh::getSurfaceProperties, mesh.cpp:311-364)
        for (uint32_t prim = 0; prim < md.primCount; prim++) {
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
    if ((rc = upload_vec(c, tris, &sc.tris))) return rc;
    if ((rc = upload_vec(c, shade, &sc.shade))) return rc;
    if ((rc = upload_vec(c, bump, &sc.bump))) return rc;
    if ((rc = upload_vec(c, mats, &sc.mats))) return rc;
    if ((rc = upload_vec(c, alpha, &sc.alpha))) return rc;
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

static int ensure_launch_resources(prt_hip_ctx* c, uint32_t blocks)
{
    uint32_t threads = blocks * PRT_BLOCK;
    if (threads > c->spillThreads) {
        if (c->spill) (void)hipFree(c->spill);
        c->spill = nullptr;
        // one area per concurrently running trace kernel: four per pipeline
        HIP_TRY(hipMalloc(&c->spill, 4 * PRT_PARTS * (size_t)threads * 2 * (PRT_STACK_MAX - PRT_STACK_LDS_PACKET) * sizeof(uint32_t)));
        c->spillThreads = threads;
    }
    return PRT_HIP_OK;
}

static int persistent_blocks(prt_hip_ctx* c)
{
    if (c->blocksPerCU == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<Q_SCATTER, false>, PRT_BLOCK, 0) != hipSuccess || nb <= 0) nb = 2;
        // Up to eight trace kernels are in flight (two pipelines x four queues) and a CU holds 8 blocks of 256 threads: 4 per
        // kernel and CU leaves room for the other pipeline's kernels (measured on C3: 1: 933 ms, 2: 627, 3: 594, 4: 588,
        // 5: 599, 6: 608, 8: 631)
        c->blocksPerCU = std::min(nb, PRT_TRACE_BPC);
    }
    return c->computeUnits * c->blocksPerCU;
}

// Carves the wavefront state of the pipelines of a pass (groups[k] pixel groups each) out of one allocation.
static int wf_layout(prt_hip_ctx* c, const uint32_t* groups, int parts, bool env, WfArgs* A)
{
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    auto entries = [](size_t slots) { return slots + (size_t)PRT_QSHARDS * PRT_BLOCK + PRT_BLOCK; };
    size_t need = 0;
    for (int k = 0; k < parts; k++) {
        const size_t g = groups[k], slots = g * 8;
        need += 3 * al(g * sizeof(uint32_t)) + al(g * sizeof(float4));
        need += (env ? 7 : 5) * al(slots * sizeof(float4));                                            // S0..S4 (S5, S6)
        need += al(slots * sizeof(float4)) + al(slots * sizeof(uint2)) + al(slots * sizeof(uint32_t)); // hits, occlusion
        need += Q_COUNT * al(entries(slots) * sizeof(uint32_t));
    }
    if (need > c->wfBytes) {
        if (c->wfBuffer) (void)hipFree(c->wfBuffer);
    if (c->frameArgs) (void)hipFree(c->frameArgs);
    prt_gather_release(c);
        c->wfBuffer = nullptr;
        c->wfBytes = 0;
        HIP_TRY(hipMalloc(&c->wfBuffer, need));
        c->wfBytes = need;
    }
    char* p = (char*)c->wfBuffer;
    auto take = [&](size_t bytes) { char* r = p; p += al(bytes); return r; };
    for (int k = 0; k < parts; k++) {
        const size_t g = groups[k], slots = g * 8;
        A[k].gRng = (uint32_t*)take(g * sizeof(uint32_t));
        A[k].gInfo = (uint32_t*)take(g * sizeof(uint32_t));
        A[k].gPixel = (uint32_t*)take(g * sizeof(uint32_t));
        A[k].gColor = (float4*)take(g * sizeof(float4));
        A[k].S0 = (float4*)take(slots * sizeof(float4));
        A[k].S1 = (float4*)take(slots * sizeof(float4));
        A[k].S2 = (float4*)take(slots * sizeof(float4));
        A[k].S3 = (float4*)take(slots * sizeof(float4));
        A[k].S4 = (float4*)take(slots * sizeof(float4));
        A[k].S5 = env ? (float4*)take(slots * sizeof(float4)) : nullptr;
        A[k].S6 = env ? (float4*)take(slots * sizeof(float4)) : nullptr;
        A[k].hitA = (float4*)take(slots * sizeof(float4));
        A[k].hitB = (uint2*)take(slots * sizeof(uint2));
        A[k].occl = (uint32_t*)take(slots * sizeof(uint32_t));
        for (int q = 0; q < Q_COUNT; q++) A[k].qE[q] = (uint32_t*)take(entries(slots) * sizeof(uint32_t));
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
        if (const char* e = getenv("PRT_FRAME_BPC")) c->frameBlocksPerCU = std::max(1, std::min(8, atoi(e)));
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

static int render_frame_kernel(prt_hip_ctx* c, const WfArgs& W0, uint64_t totalWork, hipStream_t s)
{
    const prt_render_params* p = &W0.p;
    FrameArgs A{};
    A.sc = W0.sc; A.cam = W0.cam; A.p = W0.p;
    A.x0 = W0.x0; A.y0 = W0.y0; A.x1 = W0.x1; A.y1 = W0.y1;
    A.tilesXImage = W0.tilesXImage;
    A.rtx0 = W0.rtx0; A.rty0 = W0.rty0; A.rtnx = W0.rtnx; A.rtny = W0.rtny;
    A.fullWidth = W0.fullWidth; A.firstOwned = W0.firstOwned;
    A.totalWork = (uint32_t)totalWork;
    A.totalChunks = (uint32_t)((totalWork + PRT_CHUNK - 1) / PRT_CHUNK);
    A.rgb = W0.rgb;
    A.counters = c->counters;
    A.ctrl = c->work;
    const uint32_t resident = (uint32_t)frame_blocks(c);
    const uint32_t blocks = std::max<uint32_t>(1, std::min<uint32_t>(resident, A.totalChunks));
    // a block holds at most rowsPerBlock rows at a time and takes one row per visit to the shade role: all of a small launch's
    // rows are in flight at once (a row that starts late costs a whole chain of rounds), spread evenly over the blocks
    A.rowsPerBlock = std::max<uint32_t>(1, std::min<uint32_t>(PRT_POOL_CHUNKS, (A.totalChunks + blocks - 1) / blocks));
    A.spreadRows = (A.totalChunks < (uint64_t)blocks * PRT_POOL_CHUNKS * 2 && totalWork % PRT_CHUNK == 0) ? 1u : 0u;
    if (const char* e = getenv("PRT_SPREAD")) A.spreadRows = atoi(e) && totalWork % PRT_CHUNK == 0;
    int rc = ensure_launch_resources(c, std::max<uint32_t>(resident, (uint32_t)persistent_blocks(c)));
    if (rc) return rc;
    A.spill = c->spill;
    A.spillStride = c->spillThreads;
    if ((rc = frame_layout(c, resident, c->sc.hasEnv != 0, A))) return rc;
    HIP_TRY(hipMemsetAsync(c->work, 0, PRT_PARTS * PRT_WORK_WORDS * sizeof(uint32_t), s));
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
    // The pipeline always runs on the context's own streams (the main one and the three that carry the concurrent trace
    // kernels; a foreign stream can share a hardware queue with one of those and serialise them -- 8 % on C3).  A caller's
    // stream is ordered around it with two events: work queued on it before this call is finished before the first kernel
    // starts, and whatever the caller queues next waits for the last one.
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
    WfArgs A{};
    A.sc = c->sc;
    A.cam = c->cam;
    A.p = *p;
    A.x0 = x0; A.y0 = y0; A.x1 = x1; A.y1 = y1;
    const uint32_t T = p->tileSize;
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
    A.qWork = c->work;
    A.counters = c->counters;
    static const bool useWavefrontPipeline = getenv("PRT_SCHED") && !strcmp(getenv("PRT_SCHED"), "wave"); // A/B during development
    if (!useWavefrontPipeline) {
        if (p->tileSize * p->tileSize > (1u << 20)) return fail(PRT_HIP_EINVAL, "tile too large");
        HIP_TRY(hipMemsetAsync(c->counters, 0, PRT_STAT_SHARDS * PRT_STAT_STRIDE * sizeof(unsigned long long), s));
        if (c->ringUsed == PRT_TIMING_RING) fold_timing(c);
        if (!c->evT0[c->ringUsed]) HIP_TRY(hipEventCreate(&c->evT0[c->ringUsed]));
        if (!c->evT1[c->ringUsed]) HIP_TRY(hipEventCreate(&c->evT1[c->ringUsed]));
        hipEvent_t e0 = c->evT0[c->ringUsed], e1 = c->evT1[c->ringUsed];
        c->ringUsed++;
        HIP_TRY(hipEventRecord(e0, s));
        int frc = render_frame_kernel(c, A, totalWork, s);
        if (frc) return frc;
        hipError_t fle = hipGetLastError();
        if (fle != hipSuccess) return fail(PRT_HIP_ELAUNCH, std::string("frame_kernel launch: ") + hipGetErrorString(fle));
        HIP_TRY(hipEventRecord(e1, s));
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
    const uint32_t traceBlocks = (uint32_t)persistent_blocks(c);
    int rc = ensure_launch_resources(c, traceBlocks);
    if (rc) return rc;
    A.spillStride = c->spillThreads;
    // A pass holds at most 2^26 slot references (the 26-bit owner field of a queued ray) per pipeline and whole tiles.
    const uint32_t tile2 = T * T;
    if (tile2 > (1u << 20)) return fail(PRT_HIP_EINVAL, "tile too large");
    const uint64_t totalTiles = totalWork / tile2;
    const uint64_t kMaxGroupsPerPass = 4u << 20;
    const uint64_t passTiles = std::max<uint64_t>(1, std::min<uint64_t>(std::max<uint64_t>(totalTiles, 1), kMaxGroupsPerPass / tile2));
    // two pipelines as soon as each gets a few tiles; tiny launches keep one
    const int parts = (std::min<uint64_t>(passTiles, totalTiles) >= 4 * PRT_PARTS) ? PRT_PARTS : 1;
    PartRun P[PRT_PARTS];
    const size_t spillWords = (size_t)A.spillStride * 2 * (PRT_STACK_MAX - PRT_STACK_LDS_PACKET);
    {
        uint32_t maxGroups[PRT_PARTS];
        WfArgs L[PRT_PARTS];
        for (int k = 0; k < parts; k++) maxGroups[k] = (uint32_t)(((passTiles - k + parts - 1) / parts) * tile2);
        if ((rc = wf_layout(c, maxGroups, parts, c->sc.hasEnv != 0, L))) return rc;
        for (int k = 0; k < parts; k++) {
            P[k].A = A;
            P[k].A.gRng = L[k].gRng; P[k].A.gInfo = L[k].gInfo; P[k].A.gPixel = L[k].gPixel; P[k].A.gColor = L[k].gColor;
            P[k].A.S0 = L[k].S0; P[k].A.S1 = L[k].S1; P[k].A.S2 = L[k].S2; P[k].A.S3 = L[k].S3; P[k].A.S4 = L[k].S4;
            P[k].A.S5 = L[k].S5; P[k].A.S6 = L[k].S6;
            P[k].A.hitA = L[k].hitA; P[k].A.hitB = L[k].hitB; P[k].A.occl = L[k].occl;
            for (int q = 0; q < Q_COUNT; q++) P[k].A.qE[q] = L[k].qE[q];
            P[k].A.qWork = c->work + (size_t)k * PRT_WORK_WORDS;
            P[k].A.spill = c->spill + (size_t)k * 4 * spillWords;
            P[k].A.partIndex = (uint32_t)k;
            P[k].A.partCount = (uint32_t)parts;
            // streams of pipeline k: aux[k*(1+S) - 1] (main; pipeline 0 uses the context's stream) and the S after it
            P[k].main = (k == 0) ? s : c->aux[k * (1 + PRT_SIDE_STREAMS) - 1];
            for (int j = 0; j < PRT_SIDE_STREAMS; j++) {
                P[k].side[j] = c->aux[k * (1 + PRT_SIDE_STREAMS) + j];
                P[k].join[j] = c->evJoin[k * PRT_SIDE_STREAMS + j];
            }
            P[k].fork = c->evFork[k];
            P[k].index = k;
        }
    }

    HIP_TRY(hipMemsetAsync(c->counters, 0, PRT_STAT_SHARDS * PRT_STAT_STRIDE * sizeof(unsigned long long), s));
    if (c->ringUsed == PRT_TIMING_RING) fold_timing(c);
    if (!c->evT0[c->ringUsed]) HIP_TRY(hipEventCreate(&c->evT0[c->ringUsed]));
    if (!c->evT1[c->ringUsed]) HIP_TRY(hipEventCreate(&c->evT1[c->ringUsed]));
    hipEvent_t ev0 = c->evT0[c->ringUsed], ev1 = c->evT1[c->ringUsed];
    c->ringUsed++;
    HIP_TRY(hipEventRecord(ev0, s));
    const uint32_t iterations = (p->samples / 8) * (1 + p->maxDepth) + 1;
    for (uint64_t baseTile = 0; baseTile < totalTiles; baseTile += passTiles) {
        const uint64_t tilesNow = std::min<uint64_t>(passTiles, totalTiles - baseTile);
        // pipeline 1 starts after everything queued on the main stream so far (the previous pass included)
        if (parts > 1) {
            HIP_TRY(hipEventRecord(c->evStart, s));
            for (int k = 1; k < parts; k++) HIP_TRY(hipStreamWaitEvent(P[k].main, c->evStart, 0));
        }
        int live = 0;
        for (int k = 0; k < parts; k++) {
            const uint64_t tilesOfPart = (tilesNow > (uint64_t)k) ? (tilesNow - k + parts - 1) / parts : 0;
            P[k].A.workBase = (uint32_t)(baseTile * tile2);
            P[k].A.groupCount = (uint32_t)(tilesOfPart * tile2);
            P[k].shadeBlocks = (uint32_t)(((uint64_t)P[k].A.groupCount * 8 + PRT_BLOCK - 1) / PRT_BLOCK);
            P[k].A.shardCap = ((P[k].shadeBlocks + PRT_QSHARDS - 1) / PRT_QSHARDS) * PRT_BLOCK;
            if (P[k].A.groupCount == 0) continue;
            live = k + 1;
            hipLaunchKernelGGL(init_groups_kernel, dim3((P[k].A.groupCount + 255) / 256), dim3(256), 0, P[k].main, P[k].A);
        }
        for (uint32_t it = 0; it < iterations; it++) {
            for (int k = 0; k < live; k++) {
                if (P[k].A.groupCount == 0) continue;
                if (p->countTraffic) wf_iteration<true>(P[k], traceBlocks);
                else wf_iteration<false>(P[k], traceBlocks);
            }
        }
        for (int k = 1; k < parts; k++) {
            HIP_TRY(hipEventRecord(c->evDone[k], P[k].main));
            HIP_TRY(hipStreamWaitEvent(s, c->evDone[k], 0));
        }
    }
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(PRT_HIP_ELAUNCH, std::string("wavefront launch: ") + hipGetErrorString(le));
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
    const uint32_t blocks = std::min<uint32_t>(want, (uint32_t)persistent_blocks(c));
    int rc = ensure_launch_resources(c, (uint32_t)persistent_blocks(c));
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->counters, 0, PRT_STAT_SHARDS * PRT_STAT_STRIDE * sizeof(unsigned long long), s));
    HIP_TRY(hipMemsetAsync(c->work, 0, PRT_PARTS * PRT_WORK_WORDS * sizeof(uint32_t), s));
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
    return PRT_HIP_OK;
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
#ifdef PRT_STAMP
    {
        const char* names[13] = {"header", "consume-rest", "bounce-rest", "end+camera", "emit-writes", "store", "state-loads", "hit+surface",
                                 "bump", "mats+rng", "diffuse", "emit-sync1", "emit-sync2"};
        fprintf(stderr, "shade stamps (avg cycles per wave, %llu waves):", h[23]);
        for (int k = 0; k < 13; k++) fprintf(stderr, " %s %.0f", names[k], (double)h[8 + k] / (double)(h[23] ? h[23] : 1));
        fprintf(stderr, "\n");
    }
#endif
    if (c->frameLaunched) {
        uint32_t ctrl[8 + 16 * 8] = {0};
        HIP_TRY(hipMemcpy(ctrl, c->work, sizeof(ctrl), hipMemcpyDeviceToHost));
        if (ctrl[1]) {
            std::string msg = "frame kernel: scheduler watchdog fired (a workgroup waited for work that never came);";
            for (uint32_t k = 0; k < std::min<uint32_t>(ctrl[2], 8u); k++) {
                const uint32_t* D = ctrl + 8 + 16 * k;
                char line[256];
                snprintf(line, sizeof(line), " [block %u wave %u: ready %u live %u exhausted %u lock %u, %u groups wait for %u rays, tails %u %u %u %u heads %u %u %u %u]",
                         D[0], D[1], D[2], D[3], D[4], D[5], D[6], D[7], D[8], D[9], D[10], D[11], D[12], D[13], D[14], D[15]);
                msg += line;
            }
            return fail(PRT_HIP_ELAUNCH, msg);
        }
    }
#ifdef PRT_PROFILE
    if (h[14])
        fprintf(stderr, "frame profile: waves %llu, per wave: shade %.1f%% (%.0f calls) trace %.1f%% (%.0f calls) idle/decide %.1f%% of %.2f Mcycles\n", h[14],
                100.0 * h[8] / h[13], (double)h[11] / h[14], 100.0 * h[9] / h[13], (double)h[12] / h[14], 100.0 * h[10] / h[13], h[13] / 1e6 / h[14]);
    if (!h[14] && h[16 + 3])
        for (int m = 0; m < 4; m++)
            fprintf(stderr, "  wavefront trace mode %d: %.1f M loop turns, %.1f lanes with a ray per turn, %.0f kcycles per... total %.1f Gcycles in loops => %.0f cycles per turn\n", m,
                    h[16 + 3 * m] / 1e6, (double)h[17 + 3 * m] / (double)(h[16 + 3 * m] ? h[16 + 3 * m] : 1), 0.0, h[18 + 3 * m] * 1024.0 / 1e9,
                    h[18 + 3 * m] * 1024.0 / (double)(h[16 + 3 * m] ? h[16 + 3 * m] : 1));
    if (h[14]) {
        for (int m = 0; m < 4; m++)
            fprintf(stderr, "  trace mode %d: %.1f M loop turns, %.1f lanes with a ray per turn, %.1f Gcycles in the loops => %.0f cycles per turn\n", m, h[16 + 3 * m] / 1e6,
                    (double)h[17 + 3 * m] / (double)(h[16 + 3 * m] ? h[16 + 3 * m] : 1), h[18 + 3 * m] * 1024.0 / 1e9,
                    h[18 + 3 * m] * 1024.0 / (double)(h[16 + 3 * m] ? h[16 + 3 * m] : 1));
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
    st->stackOverflow = h[7];
    fold_timing(c);
    st->kernelMs = c->accLaunches ? c->lastMs : 0.0;
    st->kernelMsSum = c->accMs;
    st->kernelLaunches = c->accLaunches;
    c->accMs = c->lastMs = 0.0;
    c->accLaunches = 0;
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
    HIP_TRY(hipMemsetAsync(c->work, 0, PRT_PARTS * PRT_WORK_WORDS * sizeof(uint32_t), c->stream));
    RaysArgs A{c->sc, n, dorg.as<float>(), ddir.as<float>(), maxT, dh.as<prt_hit>(), c->work, c->spill, c->spillThreads, c->counters};
    blocks = std::min<uint32_t>(blocks, (uint32_t)c->spillThreads / PRT_BLOCK);
    if (mode == 0) hipLaunchKernelGGL(rays_kernel<PRT_MODE_SINGLE>, dim3(blocks), dim3(PRT_BLOCK), 0, c->stream, A);
    else if (mode == 1) hipLaunchKernelGGL(rays_kernel<PRT_MODE_PACKET>, dim3(blocks), dim3(PRT_BLOCK), 0, c->stream, A);
    else if (mode == 2) hipLaunchKernelGGL(rays_kernel<PRT_MODE_OCC_SINGLE>, dim3(blocks), dim3(PRT_BLOCK), 0, c->stream, A);
    else hipLaunchKernelGGL(rays_kernel<PRT_MODE_OCC_PACKET>, dim3(blocks), dim3(PRT_BLOCK), 0, c->stream, A);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(PRT_HIP_ELAUNCH, std::string("rays_kernel launch: ") + hipGetErrorString(le));
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
