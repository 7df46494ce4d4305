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

// ============================================================================ render kernel
struct RenderArgs {
    DevScene sc;
    DevCamera cam;
    prt_render_params p;
    uint32_t x0, y0, x1, y1;
    uint32_t tilesXImage;           // image width in tiles
    uint32_t rtx0, rty0, rtnx, rtny; // tile range covering the rectangle
    uint32_t fullWidth;             // rectangle spans whole image rows -> owned tiles are an arithmetic progression
    uint32_t firstOwned, ownedCount;
    uint32_t totalWork;             // ownedCount (or rect tiles) * tileSize^2
    float* rgb;
    uint32_t* work;                 // atomic work counter
    unsigned long long* counters;   // rays, occl, nBox, nTri, nHit, nTap, nPx, overflow
    uint32_t* spill;
    uint32_t spillStride;
};

template <bool COUNT>
__global__ __launch_bounds__(PRT_BLOCK) void render_kernel(RenderArgs A)
{
    __shared__ uint32_t ldsRef[PRT_STACK_LDS * PRT_BLOCK];
    __shared__ float ldsT[PRT_STACK_LDS * PRT_BLOCK];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, slot = lane & 7u, gbase = lane & ~7u;
    const Stack st{&ldsRef[tid], &ldsT[tid], A.spill + ((size_t)blockIdx.x * PRT_BLOCK + tid), A.spillStride};
    const DevScene& sc = A.sc;
    const DevCamera& cam = A.cam;
    const uint32_t tile = A.p.tileSize, tile2 = tile * tile;
    const uint32_t samples = A.p.samples, maxDepth = A.p.maxDepth, rrDepth = A.p.rrDepth;
    const float kPi = 3.14159265358979323846f;
    const float kFar = 2.0f * sc.radius;   // path_tracer.cpp:192
    const float kEpsilon = 0.0008f;        // :193
    const uint32_t lowerMask = (1u << slot) - 1u;

    Traffic tr{0, 0, 0, 0};
    uint32_t overflow = 0;
    unsigned long long nRays = 0, nOccl = 0, nPx = 0;

    for (;;) {
        uint32_t w = 0;
        if (slot == 0) w = atomicAdd(A.work, 1u);
        w = shu(w, gbase);
        if (w >= A.totalWork) break;
        // work item -> pixel: tile-major, row-major inside a tile (main.cpp:132-138 tiles)
        uint32_t tq = w / tile2, pix = w - tq * tile2;
        uint32_t gt;
        if (A.fullWidth) {
            gt = A.firstOwned + tq * A.p.nranks;
        } else {
            uint32_t qx = tq % A.rtnx, qy = tq / A.rtnx;
            gt = (A.rty0 + qy) * A.tilesXImage + (A.rtx0 + qx);
            if (gt % A.p.nranks != A.p.rank) continue;
        }
        uint32_t tx = gt % A.tilesXImage, ty = gt / A.tilesXImage;
        uint32_t x = tx * tile + pix % tile, y = ty * tile + pix / tile;
        if (x < A.x0 || x > A.x1 || y < A.y0 || y > A.y1) continue;

        uint32_t rng = pixel_seed(x, y, cam.width, A.p.seed);
        Vec3 color = mk3(0.0f, 0.0f, 0.0f);
        if (slot == 0) nRays += samples; // path_tracer.cpp:62

        for (uint32_t pk = 0; pk < samples / 8u; pk++) {
            // ---- primary packet (camera.cpp:35-73, scene.cpp:47-63)
            DevRay pr;
            Vec3 avgDir;
            camera_packet(cam, rng, x, y, slot, gbase, pr, avgDir);
            uint32_t reverseBits = (avgDir.x < 0.0f ? 1u : 0u) | (avgDir.y < 0.0f ? 2u : 0u) | (avgDir.z < 0.0f ? 4u : 0u);
            DevHit h;
            intersect_packet<COUNT>(sc, pr, reverseBits, 100000.0f, h, st, tr, overflow);

            // ---- ComputeRadiance set-up (path_tracer.cpp:81-120): hits gathered into slots 0..alive-1 in lane order
            bool isHit = h.t != -1.0f;
            Surf5 sv{mk3(0, 0, 0), Vec2{0, 0}, 0, 0};
            Vec3 snormal = mk3(0, 0, 0), spos = mk3(0, 0, 0);
            if (isHit) {
                Surface s;
                get_surface<COUNT>(sc, h, s, tr);
                sv = Surf5{s.normal, s.uv, s.mat, s.prim};
                snormal = sample_bump<COUNT>(sc, s.mat, s, tr);
                spos = add3(scale3(h.t, pr.dir), pr.org);
            }
            uint32_t hm = group_ballot(isHit, gbase);
            uint32_t alive = __popc(hm);
            uint32_t src = gbase + ((slot < alive) ? nth_set(hm, slot) : slot);
            Surface props;
            props.normal = sh3(sv.normal, src);
            props.uv = Vec2{shf(sv.uv.x, src), shf(sv.uv.y, src)};
            props.mat = shu(sv.mat, src);
            props.prim = shu(sv.prim, src);
            uint32_t material = props.mat;
            Vec3 normal = sh3(snormal, src);
            Vec3 pos = sh3(spos, src);
            Vec3 rayDir = sh3(pr.dir, src);
            Vec3 beta = mk3(1.0f, 1.0f, 1.0f), result = mk3(0.0f, 0.0f, 0.0f);
            Vec3 lightDir = mk3(0, 0, 0), lightInt = mk3(0, 0, 0);

            uint32_t depth = 0;
            while (alive != 0 && depth < maxDepth) { // path_tracer.cpp:124
                const bool active = slot < alive;
                uint32_t rtype = 2u;
                float4 m0 = make_float4(0, 0, 0, 0), m1 = make_float4(0, 0, 0, 0);
                if (active) {
                    const float4* mp = sc.mats + 3 * (size_t)material;
                    m0 = mp[0];
                    m1 = mp[1];
                    rtype = asu(m0.w);
                    if (m1.x != 0.0f) result = add3(result, mul3(beta, mk3(m1.x, m1.y, m1.z))); // :137-139
                }
                // ---- two draws per diffuse/specular slot, in slot order (:142-144, :175-177)
                const bool draws = active && (rtype == 0u || rtype == 1u);
                uint32_t dm = group_ballot(draws, gbase);
                uint32_t pre = 2u * __popc(dm & lowerMask), tot = 2u * __popc(dm);
                uint32_t s = rng, r2b = 0, r1b = 0;
                for (uint32_t j = 0; j < tot; j++) {
                    s = xorshift32(s);
                    if (j == pre) r2b = s;
                    if (j == pre + 1u) r1b = s;
                }
                rng = s;
                Vec3 nextDir = mk3(0, 0, 0);
                bool wantLight = false;
                if (draws) {
                    Vec3 dd = diffuse_dir(normal, rng_to_float(r2b), rng_to_float(r1b));
                    if (rtype == 0u) {
                        nextDir = dd;
                        beta = mul3(beta, sample_diffuse<COUNT>(sc, material, props.uv, tr)); // :162
                        if (sc.hasLight) { // :168-172
                            lightDir = mk3(sc.lightDir[0], sc.lightDir[1], sc.lightDir[2]);
                            lightInt = mk3(sc.lightIntensity[0], sc.lightIntensity[1], sc.lightIntensity[2]);
                            wantLight = true;
                        }
                    } else {
                        Vec3 reflectDir = sub3(rayDir, scale3(dot3(normal, rayDir), scale3(2.0f, normal))); // :186
                        nextDir = add3(scale3(0.9f, reflectDir), scale3(0.1f, dd));
                    }
                }
                // ---- occlusion rays (:196-252): packet traversal when more than 2 paths are alive
                const bool directLighting = group_ballot(wantLight, gbase) != 0u;
                if (directLighting && active) {
                    const bool grouping = (alive & 0xfu) > 2u;
                    DevRay sr;
                    sr.org = add3(pos, scale3(kFar, lightDir));
                    sr.dir = mk3(-lightDir.x, -lightDir.y, -lightDir.z);
                    nRays++;
                    nOccl++;
                    bool occ;
                    if (grouping) {
                        prepare_soa(sr);
                        occ = occluded<true, COUNT>(sc, sr, kFar - kEpsilon, st, tr, overflow);
                    } else {
                        prepare_single(sr);
                        occ = occluded<false, COUNT>(sc, sr, kFar - kEpsilon, st, tr, overflow);
                    }
                    if (!occ) {
                        Vec3 lr = div3s(scale3(std_max(dot3(lightDir, normal), 0.0f), lightInt), kPi); // :229, :247
                        result = add3(result, mul3(beta, lr));
                    }
                }
                // ---- Russian roulette: one draw per alive slot in slot order when depth > rrDepth (:258-265)
                bool survive = active;
                Vec3 betaNew = beta;
                if (depth > rrDepth) {
                    uint32_t s2 = rng, ub = 0;
                    for (uint32_t j = 0; j < alive; j++) {
                        s2 = xorshift32(s2);
                        if (j == slot) ub = s2;
                    }
                    rng = s2;
                    if (active) {
                        float q = std_max(0.05f, 1.0f - length3(beta));
                        if (rng_to_float(ub) < q) survive = false;
                        else betaNew = div3s(beta, 1.0f - q);
                    }
                }
                // ---- scatter ray (:267-293)
                bool hitNext = false;
                Surf5 nv{mk3(0, 0, 0), Vec2{0, 0}, 0, 0};
                Vec3 npos = mk3(0, 0, 0), ndir = mk3(0, 0, 0);
                Surface ns;
                if (survive) {
                    ndir = normalize3(nextDir);
                    DevRay rr;
                    rr.org = pos;
                    rr.dir = ndir;
                    prepare_single(rr);
                    nRays++;
                    DevHit nh;
                    intersect_single<COUNT>(sc, rr, kFar, nh, st, tr, overflow);
                    if (nh.t != -1.0f) {
                        hitNext = true;
                        get_surface<COUNT>(sc, nh, ns, tr);
                        npos = add3(scale3(nh.t, ndir), pos);
                    }
                }
                // ---- ordered compaction into slot ci (:283-291)
                uint32_t nm = group_ballot(hitNext, gbase);
                uint32_t nAlive = __popc(nm);
                if (nAlive == 0u) break; // :295
                uint32_t ci = __popc(nm & lowerMask);
                uint32_t smat = 0;
                Vec3 snorm = mk3(0, 0, 0);
                if (hitNext) {
                    // materials[ci] = props[i].material reads the slot's PREVIOUS surface unless ci == i (:286-288)
                    smat = (ci == slot) ? ns.mat : props.mat;
                    snorm = sample_bump<COUNT>(sc, smat, ns, tr);
                    nv = Surf5{ns.normal, ns.uv, ns.mat, ns.prim};
                }
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
                if (depth > rrDepth && slot < nAlive) beta = bmoved; // beta[ci] is only written under RR (:263)
                alive = nAlive;
                depth++;
            }
            // ---- Σ result[0..7] in slot order (:303-307), then color += (:71)
            Vec3 res = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
            for (uint32_t l = 0; l < 8; l++) res = add3(res, sh3(result, gbase + l));
            color = add3(color, res);
        }
        color = div3s(color, (float)samples);       // path_tracer.cpp:28
        Vec3 c = scale3(A.p.exposure, color);       // image.cpp:45
        if (slot == 0) {
            float* px = A.rgb + ((size_t)x + (size_t)y * cam.width) * 3;
            px[0] = c.x;
            px[1] = c.y;
            px[2] = c.z;
            nPx++;
        }
    }
    // ---- statistics
    unsigned long long* C = A.counters;
    if (nRays) atomicAdd(&C[0], nRays);
    if (nOccl) atomicAdd(&C[1], nOccl);
    if (COUNT) {
        if (tr.nBox) atomicAdd(&C[2], (unsigned long long)tr.nBox);
        if (tr.nTri) atomicAdd(&C[3], (unsigned long long)tr.nTri);
        if (tr.nHit) atomicAdd(&C[4], (unsigned long long)tr.nHit);
        if (tr.nTap) atomicAdd(&C[5], (unsigned long long)tr.nTap);
    }
    if (nPx) atomicAdd(&C[6], nPx);
    if (overflow) atomicAdd(&C[7], 1ull);
}

// ============================================================================ row-level test kernels
struct RaysArgs {
    DevScene sc;
    int mode;
    uint32_t n;
    const float* org;
    const float* dir;
    float maxT;
    prt_hit* hits;
    uint32_t* spill;
    uint32_t spillStride;
    unsigned long long* counters;
};

__global__ __launch_bounds__(PRT_BLOCK) void rays_kernel(RaysArgs A)
{
    __shared__ uint32_t ldsRef[PRT_STACK_LDS * PRT_BLOCK];
    __shared__ float ldsT[PRT_STACK_LDS * PRT_BLOCK];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, gbase = lane & ~7u;
    const Stack st{&ldsRef[tid], &ldsT[tid], A.spill + ((size_t)blockIdx.x * PRT_BLOCK + tid), A.spillStride};
    uint32_t i = blockIdx.x * PRT_BLOCK + tid;
    bool valid = i < A.n;
    uint32_t ii = valid ? i : 0;
    DevRay r;
    r.org = mk3(A.org[3 * ii], A.org[3 * ii + 1], A.org[3 * ii + 2]);
    r.dir = mk3(A.dir[3 * ii], A.dir[3 * ii + 1], A.dir[3 * ii + 2]);
    Traffic tr{0, 0, 0, 0};
    uint32_t overflow = 0;
    DevHit h{0, 0, 0, 0, 0, 0};
    if (A.mode == 0) {
        prepare_single(r);
        if (valid) intersect_single<false>(A.sc, r, A.maxT, h, st, tr, overflow);
    } else if (A.mode == 1) {
        prepare_soa(r);
        Vec3 avg = mk3(0, 0, 0);
        for (uint32_t l = 0; l < 8; l++) avg = add3(avg, sh3(r.dir, gbase + l));
        avg = div3s(avg, 8.0f);
        uint32_t rev = (avg.x < 0.0f ? 1u : 0u) | (avg.y < 0.0f ? 2u : 0u) | (avg.z < 0.0f ? 4u : 0u);
        if (valid) intersect_packet<false>(A.sc, r, rev, A.maxT, h, st, tr, overflow);
    } else if (A.mode == 2) {
        prepare_single(r);
        if (valid) h.t = occluded<false, false>(A.sc, r, A.maxT, st, tr, overflow) ? 1.0f : 0.0f;
    } else {
        prepare_soa(r);
        if (valid) h.t = occluded<true, false>(A.sc, r, A.maxT, st, tr, overflow) ? 1.0f : 0.0f;
    }
    if (valid) {
        prt_hit o;
        o.t = h.t; o.i = h.i; o.j = h.j; o.k = h.k; o.primId = h.primId; o.meshId = h.meshId;
        A.hits[i] = o;
    }
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

// ============================================================================ host side of the C-ABI
namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? PRT_HIP_ENOMEM : PRT_HIP_ENODEVICE,                 \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                             \
    } while (0)

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

struct prt_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events; // one pair per render launch since the last get_stats
    size_t eventsUsed = 0;
    int computeUnits = 0;
    std::string name;
    // scene
    bool haveScene = false, haveCamera = false;
    DevScene sc{};
    DevCamera cam{};
    std::vector<void*> sceneAllocs;
    // render resources
    float* fb = nullptr;
    size_t fbPixels = 0;
    uint32_t* work = nullptr;
    unsigned long long* counters = nullptr;
    uint32_t* spill = nullptr;
    uint32_t spillThreads = 0;
    int blocksPerCU = 0;
    bool timed = false;
};

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

int prt_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int prt_hip_create(int device, prt_hip_ctx** out)
{
    if (!out) return fail(PRT_HIP_EINVAL, "out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return fail(PRT_HIP_ENODEVICE, "no HIP device: libprt_hip has no CPU path (the GPU kernels are the product)");
    if (device < 0 || device >= n) return fail(PRT_HIP_EINVAL, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    prt_hip_ctx* c = new prt_hip_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    c->computeUnits = prop.multiProcessorCount;
    c->name = prop.name[0] ? prop.name : prop.gcnArchName;
    HIP_TRY(hipStreamCreate(&c->stream));
    HIP_TRY(hipMalloc(&c->work, 256));
    HIP_TRY(hipMalloc(&c->counters, 8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->counters, 0, 8 * sizeof(unsigned long long)));
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
    (void)hipStreamSynchronize(c->stream);
    free_scene(c);
    if (c->fb) (void)hipFree(c->fb);
    if (c->work) (void)hipFree(c->work);
    if (c->counters) (void)hipFree(c->counters);
    if (c->spill) (void)hipFree(c->spill);
    for (auto& e : c->events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
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
        const uint32_t primBase = (uint32_t)(shade.size() / 4), matBase = (uint32_t)(mats.size() / 3);
        sc.primBase[m] = primBase;
        sc.hasNormals[m] = md.normals ? 1u : 0u;
        auto P = [&](uint32_t v) { return HVec3{md.positions[3 * v], md.positions[3 * v + 1], md.positions[3 * v + 2]}; };
        for (uint32_t k = 0; k < md.materialCount; k++) {
            const prt_material& mt = md.materials[k];
            if (mt.diffuseMap >= (int32_t)s->textureCount || mt.bumpMap >= (int32_t)s->textureCount)
                return fail(PRT_HIP_EINVAL, "material texture index out of range");
            if (mt.alphaTest && mt.diffuseMap < 0) return fail(PRT_HIP_EINVAL, "alphaTest material without a diffuse map");
            mats.push_back(make_float4(mt.diffuse[0], mt.diffuse[1], mt.diffuse[2], ubits(mt.reflectionType)));
            mats.push_back(make_float4(mt.emissive[0], mt.emissive[1], mt.emissive[2], ubits(mt.alphaTest)));
            mats.push_back(make_float4(ubits((uint32_t)mt.diffuseMap), ubits((uint32_t)mt.bumpMap), 0.0f, 0.0f));
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
                wnodes.push_back(make_float4(c0.lower[0], c0.lower[1], c0.lower[2], c0.upper[0]));
                wnodes.push_back(make_float4(c0.upper[1], c0.upper[2], c1.lower[0], c1.lower[1]));
                wnodes.push_back(make_float4(c1.lower[2], c1.upper[0], c1.upper[1], c1.upper[2]));
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
        HIP_TRY(hipMalloc(&c->spill, (size_t)threads * 2 * (PRT_STACK_MAX - PRT_STACK_LDS) * sizeof(uint32_t)));
        c->spillThreads = threads;
    }
    return PRT_HIP_OK;
}

static int persistent_blocks(prt_hip_ctx* c)
{
    if (c->blocksPerCU == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, render_kernel<false>, PRT_BLOCK, 0) != hipSuccess || nb <= 0) nb = 2;
        c->blocksPerCU = std::min(nb, 8);
    }
    return c->computeUnits * c->blocksPerCU;
}

int prt_hip_render(prt_hip_ctx* c, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, const prt_render_params* p, float* d_rgb,
                   void* stream)
{
    if (!c || !p) return fail(PRT_HIP_EINVAL, "NULL argument");
    if (!c->haveScene || !c->haveCamera) return fail(PRT_HIP_ESTATE, "upload a scene and set a camera first");
    const uint32_t W = c->cam.width, H = c->cam.height;
    if (x1 < x0 || y1 < y0 || x1 >= W || y1 >= H) return fail(PRT_HIP_EINVAL, "pixel rectangle outside the image");
    if (p->samples == 0 || p->tileSize == 0 || p->nranks == 0 || p->rank >= p->nranks) return fail(PRT_HIP_EINVAL, "bad render params");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
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
    RenderArgs A{};
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
    uint32_t tilesInRect = A.rtnx * A.rtny;
    if (A.fullWidth) {
        // tile ids rty0*TX .. (rty0+rtny)*TX - 1 are contiguous; this rank owns ids == rank (mod nranks)
        uint32_t lo = A.rty0 * A.tilesXImage, hi = lo + tilesInRect;
        uint32_t first = lo + ((p->rank + p->nranks - lo % p->nranks) % p->nranks);
        A.firstOwned = first;
        A.ownedCount = first < hi ? (hi - first + p->nranks - 1) / p->nranks : 0;
        A.totalWork = A.ownedCount * T * T;
    } else {
        A.totalWork = tilesInRect * T * T;
    }
    A.rgb = d_rgb;
    A.work = c->work;
    A.counters = c->counters;
    uint32_t blocks = (uint32_t)persistent_blocks(c);
    uint32_t needed = (A.totalWork * 8 + PRT_BLOCK - 1) / PRT_BLOCK;
    blocks = std::max(1u, std::min(blocks, needed));
    int rc = ensure_launch_resources(c, blocks);
    if (rc) return rc;
    A.spill = c->spill;
    A.spillStride = c->spillThreads;
    HIP_TRY(hipMemsetAsync(c->work, 0, sizeof(uint32_t), s));
    HIP_TRY(hipMemsetAsync(c->counters, 0, 8 * sizeof(unsigned long long), s));
    if (c->eventsUsed == c->events.size()) {
        hipEvent_t a = nullptr, b = nullptr;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        c->events.emplace_back(a, b);
    }
    hipEvent_t ev0 = c->events[c->eventsUsed].first, ev1 = c->events[c->eventsUsed].second;
    c->eventsUsed++;
    HIP_TRY(hipEventRecord(ev0, s));
    if (p->countTraffic) hipLaunchKernelGGL(render_kernel<true>, dim3(blocks), dim3(PRT_BLOCK), 0, s, A);
    else hipLaunchKernelGGL(render_kernel<false>, dim3(blocks), dim3(PRT_BLOCK), 0, s, A);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(PRT_HIP_ELAUNCH, std::string("render_kernel launch: ") + hipGetErrorString(le));
    HIP_TRY(hipEventRecord(ev1, s));
    c->timed = true;
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
    unsigned long long h[8];
    HIP_TRY(hipMemcpy(h, c->counters, sizeof(h), hipMemcpyDeviceToHost));
    st->raysTraced = h[0];
    st->occludedTraced = h[1];
    st->nBox = h[2];
    st->nTri = h[3];
    st->nHit = h[4];
    st->nTap = h[5];
    st->nPx = h[6];
    st->stackOverflow = h[7];
    st->kernelMs = 0.0;
    st->kernelMsSum = 0.0;
    st->kernelLaunches = 0;
    for (size_t i = 0; i < c->eventsUsed; i++) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->events[i].first, c->events[i].second) == hipSuccess) {
            st->kernelMs = ms;
            st->kernelMsSum += ms;
            st->kernelLaunches++;
        }
    }
    c->eventsUsed = 0;
    if (h[7]) return fail(PRT_HIP_ESTACK, "BVH traversal needed more than 64 stack entries (the reference asserts here, bvh.cpp:552)");
    return PRT_HIP_OK;
}

int prt_hip_trace_rays(prt_hip_ctx* c, int mode, uint32_t n, const float* org, const float* dir, float maxT, prt_hit* hits)
{
    if (!c || !org || !dir || !hits) return fail(PRT_HIP_EINVAL, "NULL argument");
    if (!c->haveScene) return fail(PRT_HIP_ESTATE, "upload a scene first");
    if (n == 0 || (n & 7u) || mode < 0 || mode > 3) return fail(PRT_HIP_EINVAL, "n must be a positive multiple of 8, mode 0..3");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t blocks = (n + PRT_BLOCK - 1) / PRT_BLOCK;
    int rc = ensure_launch_resources(c, blocks);
    if (rc) return rc;
    float *dorg = nullptr, *ddir = nullptr;
    prt_hit* dh = nullptr;
    HIP_TRY(hipMalloc(&dorg, (size_t)n * 12));
    HIP_TRY(hipMalloc(&ddir, (size_t)n * 12));
    HIP_TRY(hipMalloc(&dh, (size_t)n * sizeof(prt_hit)));
    HIP_TRY(hipMemcpy(dorg, org, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ddir, dir, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(c->counters, 0, 8 * sizeof(unsigned long long), c->stream));
    RaysArgs A{c->sc, mode, n, dorg, ddir, maxT, dh, c->spill, c->spillThreads, c->counters};
    hipLaunchKernelGGL(rays_kernel, dim3(blocks), dim3(PRT_BLOCK), 0, c->stream, A);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(PRT_HIP_ELAUNCH, std::string("rays_kernel launch: ") + hipGetErrorString(le));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(hits, dh, (size_t)n * sizeof(prt_hit), hipMemcpyDeviceToHost));
    (void)hipFree(dorg);
    (void)hipFree(ddir);
    (void)hipFree(dh);
    c->timed = false;
    return PRT_HIP_OK;
}

int prt_hip_test_leaf(prt_hip_ctx* c, uint32_t n, const float* records, float* out)
{
    if (!c || !records || !out || n == 0) return fail(PRT_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    float *din = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&din, (size_t)n * 22 * 4));
    HIP_TRY(hipMalloc(&dout, (size_t)n * 24 * 4));
    HIP_TRY(hipMemcpy(din, records, (size_t)n * 22 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(leaf_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, din, dout);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, dout, (size_t)n * 24 * 4, hipMemcpyDeviceToHost));
    (void)hipFree(din);
    (void)hipFree(dout);
    return PRT_HIP_OK;
}

int prt_hip_test_sincos(prt_hip_ctx* c, uint32_t n, const float* theta, float* s, float* cs)
{
    if (!c || !theta || !s || !cs || n == 0) return fail(PRT_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    float *dt = nullptr, *ds = nullptr, *dc = nullptr;
    HIP_TRY(hipMalloc(&dt, (size_t)n * 4));
    HIP_TRY(hipMalloc(&ds, (size_t)n * 4));
    HIP_TRY(hipMalloc(&dc, (size_t)n * 4));
    HIP_TRY(hipMemcpy(dt, theta, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sincos_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, dt, ds, dc);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(s, ds, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(cs, dc, (size_t)n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dt);
    (void)hipFree(ds);
    (void)hipFree(dc);
    return PRT_HIP_OK;
}

int prt_hip_test_powf(prt_hip_ctx* c, uint32_t n, const float* x, float* y)
{
    if (!c || !x || !y || n == 0) return fail(PRT_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    float *dx = nullptr, *dy = nullptr;
    HIP_TRY(hipMalloc(&dx, (size_t)n * 4));
    HIP_TRY(hipMalloc(&dy, (size_t)n * 4));
    HIP_TRY(hipMemcpy(dx, x, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(powf_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, dx, dy);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(y, dy, (size_t)n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dx);
    (void)hipFree(dy);
    return PRT_HIP_OK;
}

int prt_hip_test_camera(prt_hip_ctx* c, uint32_t x, uint32_t y, uint32_t state, float* out92)
{
    if (!c || !out92) return fail(PRT_HIP_EINVAL, "bad argument");
    if (!c->haveCamera) return fail(PRT_HIP_ESTATE, "set a camera first");
    HIP_TRY(hipSetDevice(c->device));
    float* d = nullptr;
    HIP_TRY(hipMalloc(&d, 92 * 4));
    hipLaunchKernelGGL(camera_kernel, dim3(1), dim3(64), 0, c->stream, c->cam, x, y, state, d);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out92, d, 92 * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return PRT_HIP_OK;
}

} // extern "C"
