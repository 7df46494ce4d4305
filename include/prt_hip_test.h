/*
 * prt_hip_test.h -- row-level entry points of the TEST build of the library (libprt_hip_test.so = the product's sources
 * compiled with -DPRT_TEST_ENTRY_POINTS).  They exist so that the parity tests can drive single rows of SURVEY.md 8(a)
 * (leaf math, the four traversals, the camera packet, sin/cos/pow) on the device; libprt_hip.so, the product, exports
 * none of them and compiles none of their kernels.
 */
#ifndef PRT_HIP_TEST_H
#define PRT_HIP_TEST_H

#include "prt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the four traversals on caller-supplied rays (host pointers) ----
 * mode 0: Scene::intersect<SingleRayHitPacket,SingleRayPacket>   (scene.cpp:47, bvh.cpp:429 single branch)
 * mode 1: Scene::intersect<RayHitPacket,RayPacket>               (packet branch; rays in groups of 8, avgDir = sum/8)
 * mode 2: Scene::occluded<bool,SingleRayPacket>                  (scene.cpp:69, bvh.cpp:576); hit.t = 1 if occluded else 0
 * mode 3: Scene::occluded<RayPacketMask,RayPacket> with a full mask
 * n rays (multiple of 8); org/dir are n*3 floats. */
int prt_hip_trace_rays(prt_hip_ctx* ctx, int mode, uint32_t n, const float* org, const float* dir, float maxT,
                       prt_hit* hits);
/* leaf math on the device (triangle.cpp:90-166, vecmath.h:1402-1518, ray.h:26-71).  in: 22 floats per record =
 * org[3] dir[3] p0[3] p1[3] p2[3] lower[3] upper[3] maxT; out: 24 floats = [0..3] t,i,j,k with SoaRay::prepare swaps,
 * [4..7] with Ray::prepare swaps, [12] box t, [13] box bool(maxT), [14] box SoA mask(maxT), [16..18] invDir,
 * [19..22] swapXZ/swapYZ (SoA), swapXZ/swapYZ (single) */
int prt_hip_test_leaf(prt_hip_ctx* ctx, uint32_t n, const float* records, float* out);
/* sin/cos of theta[i] as the kernels compute them (prt_devmath.h) */
int prt_hip_test_sincos(prt_hip_ctx* ctx, uint32_t n, const float* theta, float* sin_out, float* cos_out);
/* powf(x, 2.2f) as material.cpp:24-28 (degamma) needs it */
int prt_hip_test_powf(prt_hip_ctx* ctx, uint32_t n, const float* x, float* y);
/* Camera::GenerateJitteredRayPacket + Random on the device: out = 8 x {org[3] dir[3] invDir[3] swapXZ swapYZ},
 * avgDir[3], state after (as float bits) = 92 floats */
int prt_hip_test_camera(prt_hip_ctx* ctx, uint32_t x, uint32_t y, uint32_t state, float* out92);

#ifdef __cplusplus
}
#endif
#endif
