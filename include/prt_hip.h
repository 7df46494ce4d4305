/*
 * prt_hip.h -- C-ABI of libprt_hip.so: the MI355X (gfx950) implementation of PRT's per-pixel
 * path-tracing loop.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * The reference (amada/PRT) has no FFI of its own: its hot path is entered through one C++
 * call, PathTracer::TraceBlock(Image&, x0,y0,x1,y1, const Scene&, const Camera&, samples)
 * (path_tracer.h:20, called from main.cpp:146-147), on data owned by Scene/Bvh/Mesh.  The entry
 * points below are what that call binds to when the loop runs on the GPU; each cites the
 * reference interface it replaces (file:line under /root/reference/src).  INTEGRATION.md shows the
 * host-side binding.
 *
 * Conventions: every function returns 0 on success or a negative PRT_HIP_E* code;
 * prt_hip_last_error() gives the message (thread local).  The caller owns all host pointers; the
 * library copies what it needs during the call.  One context per device; calls on one context are
 * serialised by the caller; contexts on different devices are independent.
 */
#ifndef PRT_HIP_H
#define PRT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRT_HIP_OK 0
#define PRT_HIP_ENODEVICE (-1) /* no HIP device / HIP runtime failure: the product has no CPU path */
#define PRT_HIP_EINVAL (-2)
#define PRT_HIP_ENOMEM (-3)
#define PRT_HIP_ELAUNCH (-4)
#define PRT_HIP_ESTATE (-5)    /* scene or camera not uploaded */
#define PRT_HIP_ESTACK (-6)    /* traversal stack deeper than 64 entries (reference asserts, bvh.cpp:552) */
#define PRT_HIP_ECOMM (-7)     /* RCCL: library not found or a communicator call failed */

#define PRT_HIP_MAX_BVH 8

typedef struct prt_hip_ctx prt_hip_ctx;

/* Material fields the path reads (material.h:30-44).  reflectionType: 0 diffuse, 1 specular,
 * 2 refraction (material.h:24-28).  diffuseMap/bumpMap index prt_scene_desc.textures, -1 = none. */
typedef struct {
    float diffuse[3];
    float emissive[3];
    uint32_t reflectionType;
    uint32_t alphaTest;
    int32_t diffuseMap;
    int32_t bumpMap;
} prt_material;

/* LinearBvhNode (bvh.h:49-60) with the bit-fields widened. */
typedef struct {
    float lower[3];
    float upper[3];
    uint32_t primOrSecondNodeIndex;
    uint32_t triVectorIndex;
    uint32_t primCount; /* 0xf = internal */
    uint32_t splitAxis;
} prt_bvh_node;

/* One Bvh and the Mesh it owns (bvh.h:113-119, mesh.h:87-104), as built by Bvh::build. */
typedef struct {
    uint32_t nodeCount;
    const prt_bvh_node* nodes;     /* Bvh::m_nodes, DFS order, first child = i+1 */
    uint32_t primCount;
    const uint32_t* primRemapping; /* Bvh::m_primRemapping */
    uint32_t vertexCount;
    const uint32_t* indices;       /* 3*primCount */
    const float* positions;        /* 3*vertexCount */
    const float* normals;          /* 3*vertexCount or NULL (Mesh::hasVertexNormal) */
    const float* texcoords;        /* 2*vertexCount or NULL (Mesh::m_hasTexcoord) */
    uint32_t materialCount;
    const uint32_t* primMaterial;  /* primCount */
    const prt_material* materials;
} prt_mesh_desc;

/* Texture (texture.h:15-24), 8-bit unorm texels, `component` bytes per texel. */
typedef struct {
    int32_t width, height, component;
    const uint8_t* texels;
} prt_texture_desc;

/* What Scene holds for the path (scene.h:61-71): BVHs in Scene::add order, lights, radius. */
typedef struct {
    uint32_t meshCount;
    const prt_mesh_desc* meshes;
    uint32_t textureCount;
    const prt_texture_desc* textures;
    uint32_t hasDirectionalLight; /* Scene::isLightAvailable(kDirectional) */
    float lightDir[3];
    float lightIntensity[3];
    float radius;                 /* Scene::getRadius() */
    /* InfiniteAreaLight (light.h:28-50), tested before the directional light (path_tracer.cpp:164-173): the float RGBA
     * image (Texture::loadExr: 4 floats per texel, row 0 first) and the two CDF tables InfiniteAreaLight::create builds
     * (light.cpp:30-84).  The library copies all three. */
    uint32_t hasInfiniteAreaLight; /* Scene::isLightAvailable(kInfiniteArea) */
    int32_t envWidth, envHeight;
    const float* envTexels;        /* 4*envWidth*envHeight */
    const float* envVerticalP;     /* m_verticalP[envHeight] */
    const float* envHorizontalP;   /* m_horizontalP[envWidth*envHeight] */
} prt_scene_desc;

/* Camera after Camera::create (camera.h:17-36, 46-53). */
typedef struct {
    float pos[3], dir[3], up[3], right[3];
    uint32_t width, height;
    float invWidth, invHeight;
} prt_camera_desc;

/* The literals of the reference's loop, as parameters (SURVEY.md 5 "Config / flags"). */
typedef struct {
    uint32_t samples;  /* kSamples, main.cpp:125; a multiple of 8 (path_tracer.cpp:65) */
    uint32_t maxDepth; /* 14, path_tracer.cpp:124 */
    uint32_t rrDepth;  /* Russian roulette when depth > rrDepth; 4, path_tracer.cpp:258 */
    uint32_t seed;     /* per-pixel generator state = lowbias32(x + y*W + seed) | 1 (replaces random.h:15-17) */
    float exposure;    /* Image::m_exposure, image.cpp:45 */
    uint32_t tileSize; /* 16, main.cpp:123-124; tile t is rendered when t % nranks == rank */
    uint32_t rank, nranks;
    uint32_t countTraffic; /* also count box/triangle/surface/tap events (slower; not for timing) */
} prt_render_params;

/* stats.h:10-16 + the algorithmic-traffic events of DESIGN.md */
typedef struct {
    uint64_t raysTraced;     /* path_tracer.cpp:62,219,242,276 */
    uint64_t occludedTraced; /* path_tracer.cpp:220,243 */
    uint64_t nBox, nTri, nHit, nTap, nPx;
    /* the traversal events again, per traversal (0 primary packets, 1 scatter rays, 2 packet / 3 single occlusion rays); the taps
     * here are the alpha tests inside leaves -- nTap minus their sum are the shading taps (countTraffic launches only) */
    uint64_t modeBox[4], modeTri[4], modeTap[4];
    uint64_t stackOverflow;  /* lanes that needed more than 64 stack entries (must be 0) */
    double kernelMs;         /* HIP-event time of the last render's kernel (events on the launch stream) */
    double kernelMsSum;      /* sum over the render launches since the previous prt_hip_get_stats */
    uint64_t kernelLaunches; /* number of those launches; the event counters above are of the LAST launch */
} prt_hip_stats;

/* RayHitT (ray.h:182-198) */
typedef struct {
    float t, i, j, k;
    uint32_t primId, meshId;
} prt_hit;

/* ---- context ---- */
int prt_hip_device_count(void);
int prt_hip_create(int device, prt_hip_ctx** out);
void prt_hip_destroy(prt_hip_ctx* ctx);
const char* prt_hip_last_error(void);
/* First 16 hex digits of the SHA-256 over the kernel sources THIS library was built from (stamped at build time): a host that
 * quotes measurements (bench.py, the counter summaries under profiles/) compares what is loaded, not what lies in the tree. */
const char* prt_hip_source_sha16(void);
/* fills name (<= cap bytes) and the CU count of the context's device */
int prt_hip_device_info(prt_hip_ctx* ctx, char* name, size_t cap, int* computeUnits);

/* ---- data: replaces the pointers PathTracer reaches through const Scene& / const Camera&
 *      (scene.h:61-71 -> bvh.h:113-119 -> mesh.h:87-104; camera.h:46-53) ---- */
int prt_hip_upload_scene(prt_hip_ctx* ctx, const prt_scene_desc* scene);
int prt_hip_set_camera(prt_hip_ctx* ctx, const prt_camera_desc* camera);

/* ---- Bvh::build on the GPU: the reference's binned-SAH build (bvh.cpp:31-171) and depth-first linearisation
 * (bvh.cpp:230-299) level by level on the device, producing the IDENTICAL node array (prt_bvh_node = LinearBvhNode),
 * leaf order and primRemapping that Bvh::build produces on the host -- a mesh descriptor built from them is interchangeable
 * with one from the host builder.  Host pointers: indices 3 * primCount, positions 3 * vertexCount; nodes_out has room for
 * 2 * primCount nodes, primRemapping_out for primCount entries; buildMs (may be NULL) = device time of the build. ---- */
int prt_hip_build_bvh(prt_hip_ctx* ctx, uint32_t primCount, const uint32_t* indices, uint32_t vertexCount, const float* positions,
                      prt_bvh_node* nodes_out, uint32_t* nodeCount_out, uint32_t* primRemapping_out, double* buildMs);

/* ---- the hot path: replaces PathTracer::TraceBlock (path_tracer.cpp:17-33; pixel rectangle
 * INCLUSIVE as there) + Image::writePixel (image.cpp:44-50).  d_rgb is a DEVICE pointer to
 * width*height*3 floats (pixel (x,y) at (x + y*width)*3), or NULL for the context's own
 * framebuffer.  stream is a hipStream_t (NULL = the context's stream): the render -- ONE persistent
 * kernel launch for the whole rectangle, whatever its size -- is ordered after the work already queued
 * on it and before what is queued on it next.  Asynchronous with respect to the host;
 * prt_hip_download / prt_hip_get_stats synchronise. ---- */
int prt_hip_render(prt_hip_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                   const prt_render_params* params, float* d_rgb, void* stream);
/* GbufferVisualizer::TraceBlock (gbuffer_visualizer.cpp:17-51): one jittered camera ray per pixel, the surface's diffuse colour
 * (type 0 = kDiffuse) or bump-mapped normal * 0.5 + 0.5 (1 = kMeshNormal, 2 = kNormal), times exposure.  Same rectangle,
 * framebuffer and stream conventions as prt_hip_render; the pixel's generator state is the same function of (x, y, seed). */
int prt_hip_render_gbuffer(prt_hip_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t type, uint32_t seed,
                           float exposure, float* d_rgb, void* stream);
/* copies the rectangle (inclusive) of the context's framebuffer into a host image of the camera's size */
int prt_hip_download(prt_hip_ctx* ctx, float* rgb_host, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1);
float* prt_hip_framebuffer(prt_hip_ctx* ctx); /* device pointer, width*height*3 floats */
/* ---- image gather: the ONE exchange of the multi-GPU path.  The reference has none (one process, one
 * Image, main.cpp:107-190); tiles are dealt to ranks by tile id % nranks (prt_render_params), every rank renders the
 * tiles it owns into its own camera-sized framebuffer and the owned tiles are moved to the root: each rank packs them
 * tile-major (1/nranks of the image: 3.1 MB at 1080p with 8 ranks), the packed tiles travel, the root de-interleaves them
 * into its frame.  No reduction, no full-frame traffic, nothing to zero between frames. ---- */
#define PRT_HIP_COMM_ID_BYTES 128
/* One process per GPU (RCCL over xGMI; librccl is loaded on first use).  Rank 0 makes an id and ships its 128 bytes to the
 * other ranks over any host channel; then EVERY rank calls prt_hip_comm_init (collective).  A host that already owns an
 * ncclComm_t for these ranks hands it over with prt_hip_comm_adopt instead (the library never destroys an adopted one). */
int prt_hip_comm_unique_id(void* id128);
int prt_hip_comm_init(prt_hip_ctx* ctx, const void* id128, int rank, int nranks);
int prt_hip_comm_adopt(prt_hip_ctx* ctx, void* ncclComm);
int prt_hip_comm_destroy(prt_hip_ctx* ctx);
/* Collective, after prt_hip_render(..., params.rank = the communicator's rank, params.nranks = its size, d_rgb, stream) on every
 * rank: grouped ncclSend (owners) / ncclRecv (root), so that each link into the root carries one peer's tiles, then the
 * de-interleave kernel on the root.  d_rgb / stream as in prt_hip_render (the same buffer the render wrote); afterwards the
 * root's buffer holds the whole image, the other ranks' buffers are unchanged.
 * A failing ncclSend / ncclRecv returns PRT_HIP_ECOMM with the RCCL group closed again and marks the communicator BROKEN (RCCL leaves
 * it in an error state): every later gather on this context is refused with PRT_HIP_ECOMM until prt_hip_comm_init or
 * prt_hip_comm_adopt replaces it on every rank (a broken communicator the context owns is ended with ncclCommAbort). */
int prt_hip_gather_rccl(prt_hip_ctx* ctx, float* d_rgb, int root, void* stream);
/* bytes the context's rank contributes to a gather of its last render (what travels over xGMI) */
int prt_hip_gather_payload_bytes(prt_hip_ctx* ctx, uint64_t* bytes);
/* One process driving several contexts: context i has rendered with params.rank = i, params.nranks = n into its own
 * framebuffer (d_rgb = NULL).  The same pack and de-interleave kernels with device-to-device copies in between assemble the
 * image in context 0's framebuffer; the rectangle of it is then copied to rgb_host (camera-sized). */
int prt_hip_gather(prt_hip_ctx* const* ctxs, int n, float* rgb_host, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1);
int prt_hip_get_stats(prt_hip_ctx* ctx, prt_hip_stats* stats);

#ifdef __cplusplus
}
#endif
#endif
