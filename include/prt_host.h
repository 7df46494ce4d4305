/*
 * prt_host.h -- C entry points to the host-side scene code of libprt_hip.so (prt_amd/csrc/host):
 * the same Mesh / Bvh / Scene / Camera objects a C++ caller uses through prt.h, reachable from a
 * non-C++ host (the Python package prt_amd binds these with ctypes).  None of this is on the hot
 * path; it produces the prt_scene_desc / prt_camera_desc that prt_hip_upload_scene /
 * prt_hip_set_camera (prt_hip.h) consume.  Reference citations: file:line under /root/reference/src.
 */
#ifndef PRT_HOST_H
#define PRT_HOST_H

#include "prt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct prt_host_mesh prt_host_mesh;   /* a prt::Mesh not yet given to a Bvh */
typedef struct prt_host_scene prt_host_scene; /* a prt::Scene and the Bvhs it holds */

/* SampleModels::getCornellBox (sample_models.cpp:11-207) */
prt_host_mesh* prt_host_mesh_cornell(int box);
/* Mesh::loadObj(path, mat) when mat != NULL (mesh.cpp:151-209), else Mesh::loadObj(path) (mesh.cpp:211-300) */
prt_host_mesh* prt_host_mesh_load_obj(const char* path, const prt_material* mat);
/* Mesh::create + buffer fill (mesh.cpp:90-105, mesh.h:75-80); normals/texcoords may be NULL; untextured materials */
prt_host_mesh* prt_host_mesh_from_arrays(uint32_t primCount, uint32_t vertexCount, uint32_t materialCount,
                                         const uint32_t* indices, const float* positions, const float* normals,
                                         const float* texcoords, const uint32_t* primMaterial,
                                         const prt_material* materials);
/* seeded procedural stand-ins for assets the reference does not ship */
prt_host_mesh* prt_host_mesh_displaced_sphere(uint32_t targetTris, float radius, const float center[3],
                                              const prt_material* mat, uint32_t seed);
prt_host_mesh* prt_host_mesh_atrium(uint32_t targetTris, uint32_t seed, int alphaMasked, int bumpMapped,
                                    float emissiveFraction);
void prt_host_mesh_destroy(prt_host_mesh* m);
/* pos[i] = s*pos[i] + t  (main.cpp:40-44) */
void prt_host_mesh_transform(prt_host_mesh* m, float scale, const float translate[3]);
void prt_host_mesh_calculate_vertex_normals(prt_host_mesh* m); /* mesh.cpp:108-149 */
void prt_host_mesh_calculate_bounds(prt_host_mesh* m);         /* mesh.cpp:302-309 */
uint32_t prt_host_mesh_prim_count(const prt_host_mesh* m);

prt_host_scene* prt_host_scene_create(void); /* Scene::init (scene.cpp:9-16) */
void prt_host_scene_destroy(prt_host_scene* s);
/* Bvh::build(std::move(mesh)) (bvh.cpp:173-228) + Scene::add (scene.cpp:19-27); consumes the mesh */
int prt_host_scene_add_mesh(prt_host_scene* s, prt_host_mesh* m);
void prt_host_scene_set_directional_light(prt_host_scene* s, const float dir[3], const float intensity[3]); /* scene.h:30-35 */
/* Scene::setInfiniteAreaLight (scene.h:42-45) -> InfiniteAreaLight::create (light.cpp:30-84): from float RGBA texels (row 0 first),
 * or from a file: OpenEXR as the reference reads it through tinyexr (scan lines; NO / RLE / ZIPS / ZIP blocks) or an RGB PFM; 0, or -1 when
 * the file cannot be used */
void prt_host_scene_set_env_light(prt_host_scene* s, int32_t width, int32_t height, const float* rgba);
int prt_host_scene_load_env_light(prt_host_scene* s, const char* path);
/* the descriptor of the scene as it stands; valid until the scene is changed or destroyed */
const prt_scene_desc* prt_host_scene_describe(prt_host_scene* s);
void prt_host_scene_bbox(const prt_host_scene* s, float lowerUpper[6]);

/* Image::saveExr (image.cpp:82-139: half-float B,G,R OpenEXR) and Image::savePpm (image.cpp:52-80: tone map, gamma, 8 bit)
 * for a float RGB image of width*height*3 values, row 0 first; 0 or -1 */
int prt_host_save_exr(const char* path, uint32_t width, uint32_t height, const float* rgb, int zip /* 1: ZIP blocks (tinyexr's default), 0: raw */);
int prt_host_save_ppm(const char* path, uint32_t width, uint32_t height, const float* rgb, int tonemap);

/* Camera::create (camera.h:17-36) */
void prt_host_camera_create(const float pos[3], const float dir[3], uint32_t width, uint32_t height,
                            prt_camera_desc* out);

/* BvhBuildNode::build + Bvh::buildLinearBvhNodes (bvh.cpp:31-299) on raw arrays; outputs are malloc'ed, free with
 * prt_host_free */
int prt_host_bvh_build(uint32_t primCount, const uint32_t* indices, const float* positions, int threads,
                       prt_bvh_node** nodes, uint32_t* nodeCount, uint32_t** primRemapping);
void prt_host_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
