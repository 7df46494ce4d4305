// prt_host_capi.cpp -- C entry points of include/prt_host.h over the classes of prt.h.
#include <stdlib.h>
#include <string.h>

#include "../../../include/prt_host.h"
#include "prt.h"

using namespace prt;

struct prt_host_mesh {
    Mesh mesh;
};

struct prt_host_scene {
    Scene scene;
    std::vector<std::unique_ptr<Bvh>> bvhs;
    prt_scene_desc desc;
    Scene::DescStorage store;
};

static Material toMaterial(const prt_material& m)
{
    Material r;
    r.init();
    r.diffuse = Vector3f(m.diffuse[0], m.diffuse[1], m.diffuse[2]);
    r.emissive = Vector3f(m.emissive[0], m.emissive[1], m.emissive[2]);
    r.reflectionType = (ReflectionType)m.reflectionType;
    r.alphaTest = false;
    return r;
}

extern "C" {

prt_host_mesh* prt_host_mesh_cornell(int box)
{
    auto* m = new prt_host_mesh;
    m->mesh = SampleModels::getCornellBox(box != 0);
    return m;
}

prt_host_mesh* prt_host_mesh_load_obj(const char* path, const prt_material* mat)
{
    auto* m = new prt_host_mesh;
    if (mat) m->mesh.loadObj(path, toMaterial(*mat));
    else m->mesh.loadObj(path);
    if (m->mesh.getPrimCount() == 0) {
        delete m;
        return nullptr;
    }
    return m;
}

prt_host_mesh* prt_host_mesh_from_arrays(uint32_t primCount, uint32_t vertexCount, uint32_t materialCount, const uint32_t* indices,
                                         const float* positions, const float* normals, const float* texcoords,
                                         const uint32_t* primMaterial, const prt_material* materials)
{
    if (!indices || !positions || !primMaterial || !materials) return nullptr;
    auto* m = new prt_host_mesh;
    m->mesh.create(primCount, vertexCount, materialCount, normals != nullptr);
    memcpy(m->mesh.getIndexBuffer(), indices, (size_t)primCount * 12);
    memcpy((void*)m->mesh.getPositionBuffer(), positions, (size_t)vertexCount * 12);
    if (normals) memcpy((void*)m->mesh.getNormalBuffer(), normals, (size_t)vertexCount * 12);
    if (texcoords) {
        memcpy((void*)m->mesh.getTexcoordBuffer(), texcoords, (size_t)vertexCount * 8);
        m->mesh.setHasTexcoord(true);
    }
    memcpy(m->mesh.getPrimMateialBuffer(), primMaterial, (size_t)primCount * 4);
    for (uint32_t i = 0; i < materialCount; i++) m->mesh.getMaterialBuffer()[i] = toMaterial(materials[i]);
    m->mesh.calculateBounds();
    return m;
}

prt_host_mesh* prt_host_mesh_displaced_sphere(uint32_t targetTris, float radius, const float center[3], const prt_material* mat, uint32_t seed)
{
    auto* m = new prt_host_mesh;
    m->mesh = SampleModels::getDisplacedSphere(targetTris, radius, Vector3f(center[0], center[1], center[2]), toMaterial(*mat), seed);
    return m;
}

prt_host_mesh* prt_host_mesh_atrium(uint32_t targetTris, uint32_t seed, int alphaMasked, int bumpMapped, float emissiveFraction)
{
    auto* m = new prt_host_mesh;
    m->mesh = SampleModels::getAtrium(targetTris, seed, alphaMasked != 0, bumpMapped != 0, emissiveFraction);
    return m;
}

void prt_host_mesh_destroy(prt_host_mesh* m) { delete m; }

void prt_host_mesh_transform(prt_host_mesh* m, float s, const float t[3])
{
    auto pos = m->mesh.getPositionBuffer();
    for (uint32_t i = 0; i < m->mesh.getVertexCount(); i++) pos[i] = s * pos[i] + Vector3f(t[0], t[1], t[2]);
}

void prt_host_mesh_calculate_vertex_normals(prt_host_mesh* m) { m->mesh.calculateVertexNormals(); }
void prt_host_mesh_calculate_bounds(prt_host_mesh* m) { m->mesh.calculateBounds(); }
uint32_t prt_host_mesh_prim_count(const prt_host_mesh* m) { return m->mesh.getPrimCount(); }

prt_host_scene* prt_host_scene_create(void)
{
    auto* s = new prt_host_scene;
    s->scene.init();
    return s;
}

void prt_host_scene_destroy(prt_host_scene* s) { delete s; }

int prt_host_scene_add_mesh(prt_host_scene* s, prt_host_mesh* m)
{
    if (!s || !m) return PRT_HIP_EINVAL;
    if (s->bvhs.size() >= PRT_HIP_MAX_BVH) return PRT_HIP_EINVAL;
    std::unique_ptr<Bvh> b(new Bvh);
    b->build(std::move(m->mesh));
    s->scene.add(b.get());
    s->bvhs.push_back(std::move(b));
    delete m;
    return PRT_HIP_OK;
}

void prt_host_scene_set_directional_light(prt_host_scene* s, const float dir[3], const float intensity[3])
{
    s->scene.setDirectionalLight(Vector3f(dir[0], dir[1], dir[2]), Vector3f(intensity[0], intensity[1], intensity[2]));
}

void prt_host_scene_set_env_light(prt_host_scene* s, int32_t width, int32_t height, const float* rgba)
{
    s->scene.setInfiniteAreaLight(width, height, rgba);
}

int prt_host_scene_load_env_light(prt_host_scene* s, const char* path)
{
    s->scene.setInfiniteAreaLight(path);
    return s->scene.isLightAvailable(LightType::kInfiniteArea) ? 0 : -1;
}

const prt_scene_desc* prt_host_scene_describe(prt_host_scene* s)
{
    s->scene.describe(s->desc, s->store);
    return &s->desc;
}

static int saveImage(const char* path, uint32_t w, uint32_t h, const float* rgb, bool tonemap, bool exr, bool zip)
{
    if (!path || !rgb || w == 0 || h == 0) return -1;
    Image img(w, h, tonemap, 1.0f);
    memcpy(img.getPixels(), rgb, (size_t)w * h * 3 * sizeof(float));
    if (exr) img.saveExr(path, zip);
    else img.savePpm(path);
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    fclose(f);
    return 0;
}
int prt_host_save_exr(const char* path, uint32_t w, uint32_t h, const float* rgb, int zip) { return saveImage(path, w, h, rgb, true, true, zip != 0); }
int prt_host_save_ppm(const char* path, uint32_t w, uint32_t h, const float* rgb, int tonemap) { return saveImage(path, w, h, rgb, tonemap != 0, false, false); }

void prt_host_scene_bbox(const prt_host_scene* s, float lu[6])
{
    memcpy(lu, &s->scene.getBBox().lower, 12);
    memcpy(lu + 3, &s->scene.getBBox().upper, 12);
}

void prt_host_camera_create(const float pos[3], const float dir[3], uint32_t width, uint32_t height, prt_camera_desc* out)
{
    Camera c;
    c.create(Vector3f(pos[0], pos[1], pos[2]), Vector3f(dir[0], dir[1], dir[2]), width, height);
    c.describe(*out);
}

int prt_host_bvh_build(uint32_t primCount, const uint32_t* indices, const float* positions, int threads, prt_bvh_node** nodes,
                       uint32_t* nodeCount, uint32_t** primRemapping)
{
    if (!indices || !positions || !nodes || !nodeCount || !primRemapping) return PRT_HIP_EINVAL;
    std::vector<prt_bvh_node> n;
    std::vector<uint32_t> r;
    buildBvhArrays(primCount, indices, reinterpret_cast<const Vector3f*>(positions), n, r, threads);
    *nodes = (prt_bvh_node*)malloc(sizeof(prt_bvh_node) * std::max<size_t>(n.size(), 1));
    *primRemapping = (uint32_t*)malloc(sizeof(uint32_t) * std::max<size_t>(r.size(), 1));
    memcpy(*nodes, n.data(), n.size() * sizeof(prt_bvh_node));
    memcpy(*primRemapping, r.data(), r.size() * 4);
    *nodeCount = (uint32_t)n.size();
    return PRT_HIP_OK;
}

void prt_host_free(void* p) { free(p); }

} // extern "C"
