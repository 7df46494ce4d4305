// prt.h -- host-side C++ surface of the MI355X build.  It keeps the reference's class names,
// method names and argument meaning for everything a PRT-style main.cpp touches
// (Scene/Camera/Mesh/Material/Bvh/Image/PathTracer/SampleModels/ThreadPool), so
// that such a caller compiles against this header unchanged; PathTracer::TraceBlock hands the
// pixel rectangle to the HIP kernels through the C-ABI of include/prt_hip.h.
//
// Written from scratch: the containers are std::vector based and the SIMD wrapper types of the
// reference do not exist here (the GPU kernels replace them).  Citations are file:line under
// /root/reference/src.
#pragma once
#include <stdint.h>

#include <atomic>
#include <cmath>
#include <functional>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/prt_hip.h"

namespace prt
{

// ---- vecmath.h:165-302, 1128-1209 (scalar part; same expressions, same order) ----
struct Vector2f {
    float x, y;
    Vector2f() = default;
    Vector2f(float f) : x(f), y(f) {}
    Vector2f(float _x, float _y) : x(_x), y(_y) {}
    Vector2f operator+(const Vector2f& v) const { return Vector2f(x + v.x, y + v.y); }
    Vector2f operator-(const Vector2f& v) const { return Vector2f(x - v.x, y - v.y); }
    Vector2f operator*(const Vector2f& v) const { return Vector2f(x * v.x, y * v.y); }
    friend Vector2f operator*(float f, const Vector2f& v) { return Vector2f(f * v.x, f * v.y); }
};

struct Vector3f {
    union {
        struct { float x, y, z; };
        float v[3];
    };
    Vector3f() = default;
    Vector3f(float f) : x(f), y(f), z(f) {}
    Vector3f(float _x, float _y, float _z) : x(_x), y(_y), z(_z) {}
    void set(float _x, float _y, float _z) { x = _x; y = _y; z = _z; }
    Vector3f operator-() const { return Vector3f(-x, -y, -z); }
    Vector3f operator-(const Vector3f& o) const { return Vector3f(x - o.x, y - o.y, z - o.z); }
    Vector3f operator+(const Vector3f& o) const { return Vector3f(x + o.x, y + o.y, z + o.z); }
    Vector3f operator*(const Vector3f& o) const { return Vector3f(x * o.x, y * o.y, z * o.z); }
    Vector3f operator/(const Vector3f& o) const { return Vector3f(x / o.x, y / o.y, z / o.z); }
    Vector3f operator*(float f) const { return Vector3f(f * x, f * y, f * z); }
    friend Vector3f operator*(float f, const Vector3f& o) { return Vector3f(f * o.x, f * o.y, f * o.z); }
    friend Vector3f operator/(float f, const Vector3f& o) { return Vector3f(f / o.x, f / o.y, f / o.z); }
};

inline float dot(const Vector3f& a, const Vector3f& b) { auto v = a * b; return v.x + v.y + v.z; }
inline Vector3f cross(const Vector3f& a, const Vector3f& b)
{
    return Vector3f(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float length(const Vector3f& v) { return sqrtf(dot(v, v)); }
inline Vector3f normalize(const Vector3f& v) { float invlen = 1.0f / length(v); return invlen * v; }

// vecmath.h:1088-1113, vecmath.cpp:46-79
struct BBox {
    Vector3f lower, upper;
    static BBox init();
    void merge(const BBox& b);
    void merge(const Vector3f& p);
    float surfaceArea() const;
    Vector3f center() const { return 0.5f * (upper + lower); }
};

// ---- texture.h:15-24: 8-bit texels only on this path ----
struct Texture {
    uint16_t width = 0, height = 0;
    uint8_t component = 0;
    std::shared_ptr<std::vector<uint8_t>> texels;
    void init() { width = height = 0; component = 0; texels.reset(); }
    bool isValid() const { return texels && !texels->empty(); }
    void create(uint16_t w, uint16_t h, uint8_t comp, const uint8_t* data);
    bool isAlphaTestRequired() const; // texture.cpp:338-350
};

// ---- material.h:24-57 ----
enum class ReflectionType : uint32_t { kDiffuse = 0, kSpecular, kRefraction };

struct Material {
    Vector3f diffuse, ambient, specular, emissive;
    Texture diffuseMap, ambientMap, specularMap, emissiveMap, bumpMap;
    ReflectionType reflectionType;
    bool alphaTest;
    void init();
};

// ---- mesh.h:30-105 ----
class Mesh
{
public:
    static const uint32_t kVertexCountPerPrim = 3;
    Mesh() = default;
    Mesh(Mesh&&) = default;
    Mesh& operator=(Mesh&&) = default;
    Mesh(const Mesh&) = delete;
    Mesh& operator=(const Mesh&) = delete;

    void loadObj(const char* path);                      // mesh.cpp:211-300 (own OBJ/MTL reader; textures: see loadTexturePPM)
    void loadObj(const char* path, const Material& mat); // mesh.cpp:151-209
    void create(uint32_t primCount, uint32_t vertexCount, uint32_t materialCount, bool hasVertexNormal); // mesh.cpp:90-105
    void calculateVertexNormals(); // mesh.cpp:108-149
    void calculateBounds();        // mesh.cpp:302-309
    const BBox& getBBox() const { return m_bbox; }

    uint32_t getPrimCount() const { return (uint32_t)m_indices.size() / kVertexCountPerPrim; }
    uint32_t getIndexCount() const { return (uint32_t)m_indices.size(); }
    uint32_t getIndex(uint32_t i) const { return m_indices[i]; }
    uint32_t getVertexCount() const { return (uint32_t)m_positions.size(); }
    const Vector3f& getPosition(uint32_t i) const { return m_positions[i]; }
    const Vector2f& getTexcoord(uint32_t i) const { return m_texcoords[i]; }
    const Vector3f& getNormal(uint32_t i) const { return m_normals[i]; }
    uint32_t getPrimToMaterial(uint32_t i) const { return m_primMaterial[i]; }
    const Material& getMaterial(uint32_t i) const { return m_materials[i]; }
    uint32_t getMaterialCount() const { return (uint32_t)m_materials.size(); }
    bool hasVertexNormal() const { return m_hasVertexNormal; }
    bool hasTexcoord() const { return m_hasTexcoord; }
    void setHasTexcoord(bool b) { m_hasTexcoord = b; }

    uint32_t* getIndexBuffer() { return m_indices.data(); }
    Vector3f* getPositionBuffer() { return m_positions.data(); }
    Vector3f* getNormalBuffer() { return m_normals.data(); }
    uint32_t* getPrimMateialBuffer() { return m_primMaterial.data(); }
    Material* getMaterialBuffer() { return m_materials.data(); }
    Vector2f* getTexcoordBuffer() { return m_texcoords.data(); }
    int32_t getId() const { return (int32_t)m_id; }

private:
    friend class Scene;
    friend class Bvh;
    std::vector<uint32_t> m_indices;
    std::vector<Vector3f> m_positions, m_normals;
    std::vector<Vector2f> m_texcoords;
    std::vector<uint32_t> m_primMaterial;
    std::vector<Material> m_materials;
    BBox m_bbox;
    bool m_hasVertexNormal = false, m_hasTexcoord = false;
    uint32_t m_id = 0;
};

// ---- bvh.h:90-120: host-side build (binned SAH, bvh.cpp:21-299); traversal lives in the kernels ----
class Bvh
{
public:
    Bvh() = default;
    void build(Mesh&& mesh);
    // The same tree, built on the GPU (prt_hip_build_bvh: level-parallel binned SAH, identical node array, leaf order and
    // primRemapping).  `device` < 0 or no HIP device: falls back to build() -- scene set-up must work on any host.
    void buildOnDevice(Mesh&& mesh, int device = 0);
    const std::vector<prt_bvh_node>& getNodes() const { return m_nodes; }
    const std::vector<uint32_t>& getPrimRemapping() const { return m_primRemapping; }
    const Mesh& getMesh() const { return m_mesh; }

private:
    friend class Scene;
    Mesh m_mesh;
    std::vector<uint32_t> m_primRemapping;
    std::vector<prt_bvh_node> m_nodes;
};

// free-standing build used by Bvh::build and by the C helper API
void buildBvhArrays(uint32_t primCount, const uint32_t* indices, const Vector3f* positions, std::vector<prt_bvh_node>& nodes,
                    std::vector<uint32_t>& remap, int threads = 0);

// ---- light.h:10-25 ----
enum class LightType : uint32_t { kDirectional, kInfiniteArea };
struct DirectionalLight {
    void init() { dir = 0.0f; intensity = 0.0f; }
    Vector3f dir, intensity;
};

// ---- light.h:28-50: the image and the CDF tables; sampling (light.cpp:86-128) runs in the kernels ----
class InfiniteAreaLight
{
public:
    void init();
    void create(const char* path); // light.cpp:32: an .exr (own reader: scan lines, NO / RLE / ZIPS / ZIP), or a PFM
    void create(int32_t width, int32_t height, const float* rgba); // light.cpp:34-84 on texels already in memory
    void release();
    bool isValid() const { return !m_texels.empty(); }
    int32_t getWidth() const { return m_width; }
    int32_t getHeight() const { return m_height; }
    const std::vector<float>& getTexels() const { return m_texels; }
    const std::vector<float>& getVerticalP() const { return m_verticalP; }
    const std::vector<float>& getHorizontalP() const { return m_horizontalP; }

private:
    std::vector<float> m_texels; // RGBA float, row 0 first (Texture::loadExr, texture.cpp:303-310)
    std::vector<float> m_verticalP, m_horizontalP;
    int32_t m_width = 0, m_height = 0;
};

// ---- scene.h:11-73 ----
class Scene
{
public:
    void init();
    void add(Bvh* bvh);
    bool isLightAvailable(LightType type) const { return (m_availableLights & (1u << (uint32_t)type)) != 0; }
    void setDirectionalLight(const Vector3f& dir, const Vector3f& intensity);
    const DirectionalLight& getDirectionalLight() const { return m_directionalLight; }
    void setInfiniteAreaLight(const char* path);                                   // scene.h:42-45
    void setInfiniteAreaLight(int32_t width, int32_t height, const float* rgba);   // same, texels already in memory
    const InfiniteAreaLight& getInfiniteAreaLight() const { return m_infiniteAreaLight; }
    float getRadius() const { return m_radius; }
    const BBox& getBBox() const { return m_bbox; }
    const std::vector<Bvh*>& getBvhs() const { return m_bvh; }
    // Identity of the scene's CONTENTS for the device-side cache: drawn from one process-wide counter at construction, init()
    // and every mutation, so (address, revision) never repeats -- a new Scene in a reused stack slot is a different scene.
    uint64_t getRevision() const { return m_revision; }
    Scene();
    ~Scene(); // drops any device-side copy keyed on this object

    // Flattens the scene into the C-ABI descriptor; `store` keeps the arrays the descriptor points to alive.
    struct DescStorage {
        std::vector<prt_mesh_desc> meshes;
        std::vector<std::vector<prt_material>> materials;
        std::vector<prt_texture_desc> textures;
        std::vector<std::shared_ptr<std::vector<uint8_t>>> texelRefs;
    };
    void describe(prt_scene_desc& desc, DescStorage& store) const;

private:
    std::vector<Bvh*> m_bvh;
    uint32_t m_availableLights = 0;
    DirectionalLight m_directionalLight;
    InfiniteAreaLight m_infiniteAreaLight;
    BBox m_bbox;
    float m_radius = 0.0f;
    uint64_t m_revision;
    static uint64_t nextRevision();
};

// ---- camera.h:14-53 ----
class Camera
{
public:
    Camera() = default;
    void create(const Vector3f& pos, const Vector3f& dir, uint32_t width, uint32_t height);
    const Vector3f& getPosition() const { return m_pos; }
    const Vector3f& getDirection() const { return m_dir; }
    void describe(prt_camera_desc& d) const;

private:
    Vector3f m_pos, m_dir, m_up, m_right;
    uint32_t m_width = 0, m_height = 0;
    float m_invWidth = 0, m_invHeight = 0;
};

// ---- image.h:8-30 ----
class Image
{
public:
    Image(uint32_t width, uint32_t height, bool tonemap = true, float exposure = 1.0f);
    void writePixel(uint32_t x, uint32_t y, const Vector3f& color); // image.cpp:44-50
    void savePpm(const char* path) const;                            // image.cpp:52-80
    void saveExr(const char* path, bool zip = true) const;           // image.cpp:82-139: half-float B,G,R OpenEXR, ZIP blocks like tinyexr's default
    void savePfm(const char* path) const;
    uint32_t getWidth() const { return m_width; }
    uint32_t getHeigit() const { return m_height; }
    float getExposure() const { return m_exposure; }
    float* getPixels() { return m_pixels.data(); }
    const float* getPixels() const { return m_pixels.data(); }

private:
    std::vector<float> m_pixels;
    uint32_t m_width, m_height;
    bool m_tonemap;
    float m_exposure;
};

// ---- gbuffer_visualizer.h:15-35: the debug integrator (surface colour / normal per pixel), on the same kernels ----
class GbufferVisualizer
{
public:
    enum class Type { kDiffuse, kMeshNormal, kNormal };
    GbufferVisualizer(Type type = Type::kDiffuse, uint32_t seed = 12345, int device = 0) : m_type(type), m_seed(seed), m_device(device) {}
    void TraceBlock(Image& image, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, const Scene& scene, const Camera& camera);

private:
    Type m_type;
    uint32_t m_seed; // per-pixel generator states, replaces m_rand (gbuffer_visualizer.h:33)
    int m_device;
};

// ---- stats.h:10-33 ----
struct Stats {
    uint64_t nodesTraversed, primsTraversed, raysTraced, occludedTraced, triTested;
    void clear() { nodesTraversed = primsTraversed = raysTraced = occludedTraced = triTested = 0; }
    void merge(const Stats& o)
    {
        nodesTraversed += o.nodesTraversed; primsTraversed += o.primsTraversed; raysTraced += o.raysTraced;
        occludedTraced += o.occludedTraced; triTested += o.triTested;
    }
};

// stats.h:35-68: what main.cpp merges the per-tile tracers into (main.cpp:127-128, 148-150)
struct TotalStats {
    std::atomic<uint64_t> totalNodesTraversed, totalPrimsTraversed, totalRaysTraced, totalOccludedTraced, totalTriTested;
    void clear() { totalNodesTraversed = totalPrimsTraversed = totalRaysTraced = totalOccludedTraced = totalTriTested = 0; }
    void merge(const Stats& o)
    {
        totalNodesTraversed.fetch_add(o.nodesTraversed); totalPrimsTraversed.fetch_add(o.primsTraversed);
        totalRaysTraced.fetch_add(o.raysTraced); totalOccludedTraced.fetch_add(o.occludedTraced); totalTriTested.fetch_add(o.triTested);
    }
    void print();
};

// ---- path_tracer.h:15-38.  TraceBlock renders the INCLUSIVE rectangle on the GPU (C-ABI prt_hip_render) and
// stores the pixels through Image::writePixel's layout.  There is no CPU path: without a HIP device the call
// reports the error and aborts, like the reference's PRT_ASSERT. ----
class PathTracer
{
public:
    struct Options {
        uint32_t maxDepth = 14; // path_tracer.cpp:124
        uint32_t rrDepth = 4;   // path_tracer.cpp:258
        uint32_t seed = 12345;  // per-pixel generator state, replaces random.h:15-17
        int device = 0;
    };
    PathTracer() { m_stats.clear(); }
    explicit PathTracer(const Options& o) : m_options(o) { m_stats.clear(); }
    void TraceBlock(Image& image, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, const Scene& scene, const Camera& camera,
                    uint32_t samples);
    Stats getStats() const { return m_stats; }
    double getKernelMs() const { return m_kernelMs; }
    static void releaseDevice(); // frees the per-device context cache

private:
    Options m_options;
    Stats m_stats;
    double m_kernelMs = 0.0;
};

// ---- sample_models.h ----
class SampleModels
{
public:
    static Mesh getCornellBox(bool box = true); // sample_models.cpp:11-207
    // Seeded procedural stand-ins for the assets the reference's setups load but does not ship
    static Mesh getDisplacedSphere(uint32_t targetTris, float radius, const Vector3f& center, const Material& mat, uint32_t seed);
    static Mesh getAtrium(uint32_t targetTris, uint32_t seed, bool alphaMasked, bool bumpMapped, float emissiveFraction);
};

// ---- thread_pool.h:11-31 (main.cpp queues one task per tile) ----
class ThreadPool
{
public:
    typedef std::function<void()> Task;
    ThreadPool();
    ~ThreadPool();
    void create(int32_t threadCount);
    void queue(Task task);
    void waitAllTasksDone();
    size_t getTaskCount() const;

private:
    struct Impl;
    Impl* m_impl;
};

enum class LogLevel { kInfo, kVerbose, kError };
void logPrintf(LogLevel level, const char* format...);

} // namespace prt
