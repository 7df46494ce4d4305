// ref_harness.cpp -- drives the COMPILED REFERENCE (objects built by oracle/Makefile from the
// sources where they lie under /root/reference/src) so that its outputs can pin the oracle.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes the reference's
// headers, constructs its objects and calls its functions.  The reference keeps the members this
// harness needs private, so the reference HEADERS are included with `private`/`protected`
// redefined to `public` (layouts are unaffected; the reference's own TUs are compiled untouched).
//
// Two binaries come out of this file (see oracle/Makefile):
//   ref_core  -- links ONLY reference objects (bvh, triangle, vecmath, scene, camera, thread_pool,
//                log).  Commands: leaf, bvh, rays, camera.  The reference's mesh.cpp, material.cpp,
//                texture.cpp and image.cpp cannot be compiled in this image (they include the
//                un-vendored tinyobjloader / stb / tinyexr headers), so the few symbols they would
//                define stay unresolved and are never reached by these commands.
//   ref_path  -- additionally links path_tracer.o, sample_models.o and oracle/ref_glue.cpp, which
//                supplies the missing leaf functions by forwarding to the oracle's restatement.
//                Commands: + render, cornell.  (-DREF_WITH_GLUE)
//
// Scene file format ("PRTS", little endian) -- written by tests/prt_testlib.py:
//   char magic[4]="PRTS"; u32 version=1; u32 meshCount;
//   per mesh: u32 primCount, vertexCount, materialCount, hasNormals, hasTexcoord;
//             u32 indices[3*primCount]; f32 positions[3*vertexCount]; f32 normals[3*vertexCount] if hasNormals;
//             f32 texcoords[2*vertexCount] if hasTexcoord; u32 primMaterial[primCount];
//             materialCount x { f32 diffuse[3]; f32 emissive[3]; u32 reflectionType; u32 alphaTest; i32 diffuseMap; i32 bumpMap; }
//   u32 textureCount; per texture: i32 width, height, component; u8 texels[width*height*component];
//   u32 hasDirectional; f32 lightDir[3]; f32 lightIntensity[3];
//   f32 camPos[3]; f32 camDir[3]; u32 width; u32 height; f32 exposure;
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include <cmath>
#include <limits>
#include <vector>
#include <string>
#include <atomic>
#include <deque>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <thread>
#include <random>
#include <chrono>
#include <type_traits>
#include <immintrin.h>

#define private public
#define protected public
#include "vecmath.h"
#include "random.h"
#include "ray.h"
#include "triangle.h"
#include "mesh.h"
#include "bvh.h"
#include "scene.h"
#include "camera.h"
#include "thread_pool.h"
#ifdef REF_WITH_GLUE
#include "image.h"
#include "path_tracer.h"
#include "gbuffer_visualizer.h"
#include "sample_models.h"
#endif
#undef private
#undef protected

#include "prt_oracle.h"
#ifdef REF_WITH_GLUE
#include "ref_glue.h"
#endif

using namespace prt;

// ------------------------------------------------------------------ scene file
struct FMesh {
    uint32_t primCount, vertexCount, materialCount, hasNormals, hasTexcoord;
    std::vector<uint32_t> indices, primMaterial;
    std::vector<float> positions, normals, texcoords;
    std::vector<orc_material> materials;
};
struct FTexture { int32_t w, h, comp; std::vector<uint8_t> texels; };
struct FScene {
    std::vector<FMesh> meshes;
    std::vector<FTexture> textures;
    uint32_t hasDirectional;
    float lightDir[3], lightIntensity[3], camPos[3], camDir[3];
    uint32_t width, height;
    float exposure;
};

static void rd(FILE* f, void* p, size_t n)
{
    if (n && fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
}

static FScene loadScene(const char* path)
{
    FScene s;
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    char magic[4]; uint32_t version, meshCount;
    rd(f, magic, 4); rd(f, &version, 4); rd(f, &meshCount, 4);
    if (memcmp(magic, "PRTS", 4) || version != 1) { fprintf(stderr, "bad scene file\n"); exit(2); }
    s.meshes.resize(meshCount);
    for (auto& m : s.meshes) {
        rd(f, &m.primCount, 20);
        m.indices.resize(3 * (size_t)m.primCount); rd(f, m.indices.data(), m.indices.size() * 4);
        m.positions.resize(3 * (size_t)m.vertexCount); rd(f, m.positions.data(), m.positions.size() * 4);
        if (m.hasNormals) { m.normals.resize(3 * (size_t)m.vertexCount); rd(f, m.normals.data(), m.normals.size() * 4); }
        if (m.hasTexcoord) { m.texcoords.resize(2 * (size_t)m.vertexCount); rd(f, m.texcoords.data(), m.texcoords.size() * 4); }
        m.primMaterial.resize(m.primCount); rd(f, m.primMaterial.data(), m.primMaterial.size() * 4);
        m.materials.resize(m.materialCount); rd(f, m.materials.data(), m.materials.size() * sizeof(orc_material));
    }
    uint32_t texCount; rd(f, &texCount, 4);
    s.textures.resize(texCount);
    for (auto& t : s.textures) {
        rd(f, &t.w, 12);
        t.texels.resize((size_t)t.w * t.h * t.comp + 16);
        rd(f, t.texels.data(), (size_t)t.w * t.h * t.comp);
    }
    rd(f, &s.hasDirectional, 4); rd(f, s.lightDir, 12); rd(f, s.lightIntensity, 12);
    rd(f, s.camPos, 12); rd(f, s.camDir, 12); rd(f, &s.width, 4); rd(f, &s.height, 4); rd(f, &s.exposure, 4);
    fclose(f);
    return s;
}

// ------------------------------------------------------------------ reference objects from a scene file
struct RefScene {
    Scene* scene;
    Camera camera;
    std::vector<Bvh*> bvhs;
    std::vector<orc_mesh*> omeshes; // the oracle's view of the same meshes (for ref_glue)
    FScene* file;
};

// Fill a reference Mesh's fields directly (its allocating members live in mesh.cpp, unbuildable here).
static void fillMesh(Mesh* m, const FMesh& fm, const FScene& fs)
{
    m->m_indexCount = 3 * fm.primCount;
    m->m_vertexCount = fm.vertexCount;
    m->m_materialCount = fm.materialCount;
    m->m_indices = new uint32_t[m->m_indexCount];
    memcpy(m->m_indices, fm.indices.data(), m->m_indexCount * 4);
    m->m_texcoordIndices = new uint32_t[m->m_indexCount];
    memcpy(m->m_texcoordIndices, fm.indices.data(), m->m_indexCount * 4); // mesh.cpp:179,280: texcoord index == vertex index
    m->m_positions = new Vector3f[fm.vertexCount];
    memcpy((void*)m->m_positions, fm.positions.data(), (size_t)fm.vertexCount * 12);
    m->m_texcoords = new Vector2f[fm.vertexCount];
    memset((void*)m->m_texcoords, 0, (size_t)fm.vertexCount * 8);
    if (fm.hasTexcoord) memcpy((void*)m->m_texcoords, fm.texcoords.data(), (size_t)fm.vertexCount * 8);
    m->m_hasTexcoord = fm.hasTexcoord != 0;
    if (fm.hasNormals) {
        m->m_normals = new Vector3f[fm.vertexCount];
        memcpy((void*)m->m_normals, fm.normals.data(), (size_t)fm.vertexCount * 12);
    }
    m->m_hasVertexNormal = fm.hasNormals != 0;
    m->m_primMaterial = new uint32_t[fm.primCount];
    memcpy(m->m_primMaterial, fm.primMaterial.data(), (size_t)fm.primCount * 4);
    m->m_materials = (Material*)calloc(fm.materialCount, sizeof(Material));
    for (uint32_t i = 0; i < fm.materialCount; i++) {
        Material& d = m->m_materials[i];
        const orc_material& sm = fm.materials[i];
        d.diffuse = Vector3f(sm.diffuse[0], sm.diffuse[1], sm.diffuse[2]);
        d.emissive = Vector3f(sm.emissive[0], sm.emissive[1], sm.emissive[2]);
        d.reflectionType = (ReflectionType)sm.reflectionType;
        d.alphaTest = sm.alphaTest != 0;
        auto setTex = [&](Texture& t, int32_t idx) {
            if (idx < 0) return;
            const FTexture& ft = fs.textures[idx];
            t.width = (uint16_t)ft.w; t.height = (uint16_t)ft.h; t.component = (uint8_t)ft.comp;
            t.channel = TextureChannel::k8Unorm;
            t.texels = (void*)ft.texels.data();
        };
        setTex(d.diffuseMap, sm.diffuseMap);
        setTex(d.bumpMap, sm.bumpMap);
    }
    // Mesh::calculateBounds (mesh.cpp:302) lives in the unbuildable TU; BBox::merge is reference code (vecmath.cpp)
    BBox bbox = BBox::init();
    for (uint32_t i = 0; i < fm.vertexCount; i++) bbox.merge(m->m_positions[i]);
    m->m_bbox = bbox;
}

// Orchestration of Bvh::build (bvh.cpp:173-228) around the reference's own BvhBuildNode::build and
// Bvh::buildLinearBvhNodes; the wrapper itself cannot be called because its first statement is
// Mesh::operator=(Mesh&&) from mesh.cpp.
static Bvh* buildBvh(const Mesh* src, bool threaded)
{
    Bvh* b = (Bvh*)calloc(1, sizeof(Bvh));
    memcpy((void*)&b->m_mesh, (const void*)src, sizeof(Mesh));
    const uint32_t n = b->m_mesh.getPrimCount();
    b->m_primRemapping = new uint32_t[n];
    for (uint32_t i = 0; i < n; i++) b->m_primRemapping[i] = i;
    BvhBuildNode::BuildContext context;
    context.threadPool = nullptr;
    if (threaded) {
        context.threadPool = new ThreadPool;
        context.threadPool->create(-1);
    }
    context.nodeCount = 1;
    for (auto& c : context.primCountInNode) c = 0;
    auto root = new BvhBuildNode;
    context.mesh = &b->m_mesh;
    context.primRemapping = b->m_primRemapping;
    root->build(context, 0, (int32_t)n - 1, 0);
    if (context.threadPool) {
        context.threadPool->waitAllTasksDone();
        delete context.threadPool;
    }
    LinearBvhNode* nodes = new LinearBvhNode[context.nodeCount];
    memset((void*)nodes, 0, sizeof(LinearBvhNode) * context.nodeCount);
    uint32_t primNodeCount = 0;
    for (auto& c : context.primCountInNode) primNodeCount += c;
    b->m_triangleVectors = new TriangleVector[primNodeCount];
    int32_t triVectorIndex = 0, index = 0;
    b->buildLinearBvhNodes(nodes, &index, root, &triVectorIndex);
    b->m_nodes = nodes;
    return b;
}

static RefScene buildRef(FScene& fs, bool threadedBuild = false)
{
    RefScene r;
    r.file = &fs;
    r.scene = new Scene();
    // Scene::init (scene.cpp:9-16) minus InfiniteAreaLight::init (light.cpp -> texture.cpp, unbuildable)
    r.scene->m_directionalLight.init();
    r.scene->m_availableLights = 0;
    r.scene->m_bbox = BBox::init();
    r.scene->m_radius = std::numeric_limits<float>::max();
    for (auto& fm : fs.meshes) {
        Mesh* m = (Mesh*)calloc(1, sizeof(Mesh));
        fillMesh(m, fm, fs);
        Bvh* b = buildBvh(m, threadedBuild);
        r.bvhs.push_back(b);
        r.scene->add(b);
        r.omeshes.push_back(orc_mesh_create(fm.primCount, fm.vertexCount, fm.materialCount, fm.indices.data(),
                                            fm.positions.data(), fm.hasNormals ? fm.normals.data() : nullptr,
                                            fm.hasTexcoord ? fm.texcoords.data() : nullptr, fm.primMaterial.data(),
                                            fm.materials.data()));
    }
    if (fs.hasDirectional)
        r.scene->setDirectionalLight(Vector3f(fs.lightDir[0], fs.lightDir[1], fs.lightDir[2]),
                                     Vector3f(fs.lightIntensity[0], fs.lightIntensity[1], fs.lightIntensity[2]));
    r.camera.create(Vector3f(fs.camPos[0], fs.camPos[1], fs.camPos[2]), Vector3f(fs.camDir[0], fs.camDir[1], fs.camDir[2]),
                    fs.width, fs.height);
    return r;
}

static std::vector<float> readFloats(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<float> v(sz / 4);
    rd(f, v.data(), v.size() * 4);
    fclose(f);
    return v;
}

static void writeAll(const char* path, const void* p, size_t n)
{
    FILE* f = fopen(path, "wb");
    if (!f || (n && fwrite(p, 1, n, f) != n)) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    fclose(f);
}

// ------------------------------------------------------------------ leaf: triangle + box + prepare + rng
// in : N records of 22 floats: org[3] dir[3] p0[3] p1[3] p2[3] lower[3] upper[3] maxT
// out: N records of 24 floats:
//   [0..3]   SoA intersectTriangle with SoaRay::prepare swaps: t i j k      (triangle.cpp:90, ray.h:58)
//   [4..7]   SoA intersectTriangle with Ray::prepare swaps: t i j k         (ray.h:26)
//   [8..11]  scalar intersectTriangle: t i j k                              (triangle.cpp:8)
//   [12]     BBox::intersect scalar -> t        (vecmath.h:1402)
//   [13]     BBox::intersect scalar bool(maxT)  (vecmath.h:1449)
//   [14]     BBox::intersect SoA mask(maxT) lane 0 (vecmath.h:1504)
//   [15]     BBox::intersect SoA -> t lane 0    (vecmath.h:1482)
//   [16..18] SoaRay invDir lane 0; [19] swapXZ(soa) [20] swapYZ(soa) [21] swapXZ(single) [22] swapYZ(single) [23] 0
static int cmdLeaf(const char* in, const char* out)
{
    auto v = readFloats(in);
    size_t n = v.size() / 22;
    std::vector<float> o(n * 24, 0.0f);
    for (size_t r = 0; r < n; r++) {
        const float* p = &v[r * 22];
        float* q = &o[r * 24];
        Vector3f org(p[0], p[1], p[2]), dir(p[3], p[4], p[5]);
        Vector3f p0(p[6], p[7], p[8]), p1(p[9], p[10], p[11]), p2(p[12], p[13], p[14]);
        BBox box;
        box.lower = Vector3f(p[15], p[16], p[17]);
        box.upper = Vector3f(p[18], p[19], p[20]);
        float maxT = p[21];

        SoaRay sray;
        sray.org = SoaVector3f(org);
        sray.dir = SoaVector3f(dir);
        sray.maxT = maxT;
        sray.prepare();
        Ray ray;
        ray.org = org; ray.dir = dir; ray.maxT = maxT;
        ray.prepare();

        SoaMask all; all.setAll(true);
        auto a = intersectTriangle(all, sray.org, sray.dir, sray.swapXZ, sray.swapYZ, SoaVector3f(p0), SoaVector3f(p1), SoaVector3f(p2));
        q[0] = a.t.getLane(0);
        if (q[0] != -1.0f) { q[1] = a.i.getLane(0); q[2] = a.j.getLane(0); q[3] = a.k.getLane(0); }
        auto b = intersectTriangle(all, SoaVector3f(ray.org), SoaVector3f(ray.dir), ray.swapXZ, ray.swapYZ, SoaVector3f(p0), SoaVector3f(p1), SoaVector3f(p2));
        q[4] = b.t.getLane(0);
        if (q[4] != -1.0f) { q[5] = b.i.getLane(0); q[6] = b.j.getLane(0); q[7] = b.k.getLane(0); }
        auto c = intersectTriangle(org, dir, p0, p1, p2);
        q[8] = c.t;
        if (c.t != -1.0f) { q[9] = c.i; q[10] = c.j; q[11] = c.k; }
        q[12] = box.intersect(ray.org, ray.dir, ray.invDir);
        q[13] = box.intersect(ray.org, ray.dir, ray.invDir, maxT) ? 1.0f : 0.0f;
        q[14] = (box.intersect(sray.org, sray.dir, sray.invDir, sray.maxT).ballot() & 1) ? 1.0f : 0.0f;
        q[15] = box.intersect(sray.org, sray.dir, sray.invDir).getLane(0);
        auto inv = sray.invDir.getLane(0);
        q[16] = inv.x; q[17] = inv.y; q[18] = inv.z;
        q[19] = (sray.swapXZ.ballot() & 1) ? 1.0f : 0.0f;
        q[20] = (sray.swapYZ.ballot() & 1) ? 1.0f : 0.0f;
        q[21] = (ray.swapXZ.ballot() & 1) ? 1.0f : 0.0f;
        q[22] = (ray.swapYZ.ballot() & 1) ? 1.0f : 0.0f;
    }
    writeAll(out, o.data(), o.size() * 4);
    return 0;
}

// ------------------------------------------------------------------ bvh dump
// out: u32 meshCount; per mesh: u32 nodeCount, leafCount, primCount; nodeCount x {f32 lower[3], upper[3]; u32 primOrSecond,
//      triVectorIndex, primCount, splitAxis}; u32 primRemapping[primCount]; f32 meshBBox[6]; then f32 sceneBBox[6], f32 radius
static int cmdBvh(const char* scenePath, const char* out, bool threaded)
{
    FScene fs = loadScene(scenePath);
    RefScene r = buildRef(fs, threaded);
    std::vector<uint8_t> buf;
    auto put = [&](const void* p, size_t n) { buf.insert(buf.end(), (const uint8_t*)p, (const uint8_t*)p + n); };
    uint32_t mc = (uint32_t)r.bvhs.size();
    put(&mc, 4);
    for (Bvh* b : r.bvhs) {
        // node count: DFS from the root
        uint32_t nodeCount = 0, leafCount = 0;
        {
            std::vector<uint32_t> st{0};
            while (!st.empty()) {
                uint32_t i = st.back(); st.pop_back();
                nodeCount = std::max(nodeCount, i + 1);
                const LinearBvhNode& n = b->m_nodes[i];
                if (n.primCount == LinearBvhNode::kInternalNode) { st.push_back(i + 1); st.push_back(n.primOrSecondNodeIndex); }
                else leafCount++;
            }
        }
        uint32_t primCount = b->m_mesh.getPrimCount();
        put(&nodeCount, 4); put(&leafCount, 4); put(&primCount, 4);
        for (uint32_t i = 0; i < nodeCount; i++) {
            const LinearBvhNode& n = b->m_nodes[i];
            orc_node o;
            memcpy(o.lower, &n.bbox.lower, 12);
            memcpy(o.upper, &n.bbox.upper, 12);
            o.primOrSecondNodeIndex = n.primOrSecondNodeIndex;
            o.primCount = n.primCount;
            o.splitAxis = n.splitAxis;
            o.triVectorIndex = (n.primCount == LinearBvhNode::kInternalNode) ? 0u : (uint32_t)n.triVectorIndex;
            put(&o, sizeof(o));
        }
        put(b->m_primRemapping, (size_t)primCount * 4);
        put(&b->m_mesh.m_bbox, 24);
    }
    put(&r.scene->m_bbox, 24);
    float radius = r.scene->getRadius();
    put(&radius, 4);
    writeAll(out, buf.data(), buf.size());
    return 0;
}

// ------------------------------------------------------------------ rays
// in : N (multiple of 8) records of 7 floats: org[3] dir[3] maxT
// out: per ray 16 words: single hit {t,i,j,k (f32), primId, meshId (u32)}, occluded_single (u32),
//      packet hit {t,i,j,k,primId,meshId}, occluded_packet (u32), 2 x pad
//   Packets are consecutive groups of 8 rays; avgDir = (sum of dirs, in lane order)/8 as camera.cpp:56,70;
//   packet maxT = maxT of the group's first ray (SoaRay::maxT is per lane, but Scene::intersect uses it per lane too:
//   we set every lane's own maxT).
static int cmdRays(const char* scenePath, const char* in, const char* out)
{
    FScene fs = loadScene(scenePath);
    RefScene r = buildRef(fs);
    auto v = readFloats(in);
    size_t n = v.size() / 7;
    std::vector<uint32_t> o(n * 16, 0);
    auto putf = [](uint32_t* dst, float f) { memcpy(dst, &f, 4); };
    for (size_t g = 0; g + 8 <= n; g += 8) {
        Vector3f orgs[8], dirs[8];
        float maxTs[8];
        Vector3f avg(0.0f);
        for (int l = 0; l < 8; l++) {
            const float* p = &v[(g + l) * 7];
            orgs[l] = Vector3f(p[0], p[1], p[2]);
            dirs[l] = Vector3f(p[3], p[4], p[5]);
            maxTs[l] = p[6];
            avg = avg + dirs[l];
        }
        for (int l = 0; l < 8; l++) {
            uint32_t* q = &o[(g + l) * 16];
            Ray ray;
            ray.org = orgs[l]; ray.dir = dirs[l]; ray.maxT = maxTs[l];
            ray.prepare();
            SingleRayHitPacket hp;
            r.scene->intersect(hp, SingleRayPacket(ray));
            putf(q + 0, hp.hit.t);
            if (hp.hit.isHit()) { putf(q + 1, hp.hit.i); putf(q + 2, hp.hit.j); putf(q + 3, hp.hit.k); q[4] = hp.hit.primId; q[5] = hp.hit.meshId; }
            q[6] = r.scene->occluded<bool, SingleRayPacket>(RayPacketMask(), SingleRayPacket(ray)) ? 1u : 0u;
        }
        RayPacket pk;
        pk.rays[0].org = SoaVector3f(orgs);
        pk.rays[0].dir = SoaVector3f(dirs);
        pk.rays[0].maxT = SoaFloat(_mm256_loadu_ps(maxTs));
        pk.rays[0].prepare();
        pk.avgDir = avg / RayPacket::kSize;
        RayHitPacket hp;
        r.scene->intersect(hp, pk);
        auto om = r.scene->occluded<RayPacketMask, RayPacket>(RayPacketMask(0xff), pk);
        int obits = om.masks[0].ballot();
        for (int l = 0; l < 8; l++) {
            uint32_t* q = &o[(g + l) * 16];
            float t = hp.hits[0].t.getLane(l);
            putf(q + 7, t);
            if (t != -1.0f) {
                putf(q + 8, hp.hits[0].i.getLane(l)); putf(q + 9, hp.hits[0].j.getLane(l)); putf(q + 10, hp.hits[0].k.getLane(l));
                q[11] = (uint32_t)hp.hits[0].primId.getLane(l);
                q[12] = (uint32_t)hp.hits[0].meshId.getLane(l);
            }
            q[13] = (obits >> l) & 1;
        }
    }
    writeAll(out, o.data(), o.size() * 4);
    return 0;
}

// ------------------------------------------------------------------ camera packets + RNG
// args: x y state ; out: 8 x {org[3] dir[3] invDir[3] swapXZ swapYZ}(11 f32) + avgDir[3] + u32 state after + 4 rng floats
static int cmdCamera(const char* scenePath, uint32_t x, uint32_t y, uint32_t state, const char* out)
{
    FScene fs = loadScene(scenePath);
    Camera cam;
    cam.create(Vector3f(fs.camPos[0], fs.camPos[1], fs.camPos[2]), Vector3f(fs.camDir[0], fs.camDir[1], fs.camDir[2]), fs.width, fs.height);
    Random rng;
    rng.m_state.a = state;
    auto pk = cam.GenerateJitteredRayPacket(rng, x, y);
    std::vector<float> o;
    for (int l = 0; l < 8; l++) {
        auto org = pk.rays[0].org.getLane(l), dir = pk.rays[0].dir.getLane(l), inv = pk.rays[0].invDir.getLane(l);
        float rec[11] = {org.x, org.y, org.z, dir.x, dir.y, dir.z, inv.x, inv.y, inv.z,
                         (pk.rays[0].swapXZ.ballot() >> l) & 1 ? 1.0f : 0.0f, (pk.rays[0].swapYZ.ballot() >> l) & 1 ? 1.0f : 0.0f};
        o.insert(o.end(), rec, rec + 11);
    }
    o.push_back(pk.avgDir.x); o.push_back(pk.avgDir.y); o.push_back(pk.avgDir.z);
    float st; uint32_t s = rng.m_state.a; memcpy(&st, &s, 4);
    o.push_back(st);
    for (int i = 0; i < 2; i++) o.push_back(rng.generate());
    for (int i = 0; i < 2; i++) o.push_back(rng.generateMinus1to1());
    float cm[12] = {cam.m_pos.x, cam.m_pos.y, cam.m_pos.z, cam.m_dir.x, cam.m_dir.y, cam.m_dir.z,
                    cam.m_up.x, cam.m_up.y, cam.m_up.z, cam.m_right.x, cam.m_right.y, cam.m_right.z};
    o.insert(o.end(), cm, cm + 12);
    writeAll(out, o.data(), o.size() * 4);
    return 0;
}

#ifdef REF_WITH_GLUE
// ------------------------------------------------------------------ render (PathTracer::TraceBlock per pixel)
// out: f32 rgb[(x1-x0+1)*(y1-y0+1)*3] row-major over the rectangle, then u64 raysTraced, u64 occludedTraced
// (zero unless built with PRT_ENABLE_STATS), then f64 seconds
static int cmdRender(const char* scenePath, uint32_t spp, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t seed,
                     int threads, const char* out, const char* envPath)
{
    FScene fs = loadScene(scenePath);
    RefScene r = buildRef(fs);
    refGlueRegister(r.bvhs, r.omeshes);
    if (envPath) r.scene->setInfiniteAreaLight(envPath); // scene.h:42-45 -> InfiniteAreaLight::create (light.cpp:30-84)
    const uint32_t W = fs.width, H = fs.height;
    Image* image = refGlueMakeImage(W, H, fs.exposure);
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    const uint32_t rw = x1 - x0 + 1, rh = y1 - y0 + 1;
    std::atomic<uint32_t> next(0);
    std::atomic<uint64_t> rays(0), occl(0);
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&]() {
        // 16x16 tiles (main.cpp:123-124), one PathTracer per pixel so that its generator can be given the
        // per-pixel state (m_rand is seeded from libc rand() in the reference, random.h:15-17)
        const uint32_t tx = (rw + 15) / 16, ty = (rh + 15) / 16;
        uint64_t lr = 0, lo = 0;
        for (;;) {
            uint32_t t = next.fetch_add(1);
            if (t >= tx * ty) break;
            uint32_t bx = x0 + (t % tx) * 16, by = y0 + (t / tx) * 16;
            for (uint32_t y = by; y < by + 16 && y <= y1; y++)
                for (uint32_t x = bx; x < bx + 16 && x <= x1; x++) {
                    PathTracer tracer;
                    tracer.m_rand.m_state.a = orc_pixel_seed(x, y, W, seed);
                    tracer.TraceBlock(*image, x, y, x, y, *r.scene, r.camera, spp);
                    auto st = tracer.getStats();
                    lr += st.raysTraced; lo += st.occludedTraced;
                }
        }
        rays += lr; occl += lo;
    };
    std::vector<std::thread> pool;
    for (int i = 0; i < threads; i++) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<float> o((size_t)rw * rh * 3);
    const float* px = refGlueImagePixels(image);
    for (uint32_t y = 0; y < rh; y++)
        memcpy(&o[(size_t)y * rw * 3], &px[((size_t)(y0 + y) * W + x0) * 3], (size_t)rw * 12);
    FILE* f = fopen(out, "wb");
    fwrite(o.data(), 4, o.size(), f);
    uint64_t a = rays, b = occl;
    fwrite(&a, 8, 1, f); fwrite(&b, 8, 1, f); fwrite(&sec, 8, 1, f);
    fclose(f);
    fprintf(stderr, "ref render %ux%u spp=%u threads=%d: %.3f s, rays=%llu occl=%llu\n", rw, rh, spp, threads, sec,
            (unsigned long long)a, (unsigned long long)b);
    return 0;
}

// ------------------------------------------------------------------ GbufferVisualizer::TraceBlock per pixel (gbuffer_visualizer.cpp:17-51)
// out: f32 rgb of the rectangle, row-major
static int cmdGbuffer(const char* scenePath, int type, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t seed, const char* out)
{
    FScene fs = loadScene(scenePath);
    RefScene r = buildRef(fs);
    refGlueRegister(r.bvhs, r.omeshes);
    const uint32_t W = fs.width, H = fs.height;
    Image* image = refGlueMakeImage(W, H, fs.exposure);
    for (uint32_t y = y0; y <= y1; y++)
        for (uint32_t x = x0; x <= x1; x++) {
            GbufferVisualizer vis((GbufferVisualizer::Type)type);
            vis.m_rand.m_state.a = orc_pixel_seed(x, y, W, seed); // the visualiser's generator, per pixel like the path tracer's
            vis.TraceBlock(*image, x, y, x, y, *r.scene, r.camera);
        }
    const uint32_t rw = x1 - x0 + 1, rh = y1 - y0 + 1;
    std::vector<float> o((size_t)rw * rh * 3);
    const float* px = refGlueImagePixels(image);
    for (uint32_t y = 0; y < rh; y++) memcpy(&o[(size_t)y * rw * 3], &px[((size_t)(y0 + y) * W + x0) * 3], (size_t)rw * 12);
    writeAll(out, o.data(), o.size() * 4);
    (void)H;
    return 0;
}

// ------------------------------------------------------------------ InfiniteAreaLight::create + sample (light.cpp:30-128)
// in: env file (see ref_glue.cpp Texture::loadExr), u file = f32 pairs (u.x, u.y)
// out: i32 width, i32 height, f32 verticalP[height], f32 horizontalP[width*height], then per pair f32 dir[3], color[3]
static int cmdEnvLight(const char* envPath, const char* uPath, const char* out)
{
    InfiniteAreaLight light;
    light.init();
    light.create(envPath);
    std::vector<float> uv = readFloats(uPath);
    const float* u = uv.data();
    size_t n = uv.size() / 2;
    std::vector<float> o;
    int32_t wh[2] = {light.m_width, light.m_height};
    float tmp[2];
    memcpy(tmp, wh, 8);
    o.push_back(tmp[0]); o.push_back(tmp[1]);
    o.insert(o.end(), light.m_verticalP, light.m_verticalP + light.m_height);
    o.insert(o.end(), light.m_horizontalP, light.m_horizontalP + (size_t)light.m_width * light.m_height);
    for (size_t i = 0; i < n; i++) {
        Vector3f dir, color;
        light.sample(dir, color, Vector2f(u[2 * i], u[2 * i + 1]));
        float rec[6] = {dir.x, dir.y, dir.z, color.x, color.y, color.z};
        o.insert(o.end(), rec, rec + 6);
    }
    writeAll(out, o.data(), o.size() * 4);
    return 0;
}

// Cornell box data as the reference's SampleModels::getCornellBox builds it (sample_models.cpp:11-207):
// out: u32 primCount, vertexCount, materialCount; u32 indices[]; f32 positions[]; u32 primMaterial[]; materials as orc_material
static int cmdCornell(const char* out)
{
    Mesh mesh = SampleModels::getCornellBox(true);
    std::vector<uint8_t> buf;
    auto put = [&](const void* p, size_t n) { buf.insert(buf.end(), (const uint8_t*)p, (const uint8_t*)p + n); };
    uint32_t pc = mesh.getPrimCount(), vc = mesh.getVertexCount(), mc = mesh.getMaterialCount();
    put(&pc, 4); put(&vc, 4); put(&mc, 4);
    put(mesh.m_indices, (size_t)pc * 12);
    put(mesh.m_positions, (size_t)vc * 12);
    put(mesh.m_primMaterial, (size_t)pc * 4);
    for (uint32_t i = 0; i < mc; i++) {
        const Material& m = mesh.m_materials[i];
        orc_material o;
        memset(&o, 0, sizeof(o));
        memcpy(o.diffuse, &m.diffuse, 12);
        memcpy(o.emissive, &m.emissive, 12);
        o.reflectionType = (uint32_t)m.reflectionType;
        o.alphaTest = m.alphaTest;
        o.diffuseMap = -1; o.bumpMap = -1;
        put(&o, sizeof(o));
    }
    writeAll(out, buf.data(), buf.size());
    return 0;
}
#endif

int main(int argc, char** argv)
{
    if (argc >= 4 && !strcmp(argv[1], "leaf")) return cmdLeaf(argv[2], argv[3]);
    if (argc >= 4 && !strcmp(argv[1], "bvh")) return cmdBvh(argv[2], argv[3], argc >= 5 && !strcmp(argv[4], "threaded"));
    if (argc >= 5 && !strcmp(argv[1], "rays")) return cmdRays(argv[2], argv[3], argv[4]);
    if (argc >= 7 && !strcmp(argv[1], "camera"))
        return cmdCamera(argv[2], (uint32_t)strtoul(argv[3], 0, 10), (uint32_t)strtoul(argv[4], 0, 10), (uint32_t)strtoul(argv[5], 0, 10), argv[6]);
#ifdef REF_WITH_GLUE
    if (argc >= 11 && !strcmp(argv[1], "render"))
        return cmdRender(argv[2], (uint32_t)atoi(argv[3]), (uint32_t)atoi(argv[4]), (uint32_t)atoi(argv[5]), (uint32_t)atoi(argv[6]),
                         (uint32_t)atoi(argv[7]), (uint32_t)strtoul(argv[8], 0, 10), atoi(argv[9]), argv[10], argc >= 12 ? argv[11] : nullptr);
    if (argc >= 5 && !strcmp(argv[1], "envlight")) return cmdEnvLight(argv[2], argv[3], argv[4]);
    if (argc >= 10 && !strcmp(argv[1], "gbuffer"))
        return cmdGbuffer(argv[2], atoi(argv[3]), (uint32_t)atoi(argv[4]), (uint32_t)atoi(argv[5]), (uint32_t)atoi(argv[6]), (uint32_t)atoi(argv[7]),
                          (uint32_t)strtoul(argv[8], 0, 10), argv[9]);
    if (argc >= 3 && !strcmp(argv[1], "cornell")) return cmdCornell(argv[2]);
#endif
    fprintf(stderr,
            "usage: %s leaf <in> <out> | bvh <scene> <out> [threaded] | rays <scene> <in> <out> | camera <scene> x y state <out>"
#ifdef REF_WITH_GLUE
            " | render <scene> spp x0 y0 x1 y1 seed threads <out> | cornell <out>"
#endif
            "\n", argv[0]);
    return 1;
}
