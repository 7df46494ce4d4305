// prt_host.cpp -- host-side classes behind prt.h: scene bookkeeping, camera set-up, image
// container, thread pool and PathTracer::TraceBlock (the single entry into the GPU hot path).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

#include "prt.h"

namespace prt
{

void logPrintf(LogLevel level, const char* format...)
{
    // kVerbose lines (the reference prints BVH statistics and load messages, log.cpp:10-25) go to stderr and only when
    // PRT_VERBOSE is set, so that a host's stdout stays its own
    static const bool verbose = getenv("PRT_VERBOSE") != nullptr;
    if (level == LogLevel::kVerbose) {
        if (!verbose) return;
        va_list args;
        va_start(args, format);
        vfprintf(stderr, format, args);
        va_end(args);
        return;
    }
    if (level == LogLevel::kError) printf("ERROR: ");
    va_list args;
    va_start(args, format);
    vfprintf(stdout, format, args);
    va_end(args);
}

// ---------------------------------------------------------------- texture / material
void Texture::create(uint16_t w, uint16_t h, uint8_t comp, const uint8_t* data)
{
    width = w;
    height = h;
    component = comp;
    texels = std::make_shared<std::vector<uint8_t>>(data, data + (size_t)w * h * comp);
}

bool Texture::isAlphaTestRequired() const // texture.cpp:338-350
{
    if (!isValid() || component != 4) return false;
    for (size_t i = 0; i < (size_t)width * height; ++i)
        if ((*texels)[i * 4 + 3] < 255) return true;
    return false;
}

void Material::init() // material.cpp:30-43
{
    diffuse = Vector3f(0.0f);
    ambient = Vector3f(0.0f);
    specular = Vector3f(0.0f);
    emissive = Vector3f(0.0f);
    diffuseMap.init();
    ambientMap.init();
    specularMap.init();
    emissiveMap.init();
    bumpMap.init();
    reflectionType = ReflectionType::kDiffuse;
    alphaTest = false;
}

// ---------------------------------------------------------------- mesh
void Mesh::create(uint32_t primCount, uint32_t vertexCount, uint32_t materialCount, bool hasVertexNormal)
{
    m_indices.assign((size_t)primCount * kVertexCountPerPrim, 0);
    m_positions.assign(vertexCount, Vector3f(0.0f));
    m_texcoords.assign(vertexCount, Vector2f(0.0f));
    m_normals.clear();
    if (hasVertexNormal) m_normals.assign(vertexCount, Vector3f(0.0f));
    m_primMaterial.assign(primCount, 0);
    m_materials.resize(materialCount);
    for (auto& m : m_materials) m.init();
    m_hasVertexNormal = hasVertexNormal;
    m_hasTexcoord = false;
}

// mesh.cpp:108-149: prims in reverse order; an accumulated normal is never allowed to become zero
void Mesh::calculateVertexNormals()
{
    const uint32_t vertexCount = getVertexCount();
    m_normals.assign(vertexCount, Vector3f(0.0f));
    for (int32_t i = (int32_t)getPrimCount() - 1; i >= 0; i--) {
        uint32_t v[3];
        Vector3f p[3];
        for (uint32_t j = 0; j < 3; j++) {
            v[j] = m_indices[3 * i + j];
            p[j] = m_positions[v[j]];
        }
        auto normal = normalize(cross(p[1] - p[0], p[2] - p[0]));
        if (std::isnan(normal.x) || std::isnan(normal.y) || std::isnan(normal.z)) normal.set(0.0f, 0.0f, 0.0f);
        for (uint32_t j = 0; j < 3; j++) {
            auto n = m_normals[v[j]] + normal;
            if (length(n) > 0.0f) m_normals[v[j]] = n;
        }
    }
    for (uint32_t i = 0; i < vertexCount; i++) m_normals[i] = normalize(m_normals[i]);
    m_hasVertexNormal = true;
}

void Mesh::calculateBounds() // mesh.cpp:302-309
{
    BBox bbox = BBox::init();
    for (const auto& p : m_positions) bbox.merge(p);
    m_bbox = bbox;
}

// ---------------------------------------------------------------- scene (scene.cpp:9-27)
uint64_t Scene::nextRevision()
{
    static std::atomic<uint64_t> counter{0};
    return ++counter; // never 0, never repeated: a new Scene at a reused address cannot look like the old one
}

Scene::Scene() : m_revision(nextRevision()) {}

void Scene::init()
{
    m_directionalLight.init();
    m_infiniteAreaLight.init();
    m_availableLights = 0;
    m_bbox = BBox::init();
    m_radius = std::numeric_limits<float>::max();
    m_bvh.clear();
    m_revision = nextRevision();
}

void Scene::add(Bvh* bvh)
{
    bvh->m_mesh.m_id = (uint32_t)m_bvh.size();
    m_bvh.push_back(bvh);
    m_bbox.merge(bvh->m_mesh.getBBox());
    auto center = m_bbox.center();
    m_radius = length(m_bbox.upper - center);
    m_revision = nextRevision();
}

void Scene::setDirectionalLight(const Vector3f& dir, const Vector3f& intensity) // scene.h:30-35
{
    m_directionalLight = {dir, intensity};
    m_availableLights |= (1u << (uint32_t)LightType::kDirectional);
    m_availableLights &= ~(1u << (uint32_t)LightType::kInfiniteArea);
    m_revision = nextRevision();
}

void Scene::setInfiniteAreaLight(const char* path) // scene.h:42-45
{
    m_infiniteAreaLight.create(path);
    if (!m_infiniteAreaLight.isValid()) return;
    m_availableLights |= (1u << (uint32_t)LightType::kInfiniteArea);
    m_revision = nextRevision();
}

void Scene::setInfiniteAreaLight(int32_t width, int32_t height, const float* rgba)
{
    m_infiniteAreaLight.create(width, height, rgba);
    if (!m_infiniteAreaLight.isValid()) return;
    m_availableLights |= (1u << (uint32_t)LightType::kInfiniteArea);
    m_revision = nextRevision();
}

// ---- light.cpp:13-84 ----
static const float kPi = 3.14159265358979323846f; // vecmath.h:162
void InfiniteAreaLight::init() { release(); }

void InfiniteAreaLight::release()
{
    m_texels.clear();
    m_verticalP.clear();
    m_horizontalP.clear();
    m_width = m_height = 0;
}

// ---- OpenEXR input (texture.cpp:256-310 decodes the environment map with tinyexr's LoadEXR: RGBA float32, rows top to bottom).
// tinyexr is absent here, so the format is read directly: single-part scan-line files with NO, RLE, ZIPS or ZIP compression
// and UINT / HALF / FLOAT channels sampled 1:1.  Channels R, G, B (and A) by name; a file with one channel is grey, a missing A
// is 1, as LoadEXR fills them.  PIZ / PXR24 / B44 / DWA, tiles, deep and multi-part files are refused with a message.
bool inflateZlibBytes(const std::vector<uint8_t>& in, std::vector<uint8_t>& out); // prt_models.cpp (the PNG reader's inflate)

static float halfToFloat(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h >> 15) << 31, exp = (h >> 10) & 31u, man = h & 1023u;
    uint32_t u;
    if (exp == 0) {
        if (man == 0) {
            u = sign;
        } else { // subnormal half: normalise
            int e = -1;
            uint32_t m = man;
            do {
                e++;
                m <<= 1;
            } while (!(m & 1024u));
            u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 1023u) << 13);
        }
    } else if (exp == 31) {
        u = sign | 0x7f800000u | (man << 13);
    } else {
        u = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static bool loadExrRgba(const std::vector<uint8_t>& file, int& width, int& height, std::vector<float>& rgba, std::string& err)
{
    size_t pos = 0;
    auto need = [&](size_t n) { return pos + n <= file.size(); };
    auto rdI = [&](int32_t& v) { if (!need(4)) return false; memcpy(&v, &file[pos], 4); pos += 4; return true; };
    auto rdStr = [&](std::string& t) {
        t.clear();
        while (pos < file.size() && file[pos]) t.push_back((char)file[pos++]);
        if (pos >= file.size()) return false;
        pos++;
        return true;
    };
    int32_t magic = 0, version = 0;
    if (!rdI(magic) || !rdI(version) || magic != 20000630) { err = "not an OpenEXR file"; return false; }
    if ((version & 0xff) != 2 || (version & (0x200 | 0x800 | 0x1000))) { err = "tiled, deep or multi-part OpenEXR files are not supported"; return false; }
    struct Channel { std::string name; int32_t type; };
    std::vector<Channel> channels;
    int32_t compression = -1, x0 = 0, y0 = 0, x1 = -1, y1 = -1;
    for (;;) {
        std::string name, type;
        if (!rdStr(name)) { err = "truncated header"; return false; }
        if (name.empty()) break;
        int32_t size = 0;
        if (!rdStr(type) || !rdI(size) || size < 0 || !need((size_t)size)) { err = "truncated header"; return false; }
        const size_t end = pos + (size_t)size;
        if (name == "channels") {
            for (;;) {
                std::string cn;
                if (!rdStr(cn) || pos > end) { err = "bad channel list"; return false; }
                if (cn.empty()) break;
                int32_t ct = 0, xs = 0, ys = 0;
                if (!rdI(ct) || !need(4)) { err = "bad channel list"; return false; }
                pos += 4; // pLinear + reserved
                if (!rdI(xs) || !rdI(ys) || pos > end) { err = "bad channel list"; return false; }
                if (xs != 1 || ys != 1) { err = "subsampled channels are not supported"; return false; }
                if (ct < 0 || ct > 2) { err = "unknown channel type"; return false; }
                channels.push_back({cn, ct});
            }
        } else if (name == "compression" && size >= 1) {
            compression = file[pos];
        } else if (name == "dataWindow" && size >= 16) {
            memcpy(&x0, &file[pos], 4); memcpy(&y0, &file[pos + 4], 4); memcpy(&x1, &file[pos + 8], 4); memcpy(&y1, &file[pos + 12], 4);
        }
        pos = end;
    }
    if (channels.empty() || compression < 0 || x1 < x0 || y1 < y0) { err = "header lacks channels, compression or dataWindow"; return false; }
    if (compression > 3) { err = "only NO / RLE / ZIPS / ZIP compression is read (this file uses PIZ, PXR24, B44 or DWA)"; return false; }
    const int64_t w = (int64_t)x1 - x0 + 1, h = (int64_t)y1 - y0 + 1;
    if (w > 65536 || h > 65536) { err = "image too large"; return false; }
    width = (int)w;
    height = (int)h;
    // where each channel goes; the file stores the channels of a line one after the other, in the order of the list
    size_t lineBytes = 0;
    std::vector<size_t> chanOffset(channels.size());
    std::vector<int> target(channels.size(), -1);
    for (size_t c = 0; c < channels.size(); c++) {
        chanOffset[c] = lineBytes;
        lineBytes += (size_t)w * (channels[c].type == 1 ? 2u : 4u);
        const std::string& n = channels[c].name;
        if (n == "R") target[c] = 0;
        else if (n == "G") target[c] = 1;
        else if (n == "B") target[c] = 2;
        else if (n == "A") target[c] = 3;
    }
    const bool grey = channels.size() == 1 && target[0] < 0;
    rgba.assign((size_t)w * h * 4, 0.0f);
    for (size_t i = 3; i < rgba.size(); i += 4) rgba[i] = 1.0f;
    const uint32_t linesPerBlock = compression == 3 ? 16u : 1u, blocks = ((uint32_t)h + linesPerBlock - 1) / linesPerBlock;
    if (!need((size_t)blocks * 8)) { err = "truncated offset table"; return false; }
    const size_t table = pos;
    std::vector<uint8_t> packed, raw, tmp;
    for (uint32_t b = 0; b < blocks; b++) {
        uint64_t off = 0;
        memcpy(&off, &file[table + (size_t)b * 8], 8);
        if (file.size() < 8 || off > file.size() - 8) { err = "bad block offset"; return false; } // (no `off + 8`: a crafted offset near 2^64 wraps)
        int32_t y = 0, size = 0;
        memcpy(&y, &file[off], 4);
        memcpy(&size, &file[off + 4], 4);
        if (size < 0 || (uint64_t)size > file.size() - 8 - off || y < y0 || y > y1) { err = "bad block"; return false; }
        const uint32_t lines = std::min<uint32_t>(linesPerBlock, (uint32_t)(y1 - y + 1));
        const size_t expect = lineBytes * lines;
        const uint8_t* data = &file[off + 8];
        if ((size_t)size == expect || compression == 0) { // stored as is (also what a writer does when packing does not help)
            if ((size_t)size != expect) { err = "bad block size"; return false; }
            raw.assign(data, data + size);
        } else {
            tmp.clear();
            if (compression == 1) { // RLE: a count byte; negative = that many literal bytes, else count + 1 copies of the next byte
                for (int32_t i = 0; i < size;) {
                    const int8_t n = (int8_t)data[i++];
                    if (n < 0) {
                        const int32_t k = -(int32_t)n;
                        if (i + k > size) { err = "bad RLE block"; return false; }
                        tmp.insert(tmp.end(), data + i, data + i + k);
                        i += k;
                    } else {
                        if (i >= size) { err = "bad RLE block"; return false; }
                        tmp.insert(tmp.end(), (size_t)n + 1, data[i++]);
                    }
                }
            } else {
                packed.assign(data, data + size);
                if (!inflateZlibBytes(packed, tmp)) { err = "bad ZIP block"; return false; }
            }
            if (tmp.size() != expect) { err = "block does not unpack to its size"; return false; }
            for (size_t k = 1; k < tmp.size(); k++) tmp[k] = (uint8_t)(tmp[k - 1] + tmp[k] - 128); // predictor
            raw.resize(expect);
            const size_t half = (expect + 1) / 2; // first half: the even bytes, second half: the odd ones
            for (size_t k = 0; k < expect; k++) raw[k] = tmp[(k & 1) ? half + k / 2 : k / 2];
        }
        for (uint32_t l = 0; l < lines; l++) {
            float* row = &rgba[(size_t)(y - y0 + (int32_t)l) * w * 4];
            for (size_t c = 0; c < channels.size(); c++) {
                if (target[c] < 0 && !grey) continue;
                const uint8_t* src = &raw[(size_t)l * lineBytes + chanOffset[c]];
                for (int64_t x = 0; x < w; x++) {
                    float v;
                    if (channels[c].type == 1) {
                        uint16_t hv;
                        memcpy(&hv, src + 2 * x, 2);
                        v = halfToFloat(hv);
                    } else if (channels[c].type == 2) {
                        memcpy(&v, src + 4 * x, 4);
                    } else {
                        uint32_t u;
                        memcpy(&u, src + 4 * x, 4);
                        v = (float)u;
                    }
                    if (grey) row[4 * x] = row[4 * x + 1] = row[4 * x + 2] = v;
                    else row[4 * x + target[c]] = v;
                }
            }
        }
    }
    return true;
}

// Besides OpenEXR this build reads a PFM ("PF", width height, scale;
void InfiniteAreaLight::create(const char* path)
{
    release();
    FILE* f = fopen(path, "rb");
    if (!f) {
        logPrintf(LogLevel::kError, "Failed to open '%s'\n", path);
        return;
    }
    uint8_t head[4] = {0, 0, 0, 0};
    const size_t headGot = fread(head, 1, 4, f);
    if (headGot == 4 && head[0] == 0x76 && head[1] == 0x2f && head[2] == 0x31 && head[3] == 0x01) { // OpenEXR
        fseek(f, 0, SEEK_END);
        const long bytes = ftell(f);
        fseek(f, 0, SEEK_SET);
        std::vector<uint8_t> file(bytes > 0 ? (size_t)bytes : 0);
        const size_t got = file.empty() ? 0 : fread(file.data(), 1, file.size(), f);
        fclose(f);
        int w = 0, h = 0;
        std::vector<float> rgba;
        std::string err;
        if (got != file.size() || !loadExrRgba(file, w, h, rgba, err)) {
            logPrintf(LogLevel::kError, "Load EXR err: %s(%s)\n", got != file.size() ? "short read" : err.c_str(), path); // texture.cpp:263
            return;
        }
        logPrintf(LogLevel::kVerbose, "Loaded .exr '%s' (%d, %d)\n", path, w, h);
        create(w, h, rgba.data());
        return;
    }
    fseek(f, 0, SEEK_SET);
    char magic[3] = {0, 0, 0};
    int w = 0, h = 0;
    float scale = 0.0f;
    if (fscanf(f, "%2s %d %d %f", magic, &w, &h, &scale) != 4 || strcmp(magic, "PF") != 0 || w <= 0 || h <= 0 || scale == 0.0f) {
        logPrintf(LogLevel::kError, "'%s' is not an RGB PFM file\n", path);
        fclose(f);
        return;
    }
    fgetc(f); // the single whitespace after the header
    std::vector<float> rgb((size_t)w * h * 3);
    size_t got = fread(rgb.data(), sizeof(float), rgb.size(), f);
    fclose(f);
    if (got != rgb.size()) {
        logPrintf(LogLevel::kError, "'%s' is truncated\n", path);
        return;
    }
    if (scale > 0.0f) // big-endian samples
        for (float& v : rgb) {
            uint32_t u;
            memcpy(&u, &v, 4);
            u = (u >> 24) | ((u >> 8) & 0xff00u) | ((u << 8) & 0xff0000u) | (u << 24);
            memcpy(&v, &u, 4);
        }
    std::vector<float> rgba((size_t)w * h * 4);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float* s = &rgb[((size_t)(h - 1 - y) * w + x) * 3];
            float* d = &rgba[((size_t)y * w + x) * 4];
            d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = 1.0f;
        }
    logPrintf(LogLevel::kVerbose, "Loaded .pfm '%s' (%d, %d)\n", path, w, h);
    create(w, h, rgba.data());
}

void InfiniteAreaLight::create(int32_t width, int32_t height, const float* rgba) // light.cpp:34-84
{
    release();
    if (width <= 0 || height <= 0 || !rgba) return;
    m_width = width;
    m_height = height;
    m_texels.assign(rgba, rgba + (size_t)width * height * 4);
    m_verticalP.resize(height);
    m_horizontalP.resize((size_t)width * height);
    float* vert = m_verticalP.data();
    float* hori = m_horizontalP.data();
    const float* p = m_texels.data();
    float vsum = 0.0f;
    for (uint32_t y = 0; y < (uint32_t)height; y++) {
        float hsum = 0.0f;
        for (uint32_t x = 0; x < (uint32_t)width; x++) {
            uint32_t indexBase = x + y * (uint32_t)width;
            Vector3f c(p[4 * indexBase + 0], p[4 * indexBase + 1], p[4 * indexBase + 2]);
            float l = length(c);
            hori[indexBase] = l;
            hsum += l;
        }
        float sinPhi = std::sin(kPi * (y + 0.5f) / height);
        vert[y] = hsum * sinPhi;
        vsum += hsum * sinPhi;
        float invH = 1.0f / hsum;
        float accumH = 0.0f;
        for (uint32_t x = 0; x < (uint32_t)width; x++) {
            uint32_t indexBase = x + y * (uint32_t)width;
            float ph = accumH + invH * hori[indexBase];
            hori[indexBase] = ph;
            accumH = ph;
        }
    }
    float invV = 1.0f / vsum;
    float accumV = 0.0f;
    for (uint32_t y = 0; y < (uint32_t)height; y++) {
        float pv = accumV + invV * vert[y];
        vert[y] = pv;
        accumV = pv;
    }
}

void Scene::describe(prt_scene_desc& desc, DescStorage& store) const
{
    store.meshes.clear();
    store.materials.clear();
    store.textures.clear();
    store.texelRefs.clear();
    std::map<const std::vector<uint8_t>*, int32_t> texIndex;
    auto addTex = [&](const Texture& t) -> int32_t {
        if (!t.isValid()) return -1;
        auto it = texIndex.find(t.texels.get());
        if (it != texIndex.end()) return it->second;
        prt_texture_desc d{(int32_t)t.width, (int32_t)t.height, (int32_t)t.component, t.texels->data()};
        store.textures.push_back(d);
        store.texelRefs.push_back(t.texels);
        int32_t id = (int32_t)store.textures.size() - 1;
        texIndex[t.texels.get()] = id;
        return id;
    };
    store.materials.resize(m_bvh.size());
    for (size_t b = 0; b < m_bvh.size(); b++) {
        const Bvh* bvh = m_bvh[b];
        const Mesh& m = bvh->m_mesh;
        auto& mats = store.materials[b];
        for (uint32_t k = 0; k < m.getMaterialCount(); k++) {
            const Material& s = m.getMaterial(k);
            prt_material d;
            memcpy(d.diffuse, &s.diffuse, 12);
            memcpy(d.emissive, &s.emissive, 12);
            d.reflectionType = (uint32_t)s.reflectionType;
            d.alphaTest = s.alphaTest ? 1u : 0u;
            d.diffuseMap = addTex(s.diffuseMap);
            d.bumpMap = addTex(s.bumpMap);
            mats.push_back(d);
        }
        prt_mesh_desc md;
        md.nodeCount = (uint32_t)bvh->m_nodes.size();
        md.nodes = bvh->m_nodes.data();
        md.primCount = m.getPrimCount();
        md.primRemapping = bvh->m_primRemapping.data();
        md.vertexCount = m.getVertexCount();
        md.indices = m.m_indices.data();
        md.positions = &m.m_positions[0].x;
        md.normals = m.hasVertexNormal() ? &m.m_normals[0].x : nullptr;
        md.texcoords = m.hasTexcoord() ? &m.m_texcoords[0].x : nullptr;
        md.materialCount = m.getMaterialCount();
        md.primMaterial = m.m_primMaterial.data();
        md.materials = mats.data();
        store.meshes.push_back(md);
    }
    desc.meshCount = (uint32_t)store.meshes.size();
    desc.meshes = store.meshes.data();
    desc.textureCount = (uint32_t)store.textures.size();
    desc.textures = store.textures.data();
    desc.hasDirectionalLight = isLightAvailable(LightType::kDirectional) ? 1u : 0u;
    memcpy(desc.lightDir, &m_directionalLight.dir, 12);
    memcpy(desc.lightIntensity, &m_directionalLight.intensity, 12);
    desc.radius = m_radius;
    const bool env = isLightAvailable(LightType::kInfiniteArea) && m_infiniteAreaLight.isValid();
    desc.hasInfiniteAreaLight = env ? 1u : 0u;
    desc.envWidth = env ? m_infiniteAreaLight.getWidth() : 0;
    desc.envHeight = env ? m_infiniteAreaLight.getHeight() : 0;
    desc.envTexels = env ? m_infiniteAreaLight.getTexels().data() : nullptr;
    desc.envVerticalP = env ? m_infiniteAreaLight.getVerticalP().data() : nullptr;
    desc.envHorizontalP = env ? m_infiniteAreaLight.getHorizontalP().data() : nullptr;
}

// ---------------------------------------------------------------- camera (camera.h:17-36)
void Camera::create(const Vector3f& pos, const Vector3f& dir, uint32_t width, uint32_t height)
{
    m_pos = pos;
    m_dir = normalize(dir);
    m_width = width;
    m_height = height;
    m_invWidth = 1.0f / width;
    m_invHeight = 1.0f / height;
    auto up = Vector3f(0, 1.0f, 0);
    auto right = cross(dir, up);
    if (length(right) < 0.00001f) right = cross(dir, Vector3f(1, 0, 0));
    right = normalize(right);
    up = normalize(cross(right, dir));
    m_up = up;
    m_right = right;
}

void Camera::describe(prt_camera_desc& d) const
{
    memcpy(d.pos, &m_pos, 12);
    memcpy(d.dir, &m_dir, 12);
    memcpy(d.up, &m_up, 12);
    memcpy(d.right, &m_right, 12);
    d.width = m_width;
    d.height = m_height;
    d.invWidth = m_invWidth;
    d.invHeight = m_invHeight;
}

// ---------------------------------------------------------------- image (image.cpp:29-80)
Image::Image(uint32_t width, uint32_t height, bool tonemap, float exposure)
    : m_pixels((size_t)3 * width * height, 0.0f), m_width(width), m_height(height), m_tonemap(tonemap), m_exposure(exposure)
{
}

void Image::writePixel(uint32_t x, uint32_t y, const Vector3f& color)
{
    auto c = m_exposure * color;
    const size_t indexBase = ((size_t)x + (size_t)y * m_width) * 3;
    m_pixels[indexBase + 0] = c.x;
    m_pixels[indexBase + 1] = c.y;
    m_pixels[indexBase + 2] = c.z;
}

void Image::savePpm(const char* path) const
{
    FILE* fp = fopen(path, "wb");
    if (!fp) {
        printf("Error saving %s\n", path);
        return;
    }
    std::vector<uint8_t> pixels((size_t)3 * m_width * m_height);
    for (size_t i = 0; i < (size_t)m_width * m_height; i++) {
        for (int k = 0; k < 3; k++) {
            float c = m_pixels[3 * i + k];
            c = m_tonemap ? c / (c + 1) : c;                 // image.cpp:64
            c = std::fmin(std::fmax(c, 0.0f), 1.0f);
            pixels[3 * i + k] = (uint8_t)(powf(c, 1 / 2.2f) * 0xff); // image.cpp:65
        }
    }
    fprintf(fp, "P6\n%d %d\n255\n", m_width, m_height);
    fwrite(pixels.data(), pixels.size(), 1, fp);
    fclose(fp);
    printf("Save %s\n", path);
}

void Image::savePfm(const char* path) const
{
    FILE* fp = fopen(path, "wb");
    if (!fp) {
        printf("Error saving %s\n", path);
        return;
    }
    fprintf(fp, "PF\n%d %d\n-1.0\n", m_width, m_height);
    for (int32_t y = (int32_t)m_height - 1; y >= 0; y--) fwrite(&m_pixels[(size_t)y * m_width * 3], 4, (size_t)m_width * 3, fp);
    fclose(fp);
    printf("Save %s\n", path);
}

// float -> IEEE half, round to nearest even; overflow to infinity, NaN kept a NaN
static uint16_t floatToHalf(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u | ((x >> 13) & 0x3ffu) : 0u));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); // rounds to 2^16 or beyond
    if (x < 0x33000001u) return (uint16_t)sign;               // below half of the smallest subnormal (ties to even -> 0)
    uint32_t e = x >> 23, m = (x & 0x7fffffu) | 0x800000u;
    if (e < 113) { // half subnormal: shift the 24-bit significand down to 10 bits at exponent -14
        const uint32_t shift = 126 - e; // 14..24
        uint32_t h = m >> shift, rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((e - 112) << 10) | ((m >> 13) & 0x3ffu), rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++; // may carry into the exponent: still the right value
    return (uint16_t)(sign | h);
}

// ---- a small deflate (RFC 1951) for the EXR writer: LZ77 over a hash of 3-byte strings, fixed Huffman codes, zlib framing
namespace
{
struct BitWriter {
    std::vector<uint8_t>& out;
    uint32_t acc = 0;
    int cnt = 0;
    void put(uint32_t v, int n) // n bits, least significant first
    {
        acc |= v << cnt;
        cnt += n;
        while (cnt >= 8) {
            out.push_back((uint8_t)acc);
            acc >>= 8;
            cnt -= 8;
        }
    }
    void putCode(uint32_t code, int n) // Huffman codes go most significant bit first
    {
        uint32_t r = 0;
        for (int i = 0; i < n; i++) r |= ((code >> i) & 1u) << (n - 1 - i);
        put(r, n);
    }
    void flush()
    {
        if (cnt) out.push_back((uint8_t)acc);
        acc = 0;
        cnt = 0;
    }
};

void zlibCompress(const uint8_t* in, size_t n, std::vector<uint8_t>& out)
{
    out.clear();
    out.push_back(0x78);
    out.push_back(0x9c);
    BitWriter bw{out};
    bw.put(1, 1); // last block
    bw.put(1, 2); // fixed Huffman codes
    auto literal = [&](uint32_t v) {
        if (v < 144) bw.putCode(0x30 + v, 8);
        else bw.putCode(0x190 + (v - 144), 9);
    };
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    std::vector<int32_t> head(1 << 15, -1), prev(n ? n : 1, -1);
    auto hash3 = [&](size_t i) { return ((uint32_t)in[i] * 2654435761u ^ (uint32_t)in[i + 1] * 40503u ^ (uint32_t)in[i + 2] * 2246822519u) >> 17; };
    size_t i = 0;
    while (i < n) {
        size_t bestLen = 0, bestDist = 0;
        if (i + 3 <= n) {
            const uint32_t hsh = hash3(i);
            int tries = 32;
            for (int32_t c = head[hsh]; c >= 0 && tries-- > 0 && i - (size_t)c <= 32768; c = prev[c]) {
                size_t l = 0;
                while (l < 258 && i + l < n && in[c + l] == in[i + l]) l++;
                if (l > bestLen) {
                    bestLen = l;
                    bestDist = i - (size_t)c;
                    if (l == 258) break;
                }
            }
        }
        const size_t step = bestLen >= 3 ? bestLen : 1;
        if (bestLen >= 3) {
            int ls = 28;
            while (lbase[ls] > bestLen) ls--;
            const uint32_t sym = 257 + ls;
            if (sym < 280) bw.putCode(sym - 256, 7);
            else bw.putCode(0xc0 + (sym - 280), 8);
            bw.put((uint32_t)(bestLen - lbase[ls]), lext[ls]);
            int ds = 29;
            while (dbase[ds] > bestDist) ds--;
            bw.putCode((uint32_t)ds, 5);
            bw.put((uint32_t)(bestDist - dbase[ds]), dext[ds]);
        } else {
            literal(in[i]);
        }
        for (size_t k = 0; k < step; k++, i++)
            if (i + 3 <= n) {
                const uint32_t hsh = hash3(i);
                prev[i] = head[hsh];
                head[hsh] = (int32_t)i;
            }
    }
    bw.putCode(0, 7); // end of block
    bw.flush();
    uint32_t a = 1, bsum = 0; // Adler-32
    for (size_t k = 0; k < n; k++) {
        a = (a + in[k]) % 65521u;
        bsum = (bsum + a) % 65521u;
    }
    const uint32_t adler = (bsum << 16) | a;
    for (int sh = 24; sh >= 0; sh -= 8) out.push_back((uint8_t)(adler >> sh));
}
} // namespace

// image.cpp:82-139: three half-float channels named B, G, R (the reference converts float -> half through tinyexr, whose
// SaveEXRImageToFile compresses with ZIP by default).  Written here directly as a scan-line OpenEXR 2.0 file: magic, version,
// header attributes, offset table, then blocks of 16 scan lines (ZIP: bytes split into even / odd halves, delta-predicted,
// deflated; a block that does not shrink is stored raw, as the format prescribes) or of one raw scan line (zip = false).
void Image::saveExr(const char* path, bool zip) const
{
    FILE* fp = fopen(path, "wb");
    if (!fp) {
        printf("Error saving %s\n", path);
        return;
    }
    std::vector<uint8_t> hd;
    auto put = [&](const void* p, size_t n) { hd.insert(hd.end(), (const uint8_t*)p, (const uint8_t*)p + n); };
    auto putStr = [&](const char* t) { put(t, strlen(t) + 1); };
    auto putI = [&](int32_t v) { put(&v, 4); };
    auto putF = [&](float v) { put(&v, 4); };
    auto attr = [&](const char* name, const char* type, int32_t size) { putStr(name); putStr(type); putI(size); };
    putI(20000630);
    putI(2);
    attr("channels", "chlist", 3 * 18 + 1);
    for (const char* c : {"B", "G", "R"}) {
        putStr(c);
        putI(1); // HALF
        const uint8_t linear[4] = {0, 0, 0, 0};
        put(linear, 4);
        putI(1);
        putI(1);
    }
    hd.push_back(0);
    attr("compression", "compression", 1); hd.push_back(zip ? 3 : 0); // ZIP_COMPRESSION (16 lines per block) / NO_COMPRESSION
    attr("dataWindow", "box2i", 16); putI(0); putI(0); putI((int32_t)m_width - 1); putI((int32_t)m_height - 1);
    attr("displayWindow", "box2i", 16); putI(0); putI(0); putI((int32_t)m_width - 1); putI((int32_t)m_height - 1);
    attr("lineOrder", "lineOrder", 1); hd.push_back(0); // increasing y
    attr("pixelAspectRatio", "float", 4); putF(1.0f);
    attr("screenWindowCenter", "v2f", 8); putF(0.0f); putF(0.0f);
    attr("screenWindowWidth", "float", 4); putF(1.0f);
    hd.push_back(0);
    const uint32_t linesPerBlock = zip ? 16u : 1u, blocks = (m_height + linesPerBlock - 1) / linesPerBlock;
    std::vector<std::vector<uint8_t>> payload(blocks);
    std::vector<uint16_t> raw;
    std::vector<uint8_t> shuffled, packed;
    for (uint32_t b = 0; b < blocks; b++) {
        const uint32_t y0 = b * linesPerBlock, y1 = std::min(m_height, y0 + linesPerBlock);
        raw.assign((size_t)(y1 - y0) * m_width * 3, 0);
        for (uint32_t y = y0; y < y1; y++) {
            const float* px = &m_pixels[(size_t)y * m_width * 3];
            uint16_t* line = &raw[(size_t)(y - y0) * m_width * 3];
            for (uint32_t x = 0; x < m_width; x++) {
                line[x] = floatToHalf(px[3 * x + 2]);                       // B
                line[(size_t)m_width + x] = floatToHalf(px[3 * x + 1]);     // G
                line[(size_t)2 * m_width + x] = floatToHalf(px[3 * x + 0]); // R
            }
        }
        const uint8_t* bytes = (const uint8_t*)raw.data();
        const size_t n = raw.size() * 2;
        if (zip) {
            shuffled.resize(n);
            const size_t half = (n + 1) / 2;
            for (size_t k = 0; k < n; k++) shuffled[(k & 1) ? half + k / 2 : k / 2] = bytes[k];
            for (size_t k = n; k-- > 1;) shuffled[k] = (uint8_t)(shuffled[k] - shuffled[k - 1] + 128);
            zlibCompress(shuffled.data(), n, packed);
            if (packed.size() < n) payload[b] = packed;
            else payload[b].assign(bytes, bytes + n);
        } else {
            payload[b].assign(bytes, bytes + n);
        }
    }
    uint64_t offset = hd.size() + (uint64_t)blocks * 8;
    for (uint32_t b = 0; b < blocks; b++) {
        put(&offset, 8);
        offset += 8 + payload[b].size();
    }
    fwrite(hd.data(), 1, hd.size(), fp);
    for (uint32_t b = 0; b < blocks; b++) {
        const int32_t yy = (int32_t)(b * linesPerBlock), size = (int32_t)payload[b].size();
        fwrite(&yy, 4, 1, fp);
        fwrite(&size, 4, 1, fp);
        fwrite(payload[b].data(), 1, payload[b].size(), fp);
    }
    fclose(fp);
    printf("Saved %s\n", path);
}

// ---------------------------------------------------------------- thread pool (thread_pool.cpp)
struct ThreadPool::Impl {
    std::deque<Task> tasks;
    std::mutex mutex;
    std::condition_variable cond;
    std::vector<std::thread> workers;
    bool alive = false;
    void run()
    {
        for (;;) {
            Task task;
            {
                std::unique_lock<std::mutex> g(mutex);
                cond.wait(g, [&] { return !tasks.empty() || !alive; });
                if (tasks.empty()) return;
                task = std::move(tasks.back()); // LIFO, thread_pool.cpp:77-78
                tasks.pop_back();
            }
            task();
        }
    }
};

ThreadPool::ThreadPool() : m_impl(new Impl) {}
ThreadPool::~ThreadPool()
{
    waitAllTasksDone();
    delete m_impl;
}
void ThreadPool::create(int32_t threadCount)
{
    if (threadCount <= 0) {
        int32_t n = (int32_t)std::thread::hardware_concurrency() + threadCount;
        threadCount = n > 0 ? n : 8;
    }
    m_impl->alive = true;
    for (int32_t i = 0; i < threadCount; i++) m_impl->workers.emplace_back([this] { m_impl->run(); });
}
void ThreadPool::queue(Task task)
{
    std::unique_lock<std::mutex> g(m_impl->mutex);
    m_impl->tasks.push_back(std::move(task));
    m_impl->cond.notify_one();
}
void ThreadPool::waitAllTasksDone()
{
    {
        std::unique_lock<std::mutex> g(m_impl->mutex);
        if (!m_impl->alive) return;
        m_impl->alive = false;
    }
    m_impl->cond.notify_all();
    for (auto& w : m_impl->workers) w.join();
    m_impl->workers.clear();
}
size_t ThreadPool::getTaskCount() const
{
    std::unique_lock<std::mutex> g(m_impl->mutex);
    return m_impl->tasks.size();
}

// ---------------------------------------------------------------- PathTracer -> C-ABI
namespace
{
struct DeviceSlot {
    prt_hip_ctx* ctx = nullptr;
    const Scene* scene = nullptr;
    uint64_t sceneRevision = 0;
    prt_camera_desc camera{};
    bool haveCamera = false;
    // The reference's main.cpp calls TraceBlock once per 16x16 tile from a thread pool (main.cpp:132-160).  A launch per tile
    // would pay the whole pipeline's fixed cost 8160 times for a 1080p image, so the second rectangle asked for with the
    // same scene, camera and settings renders the WHOLE image once; that and every later rectangle is served from the
    // kept frame (pixels do not depend on how the image is cut: per-pixel generator states).
    prt_render_params frameParams{};
    bool haveFrameParams = false, frameValid = false;
    uint32_t rectCalls = 0;
    std::vector<float> frame;
    prt_hip_stats firstRectStats{};
};
std::mutex g_deviceMutex;
std::map<int, DeviceSlot> g_devices;

[[noreturn]] void die(const char* what)
{
    fprintf(stderr, "prt: %s: %s\n", what, prt_hip_last_error());
    abort(); // the reference traps on failed assertions too (prt.h:7-13)
}
} // namespace

// A Scene that goes away takes its device-side cache entries with it (the frame kept for per-tile callers included).
Scene::~Scene()
{
    std::lock_guard<std::mutex> g(g_deviceMutex);
    for (auto& kv : g_devices) {
        if (kv.second.scene != this) continue;
        kv.second.scene = nullptr;
        kv.second.sceneRevision = 0;
        kv.second.frameValid = false;
        kv.second.haveFrameParams = false;
    }
}

void TotalStats::print() // stats.h:59-66
{
    const float kToM = 1.0f / 1000000.0f;
    logPrintf(LogLevel::kInfo, "Nodes traversed: %.3fM(%llu)\n", totalNodesTraversed.load() * kToM, (unsigned long long)totalNodesTraversed.load());
    logPrintf(LogLevel::kInfo, "Prims traversed: %.3fM(%llu)\n", totalPrimsTraversed.load() * kToM, (unsigned long long)totalPrimsTraversed.load());
    logPrintf(LogLevel::kInfo, "Rays traced: %.3fM(%llu)\n", totalRaysTraced.load() * kToM, (unsigned long long)totalRaysTraced.load());
    logPrintf(LogLevel::kInfo, "Occluded traced: %.3fM(%llu)\n", totalOccludedTraced.load() * kToM, (unsigned long long)totalOccludedTraced.load());
    logPrintf(LogLevel::kInfo, "Tri tested: %.3fM(%llu)\n", totalTriTested.load() * kToM, (unsigned long long)totalTriTested.load());
}

void Bvh::buildOnDevice(Mesh&& mesh, int device)
{
    if (device < 0 || prt_hip_device_count() <= device || mesh.getPrimCount() == 0) {
        build(std::move(mesh));
        return;
    }
    std::lock_guard<std::mutex> g(g_deviceMutex);
    DeviceSlot& slot = g_devices[device];
    if (!slot.ctx && prt_hip_create(device, &slot.ctx) != PRT_HIP_OK) die("prt_hip_create");
    m_mesh = std::move(mesh);
    const uint32_t n = m_mesh.getPrimCount();
    m_nodes.resize((size_t)2 * n);
    m_primRemapping.resize(n);
    uint32_t count = 0;
    if (prt_hip_build_bvh(slot.ctx, n, m_mesh.getIndexBuffer(), m_mesh.getVertexCount(), &m_mesh.getPositionBuffer()->x, m_nodes.data(), &count,
                          m_primRemapping.data(), nullptr) != PRT_HIP_OK)
        die("prt_hip_build_bvh");
    m_nodes.resize(count);
    logPrintf(LogLevel::kVerbose, "LinearBvhNode (count=%u, built on device %d)\n", (unsigned)m_nodes.size(), device);
}

void PathTracer::releaseDevice()
{
    std::lock_guard<std::mutex> g(g_deviceMutex);
    for (auto& kv : g_devices) prt_hip_destroy(kv.second.ctx);
    g_devices.clear();
}

// The device's context with this scene and camera on it (uploaded again only when they changed).  Caller holds g_deviceMutex.
static DeviceSlot& readySlot(int device, const Scene& scene, const Camera& camera, prt_camera_desc& cd, bool& sceneChanged, bool& cameraChanged)
{
    DeviceSlot& slot = g_devices[device];
    if (!slot.ctx && prt_hip_create(device, &slot.ctx) != PRT_HIP_OK) die("prt_hip_create");
    sceneChanged = slot.scene != &scene || slot.sceneRevision != scene.getRevision();
    if (sceneChanged) {
        prt_scene_desc desc;
        Scene::DescStorage store;
        scene.describe(desc, store);
        if (prt_hip_upload_scene(slot.ctx, &desc) != PRT_HIP_OK) die("prt_hip_upload_scene");
        slot.scene = &scene;
        slot.sceneRevision = scene.getRevision();
    }
    camera.describe(cd);
    cameraChanged = !slot.haveCamera || memcmp(&cd, &slot.camera, sizeof(cd)) != 0;
    if (cameraChanged) {
        if (prt_hip_set_camera(slot.ctx, &cd) != PRT_HIP_OK) die("prt_hip_set_camera");
        slot.camera = cd;
        slot.haveCamera = true;
    }
    if (sceneChanged || cameraChanged) slot.frameValid = false;
    return slot;
}

// gbuffer_visualizer.cpp:17-26
void GbufferVisualizer::TraceBlock(Image& image, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, const Scene& scene, const Camera& camera)
{
    std::lock_guard<std::mutex> g(g_deviceMutex);
    prt_camera_desc cd;
    bool sceneChanged, cameraChanged;
    DeviceSlot& slot = readySlot(m_device, scene, camera, cd, sceneChanged, cameraChanged);
    if (image.getWidth() != cd.width || image.getHeigit() != cd.height) die("GbufferVisualizer::TraceBlock: the image and the camera differ in size");
    if (prt_hip_render_gbuffer(slot.ctx, x0, y0, x1, y1, (uint32_t)m_type, m_seed, image.getExposure(), nullptr, nullptr) != PRT_HIP_OK)
        die("prt_hip_render_gbuffer");
    if (prt_hip_download(slot.ctx, image.getPixels(), x0, y0, x1, y1) != PRT_HIP_OK) die("prt_hip_download");
    slot.frameValid = false; // the context's framebuffer no longer holds a path-traced frame
    slot.haveFrameParams = false;
}

void PathTracer::TraceBlock(Image& image, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, const Scene& scene, const Camera& camera,
                            uint32_t samples)
{
    std::lock_guard<std::mutex> g(g_deviceMutex); // calls on one context are serialised (prt_hip.h)
    prt_camera_desc cd;
    bool sceneChanged, cameraChanged;
    DeviceSlot& slot = readySlot(m_options.device, scene, camera, cd, sceneChanged, cameraChanged);
    prt_render_params p{};
    p.samples = samples;
    p.maxDepth = m_options.maxDepth;
    p.rrDepth = m_options.rrDepth;
    p.seed = m_options.seed;
    p.exposure = image.getExposure();
    p.tileSize = 16; // main.cpp:123-124
    p.rank = 0;
    p.nranks = 1;
    const uint32_t W = cd.width, H = cd.height;
    // the reference's writePixel strides by the IMAGE's width (image.cpp:44-50); rows are copied with the camera's here
    if (image.getWidth() != W || image.getHeigit() != H) die("TraceBlock: the image and the camera differ in size");
    if (sceneChanged || cameraChanged || !slot.haveFrameParams || memcmp(&p, &slot.frameParams, sizeof(p)) != 0) {
        slot.frameParams = p;
        slot.haveFrameParams = true;
        slot.frameValid = false;
        slot.rectCalls = 0;
    }
    auto copyRect = [&]() {
        const size_t row = (size_t)(x1 - x0 + 1) * 3 * sizeof(float);
        for (uint32_t y = y0; y <= y1; y++)
            memcpy(image.getPixels() + ((size_t)y * W + x0) * 3, &slot.frame[((size_t)y * W + x0) * 3], row);
    };
    if (x1 < x0 || y1 < y0 || x1 >= W || y1 >= H) die("TraceBlock: rectangle outside the image");
    if (slot.frameValid) { // a tile of a frame that has been rendered already: its rays were counted then
        copyRect();
        return;
    }
    const bool whole = (x0 == 0 && y0 == 0 && x1 == W - 1 && y1 == H - 1);
    const bool wholeNow = !whole && ++slot.rectCalls >= 2;
    const uint32_t rx0 = wholeNow ? 0 : x0, ry0 = wholeNow ? 0 : y0, rx1 = wholeNow ? W - 1 : x1, ry1 = wholeNow ? H - 1 : y1;
    if (prt_hip_render(slot.ctx, rx0, ry0, rx1, ry1, &p, nullptr, nullptr) != PRT_HIP_OK) die("prt_hip_render");
    prt_hip_stats st;
    if (wholeNow) {
        slot.frame.assign((size_t)W * H * 3, 0.0f);
        if (prt_hip_download(slot.ctx, slot.frame.data(), 0, 0, W - 1, H - 1) != PRT_HIP_OK) die("prt_hip_download");
        if (prt_hip_get_stats(slot.ctx, &st) != PRT_HIP_OK) die("prt_hip_get_stats");
        slot.frameValid = true;
        copyRect();
        // the first rectangle was rendered (and counted) on its own before: the frame's totals minus that share
        st.raysTraced -= slot.firstRectStats.raysTraced;
        st.occludedTraced -= slot.firstRectStats.occludedTraced;
    } else {
        if (prt_hip_download(slot.ctx, image.getPixels(), x0, y0, x1, y1) != PRT_HIP_OK) die("prt_hip_download");
        if (prt_hip_get_stats(slot.ctx, &st) != PRT_HIP_OK) die("prt_hip_get_stats");
        if (!whole) slot.firstRectStats = st;
    }
    m_stats.raysTraced += st.raysTraced;
    m_stats.occludedTraced += st.occludedTraced;
    m_kernelMs += st.kernelMs;
}

} // namespace prt
