// prt_models.cpp -- scene data producers: the Cornell box of the reference's default setup, an
// OBJ reader (the reference delegates to tinyobjloader, which is not vendored), and seeded
// procedural stand-ins for the assets its other setups load but does not ship.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <map>
#include <sstream>

#include "prt.h"

namespace prt
{

#include "cornell_data.inc"

static float fromBits(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// sample_models.cpp:11-207: 72 vertices, 36 triangles, 18 materials (one per quad), light emissive (17,12,4)
Mesh SampleModels::getCornellBox(bool box)
{
    Mesh mesh;
    const uint32_t primCount = box ? kCornellPrimCount : 12; // without the boxes: 5 walls + light
    const uint32_t matCount = box ? kCornellMaterialCount : 6;
    mesh.create(primCount, kCornellVertexCount, matCount, false);
    for (uint32_t i = 0; i < kCornellVertexCount; i++)
        mesh.getPositionBuffer()[i] = Vector3f(fromBits(kCornellPositionBits[3 * i]), fromBits(kCornellPositionBits[3 * i + 1]),
                                               fromBits(kCornellPositionBits[3 * i + 2]));
    auto copyPrim = [&](uint32_t dst, uint32_t src, uint32_t mat) {
        for (uint32_t j = 0; j < 3; j++) mesh.getIndexBuffer()[3 * dst + j] = kCornellIndices[3 * src + j];
        mesh.getPrimMateialBuffer()[dst] = mat;
    };
    auto copyMat = [&](uint32_t dst, uint32_t src) {
        Material& m = mesh.getMaterialBuffer()[dst];
        m.init();
        m.diffuse = Vector3f(fromBits(kCornellMaterials[src][0]), fromBits(kCornellMaterials[src][1]), fromBits(kCornellMaterials[src][2]));
        m.emissive = Vector3f(fromBits(kCornellMaterials[src][3]), fromBits(kCornellMaterials[src][4]), fromBits(kCornellMaterials[src][5]));
        m.reflectionType = (ReflectionType)kCornellMaterials[src][6];
    };
    if (box) {
        for (uint32_t p = 0; p < primCount; p++) copyPrim(p, p, kCornellPrimMaterial[p]);
        for (uint32_t m = 0; m < matCount; m++) copyMat(m, m);
    } else {
        for (uint32_t p = 0; p < 10; p++) copyPrim(p, p, kCornellPrimMaterial[p]);
        copyPrim(10, kCornellPrimCount - 2, 5);
        copyPrim(11, kCornellPrimCount - 1, 5);
        for (uint32_t m = 0; m < 5; m++) copyMat(m, m);
        copyMat(5, kCornellMaterialCount - 1);
    }
    mesh.calculateBounds();
    return mesh;
}

// ---------------------------------------------------------------- OBJ
namespace
{
struct ObjData {
    std::vector<Vector3f> positions;
    std::vector<Vector2f> texcoords; // per VERTEX, (u, 1-v), last writer wins (mesh.cpp:272-286)
    std::vector<uint32_t> indices;
    std::vector<int32_t> primMaterial;
    std::vector<std::string> materialNames;
    std::string mtllib;
};

bool parseObj(const char* path, ObjData& o)
{
    std::ifstream f(path);
    if (!f.is_open()) return false;
    std::vector<Vector2f> vt;
    std::map<std::string, int32_t> matIndex;
    int32_t curMat = -1;
    std::string line;
    struct Corner { int32_t v, t; };
    std::vector<Corner> corners;
    while (std::getline(f, line)) {
        const char* s = line.c_str();
        while (*s == ' ' || *s == '\t') s++;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            char* e;
            double x = strtod(s + 2, &e), y = strtod(e, &e), z = strtod(e, &e);
            o.positions.push_back(Vector3f((float)x, (float)y, (float)z));
        } else if (s[0] == 'v' && s[1] == 't') {
            char* e;
            double u = strtod(s + 2, &e), v = strtod(e, &e);
            vt.push_back(Vector2f((float)u, (float)v));
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            corners.clear();
            const char* p = s + 1;
            for (;;) {
                while (*p == ' ' || *p == '\t') p++;
                if (!*p || *p == '\r' || *p == '\n') break;
                char* e;
                long v = strtol(p, &e, 10);
                if (e == p) break;
                long t = 0;
                bool hasT = false;
                p = e;
                if (*p == '/') {
                    p++;
                    if (*p != '/') {
                        t = strtol(p, &e, 10);
                        hasT = e != p;
                        p = e;
                    }
                    if (*p == '/') {
                        p++;
                        (void)strtol(p, &e, 10);
                        p = e;
                    }
                }
                Corner c;
                c.v = (int32_t)(v > 0 ? v - 1 : (long)o.positions.size() + v);
                c.t = hasT ? (int32_t)(t > 0 ? t - 1 : (long)vt.size() + t) : -1;
                corners.push_back(c);
            }
            // fan triangulation 0-(k)-(k+1)
            for (size_t k = 1; k + 1 < corners.size(); k++) {
                const Corner tri[3] = {corners[0], corners[k], corners[k + 1]};
                for (const Corner& c : tri) {
                    if (o.texcoords.size() < o.positions.size()) o.texcoords.resize(o.positions.size(), Vector2f(0.0f));
                    if (c.v < 0 || (size_t)c.v >= o.positions.size()) return false;
                    o.texcoords[c.v] = (c.t >= 0 && (size_t)c.t < vt.size()) ? Vector2f(vt[c.t].x, 1.0f - vt[c.t].y) : Vector2f(0.0f);
                    o.indices.push_back((uint32_t)c.v);
                }
                o.primMaterial.push_back(curMat);
            }
        } else if (!strncmp(s, "usemtl", 6)) {
            std::istringstream is(s + 6);
            std::string name;
            is >> name;
            auto it = matIndex.find(name);
            if (it == matIndex.end()) {
                curMat = (int32_t)o.materialNames.size();
                matIndex[name] = curMat;
                o.materialNames.push_back(name);
            } else {
                curMat = it->second;
            }
        } else if (!strncmp(s, "mtllib", 6)) {
            std::istringstream is(s + 6);
            is >> o.mtllib;
        }
    }
    o.texcoords.resize(o.positions.size(), Vector2f(0.0f));
    return true;
}

// Decoders of the two 8-bit image families this build reads without stb: rows top to bottom, `depth` interleaved components.
// Binary PPM (P6), PGM (P5) and PAM (P7) ...
static bool decodePnm(FILE* f, int& w, int& h, int& depth, std::vector<uint8_t>& raw)
{
    char magic[3] = {0, 0, 0};
    int maxv = 255;
    w = h = depth = 0;
    if (fscanf(f, "%2s", magic) != 1) return false;
    if (!strcmp(magic, "P7")) {
        char key[64];
        while (fscanf(f, "%63s", key) == 1) {
            if (!strcmp(key, "WIDTH")) { if (fscanf(f, "%d", &w) != 1) break; }
            else if (!strcmp(key, "HEIGHT")) { if (fscanf(f, "%d", &h) != 1) break; }
            else if (!strcmp(key, "DEPTH")) { if (fscanf(f, "%d", &depth) != 1) break; }
            else if (!strcmp(key, "MAXVAL")) { if (fscanf(f, "%d", &maxv) != 1) break; }
            else if (!strcmp(key, "TUPLTYPE")) { if (fscanf(f, "%63s", key) != 1) break; }
            else if (!strcmp(key, "ENDHDR")) break;
        }
    } else if (!strcmp(magic, "P6") || !strcmp(magic, "P5")) {
        depth = magic[1] == '6' ? 3 : 1;
        if (fscanf(f, "%d %d %d", &w, &h, &maxv) != 3) return false;
    } else {
        return false;
    }
    fgetc(f);
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535 || maxv != 255 || depth < 1 || depth > 4) return false;
    raw.resize((size_t)w * h * depth);
    return fread(raw.data(), 1, raw.size(), f) == raw.size();
}

// ... and Truevision TGA: true-colour (type 2) and grey (type 3), raw or run-length encoded (10, 11), 8 / 24 / 32 bits per
// pixel, either row order, either column order.  Colour-mapped and 15/16-bit files are not read.  Stored BGR(A) -> RGB(A).
static bool decodeTga(FILE* f, int& w, int& h, int& depth, std::vector<uint8_t>& raw)
{
    uint8_t hd[18];
    if (fread(hd, 1, 18, f) != 18) return false;
    const int idLength = hd[0], colorMapType = hd[1], type = hd[2], bpp = hd[16], desc = hd[17];
    w = hd[12] | (hd[13] << 8);
    h = hd[14] | (hd[15] << 8);
    const bool rle = (type == 10 || type == 11), grey = (type == 3 || type == 11);
    if (colorMapType != 0 || !(type == 2 || type == 3 || rle) || w <= 0 || h <= 0) return false;
    if (!((grey && bpp == 8) || (!grey && (bpp == 24 || bpp == 32)))) return false;
    depth = bpp / 8;
    if (fseek(f, idLength, SEEK_CUR) != 0) return false;
    const size_t n = (size_t)w * h;
    std::vector<uint8_t> px(n * depth);
    if (!rle) {
        if (fread(px.data(), 1, px.size(), f) != px.size()) return false;
    } else {
        size_t i = 0;
        while (i < n) {
            int c = fgetc(f);
            if (c == EOF) return false;
            size_t count = (size_t)(c & 0x7f) + 1;
            if (i + count > n) return false;
            if (c & 0x80) { // run: one pixel repeated
                uint8_t v[4];
                if (fread(v, 1, depth, f) != (size_t)depth) return false;
                for (size_t k = 0; k < count; k++) memcpy(&px[(i + k) * depth], v, depth);
            } else if (fread(&px[i * depth], 1, count * depth, f) != count * depth) {
                return false;
            }
            i += count;
        }
    }
    const bool topDown = (desc & 0x20) != 0, rightLeft = (desc & 0x10) != 0;
    raw.resize(n * depth);
    for (int y = 0; y < h; y++) {
        const int sy = topDown ? y : h - 1 - y;
        for (int x = 0; x < w; x++) {
            const int sx = rightLeft ? w - 1 - x : x;
            const uint8_t* s = &px[((size_t)sy * w + sx) * depth];
            uint8_t* d = &raw[((size_t)y * w + x) * depth];
            if (depth == 1) d[0] = s[0];
            else {
                d[0] = s[2]; d[1] = s[1]; d[2] = s[0];
                if (depth == 4) d[3] = s[3];
            }
        }
    }
    return true;
}

// ... and PNG (see decodePng).  A plain inflate (RFC 1951: stored, fixed and dynamic Huffman blocks) and the five scanline
// filters (PNG 1.2 section 6).
namespace {
struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t buf = 0;
    int cnt = 0;
    bool fail = false;
    int bits(int k)
    {
        while (cnt < k) {
            if (pos >= n) { fail = true; return 0; }
            buf |= (uint32_t)p[pos++] << cnt;
            cnt += 8;
        }
        int v = (int)(buf & ((1u << k) - 1u));
        buf >>= k;
        cnt -= k;
        return v;
    }
};
struct Huffman {
    uint16_t count[16], symbol[288];
    void build(const uint8_t* len, int n)
    {
        memset(count, 0, sizeof(count));
        for (int i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int i = 1; i < 15; i++) offs[i + 1] = offs[i] + count[i];
        for (int i = 0; i < n; i++)
            if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    }
    int decode(BitReader& br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; len++) {
            code |= br.bits(1);
            if (br.fail) return -1;
            int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};
bool inflateZlib(const std::vector<uint8_t>& in, std::vector<uint8_t>& out)
{
    if (in.size() < 6) return false;
    BitReader br{in.data() + 2, in.size() - 2}; // skip the zlib header (CMF, FLG); the Adler-32 trailer is not checked
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
        int last = br.bits(1), type = br.bits(2);
        if (br.fail) return false;
        if (type == 0) {
            br.buf = 0; br.cnt = 0;
            if (br.pos + 4 > br.n) return false;
            size_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8);
            br.pos += 4;
            if (br.pos + len > br.n) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lens[320];
            if (type == 1) {
                for (int i = 0; i < 288; i++) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                lit.build(lens, 288);
                for (int i = 0; i < 30; i++) lens[i] = 5;
                dist.build(lens, 30);
            } else {
                int nlen = br.bits(5) + 257, ndist = br.bits(5) + 1, ncode = br.bits(4) + 4;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)br.bits(3);
                Huffman clh;
                clh.build(cl, 19);
                int i = 0;
                while (i < nlen + ndist) {
                    int sym = clh.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) lens[i++] = (uint8_t)sym;
                    else {
                        int rep = sym == 16 ? 3 + br.bits(2) : sym == 17 ? 3 + br.bits(3) : 11 + br.bits(7);
                        uint8_t v = (sym == 16 && i > 0) ? lens[i - 1] : 0;
                        if (sym == 16 && i == 0) return false;
                        if (i + rep > nlen + ndist) return false;
                        while (rep--) lens[i++] = v;
                    }
                }
                lit.build(lens, nlen);
                dist.build(lens + nlen, ndist);
            }
            for (;;) {
                int sym = lit.decode(br);
                if (sym < 0 || br.fail) return false;
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) return false;
                    int len = lbase[sym] + br.bits(lext[sym]);
                    int ds = dist.decode(br);
                    if (ds < 0 || ds >= 30) return false;
                    size_t d = dbase[ds] + (size_t)br.bits(dext[ds]);
                    if (d > out.size()) return false;
                    for (int k = 0; k < len; k++) out.push_back(out[out.size() - d]);
                }
            }
        } else {
            return false;
        }
        if (last) return !br.fail;
    }
}
} // namespace

// PNG as stb_image hands it to Texture::load (texture.cpp:218-249, 8 bits per channel requested): every colour type, bit depths
// 1 / 2 / 4 / 8 / 16, Adam7 interlace, palette and colour-key transparency (tRNS).  Samples below 8 bits are scaled to 0..255
// (x 255, x 85, x 17), 16-bit samples keep their high byte, a colour key adds an alpha channel that is 0 where the pixel
// equals the key -- the conversions stb_image applies (stbi__depth_scale_table, stbi__convert_16_to_8, stbi__compute_transparency).
static bool decodePng(FILE* f, int& w, int& h, int& depth, std::vector<uint8_t>& raw)
{
    std::vector<uint8_t> file;
    uint8_t tmp[65536];
    for (size_t n; (n = fread(tmp, 1, sizeof(tmp), f)) > 0;) file.insert(file.end(), tmp, tmp + n);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || memcmp(file.data(), sig, 8)) return false;
    auto be32 = [&](size_t o) { return ((uint32_t)file[o] << 24) | (file[o + 1] << 16) | (file[o + 2] << 8) | file[o + 3]; };
    int colorType = -1, bits = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    for (size_t o = 8; o + 12 <= file.size();) {
        uint32_t len = be32(o);
        if (o + 12 + (size_t)len > file.size()) return false;
        const uint8_t* d = &file[o + 8];
        if (!memcmp(&file[o + 4], "IHDR", 4)) {
            if (len < 13) return false;
            w = (int)be32(o + 8);
            h = (int)be32(o + 12);
            bits = d[8];
            colorType = d[9];
            interlace = d[12];
            if (d[10] != 0 || d[11] != 0 || interlace > 1) return false;
        } else if (!memcmp(&file[o + 4], "PLTE", 4)) palette.assign(d, d + len);
        else if (!memcmp(&file[o + 4], "tRNS", 4)) trns.assign(d, d + len);
        else if (!memcmp(&file[o + 4], "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!memcmp(&file[o + 4], "IEND", 4)) break;
        o += 12 + (size_t)len;
    }
    const int channels = colorType == 0 ? 1 : colorType == 2 ? 3 : colorType == 3 ? 1 : colorType == 4 ? 2 : colorType == 6 ? 4 : 0;
    if (!channels || w <= 0 || h <= 0 || w > 65535 || h > 65535) return false;
    const bool okDepth = (colorType == 0 && (bits == 1 || bits == 2 || bits == 4 || bits == 8 || bits == 16)) ||
                         (colorType == 3 && (bits == 1 || bits == 2 || bits == 4 || bits == 8)) ||
                         ((colorType == 2 || colorType == 4 || colorType == 6) && (bits == 8 || bits == 16));
    if (!okDepth) return false;
    std::vector<uint8_t> data;
    if (!inflateZlib(idat, data)) return false;
    // samples of the whole image, 16 bits each (as stored: not yet scaled)
    std::vector<uint16_t> img((size_t)w * h * channels);
    const size_t bpp = std::max<size_t>(1, (size_t)channels * bits / 8); // the filters' "corresponding byte" distance
    static const int x0s[8] = {0, 0, 4, 0, 2, 0, 1, 0}, y0s[8] = {0, 0, 0, 4, 0, 2, 0, 1}, dxs[8] = {1, 8, 8, 4, 4, 2, 2, 1}, dys[8] = {1, 8, 8, 8, 4, 4, 2, 2};
    size_t at = 0;
    std::vector<uint8_t> prev, cur;
    for (int pass = interlace ? 1 : 0; pass <= (interlace ? 7 : 0); pass++) {
        const int pw = (w - x0s[pass] + dxs[pass] - 1) / dxs[pass], ph = (h - y0s[pass] + dys[pass] - 1) / dys[pass];
        if (pw <= 0 || ph <= 0) continue;
        const size_t stride = ((size_t)pw * channels * bits + 7) / 8;
        prev.assign(stride, 0);
        cur.resize(stride);
        for (int y = 0; y < ph; y++) {
            if (at + 1 + stride > data.size()) return false;
            const uint8_t* src = &data[at];
            at += 1 + stride;
            const int ft = src[0];
            if (ft > 4) return false;
            for (size_t i = 0; i < stride; i++) {
                int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, pr = 0;
                if (ft == 1) pr = a;
                else if (ft == 2) pr = b;
                else if (ft == 3) pr = (a + b) >> 1;
                else if (ft == 4) {
                    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                    pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                }
                cur[i] = (uint8_t)(src[1 + i] + pr);
            }
            const int oy = y0s[pass] + y * dys[pass];
            for (int x = 0; x < pw; x++) {
                const int ox = x0s[pass] + x * dxs[pass];
                for (int ch = 0; ch < channels; ch++) {
                    const size_t k = (size_t)x * channels + ch;
                    uint16_t v;
                    if (bits == 16) v = (uint16_t)((cur[2 * k] << 8) | cur[2 * k + 1]);
                    else if (bits == 8) v = cur[k];
                    else v = (uint16_t)((cur[k * bits / 8] >> (8 - bits - (k * bits) % 8)) & ((1 << bits) - 1));
                    img[((size_t)oy * w + ox) * channels + ch] = v;
                }
            }
            prev.swap(cur);
        }
    }
    const size_t px = (size_t)w * h;
    if (colorType == 3) { // palette -> RGB(A)
        if (palette.size() < 3) return false;
        depth = trns.empty() ? 3 : 4;
        raw.resize(px * depth);
        for (size_t i = 0; i < px; i++) {
            size_t k = img[i];
            if (3 * k + 2 >= palette.size()) return false;
            raw[depth * i + 0] = palette[3 * k]; raw[depth * i + 1] = palette[3 * k + 1]; raw[depth * i + 2] = palette[3 * k + 2];
            if (depth == 4) raw[depth * i + 3] = k < trns.size() ? trns[k] : 255;
        }
        return true;
    }
    // colour key (tRNS with colour types 0 and 2): one 16-bit value per channel, compared with the samples as stored
    const bool keyed = (colorType == 0 && trns.size() >= 2) || (colorType == 2 && trns.size() >= 6);
    uint16_t key[3] = {0, 0, 0};
    if (keyed)
        for (int ch = 0; ch < channels; ch++) {
            key[ch] = (uint16_t)((trns[2 * ch] << 8) | trns[2 * ch + 1]);
            if (bits < 16) key[ch] &= 255; // stb_image keeps the low byte of the key for 8-bit data
        }
    static const int scale[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};
    depth = channels + (keyed ? 1 : 0);
    raw.resize(px * depth);
    for (size_t i = 0; i < px; i++) {
        bool isKey = keyed;
        for (int ch = 0; ch < channels; ch++) {
            const uint16_t v = img[i * channels + ch];
            if (keyed && v != key[ch]) isKey = false;
            raw[i * depth + ch] = bits == 16 ? (uint8_t)(v >> 8) : (uint8_t)(v * scale[bits]);
        }
        if (keyed) raw[i * depth + channels] = isKey ? 0 : 255;
    }
    return true;
}

// JPEG (baseline and extended sequential DCT, Huffman coded, 8 bits; 1 or 3 components, any sampling factors, restart
// intervals).  stb_image, which the reference decodes its textures with (texture.cpp:218-249), is not vendored; what follows
// restates ITS published integer pipeline, because the texel values are part of the image: the 12-bit fixed-point
// separable IDCT (constants x 4096, +512 >> 10 after the columns, +65536 + (128 << 17) >> 17 after the rows), chroma
// upsampling by the (3 near + 1 far + 2) >> 2 and (9, 3, 3, 1) / 16 triangle filters, and YCbCr -> RGB in 20-bit fixed point
// (1.40200, 0.71414, 0.34414 with its & 0xffff0000, 1.77200).  No stb_image is at hand to pin this against (DESIGN.md
// "parity status"); sequential and progressive Huffman-coded files; arithmetic-coded, lossless and hierarchical ones are refused.
namespace {
struct JpegHuff {
    uint8_t size[257];
    uint16_t code[257];
    uint8_t values[256];
    int maxcode[18], delta[17];
    bool build(const int* counts)
    {
        int k = 0;
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < counts[i]; j++) {
                if (k >= 256) return false;
                size[k++] = (uint8_t)(i + 1);
            }
        size[k] = 0;
        int c = 0;
        k = 0;
        for (int j = 1; j <= 16; j++) {
            delta[j] = k - c;
            if (size[k] == j) {
                while (size[k] == j) code[k++] = (uint16_t)(c++);
                if (c - 1 >= (1 << j)) return false;
            }
            maxcode[j] = c << (16 - j);
            c <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        return true;
    }
};
struct JpegComp {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dcPred = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0;
    std::vector<uint8_t> data;
    std::vector<short> coeff; // progressive files: the coefficients of every block (w2 / 8 blocks per row), refined scan by scan
};
struct JpegDec {
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t codeBuffer = 0;
    int codeBits = 0;
    uint8_t marker = 0xff; // 0xff = none pending
    bool nomore = false;
    int get8() { return pos < n ? p[pos++] : 0; }
    void grow()
    {
        do {
            unsigned b = nomore ? 0 : (unsigned)get8();
            if (b == 0xff) {
                int c = get8();
                while (c == 0xff) c = get8(); // fill bytes
                if (c != 0) {
                    marker = (uint8_t)c;
                    nomore = true;
                    return;
                }
            }
            codeBuffer |= b << (24 - codeBits);
            codeBits += 8;
        } while (codeBits <= 24);
    }
    int decode(const JpegHuff& h)
    {
        if (codeBits < 16) grow();
        const uint32_t temp = codeBuffer >> 16;
        int k;
        for (k = 1; k <= 16; k++)
            if (temp < (uint32_t)h.maxcode[k]) break;
        if (k == 17 || k > codeBits) return -1;
        const int c = (int)((codeBuffer >> (32 - k)) & ((1u << k) - 1)) + h.delta[k];
        if (c < 0 || c >= 256) return -1;
        codeBits -= k;
        codeBuffer <<= k;
        return h.values[c];
    }
    int bit()
    {
        if (codeBits < 1) grow();
        if (codeBits < 1) return 0;
        const uint32_t k = codeBuffer;
        codeBuffer <<= 1;
        --codeBits;
        return (int)(k >> 31);
    }
    int bits(int n)
    {
        if (n == 0) return 0;
        if (codeBits < n) grow();
        if (codeBits < n) return 0;
        const uint32_t k = codeBuffer >> (32 - n);
        codeBuffer <<= n;
        codeBits -= n;
        return (int)k;
    }
    int extend(int nbits) // receive + extend (ITU T.81 F.2.2.1)
    {
        if (nbits == 0) return 0;
        if (codeBits < nbits) grow();
        if (codeBits < nbits) return 0;
        const int sgn = (int)(codeBuffer >> 31);
        const uint32_t k = (codeBuffer >> (32 - nbits));
        codeBuffer <<= nbits;
        codeBits -= nbits;
        return (int)k + (sgn ? 0 : (int)((~0u << nbits) + 1));
    }
};
const uint8_t kJpegZigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                      6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                      39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};
inline uint8_t jclamp(int x) { return (unsigned)x > 255u ? (x < 0 ? 0 : 255) : (uint8_t)x; }
#define J_F2F(x) ((int)(((x) * 4096 + 0.5)))
#define J_FSH(x) ((x) * 4096)
#define J_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                               \
    int t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                     \
    p2 = s2; p3 = s6;                                                                           \
    p1 = (p2 + p3) * J_F2F(0.5411961f);                                                         \
    t2 = p1 + p3 * J_F2F(-1.847759065f);                                                        \
    t3 = p1 + p2 * J_F2F(0.765366865f);                                                         \
    p2 = s0; p3 = s4;                                                                           \
    t0 = J_FSH(p2 + p3); t1 = J_FSH(p2 - p3);                                                   \
    x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                                     \
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                                         \
    p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;                                     \
    p5 = (p3 + p4) * J_F2F(1.175875602f);                                                       \
    t0 = t0 * J_F2F(0.298631336f); t1 = t1 * J_F2F(2.053119869f);                               \
    t2 = t2 * J_F2F(3.072711026f); t3 = t3 * J_F2F(1.501321110f);                               \
    p1 = p5 + p1 * J_F2F(-0.899976223f); p2 = p5 + p2 * J_F2F(-2.562915447f);                   \
    p3 = p3 * J_F2F(-1.961570560f); p4 = p4 * J_F2F(-0.390180644f);                             \
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
void jpegIdct(uint8_t* out, int stride, const short* data)
{
    int val[64], *v = val;
    const short* d = data;
    for (int i = 0; i < 8; ++i, ++d, ++v) {
        if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
            int dcterm = d[0] * 4;
            v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
        } else {
            J_IDCT_1D(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56])
            x0 += 512; x1 += 512; x2 += 512; x3 += 512;
            v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
            v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
            v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
        }
    }
    v = val;
    uint8_t* o = out;
    for (int i = 0; i < 8; ++i, v += 8, o += stride) {
        J_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17); x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);
        o[0] = jclamp((x0 + t3) >> 17); o[7] = jclamp((x0 - t3) >> 17);
        o[1] = jclamp((x1 + t2) >> 17); o[6] = jclamp((x1 - t2) >> 17);
        o[2] = jclamp((x2 + t1) >> 17); o[5] = jclamp((x2 - t1) >> 17);
        o[3] = jclamp((x3 + t0) >> 17); o[4] = jclamp((x3 - t0) >> 17);
    }
}
// the four row resamplers: out has w * hs samples
const uint8_t* jrow1(uint8_t*, const uint8_t* near, const uint8_t*, int, int) { return near; }
const uint8_t* jrowV2(uint8_t* out, const uint8_t* near, const uint8_t* far, int w, int)
{
    for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
    return out;
}
const uint8_t* jrowH2(uint8_t* out, const uint8_t* in, const uint8_t*, int w, int)
{
    if (w == 1) {
        out[0] = out[1] = in[0];
        return out;
    }
    out[0] = in[0];
    out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    int i;
    for (i = 1; i < w - 1; ++i) {
        int n = 3 * in[i] + 2;
        out[i * 2 + 0] = (uint8_t)((n + in[i - 1]) >> 2);
        out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
    }
    out[i * 2 + 0] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
    out[i * 2 + 1] = in[w - 1];
    return out;
}
const uint8_t* jrowHV2(uint8_t* out, const uint8_t* near, const uint8_t* far, int w, int)
{
    if (w == 1) {
        out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2);
        return out;
    }
    int t1 = 3 * near[0] + far[0], t0;
    out[0] = (uint8_t)((t1 + 2) >> 2);
    for (int i = 1; i < w; ++i) {
        t0 = t1;
        t1 = 3 * near[i] + far[i];
        out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
        out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
    }
    out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
    return out;
}
const uint8_t* jrowGeneric(uint8_t* out, const uint8_t* near, const uint8_t*, int w, int hs)
{
    for (int i = 0; i < w; ++i)
        for (int j = 0; j < hs; ++j) out[i * hs + j] = near[i];
    return out;
}
} // namespace

static bool decodeJpeg(FILE* f, int& w, int& h, int& depth, std::vector<uint8_t>& raw)
{
    std::vector<uint8_t> file;
    uint8_t tmp[65536];
    for (size_t n; (n = fread(tmp, 1, sizeof(tmp), f)) > 0;) file.insert(file.end(), tmp, tmp + n);
    if (file.size() < 4 || file[0] != 0xff || file[1] != 0xd8) return false;
    JpegDec d{file.data(), file.size()};
    d.pos = 2;
    uint16_t dequant[4][64];
    JpegHuff hdc[4], hac[4];
    bool haveDc[4] = {false, false, false, false}, haveAc[4] = {false, false, false, false};
    JpegComp comp[3];
    int ncomp = 0, hmax = 1, vmax = 1, restart = 0;
    bool sawFrame = false, progressive = false;
    auto be16 = [&]() { int a = d.get8(); return (a << 8) | d.get8(); };
    for (;;) {
        int m = d.get8();
        if (m != 0xff) return false;
        while (m == 0xff) m = d.get8();
        if (m == 0xd9) return false; // EOI before any scan
        if (m == 0xdb) { // DQT
            int L = be16() - 2;
            while (L > 0) {
                int q = d.get8(), prec = q >> 4, t = q & 15;
                if (t > 3 || prec > 1) return false;
                for (int i = 0; i < 64; i++) dequant[t][kJpegZigzag[i]] = (uint16_t)(prec ? be16() : d.get8());
                L -= prec ? 129 : 65;
            }
            if (L != 0) return false;
        } else if (m == 0xc4) { // DHT
            int L = be16() - 2;
            while (L > 0) {
                int q = d.get8(), tc = q >> 4, th = q & 15, counts[16], total = 0;
                if (tc > 1 || th > 3) return false;
                for (int i = 0; i < 16; i++) total += counts[i] = d.get8();
                if (total > 256) return false;
                JpegHuff& hf = tc ? hac[th] : hdc[th];
                if (!hf.build(counts)) return false;
                for (int i = 0; i < total; i++) hf.values[i] = (uint8_t)d.get8();
                (tc ? haveAc : haveDc)[th] = true;
                L -= 17 + total;
            }
            if (L != 0) return false;
        } else if (m == 0xdd) { // DRI
            if (be16() != 4) return false;
            restart = be16();
        } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) { // SOF0 / SOF1 (sequential), SOF2 (progressive)
            progressive = (m == 0xc2);
            int L = be16();
            if (d.get8() != 8) return false;
            h = be16();
            w = be16();
            ncomp = d.get8();
            if ((ncomp != 1 && ncomp != 3) || L != 8 + 3 * ncomp || w <= 0 || h <= 0) return false;
            for (int i = 0; i < ncomp; i++) {
                comp[i].id = d.get8();
                int q = d.get8();
                comp[i].h = q >> 4;
                comp[i].v = q & 15;
                comp[i].tq = d.get8();
                if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) return false;
                hmax = std::max(hmax, comp[i].h);
                vmax = std::max(vmax, comp[i].v);
            }
            for (int i = 0; i < ncomp; i++)
                if (hmax % comp[i].h || vmax % comp[i].v) return false;
            const int mcuw = hmax * 8, mcuh = vmax * 8, mcux = (w + mcuw - 1) / mcuw, mcuy = (h + mcuh - 1) / mcuh;
            for (int i = 0; i < ncomp; i++) {
                comp[i].x = (w * comp[i].h + hmax - 1) / hmax;
                comp[i].y = (h * comp[i].v + vmax - 1) / vmax;
                comp[i].w2 = mcux * comp[i].h * 8;
                comp[i].h2 = mcuy * comp[i].v * 8;
                comp[i].data.assign((size_t)comp[i].w2 * comp[i].h2, 0);
                if (progressive) comp[i].coeff.assign((size_t)comp[i].w2 * comp[i].h2, 0);
            }
            sawFrame = true;
        } else if (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
            return false; // lossless, hierarchical, arithmetic: not decoded here
        } else if (m == 0xda) { // SOS: the one interleaved scan of a baseline file (or one scan per component)
            if (!sawFrame) return false;
            int L = be16(), ns = d.get8();
            if (ns < 1 || ns > ncomp || L != 6 + 2 * ns) return false;
            int order[3];
            for (int i = 0; i < ns; i++) {
                int id = d.get8(), q = d.get8(), which = -1;
                for (int k = 0; k < ncomp; k++)
                    if (comp[k].id == id) which = k;
                if (which < 0) return false;
                comp[which].hd = q >> 4;
                comp[which].ha = q & 15;
                if (comp[which].hd > 3 || comp[which].ha > 3) return false;
                order[i] = which;
            }
            const int specStart = d.get8(), specEnd = d.get8(), approx = d.get8(), succHigh = approx >> 4, succLow = approx & 15;
            if (progressive) {
                if (specStart > 63 || specEnd > 63 || specStart > specEnd || succHigh > 13 || succLow > 13) return false;
                if (specStart != 0 && ns != 1) return false; // AC scans carry one component
                for (int i = 0; i < ns; i++)
                    if (specStart == 0 ? (succHigh == 0 && !haveDc[comp[order[i]].hd]) : !haveAc[comp[order[i]].ha]) return false;
            } else {
                for (int i = 0; i < ns; i++)
                    if (!haveDc[comp[order[i]].hd] || !haveAc[comp[order[i]].ha]) return false;
            }
            d.codeBuffer = 0; d.codeBits = 0; d.marker = 0xff; d.nomore = false;
            for (int i = 0; i < ncomp; i++) comp[i].dcPred = 0;
            int todo = restart ? restart : 0x7fffffff;
            int eobRun = 0;
            // one block of a progressive scan (ITU T.81 G.1.2, as stb_image decodes it: DC first / refinement, AC first / refinement)
            auto blockProgressive = [&](JpegComp& c, short* data) -> bool {
                if (specStart == 0) {
                    if (specEnd != 0) return false; // a DC scan carries DC only
                    if (succHigh == 0) {
                        const int t = d.decode(hdc[c.hd]);
                        if (t < 0 || t > 15) return false;
                        const int diff = t ? d.extend(t) : 0;
                        c.dcPred += diff;
                        data[0] = (short)(c.dcPred * (1 << succLow));
                    } else if (d.bit()) {
                        data[0] = (short)(data[0] + (1 << succLow));
                    }
                    return true;
                }
                if (succHigh == 0) {
                    if (eobRun) {
                        --eobRun;
                        return true;
                    }
                    int k = specStart;
                    do {
                        const int rs = d.decode(hac[c.ha]);
                        if (rs < 0) return false;
                        const int sz = rs & 15, r = rs >> 4;
                        if (sz == 0) {
                            if (r < 15) {
                                eobRun = 1 << r;
                                if (r) eobRun += d.bits(r);
                                --eobRun;
                                break;
                            }
                            k += 16;
                        } else {
                            k += r;
                            const int zig = kJpegZigzag[k++];
                            data[zig] = (short)(d.extend(sz) * (1 << succLow));
                        }
                    } while (k <= specEnd);
                    return true;
                }
                const short bit = (short)(1 << succLow);
                auto refine = [&](short* p) {
                    if (d.bit() && (*p & bit) == 0) *p = (short)(*p > 0 ? *p + bit : *p - bit);
                };
                if (eobRun) {
                    --eobRun;
                    for (int k = specStart; k <= specEnd; ++k) {
                        short* p = &data[kJpegZigzag[k]];
                        if (*p != 0) refine(p);
                    }
                    return true;
                }
                int k = specStart;
                do {
                    const int rs = d.decode(hac[c.ha]);
                    if (rs < 0) return false;
                    int sz = rs & 15, r = rs >> 4;
                    if (sz == 0) {
                        if (r < 15) {
                            eobRun = (1 << r) - 1;
                            if (r) eobRun += d.bits(r);
                            r = 64; // to the end of the band
                        }
                    } else {
                        if (sz != 1) return false;
                        sz = d.bit() ? bit : -bit;
                    }
                    while (k <= specEnd) {
                        short* p = &data[kJpegZigzag[k++]];
                        if (*p != 0) {
                            refine(p);
                        } else {
                            if (r == 0) {
                                *p = (short)sz;
                                break;
                            }
                            --r;
                        }
                    }
                } while (k <= specEnd);
                return true;
            };
            auto block = [&](JpegComp& c, uint8_t* out) -> bool {
                if (progressive) { // `out` addresses the block's top-left sample: its coefficients live at the same block position
                    const size_t off = (size_t)(out - c.data.data());
                    const size_t by = off / ((size_t)c.w2 * 8), bx = (off % (size_t)c.w2) / 8;
                    return blockProgressive(c, &c.coeff[(by * (size_t)(c.w2 / 8) + bx) * 64]);
                }
                short data[64];
                memset(data, 0, sizeof(data));
                if (d.codeBits < 16) d.grow();
                int t = d.decode(hdc[c.hd]);
                if (t < 0 || t > 15) return false;
                int diff = t ? d.extend(t) : 0;
                int dc = c.dcPred + diff;
                c.dcPred = dc;
                data[0] = (short)(dc * dequant[c.tq][0]);
                int k = 1;
                do {
                    int rs = d.decode(hac[c.ha]);
                    if (rs < 0) return false;
                    int sz = rs & 15, r = rs >> 4;
                    if (sz == 0) {
                        if (rs != 0xf0) break; // end of block
                        k += 16;
                    } else {
                        k += r;
                        int zig = kJpegZigzag[k++];
                        data[zig] = (short)(d.extend(sz) * dequant[c.tq][zig]);
                    }
                } while (k < 64);
                jpegIdct(out, c.w2, data);
                return true;
            };
            auto restartCheck = [&]() -> bool {
                if (--todo > 0) return true;
                if (d.codeBits < 24) d.grow();
                if (!(d.marker >= 0xd0 && d.marker <= 0xd7)) return true; // no restart marker: the scan just goes on / ends
                d.codeBuffer = 0; d.codeBits = 0; d.marker = 0xff; d.nomore = false;
                for (int i = 0; i < ncomp; i++) comp[i].dcPred = 0;
                todo = restart ? restart : 0x7fffffff;
                eobRun = 0;
                return true;
            };
            if (ns == 1) { // non-interleaved: the component's own blocks, row by row
                JpegComp& c = comp[order[0]];
                const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
                for (int j = 0; j < bh; j++)
                    for (int i = 0; i < bw; i++) {
                        if (!block(c, &c.data[(size_t)c.w2 * j * 8 + i * 8])) return false;
                        if (!restartCheck()) return false;
                    }
            } else {
                const int mcux = (w + hmax * 8 - 1) / (hmax * 8), mcuy = (h + vmax * 8 - 1) / (vmax * 8);
                for (int j = 0; j < mcuy; j++)
                    for (int i = 0; i < mcux; i++) {
                        for (int k = 0; k < ns; k++) {
                            JpegComp& c = comp[order[k]];
                            for (int y = 0; y < c.v; y++)
                                for (int x = 0; x < c.h; x++)
                                    if (!block(c, &c.data[(size_t)c.w2 * ((j * c.v + y) * 8) + (i * c.h + x) * 8])) return false;
                        }
                        if (!restartCheck()) return false;
                    }
            }
            // what follows the entropy-coded data: EOI, or another scan of a non-interleaved file
            if (d.marker == 0xff) {
                while (d.pos < d.n) {
                    int x = d.get8();
                    if (x == 0xff) {
                        int y = d.get8();
                        while (y == 0xff) y = d.get8();
                        if (y != 0) { d.marker = (uint8_t)y; break; }
                    }
                }
            }
            if (d.marker == 0xd9 || d.marker == 0xff) break;
            d.pos -= 2; // step back onto the marker and parse on
            d.marker = 0xff;
        } else { // APPn, COM and the rest: skipped
            int L = be16();
            if (L < 2) return false;
            d.pos += (size_t)(L - 2);
        }
        if (d.pos >= d.n) return false;
    }
    if (progressive) // every scan is in: dequantise and transform (stb_image: stbi__jpeg_finish)
        for (int k = 0; k < ncomp; k++) {
            JpegComp& c = comp[k];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; j++)
                for (int i = 0; i < bw; i++) {
                    short* data = &c.coeff[((size_t)j * (c.w2 / 8) + i) * 64];
                    for (int q = 0; q < 64; q++) data[q] = (short)(data[q] * dequant[c.tq][q]);
                    jpegIdct(&c.data[(size_t)c.w2 * j * 8 + i * 8], c.w2, data);
                }
        }
    // ---- upsample and convert, one output row at a time
    depth = ncomp == 1 ? 1 : 3;
    raw.resize((size_t)w * h * depth);
    struct Res {
        const uint8_t* (*fn)(uint8_t*, const uint8_t*, const uint8_t*, int, int);
        const uint8_t *line0, *line1;
        int hs, vs, wLores, ystep, ypos;
        std::vector<uint8_t> buf;
    } res[3];
    for (int k = 0; k < ncomp; k++) {
        Res& r = res[k];
        r.hs = hmax / comp[k].h;
        r.vs = vmax / comp[k].v;
        r.ystep = r.vs >> 1;
        r.wLores = (w + r.hs - 1) / r.hs;
        r.ypos = 0;
        r.line0 = r.line1 = comp[k].data.data();
        r.buf.resize((size_t)w + 3 + r.hs * 2);
        r.fn = (r.hs == 1 && r.vs == 1) ? jrow1 : (r.hs == 1 && r.vs == 2) ? jrowV2 : (r.hs == 2 && r.vs == 1) ? jrowH2 : (r.hs == 2 && r.vs == 2) ? jrowHV2 : jrowGeneric;
    }
    for (int j = 0; j < h; j++) {
        const uint8_t* co[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < ncomp; k++) {
            Res& r = res[k];
            const bool bot = r.ystep >= (r.vs >> 1);
            co[k] = r.fn(r.buf.data(), bot ? r.line1 : r.line0, bot ? r.line0 : r.line1, r.wLores, r.hs);
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < comp[k].y) r.line1 += comp[k].w2;
            }
        }
        uint8_t* out = &raw[(size_t)j * w * depth];
        if (ncomp == 1) {
            memcpy(out, co[0], (size_t)w);
        } else {
#define J_FIX(x) (((int)((x) * 4096.0f + 0.5f)) << 8)
            for (int i = 0; i < w; i++) {
                int yf = (co[0][i] << 20) + (1 << 19), cr = co[2][i] - 128, cb = co[1][i] - 128;
                int r = yf + cr * J_FIX(1.40200f);
                int g = yf + (cr * -J_FIX(0.71414f)) + ((cb * -J_FIX(0.34414f)) & 0xffff0000);
                int b = yf + cb * J_FIX(1.77200f);
                out[3 * i + 0] = jclamp(r >> 20);
                out[3 * i + 1] = jclamp(g >> 20);
                out[3 * i + 2] = jclamp(b >> 20);
            }
        }
    }
    return true;
}

// Texture::load (texture.cpp:212-254) without stb_image: PNM, PNG, JPEG or TGA by content.
bool loadTexture(const std::string& path, Texture& tex, bool bump)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    int w = 0, h = 0, depth = 0;
    std::vector<uint8_t> raw;
    int c0 = fgetc(f);
    ungetc(c0, f);
    bool ok = (c0 == 'P') ? decodePnm(f, w, h, depth, raw) : (c0 == 0x89) ? decodePng(f, w, h, depth, raw) : (c0 == 0xff) ? decodeJpeg(f, w, h, depth, raw)
                                                                                                               : decodeTga(f, w, h, depth, raw);
    fclose(f);
    if (!ok || w > 65535 || h > 65535) return false;
    // texture.cpp:226-247: grey stays 1 component, everything else becomes RGBA; 3-component bump maps become
    // a 1-component height field (convertNormalToBump, texture.cpp:185-200)
    if (depth == 1) {
        tex.create((uint16_t)w, (uint16_t)h, 1, raw.data());
        return true;
    }
    std::vector<uint8_t> rgba((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        rgba[4 * i + 0] = raw[depth * i + 0];
        rgba[4 * i + 1] = depth >= 3 ? raw[depth * i + 1] : raw[depth * i];
        rgba[4 * i + 2] = depth >= 3 ? raw[depth * i + 2] : raw[depth * i];
        rgba[4 * i + 3] = depth == 4 ? raw[depth * i + 3] : (depth == 2 ? raw[depth * i + 1] : 255);
    }
    if (bump && depth == 3) {
        std::vector<uint8_t> b((size_t)w * h);
        const float kChannelScale = 1.0f / 255.0f;
        for (size_t i = 0; i < (size_t)w * h; i++) {
            float nx = 2.0f * (rgba[4 * i + 0] * kChannelScale - 0.5f);
            float ny = 2.0f * (rgba[4 * i + 1] * kChannelScale - 0.5f);
            float nz = 2.0f * (rgba[4 * i + 2] * kChannelScale - 0.5f);
            auto n = normalize(Vector3f(nx, ny, nz));
            float v = std::fmin(std::fmax(n.z * n.z * n.z * n.z, 0.0f), 1.0f);
            b[i] = (uint8_t)(0xff * v);
        }
        tex.create((uint16_t)w, (uint16_t)h, 1, b.data());
        return true;
    }
    tex.create((uint16_t)w, (uint16_t)h, 4, rgba.data());
    return true;
}

void parseMtl(const std::string& path, const std::string& dir, const std::vector<std::string>& names, std::vector<Material>& mats)
{
    std::ifstream f(path);
    if (!f.is_open()) return;
    std::map<std::string, size_t> idx;
    for (size_t i = 0; i < names.size(); i++) idx[names[i]] = i;
    Material* cur = nullptr;
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream is(line);
        std::string key;
        is >> key;
        if (key == "newmtl") {
            std::string n;
            is >> n;
            auto it = idx.find(n);
            cur = it == idx.end() ? nullptr : &mats[it->second];
        } else if (cur && key == "Kd") {
            is >> cur->diffuse.x >> cur->diffuse.y >> cur->diffuse.z;
        } else if (cur && key == "Ke") {
            is >> cur->emissive.x >> cur->emissive.y >> cur->emissive.z;
        } else if (cur && key == "Ka") {
            is >> cur->ambient.x >> cur->ambient.y >> cur->ambient.z;
        } else if (cur && key == "Ks") {
            is >> cur->specular.x >> cur->specular.y >> cur->specular.z;
        } else if (cur && (key == "map_Kd" || key == "map_bump" || key == "bump" || key == "map_Bump")) {
            std::string name, tok;
            while (is >> tok) name = tok; // last token is the file name (options precede it)
            for (auto& c : name)
                if (c == '\\') c = '/';
            bool bump = key != "map_Kd";
            Texture& t = bump ? cur->bumpMap : cur->diffuseMap;
            if (!loadTexture(dir + "/" + name, t, bump))
                logPrintf(LogLevel::kError, "texture '%s' not loaded (binary PPM/PGM/PAM, PNG, Huffman-coded JPEG and TGA are read here: stb is not vendored)\n", name.c_str());
        }
    }
    for (auto& m : mats) m.alphaTest = m.diffuseMap.isAlphaTestRequired(); // material.cpp:79
}
} // namespace

void Mesh::loadObj(const char* path, const Material& mat) // mesh.cpp:151-209
{
    ObjData o;
    if (!parseObj(path, o)) {
        logPrintf(LogLevel::kError, "Failed to load %s\n", path);
        return;
    }
    create((uint32_t)o.indices.size() / 3, (uint32_t)o.positions.size(), 1, false);
    m_indices = o.indices;
    m_positions = o.positions;
    // The reference copies only the first vertexCount*4 BYTES of the raw vt list here (mesh.cpp:200) and leaves the
    // rest of the buffer uninitialised; texcoords are never read for an untextured material, so the per-vertex
    // (u, 1-v) table of the other overload is used instead.
    m_texcoords = o.texcoords;
    std::fill(m_primMaterial.begin(), m_primMaterial.end(), 0u);
    m_materials[0] = mat;
    calculateBounds();
    m_hasTexcoord = true;
    logPrintf(LogLevel::kVerbose, "Finished loading '%s'\n", path);
}

void Mesh::loadObj(const char* path) // mesh.cpp:211-300
{
    ObjData o;
    if (!parseObj(path, o)) {
        logPrintf(LogLevel::kError, "Failed to load %s\n", path);
        return;
    }
    std::string p(path), dir = ".";
    size_t slash = p.find_last_of("/\\");
    if (slash != std::string::npos) dir = p.substr(0, slash);
    size_t matCount = std::max<size_t>(o.materialNames.size(), 1);
    create((uint32_t)o.indices.size() / 3, (uint32_t)o.positions.size(), (uint32_t)matCount, false);
    m_indices = o.indices;
    m_positions = o.positions;
    m_texcoords = o.texcoords;
    for (size_t i = 0; i < m_primMaterial.size(); i++) m_primMaterial[i] = o.primMaterial[i] < 0 ? 0u : (uint32_t)o.primMaterial[i];
    if (!o.mtllib.empty()) parseMtl(dir + "/" + o.mtllib, dir, o.materialNames, m_materials);
    calculateBounds();
    m_hasTexcoord = true;
    logPrintf(LogLevel::kVerbose, "Finished loading '%s' prim=%u, vtx=%u\n", path, getPrimCount(), getVertexCount());
}

// ---------------------------------------------------------------- procedural stand-ins
namespace
{
struct Lcg { // fixed generator so that the stand-ins are identical on every box
    uint64_t s;
    explicit Lcg(uint64_t seed) : s(seed * 6364136223846793005ull + 1442695040888963407ull) {}
    uint32_t next()
    {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return (uint32_t)(s >> 33);
    }
    float uniform() { return (float)(next() & 0xffffff) / 16777216.0f; }
    float range(float a, float b) { return a + (b - a) * uniform(); }
};
} // namespace

// Bunny-class stand-in: a lat-long sphere of ~targetTris triangles whose radius is displaced by a
// few octaves of seeded sinusoids.  2*segments*rings triangles.
Mesh SampleModels::getDisplacedSphere(uint32_t targetTris, float radius, const Vector3f& center, const Material& mat, uint32_t seed)
{
    uint32_t rings = std::max(3u, (uint32_t)std::lround(std::sqrt((double)targetTris / 4.0)));
    uint32_t segs = std::max(3u, (uint32_t)(targetTris / (2 * rings)));
    Lcg rng(seed);
    const int kWaves = 12;
    float amp[kWaves], fx[kWaves], fy[kWaves], fz[kWaves], ph[kWaves];
    for (int i = 0; i < kWaves; i++) {
        float octave = (float)(1 + i / 3);
        amp[i] = 0.12f / octave;
        fx[i] = rng.range(-3.0f, 3.0f) * octave;
        fy[i] = rng.range(-3.0f, 3.0f) * octave;
        fz[i] = rng.range(-3.0f, 3.0f) * octave;
        ph[i] = rng.range(0.0f, 6.2831853f);
    }
    auto displaced = [&](float theta, float phi) {
        Vector3f d(std::sin(theta) * std::cos(phi), std::cos(theta), std::sin(theta) * std::sin(phi));
        float r = 1.0f;
        for (int i = 0; i < kWaves; i++) r += amp[i] * std::sin(fx[i] * d.x + fy[i] * d.y + fz[i] * d.z + ph[i]);
        return center + (radius * r) * d;
    };
    const uint32_t vertexCount = 2 + (rings - 1) * segs;
    const uint32_t primCount = 2 * segs + 2 * segs * (rings - 2);
    Mesh mesh;
    mesh.create(primCount, vertexCount, 1, false);
    Vector3f* pos = mesh.getPositionBuffer();
    Vector2f* tex = mesh.getTexcoordBuffer();
    const float kPiF = 3.14159265358979323846f;
    pos[0] = displaced(0.0f, 0.0f);
    tex[0] = Vector2f(0.5f, 0.0f);
    for (uint32_t r = 1; r < rings; r++)
        for (uint32_t s = 0; s < segs; s++) {
            float theta = kPiF * (float)r / (float)rings, phi = 2.0f * kPiF * (float)s / (float)segs;
            pos[1 + (r - 1) * segs + s] = displaced(theta, phi);
            tex[1 + (r - 1) * segs + s] = Vector2f((float)s / (float)segs, (float)r / (float)rings);
        }
    pos[vertexCount - 1] = displaced(kPiF, 0.0f);
    tex[vertexCount - 1] = Vector2f(0.5f, 1.0f);
    uint32_t* idx = mesh.getIndexBuffer();
    uint32_t k = 0;
    auto ring = [&](uint32_t r, uint32_t s) { return 1 + (r - 1) * segs + (s % segs); };
    for (uint32_t s = 0; s < segs; s++) { idx[k++] = 0; idx[k++] = ring(1, s + 1); idx[k++] = ring(1, s); }
    for (uint32_t r = 1; r + 1 < rings; r++)
        for (uint32_t s = 0; s < segs; s++) {
            idx[k++] = ring(r, s); idx[k++] = ring(r, s + 1); idx[k++] = ring(r + 1, s);
            idx[k++] = ring(r, s + 1); idx[k++] = ring(r + 1, s + 1); idx[k++] = ring(r + 1, s);
        }
    for (uint32_t s = 0; s < segs; s++) { idx[k++] = vertexCount - 1; idx[k++] = ring(rings - 1, s); idx[k++] = ring(rings - 1, s + 1); }
    mesh.getMaterialBuffer()[0] = mat;
    mesh.setHasTexcoord(true);
    mesh.calculateBounds();
    return mesh;
}

// Sponza-class stand-in: a two-storey colonnaded hall ("atrium") -- floor, walls, an open roof
// ring, two rows of faceted columns joined by arches, and hanging banners/foliage cards.  With alphaMasked the cards
// (about a fifth of the triangles) carry a procedural RGBA leaf mask; with bumpMapped the floor carries a height map;
// emissiveFraction turns that share of the card materials into emitters (Zero-Day-class stand-in).
Mesh SampleModels::getAtrium(uint32_t targetTris, uint32_t seed, bool alphaMasked, bool bumpMapped, float emissiveFraction)
{
    Lcg rng(seed);
    std::vector<Vector3f> P;
    std::vector<Vector2f> T;
    std::vector<uint32_t> I, PM;
    auto vert = [&](const Vector3f& p, const Vector2f& t) { P.push_back(p); T.push_back(t); return (uint32_t)P.size() - 1; };
    auto tri = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t m) { I.push_back(a); I.push_back(b); I.push_back(c); PM.push_back(m); };
    // a grid patch from origin o spanning du, dv with nu x nv cells (winding so that cross(du,dv) is the normal)
    auto patch = [&](const Vector3f& o, const Vector3f& du, const Vector3f& dv, uint32_t nu, uint32_t nv, uint32_t m, float uvScale) {
        uint32_t base = (uint32_t)P.size();
        for (uint32_t j = 0; j <= nv; j++)
            for (uint32_t i = 0; i <= nu; i++) {
                float a = (float)i / (float)nu, b = (float)j / (float)nv;
                vert(o + a * du + b * dv, Vector2f(a * uvScale, b * uvScale));
            }
        for (uint32_t j = 0; j < nv; j++)
            for (uint32_t i = 0; i < nu; i++) {
                uint32_t v0 = base + j * (nu + 1) + i, v1 = v0 + 1, v2 = v0 + nu + 1, v3 = v2 + 1;
                tri(v0, v1, v2, m);
                tri(v1, v3, v2, m);
            }
    };
    enum { kFloor = 0, kWall, kColumn, kArch, kRoof, kCardFirst };
    const uint32_t kCardMaterials = 8;
    // budget: 25% shell, 40% columns, 15% arches, 20% cards
    const float L = 36.0f, W = 14.0f, H = 12.0f;
    uint32_t shellCells = std::max(2u, (uint32_t)std::sqrt((double)targetTris * 0.25 / 18.7));
    uint32_t nL = shellCells * 2, nW = shellCells, nH = shellCells;
    patch(Vector3f(-L / 2, 0, W / 2), Vector3f(L, 0, 0), Vector3f(0, 0, -W), nL, nW, kFloor, 8.0f);            // floor (+y)
    patch(Vector3f(-L / 2, 0, -W / 2), Vector3f(L, 0, 0), Vector3f(0, H, 0), nL, nH, kWall, 4.0f);             // back wall (+z)
    patch(Vector3f(L / 2, 0, W / 2), Vector3f(-L, 0, 0), Vector3f(0, H, 0), nL, nH, kWall, 4.0f);              // front wall (-z)
    patch(Vector3f(-L / 2, 0, W / 2), Vector3f(0, 0, -W), Vector3f(0, H, 0), nW, nH, kWall, 4.0f);             // left wall (+x)
    patch(Vector3f(L / 2, 0, -W / 2), Vector3f(0, 0, W), Vector3f(0, H, 0), nW, nH, kWall, 4.0f);              // right wall (-x)
    // roof ring: two strips along the long walls leave the middle open to the sky light
    patch(Vector3f(-L / 2, H, -W / 2), Vector3f(L, 0, 0), Vector3f(0, 0, W * 0.3f), nL, std::max(1u, nW / 3), kRoof, 4.0f);
    patch(Vector3f(-L / 2, H, W * 0.2f), Vector3f(L, 0, 0), Vector3f(0, 0, W * 0.3f), nL, std::max(1u, nW / 3), kRoof, 4.0f);
    // columns: two rows of 10, faceted cylinders with entasis; arches between neighbours
    const uint32_t colCount = 20;
    uint32_t colTris = (uint32_t)((double)targetTris * 0.40 / colCount);
    uint32_t cs = std::max(6u, (uint32_t)std::sqrt((double)colTris / 2.0 * 0.5)), ch = std::max(2u, colTris / (2 * cs));
    const float kPiF = 3.14159265358979323846f;
    for (uint32_t c = 0; c < colCount; c++) {
        float cx = -L / 2 + L * ((float)(c % 10) + 0.5f) / 10.0f, cz = (c < 10) ? -W * 0.28f : W * 0.28f;
        uint32_t base = (uint32_t)P.size();
        for (uint32_t j = 0; j <= ch; j++)
            for (uint32_t i = 0; i <= cs; i++) {
                float a = 2.0f * kPiF * (float)i / (float)cs, y = 7.0f * (float)j / (float)ch;
                float r = 0.45f * (1.0f - 0.15f * (float)j / (float)ch) + 0.03f * std::sin(8.0f * a);
                vert(Vector3f(cx + r * std::cos(a), y, cz + r * std::sin(a)), Vector2f((float)i / (float)cs * 2.0f, y));
            }
        for (uint32_t j = 0; j < ch; j++)
            for (uint32_t i = 0; i < cs; i++) {
                uint32_t v0 = base + j * (cs + 1) + i, v1 = v0 + 1, v2 = v0 + cs + 1, v3 = v2 + 1;
                tri(v0, v2, v1, kColumn);
                tri(v1, v2, v3, kColumn);
            }
    }
    uint32_t as = std::max(4u, (uint32_t)((double)targetTris * 0.15 / 108.0)); // 18 arches x 6 triangles per step
    for (uint32_t c = 0; c < colCount; c++) {
        if (c % 10 == 9) continue;
        float x0 = -L / 2 + L * ((float)(c % 10) + 0.5f) / 10.0f, x1 = x0 + L / 10.0f, cz = (c < 10) ? -W * 0.28f : W * 0.28f;
        float xc = 0.5f * (x0 + x1), rad = 0.5f * (x1 - x0);
        // a band following a semicircle from (x0,7) to (x1,7), with thickness in z and a flat top at y = 9.5
        for (int side = 0; side < 2; side++) {
            float z = cz + (side ? 0.35f : -0.35f);
            uint32_t base = (uint32_t)P.size();
            for (uint32_t i = 0; i <= as; i++) {
                float a = kPiF * (float)i / (float)as;
                float ax = xc - rad * std::cos(a), ay = 7.0f + rad * 0.9f * std::sin(a);
                vert(Vector3f(ax, ay, z), Vector2f((float)i / (float)as, 0.0f));
                vert(Vector3f(ax, 9.5f, z), Vector2f((float)i / (float)as, 1.0f));
            }
            for (uint32_t i = 0; i < as; i++) {
                uint32_t v0 = base + 2 * i, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
                if (side) { tri(v0, v2, v1, kArch); tri(v1, v2, v3, kArch); }
                else { tri(v0, v1, v2, kArch); tri(v1, v3, v2, kArch); }
            }
        }
        // underside of the arch
        uint32_t base = (uint32_t)P.size();
        for (uint32_t i = 0; i <= as; i++) {
            float a = kPiF * (float)i / (float)as;
            float ax = xc - rad * std::cos(a), ay = 7.0f + rad * 0.9f * std::sin(a);
            vert(Vector3f(ax, ay, cz - 0.35f), Vector2f((float)i / (float)as, 0.0f));
            vert(Vector3f(ax, ay, cz + 0.35f), Vector2f((float)i / (float)as, 1.0f));
        }
        for (uint32_t i = 0; i < as; i++) {
            uint32_t v0 = base + 2 * i, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
            tri(v0, v2, v1, kArch);
            tri(v1, v2, v3, kArch);
        }
    }
    // cards: small randomly oriented quads (2 triangles each), clustered like foliage / banners
    uint32_t have = (uint32_t)I.size() / 3;
    uint32_t cardTris = targetTris > have ? targetTris - have : 0;
    uint32_t cards = cardTris / 2;
    // clustered like potted plants at the column bases and hanging baskets under the arches, so that most of the
    // hall stays open to the light
    const uint32_t kClusters = 48;
    Vector3f cc[kClusters];
    float cr[kClusters];
    for (uint32_t k = 0; k < kClusters; k++) {
        uint32_t col = k % colCount;
        float cx = -L / 2 + L * ((float)(col % 10) + 0.5f) / 10.0f, cz = (col < 10) ? -W * 0.28f : W * 0.28f;
        bool hanging = k >= colCount && k < 2 * colCount;
        if (k >= 2 * colCount) { // along the walls
            cx = rng.range(-L / 2 + 2, L / 2 - 2);
            cz = (k & 1) ? -W / 2 + 0.9f : W / 2 - 0.9f;
        }
        cc[k] = Vector3f(cx + (hanging ? L / 20.0f : rng.range(-0.9f, 0.9f)), hanging ? 6.2f : rng.range(0.5f, 1.2f),
                         cz + (hanging ? 0.0f : rng.range(-0.9f, 0.9f)));
        cr[k] = hanging ? 0.55f : rng.range(0.5f, 0.9f);
    }
    for (uint32_t c = 0; c < cards; c++) {
        uint32_t k = rng.next() % kClusters;
        // rejection-free point in a ball: direction * radius^(1/3)
        float u1 = rng.range(-1.0f, 1.0f), ph = rng.range(0.0f, 2.0f * kPiF), rr = cr[k] * std::cbrt(rng.uniform());
        float sq = std::sqrt(std::fmax(0.0f, 1.0f - u1 * u1));
        Vector3f centre = cc[k] + Vector3f(rr * sq * std::cos(ph), rr * u1, rr * sq * std::sin(ph));
        float a = rng.range(0.0f, 2.0f * kPiF), tilt = rng.range(-0.9f, 0.9f), sz = rng.range(0.04f, 0.11f);
        Vector3f u(std::cos(a) * sz, std::sin(tilt) * sz, std::sin(a) * sz);
        Vector3f v(-std::sin(a) * std::sin(tilt) * sz, std::cos(tilt) * sz, std::cos(a) * std::sin(tilt) * sz);
        uint32_t m = kCardFirst + (rng.next() % kCardMaterials);
        uint32_t v0 = vert(centre - u - v, Vector2f(0, 0)), v1 = vert(centre + u - v, Vector2f(1, 0));
        uint32_t v2 = vert(centre - u + v, Vector2f(0, 1)), v3 = vert(centre + u + v, Vector2f(1, 1));
        tri(v0, v1, v2, m);
        tri(v1, v3, v2, m);
    }

    Mesh mesh;
    mesh.create((uint32_t)I.size() / 3, (uint32_t)P.size(), kCardFirst + kCardMaterials, false);
    memcpy(mesh.getIndexBuffer(), I.data(), I.size() * 4);
    memcpy((void*)mesh.getPositionBuffer(), P.data(), P.size() * sizeof(Vector3f));
    memcpy((void*)mesh.getTexcoordBuffer(), T.data(), T.size() * sizeof(Vector2f));
    memcpy(mesh.getPrimMateialBuffer(), PM.data(), PM.size() * 4);
    mesh.setHasTexcoord(true);
    Material* mats = mesh.getMaterialBuffer();
    const Vector3f stone(0.62f, 0.58f, 0.5f);
    mats[kFloor].diffuse = Vector3f(0.55f, 0.5f, 0.45f);
    mats[kWall].diffuse = stone;
    mats[kColumn].diffuse = Vector3f(0.7f, 0.68f, 0.62f);
    mats[kArch].diffuse = stone;
    mats[kRoof].diffuse = Vector3f(0.4f, 0.38f, 0.36f);
    Texture leaf, height;
    if (alphaMasked) {
        const uint32_t S = 256;
        std::vector<uint8_t> px((size_t)S * S * 4);
        Lcg tr(seed ^ 0x51ed27u);
        // blobs: alpha 255 inside a few random discs, 0 elsewhere, soft edge
        float bx[24], by[24], br[24];
        for (int i = 0; i < 24; i++) { bx[i] = tr.uniform(); by[i] = tr.uniform(); br[i] = tr.range(0.06f, 0.2f); }
        for (uint32_t y = 0; y < S; y++)
            for (uint32_t x = 0; x < S; x++) {
                float fx = ((float)x + 0.5f) / S, fy = ((float)y + 0.5f) / S, a = 0.0f;
                for (int i = 0; i < 24; i++) {
                    float dx = fx - bx[i], dy = fy - by[i];
                    float d = std::sqrt(dx * dx + dy * dy) / br[i];
                    a = std::fmax(a, std::fmin(std::fmax((1.0f - d) * 6.0f, 0.0f), 1.0f));
                }
                uint8_t* p = &px[((size_t)y * S + x) * 4];
                p[0] = (uint8_t)(40 + (tr.next() & 31));
                p[1] = (uint8_t)(120 + (tr.next() & 63));
                p[2] = (uint8_t)(30 + (tr.next() & 31));
                p[3] = (uint8_t)(255.0f * a);
            }
        leaf.create(S, S, 4, px.data());
    }
    if (bumpMapped) {
        const uint32_t S = 256;
        std::vector<uint8_t> px((size_t)S * S);
        for (uint32_t y = 0; y < S; y++)
            for (uint32_t x = 0; x < S; x++) {
                // paving stones: raised tiles with grooves
                float fx = (float)(x % 64) / 64.0f, fy = (float)(y % 64) / 64.0f;
                float e = std::fmin(std::fmin(fx, 1.0f - fx), std::fmin(fy, 1.0f - fy));
                px[(size_t)y * S + x] = (uint8_t)(255.0f * std::fmin(e * 8.0f, 1.0f));
            }
        height.create(S, S, 1, px.data());
        mats[kFloor].bumpMap = height;
    }
    Lcg mr(seed ^ 0xabcdefu);
    for (uint32_t k = 0; k < kCardMaterials; k++) {
        Material& m = mats[kCardFirst + k];
        m.diffuse = Vector3f(mr.range(0.3f, 0.9f), mr.range(0.3f, 0.9f), mr.range(0.3f, 0.9f));
        if (alphaMasked) {
            m.diffuseMap = leaf;
            m.alphaTest = leaf.isAlphaTestRequired();
        }
        if ((float)k < emissiveFraction * (float)kCardMaterials) m.emissive = Vector3f(mr.range(2.0f, 9.0f), mr.range(2.0f, 9.0f), mr.range(1.0f, 6.0f));
    }
    mesh.calculateBounds();
    return mesh;
}

// the same inflate for the OpenEXR reader's ZIP blocks (prt_host.cpp)
bool inflateZlibBytes(const std::vector<uint8_t>& in, std::vector<uint8_t>& out) { return inflateZlib(in, out); }

} // namespace prt
