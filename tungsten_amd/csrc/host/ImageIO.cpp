#include "ImageIO.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <unordered_map>

namespace tungsten_amd {

namespace ImageIO {

// ---- Radiance RGBE ---------------------------------------------------------------------------
// Texel conversion matches stb_image's stbi__hdr_convert (thirdparty/stbi/stb_image.c:5588-5613),
// which is what the reference's ImageIO::loadStbiHdr ends up calling: c * ldexp(1, e - 136).
static inline void rgbeToFloat(const uint8_t *rgbe, float *out)
{
    if (rgbe[3] != 0) {
        float f1 = float(std::ldexp(1.0f, int(rgbe[3]) - (128 + 8)));
        out[0] = rgbe[0]*f1;
        out[1] = rgbe[1]*f1;
        out[2] = rgbe[2]*f1;
    } else {
        out[0] = out[1] = out[2] = 0.0f;
    }
}

bool loadPfm(const std::string &path, std::vector<float> &rgb, int &w, int &h, std::string &err)
{
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in) { err = "cannot open file"; return false; }
    std::string magic;
    in >> magic;
    int channels = magic == "PF" ? 3 : magic == "Pf" ? 1 : 0;
    if (!channels) { err = "not a PFM file"; return false; }
    double scale;
    in >> w >> h >> scale;
    in.get();
    if (!in || w <= 0 || h <= 0) { err = "bad PFM header"; return false; }
    std::vector<float> row(size_t(w)*channels);
    rgb.resize(size_t(w)*h*3);
    for (int y = 0; y < h; ++y) {
        in.read(reinterpret_cast<char *>(row.data()), row.size()*sizeof(float));
        if (!in) { err = "truncated PFM"; return false; }
        float *dst = &rgb[size_t(h - y - 1)*w*3];
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c)
                dst[x*3 + c] = row[x*channels + (channels == 3 ? c : 0)];
    }
    return true;
}

bool loadHdr(const std::string &path, std::vector<float> &rgb, int &w, int &h, std::string &err)
{
    if (path.size() > 4 && path.substr(path.size() - 4) == ".pfm")
        return loadPfm(path, rgb, w, h, err);

    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in) { err = "cannot open file"; return false; }
    std::string line;
    std::getline(in, line);
    if (line.compare(0, 10, "#?RADIANCE") != 0 && line.compare(0, 6, "#?RGBE") != 0) { err = "not a Radiance HDR file"; return false; }
    bool formatOk = false;
    while (std::getline(in, line)) {
        if (line.empty() || line == "\r") break;
        if (line.compare(0, 23, "FORMAT=32-bit_rle_rgbe") == 0) formatOk = true;
    }
    if (!formatOk) { err = "unsupported HDR format"; return false; }
    std::getline(in, line);
    if (std::sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2) { err = "unsupported HDR data layout"; return false; }

    rgb.resize(size_t(w)*h*3);
    std::vector<uint8_t> scan(size_t(w)*4);
    for (int y = 0; y < h; ++y) {
        uint8_t hdr4[4];
        in.read(reinterpret_cast<char *>(hdr4), 4);
        if (!in) { err = "truncated HDR"; return false; }
        bool rle = w >= 8 && w < 32768 && hdr4[0] == 2 && hdr4[1] == 2 && !(hdr4[2] & 0x80);
        if (!rle) {
            // flat scanline: the 4 bytes are the first pixel
            std::memcpy(scan.data(), hdr4, 4);
            in.read(reinterpret_cast<char *>(scan.data() + 4), std::streamsize(w - 1)*4);
            if (!in) { err = "truncated HDR"; return false; }
        } else {
            if (((int(hdr4[2]) << 8) | hdr4[3]) != w) { err = "HDR scanline width mismatch"; return false; }
            for (int c = 0; c < 4; ++c) {
                int x = 0;
                while (x < w) {
                    int count = in.get();
                    if (count < 0) { err = "truncated HDR"; return false; }
                    if (count > 128) {
                        int value = in.get();
                        count -= 128;
                        if (x + count > w) { err = "corrupt HDR run"; return false; }
                        for (int i = 0; i < count; ++i) scan[size_t(x++)*4 + c] = uint8_t(value);
                    } else {
                        if (x + count > w || count == 0) { err = "corrupt HDR run"; return false; }
                        for (int i = 0; i < count; ++i) scan[size_t(x++)*4 + c] = uint8_t(in.get());
                    }
                }
            }
        }
        for (int x = 0; x < w; ++x)
            rgbeToFloat(&scan[size_t(x)*4], &rgb[(size_t(y)*w + x)*3]);
    }
    return true;
}

bool savePfm(const std::string &path, const float *img, int w, int h, int channels)
{
    if (channels != 1 && channels != 3)
        return false;
    std::ofstream out(path.c_str(), std::ios::binary);
    if (!out)
        return false;
    out << ((channels == 1) ? "Pf" : "PF") << '\n';
    out << w << " " << h << '\n';
    out << -1.0 << '\n';
    for (int y = 0; y < h; ++y)
        out.write(reinterpret_cast<const char *>(img + size_t(h - y - 1)*w*channels), std::streamsize(w)*channels*sizeof(float));
    return bool(out);
}

// ---- PNG (stored/uncompressed deflate) ---------------------------------------------------------
static uint32_t crcTable[256];
static void initCrc()
{
    static bool done = false;
    if (done) return;
    for (uint32_t n = 0; n < 256; ++n) {
        uint32_t c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crcTable[n] = c;
    }
    done = true;
}
static uint32_t crc32(const uint8_t *d, size_t n, uint32_t crc = 0)
{
    initCrc();
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = crcTable[(crc ^ d[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
static void put32(std::vector<uint8_t> &v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
static void chunk(std::vector<uint8_t> &out, const char *tag, const std::vector<uint8_t> &data)
{
    put32(out, uint32_t(data.size()));
    std::vector<uint8_t> body(tag, tag + 4);
    body.insert(body.end(), data.begin(), data.end());
    out.insert(out.end(), body.begin(), body.end());
    put32(out, crc32(body.data(), body.size()));
}

bool savePng(const std::string &path, const uint8_t *rgb, int w, int h)
{
    std::vector<uint8_t> raw;
    raw.reserve(size_t(h)*(size_t(w)*3 + 1));
    for (int y = 0; y < h; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb + size_t(y)*w*3, rgb + size_t(y + 1)*w*3);
    }
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (uint8_t c : raw) { a = (a + c) % 65521u; b = (b + a) % 65521u; }
    size_t pos = 0;
    while (pos < raw.size()) {
        size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back(n & 0xFF); z.push_back(n >> 8);
        z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
    }
    put32(z, (b << 16) | a);

    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put32(ihdr, uint32_t(w)); put32(ihdr, uint32_t(h));
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", std::vector<uint8_t>());
    std::ofstream f(path.c_str(), std::ios::binary);
    if (!f) return false;
    f.write(reinterpret_cast<const char *>(out.data()), std::streamsize(out.size()));
    return bool(f);
}

Vec3f tonemap(const std::string &op, const Vec3f &c)
{
    if (op == "linear")
        return c;
    if (op == "gamma")
        return Vec3f(std::pow(c[0], 1.0f/2.2f), std::pow(c[1], 1.0f/2.2f), std::pow(c[2], 1.0f/2.2f));
    if (op == "reinhard") {
        Vec3f r;
        for (int i = 0; i < 3; ++i) r[i] = std::pow(c[i]/(c[i] + 1.0f), 1.0f/2.2f);
        return r;
    }
    if (op == "filmic") {
        Vec3f r;
        for (int i = 0; i < 3; ++i) {
            float x = std::max(0.0f, c[i] - 0.004f);
            r[i] = (x*(6.2f*x + 0.5f))/(x*(6.2f*x + 1.7f) + 0.06f);
        }
        return r;
    }
    if (op == "pbrt") {
        Vec3f r;
        for (int i = 0; i < 3; ++i)
            r[i] = c[i] < 0.0031308f ? 12.92f*c[i] : 1.055f*std::pow(c[i], 1.0f/2.4f) - 0.055f;
        return r;
    }
    throw std::runtime_error("Invalid tonemap operator: '" + op + "'");
}

} // namespace ImageIO

namespace MeshIO {

static bool loadWo3(const std::string &path, std::vector<MeshVertex> &verts, std::vector<MeshTriangle> &tris, std::string &err)
{
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in) { err = "cannot open file"; return false; }
    uint64_t numVerts = 0, numTris = 0;
    in.read(reinterpret_cast<char *>(&numVerts), 8);
    if (!in || numVerts > (1ull << 32)) { err = "bad vertex count"; return false; }
    verts.resize(size_t(numVerts));
    in.read(reinterpret_cast<char *>(verts.data()), std::streamsize(numVerts*sizeof(MeshVertex)));
    in.read(reinterpret_cast<char *>(&numTris), 8);
    if (!in || numTris > (1ull << 32)) { err = "bad triangle count"; return false; }
    tris.resize(size_t(numTris));
    in.read(reinterpret_cast<char *>(tris.data()), std::streamsize(numTris*sizeof(MeshTriangle)));
    if (!in) { err = "truncated .wo3"; return false; }
    for (const MeshTriangle &t : tris)
        if (t.v0 >= numVerts || t.v1 >= numVerts || t.v2 >= numVerts) { err = "triangle index out of range"; return false; }
    return true;
}

bool saveWo3(const std::string &path, const std::vector<MeshVertex> &verts, const std::vector<MeshTriangle> &tris)
{
    std::ofstream out(path.c_str(), std::ios::binary);
    if (!out) return false;
    uint64_t nv = verts.size(), nt = tris.size();
    out.write(reinterpret_cast<const char *>(&nv), 8);
    out.write(reinterpret_cast<const char *>(verts.data()), std::streamsize(nv*sizeof(MeshVertex)));
    out.write(reinterpret_cast<const char *>(&nt), 8);
    out.write(reinterpret_cast<const char *>(tris.data()), std::streamsize(nt*sizeof(MeshTriangle)));
    return bool(out);
}

// Wavefront OBJ subset: v / vt / vn / f with fan triangulation, one material slot.  Vertices
// without a normal index get the constant normal (0,1,0) like io/ObjLoader.cpp:71-78.
static bool loadObj(const std::string &path, std::vector<MeshVertex> &verts, std::vector<MeshTriangle> &tris, std::string &err)
{
    std::ifstream in(path.c_str());
    if (!in) { err = "cannot open file"; return false; }
    std::vector<Vec3f> pos, nrm;
    std::vector<std::pair<float, float>> uvs;
    std::map<std::tuple<int, int, int>, uint32_t> indices;
    std::string line;
    auto fetch = [&](int p, int n, int u) -> uint32_t {
        if (p < 0) p += int(pos.size()) + 1;
        if (n < 0) n += int(nrm.size()) + 1;
        if (u < 0) u += int(uvs.size()) + 1;
        auto key = std::make_tuple(p, n, u);
        auto it = indices.find(key);
        if (it != indices.end()) return it->second;
        MeshVertex v;
        Vec3f P(0.0f), N(0.0f, 1.0f, 0.0f);
        float U = 0.0f, V = 0.0f;
        if (p && p <= int(pos.size())) P = pos[size_t(p - 1)];
        if (n && n <= int(nrm.size())) N = nrm[size_t(n - 1)];
        if (u && u <= int(uvs.size())) { U = uvs[size_t(u - 1)].first; V = uvs[size_t(u - 1)].second; }
        for (int k = 0; k < 3; ++k) { v.pos[k] = P[k]; v.normal[k] = N[k]; }
        v.uv[0] = U; v.uv[1] = V;
        uint32_t idx = uint32_t(verts.size());
        verts.push_back(v);
        indices[key] = idx;
        return idx;
    };
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string tag;
        ss >> tag;
        if (tag == "v") { Vec3f p; ss >> p[0] >> p[1] >> p[2]; pos.push_back(p); }
        else if (tag == "vn") { Vec3f n; ss >> n[0] >> n[1] >> n[2]; nrm.push_back(n); }
        else if (tag == "vt") { float u = 0, v = 0; ss >> u >> v; uvs.emplace_back(u, v); }
        else if (tag == "f") {
            std::vector<uint32_t> poly;
            std::string tok;
            while (ss >> tok) {
                int p = 0, u = 0, n = 0;
                if (std::sscanf(tok.c_str(), "%d/%d/%d", &p, &u, &n) == 3) {}
                else if (std::sscanf(tok.c_str(), "%d//%d", &p, &n) == 2) { u = 0; }
                else if (std::sscanf(tok.c_str(), "%d/%d", &p, &u) == 2) { n = 0; }
                else { std::sscanf(tok.c_str(), "%d", &p); u = n = 0; }
                poly.push_back(fetch(p, n, u));
            }
            for (size_t i = 2; i < poly.size(); ++i)
                tris.push_back(MeshTriangle{poly[0], poly[i - 1], poly[i], 0});
        }
    }
    if (tris.empty()) { err = "no faces in OBJ"; return false; }
    return true;
}

bool load(const std::string &path, std::vector<MeshVertex> &verts, std::vector<MeshTriangle> &tris, std::string &err)
{
    verts.clear(); tris.clear();
    if (path.size() > 4 && path.substr(path.size() - 4) == ".wo3")
        return loadWo3(path, verts, tris, err);
    if (path.size() > 4 && path.substr(path.size() - 4) == ".obj")
        return loadObj(path, verts, tris, err);
    err = "unknown mesh extension";
    return false;
}

// TriangleMesh::calcSmoothVertexNormals (TriangleMesh.cpp:174-231): split vertices at creases
// sharper than cos(0.15*pi), then area-weighted averaging over coincident vertices.
void recomputeNormals(std::vector<MeshVertex> &verts, std::vector<MeshTriangle> &tris)
{
    static const float SplitLimit = std::cos(PI*0.15f);
    struct Key { float p[3]; bool operator==(const Key &o) const { return p[0] == o.p[0] && p[1] == o.p[1] && p[2] == o.p[2]; } };
    struct KeyHash { size_t operator()(const Key &k) const {
        uint32_t b[3]; std::memcpy(b, k.p, 12);
        return size_t(hash32(b[0]) ^ hash32(b[1] + 0x9E3779B9u) ^ hash32(b[2] + 0x7F4A7C15u)); } };
    auto P = [&](uint32_t i) { return Vec3f(verts[i].pos[0], verts[i].pos[1], verts[i].pos[2]); };
    auto isZero = [](const Vec3f &v) { return v[0] == 0.0f && v[1] == 0.0f && v[2] == 0.0f; };

    std::vector<Vec3f> geometricN(verts.size(), Vec3f(0.0f));
    std::vector<Vec3f> accum(verts.size(), Vec3f(0.0f));
    std::unordered_multimap<Key, uint32_t, KeyHash> posToVert;
    for (uint32_t i = 0; i < verts.size(); ++i)
        posToVert.insert(std::make_pair(Key{{verts[i].pos[0], verts[i].pos[1], verts[i].pos[2]}}, i));

    for (MeshTriangle &t : tris) {
        Vec3f normal = (P(t.v1) - P(t.v0)).cross(P(t.v2) - P(t.v0));
        if (isZero(normal)) normal = Vec3f(0.0f, 1.0f, 0.0f);
        else normal.normalize();
        uint32_t *vs[3] = {&t.v0, &t.v1, &t.v2};
        for (int i = 0; i < 3; ++i) {
            Vec3f &n = geometricN[*vs[i]];
            if (isZero(n)) {
                n = normal;
            } else if (n.dot(normal) < SplitLimit) {
                verts.push_back(verts[*vs[i]]);
                geometricN.push_back(normal);
                accum.push_back(Vec3f(0.0f));
                *vs[i] = uint32_t(verts.size() - 1);
            }
        }
    }
    for (MeshTriangle &t : tris) {
        Vec3f normal = (P(t.v1) - P(t.v0)).cross(P(t.v2) - P(t.v0));
        Vec3f nN = normal.normalized();
        uint32_t vs[3] = {t.v0, t.v1, t.v2};
        for (int i = 0; i < 3; ++i) {
            auto range = posToVert.equal_range(Key{{verts[vs[i]].pos[0], verts[vs[i]].pos[1], verts[vs[i]].pos[2]}});
            for (auto it = range.first; it != range.second; ++it)
                if (geometricN[it->second].dot(nN) >= SplitLimit)
                    accum[it->second] += normal;
        }
    }
    for (uint32_t i = 0; i < verts.size(); ++i) {
        Vec3f n = isZero(accum[i]) ? geometricN[i] : accum[i].normalized();
        for (int k = 0; k < 3; ++k) verts[i].normal[k] = n[k];
    }
}

} // namespace MeshIO

} // namespace tungsten_amd
