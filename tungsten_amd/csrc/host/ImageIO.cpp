#include "ImageIO.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <iterator>
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

// Images come from scene files: untrusted input.  Nothing larger than 2^28 pixels (or 65 535 on a side) is decoded.
static bool saneImageSize(long long w, long long h) { return w > 0 && h > 0 && w <= 65535 && h <= 65535 && w*h <= (1ll << 28); }

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
    if (!in || !saneImageSize(w, h)) { err = "bad PFM header"; return false; }
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
    if (!saneImageSize(w, h)) { err = "image size out of range"; return false; }

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


// ---- PFM in (io/ImageIO.cpp:298-338): "PF" / "Pf", width height, scale, then rows bottom to top; like the reference the floats are
// taken as they lie in the file (native byte order; the scale's sign and magnitude are ignored)
bool loadPfm(const std::string &path, std::vector<float> &texels, int &w, int &h, int &channels, std::string &err)
{
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in) { err = "cannot open file"; return false; }
    std::string ident;
    in >> ident;
    if (ident == "Pf") channels = 1;
    else if (ident == "PF") channels = 3;
    else { err = "not a PFM file"; return false; }
    double scale;
    in >> w >> h >> scale;
    std::string rest;
    std::getline(in, rest);
    if (!in || !saneImageSize(w, h)) { err = "bad PFM header"; return false; }
    texels.assign(size_t(w)*h*channels, 0.0f);
    for (int y = 0; y < h; ++y)
        in.read(reinterpret_cast<char *>(&texels[size_t(h - y - 1)*w*channels]), std::streamsize(size_t(w)*channels*sizeof(float)));
    if (!in) { err = "PFM data too short"; return false; }
    return true;
}

// ---- PNG in (what the reference reads through lodepng: io/ImageIO.cpp:493-526 -> 8-bit RGBA) -----------------------
// A plain decoder of the PNG / zlib / DEFLATE specifications (RFC 2083, 1950, 1951): colour types 0, 2, 3, 4, 6 at 8 bits (16-bit
// samples keep their high byte, 1/2/4-bit grey and palette samples are unpacked), the five scanline filters, no interlacing.
namespace {

struct BitReader {
    const uint8_t *p; size_t n, pos; uint32_t bitBuf; int bitCnt;
    BitReader(const uint8_t *d, size_t len) : p(d), n(len), pos(0), bitBuf(0), bitCnt(0) {}
    bool bits(int need, uint32_t &v) {
        while (bitCnt < need) {
            if (pos >= n) return false;
            bitBuf |= uint32_t(p[pos++]) << bitCnt;
            bitCnt += 8;
        }
        v = need ? bitBuf & ((1u << need) - 1u) : 0u;
        bitBuf >>= need; bitCnt -= need;
        return true;
    }
    void alignToByte() { bitBuf = 0; bitCnt = 0; }
};

// canonical Huffman code (RFC 1951 3.2.2): symbols sorted by code length, then by value
struct Huffman {
    uint16_t count[16], symbol[288];
    bool build(const uint8_t *lengths, int n) {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) count[lengths[i]]++;
        int left = 1;
        for (int len = 1; len < 16; ++len) { left = left*2 - count[len]; if (left < 0) return false; }
        uint16_t offs[16]; offs[1] = 0;
        for (int len = 1; len < 15; ++len) offs[len + 1] = uint16_t(offs[len] + count[len]);
        for (int i = 0; i < n; ++i) if (lengths[i]) symbol[offs[lengths[i]]++] = uint16_t(i);
        return true;
    }
    int decode(BitReader &br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; ++len) {
            uint32_t b;
            if (!br.bits(1, b)) return -1;
            code |= int(b);
            int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

// `limit`: the decoder stops with an error once the output would grow beyond it (the caller knows how many bytes the image needs)
bool inflate(const uint8_t *src, size_t n, std::vector<uint8_t> &out, std::string &err, size_t limit)
{
    static const uint16_t lenBase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
    static const uint16_t lenExtra[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
    static const uint16_t distBase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
    static const uint16_t distExtra[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
    static const uint8_t clOrder[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
    if (n < 6) { err = "zlib stream too short"; return false; }
    if ((src[0] & 0x0F) != 8 || ((uint32_t(src[0]) << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) { err = "not a zlib stream"; return false; }
    BitReader br(src + 2, n - 2);
    for (;;) {
        uint32_t last, type;
        if (!br.bits(1, last) || !br.bits(2, type)) { err = "truncated deflate stream"; return false; }
        if (type == 0) {
            br.alignToByte();
            if (br.pos + 4 > br.n) { err = "truncated stored block"; return false; }
            uint32_t len = br.p[br.pos] | (uint32_t(br.p[br.pos + 1]) << 8), nlen = br.p[br.pos + 2] | (uint32_t(br.p[br.pos + 3]) << 8);
            br.pos += 4;
            if ((len ^ 0xFFFFu) != nlen || br.pos + len > br.n) { err = "bad stored block"; return false; }
            if (out.size() + len > limit) { err = "more image data than the header announces"; return false; }
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                for (int i = 0; i < 144; ++i) lengths[i] = 8;
                for (int i = 144; i < 256; ++i) lengths[i] = 9;
                for (int i = 256; i < 280; ++i) lengths[i] = 7;
                for (int i = 280; i < 288; ++i) lengths[i] = 8;
                lit.build(lengths, 288);
                for (int i = 0; i < 30; ++i) lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                uint32_t hlit, hdist, hclen;
                if (!br.bits(5, hlit) || !br.bits(5, hdist) || !br.bits(4, hclen)) { err = "truncated block header"; return false; }
                hlit += 257; hdist += 1; hclen += 4;
                if (hlit > 286 || hdist > 30) { err = "bad code counts"; return false; }
                uint8_t cl[19] = {0};
                for (uint32_t i = 0; i < hclen; ++i) { uint32_t v; if (!br.bits(3, v)) { err = "truncated code lengths"; return false; } cl[clOrder[i]] = uint8_t(v); }
                Huffman clh;
                if (!clh.build(cl, 19)) { err = "bad code-length code"; return false; }
                uint32_t i = 0;
                while (i < hlit + hdist) {
                    int sym = clh.decode(br);
                    if (sym < 0) { err = "bad code-length symbol"; return false; }
                    if (sym < 16) { lengths[i++] = uint8_t(sym); continue; }
                    uint32_t rep, prev = 0;
                    if (sym == 16) { if (i == 0) { err = "repeat without a length"; return false; } prev = lengths[i - 1]; if (!br.bits(2, rep)) { err = "truncated code lengths"; return false; } rep += 3; }
                    else if (sym == 17) { if (!br.bits(3, rep)) { err = "truncated code lengths"; return false; } rep += 3; }
                    else { if (!br.bits(7, rep)) { err = "truncated code lengths"; return false; } rep += 11; }
                    if (i + rep > hlit + hdist) { err = "too many code lengths"; return false; }
                    while (rep--) lengths[i++] = uint8_t(prev);
                }
                if (lengths[256] == 0 || !lit.build(lengths, int(hlit)) || !dist.build(lengths + hlit, int(hdist))) { err = "bad literal / distance code"; return false; }
            }
            for (;;) {
                int sym = lit.decode(br);
                if (sym < 0) { err = "bad literal / length symbol"; return false; }
                if (sym < 256) { if (out.size() >= limit) { err = "more image data than the header announces"; return false; } out.push_back(uint8_t(sym)); continue; }
                if (sym == 256) break;
                sym -= 257;
                if (sym >= 29) { err = "bad length symbol"; return false; }
                uint32_t eb, len = lenBase[sym];
                if (!br.bits(lenExtra[sym], eb)) { err = "truncated length"; return false; }
                len += eb;
                int ds = dist.decode(br);
                if (ds < 0 || ds >= 30) { err = "bad distance symbol"; return false; }
                uint32_t d = distBase[ds];
                if (!br.bits(distExtra[ds], eb)) { err = "truncated distance"; return false; }
                d += eb;
                if (d > out.size()) { err = "distance beyond the start of the data"; return false; }
                if (out.size() + len > limit) { err = "more image data than the header announces"; return false; }
                size_t from = out.size() - d;
                for (uint32_t k = 0; k < len; ++k) out.push_back(out[from + k]);
            }
        } else { err = "reserved block type"; return false; }
        if (last) break;
    }
    return true;
}

inline int paeth(int a, int b, int c) { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }

} // namespace

bool loadPng(const std::string &path, std::vector<uint8_t> &rgba, int &w, int &h, bool &hasAlpha, std::string &err)
{
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in) { err = "cannot open file"; return false; }
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0) { err = "not a PNG file"; return false; }
    auto be32 = [&](size_t o) { return (uint32_t(file[o]) << 24) | (uint32_t(file[o + 1]) << 16) | (uint32_t(file[o + 2]) << 8) | file[o + 3]; };
    std::vector<uint8_t> idat, palette, trns;
    int depth = 0, colorType = 0;
    bool haveHeader = false;
    for (size_t o = 8; o + 12 <= file.size();) {
        const uint32_t len = be32(o);
        if (o + 12 + size_t(len) > file.size()) { err = "truncated chunk"; return false; }
        const std::string type(reinterpret_cast<const char *>(&file[o + 4]), 4);
        const uint8_t *d = &file[o + 8];
        if (crc32(&file[o + 4], size_t(len) + 4) != be32(o + 8 + len)) { err = "chunk checksum mismatch"; return false; }
        if (type == "IHDR") {
            if (len != 13) { err = "bad IHDR"; return false; }
            w = int(be32(o + 8)); h = int(be32(o + 12));
            depth = d[8]; colorType = d[9];
            if (d[10] != 0 || d[11] != 0) { err = "unknown compression / filter method"; return false; }
            if (d[12] != 0) { err = "interlaced PNGs are not supported"; return false; }
            haveHeader = true;
        } else if (type == "PLTE") palette.assign(d, d + len);
        else if (type == "tRNS") trns.assign(d, d + len);
        else if (type == "IDAT") idat.insert(idat.end(), d, d + len);
        else if (type == "IEND") break;
        o += 12 + size_t(len);
    }
    if (!haveHeader || idat.empty()) { err = "missing IHDR / IDAT"; return false; }
    if (!saneImageSize(w, h)) { err = "image size out of range"; return false; }
    int channels;
    switch (colorType) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: err = "unknown colour type"; return false;
    }
    const bool depthOk = depth == 8 || (depth == 16 && colorType != 3) || ((depth == 1 || depth == 2 || depth == 4) && (colorType == 0 || colorType == 3));
    if (!depthOk) { err = "unsupported bit depth"; return false; }
    if (colorType == 3 && palette.empty()) { err = "palette image without PLTE"; return false; }
    const size_t bpp = std::max<size_t>(1, size_t(channels)*size_t(depth)/8);            // filter unit
    const size_t stride = (size_t(w)*size_t(channels)*size_t(depth) + 7)/8;
    std::vector<uint8_t> raw;
    raw.reserve((stride + 1)*size_t(h));
    if (!inflate(idat.data(), idat.size(), raw, err, (stride + 1)*size_t(h))) return false;
    if (raw.size() < (stride + 1)*size_t(h)) { err = "image data too short"; return false; }
    // scanline filters (RFC 2083 section 6), in place
    std::vector<uint8_t> prev(stride, 0);
    for (int y = 0; y < h; ++y) {
        uint8_t *line = &raw[size_t(y)*(stride + 1)];
        const int filter = line[0];
        uint8_t *cur = line + 1;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int v = cur[i];
            switch (filter) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b)/2; break;
            case 4: v += paeth(a, b, c); break;
            default: err = "unknown scanline filter"; return false;
            }
            cur[i] = uint8_t(v);
        }
        std::memcpy(prev.data(), cur, stride);
    }
    // to 8-bit RGBA
    hasAlpha = colorType == 4 || colorType == 6 || !trns.empty();
    rgba.assign(size_t(w)*h*4, 255);
    for (int y = 0; y < h; ++y) {
        const uint8_t *cur = &raw[size_t(y)*(stride + 1) + 1];
        for (int x = 0; x < w; ++x) {
            uint8_t *dst = &rgba[(size_t(y)*w + x)*4];
            auto sample = [&](int ch) -> int {                   // sample `ch` of pixel x at the image's bit depth
                const size_t idx = size_t(x)*channels + ch;
                if (depth == 8) return cur[idx];
                if (depth == 16) return (cur[idx*2] << 8) | cur[idx*2 + 1];
                const size_t bit = idx*size_t(depth);
                return (cur[bit >> 3] >> (8 - depth - int(bit & 7))) & ((1 << depth) - 1);
            };
            auto to8 = [&](int v) -> uint8_t {
                if (depth == 8) return uint8_t(v);
                if (depth == 16) return uint8_t(v >> 8);
                return uint8_t(v*255/((1 << depth) - 1));
            };
            if (colorType == 3) {
                const int idx = sample(0);
                if (size_t(idx)*3 + 2 >= palette.size()) { err = "palette index out of range"; return false; }
                dst[0] = palette[size_t(idx)*3]; dst[1] = palette[size_t(idx)*3 + 1]; dst[2] = palette[size_t(idx)*3 + 2];
                dst[3] = size_t(idx) < trns.size() ? trns[size_t(idx)] : 255;
            } else if (colorType == 0 || colorType == 4) {
                const int g = sample(0);
                dst[0] = dst[1] = dst[2] = to8(g);
                if (colorType == 4) dst[3] = to8(sample(1));
                else if (trns.size() >= 2 && g == ((trns[0] << 8) | trns[1])) dst[3] = 0;
            } else {
                const int r = sample(0), g = sample(1), b = sample(2);
                dst[0] = to8(r); dst[1] = to8(g); dst[2] = to8(b);
                if (colorType == 6) dst[3] = to8(sample(3));
                else if (trns.size() >= 6 && r == ((trns[0] << 8) | trns[1]) && g == ((trns[2] << 8) | trns[3]) && b == ((trns[4] << 8) | trns[5])) dst[3] = 0;
            }
        }
    }
    return true;
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
