#include "Scene.hpp"
#include "SkyModel.hpp"
#include "ImageIO.hpp"

#include <thread>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <sstream>

namespace tungsten_amd {

// ------------------------------------------------------------------------------------------
// JSON helpers (JsonPtr.hpp:52-65: a scalar broadcasts to every vector component)
// ------------------------------------------------------------------------------------------
static bool getVec3(const JsonValue &parent, const char *key, Vec3f &dst)
{
    const JsonValue &v = parent[key];
    if (!v) return false;
    if (v.isNumber()) { dst = Vec3f(v.asFloat()); return true; }
    if (v.isArray() && v.size() == 3) { dst = Vec3f(v[0].asFloat(), v[1].asFloat(), v[2].asFloat()); return true; }
    throw JsonLoadException(std::string("JSON: field '") + key + "' is not a 3-vector");
}

static Vec3f randomOrtho(const Vec3f &a)   // io/JsonPtr.cpp:77-89
{
    Vec3f res;
    if (std::abs(a.x()) > std::abs(a.y()))
        res = Vec3f(0.0f, 1.0f, 0.0f);
    else
        res = Vec3f(1.0f, 0.0f, 0.0f);
    return a.cross(res).normalized();
}

static void gramSchmidt(Vec3f &a, Vec3f &b, Vec3f &c)   // io/JsonPtr.cpp:91-106
{
    a.normalize();
    b -= a*a.dot(b);
    if (b.lengthSq() < 1e-5)
        b = randomOrtho(a);
    else
        b.normalize();

    c -= a*a.dot(c);
    c -= b*b.dot(c);
    if (c.lengthSq() < 1e-5)
        c = a.cross(b);
    else
        c.normalize();
}

// io/JsonPtr.cpp:108-186
static bool getTransform(const JsonValue &parent, const char *key, Mat4f &dst)
{
    const JsonValue &v = parent[key];
    if (!v) return false;
    if (v.isArray()) {
        if (v.size() != 16)
            throw JsonLoadException("JSON: matrix needs 16 elements");
        for (int i = 0; i < 16; ++i) dst[i] = v[size_t(i)].asFloat();
        return true;
    }
    if (!v.isObject())
        throw JsonLoadException("JSON: expecting a matrix value");

    Vec3f x(1.0f, 0.0f, 0.0f), y(0.0f, 1.0f, 0.0f), z(0.0f, 0.0f, 1.0f);
    Vec3f pos(0.0f);
    getVec3(v, "position", pos);

    bool explicitX = false, explicitY = false, explicitZ = false;
    Vec3f lookAt;
    if (getVec3(v, "look_at", lookAt)) {
        z = lookAt - pos;
        explicitZ = true;
    }
    explicitY = getVec3(v, "up", y);
    explicitX = getVec3(v, "x_axis", x) || explicitX;
    explicitY = getVec3(v, "y_axis", y) || explicitY;
    explicitZ = getVec3(v, "z_axis", z) || explicitZ;

    int id = (explicitZ ? 4 : 0) + (explicitY ? 2 : 0) + (explicitX ? 1 : 0);
    switch (id) {
    case 0: gramSchmidt(z, y, x); break;
    case 1: gramSchmidt(x, z, y); break;
    case 2: gramSchmidt(y, z, x); break;
    case 3: gramSchmidt(y, x, z); break;
    case 4: gramSchmidt(z, y, x); break;
    case 5: gramSchmidt(z, x, y); break;
    case 6: gramSchmidt(z, y, x); break;
    case 7: gramSchmidt(z, y, x); break;
    }
    if (x.cross(y).dot(z) < 0.0f) {
        if (!explicitX) x = -x;
        else if (!explicitY) y = -y;
        else z = -z;
    }
    Vec3f scale;
    if (getVec3(v, "scale", scale)) {
        x *= scale.x(); y *= scale.y(); z *= scale.z();
    }
    Vec3f rot;
    if (getVec3(v, "rotation", rot)) {
        Mat4f tform = Mat4f::rotYXZ(rot);
        x = tform*x; y = tform*y; z = tform*z;
    }
    dst = Mat4f(x, y, z);
    dst[3] = pos[0]; dst[7] = pos[1]; dst[11] = pos[2];
    return true;
}

// ------------------------------------------------------------------------------------------
// Textures
// ------------------------------------------------------------------------------------------
Vec3f Texture::average() const
{
    switch (type) {
    case Constant: return value;
    case Checker:  return (onColor + offColor)*0.5f;          // CheckerTexture.cpp:49-52
    default:       return scale*texAvg;                       // BitmapTexture.cpp:283-286
    }
}

Vec3f Texture::maximum() const
{
    switch (type) {
    case Constant: return value;
    case Checker:  return vmax(onColor, offColor);
    default:       return scale*texMax;
    }
}

void Texture::scaleValues(float f)
{
    switch (type) {
    case Constant: value = value*f; break;
    case Checker:  onColor = onColor*f; offColor = offColor*f; break;
    default:       scale *= f; break;
    }
}

void Texture::loadBitmap(const std::string &file)
{
    path = file;
    type = Bitmap;
    std::vector<float> rgbTexels;
    int iw = 0, ih = 0;
    std::string err;
    bool scalarDone = false;
    const bool png = file.size() >= 4 && (file.compare(file.size() - 4, 4, ".png") == 0 || file.compare(file.size() - 4, 4, ".PNG") == 0);
    if (png) {
        // ImageIO::loadLdr (io/ImageIO.cpp:493-526) + BitmapTexture::getRgb / getScalar (textures/BitmapTexture.cpp:139-154): RGB requests
        // keep the 8-bit channels (gamma-corrected through the 2.2 table unless "gamma_correct": false), scalar requests take the integer
        // average of the raw channels; a lookup turns a byte into float(byte)*(1/255) -- done once here, with the same rounding
        std::vector<uint8_t> rgba;
        bool hasAlpha = false;
        if (!ImageIO::loadPng(file, rgba, iw, ih, hasAlpha, err))
            throw std::runtime_error("Unable to load PNG texture '" + file + "': " + err);
        w = iw; h = ih;
        uint8_t gamma[256];
        for (int i = 0; i < 256; ++i)            // GammaCorrection[] (io/ImageIO.cpp:26-43) = floor(255 (i/255)^2.2)
            gamma[i] = gammaCorrect ? uint8_t(255.0*std::pow(i/255.0, 2.2)) : uint8_t(i);
        const size_t n = size_t(w)*h;
        if (rgb) {
            texels.resize(n*3);
            for (size_t i = 0; i < n; ++i)
                for (int k = 0; k < 3; ++k)
                    texels[i*3 + k] = float(gamma[rgba[i*4 + k]])*(1.0f/255.0f);
        } else {
            texels.resize(n);
            for (size_t i = 0; i < n; ++i)
                texels[i] = float(autoAlpha ? rgba[i*4 + 3] : uint8_t((int(rgba[i*4]) + int(rgba[i*4 + 1]) + int(rgba[i*4 + 2]))/3))*(1.0f/255.0f);
        }
    } else if (file.size() >= 4 && file.compare(file.size() - 4, 4, ".pfm") == 0) {
        // ImageIO::loadPfm (io/ImageIO.cpp:298-338): a scalar file feeds all three channels of an RGB request
        int ch = 0;
        if (!ImageIO::loadPfm(file, rgbTexels, iw, ih, ch, err))
            throw std::runtime_error("Unable to load PFM texture '" + file + "': " + err);
        if (ch == 1 && !rgb) {
            texels.swap(rgbTexels);              // a scalar request of a scalar file: the floats as they are
            scalarDone = true;
        } else if (ch == 1) {
            std::vector<float> grey;
            grey.swap(rgbTexels);
            rgbTexels.resize(grey.size()*3);
            for (size_t i = 0; i < grey.size(); ++i)
                rgbTexels[i*3] = rgbTexels[i*3 + 1] = rgbTexels[i*3 + 2] = grey[i];
        }
    } else if (!ImageIO::loadHdr(file, rgbTexels, iw, ih, err)) {
        throw std::runtime_error("Unable to load texture '" + file + "': " + err +
                                 " (.hdr, .pfm and .png bitmaps are read; .jpg / .exr are outside the path_tracer_hip hot-path scope)");
    }
    w = iw; h = ih;
    if (png || scalarDone) {
        // (converted above)
    } else if (rgb) {
        texels.swap(rgbTexels);
    } else {
        // TexelConversion::REQUEST_AVERAGE on an RGB HDR source (ImageIO.cpp:298-337)
        texels.resize(size_t(w)*h);
        for (size_t i = 0; i < texels.size(); ++i)
            texels[i] = (rgbTexels[i*3] + rgbTexels[i*3 + 1] + rgbTexels[i*3 + 2])/3.0f;
    }
    valid = true;
    finishBitmap();
}

// BitmapTexture::init (BitmapTexture.cpp:175-209): min/max/avg, avg accumulated as texel/(w*h)
void Texture::finishBitmap()
{
    if (rgb) {
        texMin = texMax = Vec3f(texels[0], texels[1], texels[2]);
        texAvg = Vec3f(0.0f);
        float n = float(w*h);
        for (size_t i = 0; i < size_t(w)*h; ++i) {
            Vec3f c(texels[i*3], texels[i*3 + 1], texels[i*3 + 2]);
            texMin = vmin(texMin, c);
            texMax = vmax(texMax, c);
            texAvg += c/n;
        }
    } else {
        float mn = texels[0], mx = texels[0], av = 0.0f;
        for (size_t i = 0; i < texels.size(); ++i) {
            mn = std::min(mn, texels[i]); mx = std::max(mx, texels[i]);
            av += texels[i]/float(w*h);
        }
        texMin = Vec3f(mn); texMax = Vec3f(mx); texAvg = Vec3f(av);
    }
}

// BitmapTexture::makeSamplable(MAP_SPHERICAL or MAP_UNIFORM) (BitmapTexture.cpp:400-431) followed by the
// Distribution2D constructor (sampling/Distribution2D.hpp:18-66).  Float-for-float the same
// operation order, so the tables (and therefore MIS weights) are bit-identical.  (One set of tables per texture: the environment
// maps ask for the spherical one, a thin-lens camera's aperture bitmap -- a texture of its own -- for the uniform one.)
void Texture::makeSamplable(bool spherical)
{
    if (samplable || type != Bitmap)
        return;
    std::vector<float> weights(size_t(w)*h);
    for (int y = 0, idx = 0; y < h; ++y) {
        float rowWeight = 1.0f;
        if (spherical)
            rowWeight *= std::sin((y*PI)/h);
        for (int x = 0; x < w; ++x, ++idx) {
            float wt = rgb ? std::max(texels[idx*3], std::max(texels[idx*3 + 1], texels[idx*3 + 2])) : texels[idx];
            weights[idx] = wt*rowWeight;
        }
    }
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w - 1; ++x)
            weights[x + y*w] = std::max(weights[x + y*w], weights[x + 1 + y*w]);
        if (!clamp)
            weights[y*w] = weights[w - 1 + y*w] = std::max(weights[w - 1 + y*w], weights[y*w]);
        for (int x = w - 1; x > 0; --x)
            weights[x + y*w] = std::max(weights[x + y*w], weights[x - 1 + y*w]);
    }
    for (int x = 0; x < w; ++x) {
        for (int y = 0; y < h - 1; ++y)
            weights[x + y*w] = std::max(weights[x + y*w], weights[x + (y + 1)*w]);
        if (!clamp)
            weights[x] = weights[x + (h - 1)*w] = std::max(weights[x], weights[x + (h - 1)*w]);
        for (int y = h - 1; y > 0; --y)
            weights[x + y*w] = std::max(weights[x + y*w], weights[x + (y - 1)*w]);
    }

    pdf.swap(weights);
    cdf.assign(pdf.size() + h, 0.0f);
    marginalPdf.assign(h, 0.0f);
    marginalCdf.assign(h + 1, 0.0f);
    marginalCdf[0] = 0.0f;
    for (int y = 0; y < h; ++y) {
        int idxP = y*w;
        int idxC = y*(w + 1);
        cdf[idxC] = 0.0f;
        for (int x = 0; x < w; ++x, ++idxP, ++idxC) {
            marginalPdf[y] += pdf[idxP];
            cdf[idxC + 1] = cdf[idxC] + pdf[idxP];
        }
        marginalCdf[y + 1] = marginalCdf[y] + marginalPdf[y];
    }
    for (int y = 0; y < h; ++y) {
        int idxP = y*w;
        int idxC = y*(w + 1);
        int idxTail = idxC + w;
        float rowWeight = cdf[idxTail];
        if (rowWeight < 1e-4f) {
            for (int x = 0; x < w; ++x, ++idxP, ++idxC) {
                pdf[idxP] = 1.0f/w;
                cdf[idxC] = x/float(w);
            }
        } else {
            for (int x = 0; x < w; ++x, ++idxP, ++idxC) {
                pdf[idxP] /= rowWeight;
                cdf[idxC] /= rowWeight;
            }
        }
        cdf[idxTail] = 1.0f;
    }
    float totalWeight = marginalCdf.back();
    for (float &p : marginalPdf) p /= totalWeight;
    for (float &c : marginalCdf) c /= totalWeight;
    marginalCdf.back() = 1.0f;
    samplable = true;
}

// ------------------------------------------------------------------------------------------
// BSDFs
// ------------------------------------------------------------------------------------------
namespace {
// Measured complex IORs (subset of the table the reference ships in bsdfs/ComplexIorData.hpp;
// values are physical constants from the Schubert reference cited in bsdfs/ComplexIor.cpp:3).
struct ComplexIor { const char *name; float eta[3], k[3]; };
const ComplexIor complexIors[] = {
    {"Ag", {0.1552646489f, 0.1167232965f, 0.1383806959f}, {4.8283433224f, 3.1222459278f, 2.1469504455f}},
    {"Al", {1.6574599595f, 0.8803689579f, 0.5212287346f}, {9.2238691996f, 6.2695232477f, 4.8370012281f}},
    {"Au", {0.1431189557f, 0.3749570432f, 1.4424785571f}, {3.9831604247f, 2.3857207478f, 1.6032152899f}},
    {"Be", {4.1850592788f, 3.1850604423f, 2.7840913457f}, {3.8354398268f, 3.0101260162f, 2.8690088743f}},
    {"Cr", {4.3696828663f, 2.9167024892f, 1.6547005413f}, {5.2064337956f, 4.2313645277f, 3.7549467933f}},
    {"Cu", {0.2004376970f, 0.9240334304f, 1.1022119527f}, {3.9129485033f, 2.4528477015f, 2.1421879552f}},
    {"Hg", {2.3989314904f, 1.4400254917f, 0.9095512090f}, {6.3276269444f, 4.3719414152f, 3.4217899270f}},
    {"Ir", {3.0864098394f, 2.0821938440f, 1.6178866805f}, {5.5921510077f, 4.0671757150f, 3.2672611269f}},
    {"Li", {0.2657871942f, 0.1956102432f, 0.2209198538f}, {3.5401743407f, 2.3111306542f, 1.6685930000f}},
    {"Mo", {4.4837010280f, 3.5254578255f, 2.7760769438f}, {4.1111307988f, 3.4208716252f, 3.1506031404f}},
    {"Na", {0.0602665320f, 0.0561412435f, 0.0619909494f}, {3.1792906496f, 2.1124800781f, 1.5790940266f}},
    {"Nb", {3.4201353595f, 2.7901921379f, 2.3955856658f}, {3.4413817900f, 2.7376437930f, 2.5799132708f}},
    {"Ni", {2.3672753521f, 1.6633583302f, 1.4670554172f}, {4.4988329911f, 3.0501643957f, 2.3454274399f}},
    {"Rh", {2.5857954933f, 1.8601866068f, 1.5544279524f}, {6.7822927110f, 4.7029501026f, 3.9760892461f}},
    {"Ta", {2.0625846607f, 2.3930915569f, 2.6280684948f}, {2.4080467973f, 1.7413705864f, 1.9470377016f}},
    {"TiN", {1.6484691607f, 1.1504482522f, 1.3797795097f}, {3.3684596226f, 1.9434888540f, 1.1020123347f}},
    {"V", {4.2775126218f, 3.5131538236f, 2.7611257461f}, {3.4911844504f, 2.8893580874f, 3.1116965117f}},
    {"W", {4.3707029924f, 3.3002972445f, 2.9982666528f}, {3.5006778591f, 2.6048652781f, 2.2731930614f}},
};

bool lookupComplexIor(const std::string &name, Vec3f &eta, Vec3f &k)
{
    for (const ComplexIor &c : complexIors) {
        if (name == c.name) {
            eta = Vec3f(c.eta[0], c.eta[1], c.eta[2]);
            k   = Vec3f(c.k[0], c.k[1], c.k[2]);
            return true;
        }
    }
    return false;
}

int parseDistribution(const JsonValue &v, int dflt)
{
    const JsonValue &d = v["distribution"];
    if (!d) return dflt;
    const std::string &s = d.asString();
    if (s == "beckmann") return 0;
    if (s == "phong") return 1;
    if (s == "ggx") return 2;
    throw JsonLoadException("Invalid microfacet distribution: '" + s + "'");
}

// bsdfs/Fresnel.hpp:75-96
float dielectricReflectance(float eta, float cosThetaI)
{
    if (cosThetaI < 0.0f) {
        eta = 1.0f/eta;
        cosThetaI = -cosThetaI;
    }
    float sinThetaTSq = eta*eta*(1.0f - cosThetaI*cosThetaI);
    if (sinThetaTSq > 1.0f)
        return 1.0f;
    float cosThetaT = std::sqrt(std::max(1.0f - sinThetaTSq, 0.0f));
    float Rs = (eta*cosThetaI - cosThetaT)/(eta*cosThetaI + cosThetaT);
    float Rp = (eta*cosThetaT - cosThetaI)/(eta*cosThetaT + cosThetaI);
    return (Rs*Rs + Rp*Rp)*0.5f;
}

// bsdfs/Fresnel.hpp:140-153
float computeDiffuseFresnel(float ior, const int sampleCount)
{
    double diffuseFresnel = 0.0;
    float fb = dielectricReflectance(ior, 0.0f);
    for (int i = 1; i <= sampleCount; ++i) {
        float cosThetaSq = float(i)/sampleCount;
        float fa = dielectricReflectance(ior, std::min(std::sqrt(cosThetaSq), 1.0f));
        diffuseFresnel += double(fa + fb)*(0.5/sampleCount);
        fb = fa;
    }
    return float(diffuseFresnel);
}
} // namespace

void Bsdf::prepareForRender()
{
    if (prepared)
        return;
    prepared = true;
    if (sub0) sub0->prepareForRender();
    if (sub1) sub1->prepareForRender();

    enum { GR = 1, GT = 2, DR = 4, DT = 8, SR = 16, ST = 32, FWD = 128 };
    switch (type) {
    case Lambert:         lobes = DR; break;
    case Null:            lobes = 0; break;
    case RoughConductor:  lobes = GR; break;
    case SmoothCoat:      // SmoothCoatBsdf.cpp:218-223
        scaledSigmaA = thickness*sigmaA;
        avgTransmittance = std::exp(-2.0f*scaledSigmaA.avg());
        lobes = SR | sub0->lobes;
        break;
    case Dielectric:      lobes = enableRefraction ? (SR | ST) : SR; break;
    case RoughDielectric: lobes = enableRefraction ? (GR | GT) : GR; break;
    case Mirror:          lobes = SR; break;
    case Conductor:       lobes = SR; break;
    case Plastic:         // PlasticBsdf.cpp:179-185
        lobes = SR | DR;
        scaledSigmaA = thickness*sigmaA;
        avgTransmittance = std::exp(-2.0f*scaledSigmaA.avg());
        diffuseFresnel = computeDiffuseFresnel(ior, 1000000);
        break;
    case RoughPlastic:    // RoughPlasticBsdf.cpp:215-222
        lobes = GR | DR;
        scaledSigmaA = thickness*sigmaA;
        avgTransmittance = std::exp(-2.0f*scaledSigmaA.avg());
        diffuseFresnel = computeDiffuseFresnel(ior, 1000000);
        break;
    case Mixed:           lobes = sub0->lobes | sub1->lobes; break;
    case Transparency:    lobes = FWD | sub0->lobes; break;
    case Forward:         lobes = FWD; break;
    case Error:           lobes = DR; break;
    case DiffuseTransmission:   // DiffuseTransmissionBsdf.cpp:15-19: _transmittance = 0.5 (no JSON key reads it); TgHipBsdf::eta[0]
        lobes = DT | DR;
        eta = Vec3f(0.5f, 0.0f, 0.0f); k = Vec3f(0.0f);
        break;
    case Phong:           // PhongBsdf.cpp:126-132; TgHipBsdf: eta = {exponent, diffuse ratio, -}, k = {_invExponent, _pdfFactor, _brdfFactor}
        lobes = GR | DR;
        eta = Vec3f(exponent, diffuseRatio, 0.0f);
        k = Vec3f(1.0f/(1.0f + exponent), (exponent + 1.0f)*INV_TWO_PI, (exponent + 2.0f)*INV_TWO_PI);
        break;
    case ThinSheet:       lobes = SR | FWD; break;       // ThinSheetBsdf.cpp:20-27 (enableRefraction carries _enableInterference, tex1 the thickness)
    case OrenNayar:       lobes = DR; break;             // OrenNayarBsdf.cpp:18-22
    case RoughCoat:       // RoughCoatBsdf.cpp:300-305
        scaledSigmaA = thickness*sigmaA;
        avgTransmittance = std::exp(-2.0f*scaledSigmaA.avg());
        lobes = GR | sub0->lobes;
        break;
    }
}

static std::shared_ptr<Texture> constantTexture(float v)
{
    auto t = std::make_shared<Texture>();
    t->value = Vec3f(v);
    return t;
}

std::shared_ptr<Texture> Scene::fetchTexture(const JsonValue &v, bool rgb, bool autoAlpha) const
{
    // TextureCache::fetchTexture (io/TextureCache.cpp:15-39): bitmaps are shared by (file, texel conversion, gamma_correct, interpolate, clamp)
    // -- BitmapTexture::operator< (textures/BitmapTexture.hpp:153-163) does NOT look at "scale", so of two bitmaps that differ only in their
    // scale the one fetched first serves both.  Reproduced as it is.
    auto cached = [&](const std::shared_ptr<Texture> &t) -> std::shared_ptr<Texture> {
        const std::string key = t->path + (rgb ? "|rgb" : autoAlpha ? "|auto" : "|avg") + (t->gammaCorrect ? "|g" : "|-") + (t->linear ? "|l" : "|-") + (t->clamp ? "|c" : "|-");
        for (auto &kv : _textureCache)
            if (kv.first == key) return kv.second;
        _textureCache.emplace_back(key, t);
        return t;
    };
    if (v.isString()) {
        auto t = std::make_shared<Texture>();
        t->type = Texture::Bitmap;
        t->rgb = rgb;
        t->autoAlpha = autoAlpha && !rgb;
        t->path = _srcDir.empty() ? v.asString() : _srcDir + "/" + v.asString();
        return cached(t);
    } else if (v.isNumber()) {
        return constantTexture(v.asFloat());
    } else if (v.isArray()) {
        auto t = std::make_shared<Texture>();
        if (v.size() != 3) throw JsonLoadException("JSON: expecting an RGB triple for a texture");
        t->value = Vec3f(v[0].asFloat(), v[1].asFloat(), v[2].asFloat());
        return t;
    } else if (v.isObject()) {
        std::string type = v["type"].asString();
        auto t = std::make_shared<Texture>();
        if (type == "constant") {
            Vec3f c(1.0f);
            getVec3(v, "value", c);
            t->value = c;
        } else if (type == "checker") {
            t->type = Texture::Checker;
            getVec3(v, "on_color", t->onColor);
            getVec3(v, "off_color", t->offColor);
            v.getField("res_u", t->resU);
            v.getField("res_v", t->resV);
        } else if (type == "bitmap") {
            t->type = Texture::Bitmap;
            t->rgb = rgb;
            t->autoAlpha = autoAlpha && !rgb;
            std::string file;
            if (v.getField("file", file))
                t->path = _srcDir.empty() ? file : _srcDir + "/" + file;
            v.getField("gamma_correct", t->gammaCorrect);
            v.getField("interpolate", t->linear);
            v.getField("clamp", t->clamp);
            v.getField("scale", t->scale);
            return cached(t);
        } else {
            throw JsonLoadException("Texture type '" + type + "' is outside the path_tracer_hip hot-path scope");
        }
        return t;
    }
    throw JsonLoadException("Type mismatch: Expecting a texture here");
}

std::shared_ptr<Bsdf> Scene::instantiateBsdf(const JsonValue &v) const
{
    auto b = std::make_shared<Bsdf>();
    std::string type = v["type"].asString();
    v.getField("name", b->name);
    // Bsdf::fromJson (Bsdf.cpp:19-25): scalar request.  A constant bump texture changes nothing (Primitive::setupTangentFrame ignores it,
    // Primitive.cpp:128-131); a varying one sends the shading frame through the primitive's tangent space and the map's derivatives (:133-162)
    if (const JsonValue &albedo = v["albedo"]) b->albedo = fetchTexture(albedo, true);
    else b->albedo = constantTexture(1.0f);
    if (const JsonValue &bump = v["bump"]) {
        std::shared_ptr<Texture> t = fetchTexture(bump, false);
        if (t && t->type != Texture::Constant)
            b->bump = t;
    }

    auto parseConductor = [&]() {
        Vec3f eta, k;
        bool explicitEtaK = getVec3(v, "eta", eta) && getVec3(v, "k", k);
        if (explicitEtaK) { b->eta = eta; b->k = k; }
        std::string material;
        if (v.getField("material", material) && !lookupComplexIor(material, b->eta, b->k))
            throw JsonLoadException("Unable to find material with name '" + material + "'");
        if (!explicitEtaK && material.empty())
            lookupComplexIor("Cu", b->eta, b->k);   // default material name "Cu" (RoughConductorBsdf.cpp:17-25)
    };

    if (type == "lambert") {
        b->type = Bsdf::Lambert;
    } else if (type == "null") {
        b->type = Bsdf::Null;
    } else if (type == "forward") {
        b->type = Bsdf::Forward;
    } else if (type == "mirror") {
        b->type = Bsdf::Mirror;
    } else if (type == "rough_conductor") {     // RoughConductorBsdf.cpp:31-43
        b->type = Bsdf::RoughConductor;
        parseConductor();
        b->distribution = parseDistribution(v, 2);
        if (const JsonValue &r = v["roughness"]) b->roughness = fetchTexture(r, false);
        else b->roughness = constantTexture(0.1f);
    } else if (type == "conductor") {           // ConductorBsdf.cpp:33-40
        b->type = Bsdf::Conductor;
        parseConductor();
    } else if (type == "smooth_coat") {         // SmoothCoatBsdf.cpp:12-28
        b->type = Bsdf::SmoothCoat;
        b->ior = 1.3f;
        v.getField("ior", b->ior);
        v.getField("thickness", b->thickness);
        getVec3(v, "sigma_a", b->sigmaA);
        if (const JsonValue &s = v["substrate"]) {
            b->sub0 = fetchBsdf(s);
        } else {
            b->sub0 = std::make_shared<Bsdf>();
            b->sub0->type = Bsdf::RoughConductor;
            b->sub0->albedo = constantTexture(1.0f);
            b->sub0->roughness = constantTexture(0.1f);
        }
    } else if (type == "dielectric") {
        b->type = Bsdf::Dielectric;
        v.getField("ior", b->ior);
        v.getField("enable_refraction", b->enableRefraction);
    } else if (type == "rough_dielectric") {
        b->type = Bsdf::RoughDielectric;
        v.getField("ior", b->ior);
        b->distribution = parseDistribution(v, 2);
        v.getField("enable_refraction", b->enableRefraction);
        if (const JsonValue &r = v["roughness"]) b->roughness = fetchTexture(r, false);
        else b->roughness = constantTexture(0.1f);
    } else if (type == "plastic") {
        b->type = Bsdf::Plastic;
        v.getField("ior", b->ior);
        v.getField("thickness", b->thickness);
        getVec3(v, "sigma_a", b->sigmaA);
    } else if (type == "rough_plastic") {
        b->type = Bsdf::RoughPlastic;
        v.getField("ior", b->ior);
        b->distribution = parseDistribution(v, 2);
        v.getField("thickness", b->thickness);
        getVec3(v, "sigma_a", b->sigmaA);
        if (const JsonValue &r = v["roughness"]) b->roughness = fetchTexture(r, false);
        else b->roughness = constantTexture(0.02f);
    } else if (type == "mixed") {
        b->type = Bsdf::Mixed;
        if (!v["bsdf0"] || !v["bsdf1"]) throw JsonLoadException("mixed bsdf requires bsdf0 and bsdf1");
        b->sub0 = fetchBsdf(v["bsdf0"]);
        b->sub1 = fetchBsdf(v["bsdf1"]);
        if (const JsonValue &r = v["ratio"]) b->tex1 = fetchTexture(r, false);
        else b->tex1 = constantTexture(0.5f);
    } else if (type == "diffuse_transmission") {   // DiffuseTransmissionBsdf.cpp (no fromJson of its own)
        b->type = Bsdf::DiffuseTransmission;
    } else if (type == "phong") {                  // PhongBsdf.cpp:24-29
        b->type = Bsdf::Phong;
        v.getField("exponent", b->exponent);
        v.getField("diffuse_ratio", b->diffuseRatio);
    } else if (type == "thinsheet") {              // ThinSheetBsdf.cpp:29-37
        b->type = Bsdf::ThinSheet;
        b->enableRefraction = false;               // _enableInterference
        v.getField("ior", b->ior);
        v.getField("enable_interference", b->enableRefraction);
        getVec3(v, "sigma_a", b->sigmaA);
        if (const JsonValue &t = v["thickness"]) b->tex1 = fetchTexture(t, false);
        else b->tex1 = constantTexture(0.5f);
    } else if (type == "oren_nayar") {             // OrenNayarBsdf.cpp:24-30
        b->type = Bsdf::OrenNayar;
        if (const JsonValue &r = v["roughness"]) b->roughness = fetchTexture(r, false);
        else b->roughness = constantTexture(1.0f);
    } else if (type == "rough_coat") {             // RoughCoatBsdf.cpp:26-35
        b->type = Bsdf::RoughCoat;
        b->ior = 1.3f;
        v.getField("ior", b->ior);
        v.getField("thickness", b->thickness);
        getVec3(v, "sigma_a", b->sigmaA);
        b->distribution = parseDistribution(v, 2);
        if (const JsonValue &r = v["roughness"]) b->roughness = fetchTexture(r, false);
        else b->roughness = constantTexture(0.02f);
        if (const JsonValue &sub = v["substrate"]) {
            b->sub0 = fetchBsdf(sub);
        } else {
            b->sub0 = std::make_shared<Bsdf>();
            b->sub0->type = Bsdf::RoughConductor;
            b->sub0->albedo = constantTexture(1.0f);
            b->sub0->roughness = constantTexture(0.1f);
        }
    } else if (type == "transparency") {
        b->type = Bsdf::Transparency;
        if (const JsonValue &base = v["base"]) b->sub0 = fetchBsdf(base);
        else { b->sub0 = std::make_shared<Bsdf>(); b->sub0->albedo = constantTexture(1.0f); }
        if (const JsonValue &a = v["alpha"]) b->tex1 = fetchTexture(a, false, true);   // REQUEST_AUTO (TransparencyBsdf.cpp:31)
        else b->tex1 = constantTexture(1.0f);
    } else {
        throw JsonLoadException("BSDF type '" + type + "' is outside the path_tracer_hip hot-path scope");
    }
    return b;
}

std::shared_ptr<Bsdf> Scene::fetchBsdf(const JsonValue &v) const
{
    if (v.isString()) {
        for (const auto &b : bsdfs)
            if (b->name == v.asString()) return b;
        throw JsonLoadException("Unable to find an object with name '" + v.asString() + "'");
    } else if (v.isObject()) {
        return instantiateBsdf(v);
    }
    throw JsonLoadException("Type mismatch: Expecting either an object or an object reference here");
}

// ------------------------------------------------------------------------------------------
// Media: homogeneous medium with exponential transmittance and an isotropic / Henyey-Greenstein phase function
// (media/HomogeneousMedium.cpp:19-26, Medium.cpp:20-30); everything else is rejected by name
// ------------------------------------------------------------------------------------------
void Medium::prepareForRender()   // HomogeneousMedium.cpp:43-49, ExponentialMedium.cpp:52-59, AtmosphericMedium.cpp:66-84 (the pivot: TraceableScene::flatten)
{
    unitFalloffDirection = falloffDirection.normalized();
    effectiveFalloffScale = falloffScale/radius;
    sigmaA = materialSigmaA*density;
    sigmaS = materialSigmaS*density;
    sigmaT = sigmaA + sigmaS;
    absorptionOnly = sigmaS.x() == 0.0f && sigmaS.y() == 0.0f && sigmaS.z() == 0.0f;
}

std::shared_ptr<Medium> Scene::instantiateMedium(const JsonValue &v) const
{
    auto m = std::make_shared<Medium>();
    std::string type = v["type"].asString();
    if (type != "homogeneous" && type != "exponential" && type != "atmosphere")
        throw JsonLoadException("medium type '" + type + "' is not supported by path_tracer_hip (homogeneous, exponential and atmosphere only)");
    v.getField("name", m->name);
    if (type == "atmosphere") {                     // AtmosphericMedium::fromJson (AtmosphericMedium.cpp:25-38)
        m->mediumType = 2;
        v.getField("pivot", m->pivot);
        v.getField("falloff_scale", m->falloffScale);
        v.getField("radius", m->radius);
        getVec3(v, "center", m->center);
    }
    if (type == "exponential") {                    // ExponentialMedium::fromJson (ExponentialMedium.cpp:22-31)
        m->mediumType = 1;
        v.getField("falloff_scale", m->falloffScale);
        getVec3(v, "unit_point", m->unitPoint);
        getVec3(v, "falloff_direction", m->falloffDirection);
    }
    getVec3(v, "sigma_a", m->materialSigmaA);
    getVec3(v, "sigma_s", m->materialSigmaS);
    v.getField("density", m->density);
    v.getField("max_bounces", m->maxBounce);
    // scene.fetchTransmittance (Medium.cpp:27-28); defaults from the constructors of transmittances/*.cpp
    std::function<void(const JsonValue &, int &, float *, bool)> parseTransmittance = [&](const JsonValue &t, int &type, float *p, bool nested) {
        std::string tt = t.isString() ? t.asString() : t["type"].asString();
        const bool obj = t.isObject();
        if (tt == "exponential") {
            type = 0;
        } else if (tt == "linear" || tt == "quadratic") {                       // {Linear,Quadratic}Transmittance.cpp:12-22
            type = tt == "linear" ? 1 : 2;
            p[0] = 1.0f;
            if (tt == "quadratic") p[0] = 0.75f;
            if (obj) t.getField("max_t", p[0]);
        } else if (tt == "double_exponential") {                                // DoubleExponentialTransmittance.cpp:12-23
            type = 3;
            p[0] = 0.5f; p[1] = 10.0f;
            if (obj) { t.getField("sigma_a", p[0]); t.getField("sigma_b", p[1]); }
        } else if (tt == "pulse") {                                             // PulseTransmittance.cpp:12-26
            type = 4;
            p[0] = 0.0f; p[1] = 1.0f; p[2] = 4.0f;
            int numPulses = 4;
            if (obj) { t.getField("min", p[0]); t.getField("max", p[1]); t.getField("num_pulses", numPulses); }
            p[2] = float(numPulses);
        } else if (tt == "erlang") {                                            // ErlangTransmittance.cpp:12-21
            type = 5;
            p[0] = 5.0f;
            if (obj) t.getField("rate", p[0]);
        } else if (tt == "davis") {                                             // DavisTransmittance.cpp:7-25
            type = 6;
            p[0] = 1.1f;
            if (obj) t.getField("alpha", p[0]);
            if (p[0] < 1 + 1e-6f) p[0] = 1 + 1e-6f;
        } else if (tt == "davis_weinstein") {                                   // DavisWeinsteinTransmittance.cpp:9-29
            type = 7;
            p[0] = 0.75f; p[1] = 1.0f;
            if (obj) { t.getField("h", p[0]); t.getField("c", p[1]); }
            p[0] = std::min(std::max(p[0], 0.5f), 1.0f);
        } else if (tt == "interpolated" && !nested) {                         // InterpolatedTransmittance.cpp:15-29
            type = 8;
            p[0] = 0.5f;
            if (obj) {
                t.getField("ratio", p[0]);
                if (const JsonValue &a = t["tr_a"]) parseTransmittance(a, m->subType[0], m->subP[0], true);
                if (const JsonValue &b = t["tr_b"]) parseTransmittance(b, m->subType[1], m->subP[1], true);
            }
        } else {
            throw JsonLoadException("transmittance '" + tt + "' is not supported by path_tracer_hip");
        }
    };
    if (const JsonValue &t = v["transmittance"])
        parseTransmittance(t, m->transType, m->transP, false);
    if (m->mediumType != 0 && m->transType != 0)
        throw JsonLoadException("an exponential or atmospheric medium with a non-exponential transmittance is not supported by path_tracer_hip (the "
                                "reference's ExponentialMedium / AtmosphericMedium::sampleDistance evaluate the transmittance with a flag they have not set yet)");
    if (const JsonValue &ph = v["phase_function"]) {
        std::string pt = ph.isString() ? ph.asString() : ph["type"].asString();
        if (pt == "isotropic") m->phaseType = 0;
        else if (pt == "henyey_greenstein") { m->phaseType = 1; if (ph.isObject()) ph.getField("g", m->phaseG); }
        else if (pt == "rayleigh") m->phaseType = 2;
        else throw JsonLoadException("phase function '" + pt + "' is not supported by path_tracer_hip");
    }
    return m;
}

std::shared_ptr<Medium> Scene::fetchMedium(const JsonValue &v) const
{
    if (v.isString()) {
        for (const auto &m : media)
            if (m->name == v.asString()) return m;
        throw JsonLoadException("Unable to find an object with name '" + v.asString() + "'");
    } else if (v.isObject()) {
        return instantiateMedium(v);
    }
    throw JsonLoadException("Type mismatch: Expecting either an object or an object reference here");
}

std::shared_ptr<Primitive> Scene::fetchPrimitive(const JsonValue &v) const
{
    if (v.isString()) {
        for (const auto &p : primitives)
            if (p->name == v.asString()) return p;
        throw JsonLoadException("Unable to find an object with name '" + v.asString() + "'");
    } else if (v.isObject()) {
        return instantiatePrimitive(v);
    }
    throw JsonLoadException("Type mismatch: Expecting either an object or an object reference here");
}

// ------------------------------------------------------------------------------------------
// Primitives
// ------------------------------------------------------------------------------------------
bool Primitive::isEmissive() const
{
    return (emission && emission->maximum().max() > 0.0f) || (power && power->maximum().max() > 0.0f);
}

float Primitive::powerToRadianceFactor() const
{
    switch (type) {
    case InfiniteSphere: return INV_FOUR_PI;           // InfiniteSphere.cpp:59-62
    case Skydome:        return INV_FOUR_PI;           // Skydome.cpp:63-66
    case InfiniteSphereCap: return INV_TWO_PI/(1.0f - scale[0]);   // InfiniteSphereCap.cpp:36-39
    case Point:          return INV_FOUR_PI;           // Point.cpp:25-28
    default:             return INV_PI*invArea;        // Quad.cpp:50-53, Cube.cpp, TriangleMesh.cpp:108-111
    }
}

std::shared_ptr<Primitive> Scene::instantiatePrimitive(const JsonValue &v) const
{
    auto p = std::make_shared<Primitive>();
    std::string type = v["type"].asString();
    v.getField("name", p->name);
    getTransform(v, "transform", p->transform);
    if (const JsonValue &e = v["emission"]) p->emission = fetchTexture(e, true);
    if (const JsonValue &pw = v["power"]) p->power = fetchTexture(pw, true);
    if (const JsonValue &m = v["int_medium"]) p->intMedium = fetchMedium(m);   // Primitive.cpp:30-31
    if (const JsonValue &m = v["ext_medium"]) p->extMedium = fetchMedium(m);

    auto defaultBsdf = [&]() {   // Primitive::_defaultBsdf = LambertBsdf (Primitive.cpp:11)
        auto b = std::make_shared<Bsdf>();
        b->albedo = constantTexture(1.0f);
        return b;
    };

    if (type == "mesh") {
        p->type = Primitive::Mesh;
        v.getField("file", p->file);
        v.getField("smooth", p->smooth);
        v.getField("backface_culling", p->backfaceCulling);
        v.getField("recompute_normals", p->recomputeNormals);
        if (const JsonValue &b = v["bsdf"]) {
            if (b.isArray())
                for (size_t i = 0; i < b.size(); ++i) p->bsdfs.push_back(fetchBsdf(b[i]));
            else
                p->bsdfs.push_back(fetchBsdf(b));
        } else {
            p->bsdfs.push_back(defaultBsdf());
        }
    } else if (type == "quad" || type == "cube" || type == "sphere") {
        p->type = type == "quad" ? Primitive::Quad : type == "cube" ? Primitive::Cube : Primitive::Sphere;
        if (const JsonValue &b = v["bsdf"]) p->bsdfs.push_back(fetchBsdf(b));
        else p->bsdfs.push_back(defaultBsdf());
    } else if (type == "disk") {          // Disk::fromJson (Disk.cpp:43-50)
        p->type = Primitive::Disk;
        v.getField("cone_angle", p->coneAngle);
        if (const JsonValue &b = v["bsdf"]) p->bsdfs.push_back(fetchBsdf(b));
        else p->bsdfs.push_back(defaultBsdf());
    } else if (type == "cylinder") {      // Cylinder::fromJson (Cylinder.cpp:41-48)
        p->type = Primitive::Cylinder;
        v.getField("capped", p->capped);
        if (const JsonValue &b = v["bsdf"]) p->bsdfs.push_back(fetchBsdf(b));
        else p->bsdfs.push_back(defaultBsdf());
    } else if (type == "infinite_sphere") {
        p->type = Primitive::InfiniteSphere;
        v.getField("sample", p->doSample);
    } else if (type == "skydome") {                   // Skydome::fromJson (Skydome.cpp:68-77)
        p->type = Primitive::Skydome;
        v.getField("temperature", p->skyTemperature);
        v.getField("turbidity", p->skyTurbidity);
        v.getField("intensity", p->skyIntensity);
        v.getField("sample", p->doSample);
        if (p->power)
            throw JsonLoadException("a skydome with 'power' is outside the path_tracer_hip hot-path scope");
    } else if (type == "point") {                     // Point::fromJson (Point.cpp:30-33)
        p->type = Primitive::Point;
    } else if (type == "infinite_sphere_cap") {      // InfiniteSphereCap::fromJson (InfiniteSphereCap.cpp:41-50)
        p->type = Primitive::InfiniteSphereCap;
        v.getField("sample", p->doSample);
        v.getField("cap_angle", p->capAngleDeg);
        if (v["skydome"])
            throw JsonLoadException("infinite_sphere_cap 'skydome' pivots (procedural sky) are outside the path_tracer_hip hot-path scope");
    } else if (type == "instances") {
        // Instance::fromJson (Instance.cpp:60-93)
        p->type = Primitive::Instances;
        if (p->emission || p->power)
            throw JsonLoadException("emissive 'instances' primitives are outside the path_tracer_hip hot-path scope");
        if (const JsonValue &m = v["masters"])
            for (size_t i = 0; i < m.size(); ++i)
                p->masters.push_back(fetchPrimitive(m[i]));
        for (const auto &m : p->masters)
            if (m->type != Primitive::Mesh)
                throw JsonLoadException("'instances' masters other than triangle meshes are outside the path_tracer_hip hot-path scope");
        if (v["instancesA"] || v["instancesB"])
            throw JsonLoadException("interpolated instance files (instancesA/instancesB) are outside the path_tracer_hip hot-path scope");
        if (const JsonValue &inst = v["instances"]) {
            if (inst.isString()) {
                p->instanceFile = inst.asString();
            } else {
                for (size_t i = 0; i < inst.size(); ++i) {
                    unsigned id = 0;
                    inst[i].getField("id", id);
                    Mat4f transform;
                    getTransform(inst[i], "transform", transform);
                    p->instanceId.push_back(uint8_t(id));
                    p->instancePos.push_back(transform.translation());                                  // extractTranslationVec
                    p->instanceRot.push_back(QuaternionF::fromMatrix(transform.extractRotation()));
                }
            }
        }
    } else {
        throw JsonLoadException("Primitive type '" + type + "' is outside the path_tracer_hip hot-path scope");
    }
    return p;
}

// Instance.cpp:205-232 (loadInstances) with loadLossyInstance / loadLosslessInstance (:133-171)
static void loadInstanceFile(const std::string &path, std::vector<Vec3f> &pos, std::vector<QuaternionF> &rot, std::vector<uint8_t> &ids)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f)
        throw std::runtime_error("Unable to load instances at '" + path + "'");
    auto rd = [&](void *dst, size_t n) { if (std::fread(dst, 1, n, f) != n) { std::fclose(f); throw std::runtime_error("Truncated instance file '" + path + "'"); } };
    uint32_t count = 0, compressed = 0;
    float b[6];
    rd(&count, 4); rd(&compressed, 4); rd(b, 24);     // Box3f = min, max
    Vec3f lo(b[0], b[1], b[2]), hi(b[3], b[4], b[5]);
    pos.resize(count); rot.resize(count); ids.resize(count);
    for (uint32_t i = 0; i < count; ++i) {
        if (compressed & 1u) {
            uint32_t a, bb, c;
            rd(&a, 4); rd(&bb, 4); rd(&c, 4);
            const uint32_t mask = (1u << 21) - 1;
            uint32_t x = a >> 11, y = ((a << 10) | (bb >> 22)) & mask, z = (bb >> 1) & mask;
            uint32_t r = c & 255u, axisX = (c >> 8) & 4095u, axisY = (c >> 20) & 4095u;
            float axisXf = (axisX/float(1 << 12))*2.0f - 1.0f, axisYf = (axisY/float(1 << 12))*2.0f - 1.0f;
            float rotW = TWO_PI*r/(1 << 8);
            Vec3f w(axisXf, axisYf, std::sqrt(std::max(1 - axisXf*axisXf - axisYf*axisYf, 0.0f)));
            Vec3f t = Vec3f(float(x), float(y), float(z))/float(1 << 21);
            pos[i] = lo*(Vec3f(1.0f) - t) + hi*t;                                                   // lerp (MathUtil.hpp:60-64)
            rot[i] = QuaternionF(rotW, w);
        } else {
            float pw[6];
            rd(pw, 24);
            pos[i] = Vec3f(pw[0], pw[1], pw[2]);
            Vec3f w(pw[3], pw[4], pw[5]);
            float angle = w.length();
            w = angle > 0 ? w/angle : Vec3f(0.0f, 1.0f, 0.0f);
            rot[i] = QuaternionF(angle, w);
        }
    }
    rd(ids.data(), count);
    std::fclose(f);
}

void Primitive::loadResources(const std::string &sceneDir)
{
    if (type == Instances) {
        // Instance::loadResources (Instance.cpp:265-282) reads the instance file only -- it does NOT forward to its masters,
        // and Scene::loadResources (io/Scene.cpp:281-293) only walks the scene's own primitives, so a master mesh named in
        // JSON never gets its triangles in an unmodified reference (Instance works there for masters built in memory).  Here
        // the masters are loaded; oracle/ref_harness.cpp does the same on the reference's objects before rendering goldens.
        for (auto &m : masters)
            m->loadResources(sceneDir);
        if (!instanceFile.empty())
            loadInstanceFile(sceneDir.empty() ? instanceFile : sceneDir + "/" + instanceFile, instancePos, instanceRot, instanceId);
        for (uint8_t id : instanceId)
            if (id >= masters.size())
                throw std::runtime_error("instance refers to master " + std::to_string(int(id)) + " of " + std::to_string(masters.size()));
        return;
    }
    if (type != Mesh || file.empty())
        return;
    std::string full = sceneDir.empty() ? file : sceneDir + "/" + file;
    std::string err;
    if (!MeshIO::load(full, verts, tris, err))
        throw std::runtime_error("Unable to load triangle mesh at '" + full + "': " + err);
    if (recomputeNormals)
        MeshIO::recomputeNormals(verts, tris);
}


// The reference bounds an instance by the eight rotated corners of its master's box (Instance.cpp:411-423): for a
// rotated master that box is up to sqrt(3) wider per axis than the geometry.  That box (instanceRefBounds) decides which instances the
// reference hands a ray to, and so what the ray hits; the box of the master's rotated VERTICES (instanceBounds: exact up to rounding, padded
// for the rounding of the device's own world -> master transform, never larger than the reference's) only says which of those instances
// the ray can hit at all: an instance whose tight box the ray misses is not walked (pt_kernels.h), and the wide BVH of the any-hit queries
// is built from the tight boxes.  `bounds` (the primitive's, which the scene bounds are made of) stays the reference's.
Box3f tightInstanceBox(const float *vertexPositions, size_t strideFloats, size_t numVertices, const QuaternionF &q, const Vec3f &pos, const Box3f &refBox)
{
    const Vec3f cx = q*Vec3f(1.0f, 0.0f, 0.0f), cy = q*Vec3f(0.0f, 1.0f, 0.0f), cz = q*Vec3f(0.0f, 0.0f, 1.0f);
    Vec3f lo(std::numeric_limits<float>::max()), hi(-std::numeric_limits<float>::max());
    for (size_t vi = 0; vi < numVertices; ++vi) {
        const float *v = vertexPositions + vi*strideFloats;
        Vec3f p = cx*v[0] + cy*v[1] + cz*v[2];
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
    }
    Box3f tight;
    for (int k = 0; k < 3; ++k) {
        float pad = 1e-5f*(std::max(std::fabs(lo[k]), std::fabs(hi[k])) + std::fabs(pos[k])) + 1e-6f*(hi[k] - lo[k]);
        tight.lo[k] = std::max(lo[k] + pos[k] - pad, refBox.lo[k]);
        tight.hi[k] = std::min(hi[k] + pos[k] + pad, refBox.hi[k]);
    }
    return tight;
}

void Primitive::tightenInstanceBounds()
{
    const size_t n = instancePos.size();
    auto work = [this](size_t begin, size_t end) {
        for (size_t i = begin; i < end; ++i) {
            const Primitive &m = *masters[instanceId[i]];
            if (m.tris.empty() || m.tfVerts.empty())
                continue;
            static_assert(sizeof(MeshVertex) % sizeof(float) == 0, "MeshVertex is a record of floats");
            instanceBounds[i] = tightInstanceBox(m.tfVerts[0].pos, sizeof(MeshVertex)/sizeof(float), m.tfVerts.size(), instanceRot[i], instancePos[i], instanceBounds[i]);
        }
    };
    size_t cost = 0;
    for (size_t i = 0; i < n; ++i)
        cost += masters[instanceId[i]]->tfVerts.size();
    unsigned threads = cost > (1u << 22) ? std::max(1u, std::min(16u, std::thread::hardware_concurrency())) : 1u;
    if (threads == 1) {
        work(0, n);
        return;
    }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back(work, n*t/threads, n*(t + 1)/threads);
    for (std::thread &t : pool)
        t.join();
}

void Primitive::prepareForRender()
{
    switch (type) {
    case Quad: {   // Quad.cpp:298-316
        base = transform*Vec3f(0.0f);
        edge0 = transform.transformVector(Vec3f(1.0f, 0.0f, 0.0f));
        edge1 = transform.transformVector(Vec3f(0.0f, 0.0f, 1.0f));
        base -= edge0*0.5f;
        base -= edge1*0.5f;
        Vec3f n = edge1.cross(edge0);
        area = n.length();
        invArea = 1.0f/area;
        n /= area;
        normal = n;
        invUvSq[0] = 1.0f/edge0.lengthSq();
        invUvSq[1] = 1.0f/edge1.lengthSq();
        bounds = Box3f();
        bounds.grow(base); bounds.grow(base + edge0); bounds.grow(base + edge1); bounds.grow(base + edge0 + edge1);
        break;
    } case Cube: { // Cube.cpp:353-370
        pos = transform*Vec3f(0.0f);
        scale = Mat4f::scale(transform.extractScaleVec())*Vec3f(0.5f);
        rot = transform.extractRotation();
        invRot = rot.transpose();
        faceCdf = 4.0f*Vec3f(scale.y()*scale.z(), scale.z()*scale.x(), scale.x()*scale.y());
        faceCdf[1] += faceCdf[0];
        faceCdf[2] += faceCdf[1];
        area = 2.0f*faceCdf[2];
        invArea = 1.0f/area;
        bounds = Box3f();
        for (int i = 0; i < 8; ++i)
            bounds.grow(pos + rot*Vec3f((i & 1 ? scale.x() : -scale.x()), (i & 2 ? scale.y() : -scale.y()), (i & 4 ? scale.z() : -scale.z())));
        break;
    } case Sphere: { // Sphere.cpp:285-295
        pos = transform*Vec3f(0.0f);
        float radius = (Mat4f::scale(transform.extractScaleVec())*Vec3f(1.0f)).max();
        scale = Vec3f(radius);
        rot = transform.extractRotation();
        invRot = rot.transpose();
        area = 4.0f*PI*radius*radius;
        invArea = 1.0f/area;
        bounds = Box3f();
        bounds.grow(pos - Vec3f(radius)); bounds.grow(pos + Vec3f(radius));
        break;
    } case Disk: { // Disk.cpp:303-315, bounds :283-291
        pos = transform*Vec3f(0.0f);
        float r = (Mat4f::scale(transform.extractScaleVec())*Vec3f(1.0f, 0.0f, 1.0f)).max();
        normal = transform.transformVector(Vec3f(0.0f, 1.0f, 0.0f)).normalized();
        area = r*r*PI;
        invArea = 1.0f/area;
        {   // TangentFrame(n) (math/TangentFrame.hpp:22-31)
            float sign = copysignf(1.0f, normal.z());
            const float a = -1.0f/(sign + normal.z());
            const float b = normal.x()*normal.y()*a;
            edge0 = Vec3f(1.0f + sign*normal.x()*normal.x()*a, sign*b, -sign*normal.x());   // tangent
            edge1 = Vec3f(b, sign + normal.y()*normal.y()*a, -normal.y());                  // bitangent
        }
        float coneRad = coneAngle*(PI/180.0f);
        scale = Vec3f(r, std::cos(coneRad), 0.0f);
        base = pos - normal/std::sin(coneRad);                                              // _coneBase
        bounds = Box3f();
        bounds.grow(pos - edge0*r - edge1*r); bounds.grow(pos + edge0*r - edge1*r);
        bounds.grow(pos + edge0*r + edge1*r); bounds.grow(pos - edge0*r + edge1*r);
        break;
    } case Cylinder: { // Cylinder.cpp:305-319, bounds :286-293
        rot = transform.extractRotation();
        invRot = rot.transpose();
        pos = transform*Vec3f(0.0f);
        normal = transform.up().normalized();                                               // _axis
        Vec3f sc = transform.extractScaleVec();
        float radius = 0.5f*std::max(sc.x(), sc.z()), halfHeight = 0.5f*sc.y();
        scale = Vec3f(radius, halfHeight, capped ? 1.0f : 0.0f);
        area = 2.0f*PI*radius*radius + 2.0f*PI*radius*2.0f*halfHeight;
        invArea = 1.0f/area;
        bounds = Box3f();
        bounds.grow(pos + normal*halfHeight); bounds.grow(pos - normal*halfHeight);
        bounds.lo = bounds.lo - Vec3f(radius); bounds.hi = bounds.hi + Vec3f(radius);       // Box::grow(float)
        break;
    } case InfiniteSphere: { // InfiniteSphere.cpp:280-286
        rot = transform.extractRotation();
        invRot = rot.transpose();
        break;
    } case Skydome: {        // Skydome.cpp:279-306: the sky image, baked for the direction the transform turns "up" into
        rot = Mat4f();                                   // directions map to the image unrotated (Skydome.cpp:41-60)
        invRot = Mat4f();
        const Vec3f sun = transform.transformVector(Vec3f(0.0f, 1.0f, 0.0f));
        const float sunDir[3] = {sun.x(), sun.y(), sun.z()};
        auto sky = std::make_shared<Texture>();
        sky->type = Texture::Bitmap;
        sky->w = SkydomeSizeX; sky->h = SkydomeSizeY;
        sky->rgb = true; sky->linear = true; sky->clamp = false; sky->valid = true;   // BitmapTexture(img, 512, 256, RGB_HDR, true, false)
        sky->texels = bakeSkydomeImage(sunDir, skyTemperature, skyTurbidity, skyIntensity);
        sky->finishBitmap();
        emission = sky;
        break;
    } case Point: {           // Point.cpp:183-189
        pos = transform.translation();
        // Point::_power (a Vec3f hiding Primitive::_power) is taken from _emission BEFORE Primitive::prepareForRender turns
        // a "power" texture into _emission (Point.cpp:186-188): a point light given by "power" keeps _power = 0, so
        // approximateRadiance weighs it 0 and chooseLight never picks it. Reproduced as it is.
        scale = (emission && !power) ? emission->average()*(4.0f*PI) : Vec3f(0.0f);
        break;
    } case InfiniteSphereCap: { // InfiniteSphereCap.cpp:233-249
        normal = transform.transformVector(Vec3f(0.0f, 1.0f, 0.0f)).normalized();      // _capDir
        scale = Vec3f(std::cos(capAngleDeg*(PI/180.0f)), 0.0f, 0.0f);                  // _cosCapAngle
        float sign = copysignf(1.0f, normal.z());                                     // TangentFrame(_capDir)
        const float a = -1.0f/(sign + normal.z());
        const float b = normal.x()*normal.y()*a;
        edge0 = Vec3f(1.0f + sign*normal.x()*normal.x()*a, sign*b, -sign*normal.x());
        edge1 = Vec3f(b, sign + normal.y()*normal.y()*a, -normal.y());
        break;
    } case Instances: { // Instance.cpp:392-428
        for (auto &m : masters) {
            m->prepareForRender();
            for (auto &b : m->bsdfs)
                b->prepareForRender();
        }
        QuaternionF tRot = QuaternionF::fromMatrix(transform.extractRotation());
        bounds = Box3f();
        // (round 2 tightened these boxes to the rotated vertices for its own nearest-hit walk; the reference lets a ray into an instance by
        // ITS box -- the master box's eight rotated corners -- and what it lets in decides what the ray hits, so that is what is kept)
        instanceRefBounds.assign(instancePos.size(), Box3f());
        for (size_t i = 0; i < instancePos.size(); ++i) {
            instancePos[i] = transform*instancePos[i];
            instanceRot[i] = tRot*instanceRot[i];
            const Box3f &bLocal = masters[instanceId[i]]->bounds;
            if (masters[instanceId[i]]->tris.empty() || masters[instanceId[i]]->verts.empty())
                continue;
            Box3f bGlobal;
            for (int x = 0; x < 2; ++x)
                for (int y = 0; y < 2; ++y)
                    for (int z = 0; z < 2; ++z) {
                        Vec3f t = Vec3f(float(x), float(y), float(z));
                        bGlobal.grow(instancePos[i] + instanceRot[i]*(bLocal.lo*(Vec3f(1.0f) - t) + bLocal.hi*t));   // lerp(min, max, t)
                    }
            bounds.grow(bGlobal);
            instanceRefBounds[i] = bGlobal;
        }
        instanceBounds = instanceRefBounds;
        tightenInstanceBounds();
        break;
    } case Mesh: { // TriangleMesh.cpp:524-572
        bounds = Box3f();
        for (MeshTriangle &t : tris)
            t.material = std::min(std::max(t.material, 0), int(bsdfs.size()) - 1);
        tfVerts.resize(verts.size());
        Mat4f normalTform = transform.toNormalMatrix();
        for (size_t i = 0; i < verts.size(); ++i) {
            Vec3f p = transform*Vec3f(verts[i].pos[0], verts[i].pos[1], verts[i].pos[2]);
            Vec3f n = normalTform.transformVector(Vec3f(verts[i].normal[0], verts[i].normal[1], verts[i].normal[2]));
            for (int k = 0; k < 3; ++k) { tfVerts[i].pos[k] = p[k]; tfVerts[i].normal[k] = n[k]; }
            tfVerts[i].uv[0] = verts[i].uv[0]; tfVerts[i].uv[1] = verts[i].uv[1];
            bounds.grow(p);
        }
        area = 0.0f;
        for (const MeshTriangle &t : tris) {
            Vec3f p0(tfVerts[t.v0].pos[0], tfVerts[t.v0].pos[1], tfVerts[t.v0].pos[2]);
            Vec3f p1(tfVerts[t.v1].pos[0], tfVerts[t.v1].pos[1], tfVerts[t.v1].pos[2]);
            Vec3f p2(tfVerts[t.v2].pos[0], tfVerts[t.v2].pos[1], tfVerts[t.v2].pos[2]);
            area += (p1 - p0).cross(p2 - p0).length()*0.5f;   // MathUtil::triangleArea
        }
        invArea = 1.0f/area;
        break;
    }
    }
    // Primitive::prepareForRender (Primitive.cpp:102-108)
    if (power) {
        emission = std::make_shared<Texture>(*power);
        emission->scaleValues(powerToRadianceFactor());
    }
}

// ------------------------------------------------------------------------------------------
// Camera
// ------------------------------------------------------------------------------------------
Camera::Camera()
{
    // Camera::Camera(Mat4f(), Vec2u(1000, 563)) (Camera.cpp:16-35)
    pos = transform*Vec3f(0.0f, 0.0f, 2.0f);
    lookAt = transform*Vec3f(0.0f, 0.0f, -1.0f);
    up = transform*Vec3f(0.0f, 1.0f, 0.0f);
    transform.setRight(-transform.right());
    for (float &c : filterCdf) c = 0.0f;
    precompute();
}

void Camera::fromJson(const JsonValue &v, const Scene &scene)
{
    v.getField("tonemap", tonemap);
    const JsonValue &res = v["resolution"];
    if (res) {
        if (res.isNumber()) { resX = resY = unsigned(res.asDouble()); }
        else { resX = unsigned(res[0].asDouble()); resY = unsigned(res[1].asDouble()); }
    }
    if (const JsonValue &m = v["medium"]) medium = scene.fetchMedium(m);       // Camera.cpp:49-50
    v.getField("reconstruction_filter", filterName);

    if (v["transform"]) {   // Camera.cpp:55-66
        getTransform(v, "transform", transform);
        pos = transform.translation();
        lookAt = transform.fwd() + pos;
        up = transform.up();
        getVec3(v["transform"], "up", up);
        getVec3(v["transform"], "look_at", lookAt);
        transform.setRight(-transform.right());
    }
    std::string type = "pinhole";
    v.getField("type", type);
    if (type == "thinlens") {       // ThinlensCamera::fromJson (cameras/ThinlensCamera.cpp:52-66)
        thinlens = true;
        v.getField("focus_distance", focusDist);
        v.getField("aperture_size", apertureSize);
        v.getField("cateye", catEye);
        std::string focusPivot;
        if (v.getField("focus_pivot", focusPivot) && !focusPivot.empty()) {
            // ThinlensCamera::prepareForRender (cameras/ThinlensCamera.cpp:206-218): focus on the origin of the named primitive's frame
            const Primitive *pivot = nullptr;
            for (const auto &p : scene.primitives)
                if (p->name == focusPivot) { pivot = p.get(); break; }
            if (pivot)
                focusDist = (pivot->transform*Vec3f(0.0f) - pos).length();
            else
                std::fprintf(stderr, "Warning: Focus pivot '%s' for thinlens camera not found\n", focusPivot.c_str());
        }
        if (const JsonValue &ap = v["aperture"]) {
            std::string apType;
            if (ap.isObject() && ap.getField("type", apType) && apType == "blade") {
                // BladeTexture ctor + fromJson + init (textures/BladeTexture.cpp:14-41): the default angle belongs to the default 6 blades
                blades = 6;
                bladeAngle = 0.5f*PI/6;
                ap.getField("blades", blades);
                ap.getField("angle", bladeAngle);
                if (blades < 3)
                    throw JsonLoadException("a blade aperture needs at least 3 blades");
                bladeStep = TWO_PI/blades;
                float sinAngle = std::sin(bladeStep*0.5f), cosAngle = std::cos(bladeStep*0.5f);
                const float k = std::sin(PI/blades);                           // _baseEdge = Vec2f(-sin, cos)*2.0f*sin(pi/n)
                bladeEdge[0] = -sinAngle*2.0f*k;
                bladeEdge[1] = cosAngle*2.0f*k;
            } else if (!ap.isObject() || apType != "disk") {
                // ThinlensCamera::fromJson: scene.fetchTexture(aperture, TexelConversion::REQUEST_AVERAGE) (cameras/ThinlensCamera.cpp:62-63)
                apertureTex = scene.fetchTexture(ap, false);
                if (!apertureTex || apertureTex->type != Texture::Bitmap)
                    throw JsonLoadException("thinlens apertures other than the 'disk' and 'blade' textures and bitmaps are outside the path_tracer_hip hot-path scope");
            }
        }
    } else if (type == "equirectangular") {        // EquirectangularCamera::fromJson = Camera::fromJson (no parameters of its own)
        equirectangular = true;
    } else if (type == "cubemap") {                // CubemapCamera::fromJson (cameras/CubemapCamera.cpp:126-130; the constructor's mode is the horizontal cross)
        std::string mode = "horizontal_cross";
        v.getField("mode", mode);
        cubemapMode = mode == "horizontal_cross" ? 0 : mode == "vertical_cross" ? 1 : mode == "row" ? 2 : mode == "column" ? 3 : -1;
        if (cubemapMode < 0)
            throw JsonLoadException("Invalid projection mode: '" + mode + "'");
    } else if (type != "pinhole") {
        throw JsonLoadException("Camera type '" + type + "' is outside the path_tracer_hip hot-path scope");
    }
    v.getField("fov", fovDeg);
    precompute();
}

void Camera::precompute()
{
    ratio = resY/float(resX);
    pixelSizeX = 1.0f/resX;
    invTransform = transform.invert();             // Camera::precompute (cameras/Camera.cpp:37-42)
    float fovRad = fovDeg*(PI/180.0f);            // Angle::degToRad
    planeDist = 1.0f/std::tan(fovRad*0.5f);       // PinholeCamera.cpp:28-35

    // ReconstructionFilter::precompute (ReconstructionFilter.cpp:34-58)
    enum { RES = 31 };
    auto mitchell = [](float x) {
        const float B = 1.0f/3.0f, C = 1.0f/3.0f;
        if (x < 1.0f)
            return 1.0f/6.0f*((12.0f - 9.0f*B - 6.0f*C)*x*x*x + (-18.0f + 12.0f*B + 6.0f*C)*x*x + (6.0f - 2.0f*B));
        else if (x < 2.0f)
            return 1.0f/6.0f*((-B - 6.0f*C)*x*x*x + (6.0f*B + 30.0f*C)*x*x + (-12.0f*B - 48.0f*C)*x + (8.0f*B + 24.0f*C));
        return 0.0f;
    };
    auto catmull = [](float x) {
        if (x < 1.0f) return 1.0f/6.0f*((12.0f - 3.0f)*x*x*x + (-18.0f + 3.0f)*x*x + 6.0f);
        else if (x < 2.0f) return 1.0f/6.0f*(-3.0f*x*x*x + 15.0f*x*x - 24.0f*x + 12.0f);
        return 0.0f;
    };
    auto lanczos = [](float x) {
        if (x == 0.0f) return 1.0f;
        else if (x < 2.0f) return std::sin(PI*x)*std::sin(PI*x/2.0f)/(PI*PI*x*x/2.0f);
        return 0.0f;
    };
    std::function<float(float)> eval;
    if (filterName == "dirac")      { filterType = 0; filterWidth = 0.0f; }
    else if (filterName == "box")   { filterType = 1; filterWidth = 0.5f; }
    else if (filterName == "tent")  { filterType = 2; filterWidth = 1.0f; eval = [](float x) { return 1.0f - std::abs(x); }; }
    else if (filterName == "gaussian") { filterType = 2; filterWidth = 2.0f;
        eval = [](float x) { const float Alpha = 2.0f; return std::max(std::exp(-Alpha*x*x) - std::exp(-Alpha*4.0f), 0.0f); }; }
    else if (filterName == "mitchell_netravali") { filterType = 2; filterWidth = 2.0f; eval = [=](float x) { return mitchell(std::abs(x)); }; }
    else if (filterName == "catmull_rom") { filterType = 2; filterWidth = 2.0f; eval = [=](float x) { return catmull(std::abs(x)); }; }
    else if (filterName == "lanczos") { filterType = 2; filterWidth = 2.0f; eval = [=](float x) { return lanczos(std::abs(x)); }; }
    else throw JsonLoadException("Invalid reconstruction filter: '" + filterName + "'");

    filterBinSize = filterWidth/RES;
    for (float &c : filterCdf) c = 0.0f;
    if (filterType != 2)
        return;
    float filter[RES + 1];
    float filterSum = 0.0f;
    for (int i = 0; i < RES; ++i) {
        filter[i] = eval((i*filterWidth)/RES);
        filterSum += filter[i];
    }
    filterCdf[0] = 0.0f;
    for (int i = 1; i < RES; ++i)
        filterCdf[i] = filterCdf[i - 1] + filter[i - 1]/filterSum;
    filterCdf[RES] = 1.0f;
}

void RendererSettings::fromJson(const JsonValue &v)
{
    v.getField("output_file", outputFile);
    v.getField("hdr_output_file", hdrOutputFile);
    v.getField("resume_render_file", resumeRenderFile);
    v.getField("overwrite_output_files", overwriteOutputFiles);
    v.getField("adaptive_sampling", useAdaptiveSampling);
    v.getField("enable_resume_render", enableResumeRender);
    v.getField("stratified_sampler", useSobol);
    v.getField("scene_bvh", useSceneBvh);
    v.getField("spp", spp);
    v.getField("spp_step", sppStep);
    if (const JsonValue &o = v["output_buffers"]) {   // OutputBufferSettings::fromJson (OutputBufferSettings.cpp:23-30)
        static const char *names[5] = {"color", "depth", "normal", "albedo", "visibility"};
        for (size_t i = 0; i < o.size(); ++i) {
            OutputBufferSettings b;
            std::string type = o[i]["type"].asString();
            b.type = -1;
            for (int k = 0; k < 5; ++k)
                if (type == names[k]) b.type = k;
            if (b.type < 0)
                throw JsonLoadException("Unknown output buffer type '" + type + "'");
            o[i].getField("ldr_output_file", b.ldrOutputFile);
            o[i].getField("hdr_output_file", b.hdrOutputFile);
            o[i].getField("two_buffer_variance", b.twoBufferVariance);
            o[i].getField("sample_variance", b.sampleVariance);
            outputs.push_back(b);
        }
    }
}

void IntegratorSettings::fromJson(const JsonValue &v)
{
    v.getField("type", type);
    v.getField("min_bounces", minBounces);
    v.getField("max_bounces", maxBounces);
    v.getField("enable_consistency_checks", enableConsistencyChecks);
    v.getField("enable_two_sided_shading", enableTwoSidedShading);
    v.getField("enable_light_sampling", enableLightSampling);
    v.getField("enable_volume_light_sampling", enableVolumeLightSampling);
    v.getField("low_order_scattering", lowOrderScattering);
    v.getField("include_surfaces", includeSurfaces);
    v.getField("devices", devices);
    v.getField("share_devices", shareDevices);
}

// ------------------------------------------------------------------------------------------
// Scene
// ------------------------------------------------------------------------------------------
void Scene::fromJson(const JsonValue &root)
{
    // media first: primitives and the camera refer to them by name (Scene.cpp:236-253)
    if (const JsonValue &jm = root["media"])
        for (size_t i = 0; i < jm.size(); ++i)
            media.push_back(instantiateMedium(jm[i]));

    // bsdfs may reference earlier bsdfs by name (Scene.cpp:236-253 loads them in file order)
    if (const JsonValue &jb = root["bsdfs"])
        for (size_t i = 0; i < jb.size(); ++i)
            bsdfs.push_back(instantiateBsdf(jb[i]));
    if (const JsonValue &jp = root["primitives"])
        for (size_t i = 0; i < jp.size(); ++i)
            primitives.push_back(instantiatePrimitive(jp[i]));
    if (const JsonValue &cam = root["camera"])
        camera.fromJson(cam, *this);
    if (const JsonValue &integ = root["integrator"])
        integrator.fromJson(integ);
    if (const JsonValue &rend = root["renderer"])
        renderer.fromJson(rend);
}

void Scene::loadResources()
{
    for (auto &p : primitives)
        p->loadResources(_srcDir);
    for (auto &kv : _textureCache)
        if (!kv.second->valid)
            kv.second->loadBitmap(kv.second->path);
}

std::unique_ptr<Scene> Scene::load(const std::string &jsonPath)
{
    std::ifstream in(jsonPath.c_str(), std::ios::binary);
    if (!in)
        throw std::runtime_error("Unable to open file at '" + jsonPath + "'");
    std::stringstream ss;
    ss << in.rdbuf();
    JsonValue root = JsonValue::parse(ss.str());

    std::unique_ptr<Scene> scene(new Scene());
    size_t slash = jsonPath.find_last_of('/');
    scene->_srcDir = slash == std::string::npos ? std::string() : jsonPath.substr(0, slash);
    scene->fromJson(root);
    scene->loadResources();
    return scene;
}

} // namespace tungsten_amd
