#include "SkyModel.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>

#include <dlfcn.h>

namespace tungsten_amd {

namespace {

const int NumBands = 11;          // 320, 360, ... 720 nm
const int CoeffPerBand = 1080;    // 2 albedos x 10 turbidities x 6 control points x 9 coefficients
const int RadPerBand = 120;       // 2 x 10 x 6
const int CieSamples = 471;       // math/Spectral.hpp:18-20: 360 .. 830 nm at 1 nm
const float CieMin = 360.0f, CieMax = 830.0f;
const int NumSamples = 10;        // Skydome.cpp:257: wavelengths per texel

struct Tables {
    std::vector<double> coeff, rad;
    std::vector<float> cie;
};

const Tables &tables()
{
    static Tables t;
    static std::once_flag once;
    static std::string error;
    std::call_once(once, [] {
        const std::string path = skydomeTablesPath();
        std::FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) { error = "cannot open the skydome tables '" + path + "' (set TUNGSTEN_HIP_SKYDOME_TABLES)"; return; }
        char magic[8];
        t.coeff.resize(size_t(NumBands)*CoeffPerBand);
        t.rad.resize(size_t(NumBands)*RadPerBand);
        t.cie.resize(size_t(3)*CieSamples);
        bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "TGSKY001", 8) == 0 &&
                  std::fread(t.coeff.data(), sizeof(double), t.coeff.size(), f) == t.coeff.size() &&
                  std::fread(t.rad.data(), sizeof(double), t.rad.size(), f) == t.rad.size() &&
                  std::fread(t.cie.data(), sizeof(float), t.cie.size(), f) == t.cie.size();
        char extra;
        ok = ok && std::fread(&extra, 1, 1, f) == 0;
        std::fclose(f);
        if (!ok) error = "'" + path + "' is not a skydome table file of this version";
    });
    if (!error.empty())
        throw std::runtime_error(error);
    return t;
}

// The model's state for one sun position (thirdparty/skylight/ArHosekSkyModel.h: ArHosekSkyModelState, the fields the sky radiance needs)
struct SkyState {
    double config[NumBands][9];
    double radiance[NumBands];
    double skyCorrection[NumBands];
};

// One coefficient (or the mean radiance) at the given turbidity / albedo / elevation: quintic Bezier over the six control points in the
// cube root of the normalised elevation, bilinear between the two albedo tables and the two neighbouring integer turbidities
// (ArHosekSkyModel_CookConfiguration / _CookRadianceConfiguration, ArHosekSkyModel.cpp:147-289).  `stride` = values per control point.
double bezierTerm(const double *m, int i, int stride, double e)
{
    return std::pow(1.0 - e, 5.0)*m[i] +
           5.0*std::pow(1.0 - e, 4.0)*e*m[i + stride] +
           10.0*std::pow(1.0 - e, 3.0)*std::pow(e, 2.0)*m[i + 2*stride] +
           10.0*std::pow(1.0 - e, 2.0)*std::pow(e, 3.0)*m[i + 3*stride] +
           5.0*(1.0 - e)*std::pow(e, 4.0)*m[i + 4*stride] +
           std::pow(e, 5.0)*m[i + 5*stride];
}

void cookBand(const double *coeff, const double *rad, double turbidity, double albedo, double solarElevation, double config[9], double &radiance)
{
    const int intTurbidity = int(turbidity);
    const double turbidityRem = turbidity - double(intTurbidity);
    const double e = std::pow(solarElevation/(3.141592653589793/2.0), 1.0/3.0);

    // the four corners in the order the reference adds them: (albedo 0, low turbidity), (albedo 1, low), (albedo 0, high), (albedo 1, high)
    const double w[4] = {(1.0 - albedo)*(1.0 - turbidityRem), albedo*(1.0 - turbidityRem), (1.0 - albedo)*turbidityRem, albedo*turbidityRem};
    const int corners = intTurbidity == 10 ? 2 : 4;
    for (int c = 0; c < corners; ++c) {
        const int t = c < 2 ? intTurbidity - 1 : intTurbidity;
        const double *m = coeff + (c & 1)*9*6*10 + 9*6*t;
        for (int i = 0; i < 9; ++i) {
            const double v = w[c]*bezierTerm(m, i, 9, e);
            config[i] = c == 0 ? v : config[i] + v;
        }
        const double *r = rad + (c & 1)*6*10 + 6*t;
        const double v = w[c]*bezierTerm(r, 0, 1, e);
        radiance = c == 0 ? v : radiance + v;
    }
}

// Planck's law as the model's code writes it (art_blackbody_dd_value, ArHosekSkyModel.cpp:363-376)
double blackbody(double temperature, double lambda)
{
    const double c1 = 3.74177*10E-17, c2 = 0.0143878;
    return (c1/std::pow(lambda, 5.0))*(1.0/(std::exp(c2/(lambda*temperature)) - 1.0));
}

// arhosekskymodelstate_alienworld_alloc_init (ArHosekSkyModel.cpp:402-510), without the solar-disc radius it also derives
void initState(SkyState &s, double solarElevation, double solarIntensity, double temperature, double turbidity, double albedo)
{
    const Tables &t = tables();
    // the solar spectrum the model was fitted with (Preetham's, extended into the UV; ArHosekSkyModel.cpp:386-399)
    static const double originalSolarRadiance[NumBands] = {7500.0, 12500.0, 21127.5, 26760.5, 30663.7, 27825.0, 25503.8, 25134.2, 23212.1, 21526.7, 19870.8};
    const double blackbodyScale = 3.19992*10E-11;                      // blackbody_scaling_factor (:357)
    double sunCorrection[NumBands];
    for (int wl = 0; wl < NumBands; ++wl) {
        cookBand(t.coeff.data() + size_t(wl)*CoeffPerBand, t.rad.data() + size_t(wl)*RadPerBand, turbidity, albedo, solarElevation, s.config[wl], s.radiance[wl]);
        const double lambda = (320.0 + 40.0*wl)*10E-10;
        sunCorrection[wl] = blackbody(temperature, lambda)*blackbodyScale/originalSolarRadiance[wl];
    }
    double sum = 0.0;
    for (int i = 2; i < NumBands; ++i)                                  // the nine visible bands
        sum += sunCorrection[i];
    const double ratio = sum/9.0;
    for (int i = 0; i < NumBands; ++i)
        s.skyCorrection[i] = solarIntensity*sunCorrection[i]/ratio;
}

// ArHosekSkyModel_GetRadianceInternal (ArHosekSkyModel.cpp:291-304)
double bandRadiance(const double c[9], double theta, double gamma)
{
    const double expM = std::exp(c[4]*gamma);
    const double rayM = std::cos(gamma)*std::cos(gamma);
    const double mieM = (1.0 + std::cos(gamma)*std::cos(gamma))/std::pow(1.0 + c[8]*c[8] - 2.0*c[8]*std::cos(gamma), 1.5);
    const double zenith = std::sqrt(std::cos(theta));
    return (1.0 + c[0]*std::exp(c[1]/(std::cos(theta) + 0.01)))*(c[2] + c[3]*expM + c[5]*rayM + c[6]*mieM + c[7]*zenith);
}

// arhosekskymodel_radiance (ArHosekSkyModel.cpp:519-563): linear between the two neighbouring bands
double skyRadiance(const SkyState &s, double theta, double gamma, double wavelength)
{
    const int low = int((wavelength - 320.0)/40.0);
    if (low < 0 || low >= NumBands)
        return 0.0f;
    const double interp = std::fmod((wavelength - 320.0)/40.0, 1.0);
    const double valLow = bandRadiance(s.config[low], theta, gamma)*s.radiance[low]*s.skyCorrection[low];
    if (interp < 1e-6)
        return valLow;
    double result = (1.0 - interp)*valLow;
    if (low + 1 < NumBands)
        result += interp*bandRadiance(s.config[low + 1], theta, gamma)*s.radiance[low + 1]*s.skyCorrection[low + 1];
    return result;
}

}

std::string skydomeTablesPath()
{
    if (const char *env = std::getenv("TUNGSTEN_HIP_SKYDOME_TABLES"))
        return env;
    Dl_info info;
    if (dladdr(reinterpret_cast<const void *>(&skydomeTablesPath), &info) && info.dli_fname) {
        std::string lib = info.dli_fname;
        const size_t slash = lib.find_last_of('/');
        const std::string dir = slash == std::string::npos ? "." : lib.substr(0, slash);
        return dir + "/../data/skydome_tables.bin";
    }
    return "tungsten_amd/data/skydome_tables.bin";
}

std::vector<float> bakeSkydomeImage(const float sun[3], float temperature, float turbidity, float intensity)
{
    const float PI = 3.1415926536f, TWO_PI = PI*2.0f;             // math/Angle.hpp:8-10
    const Tables &t = tables();
    // Spectral::spectralXyzWeights(NumSamples, lambdas, weights) (math/Spectral.cpp:370-392): the CIE curves resampled to NumSamples
    // equidistant wavelengths by linear ("tent") weights, normalised by the integral of y-bar
    float lambdas[NumSamples], weights[NumSamples][3];
    {
        const float delta = (CieMax - CieMin)/(NumSamples - 1);
        for (int i = 0; i < NumSamples; ++i) {
            lambdas[i] = CieMin + i*delta;
            weights[i][0] = weights[i][1] = weights[i][2] = 0.0f;
        }
        const float *X = t.cie.data(), *Y = X + CieSamples, *Z = Y + CieSamples;
        float ref = 0.0f;
        for (int i = 0; i < CieSamples; ++i) {
            const int x = int(i/delta);
            const float u = i/delta - x;
            const float entry[3] = {X[i], Y[i], Z[i]};
            for (int k = 0; k < 3; ++k) {
                weights[x][k] += (1.0f - u)*entry[k];
                if (x + 1 < NumSamples)          // (i = CieSamples - 1 lands exactly on the last sample: u == 0, nothing to spread)
                    weights[x + 1][k] += u*entry[k];
            }
            if (i < CieSamples - 1)
                ref += (Y[i] + Y[i + 1])*0.5f;
        }
        for (int i = 0; i < NumSamples; ++i)
            for (int k = 0; k < 3; ++k)
                weights[i][k] /= ref;
    }

    // Skydome::prepareForRender (Skydome.cpp:279-306)
    const float sunElevation = std::asin(std::min(std::max(sun[1], -1.0f), 1.0f));
    SkyState state;
    initState(state, sunElevation, intensity, temperature, turbidity, 0.2f);

    std::vector<float> img(size_t(SkydomeSizeX)*SkydomeSizeY*3, 0.0f);
    for (int y = 0; y < SkydomeSizeY/2; ++y) {                          // fillImage (:259-277), gammaScale = 1
        const float theta = (y + 0.5f)*PI/SkydomeSizeY;
        for (int x = 0; x < SkydomeSizeX; ++x) {
            const float phi = (x + 0.5f)*TWO_PI/SkydomeSizeX;
            const float v[3] = {std::cos(phi)*std::sin(theta), std::cos(theta), std::sin(phi)*std::sin(theta)};
            float dot = v[0]*sun[0];
            dot += v[1]*sun[1];
            dot += v[2]*sun[2];
            const float gamma = std::min(std::max(std::acos(std::min(std::max(dot, -1.0f), 1.0f))*1.0f, 0.0f), PI);
            float xyz[3] = {0.0f, 0.0f, 0.0f};
            for (int i = 0; i < NumSamples; ++i) {
                const float r = float(skyRadiance(state, theta, gamma, lambdas[i]));
                for (int k = 0; k < 3; ++k)
                    xyz[k] += weights[i][k]*r;
            }
            float *px = &img[(size_t(x) + size_t(y)*SkydomeSizeX)*3];   // Spectral::xyzToRgb (math/Spectral.hpp:21-27)
            px[0] += 3.240479f*xyz[0] + -1.537150f*xyz[1] + -0.498535f*xyz[2];
            px[1] += -0.969256f*xyz[0] + 1.875991f*xyz[1] + 0.041556f*xyz[2];
            px[2] += 0.055648f*xyz[0] + -0.204043f*xyz[1] + 1.057311f*xyz[2];
        }
    }
    for (int y = SkydomeSizeY/2; y < std::min(SkydomeSizeY/2 + 2, SkydomeSizeY); ++y)
        std::memcpy(&img[size_t(y)*SkydomeSizeX*3], &img[size_t(SkydomeSizeY/2 - 1)*SkydomeSizeX*3], size_t(SkydomeSizeX)*3*sizeof(float));
    return img;
}

}
