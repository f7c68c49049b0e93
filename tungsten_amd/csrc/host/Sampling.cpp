#include "Sampling.hpp"
#include "Math.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>

#include <dlfcn.h>

namespace tungsten_amd {

// PCG-XSH-RR 64/32 (sampling/UniformSampler.hpp:40-47)
uint32_t UniformSampler::nextI()
{
    uint64_t oldState = _state;
    _state = oldState*6364136223846793005ULL + (_sequence | 1);
    uint32_t xorShifted = uint32_t(((oldState >> 18u) ^ oldState) >> 27u);
    uint32_t rot = uint32_t(oldState >> 59u);
    return (xorShifted >> rot) | (xorShifted << (uint32_t(-int32_t(rot)) & 31));
}

// BitManip::normalizedUint (math/BitManip.hpp:47-50)
float UniformSampler::next1D()
{
    uint32_t bits = (nextI() >> 9u) | 0x3F800000u;
    float f;
    std::memcpy(&f, &bits, sizeof(f));
    return f - 1.0f;
}

// ------------------------------------------------------------------------------------------
std::string SobolMatrices::defaultPath()
{
    if (const char *env = std::getenv("TUNGSTEN_HIP_SOBOL_MATRICES"))
        return env;
    Dl_info info;
    if (dladdr(reinterpret_cast<const void *>(&SobolMatrices::defaultPath), &info) && info.dli_fname) {
        std::string lib(info.dli_fname);
        size_t slash = lib.find_last_of('/');
        std::string dir = slash == std::string::npos ? std::string(".") : lib.substr(0, slash);
        return dir + "/../data/sobol_matrices_1024x52.bin";
    }
    return "tungsten_amd/data/sobol_matrices_1024x52.bin";
}

const std::vector<uint32_t> &SobolMatrices::get()
{
    static std::vector<uint32_t> table;
    static std::mutex lock;
    std::lock_guard<std::mutex> guard(lock);
    if (!table.empty())
        return table;
    const size_t words = size_t(TGHIP_SOBOL_DIMS)*TGHIP_SOBOL_BITS;
    std::string path = defaultPath();
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f)
        throw std::runtime_error("stratified_sampler needs the Sobol' generator matrices: cannot open '" + path +
                                 "' (set TUNGSTEN_HIP_SOBOL_MATRICES)");
    std::vector<uint32_t> t(words);
    size_t got = std::fread(t.data(), sizeof(uint32_t), words, f);
    bool extra = std::fgetc(f) != EOF;
    std::fclose(f);
    // the first dimension is the van der Corput sequence: column i is bit 31 - i
    if (got != words || extra || t[0] != 0x80000000u || t[1] != 0x40000000u || t[31] != 1u)
        throw std::runtime_error("'" + path + "' is not a 1024 x 52 table of Sobol' generator matrices");
    table.swap(t);
    return table;
}

// ------------------------------------------------------------------------------------------
void PassScheduler::reset(uint32_t w, uint32_t h, uint32_t seed)
{
    _w = w;
    _h = h;
    _varianceW = (w + VarianceTileSize - 1)/VarianceTileSize;
    _varianceH = (h + VarianceTileSize - 1)/VarianceTileSize;
    _sampler = UniformSampler(hash32(seed));
    // diceTiles (:27-42): one sampler seed per tile, row-major, whichever sampler type the tile gets
    _tileSeeds.clear();
    for (uint32_t y = 0; y < h; y += TileSize)
        for (uint32_t x = 0; x < w; x += TileSize)
            _tileSeeds.push_back(hash32(_sampler.nextI()));
    TgHostSampleRecord zero;
    std::memset(&zero, 0, sizeof(zero));
    _samples.assign(size_t(_varianceW)*_varianceH, zero);
}

// SampleRecord::errorEstimate (SampleRecord.hpp:60-68)
static float errorEstimate(const TgHostSampleRecord &r)
{
    float variance = r.running_variance/float(r.sample_count - 1u);
    float meanSq = r.mean*r.mean;
    return variance/(float(r.sample_count)*(meanSq > 1e-3f ? meanSq : 1e-3f));
}

float PassScheduler::errorPercentile95()
{
    // The element std::sort would leave at position (n*95)/100 of the positive error estimates (PathTraceIntegrator.cpp:44-59).  Positive floats order
    // like their bit patterns, so the element is found by counting: a histogram over the high 16 bits locates the bin that holds it, the bin's few
    // members are then selected among -- the same value std::nth_element finds, in a third of its time (this runs between two passes, with the
    // device idle: profiles/r5_as_shipped_timeline.txt).
    std::vector<uint32_t> &bits = _errorBits, &bins = _errorBins;
    bits.clear();
    bits.reserve(_samples.size());
    bins.assign(1u << 16, 0u);
    for (TgHostSampleRecord &r : _samples) {
        r.adaptive_weight = errorEstimate(r);
        if (r.adaptive_weight > 0.0f) {                  // (NaN estimates fail the comparison, as in the reference)
            uint32_t b;
            std::memcpy(&b, &r.adaptive_weight, sizeof(b));
            bits.push_back(b);
            bins[b >> 16]++;
        }
    }
    if (bits.empty())
        return 0.0f;
    size_t k = (bits.size()*95)/100;
    uint32_t bin = 0;
    while (k >= bins[bin])
        k -= bins[bin++];
    size_t n = 0;                                        // the bin's members, compacted to the front of the list itself
    for (uint32_t b : bits)
        if ((b >> 16) == bin)
            bits[n++] = b;
    std::nth_element(bits.begin(), bits.begin() + std::ptrdiff_t(k), bits.begin() + std::ptrdiff_t(n));
    float result;
    std::memcpy(&result, &bits[k], sizeof(result));
    return result;
}

static inline float maxOf(float a, float b) { return a > b ? a : b; }   // MathUtil.hpp:23-26

void PassScheduler::dilateAdaptiveWeights()
{
    const int vw = int(_varianceW), vh = int(_varianceH);
    for (int y = 0; y < vh; ++y)
        for (int x = 0; x < vw; ++x) {
            int idx = x + y*vw;
            if (y < vh - 1) _samples[idx].adaptive_weight = maxOf(_samples[idx].adaptive_weight, _samples[idx + vw].adaptive_weight);
            if (x < vw - 1) _samples[idx].adaptive_weight = maxOf(_samples[idx].adaptive_weight, _samples[idx + 1].adaptive_weight);
        }
    for (int y = vh - 1; y >= 0; --y)
        for (int x = vw - 1; x >= 0; --x) {
            int idx = x + y*vw;
            if (y > 0) _samples[idx].adaptive_weight = maxOf(_samples[idx].adaptive_weight, _samples[idx - vw].adaptive_weight);
            if (x > 0) _samples[idx].adaptive_weight = maxOf(_samples[idx].adaptive_weight, _samples[idx - 1].adaptive_weight);
        }
}

void PassScheduler::distributeAdaptiveSamples(int spp)
{
    double totalWeight = 0.0;
    for (const TgHostSampleRecord &r : _samples)
        totalWeight += r.adaptive_weight;

    int adaptiveBudget = int(uint32_t(spp - 1)*_w*_h);
    int budgetPerTile = adaptiveBudget/int(VarianceTileSize*VarianceTileSize);
    float weightToSampleFactor = float(double(budgetPerTile)/totalWeight);

    // (PathTraceIntegrator.cpp:121-132.  The comparison's outcome is a coin toss, so it is taken without a branch: x - 0.0f is x.)
    float pixelPdf = 0.0f;
    for (TgHostSampleRecord &r : _samples) {
        float fractionalSamples = r.adaptive_weight*weightToSampleFactor;
        int adaptiveSamples = int(fractionalSamples);
        pixelPdf += fractionalSamples - float(adaptiveSamples);
        const bool extra = _sampler.next1D() < pixelPdf;
        adaptiveSamples += extra ? 1 : 0;
        pixelPdf -= extra ? 1.0f : 0.0f;
        r.next_sample_count = uint32_t(adaptiveSamples + 1);
    }
}

bool PassScheduler::generateWork(uint32_t currentSpp, uint32_t nextSpp, bool enableAdaptive)
{
    for (TgHostSampleRecord &r : _samples)
        r.sample_index += r.next_sample_count;

    int sppCount = int(nextSpp - currentSpp);
    if (enableAdaptive && currentSpp >= AdaptiveThreshold) {
        float maxError = errorPercentile95();
        if (maxError == 0.0f)
            return false;
        for (TgHostSampleRecord &r : _samples)
            r.adaptive_weight = r.adaptive_weight < maxError ? r.adaptive_weight : maxError;
        dilateAdaptiveWeights();
        distributeAdaptiveSamples(sppCount);
    } else {
        for (TgHostSampleRecord &r : _samples)
            r.next_sample_count = uint32_t(sppCount);
    }
    return true;
}

void PassScheduler::absorb(const TgHipSampleRecord *const *sources, size_t numSources)
{
    // record (rx, ry) lies inside tile (rx/4, ry/4), which belongs to shard tghip_tile_owner(tx, ty, numSources) (include/tungsten_hip.h)
    const uint32_t recsPerTile = TileSize/VarianceTileSize;
    for (uint32_t ry = 0; ry < _varianceH; ++ry)
        for (uint32_t rx = 0; rx < _varianceW; ++rx) {
            size_t i = size_t(ry)*_varianceW + rx;
            const TgHipSampleRecord &d = sources[tghip_tile_owner(rx/recsPerTile, ry/recsPerTile, uint32_t(numSources))][i];
            _samples[i].sample_count = d.sample_count;
            _samples[i].mean = d.mean;
            _samples[i].running_variance = d.running_variance;
        }
}

void PassScheduler::passArrays(std::vector<uint32_t> &index, std::vector<uint32_t> &count) const
{
    index.resize(_samples.size());
    count.resize(_samples.size());
    for (size_t i = 0; i < _samples.size(); ++i) {
        index[i] = _samples[i].sample_index;
        count[i] = _samples[i].next_sample_count;
    }
}

} // namespace tungsten_amd
