// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE.  Our own driver program that links the
// *reference implementation* (libcore.a etc. built by oracle/Makefile.ref from /root/reference)
// and produces the golden vectors under tests/golden/ that pin oracle/oracle.c:
//
//   ref_harness render  <scene.json> <seed> <spp> <out.pfm> [threads] [first sample]
//       mean radiance per pixel from the reference's own PathTracer::traceSample, with the
//       random numbers supplied by GraftPathSampler below -- the same counter-based stream
//       (keyed by seed, pixelIndex, sampleIndex) the oracle and the HIP kernels use.  Because the
//       reference's PathSampleGenerator is an abstract interface (sampling/PathSampleGenerator.hpp)
//       this needs no change to the reference: identical random numbers in, so the reference,
//       the oracle and the GPU can be compared per pixel / per sample, not just statistically.
//   ref_harness samples <scene.json> <seed> <spp> <out.bin>
//       float32[h][w][spp][3] radiance of every individual sample.
//   ref_harness units   <scene.json> <out.json>
//       known-answer vectors of the deterministic building blocks (L1 in SURVEY.md 8c).
//   ref_harness integrate <scene.json> <seed> <out.bin> [threads]
//       the reference's OWN PathTraceIntegrator pass loop (diceTiles, generateWork, adaptive sampling,
//       SampleRecord, OutputBuffer) with the tile samplers swapped for the shared counter-based ones
//       (GraftPathSampler / GraftSobolPathSampler).  Dumps the SampleRecords after every pass and the
//       final image.
//   ref_harness sobol-table <out.bin>
//       the 1024 x 52 generator matrices of the Sobol' sequence (sobol::Matrices::matrices) as u32.
//   ref_harness sky-image <scene.json> <out.bin>
//       the sky image the scene's skydome baked (512 x 256 x 3 float32).
//   ref_harness bounds <scene.json> <out.txt>
//       bounds() of every finite primitive in scene order: the items' boxes of the reference's top-level Embree geometry.
//
// Nothing here is copied from the reference; it only calls its public classes.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <utility>
#include <vector>
#include "sampling/PathSampleGenerator.hpp"
#include "sampling/UniformSampler.hpp"
#include "thread/TaskGroup.hpp"
#include "integrators/Integrator.hpp"
#include "integrators/ImageTile.hpp"
#include "integrators/path_tracer/PathTracerSettings.hpp"
#include "integrators/path_tracer/SampleRecord.hpp"
#include "integrators/path_tracer/PathTracer.hpp"
#include "math/MathUtil.hpp"
#include "math/Quaternion.hpp"
#include "bvh/BinaryBvh.hpp"
#include "primitives/Primitive.hpp"
#include <sobol/sobol.h>
// The integrate command inspects the integrator's tiles and SampleRecords, which the reference keeps
// private: open the two class definitions up (every standard/other header they use is already included).
#define private public
#define protected public
#define class struct
#include "sampling/SobolPathSampler.hpp"
#include "integrators/path_tracer/PathTraceIntegrator.hpp"
#include "primitives/Instance.hpp"
#include "textures/BitmapTexture.hpp"
#include "cameras/ThinlensCamera.hpp"
#include "primitives/Skydome.hpp"
#undef private
#undef protected
#undef class
#include "integrators/path_tracer/PathTracer.hpp"
#include "primitives/EmbreeUtil.hpp"
#include "primitives/InfiniteSphere.hpp"
#include "renderer/TraceableScene.hpp"
#include "sampling/PathSampleGenerator.hpp"
#include "sampling/UniformSampler.hpp"
#include "thread/ThreadUtils.hpp"
#include "io/DirectoryChange.hpp"
#include "io/FileUtils.hpp"
#include "io/ImageIO.hpp"
#include "io/Scene.hpp"
#include "math/MathUtil.hpp"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <thread>
#include <vector>

using namespace Tungsten;

static std::vector<std::pair<char, float>> *g_drawLog = nullptr;   // `draws` command: every number a path consumed

// The counter-based stream shared with oracle/oracle.c (sampler_start) and the HIP kernels.
class GraftPathSampler : public PathSampleGenerator
{
    UniformSampler _sampler;
    uint32 _seed;
    std::vector<float> _replay;
    size_t _replayPos = 0;
    bool _useReplay = false;

public:
    uint64 draws = 0;

    GraftPathSampler(uint32 seed) : _sampler(0), _seed(seed) {}

    void setReplay(const std::vector<float> &values) { _replay = values; _replayPos = 0; _useReplay = true; }
    size_t consumed() const { return _replayPos; }

    virtual void startPath(uint32 pixelId, uint32 sample) override
    {
        uint32 a = MathUtil::hash32(_seed) ^ pixelId;
        uint32 b = MathUtil::hash32(a) + sample;
        uint32 hi = MathUtil::hash32(b), lo = MathUtil::hash32(b ^ 0x9E3779B9u);
        _sampler = UniformSampler((uint64(hi) << 32) | lo, (uint64(pixelId) << 1) | 1u);
    }
    virtual void advancePath() override {}
    virtual void saveState(OutputStreamHandle &) override {}
    virtual void loadState(InputStreamHandle &) override {}

    virtual float next1D() override final
    {
        draws++;
        if (_useReplay)
            return _replayPos < _replay.size() ? _replay[_replayPos++] : (++_replayPos, 0.5f);
        float v = _sampler.next1D();
        if (g_drawLog) g_drawLog->push_back(std::make_pair('u', v));
        return v;
    }
    virtual bool nextBoolean(float pTrue) override final { return next1D() < pTrue; }
    virtual int nextDiscrete(int numChoices) override final { return int(next1D()*numChoices); }
    virtual Vec2f next2D() override final { float a = next1D(); float b = next1D(); return Vec2f(a, b); }
    virtual UniformSampler &uniformGenerator() override final { return _sampler; }
};

// SobolPathSampler (sampling/SobolPathSampler.hpp:20-79) with its sequential per-tile supplemental
// stream (booleans, dimensions >= 1024) replaced by the counter-based stream above; the Sobol'
// dimensions themselves are the reference's: same tile seed, scramble, permuted index.
class GraftSobolPathSampler : public PathSampleGenerator
{
    GraftPathSampler _supplemental;
    uint32 _tileSeed, _scramble = 0, _index = 0, _dimension = 0;

public:
    GraftSobolPathSampler(uint32 tileSeed, uint32 seed) : _supplemental(seed), _tileSeed(tileSeed) {}
    void setTileSeed(uint32 tileSeed) { _tileSeed = tileSeed; }
    uint64 draws() const { return _supplemental.draws; }

    virtual void startPath(uint32 pixelId, uint32 sample) override
    {
        _scramble = _tileSeed ^ MathUtil::hash32(pixelId);
        _index = sample;
        _dimension = 0;
        _supplemental.startPath(pixelId, sample);
    }
    virtual void advancePath() override {}
    virtual void saveState(OutputStreamHandle &) override {}
    virtual void loadState(InputStreamHandle &) override {}

    virtual float next1D() override final
    {
        if (_dimension >= 1024)
            return _supplemental.next1D();
        uint32 permuted = (_index & ~0xFFu) | ((_index + _scramble) & 0xFFu);
        float v = BitManip::normalizedUint(sobol::sample(permuted, _dimension++, _scramble));
        if (g_drawLog) g_drawLog->push_back(std::make_pair('s', v));
        return v;
    }
    virtual bool nextBoolean(float pTrue) override final { return _supplemental.next1D() < pTrue; }
    virtual int nextDiscrete(int numChoices) override final { return int(_supplemental.next1D()*numChoices); }
    virtual Vec2f next2D() override final { float a = next1D(); float b = next1D(); return Vec2f(a, b); }
    virtual UniformSampler &uniformGenerator() override final { return _supplemental.uniformGenerator(); }
};

struct Loaded
{
    std::unique_ptr<Scene> scene;
    std::unique_ptr<TraceableScene> ts;
    PathTraceIntegrator *pti = nullptr;
    Path dir;
};

static bool loadScene(const char *path, uint32 seed, Loaded &out)
{
    Path scenePath(path);
    out.dir = scenePath.parent();
    try {
        out.scene.reset(Scene::load(scenePath, nullptr, &out.dir));
        out.scene->loadResources();
        DirectoryChange context(out.dir);
        // Scene::loadResources only reaches the scene's own primitives and Instance::loadResources (Instance.cpp:265-282)
        // does not forward to its masters: load the master meshes here, as code that builds Instances in memory would.
        for (const std::shared_ptr<Primitive> &p : out.scene->primitives())
            if (Instance *inst = dynamic_cast<Instance *>(p.get()))
                for (std::shared_ptr<Primitive> &m : inst->_master)
                    m->loadResources();
        // A thin-lens camera makes its aperture samplable inside fromJson (ThinlensCamera::precompute, cameras/ThinlensCamera.cpp:27-35) --
        // for a bitmap aperture that is before Scene::loadResources has read the image, so the Distribution2D is built over 0 x 0 texels and
        // the unmodified `tungsten` dies with SIGSEGV at the first lens sample.  Build it again now that the texels exist, as a program
        // that assembles the camera in memory after loading the bitmap gets it.
        if (ThinlensCamera *tl = dynamic_cast<ThinlensCamera *>(out.scene->camera().get()))
            if (BitmapTexture *bmp = dynamic_cast<BitmapTexture *>(tl->_aperture.get())) {
                bmp->_distribution[MAP_UNIFORM].reset();
                tl->precompute();
            }
        out.ts.reset(out.scene->makeTraceable(seed));
    } catch (const std::exception &e) {
        std::fprintf(stderr, "ref_harness: %s\n", e.what());
        return false;
    }
    out.pti = dynamic_cast<PathTraceIntegrator *>(out.scene->integrator());
    if (!out.pti) {
        std::fprintf(stderr, "ref_harness: scene does not use the path_tracer integrator\n");
        return false;
    }
    return true;
}

static int cmdRender(int argc, char **argv, bool dumpSamples)
{
    if (argc < 6) return 2;
    uint32 seed = uint32(std::strtoul(argv[3], nullptr, 0));
    int spp = std::atoi(argv[4]);
    int threads = argc > 6 ? std::atoi(argv[6]) : int(std::thread::hardware_concurrency());
    if (threads < 1) threads = 1;
    ThreadUtils::startThreads(threads);
    Loaded L;
    if (!loadScene(argv[2], seed, L)) return 1;
    int w = L.scene->camera()->resolution().x(), h = L.scene->camera()->resolution().y();
    // "stratified_sampler": true -> the tiles' SobolPathSampler seeds, exactly as diceTiles draws them
    // (PathTraceIntegrator.cpp:27-42 after :187); [first sample] optional 7th argument
    const bool sobol = L.scene->rendererSettings().useSobol();
    const int firstSample = argc > 7 ? std::atoi(argv[7]) : 0;
    std::vector<uint32> tileSeeds;
    {
        UniformSampler dice(MathUtil::hash32(seed));
        for (int ty = 0; ty < (h + 15)/16; ++ty)
            for (int tx = 0; tx < (w + 15)/16; ++tx)
                tileSeeds.push_back(MathUtil::hash32(dice.nextI()));
    }

    std::vector<float> mean(size_t(w)*h*3, 0.0f);
    std::vector<float> samples;
    if (dumpSamples) samples.resize(size_t(w)*h*spp*3);
    std::vector<std::thread> pool;
    std::vector<uint64> draws(threads, 0);
    for (int t = 0; t < threads; ++t) {
        pool.emplace_back([&, t]() {
            PathTracer tracer(L.ts.get(), L.pti->settings(), uint32(t));
            GraftPathSampler uniformSampler(seed);
            GraftSobolPathSampler sobolSampler(0, seed);
            PathSampleGenerator &sampler = sobol ? static_cast<PathSampleGenerator &>(sobolSampler) : uniformSampler;
            for (int y = t; y < h; y += threads) {
                for (int x = 0; x < w; ++x) {
                    uint32 pixelIndex = uint32(x + y*w);
                    sobolSampler.setTileSeed(tileSeeds[size_t(y/16)*size_t((w + 15)/16) + size_t(x/16)]);
                    float sum[3] = {0, 0, 0};
                    uint32 count = 0;
                    for (int s = 0; s < spp; ++s) {
                        sampler.startPath(pixelIndex, uint32(firstSample + s));
                        Vec3f c = tracer.traceSample(Vec2u(uint32(x), uint32(y)), sampler);
                        if (dumpSamples)
                            for (int k = 0; k < 3; ++k) samples[((size_t(pixelIndex))*spp + s)*3 + k] = c[k];
                        if (std::isnan(c) || std::isinf(c))     // OutputBuffer::addSample drops these
                            continue;
                        for (int k = 0; k < 3; ++k) sum[k] += c[k];
                        count++;
                    }
                    for (int k = 0; k < 3; ++k) mean[size_t(pixelIndex)*3 + k] = count ? sum[k]/float(count) : 0.0f;
                }
            }
            draws[t] = uniformSampler.draws + sobolSampler.draws();
        });
    }
    for (auto &t : pool) t.join();
    uint64 totalDraws = 0;
    for (uint64 d : draws) totalDraws += d;
    std::fprintf(stderr, "ref_harness: %dx%d @ %d spp, %.2f random numbers per sample\n", w, h, spp,
                 double(totalDraws)/(double(w)*h*spp));

    if (dumpSamples) {
        std::ofstream out(argv[5], std::ios::binary);
        out.write(reinterpret_cast<const char *>(samples.data()), std::streamsize(samples.size()*sizeof(float)));
    } else {
        ImageIO::saveHdr(Path(argv[5]), mean.data(), w, h, 3);
    }
    return 0;
}

// ---- the reference's own pass loop ----------------------------------------------------------
// out.bin: u32 w, h, varianceW, varianceH, passes, useSobol; u32 tileSeeds[tiles] (0 for the uniform sampler);
// per pass: u32 currentSpp (after the pass), then per record {u32 sampleCount, nextSampleCount, sampleIndex;
// f32 adaptiveWeight, mean, runningVariance}; finally f32 image[h][w][3] (Camera::getLinear) .
static int cmdIntegrate(int argc, char **argv)
{
    if (argc < 5) return 2;
    uint32 seed = uint32(std::strtoul(argv[3], nullptr, 0));
    int threads = argc > 5 ? std::atoi(argv[5]) : int(std::thread::hardware_concurrency());
    if (threads < 1) threads = 1;
    ThreadUtils::startThreads(threads);
    Loaded L;
    if (!loadScene(argv[2], seed, L)) return 1;
    PathTraceIntegrator &pti = *L.pti;
    bool sobol = L.scene->rendererSettings().useSobol();
    std::vector<uint32> tileSeeds;
    for (ImageTile &tile : pti._tiles) {
        if (sobol) {
            uint32 tileSeed = static_cast<SobolPathSampler *>(tile.sampler.get())->_seed;
            tileSeeds.push_back(tileSeed);
            tile.sampler.reset(new GraftSobolPathSampler(tileSeed, seed));
        } else {
            tileSeeds.push_back(0);
            tile.sampler.reset(new GraftPathSampler(seed));
        }
    }
    std::vector<char> passes;
    uint32 numPasses = 0;
    auto put = [&](const void *p, size_t n) { passes.insert(passes.end(), (const char *)p, (const char *)p + n); };
    while (!pti.done()) {
        pti.startRender([]() {});
        pti.waitForCompletion();
        uint32 spp = pti.currentSpp();
        put(&spp, 4);
        for (const SampleRecord &r : pti._samples) {
            put(&r.sampleCount, 4); put(&r.nextSampleCount, 4); put(&r.sampleIndex, 4);
            put(&r.adaptiveWeight, 4); put(&r.mean, 4); put(&r.runningVariance, 4);
        }
        numPasses++;
    }
    uint32 w = pti._w, h = pti._h;
    std::ofstream out(argv[4], std::ios::binary);
    uint32 header[6] = {w, h, pti._varianceW, pti._varianceH, numPasses, sobol ? 1u : 0u};
    out.write((const char *)header, sizeof(header));
    out.write((const char *)tileSeeds.data(), std::streamsize(tileSeeds.size()*4));
    out.write(passes.data(), std::streamsize(passes.size()));
    for (uint32 y = 0; y < h; ++y)
        for (uint32 x = 0; x < w; ++x) {
            Vec3f c = L.scene->camera()->getLinear(x, y);
            out.write((const char *)c.data(), 12);
        }
    // scenes with renderer.output_buffers: Camera::serializeOutputBuffers (Camera.cpp:222-229) appended -- per requested
    // output, in the order color, depth, normal, albedo, visibility: _bufferA, [_bufferB], [_variance], _sampleCount
    if (!L.scene->rendererSettings().renderOutputs().empty()) {
        OutputStreamHandle handle(new std::ostringstream(std::ios::binary));
        L.scene->camera()->serializeOutputBuffers(handle);
        std::string blob = static_cast<std::ostringstream *>(handle.get())->str();
        out.write(blob.data(), std::streamsize(blob.size()));
    }
    std::fprintf(stderr, "ref_harness: integrate %ux%u, %u passes, %s sampler, adaptive %d\n", w, h, numPasses,
                 sobol ? "sobol" : "uniform", int(L.scene->rendererSettings().useAdaptiveSampling()));
    return 0;
}

// ref_harness draws <scene.json> <seed> <px> <py> <sample>: the random numbers one path consumes, in order
// ('s' = Sobol' dimension, 'u' = counter-based uniform stream), then its radiance.  Debugging aid.
static int cmdDraws(int argc, char **argv)
{
    if (argc < 7) return 2;
    uint32 seed = uint32(std::strtoul(argv[3], nullptr, 0));
    uint32 px = uint32(std::atoi(argv[4])), py = uint32(std::atoi(argv[5])), sample = uint32(std::atoi(argv[6]));
    ThreadUtils::startThreads(1);
    Loaded L;
    if (!loadScene(argv[2], seed, L)) return 1;
    uint32 w = L.scene->camera()->resolution().x(), h = L.scene->camera()->resolution().y();
    UniformSampler dice(MathUtil::hash32(seed));
    uint32 tileSeed = 0, tile = (py/16)*((w + 15)/16) + px/16;
    for (uint32 t = 0; t <= tile && t < ((w + 15)/16)*((h + 15)/16); ++t)
        tileSeed = MathUtil::hash32(dice.nextI());
    PathTracer tracer(L.ts.get(), L.pti->settings(), 0);
    GraftPathSampler uniformSampler(seed);
    GraftSobolPathSampler sobolSampler(tileSeed, seed);
    PathSampleGenerator &sampler = L.scene->rendererSettings().useSobol() ? static_cast<PathSampleGenerator &>(sobolSampler) : uniformSampler;
    std::vector<std::pair<char, float>> log;
    g_drawLog = &log;
    sampler.startPath(px + py*w, sample);
    Vec3f c = tracer.traceSample(Vec2u(px, py), sampler);
    g_drawLog = nullptr;
    for (auto &d : log) std::printf("%c %.9g\n", d.first, d.second);
    std::printf("radiance %.9g %.9g %.9g\n", c.x(), c.y(), c.z());
    return 0;
}

static int cmdSobolTable(int argc, char **argv)
{
    if (argc < 3) return 2;
    std::ofstream out(argv[2], std::ios::binary);
    out.write((const char *)sobol::Matrices::matrices, std::streamsize(sobol::Matrices::num_dimensions*sobol::Matrices::size*4));
    return out ? 0 : 1;
}

// ---- known-answer vectors -------------------------------------------------------------------
static void pv(FILE *f, const char *name, const Vec3f &v, bool comma = true)
{
    std::fprintf(f, "\"%s\": [%.9g, %.9g, %.9g]%s", name, v.x(), v.y(), v.z(), comma ? ", " : "");
}

// ref_harness sky-image <scene.json> <out.bin>: the 512 x 256 RGB image the scene's first skydome baked at prepareForRender
// (Skydome::_sky, primitives/Skydome.cpp:279-306) as float32 -- what tungsten_amd/csrc/host/SkyModel.cpp restates
static int cmdSkyImage(int argc, char **argv)
{
    if (argc < 4) return 2;
    ThreadUtils::startThreads(1);
    Loaded l;
    if (!loadScene(argv[2], 0xBA5EBA11u, l)) return 1;
    for (const std::shared_ptr<Primitive> &p : l.scene->primitives())
        if (Skydome *sky = dynamic_cast<Skydome *>(p.get())) {
            const BitmapTexture &t = *sky->_sky;
            std::ofstream out(argv[3], std::ios::binary);
            out.write(reinterpret_cast<const char *>(t._texels), size_t(t._w)*t._h*3*sizeof(float));
            std::printf("ref_harness: sky image %d x %d\n", t._w, t._h);
            return out ? 0 : 1;
        }
    std::fprintf(stderr, "ref_harness: the scene has no skydome\n");
    return 1;
}

// ref_harness bounds <scene.json> <out.txt>: per finite primitive, in scene order (the items of TraceableScene's user geometry,
// renderer/TraceableScene.hpp:101-107), its bounds() after prepareForRender as six float bit patterns -- what the library's and the oracle's
// restatements of Quad / Cube / Sphere / Disk / Cylinder::bounds are held to (tests/golden/prim_bounds.json, tests/test_top_tree.py)
static int cmdBounds(int argc, char **argv)
{
    if (argc < 4) return 2;
    ThreadUtils::startThreads(1);
    Loaded l;
    if (!loadScene(argv[2], 0xBA5EBA11u, l)) return 1;
    FILE *f = std::fopen(argv[3], "w");
    if (!f) return 1;
    for (const std::shared_ptr<Primitive> &p : l.scene->primitives()) {
        if (p->isInfinite() || p->isDirac()) continue;
        const Box3f b = p->bounds();
        const float v[6] = {b.min().x(), b.min().y(), b.min().z(), b.max().x(), b.max().y(), b.max().z()};
        for (int k = 0; k < 6; ++k) { unsigned u; std::memcpy(&u, &v[k], 4); std::fprintf(f, "%08x%s", u, k == 5 ? "\n" : " "); }
    }
    std::fclose(f);
    return 0;
}

static int cmdUnits(int argc, char **argv)
{
    if (argc < 4) return 2;
    ThreadUtils::startThreads(1);
    Loaded L;
    const uint32 seed = 0xBA5EBA11u;
    if (!loadScene(argv[2], seed, L)) return 1;
    TraceableScene &ts = *L.ts;
    Camera &cam = ts.cam();
    int w = cam.resolution().x(), h = cam.resolution().y();
    FILE *f = std::fopen(argv[3], "w");
    if (!f) return 1;
    std::fprintf(f, "{\n\"width\": %d, \"height\": %d,\n", w, h);

    // -- RNG stream (PCG + hash32 as implemented by the reference's UniformSampler / MathUtil)
    std::fprintf(f, "\"rng\": [\n");
    {
        const uint32 keys[][3] = {{seed, 0, 0}, {seed, 1, 0}, {seed, 12345, 7}, {1, 99, 255}, {0xFFFFFFFFu, 921599u, 1023u}};
        for (size_t i = 0; i < sizeof(keys)/sizeof(keys[0]); ++i) {
            GraftPathSampler s(keys[i][0]);
            s.startPath(keys[i][1], keys[i][2]);
            std::fprintf(f, "  {\"seed\": %u, \"pixel\": %u, \"sample\": %u, \"values\": [", keys[i][0], keys[i][1], keys[i][2]);
            for (int k = 0; k < 16; ++k) std::fprintf(f, "%.9g%s", s.next1D(), k == 15 ? "" : ", ");
            std::fprintf(f, "]}%s\n", i + 1 == sizeof(keys)/sizeof(keys[0]) ? "" : ",");
        }
    }
    std::fprintf(f, "],\n");

    // -- camera rays + closest hits (PinholeCamera::sampleDirection, TraceableScene::intersect)
    UniformSampler gen(0x1234567, 77);
    std::fprintf(f, "\"rays\": [\n");
    std::vector<Ray> secondary;
    std::vector<IntersectionInfo> hitInfos;
    const int NumPrimary = 192;
    for (int i = 0; i < NumPrimary; ++i) {
        uint32 px = uint32(gen.next1D()*w), py = uint32(gen.next1D()*h);
        float xi0 = gen.next1D(), xi1 = gen.next1D();
        GraftPathSampler s(seed);
        s.setReplay({xi0, xi1});
        PositionSample point;
        DirectionSample dir;
        cam.samplePosition(s, point);
        cam.sampleDirection(s, point, Vec2u(px, py), dir);
        Ray ray(point.p, dir.d);
        IntersectionTemporary data;
        IntersectionInfo info;
        bool hit = ts.intersect(ray, data, info);
        std::fprintf(f, "  {\"px\": %u, \"py\": %u, \"xi\": [%.9g, %.9g], ", px, py, xi0, xi1);
        pv(f, "o", point.p); pv(f, "d", dir.d);
        std::fprintf(f, "\"tmin\": %.9g, \"hit\": %d", ray.nearT(), hit ? 1 : 0);
        if (hit) {
            std::fprintf(f, ", \"t\": %.9g, ", ray.farT());
            pv(f, "Ng", info.Ng); pv(f, "Ns", info.Ns); pv(f, "p", info.p);
            std::fprintf(f, "\"uv\": [%.9g, %.9g], \"backside\": %d, \"prim\": \"%s\"", info.uv.x(), info.uv.y(),
                         info.primitive->hitBackside(data) ? 1 : 0, info.primitive->name().c_str());
            // a diffuse-ish secondary ray from the hit point for incoherent queries
            TangentFrame frame(info.Ns);
            Vec3f wl = SampleWarp::cosineHemisphere(Vec2f(gen.next1D(), gen.next1D()));
            if (frame.normal.dot(ray.dir()) > 0.0f) wl.z() = -wl.z();
            Ray sec(info.p, frame.toGlobal(wl), info.epsilon);
            secondary.push_back(sec);
            hitInfos.push_back(info);
        }
        std::fprintf(f, "}%s\n", (i + 1 == NumPrimary && secondary.empty()) ? "" : ",");
    }
    for (size_t i = 0; i < secondary.size(); ++i) {
        Ray ray = secondary[i];
        Vec3f o = ray.pos(), d = ray.dir();
        IntersectionTemporary data;
        IntersectionInfo info;
        bool hit = ts.intersect(ray, data, info);
        std::fprintf(f, "  {");
        pv(f, "o", o); pv(f, "d", d);
        std::fprintf(f, "\"tmin\": %.9g, \"hit\": %d", ray.nearT(), hit ? 1 : 0);
        if (hit) {
            std::fprintf(f, ", \"t\": %.9g, ", ray.farT());
            pv(f, "Ng", info.Ng); pv(f, "Ns", info.Ns); pv(f, "p", info.p);
            std::fprintf(f, "\"uv\": [%.9g, %.9g], \"backside\": %d, \"prim\": \"%s\"", info.uv.x(), info.uv.y(),
                         info.primitive->hitBackside(data) ? 1 : 0, info.primitive->name().c_str());
        }
        std::fprintf(f, "}%s\n", i + 1 == secondary.size() ? "" : ",");
    }
    std::fprintf(f, "],\n");

    // -- BSDFs: eval/pdf on a direction grid and sample() with replayed numbers, per named bsdf
    std::fprintf(f, "\"bsdfs\": [\n");
    auto &bsdfs = L.scene->bsdfs();
    for (size_t bi = 0; bi < bsdfs.size(); ++bi) {
        Bsdf &bsdf = *bsdfs[bi];
        std::fprintf(f, "  {\"name\": \"%s\", \"index\": %d, \"lobes\": %u, \"cases\": [\n", bsdf.name().c_str(), int(bi),
                     *reinterpret_cast<const uint32 *>(&bsdf.lobes()));
        const int NumCases = 48;
        for (int c = 0; c < NumCases; ++c) {
            IntersectionInfo info;
            info.Ng = info.Ns = Vec3f(0.0f, 0.0f, 1.0f);
            info.p = Vec3f(0.0f); info.w = Vec3f(0.0f, 0.0f, -1.0f);
            info.uv = Vec2f(gen.next1D(), gen.next1D());
            info.epsilon = 5e-4f; info.primitive = nullptr; info.bsdf = &bsdf;
            Vec3f wi = SampleWarp::uniformSphere(Vec2f(gen.next1D(), gen.next1D()));
            Vec3f wo = SampleWarp::uniformSphere(Vec2f(gen.next1D(), gen.next1D()));
            if (c % 3 != 0) wi.z() = std::abs(wi.z());          // mostly front-side queries
            if (c % 4 == 1) wo = Vec3f(-wi.x(), -wi.y(), wi.z()); // exact mirror direction
            std::vector<float> xi = {gen.next1D(), gen.next1D(), gen.next1D(), gen.next1D(), gen.next1D(), gen.next1D()};
            uint32 requested = (c % 5 == 4) ? uint32(BsdfLobes::AllButSpecular) : uint32(BsdfLobes::AllLobes);

            GraftPathSampler s(seed);
            s.setReplay(xi);
            TangentFrame frame(info.Ns);
            SurfaceScatterEvent ev(&info, &s, frame, wi, BsdfLobes(requested), false);
            ev.wo = wo;
            Vec3f fEval = bsdf.eval(ev, false);
            float pdf = bsdf.pdf(ev);

            SurfaceScatterEvent sv(&info, &s, frame, wi, BsdfLobes(requested), false);
            bool ok = bsdf.sample(sv, false);
            std::fprintf(f, "    {\"uv\": [%.9g, %.9g], \"requested\": %u, ", info.uv.x(), info.uv.y(), requested);
            pv(f, "wi", wi); pv(f, "wo", wo); pv(f, "f", fEval);
            std::fprintf(f, "\"pdf\": %.9g, \"xi\": [", pdf);
            for (size_t k = 0; k < xi.size(); ++k) std::fprintf(f, "%.9g%s", xi[k], k + 1 == xi.size() ? "" : ", ");
            std::fprintf(f, "], \"sample_ok\": %d, \"consumed\": %d", ok ? 1 : 0, int(s.consumed()));
            if (ok) {
                std::fprintf(f, ", ");
                pv(f, "s_wo", sv.wo); pv(f, "s_weight", sv.weight);
                std::fprintf(f, "\"s_pdf\": %.9g, \"s_lobe\": %u", sv.pdf, *reinterpret_cast<const uint32 *>(&sv.sampledLobe));
            }
            std::fprintf(f, "}%s\n", c + 1 == NumCases ? "" : ",");
        }
        std::fprintf(f, "  ]}%s\n", bi + 1 == bsdfs.size() ? "" : ",");
    }
    std::fprintf(f, "],\n");

    // -- lights: sampleDirect / approximateRadiance from surface points of the scene
    std::fprintf(f, "\"lights\": [\n");
    for (size_t li = 0; li < ts.lights().size(); ++li) {
        Primitive &light = *ts.lights()[li];
        light.makeSamplable(ts, 0);
        std::fprintf(f, "  {\"index\": %d, \"name\": \"%s\", \"cases\": [\n", int(li), light.name().c_str());
        size_t n = std::min<size_t>(hitInfos.size(), 64);
        for (size_t c = 0; c < n; ++c) {
            Vec3f p = hitInfos[c].p;
            float xi0 = gen.next1D(), xi1 = gen.next1D();
            GraftPathSampler s(seed);
            s.setReplay({xi0, xi1});
            LightSample ls;
            bool ok = light.sampleDirect(0, p, s, ls);
            std::fprintf(f, "    {");
            pv(f, "p", p);
            std::fprintf(f, "\"xi\": [%.9g, %.9g], \"approx\": %.9g, \"ok\": %d", xi0, xi1, light.approximateRadiance(0, p), ok ? 1 : 0);
            if (ok) {
                std::fprintf(f, ", ");
                pv(f, "d", ls.d);
                std::fprintf(f, "\"dist\": %.9g, \"pdf\": %.9g", std::isinf(ls.dist) ? 1e30f : ls.dist, ls.pdf);
                // directPdf + evalDirect along the sampled direction (the BSDF-sampling MIS leg)
                Ray ray(p, ls.d, 5e-4f);
                IntersectionTemporary data;
                IntersectionInfo info;
                if (light.intersect(ray, data)) {
                    info.p = ray.pos() + ray.dir()*ray.farT();
                    info.w = ray.dir();
                    light.intersectionInfo(data, info);
                    std::fprintf(f, ", \"direct_pdf\": %.9g, ", light.directPdf(0, data, info, p));
                    pv(f, "emission", light.evalDirect(data, info), false);
                }
            }
            std::fprintf(f, "}%s\n", c + 1 == n ? "" : ",");
        }
        std::fprintf(f, "  ]}%s\n", li + 1 == ts.lights().size() ? "" : ",");
    }
    std::fprintf(f, "]\n}\n");
    std::fclose(f);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: ref_harness render|samples|units|integrate|sobol-table ...\n");
        return 2;
    }
    EmbreeUtil::initDevice();
    std::string cmd = argv[1];
    int rc = 2;
    if (cmd == "render") rc = cmdRender(argc, argv, false);
    else if (cmd == "samples") rc = cmdRender(argc, argv, true);
    else if (cmd == "units") rc = cmdUnits(argc, argv);
    else if (cmd == "integrate") rc = cmdIntegrate(argc, argv);
    else if (cmd == "sobol-table") rc = cmdSobolTable(argc, argv);
    else if (cmd == "draws") rc = cmdDraws(argc, argv);
    else if (cmd == "sky-image") rc = cmdSkyImage(argc, argv);
    else if (cmd == "bounds") rc = cmdBounds(argc, argv);
    if (rc == 2) std::fprintf(stderr, "ref_harness: bad arguments\n");
    return rc;
}
