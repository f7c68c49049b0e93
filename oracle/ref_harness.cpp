// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE.  Our own driver program that links the
// *reference implementation* (libcore.a etc. built by oracle/Makefile.ref from /root/reference)
// and produces the golden vectors under tests/golden/ that pin oracle/oracle.c:
//
//   ref_harness render  <scene.json> <seed> <spp> <out.pfm> [threads]
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
//
// Nothing here is copied from the reference; it only calls its public classes.
#include "integrators/path_tracer/PathTraceIntegrator.hpp"
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
        return _sampler.next1D();
    }
    virtual bool nextBoolean(float pTrue) override final { return next1D() < pTrue; }
    virtual int nextDiscrete(int numChoices) override final { return int(next1D()*numChoices); }
    virtual Vec2f next2D() override final { float a = next1D(); float b = next1D(); return Vec2f(a, b); }
    virtual UniformSampler &uniformGenerator() override final { return _sampler; }
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

    std::vector<float> mean(size_t(w)*h*3, 0.0f);
    std::vector<float> samples;
    if (dumpSamples) samples.resize(size_t(w)*h*spp*3);
    std::vector<std::thread> pool;
    std::vector<uint64> draws(threads, 0);
    for (int t = 0; t < threads; ++t) {
        pool.emplace_back([&, t]() {
            PathTracer tracer(L.ts.get(), L.pti->settings(), uint32(t));
            GraftPathSampler sampler(seed);
            for (int y = t; y < h; y += threads) {
                for (int x = 0; x < w; ++x) {
                    uint32 pixelIndex = uint32(x + y*w);
                    float sum[3] = {0, 0, 0};
                    uint32 count = 0;
                    for (int s = 0; s < spp; ++s) {
                        sampler.startPath(pixelIndex, uint32(s));
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
            draws[t] = sampler.draws;
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

// ---- known-answer vectors -------------------------------------------------------------------
static void pv(FILE *f, const char *name, const Vec3f &v, bool comma = true)
{
    std::fprintf(f, "\"%s\": [%.9g, %.9g, %.9g]%s", name, v.x(), v.y(), v.z(), comma ? ", " : "");
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
        std::fprintf(stderr, "usage: ref_harness render|samples|units ...\n");
        return 2;
    }
    EmbreeUtil::initDevice();
    std::string cmd = argv[1];
    int rc = 2;
    if (cmd == "render") rc = cmdRender(argc, argv, false);
    else if (cmd == "samples") rc = cmdRender(argc, argv, true);
    else if (cmd == "units") rc = cmdUnits(argc, argv);
    if (rc == 2) std::fprintf(stderr, "ref_harness: bad arguments\n");
    return rc;
}
