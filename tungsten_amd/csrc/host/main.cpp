// tungsten_hip -- command-line renderer.  Same role and log lines as the reference's `tungsten`
// binary (src/tungsten/tungsten.cpp:6-24, StandaloneRenderer in src/tungsten/Shared.hpp:98-368):
// load scene -> makeTraceable(seed) -> while (!done) { startRender; waitForCompletion } -> saveOutputs.
#include "ImageIO.hpp"
#include "Integrator.hpp"
#include "Scene.hpp"
#include "TraceableScene.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace tungsten_amd;

static void usage()
{
    std::printf(
        "Usage: tungsten_hip [options] scene1 [scene2 ...]\n"
        "  -s, --seed N         random seed (default 0xBA5EBA11)\n"
        "      --spp N          override samples per pixel\n"
        "  -o, --output-file F  LDR output (PNG)\n"
        "  -e, --hdr-output-file F  HDR output (PFM)\n"
        "      --devices N      number of GPUs to shard tiles over (default 1)\n"
        "  -h, --help\n");
}

int main(int argc, char **argv)
{
    uint32_t seed = 0xBA5EBA11u;      // Shared.hpp:246
    int spp = -1, devices = 0;
    std::string outFile, hdrFile;
    std::vector<std::string> scenes;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { usage(); std::exit(2); } return argv[++i]; };
        if (a == "-s" || a == "--seed") seed = uint32_t(std::strtoul(next(), nullptr, 0));
        else if (a == "--spp") spp = std::atoi(next());
        else if (a == "-o" || a == "--output-file") outFile = next();
        else if (a == "-e" || a == "--hdr-output-file") hdrFile = next();
        else if (a == "--devices") devices = std::atoi(next());
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else scenes.push_back(a);
    }
    if (scenes.empty()) { usage(); return 2; }

    for (const std::string &path : scenes) {
        std::printf("Loading scene '%s'...\n", path.c_str());
        try {
            std::unique_ptr<Scene> scene = Scene::load(path);
            if (spp > 0) scene->renderer.spp = uint32_t(spp);
            if (!outFile.empty()) scene->renderer.outputFile = outFile;
            if (!hdrFile.empty()) scene->renderer.hdrOutputFile = hdrFile;
            if (devices > 0) scene->integrator.devices = devices;

            std::shared_ptr<Integrator> integrator = IntegratorFactory::instantiate(scene->integrator.type);
            if (PathTraceHipIntegrator *hip = dynamic_cast<PathTraceHipIntegrator *>(integrator.get()))
                hip->setSettings(scene->integrator);
            TraceableScene flattened(*scene, integrator.get(), seed);
            std::printf("Scene flattened: %u BVH nodes (depth %d), %u primitive records, %zu lights, %.3f s\n",
                        flattened.desc().num_nodes, flattened.bvhDepth(), flattened.desc().num_recs,
                        flattened.numLights(), flattened.buildSeconds());

            std::printf("Starting render...\n");
            auto t0 = std::chrono::steady_clock::now();
            while (!integrator->done()) {
                integrator->startRender([]() {});
                integrator->waitForCompletion();
                std::printf("Completed %u/%u spp\n", integrator->currentSpp(), scene->renderer.spp);
            }
            double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            double samples = double(scene->camera.resX)*scene->camera.resY*scene->renderer.spp;
            std::printf("Finished render. Render time %.3fs (%.2f Msamples/s)\n", secs, samples/secs*1e-6);
            integrator->saveOutputs();
        } catch (const std::exception &e) {
            std::fprintf(stderr, "%s\n", e.what());
            return 1;
        }
    }
    return 0;
}
