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
        "  -c, --checkpoint S   save a checkpoint (images + resume state) when S seconds of rendering have passed since the last one\n"
        "  -r, --restart        ignore a saved render state and start from 0 spp\n"
        "  -h, --help\n");
}

int main(int argc, char **argv)
{
    uint32_t seed = 0xBA5EBA11u;      // Shared.hpp:246
    int spp = -1, devices = 0;
    double checkpointInterval = 0.0;
    bool restart = false;
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
        else if (a == "-c" || a == "--checkpoint") checkpointInterval = std::atof(next());
        else if (a == "-r" || a == "--restart") restart = true;
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

            // Shared.hpp:256-279: "enable_resume_render" picks up renderer.resume_render_file unless --restart is given
            const bool resumeRender = scene->renderer.enableResumeRender && integrator->supportsResumeRender();
            if (resumeRender && !restart) {
                std::printf("Trying to resume render from saved state... ");
                if (integrator->resumeRender()) std::printf("Resume successful at %u spp\n", integrator->currentSpp());
                else                            std::printf("Resume unsuccessful. Starting from 0 spp\n");
            }

            std::printf("Starting render...\n");
            auto t0 = std::chrono::steady_clock::now();
            auto lastCheckpoint = t0;
            while (!integrator->done()) {
                integrator->startRender([]() {});
                integrator->waitForCompletion();
                std::printf("Completed %u/%u spp\n", integrator->currentSpp(), scene->renderer.spp);
                // Shared.hpp:296-311
                auto now = std::chrono::steady_clock::now();
                if (checkpointInterval > 0.0 && std::chrono::duration<double>(now - lastCheckpoint).count() > checkpointInterval) {
                    std::printf("Saving checkpoint\n");
                    integrator->saveCheckpoint();
                    if (resumeRender)
                        integrator->saveRenderResumeData();
                    lastCheckpoint = std::chrono::steady_clock::now();
                }
            }
            double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            double samples = double(scene->camera.resX)*scene->camera.resY*scene->renderer.spp;
            std::printf("Finished render. Render time %.3fs (%.2f Msamples/s)\n", secs, samples/secs*1e-6);
            integrator->saveOutputs();
            if (resumeRender)
                integrator->saveRenderResumeData();          // Shared.hpp:319-320
        } catch (const std::exception &e) {
            std::fprintf(stderr, "%s\n", e.what());
            return 1;
        }
    }
    return 0;
}
