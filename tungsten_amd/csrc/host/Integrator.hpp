// Integrator plugin surface -- the drop-in boundary on the C++ side.
//
// `Integrator` mirrors the reference's abstract class method for method
// (src/core/integrators/Integrator.hpp:16-63, Integrator.cpp:51-162); PathTraceHipIntegrator is
// the new "path_tracer_hip" entry a maintainer adds to the factory table
// (integrators/IntegratorFactory.cpp:14-23).  It replaces PathTraceIntegrator
// (integrators/path_tracer/PathTraceIntegrator.cpp) + PathTracer::traceSample + the Embree
// intersector + the per-tile CPU thread pool by calls into the extern "C" shim
// (include/tungsten_hip.h).  There is no CPU fallback: without a HIP device every render
// entry point throws std::runtime_error (the reference's FAIL convention, Debug.hpp:26-33).
#ifndef TGAMD_INTEGRATOR_HPP_
#define TGAMD_INTEGRATOR_HPP_

#include "Scene.hpp"
#include "Sampling.hpp"
#include "../../../include/tungsten_hip.h"

#include <atomic>
#include <functional>
#include <iosfwd>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace tungsten_amd {

class TraceableScene;

class Integrator
{
protected:
    TraceableScene *_scene = nullptr;
    uint32_t _currentSpp = 0, _nextSpp = 0;

    void advanceSpp();                                                    // Integrator.cpp:51-54
    void writeBuffers(const std::string &suffix, bool overwrite);        // Integrator.cpp:56-80

public:
    virtual ~Integrator() {}

    virtual void fromJson(const JsonValue &value, const Scene &scene) = 0;
    virtual void prepareForRender(TraceableScene &scene, uint32_t seed) = 0;
    virtual void teardownAfterRender() = 0;

    virtual void startRender(std::function<void()> completionCallback) = 0;
    virtual void waitForCompletion() = 0;
    virtual void abortRender() = 0;
    virtual bool supportsResumeRender() const { return false; }
    // resume payload behind the framebuffers (Integrator.hpp:25-26); the framebuffer itself goes through these two
    virtual void saveState(std::ostream &) {}
    virtual void loadState(std::istream &) {}
    virtual void restoreFramebuffer(const std::vector<float> &, const std::vector<uint32_t> &) {}
    virtual void currentFramebuffer(std::vector<float> &sum, std::vector<uint32_t> &count) { sum.clear(); count.clear(); }
    // the auxiliary output buffers (renderer.output_buffers; Camera::serialize/deserializeOutputBuffers, Camera.cpp:222-238)
    virtual void currentOutputBuffers(std::vector<TgHipAuxPixel> &aux) { aux.clear(); }
    virtual void restoreOutputBuffers(const std::vector<TgHipAuxPixel> &) {}
    void saveOutputBuffers();                                             // Camera::saveOutputBuffers (Camera.cpp:213-220)

    // per-pixel mean radiance, row-major, y down (Camera::getLinear, cameras/Camera.hpp:163-172)
    virtual const std::vector<float> &linearImage() = 0;

    // seed of the per-path sample streams: part of what a resume file must agree on (sceneHash)
    virtual uint32_t samplerSeed() const { return 0; }

    void saveOutputs();                                                   // Integrator.cpp:82-85
    void saveCheckpoint();                                                // Integrator.cpp:87-90
    void saveRenderResumeData();                                          // Integrator.cpp:108-128
    bool resumeRender();                                                  // Integrator.cpp:130-162
    bool done() const;                                                    // Integrator.hpp:44-47
    uint32_t currentSpp() const { return _currentSpp; }
    uint32_t nextSpp() const { return _nextSpp; }
};

class PathTraceHipIntegrator : public Integrator
{
    IntegratorSettings _settings;
    std::vector<tghip_ctx *> _ctxs;       // one per device ("devices" key; tiles shard round-robin)
    uint32_t _seed = 0;
    uint32_t _w = 0, _h = 0;

    std::thread _worker;                  // completion callback fires from a worker thread,
    std::exception_ptr _workerError;      // like the reference's pool thread (TaskGroup.hpp:55-75)
    std::atomic<bool> _abort{false};

    PassScheduler _scheduler;             // tile seeds, SampleRecords, adaptive sample distribution
    bool _useSobol = false, _useAdaptive = false;
    bool _useAux = false;                 // the scene asks for output buffers: passes carry TGHIP_PASS_AUX
    std::vector<uint32_t> _recordIndex, _recordCount;      // borrowed by the device until the pass completes
    std::vector<std::vector<TgHipSampleRecord>> _deviceRecords;

    std::vector<float> _sum;              // host copy of the device framebuffer (sum, count)
    std::vector<uint32_t> _count;
    std::vector<float> _linear;
    bool _imageDirty = true;

    void check(int rc, tghip_ctx *ctx, const char *what);
    void fetchFramebuffer();

public:
    PathTraceHipIntegrator();
    ~PathTraceHipIntegrator();

    void fromJson(const JsonValue &value, const Scene &scene) override;
    void prepareForRender(TraceableScene &scene, uint32_t seed) override;
    void teardownAfterRender() override;
    void startRender(std::function<void()> completionCallback) override;
    void waitForCompletion() override;
    void abortRender() override;
    const std::vector<float> &linearImage() override;
    uint32_t samplerSeed() const override { return _seed; }
    bool supportsResumeRender() const override { return true; }          // PathTraceIntegrator.cpp:215-218
    void saveState(std::ostream &out) override;                           // PathTraceIntegrator.cpp:158-172 (records + samplers)
    void loadState(std::istream &in) override;
    void restoreFramebuffer(const std::vector<float> &sum, const std::vector<uint32_t> &count) override;
    void currentFramebuffer(std::vector<float> &sum, std::vector<uint32_t> &count) override;
    void currentOutputBuffers(std::vector<TgHipAuxPixel> &aux) override;
    void restoreOutputBuffers(const std::vector<TgHipAuxPixel> &aux) override;

    const IntegratorSettings &settings() const { return _settings; }
    void setSettings(const IntegratorSettings &s) { _settings = s; }
    const PassScheduler &scheduler() const { return _scheduler; }
    tghip_ctx *context(size_t i = 0) { return i < _ctxs.size() ? _ctxs[i] : nullptr; }
    // raw accumulation buffers (sum of radiance and sample count per pixel)
    const std::vector<float> &sumBuffer() { fetchFramebuffer(); return _sum; }
    const std::vector<uint32_t> &countBuffer() { fetchFramebuffer(); return _count; }
};

// integrators/IntegratorFactory.cpp:14-23 -- the registry the host application owns.
struct IntegratorFactory
{
    static std::shared_ptr<Integrator> instantiate(const std::string &type);
    static std::vector<std::string> names();
};

} // namespace tungsten_amd

#endif
