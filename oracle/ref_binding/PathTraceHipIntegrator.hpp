// oracle/ref_binding/PathTraceHipIntegrator.hpp -- TEST INFRASTRUCTURE (compiled against /root/reference, never by the product).
//
// The class INTEGRATION.md section 2 asks a Tungsten maintainer to add: a subclass of the reference's own Integrator
// (integrators/Integrator.hpp:16-63) that drives libtungsten_hip.so through its C-ABI.  oracle/Makefile.ref links it, together
// with HipSceneFlattener and a copy of integrators/IntegratorFactory.cpp that lists it ("path_tracer_hip"; generated into
// oracle/_ref/gen/ by sed, never committed), into oracle/_ref/tungsten_hip_ref -- the reference's own `tungsten` program
// (src/tungsten/tungsten.cpp, unchanged) whose scenes may say "integrator": {"type": "path_tracer_hip", ...}.
#ifndef PATHTRACEHIPINTEGRATOR_HPP_
#define PATHTRACEHIPINTEGRATOR_HPP_

#include "integrators/Integrator.hpp"
#include "integrators/path_tracer/PathTracerSettings.hpp"

#include "HipSceneFlattener.hpp"

#include <exception>
#include <memory>
#include <thread>
#include <vector>

namespace Tungsten {

class PathTraceHipIntegrator : public Integrator
{
    PathTracerSettings _settings;            // same JSON keys as "path_tracer" (+ the optional "devices")
    int _devices;
    std::vector<tghip_ctx *> _ctxs;
    std::unique_ptr<HipSceneFlattener> _flat;
    tgh_scheduler *_scheduler;               // diceTiles / generateWork / adaptive sampling (PathTraceIntegrator.cpp:27-134), the library's host side
    uint32 _seed;
    uint32 _w, _h;
    std::thread _worker;
    std::exception_ptr _error;
    std::vector<uint32_t> _recordIndex, _recordCount;
    std::vector<float> _sum;
    std::vector<uint32_t> _count;

    void check(int rc, tghip_ctx *c, const char *what);
    void commitPass();

    virtual void saveState(OutputStreamHandle &out) override;
    virtual void loadState(InputStreamHandle &in) override;

public:
    PathTraceHipIntegrator();
    ~PathTraceHipIntegrator();

    virtual void fromJson(JsonPtr value, const Scene &scene) override;
    virtual rapidjson::Value toJson(Allocator &allocator) const override;

    virtual void prepareForRender(TraceableScene &scene, uint32 seed) override;
    virtual void teardownAfterRender() override;

    virtual void startRender(std::function<void()> completionCallback) override;
    virtual void waitForCompletion() override;
    virtual void abortRender() override;

    virtual bool supportsResumeRender() const override { return true; }    // `tungsten -c / -r` (src/tungsten/Shared.hpp:256-320): saveState / loadState below
};

}

#endif
