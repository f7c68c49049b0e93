// oracle/ref_binding/PathTraceHipIntegrator.cpp -- TEST INFRASTRUCTURE.  See PathTraceHipIntegrator.hpp.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

// commitPass() writes the pass's result straight into the camera's colour OutputBuffer, which has no bulk setter
// (cameras/OutputBuffer.hpp:104-132): open the two class definitions up
#include "OpenUp.hpp"
#define private public
#define protected public
#define class struct      /* members declared before the first access specifier (enum class -> enum struct is the same thing) */
#include "cameras/OutputBuffer.hpp"
#include "cameras/Camera.hpp"
#undef private
#undef protected
#undef class
#include "renderer/TraceableScene.hpp"
#include "io/JsonObject.hpp"
#include "io/FileUtils.hpp"
#include "Debug.hpp"

#include "PathTraceHipIntegrator.hpp"

namespace Tungsten {

PathTraceHipIntegrator::PathTraceHipIntegrator()
: _devices(1), _scheduler(nullptr), _seed(0), _w(0), _h(0)
{
}

PathTraceHipIntegrator::~PathTraceHipIntegrator()
{
    teardownAfterRender();
}

void PathTraceHipIntegrator::check(int rc, tghip_ctx *c, const char *what)
{
    if (rc != TGHIP_OK)
        FAIL("path_tracer_hip: %s: %s", what, tghip_last_error(c));
}

void PathTraceHipIntegrator::fromJson(JsonPtr value, const Scene &/*scene*/)
{
    _settings.fromJson(value);
    value.getField("devices", _devices);
}

rapidjson::Value PathTraceHipIntegrator::toJson(Allocator &allocator) const
{
    return JsonObject{_settings.toJson(allocator), allocator,
        "type", "path_tracer_hip",
        "devices", _devices
    };
}

// Integrator::saveRenderResumeData / resumeRender (integrators/Integrator.cpp:108-162) call these behind the camera's output buffers.  The
// reference's path tracer stores its SampleRecords and one sequential sampler per tile (PathTraceIntegrator.cpp:158-172); here: the
// framebuffer as the device keeps it -- radiance SUMS and sample counts: the camera's colour buffer holds means, and mean x count is not the
// sum bit for bit --, the SampleRecords, and the state of the scheduler's sampler (the per-path streams are counter-based and need none).
void PathTraceHipIntegrator::saveState(OutputStreamHandle &out)
{
    const uint64 magic = 0x3452504948474E54ull;                    // "TNGHIPR4"
    const uint64 n = uint64(_w)*_h, nr = tgh_scheduler_num_records(_scheduler), state = tgh_scheduler_sampler_state(_scheduler);
    FileUtils::streamWrite(out, magic);
    FileUtils::streamWrite(out, n);
    FileUtils::streamWrite(out, _sum.data(), size_t(n)*3);
    FileUtils::streamWrite(out, _count.data(), size_t(n));
    FileUtils::streamWrite(out, nr);
    FileUtils::streamWrite(out, tgh_scheduler_records(_scheduler), size_t(nr));
    FileUtils::streamWrite(out, state);
}
void PathTraceHipIntegrator::loadState(InputStreamHandle &in)
{
    uint64 magic = 0, n = 0, nr = 0, state = 0;
    FileUtils::streamRead(in, magic);
    FileUtils::streamRead(in, n);
    if (magic != 0x3452504948474E54ull || n != uint64(_w)*_h)
        FAIL("path_tracer_hip: the render resume state was not written by this integrator for this image size");
    FileUtils::streamRead(in, _sum.data(), size_t(n)*3);
    FileUtils::streamRead(in, _count.data(), size_t(n));
    FileUtils::streamRead(in, nr);
    if (nr != tgh_scheduler_num_records(_scheduler))
        FAIL("path_tracer_hip: the render resume state holds another number of sample records");
    TgHostSampleRecord *rec = tgh_scheduler_records(_scheduler);
    FileUtils::streamRead(in, rec, size_t(nr));
    FileUtils::streamRead(in, state);
    tgh_scheduler_set_sampler_state(_scheduler, state);
    // the merged framebuffer goes to the first device, the others restart from zero (ownership of a pixel only matters for the samples
    // still to come); every device gets the complete Welford state and keeps updating the records of its own tiles
    std::vector<TgHipSampleRecord> dev(static_cast<size_t>(nr));
    for (size_t i = 0; i < dev.size(); ++i) { dev[i].sample_count = rec[i].sample_count; dev[i].mean = rec[i].mean; dev[i].running_variance = rec[i].running_variance; }
    for (size_t d = 0; d < _ctxs.size(); ++d) {
        check(tghip_clear_framebuffer(_ctxs[d]), _ctxs[d], "tghip_clear_framebuffer");
        if (d == 0)
            check(tghip_upload_framebuffer(_ctxs[d], _sum.data(), _count.data(), size_t(n)), _ctxs[d], "tghip_upload_framebuffer");
        if (_scene->rendererSettings().useAdaptiveSampling())
            check(tghip_upload_records(_ctxs[d], dev.data(), dev.size()), _ctxs[d], "tghip_upload_records");
    }
}

// PathTraceIntegrator::prepareForRender (PathTraceIntegrator.cpp:184-201): here the device is acquired and the flattened
// scene uploaded
void PathTraceHipIntegrator::prepareForRender(TraceableScene &scene, uint32 seed)
{
    teardownAfterRender();
    _scene = &scene;
    _seed = seed;
    _currentSpp = 0;
    advanceSpp();
    scene.cam().requestColorBuffer();
    _w = scene.cam().resolution().x();
    _h = scene.cam().resolution().y();
    if (!scene.rendererSettings().renderOutputs().empty())
        FAIL("path_tracer_hip (test binding): renderer.output_buffers are committed by the stand-alone host only (INTEGRATION.md section 4)");
    if (!_settings.lowOrderScattering || !_settings.includeSurfaces)
        FAIL("path_tracer_hip: low_order_scattering / include_surfaces must keep their defaults");

    _flat.reset(new HipSceneFlattener());
    try {
        _flat->build(scene, _settings, _settings.enableVolumeLightSampling);
    } catch (const std::exception &e) {
        FAIL("%s", e.what());
    }
    // (enable_light_sampling lives in PathTracerSettings, not TraceSettings)
    const_cast<TgHipSceneDesc &>(_flat->desc()).settings.enable_light_sampling = _settings.enableLightSampling ? 1 : 0;

    // test hook: the flattened scene as a file of (name, byte count, bytes) records, written before any device is touched, so that
    // tests/test_ref_binding.py can compare it array by array with what the library's own loader makes of the same JSON -- on a box
    // without a GPU too
    if (const char *dump = std::getenv("TGHIP_REF_DUMP_DESC")) {
        const TgHipSceneDesc &d = _flat->desc();
        std::ofstream out(dump, std::ios::binary);
        auto put = [&](const char *name, const void *data, uint64_t bytes) {
            uint32_t n = uint32_t(std::strlen(name));
            out.write(reinterpret_cast<const char *>(&n), 4); out.write(name, n);
            out.write(reinterpret_cast<const char *>(&bytes), 8);
            if (bytes) out.write(static_cast<const char *>(data), std::streamsize(bytes));
        };
        put("nodes", d.nodes, uint64_t(d.num_nodes)*sizeof(TgHipBvhNode));
        put("wide_nodes", d.wide_nodes, uint64_t(d.num_wide_nodes)*sizeof(TgHipWideNode));
        put("recs", d.recs, uint64_t(d.num_recs)*sizeof(TgHipPrimRec));
        put("tri_attrs", d.tri_attrs, uint64_t(d.num_recs)*sizeof(TgHipTriAttr));
        put("objects", d.objects, uint64_t(d.num_objects)*sizeof(TgHipObject));
        put("lights", d.lights, uint64_t(d.num_lights)*4u);
        put("infinite_lights", d.infinite_lights, uint64_t(d.num_infinite_lights)*4u);
        put("bsdfs", d.bsdfs, uint64_t(d.num_bsdfs)*sizeof(TgHipBsdf));
        put("textures", d.textures, uint64_t(d.num_textures)*sizeof(TgHipTexture));
        put("texels", d.texels, d.num_texel_floats*4u);
        put("dist", d.dist, d.num_dist_floats*4u);
        put("camera", &d.camera, sizeof(d.camera));
        put("settings", &d.settings, sizeof(d.settings));
        put("bounds", d.bounds_lo, 24u);
        put("sobol", d.sobol_matrices, d.num_sobol_words*4u);
        put("media", d.media, uint64_t(d.num_media)*sizeof(TgHipMedium));
        put("light_tris", d.light_tris, d.num_light_tri_floats*4u);
        put("inst_prims", d.inst_prims, uint64_t(d.num_inst_prims)*4u);
        put("inst_leaf_boxes", d.inst_leaf_boxes, uint64_t(d.num_inst_prims)*32u);
        put("inst_tight_boxes", d.inst_tight_boxes, d.num_instances ? uint64_t(d.num_top_recs)*32u : 0u);
        const uint32_t counts[2] = {d.num_top_recs, d.num_instances};
        put("counts", counts, sizeof(counts));
    }

    // test hook (tests/test_ref_binding.py, no GPU): the plumbing around the device -- the pass loop, Integrator::saveRenderResumeData /
    // resumeRender with saveState / loadState -- without one: no context is created, a pass renders nothing
    const bool dry = std::getenv("TGHIP_REF_DRY_RUN") != nullptr;
    int available = dry ? 0 : tghip_device_count();
    if (available <= 0 && !dry)
        FAIL("path_tracer_hip: no HIP device available (there is no CPU fallback)");
    int n = dry ? 0 : std::max(1, std::min(_devices, available));
    for (int d = 0; d < n; ++d) {
        tghip_ctx *c = tghip_create(d);
        if (!c)
            FAIL("path_tracer_hip: tghip_create: %s", tghip_last_error(nullptr));
        _ctxs.push_back(c);
        check(tghip_upload_scene(c, &_flat->desc()), c, "tghip_upload_scene");
    }
    _scheduler = tgh_scheduler_create(_w, _h, seed);
    _sum.assign(size_t(_w)*_h*3, 0.0f);
    _count.assign(size_t(_w)*_h, 0u);
}

void PathTraceHipIntegrator::teardownAfterRender()
{
    if (_worker.joinable())
        _worker.join();
    for (tghip_ctx *c : _ctxs)
        tghip_destroy(c);
    _ctxs.clear();
    if (_scheduler)
        tgh_scheduler_free(_scheduler);
    _scheduler = nullptr;
    _flat.reset();
}

// the pass's samples are on the device(s): sum + count -> the camera's colour buffer (mean in _bufferA, _sampleCount), so that
// Camera::getLinear, Integrator::saveOutputs / saveCheckpoint and the HTTP server's /render work unchanged
void PathTraceHipIntegrator::commitPass()
{
    const size_t n = size_t(_w)*_h;
    bool reduced = _ctxs.size() > 1 &&
        tghip_reduce_framebuffers(_ctxs.data(), int(_ctxs.size()), 0, _sum.data(), _count.data(), n) == TGHIP_OK;
    if (!reduced) {
        std::fill(_sum.begin(), _sum.end(), 0.0f);
        std::fill(_count.begin(), _count.end(), 0u);
        std::vector<float> s(n*3);
        std::vector<uint32_t> c(n);
        for (tghip_ctx *ctx : _ctxs) {
            check(tghip_download_framebuffer(ctx, s.data(), c.data(), n), ctx, "tghip_download_framebuffer");
            for (size_t i = 0; i < n*3; ++i) _sum[i] += s[i];      // tile ownership is disjoint: x + 0
            for (size_t i = 0; i < n; ++i) _count[i] += c[i];
        }
    }
    OutputBufferVec3f *buffer = const_cast<Camera &>(_scene->cam()).colorBuffer();
    for (size_t i = 0; i < n; ++i) {
        const float cnt = float(std::max(_count[i], 1u));
        buffer->_bufferA[i] = Vec3f(_sum[i*3]/cnt, _sum[i*3 + 1]/cnt, _sum[i*3 + 2]/cnt);
        buffer->_sampleCount[i] = _count[i];
    }
}

// Asynchronous like the reference (PathTraceIntegrator.cpp:220-239): returns immediately, the completion callback fires from a
// worker thread once every device has finished its tile shard
void PathTraceHipIntegrator::startRender(std::function<void()> completionCallback)
{
    if (_worker.joinable())
        _worker.join();
    const bool sobol = _scene->rendererSettings().useSobol(), adaptive = _scene->rendererSettings().useAdaptiveSampling();
    if (done() || !tgh_scheduler_generate_work(_scheduler, _currentSpp, _nextSpp, adaptive ? 1 : 0) || _ctxs.empty()) {
        _currentSpp = _nextSpp;
        advanceSpp();
        completionCallback();
        return;
    }
    if (adaptive) {
        // renderTile (:136-156): every pixel of a record traces samples [sampleIndex, sampleIndex + nextSampleCount)
        const size_t nr = tgh_scheduler_num_records(_scheduler);
        const TgHostSampleRecord *rec = tgh_scheduler_records(_scheduler);
        _recordIndex.resize(nr); _recordCount.resize(nr);
        for (size_t i = 0; i < nr; ++i) { _recordIndex[i] = rec[i].sample_index; _recordCount[i] = rec[i].next_sample_count; }
    }
    for (size_t d = 0; d < _ctxs.size(); ++d) {
        TgHipPassDesc p;
        std::memset(&p, 0, sizeof(p));
        p.spp_begin = _currentSpp; p.spp_end = _nextSpp;
        p.seed = _seed;
        p.shard_index = uint32_t(d); p.shard_count = uint32_t(_ctxs.size());
        if (sobol) { p.flags |= TGHIP_PASS_SOBOL; p.tile_seeds = tgh_scheduler_tile_seeds(_scheduler); }
        if (adaptive) { p.flags |= TGHIP_PASS_RECORDS; p.record_index = _recordIndex.data(); p.record_count = _recordCount.data(); }
        check(tghip_render_pass(_ctxs[d], &p), _ctxs[d], "tghip_render_pass");
    }
    _error = nullptr;
    _worker = std::thread([this, completionCallback, adaptive]() {
        try {
            std::vector<int> rcs(_ctxs.size(), TGHIP_OK);
            std::vector<std::thread> drivers;                      // one host thread per device drives that device's loop
            for (size_t d = 1; d < _ctxs.size(); ++d)
                drivers.emplace_back([this, d, &rcs]() { rcs[d] = tghip_wait(_ctxs[d]); });
            rcs[0] = tghip_wait(_ctxs[0]);
            for (std::thread &t : drivers) t.join();
            for (size_t d = 0; d < _ctxs.size(); ++d) {
                if (rcs[d] == TGHIP_E_ABORTED) return;            // no finisher on abort (thread/TaskGroup.hpp:77-83)
                check(rcs[d], _ctxs[d], "tghip_wait");
            }
            if (adaptive) {
                // SampleRecord::addSample ran on the device (TGHIP_PASS_RECORDS): 12 bytes per record come back; every record is
                // non-zero on exactly the device that owns its tile
                const size_t nr = tgh_scheduler_num_records(_scheduler);
                TgHostSampleRecord *rec = tgh_scheduler_records(_scheduler);
                std::vector<TgHipSampleRecord> dev(nr);
                for (size_t d = 0; d < _ctxs.size(); ++d) {
                    check(tghip_download_records(_ctxs[d], dev.data(), nr), _ctxs[d], "tghip_download_records");
                    for (size_t i = 0; i < nr; ++i)
                        if (dev[i].sample_count) { rec[i].sample_count = dev[i].sample_count; rec[i].mean = dev[i].mean; rec[i].running_variance = dev[i].running_variance; }
                }
            }
            commitPass();
        } catch (...) {
            _error = std::current_exception();
            return;
        }
        _currentSpp = _nextSpp;
        advanceSpp();
        completionCallback();
    });
}

void PathTraceHipIntegrator::waitForCompletion()
{
    if (_worker.joinable())
        _worker.join();
    if (_error) {
        std::exception_ptr e = _error;
        _error = nullptr;
        std::rethrow_exception(e);            // TaskGroup::wait rethrows (thread/TaskGroup.hpp:70-75)
    }
}

void PathTraceHipIntegrator::abortRender()
{
    for (tghip_ctx *c : _ctxs)
        tghip_abort(c);
    if (_worker.joinable())
        _worker.join();
    _error = nullptr;
}

}
