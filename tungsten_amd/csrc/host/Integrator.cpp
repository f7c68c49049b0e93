#include "Integrator.hpp"
#include "ImageIO.hpp"
#include "TraceableScene.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <stdexcept>

namespace tungsten_amd {

// ------------------------------------------------------------------------------------------
// Integrator base (src/core/integrators/Integrator.cpp:51-85)
// ------------------------------------------------------------------------------------------
void Integrator::advanceSpp()
{
    _nextSpp = std::min(_currentSpp + _scene->rendererSettings().sppStep, _scene->rendererSettings().spp);
}

bool Integrator::done() const
{
    return _currentSpp >= _scene->rendererSettings().spp;
}

static bool fileExists(const std::string &p)
{
    FILE *f = std::fopen(p.c_str(), "rb");
    if (f) std::fclose(f);
    return f != nullptr;
}

// FileUtils-style "name1.png, name2.png, ..." when overwriting is disabled (Integrator.cpp:20-49)
static std::string incrementalFilename(const std::string &dst, const std::string &suffix, bool overwrite)
{
    size_t dot = dst.find_last_of('.');
    std::string stem = dot == std::string::npos ? dst : dst.substr(0, dot);
    std::string ext = dot == std::string::npos ? std::string() : dst.substr(dot);
    std::string base = stem + suffix;
    std::string path = base + ext;
    if (overwrite)
        return path;
    int index = 0;
    while (fileExists(path))
        path = base + std::to_string(++index) + ext;
    return path;
}

void Integrator::writeBuffers(const std::string &suffix, bool overwrite)
{
    const Camera &cam = _scene->cam();
    const RendererSettings &settings = _scene->rendererSettings();
    const std::vector<float> &hdr = linearImage();
    size_t n = size_t(cam.resX)*cam.resY;

    if (!settings.outputFile.empty()) {
        std::vector<uint8_t> ldr(n*3);
        for (size_t i = 0; i < n; ++i) {
            Vec3f c(std::max(hdr[i*3], 0.0f), std::max(hdr[i*3 + 1], 0.0f), std::max(hdr[i*3 + 2], 0.0f));
            Vec3f t = ImageIO::tonemap(cam.tonemap, c)*255.0f;
            for (int k = 0; k < 3; ++k)
                ldr[i*3 + k] = uint8_t(std::min(std::max(int(t[k]), 0), 255));
        }
        std::string path = incrementalFilename(settings.outputFile, suffix, overwrite);
        if (path.size() < 4 || path.substr(path.size() - 4) != ".png")
            path += ".png";   // only the PNG writer is in scope
        ImageIO::savePng(path, ldr.data(), int(cam.resX), int(cam.resY));
    }
    if (!settings.hdrOutputFile.empty())
        ImageIO::savePfm(incrementalFilename(settings.hdrOutputFile, suffix, overwrite), hdr.data(), int(cam.resX), int(cam.resY), 3);
    if (suffix.empty() && !settings.outputs.empty())   // Integrator.cpp:78-79
        saveOutputBuffers();
}

// OutputBuffer<T>::save for every requested output (cameras/OutputBuffer.hpp:146-189, saveLdr :56-86).  The device keeps
// every output as A / B halves + Welford sum; what a buffer without two_buffer_variance would hold in _bufferA is the
// mean of both halves.
void Integrator::saveOutputBuffers()
{
    const Camera &cam = _scene->cam();
    const int w = int(cam.resX), h = int(cam.resY);
    const size_t n = size_t(w)*h;
    std::vector<TgHipAuxPixel> aux;
    currentOutputBuffers(aux);
    if (aux.size() != n)
        return;
    static const int first[5] = {0, 3, 4, 7, 10}, channels[5] = {3, 1, 3, 3, 1};
    auto withTag = [](const std::string &file, const char *tag) {
        size_t dot = file.find_last_of('.');
        return dot == std::string::npos ? file + tag : file.substr(0, dot) + tag + file.substr(dot);
    };
    for (const OutputBufferSettings &b : _scene->rendererSettings().outputs) {
        const int ch0 = first[b.type], nch = channels[b.type];
        std::vector<float> mean(n*nch), bufA(n*nch), bufB(n*nch), var(n*nch);
        for (size_t i = 0; i < n; ++i) {
            const TgHipAuxPixel &p = aux[i];
            uint32_t cnt = p.count[b.type], cntA = (cnt + 1)/2, cntB = cnt/2;
            for (int k = 0; k < nch; ++k) {
                float a = p.a[ch0 + k], bb = p.b[ch0 + k];
                bufA[i*nch + k] = a; bufB[i*nch + k] = bb;
                mean[i*nch + k] = (a*float(cntA) + bb*float(cntB))/float(std::max(cnt, 1u));      // operator[] (:134-144)
                var[i*nch + k] = p.variance[ch0 + k]/float(cnt*std::max(1u, cnt - 1));             // save() (:178-181)
            }
        }
        auto saveHdr = [&](const std::string &file, const std::vector<float> &img) {
            if (!file.empty()) ImageIO::savePfm(file, img.data(), w, h, nch);
        };
        auto saveLdr = [&](const std::string &file, const std::vector<float> &img, bool rescale) {   // OutputBuffer::saveLdr
            if (file.empty()) return;
            float minimum = 0.0f, maximum = 0.0f;
            if (b.type == TGHIP_AUX_DEPTH) {
                for (size_t i = 0; i < n; ++i)
                    if (img[i] != std::numeric_limits<float>::infinity()) maximum = std::max(maximum, img[i]);
            } else if (b.type == TGHIP_AUX_NORMAL) {
                minimum = -1.0f; maximum = 1.0f;
            } else {
                rescale = false;
            }
            std::vector<uint8_t> ldr(n*3);
            for (size_t i = 0; i < n; ++i) {
                bool bad = false;
                float f[3];
                for (int k = 0; k < 3; ++k) {
                    f[k] = img[i*nch + (nch == 3 ? k : 0)];
                    if (rescale) f[k] = (f[k] - minimum)/(maximum - minimum);
                }
                float avg = nch == 3 ? (f[0] + f[1] + f[2])/3.0f : f[0];
                bad = std::isnan(avg) || std::isinf(avg);
                for (int k = 0; k < 3; ++k)
                    ldr[i*3 + k] = bad ? 255 : uint8_t(std::min(std::max(int(f[k]*255.0f), 0), 255));
            }
            ImageIO::savePng(file, ldr.data(), w, h);
        };
        if (b.twoBufferVariance) {
            saveHdr(b.hdrOutputFile, mean);
            if (!b.hdrOutputFile.empty()) { saveHdr(withTag(b.hdrOutputFile, "A"), bufA); saveHdr(withTag(b.hdrOutputFile, "B"), bufB); }
            saveLdr(b.ldrOutputFile, mean, true);
            if (!b.ldrOutputFile.empty()) { saveLdr(withTag(b.ldrOutputFile, "A"), bufA, true); saveLdr(withTag(b.ldrOutputFile, "B"), bufB, true); }
        } else {
            saveHdr(b.hdrOutputFile, mean);
            saveLdr(b.ldrOutputFile, mean, true);
        }
        if (b.sampleVariance) {
            if (!b.hdrOutputFile.empty()) saveHdr(withTag(b.hdrOutputFile, "Variance"), var);
            if (!b.ldrOutputFile.empty()) saveLdr(withTag(b.ldrOutputFile, "Variance"), var, false);
        }
    }
}

void Integrator::saveOutputs()
{
    writeBuffers("", _scene->rendererSettings().overwriteOutputFiles);
}

void Integrator::saveCheckpoint()
{
    writeBuffers("_checkpoint", true);
}

// The reference hashes the scene's JSON serialisation minus the renderer block (Integrator.cpp:92-106): everything that
// determines the image except how long it is rendered.  Here the same role is played by a hash of the flattened scene
// (every array the device gets + camera + integrator settings), which changes exactly when the rendered scene changes.
static uint64_t fnv1a(uint64_t h, const void *data, size_t bytes)
{
    const unsigned char *p = static_cast<const unsigned char *>(data);
    for (size_t i = 0; i < bytes; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
// Every array and every scalar block of TgHipSceneDesc goes through this one function, plus the sampler seed.
static uint64_t sceneHash(const TgHipSceneDesc &d, uint32_t seed)
{
    uint64_t h = 14695981039346656037ull;
    auto arr = [&h](const void *p, uint64_t count, size_t elem) {
        h = fnv1a(h, &count, sizeof(count));
        if (p && count) h = fnv1a(h, p, size_t(count)*elem);
    };
    arr(d.nodes, d.num_nodes, sizeof(TgHipBvhNode));
    arr(d.recs, d.num_recs, sizeof(TgHipPrimRec));
    arr(d.tri_attrs, d.num_recs, sizeof(TgHipTriAttr));
    arr(d.objects, d.num_objects, sizeof(TgHipObject));
    arr(d.lights, d.num_lights, sizeof(int32_t));
    arr(d.infinite_lights, d.num_infinite_lights, sizeof(int32_t));
    arr(d.bsdfs, d.num_bsdfs, sizeof(TgHipBsdf));
    arr(d.textures, d.num_textures, sizeof(TgHipTexture));
    arr(d.texels, d.num_texel_floats, sizeof(float));
    arr(d.dist, d.num_dist_floats, sizeof(float));
    arr(d.light_tris, d.num_light_tri_floats, sizeof(float));
    arr(d.media, d.num_media, sizeof(TgHipMedium));
    arr(nullptr, d.num_instances, 0);
    arr(nullptr, d.num_top_recs, 0);
    arr(nullptr, d.sobol_matrices ? d.num_sobol_words : 0, 0);   // (the matrices are a constant table)
    h = fnv1a(h, &d.camera, sizeof(d.camera));
    h = fnv1a(h, &d.settings, sizeof(d.settings));
    h = fnv1a(h, d.bounds_lo, sizeof(d.bounds_lo));
    h = fnv1a(h, d.bounds_hi, sizeof(d.bounds_hi));
    h = fnv1a(h, &seed, sizeof(seed));
    return h;
}

static const char ResumeMagic[8] = {'T', 'G', 'H', 'I', 'P', 'R', 'S', '1'};

// Integrator::saveRenderResumeData (Integrator.cpp:108-128): header (current spp + the two sampler switches a resumed render
// must agree on), scene hash, framebuffers, integrator state.
void Integrator::saveRenderResumeData()
{
    const RendererSettings &rs = _scene->rendererSettings();
    std::ofstream out(rs.resumeRenderFile.c_str(), std::ios::binary);
    if (!out)
        return;                                   // the reference only logs this (Integrator.cpp:112-115)
    std::vector<float> sum;
    std::vector<uint32_t> count;
    currentFramebuffer(sum, count);
    uint32_t header[5] = {_currentSpp, rs.useAdaptiveSampling ? 1u : 0u, rs.useSobol ? 1u : 0u, _scene->cam().resX, _scene->cam().resY};
    uint64_t hash = sceneHash(_scene->desc(), samplerSeed());
    out.write(ResumeMagic, sizeof(ResumeMagic));
    out.write(reinterpret_cast<const char *>(header), sizeof(header));
    out.write(reinterpret_cast<const char *>(&hash), sizeof(hash));
    out.write(reinterpret_cast<const char *>(sum.data()), std::streamsize(sum.size()*sizeof(float)));
    out.write(reinterpret_cast<const char *>(count.data()), std::streamsize(count.size()*sizeof(uint32_t)));
    if (!rs.outputs.empty()) {                    // Camera::serializeOutputBuffers (Camera.cpp:222-229)
        std::vector<TgHipAuxPixel> aux;
        currentOutputBuffers(aux);
        out.write(reinterpret_cast<const char *>(aux.data()), std::streamsize(aux.size()*sizeof(TgHipAuxPixel)));
    }
    saveState(out);
}

// Integrator::resumeRender (Integrator.cpp:130-162): false (and nothing touched) unless the file belongs to this scene
// and to the same sampler configuration.
bool Integrator::resumeRender()
{
    const RendererSettings &rs = _scene->rendererSettings();
    std::ifstream in(rs.resumeRenderFile.c_str(), std::ios::binary);
    if (!in)
        return false;
    char magic[8];
    uint32_t header[5];
    uint64_t hash = 0;
    in.read(magic, sizeof(magic));
    in.read(reinterpret_cast<char *>(header), sizeof(header));
    in.read(reinterpret_cast<char *>(&hash), sizeof(hash));
    if (!in || std::memcmp(magic, ResumeMagic, sizeof(magic)) != 0)
        return false;
    if (header[1] != (rs.useAdaptiveSampling ? 1u : 0u) || header[2] != (rs.useSobol ? 1u : 0u))
        return false;
    if (header[3] != _scene->cam().resX || header[4] != _scene->cam().resY || hash != sceneHash(_scene->desc(), samplerSeed()))
        return false;
    size_t n = size_t(header[3])*header[4];
    std::vector<float> sum(n*3);
    std::vector<uint32_t> count(n);
    in.read(reinterpret_cast<char *>(sum.data()), std::streamsize(sum.size()*sizeof(float)));
    in.read(reinterpret_cast<char *>(count.data()), std::streamsize(count.size()*sizeof(uint32_t)));
    if (!in)
        return false;
    std::vector<TgHipAuxPixel> aux;
    if (!rs.outputs.empty()) {
        aux.resize(n);
        in.read(reinterpret_cast<char *>(aux.data()), std::streamsize(n*sizeof(TgHipAuxPixel)));
        if (!in)
            return false;
    }
    restoreFramebuffer(sum, count);
    if (!aux.empty())
        restoreOutputBuffers(aux);
    loadState(in);
    if (!in)
        throw std::runtime_error("path_tracer_hip: truncated render resume state '" + rs.resumeRenderFile + "'");
    _currentSpp = header[0];
    advanceSpp();
    return true;
}

// ------------------------------------------------------------------------------------------
// PathTraceHipIntegrator
// ------------------------------------------------------------------------------------------
PathTraceHipIntegrator::PathTraceHipIntegrator() {}

PathTraceHipIntegrator::~PathTraceHipIntegrator()
{
    teardownAfterRender();
}

void PathTraceHipIntegrator::check(int rc, tghip_ctx *ctx, const char *what)
{
    if (rc != TGHIP_OK)
        throw std::runtime_error(std::string("path_tracer_hip: ") + what + " failed: " + tghip_last_error(ctx));
}

void PathTraceHipIntegrator::fromJson(const JsonValue &value, const Scene &/*scene*/)
{
    _settings.fromJson(value);
}

// PathTraceIntegrator::prepareForRender (PathTraceIntegrator.cpp:184-201): this is where the
// device is acquired and the flattened scene uploaded.
void PathTraceHipIntegrator::prepareForRender(TraceableScene &scene, uint32_t seed)
{
    teardownAfterRender();
    _scene = &scene;
    _seed = seed;
    _currentSpp = 0;
    advanceSpp();
    _w = scene.cam().resX;
    _h = scene.cam().resY;

    int available = tghip_device_count();
    if (available <= 0)
        throw std::runtime_error("path_tracer_hip: no HIP device available (there is no CPU fallback)");
    int devices = std::max(1, _settings.shareDevices ? std::min(_settings.devices, 64) : std::min(_settings.devices, available));
    for (int d = 0; d < devices; ++d) {
        tghip_ctx *ctx = tghip_create(d % available);
        if (!ctx)
            throw std::runtime_error(std::string("path_tracer_hip: tghip_create failed: ") + tghip_last_error(nullptr));
        _ctxs.push_back(ctx);
        check(tghip_upload_scene(ctx, &scene.desc()), ctx, "tghip_upload_scene");
    }
    _sum.assign(size_t(_w)*_h*3, 0.0f);
    _count.assign(size_t(_w)*_h, 0);
    _imageDirty = true;

    // PathTraceIntegrator.cpp:187,196-200: the integrator's own sampler, tile dicing, one SampleRecord per 4x4 pixels
    _useSobol = scene.rendererSettings().useSobol;
    _useAdaptive = scene.rendererSettings().useAdaptiveSampling;
    _useAux = !scene.rendererSettings().outputs.empty();   // PathTracer::_trackOutputValues (PathTracer.cpp:10)
    _scheduler.reset(_w, _h, seed);
    _deviceRecords.assign(_ctxs.size(), std::vector<TgHipSampleRecord>(_useAdaptive ? _scheduler.records().size() : 0));
}

void PathTraceHipIntegrator::teardownAfterRender()
{
    if (_worker.joinable())
        _worker.join();
    for (tghip_ctx *ctx : _ctxs)
        tghip_destroy(ctx);
    _ctxs.clear();
}

// Asynchronous like the reference (PathTraceIntegrator.cpp:220-239): returns immediately, the
// bookkeeping + callback run on a worker thread once every device finished its shard.
void PathTraceHipIntegrator::startRender(std::function<void()> completionCallback)
{
    if (_worker.joinable())
        _worker.join();
    if (!done() && _ctxs.empty())
        throw std::runtime_error("path_tracer_hip: startRender before prepareForRender");
    // PathTraceIntegrator::startRender (:220-227): nothing to do when done or when generateWork finds no work
    if (done() || !_scheduler.generateWork(_currentSpp, _nextSpp, _useAdaptive)) {
        _currentSpp = _nextSpp;
        advanceSpp();
        completionCallback();
        return;
    }

    _abort = false;
    _workerError = nullptr;
    uint32_t begin = _currentSpp, end = _nextSpp;
    for (size_t d = 0; d < _ctxs.size(); ++d) {
        TgHipPassDesc pass;
        pass.spp_begin = begin;
        pass.spp_end = end;
        pass.seed = _seed;
        pass.shard_index = uint32_t(d);
        pass.shard_count = uint32_t(_ctxs.size());
        pass.flags = (_useSobol ? TGHIP_PASS_SOBOL : 0u) | (_useAdaptive ? TGHIP_PASS_RECORDS : 0u) | (_useAux ? TGHIP_PASS_AUX : 0u);
        pass.tile_seeds = _useSobol ? _scheduler.tileSeeds().data() : nullptr;
        pass.record_index = pass.record_count = nullptr;
        if (_useAdaptive) {
            // renderTile (:136-156): every pixel of a record traces samples [sampleIndex, sampleIndex + nextSampleCount)
            if (d == 0)
                _scheduler.passArrays(_recordIndex, _recordCount);
            pass.record_index = _recordIndex.data();
            pass.record_count = _recordCount.data();
        }
        check(tghip_render_pass(_ctxs[d], &pass), _ctxs[d], "tghip_render_pass");
    }
    _imageDirty = true;
    _worker = std::thread([this, completionCallback]() {
        // one host thread per device drives that device's wavefront loop (tghip_wait blocks until the shard is
        // done; different handles may be driven from different threads, include/tungsten_hip.h)
        std::vector<int> rcs(_ctxs.size(), TGHIP_OK);
        std::vector<std::thread> drivers;
        for (size_t d = 1; d < _ctxs.size(); ++d)
            drivers.emplace_back([this, d, &rcs]() { rcs[d] = tghip_wait(_ctxs[d]); });
        rcs[0] = tghip_wait(_ctxs[0]);
        for (std::thread &t : drivers)
            t.join();
        try {
            for (size_t d = 0; d < _ctxs.size(); ++d) {
                if (rcs[d] == TGHIP_E_ABORTED || _abort)
                    return;   // no finisher / callback on abort (TaskGroup.hpp:33-41,77-83)
                check(rcs[d], _ctxs[d], "tghip_wait");
            }
            if (_useAdaptive) {
                std::vector<const TgHipSampleRecord *> sources;
                for (size_t d = 0; d < _ctxs.size(); ++d) {
                    check(tghip_download_records(_ctxs[d], _deviceRecords[d].data(), _deviceRecords[d].size()), _ctxs[d], "tghip_download_records");
                    sources.push_back(_deviceRecords[d].data());
                }
                _scheduler.absorb(sources.data(), sources.size());
            }
        } catch (...) {
            _workerError = std::current_exception();
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
    if (_workerError) {
        std::exception_ptr e = _workerError;
        _workerError = nullptr;
        std::rethrow_exception(e);     // TaskGroup::wait rethrows (TaskGroup.hpp:70-75)
    }
}

void PathTraceHipIntegrator::abortRender()
{
    _abort = true;
    for (tghip_ctx *ctx : _ctxs)
        tghip_abort(ctx);
    if (_worker.joinable())
        _worker.join();
    _workerError = nullptr;
}

void PathTraceHipIntegrator::fetchFramebuffer()
{
    if (!_imageDirty)
        return;
    waitForCompletion();
    size_t n = size_t(_w)*_h;
    if (_ctxs.size() > 1) {
        // the shards' framebuffers are summed on the device side (RCCL over xGMI) and cross PCIe once; when RCCL is not
        // available the shards are downloaded one by one and added here
        int rc = tghip_reduce_framebuffers(_ctxs.data(), int(_ctxs.size()), 0, _sum.data(), _count.data(), n);
        if (rc == TGHIP_OK) { _imageDirty = false; return; }
        // Anything else -- no librccl, several contexts on one device, a communicator that cannot be set up (no shared memory or
        // peer access in a container), a failed reduce -- falls back to the per-device download and host sum below, which needs
        // nothing but PCIe.  Said once.
        static std::atomic<bool> warned(false);
        if (rc != TGHIP_E_UNSUPPORTED && !warned.exchange(true))
            std::fprintf(stderr, "path_tracer_hip: tghip_reduce_framebuffers failed (%s); summing the shards on the host\n", tghip_last_error(_ctxs[0]));
    }
    std::fill(_sum.begin(), _sum.end(), 0.0f);
    std::fill(_count.begin(), _count.end(), 0u);
    std::vector<float> s(n*3);
    std::vector<uint32_t> c(n);
    for (tghip_ctx *ctx : _ctxs) {
        check(tghip_download_framebuffer(ctx, s.data(), c.data(), n), ctx, "tghip_download_framebuffer");
        // tile ownership is disjoint, so this sum is exact (x + 0) irrespective of device order
        for (size_t i = 0; i < n*3; ++i) _sum[i] += s[i];
        for (size_t i = 0; i < n; ++i) _count[i] += c[i];
    }
    _imageDirty = false;
}

void PathTraceHipIntegrator::currentFramebuffer(std::vector<float> &sum, std::vector<uint32_t> &count)
{
    fetchFramebuffer();
    sum = _sum;
    count = _count;
}

// Every pixel's record lives on the device that owns its tile; the others hold zeros there, so adding is exact.
void PathTraceHipIntegrator::currentOutputBuffers(std::vector<TgHipAuxPixel> &aux)
{
    waitForCompletion();
    const size_t n = size_t(_w)*_h;
    aux.assign(n, TgHipAuxPixel());
    std::memset(aux.data(), 0, n*sizeof(TgHipAuxPixel));
    std::vector<TgHipAuxPixel> dev(n);
    for (tghip_ctx *ctx : _ctxs) {
        check(tghip_download_aux(ctx, dev.data(), n), ctx, "tghip_download_aux");
        if (_ctxs.size() == 1) { aux.swap(dev); break; }
        for (size_t i = 0; i < n; ++i) {
            for (int k = 0; k < 11; ++k) { aux[i].a[k] += dev[i].a[k]; aux[i].b[k] += dev[i].b[k]; aux[i].variance[k] += dev[i].variance[k]; }
            for (int k = 0; k < 5; ++k) aux[i].count[k] += dev[i].count[k];
        }
    }
}

// Resume: a pixel's running means must continue on the device that renders its tile, so every device gets the whole state
// and only ever touches (and later reports) its own pixels -- the others are cleared again when the buffers are merged.
void PathTraceHipIntegrator::restoreOutputBuffers(const std::vector<TgHipAuxPixel> &aux)
{
    waitForCompletion();
    const uint32_t tilesX = (_w + 15)/16;
    for (size_t d = 0; d < _ctxs.size(); ++d) {
        std::vector<TgHipAuxPixel> mine(aux);
        if (_ctxs.size() > 1)
            for (uint32_t y = 0; y < _h; ++y)
                for (uint32_t x = 0; x < _w; ++x)
                    if (((x/16) + (y/16)*tilesX) % _ctxs.size() != d)
                        std::memset(&mine[size_t(x) + size_t(y)*_w], 0, sizeof(TgHipAuxPixel));
        check(tghip_upload_aux(_ctxs[d], mine.data(), mine.size()), _ctxs[d], "tghip_upload_aux");
    }
}

// The merged framebuffer goes to the first device, the others restart from zero: ownership of a pixel only matters for
// the samples still to come, the final image is the sum over devices either way.
void PathTraceHipIntegrator::restoreFramebuffer(const std::vector<float> &sum, const std::vector<uint32_t> &count)
{
    waitForCompletion();
    for (size_t d = 0; d < _ctxs.size(); ++d) {
        check(tghip_clear_framebuffer(_ctxs[d]), _ctxs[d], "tghip_clear_framebuffer");
        if (d == 0)
            check(tghip_upload_framebuffer(_ctxs[d], sum.data(), count.data(), count.size()), _ctxs[d], "tghip_upload_framebuffer");
    }
    _imageDirty = true;
}

// PathTraceIntegrator::saveState / loadState (PathTraceIntegrator.cpp:158-172): the SampleRecords, then the sampler state.
// (The reference stores one sequential sampler per tile; the per-path streams here are counter-based and need none, the
// integrator's own sampler -- the one distributeAdaptiveSamples draws from -- is stored instead.)
void PathTraceHipIntegrator::saveState(std::ostream &out)
{
    const std::vector<TgHostSampleRecord> &rec = _scheduler.records();
    uint64_t n = rec.size(), state = const_cast<PassScheduler &>(_scheduler).sampler().state();
    out.write(reinterpret_cast<const char *>(&n), sizeof(n));
    out.write(reinterpret_cast<const char *>(rec.data()), std::streamsize(n*sizeof(TgHostSampleRecord)));
    out.write(reinterpret_cast<const char *>(&state), sizeof(state));
}

void PathTraceHipIntegrator::loadState(std::istream &in)
{
    uint64_t n = 0, state = 0;
    in.read(reinterpret_cast<char *>(&n), sizeof(n));
    std::vector<TgHostSampleRecord> &rec = _scheduler.records();
    if (!in || n != rec.size()) { in.setstate(std::ios::failbit); return; }
    in.read(reinterpret_cast<char *>(rec.data()), std::streamsize(n*sizeof(TgHostSampleRecord)));
    in.read(reinterpret_cast<char *>(&state), sizeof(state));
    if (!in) return;
    _scheduler.sampler().setState(state);
    if (_useAdaptive) {
        // every device gets the complete Welford state; each keeps updating the records of its own tiles
        std::vector<TgHipSampleRecord> dev(rec.size());
        for (size_t i = 0; i < rec.size(); ++i) {
            dev[i].sample_count = rec[i].sample_count; dev[i].mean = rec[i].mean; dev[i].running_variance = rec[i].running_variance;
        }
        for (tghip_ctx *ctx : _ctxs)
            check(tghip_upload_records(ctx, dev.data(), dev.size()), ctx, "tghip_upload_records");
    }
}

const std::vector<float> &PathTraceHipIntegrator::linearImage()
{
    fetchFramebuffer();
    size_t n = size_t(_w)*_h;
    _linear.resize(n*3);
    for (size_t i = 0; i < n; ++i) {
        float inv = _count[i] ? 1.0f/float(_count[i]) : 0.0f;
        for (int k = 0; k < 3; ++k)
            _linear[i*3 + k] = _sum[i*3 + k]*inv;
    }
    return _linear;
}

// ------------------------------------------------------------------------------------------
std::shared_ptr<Integrator> IntegratorFactory::instantiate(const std::string &type)
{
    // "path_tracer" scenes are served by the HIP integrator as well: it is a drop-in for that
    // entry of the reference's table (IntegratorFactory.cpp:15).
    if (type == "path_tracer_hip" || type == "path_tracer")
        return std::make_shared<PathTraceHipIntegrator>();
    throw JsonLoadException("Integrator type '" + type + "' is outside the path_tracer_hip hot-path scope");
}

std::vector<std::string> IntegratorFactory::names()
{
    return {"path_tracer_hip", "path_tracer"};
}

} // namespace tungsten_amd
