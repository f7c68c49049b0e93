// oracle/ref_binding/HipSceneFlattener.hpp -- TEST INFRASTRUCTURE (compiled against /root/reference, never by the product).
//
// The reference-side half of the drop-in described in INTEGRATION.md section 3: walks what Tungsten's own TraceableScene
// prepared (renderer/TraceableScene.hpp:57-110: every Primitive / Bsdf / Texture after prepareForRender, the light lists,
// the camera) and fills the TgHipSceneDesc that tghip_upload_scene takes.  The acceleration structures come from
// libtungsten_hip.so (tgh_accel_build, include/tungsten_host.h) -- they take the place of the rtcCommit calls.
//
// Scope: what the BASELINE scenes and the shipped example scenes of this repository's tests use -- quad, cube, sphere, triangle
// mesh, infinite sphere (constant or bitmap emission, sampled or not) primitives; lambert, null, mirror, conductor, rough
// conductor, dielectric, rough dielectric, plastic, rough plastic, smooth coat, mixed, transparency, forward BSDFs, bump maps; constant,
// checker and bitmap textures; skydomes; the pinhole camera.  Anything else is refused with a message naming the class (the stand-alone
// host of this repository, tungsten_amd/csrc/host/TraceableScene.cpp, is the complete flattener).
#ifndef HIPSCENEFLATTENER_HPP_
#define HIPSCENEFLATTENER_HPP_

#include "tungsten_hip.h"
#include "tungsten_host.h"

#include <map>
#include <memory>
#include <string>
#include <vector>

namespace Tungsten {

class TraceableScene;
class Primitive;
class Texture;
class Bsdf;
struct TraceSettings;

class HipSceneFlattener
{
    std::vector<TgHipPrimRec> _recs;
    std::vector<TgHipTriAttr> _triAttrs;
    std::vector<TgHipObject> _objects;
    std::vector<int32_t> _lights, _infiniteLights;
    std::vector<TgHipBsdf> _bsdfs;
    std::vector<TgHipTexture> _textures;
    std::vector<float> _texels, _dist, _lightTris;
    std::vector<float> _recBounds;
    std::map<const Texture *, int32_t> _texIndex;
    std::map<const Bsdf *, int32_t> _bsdfIndex;
    tgh_accel *_accel = nullptr;
    TgHipSceneDesc _desc;

    int32_t addTexture(const Texture *t);
    void addDistribution(const Texture *t);
    int32_t addBsdf(const Bsdf *b);
    void addPrimitive(const Primitive &p, bool defaultLight, const std::vector<const Primitive *> &sampled);

public:
    HipSceneFlattener();
    ~HipSceneFlattener();
    HipSceneFlattener(const HipSceneFlattener &) = delete;
    HipSceneFlattener &operator=(const HipSceneFlattener &) = delete;

    // throws std::runtime_error for scenes outside the scope above
    void build(TraceableScene &scene, const TraceSettings &settings, bool enableVolumeLightSampling);
    const TgHipSceneDesc &desc() const { return _desc; }
};

}

#endif
