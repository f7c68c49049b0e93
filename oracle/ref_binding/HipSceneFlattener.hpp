// oracle/ref_binding/HipSceneFlattener.hpp -- TEST INFRASTRUCTURE (compiled against /root/reference, never by the product).
//
// The reference-side half of the drop-in described in INTEGRATION.md section 3: walks what Tungsten's own TraceableScene
// prepared (renderer/TraceableScene.hpp:57-110: every Primitive / Bsdf / Texture after prepareForRender, the light lists,
// the camera) and fills the TgHipSceneDesc that tghip_upload_scene takes.  The acceleration structures come from
// libtungsten_hip.so (tgh_accel_build, include/tungsten_host.h) -- they take the place of the rtcCommit calls.
//
// Scope (round 4: everything the device renders but `instances`): quad, cube, sphere, disk, cylinder, triangle-mesh primitives and emitters
// (mesh emitters with their area distribution), point lights, infinite spheres (constant or bitmap emission, sampled or not), infinite
// sphere caps, skydomes; lambert, null, mirror, conductor, rough conductor, dielectric, rough dielectric, plastic, rough plastic, smooth
// coat, mixed, transparency, forward BSDFs, bump maps; constant, checker and bitmap textures; homogeneous media with every transmittance
// and phase function, on primitives and on the camera; the pinhole and the thin-lens camera (disk, blade and bitmap apertures).  Anything
// else -- the `instances` primitive: its trees come from the stand-alone host's own flattening, tungsten_amd/csrc/host/TraceableScene.cpp
// -- is refused with a message naming the class.
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
class Medium;
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
    std::vector<TgHipMedium> _media;
    std::vector<TgHipTopNode> _topNodes;
    std::vector<const Medium *> _mediumKeys;
    std::map<const Texture *, int32_t> _texIndex;
    std::map<const Bsdf *, int32_t> _bsdfIndex;
    tgh_accel *_accel = nullptr;
    TgHipSceneDesc _desc;
    // `instances` primitives (primitives/Instance.cpp): per primitive its instance records with the reference's and the tight box of every
    // instance, the distinct master meshes in the order the instances meet them, and the box the primitive has once its masters are loaded
    struct InstanceSet { uint32_t object; std::vector<TgHipPrimRec> recs; std::vector<float> refBounds, tightBounds; };
    std::vector<InstanceSet> _instanceSets;
    std::vector<Primitive *> _masters;
    std::map<const Primitive *, std::vector<float>> _instanceBox;     // lo.xyz, hi.xyz
    std::map<const Primitive *, size_t> _objectIndex;
    void addInstances(const Primitive &p, size_t objectIndex);

    int32_t addTexture(const Texture *t);
    void addDistribution(const Texture *t);
    int32_t addBsdf(const Bsdf *b);
    int32_t addMedium(const Medium *m);
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
