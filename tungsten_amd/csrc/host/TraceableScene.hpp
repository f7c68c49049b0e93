// Render-time flattened scene.  Same role as the reference's TraceableScene
// (src/core/renderer/TraceableScene.hpp:25-274): prepares every object, builds the light lists
// (:86-110, incl. the default white environment when the scene has no emitter :97-102) and owns
// the acceleration structure.  The *backend* differs: instead of an Embree user-geometry scene
// over Primitives plus one Embree scene per mesh (:112-134, TriangleMesh.cpp:524-572) it builds
// one flat BVH2 over SoA primitive records and hands the integrator a TgHipSceneDesc to upload.
#ifndef TGAMD_TRACEABLESCENE_HPP_
#define TGAMD_TRACEABLESCENE_HPP_

#include "Scene.hpp"
#include "../../../include/tungsten_hip.h"

#include <vector>

namespace tungsten_amd {

class Integrator;

// The two trees over a single-level record array (TraceableScene::flatten, tgh_accel_build): binned-SAH BVH2, then its
// collapse into the 8-wide BVH (empty for flat-list scenes).  recs / attrs are permuted in place into the trees' order.
struct SceneAccel
{
    std::vector<TgHipBvhNode> nodes;
    std::vector<TgHipWideNode> wideNodes;
    std::vector<uint32_t> order;           // order[i] = the caller's record that ended up in slot i
    int bvhDepth = 0, wideDepth = 0;
    double sahCost = 0.0;
};
SceneAccel buildSceneAccel(std::vector<TgHipPrimRec> &recs, std::vector<TgHipTriAttr> &attrs, const std::vector<Box3f> &recBounds, bool haveInstances);

// Scenes with `instances` primitives: the inputs and the result of buildInstancedAccel (TraceableScene.cpp), which builds the scene's BVH2
// with the reference's own tree over the instances behind it, the wide BVH over the tight boxes, and the masters' subtrees.
struct InstanceSetInput
{
    uint32_t objMeta = 0;                  // the `instances` primitive's object index
    std::vector<TgHipPrimRec> recs;        // one TGHIP_REC_INSTANCE record per instance: a = position, p0 | b = rotation (w | x y z), c[0] = its master's index, meta
    std::vector<Box3f> refBounds;          // per instance: the box of its master box's eight rotated corners (Instance.cpp:409-421)
    std::vector<Box3f> tightBounds;        // per instance: the box of its geometry (Primitive::tightenInstanceBounds)
};
struct MasterInput                          // one master mesh: its triangle records (meta = the master's object index) in master space
{
    std::vector<TgHipPrimRec> recs;
    std::vector<TgHipTriAttr> attrs;
    std::vector<Box3f> bounds;
};
struct InstancedAccel
{
    std::vector<TgHipBvhNode> nodes;
    std::vector<TgHipWideNode> wideNodes;
    std::vector<uint32_t> instPrims;       // TgHipSceneDesc::inst_prims
    std::vector<float> instLeafBoxes;      // TgHipSceneDesc::inst_leaf_boxes
    std::vector<float> instTightBoxes;     // TgHipSceneDesc::inst_tight_boxes
    uint32_t numTopRecs = 0, numInstances = 0;
    int bvhDepth = 0, wideDepth = 0;
    double sahCost = 0.0;
};
// recs / attrs: in, the non-instance records (recBounds: their boxes); out, the scene's whole record array
InstancedAccel buildInstancedAccel(std::vector<TgHipPrimRec> &recs, std::vector<TgHipTriAttr> &attrs, const std::vector<Box3f> &recBounds,
                                   const std::vector<InstanceSetInput> &sets, const std::vector<MasterInput> &masters);

class TraceableScene
{
    Scene &_scene;
    Integrator *_integrator;
    uint32_t _seed;

    std::vector<TgHipBvhNode> _nodes;
    std::vector<TgHipWideNode> _wideNodes;
    std::vector<TgHipTopNode> _topNodes;
    std::vector<float> _itemBoxes;
    std::vector<int32_t> _itemObjects;
    std::vector<TgHipPrimRec> _recs;
    std::vector<TgHipTriAttr> _triAttrs;
    std::vector<TgHipObject> _objects;
    std::vector<int32_t> _lights, _infiniteLights;
    std::vector<TgHipBsdf> _bsdfs;
    std::vector<TgHipMedium> _media;
    std::vector<TgHipTexture> _textures;
    std::vector<float> _texels, _dist, _lightTris;
    std::vector<uint32_t> _instPrims;      // TgHipSceneDesc::inst_prims
    std::vector<float> _instLeafBoxes;     // TgHipSceneDesc::inst_leaf_boxes
    std::vector<float> _instTightBoxes;    // TgHipSceneDesc::inst_tight_boxes
    std::vector<std::shared_ptr<Primitive>> _allPrims;   // scene primitives (+ default light)
    TgHipSceneDesc _desc;
    Box3f _sceneBounds;
    int _bvhDepth = 0;
    int _wideDepth = 0;
    double _bvhSah = 0.0, _buildSeconds = 0.0;

    void flatten();

public:
    // integrator may be null (description-only use, e.g. by the oracle tests)
    TraceableScene(Scene &scene, Integrator *integrator, uint32_t seed);
    ~TraceableScene();

    const TgHipSceneDesc &desc() const { return _desc; }
    Scene &scene() { return _scene; }
    const Camera &cam() const { return _scene.camera; }
    const RendererSettings &rendererSettings() const { return _scene.renderer; }
    const Box3f &bounds() const { return _sceneBounds; }
    // the items of the reference's top-level Embree geometry: its _finites in scene order (TraceableScene.hpp:101-107), each with its bounds()
    // as Scene.cpp restates them (6 floats: lower, upper) and the index of its object
    const std::vector<float> &itemBoxes() const { return _itemBoxes; }
    const std::vector<int32_t> &itemObjects() const { return _itemObjects; }
    int bvhDepth() const { return _bvhDepth; }
    int wideDepth() const { return _wideDepth; }
    double bvhSahCost() const { return _bvhSah; }
    double buildSeconds() const { return _buildSeconds; }
    size_t numLights() const { return _lights.size(); }
    uint32_t seed() const { return _seed; }
};

} // namespace tungsten_amd

#endif
