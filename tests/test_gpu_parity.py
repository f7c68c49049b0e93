"""Parity of the HIP path (through the C-ABI) against the CPU oracle on the same inputs, against the
committed reference goldens, and -- at BASELINE.json's full sizes -- through size-independent properties.

Tolerances (fp32 path, DESIGN.md "Numerics"): the kernels are compiled with -ffp-contract=off, follow the oracle's operation
order and evaluate the host libm's functions and Embree's triangle arithmetic restated bit for bit; so: closest hits with identical
record ids and t/u/v; per-pixel means within 1e-4 on EVERY pixel against the oracle (against the reference's goldens: except for
the measured handful of divergent samples of ten cases), image mean rel 1e-5; integer outputs (sample counts, ray / visit counters) exact."""
import os

import numpy as np
import pytest

import oracle_lib
import scenes
import tungsten_amd as tg

pytestmark = pytest.mark.gpu

SEED = tg.DEFAULT_SEED


def _skip_mt(name):
    if name == "water_caustic" and not scenes.have_water_caustic():
        pytest.skip("water-caustic assets (assets/) not present")
    if ("materialtest" in name or name == "mesh1m") and not scenes.have_materialtest():   # mesh1m is lit by materialtest's HDRI
        pytest.skip("materialtest assets (assets/) not present")


def gpu_render(path, seed=SEED, **opts):
    r = tg.Renderer(path, seed=seed)
    for k, v in opts.items():
        r.set_option(k, v)
    r.render()
    mean, ssum, count = r.image()
    c = r.counters()
    r.close()
    return mean, ssum, count, c


def compare(mean, omean, pix_rel=2e-2, max_bad=0.01, mean_rel=5e-3):
    err = np.abs(mean - omean).max(axis=-1)/(np.abs(omean).max(axis=-1) + 1e-2)
    frac_bad = float((err > pix_rel).mean())
    assert frac_bad <= max_bad, "%.3f%% of the pixels deviate by more than %g" % (100*frac_bad, pix_rel)
    gm, om = mean.mean(axis=(0, 1)), omean.mean(axis=(0, 1))
    assert (np.abs(gm - om) <= mean_rel*om + 1e-6).all(), (gm, om)


@pytest.mark.parametrize("name", sorted(scenes.GOLDEN_CASES))
def test_gpu_matches_oracle_and_reference_golden(name, tmp_path):
    """Same scene, seed and random stream on the three implementations: GPU vs oracle per pixel, and GPU vs the
    reference's own per-sample output (tests/golden) per pixel."""
    _skip_mt(name)
    mk, kw = scenes.GOLDEN_CASES[name]
    path = mk(tmp_path, name=name + ".json", **kw)
    mean, ssum, count, c = gpu_render(path)
    flat = tg.FlattenedScene(path)
    oc = oracle_lib.OracleCounters()
    spp = kw["spp"]
    if flat.info.stratified_sampler:     # *_sobol cases: SobolPathSampler dimensions with the tiles' seeds on both sides
        osum, ocount = oracle_lib.render_pass(flat.desc, flat.width, flat.height, SEED, 0, spp, flags=tg.capi.TGHIP_PASS_SOBOL,
                                              tile_seeds=oracle_lib.dice_tiles(flat.width, flat.height, SEED)[0], counters=oc)
    else:
        osum, ocount = oracle_lib.render(flat.desc, flat.width, flat.height, 0, spp, SEED, counters=oc)
    flat.close()
    assert (count == ocount).all() and (count == spp).all()
    assert c.samples == flat.width*flat.height*spp == oc.samples
    omean = osum/np.maximum(ocount, 1)[..., None]
    # Round 4: the device evaluates the host libm's own functions and Embree's own triangle arithmetic (pt_libm.h, pt_scene.h: embreeRcp), and in
    # none of the 508 032 golden samples does it leave the oracle's path (tests/test_gpu_samples.py, profiles/r4_device_diverge.jsonl).  So the
    # images are compared at rounding level -- the device sums a pixel's samples in chunks, the oracle one by one --: EVERY pixel within 1e-4.
    compare(mean, omean, pix_rel=1e-4, max_bad=0.0, mean_rel=1e-5)
    # ray counts: the same paths, so the same rays
    if "cateye" in name or "cubemap" in name:
        # a vignetted camera sample (ThinlensCamera.cpp:119-124) -- or one of a cubemap camera's pixels outside its six faces (CubemapCamera.cpp:157-159) --
        # is a black sample without a ray in the reference; the device gives it zero throughput and lets its primary ray find that out, i.e. traces
        # one ray more per such sample
        assert 0 <= int(c.closest_rays) - int(oc.closest_rays) <= oc.samples
    else:
        assert abs(int(c.closest_rays) - int(oc.closest_rays)) <= 8
    if name == "cornell_skydome":
        # the sky image is black below the horizon; the reference tests visibility before it looks the emission up (TraceBase.cpp:163-173),
        # the device looks the emission up first and queues no shadow ray for a black one: fewer rays, the same radiance
        assert 0 <= int(oc.shadow_rays) - int(c.shadow_rays) <= 0.4*oc.shadow_rays
    elif "mesh" not in name or name == "mesh1m":
        # (a sampled mesh emitter's visibility query doubles as its light.intersect, so the device traces every such ray,
        # while the oracle only counts the ones whose light.intersect succeeded)
        # The reference traces a light sample's shadow ray BEFORE it looks at the emission it would carry (attenuatedEmission, TraceBase.cpp:
        # 144-174), the device queues none for a sample that reaches a black (back) side of its light: a few rays fewer (0.1-0.3 % in the Cornell
        # box, whose light faces down), the same radiance -- never more.
        assert 0 <= int(oc.shadow_rays) - int(c.shadow_rays) <= 0.01*oc.shadow_rays + 2
    # ... and against the reference's own per-sample output: the same (tests/test_oracle_golden.py: DIVERGING, the cases in which the oracle
    # leaves the reference's path, has been empty since the reference's top-level tree was restated)
    from test_oracle_golden import DIVERGING, PINNED, diverge_bound
    if name in PINNED:       # (a residual pinned sample by sample against the reference: tests/test_gpu_samples.py holds the device to it)
        return
    ref = np.load(os.path.join(scenes.GOLDEN, name + "_samples.npz"))["samples"].mean(axis=2)
    allowed = diverge_bound(name, 0)
    compare(mean, ref, pix_rel=1e-4, max_bad=allowed/float(ref.shape[0]*ref.shape[1]), mean_rel=3e-2 if name in DIVERGING else 1e-5)


@pytest.mark.parametrize("scene", ["cornell", "materialtest", "mesh1m", "instances"])
def test_trace_rays_matches_oracle_exactly(scene, tmp_path):
    """TraceableScene::intersect batched: identical record, identical t/u/v, and IDENTICAL node/primitive visit
    counts (the counters that feed the roofline's algorithmic bytes)."""
    _skip_mt(scene)
    mk = {"cornell": scenes.cornell, "materialtest": scenes.materialtest, "mesh1m": scenes.mesh1m, "instances": scenes.cornell_instances}[scene]
    path = mk(tmp_path, resolution=(64, 36), spp=1)
    flat = tg.FlattenedScene(path)
    d = flat.desc.contents
    rs = np.random.RandomState(5)
    n = 20000
    lo, hi = np.array(list(d.bounds_lo)), np.array(list(d.bounds_hi))
    o = lo + (hi - lo)*rs.rand(n, 3)*1.2 - 0.1*(hi - lo)
    dirs = rs.randn(n, 3)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    rays = np.concatenate([o, np.full((n, 1), 1e-4), dirs, np.full((n, 1), np.inf)], axis=1).astype(np.float32)
    rays[::7, 7] = 0.5*np.linalg.norm(hi - lo)      # finite tmax
    rays[0:3, 4:7] = [[1, 0, 0], [0, -1, 0], [0, 0, 1]]   # axis-parallel directions (inf in 1/d)
    # the device walks the scene's 8-wide BVH when it has one (single-level BVH scenes), and so does the oracle here; with instances both walk
    # the BVH2 and, behind the instance-set record, the reference's own tree over the instances in the reference's order
    wide = d.num_wide_nodes > 0 and d.num_instances == 0
    ohits, onodes, oprims = oracle_lib.trace_rays(flat.desc, rays, wide=wide)
    bhits = oracle_lib.trace_rays(flat.desc, rays)[0] if wide else ohits       # ... and the BVH2 walk finds the same hits
    r = tg.Renderer(path)
    r.set_option("count_traversal", 1)
    r.reset_counters()
    ghits, ms = r.trace_rays(rays)
    c = r.counters()
    r.close()
    flat.close()
    same = ghits["rec"] == ohits["rec"]
    assert same.all(), "record ids differ on %d rays" % (~same).sum()
    hit = same & (ohits["rec"] >= 0)
    assert hit.sum() > n//10
    for k in ("t", "u", "v"):                   # the same operations in the same order on both sides: the same floats
        assert (ghits[k][hit] == ohits[k][hit]).all(), (k, int((ghits[k][hit] != ohits[k][hit]).sum()))
    if scene == "instances":
        # same hits, fewer visits: the device does not walk the master of an instance whose tight box the ray misses (the reference's tree lets
        # a ray into an instance by the box of the instance's LEAF -- the rotated corners of up to two master boxes --, walks the master and
        # finds nothing there: pt_kernels.h: instanceReachable); the oracle walks what the reference walks
        assert c.prims_tested <= oprims and c.nodes_visited <= 1.02*onodes
    else:
        assert c.nodes_visited == onodes and c.prims_tested == oprims
    assert (ghits["rec"] == bhits["rec"]).mean() >= 0.999


@pytest.mark.gpu
def test_grid_aligned_geometry_walks_like_the_oracle(tmp_path):
    """Corner, edge and grazing rays on axis-aligned tiles at exact grid coordinates far from the origin (tests/scenes.py: tile_terraces,
    terrace_rays; tests/test_host.py holds the wide walk to the BVH2 walk's hits there): the device's wide walk visits exactly what the
    oracle's visits and returns its hits."""
    path = scenes.tile_terraces(tmp_path)
    flat = tg.FlattenedScene(path)
    rays = scenes.terrace_rays()
    ohits, onodes, oprims = oracle_lib.trace_rays(flat.desc, rays, wide=True)
    r = tg.Renderer(path)
    r.set_option("count_traversal", 1)
    r.reset_counters()
    ghits, _ = r.trace_rays(rays)
    c = r.counters()
    r.close()
    flat.close()
    assert (ohits["rec"] >= 0).mean() > 0.8
    assert (ghits["rec"] == ohits["rec"]).all()
    assert (ghits["t"] == ohits["t"]).all()
    assert c.nodes_visited == onodes and c.prims_tested == oprims


def test_empty_and_degenerate_ray_batches(tmp_path):
    path = scenes.cornell(tmp_path, resolution=(16, 9), spp=1)
    r = tg.Renderer(path)
    hits, _ = r.trace_rays(np.zeros((0, 8), np.float32))
    assert len(hits) == 0
    rays = np.array([[0, 1, 6.8, 1e-4, 0, 0, -1, 1e-3],      # tmax before anything
                     [0, 1, 6.8, 1e-4, 0, 0, 1, np.inf],     # pointing away
                     [0, 1, 6.8, 1e-4, 0, 0, -1, np.inf]], np.float32)
    hits, _ = r.trace_rays(rays)
    assert hits["rec"][0] == -1 and hits["rec"][1] == -1 and hits["rec"][2] >= 0
    r.close()


def test_passes_and_pool_geometry_do_not_change_the_image(tmp_path):
    """Splitting the samples into passes (spp_step) or shrinking the path pool changes scheduling only: every
    (pixel, sample) draws from its own stream, so the image is the same up to float summation order."""
    base, _, cnt, _ = gpu_render(scenes.cornell(tmp_path, resolution=(160, 90), spp=16))
    stepped, _, cnt2, _ = gpu_render(scenes.cornell(tmp_path, resolution=(160, 90), spp=16, spp_step=4, name="s.json"))
    small, _, cnt3, _ = gpu_render(scenes.cornell(tmp_path, resolution=(160, 90), spp=16, name="p.json"), max_slots=4096)
    assert (cnt == 16).all() and (cnt2 == 16).all() and (cnt3 == 16).all()
    assert np.allclose(base, stepped, rtol=1e-5, atol=1e-6)
    assert np.allclose(base, small, rtol=1e-5, atol=1e-6)
    again, _, _, _ = gpu_render(scenes.cornell(tmp_path, resolution=(160, 90), spp=16))
    assert (again == base).all(), "same configuration must be bit-reproducible"


@pytest.mark.parametrize("adaptive", [False, True])
def test_shards_and_batches_round_like_the_whole_pass(adaptive, tmp_path):
    """A pass picks its work-item size by its own sample count -- one sample per item when short, four when long -- and a tile shard of a pass is
    shorter than the pass, a pass cut into batches is several shorter ones.  k_resolve adds a pixel's samples up in ONE order whatever the
    items (groups of four consecutive samples first, group after group into the pixel; batches hold whole groups), so the images agree BIT
    FOR BIT: four-sample items against one-sample items, one batch against many, the sum of three shards against the whole -- at any size
    (a reviewer's finding of round 3: at production sizes the shards of a pass used to round differently from the pass)."""
    import ctypes as C
    import json
    if not scenes.have_materialtest():
        pytest.skip("materialtest assets (assets/) not present")
    kw = dict(resolution=(192, 108), spp=22)
    if adaptive:
        kw = dict(resolution=(96, 54), spp=44, spp_step=22, renderer={"adaptive_sampling": True, "stratified_sampler": True})
    path = scenes.materialtest(tmp_path, **kw)
    images = {}
    for name, opts in (("four", dict(chunk_samples=4)), ("one", dict(chunk_samples=1)), ("auto", dict()),
                       ("one, batches", dict(chunk_samples=1, max_items=40000)), ("four, batches", dict(chunk_samples=4, max_items=30000)),
                       ("one, small pool", dict(chunk_samples=1, max_slots=8192, max_items=70000))):
        img, ssum, cnt, _ = gpu_render(path, **opts)
        images[name] = ssum
        assert (cnt > 0).all()
    for name in images:
        assert images[name].tobytes() == images["four"].tobytes(), name
    # three shards on three contexts of the one device (each picks its own item size), summed on the host
    d = json.load(open(path))
    d["integrator"].update(devices=3, share_devices=True)
    shared = os.path.join(str(tmp_path), "shared.json")
    json.dump(d, open(shared, "w"))
    r = tg.Renderer(shared, seed=SEED)
    r.set_option("chunk_samples", 1, device=1)            # (one of the shards with the other item size)
    r.set_option("chunk_samples", 4, device=2)
    r.render()
    _, ssum, _ = r.image()
    r.close()
    assert ssum.tobytes() == images["four"].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["materialtest", "cornell_instances"])
def test_loop_scheduling_does_not_change_the_image(scene, tmp_path):
    """How the wavefront loop is scheduled -- parts of the pool on several streams, shading classes on streams of their own, slots per
    workgroup, pool size, host check interval, the instanced scenes' closest-hit / shading variants -- moves work between launches,
    never between (pixel, sample) streams: per-pixel sums are accumulated per work item and resolved in fixed chunk order, so every
    configuration gives the bit-identical image."""
    _skip_mt(scene)
    if scene == "materialtest":
        path = scenes.materialtest(tmp_path, resolution=(192, 108), spp=8)
    else:
        mk, kw = scenes.GOLDEN_CASES[scene]
        path = mk(tmp_path, **dict(kw, resolution=(160, 90), spp=8))
    # (renders this small would go to the tail kernel at the first host check: the loop's variants are compared with the tail kernel off,
    # the tail kernel's -- from the first check, from the middle of the render, on one stream and on eight -- against the same image)
    base, _, cnt, _ = gpu_render(path, tail_kernel=0)
    assert (cnt == 8).all() and np.isfinite(base).all()
    variants = [dict(streams=1), dict(streams=2), dict(streams=8), dict(class_streams=1), dict(streams=1, class_streams=1),
                dict(slots_per_block=512), dict(max_slots=8192), dict(check_interval=1), dict(check_interval=16), dict(blocks_per_cu=4),
                dict(grid_rounds=2), dict(leaf_batch=9), dict(streams=4, blocks_per_cu=4, check_interval=4),
                # the escaped paths shaded by a launch of their own instead of behind class 0 in one launch
                dict(merge_miss=0), dict(merge_miss=0, streams=1),
                # k_finish as a launch of its own in every iteration instead of in front of the next closest-hit launch
                dict(fold_finish=0), dict(fold_finish=0, check_interval=1), dict(fold_finish=1, check_interval=1), dict(fold_finish=1, streams=1, check_interval=3)]
    if scene == "materialtest":
        # walk time-slicing of the wide traversal kernels (PathState::suspend_*): off, the default, and settings that suspend every walk
        # after one / three / two turns of every launch -- hundreds of save / resume round trips per ray, shadow slots held and released
        variants += [dict(suspend_lanes=0), dict(suspend_lanes=64, suspend_turns=1, suspend_min_queue=0),
                     dict(suspend_lanes=64, suspend_turns=3, suspend_min_queue=0, streams=1, check_interval=1),
                     dict(suspend_lanes=8, suspend_turns=2, suspend_min_queue=0), dict(suspend_lanes=16, suspend_turns=32, suspend_min_queue=64),
                     # the sequential walk (one record OR node per turn, k_trace_*_wide<.., DECOUPLED = false>) against the default decoupled one
                     dict(decouple=0), dict(decouple=0, suspend_lanes=64, suspend_turns=1, suspend_min_queue=0),
                     # how much of the top of the wide tree the traversal kernels keep in LDS
                     dict(lds_nodes=0), dict(lds_nodes=1), dict(lds_nodes=9), dict(lds_nodes=585)]
    if scene == "cornell_instances":
        # (wide_closest = 1 is not among them: the wide walk returns the NEAREST instance hit, the default walks the reference's own tree over
        # the instances in the reference's order and returns the LAST one, as Instance::intersect does)
        variants += [dict(inst_dyn=0), dict(inst_simple=0), dict(wide_shadow=0), dict(leaf_batch_bvh2=1), dict(inst_dyn=0, wide_shadow=0)]
    variants = [dict(v, tail_kernel=0) for v in variants]
    variants += [dict(tail_kernel=1), dict(tail_kernel=1, streams=1), dict(tail_kernel=1, streams=8), dict(tail_kernel=1, tail_threshold=3000, check_interval=2),
                 dict(tail_kernel=1, tail_threshold=30000, check_interval=1, streams=2), dict(tail_kernel=1, slots_per_block=512),
                 dict(tail_kernel=1, suspend_lanes=64, suspend_turns=2, suspend_min_queue=0), dict(tail_kernel=1, max_slots=8192, tail_threshold=1 << 30)]
    for opts in variants:
        img, _, c, _ = gpu_render(path, **opts)
        assert (c == 8).all(), opts
        assert (img == base).all(), "image changed with %r" % (opts,)
    # samples per work item: the same samples in other partial sums (the order of the float additions changes, nothing else); one
    # sample per item is also the configuration in which the parts of the pool drift furthest apart (the liveness word takes the
    # MAXIMUM of the iteration tags for that reason)
    for chunk in (1, 2, 3):
        for opts in (dict(chunk_samples=chunk), dict(chunk_samples=chunk, check_interval=4), dict(chunk_samples=chunk, streams=4, blocks_per_cu=4)):
            img, _, c, _ = gpu_render(path, **opts)
            assert (c == 8).all(), opts
            assert np.allclose(img, base, rtol=1e-5, atol=1e-6), opts


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell_fog", "cornell_smoke", "cornell_fog_smoke_sobol", "cornell_fog_rayleigh", "cornell_fog_davis",
                                  "volumetric_caustic", "non_exponential_linear", "non_exponential_area_lights"])
def test_lean_media_shading_variant_is_the_full_one(name, tmp_path):
    """Media scenes whose surfaces are Lambert / null / forward / dielectric / mirror are shaded by k_shade<MASK_MEDIA> (no scratch) instead
    of k_shade<BSDF_MASK_ALL>: a narrower instantiation of the same shadeBody -- the same image bit for bit, the same rays."""
    mk, kw = scenes.GOLDEN_CASES[name]
    path = mk(tmp_path, name=name + ".json", **dict(kw, resolution=(160, 90)))
    full, _, cf, kf = gpu_render(path, media_lean=0)
    lean, _, cl, kl = gpu_render(path, media_lean=1)
    assert (cf == cl).all() and np.isfinite(full).all()
    assert (full == lean).all()
    assert (kf.closest_rays, kf.shadow_rays, kf.samples) == (kl.closest_rays, kl.shadow_rays, kl.samples)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "cornell_sun_sky", "cornell_skydome", "cornell_png_textures", "zoo_a", "cornell_fog", "cornell_instances", "materialtest", "cornell_mesh_light", "cornell_bump"])
def test_table_placement_does_not_change_the_image(name, tmp_path):
    """The shading kernels read objects / bsdfs / textures / light lists from an LDS copy (pt_kernels.h: stageSceneTables), the sampled environment
    map's marginal tables included; a scene whose tables do not fit shades with the one GLOBAL_TABLES variant (every class, no fused and no tail
    launches), and an environment map whose marginal tables do not fit is sampled through its global tables.  Options `lds_tables` = 0 and
    `env_lds` = 0 force those two paths on scenes that do fit: the same image bit for bit, the same rays."""
    _skip_mt(name)
    if name in scenes.GOLDEN_CASES:
        mk, kw = scenes.GOLDEN_CASES[name]
        path = mk(tmp_path, name=name + ".json", **dict(kw, resolution=(160, 90)))
    else:
        path = getattr(scenes, name)(tmp_path, resolution=(160, 90), spp=8)
    base, _, cb, kb = gpu_render(path)
    assert np.isfinite(base).all()
    for opts in (dict(lds_tables=0), dict(env_lds=0), dict(lds_tables=0, env_lds=0)):
        img, _, c, k = gpu_render(path, **opts)
        assert (c == cb).all(), opts
        assert (img == base).all(), "image changed with %r" % (opts,)
        assert (k.closest_rays, k.shadow_rays, k.samples) == (kb.closest_rays, kb.shadow_rays, kb.samples), opts


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["materialtest", "mesh1m"])
def test_hoisted_quad_does_not_change_the_image(name, tmp_path):
    """A scene of triangles with exactly one quad (the ground plane of materialtest and mesh1m): the decoupled walks test the quad once per ray before
    the walk and skip it inside (DeviceScene::hoisted_rec); option `hoist_quad` = 0 leaves it in the walk.  A quad accepts t <= tmax and a triangle
    t < tmax, so the quad wins a tie whichever is tested first: the same image bit for bit, the same rays -- on the loop and on the tail kernel."""
    _skip_mt(name)
    path = getattr(scenes, name)(tmp_path, resolution=(320, 180), spp=8)
    base, _, cb, kb = gpu_render(path, hoist_quad=0)
    assert np.isfinite(base).all()
    for opts in (dict(hoist_quad=1), dict(hoist_quad=1, tail_kernel=1, tail_threshold=1 << 30), dict(hoist_quad=1, suspend_lanes=64, suspend_turns=2, suspend_min_queue=0)):
        img, _, c, k = gpu_render(path, **opts)
        assert (c == cb).all(), opts
        assert (img == base).all(), "image changed with %r" % (opts,)
        assert (k.closest_rays, k.shadow_rays, k.samples) == (kb.closest_rays, kb.shadow_rays, kb.samples), opts


@pytest.mark.gpu
def test_tail_kernel_traces_the_rays_the_loop_traces(tmp_path):
    """k_tail (one launch per part in which every workgroup iterates over its own slots) against the launch-per-step loop: the same image,
    the same samples, the same closest-hit and shadow rays -- entered at the first host check, and half way through the render."""
    _skip_mt("materialtest")
    path = scenes.materialtest(tmp_path, resolution=(320, 180), spp=16)
    res = []
    for opts in (dict(tail_kernel=0), dict(tail_kernel=1, tail_threshold=1 << 30), dict(tail_kernel=1, tail_threshold=100000, check_interval=2)):
        r = tg.Renderer(path, seed=SEED)
        for k, v in opts.items():
            r.set_option(k, v)
        r.render()
        mean, _, cnt = r.image()
        c = r.counters()
        res.append((mean.copy(), cnt.copy(), (c.samples, c.closest_rays, c.shadow_rays, c.shadow_slots), c.tail_launches, c.iterations))
        r.close()
    assert res[0][3] == 0 and res[1][3] >= 1 and res[2][3] >= 1
    assert res[1][4] == 0 and 0 < res[2][4] < res[0][4]          # (iterations count the loop's launches only)
    for other in res[1:]:
        assert other[2] == res[0][2]
        assert (other[0] == res[0][0]).all() and (other[1] == res[0][1]).all()


@pytest.mark.gpu
def test_suspended_walks_visit_what_uninterrupted_walks_visit(tmp_path):
    """A suspended walk continues exactly where it stopped: ray, node-visit and record-test counts of a render that suspends every walk
    after two turns per launch equal those of the render that never suspends (and the image is the same, bit for bit)."""
    _skip_mt("materialtest")
    path = scenes.materialtest(tmp_path, resolution=(160, 90), spp=4)
    res = []
    for opts in (dict(suspend_lanes=0), dict(suspend_lanes=64, suspend_turns=2, suspend_min_queue=0)):
        r = tg.Renderer(path, seed=SEED)
        r.set_option("count_traversal", 1)
        for k, v in opts.items():
            r.set_option(k, v)
        r.render()
        mean, _, cnt = r.image()
        c = r.counters()
        res.append((mean.copy(), cnt.copy(), (c.samples, c.closest_rays, c.shadow_rays, c.shadow_slots, c.nodes_visited, c.prims_tested,
                                              c.nodes_visited_shadow, c.prims_tested_shadow), c.iterations))
        r.close()
    assert res[0][2] == res[1][2]
    assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all()
    assert res[1][3] > res[0][3]                    # the time-sliced render really took more, shorter launches


@pytest.mark.gpu
def test_instanced_shadow_walk_agrees_with_the_bvh2_walk(tmp_path):
    """The wide shadow walk of instanced scenes (k_trace_shadow_wide<., ., INST>) gives the image of the two-level BVH2 shadow walk bit
    for bit, with and without counting its visits -- on the crowded instance scene of the goldens and on instances10k (BASELINE
    configs[4]'s scene at 2 spp), where the variant without PT_TURN_JOIN (pt_wavefront.h) loses occluders in 44 % of the pixels."""
    mk, kw = scenes.GOLDEN_CASES["cornell_instances"]
    path = mk(tmp_path, **dict(kw, resolution=(160, 90), spp=8))
    bvh2, _, cnt, _ = gpu_render(path, wide_shadow=0)
    plain, _, _, _ = gpu_render(path)
    counting, _, _, c = gpu_render(path, count_traversal=1)
    assert (cnt == 8).all() and c.nodes_visited_shadow > 0
    assert (plain == bvh2).all()
    assert (counting == bvh2).all()
    # (round 6: the default is k_trace_shadow_fast_inst -- the same walk inside the fast kernel's slot handling; inst_shadow_fast = 0: k_trace_shadow_wide<., ., INST>)
    for opts in (dict(inst_shadow_fast=0), dict(inst_shadow_fast=0, count_traversal=1)):
        img, _, _, _ = gpu_render(path, **opts)
        assert (img == bvh2).all(), opts
    if not scenes.have_materialtest():
        pytest.skip("instances10k needs the materialtest assets (assets/)")
    big = scenes.instances10k(tmp_path, resolution=(1920, 1080), spp=2)
    r = tg.Renderer(big, seed=SEED)
    try:
        images = {}
        for name, opts in (("bvh2", dict(wide_shadow=0)), ("wide", dict(wide_shadow=1)), ("wide, counting", dict(count_traversal=1)),
                           ("round 5's kernel", dict(count_traversal=0, inst_shadow_fast=0))):
            for k, v in opts.items():
                r.set_option(k, v)
            images[name] = _one_pass(r, 2)
    finally:
        r.close()
    assert images["bvh2"].mean() > 0.1
    assert (images["wide"] == images["bvh2"]).all(), float((images["wide"] != images["bvh2"]).any(axis=-1).mean())
    assert (images["wide, counting"] == images["bvh2"]).all()
    assert (images["round 5's kernel"] == images["bvh2"]).all()


@pytest.mark.parametrize("scene", ["materialtest", "mesh1m"])
def test_hinted_kernels_are_deterministic_and_agree_with_the_plain_walks(scene, tmp_path):
    """The kernels whose path-pool stores carry the non-temporal hint (PT_NT_STATE: the wavefront k_shade launches, the decoupled walks, the folded
    k_finish) at the size where the pool is megabytes and four parts share the chip: the default build renders the same image twice, bit for
    bit, and that image is the one of the kernels WITHOUT the hint -- the sequential wide walks (decouple = 0, stand-alone k_finish), bit for bit, and
    the BVH2 walks (wide_bvh = 0) but for the few pixels in which two triangles are hit at the same distance.  What the hint may and may not do was measured on the device (tools/ubench_nt_coherence.hip,
    profiles/r6_ubench_nt_coherence.txt: a plain load after a non-temporal store never reads a stale line, same lane or another wave of the
    workgroup); this test holds the product's kernels to it (round 5's k_trace_shadow_wide failure showed as run-to-run differences)."""
    _skip_mt(scene)
    path = (scenes.materialtest if scene == "materialtest" else scenes.mesh1m)(tmp_path, resolution=(960, 540), spp=4)
    r = tg.Renderer(path, seed=SEED)
    try:
        first = _one_pass(r, 4)
        again = _one_pass(r, 4)
        images = {}
        for name, opts in (("sequential wide walks", dict(decouple=0, fold_finish=0)), ("BVH2 walks", dict(decouple=1, fold_finish=1, wide_bvh=0))):
            for k, v in opts.items():
                r.set_option(k, v)
            images[name] = _one_pass(r, 4)
    finally:
        r.close()
    assert np.isfinite(first).all() and first.mean() > 0.05
    assert (first == again).all(), "two renders of the default build differ in %.4f %% of the pixels" % (100.0*float((first != again).any(axis=-1).mean()))
    for name, img in images.items():
        differ = (img != first).any(axis=-1)
        # (the BVH2 walk meets hits at EQUAL distances in another order than the wide walk: one or two pixels of 518 400 take the other triangle of an
        # edge -- measured 1 / 2 on materialtest / mesh1m, round 6; the wide walks must agree with each other in every pixel)
        allowed = 4 if name == "BVH2 walks" else 0
        assert int(differ.sum()) <= allowed, "%s: %d pixels (%.4f %%) differ" % (name, int(differ.sum()), 100.0*float(differ.mean()))


def _one_pass(r, spp, seed=SEED):
    """one plain pass [0, spp) into a cleared framebuffer on the renderer's first context; returns the radiance sums"""
    import ctypes as C
    ctx = r.context()
    n = r.width*r.height
    ssum = np.empty((r.height, r.width, 3), np.float32)
    cnt = np.empty((r.height, r.width), np.uint32)
    p = tg.TgHipPassDesc(0, spp, seed & 0xFFFFFFFF, 0, 1, 0)
    for rc in (tg.lib.tghip_clear_framebuffer(ctx), tg.lib.tghip_render_pass(ctx, C.byref(p)), tg.lib.tghip_wait(ctx),
               tg.lib.tghip_download_framebuffer(ctx, ssum.ctypes.data, cnt.ctypes.data, n)):
        assert rc == 0, tg.lib.tghip_last_error(ctx).decode()
    assert (cnt == spp).all()
    return ssum


def test_tile_shards_partition_the_image(tmp_path):
    """The multi-GPU decomposition (16x16 tiles round-robin over shards) on one device: shard
    framebuffers are disjoint and sum to the unsharded image exactly."""
    import ctypes as C
    path = scenes.cornell(tmp_path, resolution=(100, 50), spp=4)
    r = tg.Renderer(path)
    ctx = r.context()
    n = 100*50

    def run(idx, cnt):
        assert tg.lib.tghip_clear_framebuffer(ctx) == 0
        p = tg.TgHipPassDesc(0, 4, 99, idx, cnt, 0)
        assert tg.lib.tghip_render_pass(ctx, C.byref(p)) == 0
        assert tg.lib.tghip_wait(ctx) == 0
        s, c = np.empty((n, 3), np.float32), np.empty(n, np.uint32)
        assert tg.lib.tghip_download_framebuffer(ctx, s.ctypes.data, c.ctypes.data, n) == 0
        return s, c
    whole, wc = run(0, 1)
    acc, cnt = np.zeros_like(whole), np.zeros_like(wc)
    for i in range(3):
        s, c = run(i, 3)
        assert ((c > 0) & (cnt > 0)).sum() == 0
        assert (s[c == 0] == 0).all(), "a shard wrote radiance into pixels it does not own"
        acc += s
        cnt += c
    r.close()
    assert (cnt == wc).all() and (wc == 4).all()
    assert (acc == whole).all()


@pytest.mark.parametrize("scene,res,spp", [("cornell", (1280, 720), 32), ("materialtest", (1280, 720), 8)])
def test_full_size_properties(scene, res, spp, tmp_path):
    """BASELINE.json sizes (1280x720): every pixel receives exactly spp finite samples, the counters add up,
    the image mean agrees with the oracle on a bounded sub-sample of rows, and is finite everywhere."""
    _skip_mt(scene)
    mk = scenes.cornell if scene == "cornell" else scenes.materialtest
    path = mk(tmp_path, resolution=res, spp=spp)
    mean, ssum, count, c = gpu_render(path)
    w, h = res
    assert (count == spp).all()
    assert c.samples == w*h*spp
    assert np.isfinite(mean).all() and (mean >= 0).all()
    assert 3.0 <= (c.closest_rays + c.shadow_rays)/c.samples <= 6.0     # 4.2-4.4 rays per sample
    # oracle on 16 full rows (one tile row): same pixels, same streams
    flat = tg.FlattenedScene(path)
    row0 = (h//2)//16
    tiles_x = (w + 15)//16
    # shard trick: with shard_count = number of tile rows*..., pick tiles of one row via a pass over the whole image
    # restricted by the oracle's own per-sample entry point
    ys = range(row0*16, row0*16 + 16)
    xs = range(0, w, 5)
    om = np.zeros((len(ys), len(xs), 3), np.float32)
    for iy, y in enumerate(ys):
        for ix, x in enumerate(xs):
            acc = np.zeros(3, np.float64)
            for s in range(spp):
                acc += oracle_lib.trace_sample(flat.desc, SEED, x, y, s)
            om[iy, ix] = acc/spp
    flat.close()
    gm = mean[row0*16:row0*16 + 16, ::5]
    compare(gm, om, pix_rel=1e-4, max_bad=0.0, mean_rel=1e-5)


def test_errors_and_abort(tmp_path):
    import ctypes as C
    ctx = tg.lib.tghip_create(0)
    assert ctx
    p = tg.TgHipPassDesc(0, 1, 1, 0, 1, 0)
    assert tg.lib.tghip_render_pass(ctx, C.byref(p)) == -4           # TGHIP_E_NOSCENE
    assert b"before upload" in tg.lib.tghip_last_error(ctx)
    assert tg.lib.tghip_set_option(ctx, b"no_such_option", 1) == -1
    flat = tg.FlattenedScene(scenes.cornell(tmp_path, resolution=(64, 36), spp=1))
    bad = tg.TgHipSceneDesc.from_buffer_copy(flat.desc.contents)
    bad.abi_version = 77
    assert tg.lib.tghip_upload_scene(ctx, C.byref(bad)) == -1
    assert tg.lib.tghip_upload_scene(ctx, flat.desc) == 0
    p = tg.TgHipPassDesc(4, 2, 1, 0, 1, 0)                            # spp_end < spp_begin
    assert tg.lib.tghip_render_pass(ctx, C.byref(p)) == -1
    p = tg.TgHipPassDesc(0, 2, 1, 5, 2, 0)                            # shard_index >= shard_count
    assert tg.lib.tghip_render_pass(ctx, C.byref(p)) == -1
    p = tg.TgHipPassDesc(0, 0, 1, 0, 1, 0)                            # empty pass is a no-op
    assert tg.lib.tghip_render_pass(ctx, C.byref(p)) == 0 and tg.lib.tghip_wait(ctx) == 0
    tg.lib.tghip_destroy(ctx)
    flat.close()
    assert tg.lib.tghip_create(10**6) is None


def test_abort_stops_a_running_pass(tmp_path):
    """tghip_abort (PathTraceIntegrator::abortRender, PathTraceIntegrator.cpp:246-256): a pass that would run for
    seconds returns TGHIP_E_ABORTED shortly after the flag is raised from another thread; the handle stays usable."""
    import ctypes as C
    import threading
    import time
    path = scenes.cornell(tmp_path, resolution=(1280, 720), spp=8192)
    flat = tg.FlattenedScene(path)
    ctx = tg.lib.tghip_create(0)
    assert tg.lib.tghip_upload_scene(ctx, flat.desc) == 0
    p = tg.TgHipPassDesc(0, 8192, SEED, 0, 1, 0)
    assert tg.lib.tghip_render_pass(ctx, C.byref(p)) == 0
    result = {}

    def waiter():
        t0 = time.time()
        result["rc"] = tg.lib.tghip_wait(ctx)
        result["secs"] = time.time() - t0
    th = threading.Thread(target=waiter)
    th.start()
    time.sleep(0.3)
    assert th.is_alive(), "the pass finished before it could be aborted"
    assert tg.lib.tghip_abort(ctx) == 0
    th.join(timeout=30)
    assert not th.is_alive()
    assert result["rc"] == -5, result                       # TGHIP_E_ABORTED
    assert result["secs"] < 2.0, result                     # the full pass takes ~4 s
    # partial results are kept (every finished sample was flushed) and a new pass works
    n = 1280*720
    s, c = np.empty((n, 3), np.float32), np.empty(n, np.uint32)
    assert tg.lib.tghip_download_framebuffer(ctx, s.ctypes.data, c.ctypes.data, n) == 0
    assert np.isfinite(s).all() and 0 < int(c.max()) < 8192
    assert tg.lib.tghip_clear_framebuffer(ctx) == 0
    p = tg.TgHipPassDesc(0, 2, SEED, 0, 1, 0)
    assert tg.lib.tghip_render_pass(ctx, C.byref(p)) == 0 and tg.lib.tghip_wait(ctx) == 0
    assert tg.lib.tghip_download_framebuffer(ctx, s.ctypes.data, c.ctypes.data, n) == 0
    assert (c == 2).all()
    tg.lib.tghip_destroy(ctx)
    flat.close()


def test_integrator_loop_and_outputs(tmp_path):
    """The CLI path (Shared.hpp:281-317): while (!done) { startRender; waitForCompletion } over several passes, then
    saveOutputs writes the PNG (tonemapped) and the PFM (mean radiance, bottom row first)."""
    png, pfm = str(tmp_path/"out.png"), str(tmp_path/"out.pfm")
    path = scenes.cornell(tmp_path, resolution=(96, 54), spp=12, spp_step=5,
                          renderer={"output_file": png, "hdr_output_file": pfm, "overwrite_output_files": True})
    r = tg.Renderer(path)
    steps = 0
    while not r.step():
        steps += 1
    assert steps == 2                                        # passes of 5, 5, 2 spp
    mean, ssum, count = r.image()
    assert (count == 12).all()
    r.save_outputs()
    r.close()
    assert os.path.getsize(png) > 1000
    assert np.allclose(tg.load_pfm(pfm), mean, rtol=1e-6, atol=1e-7)


def test_large_passes_are_split_into_batches(tmp_path):
    """A pass whose work items exceed `max_items` is split by sample range first, then by tiles (tghip_wait): every
    pixel still receives exactly spp samples and the image equals the unsplit render up to summation order."""
    path = scenes.cornell(tmp_path, resolution=(320, 180), spp=16)
    base, _, cnt, c0 = gpu_render(path)
    assert (cnt == 16).all()
    for max_items in (1 << 16, 1 << 13):        # 230 400 items in one piece; forces sample-range and tile splits
        img, _, cnt2, c = gpu_render(path, max_items=max_items)
        assert (cnt2 == 16).all()
        assert c.samples == 320*180*16
        assert np.allclose(img, base, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_instanced_closest_hit_walk_through_wide_masters_agrees_with_the_bvh2_walk(tmp_path):
    """Round 6: k_trace_closest_instw walks the masters' subtrees through the 8-wide BVH (a nearest-hit query may be any correct one:
    primitives/Instance.cpp:296-303 fixes the order between instances only) and votes on the phase of its turn.  Same image, bit for bit, as the
    three-level BVH2 walk (option inst_wide = 0), whatever the vote's threshold and the refill level, with and without the visit counters -- on the
    crowded instance scene of the goldens, on the tie scene, and on instances10k at 2 spp."""
    for case in ("cornell_instances", "cornell_instance_ties"):
        mk, kw = scenes.GOLDEN_CASES[case]
        path = mk(tmp_path, name=case + ".json", **dict(kw, resolution=(160, 90), spp=8))
        bvh2, _, cnt, _ = gpu_render(path, inst_wide=0)
        assert (cnt == 8).all() and bvh2.mean() > 0.01
        for opts in (dict(), dict(inst_phase_min=1), dict(inst_phase_min=64, inst_refill_at=32), dict(count_traversal=1), dict(inst_dyn=0)):
            img, _, _, c = gpu_render(path, **opts)
            assert (img == bvh2).all(), (case, opts, float((img != bvh2).any(axis=-1).mean()))
    if not scenes.have_materialtest():
        pytest.skip("instances10k needs the materialtest assets (assets/)")
    big = scenes.instances10k(tmp_path, resolution=(1920, 1080), spp=2)
    r = tg.Renderer(big, seed=SEED)
    try:
        images = {}
        for name, opts in (("bvh2", dict(inst_wide=0)), ("wide", dict(inst_wide=1)), ("wide, counting", dict(count_traversal=1)),
                           ("wide, vote 1", dict(count_traversal=0, inst_phase_min=1)), ("wide, vote 40", dict(inst_phase_min=40, inst_refill_at=56))):
            for k, v in opts.items():
                r.set_option(k, v)
            images[name] = _one_pass(r, 2)
    finally:
        r.close()
    assert images["bvh2"].mean() > 0.1
    # every variant of the new kernel: one image (a lane's walk is a function of its ray)
    for name in images:
        if name != "bvh2":
            assert (images[name] == images["wide"]).all(), (name, float((images[name] != images["wide"]).any(axis=-1).mean()))
    # against the BVH2 walk of the masters: 4.1 M paths through 10 000 copies of a lat-long sphere, whose poles are fans of a hundred slivers around one
    # vertex -- where two triangles of a master answer a ray at distances one rounding apart, the first one tested keeps the hit (triTest accepts
    # T < |den| tmax and rounds t afterwards), and the two walks test in different orders.  Measured: 63 of 2 073 600 pixels.  (The reference's order
    # inside a master is Embree's, which neither walk restates: DESIGN.md "ties".)
    differ = float((images["wide"] != images["bvh2"]).any(axis=-1).mean())
    assert differ <= 1e-4, differ
    assert np.allclose(images["wide"].mean(axis=(0, 1)), images["bvh2"].mean(axis=(0, 1)), rtol=1e-5)


def test_many_instances_match_oracle(tmp_path):
    """BASELINE configs[4] in small: 2 500 instances of a 1 800-triangle mesh (four masters / materials) -- a deep three-level
    walk (the scene's tree, the reference's own tree over the instances in the reference's order, the master's subtree, on one stack)
    against the oracle's recursion."""
    path = scenes.instances10k(tmp_path, resolution=(64, 36), spp=4, count=2500, n_lat=30, n_lon=30)
    mean, ssum, count, c = gpu_render(path, count_traversal=1)
    flat = tg.FlattenedScene(path)
    assert flat.desc.contents.num_instances == 2500 and flat.desc.contents.num_top_recs == 2502      # the floor, 2 500 instance records, the set
    oc = oracle_lib.OracleCounters()
    osum, ocount = oracle_lib.render(flat.desc, flat.width, flat.height, 0, 4, SEED, counters=oc)
    flat.close()
    assert (count == ocount).all() and (count == 4).all()
    compare(mean, osum/np.maximum(ocount, 1)[..., None], pix_rel=1e-4, max_bad=0.0, mean_rel=1e-5)
    assert abs(int(c.closest_rays) - int(oc.closest_rays)) <= 8
    # (exact node / record visit counts of the two-level walk: test_trace_rays_matches_oracle_exactly[instances]; a pass's
    # totals differ because the device answers shadow rays with any-hit queries, the oracle with closest-hit walks)


def _shard_contexts(path, n, spp, seed=SEED):
    """n contexts (device i % device_count... one per device) that each render tile shard i of n; returns (flat, ctxs)."""
    import ctypes as C
    flat = tg.FlattenedScene(path)
    ctxs = []
    for i in range(n):
        ctx = tg.lib.tghip_create(i)
        assert ctx, tg.lib.tghip_last_error(None)
        assert tg.lib.tghip_upload_scene(ctx, flat.desc) == 0
        p = tg.TgHipPassDesc(0, spp, seed, i, n, 0)
        assert tg.lib.tghip_render_pass(ctx, C.byref(p)) == 0
        ctxs.append(ctx)
    return flat, ctxs


@pytest.mark.parametrize("n", [1, 2])
def test_rccl_framebuffer_reduce_behind_the_c_abi(n, tmp_path):
    """tghip_reduce_framebuffers: the tile shards of n in-process contexts, one per device, summed by RCCL into
    the root's scratch buffer -- bit-identical to the unsharded render (disjoint tiles: x + 0).  n = 1 runs the whole RCCL path
    (library load, communicator, ncclReduce) on the one device every box has; n = 2 needs two devices."""
    import ctypes as C
    if tg.device_count() < n:
        pytest.skip("needs %d HIP devices" % n)
    w, h, spp = 200, 120, 4
    path = scenes.cornell(tmp_path, resolution=(w, h), spp=spp)
    whole, wsum, wcount, _ = gpu_render(path)
    flat, ctxs = _shard_contexts(path, n, spp)
    npix = w*h
    s, c = np.full((npix, 3), -1, np.float32), np.full(npix, 7, np.uint32)
    arr = (C.c_void_p*n)(*ctxs)
    rc = tg.lib.tghip_reduce_framebuffers(arr, n, 0, s.ctypes.data, c.ctypes.data, npix)
    assert rc == 0, tg.lib.tghip_last_error(ctxs[0])
    assert (c.reshape(h, w) == wcount).all() and (wcount == spp).all()
    assert s.reshape(h, w, 3).tobytes() == wsum.tobytes()
    # the shards' own framebuffers are untouched (a second reduce gives the same image) and wrong arguments are refused
    s2, c2 = np.empty_like(s), np.empty_like(c)
    assert tg.lib.tghip_reduce_framebuffers(arr, n, 0, s2.ctypes.data, c2.ctypes.data, npix) == 0
    assert s2.tobytes() == s.tobytes() and (c2 == c).all()
    assert tg.lib.tghip_reduce_framebuffers(arr, n, n, s.ctypes.data, c.ctypes.data, npix) == -1
    assert tg.lib.tghip_reduce_framebuffers(arr, n, 0, s.ctypes.data, c.ctypes.data, npix + 1) == -1
    for ctx in ctxs:
        tg.lib.tghip_destroy(ctx)
    flat.close()


def test_rank_communicator_reduce_behind_the_c_abi(tmp_path):
    """tghip_comm_unique_id / tghip_comm_init_rank / tghip_reduce_framebuffer_rank: the exchange step of one process per GPU (bench.py --gpus N under
    torch.distributed.run) with a world of ONE rank -- the whole RCCL path (id, ncclCommInitRank, grouped ncclReduce into the root's scratch image,
    download) on the one device every box has; the merged image is the render's, bit for bit, the context's own framebuffer untouched."""
    import ctypes as C
    w, h, spp = 200, 120, 4
    path = scenes.cornell(tmp_path, resolution=(w, h), spp=spp)
    whole, wsum, wcount, _ = gpu_render(path)
    flat, ctxs = _shard_contexts(path, 1, spp)
    ctx, npix = ctxs[0], w*h
    s, c = np.full((npix, 3), -1, np.float32), np.full(npix, 7, np.uint32)
    assert tg.lib.tghip_reduce_framebuffer_rank(ctx, 0, s.ctypes.data, c.ctypes.data, npix) == -1          # no communicator yet
    assert b"tghip_comm_init_rank" in tg.lib.tghip_last_error(ctx)
    ident = (C.c_ubyte*tg.capi.TGHIP_COMM_ID_BYTES)()
    assert tg.lib.tghip_comm_unique_id(ident, 64) == -1                                                    # buffer too small
    assert tg.lib.tghip_comm_unique_id(ident, len(ident)) == 0 and any(ident)
    assert tg.lib.tghip_comm_init_rank(ctx, ident, len(ident), 1, 1) == -1                                 # rank outside the world
    assert tg.lib.tghip_comm_init_rank(ctx, ident, len(ident), 1, 0) == 0, tg.lib.tghip_last_error(ctx)
    assert tg.lib.tghip_reduce_framebuffer_rank(ctx, 0, None, None, npix) == 0                             # (merged image stays in HBM)
    assert tg.lib.tghip_reduce_framebuffer_rank(ctx, 0, s.ctypes.data, c.ctypes.data, npix) == 0, tg.lib.tghip_last_error(ctx)
    assert (c.reshape(h, w) == wcount).all() and s.reshape(h, w, 3).tobytes() == wsum.tobytes()
    assert tg.lib.tghip_reduce_framebuffer_rank(ctx, 1, s.ctypes.data, c.ctypes.data, npix) == -1          # root outside the world
    assert tg.lib.tghip_reduce_framebuffer_rank(ctx, 0, s.ctypes.data, c.ctypes.data, npix + 1) == -1
    fs, fc = np.empty((npix, 3), np.float32), np.empty(npix, np.uint32)
    assert tg.lib.tghip_download_framebuffer(ctx, fs.ctypes.data, fc.ctypes.data, npix) == 0 and fs.tobytes() == wsum.tobytes()
    # a second communicator on the same context replaces the first
    assert tg.lib.tghip_comm_unique_id(ident, len(ident)) == 0
    assert tg.lib.tghip_comm_init_rank(ctx, ident, len(ident), 1, 0) == 0
    assert tg.lib.tghip_reduce_framebuffer_rank(ctx, 0, s.ctypes.data, c.ctypes.data, npix) == 0 and s.reshape(h, w, 3).tobytes() == wsum.tobytes()
    tg.lib.tghip_destroy(ctx)
    flat.close()


def test_a_failing_reduce_falls_back_to_the_host_sum(tmp_path, capfd):
    """tghip_reduce_framebuffers made to fail (the "fail_reduce" fault-injection option): the C-ABI call reports TGHIP_E_HIP with a message and
    leaves the shards' framebuffers alone, and the integrator's exchange step (Integrator.cpp: fetchFramebuffer) falls back to per-device
    downloads summed on the host -- the image of the unsharded render bit for bit -- and says so once on stderr."""
    import ctypes as C
    import json
    w, h, spp = 200, 120, 4
    path = scenes.cornell(tmp_path, resolution=(w, h), spp=spp)
    whole, wsum, wcount, _ = gpu_render(path)
    # behind the C-ABI, one rank: the working path first, then the injected failure, then the working path again
    flat, ctxs = _shard_contexts(path, 1, spp)
    npix = w*h
    s, c = np.empty((npix, 3), np.float32), np.empty(npix, np.uint32)
    arr = (C.c_void_p*1)(*ctxs)
    assert tg.lib.tghip_reduce_framebuffers(arr, 1, 0, s.ctypes.data, c.ctypes.data, npix) == 0
    assert tg.lib.tghip_set_option(ctxs[0], b"fail_reduce", 1) == 0
    s2 = np.full_like(s, -1)
    assert tg.lib.tghip_reduce_framebuffers(arr, 1, 0, s2.ctypes.data, c.ctypes.data, npix) == -3      # TGHIP_E_HIP
    assert b"forced failure" in tg.lib.tghip_last_error(ctxs[0]) and (s2 == -1).all()
    assert tg.lib.tghip_set_option(ctxs[0], b"fail_reduce", 0) == 0
    assert tg.lib.tghip_reduce_framebuffers(arr, 1, 0, s2.ctypes.data, c.ctypes.data, npix) == 0 and s2.tobytes() == s.tobytes() == wsum.tobytes()
    tg.lib.tghip_destroy(ctxs[0])
    flat.close()
    # through the integrator: two shards, the reduce fails, the host sums
    d = json.load(open(path))
    d["integrator"].update(devices=2, share_devices=True)
    shared = os.path.join(str(tmp_path), "two.json")
    json.dump(d, open(shared, "w"))
    r = tg.Renderer(shared, seed=SEED)
    r.set_option("fail_reduce", 1)
    r.render()
    mean, ssum, count = r.image()
    r.close()
    assert (count == wcount).all() and ssum.tobytes() == wsum.tobytes()
    assert "summing the shards on the host" in capfd.readouterr().err


@pytest.mark.gpu
@pytest.mark.parametrize("scene,adaptive", [("cornell", False), ("materialtest", False), ("materialtest", True)])
def test_several_contexts_driven_by_host_threads_on_one_device(scene, adaptive, tmp_path):
    """The in-process multi-device path of the integrator ("devices": N -- N contexts, each rendering its tile shard and driven by
    its own host thread, merged by tghip_reduce_framebuffers or, as here where the contexts share a device, by the host sum)
    exercised on ONE GPU ("share_devices"): three shards give the image of the unsharded render bit for bit, records included."""
    import json
    _skip_mt(scene)
    kw = dict(resolution=(192, 108), spp=8)
    if adaptive:      # (the scheduler starts to move samples once every record has 16: three passes of 16)
        kw = dict(resolution=(96, 54), spp=48, spp_step=16, renderer={"adaptive_sampling": True, "stratified_sampler": True})
    path = scenes.materialtest(tmp_path, **kw) if scene == "materialtest" else scenes.cornell(tmp_path, **kw)
    base, _, cnt, _ = gpu_render(path)
    d = json.load(open(path))
    d["integrator"].update(devices=3, share_devices=True)
    shared = os.path.join(str(tmp_path), "shared.json")
    json.dump(d, open(shared, "w"))
    r = tg.Renderer(shared, seed=SEED)
    assert r.context(2), "three contexts expected"
    r.render()
    mean, _, count = r.image()
    r.close()
    assert (count == cnt).all()
    if adaptive:
        assert count.min() < count.max()                 # (the pass scheduler really moved samples)
    assert (mean == base).all()


def test_upload_refuses_instance_descriptors_with_a_bad_top_record_count(tmp_path):
    """A scene with instances indexes recs[] and inst_tight_boxes[] by num_top_recs (tghip_upload_scene: the instance trees' leaves, the tight-box
    upload): a count of 0 or beyond num_recs, or missing tight boxes, is refused with TGHIP_E_INVALID before anything is dereferenced."""
    import ctypes as C
    mk, kw = scenes.GOLDEN_CASES["cornell_instances"]
    flat = tg.FlattenedScene(mk(tmp_path, **dict(kw, resolution=(32, 18), spp=1)))
    d = flat.desc.contents
    assert d.num_instances > 0 and 0 < d.num_top_recs <= d.num_recs
    ctx = tg.lib.tghip_create(0)
    assert ctx
    assert tg.lib.tghip_upload_scene(ctx, flat.desc) == 0
    for field, value in (("num_top_recs", 0), ("num_top_recs", d.num_recs + 1), ("inst_tight_boxes", None)):
        bad = tg.TgHipSceneDesc.from_buffer_copy(d)
        if value is None:
            bad.inst_tight_boxes = C.cast(None, type(bad.inst_tight_boxes))
        else:
            setattr(bad, field, value)
        assert tg.lib.tghip_upload_scene(ctx, C.byref(bad)) == -1, (field, value)
        assert b"malformed scene description" in tg.lib.tghip_last_error(ctx)
    tg.lib.tghip_destroy(ctx)
    flat.close()
