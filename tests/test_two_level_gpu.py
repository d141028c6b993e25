"""Two-level trees (HRT_CTX_TWO_LEVEL): the reference's IAS over shared GASes (src/Global/RendererImpl.cu:174-242; GAS chosen by
shapeID, src/Global/RendererTime.cu:116-130; one IAS per time step, :87-151) kept as a structure -- a top level over the instances whose
leaves are transform nodes, one object-space tree per unique BLAS behind it -- and traced by k_fused<.., INSTANCED>.

Parity bar: hit records and frames BIT-EXACT against the oracle's INSTANCED mode (oracle.c oracle_scene_create_mode: the ray goes into
the instance's object space, as at an IAS leaf); the flattened trees of every other test stay pinned to the FLATTENED mode.
Size-independent properties at DEM sizes: memory and update cost grow with instances + unique primitives."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _two_level(hrt, gpu_available, flags=0):
    if not gpu_available:
        pytest.skip("no GPU in this container")
    return hrt.Renderer(0, hrt.CTX_TWO_LEVEL | flags)


def _hits_match(oracle, r, scene, n_rays=30000, seed=31, scale=1.6, any_hit=False, brute=True):
    o, d = oracle.random_rays(n_rays, seed, scale)
    t, u, v, prim, inst = r.trace_rays(o, d, any_hit=any_hit)
    rt, ru, rv, rprim, rinst = oracle.OracleScene(scene, force_brute=brute, instanced=True).trace(o, d)
    if any_hit:      # which hit an any-hit query reports is the traversal's business: hit / no hit is not
        assert np.array_equal(prim != 0xFFFFFFFF, rprim != 0xFFFFFFFF)
        return
    assert np.array_equal(prim, rprim) and np.array_equal(inst, rinst)
    assert np.array_equal(t.view(np.uint32), rt.view(np.uint32))
    assert np.array_equal(u.view(np.uint32), ru.view(np.uint32)) and np.array_equal(v.view(np.uint32), rv.view(np.uint32))


def _frame_matches(oracle, r, scene, w, h, salt, spp=1, states=None):
    if states is None:
        r.set_frame(w, h, salt, linear=True)
        states = oracle.rng_init(w, h, salt)
    r.render(spp)
    ref = oracle.OracleScene(scene, instanced=True).render(w, h, states, spp)
    assert np.array_equal(r.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32)), "linear radiance must be bit-exact"
    assert np.array_equal(r.color.cpu().numpy().view(np.uint32), ref["color"].view(np.uint32))
    return states, ref


def test_two_level_hits_and_frames_against_the_instanced_oracle(hrt, oracle, gpu_available):
    """All four programs, transformed and identity instances, spheres and triangles, BLASes shared by many instances: hit records
    (t, u, v, primitive, instance) against the instanced oracle's brute force, any-hit queries, rendered frames at several spp."""
    r = _two_level(hrt, gpu_available)
    try:
        for scene, (w, h, spp) in ((hrt.scenes.mixed_test_scene(2500, 50, 13, 120, 80, 2), (120, 80, 2)),
                                   (hrt.scenes.cornell_box(96, 96, 2), (96, 96, 2)),
                                   (hrt.scenes.sphere_in_box(64, 64, 3), (64, 64, 3)),
                                   (hrt.scenes.particle_scene(40, 100, 70, 1, frame=2), (100, 70, 1))):
            r.load_scene(scene)
            s = r.stats()
            _hits_match(oracle, r, scene)
            _hits_match(oracle, r, scene, any_hit=True)
            _frame_matches(oracle, r, scene, w, h, 77, spp)
            s = r.stats()
            assert s.fused_fallback_launches == 0
    finally:
        r.close()


def test_two_level_differs_from_the_flattened_mode_only_in_rounding(hrt, oracle, gpu_available):
    """The same scene through both structures: the same primitives are hit, t agrees to a few ULP, and each structure is bit-exact
    against ITS oracle mode -- the two canonical modes are one geometry, two roundings."""
    scene = hrt.scenes.particle_scene(60, 96, 64, 1, frame=3)
    o, d = oracle.random_rays(40000, 5)
    out = {}
    for name, flags in (("flat", 0), ("two", hrt.CTX_TWO_LEVEL)):
        if not gpu_available:
            pytest.skip("no GPU")
        r = hrt.Renderer(0, flags)
        try:
            r.load_scene(scene)
            out[name] = r.trace_rays(o, d)
        finally:
            r.close()
    ref_flat = oracle.OracleScene(scene, force_brute=True).trace(o, d)
    ref_two = oracle.OracleScene(scene, force_brute=True, instanced=True).trace(o, d)
    for got, ref in ((out["flat"], ref_flat), (out["two"], ref_two)):
        assert np.array_equal(got[3], ref[3]) and np.array_equal(got[0].view(np.uint32), ref[0].view(np.uint32))
    hit = out["flat"][3] != 0xFFFFFFFF
    same = (out["flat"][3] == out["two"][3]) & (out["flat"][4] == out["two"][4])
    assert same.mean() > 0.999                       # (an edge-on hit may go to the neighbour triangle in one of the roundings)
    both = hit & same
    rel = np.abs(out["flat"][0][both].astype(np.float64) - out["two"][0][both]) / out["flat"][0][both]
    assert rel.max() < 1e-4 and np.median(rel) < 1e-6
    assert (out["flat"][0].view(np.uint32) != out["two"][0].view(np.uint32)).any()      # and they ARE two roundings


def test_two_level_animation_refits_the_top_level_only(hrt, oracle, gpu_available):
    """updateIAS on a two-level tree: new transforms every frame, the refit rewrites transform nodes and the boxes above them; every
    frame bit-exact against the instanced oracle on that frame's scene, RNG streams carried across frames (Q8); an update with the
    build's transforms rewrites the very bytes; synchronous and asynchronous updates."""
    for flags in (0, hrt.CTX_ASYNC_UPDATE):
        r = _two_level(hrt, gpu_available, flags)
        try:
            n_p, w, h = 40, 72, 48
            scene = hrt.scenes.particle_scene(n_p, w, h, 1, frame=0)
            r.load_scene(scene)
            from test_gpu_parity import _download_tree
            nodes0, prims0 = _download_tree(hrt, r)
            r.update_instances([it["transform"] for it in scene["instances"]])
            r.update_instances([it["transform"] for it in scene["instances"]])
            nodes1, prims1 = _download_tree(hrt, r)
            assert np.array_equal(nodes0, nodes1) and np.array_equal(prims0, prims1)
            before = r.stats()
            ground = scene["instances"][0]["transform"]
            r.set_frame(w, h, 5, linear=True)
            states = oracle.rng_init(w, h, 5)
            for frame in (1, 2, 5):
                poses = hrt.scenes.particle_poses(n_p, frame)
                r.update_instances([ground] + poses)
                for it, m in zip(scene["instances"][1:], poses):
                    it["transform"] = m
                _frame_matches(oracle, r, scene, w, h, 5, 1, states)
                _hits_match(oracle, r, scene, 8000)
            after = r.stats()
            assert after.tlas_refits == before.tlas_refits + 3 and after.tlas_rebuilds == before.tlas_rebuilds
        finally:
            r.close()


def test_two_level_rebuild_when_the_particles_start_on_top_of_each_other(hrt, oracle, gpu_available, monkeypatch):
    """The reference builds each file's IAS with identity transforms and poses it afterwards (RendererTime.cu:111-127): the first
    update rebuilds -- at once when it sees how far the instances have gone, else (HRT_REFIT_MOVED_FAR=0) after a refit that is
    checked on the spot -- as a two-level tree again."""
    monkeypatch.setenv("HRT_REFIT_MOVED_FAR", "0")
    r = _two_level(hrt, gpu_available)
    try:
        n_p, w, h = 30, 72, 48
        scene = hrt.scenes.particle_scene(n_p, w, h, 1, frame=0)
        posed = [it["transform"].copy() for it in scene["instances"]]
        for it in scene["instances"][1:]:
            it["transform"] = hrt.scenes.IDENTITY.copy()
        r.load_scene(scene)
        _hits_match(oracle, r, scene, 5000)
        r.update_instances(posed)
        s = r.stats()
        assert s.tlas_refits == 1 and s.tlas_rebuilds == 2 and s.tlas_refit_ratio > 1.5
        for it, m in zip(scene["instances"], posed):
            it["transform"] = m
        _hits_match(oracle, r, scene)
        _frame_matches(oracle, r, scene, w, h, 11)
    finally:
        r.close()


def test_two_level_memory_and_update_cost_grow_with_instances(hrt, oracle, gpu_available):
    """2000 particles of three shared shapes: the flattened tree holds a record per instance x primitive, the two-level tree the
    three shapes' records plus a transform node and a share of a box node per instance; an update touches the top level only.
    Frame and hit records bit-exact against the instanced oracle at this size too."""
    if not gpu_available:
        pytest.skip("no GPU")
    scene = hrt.scenes.particle_cloud(2000, 160, 120, 1)
    flat_prims = sum(len(it["vertices"]) for it in scene["instances"] if it["geometry"] == "triangles") + 1
    unique_prims = sum(len(v) for v in {id(it["vertices"]): it["vertices"] for it in scene["instances"] if it["geometry"] == "triangles"}.values()) + 1
    sizes, upd = {}, {}
    for name, flags in (("flat", 0), ("two", hrt.CTX_TWO_LEVEL)):
        r = hrt.Renderer(0, flags)
        try:
            r.load_scene(scene)
            r.set_frame(160, 120, 3, linear=True)
            r.render(1)
            s = r.stats()
            sizes[name] = (s.bvh_alloc_bytes, s.bvh_nodes, s.bvh_triangles + s.bvh_spheres)
            xf = [it["transform"] for it in scene["instances"]]
            r.update_instances(xf)
            r._torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                r.update_instances(xf)
            r._torch.cuda.synchronize()
            upd[name] = (time.perf_counter() - t0) / 5
            if name == "two":
                _hits_match(oracle, r, scene, 20000, scale=2.5, brute=False)
                _frame_matches(oracle, r, scene, 160, 120, 3)
        finally:
            r.close()
    assert sizes["flat"][2] == flat_prims and sizes["two"][2] == unique_prims
    # 2001 transform nodes + ~300 box nodes + the three shapes: a few hundred KB against tens of MB
    assert sizes["two"][0] < sizes["flat"][0] / 20, sizes
    assert sizes["two"][0] < 2001 * 400 + unique_prims * 200, sizes
    print("two-level: %d bytes, %d nodes; flattened: %d bytes, %d nodes; update %.3f ms vs %.3f ms" %
          (sizes["two"][0], sizes["two"][1], sizes["flat"][0], sizes["flat"][1], upd["two"] * 1e3, upd["flat"] * 1e3))


def test_two_level_hundred_thousand_particles(hrt, oracle, gpu_available):
    """A DEM-sized run: 10^5 particles of three shared shapes (3.3 M triangles flattened, 80 unique): built and updated in time and
    memory proportional to the instances, hit records and a frame bit-exact against the instanced oracle."""
    r = _two_level(hrt, gpu_available)
    try:
        n_p = 100_000
        scene = hrt.scenes.particle_cloud(n_p, 192, 128, 1, subdiv=1)
        t0 = time.perf_counter()
        r.load_scene(scene)
        t_load = time.perf_counter() - t0
        s0 = r.stats()
        _hits_match(oracle, r, scene, 60000, scale=6.0, brute=False)
        _frame_matches(oracle, r, scene, 192, 128, 9)
        s = r.stats()
        assert s.bvh_triangles + s.bvh_spheres == 32 + 8 + 12 + 1
        assert s.bvh_alloc_bytes < (n_p + 1) * 300, s.bvh_alloc_bytes      # ~ a transform node + a seventh of a box node + their boxes per instance
        xf = [it["transform"] for it in scene["instances"]]
        r._torch.cuda.synchronize()
        print("10^5 particles: load_scene %.2f s, tree %d bytes, %d nodes, depth %d" % (t_load, s.bvh_alloc_bytes, s.bvh_nodes, s.bvh_depth))
        assert s.fused_fallback_launches == 0 and s0.tlas_rebuilds == 1
        # a synchronous update at this size: the instance array comes back through pinned memory and the host derives the tables on a few
        # threads (hrt_accel.cpp download_instances / instance_tables) -- the same tables, the same frame as the oracle's of the moved scene
        for it in scene["instances"][1:]:
            m = it["transform"].copy(); m[3] += np.float32(0.004); m[11] -= np.float32(0.006); it["transform"] = m
        r.update_instances([it["transform"] for it in scene["instances"]])
        after = r.stats()
        assert after.tlas_refits == s.tlas_refits + 1 and after.tlas_rebuilds == s.tlas_rebuilds
        _frame_matches(oracle, r, scene, 192, 128, 10)
    finally:
        r.close()


def test_two_level_is_chosen_by_size_and_refused_where_it_cannot_be_traced(hrt, oracle, gpu_available, monkeypatch):
    """Without the flag a scene is flattened unless that would leave the caches (threshold lowered here); a counting context and the
    other execution modes flatten whatever is asked for; HRT_TWO_LEVEL=-1 never builds one."""
    if not gpu_available:
        pytest.skip("no GPU")
    scene = hrt.scenes.particle_cloud(500, 96, 64, 1)

    def nodes_of(flags):
        r = hrt.Renderer(0, flags)
        try:
            r.load_scene(scene)
            r.set_frame(96, 64, 1, linear=True)
            r.render(1)
            return r.stats().bvh_triangles
        finally:
            r.close()
    flat = nodes_of(0)
    assert nodes_of(hrt.CTX_TWO_LEVEL) < flat / 50
    assert nodes_of(hrt.CTX_TWO_LEVEL | hrt.CTX_COUNT) == flat
    monkeypatch.setenv("HRT_TWO_LEVEL_MIN_PRIMS", "1000")
    assert nodes_of(0) < flat / 50
    monkeypatch.setenv("HRT_TWO_LEVEL", "-1")
    assert nodes_of(0) == flat
    monkeypatch.delenv("HRT_TWO_LEVEL")
    monkeypatch.setenv("HRT_FUSED", "0")
    assert nodes_of(hrt.CTX_TWO_LEVEL) == flat


def test_rays_at_the_edge_of_an_instances_bounding_sphere(hrt, oracle, gpu_available):
    """A ray that misses an instance's bounding sphere does not enter it (transform node words 0-2, 7; fused.hip).  The test is culling
    only and has to stay so at its edges: rays that graze the sphere or the farthest vertices, rays that start inside it or on it,
    origins 10^5 radii away (where the terms of the test round coarsely), instances that are scaled unevenly and sheared (the sphere
    is tested in OBJECT space: any affine map).  Hit records bit-exact against the instanced oracle's brute force."""
    r = _two_level(hrt, gpu_available)
    try:
        scene = hrt.scenes.particle_scene(9, 64, 48, 1, frame=1)
        rng = np.random.default_rng(17)
        for k, it in enumerate(scene["instances"][1:]):           # (the ground sphere stays as it is)
            m = it["transform"].reshape(3, 4).astype(np.float64)
            lin = m[:, :3] @ np.diag([1.0 + 0.8 * (k % 3), 0.5 + 0.25 * (k % 2), 1.0]) @ np.array([[1, 0.3 * (k % 2), 0], [0, 1, 0], [0.2 * (k % 3 == 0), 0, 1]])
            it["transform"] = np.concatenate([lin, m[:, 3:]], axis=1).astype(np.float32).reshape(12)
        r.load_scene(scene)
        assert r.stats().fused_fallback_launches == 0
        origins, dirs = [], []
        for it in scene["instances"][1:]:
            m = it["transform"].reshape(3, 4).astype(np.float64)
            v = it["vertices"].reshape(-1, 3).astype(np.float64)
            c_o = 0.5 * (v.min(0) + v.max(0)); rad = np.linalg.norm(v - c_o, axis=1).max()
            far = v[np.argsort(-np.linalg.norm(v - c_o, axis=1))[:6]]                    # the vertices that define the sphere
            n = 400
            u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
            t = np.cross(u, rng.normal(size=(n, 3))); t /= np.linalg.norm(t, axis=1, keepdims=True)
            scale = rng.choice([0.98, 0.999, 1.0, 1.0001, 1.001, 1.02], size=(n, 1))
            target = np.where(rng.random((n, 1)) < 0.3, far[rng.integers(0, len(far), n)], c_o + u * rad * scale)   # on / just off the sphere, or a far vertex
            dist = rng.choice([0.0, 0.5, 1.0, 3.0, 1e2, 1e5], size=(n, 1)) * rad                # start inside, on, near, very far
            o_obj = target - t * dist
            d_obj = t * rng.choice([1.0, 1e-3, 37.0], size=(n, 1))                              # (t is shared by both spaces: unnormalised directions)
            origins.append(o_obj @ m[:, :3].T + m[:, 3]); dirs.append(d_obj @ m[:, :3].T)      # object -> world
        o = np.concatenate(origins).astype(np.float32); d = np.concatenate(dirs).astype(np.float32)
        t_, u_, v_, prim, inst = r.trace_rays(o, d)
        rt, ru, rv, rprim, rinst = oracle.OracleScene(scene, force_brute=True, instanced=True).trace(o, d)
        assert np.array_equal(prim, rprim) and np.array_equal(inst, rinst) and np.array_equal(t_.view(np.uint32), rt.view(np.uint32))
        assert (prim != 0xFFFFFFFF).sum() > 500                                                 # (and plenty of them do hit)
    finally:
        r.close()


@pytest.mark.parametrize("flags_name", ["flattened", "two_level"])
def test_every_leaf_hold_gives_the_same_bits(hrt, oracle, gpu_available, monkeypatch, flags_name):
    """HRT_LEAF_HOLD (how many leaf groups a lane may queue before its node work waits; by scene when unset: 4 / 2 / 1) and the
    regeneration threshold change WHEN primitives are tested, never what is hit: frames and ray counts are the oracle's at every value."""
    if not gpu_available:
        pytest.skip("no GPU in this container")
    w, h, spp = 128, 96, 4
    scene = hrt.scenes.particle_cloud(400, w, h, spp)
    flags = hrt.CTX_TWO_LEVEL if flags_name == "two_level" else 0
    ref = oracle.OracleScene(scene, instanced=flags != 0).render(w, h, oracle.rng_init(w, h, 5), spp)
    for env in ({}, {"HRT_LEAF_HOLD": "1"}, {"HRT_LEAF_HOLD": "2"}, {"HRT_LEAF_HOLD": "3"}, {"HRT_LEAF_HOLD": "4"},
                {"HRT_LEAF_HOLD": "1", "HRT_REFILL_THRESHOLD": "1"}, {"HRT_LEAF_HOLD": "1", "HRT_REFILL_THRESHOLD": "64", "HRT_POSTPONE_PCT": "100"}):
        for k in ("HRT_LEAF_HOLD", "HRT_REFILL_THRESHOLD", "HRT_POSTPONE_PCT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = hrt.Renderer(0, flags)
        try:
            r.load_scene(scene)
            r.set_frame(w, h, 5, linear=True)
            r.reset_stats()
            r.render(spp)
            assert np.array_equal(r.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32)), env
            assert int(r.stats().rays) == ref["rays"] and r.stats().fused_fallback_launches == 0
        finally:
            r.close()
