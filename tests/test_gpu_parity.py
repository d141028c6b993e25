"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, same seeded inputs.

Bars (stated per test):
  * integer / index work (RNG states, hit primitive + instance, ray counts): bit-exact
  * hit t,u,v and the linear radiance: bit-exact (all control-flow arithmetic is pinned)
  * instance transforms from the pose kernel: bit-exact.  sinf / cosf / acosf / asinf / atan2f are pinned on both sides as the
    correctly rounded float (oracle: libm double + __float128 near ties; kernel: csrc/cr_trig.h), checked over every float
    (test_trig_pin_every_float_on_the_gpu); frames rendered from posed instances are compared with the oracle rendering
    from ITS OWN transforms.  No parity test carries a tolerance.
  * sRGB colour (float and 8-bit): bit-exact.  The shader's powf(c, 1/2.4f) is pinned on both sides as the correctly
    rounded float of c^y (oracle: libm double pow + __float128 near ties; kernels: csrc/srgb_pow.h), checked over every
    float in [0, 1] (test_color_conversion_all_floats_bit_exact); libm's powf itself is within 1 ULP of that
    (tests/test_oracle_cpu.py), well inside north_star's 1e-5.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ulp_diff(a, b):
    ai = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    bi = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.abs(ai - bi)


def _render_both(hrt, oracle, renderer, scene, width, height, spp, rows=None, tile=None):
    salt = hrt.scenes.SEED_SALT
    renderer.load_scene(scene)
    renderer.set_frame(width, height, salt, aov=True, linear=True)
    renderer.render(spp, tile=tile)
    osc = oracle.OracleScene(scene)
    states = oracle.rng_init(width, height, salt)
    ref = osc.render(width, height, states, spp, rows=rows)
    return ref, states


def _check_image(renderer, ref, rows=None):
    color = renderer.color.cpu().numpy()
    linear = renderer.linear.cpu().numpy()
    sel = slice(None) if rows is None else np.asarray(rows)
    assert np.array_equal(linear[sel].view(np.uint32), ref["linear"][sel].view(np.uint32)), "linear radiance must be bit-exact"
    assert np.array_equal(color[sel].view(np.uint32), ref["color"][sel].view(np.uint32)), "sRGB colour must be bit-exact"
    # quirk Q3: AOVs are always zero
    assert np.array_equal(renderer.albedo.cpu().numpy()[sel], ref["albedo"][sel])
    assert np.array_equal(renderer.normal.cpu().numpy()[sel], ref["normal"][sel])


def test_rng_init_bit_exact(hrt, oracle, renderer):
    # 70x33: neither dimension is a multiple of 16 (the reference's quirk Q9 case)
    w, h = 70, 33
    renderer.set_frame(w, h, hrt.scenes.SEED_SALT, aov=False)
    got = renderer.rng_states_numpy()
    want = oracle.rng_init(w, h, hrt.scenes.SEED_SALT)
    assert np.array_equal(got[:, :6], want[:, :6])
    assert np.array_equal(got, want)


def test_rng_init_other_salt_and_large_index(hrt, oracle, renderer):
    w, h = 1024, 130            # indices up to 2^17: exercises many jump matrices
    renderer.set_frame(w, h, 12345678901234567, aov=False)
    got = renderer.rng_states_numpy()
    want = oracle.rng_init(w, h, 12345678901234567)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("any_hit", [False, True])
def test_traverse_matches_bruteforce_triangles(hrt, oracle, renderer, any_hit):
    scene = hrt.scenes.random_soup(6000, 0.08, 5)
    renderer.load_scene(scene)
    o, d = oracle.random_rays(40000, 11)
    t, u, v, prim, inst = renderer.trace_rays(o, d, any_hit=any_hit)
    brute = oracle.OracleScene(scene, force_brute=True)
    rt, ru, rv, rprim, rinst = brute.trace(o, d, any_hit=any_hit)
    if any_hit:
        # any-hit: which primitive is found first is traversal-order dependent; hit/miss is not
        assert np.array_equal(prim != 0xFFFFFFFF, rprim != 0xFFFFFFFF)
    else:
        assert np.array_equal(prim, rprim) and np.array_equal(inst, rinst)
        assert np.array_equal(t.view(np.uint32), rt.view(np.uint32))
        assert np.array_equal(u.view(np.uint32), ru.view(np.uint32))
        assert np.array_equal(v.view(np.uint32), rv.view(np.uint32))
    assert (rprim != 0xFFFFFFFF).mean() > 0.3


def test_rays_with_signed_zero_direction_components(hrt, oracle, renderer):
    """A direction component of -0.0 -- a ray mirrored by an axis-aligned wall -- counts as positive in the slab test (`d < 0`); its guarded
    reciprocal was negative (copysignf), near and far swapped, every box was culled and the ray left the room: one pixel of a Cornell box
    at 64x94, found in round 4 by tools/stress_modes.py.  Axis-parallel and in-plane rays with every combination of +0.0 / -0.0, closest
    and any-hit, against brute force; and that frame against the oracle."""
    scene = hrt.scenes.cornell_box(64, 94, 3)
    renderer.load_scene(scene)
    o, d = oracle.axis_parallel_rays(60000, 7)
    brute = oracle.OracleScene(scene, force_brute=True)
    t, u, v, prim, inst = renderer.trace_rays(o, d)
    rt, ru, rv, rprim, rinst = brute.trace(o, d)
    assert np.array_equal(prim, rprim) and np.array_equal(inst, rinst) and np.array_equal(t.view(np.uint32), rt.view(np.uint32))
    a = renderer.trace_rays(o, d, any_hit=True)
    assert np.array_equal(a[3] != 0xFFFFFFFF, rprim != 0xFFFFFFFF)
    renderer.set_frame(64, 94, 525075280, linear=True)
    renderer.render(3)
    ref = oracle.OracleScene(scene).render(64, 94, oracle.rng_init(64, 94, 525075280), 3)
    assert np.array_equal(renderer.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32))


def test_traverse_mixed_scene_with_transforms(hrt, oracle, renderer):
    scene = hrt.scenes.mixed_test_scene(3000, 60, 9)
    renderer.load_scene(scene)
    o, d = oracle.random_rays(30000, 21)
    t, u, v, prim, inst = renderer.trace_rays(o, d)
    rt, ru, rv, rprim, rinst = oracle.OracleScene(scene, force_brute=True).trace(o, d)
    assert np.array_equal(prim, rprim) and np.array_equal(inst, rinst)
    assert np.array_equal(t.view(np.uint32), rt.view(np.uint32))
    assert np.array_equal(u.view(np.uint32), ru.view(np.uint32)) and np.array_equal(v.view(np.uint32), rv.view(np.uint32))
    assert set(np.unique(rinst[rinst != 0xFFFFFFFF])) == {0, 1, 2, 3, 4}


def test_traverse_counts_agree_with_cpu_walk(hrt, oracle, renderer):
    """Node-visit / primitive-test counters (HRT_CTX_COUNT) against a CPU walk of the same BVH bytes.
    The counting build of the kernel walks in the canonical order (all leaves of a node before the next
    node), which is what the CPU walker does, so the totals agree exactly -- and the hits still match."""
    import ctypes as C
    scene = hrt.scenes.random_soup(20000, 0.05, 3)
    renderer.set_flags(hrt.CTX_COUNT)
    renderer.load_scene(scene)
    o, d = oracle.random_rays(20000, 5)
    renderer.reset_stats()
    got = renderer.trace_rays(o, d)
    s = renderer.stats()
    blob = hrt.BvhBlob()
    assert renderer.lib.hrt_tlas_download(renderer.ctx, renderer.tlas, C.byref(blob)) == 0
    res = oracle.bvh8_trace(blob.nodes, blob.triangles, o, d)
    renderer.lib.hrt_host_free(C.byref(blob))
    assert s.node_visits == res[5] and s.prim_tests == res[6]
    assert np.array_equal(got[3], res[3]) and np.array_equal(got[0].view(np.uint32), res[0].view(np.uint32))
    # the production build (no counting) must give the same hits
    renderer.set_flags(0)
    fast = renderer.trace_rays(o, d)
    assert np.array_equal(fast[3], res[3]) and np.array_equal(fast[0].view(np.uint32), res[0].view(np.uint32))


def test_render_cornell_c1(hrt, oracle, renderer):
    """BASELINE configs[0]: Cornell box, 32 triangles, 256x256, 1 spp."""
    scene = hrt.scenes.cornell_box(256, 256, 1)
    renderer.reset_stats()
    ref, states = _render_both(hrt, oracle, renderer, scene, 256, 256, 1)
    _check_image(renderer, ref)
    assert np.array_equal(renderer.rng_states_numpy(), states), "RNG streams must end in the same state"
    s = renderer.stats()
    assert s.rays == ref["rays"] and s.paths == 256 * 256


def test_render_sphere_in_box_c2_spp(hrt, oracle, renderer):
    """BASELINE configs[1] at its own size: sphere + walls, 512 x 512, 16 spp (4.2 M paths: a few seconds of oracle)."""
    scene = hrt.scenes.sphere_in_box()
    assert (scene["width"], scene["height"], scene["spp"]) == (512, 512, 16)
    renderer.reset_stats()
    ref, states = _render_both(hrt, oracle, renderer, scene, 512, 512, 16)
    _check_image(renderer, ref)
    assert np.array_equal(renderer.rng_states_numpy(), states)
    assert renderer.stats().rays == ref["rays"]


def test_render_mixed_programs_and_transforms(hrt, oracle, renderer):
    """All four closest-hit programs, fuzz > 0 and = 0, transformed instances (quirks Q1/Q2), ragged frame."""
    scene = hrt.scenes.mixed_test_scene(2000, 40, 7, 97, 61, 3)
    ref, states = _render_both(hrt, oracle, renderer, scene, 97, 61, 3)
    _check_image(renderer, ref)
    assert np.array_equal(renderer.rng_states_numpy(), states)


def test_render_soup_small(hrt, oracle, renderer):
    scene = hrt.scenes.random_soup(50000, 0.04, 4, 320, 180, 2)
    renderer.reset_stats()
    ref, states = _render_both(hrt, oracle, renderer, scene, 320, 180, 2)
    _check_image(renderer, ref)
    assert renderer.stats().rays == ref["rays"]


def test_persistent_rng_across_launches(hrt, oracle, renderer):
    """Two 1-spp launches continue the per-pixel streams (quirk Q8), like two frames of the reference."""
    scene = hrt.scenes.cornell_box(64, 64, 1)
    salt = hrt.scenes.SEED_SALT
    renderer.load_scene(scene)
    renderer.set_frame(64, 64, salt, linear=True)
    osc = oracle.OracleScene(scene)
    states = oracle.rng_init(64, 64, salt)
    for _ in range(2):
        renderer.render(1)
        ref = osc.render(64, 64, states, 1)
        assert np.array_equal(renderer.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32))


def test_tile_union_equals_full_frame(hrt, oracle, renderer):
    """Multi-GPU split on one GPU: the stripes of 3 'ranks' rendered one after another give the
    1-GPU image bit-for-bit, rows outside a tile stay untouched, and the oracle agrees per tile."""
    scene = hrt.scenes.mixed_test_scene(1500, 30, 3, 80, 50, 2)
    salt = hrt.scenes.SEED_SALT
    ref, _ = _render_both(hrt, oracle, renderer, scene, 80, 50, 2)
    full_color = renderer.color.cpu().numpy().copy()
    full_linear = renderer.linear.cpu().numpy().copy()
    renderer.set_frame(80, 50, salt, linear=True)
    acc = np.zeros_like(full_color)
    for rank in range(3):
        renderer.color.zero_()
        tile = hrt.tile_for_rank(50, rank, 3, stripe_rows=4)
        renderer.render(2, tile=tile)
        part = renderer.color.cpu().numpy()
        rows = [y for y in range(50) if (y // 4) % 3 == rank]
        others = [y for y in range(50) if (y // 4) % 3 != rank]
        assert np.all(part[others] == 0)
        assert np.array_equal(part[rows], full_color[rows])
        acc += part                                       # what the RCCL reduce(sum) does
    assert np.array_equal(acc, full_color)
    assert np.array_equal(renderer.linear.cpu().numpy()[[y for y in range(50) if (y // 4) % 3 == 2]],
                          full_linear[[y for y in range(50) if (y // 4) % 3 == 2]])


def test_two_ranks_as_two_contexts_on_one_device(hrt, oracle, gpu_available):
    """The N = 2 device path without a second GPU: two contexts on device 0 -- each with its own scene copy, tree, RNG states and
    frame, as two ranks have -- render their stripes of the same frame at 16 spp (small tiles: the cost-ordered probe launch runs
    too); the reduce is a plain add.  The sum is the one-context frame bit for bit, the oracle's too, and the rays add up."""
    if not gpu_available:
        pytest.skip("no GPU")
    scene = hrt.scenes.mixed_test_scene(3000, 40, 11, 192, 128, 16)
    w, h, spp, salt = 192, 128, 16, hrt.scenes.SEED_SALT
    ranks = [hrt.Renderer(0, 0) for _ in range(2)]
    try:
        for r in ranks:
            r.load_scene(scene)
            r.set_frame(w, h, salt, linear=True)
        for k, r in enumerate(ranks):                      # enqueue both before waiting for either: two contexts in flight on one card
            r.color.zero_(); r.linear.zero_()
            r.render(spp, tile=hrt.tile_for_rank(h, k, 2), sync=False)
        for r in ranks:
            r._torch.cuda.synchronize()
        total = ranks[0].color + ranks[1].color            # ncclReduce(sum) / torch.distributed.reduce stand-in: x + 0 is exact
        total_lin = ranks[0].linear + ranks[1].linear
        rays = sum(r.stats().rays for r in ranks)
        one = hrt.Renderer(0, 0)
        try:
            one.load_scene(scene)
            one.set_frame(w, h, salt, linear=True)
            one.render(spp)
            assert np.array_equal(total.cpu().numpy().view(np.uint32), one.color.cpu().numpy().view(np.uint32))
            assert rays == one.stats().rays
        finally:
            one.close()
        ref = oracle.OracleScene(scene).render(w, h, oracle.rng_init(w, h, salt), spp)
        lin = total_lin.cpu().numpy()
        lin[..., 3] = 1.0                                   # (each rank writes alpha 1 in its own rows only)
        assert np.array_equal(lin.view(np.uint32), ref["linear"].view(np.uint32)) and rays == ref["rays"]
        for k, r in enumerate(ranks):                      # a rank's rows outside its stripes stayed zero
            others = [y for y in range(h) if (y // 8) % 2 != k]
            assert not r.color.cpu().numpy()[others].any()
    finally:
        for r in ranks:
            r.close()


def test_tile_then_empty_tile_then_same_tile(hrt, renderer):
    """The cached row list of a tile must not survive a launch that replaced it: tile A, an empty tile (nothing to
    render, returns at once), tile A again -- the third launch renders A afresh instead of returning with a stale frame."""
    scene = hrt.scenes.cornell_box(48, 40, 1)
    renderer.load_scene(scene)
    a = hrt.Tile(0, 24, 1, 1, 0)
    empty = hrt.Tile(8, 8, 1, 1, 0)
    renderer.set_frame(48, 40, 3, aov=False)
    renderer.render(1, tile=a)
    want = renderer.color.cpu().numpy().copy()
    assert want[:24].any() and not want[24:].any()
    renderer.set_frame(48, 40, 3, aov=False)              # same seed, zeroed frame
    renderer.render(1, tile=empty)
    assert not renderer.color.cpu().numpy().any()
    renderer.render(1, tile=a)
    assert np.array_equal(renderer.color.cpu().numpy(), want)


def test_to_rgba8(hrt, oracle, renderer):
    scene = hrt.scenes.cornell_box(96, 96, 1)
    _render_both(hrt, oracle, renderer, scene, 96, 96, 1)
    got = renderer.to_rgba8().cpu().numpy()
    src = renderer.color.cpu().numpy()
    want = np.zeros((96, 96, 4), np.uint8)
    oracle.lib().oracle_to_rgba8(src.ctypes.data, want.ctypes.data, 96, 96)
    # second sRGB encode (quirk Q7): byte output is bit-exact (shared correctly rounded pow)
    assert np.array_equal(got, want)


def test_color_conversion_all_floats_bit_exact(hrt, oracle, gpu_available):
    """colorToFloat4 and colorToUchar4 (DeviceFunctions.cuh:153-212) on the GPU against the oracle for EVERY float in
    [0, 1] (1 065 353 217 values, three per colour), plus the values the clamp has to deal with (negative, > 1, inf, NaN).
    Bar: bit-exact, floats and bytes.  (-0 is left out: C leaves the sign of fmaxf(0, -0) open -- libm returns -0, v_max_f32
    +0 -- and radiance is a sum of products of non-negative numbers, never -0.)"""
    import torch
    if not gpu_available:
        pytest.skip("no GPU")
    renderer = hrt.Renderer(0, 0)
    lib, L = renderer.lib, oracle.lib()
    one = 0x3F800000
    per = 3 * (1 << 23)                                   # floats per chunk (8 M colours)
    edge = np.array([-1.0, 1.5, np.inf, -np.inf, np.nan, 0.0031308, 0.00313080009, 0.00313079986, 1e-45, 1.0,
                     np.nextafter(np.float32(1), np.float32(0))], dtype=np.float32)
    first = 0
    while first <= one:
        n = min(per, one + 1 - first)
        bits = torch.arange(first, first + n, dtype=torch.int32, device=renderer.device)
        vals = bits.view(torch.float32)
        if first == 0:
            vals = torch.cat([vals, torch.from_numpy(edge).to(renderer.device)])
        pad = (-vals.numel()) % 3
        if pad:
            vals = torch.cat([vals, torch.zeros(pad, dtype=torch.float32, device=renderer.device)])
        m = vals.numel() // 3
        src = torch.ones((m, 4), dtype=torch.float32, device=renderer.device)
        src[:, :3] = vals.view(m, 3)
        dst = torch.empty_like(src)
        rgba = torch.empty((m, 4), dtype=torch.uint8, device=renderer.device)
        renderer._check(lib.hrt_color_to_float4(renderer.ctx, src.data_ptr(), dst.data_ptr(), m, renderer._stream()), "hrt_color_to_float4")
        renderer._check(lib.hrt_to_rgba8(renderer.ctx, src.data_ptr(), rgba.data_ptr(), m, 1, renderer._stream()), "hrt_to_rgba8")
        h_src = src.cpu().numpy()
        want = np.empty_like(h_src)
        L.oracle_color_to_float4_n(h_src.ctypes.data, want.ctypes.data, m)
        want8 = np.empty((m, 4), np.uint8)
        L.oracle_to_rgba8(h_src.ctypes.data, want8.ctypes.data, m, 1)
        got = dst.cpu().numpy()
        bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
        assert len(bad) == 0, [(float.hex(float(h_src[i, j])), float.hex(float(got[i, j])), float.hex(float(want[i, j]))) for i, j in bad[:8]]
        assert np.array_equal(rgba.cpu().numpy(), want8), hex(first)
        first += n
    renderer.close()


def test_empty_scene_and_all_miss(hrt, oracle, renderer):
    """No instances: every ray misses, the frame is colorToFloat4(background)."""
    scene = {"instances": [], "camera": hrt.scenes._soup_camera(), "background": hrt.scenes.BACKGROUND}
    ref, _ = _render_both(hrt, oracle, renderer, scene, 40, 30, 1)
    _check_image(renderer, ref)
    assert ref["rays"] == 40 * 30
    bg = np.array([float.fromhex(x) for x in ("0x1.b56792p-1", "0x1.d00ab6p-1", "0x1.e8ccbep-1")], dtype=np.float32)   # SURVEY.md 8(c) probe of the reference
    assert np.array_equal(renderer.color.cpu().numpy()[..., :3], np.broadcast_to(bg, (30, 40, 3)))


def _download_tree(hrt, renderer):
    import ctypes as C
    blob = hrt.BvhBlob()
    assert renderer.lib.hrt_tlas_download(renderer.ctx, renderer.tlas, C.byref(blob)) == 0
    nodes = np.ctypeslib.as_array(C.cast(blob.nodes, C.POINTER(C.c_uint8)), shape=(blob.n_nodes * 80,)).copy()
    prims = np.ctypeslib.as_array(C.cast(blob.triangles, C.POINTER(C.c_uint8)), shape=(max(blob.n_triangles, 1) * 48,)).copy()
    renderer.lib.hrt_host_free(C.byref(blob))
    return nodes, prims


def _moved_scene_matches_oracle(hrt, oracle, renderer, scene, width, height, salt, spp=1):
    renderer.set_frame(width, height, salt, linear=True)
    renderer.render(spp)
    osc = oracle.OracleScene(scene)
    st = oracle.rng_init(width, height, salt)
    ref = osc.render(width, height, st, spp)
    assert np.array_equal(renderer.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32))
    o, d = oracle.random_rays(20000, 31)
    t, u, v, prim, inst = renderer.trace_rays(o, d)
    rt, ru, rv, rprim, rinst = oracle.OracleScene(scene, force_brute=True).trace(o, d)
    assert np.array_equal(prim, rprim) and np.array_equal(inst, rinst) and np.array_equal(t.view(np.uint32), rt.view(np.uint32))


def test_refit_with_unchanged_transforms_reproduces_the_built_tree(hrt, renderer):
    """updateIAS with the transforms it was built with: the device refit (same arithmetic as the host
    flatten + quantisation) must write back the very bytes the builder produced -- nodes and records."""
    scene = hrt.scenes.mixed_test_scene(3000, 40, 5)
    renderer.load_scene(scene)
    nodes0, prims0 = _download_tree(hrt, renderer)
    before = renderer.stats()
    renderer.update_instances([it["transform"] for it in scene["instances"]])
    nodes1, prims1 = _download_tree(hrt, renderer)
    after = renderer.stats()
    assert after.tlas_refits == before.tlas_refits + 1 and after.tlas_rebuilds == before.tlas_rebuilds
    assert np.array_equal(prims0, prims1)
    assert np.array_equal(nodes0, nodes1)


def test_update_instances_refit_matches_oracle(hrt, oracle, gpu_available, monkeypatch):
    """updateIAS path (RendererMesh.cu:379-401): move, rotate and scale instances, refit on the device,
    re-render, compare with the oracle on the moved scene -- triangles and spheres (Q1/Q2 included).  The
    quality guard is switched off so that it is the refitted tree that is traced, however far things move."""
    if not gpu_available:
        pytest.skip("no GPU")
    monkeypatch.setenv("HRT_REFIT_REBUILD_RATIO", "1e30")
    r = hrt.Renderer(0, hrt.CTX_COUNT)
    try:
        scene = hrt.scenes.mixed_test_scene(900, 20, 5, 64, 48, 1)
        r.load_scene(scene)
        before = r.stats()
        moved = [it["transform"].copy() for it in scene["instances"]]
        moved[0][3] += 0.25
        moved[1] = hrt.scenes.rigid_transform((0.1, -0.2, 0.05), (0.3, 1.0, 0.2), 0.7, 1.1)
        moved[3][7] -= 0.2
        moved[4] = hrt.scenes.rigid_transform((-0.1, 0.1, 0.0), (0.0, 0.0, 1.0), 0.4)
        r.update_instances(moved)
        after = r.stats()
        assert after.tlas_refits == before.tlas_refits + 1 and after.tlas_rebuilds == before.tlas_rebuilds
        assert after.tlas_refit_ratio > 1.0                 # the boxes did grow
        for it, m in zip(scene["instances"], moved):
            it["transform"] = m
        _moved_scene_matches_oracle(hrt, oracle, r, scene, 64, 48, 99)
    finally:
        r.close()


def test_update_instances_rebuild_path(hrt, oracle, monkeypatch):
    """HRT_REFIT=0: every update rebuilds the tree on the host (the fallback when handles change)."""
    monkeypatch.setenv("HRT_REFIT", "0")
    r = hrt.Renderer(0, 0)
    try:
        scene = hrt.scenes.mixed_test_scene(900, 20, 5, 64, 48, 1)
        r.load_scene(scene)
        moved = [it["transform"].copy() for it in scene["instances"]]
        moved[0][3] += 0.25
        moved[3][7] -= 0.2
        r.update_instances(moved)
        s = r.stats()
        assert s.tlas_refits == 0 and s.tlas_rebuilds == 2
        for it, m in zip(scene["instances"], moved):
            it["transform"] = m
        _moved_scene_matches_oracle(hrt, oracle, r, scene, 64, 48, 99)
    finally:
        r.close()


def test_particle_animation_refit_every_frame(hrt, oracle, renderer):
    """The reference's Time mode in small: particles instancing shared shapes over the huge ground sphere,
    new transforms every frame (RendererTime.cu:436-480), updateIAS, one launch -- each frame bit-exact
    against the oracle on that frame's scene; the RNG streams carry over from frame to frame (Q8)."""
    w, h, salt = 72, 48, 4242
    scene = hrt.scenes.particle_scene(27, w, h, 1, frame=0)
    renderer.load_scene(scene)
    renderer.set_frame(w, h, salt, linear=True)
    states = oracle.rng_init(w, h, salt)
    n_p = 27
    for frame in range(4):
        poses = hrt.scenes.particle_poses(n_p, frame)
        ground = scene["instances"][0]["transform"]
        renderer.update_instances([ground] + poses)
        for it, m in zip(scene["instances"][1:], poses):
            it["transform"] = m
        renderer.render(1)
        ref = oracle.OracleScene(scene).render(w, h, states, 1)
        assert np.array_equal(renderer.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32)), "frame %d" % frame
        assert np.array_equal(renderer.rng_states_numpy(), states)
    s = renderer.stats()
    assert s.tlas_refits >= 4


def test_asynchronous_update_frames(hrt, oracle, gpu_available, monkeypatch):
    """HRT_CTX_ASYNC_UPDATE: hrt_tlas_update derives the instance tables on the device (k_instance_tables: object->world,
    inverse in double, identity flags, scene scale) and only enqueues the refit -- no read-back of the instance array.  The
    frames of a particle animation (shared shapes, the huge ground sphere, spheres under rotation in the mixed scene) are
    the oracle's bit for bit, i.e. the device tables equal the host's; a changed BLAS handle (a broken promise) is detected
    on the device and rebuilt at the next update."""
    if not gpu_available:
        pytest.skip("no GPU")
    import torch
    monkeypatch.setenv("HRT_REFIT_REBUILD_RATIO", "1e30")      # the quality guard is not what is tested here
    r = hrt.Renderer(0, hrt.CTX_ASYNC_UPDATE)
    try:
        w, h, salt, n_p = 72, 48, 99, 27
        scene = hrt.scenes.particle_scene(n_p, w, h, 1, frame=0)
        r.load_scene(scene)
        r.set_frame(w, h, salt, linear=True)
        states = oracle.rng_init(w, h, salt)
        ground = scene["instances"][0]["transform"]
        for frame in range(1, 5):
            poses = hrt.scenes.particle_poses(n_p, frame)
            r.update_instances([ground] + poses)
            for it, m in zip(scene["instances"][1:], poses):
                it["transform"] = m
            r.render(1)
            ref = oracle.OracleScene(scene).render(w, h, states, 1)
            assert np.array_equal(r.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32)), frame
        s = r.stats()
        assert s.tlas_refits == 4 and s.tlas_rebuilds == 1
        # spheres under a rotation + scale (the inverse matters), triangles under a shear-free affine map
        scene = hrt.scenes.mixed_test_scene(900, 20, 5, 64, 48, 1)
        r.load_scene(scene)
        moved = [it["transform"].copy() for it in scene["instances"]]
        moved[0][3] += 0.25
        moved[1] = hrt.scenes.rigid_transform((0.1, -0.2, 0.05), (0.3, 1.0, 0.2), 0.7, 1.1)
        moved[3] = hrt.scenes.rigid_transform((0.05, 0.02, -0.1), (1.0, 0.2, 0.1), 1.3, 0.8)
        moved[4] = hrt.scenes.rigid_transform((-0.1, 0.1, 0.0), (0.0, 0.0, 1.0), 0.4)
        before = r.stats()
        r.update_instances(moved)
        for it, m in zip(scene["instances"], moved):
            it["transform"] = m
        _moved_scene_matches_oracle(hrt, oracle, r, scene, 64, 48, 7)
        assert r.stats().tlas_refits == before.tlas_refits + 1
        # a broken promise: another BLAS handle in the array.  This update still refits (nothing is read back) ...
        tri = [i for i, it in enumerate(scene["instances"]) if it["geometry"] == "triangles"]
        r._h_inst[tri[0]].traversableHandle, r._h_inst[tri[1]].traversableHandle = r._h_inst[tri[1]].traversableHandle, r._h_inst[tri[0]].traversableHandle
        r._d_inst.copy_(torch.from_numpy(np.frombuffer(bytes(r._h_inst), dtype=np.uint8).copy()))
        mid = r.stats()
        r._check(r.lib.hrt_tlas_update(r.ctx, r.tlas, r._d_inst.data_ptr(), len(moved), r._stream()), "update")
        assert r.stats().tlas_rebuilds == mid.tlas_rebuilds
        # ... and the next one has the device's verdict and rebuilds with the handles as they are now
        r._check(r.lib.hrt_tlas_update(r.ctx, r.tlas, r._d_inst.data_ptr(), len(moved), r._stream()), "update")
        assert r.stats().tlas_rebuilds == mid.tlas_rebuilds + 1
        a, b = scene["instances"][tri[0]], scene["instances"][tri[1]]
        for key in ("vertices", "normals"):
            a[key], b[key] = b[key], a[key]
        # (the SBT records still point at the normals of the instances' old geometry: swap them back into place for the oracle)
        for key in ("normals",):
            a[key], b[key] = b[key], a[key]
        o, d = oracle.random_rays(20000, 31)
        t, u, v, prim, inst = r.trace_rays(o, d)
        rt, ru, rv, rprim, rinst = oracle.OracleScene(scene, force_brute=True).trace(o, d)
        assert np.array_equal(prim, rprim) and np.array_equal(inst, rinst) and np.array_equal(t.view(np.uint32), rt.view(np.uint32))
    finally:
        r.close()


def test_asynchronous_update_notices_changed_sbt_offsets(hrt, oracle, gpu_available, monkeypatch):
    """HRT_CTX_ASYNC_UPDATE and a changed sbtOffset (a hit-group record per instance, RendererMesh.cu:130-140): nothing is read
    back during the update that carries the change, but k_instance_tables compares the offsets with the build's on the device,
    and the NEXT update takes the synchronous path, re-reads them and re-derives the material tables -- no rebuild.  The frame
    after that is the oracle's on the scene with the two instances' materials (and normal arrays) exchanged."""
    if not gpu_available:
        pytest.skip("no GPU")
    import torch
    monkeypatch.setenv("HRT_REFIT_REBUILD_RATIO", "1e30")
    r = hrt.Renderer(0, hrt.CTX_ASYNC_UPDATE)
    try:
        w, h = 64, 48
        scene = hrt.scenes.mixed_test_scene(900, 20, 5, w, h, 1)
        r.load_scene(scene)
        xf = [it["transform"] for it in scene["instances"]]
        r.update_instances(xf)                                 # first update after the build: synchronous by design
        r.update_instances(xf)                                 # asynchronous from here on
        r._h_inst[0].sbtOffset, r._h_inst[2].sbtOffset = 2, 0
        r._d_inst.copy_(torch.from_numpy(np.frombuffer(bytes(r._h_inst), dtype=np.uint8).copy()))
        before = r.stats()
        r._check(r.lib.hrt_tlas_update(r.ctx, r.tlas, r._d_inst.data_ptr(), len(xf), r._stream()), "update")   # detected on the device
        r._check(r.lib.hrt_tlas_update(r.ctx, r.tlas, r._d_inst.data_ptr(), len(xf), r._stream()), "update")   # acted on
        after = r.stats()
        assert after.tlas_rebuilds == before.tlas_rebuilds and after.tlas_refits == before.tlas_refits + 2
        a, b = scene["instances"][0], scene["instances"][2]
        for key in ("material", "albedo", "fuzz", "normals"):
            a[key], b[key] = b[key], a[key]
        _moved_scene_matches_oracle(hrt, oracle, r, scene, w, h, 5)
        # and the update after that is asynchronous again (the device's copy of the offsets was refreshed)
        r._check(r.lib.hrt_tlas_update(r.ctx, r.tlas, r._d_inst.data_ptr(), len(xf), r._stream()), "update")
        r._check(r.lib.hrt_tlas_update(r.ctx, r.tlas, r._d_inst.data_ptr(), len(xf), r._stream()), "update")
        _moved_scene_matches_oracle(hrt, oracle, r, scene, w, h, 6)
        assert r.stats().tlas_rebuilds == before.tlas_rebuilds
    finally:
        r.close()


def test_pose_instances_matches_oracle(hrt, oracle, renderer):
    """hrt_pose_instances (slerp -> quatToEuler -> constructTransformMatrix on the device, RendererTime.cu:436-472)
    against the oracle: EVERY entry of every transform bit-exact.  sinf / cosf / acosf / asinf / atan2f are pinned on both
    sides as the correctly rounded float of the exact value (csrc/cr_trig.h; oracle: libm double + __float128), the rest
    is float arithmetic in the reference's order."""
    n = 300
    scene = hrt.scenes.particle_scene(n, 32, 32, 1, subdiv=0)
    renderer.load_scene(scene)
    cur, nxt = hrt.scenes.particle_states(n, 0), hrt.scenes.particle_states(n, 1)
    s = np.float32(np.sqrt(0.5))
    nxt[0, :4] = cur[0, :4]                                   # identical quaternions: normalised-lerp branch
    nxt[1, :4] = cur[1, :4] + np.float32([1e-3, 0, -1e-3, 0])  # nearly identical
    nxt[2, :4] = -cur[2, :4]                                  # dot < 0: negated
    cur[3, :4] = nxt[3, :4] = np.float32([s, 0, s, 0]) * np.float32(1.0000002)   # |sinp| >= 1 after the placement shuffle
    cur[4, :4] = nxt[4, :4] = [0, 0, 1, 0]                    # identity as quatToEuler sees it
    for dur, frame, count, off, sc in ((0.5, 0, 120, (0, 0, 0), (1, 1, 1)), (0.5, 59, 120, (0, 0, 0), (1, 1, 1)),
                                       (0.5, 119, 120, (0.1, 0.2, -0.3), (1.5, 0.5, 2.0)), (2.0, 0, 1, (0, 0, 0), (1, 1, 1))):
        renderer.pose_instances(cur, nxt, dur, frame, count, first_instance=1, offset=off, scale=sc, update=False)
        got = renderer.instance_transforms()
        want = oracle.pose_transforms(cur, nxt, dur, frame, count, off, sc)
        assert np.array_equal(got[0], scene["instances"][0]["transform"])           # the extra geometry is not touched
        assert np.array_equal(got[1:].view(np.uint32), want.view(np.uint32)), (dur, frame, count)
    # many more particles, arbitrary unit quaternions
    rng = np.random.default_rng(17)
    n2 = 20000
    scene = hrt.scenes.particle_scene(n2, 32, 32, 1, subdiv=0)
    renderer.load_scene(scene)
    cur, nxt = hrt.scenes.particle_states(n2, 0), hrt.scenes.particle_states(n2, 1)
    for st in (cur, nxt):
        q = rng.normal(size=(n2, 4))
        st[:, :4] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    for frame in (0, 7, 29):
        renderer.pose_instances(cur, nxt, 0.25, frame, 30, first_instance=1, update=False)
        want = oracle.pose_transforms(cur, nxt, 0.25, frame, 30)
        assert np.array_equal(renderer.instance_transforms()[1:].view(np.uint32), want.view(np.uint32)), frame


def _gpu_trig(hrt, r, which, a=None, b=None, first=0, stride=1, count=0, slow=False):
    import torch
    if a is not None:
        ta = torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(r.device)
        tb = torch.from_numpy(np.ascontiguousarray(b, np.float32)).to(r.device) if b is not None else None
        count = ta.numel()
    else:
        ta = tb = None
    out = torch.empty(count, dtype=torch.float32, device=r.device)
    r._check(r.lib.hrt_debug_trig(r.ctx, which, ta.data_ptr() if ta is not None else None, tb.data_ptr() if tb is not None else None,
                                  first, stride, count, int(slow), out.data_ptr(), r._stream()), "hrt_debug_trig")
    return out.cpu().numpy()


def test_trig_pin_every_float_on_the_gpu(hrt, oracle, gpu_available):
    """The pose kernel's sinf / cosf over ALL 2^32 float bit patterns, acosf / asinf over every float in [-1, 1] (and a strided
    sweep of the rest: NaN), atan2f over 20 M pairs: the device's values (ocml double + the double-double slow path of
    csrc/cr_trig.h) are the oracle's (libm double + __float128) bit for bit.  Then strided sweeps with every value forced
    through the slow path."""
    if not gpu_available:
        pytest.skip("no GPU")
    from test_pose_cpu import atan2_test_pairs, _same_bits
    r = hrt.Renderer(0, 0)
    try:
        chunk = 1 << 26
        ref = np.empty(chunk, np.float32)
        for which in (oracle.TRIG_SIN, oracle.TRIG_COS):
            for first in range(0, 1 << 32, chunk):
                got = _gpu_trig(hrt, r, which, first=first, count=chunk)
                assert _same_bits(got, oracle.trig_bits(which, first, 1, chunk, out=ref)), (which, hex(first))
        for which in (oracle.TRIG_ACOS, oracle.TRIG_ASIN):
            for sign in (0, 0x80000000):
                for first in range(0, 0x3F800000 + 1, chunk):
                    n = min(chunk, 0x3F800000 + 1 - first)
                    got = _gpu_trig(hrt, r, which, first=sign + first, count=n)
                    assert _same_bits(got, oracle.trig_bits(which, sign + first, 1, n, out=ref)), (which, hex(sign + first))
            got = _gpu_trig(hrt, r, which, first=0, stride=1021, count=(1 << 32) // 1021)
            assert _same_bits(got, oracle.trig_bits(which, 0, 1021, (1 << 32) // 1021)), which
        y, x = atan2_test_pairs(20_000_000, 11)
        assert _same_bits(_gpu_trig(hrt, r, oracle.TRIG_ATAN2, y, x), oracle.trig(oracle.TRIG_ATAN2, y, x))
        # every value through the double-double path
        for which in (oracle.TRIG_SIN, oracle.TRIG_COS):
            got = _gpu_trig(hrt, r, which, first=0, stride=61, count=(1 << 32) // 61, slow=True)
            assert _same_bits(got, oracle.trig_bits(which, 0, 61, (1 << 32) // 61)), which
        for which in (oracle.TRIG_ACOS, oracle.TRIG_ASIN):
            for sign in (0, 0x80000000):
                n = 0x3F800000 // 17 + 1
                got = _gpu_trig(hrt, r, which, first=sign, stride=17, count=n, slow=True)
                assert _same_bits(got, oracle.trig_bits(which, sign, 17, n)), (which, sign)
        y, x = atan2_test_pairs(4_000_000, 12)
        assert _same_bits(_gpu_trig(hrt, r, oracle.TRIG_ATAN2, y, x, slow=True), oracle.trig(oracle.TRIG_ATAN2, y, x))
    finally:
        r.close()


def _mesh_mode_transforms(oracle, velocities, dur, frame, count, off, sc):
    """Mesh mode's update on the oracle side (RendererMesh.cu:379-391): shift = offset + (velocity * duration / frames) * frame,
    rotation (0, 0, 0), through the oracle's constructTransformMatrix."""
    f = np.float32
    out = np.zeros((len(velocities), 12), np.float32)
    for i, vel in enumerate(np.asarray(velocities, np.float32)):
        per_frame = ((vel * f(dur)).astype(f) / f(count)).astype(f)
        shift = (np.asarray(off, f) + (per_frame * f(frame)).astype(f)).astype(f)
        out[i] = oracle.construct_transform(shift, (0, 0, 0), sc)
    return out


def test_pose_instances_mesh_mode(hrt, oracle, renderer):
    """Mesh mode's per-frame update (RendererMesh.cu:379-391): every particle drifts by velocity * duration / frames per
    frame, no rotation, the particle's position is not added (its geometry is already posed).  Bit-exact against
    constructTransformMatrix(offset + shift * frame, {0,0,0}, scale) of the oracle."""
    n = 64
    scene = hrt.scenes.particle_scene(n, 32, 32, 1, subdiv=0)
    renderer.load_scene(scene)
    st = hrt.scenes.particle_states(n, 0)
    st[:, 7:10] = np.random.default_rng(3).uniform(-2, 2, (n, 3)).astype(np.float32)
    f = np.float32
    for dur, frame, count, off, sc in ((0.01, 0, 9, (0, 0, 0), (1, 1, 1)), (0.01, 8, 9, (0.5, -1, 2), (1, 1, 1)), (0.25, 77, 120, (0, 0, 0), (2, 0.5, 1.5))):
        renderer.pose_instances(st, st, dur, frame, count, first_instance=1, offset=off, scale=sc, update=False, mesh_mode=True)
        got = renderer.instance_transforms()[1:]
        for i in range(n):
            per_frame = ((st[i, 7:10] * f(dur)).astype(f) / f(count)).astype(f)
            shift = (np.asarray(off, f) + (per_frame * f(frame)).astype(f)).astype(f)
            want = oracle.construct_transform(shift, (0, 0, 0), sc)
            assert np.array_equal(got[i].view(np.uint32), want.view(np.uint32)), (i, frame)


def test_time_mode_frames_pose_refit_render(hrt, oracle, renderer):
    """The whole Time-mode frame on the device: pose kernel -> updateIAS (refit) -> launch, several frames across two
    time steps; every frame's image is bit-exact against the oracle rendering the scene with the transforms the
    pose kernel produced (read back), and the RNG streams carry over (Q8)."""
    w, h, salt, n = 72, 48, 777, 27
    scene = hrt.scenes.particle_scene(n, w, h, 1)
    renderer.load_scene(scene)
    renderer.set_frame(w, h, salt, linear=True)
    states = oracle.rng_init(w, h, salt)
    steps = [hrt.scenes.particle_states(n, k) for k in range(3)]
    for step in range(2):
        for frame in (0, 3, 5):
            renderer.pose_instances(steps[step], steps[step + 1], 0.05, frame, 6, first_instance=1)
            renderer.render(1)
            want = oracle.pose_transforms(steps[step], steps[step + 1], 0.05, frame, 6)
            assert np.array_equal(renderer.instance_transforms()[1:].view(np.uint32), want.view(np.uint32))
            for it, m in zip(scene["instances"][1:], want):
                it["transform"] = m.copy()
            ref = oracle.OracleScene(scene).render(w, h, states, 1)
            assert np.array_equal(renderer.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32)), (step, frame)
    assert np.array_equal(renderer.rng_states_numpy(), states)
    assert renderer.stats().tlas_refits == 6


def test_real_data_time_mode_frames(hrt, oracle, renderer):
    """The reference's shipped sample (tests/golden/files: config.json, STL shapes, particle VTK steps, series) through
    the whole chain: readers -> scene as RendererTime::commitRendererData assembles it -> per frame pose kernel,
    updateIAS (refit), launch.  Window reduced to 300x200 (config: 1200x800) so the scalar oracle finishes in seconds;
    every frame bit-exact against the oracle rendering with its OWN transforms (oracle.pose_transforms), which the pose
    kernel's equal bit for bit."""
    import importlib
    from pathlib import Path
    io = importlib.import_module("nvidia-optix-ray-tracer_amd.io")
    tm = io.time_mode_scene(Path(__file__).resolve().parent / "golden" / "files" / "config.json", width=300, height=200)
    scene, cfg = tm["scene"], tm["config"]
    w, h, salt = 300, 200, hrt.scenes.SEED_SALT
    renderer.load_scene(scene)
    renderer.set_frame(w, h, salt, linear=True)
    states = oracle.rng_init(w, h, salt)
    hits = 0
    for file_index, frame in ((0, 0), (0, 8), (1, 4)):
        cur = tm["states"][file_index]
        nxt = tm["states"][min(file_index + 1, len(tm["states"]) - 1)]            # RendererTime.cu:443-447
        renderer.pose_instances(cur, nxt, float(tm["durations"][file_index]), frame, tm["frame_counts"][file_index],
                                first_instance=tm["n_extra"], offset=cfg["particle-shift"], scale=cfg["particle-scale"])
        renderer.render(1)
        xf = renderer.instance_transforms()
        want = oracle.pose_transforms(cur, nxt, float(tm["durations"][file_index]), frame, tm["frame_counts"][file_index],
                                      cfg["particle-shift"], cfg["particle-scale"])
        n_extra = tm["n_extra"]
        assert np.array_equal(xf[n_extra:].view(np.uint32), want.view(np.uint32))
        assert np.array_equal(xf[0], scene["instances"][0]["transform"])
        for it, m in zip(scene["instances"][n_extra:], want):
            it["transform"] = m.copy()
        ref = oracle.OracleScene(scene).render(w, h, states, 1)
        assert np.array_equal(renderer.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32)), (file_index, frame)
        hits += int(ref["rays"]) - w * h
    assert hits > w * h // 2                                   # the ground sphere fills the lower half of the frame
    assert np.array_equal(renderer.rng_states_numpy(), states)


def test_mesh_mode_frames_and_cpp_driver(hrt, oracle, gpu_available, tmp_path):
    """Mesh mode end to end (the reference's second program mode, src/Global/RendererMesh.cu; its launch site :416-419 is the one
    the boundary replaces): a synthetic data set in the reference's on-disk formats (series, particleN.cache, metadata.cache,
    config.json) -> io.mesh_mode_scene assembles what commitRendererData does -> per frame the Mesh-mode pose kernel, updateIAS,
    launch.  Every frame of every file bit-exact against the oracle; then the C++ driver hrt_mesh_render (loader threads, device
    GAS / IAS builds, the same frame loop) plays the same data and its last frame's bytes are the oracle's."""
    import importlib
    import subprocess
    from pathlib import Path
    if not gpu_available:
        pytest.skip("no GPU")
    io = importlib.import_module("nvidia-optix-ray-tracer_amd.io")
    w, h, salt = 160, 120, hrt.scenes.SEED_SALT
    cfg_path = io.write_mesh_mode_sample(tmp_path, n_files=3, n_particles=12, width=w, height=h)
    mm = io.mesh_mode_scene(cfg_path)
    cfg, n_extra = mm["config"], mm["n_extra"]
    assert len(mm["scenes"]) == 3 and mm["frame_counts"] == [3, 6, 6]
    r = hrt.Renderer(0, 0)
    try:
        states = None
        for k, scene in enumerate(mm["scenes"]):
            r.load_scene(scene)
            if states is None:
                r.set_frame(w, h, salt, linear=True)
                states = oracle.rng_init(w, h, salt)
            st = np.zeros((len(mm["velocities"][k]), 12), np.float32)
            st[:, 7:10] = mm["velocities"][k]
            for frame in range(mm["frame_counts"][k]):
                r.pose_instances(st, st, float(mm["durations"][k]), frame, mm["frame_counts"][k], first_instance=n_extra,
                                 offset=cfg["particle-shift"], scale=cfg["particle-scale"], mesh_mode=True)
                r.render(1)
                xf = r.instance_transforms()
                assert np.array_equal(xf[0], scene["instances"][0]["transform"])
                want = _mesh_mode_transforms(oracle, mm["velocities"][k], float(mm["durations"][k]), frame, mm["frame_counts"][k],
                                             cfg["particle-shift"], cfg["particle-scale"])
                assert np.array_equal(xf[n_extra:].view(np.uint32), want.view(np.uint32))
                for it, m in zip(scene["instances"][n_extra:], want):      # the oracle renders from its own transforms
                    it["transform"] = m.copy()
                ref = oracle.OracleScene(scene).render(w, h, states, 1)
                assert np.array_equal(r.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32)), (k, frame)
                assert ref["rays"] > w * h
        assert np.array_equal(r.rng_states_numpy(), states)
        want = np.zeros((h, w, 4), np.uint8)
        oracle.lib().oracle_to_rgba8(np.ascontiguousarray(ref["color"]).ctypes.data, want.ctypes.data, w, h)
    finally:
        r.close()
    exe = Path(__file__).resolve().parent.parent / "nvidia-optix-ray-tracer_amd" / "lib" / "hrt_mesh_render"
    assert exe.exists(), "run `make tools`"
    out = tmp_path / "last.ppm"
    p = subprocess.run([str(exe), str(cfg_path), str(tmp_path / "bin"), "-1", str(out)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "15 frames 160x120" in p.stdout and "3 files loaded" in p.stdout, p.stdout
    header = b"P6\n160 120\n255\n"
    raw = out.read_bytes()
    assert raw.startswith(header)
    got = np.frombuffer(raw[len(header):], np.uint8).reshape(h, w, 3)
    assert np.array_equal(got, want[..., :3])


def test_tree_over_instances_parity(hrt, oracle, gpu_available, monkeypatch):
    """HRT_TLAS_INSTANCED=1: hrt_tlas_build makes a top tree over the instances whose leaves are per-instance copies of
    object-space template trees, and the device refit fills in every box and world-space record.  Same bits as the
    oracle: rendered images, hit records against brute force, an animation refitted every frame; an update with
    unchanged transforms rewrites the very bytes the build produced; BLASes shared by many instances."""
    if not gpu_available:
        pytest.skip("no GPU")
    monkeypatch.setenv("HRT_TLAS_INSTANCED", "1")
    r = hrt.Renderer(0, 0)
    try:
        for scene, (w, h, spp) in ((hrt.scenes.mixed_test_scene(2500, 50, 13, 120, 80, 2), (120, 80, 2)),
                                   (hrt.scenes.cornell_box(96, 96, 2), (96, 96, 2)),
                                   (hrt.scenes.particle_scene(40, 100, 70, 1, frame=2), (100, 70, 1))):
            r.load_scene(scene)
            _moved_scene_matches_oracle(hrt, oracle, r, scene, w, h, 31, spp)
        # refit of a tree over instances: identical transforms -> identical bytes; moved -> oracle on the moved scene
        n_p = 40
        scene = hrt.scenes.particle_scene(n_p, 72, 48, 1, frame=0)
        r.load_scene(scene)
        nodes0, prims0 = _download_tree(hrt, r)
        before = r.stats()
        r.update_instances([it["transform"] for it in scene["instances"]])
        nodes1, prims1 = _download_tree(hrt, r)
        assert np.array_equal(nodes0, nodes1) and np.array_equal(prims0, prims1)
        ground = scene["instances"][0]["transform"]
        r.set_frame(72, 48, 5, linear=True)
        states = oracle.rng_init(72, 48, 5)
        for frame in (1, 2, 5):
            poses = hrt.scenes.particle_poses(n_p, frame)
            r.update_instances([ground] + poses)
            for it, m in zip(scene["instances"][1:], poses):
                it["transform"] = m
            r.render(1)
            ref = oracle.OracleScene(scene).render(72, 48, states, 1)
            assert np.array_equal(r.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32)), frame
        after = r.stats()
        assert after.tlas_refits == before.tlas_refits + 4 and after.tlas_rebuilds == before.tlas_rebuilds
    finally:
        r.close()


def test_rebuild_during_animation_uses_the_tree_over_instances(hrt, oracle, gpu_available, monkeypatch):
    """The reference builds each file's IAS with identity transforms and poses it afterwards (RendererTime.cu:111-127):
    all particles start on top of each other.  The first update sees that most instances are further from where the tree
    was built than they are wide and rebuilds at once -- as a tree over instances (milliseconds), without refitting the
    useless tree first; with that check off (HRT_REFIT_MOVED_FAR=0) the refit is done, checked on the spot, and the tree
    rebuilt within the same update.  Either way no frame is traced through the degraded tree.  Image parity after the
    rebuild and after a further refit."""
    if not gpu_available:
        pytest.skip("no GPU")
    for moved_far_check in (True, False):
        monkeypatch.setenv("HRT_REFIT_MOVED_FAR", "1" if moved_far_check else "0")
        r = hrt.Renderer(0, 0)
        try:
            n_p, w, h = 30, 72, 48
            scene = hrt.scenes.particle_scene(n_p, w, h, 1, frame=0)
            posed = [it["transform"].copy() for it in scene["instances"]]
            for it in scene["instances"][1:]:
                it["transform"] = hrt.scenes.IDENTITY.copy()
            r.load_scene(scene)                                   # merged build over the overlapping particles
            r.update_instances(posed)                             # rebuilt: at once, or after a refit whose boxes explode
            s = r.stats()
            if moved_far_check:
                assert s.tlas_refits == 0 and s.tlas_rebuilds == 2
            else:
                assert s.tlas_refits == 1 and s.tlas_rebuilds == 2 and s.tlas_refit_ratio > 1.5
            for it, m in zip(scene["instances"], posed):
                it["transform"] = m
            _moved_scene_matches_oracle(hrt, oracle, r, scene, w, h, 11)
            r.update_instances(posed)                             # and the rebuilt tree refits
            s = r.stats()
            assert s.tlas_refits == (1 if moved_far_check else 2) and s.tlas_rebuilds == 2 and s.tlas_refit_ratio < 1.01
            _moved_scene_matches_oracle(hrt, oracle, r, scene, w, h, 12)
        finally:
            r.close()


def test_refit_quality_guard_rebuilds(hrt, oracle, monkeypatch):
    """A refit keeps the topology; when the instances have moved so far that the boxes' area sum passes
    HRT_REFIT_REBUILD_RATIO x the built tree's, the next update rebuilds -- and the image is still the oracle's."""
    monkeypatch.setenv("HRT_REFIT_REBUILD_RATIO", "1.2")
    r = hrt.Renderer(0, 0)
    try:
        w, h = 64, 48
        scene = hrt.scenes.particle_scene(12, w, h, 1, frame=0)
        r.load_scene(scene)
        ground = scene["instances"][0]["transform"]
        far = [hrt.scenes.rigid_transform((0.9 * np.cos(i), 0.9 * np.sin(i), 0.3 + 0.05 * i), (0, 0, 1), 0.3 * i) for i in range(12)]
        near = hrt.scenes.particle_poses(12, 1)
        r.update_instances([ground] + near)          # first refit after the build: checked on the spot, fine
        r.update_instances([ground] + far)           # refit, boxes grow a lot (checked at the next update)
        s = r.stats()
        assert s.tlas_refits == 2 and s.tlas_rebuilds == 1, s.tlas_refit_ratio
        r.update_instances([ground] + far)           # sees the degraded tree -> rebuild
        s = r.stats()
        assert s.tlas_refits == 2 and s.tlas_rebuilds == 2 and s.tlas_refit_ratio > 1.2, s.tlas_refit_ratio
        for it, m in zip(scene["instances"][1:], far):
            it["transform"] = m
        _moved_scene_matches_oracle(hrt, oracle, r, scene, w, h, 7)
    finally:
        r.close()


def test_full_size_properties_1080p(hrt, renderer):
    """BASELINE full size (1920x1080, 100k triangles): size-independent properties only.
    determinism, 1..5 rays per path, tile idempotence, finite output, alpha = 1."""
    scene = hrt.scenes.soup_100k(1920, 1080, 2)
    renderer.load_scene(scene)
    renderer.set_frame(1920, 1080, hrt.scenes.SEED_SALT, aov=False)
    renderer.reset_stats()
    renderer.render(2)
    a = renderer.color.cpu().numpy().copy()
    s = renderer.stats()
    assert s.paths == 1920 * 1080 * 2
    assert s.paths <= s.rays <= 5 * s.paths
    assert s.rays_any <= s.paths
    assert np.isfinite(a).all() and (a[..., 3] == 1).all() and a.min() >= 0 and a.max() <= 1
    renderer.set_frame(1920, 1080, hrt.scenes.SEED_SALT, aov=False)
    renderer.render(2)
    assert np.array_equal(a, renderer.color.cpu().numpy()), "same seed, same image"
    # stripes of 8 'ranks' reassemble the frame
    renderer.set_frame(1920, 1080, hrt.scenes.SEED_SALT, aov=False)
    renderer.color.zero_()
    for rank in range(8):
        renderer.render(2, tile=hrt.tile_for_rank(1080, rank, 8), sync=False)
    import torch
    torch.cuda.synchronize()
    assert np.array_equal(a, renderer.color.cpu().numpy())


def test_full_size_c3_frame_against_the_oracle(hrt, oracle, gpu_available):
    """BASELINE configs[2] geometry at its full size (100 000 triangles, 1920x1080), one sample per pixel, production
    kernels: every one of the 2 073 600 pixels' linear radiance, the final RNG states and the ray count are bit-exact
    against the oracle (a few seconds of CPU on the box's host threads)."""
    if not gpu_available:
        pytest.skip("no GPU")
    r = hrt.Renderer(0, 0)
    try:
        w, h = 1920, 1080
        scene = hrt.scenes.soup_100k(w, h, 1)
        r.load_scene(scene)
        r.set_frame(w, h, hrt.scenes.SEED_SALT, aov=False, linear=True)
        r.render(1)
        states = oracle.rng_init(w, h, hrt.scenes.SEED_SALT)
        ref = oracle.OracleScene(scene).render(w, h, states, 1)
        assert np.array_equal(r.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32))
        assert np.array_equal(r.color.cpu().numpy().view(np.uint32), ref["color"].view(np.uint32))
        assert np.array_equal(r.rng_states_numpy(), states)
        assert r.stats().rays == ref["rays"]
    finally:
        r.close()


@pytest.mark.parametrize("config", ["C4", "C5"])
def test_full_size_c4_c5_frames_against_the_oracle(hrt, oracle, gpu_available, config):
    """BASELINE configs[3] and configs[4] as SURVEY 8(d) defines them -- 1 M triangles at 1920x1080; C5 with its real 8
    materials (4 rough + 4 metal with fuzz 0 / 0.1 / 0.3 / 0.5, one instance each) -- one sample per pixel, production
    kernels: linear radiance, sRGB colour, final RNG states and ray count of all 2 073 600 pixels bit-exact against the
    oracle."""
    if not gpu_available:
        pytest.skip("no GPU")
    r = hrt.Renderer(0, 0)
    try:
        w, h = 1920, 1080
        scene = hrt.scenes.soup_1m(w, h, 1) if config == "C4" else hrt.scenes.soup_1m_8mat(w, h, 1)
        if config == "C5":
            mats = [(it["material"], float(it["fuzz"])) for it in scene["instances"]]
            assert len(mats) == 8 and sum(m == "rough" for m, _ in mats) == 4
            assert sorted(f for m, f in mats if m == "metal") == pytest.approx([0.0, 0.1, 0.3, 0.5])
        r.load_scene(scene)
        r.set_frame(w, h, hrt.scenes.SEED_SALT, aov=False, linear=True)
        r.reset_stats()
        r.render(1)
        states = oracle.rng_init(w, h, hrt.scenes.SEED_SALT)
        ref = oracle.OracleScene(scene).render(w, h, states, 1)
        assert np.array_equal(r.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32))
        assert np.array_equal(r.color.cpu().numpy().view(np.uint32), ref["color"].view(np.uint32))
        assert np.array_equal(r.rng_states_numpy(), states)
        assert r.stats().rays == ref["rays"]
    finally:
        r.close()


def test_full_size_c4_c5_properties(hrt, gpu_available):
    """BASELINE configs[3] / [4] geometry at full size (1 M triangles, 1920x1080; spp reduced to 2): size-independent
    properties with the production kernels -- determinism, 1..5 rays per path, at most one any-hit ray per path, the
    union of 4 stripe tiles is the full frame bit for bit, and the 8-instance split of the same triangles (C5) with one
    material gives the very image of the single instance (C4): the hit and everything after it do not depend on how
    the triangles are grouped into instances."""
    if not gpu_available:
        pytest.skip("no GPU")
    import torch
    r = hrt.Renderer(0, 0)
    try:
        w, h, spp = 1920, 1080, 2
        c4 = hrt.scenes.soup_1m(w, h, spp)
        r.load_scene(c4)
        r.set_frame(w, h, hrt.scenes.SEED_SALT, aov=False)
        r.render(spp)
        full = r.color.cpu().numpy().copy()
        s = r.stats()
        assert s.paths == w * h * spp and s.paths <= s.rays <= 5 * s.paths and s.rays_any <= s.paths
        assert s.bvh_triangles == 1_000_000
        assert np.isfinite(full).all() and (full[..., 3] == 1).all() and 0 <= full.min() and full.max() <= 1
        r.set_frame(w, h, hrt.scenes.SEED_SALT, aov=False)
        r.color.zero_()
        for rank in range(4):
            r.render(spp, tile=hrt.tile_for_rank(h, rank, 4), sync=False)
        torch.cuda.synchronize()
        assert np.array_equal(full, r.color.cpu().numpy())
        # C5's instance split with C4's single material
        c5 = hrt.scenes.soup_1m_8mat(w, h, spp)
        assert len(c5["instances"]) == 8
        for it in c5["instances"]:
            it["material"], it["albedo"], it["fuzz"] = c4["instances"][0]["material"], c4["instances"][0]["albedo"], 0.0
        r.load_scene(c5)
        r.set_frame(w, h, hrt.scenes.SEED_SALT, aov=False)
        r.reset_stats()
        r.render(spp)
        assert np.array_equal(full, r.color.cpu().numpy())
        assert r.stats().rays == s.rays
    finally:
        r.close()


def test_device_build_degenerate_geometry(hrt, oracle, gpu_available, monkeypatch):
    """The device build (PLOC, build.hip) on geometry that stresses it: NaN / Inf vertices (such primitives are left out of
    the tree and never hit), zero-area and duplicated triangles (coinciding Morton codes, equal boxes), one huge triangle
    among tiny ones, a sphere BLAS under a rotation -- hit records bit-exact against the brute-force oracle, and the image
    too.  The same scene through the builders with spatial splits (HRT_CTX_FAST_TRACE: the device's top-down phase, which has to
    leave the 41 coinciding triangles to PLOC as one cell and cuts the huge one many times; the host's) gives the same bits: the
    result does not depend on the tree."""
    if not gpu_available:
        pytest.skip("no GPU")
    rng = np.random.default_rng(11)
    scene = hrt.scenes.mixed_test_scene(6000, 30, 21, 96, 64, 2)         # (above 4096 primitives, so that HRT_CTX_FAST_TRACE does split on the device)
    tri = [it for it in scene["instances"] if it["geometry"] == "triangles"][0]
    v = tri["vertices"].reshape(-1, 9).copy()
    v[5, 3] = np.nan; v[17, 0] = np.inf; v[40, 8] = -np.inf               # non-finite primitives
    v[60] = np.tile(v[60, :3], 3)                                         # a point
    v[61, 3:6] = v[61, :3]                                                # a segment
    v[100:140] = v[99]                                                    # 41 copies of one triangle
    v[200] = [-5, -5, 0.2, 5, -5, 0.2, 0, 7, 0.2]                         # one huge triangle
    tri["vertices"] = v.reshape(tri["vertices"].shape)
    images = []
    for flags, builder in ((0, "device"), (hrt.CTX_FAST_TRACE, "device"), (hrt.CTX_FAST_TRACE, "host")):
        monkeypatch.setenv("HRT_FAST_TRACE_BUILD", builder)
        r = hrt.Renderer(0, flags)
        try:
            r.load_scene(scene)
            o, d = oracle.random_rays(30000, 3)
            t, u, vv, prim, inst = r.trace_rays(o, d)
            rt, ru, rv, rprim, rinst = oracle.OracleScene(scene, force_brute=True).trace(o, d)
            assert np.array_equal(prim, rprim) and np.array_equal(inst, rinst)
            assert np.array_equal(t.view(np.uint32), rt.view(np.uint32)) and np.array_equal(u.view(np.uint32), ru.view(np.uint32))
            r.set_frame(96, 64, 5, linear=True)
            r.render(2)
            images.append(r.linear.cpu().numpy().copy())
            s = r.stats()
            assert s.bvh_triangles + s.bvh_spheres > 0
        finally:
            r.close()
    ref = oracle.OracleScene(scene, force_brute=True).render(96, 64, oracle.rng_init(96, 64, 5), 2)      # (the oracle's own BVH is not meant for NaNs)
    assert np.array_equal(images[0].view(np.uint32), ref["linear"].view(np.uint32))
    assert np.array_equal(images[0].view(np.uint32), images[1].view(np.uint32))
    assert np.array_equal(images[0].view(np.uint32), images[2].view(np.uint32))


def test_device_build_is_fast_and_keeps_the_geometry_on_the_device(hrt, gpu_available):
    """1 M triangles: hrt_blas_build_triangles + hrt_tlas_build on the device -- no host copy of the vertices, a tree of
    BVH8 quality (a seventh as many nodes as triangles), well under a tenth of a second here (measured: ~16 ms for the two
    calls, profiles/r02_build_bench.txt; the bar is loose because the box is shared)."""
    if not gpu_available:
        pytest.skip("no GPU")
    import time
    import torch
    r = hrt.Renderer(0, 0)
    try:
        scene = hrt.scenes.soup_1m(64, 64, 1)
        r.load_scene(scene)                                  # warm-up (allocator, code objects)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.load_scene(scene)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        s = r.stats()
        r.set_frame(64, 64, 1, aov=False); r.render(1); s = r.stats()
        assert s.bvh_triangles == 1_000_000 and 100_000 < s.bvh_nodes < 250_000
        assert dt < 0.25, dt
    finally:
        r.close()


def test_concurrent_blas_builds(hrt, oracle, renderer):
    """The reference builds its GASes from several host threads, one stream each (RendererMesh.cu:98-100, 205-219):
    hrt_blas_build_* must be safe to call concurrently on one context."""
    import ctypes as C
    import threading
    import torch
    scene = hrt.scenes.mixed_test_scene(1200, 24, 4, 64, 48, 1)
    tri = [it for it in scene["instances"] if it["geometry"] == "triangles"]
    handles = [[None] * 8 for _ in tri]
    verts = [torch.from_numpy(np.ascontiguousarray(it["vertices"].reshape(-1, 3))).cuda() for it in tri]
    streams = [torch.cuda.Stream() for _ in range(8)]
    torch.cuda.synchronize()

    def work(k):
        for j, v in enumerate(verts):
            hdl = C.c_uint64()
            rc = renderer.lib.hrt_blas_build_triangles(renderer.ctx, v.data_ptr(), v.shape[0], C.c_void_p(streams[k].cuda_stream), C.byref(hdl))
            handles[j][k] = (rc, hdl.value)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    flat = [x for row in handles for x in row]
    assert all(rc == 0 for rc, _ in flat)
    assert len({hd for _, hd in flat}) == len(flat), "every build returns its own handle"
    # any of the concurrently built handles renders like a sequential build
    renderer.load_scene(scene)
    renderer.set_frame(64, 48, 5, linear=True)
    renderer.render(1)
    want = renderer.linear.cpu().numpy().copy()
    j = 0
    for i, it in enumerate(scene["instances"]):
        if it["geometry"] == "triangles":
            renderer._h_inst[i].traversableHandle = handles[j][(3 * j + 1) % 8][1]
            j += 1
    renderer._d_inst.copy_(torch.from_numpy(np.frombuffer(bytes(renderer._h_inst), dtype=np.uint8).copy()))
    renderer._check(renderer.lib.hrt_tlas_update(renderer.ctx, renderer.tlas, renderer._d_inst.data_ptr(), len(scene["instances"]), renderer._stream()), "update")
    assert renderer.stats().tlas_rebuilds == 2              # changed handles: not a refit
    renderer.set_frame(64, 48, 5, linear=True)
    renderer.render(1)
    assert np.array_equal(want.view(np.uint32), renderer.linear.cpu().numpy().view(np.uint32))


def test_errors_are_loud(hrt, renderer):
    import ctypes as C
    lib = renderer.lib
    rg = hrt.RayGenParams()
    gp = hrt.GlobalParams(0xdead, None)
    assert lib.hrt_render_launch(renderer.ctx, C.byref(gp), C.byref(rg), 1, None, None) < 0
    assert len(lib.hrt_last_error(renderer.ctx)) > 0
    bad = C.c_uint64()
    assert lib.hrt_blas_build_triangles(renderer.ctx, None, 4, None, C.byref(bad)) == -1
    assert lib.hrt_blas_build_triangles(renderer.ctx, None, 3, None, C.byref(bad)) < 0          # NULL vertices
    assert lib.hrt_tlas_update(renderer.ctx, 0xbeef, None, 0, None) < 0 and b"TLAS" in lib.hrt_last_error(renderer.ctx)
    assert lib.hrt_blas_destroy(renderer.ctx, 0xbeef) < 0
    # an instance that names a handle which is not a BLAS
    inst = hrt.Instance()
    inst.traversableHandle = 0x7777
    inst.visibilityMask = 1
    import torch
    d = torch.from_numpy(np.frombuffer(bytes(inst), dtype=np.uint8).copy()).cuda()
    tl = C.c_uint64()
    assert lib.hrt_tlas_build(renderer.ctx, d.data_ptr(), 1, None, C.byref(tl)) < 0 and b"BLAS" in lib.hrt_last_error(renderer.ctx)
    # update must keep the instance count
    scene = hrt.scenes.cornell_box(16, 16, 1)
    renderer.load_scene(scene)
    assert lib.hrt_tlas_update(renderer.ctx, renderer.tlas, renderer._d_inst.data_ptr(), 99, None) < 0
    # pose: zero frame count, misaligned arrays
    pp = hrt.PoseParams(0.5, 0, 0, (C.c_float * 3)(0, 0, 0), (C.c_float * 3)(1, 1, 1))
    assert lib.hrt_pose_instances(renderer.ctx, renderer._d_inst.data_ptr(), 0, 1, d.data_ptr(), d.data_ptr(), C.byref(pp), None) < 0
    pp.frame_count = 4
    assert lib.hrt_pose_instances(renderer.ctx, renderer._d_inst.data_ptr() + 4, 0, 1, d.data_ptr(), d.data_ptr(), C.byref(pp), None) < 0


MODES = {
    "default": {},
    "wavefront": {"HRT_FUSED": "0"},
    "wavefront-samples-replayed-from-a-graph": {"HRT_FUSED": "0", "HRT_WAVEFRONT_GRAPH": "1"},
    "wavefront-round-1-traverse-kernel": {"HRT_FUSED": "0", "HRT_WAVEFRONT_LEAN": "0"},
    "wavefront-by-tile-size": {"HRT_FUSED": "-1", "HRT_FUSED_MAX_PIXELS": "16000"},
    "wavefront-lds-dma-gather": {"HRT_FUSED": "0", "HRT_LDS_GATHER": "1"},
    "wavefront-substreams": {"HRT_FUSED": "0", "HRT_SUBSTREAMS": "3", "HRT_SUBSTREAM_MIN_PIXELS": "1024"},
    "wavefront-no-tail-split-small-slices": {"HRT_FUSED": "0", "HRT_TAIL_SPLIT": "0", "HRT_FETCH_CHUNK": "16", "HRT_REFILL_THRESHOLD": "4"},
    "fused-two-samples-per-launch": {"HRT_FUSED_MAX_SPP": "2"},
    "fused-round-1-kernel": {"HRT_FUSED": "2"},
    "fused-no-tail-splitting": {"HRT_TAIL_SPLIT": "0"},
    "fused-tail-splitting-eager-regeneration-few-waves": {"HRT_TAIL_REGEN": "1", "HRT_TRAVERSE_BLOCKS_PER_CU": "2", "HRT_FETCH_CHUNK": "16"},
    "fused-tail-splitting-lazy-regeneration": {"HRT_TAIL_REGEN": "40", "HRT_TRAVERSE_BLOCKS_PER_CU": "1"},
    "fused-deep-trees-take-round-1-kernel": {"HRT_FUSED_MAX_DEPTH": "1"},
    "fused-leaf-quorum-lazy-leaf-passes": {"HRT_LEAF_QUORUM": "12", "HRT_POSTPONE_PCT": "70"},
    "fused-eager-leaf-passes-one-wave-per-simd": {"HRT_POSTPONE_PCT": "0", "HRT_TRAVERSE_BLOCKS_PER_CU": "4", "HRT_REFILL_THRESHOLD": "48"},
    "fused-cost-ordered-slices-off": {"HRT_FUSED_LPT": "0"},
    "fused-small-slices": {"HRT_FETCH_CHUNK": "16", "HRT_REFILL_THRESHOLD": "4", "HRT_TRAVERSE_BLOCKS_PER_CU": "3"},
    "aligned-records": {"HRT_NODE_STRIDE": "128", "HRT_PRIM_STRIDE": "64"},
    "host-sah-build": {"HRT_BUILD": "host"},
    "device-build-wide-ploc-search": {"HRT_PLOC_RADIUS": "100", "HRT_BVH_CPRIM": "1.0"},
    "device-build-narrow-ploc-search-aligned-records": {"HRT_PLOC_RADIUS": "1", "HRT_NODE_STRIDE": "128", "HRT_PRIM_STRIDE": "64"},
}


@pytest.mark.parametrize("mode", sorted(MODES))
def test_every_execution_mode_is_bit_exact(hrt, oracle, gpu_available, monkeypatch, mode):
    """The production (non-counting) kernels in every execution mode -- wavefront default, fused path
    mode, LDS-DMA gathers, sub-tile streams, tuning extremes -- against the oracle: linear radiance,
    final RNG states and ray counts bit-exact, on a scene with all four programs and on the Cornell box."""
    if not gpu_available:
        pytest.skip("no GPU")
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    r = hrt.Renderer(0, 0)                      # flags 0: the production kernels (counting builds walk canonically)
    try:
        for scene, (w, h, spp) in ((hrt.scenes.mixed_test_scene(2500, 50, 13, 150, 90, 3), (150, 90, 3)),
                                   (hrt.scenes.cornell_box(128, 128, 5), (128, 128, 5)),
                                   (hrt.scenes.random_soup(30000, 0.05, 8, 200, 120, 2), (200, 120, 2))):
            salt = hrt.scenes.SEED_SALT
            r.load_scene(scene)
            r.set_frame(w, h, salt, aov=True, linear=True)
            r.reset_stats()
            r.render(spp)
            osc = oracle.OracleScene(scene)
            states = oracle.rng_init(w, h, salt)
            ref = osc.render(w, h, states, spp)
            assert np.array_equal(r.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32)), mode
            assert np.array_equal(r.rng_states_numpy(), states), mode
            assert r.stats().rays == ref["rays"], mode
            assert not r.albedo.cpu().numpy().any() and not r.normal.cpu().numpy().any()
            # a second launch continues the streams (quirk Q8)
            r.render(1)
            ref2 = osc.render(w, h, states, 1)
            assert np.array_equal(r.linear.cpu().numpy().view(np.uint32), ref2["linear"].view(np.uint32)), mode
    finally:
        r.close()


@pytest.mark.parametrize("builder", ["host", "device"])
def test_fast_trace_tree_with_spatial_splits(hrt, oracle, gpu_available, monkeypatch, builder):
    """HRT_CTX_FAST_TRACE (the reference's PREFER_FAST_TRACE on its static geometry, RendererImpl.cu:94): a tree with spatial splits
    (SBVH) -- triangles referenced from several leaves, each with the box of its part -- from the host builder, or
    (HRT_FAST_TRACE_BUILD=device) from the device's top-down phase + PLOC within the cells (build_split.hip).  Duplicates cannot change the
    canonical hit: image, RNG states, ray counts and hit records are the oracle's; the tree does hold more records than the scene
    has triangles; walked on the CPU it finds what brute force finds; and the first update replaces the split tree by a device-built
    one without splits (a refit would fall back to whole-primitive boxes)."""
    if not gpu_available:
        pytest.skip("no GPU")
    monkeypatch.setenv("HRT_FAST_TRACE_BUILD", builder)
    r = hrt.Renderer(0, hrt.CTX_FAST_TRACE)
    try:
        w, h, spp = 160, 100, 2
        scene = hrt.scenes.random_soup(60000, 0.06, 9, w, h, spp)
        r.load_scene(scene)
        _moved_scene_matches_oracle(hrt, oracle, r, scene, w, h, 3, spp)
        st = r.stats()
        assert st.bvh_triangles == 60000 and st.bvh_bytes > st.bvh_nodes * 80 + 60000 * 48 * 1.05       # duplicated references
        blob = hrt.BvhBlob()
        import ctypes as C
        r._check(r.lib.hrt_tlas_download(r.ctx, r.tlas, C.byref(blob)), "download")
        assert blob.n_triangles > 63000
        o, d = oracle.random_rays(4000, 77)
        want = oracle.OracleScene(scene, force_brute=True).trace(o, d)
        got = oracle.bvh8_trace(blob.nodes, blob.triangles, o, d)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[3], want[3])      # distance and primitive: the tree loses nothing
        r.lib.hrt_host_free(C.byref(blob))
        scene2 = hrt.scenes.mixed_test_scene(6000, 30, 5, w, h, spp)        # (above 4096 primitives: below, the device leaves the splits out)
        r.load_scene(scene2)
        _moved_scene_matches_oracle(hrt, oracle, r, scene2, w, h, 4, spp)
        moved = [it["transform"].copy() for it in scene2["instances"]]
        moved[0][3] += 0.1
        before = r.stats()
        assert before.bvh_bytes - before.bvh_nodes * 80 > (6000 + 30) * 48          # this tree has split references too
        r.update_instances(moved)
        after = r.stats()
        assert after.tlas_rebuilds == before.tlas_rebuilds + 1      # a tree with split references is not refitted: the update rebuilds (device build)
        for it, m in zip(scene2["instances"], moved):
            it["transform"] = m
        _moved_scene_matches_oracle(hrt, oracle, r, scene2, w, h, 5, spp)
        r.update_instances(moved)
        assert r.stats().tlas_refits == after.tlas_refits + 1        # ... and from then on it is an ordinary refittable tree
    finally:
        r.close()


@pytest.mark.parametrize("case", ["identical", "points", "two-clusters-far-apart"])
def test_device_build_thousands_of_coinciding_primitives(hrt, oracle, gpu_available, case):
    """6000 copies of one triangle among 6000 others: no plane separates them (they stay one cell of the top-down phase) and every merged
    area ties, so PLOC's "lowest position wins" rule alone would pair one couple a round and hand the collapse a chain (depth 860: the
    build used to fail on the traversal stack's depth limit); position i ^ 1 wins ties instead and the copies halve every round.  And a
    scene of two clusters 5000 units apart (the scene's box says nothing about either).  Default build and HRT_CTX_FAST_TRACE; hits as
    brute force finds them."""
    if not gpu_available:
        pytest.skip("no GPU")
    scene = hrt.scenes.random_soup(12000, 0.1, 3, 96, 64, 1)
    v = scene["instances"][0]["vertices"]
    o, d = oracle.random_rays(3000, 9)
    if case == "identical":
        v[:6000] = v[0]
    elif case == "points":
        v[:6000] = np.float32(0.25)          # 6000 triangles shrunk to one point: boxes without area (the collapse's costs would all tie)
    else:
        v[:6000] += np.float32(5000.0)
        o[:1500] += np.float32(5000.0)
    scene["instances"][0]["normals"] = hrt.scenes.face_normals(v)
    rt, ru, rv, rprim, rinst = oracle.OracleScene(scene, force_brute=True).trace(o, d)
    assert (rprim != 0xFFFFFFFF).sum() > 1000
    for flags in (0, hrt.CTX_FAST_TRACE):
        r = hrt.Renderer(0, flags)
        try:
            r.load_scene(scene)
            t, u, vv, prim, inst = r.trace_rays(o, d)
            assert np.array_equal(prim, rprim) and np.array_equal(t.view(np.uint32), rt.view(np.uint32)) and np.array_equal(u.view(np.uint32), ru.view(np.uint32))
            r.set_frame(96, 64, 3, aov=False)
            r.render(1)
            assert r.stats().bvh_depth <= 12
        finally:
            r.close()


def test_default_build_shapes_the_reference_kind_of_scene(hrt, oracle, gpu_available, monkeypatch):
    """The reference's kind of scene: hundreds of particles instancing a few closed shapes over a ground sphere of radius 1000.  PLOC
    alone (bottom-up merges in Morton order, search radius 2) has no view of such a scene's large-scale structure and its tree costs
    several times the node visits; the default device build therefore starts with the top-down phase (object splits only: the tree
    stays refittable).  Node visits per closest-hit ray of the counting pass: at most 9 by default, more than twice that with
    HRT_BUILD_TOPDOWN=0; the same hits either way, and the default tree is refitted, not rebuilt, by an update."""
    if not gpu_available:
        pytest.skip("no GPU")
    w, h = 320, 200
    scene = hrt.scenes.particle_scene(400, w, h, 1, subdiv=3)
    visits = {}
    hits = {}
    for topdown in ("1", "0"):
        monkeypatch.setenv("HRT_BUILD_TOPDOWN", topdown)
        r = hrt.Renderer(0, hrt.CTX_COUNT)
        try:
            r.load_scene(scene)
            r.set_frame(w, h, hrt.scenes.SEED_SALT, aov=False)
            r.render(1)
            st = r.stats()
            assert st.bvh_triangles > 4096
            visits[topdown] = st.node_visits_closest / max(st.rays_closest, 1)
            o, d = oracle.random_rays(4000, 5)
            hits[topdown] = r.trace_rays(o, d)
            if topdown == "1":
                before = r.stats()
                r.update_instances([it["transform"] for it in scene["instances"]])      # (identity move: the first update checks the refit on the spot)
                after = r.stats()
                assert after.tlas_refits == before.tlas_refits + 1 and after.tlas_rebuilds == before.tlas_rebuilds
        finally:
            r.close()
    assert visits["1"] <= 9.0 and visits["0"] >= 2.0 * visits["1"], visits
    assert all(np.array_equal(a, b) for a, b in zip(hits["1"], hits["0"]))
    rt, ru, rv, rprim, rinst = oracle.OracleScene(scene, force_brute=True).trace(*oracle.random_rays(4000, 5))
    assert np.array_equal(hits["1"][3], rprim) and np.array_equal(hits["1"][4], rinst) and np.array_equal(hits["1"][0].view(np.uint32), rt.view(np.uint32))


def test_device_split_build_with_full_segment_tables(hrt, oracle, gpu_available, monkeypatch):
    """The top-down phase keeps its segments in tables sized for twice what balanced splits make.  When they fill up (here: forced,
    HRT_SBVH_SEG_CAP) the level is planned again with every segment left as a cell, and PLOC builds the rest -- cells of thousands of
    references instead of a handful.  Same hits, same image."""
    if not gpu_available:
        pytest.skip("no GPU")
    monkeypatch.setenv("HRT_SBVH_SEG_CAP", "40")
    r = hrt.Renderer(0, hrt.CTX_FAST_TRACE)
    try:
        w, h, spp = 128, 80, 2
        scene = hrt.scenes.random_soup(30000, 0.08, 12, w, h, spp)
        r.load_scene(scene)
        _moved_scene_matches_oracle(hrt, oracle, r, scene, w, h, 6, spp)
        nodes, prims = _download_tree(hrt, r)
        assert prims.size > 30000 * 48          # the first levels did split references
    finally:
        r.close()


def test_device_build_falls_back_to_ploc_when_the_top_down_phase_gives_up(hrt, oracle, gpu_available, monkeypatch):
    """The top-down phase runs in every default build above 4096 primitives, rebuilds inside hrt_tlas_update included.  When it gives
    up (here: on request, HRT_SBVH_SEG_CAP=0; in the field: tables outgrown twice, no room for its temporaries) the build goes on with
    PLOC alone instead of failing: the tree HRT_BUILD_TOPDOWN=0 builds (as many nodes, the same records), same hits, same image -- also under HRT_CTX_FAST_TRACE."""
    if not gpu_available:
        pytest.skip("no GPU")
    w, h, spp = 128, 80, 2
    scene = hrt.scenes.random_soup(30000, 0.08, 12, w, h, spp)
    trees = {}
    for name, env, flags in (("ploc", {"HRT_BUILD_TOPDOWN": "0"}, 0), ("gave-up", {"HRT_SBVH_SEG_CAP": "0"}, 0), ("gave-up-fast-trace", {"HRT_SBVH_SEG_CAP": "0"}, hrt.CTX_FAST_TRACE)):
        for k in ("HRT_BUILD_TOPDOWN", "HRT_SBVH_SEG_CAP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = hrt.Renderer(0, flags)
        try:
            r.load_scene(scene)
            _moved_scene_matches_oracle(hrt, oracle, r, scene, w, h, 6, spp)
            trees[name] = _download_tree(hrt, r)
            moved = scene["instances"][0]["transform"].copy()
            r.update_instances([moved])                          # and an update refits such a tree (it has no split references)
            assert r.stats().tlas_refits == 1
        finally:
            r.close()
    # the tree PLOC alone builds: as many nodes, the same records (the emission hands out node and record blocks from atomic cursors,
    # so their ORDER within a level may differ from build to build)
    def records(t):
        rec = t[1].reshape(-1, 48)
        return rec[np.lexsort(rec.T[::-1])]
    for name in ("gave-up", "gave-up-fast-trace"):
        assert trees[name][0].size == trees["ploc"][0].size, name
        assert np.array_equal(records(trees[name]), records(trees["ploc"])), name


def test_device_split_build_is_deterministic_and_fast(hrt, gpu_available):
    """The device build with spatial splits takes its positions from prefix sums and its bounds from ordered-integer min / max: two
    builds of the same scene give the same tree (the same nodes and records, up to the order of the blocks they are stored in).  1 M triangles build in well under a second (measured: ~45 ms
    against 1.3 s for the host builder; the bar is loose because the box is shared), into a tree with more records than triangles."""
    if not gpu_available:
        pytest.skip("no GPU")
    import time
    import torch
    r = hrt.Renderer(0, hrt.CTX_FAST_TRACE)
    try:
        scene = hrt.scenes.random_soup(200000, 0.03, 4, 64, 64, 1)
        def signature():
            # the emission hands out child and record blocks from atomic cursors, so the ORDER of the blocks differs from build to
            # build; what the blocks hold does not: nodes without their two block offsets, and records, as sorted rows
            nodes, prims = _download_tree(hrt, r)
            nd = nodes.reshape(-1, 80).copy(); nd[:, 16:24] = 0
            pr = prims.reshape(-1, 48)
            return nd[np.lexsort(nd.T[::-1])], pr[np.lexsort(pr.T[::-1])]
        r.load_scene(scene)
        n0, p0 = signature()
        r.load_scene(scene)
        n1, p1 = signature()
        assert np.array_equal(n0, n1) and np.array_equal(p0, p1)
        assert p0.size > 200000 * 48 * 1.05
        big = hrt.scenes.soup_1m(64, 64, 1)
        r.load_scene(big)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.load_scene(big)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        assert dt < 1.0, dt
        assert _download_tree(hrt, r)[1].size > 1000000 * 48 * 1.1
    finally:
        r.close()


def test_wavefront_mode_replays_a_captured_graph(hrt, oracle, gpu_available, monkeypatch):
    """Wavefront mode (separate generate / traverse / bin / shade / accumulate kernels) with HRT_WAVEFRONT_GRAPH=1: the launches of
    samples 2 and 3 are captured as a hipGraph and replayed for the following pairs.  10 samples = samples 0 and 1 enqueued, 3 replays,
    samples 8 and 9 enqueued: image, RNG states and ray count are the oracle's, and the replays did happen."""
    if not gpu_available:
        pytest.skip("no GPU")
    monkeypatch.setenv("HRT_FUSED", "0")
    monkeypatch.setenv("HRT_WAVEFRONT_GRAPH", "1")
    r = hrt.Renderer(0, 0)
    try:
        w, h, spp, salt = 150, 90, 10, 77
        scene = hrt.scenes.mixed_test_scene(2500, 50, 13, w, h, spp)
        r.load_scene(scene)
        r.set_frame(w, h, salt, aov=False, linear=True)
        r.reset_stats()
        r.render(spp)
        states = oracle.rng_init(w, h, salt)
        ref = oracle.OracleScene(scene).render(w, h, states, spp)
        st = r.stats()
        assert st.graph_replays == 3 and st.rays == ref["rays"]
        assert np.array_equal(r.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32))
        assert np.array_equal(r.rng_states_numpy(), states)
    finally:
        r.close()


@pytest.mark.parametrize("builder", ["host", "device"])
def test_deep_trees_at_and_beyond_the_path_kernels_node_stack(hrt, oracle, gpu_available, monkeypatch, builder):
    """k_fused keeps one sibling group per tree level in LDS, 12 at most, and has no overflow path; deeper trees take round 1's
    path kernel by themselves.  Two scenes built to be deep (scenes.growing_chain): with the host builder 12 levels below the
    root -- the stack used to its last entry -- and 13; with the device builder whatever PLOC makes of them.  Image, RNG states and
    ray counts are the oracle's in every case."""
    if not gpu_available:
        pytest.skip("no GPU")
    if builder == "host":
        monkeypatch.setenv("HRT_BUILD", "host")
    depths = []
    r = hrt.Renderer(0, 0)
    try:
        for n in (210, 300):
            scene = hrt.scenes.growing_chain(n)
            w, h, spp = scene["width"], scene["height"], scene["spp"]
            r.load_scene(scene)
            r.set_frame(w, h, hrt.scenes.SEED_SALT, aov=False, linear=True)
            r.reset_stats()
            r.render(spp)
            states = oracle.rng_init(w, h, hrt.scenes.SEED_SALT)
            ref = oracle.OracleScene(scene).render(w, h, states, spp)
            assert ref["rays"] > 3 * w * h * spp                       # the paths do bounce around in there
            assert np.array_equal(r.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32)), n
            assert np.array_equal(r.rng_states_numpy(), states), n
            st = r.stats()
            assert st.rays == ref["rays"], n
            depths.append(int(st.bvh_depth))
    finally:
        r.close()
    if builder == "host":
        assert depths == [12, 13], depths          # = kFusedMaxDepth and one more: both kernels ran


def test_a_tree_too_deep_to_traverse_is_built_again_by_position(hrt, oracle, gpu_available):
    """393 triangles whose size grows by 7 % each, over twelve orders of magnitude: nearest-neighbour clustering (like SAH) takes one
    triangle off the rest at every level and the tree is deeper than any kernel's stack holds -- `hrt_tlas_build` used to say so and
    give up (found by tools/stress_modes.py).  The device build now notices and builds again by position (every cluster with its Morton
    neighbour: log2(n) levels); the image is the oracle's either way."""
    if not gpu_available:
        pytest.skip("no GPU")
    r = hrt.Renderer(0, 0)
    try:
        for n, growth in ((393, 1.0728), (369, 1.0777)):
            scene = hrt.scenes.growing_chain(n, growth, 96, 64, 2)
            r.load_scene(scene)
            r.set_frame(96, 64, 11, linear=True)
            r.render(2)
            st = r.stats()
            assert st.bvh_depth <= 12 and st.bvh_triangles == n, (int(st.bvh_depth), int(st.bvh_triangles))
            ref = oracle.OracleScene(scene).render(96, 64, oracle.rng_init(96, 64, 11), 2)
            assert np.array_equal(r.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32)), n
            o, d = oracle.random_rays(20000, 3, 40.0)
            got = r.trace_rays(o, d); want = oracle.OracleScene(scene, force_brute=True).trace(o, d)
            assert np.array_equal(got[3], want[3]) and np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))
    finally:
        r.close()


def test_record_arrays_beyond_32_bit_offsets_take_round_ones_kernel(hrt, oracle, gpu_available, monkeypatch):
    """k_fused addresses nodes and records by 32-bit byte offsets; arrays of 4 GiB and more take round 1's path kernel, which
    uses 64-bit addresses (csrc/hrt_api.cpp: fits_fused_kernel).  HRT_FUSED_MAX_BYTES lowers the limit so that a small scene
    drives that fallback: image, RNG states, ray count and hit records are the oracle's, and the fallback did run."""
    if not gpu_available:
        pytest.skip("no GPU")
    monkeypatch.setenv("HRT_FUSED_MAX_BYTES", "65536")
    r = hrt.Renderer(0, 0)
    try:
        w, h, spp = 96, 64, 2
        scene = hrt.scenes.mixed_test_scene(6000, 40, 5, w, h, spp)
        r.load_scene(scene)
        assert r.stats().fused_fallback_launches == 0
        _moved_scene_matches_oracle(hrt, oracle, r, scene, w, h, 23, spp)
        st = r.stats()
        assert st.bvh_bytes > 65536 and st.fused_fallback_launches >= 2          # the render and hrt_trace_rays
        assert np.array_equal(r.rng_states_numpy(), _oracle_states_after(oracle, scene, w, h, 23, spp))
    finally:
        r.close()
    monkeypatch.delenv("HRT_FUSED_MAX_BYTES")
    r = hrt.Renderer(0, 0)
    try:
        r.load_scene(scene)
        _moved_scene_matches_oracle(hrt, oracle, r, scene, w, h, 23, spp)
        assert r.stats().fused_fallback_launches == 0
    finally:
        r.close()


def _oracle_states_after(oracle, scene, w, h, salt, spp):
    st = oracle.rng_init(w, h, salt)
    oracle.OracleScene(scene).render(w, h, st, spp)
    return st


def test_eight_million_triangles_out_of_the_infinity_cache(hrt, oracle, gpu_available):
    """C4's law at 8 M triangles: 96 MB of nodes + 512 MB of records (+ 288 MB of normals) -- the tree no longer fits the 256 MiB
    Infinity Cache, so the kernel's loads meet HBM (profiles/r03_large_scenes.txt has the rates).  Device build, then a row
    subset of the 1920x1080 frame at 1 spp against the oracle (bit-exact, RNG states of the rendered rows included) and
    5000 rays against the oracle's brute force over all 8 M triangles."""
    if not gpu_available:
        pytest.skip("no GPU")
    W, H = 1920, 1080
    scene = hrt.scenes.soup_large(8_000_000, W, H, 1)
    r = hrt.Renderer(0, 0)
    try:
        r.load_scene(scene)
        r.set_frame(W, H, hrt.scenes.SEED_SALT, aov=False, linear=True)
        tile = hrt.Tile(0, H, 1, 45, 7)                        # rows 7, 52, 97, ...
        rows = np.arange(7, H, 45, dtype=np.uint32)
        r.reset_stats()
        r.render(1, tile=tile)
        st = r.stats()
        assert st.bvh_triangles == 8_000_000 and st.bvh_bytes > 450e6 and st.fused_fallback_launches == 0
        osc = oracle.OracleScene(scene)
        states = oracle.rng_init(W, H, hrt.scenes.SEED_SALT)
        ref = osc.render(W, H, states, 1, rows=rows)
        assert st.rays == ref["rays"] and ref["rays"] > 2 * len(rows) * W
        assert np.array_equal(r.linear.cpu().numpy()[rows].view(np.uint32), ref["linear"][rows].view(np.uint32))
        assert np.array_equal(r.color.cpu().numpy()[rows].view(np.uint32), ref["color"][rows].view(np.uint32))
        got_states = r.rng_states_numpy().reshape(H, W, 12)
        assert np.array_equal(got_states[rows], states.reshape(H, W, 12)[rows])
        o, d = oracle.random_rays(5000, 41)
        t, u, v, prim, inst = r.trace_rays(o, d)
        rt, ru, rv, rprim, rinst = oracle.OracleScene(scene, force_brute=True).trace(o, d)
        assert (prim != 0xFFFFFFFF).sum() > 2500
        assert np.array_equal(prim, rprim) and np.array_equal(inst, rinst)
        assert np.array_equal(t.view(np.uint32), rt.view(np.uint32)) and np.array_equal(u.view(np.uint32), ru.view(np.uint32)) and np.array_equal(v.view(np.uint32), rv.view(np.uint32))
        # the same scene under HRT_CTX_FAST_TRACE: the device build with spatial splits (8 M primitives through ~25 split levels,
        # ~11 M records) -- the same rows and the same hits, bit for bit
        linear0, states0 = r.linear.cpu().numpy()[rows].copy(), got_states[rows].copy()
        r2 = hrt.Renderer(0, hrt.CTX_FAST_TRACE)
        try:
            r2.load_scene(scene)
            r2.set_frame(W, H, hrt.scenes.SEED_SALT, aov=False, linear=True)
            r2.render(1, tile=tile)
            st2 = r2.stats()
            assert st2.rays == st.rays and st2.bvh_bytes > st.bvh_bytes * 1.2 and st2.fused_fallback_launches == 0
            assert np.array_equal(r2.linear.cpu().numpy()[rows].view(np.uint32), linear0.view(np.uint32))
            assert np.array_equal(r2.rng_states_numpy().reshape(H, W, 12)[rows], states0)
            t2, u2, v2, prim2, inst2 = r2.trace_rays(o, d)
            assert np.array_equal(prim2, prim) and np.array_equal(t2.view(np.uint32), t.view(np.uint32)) and np.array_equal(u2.view(np.uint32), u.view(np.uint32))
        finally:
            r2.close()
    finally:
        r.close()


def test_cost_ordered_slices_bit_exact(hrt, oracle, gpu_available):
    """Few pixels per lane and many samples: the first samples run as a probe launch that measures the slices, the rest
    of the render hands the slices out slowest first.  The order is free -- image, RNG states and ray count are those of
    the oracle -- and the render did take the two launches."""
    if not gpu_available:
        pytest.skip("no GPU")
    r = hrt.Renderer(0, 0)
    try:
        w, h, spp = 512, 288, 16
        scene = hrt.scenes.random_soup(30000, 0.05, 8, w, h, spp)
        r.load_scene(scene)
        r.set_frame(w, h, hrt.scenes.SEED_SALT, aov=False, linear=True)
        r.reset_stats()
        r.render(spp)
        s = r.stats()
        assert s.kernel_launches[hrt.K_PATHS] == 2          # probe + the rest
        states = oracle.rng_init(w, h, hrt.scenes.SEED_SALT)
        ref = oracle.OracleScene(scene).render(w, h, states, spp)
        assert np.array_equal(r.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32))
        assert np.array_equal(r.rng_states_numpy(), states) and s.rays == ref["rays"]
    finally:
        r.close()


def test_fused_mode_tiles(hrt, oracle, gpu_available, monkeypatch):
    """Fused path mode on stripe tiles: union == full frame, rows outside the tile untouched."""
    if not gpu_available:
        pytest.skip("no GPU")
    r = hrt.Renderer(0, 0)                      # fused path mode is the default of the production build
    try:
        scene = hrt.scenes.mixed_test_scene(1500, 30, 3, 80, 50, 2)
        r.load_scene(scene)
        ref = oracle.OracleScene(scene).render(80, 50, oracle.rng_init(80, 50, 7), 2)
        r.set_frame(80, 50, 7, linear=True)
        acc = np.zeros((50, 80, 4), np.float32)
        for rank in range(3):
            r.color.zero_()
            r.render(2, tile=hrt.tile_for_rank(50, rank, 3, stripe_rows=4))
            part = r.color.cpu().numpy()
            others = [y for y in range(50) if (y // 4) % 3 != rank]
            assert np.all(part[others] == 0)
            acc += part
        assert np.array_equal(r.linear.cpu().numpy().view(np.uint32), ref["linear"].view(np.uint32))
        assert np.array_equal(acc.view(np.uint32), ref["color"].view(np.uint32))
    finally:
        r.close()


def test_cpp_multi_gpu_host_with_rccl(gpu_available):
    """csrc/host/multi_gpu.hpp through the hrt_render driver: one process, N devices (here N = 1, the box has one GPU),
    ncclCommInitAll, per-device context + stream, stripes through HrtTile, one ncclReduce(sum) of the float4 frame into
    device 0 -- and --check renders the frame again on one context, untiled, and compares the bits."""
    if not gpu_available:
        pytest.skip("no GPU")
    import subprocess
    import tempfile
    from pathlib import Path
    exe = Path(__file__).resolve().parent.parent / "nvidia-optix-ray-tracer_amd" / "lib" / "hrt_render"
    assert exe.exists(), "run `make tools`"
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([str(exe), "20000", "320", "200", "3", str(Path(tmp) / "f.ppm"), "--gpus", "1", "--check"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "IS BIT-IDENTICAL TO" in r.stdout and "1 GPU(s)" in r.stdout
        assert (Path(tmp) / "f.ppm").stat().st_size == 320 * 200 * 3 + len("P6\n320 200\n255\n")
