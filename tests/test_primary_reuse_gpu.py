"""HRT_CTX_REUSE_PRIMARY: the reference's raygen has no pixel jitter (shader/Shader.cu:249-261 -- the pixel centre, every sample), so
a pixel's primary ray finds the same hit in each of a launch's samples.  With the flag the path kernel (k_fused<.., REUSE>, fused.hip)
traverses it once per launch and shades the later samples from the cached hit record.

Bar: the image is the SAME BITS as without the flag (and as the oracle's), and HrtStats.rays counts the traversed rays only: it falls by
exactly one primary ray per pixel and reused sample."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render(hrt, scene, w, h, spp, flags, salt=91, tile=None):
    r = hrt.Renderer(0, flags)
    try:
        r.load_scene(scene)
        r.set_frame(w, h, salt, linear=True)
        r.reset_stats()
        r.render(spp, tile=tile)
        s = r.stats()
        return r.linear.cpu().numpy().copy(), r.color.cpu().numpy().copy(), int(s.rays), int(s.paths), int(s.fused_fallback_launches)
    finally:
        r.close()


@pytest.mark.parametrize("name", ["mixed", "cornell", "sky", "spheres"])
def test_reused_primary_hits_give_the_same_bits_and_fewer_rays(hrt, oracle, gpu_available, name):
    """All four programs (spheres and triangles, rough and metal), a frame that is mostly background (primary misses: the pixel's
    samples are all the background colour), a closed box: frames with the flag against frames without, and against the oracle."""
    if not gpu_available:
        pytest.skip("no GPU in this container")
    w, h, spp = 160, 96, 12
    scene = {"mixed": lambda: hrt.scenes.mixed_test_scene(3000, 40, 5, w, h, spp),
             "cornell": lambda: hrt.scenes.cornell_box(w, h, spp),
             "sky": lambda: hrt.scenes.mixed_test_scene(60, 3, 8, w, h, spp),
             "spheres": lambda: hrt.scenes.sphere_in_box(w, h, spp)}[name]()
    lin0, col0, rays0, paths0, _ = _render(hrt, scene, w, h, spp, 0)
    lin1, col1, rays1, paths1, fb = _render(hrt, scene, w, h, spp, hrt.CTX_REUSE_PRIMARY)
    assert fb == 0
    assert np.array_equal(lin0.view(np.uint32), lin1.view(np.uint32)) and np.array_equal(col0.view(np.uint32), col1.view(np.uint32))
    assert paths0 == paths1 == w * h * spp
    assert rays0 - rays1 == w * h * (spp - 1)               # one primary ray per pixel instead of one per sample; every bounce still traced
    ref = oracle.OracleScene(scene).render(w, h, oracle.rng_init(w, h, 91), spp)
    assert np.array_equal(lin1.view(np.uint32), ref["linear"].view(np.uint32)) and rays0 == ref["rays"]


def test_reuse_across_the_launches_of_a_long_render_and_the_probe_launch(hrt, gpu_available, monkeypatch):
    """A render cut into several launches (HRT_FUSED_MAX_SPP) traverses the primary ray once per launch -- the running sums and RNG
    states carry over, the cache does not have to; a small tile with many slices also takes the cost-ordering probe launch first (whose
    background pixels are not short-cut: their cost is being measured)."""
    if not gpu_available:
        pytest.skip("no GPU in this container")
    w, h, spp = 512, 256, 16
    scene = hrt.scenes.mixed_test_scene(4000, 30, 3, w, h, spp)
    lin0, _, rays0, _, _ = _render(hrt, scene, w, h, spp, 0)
    lin1, _, rays1, _, _ = _render(hrt, scene, w, h, spp, hrt.CTX_REUSE_PRIMARY)        # probe launch (2 spp) + one launch of 14
    assert np.array_equal(lin0.view(np.uint32), lin1.view(np.uint32))
    assert rays0 - rays1 == w * h * (spp - 2)
    monkeypatch.setenv("HRT_FUSED_MAX_SPP", "5")
    monkeypatch.setenv("HRT_FUSED_LPT", "0")
    lin2, _, rays2, _, _ = _render(hrt, scene, w, h, spp, hrt.CTX_REUSE_PRIMARY)        # launches of 5, 5, 5, 1 samples
    assert np.array_equal(lin0.view(np.uint32), lin2.view(np.uint32))
    assert rays0 - rays2 == w * h * (spp - 4)
    monkeypatch.delenv("HRT_FUSED_MAX_SPP")
    monkeypatch.setenv("HRT_REUSE_PRIMARY", "1")                                         # the knob: the flag for every context
    lin3, _, rays3, _, _ = _render(hrt, scene, w, h, spp, 0)
    assert np.array_equal(lin0.view(np.uint32), lin3.view(np.uint32)) and rays0 - rays3 == w * h * (spp - 1)


def test_reuse_on_a_tile_and_through_a_two_level_tree(hrt, oracle, gpu_available):
    """A rank's stripes of a frame (the multi-GPU split), and the INSTANCED instantiation of the kernel (an IAS over shared GASes):
    same bits as without the flag in both."""
    if not gpu_available:
        pytest.skip("no GPU in this container")
    w, h, spp = 192, 128, 8
    scene = hrt.scenes.mixed_test_scene(3000, 40, 11, w, h, spp)
    tile = hrt.tile_for_rank(h, 1, 2)
    lin0, _, rays0, _, _ = _render(hrt, scene, w, h, spp, 0, tile=tile)
    lin1, _, rays1, _, _ = _render(hrt, scene, w, h, spp, hrt.CTX_REUSE_PRIMARY, tile=tile)
    assert np.array_equal(lin0.view(np.uint32), lin1.view(np.uint32))
    n_rows = sum(1 for y in range(h) if (y // 8) % 2 == 1)
    assert rays0 - rays1 == w * n_rows * (spp - 1)
    cloud = hrt.scenes.particle_cloud(300, w, h, spp)
    lin2, _, rays2, _, _ = _render(hrt, cloud, w, h, spp, hrt.CTX_TWO_LEVEL)
    lin3, _, rays3, _, fb = _render(hrt, cloud, w, h, spp, hrt.CTX_TWO_LEVEL | hrt.CTX_REUSE_PRIMARY)
    assert fb == 0 and np.array_equal(lin2.view(np.uint32), lin3.view(np.uint32)) and rays2 - rays3 == w * h * (spp - 1)
    ref = oracle.OracleScene(cloud, instanced=True).render(w, h, oracle.rng_init(w, h, 91), spp)
    assert np.array_equal(lin3.view(np.uint32), ref["linear"].view(np.uint32))
