#!/usr/bin/env python3
"""Lane utilisation of the fused path kernel per phase (C4, 16 spp).  Needs a library built with the counters compiled in:
    touch nvidia-optix-ray-tracer_amd/csrc/kernels.hip && make lib HIPFLAGS="$(make -pn | sed -n 's/^HIPFLAGS := //p') -DHRT_LANE_STATS"
(the counters reuse HrtStats.debug[] and the two counting fields the production kernels leave at zero)."""
import importlib, sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
if len(sys.argv) > 1 and sys.argv[1] == "particles":       # tools/lane_stats.py particles N: the reference's kind of scene, 1200x800
    n_p = int(sys.argv[2])
    scene = hrt.scenes.particle_scene(n_p, 1200, 800, 16, subdiv=2 if n_p <= 100 else 3)
    r = hrt.Renderer(0, 0); r.load_scene(scene); r.set_frame(1200, 800, hrt.scenes.SEED_SALT, aov=False)
else:
    scene = hrt.scenes.soup_1m(1920, 1080, 16)
    r = hrt.Renderer(0, hrt.CTX_FAST_TRACE); r.load_scene(scene); r.set_frame(1920, 1080, hrt.scenes.SEED_SALT, aov=False)      # (the tree bench.py times)
r.render(2); r.reset_stats(); r.render(16)
s = r.stats()
it, alive, node, prim = s.debug[0], s.debug[1], s.debug[2], s.debug[3]
ppass, regen = s.node_visits_closest, s.prim_tests_closest
print("rays", s.rays, "wave iterations", it, "iters/ray-lane", it*64/s.rays)
print("alive lanes/iter %.1f  node lanes/iter %.1f  prim lanes/iter %.1f" % (alive/it, node/it, prim/it))
print("prim passes per iteration %.3f  lanes per prim pass %.1f  regens per iteration %.4f (every %.1f iterations)" % (ppass/it, prim/max(ppass,1), regen/it, it/max(regen,1)))
print("node steps per ray %.2f  prim tests per ray %.2f" % (node/s.rays, prim/s.rays))
