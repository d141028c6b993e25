import importlib, sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
for with_sphere in (False, True, False, True):
    scene = hrt.scenes.soup_1m(1920, 1080, 16)
    if with_sphere:   # one tiny sphere at (x, x, x), x = argv[1] (default 50: far outside the soup; 0: inside): the scene then takes the sphere-capable kernel -- and, far away, stretches the root
        scene["instances"].append(hrt.scenes._sphere_instance([[float(sys.argv[1]) if len(sys.argv) > 1 else 50.0] * 3], [1e-4], [0.5, 0.5, 0.5]))
    r = hrt.Renderer(0, hrt.CTX_TIMING | hrt.CTX_FAST_TRACE)
    r.load_scene(scene); r.set_frame(1920, 1080, hrt.scenes.SEED_SALT, aov=False)
    r.render(16, sync=True); r.reset_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): r.render(16, sync=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = r.stats()
    rate = s.rays / dt / 1e6
    r.set_flags(hrt.CTX_COUNT); r.reset_stats(); r.render(1); c = r.stats()
    print("C4 at 16 spp", "with one sphere (k_fused<true>)" if with_sphere else "triangles only (k_fused<false>)", round(rate, 1), "Mrays/s;",
          f"{c.node_visits_closest / max(c.rays_closest, 1):.2f} nodes + {c.prim_tests_closest / max(c.rays_closest, 1):.2f} primitives per closest-hit ray; {c.bvh_nodes} nodes, depth {c.bvh_depth}", flush=True)
    r.close()
