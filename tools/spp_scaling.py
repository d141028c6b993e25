#!/usr/bin/env python3
"""Rate against samples per launch on the particle scenes (the reference's layout: particles instancing a few shapes over a huge ground
sphere), and what the tree is worth there: tools/spp_scaling.py [--fast-trace] [--spp 1,4,16]   (builder knobs by environment)."""
import argparse, importlib, sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--fast-trace", action="store_true"); ap.add_argument("--spp", default="1,4,16"); ap.add_argument("--particles", default="25,2000")
a = ap.parse_args()
for n_p in (int(x) for x in a.particles.split(",")):
    scene = hrt.scenes.particle_scene(n_p, 1200, 800, 1, subdiv=2 if n_p <= 100 else 3)
    r = hrt.Renderer(0, hrt.CTX_TIMING | (hrt.CTX_FAST_TRACE if a.fast_trace else 0))
    t0 = time.perf_counter(); r.load_scene(scene); load_ms = (time.perf_counter() - t0) * 1e3
    r.set_frame(1200, 800, hrt.scenes.SEED_SALT, aov=False)
    for spp in (int(x) for x in a.spp.split(",")):
        r.render(spp, sync=True); r.reset_stats()
        reps = max(2, 64 // spp)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): r.render(spp, sync=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        s = r.stats()
        print(json.dumps({"scene": scene["name"], "spp": spp, "ms_per_launch": round(dt / reps * 1e3, 3), "Mrays_per_s": round(s.rays / dt / 1e6, 1),
                          "bvh_nodes": int(s.bvh_nodes), "load_ms": round(load_ms, 2)}), flush=True)
    r.set_flags(hrt.CTX_COUNT); r.reset_stats(); r.render(1); c = r.stats()
    print(f"   {c.node_visits_closest / max(c.rays_closest, 1):.2f} nodes + {c.prim_tests_closest / max(c.rays_closest, 1):.2f} primitives per closest-hit ray", flush=True)
    r.close()
