#!/usr/bin/env python3
"""Two-level trees against flattened ones on the reference's kind of scene (particles instancing a few shared shapes over the ground
sphere): memory, build and update cost, Mrays/s, and -- with the counters compiled in (make stats; HRT_LIB=.../libhrt_stats.so) --
node steps per ray.  One JSON line per (scene, structure).
Usage: tools/two_level_bench.py [--particles 2000,100000] [--subdiv 3] [--spp 1,4] [--structures flat,two] [--frames 10] [--render-only]"""
import argparse, ctypes as C, importlib, json, os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")

ap = argparse.ArgumentParser()
ap.add_argument("--particles", default="2000,100000"); ap.add_argument("--subdiv", type=int, default=3); ap.add_argument("--spp", default="1,4")
ap.add_argument("--structures", default="flat,two"); ap.add_argument("--frames", type=int, default=10); ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080); ap.add_argument("--render-only", action="store_true"); ap.add_argument("--async-update", action="store_true")
ap.add_argument("--scene", default="cloud", help="cloud (jittered cubic grid) or column (scenes.particle_scene: the 5-wide grid of the shipped sample)")
a = ap.parse_args()
stats_build = "stats" in os.environ.get("HRT_LIB", "")

for n_p in (int(x) for x in a.particles.split(",")):
    scene = (hrt.scenes.particle_cloud(n_p, a.width, a.height, 1, subdiv=a.subdiv) if a.scene == "cloud"
             else hrt.scenes.particle_scene(n_p, a.width, a.height, 1, subdiv=a.subdiv))
    flat_prims = sum(len(it["vertices"]) if it["geometry"] == "triangles" else len(it["radii"]) for it in scene["instances"])
    for structure in a.structures.split(","):
        os.environ["HRT_TWO_LEVEL"] = "0" if structure == "two" else "-1"      # (read when the context is created; without it the size rule decides)
        r = hrt.Renderer(0, (hrt.CTX_TWO_LEVEL if structure == "two" else 0) | (hrt.CTX_ASYNC_UPDATE if a.async_update else 0))
        t0 = time.perf_counter(); r.load_scene(scene); torch.cuda.synchronize(); load_s = time.perf_counter() - t0
        n = len(scene["instances"]); st = r._stream()
        out = {"scene": scene["name"], "structure": structure, "async_update": a.async_update, "instances": n, "flattened_primitives": flat_prims, "load_scene_s": round(load_s, 3)}
        if not a.render_only:
            # hrt_tlas_build alone (the instance array is on the device already, the BLASes exist)
            tl = C.c_uint64(); t0 = time.perf_counter()
            r._check(r.lib.hrt_tlas_build(r.ctx, r._d_inst.data_ptr(), n, st, C.byref(tl)), "hrt_tlas_build"); torch.cuda.synchronize()
            out["tlas_build_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
            r._check(r.lib.hrt_tlas_destroy(r.ctx, tl.value), "hrt_tlas_destroy")
            # hrt_tlas_update: every particle moved a little (refit), call -> GPU done
            raw = np.frombuffer(bytes(r._h_inst), dtype=np.uint8).reshape(-1, 80).copy()
            bufs = []
            for f in range(a.frames + 1):
                m = raw.copy(); xf = m[:, :48].view(np.float32).reshape(-1, 12)
                xf[1:, 3] += np.float32(0.002 * f); xf[1:, 11] -= np.float32(0.003 * f)
                bufs.append(r._dev(m.reshape(-1)))
            r._check(r.lib.hrt_tlas_update(r.ctx, r.tlas, bufs[0].data_ptr(), n, st), "hrt_tlas_update"); torch.cuda.synchronize()
            s0 = r.stats(); t0 = time.perf_counter()
            for f in range(1, a.frames + 1):
                r._check(r.lib.hrt_tlas_update(r.ctx, r.tlas, bufs[f].data_ptr(), n, st), "hrt_tlas_update")
            torch.cuda.synchronize(); out["tlas_update_ms"] = round((time.perf_counter() - t0) * 1e3 / a.frames, 4)
            s1 = r.stats(); out["refits"] = int(s1.tlas_refits - s0.tlas_refits); out["rebuilds"] = int(s1.tlas_rebuilds - s0.tlas_rebuilds)
        r.set_frame(a.width, a.height, hrt.scenes.SEED_SALT, aov=False)
        r.render(1)
        s = r.stats()
        out.update({"bvh_alloc_bytes": int(s.bvh_alloc_bytes), "bvh_nodes": int(s.bvh_nodes), "records": int(s.bvh_triangles + s.bvh_spheres), "depth": int(s.bvh_depth)})
        for spp in (int(x) for x in a.spp.split(",")):
            r.render(spp, sync=True); r.reset_stats()
            reps = max(3, 32 // spp)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): r.render(spp, sync=False)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            s = r.stats()
            out["spp%d" % spp] = {"ms_per_launch": round(dt / reps * 1e3, 3), "Mrays_per_s": round(s.rays / dt / 1e6, 1), "rays_per_path": round(s.rays / max(s.paths, 1), 3)}
            if stats_build:
                it, node, prim = s.debug[0], s.debug[2], s.debug[3]
                out["spp%d" % spp].update({"node_steps_per_ray": round(node / s.rays, 2), "prim_tests_per_ray": round(prim / s.rays, 2),
                                            "wave_iterations_per_ray_lane": round(it * 64 / s.rays, 2), "lanes_alive_per_iteration": round(s.debug[1] / it, 1),
                                            "instances_entered_per_ray": round((s.node_visits - s.node_visits_closest) / s.rays, 2)})
        print(json.dumps(out), flush=True)
        r.close()
