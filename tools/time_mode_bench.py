#!/usr/bin/env python3
"""Time-mode frame loop of the reference (src/Global/RendererTime.cu:425-500) on one MI355X:
per frame  pose kernel -> hrt_tlas_update (device refit) -> hrt_render_launch (1 spp) -> hrt_to_rgba8.
Prints one JSON line per scene with ms per frame, frames/s and Mrays/s (window 1200x800 as files/config.json).
Usage: python tools/time_mode_bench.py [--frames 200]"""
import argparse, importlib, json, os, sys, time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")


def run(n_particles, subdiv, frames, w=1200, h=800, two_level=False):
    import torch
    scene = hrt.scenes.particle_scene(n_particles, w, h, 1, subdiv=subdiv)
    r = hrt.Renderer(0, (0 if os.environ.get("HRT_SYNC_UPDATE") else hrt.CTX_ASYNC_UPDATE) | (hrt.CTX_TWO_LEVEL if two_level else 0))
    r.load_scene(scene)
    r.set_frame(w, h, hrt.scenes.SEED_SALT, aov=False)
    cur = r._dev(hrt.scenes.particle_states(n_particles, 0))
    nxt = r._dev(hrt.scenes.particle_states(n_particles, 1))

    def frame(f):
        r.pose_instances(cur, nxt, 0.5, f, frames, first_instance=1)
        r.render(1, sync=True)                       # the reference synchronises after every launch (:500)
        r.to_rgba8()

    for f in range(3):
        frame(f)
    torch.cuda.synchronize()
    r.reset_stats()
    t0 = time.perf_counter()
    for f in range(frames):
        frame(f)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s = r.stats()
    print(json.dumps({"scene": scene["name"], "structure": "two-level" if two_level else "flattened", "window": [w, h], "triangles": int(s.bvh_triangles), "tree_bytes": int(s.bvh_alloc_bytes), "frames": frames,
                      "ms_per_frame": round(dt * 1e3 / frames, 4), "fps": round(frames / dt, 1),
                      "Mrays_per_s": round(s.rays / dt / 1e6, 1), "rays_per_frame": int(s.rays // frames),
                      "tlas_refits": int(s.tlas_refits), "tlas_rebuilds": int(s.tlas_rebuilds)}), flush=True)
    r.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--dem", type=int, default=0, help="N: the loop at DEM scale instead -- N particles (subdiv 3), flattened against two-level tree")
    a = ap.parse_args()
    if a.dem:
        os.environ["HRT_TWO_LEVEL"] = "-1"; run(a.dem, 3, a.frames)
        os.environ["HRT_TWO_LEVEL"] = "0"; run(a.dem, 3, a.frames, two_level=True)
        sys.exit(0)
    for instanced in ("0", "1"):                     # hrt_tlas_build: merged tree / tree over instances
        os.environ["HRT_TLAS_INSTANCED"] = instanced
        print("HRT_TLAS_INSTANCED=" + instanced, flush=True)
        run(25, 2, a.frames)
        run(2000, 3, a.frames)
