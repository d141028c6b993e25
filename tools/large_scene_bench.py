#!/usr/bin/env python3
"""Out-of-cache datapoints: C4's soup law at 1 M / 8 M / 32 M triangles (tree + records beyond the 256 MiB Infinity Cache
from 8 M on).  One JSON line per run: device build time, Mrays/s at 1920x1080 x spp, nodes / primitives per ray (counting
pass), the algorithmic-bytes fraction of the HBM peak.  tools/profile_large.sh wraps it in rocprofv3 --pmc passes for the
memory-side counters (fabric bytes per launch, TCC hit rate)."""
import os, argparse, ctypes as C, importlib, json, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=8_000_000)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--no-count", action="store_true", help="skip the counting pass and the build timing (PMC runs)")
ap.add_argument("--fast-trace", action="store_true", help="HRT_CTX_FAST_TRACE: the build with spatial splits (on the device; HRT_FAST_TRACE_BUILD=host: the host builder) instead of the default device build")
ap.add_argument("--cache-dir", default="/tmp/hrt_scenes")
args = ap.parse_args()

import torch
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
W, H = 1920, 1080
cache = Path(args.cache_dir) / f"soup_{args.tris}.npz"
if args.tris == 1_000_000:
    scene = hrt.scenes.soup_1m(W, H, args.spp)
elif cache.exists():
    d = np.load(cache)
    scene = hrt.scenes.soup_large(1000, W, H, args.spp)
    scene["instances"][0]["vertices"], scene["instances"][0]["normals"] = d["v"], d["n"]
    scene["name"] = f"soup-{args.tris // 1_000_000}M"
else:
    scene = hrt.scenes.soup_large(args.tris, W, H, args.spp)
    try:
        cache.parent.mkdir(parents=True, exist_ok=True)
        np.savez(cache, v=scene["instances"][0]["vertices"], n=scene["instances"][0]["normals"])
    except OSError:
        pass
r = hrt.Renderer(0, hrt.CTX_TIMING | (hrt.CTX_FAST_TRACE if args.fast_trace else 0))
t0 = time.perf_counter(); r.load_scene(scene); load_s = time.perf_counter() - t0
out = {"scene": scene["name"], "triangles": args.tris, "edge": hrt.scenes.soup_law_edge(args.tris) if args.tris != 1_000_000 else 0.014,
       "builder": ("device top-down SAH with spatial splits + PLOC" if os.environ.get("HRT_FAST_TRACE_BUILD", "device") == "device" else "host binned SAH with spatial splits") if args.fast_trace else ("device top-down SAH (object splits) + PLOC: the default build" if os.environ.get("HRT_BUILD_TOPDOWN", "1") != "0" else "device PLOC alone"),
       "load_scene_s": round(load_s, 3)}
if not args.no_count and (not args.fast_trace or os.environ.get("HRT_FAST_TRACE_BUILD", "device") == "device"):
    # the TLAS build alone: the same instance array once more, timed from call to stream idle
    best = 1e9
    for _ in range(2):
        h = C.c_uint64()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r._check(r.lib.hrt_tlas_build(r.ctx, r._d_inst.data_ptr(), r._n_inst, r._stream(), C.byref(h)), "hrt_tlas_build")
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
        r._check(r.lib.hrt_tlas_destroy(r.ctx, h.value), "hrt_tlas_destroy")
    out["device_build_ms"] = round(best * 1e3, 2)
r.set_frame(W, H, hrt.scenes.SEED_SALT, aov=False)
r.render(min(args.spp, 2), sync=True)                     # warm-up
r.reset_stats()
t0 = time.perf_counter()
for _ in range(args.steps):
    r.render(args.spp, sync=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = r.stats()
out.update({"spp": args.spp, "steps": args.steps, "mrays_per_s": round(st.rays / dt / 1e6, 1), "ms_per_step": round(dt / args.steps * 1e3, 2),
            "rays_per_step": int(st.rays // args.steps), "kernel_ms_per_step": round(st.kernel_ms[hrt.K_PATHS] / args.steps, 2),
            "bvh_nodes": int(st.bvh_nodes), "bvh_depth": int(st.bvh_depth), "bvh_bytes": int(st.bvh_bytes), "bvh_alloc_bytes": int(st.bvh_alloc_bytes),
            "tree_plus_normals_MB": round((st.bvh_nodes * 80 + st.bvh_triangles * 64 + st.bvh_triangles * 36) / 1e6, 1),
            "fused_fallback_launches": int(st.fused_fallback_launches)})
if not args.no_count:
    r.set_flags(hrt.CTX_COUNT)
    r.reset_stats()
    r.render(1, sync=True)
    sc = r.stats()
    n_c, p_c = sc.node_visits_closest / max(sc.rays_closest, 1), sc.prim_tests_closest / max(sc.rays_closest, 1)
    n_a = (sc.node_visits - sc.node_visits_closest) / max(sc.rays_any, 1)
    p_a = (sc.prim_tests - sc.prim_tests_closest) / max(sc.rays_any, 1)
    bytes_per_launch = ((80 * n_c + 48 * p_c) * st.rays_closest + (80 * n_a + 48 * p_a) * st.rays_any) / args.steps
    out.update({"nodes_per_ray": round(n_c, 2), "prims_per_ray": round(p_c, 2), "algorithmic_bytes_per_launch": round(bytes_per_launch),
                "algorithmic_frac_of_8TBs": round(bytes_per_launch / (st.kernel_ms[hrt.K_PATHS] / args.steps * 1e-3) / 8e12, 4)})
print(json.dumps(out), flush=True)
r.close()
