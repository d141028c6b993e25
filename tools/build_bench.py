#!/usr/bin/env python3
"""Acceleration-structure build: device (the default: top-down object splits + PLOC in the cells), ploc (PLOC alone, HRT_BUILD_TOPDOWN=0), split (HRT_CTX_FAST_TRACE: the device's top-down phase with spatial splits +
PLOC in the cells, build_split.hip), host (binned SAH, bvh8_build.cpp) and host-split (the host builder with spatial splits) -- build time of
hrt_tlas_build (call to completion) and what the tree is worth when traced (C3 / C4 frame, a few spp).
Usage: tools/build_bench.py [C3|C4] [spp]      (HRT_BUILD_BENCH_MODES=device,split,host,host-split selects)"""
import importlib, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
scene = hrt.scenes.BASELINE_CONFIGS[cfg]()
W, H = scene["width"], scene["height"]
for mode in os.environ.get("HRT_BUILD_BENCH_MODES", "device,ploc,split,host,host-split").split(","):
    os.environ.pop("HRT_BUILD", None); os.environ.pop("HRT_FAST_TRACE_BUILD", None); os.environ.pop("HRT_BUILD_TOPDOWN", None)
    if mode == "ploc":
        os.environ["HRT_BUILD_TOPDOWN"] = "0"
    if mode == "host":
        os.environ["HRT_BUILD"] = "host"
    if mode == "host-split":
        os.environ["HRT_FAST_TRACE_BUILD"] = "host"
    r = hrt.Renderer(0, hrt.CTX_TIMING | (hrt.CTX_FAST_TRACE if mode in ("split", "host-split") else 0))
    r.load_scene(scene)                       # warm-up build (allocator, code objects)
    times = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.load_scene(scene)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    r.set_frame(W, H, hrt.scenes.SEED_SALT, aov=False)
    r.render(2); r.reset_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r.render(spp)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = r.stats()
    r.set_flags(hrt.CTX_COUNT); r.reset_stats(); r.render(1); c = r.stats()
    walk = f"{c.node_visits_closest / max(c.rays_closest, 1):.2f} nodes + {c.prim_tests_closest / max(c.rays_closest, 1):.2f} primitives per closest-hit ray"
    print(f"{cfg} {mode:10s} build: load_scene (BLAS copies + TLAS build) best of 3 = {min(times)*1e3:8.2f} ms; tree {s.bvh_nodes} nodes, {s.bvh_bytes/1e6:.1f} MB; "
          f"trace {s.rays/dt/1e6:8.1f} Mrays/s at {spp} spp; {walk}", flush=True)
    r.close()
