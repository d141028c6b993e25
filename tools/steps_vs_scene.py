#!/usr/bin/env python3
"""Traversal steps/s for incoherent rays as a function of BVH size (cache residency)."""
import importlib, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
rng = np.random.default_rng(0)
n = 2_000_000
o = rng.uniform(-1, 1, (n, 3)).astype(np.float32); d = rng.normal(size=(n, 3)).astype(np.float32)
for ntri, edge in ((300, 0.25), (3000, 0.1), (30000, 0.05), (300000, 0.022), (1000000, 0.014), (4000000, 0.009)):
    scene = hrt.scenes.random_soup(ntri, edge, 3)
    r = hrt.Renderer(0, hrt.CTX_COUNT)
    r.load_scene(scene)
    r.reset_stats(); r.trace_rays(o, d); c = r.stats()
    steps = c.node_visits + c.prim_tests
    r.set_flags(hrt.CTX_TIMING)
    r.trace_rays(o, d); r.reset_stats()
    for _ in range(3):
        r.trace_rays(o, d)
    ms = r.stats().kernel_ms[hrt.K_TRAVERSE] / 3
    print(f"{ntri:8d} tris, bvh {c.bvh_bytes/1e6:7.2f} MB: {c.node_visits/n:6.1f} nodes + {c.prim_tests/n:5.1f} prims per ray, "
          f"{ms*1e3:8.1f} us, {n/ms/1e3:7.1f} Mrays/s, {steps/ms/1e6:6.1f} G steps/s", flush=True)
    r.close()
