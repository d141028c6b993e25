#!/usr/bin/env python3
"""Micro-benchmark of the traverse kernel alone through hrt_trace_rays: time vs number of rays."""
import importlib, sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
scene = hrt.scenes.soup_1m() if len(sys.argv) < 2 else hrt.scenes.random_soup(int(sys.argv[1]), 0.03, 1)
r = hrt.Renderer(0, hrt.CTX_TIMING)
r.load_scene(scene)
rng = np.random.default_rng(0)
for kind in ("inside", "primary"):
    for n in (1000, 50_000, 200_000, 800_000, 2_000_000):
        if kind == "inside":
            o = rng.uniform(-1, 1, (n, 3)).astype(np.float32); d = rng.normal(size=(n, 3)).astype(np.float32)
        else:
            o = np.tile(np.array([0, 0, 3.5], np.float32), (n, 1))
            d = np.stack([rng.uniform(-.33, .33, n), rng.uniform(-.19, .19, n), -np.ones(n)], 1).astype(np.float32)
        r.trace_rays(o, d)
        r.reset_stats()
        for _ in range(5):
            r.trace_rays(o, d)
        s = r.stats()
        ms = s.kernel_ms[hrt.K_TRAVERSE] / 5
        print(f"{kind:8s} n={n:8d}  {ms*1e3:9.1f} us/launch  {n/ms/1e3:8.1f} Mrays/s", flush=True)
