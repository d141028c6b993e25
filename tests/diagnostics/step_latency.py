#!/usr/bin/env python3
"""How long does ONE traversal step take?  Trace copies of one long ray (steps known from the CPU
walk of the same BVH bytes) in a single wave, then at full occupancy."""
import importlib, sys, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle_py as O
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
scene = hrt.scenes.soup_1m()
r = hrt.Renderer(0, hrt.CTX_TIMING)
r.load_scene(scene)
blob = hrt.BvhBlob(); r.lib.hrt_tlas_download(r.ctx, r.tlas, C.byref(blob))
rng = np.random.default_rng(1)
n = 100000
o = rng.uniform(-1, 1, (n, 3)).astype(np.float32); d = rng.normal(size=(n, 3)).astype(np.float32)
pr = np.zeros(n, np.uint32)
res = O.bvh8_trace(blob.nodes, blob.triangles, o, d, per_ray_nodes=pr)
for pick, name in ((int(np.argmax(pr)), "max"), (int(np.argsort(pr)[n // 2]), "median")):
    one = O.bvh8_trace(blob.nodes, blob.triangles, o[pick:pick + 1], d[pick:pick + 1])
    nodes, prims = one[5], one[6]
    for copies in (64, 256 * 1536, 4 * 256 * 1536):
        oo = np.repeat(o[pick:pick + 1], copies, 0); dd = np.repeat(d[pick:pick + 1], copies, 0)
        r.trace_rays(oo, dd); r.reset_stats()
        for _ in range(3):
            r.trace_rays(oo, dd)
        ms = r.stats().kernel_ms[hrt.K_TRAVERSE] / 3
        print(f"{name}: {nodes} node steps + {prims} prim tests, {copies} copies: {ms*1e3:.1f} us "
              f"-> {ms*1e3/ (nodes):.3f} us per node step (serial), {copies*(nodes+prims)/ms/1e6:.1f} G steps/s", flush=True)
