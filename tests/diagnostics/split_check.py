#!/usr/bin/env python3
"""Does tail splitting shorten a lone long ray?  Trace 1 / 8 / 64 copies of the longest ray of a sample."""
import importlib, sys, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle_py as O
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
scene = hrt.scenes.soup_1m()
r = hrt.Renderer(0, hrt.CTX_TIMING | hrt.CTX_COUNT)
r.load_scene(scene)
blob = hrt.BvhBlob(); r.lib.hrt_tlas_download(r.ctx, r.tlas, C.byref(blob))
rng = np.random.default_rng(1)
n = 100000
o = rng.uniform(-1, 1, (n, 3)).astype(np.float32); d = rng.normal(size=(n, 3)).astype(np.float32)
pr = np.zeros(n, np.uint32)
O.bvh8_trace(blob.nodes, blob.triangles, o, d, per_ray_nodes=pr)
pick = int(np.argmax(pr))
one = O.bvh8_trace(blob.nodes, blob.triangles, o[pick:pick + 1], d[pick:pick + 1])
print("longest ray:", one[5], "node steps,", one[6], "prim tests")
for copies in (1, 8, 64, 65, 4096):
    oo = np.repeat(o[pick:pick + 1], copies, 0); dd = np.repeat(d[pick:pick + 1], copies, 0)
    r.trace_rays(oo, dd); r.reset_stats()
    for _ in range(3):
        res = r.trace_rays(oo, dd)
    s = r.stats()
    ok = (res[3] == one[3][0]).all() and (res[0] == one[0][0]).all()
    print(f"{copies:5d} copies: {s.kernel_ms[hrt.K_TRAVERSE]/3*1e3:8.1f} us/launch, nodes/ray {s.node_visits/3/copies:7.1f}, prims/ray {s.prim_tests/3/copies:6.1f}, same hit: {ok}", flush=True)
