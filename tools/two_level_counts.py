#!/usr/bin/env python3
"""Canonical (CPU walk of the downloaded tree: every leaf tested when its node is visited, nearest child first) against executed (the path
kernel's counters, make stats) node steps and primitive tests per ray, for the flattened and the two-level tree of one particle cloud
and the same rays.  HRT_LIB=.../libhrt_stats.so tools/two_level_counts.py [n_particles] [subdiv]"""
import importlib, sys, os, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
import oracle_py as O
from test_gpu_parity import _download_tree
n_p = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
subdiv = int(sys.argv[2]) if len(sys.argv) > 2 else 3
scene = hrt.scenes.particle_cloud(n_p, 64, 64, 1, subdiv=subdiv)
ext = 0.2 * int(np.ceil(n_p ** (1 / 3)))
rng = np.random.default_rng(3)
n = 200000
o = rng.uniform(-0.6 * ext, 0.6 * ext, size=(n, 3)).astype(np.float32); o[:, 2] = np.abs(o[:, 2]) + 0.05
d = rng.normal(size=(n, 3)).astype(np.float32)
for structure in ("flat", "two"):
    os.environ["HRT_TWO_LEVEL"] = "-1" if structure == "flat" else "0"
    r = hrt.Renderer(0, hrt.CTX_TWO_LEVEL if structure == "two" else 0)
    r.load_scene(scene)
    nodes, prims = _download_tree(hrt, r)
    inv = np.stack([np.linalg.inv(np.vstack([it["transform"].reshape(3, 4).astype(np.float64), [0, 0, 0, 1]]))[:3].reshape(12) for it in scene["instances"]]).astype(np.float32)
    ident = np.array([int(np.array_equal(it["transform"], hrt.scenes.IDENTITY)) for it in scene["instances"]], dtype=np.uint32)
    res = O.bvh8_trace(nodes.ctypes.data, prims.ctypes.data, o, d, inst_inv=inv, inst_identity=ident)
    extra = ""
    if structure == "two":      # the same walk with the bounding-sphere test at the transform nodes switched off (radius < 0: none)
        w = np.frombuffer(nodes.tobytes(), dtype=np.uint32).reshape(-1, 20).copy()
        w[w[:, 3] == 0, 7] = np.float32(-1.0).view(np.uint32)
        off = O.bvh8_trace(w.ctypes.data, prims.ctypes.data, o, d, inst_inv=inv, inst_identity=ident)
        extra = f" (without the bounding-sphere test at the transform nodes: {off[5] / n:.2f} + {off[6] / n:.2f}; same hits: {bool(np.array_equal(off[3], res[3]))})"
    r.reset_stats()
    got = r.trace_rays(o, d)
    s = r.stats()
    line = f"{scene['name']} {structure}: canonical {res[5] / n:.2f} node visits + {res[6] / n:.2f} primitive tests per ray{extra}; hit fraction {(res[3] != 0xffffffff).mean():.2f}"
    if "stats" in os.environ.get("HRT_LIB", ""):
        line += f"; executed {s.debug[2] / n:.2f} node steps + {s.debug[3] / n:.2f} primitive tests in {s.debug[0] * 64 / n:.1f} lane iterations per ray, {(s.node_visits - s.node_visits_closest) / n:.2f} instances entered"
    print(line, "; same primitives as the CPU walk:", bool(np.array_equal(got[3], res[3])), flush=True)
    r.close()
