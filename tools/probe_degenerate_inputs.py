#!/usr/bin/env python3
"""Builds that stress the device builder (default and HRT_CTX_FAST_TRACE) on odd inputs -- a planar grid, a line of tiny triangles, coordinates
of 1e5, thousands of points, sizes around the small-build limit, huge triangles among small ones, thousands of spheres -- and compares 2500 rays' hits
with brute force.  Prints one line per case; exits 1 when anything differs.  (GPU; the cases that earned a test are in tests/test_gpu_parity.py.)"""
import importlib, sys, traceback
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle_py as oracle
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
rng = np.random.default_rng(1)

def soup(n, edge, seed):
    return hrt.scenes.random_soup(n, edge, seed, 96, 64, 1)

def cases():
    s = soup(10000, 0.05, 1); v = s["instances"][0]["vertices"]
    g = np.stack(np.meshgrid(np.arange(100), np.arange(100), indexing="ij"), -1).reshape(-1, 2).astype(np.float32) * 0.02 - 1
    v[:, 0, :] = np.c_[g, np.zeros(len(g))]; v[:, 1, :] = v[:, 0, :] + [0.02, 0, 0]; v[:, 2, :] = v[:, 0, :] + [0, 0.02, 0]
    yield "planar grid", s
    s = soup(9000, 0.05, 2); v = s["instances"][0]["vertices"]
    x = np.linspace(-1, 1, len(v), dtype=np.float32)
    v[:, 0, :] = np.c_[x, 0 * x, 0 * x]; v[:, 1, :] = v[:, 0, :] + [1e-3, 0, 0]; v[:, 2, :] = v[:, 0, :] + [0, 1e-3, 0]
    yield "a line of tiny triangles", s
    s = soup(8000, 0.05, 3); s["instances"][0]["vertices"] *= np.float32(1e5)
    yield "coordinates of 1e5", s
    s = soup(9000, 0.05, 4); s["instances"][0]["vertices"][:5000] = 0.25
    yield "5000 points", s
    for n in (4096, 4097, 4100):
        yield f"n = {n}", soup(n, 0.08, 5)
    s = soup(9000, 0.03, 6); v = s["instances"][0]["vertices"]
    v[:600] = (rng.uniform(-1, 1, (600, 3, 3)) * 3).astype(np.float32)
    yield "600 huge triangles among small ones", s
    s = hrt.scenes.mixed_test_scene(5000, 6000, 9, 96, 64, 1)
    yield "6000 spheres and 5000 triangles", s

bad = 0
for name, scene in cases():
    for inst in scene["instances"]:
        if inst["geometry"] == "triangles":
            inst["normals"] = hrt.scenes.face_normals(inst["vertices"])
    o, d = oracle.random_rays(2500, 11)
    if "1e5" in name:
        o *= np.float32(1e5)
    try:
        want = oracle.OracleScene(scene, force_brute=True).trace(o, d)
    except Exception as e:
        print(name, "oracle failed", e); continue
    for flags, fn in ((0, "default"), (hrt.CTX_FAST_TRACE, "fast-trace")):
        try:
            r = hrt.Renderer(0, flags)
            r.load_scene(scene)
            t, u, vv, prim, inst = r.trace_rays(o, d)
            ok = np.array_equal(prim, want[3]) and np.array_equal(inst, want[4]) and np.array_equal(t.view(np.uint32), want[0].view(np.uint32))
            r.set_frame(96, 64, 3, aov=False); r.render(1); st = r.stats()
            print(f"{name:40s} {fn:10s} hits {int((prim != 0xFFFFFFFF).sum()):5d} parity {ok} nodes {st.bvh_nodes} depth {st.bvh_depth}", flush=True)
            bad += 0 if ok else 1
            r.close()
        except Exception as e:
            bad += 1
            print(f"{name:40s} {fn:10s} FAILED: {str(e)[:200]}", flush=True)
print("problems:", bad)
sys.exit(1 if bad else 0)
