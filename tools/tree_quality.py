#!/usr/bin/env python3
"""Tree quality without a GPU: node visits / primitive tests per ray of a host-built BVH8, counted by the CPU walker of the
test oracle over a realistic ray set -- the rays of path-traced samples of the scene (primary rays of a pixel subset and their
rough bounces up to depth 5, generated here with numpy random numbers: only the distribution matters).
    python tools/tree_quality.py [n_triangles=1000000] [n_pixels=60000] [--empty]
Environment knobs of the host builder (csrc/bvh8_build.cpp) select the variant; prints one line."""
import ctypes as C, importlib, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_py
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")


def path_rays(scene, n_pixels, seed=1):
    """(origins, directions) of all rays of one sample of n_pixels random pixels, depth <= 5, rough bounces."""
    rng = np.random.default_rng(seed)
    W, H = scene["width"], scene["height"]
    cam = scene["camera"]
    u, v, w = hrt.configure_camera(cam["center"], cam["target"], cam["up"], cam.get("opengl", True))
    px = rng.integers(0, W, n_pixels); py = rng.integers(0, H, n_pixels)
    ndcx = ((px + 0.5) / W * 2 - 1) * (W / H); ndcy = (py + 0.5) / H * 2 - 1
    d = ndcx[:, None] * u[None] + ndcy[:, None] * v[None] + w[None]
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o = np.repeat(np.asarray(cam["center"], np.float32)[None], n_pixels, 0)
    osc = oracle_py.OracleScene(scene)
    normals = scene["instances"][0]["normals"][:, 0, :]
    all_o, all_d = [], []
    for depth in range(1, 6):
        all_o.append(o); all_d.append(d)
        t, uu, vv, prim, inst = osc.trace(o, d)
        hit = prim != 0xFFFFFFFF
        if depth == 5 or not hit.any():
            break
        o, d, t, prim = o[hit], d[hit], t[hit], prim[hit]
        p = (o + t[:, None] * d).astype(np.float32)
        n = normals[prim].copy()
        flip = (n * d).sum(1) >= 0
        n[flip] = -n[flip]
        r = rng.uniform(-1, 1, (len(p), 3))
        r = r / np.linalg.norm(r, axis=1, keepdims=True)
        d = (n + r).astype(np.float32)
        o = p
    return np.concatenate(all_o), np.concatenate(all_d)


def build(verts):
    lib = hrt.load_library()
    v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3, 3)
    blob = hrt.BvhBlob()
    t0 = time.time()
    rc = lib.hrt_host_build_bvh8(v.ctypes.data, v.shape[0], C.byref(blob))
    assert rc == 0, lib.hrt_last_error(None)
    return lib, blob, time.time() - t0


if __name__ == "__main__":
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_tri = int(pos[0]) if len(pos) > 0 else 1_000_000
    n_px = int(pos[1]) if len(pos) > 1 else 60000
    scene = hrt.scenes.soup_1m(1920, 1080, 1) if n_tri == 1_000_000 else hrt.scenes.random_soup(n_tri, hrt.scenes.soup_law_edge(n_tri), 1, 1920, 1080, 1)
    cache = Path("/tmp") / f"tq_rays_{n_tri}_{n_px}.npz"
    if cache.exists():
        z = np.load(cache); o, d = z["o"], z["d"]
    else:
        o, d = path_rays(scene, n_px); np.savez(cache, o=o, d=d)
    lib, blob, secs = build(scene["instances"][0]["vertices"])
    res = oracle_py.bvh8_trace(blob.nodes, blob.triangles, o, d)
    nodes, prims = res[5], res[6]
    hits = int((res[3] != 0xFFFFFFFF).sum())
    knobs = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("HRT_"))
    print(f"{knobs or 'default':60s} build {secs:6.2f}s nodes {blob.n_nodes:8d} refs {blob.n_triangles:8d} | rays {len(o)} hits {hits} | "
          f"nodes/ray {nodes / len(o):6.3f} prims/ray {prims / len(o):6.3f} | est instr/ray {213 * nodes / len(o) + 83 * prims / len(o):7.1f}")
    if "--line-mates" in sys.argv:
        # what a 64-byte node would save: the L2s fetch 128-byte lines, so two 64-byte nodes share one, and a visit costs no new
        # line when the same ray has been at the node's neighbour (the ray's own reuse only: an upper bound on the lines of a ray)
        m = oracle_py.bvh8_trace.last_line_mates
        print(f"  node visits {nodes / len(o):.2f} per ray, of which {m / len(o):.2f} ({100.0 * m / nodes:.1f} %) at the array neighbour of a node the ray visited before")
    if "--empty" in sys.argv:
        # what the visits are made of: visits that find nothing to enter or test, for miss rays, hit rays, and hit rays culled with
        # their final hit distance from the start (the best any traversal order could do)
        t, prim = res[0], res[3]
        hit = prim != 0xFFFFFFFF
        def run(oo, dd, **kw):
            r = oracle_py.bvh8_trace(blob.nodes, blob.triangles, oo, dd, **kw)
            return r[5] / len(oo), oracle_py.bvh8_trace.last_empty_visits / len(oo)
        na, ea = run(o, d); nm, em = run(o[~hit], d[~hit]); nh, eh = run(o[hit], d[hit])
        d2 = (d[hit] * (t[hit] * np.float32(1.00001))[:, None]).astype(np.float32)
        nb, eb = run(o[hit], d2, tmax=1.0)
        print(f"  empty visits: all rays {ea:.3f} of {na:.3f} | miss rays {em:.3f} of {nm:.3f} | hit rays {eh:.3f} of {nh:.3f} | hit rays culled with t_hit from the start {eb:.3f} of {nb:.3f}")
    lib.hrt_host_free(C.byref(blob))
