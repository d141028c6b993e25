#!/usr/bin/env python3
"""How full are the 8-wide nodes, and what would a narrower node cost?  For a soup scene: histogram of filled child slots per node of
the default tree, then the same scene built with at most W children per node (HRT_BVH_WIDTH = 8, 6, 5, 4: still stored in 80-byte
8-slot nodes, so only the VISITS change): nodes, node visits + primitive tests per ray (counting pass), Mrays/s at 16 spp.
Usage: tools/node_width.py [triangles=1000000] [--fast-trace]"""
import importlib, json, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
from test_gpu_parity import _download_tree
n_tri = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1_000_000
flags = hrt.CTX_FAST_TRACE if "--fast-trace" in sys.argv else 0
W, H, spp = 1920, 1080, 16
scene = hrt.scenes.soup_1m(W, H, spp) if n_tri == 1_000_000 else hrt.scenes.soup_large(n_tri, W, H, spp)
for width in (8, 6, 5, 4):
    os.environ["HRT_BVH_WIDTH"] = str(width)
    r = hrt.Renderer(0, flags)
    r.load_scene(scene)
    nodes, prims = _download_tree(hrt, r)
    meta = nodes.reshape(-1, 80)[:, 24:32]
    filled = (meta != 0).sum(axis=1)
    hist = np.bincount(filled, minlength=9)
    r.set_frame(W, H, hrt.scenes.SEED_SALT, aov=False)
    r.render(2); r.reset_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): r.render(spp, sync=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = r.stats()
    r.set_flags(hrt.CTX_COUNT | flags); r.reset_stats(); r.render(1); c = r.stats()
    print(json.dumps({"scene": scene["name"], "fast_trace": bool(flags), "max_children": width, "nodes": int(len(filled)), "mean_filled_slots": round(float(filled.mean()), 2),
                      "filled_slots_histogram_0_to_8": [int(x) for x in hist], "nodes_per_ray": round(c.node_visits_closest / max(c.rays_closest, 1), 2),
                      "prims_per_ray": round(c.prim_tests_closest / max(c.rays_closest, 1), 2), "slab_tests_per_ray_if_stored_W_wide": round(width * c.node_visits_closest / max(c.rays_closest, 1), 1),
                      "Mrays_per_s_in_80_byte_nodes": round(s.rays / dt / 1e6, 1)}), flush=True)
    r.close()
