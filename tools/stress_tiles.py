#!/usr/bin/env python3
"""Random tile splits: a random scene rendered as the N stripes-tiles of an N-rank job (one context per rank, as ranks have), with and without primary-hit reuse,
small and large sample counts (the cost-ordered probe launch, renders cut into several launches); the sum of the ranks' frames against the oracle's full frame.
    tools/stress_tiles.py [n_scenes=20] [seed=1]"""
import importlib, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
import oracle_py as oracle
n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for k in range(n_scenes):
    w, h = int(rng.integers(33, 300)), int(rng.integers(9, 200)); spp = int(rng.choice([1, 2, 5, 17, 33]))
    kind = rng.integers(0, 3)
    if kind == 0: scene = hrt.scenes.mixed_test_scene(int(rng.integers(10, 3000)), int(rng.integers(1, 40)), int(rng.integers(1, 1000)), w, h, spp)
    elif kind == 1: scene = hrt.scenes.particle_cloud(int(rng.integers(4, 300)), w, h, spp, seed=int(rng.integers(1, 100)))
    else: scene = hrt.scenes.cornell_box(w, h, spp)
    n_ranks = int(rng.integers(1, 7)); salt = int(rng.integers(1, 1 << 30))
    flags = (hrt.CTX_REUSE_PRIMARY if rng.random() < 0.5 else 0) | (hrt.CTX_TWO_LEVEL if rng.random() < 0.3 else 0)
    os.environ["HRT_FUSED_MAX_SPP"] = str(int(rng.choice([512, 7, 3]))); os.environ["HRT_FUSED_LPT"] = str(int(rng.choice([2, 0, 5])))
    total = None; rays = 0
    try:
        for rank in range(n_ranks):
            r = hrt.Renderer(0, flags)
            try:
                r.load_scene(scene); r.set_frame(w, h, salt, linear=True)
                r.linear.zero_()
                r.render(spp, tile=hrt.tile_for_rank(h, rank, n_ranks))
                lin = r.linear.cpu().numpy().copy()
                total = lin if total is None else total + lin
            finally:
                r.close()
        total[..., 3] = 1.0
        ref = oracle.OracleScene(scene, instanced=bool(flags & hrt.CTX_TWO_LEVEL)).render(w, h, oracle.rng_init(w, h, salt), spp)["linear"]
        ok = np.array_equal(total.view(np.uint32), ref.view(np.uint32))
    except hrt.HrtError as e:
        ok = False; print("ERROR", str(e)[:200], flush=True)
    if not ok:
        bad += 1; print("MISMATCH", scene["name"], (w, h, spp), "ranks", n_ranks, "flags", flags, os.environ["HRT_FUSED_MAX_SPP"], os.environ["HRT_FUSED_LPT"], flush=True)
    print(time.strftime("%H:%M:%S"), k, scene["name"], (w, h, spp), "ranks", n_ranks, "flags", flags, "ok" if not bad else f"{bad} bad so far", flush=True)
print("stress:", "all frames bit-exact" if not bad else f"{bad} FAILURES")
sys.exit(1 if bad else 0)
