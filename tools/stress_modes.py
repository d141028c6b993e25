#!/usr/bin/env python3
"""Random small scenes through every combination of tree structure (flattened / two-level), primary-hit reuse and leaf-hold value, frames
compared bit for bit with the oracle (its FLATTENED / INSTANCED mode).  A one-off stress run, not part of the suites.
    tools/stress_modes.py [n_scenes=24] [seed=1]"""
import importlib, itertools, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
import oracle_py as oracle

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for k in range(n_scenes):
    kind = rng.integers(0, 8)
    w, h, spp = int(rng.integers(40, 160)), int(rng.integers(30, 120)), int(rng.integers(1, 7))
    if os.environ.get("STRESS_BIG"): w, h, spp = int(rng.integers(300, 800)), int(rng.integers(200, 500)), int(rng.integers(1, 4))      # (fewer, larger frames)
    if kind == 0: scene = hrt.scenes.mixed_test_scene(int(rng.integers(10, 4000)), int(rng.integers(1, 60)), int(rng.integers(1, 1000)), w, h, spp)
    elif kind == 1: scene = hrt.scenes.particle_cloud(int(rng.integers(4, 900)), w, h, spp, subdiv=int(rng.integers(0, 3)), seed=int(rng.integers(1, 100)))
    elif kind == 2: scene = hrt.scenes.particle_scene(int(rng.integers(1, 300)), w, h, spp, frame=int(rng.integers(0, 5)))
    elif kind == 3: scene = hrt.scenes.cornell_box(w, h, spp)
    elif kind == 4: scene = hrt.scenes.random_soup(int(rng.integers(1, 30000)), float(rng.uniform(0.01, 0.3)), int(rng.integers(1, 1000)), w, h, spp)
    elif kind == 5: scene = hrt.scenes.sphere_in_box(w, h, spp)
    elif kind == 6:
        n_chain = int(rng.integers(3, 400))
        scene = hrt.scenes.growing_chain(n_chain, float(min(rng.uniform(1.01, 1.3), 1e12 ** (1.0 / n_chain))), w, h, spp)      # (sizes stay finite floats)
    else:
        # a room of mirrors: axis-aligned metal walls without fuzz send rays along the axes, into edges and corners, with components of +-0.0
        scene = hrt.scenes.cornell_box(w, h, spp) if rng.random() < 0.5 else hrt.scenes.sphere_in_box(w, h, spp)
        scene["name"] += "-mirrors"
        for it in scene["instances"]:
            if rng.random() < 0.7: it["material"], it["fuzz"] = "metal", float(rng.choice([0.0, 0.0, 0.05]))
    salt = int(rng.integers(1, 1 << 30))
    refs = {}
    combos = list(itertools.product((0, 1), (0, 1), ("", "1", "3")))
    if os.environ.get("STRESS_OTHER_MODES"):      # the other execution modes (flattened trees): wavefront pipeline, round 1's path kernel, the counting build
        combos = [(0, 0, "fused0"), (0, 0, "fused2"), (0, 0, "count"), (0, 0, "fused-1")]
    for two, reuse, hold in combos:
        ctx_flags = 0
        os.environ.pop("HRT_FUSED", None)
        if hold.startswith("fused"): os.environ["HRT_FUSED"] = hold[5:]; hold = ""
        elif hold == "count": ctx_flags = hrt.CTX_COUNT; hold = ""
        if hold: os.environ["HRT_LEAF_HOLD"] = hold
        else: os.environ.pop("HRT_LEAF_HOLD", None)
        os.environ["HRT_REFILL_THRESHOLD"] = str(int(rng.integers(1, 64)))
        r = hrt.Renderer(0, (hrt.CTX_TWO_LEVEL if two else 0) | (hrt.CTX_REUSE_PRIMARY if reuse else 0) | ctx_flags)
        try:
            r.load_scene(scene); r.set_frame(w, h, salt, linear=True); r.render(spp)
            got = r.linear.cpu().numpy().view(np.uint32).copy()
        except hrt.HrtError as e:
            bad += 1
            print("ERROR", scene["name"], (w, h, spp), "two" if two else "flat", str(e)[:160], flush=True)
            continue
        finally:
            r.close()
        if two not in refs:
            refs[two] = oracle.OracleScene(scene, instanced=bool(two)).render(w, h, oracle.rng_init(w, h, salt), spp)["linear"].view(np.uint32)
        ok = np.array_equal(got, refs[two])
        if not ok:
            bad += 1
            print("MISMATCH", scene["name"], (w, h, spp), "two" if two else "flat", "reuse" if reuse else "", "hold", hold or "auto", int((got != refs[two]).sum()), "words", flush=True)
    print(time.strftime("%H:%M:%S"), k, scene["name"], (w, h, spp), "ok" if not bad else f"{bad} mismatches so far", flush=True)
print("stress:", "all frames bit-exact" if not bad else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
