#!/usr/bin/env python3
"""Random animations: a random scene, then a sequence of hrt_tlas_update calls with random transforms -- small steps (refits), now and then a jump across
the scene (the quality guard rebuilds), uneven scales and shears, an instance shrunk to nothing -- synchronous and asynchronous updates, flattened and
two-level trees, other execution modes; after every update the frame is compared bit for bit with the oracle's render of the moved scene.
    tools/stress_updates.py [n_scenes=20] [seed=1]"""
import importlib, os, sys, time, copy
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

def jitter(m, rng, big):
    m = m.reshape(3, 4).astype(np.float64).copy()
    ang = rng.normal(0, 1.5 if big else 0.05); ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
    lin = R @ m[:, :3]
    kind = rng.random()
    if kind < 0.1: lin = lin @ np.diag(rng.uniform(0.5, 1.8, 3))
    elif kind < 0.15: lin = lin @ np.array([[1, rng.uniform(-0.4, 0.4), 0], [0, 1, 0], [0, rng.uniform(-0.4, 0.4), 1]])
    elif kind < 0.17: lin = lin * 1e-9
    m[:, :3] = lin
    m[:, 3] += rng.normal(0, 1.0 if big else 0.02, 3)
    return m.astype(np.float32).reshape(12)

for k in range(n_scenes):
    w, h, spp = int(rng.integers(40, 120)), int(rng.integers(30, 90)), int(rng.integers(1, 4))
    kind = rng.integers(0, 3)
    if kind == 0: scene = hrt.scenes.mixed_test_scene(int(rng.integers(10, 2000)), int(rng.integers(1, 40)), int(rng.integers(1, 1000)), w, h, spp)
    elif kind == 1: scene = hrt.scenes.particle_cloud(int(rng.integers(4, 500)), w, h, spp, subdiv=int(rng.integers(0, 3)), seed=int(rng.integers(1, 100)))
    else: scene = hrt.scenes.particle_scene(int(rng.integers(1, 200)), w, h, spp, frame=int(rng.integers(0, 5)))
    flags = (hrt.CTX_TWO_LEVEL if rng.random() < 0.4 else 0) | (hrt.CTX_ASYNC_UPDATE if rng.random() < 0.5 else 0)
    mode = str(rng.choice(["1", "1", "1", "2", "0"]))
    if flags & hrt.CTX_TWO_LEVEL: mode = "1"
    os.environ["HRT_FUSED"] = mode
    salt = int(rng.integers(1, 1 << 30))
    scene = copy.deepcopy(scene)
    r = hrt.Renderer(0, flags)
    st = None
    try:
        r.load_scene(scene); r.set_frame(w, h, salt, linear=True)
        states = oracle.rng_init(w, h, salt)
        for step in range(int(rng.integers(2, 7))):
            big = rng.random() < 0.2
            tr = [jitter(it["transform"], rng, big) if rng.random() < 0.8 else it["transform"] for it in scene["instances"]]
            for it, m in zip(scene["instances"], tr): it["transform"] = m
            r.update_instances(tr)
            r.render(spp)
            got = r.linear.cpu().numpy().view(np.uint32)
            ref = oracle.OracleScene(scene, instanced=bool(flags & hrt.CTX_TWO_LEVEL)).render(w, h, states, spp)["linear"].view(np.uint32)
            if not np.array_equal(got, ref):
                bad += 1
                print("MISMATCH", scene["name"], (w, h, spp), "flags", flags, "HRT_FUSED", mode, "step", step, "big" if big else "", int((got != ref).sum()), "words", flush=True)
                break
        st = r.stats()
    except hrt.HrtError as e:
        bad += 1; print("ERROR", scene["name"], "flags", flags, "HRT_FUSED", mode, str(e)[:160], flush=True)
    finally:
        r.close()
    print(time.strftime("%H:%M:%S"), k, scene["name"], (w, h, spp), "flags", flags, "mode", mode, "refits", int(st.tlas_refits) if st else "-", "rebuilds", int(st.tlas_rebuilds) if st else "-", "ok" if not bad else f"{bad} bad so far", flush=True)
print("stress:", "all frames bit-exact" if not bad else f"{bad} FAILURES")
sys.exit(1 if bad else 0)
