#!/usr/bin/env python3
"""Dump the transform nodes of a small two-level tree and compare the kernel's hits with the CPU walk of the same bytes."""
import importlib, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
def say(*a): print(time.strftime("%H:%M:%S"), *a, flush=True)
import torch
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
import oracle_py as O
from test_gpu_parity import _download_tree
r = hrt.Renderer(0, hrt.CTX_TWO_LEVEL)
scene = hrt.scenes.particle_scene(12, 64, 48, 1)
r.load_scene(scene); torch.cuda.synchronize()
nodes, prims = _download_tree(hrt, r)
w = np.frombuffer(nodes.tobytes(), dtype=np.uint32).reshape(-1, 20)
f = w.view(np.float32)
say("nodes", len(w))
for i in range(len(w)):
    if w[i, 3] == 0:
        say("xform node", i, "centre", f[i, 0:3], "radius", f[i, 7], "root", w[i, 4], "inst", w[i, 5], "identity", w[i, 6], "inv row0", f[i, 8:12])
o, d = O.random_rays(2000, 5, 1.6)
inv = np.stack([np.linalg.inv(np.vstack([it["transform"].reshape(3, 4).astype(np.float64), [0, 0, 0, 1]]))[:3].reshape(12) for it in scene["instances"]]).astype(np.float32)
ident = np.array([int(np.array_equal(it["transform"], hrt.scenes.IDENTITY)) for it in scene["instances"]], dtype=np.uint32)
res = O.bvh8_trace(nodes.ctypes.data, prims.ctypes.data, o, d, inst_inv=inv, inst_identity=ident)
say("cpu walk hits", int((res[3] != 0xffffffff).sum()))
got = r.trace_rays(o, d); torch.cuda.synchronize()
say("kernel hits", int((got[3] != 0xffffffff).sum()))
rt = O.OracleScene(scene, force_brute=True, instanced=True).trace(o, d)
say("brute force hits", int((rt[3] != 0xffffffff).sum()))
r.close()
