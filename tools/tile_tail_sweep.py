#!/usr/bin/env python3
"""The stripe tile of rank 0 of N (C4, 256 spp) rendered alone on one GPU, per knob setting: what the slowest pixels' sample
chains cost at the end of a small tile, and what tail splitting / the wave count / the regeneration rule do about it.
Usage: tools/tile_tail_sweep.py "N,N,..." "VAR=a,b" ["VAR2=x,y" ...]        (the scene is generated once)"""
import importlib, itertools, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
ns = [int(x) for x in sys.argv[1].split(",")]
specs = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[2:]]
scene = hrt.scenes.soup_1m()
W, H, SPP = 1920, 1080, int(os.environ.get("TILE_SPP", "256"))
for combo in itertools.product(*[v for _, v in specs]):
    for (k, _), v in zip(specs, combo):
        os.environ[k] = v
    r = hrt.Renderer(0, 0)
    r.load_scene(scene)
    out = []
    for n in ns:
        r.set_frame(W, H, hrt.scenes.SEED_SALT, aov=False)
        tile = hrt.tile_for_rank(H, 0, n) if n > 1 else None
        r.render(2, tile=tile)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.render(SPP, tile=tile)
        torch.cuda.synchronize(); out.append("N=%d %.1f ms" % (n, (time.perf_counter() - t0) * 1e3))
    r.close()
    print(" ".join("%s=%s" % (k, v) for (k, _), v in zip(specs, combo)), ":", "  ".join(out), flush=True)
