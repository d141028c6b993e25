#!/usr/bin/env python3
"""Single-GPU estimate of the strong-scaling tile split: render stripe tile `rank 0 of N` alone and
report its time; N GPUs would each do this concurrently (plus one 33 MB reduce)."""
import importlib, sys, time, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
scene = hrt.scenes.soup_1m()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
r = hrt.Renderer(0, 0)
r.load_scene(scene)
W, H = 1920, 1080
base = None
for n in [int(x) for x in os.environ.get("TILE_NS", "1,2,4,8").split(",")]:
    r.set_frame(W, H, hrt.scenes.SEED_SALT, aov=False)
    tile = hrt.tile_for_rank(H, 0, n) if n > 1 else None
    r.render(2, tile=tile)                       # warm-up
    r.reset_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r.render(spp, tile=tile)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = r.stats()
    base = base or dt
    print(f"N={n}: tile of rank 0 = {s.paths // spp} px, {dt*1e3:8.1f} ms for {spp} spp, {s.rays/dt/1e6:8.1f} Mrays/s per GPU, "
          f"projected {n*s.rays/dt/1e6:8.1f} Mrays/s, speed-up {base/dt:4.2f}x of ideal {n}x", flush=True)
