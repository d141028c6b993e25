#!/usr/bin/env python3
"""Render only the rank-0-of-8 stripe tile (what one GPU does in the 8-GPU split), for profiling."""
import importlib, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
scene = hrt.scenes.soup_1m()
r = hrt.Renderer(0, hrt.CTX_TIMING)
r.load_scene(scene)
r.set_frame(1920, 1080, hrt.scenes.SEED_SALT, aov=False)
tile = hrt.tile_for_rank(1080, 0, n) if n > 1 else None
r.render(2, tile=tile); r.reset_stats()
torch.cuda.synchronize(); t0 = time.perf_counter()
r.render(spp, tile=tile)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
s = r.stats()
print(f"N={n}: {dt*1e3:.1f} ms for {spp} spp = {dt/spp*1e3:.3f} ms/sample; kernel ms:",
      {hrt.KERNEL_NAMES[k]: (round(s.kernel_ms[k], 2), int(s.kernel_launches[k])) for k in range(hrt.K_COUNT)})
