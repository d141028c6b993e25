#!/usr/bin/env python3
"""Mesh-mode frame loop through the C++ driver on a synthetic data set in the reference's on-disk formats (the reference ships no
Mesh-mode sample): tools/mesh_mode_bench.py [n_files n_particles width height].  Prints the driver's own lines: cache loading with
loader threads (one device GAS build per particle, one IAS per file), then ms per frame of pose kernel -> updateIAS -> launch -> 8 bit."""
import importlib, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
io = importlib.import_module("nvidia-optix-ray-tracer_amd.io")
n_files, n_particles, w, h = (int(x) for x in (sys.argv[1:5] + ["6", "2000", "1200", "800"][len(sys.argv) - 1:]))
with tempfile.TemporaryDirectory() as tmp:
    cfg = io.write_mesh_mode_sample(tmp, n_files=n_files, n_particles=n_particles, width=w, height=h)
    for env in ({}, {"HRT_MESH_RENDER_SYNC_UPDATE": "1"}):
        print("==", env or "asynchronous updates (default)", flush=True)
        p = subprocess.run([str(ROOT / "nvidia-optix-ray-tracer_amd" / "lib" / "hrt_mesh_render"), str(cfg), str(Path(tmp) / "bin"), "-1", str(Path(tmp) / "f.ppm")],
                           capture_output=True, text=True, env={**__import__("os").environ, **env})
        print("\n".join(l for l in p.stdout.splitlines() if not l.startswith("[")), p.stderr[-500:], flush=True)
