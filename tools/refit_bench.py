#!/usr/bin/env python3
"""Per-frame updateIAS cost: device refit (hrt_tlas_update) vs a host rebuild, on one MI355X.

Prints one JSON line per scene: wall time of hrt_tlas_update, the refit kernels' HIP-event time, the
algorithmic bytes they move (per node 80 B read + 80 B written + 24 B box written + 24 B box read by the
parent; per triangle 36 B source + 48 B record written) and the resulting GB/s against the 8 TB/s HBM peak.
Usage: python tools/refit_bench.py [--frames 20]"""
import argparse, importlib, json, os, sys, time
from pathlib import Path
import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")


def run(name, scene, poses_of, frames, mode):
    """mode: "refit" (device refit), "rebuild" (HRT_REFIT=0: tree over instances, the default for rebuilds during updates)
    or "rebuild-merged" (HRT_TLAS_INSTANCED=-1: full SAH build over all primitives)."""
    refit = mode == "refit"
    os.environ["HRT_REFIT"] = "1" if refit else "0"
    os.environ["HRT_TLAS_INSTANCED"] = "-1" if mode == "rebuild-merged" else "0"
    import ctypes as C
    import torch
    r = hrt.Renderer(0, hrt.CTX_TIMING)
    r.load_scene(scene)
    r.set_frame(64, 64, 1, aov=False)
    r.render(1)
    s0 = r.stats(reset=True)
    n_nodes, n_tris = s0.bvh_nodes, s0.bvh_triangles

    def instances_on_device(f):                             # what the reference's cudaMemcpy of pin_instances leaves behind
        for i, m in enumerate(poses_of(f)):
            for k in range(12):
                r._h_inst[i].transform[k] = float(m[k])
        return r._dev(np.frombuffer(bytes(r._h_inst), dtype=np.uint8).copy())

    bufs = [instances_on_device(f) for f in range(frames + 1)]
    n = len(scene["instances"])
    st = r._stream()
    r._check(r.lib.hrt_tlas_update(r.ctx, r.tlas, bufs[0].data_ptr(), n, st), "hrt_tlas_update")      # warm-up
    torch.cuda.synchronize()
    r.reset_stats()
    t0 = time.perf_counter()
    for f in range(1, frames + 1):
        r._check(r.lib.hrt_tlas_update(r.ctx, r.tlas, bufs[f].data_ptr(), n, st), "hrt_tlas_update")
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    s = r.stats()
    out = {"scene": name, "mode": mode, "instances": n,
           "bvh_nodes": int(n_nodes), "triangles": int(n_tris), "frames": frames,
           "update_call_ms": round((t1 - t0) * 1e3 / frames, 4), "update_done_ms": round((t2 - t0) * 1e3 / frames, 4),
           "tlas_refits": int(s.tlas_refits), "tlas_rebuilds": int(s.tlas_rebuilds), "refit_ratio": round(s.tlas_refit_ratio, 3)}
    if refit and s.kernel_launches[hrt.K_REFIT]:
        k_ms = s.kernel_ms[hrt.K_REFIT] / s.kernel_launches[hrt.K_REFIT]
        bytes_ = n_nodes * (80 + 80 + 24 + 24) + n_tris * (36 + 48)
        out.update({"refit_kernels_ms": round(k_ms, 4), "algorithmic_bytes": int(bytes_),
                    "achieved_GBps": round(bytes_ / (k_ms * 1e-3) / 1e9, 1), "hbm_peak_GBps": 8000})
    r.close()
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--only", default="", help="one case, e.g. particles-25:rebuild-merged (with --frames rebuilds: for API-level profiles)")
    a = ap.parse_args()
    if a.only:
        name, mode = a.only.split(":")
        n_p, sub = (25, 2) if name == "particles-25" else (2000, 3)
        sc = hrt.scenes.particle_scene(n_p, 1200, 800, 1, 0, subdiv=sub)
        ground = sc["instances"][0]["transform"]
        cache = {f: [ground] + hrt.scenes.particle_poses(n_p, f) for f in range(a.frames + 1)}
        run(name, sc, lambda f: cache[f], a.frames, mode)
        return
    # the reference's Time-mode structure: particles instancing shared shapes + ground sphere
    for n_p, sub in ((25, 2), (2000, 3)):
        sc = hrt.scenes.particle_scene(n_p, 1200, 800, 1, 0, subdiv=sub)
        ground = sc["instances"][0]["transform"]
        poses = lambda f, n_p=n_p, ground=ground: [ground] + hrt.scenes.particle_poses(n_p, f)   # noqa: E731
        cache = {}
        cached = lambda f, poses=poses, cache=cache: cache.setdefault(f, poses(f))               # noqa: E731
        for f in range(a.frames + 1):
            cached(f)
        run("particles-%d" % n_p, sc, cached, a.frames, "refit")
        run("particles-%d" % n_p, sc, cached, min(a.frames, 5), "rebuild")
        run("particles-%d" % n_p, sc, cached, min(a.frames, 5), "rebuild-merged")
    # one big static instance that moves as a whole (C4 geometry)
    sc = hrt.scenes.soup_1m(1920, 1080, 1)
    base = sc["instances"][0]["transform"]
    def poses(f, base=base):
        m = base.copy(); m[3] += 0.01 * f
        return [m]
    run("soup-1m", sc, poses, a.frames, "refit")
    run("soup-1m", sc, poses, 1, "rebuild")


if __name__ == "__main__":
    main()
