#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/.

Provenance (what each file pins):
  rocrand_xorwow_vectors.json  outputs of rocRAND's host-callable xorwow engine (ROCm 7.2 headers,
                               tests/helpers/rocrand_xorwow_ref.cpp): a THIRD-PARTY implementation of the
                               XORWOW recurrence + 2^67 sub-sequence skip-ahead.  Pins the oracle's GF(2) algebra.
  reference_kat.json           the one number the reference itself yielded in this container (SURVEY.md 8c probe:
                               colorToFloat4 of the miss colour), plus struct sizes measured there.
  pose_vectors.json            transforms of the Time-mode pose pipeline produced by oracle/oracle.c (slerp -> quatToEuler ->
                               constructTransformMatrix) for fixed particle states; pins the oracle against drift only.
  oracle_*.npz                 images / hit records produced by oracle/oracle.c itself.  They pin the oracle
                               against drift and let the GPU tests run without re-rendering; they are NOT
                               reference outputs (the reference cannot be built or run here: "parity unpinned").
Run from the repo root:  python tests/golden/make_golden.py
"""
import ctypes as C
import importlib
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_py as O  # noqa: E402

hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
G = ROOT / "tests" / "golden"


def rocrand_vectors():
    so = ROOT / "tests" / "helpers" / "librocrand_ref.so"
    if not so.exists():
        subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                               str(so.with_name("rocrand_xorwow_ref.cpp")), "-o", str(so)])
    R = C.CDLL(str(so))
    R.rocrand_xorwow_draw.argtypes = [C.c_ulonglong, C.c_ulonglong, C.c_ulonglong, C.c_void_p, C.c_int]
    out = []
    for seed, sub in ((0, 0), (1234, 1), (99, 2), (7, 3), (0x5EED0000C0FFEE, 12345), (1, 0xFFFFF), (5, 2073599),
                      (0xFFFFFFFFFFFFFFFF, 0x7FFFFFFF), (42, 65536)):
        ref = np.zeros(8, np.uint32)
        R.rocrand_xorwow_draw(seed, sub, 0, ref.ctypes.data, 8)
        out.append({"seed": seed, "subsequence": sub, "draws": [int(x) for x in ref]})
    (G / "rocrand_xorwow_vectors.json").write_text(json.dumps({
        "source": "rocrand_device::xorwow_engine(seed, subsequence, 0), /opt/rocm/include/rocrand/rocrand_xorwow.h (ROCm 7.2)",
        "scramble_constants": {"s0_xor": 0x2c7f967f, "s1_xor": 0xa03697cb, "t0_mul": 1228688033, "t1_mul": 2073658381},
        "vectors": out}, indent=1))


def oracle_images():
    for name, scene in (("c1_64", hrt.scenes.cornell_box(64, 64, 1)),
                        ("c2_48_spp4", hrt.scenes.sphere_in_box(48, 48, 4)),
                        ("mixed_61x37_spp2", hrt.scenes.mixed_test_scene(600, 16, 7, 61, 37, 2))):
        W, H, spp = scene["width"], scene["height"], scene["spp"]
        osc = O.OracleScene(scene)
        st = O.rng_init(W, H, hrt.scenes.SEED_SALT)
        r = osc.render(W, H, st, spp)
        np.savez_compressed(G / f"oracle_{name}.npz", linear=r["linear"], color=r["color"], rays=np.uint64(r["rays"]),
                            states_tail=st[-64:], width=W, height=H, spp=spp)
    # hit records of the canonical intersector on a small mixed scene (brute force)
    scene = hrt.scenes.mixed_test_scene(600, 16, 7)
    o, d = O.random_rays(4000, 3)
    t, u, v, prim, inst = O.OracleScene(scene, force_brute=True).trace(o, d)
    np.savez_compressed(G / "oracle_hits_mixed.npz", t=t, u=u, v=v, prim=prim, inst=inst)


def pose_vectors():
    cur, nxt = hrt.scenes.particle_states(8, 0), hrt.scenes.particle_states(8, 1)
    cases = []
    for dur, frame, count, off, sc in ((0.5, 0, 120, (0, 0, 0), (1, 1, 1)), (0.5, 77, 120, (0, 0, 0), (1, 1, 1)),
                                       (0.25, 119, 120, (0.5, -0.25, 1.0), (1.5, 1.5, 1.5)), (1.0, 0, 1, (0, 0, 0), (1, 2, 3))):
        out = O.pose_transforms(cur, nxt, dur, frame, count, off, sc)
        cases.append({"duration": dur, "frame": frame, "frame_count": count, "offset": list(off), "scale": list(sc),
                      "transforms_hex": [int(x) for x in out.view(np.uint32).reshape(-1)]})
    (G / "pose_vectors.json").write_text(json.dumps({
        "source": "oracle/oracle.c: oracle_pose_transforms on nvidia-optix-ray-tracer_amd.scenes.particle_states(8, 0 / 1)",
        "current": [[float(x) for x in row] for row in cur], "next": [[float(x) for x in row] for row in nxt],
        "cases": cases}, indent=0))


if __name__ == "__main__":
    rocrand_vectors()
    pose_vectors()
    (G / "reference_kat.json").write_text(json.dumps({
        "source": "SURVEY.md 8(c): the reference's own colorToFloat4 (include/Global/DeviceFunctions.cuh:188-209) evaluated "
                  "in the survey container on the miss colour (0.7, 0.8, 0.9) of src/Global/RendererMesh.cu:262, and sizeof() "
                  "of the reference structs",
        "colorToFloat4_background_hex": ["0x1.b56792p-1", "0x1.d00ab6p-1", "0x1.e8ccbep-1", "0x1p+0"],
        "sizeof": {"GlobalParams": 16, "RayGenParams": 80, "MissParams": 12, "HitGroupParams": 32}}, indent=1))
    oracle_images()
    print("golden fixtures written to", G)
