"""CPU tests of the product's host side: the C-ABI library loads and exports what include/hrt.h
declares, struct layouts match the reference's, there is no CPU fallback, the host BVH8 builder
produces a valid tree whose CPU walk (oracle-side walker, test infrastructure) equals brute force."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol(hrt):
    lib = hrt.load_library()
    header = (ROOT / "include" / "hrt.h").read_text()
    declared = sorted(set(re.findall(r"\b(hrt_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations found in include/hrt.h"
    for name in declared:
        assert hasattr(lib, name), f"libhrt.so does not export {name}"
    assert sorted(hrt.EXPORTS) == declared
    assert b"gfx950" in lib.hrt_version()


def test_struct_layouts_match_reference(hrt):
    assert C.sizeof(hrt.GlobalParams) == 16
    assert C.sizeof(hrt.RayGenParams) == 80
    assert [getattr(hrt.RayGenParams, f).offset for f in ("width", "height", "colorBuffer", "albedoBuffer", "normalBuffer",
                                                           "cameraCenter", "cameraU", "cameraV", "cameraW")] == [0, 4, 8, 16, 24, 32, 44, 56, 68]
    assert C.sizeof(hrt.MissParams) == 12 and C.sizeof(hrt.HitGroupParams) == 32
    assert hrt.HitGroupParams.albedo.offset == 16 and hrt.HitGroupParams.fuzz.offset == 28
    assert C.sizeof(hrt.SbtRecord) == 64 and hrt.SbtRecord.data.offset == 32          # OPTIX_SBT_RECORD_HEADER_SIZE
    assert C.sizeof(hrt.Instance) == 80 and hrt.Instance.sbtOffset.offset == 52 and hrt.Instance.traversableHandle.offset == 64


def test_no_cpu_fallback(hrt, gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    lib = hrt.load_library()
    ctx = C.c_void_p()
    assert lib.hrt_ctx_create(0, 0, C.byref(ctx)) == -2            # HRT_ERR_NO_DEVICE
    assert not ctx.value and b"no CPU path" in lib.hrt_last_error(None)
    with pytest.raises(hrt.HrtError):
        hrt.Renderer(0)


def test_tuning_knobs_come_from_one_list():
    """csrc/knobs.def is THE list of environment knobs: hrt_ctx_create parses it, INTEGRATION.md's table is generated from it
    (tools/knob_table.py --write), and no library source reads an HRT_* variable that is not in it."""
    import sys
    sys.path.insert(0, str(ROOT / "tools"))
    import knob_table
    names = {k[0] for k in knob_table.knobs() if k[0]}
    assert len(names) >= 30 and len(names) == sum(1 for k in knob_table.knobs() if k[0]), "duplicate knob"
    assert knob_table.current((ROOT / "INTEGRATION.md").read_text()) == knob_table.table(), "run tools/knob_table.py --write"
    csrc = ROOT / "nvidia-optix-ray-tracer_amd" / "csrc"
    read = set()
    sites = 0
    for f in list(csrc.glob("*.cpp")) + list(csrc.glob("*.hip")) + list(csrc.glob("*.h")) + list(csrc.glob("*.hpp")):
        text = f.read_text()
        sites += text.count("getenv(")
        read |= set(re.findall(r'getenv\("(HRT_[A-Z0-9_]+)"\)', text))
    assert read <= names, sorted(read - names)
    assert sites <= 12, sites                        # (round 3: 59 getenv sites; the list is parsed by one of them)


def test_sbt_header_packing(hrt):
    lib = hrt.load_library()
    rec = hrt.SbtRecord()
    assert lib.hrt_sbt_record_pack_header(hrt.PROGRAM_TRIANGLE_METAL, C.byref(rec)) == 0
    assert bytes(rec.header[:4]) == b"HRT\x03" and not any(rec.header[4:])
    assert lib.hrt_sbt_record_pack_header(7, C.byref(rec)) == -1
    assert lib.hrt_sbt_record_pack_header(0, None) == -1


def _host_bvh_lib(hrt):
    """libhrt.so, or -- under `make asan-test` -- the sanitizer build of the host builder alone (HRT_HOST_BVH_LIB)."""
    import os
    path = os.environ.get("HRT_HOST_BVH_LIB")
    if not path:
        return hrt.load_library()
    lib = C.CDLL(path)
    lib.hrt_last_error.restype = C.c_char_p
    lib.hrt_last_error.argtypes = [C.c_void_p]
    lib.hrt_host_build_bvh8.restype = C.c_int
    lib.hrt_host_build_bvh8.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(hrt.BvhBlob)]
    lib.hrt_host_free.restype = None
    lib.hrt_host_free.argtypes = [C.POINTER(hrt.BvhBlob)]
    return lib


def _build(hrt, tris):
    lib = _host_bvh_lib(hrt)
    v = np.ascontiguousarray(tris, np.float32).reshape(-1, 3, 3)
    blob = hrt.BvhBlob()
    rc = lib.hrt_host_build_bvh8(v.ctypes.data, v.shape[0], C.byref(blob))
    assert rc == 0, lib.hrt_last_error(None)
    return lib, blob


@pytest.mark.parametrize("n", [1, 2, 3, 4, 9, 200, 5000])
def test_host_bvh8_build_and_cpu_walk(hrt, oracle, n):
    scene = hrt.scenes.random_soup(n, 0.2, 13)
    lib, blob = _build(hrt, scene["instances"][0]["vertices"])
    assert blob.n_triangles == n and blob.n_nodes >= 1
    o, d = oracle.random_rays(3000, n)
    want = oracle.OracleScene(scene, force_brute=True).trace(o, d)
    got = oracle.bvh8_trace(blob.nodes, blob.triangles, o, d)
    assert all(np.array_equal(a, b) for a, b in zip(got[:5], want))
    any_w = oracle.OracleScene(scene, force_brute=True).trace(o, d, any_hit=True)
    any_g = oracle.bvh8_trace(blob.nodes, blob.triangles, o, d, any_hit=True)
    assert np.array_equal(any_g[3] != 0xFFFFFFFF, any_w[3] != 0xFFFFFFFF)
    lib.hrt_host_free(C.byref(blob))
    assert not blob.nodes


def test_cpu_walk_of_rays_with_signed_zero_direction_components(hrt, oracle):
    """A direction component of -0.0 (a ray mirrored by an axis-aligned wall) counts as positive in the slab test's near / far choice
    (`d < 0`), so its guarded reciprocal has to be positive too: with copysignf it was negative, every box was culled, and one pixel of
    a Cornell box leaked light (round 4, tools/stress_modes.py).  The CPU walker of the product's tree shares that arithmetic."""
    scene = hrt.scenes.cornell_box(64, 64, 1)
    verts = np.concatenate([it["vertices"].reshape(-1, 3, 3) for it in scene["instances"]])
    lib, blob = _build(hrt, verts)
    o, d = oracle.axis_parallel_rays(20000, 3)
    assert (np.signbit(d) & (d == 0)).any() and ((d == 0) & ~np.signbit(d)).any()
    one = dict(scene); one["instances"] = [dict(scene["instances"][0], vertices=verts, normals=np.zeros((len(verts), 3, 3), np.float32))]
    want = oracle.OracleScene(one, force_brute=True).trace(o, d)
    got = oracle.bvh8_trace(blob.nodes, blob.triangles, o, d)
    assert np.array_equal(got[3], want[3]) and np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))
    assert (want[3] != 0xFFFFFFFF).mean() > 0.4             # (the room is open towards the camera)
    lib.hrt_host_free(C.byref(blob))


def test_oracle_renders_the_same_image_through_the_products_bvh8(hrt, oracle):
    """bench.py's cpu_baseline walks the product's BVH8 bytes ("same BVH bytes as the GPU run"): an oracle scene
    with the product's tree attached renders the very image it renders through its own BVH2, rays counted alike -- the closest hit is
    canonical -- and reports the product tree's node visits."""
    scene = hrt.scenes.random_soup(3000, 0.15, 5, 96, 64, 2)
    lib, blob = _build(hrt, scene["instances"][0]["vertices"])
    nodes = np.ctypeslib.as_array(C.cast(blob.nodes, C.POINTER(C.c_uint8)), shape=(blob.n_nodes * 80,)).copy()
    prims = np.ctypeslib.as_array(C.cast(blob.triangles, C.POINTER(C.c_uint8)), shape=(blob.n_triangles * 48,)).copy()
    lib.hrt_host_free(C.byref(blob))
    osc = oracle.OracleScene(scene)
    s1, s2 = oracle.rng_init(96, 64, 3), oracle.rng_init(96, 64, 3)
    a = osc.render(96, 64, s1, 2)
    osc.attach_bvh8(nodes, prims)
    b = osc.render(96, 64, s2, 2)
    assert np.array_equal(a["linear"].view(np.uint32), b["linear"].view(np.uint32)) and a["rays"] == b["rays"] and np.array_equal(s1, s2)
    assert 0 < b["node_visits"] < a["node_visits"]           # an 8-wide SAH tree against a median-split BVH2
    osc.attach_bvh8(None, None)
    s3 = oracle.rng_init(96, 64, 3)
    assert osc.render(96, 64, s3, 2)["node_visits"] == a["node_visits"]


def test_host_bvh8_degenerate_inputs(hrt, oracle):
    """Duplicates, zero-area triangles, identical centroids, huge coordinate range."""
    rng = np.random.default_rng(3)
    base = rng.uniform(-1, 1, (50, 3, 3)).astype(np.float32)
    tris = np.concatenate([base, base[:10], np.zeros((8, 3, 3), np.float32),                  # duplicates, all-zero triangles
                           np.repeat(rng.uniform(-1, 1, (1, 3, 3)).astype(np.float32), 40, 0),   # 40 identical triangles
                           (base[:5] * 1000 + 500).astype(np.float32)])                       # far away and large
    scene = {"instances": [hrt.scenes._tri_instance(tris, hrt.scenes.WHITE)], "camera": hrt.scenes._soup_camera(),
             "background": hrt.scenes.BACKGROUND}
    lib, blob = _build(hrt, tris)
    o, d = oracle.random_rays(4000, 17)
    want = oracle.OracleScene(scene, force_brute=True).trace(o, d)
    got = oracle.bvh8_trace(blob.nodes, blob.triangles, o, d)
    assert all(np.array_equal(a, b) for a, b in zip(got[:5], want))
    lib.hrt_host_free(C.byref(blob))


@pytest.mark.parametrize("n,edge", [(3, 0.5), (40, 0.6), (3000, 0.25), (20000, 0.08)])
def test_host_bvh8_spatial_splits_keep_the_canonical_hit(hrt, oracle, monkeypatch, n, edge):
    """SBVH (HRT_SBVH=1; what hrt_tlas_build uses under HRT_CTX_FAST_TRACE): nodes may cut their references with a plane, so a
    triangle can be referenced from several leaves, each holding the box of its part.  The builder's own validation (every
    reference's box inside every ancestor slot) passes, large overlapping triangles do get split, and the CPU walk of the tree
    still gives the brute-force closest hit for every ray, any-hit included: duplicates cannot change (t, instance, primitive)."""
    scene = hrt.scenes.random_soup(n, edge, 21)
    monkeypatch.setenv("HRT_SBVH", "1")
    lib, blob = _build(hrt, scene["instances"][0]["vertices"])
    assert blob.n_triangles >= n
    if n >= 3000:
        assert blob.n_triangles > 1.02 * n                         # references were duplicated
    o, d = oracle.random_rays(6000, n)
    want = oracle.OracleScene(scene, force_brute=True).trace(o, d)
    got = oracle.bvh8_trace(blob.nodes, blob.triangles, o, d)
    assert all(np.array_equal(a, b) for a, b in zip(got[:5], want))
    any_w = oracle.OracleScene(scene, force_brute=True).trace(o, d, any_hit=True)
    any_g = oracle.bvh8_trace(blob.nodes, blob.triangles, o, d, any_hit=True)
    assert np.array_equal(any_g[3] != 0xFFFFFFFF, any_w[3] != 0xFFFFFFFF)
    split_nodes, split_counts = got[5], got[6]
    lib.hrt_host_free(C.byref(blob))
    monkeypatch.setenv("HRT_SBVH", "0")
    lib, blob = _build(hrt, scene["instances"][0]["vertices"])
    assert blob.n_triangles == n
    plain = oracle.bvh8_trace(blob.nodes, blob.triangles, o, d)
    assert all(np.array_equal(a, b) for a, b in zip(plain[:5], want))
    if n >= 3000:
        assert split_nodes < plain[5]                             # and the split tree is the cheaper one to walk
    lib.hrt_host_free(C.byref(blob))


def test_host_bvh8_spatial_split_budget_and_determinism(hrt, oracle, monkeypatch):
    """The references spatial splits may add are a budget handed down the tree (HRT_SBVH_BUDGET x n for the root, shared among
    children in proportion to their sizes): the tree never holds more than n + budget references, whatever the worker threads'
    timing, and two builds of the same input are the same bytes."""
    n = 150000                                                    # large enough for the builder to fork worker threads
    scene = hrt.scenes.random_soup(n, 0.03, 5)
    v = scene["instances"][0]["vertices"]
    monkeypatch.setenv("HRT_SBVH", "1")
    sizes = {}
    for budget in ("0", "0.02", "2"):
        monkeypatch.setenv("HRT_SBVH_BUDGET", budget)
        lib, blob = _build(hrt, v)
        sizes[budget] = int(blob.n_triangles)
        assert n <= blob.n_triangles <= n + int(float(budget) * n)
        nodes = np.ctypeslib.as_array(C.cast(blob.nodes, C.POINTER(C.c_uint8)), (blob.n_nodes * 80,)).copy()
        prims = np.ctypeslib.as_array(C.cast(blob.triangles, C.POINTER(C.c_uint8)), (blob.n_triangles * 48,)).copy()
        lib.hrt_host_free(C.byref(blob))
        lib, blob = _build(hrt, v)
        assert blob.n_triangles == sizes[budget]
        assert np.array_equal(nodes, np.ctypeslib.as_array(C.cast(blob.nodes, C.POINTER(C.c_uint8)), (blob.n_nodes * 80,)))
        assert np.array_equal(prims, np.ctypeslib.as_array(C.cast(blob.triangles, C.POINTER(C.c_uint8)), (blob.n_triangles * 48,)))
        lib.hrt_host_free(C.byref(blob))
    assert sizes["0"] == n and sizes["0"] < sizes["0.02"] < sizes["2"]


def test_empty_bvh(hrt, oracle):
    lib = _host_bvh_lib(hrt)
    blob = hrt.BvhBlob()
    assert lib.hrt_host_build_bvh8(None, 0, C.byref(blob)) == 0
    assert blob.n_nodes == 1 and blob.n_triangles == 0
    got = oracle.bvh8_trace(blob.nodes, blob.triangles, [[0, 0, 3]], [[0, 0, -1]])
    assert got[3][0] == 0xFFFFFFFF
    lib.hrt_host_free(C.byref(blob))


def test_splitmix64_known_answers(hrt):
    out = hrt.scenes.splitmix64(0, 3)
    assert [int(x) for x in out] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]
    u = hrt.scenes.uniform_f32(1, 1000, -2.0, 3.0)
    assert u.dtype == np.float32 and u.min() >= -2.0 and u.max() < 3.0


def test_scenes_are_deterministic_and_sized(hrt):
    c1 = hrt.scenes.cornell_box()
    assert sum(len(i["vertices"]) for i in c1["instances"]) == 32 and (c1["width"], c1["height"], c1["spp"]) == (256, 256, 1)
    c2 = hrt.scenes.sphere_in_box()
    assert sum(len(i["vertices"]) for i in c2["instances"] if i["geometry"] == "triangles") == 12
    assert [i["geometry"] for i in c2["instances"]].count("spheres") == 1 and (c2["width"], c2["spp"]) == (512, 16)
    a, b = hrt.scenes.random_soup(1000, 0.03, 1), hrt.scenes.random_soup(1000, 0.03, 1)
    assert np.array_equal(a["instances"][0]["vertices"], b["instances"][0]["vertices"])
    v = a["instances"][0]["vertices"]
    assert v.shape == (1000, 3, 3) and np.abs(v.mean(axis=1)).max() <= 1.0 + 1e-6
    n = a["instances"][0]["normals"]
    assert np.allclose(np.linalg.norm(n[:, 0], axis=1), 1, atol=1e-5) and np.array_equal(n[:, 0], n[:, 2])
    c5 = hrt.scenes.soup_1m_8mat(n_triangles=800)
    assert len(c5["instances"]) == 8 and [i["material"] for i in c5["instances"]] == ["rough"] * 4 + ["metal"] * 4
    assert [i["fuzz"] for i in c5["instances"][4:]] == [0.0, 0.1, 0.3, 0.5]
    assert sum(len(i["vertices"]) for i in c5["instances"]) == 800


def test_tiles_partition_the_frame(hrt):
    for height, world in ((1080, 8), (1080, 4), (50, 3), (7, 2), (16, 1)):
        seen = np.zeros(height, int)
        for rank in range(world):
            t = hrt.tile_for_rank(height, rank, world)
            rows = [y for y in range(t.y_begin, t.y_end) if (y // t.stripe_rows) % t.stripe_period == t.stripe_phase]
            seen[rows] += 1
        assert (seen == 1).all()
    sizes = []
    for rank in range(8):
        t = hrt.tile_for_rank(1080, rank, 8)
        sizes.append(sum(1 for y in range(1080) if (y // t.stripe_rows) % t.stripe_period == t.stripe_phase))
    assert max(sizes) - min(sizes) <= 8


def test_hand_issued_loads_are_not_touched_before_their_wait():
    """The traversal kernels issue their node / primitive loads from inline asm and retire them with hand-counted
    s_waitcnt: between the two the compiler must not read, copy or spill the destination registers (it does not know they
    are in flight).  tools/audit_asm_loads.py compiles kernels.hip, fused.hip and fused_queue.hip to assembly with the Makefile's own flags
    and checks every instantiation; any compiler or flag change that breaks the invariant fails here."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "audit_asm_loads.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 hazardous instructions" in r.stdout and "hazardous instructions" in r.stdout


def test_path_kernel_hot_loop_keeps_its_instruction_budget():
    """k_fused is bound by instruction issue: its traversal loop is 470 instructions for triangles-only
    scenes -- 171 of the single-pipe kind -- with no scratch access, at 4 waves per SIMD (<= 128 VGPRs).  tools/loop_stats.py
    compiles fused.hip with the Makefile's flags and counts; a compiler, flag or source change that fattens the loop, spills inside
    it or costs a wave of occupancy fails here, before anybody has to find it in a benchmark."""
    import re
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "tools"))
    import loop_stats
    found = {name: (cs, ops) for name, cs, ops in loop_stats.loops("nvidia-optix-ray-tracer_amd/csrc/fused.hip", "_ZN3hrt7k_fused")}
    tri = [v for k, v in found.items() if "ILb0ELb0ELb0E" in k]          # <HAS_SPHERES = false, INSTANCED = false, REUSE = false>: C4's kernel
    assert len(found) == 8 and len(tri) == 1, sorted(found)
    cs, ops = tri[0]
    assert sum(cs.values()) <= 480, dict(cs)
    assert cs["valu_complex"] <= 175 and cs["salu"] <= 110, dict(cs)
    assert not any(op.startswith("scratch_") or op.startswith("buffer_") for op in ops), sorted(ops)
    text = Path("/tmp/hrt_loops_fused.s").read_text()
    # 4 waves per SIMD (<= 128 VGPRs) for the one-level kernels; the INSTANCED instantiations are compiled for 3 (<= 168)
    counts = sorted(int(v) for v in re.findall(r"\.vgpr_count:\s+(\d+)", text))
    assert len(counts) == 8 and counts[3] <= 128 and counts[7] <= 168, counts
    # ... and what they keep in scratch stays what it is (kernel constants, reloaded around the shading): a 12th register spilled by the C4
    # kernel was a reload in every regeneration (round 4: a second SGPR operand of the bookkeeping sequence had done that)
    spills = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)}
    by_kind = {k: v for k, v in spills.items() if "k_fused" in k}
    assert len(by_kind) == 8, sorted(spills)
    assert max(v for k, v in by_kind.items() if "ILb0ELb0ELb0E" in k) <= 11 and max(v for k, v in by_kind.items() if "ILb1ELb0ELb0E" in k) <= 14, by_kind
    assert all(v == 0 for k, v in by_kind.items() if "ELb1ELb" in k.split("k_fusedILb")[1][:9]), by_kind      # INSTANCED: nothing spilled at 3 waves per SIMD
    # the REUSE instantiations (HRT_CTX_REUSE_PRIMARY) keep the same loop: nothing of theirs is inside it
    for k, (cs_k, ops_k) in found.items():
        if "ILb0ELb0ELb1E" in k:
            assert sum(cs_k.values()) <= 480 and not any(op.startswith("scratch_") for op in ops_k), (k, dict(cs_k))


def test_context_flags_of_the_binding_are_the_headers(hrt):
    """The ctypes binding restates the context flags of include/hrt.h as Python constants: every HRT_CTX_* of the header has its
    CTX_* twin with the same value, no two flags share a bit."""
    import re
    from pathlib import Path
    text = (Path(__file__).resolve().parent.parent / "include" / "hrt.h").read_text()
    flags = {m.group(1): int(m.group(2), 16) for m in re.finditer(r"#define\s+HRT_CTX_(\w+)\s+0x([0-9a-fA-F]+)u", text)}
    assert len(flags) >= 6 and len(set(flags.values())) == len(flags), flags
    for name, value in flags.items():
        assert getattr(hrt, "CTX_" + name) == value, name
        assert value & (value - 1) == 0, name


def test_tools_compile():
    """The measurement and stress scripts under tools/ are run by hand on the GPU box: at least they parse."""
    import py_compile
    from pathlib import Path
    for path in sorted((Path(__file__).resolve().parent.parent / "tools").glob("*.py")):
        py_compile.compile(str(path), doraise=True, cfile=f"/tmp/hrt_pyc_{path.stem}.pyc")
