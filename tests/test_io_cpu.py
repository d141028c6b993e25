"""Host-side readers of the reference's input formats (include/hrt_io.h) on the reference's
own shipped sample data (tests/golden/files/: config.json, STL shapes, particle VTK time steps, series file)."""
import ctypes as C
import importlib
import json
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
FILES = ROOT / "tests" / "golden" / "files"


@pytest.fixture(scope="module")
def io(hrt):
    return importlib.import_module("nvidia-optix-ray-tracer_amd.io")


def test_io_library_exports_every_declared_symbol(io):
    header = (ROOT / "include" / "hrt_io.h").read_text()
    declared = sorted(set(re.findall(r"\b(hrt_io_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(io.IO_EXPORTS)
    for name in declared:
        assert hasattr(io.lib(), name)
    assert C.sizeof(io.IoSphere) == 80


def test_stl_shapes(io):
    counts = []
    for k in range(8):
        m = io.read_stl(FILES / "shape" / "separated" / ("shape_%010d.stl" % k))
        v, n, fn = m["vertices"], m["normals"], m["file_normals"]
        counts.append(len(v))
        assert v.dtype == np.float32 and v.shape[1:] == (3, 3) and n.shape == v.shape and fn.shape == (len(v), 3)
        # per-vertex layout: the facet's unit geometric normal three times (what Shader.cu:140-142 indexes)
        assert np.array_equal(n[:, 0], n[:, 1]) and np.array_equal(n[:, 0], n[:, 2])
        assert np.abs(np.linalg.norm(n[:, 0].astype(np.float64), axis=1) - 1).max() < 1e-6
        # the normal is the winding's, oriented by vtkPolyDataNormals' consistency / auto-orientation pass as the reference sets it
        # (VTKReaderImpl.cpp:279-285).  The two closed particle shapes are wound outwards already: nothing is reversed and the
        # normal agrees with the one the file states (to its 6 printed digits).  The wall quads are open surfaces, which that pass
        # orients by its seed rule alone -- the leftmost triangle's normal must not point towards +x -- so a wall facing +x comes
        # out facing -x (the shader turns normals towards the ray anyway, Shader.cu:152-155); both triangles of a quad agree.
        if k < 2:
            assert np.abs(n[:, 0] - fn).max() < 1e-4    # vertices are printed with 6 digits, so is the stated normal
        else:
            sign = -1.0 if abs(fn[0, 0]) > 1e-6 and fn[0, 0] > 0 else 1.0
            assert np.abs(n[:, 0] - sign * fn).max() < 1e-4 and np.array_equal(n[0, 0], n[1, 0])
            assert n[0, 0, 0] <= 0
    assert counts == [252, 396, 2, 2, 2, 2, 2, 2]
    # the two particle shapes are closed surfaces: every undirected edge belongs to exactly two triangles, and the
    # normals point away from the centroid (star-shaped bodies)
    for k in (0, 1):
        m = io.read_stl(FILES / "shape" / "separated" / ("shape_%010d.stl" % k))
        v = m["vertices"]
        edges = {}
        for t in v:
            for a, b in ((0, 1), (1, 2), (2, 0)):
                key = tuple(sorted((tuple(t[a]), tuple(t[b]))))
                edges[key] = edges.get(key, 0) + 1
        assert set(edges.values()) == {2}
        c = v.reshape(-1, 3).mean(axis=0)
        assert (((v.mean(axis=1) - c) * m["normals"][:, 0]).sum(axis=1) > 0).all()
    # first vertex of the first file, as float(strtod(...))
    m0 = io.read_stl(FILES / "shape" / "separated" / "shape_0000000000.stl")
    assert np.array_equal(m0["vertices"][0, 0], np.float32([0.0240016, 0.0302721, -0.0297948]))


def test_stl_errors(io, tmp_path):
    with pytest.raises(io.IoError, match="cannot open"):
        io.read_stl(tmp_path / "missing.stl")
    bad = tmp_path / "bad.stl"
    bad.write_text("solid x\nfacet normal 0 0 1\nouter loop\nvertex 0 0 0\nvertex 1 0 0\nendloop\nendfacet\nendsolid x\n")
    with pytest.raises(io.IoError, match="bad vertex"):
        io.read_stl(bad)
    empty = tmp_path / "empty.stl"
    empty.write_text("solid x\nendsolid x\n")
    assert len(io.read_stl(empty)["vertices"]) == 0


def test_particle_vtk(io):
    steps = [io.read_particle_vtk(FILES / "particle" / ("particle_%015d.vtk" % k)) for k in (0, 100, 200)]
    for s in steps:
        st = s["states"]
        assert st.shape == (25, 12) and st.dtype == np.float32
        assert np.array_equal(s["ids"], np.arange(25, dtype=np.uint64))
        assert set(s["shape_ids"].tolist()) <= {0, 1}
        assert np.abs(np.linalg.norm(st[:, :4].astype(np.float64), axis=1) - 1).max() < 1e-5      # unit quaternions
        assert not st[:, 10:].any()
    s0 = steps[0]["states"]
    # first record of the file: quat components in file order, position, velocity (0, 0, -2)
    assert np.array_equal(s0[0, :4], np.float32([0.987695, 0.113077, 0.0865597, 0.064658]))
    assert np.array_equal(s0[0, 4:7], np.float32([-0.403633, -0.403633, 0.396367]))
    assert np.array_equal(s0[:, 7:10], np.tile(np.float32([0, 0, -2]), (25, 1)))
    assert np.array_equal(steps[0]["shape_ids"][:5], [0, 1, 0, 1, 1])
    # the particles fall: z decreases from step to step by about v * dt = 0.02
    dz = steps[1]["states"][:, 6] - s0[:, 6]
    assert (dz < 0).all() and np.abs(dz + 0.02).max() < 5e-3


def test_particle_vtk_errors(io, tmp_path):
    src = (FILES / "particle" / "particle_000000000000000.vtk").read_text()
    cut = tmp_path / "cut.vtk"
    cut.write_text(src[: len(src) // 3])
    with pytest.raises(io.IoError):
        io.read_particle_vtk(cut)
    noquat = tmp_path / "noquat.vtk"
    noquat.write_text(src.replace("SCALARS quat double 4", "SCALARS tauq double 4"))
    with pytest.raises(io.IoError, match="quat"):
        io.read_particle_vtk(noquat)
    binary = tmp_path / "bin.vtk"
    binary.write_text(src.replace("ASCII", "BINARY", 1))
    with pytest.raises(io.IoError, match="ASCII"):
        io.read_particle_vtk(binary)


def test_series_durations(io, tmp_path):
    files, dur = io.read_series(str(FILES) + "/", "particle.vtk.series")
    assert [Path(f).name for f in files] == ["particle_%015d.vtk" % k for k in (0, 100, 200)]
    assert all(f.startswith(str(FILES) + "/particle/") for f in files)
    t = np.float32([0.0, 1e-2, 2e-2])
    assert np.array_equal(dur, np.float32([t[1] - t[0], t[2] - t[1], t[2] - t[1]]))     # the last one repeats (VTKTimeReader.cu:80-81)
    one = tmp_path / "one.series"
    one.write_text(json.dumps({"file-series-version": "1.0", "files": [{"name": "a.vtk", "time": 3.5}]}))
    files, dur = io.read_series(str(tmp_path) + "/", "one.series")
    assert files == [str(tmp_path) + "/a.vtk"] and np.array_equal(dur, np.float32([1000.0]))
    bad = tmp_path / "bad.series"
    bad.write_text('{"file-series-version": "1.0", "files": 3}')
    with pytest.raises(io.IoError, match="files"):
        io.read_series(str(tmp_path) + "/", "bad.series")


def test_color_ramp(io):
    r = io.bake_color_ramp("terrain", 25)
    assert r.shape == (25, 3)
    assert np.array_equal(r[0], np.float32([0.149, 0.149, 0.149])) and np.array_equal(r[24], np.float32([0.996, 0.922, 0.545]))
    # u = 12/24 = 0.5 is a stop: lower + (upper - lower) * 1
    lo, hi = np.float32([0.114, 0.451, 0.208]), np.float32([0.639, 0.784, 0.325])
    assert np.array_equal(r[12], lo + (hi - lo) * np.float32(1.0))
    # between stops: plain float32 lerp with t = (u - p0) / (p1 - p0)
    u = np.float32(7) / np.float32(24)
    t = (u - np.float32(0.25)) / (np.float32(0.5) - np.float32(0.25))
    assert np.array_equal(r[7], lo + (hi - lo) * t)
    assert np.array_equal(io.bake_color_ramp("TeRRain", 25), r)                         # case-insensitive
    assert np.array_equal(io.bake_color_ramp("no-such-preset", 5), io.bake_color_ramp("viridis", 5))
    assert np.array_equal(io.bake_color_ramp("grayscale", 1), np.float32([[0.95, 0.95, 0.95]]))   # count 1: the last stop
    assert io.bake_color_ramp("plasma", 0).shape == (0, 3)
    g = io.bake_color_ramp("grayscale", 11)
    assert np.allclose(g[:, 0], np.linspace(0.05, 0.95, 11), atol=1e-6)


def test_construct_transform_equals_oracle(io, oracle):
    rng = np.random.default_rng(8)
    for _ in range(100):
        s, r, c = rng.uniform(-3, 3, 3), rng.uniform(-180, 180, 3), rng.uniform(0.2, 2, 3)
        assert np.array_equal(io.construct_transform(s, r, c).view(np.uint32), oracle.construct_transform(s, r, c).view(np.uint32))


def test_config(io, tmp_path):
    c = io.load_config(FILES / "config.json")
    assert c["mesh"] is False and c["api"] == "VK" and c["opengl"] is False and c["window"] == (1200, 800)
    assert c["series-name"] == "particle.vtk.series" and c["stl-path"] == "../files/shape/separated/"
    assert c["particle-material-preset"] == "terrain" and c["fps"] == 240 and c["render-speed-ratio"] == 4
    assert np.array_equal(c["roughs"], np.float32([[0.65, 0.05, 0.05], [0.73, 0.73, 0.73], [0.12, 0.45, 0.15], [0.70, 0.60, 0.50]]))
    assert np.array_equal(c["metals"], np.float32([[0.8, 0.85, 0.88, 0.0]]))
    assert len(c["spheres"]) == 1
    s = c["spheres"][0]
    assert s["radius"] == 1000.0 and not s["metal"] and s["material_index"] == 3
    assert np.array_equal(s["transform"], np.float32([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, -1000.5]))
    assert np.array_equal(c["camera-center"], np.float32([5, 0, 0])) and np.array_equal(c["up-direction"], np.float32([0, 0, 1]))
    src = json.loads((FILES / "config.json").read_text())
    for api, msg in (("D3D12", "Windows"), ("Metal", "Invalid api")):
        src["loop-data"]["api"] = api
        p = tmp_path / "c.json"
        p.write_text(json.dumps(src))
        with pytest.raises(io.IoError, match=msg):
            io.load_config(p)
    src["loop-data"]["api"] = "OGL"
    del src["stl-path"]
    p.write_text(json.dumps(src))
    with pytest.raises(io.IoError, match="stl-path"):
        io.load_config(p)
    p.write_text("{ not json")
    with pytest.raises(io.IoError, match="parse error"):
        io.load_config(p)


def test_mesh_cache_round_trip_and_layout(io, tmp_path):
    """particleN.cache (VTKMeshReader.cuh:15-23): [u64 n] { [u64 id] [float3 vel] [u64 vertices] [float3 xN] [float3 xN] }."""
    shapes = [io.read_stl(FILES / "shape" / "separated" / ("shape_%010d.stl" % k)) for k in (0, 2)]
    parts = [{"id": 7, "velocity": np.float32([0, 0, -2]), "vertices": shapes[0]["vertices"], "normals": shapes[0]["normals"]},
             {"id": 3, "velocity": np.float32([1, 2, 3]), "vertices": shapes[1]["vertices"], "normals": shapes[1]["normals"]},
             {"id": 9, "velocity": np.float32([0, 0, 0]), "vertices": np.zeros((0, 3, 3), np.float32), "normals": np.zeros((0, 3, 3), np.float32)}]
    path = tmp_path / "particle0.cache"
    io.write_mesh_cache(path, parts)
    raw = path.read_bytes()
    assert len(raw) == 8 + sum(8 + 12 + 8 + 2 * 36 * len(p["vertices"]) for p in parts)
    assert int.from_bytes(raw[:8], "little") == 3 and int.from_bytes(raw[8:16], "little") == 7
    assert np.array_equal(np.frombuffer(raw[16:28], np.float32), [0, 0, -2])
    assert int.from_bytes(raw[28:36], "little") == 3 * 252
    assert np.array_equal(np.frombuffer(raw[36:36 + 36], np.float32).reshape(3, 3), shapes[0]["vertices"][0])
    back = io.read_mesh_cache(path)
    assert [p["id"] for p in back] == [7, 3, 9]
    for a, b in zip(parts, back):
        assert np.array_equal(a["vertices"], b["vertices"]) and np.array_equal(a["normals"], b["normals"]) and np.array_equal(a["velocity"], b["velocity"])
    (tmp_path / "short.cache").write_bytes(raw[: len(raw) // 2])
    with pytest.raises(io.IoError, match="truncated"):
        io.read_mesh_cache(tmp_path / "short.cache")


def test_mesh_cache_reader_against_hand_assembled_bytes(io, tmp_path):
    """A pin for the cache reader that is not its own writer: the bytes of a two-particle cache file put together field by
    field from the layout the reference documents (include/Util/VTKMeshReader.cuh:15-23) and writes (VTKMeshReader.cu:54-72:
    every size a fixed-width little-endian uint64, float3 = 12 bytes, vertices then normals)."""
    import struct
    v0 = [(0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0), (1.0, 0.0, 1.0), (0.0, 1.0, 1.0)]
    n0 = [(0.0, 0.0, 1.0)] * 6
    v1 = [(2.0, 2.0, 2.0), (3.0, 2.0, 2.0), (2.0, 3.0, 2.5)]
    n1 = [(0.0, -0.4472136, 0.8944272)] * 3
    blob = struct.pack("<Q", 2)
    blob += struct.pack("<Q", 41) + struct.pack("<3f", 0.5, -1.5, 2.0) + struct.pack("<Q", 6)
    blob += b"".join(struct.pack("<3f", *p) for p in v0) + b"".join(struct.pack("<3f", *p) for p in n0)
    blob += struct.pack("<Q", 7) + struct.pack("<3f", 0.0, 0.0, -9.81) + struct.pack("<Q", 3)
    blob += b"".join(struct.pack("<3f", *p) for p in v1) + b"".join(struct.pack("<3f", *p) for p in n1)
    assert len(blob) == 8 + (8 + 12 + 8 + 2 * 72) + (8 + 12 + 8 + 2 * 36)
    path = tmp_path / "particle17.cache"
    path.write_bytes(blob)
    parts = io.read_mesh_cache(path)
    assert [p["id"] for p in parts] == [41, 7]
    assert np.array_equal(parts[0]["velocity"], np.float32([0.5, -1.5, 2.0])) and np.array_equal(parts[1]["velocity"], np.float32([0.0, 0.0, -9.81]))
    assert parts[0]["vertices"].shape == (2, 3, 3) and parts[1]["vertices"].shape == (1, 3, 3)      # triangleCount = vertexCount / 3, VTKMeshReader.cu:254
    assert np.array_equal(parts[0]["vertices"].reshape(-1, 3), np.float32(v0)) and np.array_equal(parts[0]["normals"].reshape(-1, 3), np.float32(n0))
    assert np.array_equal(parts[1]["vertices"].reshape(-1, 3), np.float32(v1)) and np.array_equal(parts[1]["normals"].reshape(-1, 3), np.float32(n1))
    # and the writer reproduces these very bytes
    io.write_mesh_cache(tmp_path / "again.cache", parts)
    assert (tmp_path / "again.cache").read_bytes() == blob
    # a vertex count that is not a multiple of 3 cannot be a triangle list
    bad = bytearray(blob); bad[28:36] = struct.pack("<Q", 5)
    (tmp_path / "bad.cache").write_bytes(bytes(bad))
    with pytest.raises(io.IoError):
        io.read_mesh_cache(tmp_path / "bad.cache")


def test_metadata_cache(io, tmp_path):
    """metadata.cache (VTKMeshReader.cuh:23): the largest cell count of the series as decimal text, `metaData << maxCellCount` /
    `metaData >> maxCount` (VTKMeshReader.cu:203, :277); the directory argument ends with a separator like cache-path does."""
    d = str(tmp_path) + "/"
    (tmp_path / "metadata.cache").write_text("12345")                # what the reference's cache run leaves behind
    assert io.read_metadata_cache(d) == 12345
    (tmp_path / "metadata.cache").write_text("  77\n")                # operator>> skips white space
    assert io.read_metadata_cache(d) == 77
    io.write_metadata_cache(d, 2 ** 40 + 5)
    assert (tmp_path / "metadata.cache").read_bytes() == str(2 ** 40 + 5).encode()       # no newline, no padding
    assert io.read_metadata_cache(d) == 2 ** 40 + 5
    (tmp_path / "metadata.cache").write_text("cells")
    with pytest.raises(io.IoError, match="no cell count"):
        io.read_metadata_cache(d)
    with pytest.raises(io.IoError, match="cannot open"):
        io.read_metadata_cache(str(tmp_path / "nowhere") + "/")


MESH_VTK = """# vtk DataFile Version 2.0
two particles as triangle strips
ASCII
DATASET POLYDATA
POINTS 9 double
0 0 0  1 0 0  0 1 0  1 1 0  0 2 0
5 5 5  6 5 5  5 6 5  5 5 6
TRIANGLE_STRIPS 2 11
5 0 1 2 3 4
4 5 6 7 8
CELL_DATA 2
SCALARS id int 1
LOOKUP_TABLE default
12 3
VECTORS vel double
0 0 -1  0.5 0.25 0
"""


def test_vtk_mesh_file_strips_to_triangles(io, tmp_path):
    """readVTKMeshFile (VTKReaderImpl.cpp:24-137): a strip of k points gives k - 2 triangles, the odd ones with their last two
    vertices swapped (:96-104) so that the winding stays the same; ids / velocities from CELL_DATA; every vertex carries its
    point's normal (normalised sum of the unit normals of the triangles around the point).  The result written as a cache
    file is what the reference's cache run would write, and reads back the same."""
    p = tmp_path / "mesh_000.vtk"
    p.write_text(MESH_VTK)
    parts, cells = io.read_vtk_mesh_file(p)
    assert cells == 2 and [q["id"] for q in parts] == [12, 3]
    assert np.array_equal(parts[0]["velocity"], np.float32([0, 0, -1])) and np.array_equal(parts[1]["velocity"], np.float32([0.5, 0.25, 0]))
    a = parts[0]["vertices"]
    assert a.shape == (3, 3, 3)
    assert np.array_equal(a[0], np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]]))
    assert np.array_equal(a[1], np.float32([[1, 0, 0], [1, 1, 0], [0, 1, 0]]))          # odd triangle: (v1, v3, v2)
    assert np.array_equal(a[2], np.float32([[0, 1, 0], [1, 1, 0], [0, 2, 0]]))
    for t in a:                                                                          # one winding throughout: all face +z
        assert np.cross(t[1] - t[0], t[2] - t[0])[2] > 0
    assert np.array_equal(parts[0]["normals"], np.broadcast_to(np.float32([0, 0, 1]), (3, 3, 3)))   # a flat strip
    b, bn = parts[1]["vertices"], parts[1]["normals"]
    assert b.shape == (2, 3, 3) and np.array_equal(b[1], np.float32([[6, 5, 5], [5, 5, 6], [5, 6, 5]]))
    n_face0 = np.float64([0, 0, 1]); n_face1 = np.cross([-1, 0, 1], [-1, 1, 0]); n_face1 = n_face1 / np.linalg.norm(n_face1)
    shared = (n_face0 + n_face1) / np.linalg.norm(n_face0 + n_face1)
    assert np.allclose(bn[0][0], n_face0) and np.allclose(bn[0][1], shared, atol=1e-7) and np.allclose(bn[1][1], n_face1, atol=1e-7)
    io.write_mesh_cache(tmp_path / "particle0.cache", parts)
    back = io.read_mesh_cache(tmp_path / "particle0.cache")
    assert all(np.array_equal(x["vertices"], y["vertices"]) and np.array_equal(x["normals"], y["normals"]) for x, y in zip(parts, back))
    # the reference refuses any other cell type (:73-77) and files without the cell data (:49-52)
    (tmp_path / "poly.vtk").write_text(MESH_VTK.replace("TRIANGLE_STRIPS 2 11", "POLYGONS 2 11"))
    with pytest.raises(io.IoError, match="illegal cell type"):
        io.read_vtk_mesh_file(tmp_path / "poly.vtk")
    (tmp_path / "nodata.vtk").write_text(MESH_VTK.split("CELL_DATA")[0])
    with pytest.raises(io.IoError, match="cell data"):
        io.read_vtk_mesh_file(tmp_path / "nodata.vtk")


def test_time_mode_scene_assembly(io):
    """RendererTime::commitRendererData in small: extra sphere first, then the particles of file 0 sharing shapes."""
    tm = io.time_mode_scene(FILES / "config.json")
    sc = tm["scene"]
    assert len(sc["instances"]) == 26 and tm["n_extra"] == 1
    assert sc["instances"][0]["geometry"] == "spheres" and np.array_equal(sc["instances"][0]["albedo"], np.float32([0.70, 0.60, 0.50]))
    assert [it["shape"] for it in sc["instances"][1:6]] == [0, 1, 0, 1, 1]
    assert all(np.array_equal(it["albedo"], tm["ramp"][i]) for i, it in enumerate(sc["instances"][1:]))
    assert tm["frame_counts"] == [9, 9, 9]                     # size_t(0.01f * float(240 * 4)), RendererTime.cu:427-428
    assert sc["width"] == 1200 and sc["height"] == 800 and sc["camera"]["opengl"] is False


def test_mesh_mode_cache_run_through_the_cpp_driver(io, tmp_path):
    """hrt_mesh_render with "cache": true is RendererMesh::writeCacheFilesAndExit (src/Util/VTKMeshReader.cu:146-215): every VTK
    file of the series -> particleN.cache, the largest cell count -> metadata.cache, with loader threads; no GPU involved.  The
    files it writes are what the reader + writer give one by one, and the data set then assembles as a Mesh-mode scene."""
    import subprocess
    from pathlib import Path
    exe = Path(__file__).resolve().parent.parent / "nvidia-optix-ray-tracer_amd" / "lib" / "hrt_mesh_render"
    if not exe.exists():
        pytest.skip("run `make tools`")
    cfg_path = io.write_mesh_mode_sample(tmp_path, n_files=3, n_particles=7, as_vtk=True)
    p = subprocess.run([str(exe), str(cfg_path), str(tmp_path / "bin")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert io.read_metadata_cache(str(tmp_path / "cache") + "/") == 7
    for k in range(3):
        particles, cells = io.read_vtk_mesh_file(tmp_path / "files" / f"mesh_{k}.vtk")
        assert cells == 7 and all(len(q["vertices"]) == 4 for q in particles)          # strips of 6 points
        io.write_mesh_cache(str(tmp_path / "expect.cache"), particles)
        assert (tmp_path / "cache" / f"particle{k}.cache").read_bytes() == (tmp_path / "expect.cache").read_bytes()
    # flip the config to the render run: the same directory now is a complete Mesh-mode data set
    import json
    cfg = json.loads(cfg_path.read_text()); cfg["cache"] = False; cfg_path.write_text(json.dumps(cfg))
    mm = io.mesh_mode_scene(cfg_path)
    assert len(mm["scenes"]) == 3 and all(len(s["instances"]) == 1 + 7 for s in mm["scenes"])
    assert mm["velocities"][0].shape == (7, 3) and mm["frame_counts"] == [3, 6, 6]


def _write_stl(path, tris):
    with open(path, "w") as f:
        f.write("solid t\n")
        for t in tris:
            f.write("facet normal 0 0 0\nouter loop\n")
            for v in t:
                f.write("vertex %r %r %r\n" % tuple(float(x) for x in v))
            f.write("endloop\nendfacet\n")
        f.write("endsolid t\n")


def _cube_triangles():
    """12 triangles of the unit cube, wound outwards."""
    c = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32)      # index = 4x + 2y + z
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]   # -x +x -y +y -z +z, outward
    tris = []
    for a, b, cc, d in quads:
        tris += [[c[a], c[b], c[cc]], [c[a], c[cc], c[d]]]
    return np.array(tris, np.float32)


def test_normals_consistency_and_auto_orientation(io, tmp_path):
    """vtkPolyDataNormals as the reference configures it (Consistency + AutoOrientNormals: VTKReaderImpl.cpp:54-60, :279-285),
    restated in scene_io.cpp.  The reference keeps the file's vertex order and takes only the normals from the filter, so what
    changes is the normal's sign: (1) a closed shape with ONE face wound the wrong way gets that face's normal turned outwards,
    (2) an inside-out closed shape gets all of them turned, (3) a correctly wound one is left alone, (4) two shapes in one file
    are oriented independently; Mesh-mode strips: (5) a strip stored backwards contributes its point normals with the sign of
    the component it is connected to."""
    cube = _cube_triangles()
    centre = np.float32([0.5, 0.5, 0.5])

    def outward(m):
        v, n = m["vertices"], m["normals"][:, 0]
        c = v.reshape(-1, 3).mean(axis=0) if len(v) <= 12 else None
        return v, n, c

    def all_outward(v, n, c):
        return bool((((v.mean(axis=1) - c) * n).sum(axis=1) > 0).all())
    # (3) as wound
    _write_stl(tmp_path / "cube.stl", cube)
    m = io.read_stl(tmp_path / "cube.stl")
    assert np.array_equal(m["vertices"], cube) and all_outward(m["vertices"], m["normals"][:, 0], centre)
    good_normals = m["normals"].copy()
    # (1) one face reversed: vertices as in the file, normal as for the correct winding
    one = cube.copy(); one[5] = one[5][::-1]
    _write_stl(tmp_path / "one.stl", one)
    m = io.read_stl(tmp_path / "one.stl")
    assert np.array_equal(m["vertices"], one)
    assert all_outward(m["vertices"], m["normals"][:, 0], centre) and np.allclose(m["normals"], good_normals, atol=1e-7)
    # (2) inside out
    inside_out = cube[:, ::-1].copy()
    _write_stl(tmp_path / "inv.stl", inside_out)
    m = io.read_stl(tmp_path / "inv.stl")
    assert np.array_equal(m["vertices"], inside_out) and all_outward(m["vertices"], m["normals"][:, 0], centre)
    # (4) two components, the second inside out and shifted
    two = np.concatenate([cube, inside_out + np.float32([3, 0.25, -1])])
    _write_stl(tmp_path / "two.stl", two)
    m = io.read_stl(tmp_path / "two.stl")
    assert all_outward(m["vertices"][:12], m["normals"][:12, 0], centre)
    assert all_outward(m["vertices"][12:], m["normals"][12:, 0], centre + np.float32([3, 0.25, -1]))
    # (5) Mesh mode: a closed tetrahedron as ONE strip (0 1 2 3 0 1 gives its four faces), once as is and once with the strip reversed;
    # the point normals must point away from the centroid either way
    tet = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
    for order in ("0 1 2 3 0 1", "1 0 3 2 1 0"):
        text = ("# vtk DataFile Version 2.0\ntet\nASCII\nDATASET POLYDATA\nPOINTS 4 double\n" + " ".join("%g %g %g" % tuple(p) for p in tet) +
                "\nTRIANGLE_STRIPS 1 7\n6 " + order + "\nCELL_DATA 1\nSCALARS id int 1\nLOOKUP_TABLE default\n5\nVECTORS vel double\n0 0 0\n"
                "FIELD FieldData 1\ntime 1 1 double\n0.25\nMETADATA\nINFORMATION 0\n\n")
        (tmp_path / "tet.vtk").write_text(text)
        parts, cells = io.read_vtk_mesh_file(tmp_path / "tet.vtk")          # (FIELD / METADATA blocks are read past, as vtkPolyDataReader does)
        assert cells == 1 and parts[0]["vertices"].shape == (4, 3, 3)
        v, n = parts[0]["vertices"].reshape(-1, 3), parts[0]["normals"].reshape(-1, 3)
        assert (((v - tet.mean(axis=0)) * n).sum(axis=1) > 0).all(), order
