"""ctypes face of libhrt_io.so (include/hrt_io.h): the reference's input formats read on the host, and the
assembly of a Time-mode scene from them the way RendererTime::commitRendererData does
(src/Global/RendererTime.cu:160-290)."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_LIB = None
IO_EXPORTS = ["hrt_io_last_error", "hrt_io_read_stl", "hrt_io_free_mesh", "hrt_io_read_particle_vtk", "hrt_io_free_particles",
              "hrt_io_read_series", "hrt_io_free_series", "hrt_io_bake_color_ramp", "hrt_io_construct_transform",
              "hrt_io_load_config", "hrt_io_free_config", "hrt_io_read_mesh_cache", "hrt_io_write_mesh_cache", "hrt_io_free_mesh_cache",
              "hrt_io_read_metadata_cache", "hrt_io_write_metadata_cache", "hrt_io_read_vtk_mesh_file"]


class IoMesh(C.Structure):
    _fields_ = [("vertices", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)), ("file_normals", C.POINTER(C.c_float)),
                ("n_triangles", C.c_uint64)]


class IoParticles(C.Structure):
    _fields_ = [("states", C.POINTER(C.c_float)), ("ids", C.POINTER(C.c_uint64)), ("shape_ids", C.POINTER(C.c_uint64)), ("n", C.c_uint64)]


class IoSeries(C.Structure):
    _fields_ = [("files", C.POINTER(C.c_char_p)), ("durations", C.POINTER(C.c_float)), ("n", C.c_uint64)]


class IoSphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("metal", C.c_int32), ("material_index", C.c_uint64),
                ("transform", C.c_float * 12)]


class IoConfig(C.Structure):
    _fields_ = [("mesh", C.c_int32), ("cache", C.c_int32), ("debug_mode", C.c_int32), ("api_is_opengl", C.c_int32),
                ("series_path", C.c_char_p), ("series_name", C.c_char_p), ("cache_path", C.c_char_p), ("stl_path", C.c_char_p),
                ("particle_material_preset", C.c_char_p), ("api", C.c_char_p),
                ("cache_process_thread_count", C.c_uint64),
                ("roughs", C.POINTER(C.c_float)), ("n_roughs", C.c_uint64),
                ("metals", C.POINTER(C.c_float)), ("n_metals", C.c_uint64),
                ("spheres", C.POINTER(IoSphere)), ("n_spheres", C.c_uint64),
                ("window_width", C.c_int32), ("window_height", C.c_int32),
                ("fps", C.c_uint64), ("render_speed_ratio", C.c_uint64), ("camera_initial_speed_ratio", C.c_uint64),
                ("camera_center", C.c_float * 3), ("camera_target", C.c_float * 3), ("up_direction", C.c_float * 3),
                ("particle_shift", C.c_float * 3), ("particle_scale", C.c_float * 3),
                ("mouse_sensitivity", C.c_float), ("camera_pitch_limit_degree", C.c_float), ("camera_speed_stride", C.c_float)]


class IoMeshCache(C.Structure):
    _fields_ = [("n_particles", C.c_uint64), ("ids", C.POINTER(C.c_uint64)), ("velocities", C.POINTER(C.c_float)),
                ("first_triangle", C.POINTER(C.c_uint64)), ("vertices", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float))]


def lib():
    global _LIB
    if _LIB is None:
        import os
        path = Path(os.environ["HRT_IO_LIB"]) if os.environ.get("HRT_IO_LIB") else Path(__file__).resolve().parent / "lib" / "libhrt_io.so"   # HRT_IO_LIB: the sanitizer build (make asan-test)
        if not path.exists():
            raise RuntimeError(f"{path} is missing: run `make lib` (or __graft_entry__.build())")
        L = C.CDLL(str(path))
        L.hrt_io_last_error.restype = C.c_char_p
        for name in IO_EXPORTS:
            if name.startswith("hrt_io_free"):
                getattr(L, name).restype = None
        L.hrt_io_bake_color_ramp.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p]
        L.hrt_io_construct_transform.argtypes = [C.c_void_p] * 4
        L.hrt_io_read_series.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p]
        for name in ("hrt_io_read_stl", "hrt_io_read_particle_vtk", "hrt_io_load_config", "hrt_io_read_mesh_cache", "hrt_io_write_mesh_cache"):
            getattr(L, name).argtypes = [C.c_char_p, C.c_void_p]
        L.hrt_io_read_metadata_cache.argtypes = [C.c_char_p, C.POINTER(C.c_uint64)]
        L.hrt_io_write_metadata_cache.argtypes = [C.c_char_p, C.c_uint64]
        L.hrt_io_read_vtk_mesh_file.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(C.c_uint64)]
        _LIB = L
    return _LIB


class IoError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise IoError(lib().hrt_io_last_error().decode(errors="replace"))


def _arr(ptr, n, dtype):
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).astype(dtype, copy=True) if n else np.zeros(0, dtype)


def read_stl(path):
    """-> dict(vertices (n,3,3) f32, normals (n,3,3) f32, file_normals (n,3) f32)."""
    m = IoMesh()
    _check(lib().hrt_io_read_stl(os.fsencode(path), C.byref(m)))
    n = m.n_triangles
    out = {"vertices": _arr(m.vertices, 9 * n, np.float32).reshape(n, 3, 3), "normals": _arr(m.normals, 9 * n, np.float32).reshape(n, 3, 3),
           "file_normals": _arr(m.file_normals, 3 * n, np.float32).reshape(n, 3)}
    lib().hrt_io_free_mesh(C.byref(m))
    return out


def read_particle_vtk(path):
    """-> dict(states (n,12) f32 in HrtParticleState layout, ids (n,) u64, shape_ids (n,) u64)."""
    p = IoParticles()
    _check(lib().hrt_io_read_particle_vtk(os.fsencode(path), C.byref(p)))
    out = {"states": _arr(p.states, 12 * p.n, np.float32).reshape(p.n, 12), "ids": _arr(p.ids, p.n, np.uint64), "shape_ids": _arr(p.shape_ids, p.n, np.uint64)}
    lib().hrt_io_free_particles(C.byref(p))
    return out


def read_series(directory, name):
    """-> (list of paths, durations (n,) f32); directory is prepended verbatim as the reference does."""
    s = IoSeries()
    _check(lib().hrt_io_read_series(os.fsencode(directory), os.fsencode(name), C.byref(s)))
    files = [s.files[i].decode() for i in range(s.n)]
    dur = _arr(s.durations, s.n, np.float32)
    lib().hrt_io_free_series(C.byref(s))
    return files, dur


def bake_color_ramp(preset, count):
    out = np.zeros((count, 3), np.float32)
    _check(lib().hrt_io_bake_color_ramp(preset.encode(), count, out.ctypes.data))
    return out


def construct_transform(shift, rotate_deg, scale):
    a, b, c = (np.ascontiguousarray(x, dtype=np.float32) for x in (shift, rotate_deg, scale))
    out = np.zeros(12, np.float32)
    _check(lib().hrt_io_construct_transform(a.ctypes.data, b.ctypes.data, c.ctypes.data, out.ctypes.data))
    return out


def load_config(path):
    c = IoConfig()
    _check(lib().hrt_io_load_config(os.fsencode(path), C.byref(c)))
    out = {"mesh": bool(c.mesh), "cache": bool(c.cache), "debug-mode": bool(c.debug_mode), "api": c.api.decode(), "opengl": bool(c.api_is_opengl),
           "series-path": c.series_path.decode(), "series-name": c.series_name.decode(), "cache-path": c.cache_path.decode(),
           "stl-path": c.stl_path.decode(), "particle-material-preset": c.particle_material_preset.decode(),
           "cache-process-thread-count": int(c.cache_process_thread_count),
           "roughs": _arr(c.roughs, 3 * c.n_roughs, np.float32).reshape(-1, 3), "metals": _arr(c.metals, 4 * c.n_metals, np.float32).reshape(-1, 4),
           "spheres": [{"center": np.array(c.spheres[i].center, np.float32), "radius": float(c.spheres[i].radius), "metal": bool(c.spheres[i].metal),
                        "material_index": int(c.spheres[i].material_index), "transform": np.array(c.spheres[i].transform, np.float32)}
                       for i in range(c.n_spheres)],
           "window": (int(c.window_width), int(c.window_height)), "fps": int(c.fps), "render-speed-ratio": int(c.render_speed_ratio),
           "camera-center": np.array(c.camera_center, np.float32), "camera-target": np.array(c.camera_target, np.float32),
           "up-direction": np.array(c.up_direction, np.float32), "particle-shift": np.array(c.particle_shift, np.float32),
           "particle-scale": np.array(c.particle_scale, np.float32), "mouse-sensitivity": float(c.mouse_sensitivity),
           "camera-pitch-limit-degree": float(c.camera_pitch_limit_degree), "camera-speed-stride": float(c.camera_speed_stride),
           "camera-initial-speed-ratio": int(c.camera_initial_speed_ratio)}
    lib().hrt_io_free_config(C.byref(c))
    return out


def read_mesh_cache(path):
    """Mesh-mode particleN.cache -> list of dict(id, velocity (3,), vertices (t,3,3), normals (t,3,3))."""
    m = IoMeshCache()
    _check(lib().hrt_io_read_mesh_cache(os.fsencode(path), C.byref(m)))
    n = m.n_particles
    first = _arr(m.first_triangle, n + 1, np.uint64)
    total = int(first[-1]) if n else 0
    ids, vel = _arr(m.ids, n, np.uint64), _arr(m.velocities, 3 * n, np.float32).reshape(n, 3)
    v, nm = _arr(m.vertices, 9 * total, np.float32).reshape(total, 3, 3), _arr(m.normals, 9 * total, np.float32).reshape(total, 3, 3)
    lib().hrt_io_free_mesh_cache(C.byref(m))
    return [{"id": int(ids[i]), "velocity": vel[i], "vertices": v[int(first[i]):int(first[i + 1])], "normals": nm[int(first[i]):int(first[i + 1])]}
            for i in range(n)]


def _mesh_cache_to_list(m):
    n = m.n_particles
    first = _arr(m.first_triangle, n + 1, np.uint64)
    total = int(first[-1]) if n else 0
    ids, vel = _arr(m.ids, n, np.uint64), _arr(m.velocities, 3 * n, np.float32).reshape(n, 3)
    v, nm = _arr(m.vertices, 9 * total, np.float32).reshape(total, 3, 3), _arr(m.normals, 9 * total, np.float32).reshape(total, 3, 3)
    return [{"id": int(ids[i]), "velocity": vel[i], "vertices": v[int(first[i]):int(first[i + 1])], "normals": nm[int(first[i]):int(first[i + 1])]}
            for i in range(n)]


def read_vtk_mesh_file(path):
    """Mesh-mode VTK file (triangle strips + CELL_DATA id / vel; VTKReaderImpl.cpp:24-137) -> (particles as read_mesh_cache
    returns them, cell count)."""
    m = IoMeshCache()
    cells = C.c_uint64(0)
    _check(lib().hrt_io_read_vtk_mesh_file(os.fsencode(path), C.byref(m), C.byref(cells)))
    out = _mesh_cache_to_list(m)
    lib().hrt_io_free_mesh_cache(C.byref(m))
    return out, int(cells.value)


def read_metadata_cache(directory):
    """metadata.cache of a cache directory (path with trailing separator, as in config.json's cache-path) -> max cell count."""
    v = C.c_uint64(0)
    _check(lib().hrt_io_read_metadata_cache(os.fsencode(str(directory)), C.byref(v)))
    return int(v.value)


def write_metadata_cache(directory, max_cell_count):
    _check(lib().hrt_io_write_metadata_cache(os.fsencode(str(directory)), int(max_cell_count)))


def write_mesh_cache(path, particles):
    n = len(particles)
    ids = np.array([p["id"] for p in particles], np.uint64)
    vel = np.ascontiguousarray([p["velocity"] for p in particles], dtype=np.float32).reshape(n, 3) if n else np.zeros((0, 3), np.float32)
    first = np.zeros(n + 1, np.uint64)
    for i, p in enumerate(particles):
        first[i + 1] = first[i] + np.uint64(len(p["vertices"]))
    v = np.ascontiguousarray(np.concatenate([np.asarray(p["vertices"], np.float32).reshape(-1, 9) for p in particles]) if n else np.zeros((0, 9), np.float32))
    nm = np.ascontiguousarray(np.concatenate([np.asarray(p["normals"], np.float32).reshape(-1, 9) for p in particles]) if n else np.zeros((0, 9), np.float32))
    m = IoMeshCache(n, ids.ctypes.data_as(C.POINTER(C.c_uint64)), vel.ctypes.data_as(C.POINTER(C.c_float)),
                    first.ctypes.data_as(C.POINTER(C.c_uint64)), v.ctypes.data_as(C.POINTER(C.c_float)), nm.ctypes.data_as(C.POINTER(C.c_float)))
    _check(lib().hrt_io_write_mesh_cache(os.fsencode(path), C.byref(m)))


def time_mode_scene(config_path, base_dir=None, width=None, height=None):
    """What RendererTime::commitRendererData assembles (src/Global/RendererTime.cu:160-290), as a scene dict for
    Renderer.load_scene plus the per-file particle states for Renderer.pose_instances:
      instances = the config's extra spheres (transform = their static matrix, Main.cu:5-9), then one instance per
      particle of file 0 sharing its shape's BLAS, rough albedo = ramp[particle id] (:246-266), identity transform
      until the first pose update (:111-127).
    Paths in the config are relative to the executable's directory in the reference; ``base_dir`` stands for it
    (default: the directory of the config file + "/../bin", i.e. "../files/" resolves next to the config)."""
    from . import scenes
    cfg = load_config(config_path)
    base = Path(base_dir) if base_dir is not None else Path(config_path).resolve().parent.parent / "bin"
    resolve = lambda p: p if os.path.isabs(p) else os.path.normpath(str(base / p)) + ("/" if p.endswith("/") else "")   # noqa: E731
    files, durations = read_series(resolve(cfg["series-path"]), cfg["series-name"])
    stl_dir = Path(resolve(cfg["stl-path"]))
    shapes = [read_stl(p) for p in sorted(stl_dir.iterdir()) if p.is_file() and p.suffix == ".stl"]
    steps = [read_particle_vtk(f) for f in files]
    max_particles = max(len(s["ids"]) for s in steps)
    ramp = bake_color_ramp(cfg["particle-material-preset"], max_particles)
    inst = []
    for sp in cfg["spheres"]:
        if sp["metal"]:
            m = cfg["metals"][sp["material_index"]]
            inst.append(scenes._sphere_instance([sp["center"]], [sp["radius"]], m[:3], "metal", float(m[3]), sp["transform"]))
        else:
            inst.append(scenes._sphere_instance([sp["center"]], [sp["radius"]], cfg["roughs"][sp["material_index"]], "rough", 0.0, sp["transform"]))
    first = steps[0]
    for i in range(len(first["ids"])):
        sid = int(first["shape_ids"][i])
        it = {"geometry": "triangles", "vertices": shapes[sid]["vertices"], "normals": shapes[sid]["normals"], "material": "rough",
              "albedo": ramp[int(first["ids"][i])].copy(), "fuzz": 0.0, "transform": scenes.IDENTITY.copy(), "shape": sid}
        inst.append(it)
    w, h = cfg["window"]
    scene = {"name": "time-mode:" + cfg["series-name"], "instances": inst,
             "camera": {"center": cfg["camera-center"], "target": cfg["camera-target"], "up": cfg["up-direction"], "opengl": cfg["opengl"]},
             "background": scenes.BACKGROUND.copy(), "width": width or w, "height": height or h, "spp": 1}
    frames = [int(np.float32(d) * np.float32(cfg["fps"] * cfg["render-speed-ratio"])) for d in durations]   # RendererTime.cu:427-428
    return {"scene": scene, "config": cfg, "states": [s["states"] for s in steps], "durations": durations, "frame_counts": frames,
            "n_extra": len(cfg["spheres"]), "shapes": shapes, "ramp": ramp}


def mesh_mode_scene(config_path, base_dir=None, width=None, height=None):
    """What RendererMesh::commitRendererData assembles (src/Global/RendererMesh.cu:160-310), one scene dict per VTK file for
    Renderer.load_scene plus the drift velocities for Renderer.pose_instances(..., mesh_mode=True):
      instances = the config's extra spheres (their static matrices), then one instance per particle of the file with its OWN
      geometry (one GAS per particle, :107-114), rough albedo = ramp[particle id] where the ramp has metadata.cache's
      max-cell-count colours (:222-232), identity transform until the first frame."""
    from . import scenes
    cfg = load_config(config_path)
    base = Path(base_dir) if base_dir is not None else Path(config_path).resolve().parent.parent / "bin"
    resolve = lambda p: p if os.path.isabs(p) else os.path.normpath(str(base / p)) + ("/" if p.endswith("/") else "")   # noqa: E731
    files, durations = read_series(resolve(cfg["series-path"]), cfg["series-name"])
    cache_dir = resolve(cfg["cache-path"])
    ramp = bake_color_ramp(cfg["particle-material-preset"], read_metadata_cache(cache_dir))
    extra = []
    for sp in cfg["spheres"]:
        if sp["metal"]:
            m = cfg["metals"][sp["material_index"]]
            extra.append(scenes._sphere_instance([sp["center"]], [sp["radius"]], m[:3], "metal", float(m[3]), sp["transform"]))
        else:
            extra.append(scenes._sphere_instance([sp["center"]], [sp["radius"]], cfg["roughs"][sp["material_index"]], "rough", 0.0, sp["transform"]))
    w, h = cfg["window"]
    out_scenes, velocities = [], []
    for k in range(len(files)):
        particles = read_mesh_cache(os.path.join(cache_dir, f"particle{k}.cache"))
        inst = [dict(e) for e in extra]
        for p in particles:
            inst.append({"geometry": "triangles", "vertices": p["vertices"], "normals": p["normals"], "material": "rough",
                         "albedo": ramp[p["id"]].copy(), "fuzz": 0.0, "transform": scenes.IDENTITY.copy()})
        out_scenes.append({"name": f"mesh-mode:{cfg['series-name']}:{k}", "instances": inst,
                           "camera": {"center": cfg["camera-center"], "target": cfg["camera-target"], "up": cfg["up-direction"], "opengl": cfg["opengl"]},
                           "background": scenes.BACKGROUND.copy(), "width": width or w, "height": height or h, "spp": 1})
        velocities.append(np.array([p["velocity"] for p in particles], np.float32).reshape(-1, 3))
    frames = [int(np.float32(d) * np.float32(cfg["fps"] * cfg["render-speed-ratio"])) for d in durations]   # RendererMesh.cu:366-367
    return {"scenes": out_scenes, "config": cfg, "velocities": velocities, "durations": durations, "frame_counts": frames, "n_extra": len(cfg["spheres"])}


def write_mesh_mode_sample(root, n_files=3, n_particles=12, seed=11, width=160, height=120, as_vtk=False):
    """A small synthetic Mesh-mode data set in the reference's on-disk formats under `root` (the reference ships only a Time-mode
    sample): bin/ (the "executable's" directory the config's paths are relative to), files/mesh.vtk.series, cache/particleN.cache +
    cache/metadata.cache -- or, with as_vtk, files/mesh_N.vtk (triangle strips with CELL_DATA id / vel) for the cache run to convert --
    and config.json.  Particles: small closed blobs (cache form) or zigzag ribbons of 6 points = 4 triangles (VTK form: one strip
    per particle) that move between the files and drift within one; a ground sphere from the config.  Returns the config path."""
    import json
    from . import scenes
    root = Path(root)
    for d in ("bin", "files", "cache"):
        (root / d).mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(seed)
    base_pos = rng.uniform(-1.2, 1.2, (n_particles, 3)).astype(np.float32) * np.array([1, 1, 0.4], np.float32)
    vel = rng.uniform(-0.6, 0.6, (n_particles, 3)).astype(np.float32)
    shapes = [scenes._blob_shape(1, 0.12 + 0.02 * (i % 4), seed + i) for i in range(n_particles)]
    times = [0.0]
    for k in range(n_files):
        times.append(times[-1] + 0.01 * (k + 1))
    max_cells = 0
    for k in range(n_files):
        particles = []
        ids = rng.permutation(n_particles)                              # the files list the particles in no particular order
        for i in range(n_particles):
            v = (shapes[i] + (base_pos[i] + vel[i] * np.float32(times[k]))).astype(np.float32)
            particles.append({"id": int(ids[i]), "velocity": vel[i], "vertices": v, "normals": scenes.face_normals(v)})
        max_cells = max(max_cells, int(ids.max()) + 1)
        if as_vtk:
            ribbons = []
            for i in range(n_particles):
                j = np.arange(6, dtype=np.float32)
                pts = np.stack([0.08 * j, 0.1 * (j % 2), 0.02 * j * (i % 3)], axis=1).astype(np.float32) + (base_pos[i] + vel[i] * np.float32(times[k])).astype(np.float32)
                ribbons.append({"id": int(ids[i]), "velocity": vel[i], "points": pts})
            _write_vtk_mesh_file(root / "files" / f"mesh_{k}.vtk", ribbons)
        else:
            write_mesh_cache(str(root / "cache" / f"particle{k}.cache"), particles)
    if not as_vtk:
        write_metadata_cache(str(root / "cache") + "/", max_cells)
    series = {"file-series-version": "1.0", "files": [{"name": f"mesh_{k}.vtk", "time": times[k]} for k in range(n_files)]}
    (root / "files" / "mesh.vtk.series").write_text(json.dumps(series, indent=1))
    cfg = {"mesh": True, "series-path": "../files/", "series-name": "mesh.vtk.series", "cache-path": "../cache/", "stl-path": "../files/",
           "cache": bool(as_vtk), "debug-mode": False, "cache-process-thread-count": 3, "particle-material-preset": "viridis",
           "roughs": [{"albedo": [0.70, 0.60, 0.50]}], "metals": [{"albedo": [0.8, 0.85, 0.88], "fuzz": 0.1}],
           "spheres": [{"center": [0.0, 0.0, 0.0], "radius": 100.0, "mat-type": "ROUGH", "mat-index": 0, "shift": [0.0, 0.0, -100.6],
                        "rotate": [0.0, 0.0, 0.0], "scale": [1.0, 1.0, 1.0]}],
           "triangles": [],
           "loop-data": {"api": "VK", "window-width": width, "window-height": height, "fps": 100, "camera-center": [4.0, 0.5, 1.0],
                         "camera-target": [0.0, 0.0, 0.0], "up-direction": [0.0, 0.0, 1.0], "camera-pitch-limit-degree": 85.0,
                         "camera-speed-stride": 0.002, "camera-initial-speed-ratio": 10, "mouse-sensitivity": 0.002, "render-speed-ratio": 3,
                         "particle-shift": [0.1, 0.0, 0.05], "particle-scale": [1.0, 1.0, 1.0]}}
    path = root / "files" / "config.json"
    path.write_text(json.dumps(cfg, indent=1))
    return path


def _write_vtk_mesh_file(path, ribbons):
    """Legacy ASCII POLYDATA with one TRIANGLE_STRIP per particle and CELL_DATA id / vel: the input of the cache run
    (vtk_reader::readVTKMeshFile, src/Util/VTKReaderImpl.cpp:24-137).  ribbons: dict(id, velocity, points (m,3)): strip k of m points
    is m - 2 triangles."""
    pts, strips = [], []
    for r in ribbons:
        b = len(pts)
        pts.extend(np.asarray(r["points"], np.float32).tolist())
        strips.append(list(range(b, len(pts))))
    with open(path, "w") as f:
        f.write("# vtk DataFile Version 3.0\nmesh-mode sample\nASCII\nDATASET POLYDATA\n")
        f.write(f"POINTS {len(pts)} float\n")
        for q in pts:
            f.write("%.9g %.9g %.9g\n" % tuple(q))
        f.write(f"TRIANGLE_STRIPS {len(strips)} {sum(len(s) + 1 for s in strips)}\n")
        for s in strips:
            f.write(" ".join(str(x) for x in [len(s)] + s) + "\n")
        f.write(f"CELL_DATA {len(strips)}\nSCALARS id int 1\nLOOKUP_TABLE default\n")
        f.write("\n".join(str(int(r["id"])) for r in ribbons) + "\n")
        f.write("VECTORS vel float\n")
        for r in ribbons:
            f.write("%.9g %.9g %.9g\n" % tuple(np.asarray(r["velocity"], np.float32)))
