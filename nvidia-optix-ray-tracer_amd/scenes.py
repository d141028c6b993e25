"""Synthetic scenes of BASELINE.json / SURVEY.md 8(d) (the reference ships no benchmark scenes).

All generators are deterministic: SplitMix64 with a documented seed, float32 output.  Normals
are the normalised face normal replicated three times, and every instance has the identity
transform unless a test asks otherwise, so the reference's quirks Q1/Q2/Q5 do not matter.

A scene is a plain dict::

    {"instances": [ {geometry: "triangles"|"spheres", vertices (n,3,3) f32, normals (n,3,3) f32,
                     centers (n,3) f32, radii (n,) f32, material: "rough"|"metal",
                     albedo (3,) f32, fuzz float, transform (12,) f32}, ... ],
     "camera": {center, target, up  (3,) f32, "opengl": True},
     "background": (3,) f32, "width": int, "height": int, "spp": int, "name": str}
"""
from __future__ import annotations

import numpy as np

SEED_SALT = 0x5EED0000C0FFEE          # replaces clock64() in the reference's curand_init (quirk Q8)
BACKGROUND = np.array([0.7, 0.8, 0.9], dtype=np.float32)   # src/Global/RendererMesh.cu:262
IDENTITY = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float32)

# rough albedos of the reference's shipped config (files/config.json:11-16)
RED = np.array([0.65, 0.05, 0.05], dtype=np.float32)
WHITE = np.array([0.73, 0.73, 0.73], dtype=np.float32)
GREEN = np.array([0.12, 0.45, 0.15], dtype=np.float32)
SAND = np.array([0.70, 0.60, 0.50], dtype=np.float32)
STEEL = np.array([0.8, 0.85, 0.88], dtype=np.float32)       # files/config.json:18


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n outputs of SplitMix64 started at ``seed`` (vectorised; uint64 wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform_f32(seed: int, n: int, lo: float, hi: float) -> np.ndarray:
    """n float32 values in [lo, hi): top 24 bits of SplitMix64 scaled by 2^-24."""
    u = (splitmix64(seed, n) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    return (np.float32(lo) + (np.float32(hi) - np.float32(lo)) * u).astype(np.float32)


def face_normals(vertices: np.ndarray) -> np.ndarray:
    """(n,3,3) vertices -> (n,3,3) normals: normalised face normal replicated per vertex."""
    v = vertices.astype(np.float32)
    e1 = v[:, 1] - v[:, 0]
    e2 = v[:, 2] - v[:, 0]
    n = np.cross(e1, e2).astype(np.float32)
    with np.errstate(over="ignore"):          # huge triangles (growing_chain): the squares overflow, the normal becomes 0 -- for everyone alike
        ln = np.sqrt((n * n).sum(axis=1, dtype=np.float32)).astype(np.float32)
    ok = ln > 0
    n[ok] = (n[ok] / ln[ok, None]).astype(np.float32)
    n[~ok] = np.array([0, 0, 1], dtype=np.float32)
    return np.repeat(n[:, None, :], 3, axis=1).astype(np.float32).copy()


def _tri_instance(vertices, albedo, material="rough", fuzz=0.0, transform=None):
    vertices = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3, 3)
    return {"geometry": "triangles", "vertices": vertices, "normals": face_normals(vertices),
            "material": material, "albedo": np.asarray(albedo, dtype=np.float32), "fuzz": float(fuzz),
            "transform": IDENTITY.copy() if transform is None else np.asarray(transform, dtype=np.float32)}


def _sphere_instance(centers, radii, albedo, material="rough", fuzz=0.0, transform=None):
    return {"geometry": "spheres", "centers": np.ascontiguousarray(centers, dtype=np.float32).reshape(-1, 3),
            "radii": np.ascontiguousarray(radii, dtype=np.float32).reshape(-1),
            "material": material, "albedo": np.asarray(albedo, dtype=np.float32), "fuzz": float(fuzz),
            "transform": IDENTITY.copy() if transform is None else np.asarray(transform, dtype=np.float32)}


def _quad(a, b, c, d):
    return [[a, b, c], [a, c, d]]


def _box(lo, hi, skip_bottom=True):
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    p = lambda x, y, z: [x, y, z]  # noqa: E731
    tris = []
    tris += _quad(p(x0, y1, z0), p(x1, y1, z0), p(x1, y1, z1), p(x0, y1, z1))   # top
    tris += _quad(p(x0, y0, z0), p(x1, y0, z0), p(x1, y1, z0), p(x0, y1, z0))   # front
    tris += _quad(p(x0, y0, z1), p(x0, y1, z1), p(x1, y1, z1), p(x1, y0, z1))   # back
    tris += _quad(p(x0, y0, z0), p(x0, y1, z0), p(x0, y1, z1), p(x0, y0, z1))   # left
    tris += _quad(p(x1, y0, z0), p(x1, y0, z1), p(x1, y1, z1), p(x1, y1, z0))   # right
    if not skip_bottom:
        tris += _quad(p(x0, y0, z0), p(x0, y0, z1), p(x1, y0, z1), p(x1, y0, z0))
    return tris


def _cornell_walls():
    """Unit Cornell box, open towards -z.  Returns (white, red, green) triangle lists."""
    p = lambda x, y, z: [x, y, z]  # noqa: E731
    white = []
    white += _quad(p(0, 0, 0), p(1, 0, 0), p(1, 0, 1), p(0, 0, 1))      # floor
    white += _quad(p(0, 1, 0), p(0, 1, 1), p(1, 1, 1), p(1, 1, 0))      # ceiling
    white += _quad(p(0, 0, 1), p(1, 0, 1), p(1, 1, 1), p(0, 1, 1))      # back wall
    red = _quad(p(0, 0, 0), p(0, 0, 1), p(0, 1, 1), p(0, 1, 0))         # left wall
    green = _quad(p(1, 0, 0), p(1, 1, 0), p(1, 1, 1), p(1, 0, 1))       # right wall
    return white, red, green


def _cornell_camera():
    # (278,273,-800) -> (278,273,0) scaled by 1/555; |W| = 1/tan(20 deg) for a 40 degree vertical field of view
    center = np.array([278.0 / 555.0, 273.0 / 555.0, -800.0 / 555.0], dtype=np.float32)
    w = np.float32(1.0 / np.tan(np.deg2rad(20.0)))
    target = (center + np.array([0, 0, w], dtype=np.float32)).astype(np.float32)
    return {"center": center, "target": target, "up": np.array([0, 1, 0], dtype=np.float32), "opengl": True}


def cornell_box(width=256, height=256, spp=1):
    """C1: Cornell box, 32 triangles (5 walls x 2, two boxes x 10, ceiling panel x 2), all rough.
    The reference has no emitters: light comes from the miss colour through the open front."""
    white, red, green = _cornell_walls()
    white = list(white)
    white += _box((0.13, 0.0, 0.12), (0.43, 0.30, 0.42))               # short box
    white += _box((0.52, 0.0, 0.45), (0.82, 0.60, 0.75))               # tall box
    white += _quad([0.35, 0.999, 0.35], [0.35, 0.999, 0.65], [0.65, 0.999, 0.65], [0.65, 0.999, 0.35])   # ceiling panel
    inst = [_tri_instance(white, WHITE), _tri_instance(red, RED), _tri_instance(green, GREEN)]
    assert sum(len(i["vertices"]) for i in inst) == 32
    return {"name": "C1-cornell-32tri", "instances": inst, "camera": _cornell_camera(), "background": BACKGROUND.copy(),
            "width": width, "height": height, "spp": spp}


def sphere_in_box(width=512, height=512, spp=16):
    """C2: one rough sphere r = 0.3 at the box centre + the C1 walls and ceiling panel (12 triangles)."""
    white, red, green = _cornell_walls()
    white = list(white) + _quad([0.35, 0.999, 0.35], [0.35, 0.999, 0.65], [0.65, 0.999, 0.65], [0.65, 0.999, 0.35])
    inst = [_tri_instance(white, WHITE), _tri_instance(red, RED), _tri_instance(green, GREEN),
            _sphere_instance([[0.5, 0.5, 0.5]], [0.3], WHITE)]
    assert sum(len(i["vertices"]) for i in inst if i["geometry"] == "triangles") == 12
    return {"name": "C2-sphere-in-box", "instances": inst, "camera": _cornell_camera(), "background": BACKGROUND.copy(),
            "width": width, "height": height, "spp": spp}


def _random_triangles(n, edge, seed):
    c = uniform_f32(seed * 1000 + 1, 3 * n, -1.0, 1.0).reshape(n, 3)
    e1 = uniform_f32(seed * 1000 + 2, 3 * n, -edge, edge).reshape(n, 3)
    e2 = uniform_f32(seed * 1000 + 3, 3 * n, -edge, edge).reshape(n, 3)
    third = np.float32(1.0 / 3.0)
    v0 = (c - (e1 + e2) * third).astype(np.float32)          # centroid stays at c
    return np.stack([v0, (v0 + e1).astype(np.float32), (v0 + e2).astype(np.float32)], axis=1).astype(np.float32)


def growing_chain(n_triangles=210, growth=1.1, width=96, height=64, spp=2):
    """A deep tree on purpose: triangles whose size and distance from the origin grow geometrically, so that every SAH split takes
    one triangle off the rest.  210 triangles at 1.1 give a BVH8 of 12 levels below the root with the host builder (the most
    the path kernel's LDS node stack holds), 300 give 13 (round 1's kernel takes over).  One metal material with fuzz."""
    k = np.arange(n_triangles, dtype=np.float64)
    s = 0.004 * growth ** k
    c = np.stack([3.0 * s, 0.3 * s, 0.0 * s], axis=1)
    v = np.stack([c + np.stack([-s, -s, 0.0 * s], 1), c + np.stack([s, -s, 0.0 * s], 1), c + np.stack([0.0 * s, s, 0.2 * s], 1)], axis=1)
    cam = {"center": np.array([-20, 5, 40], dtype=np.float32), "target": np.array([60, 6, 0], dtype=np.float32),
           "up": np.array([0, 1, 0], dtype=np.float32), "opengl": True}
    return {"name": f"chain-{n_triangles}", "instances": [_tri_instance(v, WHITE, material="metal", fuzz=0.3)], "camera": cam,
            "background": BACKGROUND.copy(), "width": width, "height": height, "spp": spp}


def _soup_camera():
    return {"center": np.array([0, 0, 3.5], dtype=np.float32), "target": np.array([0, 0, 0], dtype=np.float32),
            "up": np.array([0, 1, 0], dtype=np.float32), "opengl": True}


def random_soup(n_triangles=100_000, edge=0.03, seed=1, width=1920, height=1080, spp=64, name=None):
    """C3 / C4: n random triangles, centroid ~ U[-1,1]^3, two edge vectors ~ U[-edge,edge]^3, one rough material."""
    v = _random_triangles(n_triangles, edge, seed)
    return {"name": name or f"soup-{n_triangles}", "instances": [_tri_instance(v, WHITE)], "camera": _soup_camera(),
            "background": BACKGROUND.copy(), "width": width, "height": height, "spp": spp}


def soup_100k(width=1920, height=1080, spp=64):
    return random_soup(100_000, 0.03, 1, width, height, spp, "C3-soup-100k")


def soup_1m(width=1920, height=1080, spp=256):
    return random_soup(1_000_000, 0.014, 2, width, height, spp, "C4-soup-1M")


def soup_law_edge(n_triangles):
    """Edge scale of the soup law C3 / C4 follow: constant expected overlap, edge ~ n^(-1/3) (0.03 at 100 k, 0.014 at 1 M)."""
    return round(0.03 * (1.0e5 / n_triangles) ** (1.0 / 3.0), 4)


def soup_large(n_triangles, width=1920, height=1080, spp=16):
    """C4's law at 8 M / 32 M triangles: tree + records exceed the 256 MiB Infinity Cache (the out-of-cache datapoints
    of profiles/r03_large_scenes.txt).  Seed = 2 + log2(n / 1 M)."""
    seed = 2 + max(0, int(round(np.log2(n_triangles / 1.0e6))))
    return random_soup(n_triangles, soup_law_edge(n_triangles), seed, width, height, spp, f"soup-{n_triangles // 1_000_000}M")


def soup_1m_8mat(width=1920, height=1080, spp=1024, n_triangles=1_000_000):
    """C5: the C4 geometry dealt round-robin into 8 instances: 4 rough (config.json albedos) +
    4 metal (fuzz 0, 0.1, 0.3, 0.5; albedo 0.8, 0.85, 0.88).  The reference has neither glass nor
    emitters / light sampling, so BASELINE.json's "glass" and "NEE" have no counterpart here."""
    v = _random_triangles(n_triangles, 0.014, 2)
    inst = []
    for k, alb in enumerate([RED, WHITE, GREEN, SAND]):
        inst.append(_tri_instance(v[k::8], alb))
    for k, fz in enumerate([0.0, 0.1, 0.3, 0.5]):
        inst.append(_tri_instance(v[4 + k::8], STEEL, "metal", fz))
    return {"name": "C5-soup-1M-8mat", "instances": inst, "camera": _soup_camera(), "background": BACKGROUND.copy(),
            "width": width, "height": height, "spp": spp}


def mixed_test_scene(n_triangles=2000, n_spheres=40, seed=7, width=96, height=64, spp=2, transforms=True):
    """Small scene that exercises all four programs, several instances, non-identity transforms
    (quirks Q1/Q2 included) and spheres sharing one BLAS: parity-test material, not a benchmark."""
    v = _random_triangles(n_triangles, 0.12, seed)
    c = uniform_f32(seed * 1000 + 11, 3 * n_spheres, -0.9, 0.9).reshape(n_spheres, 3)
    r = uniform_f32(seed * 1000 + 12, n_spheres, 0.03, 0.15)
    shift = np.array([1, 0, 0, 0.15, 0, 1, 0, -0.1, 0, 0, 1, 0.05], dtype=np.float32)
    ang = np.float32(0.3)
    rot = np.array([np.cos(ang), -np.sin(ang), 0, 0.05, np.sin(ang), np.cos(ang), 0, 0, 0, 0, 1.25, -0.1], dtype=np.float32)
    h = n_spheres // 2
    inst = [
        _tri_instance(v[0::3], RED),
        _tri_instance(v[1::3], STEEL, "metal", 0.0, rot if transforms else None),
        _tri_instance(v[2::3], STEEL, "metal", 0.3),
        _sphere_instance(c[:h], r[:h], GREEN, "rough", 0.0, shift if transforms else None),
        _sphere_instance(c[h:], r[h:], STEEL, "metal", 0.2),
    ]
    return {"name": "mixed-test", "instances": inst, "camera": _soup_camera(), "background": BACKGROUND.copy(),
            "width": width, "height": height, "spp": spp}


def _blob_shape(subdiv, radius, seed):
    """A closed, lumpy triangle mesh: an octahedron subdivided ``subdiv`` times, vertices pushed out to
    ``radius`` x (1 +- 25 % noise keyed on the vertex direction) -- a stand-in for the reference's STL particles."""
    o = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float64)
    faces = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    tris = np.array([[o[a], o[b], o[c]] for a, b, c in faces])
    for _ in range(subdiv):
        a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
        ab, bc, ca = (a + b) / 2, (b + c) / 2, (c + a) / 2
        tris = np.concatenate([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)])
    d = tris / np.linalg.norm(tris, axis=2, keepdims=True)
    k = np.array([1.7, 2.3, 2.9]) + seed
    bump = 1.0 + 0.25 * np.sin((d * k).sum(axis=2) * 3.0)
    return (d * bump[..., None] * radius).astype(np.float32)


def rigid_transform(position, axis, angle, scale=1.0):
    """3x4 row-major transform: rotation by ``angle`` (radians) about ``axis``, uniform scale, then translation."""
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    x, y, z = axis
    c, s_ = np.cos(angle), np.sin(angle)
    r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s_, x * z * (1 - c) + y * s_],
                  [y * x * (1 - c) + z * s_, c + y * y * (1 - c), y * z * (1 - c) - x * s_],
                  [z * x * (1 - c) - y * s_, z * y * (1 - c) + x * s_, c + z * z * (1 - c)]]) * scale
    m = np.concatenate([r, np.asarray(position, dtype=np.float64).reshape(3, 1)], axis=1)
    return m.astype(np.float32).reshape(12)


def particle_poses(n_particles, frame, seed=5):
    """Transforms of the particles of ``particle_scene`` at an animation frame: a 5-wide grid 0.2 apart (the layout
    of the reference's files/particle/*.vtk), falling along -z and tumbling, as in its Time mode."""
    ax = uniform_f32(seed * 77 + 1, 3 * n_particles, -1.0, 1.0).reshape(n_particles, 3) + np.float32(1e-3)
    w = uniform_f32(seed * 77 + 2, n_particles, -0.4, 0.4)
    out = []
    for i in range(n_particles):
        gx, gy = i % 5, (i // 5) % 5
        layer = i // 25
        pos = (-0.4 + 0.2 * gx + 0.01 * np.sin(i), -0.4 + 0.2 * gy, 0.4 + 0.2 * layer - 0.02 * frame)
        out.append(rigid_transform(pos, ax[i], float(w[i]) * frame + 0.1 * i))
    return out


def particle_states(n_particles, step, seed=5):
    """(n, 12) float32 particle states of time step ``step`` in HrtParticleState layout (quat.xyzw as the four
    file components, position, velocity, 2 pad): the 5-wide grid of files/particle/*.vtk, velocity (0, 0, -2) as in
    that data, unit quaternions drifting from step to step."""
    st = np.zeros((n_particles, 12), dtype=np.float32)
    a = uniform_f32(seed * 131 + 3, 4 * n_particles, -1.0, 1.0).reshape(n_particles, 4)
    b = uniform_f32(seed * 131 + 4, 4 * n_particles, -1.0, 1.0).reshape(n_particles, 4)
    q = a + np.float32(0.35 * step) * b
    q = q / np.sqrt((q.astype(np.float64) ** 2).sum(axis=1, keepdims=True))
    st[:, 0:4] = q.astype(np.float32)
    i = np.arange(n_particles)
    st[:, 4] = -0.4 + 0.2 * (i % 5)
    st[:, 5] = -0.4 + 0.2 * ((i // 5) % 5)
    st[:, 6] = 0.4 + 0.2 * (i // 25) - 0.02 * step
    st[:, 9] = -2.0
    return st


def particle_scene(n_particles=25, width=96, height=64, spp=1, frame=0, subdiv=2, seed=5):
    """The structure of the reference's Time-mode scenes (files/config.json, files/particle/*.vtk): the extra
    geometry first (a huge ground sphere shifted by its instance transform, quirk Q1), then the particles
    (RendererTime.cu:436: instances [addGeoCount, instanceCount)) instancing a few shared shapes (``shape`` =
    BLAS to share, RendererTime.cu:116-130), one transform each per frame."""
    shapes = [_blob_shape(subdiv, 0.06, seed), _blob_shape(max(subdiv - 1, 0), 0.05, seed + 1),
              np.asarray(_box((-0.04, -0.04, -0.04), (0.04, 0.04, 0.04), skip_bottom=False), dtype=np.float32)]
    albedos = [RED, WHITE, GREEN, SAND]
    poses = particle_poses(n_particles, frame, seed)
    ground = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, -1000.5], dtype=np.float32)       # files/config.json:26
    inst = [_sphere_instance([[0, 0, 0]], [1000.0], SAND, "rough", 0.0, ground)]
    for i in range(n_particles):
        sid = i % len(shapes)
        metal = i % 4 == 3
        it = _tri_instance(shapes[sid], STEEL if metal else albedos[i % 4], "metal" if metal else "rough", 0.1 if metal else 0.0, poses[i])
        it["shape"] = sid
        inst.append(it)
    cam = {"center": np.array([2.2, 0.3, 0.9], dtype=np.float32), "target": np.array([0, 0, 0.2], dtype=np.float32),
           "up": np.array([0, 0, 1], dtype=np.float32), "opengl": False}
    return {"name": "particles-%d" % n_particles, "instances": inst, "camera": cam, "background": BACKGROUND.copy(),
            "width": width, "height": height, "spp": spp}


def particle_cloud(n_particles=2000, width=96, height=64, spp=1, subdiv=2, seed=9, spacing=0.2):
    """A DEM-sized scene of the reference's kind: ``n_particles`` instances of three shared shapes on a jittered cubic grid (``spacing``
    apart, randomly rotated) over the ground sphere -- what ``particle_scene`` is to the shipped 25-particle sample, this is to a run
    with 10^3 .. 10^5 particles.  Flattened it has ~57 triangles per particle at subdiv 2 (~217 at subdiv 3); as an IAS over three GASes it has the three shapes."""
    shapes = [_blob_shape(subdiv, 0.06, seed), _blob_shape(max(subdiv - 1, 0), 0.05, seed + 1),
              np.asarray(_box((-0.04, -0.04, -0.04), (0.04, 0.04, 0.04), skip_bottom=False), dtype=np.float32)]
    normals = [face_normals(v) for v in shapes]
    side = int(np.ceil(n_particles ** (1.0 / 3.0)))
    i = np.arange(n_particles)
    gx, gy, gz = i % side, (i // side) % side, i // (side * side)
    jit = uniform_f32(seed * 53 + 1, 3 * n_particles, -0.25, 0.25).reshape(n_particles, 3) * np.float32(spacing)
    half = 0.5 * spacing * (side - 1)
    pos = np.stack([gx * spacing - half, gy * spacing - half, gz * spacing + 0.1], axis=1).astype(np.float64) + jit
    ax = uniform_f32(seed * 53 + 2, 3 * n_particles, -1.0, 1.0).reshape(n_particles, 3).astype(np.float64) + 1e-3
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    ang = uniform_f32(seed * 53 + 3, n_particles, 0.0, 6.28).astype(np.float64)
    c, s_ = np.cos(ang), np.sin(ang)
    x, y, z = ax[:, 0], ax[:, 1], ax[:, 2]
    rot = np.stack([c + x * x * (1 - c), x * y * (1 - c) - z * s_, x * z * (1 - c) + y * s_, pos[:, 0],
                    y * x * (1 - c) + z * s_, c + y * y * (1 - c), y * z * (1 - c) - x * s_, pos[:, 1],
                    z * x * (1 - c) - y * s_, z * y * (1 - c) + x * s_, c + z * z * (1 - c), pos[:, 2]], axis=1).astype(np.float32)
    albedos = [RED, WHITE, GREEN, SAND]
    ground = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, -1000.5], dtype=np.float32)
    inst = [_sphere_instance([[0, 0, 0]], [1000.0], SAND, "rough", 0.0, ground)]
    for k in range(n_particles):
        sid = k % len(shapes)
        metal = k % 4 == 3
        inst.append({"geometry": "triangles", "vertices": shapes[sid], "normals": normals[sid], "albedo": STEEL if metal else albedos[k % 4],
                     "material": "metal" if metal else "rough", "fuzz": 0.1 if metal else 0.0, "transform": rot[k], "shape": sid})
    ext = spacing * side
    cam = {"center": np.array([1.6 * ext + 1.0, 0.35 * ext, 0.9 * ext + 0.4], dtype=np.float32), "target": np.array([0, 0, 0.45 * ext], dtype=np.float32),
           "up": np.array([0, 0, 1], dtype=np.float32), "opengl": False}
    return {"name": "cloud-%d" % n_particles, "instances": inst, "camera": cam, "background": BACKGROUND.copy(),
            "width": width, "height": height, "spp": spp}


BASELINE_CONFIGS = {"C1": cornell_box, "C2": sphere_in_box, "C3": soup_100k, "C4": soup_1m, "C5": soup_1m_8mat}
