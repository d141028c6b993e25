"""ctypes wrapper around oracle/liboracle.so -- TEST INFRASTRUCTURE (see oracle/oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "oracle" / "liboracle.so"


class OracleInstance(C.Structure):
    _fields_ = [("transform", C.c_float * 12), ("geometry", C.c_int32), ("material", C.c_int32),
                ("albedo", C.c_float * 3), ("fuzz", C.c_float), ("n_prims", C.c_uint32),
                ("vertices", C.c_void_p), ("normals", C.c_void_p), ("centers", C.c_void_p), ("radii", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB.exists():
        subprocess.check_call(["make", "-C", str(ROOT / "oracle")])
    import os
    L = C.CDLL(os.environ.get("HRT_ORACLE_LIB") or str(LIB))      # HRT_ORACLE_LIB: the sanitizer build (make asan-test)
    L.oracle_scene_create.restype = C.c_void_p
    L.oracle_scene_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.oracle_scene_create_mode.restype = C.c_void_p
    L.oracle_scene_create_mode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.oracle_scene_destroy.argtypes = [C.c_void_p]
    L.oracle_scene_attach_bvh8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_render.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_bvh8_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                    C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_rng_init.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    L.oracle_rng_init_one.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    L.oracle_rng_init_generic.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.oracle_rng_uniform.restype = C.c_float
    L.oracle_rng_uniform.argtypes = [C.c_void_p]
    L.oracle_rng_next.restype = C.c_uint32
    L.oracle_rng_next.argtypes = [C.c_void_p]
    L.oracle_color_to_float4.argtypes = [C.c_void_p, C.c_void_p]
    L.oracle_color_to_uchar4.argtypes = [C.c_void_p, C.c_void_p]
    L.oracle_to_rgba8.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    L.oracle_color_to_float4_n.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.oracle_pow_inv_gamma_bits.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p]
    L.oracle_pow_inv_gamma.restype = C.c_float
    L.oracle_pow_inv_gamma.argtypes = [C.c_float]
    L.oracle_pow_inv_gamma_libm_powf.restype = C.c_float
    L.oracle_pow_inv_gamma_libm_powf.argtypes = [C.c_float]
    L.oracle_configure_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_construct_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_slerp.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    L.oracle_quat_to_euler.argtypes = [C.c_void_p, C.c_void_p]
    L.oracle_pose_transforms.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_trig.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
    L.oracle_trig_bits.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p]
    L.oracle_trig_libm.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.oracle_num_threads.restype = C.c_int
    L.oracle_set_threads.argtypes = [C.c_int]
    L.oracle_init()
    _lib = L
    return L


def _p(a):
    return a.ctypes.data if a is not None else None


class OracleScene:
    """Oracle-side scene built from the same dict the product's Renderer.load_scene takes.
    instanced=True: the INSTANCED canonical mode (oracle.c: rays go into the instance's object space, as at an IAS leaf) -- what a
    two-level TLAS of the product is pinned to; the default is the FLATTENED mode (triangles transformed to world space)."""

    def __init__(self, scene, force_brute=False, instanced=False):
        L = lib()
        insts = scene["instances"]
        self._keep = []
        arr = (OracleInstance * max(len(insts), 1))()
        for i, it in enumerate(insts):
            for k in range(12):
                arr[i].transform[k] = float(it["transform"][k])
            arr[i].material = 0 if it["material"] == "rough" else 1
            for k in range(3):
                arr[i].albedo[k] = float(it["albedo"][k])
            arr[i].fuzz = float(it["fuzz"]) if it["material"] == "metal" else 0.0
            if it["geometry"] == "triangles":
                v = np.ascontiguousarray(it["vertices"], dtype=np.float32)
                n = np.ascontiguousarray(it["normals"], dtype=np.float32)
                self._keep += [v, n]
                arr[i].geometry = 1
                arr[i].n_prims = v.shape[0]
                arr[i].vertices = v.ctypes.data
                arr[i].normals = n.ctypes.data
            else:
                c = np.ascontiguousarray(it["centers"], dtype=np.float32)
                r = np.ascontiguousarray(it["radii"], dtype=np.float32)
                self._keep += [c, r]
                arr[i].geometry = 0
                arr[i].n_prims = r.shape[0]
                arr[i].centers = c.ctypes.data
                arr[i].radii = r.ctypes.data
        self._arr = arr
        self.scene = scene
        self.handle = C.c_void_p(L.oracle_scene_create_mode(arr, len(insts), int(force_brute), int(instanced)))

    def camera12(self):
        L = lib()
        cam = self.scene["camera"]
        c = np.ascontiguousarray(cam["center"], dtype=np.float32)
        t = np.ascontiguousarray(cam["target"], dtype=np.float32)
        up = np.ascontiguousarray(cam["up"], dtype=np.float32)
        out = np.zeros(12, dtype=np.float32)
        out[0:3] = c
        u, v, w = out[3:6], out[6:9], out[9:12]
        uu, vv, ww = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
        L.oracle_configure_camera(_p(c), _p(t), _p(up), int(cam.get("opengl", True)), _p(uu), _p(vv), _p(ww))
        u[:], v[:], w[:] = uu, vv, ww
        return out

    def render(self, width, height, states, spp=1, rows=None, want_linear=True):
        """states: (H*W, 12) uint32, updated in place.  Returns dict of (H,W,4) float32 arrays + counters."""
        L = lib()
        cam = self.camera12()
        bg = np.ascontiguousarray(self.scene["background"], dtype=np.float32)
        color = np.zeros((height, width, 4), np.float32)
        albedo = np.full((height, width, 4), 7.0, np.float32)
        normal = np.full((height, width, 4), 7.0, np.float32)
        linear = np.zeros((height, width, 4), np.float32) if want_linear else None
        cnt = np.zeros(3, np.uint64)
        rows_a = None if rows is None else np.ascontiguousarray(rows, dtype=np.uint32)
        L.oracle_render(self.handle, _p(cam), width, height, _p(states), _p(bg), spp,
                        _p(rows_a), 0 if rows_a is None else rows_a.shape[0],
                        _p(color), _p(albedo), _p(normal), _p(linear), _p(cnt))
        return {"color": color, "albedo": albedo, "normal": normal, "linear": linear,
                "rays": int(cnt[0]), "node_visits": int(cnt[1]), "prim_tests": int(cnt[2])}

    def pixel_path(self, width, height, states, x, y, cap=16):
        """One sample of one pixel with its rays logged (a debugging aid): returns (rays (n, 10): origin, direction, any-hit flag, t or -1,
        primitive and instance as uint32 bits; linear radiance (3,)).  The pixel's RNG state in `states` advances as in a render."""
        L = lib()
        L.oracle_debug_pixel_path.restype = C.c_uint32
        L.oracle_debug_pixel_path.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        cam = self.camera12(); bg = np.ascontiguousarray(self.scene["background"], dtype=np.float32)
        log = np.zeros((cap, 10), np.float32); res = np.zeros(3, np.float32)
        n = L.oracle_debug_pixel_path(self.handle, _p(cam), width, height, _p(states), _p(bg), x, y, _p(log), cap, _p(res))
        return log[:n], res

    def trace(self, origins, directions, tmin=1e-6, tmax=1e16, any_hit=False):
        L = lib()
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        t, u, v = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        prim, inst = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        L.oracle_trace_rays(self.handle, _p(o), _p(d), n, tmin, tmax, int(any_hit), _p(t), _p(u), _p(v), _p(prim), _p(inst))
        return t, u, v, prim, inst

    def attach_bvh8(self, nodes, prims):
        """Walk the PRODUCT's packed BVH8 (uint8 arrays as hrt_tlas_download / hrt_host_build_bvh8 return them, for these very
        instances) instead of the oracle's own BVH2: same image (the closest hit is canonical), the product's bytes.  None detaches."""
        if nodes is None:
            lib().oracle_scene_attach_bvh8(self.handle, None, None)
            self._bvh8 = None
            return
        self._bvh8 = (np.ascontiguousarray(nodes, dtype=np.uint8), np.ascontiguousarray(prims, dtype=np.uint8))
        lib().oracle_scene_attach_bvh8(self.handle, self._bvh8[0].ctypes.data, self._bvh8[1].ctypes.data)

    def close(self):
        if self.handle:
            lib().oracle_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def construct_transform(shift, rotate_deg, scale):
    a, b, c = (np.ascontiguousarray(x, dtype=np.float32) for x in (shift, rotate_deg, scale))
    out = np.zeros(12, np.float32)
    lib().oracle_construct_transform(_p(a), _p(b), _p(c), _p(out))
    return out


def slerp(q1, q2, t):
    a, b = np.ascontiguousarray(q1, dtype=np.float32), np.ascontiguousarray(q2, dtype=np.float32)
    out = np.zeros(4, np.float32)
    lib().oracle_slerp(_p(a), _p(b), float(t), _p(out))
    return out


def quat_to_euler(q):
    a = np.ascontiguousarray(q, dtype=np.float32)
    out = np.zeros(3, np.float32)
    lib().oracle_quat_to_euler(_p(a), _p(out))
    return out


def pose_transforms(current, nxt, duration, frame, frame_count, offset=(0, 0, 0), scale=(1, 1, 1)):
    """(n, 12) float32 particle states of two time steps -> (n, 12) transforms of the frame."""
    cur = np.ascontiguousarray(current, dtype=np.float32)
    nx = np.ascontiguousarray(nxt, dtype=np.float32)
    off, sc = np.ascontiguousarray(offset, dtype=np.float32), np.ascontiguousarray(scale, dtype=np.float32)
    out = np.zeros((cur.shape[0], 12), np.float32)
    lib().oracle_pose_transforms(_p(cur), _p(nx), cur.shape[0], float(duration), int(frame), int(frame_count), _p(off), _p(sc), _p(out))
    return out


TRIG_SIN, TRIG_COS, TRIG_ACOS, TRIG_ASIN, TRIG_ATAN2 = range(5)


def trig(which, a, b=None, force_exact=False, libm=False):
    """The pose pipeline's float transcendentals as the oracle pins them (correctly rounded): f(a) or atan2(a, b) elementwise.
    force_exact decides every value in __float128; libm=True is the platform's float libm (tolerance cross-check)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = None if b is None else np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty(a.shape, np.float32)
    if libm:
        lib().oracle_trig_libm(which, _p(a), _p(b if b is not None else a), a.size, _p(out))
    else:
        lib().oracle_trig(which, _p(a), _p(b), a.size, int(force_exact), _p(out))
    return out


def trig_bits(which, first, stride, count, force_exact=False, out=None):
    """... over the floats whose bit patterns are first, first + stride, ... (one-argument functions)."""
    if out is None:
        out = np.empty(count, np.float32)
    lib().oracle_trig_bits(which, int(first) & 0xFFFFFFFF, int(stride), int(count), int(force_exact), _p(out))
    return out[:count]


def rng_init(width, height, salt):
    st = np.zeros((width * height, 12), dtype=np.uint32)
    lib().oracle_rng_init(_p(st), width, height, salt)
    return st


def bvh8_trace(nodes_ptr, prims_ptr, origins, directions, tmin=1e-6, tmax=1e16, any_hit=False,
               inst_inv=None, inst_identity=None, per_ray_nodes=None):
    """Walk a packed BVH8 blob of the product on the CPU.  Returns (t,u,v,prim,inst, node_visits, prim_tests)."""
    L = lib()
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
    n = o.shape[0]
    t, u, v = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    prim, inst = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    cnt = np.zeros(4, np.uint64)
    L.oracle_bvh8_trace(nodes_ptr, prims_ptr, _p(inst_inv), _p(inst_identity), _p(o), _p(d), n, tmin, tmax, int(any_hit),
                        _p(t), _p(u), _p(v), _p(prim), _p(inst), _p(cnt), _p(per_ray_nodes))
    bvh8_trace.last_empty_visits = int(cnt[2])       # node visits that found nothing to enter or test (tools/tree_quality.py)
    bvh8_trace.last_line_mates = int(cnt[3])         # node visits whose array neighbour (index ^ 1) the ray had visited before
    return t, u, v, prim, inst, int(cnt[0]), int(cnt[1])


def axis_parallel_rays(n, seed, lo=(0.05, 0.05, -0.95), hi=(0.95, 0.95, 0.95)):
    """Rays along the coordinate axes from points inside a box, their other two direction components every combination of +0.0 and -0.0
    (what a mirror reflection off an axis-aligned wall produces), and rays in the coordinate planes (one component +-0.0).  A reciprocal
    whose sign disagrees with `d < 0` for -0.0 swaps near and far in a slab test: such a ray then leaves a closed room."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = np.zeros((n, 3), np.float32)
    axis = rng.integers(0, 3, n); sign = rng.choice(np.float32([-1.0, 1.0]), n)
    zeros = rng.choice(np.float32([0.0, -0.0]), (n, 3))
    d[:] = zeros
    planar = rng.random(n) < 0.4                       # two components free, one a signed zero
    d[planar] = rng.normal(size=(int(planar.sum()), 3)).astype(np.float32)
    d[planar, axis[planar]] = zeros[planar, axis[planar]]
    d[~planar, axis[~planar]] = sign[~planar] * rng.choice(np.float32([1.0, 0.25, 7.0]), int((~planar).sum()))
    return o, d


def random_rays(n, seed, scale=1.6):
    """Deterministic test rays: origins in a cube around the scenes, directions pointing roughly inwards."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-scale, scale, size=(n, 3)).astype(np.float32)
    tgt = rng.uniform(-0.8, 0.8, size=(n, 3)).astype(np.float32)
    d = (tgt - o).astype(np.float32)
    # a few axis-parallel / zero-component directions: the slab test's edge cases
    k = max(1, n // 50)
    d[:k, 0] = 0.0
    d[k:2 * k, 1] = 0.0
    d[2 * k:3 * k, 1:] = 0.0
    return o, d
