"""MI355X-native wavefront path tracer: thin Python binding over the C ABI (include/hrt.h).

The product is ``lib/libhrt.so`` (hand-written HIP kernels for gfx950 + a C++ host layer).
This module only loads it with ctypes and mirrors the structs of ``include/hrt_params.h``;
there is no Python or CPU compute path -- if the library is missing, or no gfx950 device is
present, the calls raise.

The directory name contains hyphens, so import it with::

    import importlib; hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["HRT_LIB"]) if os.environ.get("HRT_LIB") else _PKG_DIR / "lib" / "libhrt.so"   # HRT_LIB: an instrumented build (tools/lane_stats.py)

# ---- struct mirrors (include/hrt_params.h) ---------------------------------------------


class Float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class GlobalParams(C.Structure):
    _fields_ = [("handle", C.c_uint64), ("stateArray", C.c_void_p)]


class RayGenParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32),
                ("colorBuffer", C.c_void_p), ("albedoBuffer", C.c_void_p), ("normalBuffer", C.c_void_p),
                ("cameraCenter", Float3), ("cameraU", Float3), ("cameraV", Float3), ("cameraW", Float3)]


class MissParams(C.Structure):
    _fields_ = [("backgroundColor", Float3)]


class HitGroupParams(C.Structure):
    # union{sphere{centers, radii} | triangles{vertexNormals}} + union{rough{albedo} | metal{albedo, fuzz}}
    _fields_ = [("ptr0", C.c_void_p), ("ptr1", C.c_void_p), ("albedo", Float3), ("fuzz", C.c_float)]


class SbtRecord(C.Structure):
    _fields_ = [("header", C.c_ubyte * 32), ("data", HitGroupParams)]


class Instance(C.Structure):
    _fields_ = [("transform", C.c_float * 12), ("instanceId", C.c_uint32), ("sbtOffset", C.c_uint32),
                ("visibilityMask", C.c_uint32), ("flags", C.c_uint32), ("traversableHandle", C.c_uint64),
                ("pad", C.c_uint32 * 2)]


class PoseParams(C.Structure):
    """HrtPoseParams: frame-loop scalars of the Time-mode pose update (src/Global/RendererTime.cu:425-470)."""
    _fields_ = [("duration", C.c_float), ("frame", C.c_uint32), ("frame_count", C.c_uint32),
                ("particle_offset", C.c_float * 3), ("particle_scale", C.c_float * 3), ("mesh_mode", C.c_uint32)]


PARTICLE_STATE_FLOATS = 12      # HrtParticleState as floats: quat.xyzw, position.xyz, velocity.xyz, 2 pad


class Tile(C.Structure):
    _fields_ = [("y_begin", C.c_uint32), ("y_end", C.c_uint32), ("stripe_rows", C.c_uint32),
                ("stripe_period", C.c_uint32), ("stripe_phase", C.c_uint32)]


K_GENERATE, K_TRAVERSE, K_TRAVERSE_ANY, K_BIN, K_SHADE, K_ACCUMULATE, K_FINALIZE, K_PATHS, K_REFIT, K_COUNT = range(10)
KERNEL_NAMES = ["generate", "traverse", "traverse_any", "bin", "shade", "accumulate", "finalize", "paths", "refit"]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_any", C.c_uint64),
                ("paths", C.c_uint64), ("node_visits", C.c_uint64), ("prim_tests", C.c_uint64),
                ("node_visits_closest", C.c_uint64), ("prim_tests_closest", C.c_uint64),
                ("kernel_ms", C.c_double * K_COUNT), ("kernel_launches", C.c_uint64 * K_COUNT),
                ("bvh_nodes", C.c_uint64), ("bvh_triangles", C.c_uint64), ("bvh_spheres", C.c_uint64),
                ("bvh_bytes", C.c_uint64), ("debug", C.c_uint64 * 4),
                ("tlas_refits", C.c_uint64), ("tlas_rebuilds", C.c_uint64), ("tlas_refit_ratio", C.c_double),
                ("bvh_depth", C.c_uint64), ("fused_fallback_launches", C.c_uint64), ("graph_replays", C.c_uint64), ("bvh_alloc_bytes", C.c_uint64)]


class BvhBlob(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("n_nodes", C.c_uint64), ("triangles", C.c_void_p),
                ("n_triangles", C.c_uint64), ("bounds", C.c_float * 6)]


assert C.sizeof(GlobalParams) == 16 and C.sizeof(RayGenParams) == 80 and C.sizeof(MissParams) == 12
assert C.sizeof(HitGroupParams) == 32 and C.sizeof(SbtRecord) == 64 and C.sizeof(Instance) == 80

PROGRAM_SPHERE_ROUGH, PROGRAM_SPHERE_METAL, PROGRAM_TRIANGLE_ROUGH, PROGRAM_TRIANGLE_METAL = range(4)
CTX_TIMING, CTX_COUNT, CTX_FAST_TRACE, CTX_ASYNC_UPDATE, CTX_TWO_LEVEL, CTX_REUSE_PRIMARY = 1, 2, 4, 8, 16, 32

# every symbol include/hrt.h declares (checked by the CPU test-suite)
EXPORTS = [
    "hrt_ctx_create", "hrt_ctx_destroy", "hrt_ctx_set_flags", "hrt_last_error", "hrt_version",
    "hrt_blas_build_triangles", "hrt_blas_build_spheres", "hrt_blas_destroy",
    "hrt_tlas_build", "hrt_tlas_update", "hrt_tlas_destroy", "hrt_pose_instances",
    "hrt_sbt_record_pack_header", "hrt_materials_set", "hrt_miss_set",
    "hrt_rng_init", "hrt_rng_free", "hrt_render_launch", "hrt_sync", "hrt_to_rgba8", "hrt_color_to_float4",
    "hrt_stats_reset", "hrt_stats_get", "hrt_trace_rays", "hrt_debug_set_linear_output",
    "hrt_host_build_bvh8", "hrt_tlas_download", "hrt_host_free", "hrt_debug_trig",
]


class HrtError(RuntimeError):
    pass


_lib = None


def load_library():
    """dlopen libhrt.so.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise HrtError(f"{LIB_PATH} is missing: run `make lib` (or __graft_entry__.build()); there is no CPU fallback")
    # Load order matters in a Python process: PyTorch ships its own copy of the HIP runtime.  If libhrt.so came first it
    # would pull in /opt/rocm's libamdhip64, and the process would end up with two runtimes (observed: hipGetDeviceCount
    # then reports no device).  Importing torch first makes libhrt.so bind to the runtime torch has already loaded --
    # the same one the tensors that carry the device memory live in.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
    lib.hrt_last_error.restype = C.c_char_p
    lib.hrt_last_error.argtypes = [C.c_void_p]
    lib.hrt_version.restype = C.c_char_p
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("hrt_last_error", "hrt_version", "hrt_host_free"):
            fn.restype = C.c_int
    lib.hrt_ctx_create.argtypes = [C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
    lib.hrt_ctx_destroy.argtypes = [C.c_void_p]
    lib.hrt_ctx_set_flags.argtypes = [C.c_void_p, C.c_uint32]
    lib.hrt_blas_build_triangles.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.hrt_blas_build_spheres.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.hrt_blas_destroy.argtypes = [C.c_void_p, C.c_uint64]
    lib.hrt_tlas_build.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.hrt_tlas_update.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.hrt_tlas_destroy.argtypes = [C.c_void_p, C.c_uint64]
    lib.hrt_pose_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hrt_sbt_record_pack_header.argtypes = [C.c_int, C.c_void_p]
    lib.hrt_materials_set.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.hrt_miss_set.argtypes = [C.c_void_p, C.POINTER(MissParams)]
    lib.hrt_rng_init.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.hrt_rng_free.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hrt_render_launch.argtypes = [C.c_void_p, C.POINTER(GlobalParams), C.POINTER(RayGenParams), C.c_uint32,
                                      C.POINTER(Tile), C.c_void_p]
    lib.hrt_sync.argtypes = [C.c_void_p, C.c_void_p]
    lib.hrt_to_rgba8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.hrt_color_to_float4.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.hrt_stats_reset.argtypes = [C.c_void_p]
    lib.hrt_stats_get.argtypes = [C.c_void_p, C.POINTER(Stats)]
    lib.hrt_trace_rays.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_float,
                                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hrt_debug_set_linear_output.argtypes = [C.c_void_p, C.c_void_p]
    lib.hrt_debug_trig.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
    lib.hrt_host_build_bvh8.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(BvhBlob)]
    lib.hrt_tlas_download.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(BvhBlob)]
    lib.hrt_host_free.argtypes = [C.POINTER(BvhBlob)]
    lib.hrt_host_free.restype = None
    _lib = lib
    return lib


from .host import Renderer, configure_camera, tile_for_rank, reduce_tiles  # noqa: E402  (host-side mirror of the launch sequence)
from . import scenes  # noqa: E402

__all__ = ["load_library", "Renderer", "configure_camera", "tile_for_rank", "reduce_tiles", "scenes", "HrtError",
           "GlobalParams", "RayGenParams", "MissParams", "HitGroupParams", "SbtRecord", "Instance", "Tile", "Stats"]
