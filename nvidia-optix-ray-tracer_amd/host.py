"""Host-side mirror of the reference's launch sequence, over the C ABI.

What the reference's ``RendererMesh::commitRendererData`` / ``startRender`` do around the
launch (src/Global/RendererMesh.cu:93-160 GAS + instances + IAS, :247-305 SBT, :323-324 RNG,
:328-333 camera, :395-419 per-frame upload + launch) is restated here step by step, each step
being one call into libhrt.so.  PyTorch is used only to own device memory and streams.
The C++ twin of this file is csrc/host/renderer_host.hpp.
"""
from __future__ import annotations

import ctypes as C

import numpy as np


def _f32(x):
    return np.float32(x)


def _normalize(v):
    """normalize(), include/Global/DeviceFunctions.cuh:397-404 with rsqrtf pinned as 1/sqrtf."""
    v = np.asarray(v, dtype=np.float32)
    len2 = _f32(_f32(_f32(v[0] * v[0]) + _f32(v[1] * v[1])) + _f32(v[2] * v[2]))
    if len2 <= _f32(_f32(1e-6) * _f32(1e-6)):
        return np.array([0, 0, 1], dtype=np.float32)
    inv = _f32(_f32(1.0) / np.sqrt(len2, dtype=np.float32))
    return np.array([v[0] * inv, v[1] * inv, v[2] * inv], dtype=np.float32)


def _cross(a, b):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    return np.array([_f32(_f32(a[1] * b[2]) - _f32(a[2] * b[1])),
                     _f32(_f32(a[2] * b[0]) - _f32(a[0] * b[2])),
                     _f32(_f32(a[0] * b[1]) - _f32(a[1] * b[0]))], dtype=np.float32)


def configure_camera(center, target, up, opengl=True):
    """SDL_GraphicsWindowConfigureCamera, src/GraphicsAPI/SDL_GraphicsWindow.cu:4-14.
    Returns (U, V, W): W = target - center is NOT normalised (its length sets the field of view)."""
    center = np.asarray(center, dtype=np.float32)
    target = np.asarray(target, dtype=np.float32)
    up_dir = _normalize(up)
    if not opengl:
        up_dir = (-up_dir).astype(np.float32)
    w = (target - center).astype(np.float32)
    u = _normalize(_cross(w, up_dir))
    v = _normalize(_cross(u, w))
    return u, v, w


def tile_for_rank(height, rank, world_size, stripe_rows=8):
    """Row stripes of the multi-GPU split: stripe k (stripe_rows rows) belongs to rank k % world_size.
    Interleaving balances the load (the middle of the frame is the expensive part)."""
    from . import Tile
    if world_size <= 1:
        return Tile(0, height, 1, 1, 0)
    return Tile(0, height, stripe_rows, world_size, rank)


def reduce_tiles(frame, dst=0):
    """The one exchange step of the multi-GPU split: sum the per-rank frames (each zero outside its
    own row stripes, so x + 0 is exact) into rank ``dst`` over torch.distributed (RCCL on GPUs)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if frame.is_cuda and dist.get_backend() != "nccl":
            # rehearsal on a box with fewer cards than ranks (gloo has no CUDA reduce): stage through the host
            staged = frame.cpu()
            dist.reduce(staged, dst=dst, op=dist.ReduceOp.SUM)
            if dist.get_rank() == dst:
                frame.copy_(staged)
        else:
            dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM)
    return frame


class Renderer:
    """One context + one scene: build, then ``render()`` as often as needed."""

    def __init__(self, device=0, flags=0):
        import torch
        from . import load_library, HrtError
        self._torch = torch
        self._err = HrtError
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise HrtError("no GPU visible: the renderer has no CPU path")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        ctx = C.c_void_p()
        rc = self.lib.hrt_ctx_create(device, flags, C.byref(ctx))
        if rc != 0:
            raise HrtError(f"hrt_ctx_create: {self.lib.hrt_last_error(None).decode()}")
        self.ctx = ctx
        self._keep = []             # device tensors the SBT records point into
        self._blas = []
        self._n_inst = None
        self.tlas = None
        self.states = None
        self.width = self.height = 0
        self.color = self.albedo = self.normal = self.linear = None
        self.cam = None

    # ---- helpers ----
    def _check(self, rc, what):
        if rc != 0:
            raise self._err(f"{what}: {self.lib.hrt_last_error(self.ctx).decode()} (status {rc})")

    def _dev(self, arr):
        t = self._torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)
        return t

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    # ---- scene ----
    def load_scene(self, scene):
        """GAS per instance -> instances (sbtOffset = i) -> IAS -> SBT records -> miss record."""
        from . import Instance, SbtRecord, MissParams, Float3
        from . import PROGRAM_SPHERE_ROUGH, PROGRAM_SPHERE_METAL, PROGRAM_TRIANGLE_ROUGH, PROGRAM_TRIANGLE_METAL
        lib, st = self.lib, self._stream()
        insts = scene["instances"]
        n = len(insts)
        h_inst = (Instance * max(n, 1))()
        h_rec = (SbtRecord * max(n, 1))()
        shared = {}                                        # "shape" id -> (GAS handle, normals): RendererTime.cu:116-130
        for i, it in enumerate(insts):
            handle = C.c_uint64()
            if it["geometry"] == "triangles":
                key = it.get("shape")
                if key is not None and key in shared:
                    handle.value, normals = shared[key]
                else:
                    verts = self._dev(it["vertices"].reshape(-1, 3))
                    normals = self._dev(it["normals"].reshape(-1, 3))
                    self._check(lib.hrt_blas_build_triangles(self.ctx, verts.data_ptr(), verts.shape[0], st, C.byref(handle)),
                                "hrt_blas_build_triangles")
                    del verts                              # the reference frees vertices after the build too
                    self._keep.append(normals)
                    if key is not None:
                        shared[key] = (handle.value, normals)
                h_rec[i].data.ptr0 = normals.data_ptr()
                prog = PROGRAM_TRIANGLE_ROUGH if it["material"] == "rough" else PROGRAM_TRIANGLE_METAL
            else:
                centers = self._dev(it["centers"])
                radii = self._dev(it["radii"])
                self._check(lib.hrt_blas_build_spheres(self.ctx, centers.data_ptr(), radii.data_ptr(), radii.shape[0], st,
                                                       C.byref(handle)), "hrt_blas_build_spheres")
                self._keep += [centers, radii]
                h_rec[i].data.ptr0 = centers.data_ptr()
                h_rec[i].data.ptr1 = radii.data_ptr()
                prog = PROGRAM_SPHERE_ROUGH if it["material"] == "rough" else PROGRAM_SPHERE_METAL
            self._blas.append(handle.value)
            self._check(lib.hrt_sbt_record_pack_header(prog, C.byref(h_rec[i], 0)), "hrt_sbt_record_pack_header")
            a = it["albedo"]
            h_rec[i].data.albedo = Float3(float(a[0]), float(a[1]), float(a[2]))
            h_rec[i].data.fuzz = float(it["fuzz"]) if it["material"] == "metal" else 0.0
            for k in range(12):
                h_inst[i].transform[k] = float(it["transform"][k])
            h_inst[i].instanceId = 0
            h_inst[i].sbtOffset = i
            h_inst[i].visibilityMask = 1
            h_inst[i].flags = 0
            h_inst[i].traversableHandle = handle.value
        d_inst = self._dev(np.frombuffer(bytes(h_inst), dtype=np.uint8).copy())
        tl = C.c_uint64()
        self._check(lib.hrt_tlas_build(self.ctx, d_inst.data_ptr(), n, st, C.byref(tl)), "hrt_tlas_build")
        self.tlas = tl.value
        self._d_inst = d_inst
        self._h_inst = h_inst
        self._n_inst = n
        self._check(lib.hrt_materials_set(self.ctx, h_rec, n), "hrt_materials_set")
        bg = scene.get("background", np.array([0.7, 0.8, 0.9], dtype=np.float32))
        miss = MissParams(Float3(float(bg[0]), float(bg[1]), float(bg[2])))
        self._check(lib.hrt_miss_set(self.ctx, C.byref(miss)), "hrt_miss_set")
        cam = scene["camera"]
        self.set_camera(cam["center"], cam["target"], cam["up"], cam.get("opengl", True))
        return self

    def update_instances(self, transforms):
        """Per-frame instance update + updateIAS (src/Global/RendererMesh.cu:379-401)."""
        for i, m in enumerate(transforms):
            for k in range(12):
                self._h_inst[i].transform[k] = float(m[k])
        self._d_inst.copy_(self._torch.from_numpy(np.frombuffer(bytes(self._h_inst), dtype=np.uint8).copy()))
        self._check(self.lib.hrt_tlas_update(self.ctx, self.tlas, self._d_inst.data_ptr(), len(transforms), self._stream()),
                    "hrt_tlas_update")

    def pose_instances(self, current, nxt, duration, frame, frame_count, first_instance=0,
                       offset=(0.0, 0.0, 0.0), scale=(1.0, 1.0, 1.0), update=True, mesh_mode=False):
        """Time mode's per-frame pose step on the device (src/Global/RendererTime.cu:436-480): ``current`` / ``nxt`` are
        (n, 12) float32 particle states (quat.xyzw, position, velocity, 2 pad) of this and the next time step; writes the
        transforms of instances [first_instance, first_instance + n) in device memory, then updateIAS."""
        from . import PoseParams
        cur = current if hasattr(current, "data_ptr") else self._dev(np.ascontiguousarray(current, dtype=np.float32))
        nx = nxt if hasattr(nxt, "data_ptr") else self._dev(np.ascontiguousarray(nxt, dtype=np.float32))
        n = cur.shape[0]
        pp = PoseParams(float(duration), int(frame), int(frame_count),
                        (C.c_float * 3)(*[float(x) for x in offset]), (C.c_float * 3)(*[float(x) for x in scale]), 1 if mesh_mode else 0)
        self._check(self.lib.hrt_pose_instances(self.ctx, self._d_inst.data_ptr(), first_instance, n, cur.data_ptr(), nx.data_ptr(),
                                                C.byref(pp), self._stream()), "hrt_pose_instances")
        if update:
            self._check(self.lib.hrt_tlas_update(self.ctx, self.tlas, self._d_inst.data_ptr(), self._n_inst,
                                                 self._stream()), "hrt_tlas_update")

    def instance_transforms(self):
        """The transforms currently in the device instance array, (n, 12) float32."""
        raw = self._d_inst.cpu().numpy().view(np.uint8).reshape(-1, 80)
        return raw[: self._n_inst, :48].copy().view(np.float32).reshape(-1, 12)

    def set_camera(self, center, target, up, opengl=True):
        u, v, w = configure_camera(center, target, up, opengl)
        self.cam = (np.asarray(center, dtype=np.float32), u, v, w)

    def set_frame(self, width, height, seed_salt, aov=True, linear=False):
        """RNG states + output buffers (initDeviceRandomGenerators, denoiser input buffers)."""
        torch = self._torch
        if self.states is not None:
            self._check(self.lib.hrt_rng_free(self.ctx, self.states, self._stream()), "hrt_rng_free")
            self.states = None
        st = C.c_void_p()
        self._check(self.lib.hrt_rng_init(self.ctx, width, height, seed_salt, self._stream(), C.byref(st)), "hrt_rng_init")
        self.states = st
        self.width, self.height = width, height
        self.color = torch.zeros((height, width, 4), dtype=torch.float32, device=self.device)
        self.albedo = torch.full((height, width, 4), 7.0, dtype=torch.float32, device=self.device) if aov else None
        self.normal = torch.full((height, width, 4), 7.0, dtype=torch.float32, device=self.device) if aov else None
        self.linear = torch.zeros((height, width, 4), dtype=torch.float32, device=self.device) if linear else None
        self._check(self.lib.hrt_debug_set_linear_output(self.ctx, self.linear.data_ptr() if linear else None),
                    "hrt_debug_set_linear_output")
        return self

    def rng_states_numpy(self):
        """Copy of the device RNG states as a (H*W, 12) uint32 array (48 bytes per state)."""
        self._torch.cuda.synchronize(self.device)
        n = self.width * self.height
        out = np.empty((n, 12), dtype=np.uint32)
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rc = hip.hipMemcpy(out.ctypes.data, self.states, n * 48, 2)      # hipMemcpyDeviceToHost
        if rc != 0:
            raise self._err(f"hipMemcpy of the RNG states failed ({rc})")
        return out

    # ---- the launch ----
    def render(self, spp=1, tile=None, sync=True):
        from . import GlobalParams, RayGenParams, Float3
        center, u, v, w = self.cam
        params = GlobalParams(self.tlas, self.states)
        rg = RayGenParams()
        rg.width, rg.height = self.width, self.height
        rg.colorBuffer = self.color.data_ptr()
        rg.albedoBuffer = self.albedo.data_ptr() if self.albedo is not None else None
        rg.normalBuffer = self.normal.data_ptr() if self.normal is not None else None
        rg.cameraCenter = Float3(*[float(x) for x in center])
        rg.cameraU = Float3(*[float(x) for x in u])
        rg.cameraV = Float3(*[float(x) for x in v])
        rg.cameraW = Float3(*[float(x) for x in w])
        st = self._stream()
        self._check(self.lib.hrt_render_launch(self.ctx, C.byref(params), C.byref(rg), spp,
                                               C.byref(tile) if tile is not None else None, st), "hrt_render_launch")
        if sync:
            self._check(self.lib.hrt_sync(self.ctx, st), "hrt_sync")

    def to_rgba8(self):
        torch = self._torch
        out = torch.empty((self.height, self.width, 4), dtype=torch.uint8, device=self.device)
        self._check(self.lib.hrt_to_rgba8(self.ctx, self.color.data_ptr(), out.data_ptr(), self.width, self.height,
                                          self._stream()), "hrt_to_rgba8")
        self._torch.cuda.synchronize(self.device)
        return out

    def trace_rays(self, origins, directions, tmin=1e-6, tmax=1e16, any_hit=False):
        torch = self._torch
        o = self._dev(np.asarray(origins, dtype=np.float32).reshape(-1, 3))
        d = self._dev(np.asarray(directions, dtype=np.float32).reshape(-1, 3))
        n = o.shape[0]
        t = torch.empty(n, dtype=torch.float32, device=self.device)
        u = torch.empty_like(t)
        v = torch.empty_like(t)
        prim = torch.empty(n, dtype=torch.int32, device=self.device)
        inst = torch.empty(n, dtype=torch.int32, device=self.device)
        self._check(self.lib.hrt_trace_rays(self.ctx, self.tlas, o.data_ptr(), d.data_ptr(), n, tmin, tmax, int(any_hit),
                                            t.data_ptr(), u.data_ptr(), v.data_ptr(), prim.data_ptr(), inst.data_ptr(),
                                            self._stream()), "hrt_trace_rays")
        return (t.cpu().numpy(), u.cpu().numpy(), v.cpu().numpy(),
                prim.cpu().numpy().view(np.uint32), inst.cpu().numpy().view(np.uint32))

    def stats(self, reset=False):
        from . import Stats
        s = Stats()
        self._check(self.lib.hrt_stats_get(self.ctx, C.byref(s)), "hrt_stats_get")
        if reset:
            self._check(self.lib.hrt_stats_reset(self.ctx), "hrt_stats_reset")
        return s

    def set_flags(self, flags):
        self._check(self.lib.hrt_ctx_set_flags(self.ctx, flags), "hrt_ctx_set_flags")

    def reset_stats(self):
        self._check(self.lib.hrt_stats_reset(self.ctx), "hrt_stats_reset")

    def close(self):
        if getattr(self, "ctx", None):
            if self.states is not None:
                self.lib.hrt_rng_free(self.ctx, self.states, None)
                self.states = None
            self.lib.hrt_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
