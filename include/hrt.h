/*
 * hrt.h -- C ABI of the MI355X wavefront path tracer (libhrt.so).
 *
 * Drop-in boundary for ONE path of the reference: the per-frame ray-tracing
 * launch and the acceleration-structure calls that feed it.  Every entry point
 * names the reference call site it replaces (paths relative to the reference
 * tree).  Plain pointers and sizes only; "stream" is a hipStream_t passed as
 * void* (NULL = the null stream, which is what the reference uses).
 *
 * Conventions
 *   - every function returns HRT_OK (0) or a negative HrtStatus; the message of
 *     the last failure on a context is hrt_last_error(ctx).  (The reference
 *     logs and exit()s through its check macros, include/Global/HostFunctions.cuh:147-166;
 *     a library cannot, so the caller re-wraps.)
 *   - d_* parameters are DEVICE pointers owned by the caller, h_* are host pointers.
 *   - acceleration-structure memory is owned by the handle (reference:
 *     cleanupAccelerationStructure, src/Global/RendererImpl.cu:244-266).
 *   - *_build calls are thread-safe per context (reference builds from several
 *     loader threads, src/Global/RendererMesh.cu:93-100); hrt_render_launch is
 *     called from one thread at a time per context.
 *   - there is NO CPU fallback: without a gfx950 device every compute entry
 *     point fails with HRT_ERR_NO_DEVICE.
 */
#ifndef HRT_H
#define HRT_H

#include <stdint.h>
#include "hrt_params.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum HrtStatus {
    HRT_OK               =  0,
    HRT_ERR_INVALID      = -1,   /* bad argument                                 */
    HRT_ERR_NO_DEVICE    = -2,   /* no HIP device / wrong architecture           */
    HRT_ERR_HIP          = -3,   /* a HIP runtime call failed                    */
    HRT_ERR_OOM          = -4,
    HRT_ERR_STATE        = -5    /* call order (e.g. launch before materials_set)*/
} HrtStatus;

typedef struct HrtContext HrtContext;

/* context flags */
#define HRT_CTX_TIMING   0x1u    /* bracket every kernel with HIP events (hrt_get_stats) */
#define HRT_CTX_COUNT    0x2u    /* traverse kernel counts node visits / primitive tests */
#define HRT_CTX_ASYNC_UPDATE 0x8u /* hrt_tlas_update never reads the instance array back: the per-instance tables (object->world, its inverse, the
                                    scene scale) are derived on the device and the refit is only enqueued -- the whole Time-mode frame (pose kernel,
                                    update, launch) then runs without a host synchronisation.  The caller promises what updateIAS requires anyway
                                    (OPTIX_BUILD_OPERATION_UPDATE, RendererImpl.cu:210-242): same BLAS handles, visibility bits and sbtOffsets as
                                    at the build.  A broken promise and a tree that has degraded past the rebuild ratio are both detected on the
                                    device and acted on at the NEXT update (one frame late), which then takes the synchronous path: a changed handle /
                                    visibility bit or a degraded tree rebuilds, a changed sbtOffset refreshes the material tables.  The FIRST update
                                    after a build is synchronous in either mode (one small read-back per build): the reference builds every file's IAS
                                    with identity transforms and poses it afterwards, so that update is the one that has to rebuild. */
#define HRT_CTX_TWO_LEVEL 0x10u  /* hrt_tlas_build keeps the reference's structure, an IAS over shared GASes (RendererImpl.cu:174-206; GAS by shapeID,
                                    RendererTime.cu:116-130): a top level over the instances whose leaves are TRANSFORM NODES, one object-space tree per
                                    unique BLAS behind it -- memory, build and update cost grow with instances + unique primitives instead of
                                    instances x primitives; a ray entering an instance is transformed into its object space (csrc/bvh8.h, fused.hip).
                                    Without the flag such a tree is built when the flattened one would leave the caches (more than 4 M flattened
                                    primitives, at least 4 per unique one; HRT_TWO_LEVEL=1 / -1: always / never).  Hits of such a tree are pinned to the
                                    oracle's INSTANCED mode (object-space triangle test), those of a flattened tree to its FLATTENED mode: the two
                                    differ in rounding, not in geometry.  Trees too deep for the path kernel's stack, and contexts that count
                                    (HRT_CTX_COUNT) or run another execution mode (HRT_FUSED != 1), flatten. */
#define HRT_CTX_REUSE_PRIMARY 0x20u /* hrt_render_launch with spp > 1: the reference's raygen has no pixel jitter (shader/Shader.cu:249-261), so a pixel's
                                    primary ray hits the same thing in every sample.  With this flag (or HRT_REUSE_PRIMARY=1) the path kernel traverses it for
                                    the first sample a launch takes of the pixel and shades the later samples from that hit record: the same image bit
                                    for bit, a third fewer rays on the reference's scenes.  HrtStats.rays counts TRAVERSED rays only, so Mrays/s figures
                                    with and without the flag are not comparable -- compare times.  Off by default (all published figures are without);
                                    ignored by the other execution modes (HRT_FUSED != 1, HRT_CTX_COUNT). */
#define HRT_CTX_FAST_TRACE 0x4u  /* hrt_tlas_build prefers trace speed to build speed: the reference's OPTIX_BUILD_FLAG_PREFER_FAST_TRACE
                                    (its GAS builds, RendererImpl.cu:94,118,144).  The tree is then built WITH SPATIAL SPLITS: on the device
                                    (csrc/build_split.hip: top-down SAH splits of references level by level, PLOC within the cells that
                                    remain; 1 M triangles: ~15 ms against ~9 ms for the default build -- the same phase with object splits only,
                                    which keeps the tree refittable --, 9.5 % fewer node visits per ray,
                                    1.46 records per triangle, ~1.2 KB of working memory per triangle), or -- environment
                                    HRT_FAST_TRACE_BUILD=host -- by the host's binned-SAH builder from a host copy of the geometry (the same
                                    rules, ~1.3 s).  Such a tree is for static scenes: the first hrt_tlas_update replaces it by a
                                    default-built one (a refit cannot keep the split references' boxes), and rebuilds inside hrt_tlas_update
                                    are always default builds (the reference's IAS flags: ALLOW_UPDATE | PREFER_FAST_BUILD, RendererImpl.cu:180). */

/* replaces createContext / destroyContext, src/Global/RendererImpl.cu:6-27 */
int  hrt_ctx_create(int device_id, uint32_t flags, HrtContext **out_ctx);
int  hrt_ctx_destroy(HrtContext *ctx);
int  hrt_ctx_set_flags(HrtContext *ctx, uint32_t flags);   /* switch HRT_CTX_TIMING / HRT_CTX_COUNT / HRT_CTX_REUSE_PRIMARY at run time */
const char *hrt_last_error(const HrtContext *ctx);        /* ctx may be NULL: creation errors */
const char *hrt_version(void);

/* replaces buildGASForTriangles / buildGASForParticle, src/Global/RendererImpl.cu:89-111,139-172.
 * d_vertices: float3, stride 12, every 3 consecutive vertices are one triangle (no index
 * buffer, as the reference).  n_vertices must be a multiple of 3.  The caller may free
 * d_vertices after the call returns (Mesh mode does, src/Global/RendererMesh.cu:116). */
int  hrt_blas_build_triangles(HrtContext *ctx, const HrtFloat3 *d_vertices, uint32_t n_vertices,
                              void *stream, HrtTraversable *out_blas);
/* replaces buildGASForSpheres, src/Global/RendererImpl.cu:113-138 */
int  hrt_blas_build_spheres(HrtContext *ctx, const HrtFloat3 *d_centers, const float *d_radii,
                            uint32_t n_spheres, void *stream, HrtTraversable *out_blas);
int  hrt_blas_destroy(HrtContext *ctx, HrtTraversable blas);

/* replaces buildIAS / updateIAS, src/Global/RendererImpl.cu:174-242.  d_instances lives in
 * device memory (reference: cudaMemcpy H2D then build, src/Global/RendererMesh.cu:151-160).
 * hrt_tlas_build flattens the instances into one world-space BVH8, built on the device (top-down SAH object splits down to
 * cells of a few references, csrc/build_split.hip -- with spatial splits under HRT_CTX_FAST_TRACE --, then Morton sort, PLOC
 * within the cells and the optimal 8-wide collapse, csrc/build.hip; scenes of at most 4096 primitives: PLOC alone).  hrt_tlas_update takes
 * the same number of instances: when only transforms (and sbtOffsets) changed, the tree is refitted on the
 * device, asynchronously on `stream` after one small read-back of the instance array; a changed BLAS handle
 * or visibility mask, or a refitted tree whose boxes have grown too far, rebuilds it (as a tree over
 * instances, which costs milliseconds).  A BLAS may be destroyed while a TLAS still instances it. */
int  hrt_tlas_build(HrtContext *ctx, const HrtInstance *d_instances, uint32_t n_instances,
                    void *stream, HrtTraversable *out_tlas);
int  hrt_tlas_update(HrtContext *ctx, HrtTraversable tlas, const HrtInstance *d_instances,
                     uint32_t n_instances, void *stream);
int  hrt_tlas_destroy(HrtContext *ctx, HrtTraversable tlas);

/* replaces the per-frame host loop of Time mode, src/Global/RendererTime.cu:436-472: for particle i,
 * slerp(cur.quat, next.quat, factor) (:296-340) -> quatToEuler (:343-370) -> constructTransformMatrix(offset + shift,
 * rotate, scale) (include/Global/DeviceFunctions.cuh:133-148), written to d_instances[first_instance + i].transform
 * on the device -- no host loop, no H2D copy of the instance array.  Follow with hrt_tlas_update. */
int  hrt_pose_instances(HrtContext *ctx, HrtInstance *d_instances, uint32_t first_instance, uint32_t n_particles,
                        const HrtParticleState *d_current, const HrtParticleState *d_next,
                        const HrtPoseParams *h_params, void *stream);

/* replaces optixSbtRecordPackHeader as used at src/Global/RendererImpl.cu:514-560 */
int  hrt_sbt_record_pack_header(HrtProgram program, void *record_header);
/* replaces the hit-group SBT upload, src/Global/RendererMesh.cu:283-305: record i belongs
 * to the instance whose sbtOffset is i.  h_records is host memory; it is copied. */
int  hrt_materials_set(HrtContext *ctx, const HrtSbtRecord *h_records, uint32_t n_records);
/* replaces the miss record of createRaygenMissSBTRecord, src/Global/RendererImpl.cu:472-492 */
int  hrt_miss_set(HrtContext *ctx, const HrtMissParams *h_miss);

/* replaces RandomGenerator::initDeviceRandomGenerators / freeDeviceRandomGenerators,
 * src/Global/HostFunctions.cu:122-140: allocates W*H states and initialises state i with
 * curand_init(seed = i ^ seed_salt, subsequence = i, offset = 0).  seed_salt pins the
 * reference's clock64() term (quirk Q8); the kernel is bounds-checked and indexes by the
 * frame width (fixes quirk Q9). */
int  hrt_rng_init(HrtContext *ctx, uint32_t width, uint32_t height, uint64_t seed_salt,
                  void *stream, HrtRngState **out_d_states);
int  hrt_rng_free(HrtContext *ctx, HrtRngState *d_states, void *stream);

/* replaces optixLaunch(pipeline, stream, dev_params, sizeof(GlobalParams), &sbt, W, H, 1)
 * + cudaDeviceSynchronize, src/Global/RendererMesh.cu:416-419 / RendererTime.cu:497-500.
 * h_params / h_raygen are HOST copies of the two blocks the reference uploads each frame
 * (RendererMesh.cu:403-413).  spp successive samples are taken on the persistent per-pixel
 * RNG streams; the colour written is colorToFloat4(mean of the samples), which for spp = 1
 * is exactly the reference's frame.  tile == NULL renders the whole frame; rows outside the
 * tile are left untouched.  The call returns after the work has been ENQUEUED on stream;
 * hrt_sync (or any stream sync) completes it.  (Renders of >= 16 samples on small tiles synchronise
 * the stream once in the middle: a probe launch orders the pixel slices by cost for the rest.) */
int  hrt_render_launch(HrtContext *ctx, const HrtGlobalParams *h_params,
                       const HrtRayGenParams *h_raygen, uint32_t spp,
                       const HrtTile *tile, void *stream);
int  hrt_sync(HrtContext *ctx, void *stream);

/* replaces convertFloat4ToUchar4Kernel, src/Global/RendererImpl.cu:672-678 (second sRGB
 * encode: quirk Q7) */
int  hrt_to_rgba8(HrtContext *ctx, const HrtFloat4 *d_src, HrtUchar4 *d_dst,
                  uint32_t width, uint32_t height, void *stream);

/* colorToFloat4 (include/Global/DeviceFunctions.cuh:188-209) over n colours: the conversion raygen applies to its
 * result (shader/Shader.cu:270), as a call of its own.  Both conversions pin the shader's powf(c, 1/2.4f) as the
 * correctly rounded float of c^y, y = (double)(1.0f/2.4f) (csrc/srgb_pow.h), so the float image and the byte image are
 * bit-exact against the oracle. */
int  hrt_color_to_float4(HrtContext *ctx, const HrtFloat4 *d_src, HrtFloat4 *d_dst, uint32_t n, void *stream);

/* ---- measurement (no reference counterpart: the reference has no timers) ------------- */
enum { HRT_K_GENERATE = 0, HRT_K_TRAVERSE, HRT_K_TRAVERSE_ANY, HRT_K_BIN, HRT_K_SHADE,
       HRT_K_ACCUMULATE, HRT_K_FINALIZE, HRT_K_PATHS /* fused path mode */, HRT_K_REFIT /* hrt_tlas_update */, HRT_K_COUNT };

typedef struct HrtStats {
    uint64_t rays;                         /* trace calls since the last reset (1..5 per pixel-sample) */
    uint64_t rays_closest, rays_any;       /* split by traverse kernel                                 */
    uint64_t paths;                        /* pixel-samples                                            */
    uint64_t node_visits, prim_tests;      /* HRT_CTX_COUNT only, closest+any                          */
    uint64_t node_visits_closest, prim_tests_closest;   /* the closest-hit kernel's share            */
    double   kernel_ms[HRT_K_COUNT];       /* HRT_CTX_TIMING only: summed HIP-event time per kernel    */
    uint64_t kernel_launches[HRT_K_COUNT];
    uint64_t bvh_nodes, bvh_triangles, bvh_spheres;   /* of the TLAS last launched               */
    uint64_t bvh_bytes;
    uint64_t debug[4];                     /* HRT_CTX_COUNT, closest-hit kernel: wave iterations, wave leaf passes,
                                              sum of alive lanes over iterations, reserved                     */
    uint64_t tlas_refits, tlas_rebuilds;   /* hrt_tlas_update calls served by the device refit / builds + rebuilds    */
    double   tlas_refit_ratio;             /* quality sum of the last refitted tree checked / that of the built tree  */
    uint64_t bvh_depth;                    /* levels below the root of the TLAS last launched (trees deeper than 12 take round 1's path kernel) */
    uint64_t fused_fallback_launches;      /* launches since the context was created that took round 1's path kernel because the tree did not
                                              fit k_fused (deeper than 12 levels, or node / record arrays beyond 4 GiB = 32-bit byte offsets) */
    uint64_t graph_replays;                /* wavefront mode: sample pairs replayed from a captured hipGraph since the context was created */
    uint64_t bvh_alloc_bytes;              /* device memory the TLAS last launched holds for nodes, node boxes and records (bvh_bytes: the packed payload) */
} HrtStats;

int  hrt_stats_reset(HrtContext *ctx);
int  hrt_stats_get(HrtContext *ctx, HrtStats *out);       /* synchronises the device */

/* ---- introspection used by the parity tests ------------------------------------------ */
/* Trace n rays (origin/direction float3 arrays on the device, tmin/tmax as in the shader)
 * through a TLAS and write t,u,v (float) and prim,inst (u32; 0xffffffff on miss) per ray.
 * The rays go through the kernel hrt_render_launch itself runs in the context's configuration: the
 * fused path kernel by default (each ray stands in for a pixel that is traced once and not shaded),
 * the wavefront traverse kernel under HRT_CTX_COUNT or HRT_FUSED=0. */
int  hrt_trace_rays(HrtContext *ctx, HrtTraversable tlas, const HrtFloat3 *d_origins,
                    const HrtFloat3 *d_directions, uint32_t n_rays, float tmin, float tmax,
                    int any_hit, float *d_t, float *d_u, float *d_v,
                    uint32_t *d_prim, uint32_t *d_inst, void *stream);

/* When set (non-NULL), every following launch also writes the linear mean radiance (the value
 * colorToFloat4 is applied to, shader/Shader.cu:270) as W*H float4, so that tests can compare
 * it bit-for-bit with the oracle.  Pass NULL to switch it off. */
int  hrt_debug_set_linear_output(HrtContext *ctx, HrtFloat4 *d_linear);

/* sinf (0) / cosf (1) / acosf (2) / asinf (3) / atan2f (4) as hrt_pose_instances evaluates them (csrc/cr_trig.h: the correctly
 * rounded float of the exact value, the pin shared with the oracle), over n arguments: d_a[i] (atan2: y = d_a[i], x = d_b[i]), or,
 * when d_a is NULL, the floats whose bit patterns are first_bits + i * stride_bits.  force_slow != 0 decides every value in
 * double-double arithmetic (the path one call in ~500 000 takes). */
int  hrt_debug_trig(HrtContext *ctx, int function, const float *d_a, const float *d_b, uint32_t first_bits, uint32_t stride_bits,
                    uint64_t n, int force_slow, float *d_out, void *stream);

/* Host-only BVH8 build over triangles given as 9 floats each (no GPU needed): returns the
 * packed node and primitive blobs the device kernels traverse.  Free with hrt_host_free. */
typedef struct HrtBvhBlob {
    void    *nodes;      uint64_t n_nodes;      /* 80-byte packed BVH8 nodes             */
    void    *triangles;  uint64_t n_triangles;  /* 48-byte triangle records, leaf order  */
    float    bounds[6];
} HrtBvhBlob;
int  hrt_host_build_bvh8(const float *h_triangles, uint32_t n_triangles, HrtBvhBlob *out);
/* Copy the flattened world-space BVH of a TLAS, as it is in device memory now (after any refit), back to
 * the host (same blob format). */
int  hrt_tlas_download(HrtContext *ctx, HrtTraversable tlas, HrtBvhBlob *out);
void hrt_host_free(HrtBvhBlob *blob);

#ifdef __cplusplus
}
#endif
#endif /* HRT_H */
