/*
 * hrt_params.h -- launch-parameter / SBT / instance structs of the render call.
 *
 * These are the byte-compatible mirrors of the structs the reference hands to
 * optixLaunch and to the acceleration-structure builders.  A reference-side
 * caller can memcpy its own structs into these (or pass pointers directly).
 *
 *   HrtGlobalParams   <- GlobalParams      include/Global/Shader.cuh:21-24   (16 B)
 *   HrtRayGenParams   <- RayGenParams      include/Global/Shader.cuh:27-35   (80 B)
 *   HrtMissParams     <- MissParams        include/Global/Shader.cuh:38-40   (12 B)
 *   HrtHitGroupParams <- HitGroupParams    include/Global/Shader.cuh:43-70   (32 B)
 *   HrtSbtRecord      <- SbtRecord<HitGroupParams>  include/Global/RendererImpl.cuh:9-16 (64 B)
 *   HrtInstance       <- OptixInstance as filled at src/Global/RendererMesh.cu:131-144 (80 B)
 *   HrtRngState       <- curandState (XORWOW) as allocated at src/Global/HostFunctions.cu:133 (48 B)
 *   HrtParticleState  <- the fields of RendererTimeParticleReference the frame loop reads,
 *                        include/Global/RendererImpl.cuh:93-99 (48 B)
 *
 * Plain C, no HIP/torch types.  float3 is three packed floats (12 B, align 4);
 * float4 is four floats (16 B, align 16) exactly as the CUDA vector types.
 */
#ifndef HRT_PARAMS_H
#define HRT_PARAMS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- compile-time constants of the reference shader ---------------------------------- */
#define HRT_RAY_TRACE_DEPTH   5u        /* rayTraceDepth, Shader.cuh:8                      */
#define HRT_FLOAT_ZERO_VALUE  1e-6f     /* FLOAT_ZERO_VALUE, DeviceFunctions.cuh:18 (tMin)  */
#define HRT_FLOAT_INF_VALUE   1e16f     /* FLOAT_INFINITY_VALUE, DeviceFunctions.cuh:19     */

typedef struct HrtFloat3 { float x, y, z; } HrtFloat3;                       /* 12 B */
typedef struct HrtFloat4 { float x, y, z, w; } __attribute__((aligned(16))) HrtFloat4;
typedef struct HrtUchar4 { unsigned char x, y, z, w; } HrtUchar4;

/* Opaque traversable handle (OptixTraversableHandle is an unsigned long long too). */
typedef uint64_t HrtTraversable;

/* XORWOW generator state, laid out like cuRAND's curandStateXORWOW_t so that
 * GlobalParams.stateArray keeps its element size (48 B) and field order. */
typedef struct HrtRngState {
    uint32_t d;                 /* Weyl counter                                   */
    uint32_t v[5];              /* xorshift words                                 */
    int32_t  boxmuller_flag;
    int32_t  boxmuller_flag_double;
    float    boxmuller_extra;
    float    _pad;
    double   boxmuller_extra_double;
} HrtRngState;                  /* 48 B */

/* GlobalParams: the __constant__ "params" block of the launch. */
typedef struct HrtGlobalParams {
    HrtTraversable handle;      /* TLAS returned by hrt_tlas_build                */
    HrtRngState   *stateArray;  /* device pointer, one state per pixel (y*W+x)    */
} HrtGlobalParams;              /* 16 B */

/* RayGenParams: payload of the raygen SBT record. */
typedef struct HrtRayGenParams {
    uint32_t   width, height;
    HrtFloat4 *colorBuffer;     /* device, W*H float4: sRGB-encoded radiance      */
    HrtFloat4 *albedoBuffer;    /* device, W*H float4 (always zero: quirk Q3) or NULL */
    HrtFloat4 *normalBuffer;    /* device, W*H float4 (always zero: quirk Q3) or NULL */
    HrtFloat3  cameraCenter;
    HrtFloat3  cameraU, cameraV, cameraW;
} HrtRayGenParams;              /* 80 B */

typedef struct HrtMissParams {
    HrtFloat3 backgroundColor;
} HrtMissParams;                /* 12 B */

/* HitGroupParams: per-instance shading data. */
typedef struct HrtHitGroupParams {
    union {
        struct { HrtFloat3 *centers; float *radii; } sphere;     /* device pointers */
        struct { HrtFloat3 *vertexNormals; } triangles;          /* device pointer, 3 per triangle */
    };
    union {
        struct { HrtFloat3 albedo; } rough;
        struct { HrtFloat3 albedo; float fuzz; } metal;
    };
} HrtHitGroupParams;            /* 32 B */

/* The four closest-hit programs of the reference (shader/Shader.cu:297-310).
 * The id is what hrt_sbt_record_pack_header writes into the record header, the
 * role optixSbtRecordPackHeader(programGroup, &record) plays in the reference
 * (src/Global/RendererImpl.cu:514-560). */
typedef enum HrtProgram {
    HRT_PROGRAM_SPHERE_ROUGH   = 0,
    HRT_PROGRAM_SPHERE_METAL   = 1,
    HRT_PROGRAM_TRIANGLE_ROUGH = 2,
    HRT_PROGRAM_TRIANGLE_METAL = 3,
    HRT_PROGRAM_COUNT          = 4
} HrtProgram;

#define HRT_SBT_RECORD_HEADER_SIZE 32   /* OPTIX_SBT_RECORD_HEADER_SIZE */

typedef struct HrtSbtRecord {
    unsigned char     header[HRT_SBT_RECORD_HEADER_SIZE];   /* opaque; packed by hrt_sbt_record_pack_header */
    HrtHitGroupParams data;
} __attribute__((aligned(16))) HrtSbtRecord;                /* 64 B */

/* One instance of the IAS, in the OptixInstance layout. */
typedef struct HrtInstance {
    float          transform[12];       /* row-major 3x4 object->world                */
    uint32_t       instanceId;
    uint32_t       sbtOffset;           /* index of this instance's HrtSbtRecord      */
    uint32_t       visibilityMask;      /* reference always sets 1 and traces mask 1  */
    uint32_t       flags;
    HrtTraversable traversableHandle;   /* BLAS from hrt_blas_build_*                 */
    uint32_t       pad[2];
} __attribute__((aligned(16))) HrtInstance;                 /* 80 B */

/* Rows of the frame a launch renders.  The reference always renders the whole
 * frame; the multi-GPU tile split renders a subset of rows per GPU.  A row y of
 * [y_begin, y_end) belongs to the tile iff (y / stripe_rows) % stripe_period == stripe_phase.
 * {0, H, 1, 1, 0} is the full frame.  Pixel/RNG indices stay global (y*W+x). */
typedef struct HrtTile {
    uint32_t y_begin, y_end;
    uint32_t stripe_rows, stripe_period, stripe_phase;
} HrtTile;

/* One particle of one time step: what src/Global/RendererTime.cu:436-472 reads per particle and frame.
 * quat keeps the reference's float4 field order as loaded at src/Util/VTKReaderImpl.cpp:194-200
 * (x,y,z,w = file components 0..3). */
typedef struct HrtParticleState {
    HrtFloat4 quat;
    HrtFloat3 position;
    HrtFloat3 velocity;
    uint32_t  _pad[2];
} HrtParticleState;             /* 48 B */

/* Frame-loop scalars of the pose update (src/Global/RendererTime.cu:425-470). */
typedef struct HrtPoseParams {
    float     duration;         /* data.durations[currentFileIndex]                               */
    uint32_t  frame;            /* frameCount                                                     */
    uint32_t  frame_count;      /* frameCountThisFile                                             */
    HrtFloat3 particle_offset;  /* loopData.particleOffset                                        */
    HrtFloat3 particle_scale;   /* loopData.particleScale                                         */
    uint32_t  mesh_mode;        /* 0: Time mode (position + velocity, slerp of the quaternions);
                                   1: Mesh mode, src/Global/RendererMesh.cu:379-391 -- the geometry of
                                   a file is already posed, the instance only drifts:
                                   constructTransformMatrix(offset + (velocity * duration / frame_count)
                                   * frame, {0,0,0}, scale); position and quat are not read          */
} HrtPoseParams;

#ifdef __cplusplus
}
/* layout checks (values verified against the reference structs) */
static_assert(sizeof(HrtGlobalParams) == 16, "GlobalParams is 16 B");
static_assert(sizeof(HrtRayGenParams) == 80, "RayGenParams is 80 B");
static_assert(sizeof(HrtMissParams) == 12, "MissParams is 12 B");
static_assert(sizeof(HrtHitGroupParams) == 32, "HitGroupParams is 32 B");
static_assert(sizeof(HrtSbtRecord) == 64, "SbtRecord<HitGroupParams> is 64 B");
static_assert(sizeof(HrtInstance) == 80, "OptixInstance is 80 B");
static_assert(sizeof(HrtRngState) == 48, "curandState is 48 B");
static_assert(sizeof(HrtParticleState) == 48, "HrtParticleState is 48 B");
#endif

#endif /* HRT_PARAMS_H */
