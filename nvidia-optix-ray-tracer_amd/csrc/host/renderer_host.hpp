// renderer_host.hpp -- C++ host-side mirror of the reference's renderer-implementation layer
// over the C ABI (include/hrt.h).  Same function names and argument meaning as the reference's
// include/Global/RendererImpl.cuh:153-253 for the path that was rebuilt; errors follow the
// reference's convention (log + exit, include/Global/HostFunctions.cuh:147-166), which is what a
// reference-side caller expects from these helpers.  Header-only; link against libhrt.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <tuple>
#include <utility>
#include <vector>

#include "hrt.h"

namespace project {

constexpr int HRT_ERROR_EXIT_CODE = -200;   // the reference's OptiX error exit code (HostFunctions.cu:30-50)

#define hrtCheckError(ctx, call)                                                                  \
    do {                                                                                          \
        const int _rc = (call);                                                                   \
        if (_rc != HRT_OK) {                                                                      \
            std::fprintf(stderr, "[hrt] %s failed (%d): %s\n", #call, _rc, hrt_last_error(ctx)); \
            std::exit(project::HRT_ERROR_EXIT_CODE);                                              \
        }                                                                                         \
    } while (0)

typedef std::pair<HrtTraversable, void *> GAS;                 // (handle, unused) -- RendererImpl.cuh:20
typedef std::tuple<HrtTraversable, void *, size_t> IAS;        // RendererImpl.cuh:21

struct RendererSphere { int materialType; size_t materialIndex; HrtFloat3 *dev_centers; float *dev_radii; size_t count; };
struct RendererTriangle { int materialType; size_t materialIndex; HrtFloat3 *dev_vertices; HrtFloat3 *dev_normals; size_t count; };

// createContext / destroyContext, src/Global/RendererImpl.cu:6-27
inline HrtContext *createContext(int device = 0, bool isDebugMode = false) {
    HrtContext *ctx = nullptr;
    const int rc = hrt_ctx_create(device, isDebugMode ? (HRT_CTX_TIMING | HRT_CTX_COUNT) : 0u, &ctx);
    if (rc != HRT_OK) { std::fprintf(stderr, "[hrt] createContext: %s\n", hrt_last_error(nullptr)); std::exit(HRT_ERROR_EXIT_CODE); }
    return ctx;
}
inline void destroyContext(HrtContext *&ctx) { hrt_ctx_destroy(ctx); ctx = nullptr; }

// buildGASForSpheres / buildGASForTriangles, src/Global/RendererImpl.cu:113-172 (count = triangles, Q6 fixed: 3 vertices each)
inline GAS buildGASForSpheres(HrtContext *ctx, const RendererSphere &s, hipStream_t stream = nullptr) {
    HrtTraversable h = 0;
    hrtCheckError(ctx, hrt_blas_build_spheres(ctx, s.dev_centers, s.dev_radii, (uint32_t)s.count, stream, &h));
    return {h, nullptr};
}
inline GAS buildGASForTriangles(HrtContext *ctx, const RendererTriangle &t, hipStream_t stream = nullptr) {
    HrtTraversable h = 0;
    hrtCheckError(ctx, hrt_blas_build_triangles(ctx, t.dev_vertices, (uint32_t)(3 * t.count), stream, &h));
    return {h, nullptr};
}
// buildIAS / updateIAS, src/Global/RendererImpl.cu:174-242
inline IAS buildIAS(HrtContext *ctx, const HrtInstance *dev_instances, size_t instanceCount, hipStream_t stream = nullptr) {
    HrtTraversable h = 0;
    hrtCheckError(ctx, hrt_tlas_build(ctx, dev_instances, (uint32_t)instanceCount, stream, &h));
    return {h, nullptr, instanceCount};
}
inline void updateIAS(HrtContext *ctx, IAS &ias, const HrtInstance *dev_instances, size_t instanceCount, hipStream_t stream = nullptr) {
    hrtCheckError(ctx, hrt_tlas_update(ctx, std::get<0>(ias), dev_instances, (uint32_t)instanceCount, stream));
}
inline void cleanupAccelerationStructure(HrtContext *ctx, GAS &g) { hrt_blas_destroy(ctx, g.first); g = {}; }
inline void cleanupAccelerationStructure(HrtContext *ctx, IAS &i) { hrt_tlas_destroy(ctx, std::get<0>(i)); i = {}; }

// SBT records, src/Global/RendererImpl.cu:472-575.  The four OptixProgramGroup arguments of the reference become the
// HrtProgram ids hrt_sbt_record_pack_header takes; the records keep their layout and are handed to hrt_materials_set
// (record i <-> instance with sbtOffset i) where the reference copies them to the device (RendererMesh.cu:283-305).
enum MaterialType { ROUGH = 0, METAL = 1 };                                   // include/Global/Shader.cuh:16-18
struct RendererMaterial { std::vector<HrtFloat3> roughs; std::vector<std::pair<HrtFloat3, float>> metals; };   // RendererImpl.cuh:109-112
typedef HrtSbtRecord HitGroupSbtRecord;

inline void packMaterial(HrtContext *ctx, HitGroupSbtRecord &record, bool sphere, int materialType, size_t materialIndex,
                         const RendererMaterial &globalMaterials) {
    if (materialType == ROUGH) {
        hrtCheckError(ctx, hrt_sbt_record_pack_header(sphere ? HRT_PROGRAM_SPHERE_ROUGH : HRT_PROGRAM_TRIANGLE_ROUGH, &record));
        record.data.rough.albedo = globalMaterials.roughs.at(materialIndex);
    } else {
        hrtCheckError(ctx, hrt_sbt_record_pack_header(sphere ? HRT_PROGRAM_SPHERE_METAL : HRT_PROGRAM_TRIANGLE_METAL, &record));
        record.data.metal.albedo = globalMaterials.metals.at(materialIndex).first;
        record.data.metal.fuzz = globalMaterials.metals.at(materialIndex).second;
    }
}
// createAddSphereTriangleSBTRecord, RendererImpl.cu:494-552: spheres first, then triangles
inline std::vector<HitGroupSbtRecord> createAddSphereTriangleSBTRecord(HrtContext *ctx, const std::vector<RendererSphere> &spheres,
                                                                       const std::vector<RendererTriangle> &triangles,
                                                                       const RendererMaterial &globalMaterials) {
    std::vector<HitGroupSbtRecord> records;
    records.reserve(spheres.size() + triangles.size());
    for (const auto &sphere : spheres) {
        HitGroupSbtRecord record = {};
        record.data.sphere.centers = sphere.dev_centers; record.data.sphere.radii = sphere.dev_radii;
        packMaterial(ctx, record, true, sphere.materialType, sphere.materialIndex, globalMaterials);
        records.push_back(record);
    }
    for (const auto &triangle : triangles) {
        HitGroupSbtRecord record = {};
        record.data.triangles.vertexNormals = triangle.dev_normals;
        packMaterial(ctx, record, false, triangle.materialType, triangle.materialIndex, globalMaterials);
        records.push_back(record);
    }
    return records;
}
// createVTKParticleSBTRecord, RendererImpl.cu:554-575: (rough material index, vertex normals) per particle, all ROUGH
inline std::vector<HitGroupSbtRecord> createVTKParticleSBTRecord(HrtContext *ctx, const std::vector<std::pair<size_t, HrtFloat3 *>> &particles,
                                                                 const RendererMaterial &globalMaterials) {
    std::vector<HitGroupSbtRecord> records;
    records.reserve(particles.size());
    for (const auto &particle : particles) {
        HitGroupSbtRecord record = {};
        record.data.triangles.vertexNormals = particle.second;
        packMaterial(ctx, record, false, ROUGH, particle.first, globalMaterials);
        records.push_back(record);
    }
    return records;
}
// the miss half of createRaygenMissSBTRecord, RendererImpl.cu:472-492 (the raygen record's payload is HrtRayGenParams, passed per launch)
inline void createMissSBTRecord(HrtContext *ctx, const HrtFloat3 &backgroundColor) {
    const HrtMissParams miss{backgroundColor};
    hrtCheckError(ctx, hrt_miss_set(ctx, &miss));
}

// RandomGenerator::initDeviceRandomGenerators, src/Global/HostFunctions.cu:128-140 (seedSalt pins clock64())
struct RandomGenerator {
    static void initDeviceRandomGenerators(HrtContext *ctx, HrtRngState *&dev_stateArray, size_t x, size_t y,
                                           unsigned long long seedSalt, hipStream_t stream = nullptr) {
        hrtCheckError(ctx, hrt_rng_init(ctx, (uint32_t)x, (uint32_t)y, seedSalt, stream, &dev_stateArray));
    }
    static void freeDeviceRandomGenerators(HrtContext *ctx, HrtRngState *&dev_stateArray, hipStream_t stream = nullptr) {
        hrtCheckError(ctx, hrt_rng_free(ctx, dev_stateArray, stream));
        dev_stateArray = nullptr;
    }
};

// SDL_GraphicsWindowConfigureCamera, src/GraphicsAPI/SDL_GraphicsWindow.cu:4-14
struct SDL_GraphicsWindowCamera { HrtFloat3 upDirection, cameraCenter, cameraTarget, cameraU, cameraV, cameraW; };
inline HrtFloat3 hostNormalize(HrtFloat3 v) {
    const float len2 = v.x * v.x + v.y * v.y + v.z * v.z;
    if (len2 <= HRT_FLOAT_ZERO_VALUE * HRT_FLOAT_ZERO_VALUE) return {0.0f, 0.0f, 1.0f};
    const float inv = 1.0f / std::sqrt(len2);
    return {v.x * inv, v.y * inv, v.z * inv};
}
inline HrtFloat3 hostCross(HrtFloat3 a, HrtFloat3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline SDL_GraphicsWindowCamera SDL_GraphicsWindowConfigureCamera(const HrtFloat3 &center, const HrtFloat3 &target, const HrtFloat3 &up, bool isOpenGL) {
    SDL_GraphicsWindowCamera camera{};
    camera.upDirection = hostNormalize(up); camera.cameraCenter = center; camera.cameraTarget = target;
    if (!isOpenGL) camera.upDirection = {-camera.upDirection.x, -camera.upDirection.y, -camera.upDirection.z};
    camera.cameraW = {target.x - center.x, target.y - center.y, target.z - center.z};
    camera.cameraU = hostNormalize(hostCross(camera.cameraW, camera.upDirection));
    camera.cameraV = hostNormalize(hostCross(camera.cameraU, camera.cameraW));
    return camera;
}

// the launch: optixLaunch + cudaDeviceSynchronize, src/Global/RendererMesh.cu:416-419
inline void launch(HrtContext *ctx, const HrtGlobalParams &params, const HrtRayGenParams &raygen, unsigned int spp = 1,
                   const HrtTile *tile = nullptr, hipStream_t stream = nullptr) {
    hrtCheckError(ctx, hrt_render_launch(ctx, &params, &raygen, spp, tile, stream));
    hrtCheckError(ctx, hrt_sync(ctx, stream));
}

}  // namespace project
