// multi_gpu.hpp -- the multi-GPU frame in C++ with RCCL called directly: ONE process, N devices of one node.
//
// The reference is single-GPU (docs/technical-details.md:327); north_star splits the image into tiles across the GPUs of a
// node with an RCCL reduce of per-tile radiance.  Here: one context + one stream per device, the scene and its tree replicated
// (66 MB of 288 GB), device k renders the interleaved 8-row stripes (y / 8) % N == k of the frame into its own zeroed W x H
// float4 buffer (hrt_render_launch with an HrtTile; the RNG stream is keyed by the global pixel index, shader/Shader.cu:97, so
// the union is the 1-GPU image bit for bit), then ONE collective per frame: ncclReduce(sum) of the float4 frames into device
// 0 over xGMI (x + 0 is exact; 33 MB at 1080p).  No other exchange exists on the path.  The Python twin of this file is
// host.py (tile_for_rank / reduce_tiles over torch.distributed, which bench.py --gpus N uses under torchrun).
#pragma once
#include <rccl/rccl.h>

#include <cstring>
#include <thread>

#include "renderer_host.hpp"

namespace project {

#define ncclCheck(call)                                                                                       \
    do {                                                                                                      \
        const ncclResult_t _r = (call);                                                                       \
        if (_r != ncclSuccess) { std::fprintf(stderr, "[rccl] %s: %s\n", #call, ncclGetErrorString(_r)); std::exit(-400); } \
    } while (0)
#define hipCheckM(x) do { hipError_t _e = (x); if (_e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(_e)); std::exit(-100); } } while (0)

// what every device holds
struct DeviceFrame {
    int device = 0;
    HrtContext *ctx = nullptr;
    hipStream_t stream = nullptr;
    HrtFloat3 *dev_vertices = nullptr, *dev_normals = nullptr;
    HrtInstance *dev_instances = nullptr;
    GAS gas{}; IAS ias{};
    HrtRngState *dev_stateArray = nullptr;
    HrtFloat4 *color = nullptr;
};

struct MultiGpuRenderer {
    uint32_t W = 0, H = 0;
    std::vector<DeviceFrame> dev;
    std::vector<ncclComm_t> comms;

    // scene: one triangle soup, one rough material (the C3 / C4 layout); replicated on every device
    void create(int n_gpus, const std::vector<HrtFloat3> &verts, const std::vector<HrtFloat3> &normals, uint32_t width, uint32_t height,
                unsigned long long seedSalt, uint32_t ctx_flags) {
        W = width; H = height;
        int visible = 0;
        hipCheckM(hipGetDeviceCount(&visible));
        if (n_gpus < 1 || n_gpus > visible) { std::fprintf(stderr, "--gpus %d: %d device(s) visible\n", n_gpus, visible); std::exit(2); }
        dev.resize(n_gpus); comms.resize(n_gpus);
        std::vector<int> ids(n_gpus);
        for (int k = 0; k < n_gpus; ++k) ids[k] = k;
        ncclCheck(ncclCommInitAll(comms.data(), n_gpus, ids.data()));
        std::vector<std::thread> th;
        for (int k = 0; k < n_gpus; ++k)
            th.emplace_back([&, k] {
                DeviceFrame &d = dev[k];
                d.device = k;
                hipCheckM(hipSetDevice(k));
                hipCheckM(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
                d.ctx = createContext(k, false);
                hrtCheckError(d.ctx, hrt_ctx_set_flags(d.ctx, ctx_flags));
                const size_t vb = verts.size() * sizeof(HrtFloat3);
                hipCheckM(hipMalloc((void **)&d.dev_vertices, vb)); hipCheckM(hipMalloc((void **)&d.dev_normals, vb));
                hipCheckM(hipMemcpyAsync(d.dev_vertices, verts.data(), vb, hipMemcpyHostToDevice, d.stream));
                hipCheckM(hipMemcpyAsync(d.dev_normals, normals.data(), vb, hipMemcpyHostToDevice, d.stream));
                RendererTriangle tri{0, 0, d.dev_vertices, d.dev_normals, verts.size() / 3};
                d.gas = buildGASForTriangles(d.ctx, tri, d.stream);
                hipCheckM(hipStreamSynchronize(d.stream));
                hipCheckM(hipFree(d.dev_vertices)); d.dev_vertices = nullptr;            // RendererMesh.cu:116
                HrtInstance inst{}; const float idm[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
                std::memcpy(inst.transform, idm, sizeof idm);
                inst.sbtOffset = 0; inst.visibilityMask = 1; inst.traversableHandle = d.gas.first;
                hipCheckM(hipMalloc((void **)&d.dev_instances, sizeof inst));
                hipCheckM(hipMemcpyAsync(d.dev_instances, &inst, sizeof inst, hipMemcpyHostToDevice, d.stream));
                d.ias = buildIAS(d.ctx, d.dev_instances, 1, d.stream);
                HrtSbtRecord rec{};
                hrtCheckError(d.ctx, hrt_sbt_record_pack_header(HRT_PROGRAM_TRIANGLE_ROUGH, &rec));
                rec.data.triangles.vertexNormals = d.dev_normals;
                rec.data.rough.albedo = {0.73f, 0.73f, 0.73f};
                hrtCheckError(d.ctx, hrt_materials_set(d.ctx, &rec, 1));
                createMissSBTRecord(d.ctx, {0.7f, 0.8f, 0.9f});
                RandomGenerator::initDeviceRandomGenerators(d.ctx, d.dev_stateArray, W, H, seedSalt, d.stream);
                hipCheckM(hipMalloc((void **)&d.color, sizeof(HrtFloat4) * (size_t)W * H));
            });
        for (auto &t : th) t.join();
    }

    // one frame: every device its stripes, then the one reduce into device 0.  Returns when device 0's frame is complete.
    void render(const SDL_GraphicsWindowCamera &cam, uint32_t spp) {
        const int n = (int)dev.size();
        std::vector<std::thread> th;
        for (int k = 0; k < n; ++k)
            th.emplace_back([&, k] {
                DeviceFrame &d = dev[k];
                hipCheckM(hipSetDevice(d.device));
                hipCheckM(hipMemsetAsync(d.color, 0, sizeof(HrtFloat4) * (size_t)W * H, d.stream));       // rows of the other devices must be zero for the sum
                HrtGlobalParams params{std::get<0>(d.ias), d.dev_stateArray};
                HrtRayGenParams raygen{};
                raygen.width = W; raygen.height = H; raygen.colorBuffer = d.color;
                raygen.cameraCenter = cam.cameraCenter; raygen.cameraU = cam.cameraU; raygen.cameraV = cam.cameraV; raygen.cameraW = cam.cameraW;
                const HrtTile tile{0, H, n > 1 ? 8u : 1u, (uint32_t)n, (uint32_t)k};
                hrtCheckError(d.ctx, hrt_render_launch(d.ctx, &params, &raygen, spp, n > 1 ? &tile : nullptr, d.stream));
                ncclCheck(ncclReduce(d.color, d.color, (size_t)W * H * 4, ncclFloat, ncclSum, 0, comms[k], d.stream));
                hipCheckM(hipStreamSynchronize(d.stream));
            });
        for (auto &t : th) t.join();
    }

    uint64_t rays() {
        uint64_t total = 0;
        for (DeviceFrame &d : dev) { HrtStats st{}; hrtCheckError(d.ctx, hrt_stats_get(d.ctx, &st)); total += st.rays; }
        return total;
    }
    void reset_stats() { for (DeviceFrame &d : dev) hrtCheckError(d.ctx, hrt_stats_reset(d.ctx)); }

    void destroy() {
        for (size_t k = 0; k < dev.size(); ++k) {
            DeviceFrame &d = dev[k];
            hipCheckM(hipSetDevice(d.device));
            RandomGenerator::freeDeviceRandomGenerators(d.ctx, d.dev_stateArray, d.stream);
            cleanupAccelerationStructure(d.ctx, d.ias); cleanupAccelerationStructure(d.ctx, d.gas);
            hipCheckM(hipFree(d.dev_normals)); hipCheckM(hipFree(d.dev_instances)); hipCheckM(hipFree(d.color));
            destroyContext(d.ctx);
            hipCheckM(hipStreamDestroy(d.stream));
            ncclCheck(ncclCommDestroy(comms[k]));
        }
        dev.clear(); comms.clear();
    }
};

}  // namespace project
