// hrt_render.cpp -- small C++ driver playing the role of the reference's startRender loop
// (src/Global/RendererMesh.cu:307-434) for one frame: builds a synthetic triangle soup, renders
// it through the host mirror (renderer_host.hpp) and writes a binary PPM of the 8-bit image.
//   hrt_render [n_triangles=100000] [width=960] [height=540] [spp=4] [out.ppm]
#include "renderer_host.hpp"

#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>

using namespace project;

static uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static float uni(uint64_t &s, float lo, float hi) { return lo + (hi - lo) * ((float)(splitmix(s) >> 40) * 0x1p-24f); }

#define hipCheck(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(-100); } } while (0)

int main(int argc, char **argv) {
    const uint32_t n_tri = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 100000u;
    const uint32_t W = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 960u, H = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 540u;
    const uint32_t spp = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 4u;
    const std::string out = argc > 5 ? argv[5] : "hrt_render.ppm";

    std::vector<HrtFloat3> verts(3 * (size_t)n_tri), normals(3 * (size_t)n_tri);
    uint64_t s = 42; const float edge = 0.6f / std::cbrt((float)n_tri);
    for (uint32_t i = 0; i < n_tri; ++i) {
        const HrtFloat3 c{uni(s, -1, 1), uni(s, -1, 1), uni(s, -1, 1)};
        const HrtFloat3 e1{uni(s, -edge, edge), uni(s, -edge, edge), uni(s, -edge, edge)}, e2{uni(s, -edge, edge), uni(s, -edge, edge), uni(s, -edge, edge)};
        verts[3 * i] = c; verts[3 * i + 1] = {c.x + e1.x, c.y + e1.y, c.z + e1.z}; verts[3 * i + 2] = {c.x + e2.x, c.y + e2.y, c.z + e2.z};
        const HrtFloat3 n = hostNormalize(hostCross(e1, e2));
        normals[3 * i] = normals[3 * i + 1] = normals[3 * i + 2] = n;
    }

    HrtContext *ctx = createContext(0, false);
    hrtCheckError(ctx, hrt_ctx_set_flags(ctx, HRT_CTX_TIMING));      // production kernels, HIP-event timing
    RendererTriangle tri{0, 0, nullptr, nullptr, n_tri};
    hipCheck(hipMalloc((void **)&tri.dev_vertices, verts.size() * sizeof(HrtFloat3)));
    hipCheck(hipMalloc((void **)&tri.dev_normals, normals.size() * sizeof(HrtFloat3)));
    hipCheck(hipMemcpy(tri.dev_vertices, verts.data(), verts.size() * sizeof(HrtFloat3), hipMemcpyHostToDevice));
    hipCheck(hipMemcpy(tri.dev_normals, normals.data(), normals.size() * sizeof(HrtFloat3), hipMemcpyHostToDevice));
    const auto t0 = std::chrono::steady_clock::now();
    GAS gas = buildGASForTriangles(ctx, tri);
    hipCheck(hipFree(tri.dev_vertices));                                    // RendererMesh.cu:116

    HrtInstance inst{}; const float idm[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    std::memcpy(inst.transform, idm, sizeof idm);
    inst.sbtOffset = 0; inst.visibilityMask = 1; inst.traversableHandle = gas.first;
    HrtInstance *dev_instances = nullptr;
    hipCheck(hipMalloc((void **)&dev_instances, sizeof inst));
    hipCheck(hipMemcpy(dev_instances, &inst, sizeof inst, hipMemcpyHostToDevice));
    IAS ias = buildIAS(ctx, dev_instances, 1);
    const auto t1 = std::chrono::steady_clock::now();

    HrtSbtRecord rec{};
    hrtCheckError(ctx, hrt_sbt_record_pack_header(HRT_PROGRAM_TRIANGLE_ROUGH, &rec));
    rec.data.triangles.vertexNormals = tri.dev_normals;
    rec.data.rough.albedo = {0.73f, 0.73f, 0.73f};
    hrtCheckError(ctx, hrt_materials_set(ctx, &rec, 1));
    const HrtMissParams miss{{0.7f, 0.8f, 0.9f}};
    hrtCheckError(ctx, hrt_miss_set(ctx, &miss));

    HrtRngState *dev_stateArray = nullptr;
    RandomGenerator::initDeviceRandomGenerators(ctx, dev_stateArray, W, H, 0x5EED0000C0FFEEull);
    const auto cam = SDL_GraphicsWindowConfigureCamera({0, 0, 3.5f}, {0, 0, 0}, {0, 1, 0}, true);

    HrtFloat4 *color = nullptr; HrtUchar4 *rgba = nullptr;
    hipCheck(hipMalloc((void **)&color, sizeof(HrtFloat4) * (size_t)W * H));
    hipCheck(hipMalloc((void **)&rgba, sizeof(HrtUchar4) * (size_t)W * H));
    HrtGlobalParams params{std::get<0>(ias), dev_stateArray};
    HrtRayGenParams raygen{};
    raygen.width = W; raygen.height = H; raygen.colorBuffer = color;
    raygen.cameraCenter = cam.cameraCenter; raygen.cameraU = cam.cameraU; raygen.cameraV = cam.cameraV; raygen.cameraW = cam.cameraW;

    launch(ctx, params, raygen, 1);                                         // warm-up frame
    hrtCheckError(ctx, hrt_stats_reset(ctx));
    const auto t2 = std::chrono::steady_clock::now();
    launch(ctx, params, raygen, spp);
    const auto t3 = std::chrono::steady_clock::now();
    HrtStats st{};
    hrtCheckError(ctx, hrt_stats_get(ctx, &st));
    const double ms = std::chrono::duration<double, std::milli>(t3 - t2).count();
    std::printf("build %.1f ms | %u tris %ux%u %u spp: %.2f ms, %.1f Mrays/s, %.2f rays/path\n",
                std::chrono::duration<double, std::milli>(t1 - t0).count(), n_tri, W, H, spp, ms, st.rays / ms * 1e-3,
                (double)st.rays / st.paths);
    for (int k = 0; k < HRT_K_COUNT; ++k)
        if (st.kernel_launches[k]) std::printf("  kernel class %d: %.3f ms in %llu launches\n", k, st.kernel_ms[k], (unsigned long long)st.kernel_launches[k]);

    hrtCheckError(ctx, hrt_to_rgba8(ctx, color, rgba, W, H, nullptr));
    std::vector<HrtUchar4> host((size_t)W * H);
    hipCheck(hipMemcpy(host.data(), rgba, host.size() * sizeof(HrtUchar4), hipMemcpyDeviceToHost));
    if (FILE *f = std::fopen(out.c_str(), "wb")) {
        std::fprintf(f, "P6\n%u %u\n255\n", W, H);
        for (uint32_t y = H; y-- > 0;) for (uint32_t x = 0; x < W; ++x) std::fwrite(&host[(size_t)y * W + x], 1, 3, f);
        std::fclose(f);
    }
    RandomGenerator::freeDeviceRandomGenerators(ctx, dev_stateArray);
    cleanupAccelerationStructure(ctx, ias); cleanupAccelerationStructure(ctx, gas);
    hipCheck(hipFree(tri.dev_normals)); hipCheck(hipFree(dev_instances)); hipCheck(hipFree(color)); hipCheck(hipFree(rgba));
    destroyContext(ctx);
    return 0;
}
