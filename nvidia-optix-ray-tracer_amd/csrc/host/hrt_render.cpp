// hrt_render.cpp -- small C++ driver playing the role of the reference's startRender loop
// (src/Global/RendererMesh.cu:307-434) for one frame: builds a synthetic triangle soup, renders it through the host mirror
// (renderer_host.hpp; with --gpus N through multi_gpu.hpp: one process, N devices, stripes + one ncclReduce) and writes a
// binary PPM of the 8-bit image.
//   hrt_render [n_triangles=100000] [width=960] [height=540] [spp=4] [out.ppm] [--gpus N] [--check]
// --check renders the frame once more on device 0 alone, untiled, and fails unless the reduced frame is the same bits.
#include "multi_gpu.hpp"

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

int main(int argc, char **argv) {
    std::vector<std::string> pos;
    int n_gpus = 1; bool check = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--gpus" && i + 1 < argc) n_gpus = std::atoi(argv[++i]);
        else if (a == "--check") check = true;
        else pos.push_back(a);
    }
    const uint32_t n_tri = pos.size() > 0 ? (uint32_t)std::atoi(pos[0].c_str()) : 100000u;
    const uint32_t W = pos.size() > 1 ? (uint32_t)std::atoi(pos[1].c_str()) : 960u, H = pos.size() > 2 ? (uint32_t)std::atoi(pos[2].c_str()) : 540u;
    const uint32_t spp = pos.size() > 3 ? (uint32_t)std::atoi(pos[3].c_str()) : 4u;
    const std::string out = pos.size() > 4 ? pos[4] : "hrt_render.ppm";
    const unsigned long long salt = 0x5EED0000C0FFEEull;

    std::vector<HrtFloat3> verts(3 * (size_t)n_tri), normals(3 * (size_t)n_tri);
    uint64_t s = 42; const float edge = 0.6f / std::cbrt((float)n_tri);
    for (uint32_t i = 0; i < n_tri; ++i) {
        const HrtFloat3 c{uni(s, -1, 1), uni(s, -1, 1), uni(s, -1, 1)};
        const HrtFloat3 e1{uni(s, -edge, edge), uni(s, -edge, edge), uni(s, -edge, edge)}, e2{uni(s, -edge, edge), uni(s, -edge, edge), uni(s, -edge, edge)};
        verts[3 * i] = c; verts[3 * i + 1] = {c.x + e1.x, c.y + e1.y, c.z + e1.z}; verts[3 * i + 2] = {c.x + e2.x, c.y + e2.y, c.z + e2.z};
        const HrtFloat3 n = hostNormalize(hostCross(e1, e2));
        normals[3 * i] = normals[3 * i + 1] = normals[3 * i + 2] = n;
    }
    const auto cam = SDL_GraphicsWindowConfigureCamera({0, 0, 3.5f}, {0, 0, 0}, {0, 1, 0}, true);

    MultiGpuRenderer mg;
    const auto t0 = std::chrono::steady_clock::now();
    mg.create(n_gpus, verts, normals, W, H, salt, HRT_CTX_TIMING);       // production kernels, HIP-event timing
    const auto t1 = std::chrono::steady_clock::now();
    mg.render(cam, 1);                                                    // warm-up frame (advances the RNG streams)
    mg.reset_stats();
    const auto t2 = std::chrono::steady_clock::now();
    mg.render(cam, spp);
    const auto t3 = std::chrono::steady_clock::now();
    const uint64_t rays = mg.rays();
    const double ms = std::chrono::duration<double, std::milli>(t3 - t2).count();
    std::printf("%d GPU(s) | scene + build %.1f ms | %u tris %ux%u %u spp: %.2f ms per frame incl. the reduce, %.1f Mrays/s\n", n_gpus,
                std::chrono::duration<double, std::milli>(t1 - t0).count(), n_tri, W, H, spp, ms, rays / ms * 1e-3);

    DeviceFrame &d0 = mg.dev[0];
    hipCheckM(hipSetDevice(d0.device));
    std::vector<HrtFloat4> frame((size_t)W * H);
    hipCheckM(hipMemcpy(frame.data(), d0.color, frame.size() * sizeof(HrtFloat4), hipMemcpyDeviceToHost));
    int rc = 0;
    if (check) {
        // the same two launches on device 0 alone, whole frame, fresh streams: the reference image of the split
        HrtRngState *states = nullptr; HrtFloat4 *color = nullptr;
        RandomGenerator::initDeviceRandomGenerators(d0.ctx, states, W, H, salt, d0.stream);
        hipCheckM(hipMalloc((void **)&color, sizeof(HrtFloat4) * (size_t)W * H));
        HrtGlobalParams params{std::get<0>(d0.ias), states};
        HrtRayGenParams raygen{};
        raygen.width = W; raygen.height = H; raygen.colorBuffer = color;
        raygen.cameraCenter = cam.cameraCenter; raygen.cameraU = cam.cameraU; raygen.cameraV = cam.cameraV; raygen.cameraW = cam.cameraW;
        launch(d0.ctx, params, raygen, 1, nullptr, d0.stream);
        launch(d0.ctx, params, raygen, spp, nullptr, d0.stream);
        std::vector<HrtFloat4> ref((size_t)W * H);
        hipCheckM(hipMemcpy(ref.data(), color, ref.size() * sizeof(HrtFloat4), hipMemcpyDeviceToHost));
        const bool same = std::memcmp(ref.data(), frame.data(), ref.size() * sizeof(HrtFloat4)) == 0;
        std::printf("check: reduced frame of %d GPU(s) %s the single-context frame\n", n_gpus, same ? "IS BIT-IDENTICAL TO" : "DIFFERS FROM");
        rc = same ? 0 : 1;
        RandomGenerator::freeDeviceRandomGenerators(d0.ctx, states, d0.stream);
        hipCheckM(hipFree(color));
    }
    HrtUchar4 *rgba = nullptr;
    hipCheckM(hipMalloc((void **)&rgba, sizeof(HrtUchar4) * (size_t)W * H));
    hrtCheckError(d0.ctx, hrt_to_rgba8(d0.ctx, d0.color, rgba, W, H, d0.stream));
    hipCheckM(hipStreamSynchronize(d0.stream));
    std::vector<HrtUchar4> host((size_t)W * H);
    hipCheckM(hipMemcpy(host.data(), rgba, host.size() * sizeof(HrtUchar4), hipMemcpyDeviceToHost));
    if (FILE *f = std::fopen(out.c_str(), "wb")) {
        std::fprintf(f, "P6\n%u %u\n255\n", W, H);
        for (uint32_t y = H; y-- > 0;) for (uint32_t x = 0; x < W; ++x) std::fwrite(&host[(size_t)y * W + x], 1, 3, f);
        std::fclose(f);
    }
    hipCheckM(hipFree(rgba));
    mg.destroy();
    return rc;
}
