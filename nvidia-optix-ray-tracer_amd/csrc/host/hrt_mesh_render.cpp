// hrt_mesh_render.cpp -- the reference's Mesh-mode program flow, headless, in the reference's own language:
// main() (src/Global/Main.cu:21-36) -> RendererMesh::writeCacheFilesAndExit (src/Util/VTKMeshReader.cu:146-215) when the
// config says "cache", else RendererMesh::commitRendererData (src/Global/RendererMesh.cu:160-310) -> RendererMesh::startRender's
// frame loop (:312-440) with the window, camera input and denoiser left out.
//   hrt_mesh_render <config.json> [exe_dir] [max_frames=one pass over the files] [out.ppm] [width height]
// exe_dir is the directory the config's relative paths are relative to (the reference runs from bin/).
// Loading: up to cache-process-thread-count loader threads, each with its own stream, read particleN.cache and build one GAS per
// particle plus the file's IAS ON THE DEVICE (readVTKFileCache, :93-157).  Per frame: every particle drifts by its velocity
// (hrt_pose_instances in Mesh mode replaces the host loop :379-391 and the H2D copy :395-397) -> updateIAS -> launch + sync ->
// convert to 8 bit; the last frame is written as PPM.
#include "renderer_host.hpp"
#include "hrt_io.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <deque>
#include <string>
#include <thread>

using namespace project;

#define hipCheck(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(-100); } } while (0)
#define ioCheck(x) do { if ((x) != 0) { std::fprintf(stderr, "%s\n", hrt_io_last_error()); std::exit(-1); } } while (0)   // VTK_READER_ERROR_EXIT_CODE

static std::string join(const std::string &base, const std::string &p) { return (!p.empty() && p[0] == '/') ? p : base + "/" + p; }

template <typename T> static T *toDevice(const T *host, size_t count, hipStream_t stream = nullptr) {
    T *dev = nullptr;
    hipCheck(hipMalloc((void **)&dev, std::max<size_t>(1, count) * sizeof(T)));
    if (count) hipCheck(hipMemcpyAsync(dev, host, count * sizeof(T), hipMemcpyHostToDevice, stream));
    return dev;
}

// bounded pool of loader threads, oldest joined first: the reference's std::deque<std::thread> scheme (RendererMesh.cu:204-218)
template <typename F> static void forEachFile(size_t fileCount, size_t maxThreads, F &&work) {
    const size_t threadCount = std::max<size_t>(1, std::min<size_t>(std::thread::hardware_concurrency(), maxThreads));
    std::deque<std::thread> workers;
    for (size_t i = 0; i < fileCount; ++i) {
        workers.emplace_back(work, i);
        if (workers.size() == threadCount) { workers.front().join(); workers.pop_front(); }
    }
    while (!workers.empty()) { workers.front().join(); workers.pop_front(); }
}

// one VTK time step as Mesh mode keeps it: a GAS and the vertex normals per particle, the drift velocities, the IAS over all of them
struct FileData {
    std::vector<GAS> gas;                      // extra geometry first, then the particles (the order of the SBT records)
    std::vector<HrtFloat3 *> dev_normals;      // per particle
    std::vector<uint64_t> ids;                 // per particle: index into the ramp
    HrtInstance *dev_instances = nullptr;
    HrtParticleState *dev_states = nullptr;    // only the velocity is read in Mesh mode
    IAS ias{};
    std::vector<HrtSbtRecord> records;
    size_t instanceCount = 0, particleCount = 0, triangleCount = 0;
};

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s <config.json> [exe_dir] [max_frames] [out.ppm] [width height]\n", argv[0]); return 2; }
    const std::string configPath = argv[1];
    const std::string exeDir = argc > 2 ? argv[2] : ".";
    const long maxFrames = argc > 3 ? std::atol(argv[3]) : -1;
    const std::string out = argc > 4 ? argv[4] : "hrt_mesh_render.ppm";

    HrtIoConfig cfg;
    ioCheck(hrt_io_load_config(configPath.c_str(), &cfg));
    if (!cfg.mesh) { std::fprintf(stderr, "config selects Time mode; this driver plays Mesh mode (hrt_time_render plays the other)\n"); return 2; }
    const uint32_t W = argc > 6 ? (uint32_t)std::atoi(argv[5]) : (uint32_t)cfg.window_width;
    const uint32_t H = argc > 6 ? (uint32_t)std::atoi(argv[6]) : (uint32_t)cfg.window_height;
    const size_t maxThreads = std::max<size_t>(1, (size_t)cfg.cache_process_thread_count);
    std::string cacheDir = join(exeDir, cfg.cache_path);
    if (cacheDir.empty() || cacheDir.back() != '/') cacheDir += '/';

    HrtIoSeries series;
    ioCheck(hrt_io_read_series(join(exeDir, cfg.series_path).c_str(), cfg.series_name, &series));

    if (cfg.cache) {
        // writeCacheFilesAndExit: every VTK file of the series -> particleN.cache, the largest cell count -> metadata.cache
        std::atomic<uint64_t> maxCellCount(0), done(0);
        forEachFile(series.n, maxThreads, [&](size_t i) {
            HrtIoMeshCache m; uint64_t cells = 0;
            ioCheck(hrt_io_read_vtk_mesh_file(series.files[i], &m, &cells));
            ioCheck(hrt_io_write_mesh_cache((cacheDir + "particle" + std::to_string(i) + ".cache").c_str(), &m));
            hrt_io_free_mesh_cache(&m);
            uint64_t seen = maxCellCount.load();
            while (cells > seen && !maxCellCount.compare_exchange_weak(seen, cells)) {}
            std::printf("[%llu/%llu] wrote the cache of %s\n", (unsigned long long)++done, (unsigned long long)series.n, series.files[i]);
        });
        ioCheck(hrt_io_write_metadata_cache(cacheDir.c_str(), maxCellCount.load()));
        std::printf("%llu cache files written to %s, max cell count %llu\n", (unsigned long long)series.n, cacheDir.c_str(), (unsigned long long)maxCellCount.load());
        hrt_io_free_series(&series); hrt_io_free_config(&cfg);
        return 0;
    }

    HrtContext *ctx = createContext(0, false);
    // the frame loop only moves instances: pose kernel, updateIAS and the launch are enqueued back to back with no read-back in between
    const uint32_t flags = (std::getenv("HRT_MESH_RENDER_KERNEL_TIMES") ? HRT_CTX_TIMING : 0u) | (std::getenv("HRT_MESH_RENDER_SYNC_UPDATE") ? 0u : HRT_CTX_ASYNC_UPDATE);
    hrtCheckError(ctx, hrt_ctx_set_flags(ctx, flags));

    // extra geometry: one GAS per sphere of the config (buildAddDataGAS, RendererMesh.cu:79-91), shared by every file
    std::vector<GAS> addGAS;
    std::vector<HrtFloat3 *> sphereCenters; std::vector<float *> sphereRadii;
    std::vector<RendererSphere> addSpheres;
    for (uint64_t i = 0; i < cfg.n_spheres; ++i) {
        const HrtFloat3 c{cfg.spheres[i].center[0], cfg.spheres[i].center[1], cfg.spheres[i].center[2]};
        RendererSphere s{cfg.spheres[i].metal ? METAL : ROUGH, (size_t)cfg.spheres[i].material_index, toDevice(&c, 1), toDevice(&cfg.spheres[i].radius, 1), 1};
        sphereCenters.push_back(s.dev_centers); sphereRadii.push_back(s.dev_radii);
        addGAS.push_back(buildGASForSpheres(ctx, s));
        addSpheres.push_back(s);
    }
    const size_t addGeoCount = addGAS.size();

    // the loader threads (readVTKFileCache): cache -> device, one GAS per particle, the instances, the IAS -- all on the thread's stream
    std::vector<FileData> perFile(series.n);
    std::atomic<size_t> processed(0);
    const auto tLoad = std::chrono::steady_clock::now();
    forEachFile(series.n, maxThreads, [&](size_t f) {
        hipStream_t stream = nullptr;
        hipCheck(hipStreamCreate(&stream));
        HrtIoMeshCache m;
        ioCheck(hrt_io_read_mesh_cache((cacheDir + "particle" + std::to_string(f) + ".cache").c_str(), &m));
        FileData &fd = perFile[f];
        fd.gas = addGAS;
        std::vector<HrtParticleState> states(m.n_particles);
        for (uint64_t p = 0; p < m.n_particles; ++p) {
            const uint64_t first = m.first_triangle[p], count = m.first_triangle[p + 1] - first;
            RendererTriangle t{ROUGH, 0, toDevice(reinterpret_cast<const HrtFloat3 *>(m.vertices + 9 * first), 3 * count, stream),
                               toDevice(reinterpret_cast<const HrtFloat3 *>(m.normals + 9 * first), 3 * count, stream), (size_t)count};
            fd.gas.push_back(buildGASForTriangles(ctx, t, stream));
            hipCheck(hipStreamSynchronize(stream));        // the GAS has its own copy of the vertices (hrt.h): free them, as :116 does
            hipCheck(hipFree(t.dev_vertices));
            fd.dev_normals.push_back(t.dev_normals);
            fd.ids.push_back(m.ids[p]);
            std::memset(&states[p], 0, sizeof states[p]);
            states[p].velocity = {m.velocities[3 * p], m.velocities[3 * p + 1], m.velocities[3 * p + 2]};
            fd.triangleCount += count;
        }
        fd.particleCount = m.n_particles;
        fd.instanceCount = fd.gas.size();
        const float identity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        std::vector<HrtInstance> inst(fd.instanceCount);
        for (size_t i = 0; i < fd.instanceCount; ++i) {
            std::memset(&inst[i], 0, sizeof inst[i]);
            // (the reference starts every instance at the identity, :130-134, and lets its per-frame callback place instance 0,
            // Main.cu:5-9; the spheres' static matrices are applied here once instead)
            std::memcpy(inst[i].transform, i < addGeoCount ? cfg.spheres[i].transform : identity, sizeof identity);
            inst[i].sbtOffset = (uint32_t)i; inst[i].visibilityMask = 1; inst[i].traversableHandle = fd.gas[i].first;
        }
        fd.dev_instances = toDevice(inst.data(), inst.size(), stream);
        fd.dev_states = toDevice(states.data(), states.size(), stream);
        hipCheck(hipStreamSynchronize(stream));            // the staging vectors go out of scope
        fd.ias = buildIAS(ctx, fd.dev_instances, fd.instanceCount, stream);
        hipCheck(hipStreamSynchronize(stream));
        hrt_io_free_mesh_cache(&m);
        hipCheck(hipStreamDestroy(stream));
        std::printf("[%zu/%llu] read the cache of file %zu: %zu particles, %zu triangles\n", ++processed, (unsigned long long)series.n, f, fd.particleCount, fd.triangleCount);
    });
    const double loadMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tLoad).count();

    // materials: the config's, then the baked ramp with one colour per cell of the largest file (:222-232)
    uint64_t maxCellCount = 0;
    ioCheck(hrt_io_read_metadata_cache(cacheDir.c_str(), &maxCellCount));
    std::vector<float> ramp(3 * std::max<uint64_t>(1, maxCellCount));
    ioCheck(hrt_io_bake_color_ramp(cfg.particle_material_preset, maxCellCount, ramp.data()));
    RendererMaterial materials;
    for (uint64_t i = 0; i < cfg.n_roughs; ++i) materials.roughs.push_back({cfg.roughs[3 * i], cfg.roughs[3 * i + 1], cfg.roughs[3 * i + 2]});
    for (uint64_t i = 0; i < cfg.n_metals; ++i) materials.metals.push_back({{cfg.metals[4 * i], cfg.metals[4 * i + 1], cfg.metals[4 * i + 2]}, cfg.metals[4 * i + 3]});
    const size_t materialOffset = materials.roughs.size();
    for (uint64_t i = 0; i < maxCellCount; ++i) materials.roughs.push_back({ramp[3 * i], ramp[3 * i + 1], ramp[3 * i + 2]});
    const std::vector<HitGroupSbtRecord> addGeoRecord = createAddSphereTriangleSBTRecord(ctx, addSpheres, {}, materials);
    for (auto &fd : perFile) {                      // per file: the records of its instances, extra geometry first (:262-282)
        std::vector<std::pair<size_t, HrtFloat3 *>> particleSBTData;
        for (size_t p = 0; p < fd.particleCount; ++p) {
            if (fd.ids[p] >= maxCellCount) { std::fprintf(stderr, "particle id %llu beyond metadata.cache's cell count %llu\n", (unsigned long long)fd.ids[p], (unsigned long long)maxCellCount); return -1; }
            particleSBTData.emplace_back(fd.ids[p] + materialOffset, fd.dev_normals[p]);
        }
        fd.records = createVTKParticleSBTRecord(ctx, particleSBTData, materials);
        fd.records.insert(fd.records.begin(), addGeoRecord.begin(), addGeoRecord.end());
    }
    createMissSBTRecord(ctx, {0.7f, 0.8f, 0.9f});
    size_t totalTriangles = 0; for (auto &fd : perFile) totalTriangles += fd.triangleCount;
    std::printf("%llu files loaded in %.1f ms (%zu loader threads): %zu triangles in all, %llu ramp colours\n", (unsigned long long)series.n, loadMs,
                std::min<size_t>(std::thread::hardware_concurrency(), maxThreads), totalTriangles, (unsigned long long)maxCellCount);

    HrtRngState *dev_stateArray = nullptr;
    RandomGenerator::initDeviceRandomGenerators(ctx, dev_stateArray, W, H, 0x5EED0000C0FFEEull);
    const auto camera = SDL_GraphicsWindowConfigureCamera({cfg.camera_center[0], cfg.camera_center[1], cfg.camera_center[2]},
                                                          {cfg.camera_target[0], cfg.camera_target[1], cfg.camera_target[2]},
                                                          {cfg.up_direction[0], cfg.up_direction[1], cfg.up_direction[2]}, cfg.api_is_opengl != 0);
    HrtFloat4 *color = nullptr; HrtUchar4 *rgba = nullptr;
    hipCheck(hipMalloc((void **)&color, sizeof(HrtFloat4) * (size_t)W * H));
    hipCheck(hipMalloc((void **)&rgba, sizeof(HrtUchar4) * (size_t)W * H));
    HrtRayGenParams raygen{};
    raygen.width = W; raygen.height = H; raygen.colorBuffer = color;
    raygen.cameraCenter = camera.cameraCenter; raygen.cameraU = camera.cameraU; raygen.cameraV = camera.cameraV; raygen.cameraW = camera.cameraW;

    long frames = 0;
    hrtCheckError(ctx, hrt_stats_reset(ctx));
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t f = 0; f < series.n && (maxFrames < 0 || frames < maxFrames); ++f) {     // (the reference loops the animation; one pass here)
        FileData &fd = perFile[f];
        hrtCheckError(ctx, hrt_materials_set(ctx, fd.records.data(), (uint32_t)fd.records.size()));
        const size_t frameCountPerFile = (size_t)(series.durations[f] * (float)(cfg.fps * cfg.render_speed_ratio));   // :366-367
        HrtPoseParams pose{};
        pose.duration = series.durations[f]; pose.frame_count = (uint32_t)frameCountPerFile; pose.mesh_mode = 1;
        std::memcpy(&pose.particle_offset, cfg.particle_shift, 12); std::memcpy(&pose.particle_scale, cfg.particle_scale, 12);
        for (size_t frame = 0; frame < frameCountPerFile && (maxFrames < 0 || frames < maxFrames); ++frame, ++frames) {
            pose.frame = (uint32_t)frame;
            hrtCheckError(ctx, hrt_pose_instances(ctx, fd.dev_instances, (uint32_t)addGeoCount, (uint32_t)fd.particleCount, fd.dev_states, fd.dev_states, &pose, nullptr));
            updateIAS(ctx, fd.ias, fd.dev_instances, fd.instanceCount);
            const HrtGlobalParams params{std::get<0>(fd.ias), dev_stateArray};
            hrtCheckError(ctx, hrt_render_launch(ctx, &params, &raygen, 1, nullptr, nullptr));      // (launch, conversion, then the frame's one synchronisation)
            hrtCheckError(ctx, hrt_to_rgba8(ctx, color, rgba, W, H, nullptr));
            hrtCheckError(ctx, hrt_sync(ctx, nullptr));
        }
    }
    hipCheck(hipDeviceSynchronize());
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    HrtStats st{};
    hrtCheckError(ctx, hrt_stats_get(ctx, &st));
    for (int k = 0; k < HRT_K_COUNT; ++k)
        if (st.kernel_launches[k] && st.kernel_ms[k] > 0.0) std::printf("  kernel class %d: %.3f ms in %llu launches\n", k, st.kernel_ms[k], (unsigned long long)st.kernel_launches[k]);
    std::printf("%ld frames %ux%u: %.3f ms/frame (%.0f frames/s), %.1f Mrays/s, refits %llu rebuilds %llu\n", frames, W, H, ms / std::max(1l, frames),
                frames / ms * 1e3, st.rays / ms * 1e-3, (unsigned long long)st.tlas_refits, (unsigned long long)st.tlas_rebuilds);

    std::vector<HrtUchar4> host((size_t)W * H);
    hipCheck(hipMemcpy(host.data(), rgba, host.size() * sizeof(HrtUchar4), hipMemcpyDeviceToHost));
    if (FILE *fp = std::fopen(out.c_str(), "wb")) {
        std::fprintf(fp, "P6\n%u %u\n255\n", W, H);
        for (uint32_t y = 0; y < H; ++y) for (uint32_t x = 0; x < W; ++x) std::fwrite(&host[(size_t)y * W + x], 1, 3, fp);
        std::fclose(fp);
    }

    RandomGenerator::freeDeviceRandomGenerators(ctx, dev_stateArray);
    for (auto &fd : perFile) {
        cleanupAccelerationStructure(ctx, fd.ias);
        for (size_t i = addGeoCount; i < fd.gas.size(); ++i) cleanupAccelerationStructure(ctx, fd.gas[i]);
        for (auto *p : fd.dev_normals) hipCheck(hipFree(p));
        hipCheck(hipFree(fd.dev_instances)); hipCheck(hipFree(fd.dev_states));
    }
    for (auto &g : addGAS) cleanupAccelerationStructure(ctx, g);
    for (auto *p : sphereCenters) hipCheck(hipFree(p));
    for (auto *p : sphereRadii) hipCheck(hipFree(p));
    hrt_io_free_series(&series); hrt_io_free_config(&cfg);
    hipCheck(hipFree(color)); hipCheck(hipFree(rgba));
    destroyContext(ctx);
    return 0;
}
