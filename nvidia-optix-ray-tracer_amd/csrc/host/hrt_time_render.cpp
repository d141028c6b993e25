// hrt_time_render.cpp -- the reference's Time-mode program flow, headless, in the reference's own language:
// main() (src/Global/Main.cu:12-50) -> RendererTime::commitRendererData (src/Global/RendererTime.cu:160-290)
// -> RendererTime::startRender's frame loop (:373-520) with the window, camera input and denoiser left out.
//   hrt_time_render <config.json> [exe_dir] [max_frames=all] [out.ppm] [width height]
// exe_dir is the directory the config's relative paths are relative to (the reference runs from bin/).
// Per frame: hrt_pose_instances -> updateIAS -> launch + sync -> convert to 8 bit; the last frame is written as PPM.
#include "renderer_host.hpp"
#include "hrt_io.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <dirent.h>
#include <string>

using namespace project;

#define hipCheck(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(-100); } } while (0)
#define ioCheck(x) do { if ((x) != 0) { std::fprintf(stderr, "%s\n", hrt_io_last_error()); std::exit(-1); } } while (0)   // VTK_READER_ERROR_EXIT_CODE

static std::string join(const std::string &base, const std::string &p) { return (!p.empty() && p[0] == '/') ? p : base + "/" + p; }

template <typename T> static T *toDevice(const T *host, size_t count) {
    T *dev = nullptr;
    hipCheck(hipMalloc((void **)&dev, std::max<size_t>(1, count) * sizeof(T)));
    if (count) hipCheck(hipMemcpy(dev, host, count * sizeof(T), hipMemcpyHostToDevice));
    return dev;
}

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s <config.json> [exe_dir] [max_frames] [out.ppm] [width height]\n", argv[0]); return 2; }
    const std::string configPath = argv[1];
    const std::string exeDir = argc > 2 ? argv[2] : ".";
    const long maxFrames = argc > 3 ? std::atol(argv[3]) : -1;
    const std::string out = argc > 4 ? argv[4] : "hrt_time_render.ppm";

    HrtIoConfig cfg;
    ioCheck(hrt_io_load_config(configPath.c_str(), &cfg));
    if (cfg.mesh) { std::fprintf(stderr, "config selects Mesh mode; this driver plays Time mode\n"); return 2; }
    const uint32_t W = argc > 6 ? (uint32_t)std::atoi(argv[5]) : (uint32_t)cfg.window_width;
    const uint32_t H = argc > 6 ? (uint32_t)std::atoi(argv[6]) : (uint32_t)cfg.window_height;

    HrtContext *ctx = createContext(0, false);
    // the frame loop below only moves instances (same shapes, same records): updates may be asynchronous -- pose kernel, updateIAS and
    // the launch are then enqueued back to back with no read-back in between (HRT_TIME_RENDER_SYNC_UPDATE=1: the synchronous update)
    const uint32_t flags = (std::getenv("HRT_TIME_RENDER_KERNEL_TIMES") ? HRT_CTX_TIMING : 0u) | (std::getenv("HRT_TIME_RENDER_SYNC_UPDATE") ? 0u : HRT_CTX_ASYNC_UPDATE);
    hrtCheckError(ctx, hrt_ctx_set_flags(ctx, flags));

    // extra geometry first (buildAddDataGAS, RendererTime.cu:73-84), then one GAS per STL shape in name order (:183-190)
    std::vector<GAS> gasAll;
    std::vector<HrtFloat3 *> sphereCenters; std::vector<float *> sphereRadii;
    for (uint64_t i = 0; i < cfg.n_spheres; ++i) {
        const HrtFloat3 c{cfg.spheres[i].center[0], cfg.spheres[i].center[1], cfg.spheres[i].center[2]};
        RendererSphere s{cfg.spheres[i].metal, (size_t)cfg.spheres[i].material_index, toDevice(&c, 1), toDevice(&cfg.spheres[i].radius, 1), 1};
        sphereCenters.push_back(s.dev_centers); sphereRadii.push_back(s.dev_radii);
        gasAll.push_back(buildGASForSpheres(ctx, s));
    }
    const size_t addGeoCount = gasAll.size();
    std::vector<std::string> stlNames;
    const std::string stlDir = join(exeDir, cfg.stl_path);
    if (DIR *d = opendir(stlDir.c_str())) {
        while (dirent *e = readdir(d)) { const std::string n = e->d_name; if (n.size() > 4 && n.substr(n.size() - 4) == ".stl") stlNames.push_back(n); }
        closedir(d);
    } else { std::fprintf(stderr, "cannot open STL directory %s\n", stlDir.c_str()); return -1; }
    std::sort(stlNames.begin(), stlNames.end());
    std::vector<HrtFloat3 *> shapeNormals;
    for (const auto &n : stlNames) {
        HrtIoMesh m;
        ioCheck(hrt_io_read_stl((stlDir + "/" + n).c_str(), &m));
        RendererTriangle t{0, 0, toDevice(reinterpret_cast<HrtFloat3 *>(m.vertices), 3 * m.n_triangles),
                           toDevice(reinterpret_cast<HrtFloat3 *>(m.normals), 3 * m.n_triangles), (size_t)m.n_triangles};
        gasAll.push_back(buildGASForTriangles(ctx, t));
        hipCheck(hipFree(t.dev_vertices));
        shapeNormals.push_back(t.dev_normals);
        hrt_io_free_mesh(&m);
    }

    HrtIoSeries series;
    ioCheck(hrt_io_read_series((join(exeDir, cfg.series_path)).c_str(), cfg.series_name, &series));
    std::vector<HrtIoParticles> files(series.n);
    uint64_t maxParticles = 0;
    for (uint64_t f = 0; f < series.n; ++f) { ioCheck(hrt_io_read_particle_vtk(series.files[f], &files[f])); maxParticles = std::max(maxParticles, files[f].n); }
    std::printf("%llu extra spheres, %zu shapes, %llu files, %llu particles\n", (unsigned long long)cfg.n_spheres, stlNames.size(),
                (unsigned long long)series.n, (unsigned long long)maxParticles);

    // materials: config roughs, then the baked ramp, one colour per particle id (RendererTime.cu:246-256)
    std::vector<float> ramp(3 * maxParticles);
    ioCheck(hrt_io_bake_color_ramp(cfg.particle_material_preset, maxParticles, ramp.data()));
    RendererMaterial materials;
    for (uint64_t i = 0; i < cfg.n_roughs; ++i) materials.roughs.push_back({cfg.roughs[3 * i], cfg.roughs[3 * i + 1], cfg.roughs[3 * i + 2]});
    for (uint64_t i = 0; i < cfg.n_metals; ++i) materials.metals.push_back({{cfg.metals[4 * i], cfg.metals[4 * i + 1], cfg.metals[4 * i + 2]}, cfg.metals[4 * i + 3]});
    const size_t materialOffset = materials.roughs.size();
    for (uint64_t i = 0; i < maxParticles; ++i) materials.roughs.push_back({ramp[3 * i], ramp[3 * i + 1], ramp[3 * i + 2]});
    std::vector<RendererSphere> addSpheres;
    for (uint64_t i = 0; i < cfg.n_spheres; ++i)
        addSpheres.push_back({cfg.spheres[i].metal ? METAL : ROUGH, (size_t)cfg.spheres[i].material_index, sphereCenters[i], sphereRadii[i], 1});
    const std::vector<HitGroupSbtRecord> addGeoRecord = createAddSphereTriangleSBTRecord(ctx, addSpheres, {}, materials);

    // per file: instances (identity until the first frame), IAS, SBT records (:96-140, :258-288), particle states on the device
    struct FileData { HrtInstance *dev_instances; IAS ias; std::vector<HrtSbtRecord> records; HrtParticleState *dev_states; size_t instanceCount; };
    std::vector<FileData> perFile(series.n);
    const float identity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (uint64_t f = 0; f < series.n; ++f) {
        const HrtIoParticles &p = files[f];
        const size_t count = addGeoCount + p.n;
        std::vector<HrtInstance> inst(count);
        std::vector<std::pair<size_t, HrtFloat3 *>> particleSBTData;
        for (size_t i = 0; i < count; ++i) {
            std::memset(&inst[i], 0, sizeof inst[i]);
            std::memcpy(inst[i].transform, i < addGeoCount ? cfg.spheres[i].transform : identity, sizeof identity);
            inst[i].sbtOffset = (uint32_t)i; inst[i].visibilityMask = 1;
            if (i < addGeoCount) inst[i].traversableHandle = gasAll[i].first;
            else {
                const size_t k = i - addGeoCount;
                inst[i].traversableHandle = gasAll[addGeoCount + p.shape_ids[k]].first;
                particleSBTData.emplace_back(p.ids[k] + materialOffset, shapeNormals[p.shape_ids[k]]);
            }
        }
        std::vector<HitGroupSbtRecord> rec = createVTKParticleSBTRecord(ctx, particleSBTData, materials);
        rec.insert(rec.begin(), addGeoRecord.begin(), addGeoRecord.end());
        perFile[f].dev_instances = toDevice(inst.data(), count);
        perFile[f].ias = buildIAS(ctx, perFile[f].dev_instances, count);
        perFile[f].records = std::move(rec);
        perFile[f].dev_states = toDevice(p.states, p.n);
        perFile[f].instanceCount = count;
    }
    createMissSBTRecord(ctx, {0.7f, 0.8f, 0.9f});

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

    {   // one untimed launch: the first one pays for loading the code objects
        hrtCheckError(ctx, hrt_materials_set(ctx, perFile[0].records.data(), (uint32_t)perFile[0].records.size()));
        const HrtGlobalParams params{std::get<0>(perFile[0].ias), dev_stateArray};
        launch(ctx, params, raygen, 1);
        hrtCheckError(ctx, hrt_to_rgba8(ctx, color, rgba, W, H, nullptr));
    }
    long frames = 0;
    hrtCheckError(ctx, hrt_stats_reset(ctx));
    // HRT_TIME_RENDER_BREAKDOWN=1: where a frame's time goes -- every step bracketed by device synchronisations (so the frames are slower
    // than in a normal run, where the steps are only enqueued), summed separately for the first frame of a file (materials, the update
    // that turns the identity-built IAS into the posed scene) and the others
    const bool breakdown = std::getenv("HRT_TIME_RENDER_BREAKDOWN") != nullptr;
    enum { kMaterials, kPose, kUpdate, kLaunch, kRgba, kSteps };
    const char *stepName[kSteps] = {"hrt_materials_set", "hrt_pose_instances", "hrt_tlas_update", "hrt_render_launch", "hrt_to_rgba8 + sync"};
    double stepMs[2][kSteps] = {{0}}; long stepFrames[2] = {0, 0};
    auto timed = [&](int first, int step, auto &&fn) {
        if (!breakdown) { fn(); return; }
        hipCheck(hipDeviceSynchronize());
        const auto a = std::chrono::steady_clock::now();
        fn();
        hipCheck(hipDeviceSynchronize());
        stepMs[first][step] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    };
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t f = 0; f < series.n && (maxFrames < 0 || frames < maxFrames); ++f) {
        FileData &fd = perFile[f];
        timed(1, kMaterials, [&] { hrtCheckError(ctx, hrt_materials_set(ctx, fd.records.data(), (uint32_t)fd.records.size())); });
        const uint64_t next = f + 1 < series.n ? f + 1 : f;                                   // :443-447
        const size_t frameCountThisFile = (size_t)(series.durations[f] * (float)(cfg.fps * cfg.render_speed_ratio));   // :427-428
        HrtPoseParams pose{};
        pose.duration = series.durations[f]; pose.frame_count = (uint32_t)frameCountThisFile;
        std::memcpy(&pose.particle_offset, cfg.particle_shift, 12); std::memcpy(&pose.particle_scale, cfg.particle_scale, 12);
        for (size_t frame = 0; frame < frameCountThisFile && (maxFrames < 0 || frames < maxFrames); ++frame, ++frames) {
            pose.frame = (uint32_t)frame;
            const int first = frame == 0 ? 1 : 0;
            ++stepFrames[first];
            timed(first, kPose, [&] { hrtCheckError(ctx, hrt_pose_instances(ctx, fd.dev_instances, (uint32_t)addGeoCount, (uint32_t)files[f].n, fd.dev_states,
                                                                            perFile[next].dev_states, &pose, nullptr)); });
            timed(first, kUpdate, [&] { updateIAS(ctx, fd.ias, fd.dev_instances, fd.instanceCount); });
            const HrtGlobalParams params{std::get<0>(fd.ias), dev_stateArray};
            // (launch, conversion, then the frame's one synchronisation -- the reference synchronises between the two, RendererTime.cu:497-515,
            // because its denoiser runs on the host's schedule; nothing here needs the frame before the bytes exist)
            timed(first, kLaunch, [&] { hrtCheckError(ctx, hrt_render_launch(ctx, &params, &raygen, 1, nullptr, nullptr)); });
            timed(first, kRgba, [&] { hrtCheckError(ctx, hrt_to_rgba8(ctx, color, rgba, W, H, nullptr)); hrtCheckError(ctx, hrt_sync(ctx, nullptr)); });
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

    if (breakdown)
        for (int first = 1; first >= 0; --first) {
            std::printf("  %s (%ld): ", first ? "first frame of a file" : "other frames", stepFrames[first]);
            for (int k = 0; k < kSteps; ++k) std::printf("%s %.3f ms%s", stepName[k], stepMs[first][k] / std::max(1l, stepFrames[first]), k + 1 < kSteps ? ", " : "\n");
        }

    std::vector<HrtUchar4> host((size_t)W * H);
    hipCheck(hipMemcpy(host.data(), rgba, host.size() * sizeof(HrtUchar4), hipMemcpyDeviceToHost));
    if (FILE *fp = std::fopen(out.c_str(), "wb")) {
        std::fprintf(fp, "P6\n%u %u\n255\n", W, H);
        for (uint32_t y = 0; y < H; ++y) for (uint32_t x = 0; x < W; ++x) std::fwrite(&host[(size_t)y * W + x], 1, 3, fp);
        std::fclose(fp);
    }

    RandomGenerator::freeDeviceRandomGenerators(ctx, dev_stateArray);
    for (auto &fd : perFile) { cleanupAccelerationStructure(ctx, fd.ias); hipCheck(hipFree(fd.dev_instances)); hipCheck(hipFree(fd.dev_states)); }
    for (auto &g : gasAll) cleanupAccelerationStructure(ctx, g);
    for (auto *p : shapeNormals) hipCheck(hipFree(p));
    for (auto *p : sphereCenters) hipCheck(hipFree(p));
    for (auto *p : sphereRadii) hipCheck(hipFree(p));
    for (auto &p : files) hrt_io_free_particles(&p);
    hrt_io_free_series(&series); hrt_io_free_config(&cfg);
    hipCheck(hipFree(color)); hipCheck(hipFree(rgba));
    destroyContext(ctx);
    return 0;
}
