// hrt_api.cpp -- C ABI of libhrt.so: context, acceleration structures, materials, RNG, launch.
// Entry points and the reference call sites they replace are documented in include/hrt.h.
// There is deliberately no CPU fallback anywhere in this file: every compute path ends in a
// HIP kernel launch, and context creation fails when no gfx950 device is present.
#include "hrt_internal.hpp"

namespace hrt {

// Errors: *_build calls may run on several loader threads of one context (RendererMesh.cu:93-100), so a thread's last failure
// is kept per thread; hrt_last_error answers with the calling thread's own failure on that context when it has one, else with
// the context's most recent failure from any thread (copied under a lock).
namespace {
thread_local std::string g_create_error, g_thread_error, g_error_copy;
thread_local const HrtContext *g_thread_error_ctx = nullptr;
std::mutex g_error_mu;
}
const char *create_error() { return g_create_error.c_str(); }

int fail(HrtContext *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (ctx) {
        g_thread_error = buf; g_thread_error_ctx = ctx;
        std::lock_guard<std::mutex> lk(g_error_mu);
        ctx->error = buf;
    } else g_create_error = buf;
    return code;
}
const char *last_error_of(const HrtContext *ctx) {
    if (g_thread_error_ctx == ctx) return g_thread_error.c_str();
    std::lock_guard<std::mutex> lk(g_error_mu);
    g_error_copy = ctx->error;
    return g_error_copy.c_str();
}

// ---- XORWOW sub-sequence jump matrices: T^(2^(67+k)), k = 0..31, 160 columns x 5 words ----
struct Gf2m { uint32_t col[160][5]; };
void gf2_apply(const Gf2m &m, const uint32_t in[5], uint32_t out[5]) {
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < 5; ++w)
        for (int b = 0; b < 32; ++b)
            if ((in[w] >> b) & 1u)
                for (int k = 0; k < 5; ++k) r[k] ^= m.col[w * 32 + b][k];
    std::memcpy(out, r, sizeof r);
}
void gf2_square(Gf2m &m) {
    Gf2m t;
    for (int i = 0; i < 160; ++i) gf2_apply(m, m.col[i], t.col[i]);
    m = t;
}
std::vector<uint32_t> make_jump_tables() {
    Gf2m t;
    for (int i = 0; i < 160; ++i) {
        uint32_t v[5] = {0, 0, 0, 0, 0};
        v[i / 32] = 1u << (i % 32);
        const uint32_t x = v[0] ^ (v[0] >> 2);           // one xorshift step of XORWOW
        const uint32_t n4 = (v[4] ^ (v[4] << 4)) ^ (x ^ (x << 1));
        t.col[i][0] = v[1]; t.col[i][1] = v[2]; t.col[i][2] = v[3]; t.col[i][3] = v[4]; t.col[i][4] = n4;
    }
    for (int k = 0; k < 67; ++k) gf2_square(t);
    std::vector<uint32_t> out(32 * 800);
    for (int k = 0; k < 32; ++k) {
        std::memcpy(&out[(size_t)k * 800], t.col, sizeof t.col);
        gf2_square(t);
    }
    return out;
}

int ensure_workspace(HrtContext *ctx, uint32_t n, uint32_t height) {
    Workspace &w = ctx->ws;
    if (n > w.capacity) {
        for (SampleSet &st : w.set) {
            void *ptrs[] = {st.rays[0], st.rays[1], st.hit_tuvp, st.hit_inst, st.bin_items, st.chain, st.result};
            for (void *p : ptrs) if (p) (void)hipFree(p);
            st = SampleSet{{nullptr, nullptr}, nullptr, nullptr, nullptr, nullptr, nullptr, st.stages};
        }
        if (w.accum) (void)hipFree(w.accum);
        w.accum = nullptr; w.capacity = 0;
        for (SampleSet &st : w.set) {
            HIP_TRY(ctx, hipMalloc((void **)&st.rays[0], sizeof(RayRec) * (size_t)n));
            HIP_TRY(ctx, hipMalloc((void **)&st.rays[1], sizeof(RayRec) * (size_t)n));
            HIP_TRY(ctx, hipMalloc((void **)&st.hit_tuvp, sizeof(float4) * (size_t)n));
            HIP_TRY(ctx, hipMalloc((void **)&st.hit_inst, sizeof(uint32_t) * (size_t)n));
            HIP_TRY(ctx, hipMalloc((void **)&st.bin_items, sizeof(uint32_t) * (size_t)n * kNumBins));
            HIP_TRY(ctx, hipMalloc((void **)&st.chain, sizeof(uint32_t) * 4 * (size_t)n));
            HIP_TRY(ctx, hipMalloc((void **)&st.result, sizeof(float4) * (size_t)n));
        }
        HIP_TRY(ctx, hipMalloc((void **)&w.accum, sizeof(float4) * (size_t)n));
        w.capacity = n;
    }
    if (height > w.rows_capacity) {
        if (w.rows) (void)hipFree(w.rows);
        w.rows_capacity = 0;
        HIP_TRY(ctx, hipMalloc((void **)&w.rows, sizeof(uint32_t) * (size_t)height));
        w.rows_capacity = height;
    }
    for (SampleSet &st : w.set)
        if (!st.stages) { HIP_TRY(ctx, hipMalloc((void **)&st.stages, sizeof(StageCounters) * (kRayTraceDepth + 1) * kMaxSubTiles)); ctx->fused_counters_clean = false; }
    return HRT_OK;
}

void drain_spans(HrtContext *ctx) {
    for (const TimedSpan &sp : ctx->spans) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) { ctx->kernel_ms[sp.kind] += ms; ctx->kernel_launches[sp.kind]++; }
    }
    ctx->spans.clear();
    ctx->events_used = 0;
}

}  // namespace hrt


// ======================================================================================
extern "C" {

const char *hrt_version(void) { return "hrt 0.1 (gfx950 wavefront path tracer)"; }

const char *hrt_last_error(const HrtContext *ctx) { return ctx ? hrt::last_error_of(ctx) : hrt::create_error(); }

int hrt_ctx_create(int device_id, uint32_t flags, HrtContext **out_ctx) {
    if (!out_ctx) return fail(nullptr, HRT_ERR_INVALID, "out_ctx is NULL");
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(nullptr, HRT_ERR_NO_DEVICE, "no HIP device visible: libhrt has no CPU path");
    if (device_id < 0 || device_id >= n) return fail(nullptr, HRT_ERR_INVALID, "device %d out of range (%d devices)", device_id, n);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return fail(nullptr, HRT_ERR_HIP, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, HRT_ERR_NO_DEVICE, "device %d is %s; libhrt carries gfx950 code only", device_id, prop.gcnArchName);
    if (hipSetDevice(device_id) != hipSuccess) return fail(nullptr, HRT_ERR_HIP, "hipSetDevice(%d) failed", device_id);
    std::unique_ptr<HrtContext> ctx(new HrtContext());
    ctx->device = device_id; ctx->flags = flags; ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const std::vector<uint32_t> jump = make_jump_tables();
    if (hipMalloc((void **)&ctx->d_jump, jump.size() * sizeof(uint32_t)) != hipSuccess ||
        hipMemcpy(ctx->d_jump, jump.data(), jump.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc((void **)&ctx->d_stats, sizeof(DeviceStats)) != hipSuccess ||
        hipMemset(ctx->d_stats, 0, sizeof(DeviceStats)) != hipSuccess)
        return fail(nullptr, HRT_ERR_HIP, "context allocation failed: %s", hipGetErrorString(hipGetLastError()));
    // the tuning knobs: one list (knobs.def), parsed here, documented from there (tools/knob_table.py)
#define HRT_KNOB(NAME, DEFAULT, DOC, ...) if (const char *e = std::getenv(NAME)) { __VA_ARGS__; }
#define HRT_HOST_KNOB(NAME, DEFAULT, DOC)
#include "knobs.def"
#undef HRT_KNOB
#undef HRT_HOST_KNOB
    *out_ctx = ctx.release();
    return HRT_OK;
}

int hrt_ctx_set_flags(HrtContext *ctx, uint32_t flags) {
    if (!ctx) return HRT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    HIP_TRY(ctx, hipDeviceSynchronize());
    drain_spans(ctx);
    ctx->flags = flags;
    return HRT_OK;
}

int hrt_ctx_destroy(HrtContext *ctx) {
    if (!ctx) return HRT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto &kv : ctx->tlas) { free_tlas_device(ctx, *kv.second); free_tlas_host(*kv.second); }
    pool_drain(ctx);
    Workspace &w = ctx->ws;
    for (SampleSet &st : w.set) {
        void *sp[] = {st.rays[0], st.rays[1], st.hit_tuvp, st.hit_inst, st.bin_items, st.chain, st.result, st.stages};
        for (void *p : sp) if (p) (void)hipFree(p);
    }
    if (ctx->pin_stage) (void)hipHostFree(ctx->pin_stage);
    void *ptrs[] = {w.accum, w.slice_cost, w.slice_order, w.primary_cache, w.rows, ctx->d_jump, ctx->d_stats, ctx->d_hitgroups, ctx->d_inst_program};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (const ScratchArena &a : ctx->scratch_free) (void)hipFree(a.p);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->sub_done) (void)hipEventDestroy(e);
    for (hipStream_t st : ctx->sub_streams) (void)hipStreamDestroy(st);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_graph_done) (void)hipEventDestroy(ctx->ev_graph_done);
    if (ctx->graph_stream) (void)hipStreamDestroy(ctx->graph_stream);
    delete ctx;
    return HRT_OK;
}

// ---- SBT ------------------------------------------------------------------------------
int hrt_sbt_record_pack_header(HrtProgram program, void *record_header) {
    if (!record_header || (int)program < 0 || (int)program >= (int)HRT_PROGRAM_COUNT) return HRT_ERR_INVALID;
    unsigned char *h = (unsigned char *)record_header;
    std::memset(h, 0, HRT_SBT_RECORD_HEADER_SIZE);
    h[0] = 'H'; h[1] = 'R'; h[2] = 'T'; h[3] = (unsigned char)program;
    return HRT_OK;
}

int hrt_materials_set(HrtContext *ctx, const HrtSbtRecord *h_records, uint32_t n_records) {
    if (!ctx || (n_records && !h_records)) return HRT_ERR_INVALID;
    for (uint32_t i = 0; i < n_records; ++i) {
        const unsigned char *h = h_records[i].header;
        if (h[0] != 'H' || h[1] != 'R' || h[2] != 'T' || h[3] >= HRT_PROGRAM_COUNT)
            return fail(ctx, HRT_ERR_INVALID, "SBT record %u: header not packed by hrt_sbt_record_pack_header", i);
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->records.assign(h_records, h_records + n_records);
    ctx->have_records = true;
    ctx->materials_generation++;
    return HRT_OK;
}

int hrt_miss_set(HrtContext *ctx, const HrtMissParams *h_miss) {
    if (!ctx || !h_miss) return HRT_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->miss = *h_miss;
    return HRT_OK;
}

// ---- RNG ------------------------------------------------------------------------------
int hrt_rng_init(HrtContext *ctx, uint32_t width, uint32_t height, uint64_t seed_salt, void *stream, HrtRngState **out_d_states) {
    if (!ctx || !out_d_states) return HRT_ERR_INVALID;
    const uint64_t n64 = (uint64_t)width * height;
    if (n64 == 0 || n64 > 0xffffffffull) return fail(ctx, HRT_ERR_INVALID, "frame %ux%u is empty or exceeds 2^32 pixels", width, height);
    (void)hipSetDevice(ctx->device);
    RngState *d = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d, sizeof(RngState) * n64));
    launch_rng_init(d, (uint32_t)n64, seed_salt, ctx->d_jump, (hipStream_t)stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));      // reference: cudaStreamSynchronize, HostFunctions.cu:135
    *out_d_states = reinterpret_cast<HrtRngState *>(d);
    return HRT_OK;
}

int hrt_rng_free(HrtContext *ctx, HrtRngState *d_states, void *stream) {
    if (!ctx) return HRT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(ctx, hipFree(d_states));
    return HRT_OK;
}

// ---- the launch -----------------------------------------------------------------------
// k_fused (fused.hip) keeps one sibling group per tree level in LDS without an overflow path and addresses the records with
// 32-bit byte offsets; a tree outside either limit takes round 1's fused kernel (kernels.hip), which has neither.
static bool fits_fused_kernel(const HrtContext *ctx, const Tlas &t) {
    return t.max_depth <= (uint32_t)ctx->fused_max_depth && (uint64_t)t.n_nodes * t.node_stride < ctx->fused_max_bytes &&
           (uint64_t)std::max(t.n_prims, 1u) * t.prim_stride < ctx->fused_max_bytes;
}
static int refresh_tables(HrtContext *ctx, uint64_t handle, Tlas *t, hipStream_t s) {
    if (ctx->table_tlas == handle && ctx->table_tlas_gen == t->generation && ctx->table_mat_gen == ctx->materials_generation) return HRT_OK;
    const uint32_t n = std::max(t->n_instances, 1u);
    std::vector<HitGroup> hg(n); std::vector<uint32_t> prog(n, 0);
    for (auto &p : ctx->program_present) p = false;
    for (uint32_t i = 0; i < t->n_instances; ++i) {
        const uint32_t off = t->sbt_offset[i];
        if (off >= ctx->records.size()) return fail(ctx, HRT_ERR_STATE, "instance %u: sbtOffset %u has no SBT record (%zu set)", i, off, ctx->records.size());
        const HrtSbtRecord &r = ctx->records[off];
        const uint32_t program = r.header[3];
        const bool sphere_prog = program == HRT_PROGRAM_SPHERE_ROUGH || program == HRT_PROGRAM_SPHERE_METAL;
        if (sphere_prog != (t->kind[i] == kPrimKindSphere))
            return fail(ctx, HRT_ERR_STATE, "instance %u: program %u does not match its geometry", i, program);
        std::memcpy(&hg[i], &r.data, sizeof(HitGroup));
        prog[i] = program;
        ctx->program_present[program] = true;
    }
    if (n > ctx->table_capacity) {
        if (ctx->d_hitgroups) { (void)hipFree(ctx->d_hitgroups); (void)hipFree(ctx->d_inst_program); }
        ctx->table_capacity = 0;
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_hitgroups, sizeof(HitGroup) * n));
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_inst_program, sizeof(uint32_t) * n));
        ctx->table_capacity = n;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_hitgroups, hg.data(), sizeof(HitGroup) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_inst_program, prog.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    ctx->table_tlas = handle; ctx->table_tlas_gen = t->generation; ctx->table_mat_gen = ctx->materials_generation;
    return HRT_OK;
}

// What both execution modes of a launch work from (hrt_render_launch has checked the arguments, found the tree, refreshed the material
// tables and uploaded the tile's row list).
struct LaunchFrame {
    HrtContext *ctx; Tlas *t; const HrtGlobalParams *h_params; const HrtRayGenParams *rg;
    uint32_t spp, n;               // samples per pixel, pixels of the tile
    hipStream_t s;
};

// ---- fused path mode: ONE launch, every lane owns a pixel and runs all its samples (generate, traverse, shade, accumulate in place).
//      No stage barriers.  The default (HRT_FUSED=1); HRT_FUSED=2 and trees outside k_fused's limits: round 1's path kernel. ----
static int render_fused(const LaunchFrame &f) {
    HrtContext *ctx = f.ctx; Tlas *t = f.t; const HrtGlobalParams *h_params = f.h_params; const HrtRayGenParams *rg = f.rg;
    const uint32_t spp = f.spp, n = f.n; hipStream_t s = f.s;
    Workspace &w = ctx->ws;
    StageCounters *stg = w.set[0].stages;
    TraverseArgs ta{};
    ta.nodes = t->d_nodes; ta.prims = t->d_prims; ta.node_stride = t->node_stride; ta.prim_stride = t->prim_stride;
    ta.fetch_counter = stg[0].fetch;
    ta.inst_inv = t->d_inst_inv; ta.inst_identity = t->d_inst_identity;
    ta.tmin = kFloatZero; ta.tmax = kFloatInfinity;
    ta.refill_threshold = ctx->fused_refill_threshold; ta.postpone_pct = ctx->postpone_pct; ta.leaf_quorum = ctx->leaf_quorum; ta.tail_regen = ctx->fused_tail_regen; ta.tail_split = ctx->tail_split; ta.fetch_chunk = (uint32_t)ctx->fused_fetch_chunk;
    // How many leaf groups a lane may queue before its node work waits for primitive tests.  A queued group is tested whatever has been hit
    // meanwhile, so the more primitive tests a ray makes per node visit the sooner they should follow their node: soups (7 tests per 21 visits)
    // run the stack full, scenes of many bodies (19 per 22 in a dense cloud) want 2, two-level trees (30 per 28, all of them inside instances) 1 --
    // profiles/r04_leaf_hold.txt: +2 % / +7 %, C4 unchanged; the shipped sample's 25 particles (1.4 tests per 5.4 visits) are a soup in this respect.
    ta.leaf_hold = ctx->leaf_hold > 0 ? ctx->leaf_hold : t->two_level ? 1 : (t->scene_of_bodies && t->n_instances >= 256u) ? 2 : 4;
    if (ctx->refill_auto && t->two_level) ta.refill_threshold = 12;      // (a two-level ray is long and its lanes finish far apart: 12 against 20 is +1.4-2.3 %, flattened trees lose 4 %)
    PathArgs &pa = ta.path;
    pa.rows = w.rows; pa.first_pixel = 0; pa.n_tile_pixels = n; pa.width = rg->width; pa.height = rg->height; pa.spp = spp;
    std::memcpy(pa.center, &rg->cameraCenter, 12); std::memcpy(pa.U, &rg->cameraU, 12);
    std::memcpy(pa.V, &rg->cameraV, 12); std::memcpy(pa.W, &rg->cameraW, 12);
    pa.bg[0] = ctx->miss.backgroundColor.x; pa.bg[1] = ctx->miss.backgroundColor.y; pa.bg[2] = ctx->miss.backgroundColor.z;
    pa.states = reinterpret_cast<RngState *>(h_params->stateArray);
    pa.hitgroups = ctx->d_hitgroups; pa.inst_program = ctx->d_inst_program; pa.accum = w.accum;
    pa.rays_closest = &ctx->d_stats->rays_closest; pa.rays_any = &ctx->d_stats->rays_any;
    // Waves per CU: a lane runs its pixel's samples one after the other, so a small tile (the multi-GPU split) ends
    // with its slowest pixels; about 1.4 pixels per lane lets the lanes that drew cheap pixels take a second one
    // while fewer waves share each SIMD (measured, profiles/r01_sweep_tile_waves.txt: 1/8 of the C4 frame takes
    // 172 ms on 12 waves per CU, 210 ms on 16).  A full frame has many pixels per lane and keeps the maximum: 16 waves
    // per CU = 4 per SIMD for k_fused (122 VGPRs, nothing spilled: 3120 Mrays/s on C4; compiled for 5 waves it spills 95
    // registers around the shading: 2560), 20 = 5 per SIMD for round 1's kernel (96 VGPRs, a few spills: 2905).
    // k_fused (fused.hip) keeps one sibling group per tree level in LDS and has no overflow path: deeper trees, and HRT_FUSED=2, take round 1's kernel
    const bool lean = ctx->fused != 2 && fits_fused_kernel(ctx, *t);
    if (t->two_level && !lean) return fail(ctx, HRT_ERR_STATE, "a two-level TLAS needs the default path kernel (HRT_FUSED=1)");
    if (lean) ta.postpone_pct = ctx->fused_postpone_pct;
    else if (ctx->fused != 2) ctx->fused_fallback_launches++;        // the tree does not fit k_fused
    uint32_t blocks_per_cu = lean ? std::min<uint32_t>((uint32_t)ctx->fused_blocks_per_cu, (uint32_t)(t->two_level ? kFusedInstancedBlocksPerCu : kFusedBlocksPerCu)) : (uint32_t)ctx->fused_blocks_per_cu;
    if (ctx->traverse_blocks_auto) {
        const uint32_t fit = (uint32_t)((10ull * n + 14ull * 64ull * (uint64_t)ctx->n_cu - 1ull) / (14ull * 64ull * (uint64_t)ctx->n_cu));
        blocks_per_cu = std::min(blocks_per_cu, std::max(fit, 4u));
    }
    const uint32_t grid = std::min<uint32_t>((uint32_t)ctx->n_cu * blocks_per_cu, (n + 63u) / 64u);
    if (t->two_level) ta.tail_split = 0;        // (the pieces of a split ray would have to carry the instance they are in)
    // HRT_CTX_REUSE_PRIMARY: the lanes' slots for their pixels' primary hits (k_fused<.., REUSE>, fused.hip)
    if (lean && spp > 1u && ((ctx->flags & HRT_CTX_REUSE_PRIMARY) != 0u || ctx->reuse_primary)) {
        if (grid * 64u > w.primary_cache_lanes) {
            if (w.primary_cache) (void)hipFree(w.primary_cache);
            w.primary_cache = nullptr; w.primary_cache_lanes = 0;
            HIP_TRY(ctx, hipMalloc((void **)&w.primary_cache, sizeof(float4) * 2u * grid * 64u));
            w.primary_cache_lanes = grid * 64u;
        }
        pa.primary_cache = w.primary_cache;
    }
    auto launch = [&]() {
        if (lean && t->two_level) launch_fused_instanced(ta, t->has_spheres, grid, s);
        else if (lean) launch_fused(ta, t->has_spheres, grid, s);
        else launch_paths_v1(ta, t->has_spheres, grid, s);
    };
    // Longest-processing-time-first: a pixel's samples run one after the other in one lane, so a render ends with
    // whatever pixels were started last.  For renders of many samples the first sample is a probe launch of its own
    // that records how long each slice's pixels took over their sample; the slices are then handed out slowest first,
    // and the render ends on pixels whose paths leave the scene at once.  (The image does not depend on the order.)
    uint32_t done_spp = 0;
    const uint32_t n_slices = (n + ta.fetch_chunk - 1u) / ta.fetch_chunk;
    // Worth it when a lane gets only a few pixels (the tiles of the multi-GPU split: 1/4 of the C4 frame 238 -> 219 ms);
    // a full frame has ~6 pixels per lane, a tail of a few percent, and keeps its single launch.
    if (ctx->fused_lpt && spp >= 16u && n_slices >= 4096u && (uint64_t)n < 4ull * 64ull * (uint64_t)grid) {
        if (n_slices > w.slice_capacity) {
            for (uint32_t *p : {w.slice_cost, w.slice_order}) if (p) (void)hipFree(p);
            w.slice_cost = w.slice_order = nullptr; w.slice_capacity = 0;
            HIP_TRY(ctx, hipMalloc((void **)&w.slice_cost, sizeof(uint32_t) * n_slices));
            HIP_TRY(ctx, hipMalloc((void **)&w.slice_order, sizeof(uint32_t) * n_slices));
            w.slice_capacity = n_slices;
        }
        HIP_TRY(ctx, hipMemsetAsync(w.slice_cost, 0, sizeof(uint32_t) * n_slices, s));
        HIP_TRY(ctx, hipMemsetAsync(stg, 0, sizeof(StageCounters), s));
        const uint32_t probe_spp = (uint32_t)std::min<int>(std::max(ctx->fused_lpt, 1), (int)spp / 4);
        pa.spp = probe_spp; pa.continue_sum = 0u; pa.slice_cost = w.slice_cost; pa.slice_order = nullptr;
        { Timer tm(ctx, s, HRT_K_PATHS); launch(); }
        ctx->fused_counters_clean = false;
        ctx->h_slice_cost.resize(n_slices); ctx->h_slice_order.resize(n_slices);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_slice_cost.data(), w.slice_cost, sizeof(uint32_t) * n_slices, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
        for (uint32_t i = 0; i < n_slices; ++i) ctx->h_slice_order[i] = i;
        std::stable_sort(ctx->h_slice_order.begin(), ctx->h_slice_order.end(),
                         [&](uint32_t x, uint32_t y) { return ctx->h_slice_cost[x] > ctx->h_slice_cost[y]; });
        HIP_TRY(ctx, hipMemcpyAsync(w.slice_order, ctx->h_slice_order.data(), sizeof(uint32_t) * n_slices, hipMemcpyHostToDevice, s));
        pa.slice_cost = nullptr; pa.slice_order = w.slice_order;
        done_spp = probe_spp;
    }
    // very long renders are cut into launches of at most fused_max_spp samples (a launch should stay in the
    // range of seconds); the RNG states and the running sums carry over, so the result is the same bits
    while (done_spp < spp) {
        const uint32_t now = std::min<uint32_t>(spp - done_spp, (uint32_t)ctx->fused_max_spp);
        pa.spp = now; pa.continue_sum = done_spp > 0 ? 1u : 0u;
        // (the slice counters: zeroed by the finalize kernel of the previous launch when that was a path-kernel launch too)
        if (!ctx->fused_counters_clean) HIP_TRY(ctx, hipMemsetAsync(stg, 0, sizeof(StageCounters), s));
        ctx->fused_counters_clean = false;
        { Timer tm(ctx, s, HRT_K_PATHS); launch(); }
        done_spp += now;
    }
    FinalizeArgs fa{};
    fa.accum = w.accum; fa.rows = w.rows; fa.n_tile_pixels = n; fa.width = rg->width; fa.spp = spp;
    fa.color = reinterpret_cast<float4 *>(rg->colorBuffer); fa.albedo = reinterpret_cast<float4 *>(rg->albedoBuffer);
    fa.normal = reinterpret_cast<float4 *>(rg->normalBuffer); fa.linear = ctx->d_linear;
    constexpr uint32_t kCounterWords = (uint32_t)(sizeof(StageCounters) / sizeof(uint32_t));
    if (n >= kCounterWords) { fa.reset_counters = reinterpret_cast<uint32_t *>(stg); fa.n_reset = kCounterWords; }
    { Timer tm(ctx, s, HRT_K_FINALIZE); launch_finalize(fa, s); }
    ctx->fused_counters_clean = fa.reset_counters != nullptr;
    HIP_TRY(ctx, hipGetLastError());
    ctx->paths += (uint64_t)n * spp;
    ctx->last_tlas = h_params->handle;
    return HRT_OK;
}

// ---- wavefront mode (HRT_FUSED=0, and always under HRT_CTX_COUNT): the north-star pipeline with separate kernels -- generate, traverse,
//      bin by material, shade per program, accumulate -- every kernel reading its input count from device memory, so a launch is a
//      fixed sequence of enqueues with no host synchronisation ----
static int render_wavefront(const LaunchFrame &f, bool count) {
    HrtContext *ctx = f.ctx; Tlas *t = f.t; const HrtGlobalParams *h_params = f.h_params; const HrtRayGenParams *rg = f.rg;
    const uint32_t spp = f.spp, n = f.n; hipStream_t s = f.s;
    Workspace &w = ctx->ws;
    int rc = HRT_OK;

    ctx->fused_counters_clean = false;      // (the wavefront schedule below shares the counters' memory)
    // ---- sub-tiles: contiguous ranges of the tile's pixels, each on its own stream.  A traverse
    //      launch ends with a tail (the longest rays, ~0.3 ms) during which most CUs idle; with
    //      2-3 independent sub-tiles in flight one sub-tile's tail overlaps another's bulk. ----
    // 0 = automatic: a full frame keeps one stream (clean per-kernel timing, the GPU is full anyway);
    // the smaller tiles of the multi-GPU split are latency-bound per stage and gain from 2-3 sub-tiles
    // in flight (measured on the 1/8 tile: 41.4 -> 37.6 ms per 32 spp with 3).
    uint32_t S = ctx->substreams > 0 ? (uint32_t)ctx->substreams : (n >= 1200000u ? 1u : (n >= 600000u ? 2u : 3u));
    while (S > 1 && n / S < (uint32_t)ctx->substream_min_pixels) --S;
    if (S > 1) {
        while (ctx->sub_streams.size() < S) {
            hipStream_t st; HIP_TRY(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); ctx->sub_streams.push_back(st);
            hipEvent_t ev; HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming)); ctx->sub_done.push_back(ev);
        }
        if (!ctx->ev_begin) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_begin, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, s));
        for (uint32_t k = 0; k < S; ++k) HIP_TRY(ctx, hipStreamWaitEvent(ctx->sub_streams[k], ctx->ev_begin, 0));
    }
    struct Sub { uint32_t j0, n; hipStream_t st; uint32_t index; uint32_t grid_wide, grid_trav; };
    if (S > (uint32_t)kMaxSubTiles) S = kMaxSubTiles;
    // hipGraph replay of the per-sample launch sequence (below).  A capture cannot run on the legacy null stream -- the one the
    // reference, and a caller that passes NULL, uses -- so the pipeline then runs on a stream of the context's own, ordered after
    // the caller's work by an event and joined back before the finalize kernel.
    const bool want_graph = ctx->wavefront_graph && S == 1 && (ctx->flags & HRT_CTX_TIMING) == 0 && spp >= 5u;
    hipStream_t ws = s;
    if (want_graph && s == nullptr) {
        if (!ctx->graph_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->graph_stream, hipStreamNonBlocking));
        if (!ctx->ev_begin) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_begin, hipEventDisableTiming));
        if (!ctx->ev_graph_done) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_graph_done, hipEventDisableTiming));
        ws = ctx->graph_stream;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, s));
        HIP_TRY(ctx, hipStreamWaitEvent(ws, ctx->ev_begin, 0));
    }
    std::vector<Sub> subs(S);
    for (uint32_t k = 0; k < S; ++k) {
        const uint32_t j0 = (uint32_t)((uint64_t)n * k / S), j1 = (uint32_t)((uint64_t)n * (k + 1) / S);
        subs[k].j0 = j0; subs[k].n = j1 - j0; subs[k].st = S > 1 ? ctx->sub_streams[k] : ws; subs[k].index = k;
        subs[k].grid_wide = std::min<uint32_t>((uint32_t)ctx->n_cu * 8u, (subs[k].n + 255u) / 256u);
        subs[k].grid_trav = std::min<uint32_t>((uint32_t)ctx->n_cu * (uint32_t)ctx->traverse_blocks_per_cu, (2u * subs[k].n + 63u) / 64u);
    }

    // ---- schedule.  Class (s, d) = the rays of sample s at depth d.  Samples are sequential per pixel
    //      (one persistent RNG stream), but two things are off that chain: the primary rays of sample
    //      s+1 draw no random numbers, and the depth-5 rays of sample s only decide black / background.
    //      They ride along with the launches of the chain:
    //          T{(s,2),(s-1,5)}   T{(s,3)}   T{(s,4),(s+1,1)}
    //      so a sample costs three traversal launches instead of five, and the under-filled depth-5
    //      launch disappears.  Shading order (RNG) and the order of the per-sample sums stay sequential. ----
    auto stages_of = [&](uint32_t sample, const Sub &sb) { return w.set[sample & 1u].stages + (size_t)sb.index * (kRayTraceDepth + 1); };
    auto seg_of = [&](uint32_t sample, uint32_t depth, const Sub &sb) {
        SampleSet &st = w.set[sample & 1u];
        TraverseSeg g{};
        g.rays = st.rays[(depth - 1u) & 1u] + sb.j0;
        g.n_ptr = depth == 1 ? nullptr : &stages_of(sample, sb)[depth - 1].bin_count[1];
        g.n = sb.n;
        g.any_hit = depth >= kRayTraceDepth ? 1u : 0u;      // a hit at the depth limit is black whatever it is (Shader.cu:102-107)
        g.hit_tuvp = st.hit_tuvp + sb.j0; g.hit_inst = st.hit_inst + sb.j0;
        g.count_nodes = g.any_hit ? &ctx->d_stats->nodes_any : &ctx->d_stats->nodes_closest;
        g.count_prims = g.any_hit ? &ctx->d_stats->prims_any : &ctx->d_stats->prims_closest;
        return g;
    };
    auto do_generate = [&](uint32_t sample, const Sub &sb) -> int {
        HIP_TRY(ctx, hipMemsetAsync(stages_of(sample, sb), 0, sizeof(StageCounters) * (kRayTraceDepth + 1), sb.st));
        GenerateArgs ga{};
        ga.rays = w.set[sample & 1u].rays[0] + sb.j0; ga.rows = w.rows; ga.first_pixel = sb.j0; ga.n_tile_pixels = sb.n;
        ga.width = rg->width; ga.height = rg->height;
        std::memcpy(ga.center, &rg->cameraCenter, 12); std::memcpy(ga.U, &rg->cameraU, 12);
        std::memcpy(ga.V, &rg->cameraV, 12); std::memcpy(ga.W, &rg->cameraW, 12);
        Timer tm(ctx, sb.st, HRT_K_GENERATE); launch_generate(ga, sb.st);
        return HRT_OK;
    };
    // one traverse launch over class (sa, da) and, optionally, class (sb_, db)
    auto do_traverse = [&](const Sub &sb, uint32_t sa, uint32_t da, bool second, uint32_t sb_, uint32_t db) {
        TraverseArgs ta{};
        ta.nodes = t->d_nodes; ta.prims = t->d_prims; ta.node_stride = t->node_stride; ta.prim_stride = t->prim_stride;
        ta.seg[0] = seg_of(sa, da, sb);
        if (second) ta.seg[1] = seg_of(sb_, db, sb);
        ta.fetch_counter = stages_of(sa, sb)[da].fetch;
        ta.inst_inv = t->d_inst_inv; ta.inst_identity = t->d_inst_identity;
        ta.tmin = kFloatZero; ta.tmax = kFloatInfinity;      // Shader.cu:232, :266
        ta.refill_threshold = ctx->refill_threshold; ta.postpone_pct = ctx->postpone_pct; ta.leaf_quorum = ctx->leaf_quorum; ta.leaf_hold = 4; ta.tail_regen = ctx->fused_tail_regen; ta.tail_split = ctx->tail_split; ta.fetch_chunk = (uint32_t)ctx->fetch_chunk;
        Timer tm(ctx, sb.st, (da >= kRayTraceDepth && !second) ? HRT_K_TRAVERSE_ANY : HRT_K_TRAVERSE);
        // production traversal: the loop of the path kernel over the ray queues (k_trace_queue); the counting build, the LDS-DMA gather
        // mode and trees that do not fit k_fused's stacks / offsets keep round 1's k_traverse
        if (!count && ctx->lds_gather == 0 && ctx->wavefront_lean && fits_fused_kernel(ctx, *t)) {
            ta.postpone_pct = ctx->fused_postpone_pct; ta.refill_threshold = ctx->fused_refill_threshold;
            const uint32_t grid = std::min<uint32_t>((uint32_t)ctx->n_cu * (uint32_t)std::min(ctx->traverse_blocks_per_cu, kFusedBlocksPerCu), (2u * sb.n + 63u) / 64u);
            launch_trace_queue(ta, t->has_spheres, grid, sb.st);
        } else launch_traverse(ta, count, t->has_spheres, ctx->lds_gather != 0, sb.grid_trav, sb.st);
    };
    // everything that follows the traversal of class (sample, depth): binning, shading, path ends
    auto do_after = [&](const Sub &sb, uint32_t sample, uint32_t depth) {
        SampleSet &st = w.set[sample & 1u];
        StageCounters *stg = stages_of(sample, sb);
        hipStream_t strm = sb.st;
        uint32_t *bin_items = st.bin_items + (size_t)sb.j0 * kNumBins;
        const RayRec *rays_in = st.rays[(depth - 1u) & 1u] + sb.j0;
        BinArgs ba{};
        ba.n_rays_ptr = depth == 1 ? nullptr : &stg[depth - 1].bin_count[1]; ba.n_rays = sb.n;
        ba.hit_inst = st.hit_inst + sb.j0; ba.inst_program = ctx->d_inst_program; ba.depth = depth;
        ba.bin_count = stg[depth].bin_count; ba.bin_items = bin_items; ba.bin_stride = sb.n;
        ba.total_rays = depth >= kRayTraceDepth ? &ctx->d_stats->rays_any : &ctx->d_stats->rays_closest;
        { Timer tm(ctx, strm, HRT_K_BIN); launch_bin(ba, sb.grid_wide, strm); }
        if (depth < kRayTraceDepth) {
            ShadeArgs sa{};
            sa.bin_count = stg[depth].bin_count; sa.bin_items = bin_items; sa.bin_stride = sb.n;
            sa.rays_in = rays_in; sa.rays_out = st.rays[depth & 1u] + sb.j0;
            sa.hit_tuvp = st.hit_tuvp + sb.j0; sa.hit_inst = st.hit_inst + sb.j0; sa.hitgroups = ctx->d_hitgroups;
            sa.states = reinterpret_cast<RngState *>(h_params->stateArray);
            sa.chain = st.chain; sa.depth = depth;
            for (int p = 0; p < (int)kNumPrograms; ++p)
                if (ctx->program_present[p]) { Timer tm(ctx, strm, HRT_K_SHADE); launch_shade(sa, p, sb.grid_wide, strm); }
        }
        AccumArgs aa{};
        aa.bin_count = stg[depth].bin_count; aa.bin_items = bin_items; aa.rays_in = rays_in; aa.hit_inst = st.hit_inst + sb.j0;
        aa.hitgroups = ctx->d_hitgroups; aa.chain = st.chain; aa.result = st.result;
        aa.bg[0] = ctx->miss.backgroundColor.x; aa.bg[1] = ctx->miss.backgroundColor.y; aa.bg[2] = ctx->miss.backgroundColor.z;
        aa.depth = depth;
        { Timer tm(ctx, strm, HRT_K_ACCUMULATE); launch_accumulate(aa, sb.grid_wide, strm); }
        if (depth >= kRayTraceDepth) {       // the sample is complete: add it, in sample order
            Timer tm(ctx, strm, HRT_K_ACCUMULATE);
            launch_sum(w.accum + sb.j0, st.result + sb.j0, sb.n, sample == 0 ? 1u : 0u, strm);
        }
    };

    for (const Sub &sb : subs) { rc = do_generate(0, sb); if (rc != HRT_OK) return rc; }
    for (const Sub &sb : subs) do_traverse(sb, 0, 1, false, 0, 0);
    auto sample_body = [&](uint32_t sample) -> int {
        const bool prev = sample > 0, next = sample + 1 < spp;
        for (const Sub &sb : subs) do_after(sb, sample, 1);
        for (const Sub &sb : subs) do_traverse(sb, sample, 2, prev, sample - 1, kRayTraceDepth);
        if (prev) for (const Sub &sb : subs) do_after(sb, sample - 1, kRayTraceDepth);
        for (const Sub &sb : subs) do_after(sb, sample, 2);
        for (const Sub &sb : subs) do_traverse(sb, sample, 3, false, 0, 0);
        for (const Sub &sb : subs) do_after(sb, sample, 3);
        if (next) for (const Sub &sb : subs) { const int r = do_generate(sample + 1, sb); if (r != HRT_OK) return r; }
        for (const Sub &sb : subs) do_traverse(sb, sample, 4, next, sample + 1, 1);
        for (const Sub &sb : subs) do_after(sb, sample, 4);
        return HRT_OK;
    };
    uint32_t sample = 0;
    rc = sample_body(sample++);
    if (rc != HRT_OK) return rc;
    // Samples 2 .. spp - 2 enqueue the same ~45 operations on the same buffers, alternating between the two workspace sets:
    // a pair of them is captured once as a hipGraph and replayed (HRT_WAVEFRONT_GRAPH=1; one stream, no per-kernel event
    // timers inside a graph, so not under HRT_CTX_TIMING).  Every kernel reads its ray count from device memory, so the
    // replayed launches are the eager ones, and the image is the same bits.  (Sample 1 is not like the others: its launches
    // add sample 0 to the running sum with the "initialise" flag set.)
    if (want_graph) {
        rc = sample_body(sample++);
        if (rc != HRT_OK) return rc;
        const uint32_t pairs = (spp - 3u) / 2u;
        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
        HIP_TRY(ctx, hipStreamBeginCapture(ws, hipStreamCaptureModeThreadLocal));
        int rc1 = sample_body(2u);
        if (rc1 == HRT_OK) rc1 = sample_body(3u);
        const hipError_t ce = hipStreamEndCapture(ws, &graph);
        if (rc1 != HRT_OK) { if (graph) (void)hipGraphDestroy(graph); return rc1; }
        HIP_TRY(ctx, ce);
        hipError_t ge = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        for (uint32_t k = 0; k < pairs && ge == hipSuccess; ++k) ge = hipGraphLaunch(exec, ws);
        // (the executable graph must outlive its launches: the stream is drained before it is destroyed)
        if (ge == hipSuccess) ge = hipStreamSynchronize(ws);
        if (exec) (void)hipGraphExecDestroy(exec);
        (void)hipGraphDestroy(graph);
        HIP_TRY(ctx, ge);
        sample += 2u * pairs;
        ctx->graph_replays += pairs;
    }
    for (; sample < spp; ++sample) { rc = sample_body(sample); if (rc != HRT_OK) return rc; }
    for (const Sub &sb : subs) do_traverse(sb, spp - 1, kRayTraceDepth, false, 0, 0);
    for (const Sub &sb : subs) do_after(sb, spp - 1, kRayTraceDepth);
    if (ws != s) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_graph_done, ws));
        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_graph_done, 0));
    }
    if (S > 1) {
        for (uint32_t k = 0; k < S; ++k) {
            HIP_TRY(ctx, hipEventRecord(ctx->sub_done[k], ctx->sub_streams[k]));
            HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->sub_done[k], 0));
        }
    }
    FinalizeArgs fa{};
    fa.accum = w.accum; fa.rows = w.rows; fa.n_tile_pixels = n; fa.width = rg->width; fa.spp = spp;
    fa.color = reinterpret_cast<float4 *>(rg->colorBuffer); fa.albedo = reinterpret_cast<float4 *>(rg->albedoBuffer);
    fa.normal = reinterpret_cast<float4 *>(rg->normalBuffer); fa.linear = ctx->d_linear;
    { Timer tm(ctx, s, HRT_K_FINALIZE); launch_finalize(fa, s); }
    HIP_TRY(ctx, hipGetLastError());
    ctx->paths += (uint64_t)n * spp;
    ctx->last_tlas = h_params->handle;
    return HRT_OK;
}

int hrt_render_launch(HrtContext *ctx, const HrtGlobalParams *h_params, const HrtRayGenParams *rg, uint32_t spp,
                      const HrtTile *tile, void *stream) {
    if (!ctx || !h_params || !rg) return HRT_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    (void)hipSetDevice(ctx->device);
    if (spp == 0) return fail(ctx, HRT_ERR_INVALID, "spp must be >= 1");
    if (rg->width == 0 || rg->height == 0 || (uint64_t)rg->width * rg->height > 0xffffffffull) return fail(ctx, HRT_ERR_INVALID, "bad frame size %ux%u", rg->width, rg->height);
    if (!rg->colorBuffer) return fail(ctx, HRT_ERR_INVALID, "RayGenParams.colorBuffer is NULL");
    if (!h_params->stateArray) return fail(ctx, HRT_ERR_INVALID, "GlobalParams.stateArray is NULL");
    if (!ctx->have_records) return fail(ctx, HRT_ERR_STATE, "hrt_materials_set has not been called");
    Tlas *t;
    { std::lock_guard<std::mutex> lk(ctx->mu); auto it = ctx->tlas.find(h_params->handle);
      if (it == ctx->tlas.end()) return fail(ctx, HRT_ERR_INVALID, "GlobalParams.handle 0x%llx is not a TLAS", (unsigned long long)h_params->handle);
      t = it->second.get(); }
    int rc = refresh_tables(ctx, h_params->handle, t, s);
    if (rc != HRT_OK) return rc;

    // ---- tile rows ----
    HrtTile tl = tile ? *tile : HrtTile{0, rg->height, 1, 1, 0};
    if (tl.stripe_rows == 0 || tl.stripe_period == 0 || tl.stripe_phase >= tl.stripe_period || tl.y_begin > tl.y_end || tl.y_end > rg->height)
        return fail(ctx, HRT_ERR_INVALID, "bad tile");
    const bool same_rows = ctx->rows_w == rg->width && ctx->rows_h == rg->height && std::memcmp(&ctx->rows_tile, &tl, sizeof tl) == 0 && ctx->ws.rows;
    if (!same_rows) {
        ctx->rows_w = ctx->rows_h = 0;       // the cached row list is being replaced: its key is valid again only once the upload below has succeeded
        ctx->h_rows.clear();
        for (uint32_t y = tl.y_begin; y < tl.y_end; ++y)
            if ((y / tl.stripe_rows) % tl.stripe_period == tl.stripe_phase) ctx->h_rows.push_back(y);
    }
    const uint32_t n_rows = (uint32_t)ctx->h_rows.size();
    const uint64_t n64 = (uint64_t)n_rows * rg->width;
    if (n64 == 0) return HRT_OK;
    const uint32_t n = (uint32_t)n64;
    rc = ensure_workspace(ctx, n, rg->height);
    if (rc != HRT_OK) return rc;
    Workspace &w = ctx->ws;
    if (!same_rows) {
        HIP_TRY(ctx, hipMemcpyAsync(w.rows, ctx->h_rows.data(), sizeof(uint32_t) * n_rows, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
        ctx->rows_tile = tl; ctx->rows_w = rg->width; ctx->rows_h = rg->height;
    }

    const bool count = (ctx->flags & HRT_CTX_COUNT) != 0;
    if (t->two_level && (count || ctx->fused != 1)) return fail(ctx, HRT_ERR_STATE, "a two-level TLAS is traced by the default path kernel only (not under HRT_CTX_COUNT / HRT_FUSED != 1)");
    const LaunchFrame frame{ctx, t, h_params, rg, spp, n, s};
    const bool use_fused = !count && (ctx->fused > 0 || (ctx->fused < 0 && n <= (uint32_t)ctx->fused_max_pixels));
    return use_fused ? render_fused(frame) : render_wavefront(frame, count);
}


int hrt_sync(HrtContext *ctx, void *stream) {
    if (!ctx) return HRT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    return HRT_OK;
}

int hrt_to_rgba8(HrtContext *ctx, const HrtFloat4 *d_src, HrtUchar4 *d_dst, uint32_t width, uint32_t height, void *stream) {
    if (!ctx || !d_src || !d_dst) return HRT_ERR_INVALID;
    const uint64_t n = (uint64_t)width * height;
    if (n > 0xffffffffull) return fail(ctx, HRT_ERR_INVALID, "frame too large");
    (void)hipSetDevice(ctx->device);
    launch_to_rgba8(reinterpret_cast<const float4 *>(d_src), reinterpret_cast<uchar4 *>(d_dst), (uint32_t)n, (hipStream_t)stream);
    HIP_TRY(ctx, hipGetLastError());
    return HRT_OK;
}

int hrt_color_to_float4(HrtContext *ctx, const HrtFloat4 *d_src, HrtFloat4 *d_dst, uint32_t n, void *stream) {
    if (!ctx || (n && (!d_src || !d_dst))) return HRT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    launch_color_to_float4(reinterpret_cast<const float4 *>(d_src), reinterpret_cast<float4 *>(d_dst), n, (hipStream_t)stream);
    HIP_TRY(ctx, hipGetLastError());
    return HRT_OK;
}

// ---- measurement ----------------------------------------------------------------------
int hrt_stats_reset(HrtContext *ctx) {
    if (!ctx) return HRT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemset(ctx->d_stats, 0, sizeof(DeviceStats)));
    drain_spans(ctx);
    for (int k = 0; k < HRT_K_COUNT; ++k) { ctx->kernel_ms[k] = 0.0; ctx->kernel_launches[k] = 0; }
    ctx->paths = 0;
    return HRT_OK;
}

int hrt_stats_get(HrtContext *ctx, HrtStats *out) {
    if (!ctx || !out) return HRT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    HIP_TRY(ctx, hipDeviceSynchronize());
    drain_spans(ctx);
    DeviceStats ds;
    HIP_TRY(ctx, hipMemcpy(&ds, ctx->d_stats, sizeof ds, hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof *out);
    out->rays_closest = ds.rays_closest; out->rays_any = ds.rays_any; out->rays = ds.rays_closest + ds.rays_any;
    out->paths = ctx->paths; out->node_visits = ds.nodes_closest + ds.nodes_any; out->prim_tests = ds.prims_closest + ds.prims_any;
    out->node_visits_closest = ds.nodes_closest; out->prim_tests_closest = ds.prims_closest;
    for (int k = 0; k < 4; ++k) out->debug[k] = ds.debug[k];
    out->tlas_refits = ctx->tlas_refits; out->tlas_rebuilds = ctx->tlas_rebuilds; out->tlas_refit_ratio = ctx->tlas_refit_ratio;
    for (int k = 0; k < HRT_K_COUNT; ++k) { out->kernel_ms[k] = ctx->kernel_ms[k]; out->kernel_launches[k] = ctx->kernel_launches[k]; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->tlas.find(ctx->last_tlas);
    if (it != ctx->tlas.end()) {
        const Tlas &tl = *it->second;
        out->bvh_nodes = tl.n_nodes; out->bvh_triangles = tl.n_triangles; out->bvh_spheres = tl.n_spheres; out->bvh_depth = tl.max_depth;
        out->bvh_bytes = (uint64_t)tl.n_nodes * sizeof(Bvh8Node) + (uint64_t)tl.n_prims * sizeof(PrimRecord);
        out->bvh_alloc_bytes = tl.alloc_bytes;
    }
    out->fused_fallback_launches = ctx->fused_fallback_launches; out->graph_replays = ctx->graph_replays;
    return HRT_OK;
}

// ---- introspection for the parity tests -----------------------------------------------
int hrt_trace_rays(HrtContext *ctx, HrtTraversable tlas, const HrtFloat3 *d_origins, const HrtFloat3 *d_directions, uint32_t n_rays,
                   float tmin, float tmax, int any_hit, float *d_t, float *d_u, float *d_v, uint32_t *d_prim, uint32_t *d_inst, void *stream) {
    if (!ctx || !d_origins || !d_directions || !d_t || !d_u || !d_v || !d_prim || !d_inst) return HRT_ERR_INVALID;
    if (n_rays == 0) return HRT_OK;
    hipStream_t s = (hipStream_t)stream;
    (void)hipSetDevice(ctx->device);
    Tlas *t;
    { std::lock_guard<std::mutex> lk(ctx->mu); auto it = ctx->tlas.find(tlas); if (it == ctx->tlas.end()) return fail(ctx, HRT_ERR_INVALID, "unknown TLAS handle"); t = it->second.get(); }
    RayRec *rays = nullptr; float4 *tuvp = nullptr; uint32_t *inst = nullptr, *fetch = nullptr;
    struct Scratch { void **p[4]; ~Scratch() { for (void **q : p) if (*q) (void)hipFree(*q); } }
        scratch{{(void **)&rays, (void **)&tuvp, (void **)&inst, (void **)&fetch}};     // freed on every way out
    HIP_TRY(ctx, hipMalloc((void **)&rays, sizeof(RayRec) * (size_t)n_rays));
    HIP_TRY(ctx, hipMalloc((void **)&tuvp, sizeof(float4) * (size_t)n_rays));
    HIP_TRY(ctx, hipMalloc((void **)&inst, sizeof(uint32_t) * (size_t)n_rays));
    HIP_TRY(ctx, hipMalloc((void **)&fetch, sizeof(uint32_t) * 8 * 32));
    HIP_TRY(ctx, hipMemsetAsync(fetch, 0, sizeof(uint32_t) * 8 * 32, s));
    launch_pack_rays(reinterpret_cast<const float *>(d_origins), reinterpret_cast<const float *>(d_directions), n_rays, rays, s);
    TraverseArgs ta{};
    ta.nodes = t->d_nodes; ta.prims = t->d_prims; ta.node_stride = t->node_stride; ta.prim_stride = t->prim_stride;
    ta.fetch_counter = fetch; ta.inst_inv = t->d_inst_inv; ta.inst_identity = t->d_inst_identity;
    ta.tmin = tmin; ta.tmax = tmax; ta.postpone_pct = ctx->postpone_pct; ta.leaf_quorum = ctx->leaf_quorum; ta.leaf_hold = ctx->leaf_hold > 0 ? ctx->leaf_hold : t->two_level ? 1 : (t->scene_of_bodies && t->n_instances >= 256u) ? 2 : 4; ta.tail_regen = ctx->fused_tail_regen;
    const bool count = (ctx->flags & HRT_CTX_COUNT) != 0;
    if (t->two_level && (count || ctx->fused != 1 || !fits_fused_kernel(ctx, *t))) return fail(ctx, HRT_ERR_STATE, "a two-level TLAS is traced by the default path kernel only");
    if (!count && ctx->fused > 0) {
        // the production configuration: the rays go through the very kernel hrt_render_launch runs (fused path kernel, v_rcp_f32
        // slab test, regeneration thresholds), each ray standing in for a pixel that is traced once and not shaded
        ta.refill_threshold = ctx->fused_refill_threshold; ta.tail_split = ctx->tail_split; ta.fetch_chunk = (uint32_t)ctx->fused_fetch_chunk;
        PathArgs &pa = ta.path;
        pa.n_tile_pixels = n_rays; pa.first_pixel = 0; pa.width = n_rays; pa.height = 1; pa.spp = 1;
        pa.trace_rays = rays; pa.trace_tuvp = tuvp; pa.trace_inst = inst; pa.trace_any = any_hit ? 1u : 0u;
        pa.rays_closest = &ctx->d_stats->rays_closest; pa.rays_any = &ctx->d_stats->rays_any;
        const bool lean = ctx->fused != 2 && fits_fused_kernel(ctx, *t);
        if (lean) ta.postpone_pct = ctx->fused_postpone_pct;
        else if (ctx->fused != 2) ctx->fused_fallback_launches++;
        const uint32_t blocks_per_cu = lean ? std::min<uint32_t>((uint32_t)ctx->fused_blocks_per_cu, (uint32_t)(t->two_level ? kFusedInstancedBlocksPerCu : kFusedBlocksPerCu)) : (uint32_t)ctx->fused_blocks_per_cu;
        const uint32_t grid = std::min<uint32_t>((uint32_t)ctx->n_cu * blocks_per_cu, (n_rays + 63u) / 64u);
        Timer tm(ctx, s, HRT_K_PATHS);
        if (t->two_level) { ta.tail_split = 0; launch_fused_instanced(ta, t->has_spheres, grid, s); }
        else if (lean) launch_fused(ta, t->has_spheres, grid, s); else launch_paths_v1(ta, t->has_spheres, grid, s);
    } else {
        ta.seg[0].rays = rays; ta.seg[0].n_ptr = nullptr; ta.seg[0].n = n_rays; ta.seg[0].any_hit = any_hit ? 1u : 0u;
        ta.seg[0].hit_tuvp = tuvp; ta.seg[0].hit_inst = inst;
        ta.seg[0].count_nodes = any_hit ? &ctx->d_stats->nodes_any : &ctx->d_stats->nodes_closest;
        ta.seg[0].count_prims = any_hit ? &ctx->d_stats->prims_any : &ctx->d_stats->prims_closest;
        ta.refill_threshold = ctx->refill_threshold; ta.tail_split = ctx->tail_split; ta.fetch_chunk = (uint32_t)ctx->fetch_chunk;
        const uint32_t grid = std::min<uint32_t>((uint32_t)ctx->n_cu * (uint32_t)ctx->traverse_blocks_per_cu, (n_rays + 63u) / 64u);
        Timer tm(ctx, s, any_hit ? HRT_K_TRAVERSE_ANY : HRT_K_TRAVERSE);
        launch_traverse(ta, count, t->has_spheres, ctx->lds_gather != 0, grid, s);
    }
    launch_unpack_hits(tuvp, inst, n_rays, d_t, d_u, d_v, d_prim, d_inst, s);
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(ctx, HRT_ERR_HIP, "hrt_trace_rays: %s", hipGetErrorString(e));
    ctx->last_tlas = tlas;
    return HRT_OK;
}

int hrt_debug_set_linear_output(HrtContext *ctx, HrtFloat4 *d_linear) {
    if (!ctx) return HRT_ERR_INVALID;
    ctx->d_linear = reinterpret_cast<float4 *>(d_linear);
    return HRT_OK;
}

}  // extern "C"
