// hrt_internal.hpp -- what the translation units of libhrt.so share: the objects behind the opaque handles of
// include/hrt.h (context, BLAS, TLAS, workspace), error plumbing and the helpers that cross file boundaries.
// hrt_api.cpp: context, materials, RNG, the launch, measurement.  hrt_accel.cpp: acceleration structures
// (build, per-frame refit, trees over instances, poses, download).
#pragma once
#include "../../include/hrt.h"
#include "bvh8.h"
#include "bvh8_geom.h"
#include "device_types.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>


namespace hrt {

struct Blas {
    uint32_t kind = kPrimKindTriangle;
    uint32_t n_prims = 0;
    // host copies of the geometry: only filled when the HOST builder is asked for (HRT_BUILD=host), from d_verts
    std::vector<float> verts;        // triangles: 9 floats each (object space)
    std::vector<float> centers;      // spheres: 3 floats each
    std::vector<float> radii;
    bool host_geometry = false;
    float *d_verts = nullptr;        // device copy of the geometry (triangles: 9 floats each; spheres: {cx, cy, cz, r}): the device
                                     // build and every refit derive the world-space records from it
                                     // (the caller may free its vertex buffer after the build, RendererMesh.cu:116)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};   // object-space bounds
    // object-space BVH8 of this geometry alone: the per-instance subtree of the trees over instances (built on first use)
    std::mutex tmpl_mu; bool tmpl_built = false; Bvh8 tmpl;
    // ... and its device copy (topology, packed 80 / 48 byte records): what a two-level TLAS copies in behind its top level
    unsigned char *d_tmpl_nodes = nullptr, *d_tmpl_prims = nullptr;
    float bsphere[4] = {0, 0, 0, -1.0f}; bool bsphere_ready = false;      // bounding sphere (centre of the box, largest vertex distance): the instances' tight bound
    ~Blas() { if (d_verts) (void)hipFree(d_verts); if (d_tmpl_nodes) (void)hipFree(d_tmpl_nodes); if (d_tmpl_prims) (void)hipFree(d_tmpl_prims); }
};

struct Tlas {
    uint32_t n_instances = 0;
    std::vector<uint32_t> sbt_offset;      // per instance
    std::vector<uint32_t> kind;            // per instance: triangle / sphere BLAS
    Bvh8 bvh;                               // host copy of a HOST-built tree (empty for device builds)
    uint32_t n_nodes = 0, n_prims = 0, n_triangles = 0, n_spheres = 0, max_depth = 0;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    void *d_nodes = nullptr, *d_prims = nullptr;
    float *d_inst_inv = nullptr;
    uint32_t *d_inst_identity = nullptr;
    bool has_spheres = false;
    uint32_t node_stride = 80, prim_stride = 48;
    uint64_t alloc_bytes = 0;               // device memory of nodes + node boxes / reference areas + records, as allocated
    uint64_t generation = 0;
    // refit (hrt_tlas_update): what must stay the same, and the device tables the refit kernel reads
    std::vector<std::shared_ptr<Blas>> blas_refs;       // keeps the source geometry alive
    std::vector<uint64_t> sig_handle; std::vector<uint32_t> sig_visibility;
    std::vector<float> h_xf, h_inv; std::vector<uint32_t> h_ident;   // staging of the per-instance uploads
    float *d_node_box = nullptr, *d_node_ref = nullptr, *d_inst_xf = nullptr, *d_area = nullptr;
    uint32_t *d_order = nullptr;                         // trees over instances: refit order (NULL: breadth-first index ranges)
    uint32_t *d_inst_first = nullptr, *d_inst_kind = nullptr;   // device build: global number of each instance's first primitive; geometry kind
    // asynchronous updates (HRT_CTX_ASYNC_UPDATE): what the tree was built with, for the device-side tables kernel, and its verdict
    unsigned long long *d_sig_handle = nullptr; uint32_t *d_sig_visibility = nullptr, *d_sig_sbt = nullptr; float *d_blas_box = nullptr;
    uint32_t *d_update_flags = nullptr;                  // [0] scene scale (float bits), [1] bit 0: handle / visibility changed, bit 1: an sbtOffset changed
    bool built_posed = false;                            // built by an update, for the poses of that frame (not the reference's identity-posed load-time build): asynchronous updates may follow at once
    bool async_words_ready = false;                      // d_update_flags / d_area are in their initial state (the asynchronous update's epilogue leaves them so)
    uint32_t *h_update_flags = nullptr;                  // pinned: [0..1] copy of the above after the last update, [2..3] their initial values
    std::vector<std::pair<uint32_t, uint32_t>> phases;   // (first, count) in processing order, children before parents
    bool instanced = false;
    // TWO-LEVEL tree (bvh8.h: transform nodes; the reference's IAS over shared GASes): nodes [0, n_top_nodes) are the top level -- box nodes and
    // one transform node per visible instance, what an update refits --, behind them one object-space tree per UNIQUE BLAS; d_prims holds those
    // trees' records only.  Memory and update cost: instances + unique primitives.
    bool two_level = false; uint32_t n_top_nodes = 0, n_unique_blas = 0;
    bool scene_of_bodies = false;        // at least 4 visible instances of fewer than 20 000 primitives each on average (the reference's kind of scene)
    uint32_t *d_inst_root = nullptr;                     // per instance: node index of its BLAS's root
    float *d_blas_bound = nullptr;                       // per instance, 10 floats: its BLAS's object-space box and bounding sphere (the top level's "geometry")
    float built_reach = 1.0f;                            // the largest object-space |coordinate| a ray origin was assumed to have when the BLAS trees were padded
    bool has_split_refs = false;                         // built with spatial splits (HRT_CTX_FAST_TRACE): a refit would recompute the leaf boxes from whole primitives, so the first update rebuilds instead
    const void **d_inst_src = nullptr;
    float *h_area = nullptr;                             // pinned: area sum of the last refit
    hipEvent_t area_ready = nullptr; bool area_pending = false;
    float *d_rec_box = nullptr;                          // small trees: 6 floats per record, the refit's per-record pass (refit.hip k_refit_records)
    uint64_t refits = 0, rebuilds = 0, refits_since_build = 0;
};

// per-depth counters, zeroed once per sample: bin sizes + 8 slice counters on 128-byte lines of their own
struct StageCounters { uint32_t bin_count[32]; uint32_t fetch[8 * 32]; };
static_assert(sizeof(StageCounters) == 128 + 8 * 128, "stage counters layout");

struct DeviceStats { uint64_t rays_closest, rays_any, nodes_closest, prims_closest, nodes_any, prims_any; uint64_t debug[4]; };

// Everything one sample needs besides the per-pixel state.  Two sets (sample parity): the stages of
// two consecutive samples overlap in time (hrt_render_launch), never more.
struct SampleSet {
    RayRec *rays[2] = {nullptr, nullptr};  // depth d reads rays[(d-1)&1], shade writes rays[d&1]
    float4 *hit_tuvp = nullptr; uint32_t *hit_inst = nullptr;
    uint32_t *bin_items = nullptr;          // kNumBins x n ray indices
    uint32_t *chain = nullptr;              // 4 instance indices per tile pixel
    float4 *result = nullptr;               // the sample's linear radiance per tile pixel
    StageCounters *stages = nullptr;        // [sub-tile][kRayTraceDepth + 1]
};
struct Workspace {
    uint32_t capacity = 0, rows_capacity = 0;
    SampleSet set[2];
    float4 *accum = nullptr;
    uint32_t *rows = nullptr;
    uint32_t *slice_cost = nullptr, *slice_order = nullptr; uint32_t slice_capacity = 0;   // fused mode: cost-ordered slices
    float4 *primary_cache = nullptr; uint32_t primary_cache_lanes = 0;                     // HRT_CTX_REUSE_PRIMARY: two float4 per lane of the path kernel's grid
};

struct TimedSpan { int kind; hipEvent_t a, b; };

// working memory of the device builds, kept by the context between builds (a few arenas: builds may run on several loader threads)
struct ScratchArena { void *p = nullptr; size_t bytes = 0; };

}  // namespace hrt

namespace hrt { extern int g_instance_table_threads; }      // host threads that derive the per-instance tables of 32768 instances and more (HRT_TABLE_THREADS)
using namespace hrt;

struct HrtContext {
    int device = 0;
    uint32_t flags = 0;
    int n_cu = 256;
    std::string error;                          // last failure of any thread (under the error lock of hrt_api.cpp); a thread asks for its own first
    std::mutex mu;
    std::unordered_map<uint64_t, std::shared_ptr<hrt::Blas>> blas;
    std::unordered_map<uint64_t, std::unique_ptr<hrt::Tlas>> tlas;
    uint64_t next_handle = 0x1000;
    std::mutex pin_mu; void *pin_stage = nullptr; size_t pin_bytes = 0;      // pinned staging for the instance array of large synchronous updates (hrt_accel.cpp download_instances)
    std::mutex scratch_mu; std::vector<hrt::ScratchArena> scratch_free;      // scratch_acquire / scratch_release (hrt_accel.cpp)
    std::mutex pool_mu; std::vector<hrt::ScratchArena> pool_free; std::unordered_map<void *, size_t> pool_live; size_t pool_bytes = 0;   // pool_alloc / pool_release: the trees' device memory
    // materials
    std::vector<HrtSbtRecord> records;
    HrtMissParams miss{{0.7f, 0.8f, 0.9f}};      // reference default, src/Global/RendererMesh.cu:262
    bool have_records = false;
    uint64_t materials_generation = 0;
    // per-launch device tables derived from (tlas, records)
    HitGroup *d_hitgroups = nullptr; uint32_t *d_inst_program = nullptr; uint32_t table_capacity = 0;
    uint64_t table_tlas = 0, table_tlas_gen = 0, table_mat_gen = ~0ull;
    bool program_present[kNumPrograms] = {false, false, false, false};
    // rng
    uint32_t *d_jump = nullptr;
    // workspace + stats
    Workspace ws;
    std::vector<uint32_t> h_slice_cost, h_slice_order;
    std::vector<uint32_t> h_rows; HrtTile rows_tile{0, 0, 0, 0, 0}; uint32_t rows_w = 0, rows_h = 0;
    DeviceStats *d_stats = nullptr;
    uint64_t paths = 0;
    uint64_t last_tlas = 0;
    std::vector<TimedSpan> spans; std::vector<hipEvent_t> event_pool; size_t events_used = 0;
    double kernel_ms[HRT_K_COUNT] = {0}; uint64_t kernel_launches[HRT_K_COUNT] = {0};
    float4 *d_linear = nullptr;
    int refill_threshold = 16;                  // wavefront mode; fused mode: fused_refill_threshold
    int fused_refill_threshold = 20, fused_fetch_chunk = 16;   // measured optimum of the fused path mode (profiles/r01_sweep_fused_*.txt, r02_sweep_fused_knobs.txt, r03_sweep_fused_knobs.txt: 20 is 1 % ahead of 16)
    int traverse_blocks_per_cu = 16;            // one-wave workgroups of the wavefront traverse kernel per CU
    int fused_blocks_per_cu = 20;               // ... of the fused path kernels: round 1's takes 5 waves per SIMD (96 VGPRs), k_fused is capped at kFusedBlocksPerCu
    bool traverse_blocks_auto = true;           // fused mode: fewer of them for small tiles (not when the env knob is set)
    int postpone_pct = 25;                      // wavefront kernels and round 1 path kernel; k_fused: fused_postpone_pct
    int fused_postpone_pct = 40;                // (r02_sweep_lanes_active.txt: same speed as 25, 65 % of the lanes active instead of 63.4 %)
    int fused_max_depth = kFusedMaxDepth;       // deeper trees take round 1's fused kernel (HRT_FUSED_MAX_DEPTH lowers it: tests)
    uint64_t fused_max_bytes = 1ull << 32;      // k_fused addresses nodes and records by 32-bit byte offsets: larger arrays take round 1's kernel (HRT_FUSED_MAX_BYTES lowers it: tests)
    bool fused_counters_clean = false;          // the path kernel's slice counters are zero (the previous launch's finalize kernel left them so)
    uint64_t fused_fallback_launches = 0;       // launches that took round 1's path kernel because the tree did not fit k_fused
    int wavefront_lean = 1;                     // wavefront mode traverses with k_trace_queue (the loop of k_fused); 0: round 1's k_traverse (HRT_WAVEFRONT_LEAN)
    int wavefront_graph = 0;                    // wavefront mode: 1 = replay a captured pair of samples as a hipGraph; 0 (default) = enqueue every launch: the launches
                                                // are not what limits the mode (traversal is 85 % of its GPU time), profiles/r03_wavefront.txt
    uint64_t graph_replays = 0;
    int fused_tail_regen = 12;                  // k_fused, tile used up: finished rays that wait before a regeneration (HRT_TAIL_REGEN; 1/8 of C4: 142 ms with 1, 129 with 8..16)
    bool refill_auto = true;                    // HRT_REFILL_THRESHOLD not set: two-level launches regenerate at 12 waiting lanes (profiles/r04_leaf_hold.txt)
    int leaf_hold = 0;                          // HRT_LEAF_HOLD: leaf groups a lane may queue before its node work waits for primitive tests; 0 = by scene (render_fused)
    int leaf_quorum = 1;                        // k_fused: lanes with nothing but leaf work wait until this many of them have gathered (HRT_LEAF_QUORUM)
    int tail_split = 1;
    int node_stride = 80, prim_stride = 64;     // bytes between records in HBM (80/48 packed; 128/64 = one cache line each)
    bool node_stride_auto = true;               // no HRT_NODE_STRIDE given: trees beyond the Infinity Cache (> 3.5 M primitives) get 128-byte nodes -- a packed
                                                // 80-byte node straddles two 128-byte lines two times in five, which only costs once the lines come from HBM
                                                // (32 M triangles: +3.4 %, 8 M: +1 %, C4: -3 %; profiles/r03_large_scenes_node_stride.txt)
    int fused = 1;                              // 1: fused persistent path kernel k_fused (default), 2: round 1's fused kernel, 0: wavefront kernels, -1: fused only for small tiles
    int fused_max_pixels = 700000;
    bool reuse_primary = false;                 // HRT_REUSE_PRIMARY: every launch as under HRT_CTX_REUSE_PRIMARY
    int fused_lpt = 2;                          // samples of the probe launch that orders the slices by cost for the rest of the render (0: off)
    int fused_max_spp = 512;                    // samples per fused launch
    int lds_gather = 0;                         // 1: cooperative LDS-DMA gathers, 0: per-lane register loads
    int fetch_chunk = 64;
    int substream_min_pixels = 32768;
    int two_level = 0;                          // hrt_tlas_build: 1 = two-level trees (transform nodes over shared BLASes) whenever the path kernel can take them,
                                                // -1 = never, 0 = when flattening would leave the caches: more than two_level_min_prims flattened primitives,
                                                // at least two_level_min_share times the unique ones (HRT_TWO_LEVEL, HRT_CTX_TWO_LEVEL)
    uint64_t two_level_min_prims = 4000000ull; float two_level_min_share = 4.0f;
    int tlas_instanced = 0;                     // 1: hrt_tlas_build makes trees over instances too, 0: only rebuilds during updates do, -1: never
    int build_on_device = 1;                    // 1: PLOC build on the GPU (build.hip), 0: binned-SAH build on the host (HRT_BUILD=host; needs a host copy of the geometry)
    float quant_guard = 1.25f;                  // device builds (HRT_QUANT_GUARD; 0: off): see build.hip emit_item
    int build_topdown = 1;                      // device builds of more than 4096 primitives start with the top-down phase of build_split.hip (object splits; HRT_BUILD_TOPDOWN=0: PLOC alone)
    int fast_trace_on_device = 1;               // HRT_CTX_FAST_TRACE builds: 1 = on the device with spatial splits (build_split.hip), 0 = the host builder (HRT_FAST_TRACE_BUILD=device|host)
    float split_budget = 1.0f, split_alpha = 1e-5f, split_bias = 0.95f, split_cut_bias = 1.0f; int split_cell_refs = 16;      // the device's spatial splits (HRT_SBVH_BUDGET / _ALPHA / _BIAS / _CELL_REFS)
    int ploc_radius = 2;                        // device build: nearest-neighbour search radius of the PLOC rounds (positions in Morton order); 2 traces fastest
                                                // on the soup scenes (C4: 24.1 node visits per ray, 16: 28.3, 64: 36.6 -- profiles/r02_build_bench.txt)
    int build_width = 8;                        // children per node at most (HRT_BVH_WIDTH)
    float build_c_node = 1.0f, build_c_prim = 0.45f;   // collapse costs (bvh8_build.cpp has the same defaults)
    // ... and the primitive's cost for scenes of BODIES (several instances of small meshes: the reference's kind): their rays test as many primitives as they
    // visit nodes, a lane tests one primitive per iteration, and the collapse's 0.45 -- right for a soup, whose rays visit three nodes per test -- makes leaves too
    // fat: 2.0 is 10-12 % faster on 2000 particles, 5-10 % on 25, 3 % on the shipped sample's loop, 8 % on two-level trees, and 1 % SLOWER on C4
    // (profiles/r04_cprim_sweep.txt).  HRT_BVH_CPRIM sets both.
    float build_c_prim_bodies = 2.0f;
    bool build_verbose = false;                 // HRT_BUILD_VERBOSE: builds and updates report on stderr
    int refit = 1;                              // hrt_tlas_update: 1 = device refit when only transforms changed, 0 = always rebuild
    int refit_moved_far_check = 1;              // the first update after a build rebuilds without refitting first when most instances have moved further than their size (HRT_REFIT_MOVED_FAR=0: always refit first)
    float refit_rebuild_ratio = 1.5f;           // rebuild when the refitted tree's weighted mean node area has grown by this factor
    std::atomic<uint64_t> tlas_refits{0}, tlas_rebuilds{0}; std::atomic<double> tlas_refit_ratio{1.0};   // (builds run on several loader threads)
    int substreams = 0;                         // sub-tiles rendered on their own HIP streams so that one's tail overlaps another's bulk
    std::vector<hipStream_t> sub_streams; std::vector<hipEvent_t> sub_done; hipEvent_t ev_begin = nullptr;
    hipStream_t graph_stream = nullptr; hipEvent_t ev_graph_done = nullptr;     // wavefront mode's graph capture when the caller's stream is the null stream
};

namespace hrt {

// formats the message into ctx->error (or the creating thread's error when ctx is NULL) and returns code
int fail(HrtContext *ctx, int code, const char *fmt, ...);
const char *create_error();
const char *last_error_of(const HrtContext *ctx);

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(ctx, _e == hipErrorOutOfMemory ? HRT_ERR_OOM : HRT_ERR_HIP,          \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

constexpr int kMaxSubTiles = 8;

struct Timer {
    HrtContext *ctx; hipStream_t s; bool on; TimedSpan span{};
    Timer(HrtContext *c, hipStream_t st, int kind) : ctx(c), s(st), on((c->flags & HRT_CTX_TIMING) != 0) {
        if (!on) { ctx->kernel_launches[kind]++; return; }
        span.kind = kind; span.a = next(); span.b = next();
        (void)hipEventRecord(span.a, s);
    }
    ~Timer() { if (on) { (void)hipEventRecord(span.b, s); ctx->spans.push_back(span); } }
    hipEvent_t next() {
        if (ctx->events_used == ctx->event_pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); ctx->event_pool.push_back(e); }
        return ctx->event_pool[ctx->events_used++];
    }
};

void drain_spans(HrtContext *ctx);

// hrt_accel.cpp
hrt::ScratchArena scratch_acquire(HrtContext *ctx, size_t bytes);      // {nullptr, 0} when the device is out of memory
void scratch_release(HrtContext *ctx, hrt::ScratchArena a);
void free_tlas_device(HrtContext *ctx, Tlas &t);
hipError_t pool_alloc(HrtContext *ctx, void **p, size_t bytes);
void pool_release(HrtContext *ctx, void *p);
void pool_drain(HrtContext *ctx);
void free_tlas_host(Tlas &t);

}  // namespace hrt
