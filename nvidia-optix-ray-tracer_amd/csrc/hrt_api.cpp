// hrt_api.cpp -- C ABI of libhrt.so: context, acceleration structures, materials, RNG, launch.
// Entry points and the reference call sites they replace are documented in include/hrt.h.
// There is deliberately no CPU fallback anywhere in this file: every compute path ends in a
// HIP kernel launch, and context creation fails when no gfx950 device is present.
#include "../../include/hrt.h"
#include "bvh8.h"
#include "bvh8_geom.h"
#include "device_types.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

using namespace hrt;

namespace {

thread_local std::string g_create_error;

struct Blas {
    uint32_t kind = kPrimKindTriangle;
    uint32_t n_prims = 0;
    std::vector<float> verts;        // triangles: 9 floats each (object space)
    std::vector<float> centers;      // spheres: 3 floats each
    std::vector<float> radii;
    float *d_verts = nullptr;        // device copy of verts: the refit re-derives the world-space records from it
                                     // (the caller may free its vertex buffer after the build, RendererMesh.cu:116)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};   // object-space bounds
    // object-space BVH8 of this geometry alone: the per-instance subtree of the trees over instances (built on first use)
    std::mutex tmpl_mu; bool tmpl_built = false; Bvh8 tmpl;
    ~Blas() { if (d_verts) (void)hipFree(d_verts); }
};

struct Tlas {
    uint32_t n_instances = 0;
    std::vector<uint32_t> sbt_offset;      // per instance
    std::vector<uint32_t> kind;            // per instance: triangle / sphere BLAS
    Bvh8 bvh;                               // host copy (download / stats)
    void *d_nodes = nullptr, *d_prims = nullptr;
    float *d_inst_inv = nullptr;
    uint32_t *d_inst_identity = nullptr;
    bool has_spheres = false;
    uint32_t node_stride = 80, prim_stride = 48;
    uint64_t generation = 0;
    // refit (hrt_tlas_update): what must stay the same, and the device tables the refit kernel reads
    std::vector<std::shared_ptr<Blas>> blas_refs;       // keeps the source geometry alive
    std::vector<uint64_t> sig_handle; std::vector<uint32_t> sig_visibility;
    std::vector<float> h_xf, h_inv; std::vector<uint32_t> h_ident;   // staging of the per-instance uploads
    float *d_node_box = nullptr, *d_node_ref = nullptr, *d_inst_xf = nullptr, *d_area = nullptr;
    uint32_t *d_order = nullptr;                         // trees over instances: refit order (NULL: breadth-first index ranges)
    std::vector<std::pair<uint32_t, uint32_t>> phases;   // (first, count) in processing order, children before parents
    bool instanced = false;
    const void **d_inst_src = nullptr;
    float *h_area = nullptr;                             // pinned: area sum of the last refit
    hipEvent_t area_ready = nullptr; bool area_pending = false;
    uint64_t refits = 0, rebuilds = 0;
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
};

struct TimedSpan { int kind; hipEvent_t a, b; };

}  // namespace

struct HrtContext {
    int device = 0;
    uint32_t flags = 0;
    int n_cu = 256;
    std::string error;
    std::mutex mu;
    std::unordered_map<uint64_t, std::shared_ptr<Blas>> blas;
    std::unordered_map<uint64_t, std::unique_ptr<Tlas>> tlas;
    uint64_t next_handle = 0x1000;
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
    std::vector<uint32_t> h_rows; HrtTile rows_tile{0, 0, 0, 0, 0}; uint32_t rows_w = 0, rows_h = 0;
    DeviceStats *d_stats = nullptr;
    uint64_t paths = 0;
    uint64_t last_tlas = 0;
    std::vector<TimedSpan> spans; std::vector<hipEvent_t> event_pool; size_t events_used = 0;
    double kernel_ms[HRT_K_COUNT] = {0}; uint64_t kernel_launches[HRT_K_COUNT] = {0};
    float4 *d_linear = nullptr;
    int refill_threshold = 16;                  // wavefront mode; fused mode: fused_refill_threshold
    int fused_refill_threshold = 24, fused_fetch_chunk = 16;   // measured optimum of the fused path mode (profiles/r01_sweep_fused_*.txt)
    int traverse_blocks_per_cu = 16;            // one-wave workgroups of the traverse kernel per CU
    bool traverse_blocks_auto = true;           // fused mode: fewer of them for small tiles (not when the env knob is set)
    int postpone_pct = 25;
    int tail_split = 1;
    int node_stride = 80, prim_stride = 64;     // bytes between records in HBM (80/48 packed; 128/64 = one cache line each)
    int fused = 1;                              // 1: fused persistent path mode (default), 0: wavefront kernels, -1: fused only for small tiles
    int fused_max_pixels = 700000;
    int fused_max_spp = 512;                    // samples per fused launch
    int lds_gather = 0;                         // 1: cooperative LDS-DMA gathers, 0: per-lane register loads
    int fetch_chunk = 64;
    int substream_min_pixels = 32768;
    int tlas_instanced = 0;                     // 1: hrt_tlas_build makes trees over instances too, 0: only rebuilds during updates do, -1: never
    int refit = 1;                              // hrt_tlas_update: 1 = device refit when only transforms changed, 0 = always rebuild
    float refit_rebuild_ratio = 1.5f;           // rebuild when the refitted tree's weighted mean node area has grown by this factor
    uint64_t tlas_refits = 0, tlas_rebuilds = 0; double tlas_refit_ratio = 1.0;
    int substreams = 0;                         // sub-tiles rendered on their own HIP streams so that one's tail overlaps another's bulk
    std::vector<hipStream_t> sub_streams; std::vector<hipEvent_t> sub_done; hipEvent_t ev_begin = nullptr;
};

namespace {

int fail(HrtContext *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (ctx) ctx->error = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(ctx, _e == hipErrorOutOfMemory ? HRT_ERR_OOM : HRT_ERR_HIP,          \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

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

// ---- transforms (fixed operation order; DESIGN.md "instances") ----
bool is_identity(const float *m) {
    static const float id[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    return std::memcmp(m, id, sizeof id) == 0;
}
void invert_affine(const float *m, float *o) {
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C;
    const double r = 1.0 / det;
    const double n00 = A * r, n01 = -(b * i - c * h) * r, n02 = (b * f - c * e) * r;
    const double n10 = B * r, n11 = (a * i - c * g) * r, n12 = -(a * f - c * d) * r;
    const double n20 = C * r, n21 = -(a * h - b * g) * r, n22 = (a * e - b * d) * r;
    const double tx = m[3], ty = m[7], tz = m[11];
    o[0] = (float)n00; o[1] = (float)n01; o[2] = (float)n02; o[3] = (float)(-(n00 * tx + n01 * ty + n02 * tz));
    o[4] = (float)n10; o[5] = (float)n11; o[6] = (float)n12; o[7] = (float)(-(n10 * tx + n11 * ty + n12 * tz));
    o[8] = (float)n20; o[9] = (float)n21; o[10] = (float)n22; o[11] = (float)(-(n20 * tx + n21 * ty + n22 * tz));
}

void free_tlas_device(Tlas &t) {
    if (t.d_nodes) (void)hipFree(t.d_nodes);
    if (t.d_prims) (void)hipFree(t.d_prims);
    if (t.d_inst_inv) (void)hipFree(t.d_inst_inv);
    if (t.d_inst_identity) (void)hipFree(t.d_inst_identity);
    if (t.d_node_box) (void)hipFree(t.d_node_box);
    if (t.d_node_ref) (void)hipFree(t.d_node_ref);
    if (t.d_order) (void)hipFree(t.d_order);
    if (t.d_inst_xf) (void)hipFree(t.d_inst_xf);
    if (t.d_area) (void)hipFree(t.d_area);
    if (t.d_inst_src) (void)hipFree((void *)t.d_inst_src);
    t.d_nodes = t.d_prims = nullptr; t.d_inst_inv = nullptr; t.d_inst_identity = nullptr;
    t.d_node_box = t.d_node_ref = t.d_inst_xf = t.d_area = nullptr; t.d_inst_src = nullptr; t.d_order = nullptr;
    t.area_pending = false;
}
void free_tlas_host(Tlas &t) {
    if (t.h_area) (void)hipHostFree(t.h_area);
    if (t.area_ready) (void)hipEventDestroy(t.area_ready);
    t.h_area = nullptr; t.area_ready = nullptr;
}

// Per-instance tables of a set of instances: object->world, world->object, identity flags, and the
// largest |coordinate| of the transformed BLAS boxes (what the padding of the tree is derived from).
float instance_tables(const std::vector<HrtInstance> &inst, const std::vector<std::shared_ptr<Blas>> &blas,
                      std::vector<float> &xf, std::vector<float> &inv, std::vector<uint32_t> &ident) {
    const size_t n = inst.size();
    xf.assign(12 * std::max<size_t>(n, 1), 0.0f); inv.assign(12 * std::max<size_t>(n, 1), 0.0f); ident.assign(std::max<size_t>(n, 1), 1u);
    float smax = 1.0f;
    for (size_t i = 0; i < n; ++i) {
        const float *m = inst[i].transform;
        std::memcpy(&xf[12 * i], m, 12 * sizeof(float));
        const bool id = is_identity(m);
        ident[i] = id ? 1u : 0u;
        invert_affine(m, &inv[12 * i]);
        const Blas &b = *blas[i];
        if ((inst[i].visibilityMask & 1u) == 0 || !(b.lo[0] <= b.hi[0])) continue;
        for (int c = 0; c < 8; ++c) {
            const float q[3] = {(c & 1) ? b.hi[0] : b.lo[0], (c & 2) ? b.hi[1] : b.lo[1], (c & 4) ? b.hi[2] : b.lo[2]};
            float w[3];
            if (id) { w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; } else xf_point(m, q, w);
            for (int a = 0; a < 3; ++a) if (std::isfinite(w[a])) smax = std::max(smax, std::fabs(w[a]));
        }
    }
    return smax;
}

void launch_refit_phases(RefitArgs ra, const std::vector<std::pair<uint32_t, uint32_t>> &phases, hipStream_t s);

// Object-space BVH8 of one BLAS (built once): the subtree every instance of it gets in a tree over instances.
void ensure_template(Blas &b) {
    std::lock_guard<std::mutex> lk(b.tmpl_mu);
    if (b.tmpl_built) return;
    std::vector<BuildPrim> prims;
    prims.reserve(b.n_prims);
    for (uint32_t p = 0; p < b.n_prims; ++p) {
        BuildPrim bp; std::memset(&bp, 0, sizeof bp);
        if (b.kind == kPrimKindTriangle) {
            triangle_world(&b.verts[9 * (size_t)p], nullptr, true, bp.rec.a, bp.rec.b, bp.rec.c, bp.lo, bp.hi);
            bp.rec.kind = kPrimKindTriangle;
        } else {
            const float *c = &b.centers[3 * (size_t)p];
            bp.rec.a[0] = c[0]; bp.rec.a[1] = c[1]; bp.rec.a[2] = c[2]; bp.rec.b[0] = b.radii[p]; bp.rec.kind = kPrimKindSphere;
            sphere_world_bounds(c, b.radii[p], nullptr, true, bp.lo, bp.hi);
        }
        bp.rec.prim = p;
        if (finite_box(bp.lo, bp.hi)) prims.push_back(bp);
    }
    build_bvh8(prims, b.tmpl, 0);
    b.tmpl_built = true;
}

// Build a TLAS on the host and upload it together with the tables the device refit needs (hrt_tlas_update).
//  * merged (default of hrt_tlas_build): every instance is flattened into world space and ONE tree is built over all
//    primitives -- the best tree, at the price of a full SAH build;
//  * instanced: a top tree over the instances' boxes whose leaves are per-instance copies of object-space template
//    trees; only the topology comes from the host (milliseconds for thousands of instances), the device refit
//    computes every box and world-space record.  The shape the reference's own scenes have (particles instancing a
//    few shapes); used when a refitted tree has degraded and has to be rebuilt while frames are being rendered.
// Either way the result is one world-space BVH8: the traversal kernels do not know the difference.
int build_tlas_into(HrtContext *ctx, Tlas &t, const std::vector<HrtInstance> &inst, hipStream_t s, bool instanced) {
    const uint32_t n = (uint32_t)inst.size();
    std::vector<std::shared_ptr<Blas>> refs(n);
    size_t total = 0;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        for (uint32_t i = 0; i < n; ++i) {
            auto it = ctx->blas.find(inst[i].traversableHandle);
            if (it == ctx->blas.end()) return fail(ctx, HRT_ERR_INVALID, "instance %u: unknown BLAS handle 0x%llx", i, (unsigned long long)inst[i].traversableHandle);
            refs[i] = it->second;
            total += it->second->n_prims;
        }
    }
    const float scene_scale = instance_tables(inst, refs, t.h_xf, t.h_inv, t.h_ident);
    t.sbt_offset.assign(n, 0); t.kind.assign(n, 0); t.has_spheres = false;
    t.sig_handle.assign(n, 0); t.sig_visibility.assign(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
        t.sbt_offset[i] = inst[i].sbtOffset;
        t.kind[i] = refs[i]->kind;
        t.sig_handle[i] = inst[i].traversableHandle; t.sig_visibility[i] = inst[i].visibilityMask & 1u;
        if ((inst[i].visibilityMask & 1u) != 0 && refs[i]->kind == kPrimKindSphere && refs[i]->n_prims) t.has_spheres = true;
    }
    std::vector<uint32_t> order;
    t.phases.clear();
    if (instanced) {
        std::vector<const Bvh8 *> tmpl(n, nullptr);
        std::vector<float> box(6 * (size_t)std::max(n, 1u), 0.0f);
        for (uint32_t i = 0; i < n; ++i) {
            Blas &b = *refs[i];
            if ((inst[i].visibilityMask & 1u) == 0 || !(b.lo[0] <= b.hi[0])) continue;      // the reference traces with mask 1 (Shader.cu:71)
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int c = 0; c < 8; ++c) {
                const float q[3] = {(c & 1) ? b.hi[0] : b.lo[0], (c & 2) ? b.hi[1] : b.lo[1], (c & 4) ? b.hi[2] : b.lo[2]};
                float w[3];
                if (t.h_ident[i]) { w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; } else xf_point(inst[i].transform, q, w);
                for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], w[a]); hi[a] = std::fmax(hi[a], w[a]); }
            }
            if (!finite_box(lo, hi)) continue;                                                 // a NaN transform: nothing to hit
            ensure_template(b);
            tmpl[i] = &b.tmpl;
            for (int a = 0; a < 3; ++a) { box[6 * (size_t)i + a] = lo[a]; box[6 * (size_t)i + 3 + a] = hi[a]; }
        }
        InstancedTree it;
        assemble_instanced_bvh8(tmpl, box, it);
        t.bvh = Bvh8();
        t.bvh.nodes = std::move(it.nodes); t.bvh.prims = std::move(it.prims);
        t.bvh.n_triangles = it.n_triangles; t.bvh.n_spheres = it.n_spheres; t.bvh.max_depth = it.max_depth;
        t.bvh.node_box.assign(6 * t.bvh.nodes.size(), 0.0f);
        t.bvh.node_ref.assign(2 * t.bvh.nodes.size(), 0.0f);
        for (size_t i = 0; i < it.weight.size(); ++i) t.bvh.node_ref[2 * i] = it.weight[i];
        order = std::move(it.order);
        for (size_t h = 0; h + 1 < it.phase_begin.size(); ++h) t.phases.emplace_back(it.phase_begin[h], it.phase_begin[h + 1] - it.phase_begin[h]);
    } else {
        std::vector<BuildPrim> prims;
        prims.reserve(total);
        for (uint32_t i = 0; i < n; ++i) {
            const Blas *b = refs[i].get();
            const float *m = inst[i].transform;
            const bool id = t.h_ident[i] != 0u;
            if ((inst[i].visibilityMask & 1u) == 0) continue;       // the reference traces with mask 1 (Shader.cu:71)
            for (uint32_t p = 0; p < b->n_prims; ++p) {
                BuildPrim bp; std::memset(&bp, 0, sizeof bp);
                if (b->kind == kPrimKindTriangle) {
                    triangle_world(&b->verts[9 * (size_t)p], m, id, bp.rec.a, bp.rec.b, bp.rec.c, bp.lo, bp.hi);
                    bp.rec.prim = p; bp.rec.inst = i; bp.rec.kind = kPrimKindTriangle;
                } else {
                    const float *c = &b->centers[3 * (size_t)p];
                    bp.rec.a[0] = c[0]; bp.rec.a[1] = c[1]; bp.rec.a[2] = c[2]; bp.rec.prim = p;
                    bp.rec.b[0] = b->radii[p]; bp.rec.inst = i; bp.rec.kind = kPrimKindSphere;
                    sphere_world_bounds(c, b->radii[p], m, id, bp.lo, bp.hi);
                }
                // NaN / Inf geometry never hits anything; keep it out of the tree
                if (!finite_box(bp.lo, bp.hi)) continue;
                prims.push_back(bp);
            }
        }
        build_bvh8(prims, t.bvh, 0, scene_scale);
        for (size_t l = t.bvh.level_begin.size() - 1; l-- > 0;) t.phases.emplace_back(t.bvh.level_begin[l], t.bvh.level_begin[l + 1] - t.bvh.level_begin[l]);
    }
    if (2 * t.bvh.max_depth + 2 > (uint32_t)(8 + 56))
        return fail(ctx, HRT_ERR_INVALID, "BVH depth %u exceeds the traversal stack", t.bvh.max_depth);

    free_tlas_device(t);
    t.n_instances = n;
    t.blas_refs = std::move(refs);
    t.node_stride = (uint32_t)ctx->node_stride; t.prim_stride = (uint32_t)ctx->prim_stride;
    const size_t n_nodes = t.bvh.nodes.size(), n_prims = t.bvh.prims.size();
    const size_t nb = (size_t)t.node_stride * n_nodes;
    const size_t pb = (size_t)t.prim_stride * std::max<size_t>(n_prims, 1);
    std::vector<const void *> src(std::max(n, 1u), nullptr);
    for (uint32_t i = 0; i < n; ++i) src[i] = t.blas_refs[i]->d_verts;
    HIP_TRY(ctx, hipMalloc(&t.d_nodes, nb));
    HIP_TRY(ctx, hipMalloc(&t.d_prims, pb));
    HIP_TRY(ctx, hipMalloc((void **)&t.d_inst_inv, sizeof(float) * t.h_inv.size()));
    HIP_TRY(ctx, hipMalloc((void **)&t.d_inst_xf, sizeof(float) * t.h_xf.size()));
    HIP_TRY(ctx, hipMalloc((void **)&t.d_inst_identity, sizeof(uint32_t) * t.h_ident.size()));
    HIP_TRY(ctx, hipMalloc((void **)&t.d_inst_src, sizeof(void *) * src.size()));
    HIP_TRY(ctx, hipMalloc((void **)&t.d_node_box, sizeof(float) * std::max<size_t>(t.bvh.node_box.size(), 6)));
    HIP_TRY(ctx, hipMalloc((void **)&t.d_node_ref, sizeof(float) * std::max<size_t>(t.bvh.node_ref.size(), 2)));
    HIP_TRY(ctx, hipMalloc((void **)&t.d_area, sizeof(float)));
    if (!order.empty()) HIP_TRY(ctx, hipMalloc((void **)&t.d_order, sizeof(uint32_t) * order.size()));
    if (!t.h_area) HIP_TRY(ctx, hipHostMalloc((void **)&t.h_area, sizeof(float), hipHostMallocDefault));
    if (!t.area_ready) HIP_TRY(ctx, hipEventCreateWithFlags(&t.area_ready, hipEventDisableTiming));
    if (t.node_stride == sizeof(Bvh8Node) && t.prim_stride == sizeof(PrimRecord)) {
        HIP_TRY(ctx, hipMemcpyAsync(t.d_nodes, t.bvh.nodes.data(), nb, hipMemcpyHostToDevice, s));
        if (n_prims) HIP_TRY(ctx, hipMemcpyAsync(t.d_prims, t.bvh.prims.data(), sizeof(PrimRecord) * n_prims, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
    } else {
        std::vector<unsigned char> hn(nb, 0), hp(pb, 0);
        for (size_t i = 0; i < n_nodes; ++i) std::memcpy(&hn[i * t.node_stride], &t.bvh.nodes[i], sizeof(Bvh8Node));
        for (size_t i = 0; i < n_prims; ++i) std::memcpy(&hp[i * t.prim_stride], &t.bvh.prims[i], sizeof(PrimRecord));
        HIP_TRY(ctx, hipMemcpyAsync(t.d_nodes, hn.data(), nb, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(t.d_prims, hp.data(), pb, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
    }
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_inv, t.h_inv.data(), sizeof(float) * t.h_inv.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_xf, t.h_xf.data(), sizeof(float) * t.h_xf.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_identity, t.h_ident.data(), sizeof(uint32_t) * t.h_ident.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync((void *)t.d_inst_src, src.data(), sizeof(void *) * src.size(), hipMemcpyHostToDevice, s));
    if (!t.bvh.node_box.empty())
        HIP_TRY(ctx, hipMemcpyAsync(t.d_node_box, t.bvh.node_box.data(), sizeof(float) * t.bvh.node_box.size(), hipMemcpyHostToDevice, s));
    if (!t.bvh.node_ref.empty())
        HIP_TRY(ctx, hipMemcpyAsync(t.d_node_ref, t.bvh.node_ref.data(), sizeof(float) * t.bvh.node_ref.size(), hipMemcpyHostToDevice, s));
    if (!order.empty()) HIP_TRY(ctx, hipMemcpyAsync(t.d_order, order.data(), sizeof(uint32_t) * order.size(), hipMemcpyHostToDevice, s));
    if (instanced && n_prims) {
        // the device computes what the host left blank: world-space records, boxes, origins, exponents, quantised
        // children, and the built areas the quality guard compares later refits with
        RefitArgs ra{};
        ra.nodes = reinterpret_cast<unsigned char *>(t.d_nodes); ra.node_stride = t.node_stride;
        ra.prims = reinterpret_cast<unsigned char *>(t.d_prims); ra.prim_stride = t.prim_stride;
        ra.node_box = t.d_node_box; ra.node_ref = t.d_node_ref; ra.inst_xf = t.d_inst_xf; ra.inst_identity = t.d_inst_identity;
        ra.inst_src = t.d_inst_src; ra.pad = 4e-6f * std::max(1.0f, scene_scale); ra.order = t.d_order; ra.write_reference = 1u;
        launch_refit_phases(ra, t.phases, s);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    t.instanced = instanced;
    t.generation++;
    t.rebuilds++; ctx->tlas_rebuilds++;
    return HRT_OK;
}

int download_instances(HrtContext *ctx, const HrtInstance *d_instances, uint32_t n, hipStream_t s, std::vector<HrtInstance> &inst) {
    inst.resize(n);
    if (n) {
        HIP_TRY(ctx, hipMemcpyAsync(inst.data(), d_instances, sizeof(HrtInstance) * (size_t)n, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
    }
    return HRT_OK;
}

constexpr int kMaxSubTiles = 8;

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
        if (!st.stages) HIP_TRY(ctx, hipMalloc((void **)&st.stages, sizeof(StageCounters) * (kRayTraceDepth + 1) * kMaxSubTiles));
    return HRT_OK;
}

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

void drain_spans(HrtContext *ctx) {
    for (const TimedSpan &sp : ctx->spans) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) { ctx->kernel_ms[sp.kind] += ms; ctx->kernel_launches[sp.kind]++; }
    }
    ctx->spans.clear();
    ctx->events_used = 0;
}

// The phases of a refit, children before parents: wide phases get a launch each, the narrow ones at the end (the top
// of the tree, or all of a small tree) run in one single-workgroup launch.
void launch_refit_phases(RefitArgs ra, const std::vector<std::pair<uint32_t, uint32_t>> &phases, hipStream_t s) {
    size_t tail = phases.size();
    while (tail > 0 && phases.size() - tail < kRefitTopLevels && phases[tail - 1].second <= kRefitTopLevelNodes) --tail;
    for (size_t i = 0; i < tail; ++i) { ra.first_node = phases[i].first; ra.n_nodes = phases[i].second; launch_refit_level(ra, s); }
    RefitLevels top{};
    for (size_t i = tail; i < phases.size(); ++i) { top.first[top.n_levels] = phases[i].first; top.count[top.n_levels] = phases[i].second; ++top.n_levels; }
    launch_refit_top(ra, top, s);
}

// Device refit of a built tree under new instance transforms: upload the per-instance tables, then one
// k_refit_level launch per tree level, deepest first.  Asynchronous on s.
int refit_tlas(HrtContext *ctx, Tlas &t, const std::vector<HrtInstance> &inst, hipStream_t s) {
    const float scene_scale = instance_tables(inst, t.blas_refs, t.h_xf, t.h_inv, t.h_ident);
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_inv, t.h_inv.data(), sizeof(float) * t.h_inv.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_xf, t.h_xf.data(), sizeof(float) * t.h_xf.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_identity, t.h_ident.data(), sizeof(uint32_t) * t.h_ident.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(t.d_area, 0, sizeof(float), s));
    RefitArgs ra{};
    ra.nodes = reinterpret_cast<unsigned char *>(t.d_nodes); ra.node_stride = t.node_stride;
    ra.prims = reinterpret_cast<unsigned char *>(t.d_prims); ra.prim_stride = t.prim_stride;
    ra.node_box = t.d_node_box; ra.node_ref = t.d_node_ref; ra.inst_xf = t.d_inst_xf; ra.inst_identity = t.d_inst_identity; ra.inst_src = t.d_inst_src;
    ra.pad = 4e-6f * std::max(1.0f, scene_scale);
    ra.area_sum = t.d_area;
    ra.order = t.d_order;
    { Timer tm(ctx, s, HRT_K_REFIT); launch_refit_phases(ra, t.phases, s); }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(t.h_area, t.d_area, sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipEventRecord(t.area_ready, s));
    t.area_pending = true;
    t.refits++; ctx->tlas_refits++;
    return HRT_OK;
}

}  // namespace

// ======================================================================================
extern "C" {

const char *hrt_version(void) { return "hrt 0.1 (gfx950 wavefront path tracer)"; }

const char *hrt_last_error(const HrtContext *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

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
    if (const char *e = std::getenv("HRT_TRAVERSE_BLOCKS_PER_CU")) { const int v = std::atoi(e); if (v >= 1 && v <= 32) { ctx->traverse_blocks_per_cu = v; ctx->traverse_blocks_auto = false; } }
    if (const char *e = std::getenv("HRT_FETCH_CHUNK")) { const int v = std::atoi(e); if (v >= 8 && v <= 4096) { ctx->fetch_chunk = v; ctx->fused_fetch_chunk = v; } }
    if (const char *e = std::getenv("HRT_NODE_STRIDE")) { const int v = std::atoi(e); if (v >= 80 && v <= 256 && v % 16 == 0) ctx->node_stride = v; }
    if (const char *e = std::getenv("HRT_PRIM_STRIDE")) { const int v = std::atoi(e); if (v >= 48 && v <= 256 && v % 16 == 0) ctx->prim_stride = v; }
    if (const char *e = std::getenv("HRT_FUSED")) ctx->fused = std::atoi(e);
    if (const char *e = std::getenv("HRT_TLAS_INSTANCED")) ctx->tlas_instanced = std::atoi(e);
    if (const char *e = std::getenv("HRT_REFIT")) ctx->refit = std::atoi(e) != 0;
    if (const char *e = std::getenv("HRT_REFIT_REBUILD_RATIO")) { const double v = std::atof(e); if (v >= 1.0) ctx->refit_rebuild_ratio = (float)v; }
    if (const char *e = std::getenv("HRT_FUSED_MAX_SPP")) { const int v = std::atoi(e); if (v >= 1) ctx->fused_max_spp = v; }
    if (const char *e = std::getenv("HRT_FUSED_MAX_PIXELS")) { const int v = std::atoi(e); if (v > 0) ctx->fused_max_pixels = v; }
    if (const char *e = std::getenv("HRT_LDS_GATHER")) ctx->lds_gather = std::atoi(e) != 0;
    if (const char *e = std::getenv("HRT_TAIL_SPLIT")) ctx->tail_split = std::atoi(e) != 0;
    if (const char *e = std::getenv("HRT_SUBSTREAM_MIN_PIXELS")) { const int v = std::atoi(e); if (v >= 1024) ctx->substream_min_pixels = v; }
    if (const char *e = std::getenv("HRT_SUBSTREAMS")) { const int v = std::atoi(e); if (v >= 0 && v <= 8) ctx->substreams = v; }
    if (const char *e = std::getenv("HRT_POSTPONE_PCT")) { const int v = std::atoi(e); if (v >= 0 && v <= 100) ctx->postpone_pct = v; }
    if (const char *e = std::getenv("HRT_REFILL_THRESHOLD")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) { ctx->refill_threshold = v; ctx->fused_refill_threshold = v; } }
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
    for (auto &kv : ctx->tlas) { free_tlas_device(*kv.second); free_tlas_host(*kv.second); }
    Workspace &w = ctx->ws;
    for (SampleSet &st : w.set) {
        void *sp[] = {st.rays[0], st.rays[1], st.hit_tuvp, st.hit_inst, st.bin_items, st.chain, st.result, st.stages};
        for (void *p : sp) if (p) (void)hipFree(p);
    }
    void *ptrs[] = {w.accum, w.rows, ctx->d_jump, ctx->d_stats, ctx->d_hitgroups, ctx->d_inst_program};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->sub_done) (void)hipEventDestroy(e);
    for (hipStream_t st : ctx->sub_streams) (void)hipStreamDestroy(st);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    delete ctx;
    return HRT_OK;
}

// ---- acceleration structures ---------------------------------------------------------
int hrt_blas_build_triangles(HrtContext *ctx, const HrtFloat3 *d_vertices, uint32_t n_vertices, void *stream, HrtTraversable *out_blas) {
    if (!ctx || !out_blas) return HRT_ERR_INVALID;
    if (n_vertices % 3 != 0) return fail(ctx, HRT_ERR_INVALID, "n_vertices (%u) is not a multiple of 3", n_vertices);
    if (n_vertices && !d_vertices) return fail(ctx, HRT_ERR_INVALID, "d_vertices is NULL");
    (void)hipSetDevice(ctx->device);
    std::shared_ptr<Blas> b(new Blas());
    b->kind = kPrimKindTriangle; b->n_prims = n_vertices / 3;
    b->verts.resize(3 * (size_t)n_vertices);
    if (n_vertices) {
        const size_t bytes = sizeof(float) * 3 * (size_t)n_vertices;
        HIP_TRY(ctx, hipMalloc((void **)&b->d_verts, bytes));
        HIP_TRY(ctx, hipMemcpyAsync(b->d_verts, d_vertices, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        HIP_TRY(ctx, hipMemcpyAsync(b->verts.data(), d_vertices, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
        HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
        for (size_t v = 0; v < n_vertices; ++v) {
            const float *q = &b->verts[3 * v];
            if (!(std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2]))) continue;
            for (int a = 0; a < 3; ++a) { b->lo[a] = std::fmin(b->lo[a], q[a]); b->hi[a] = std::fmax(b->hi[a], q[a]); }
        }
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    const uint64_t h = ctx->next_handle++;
    ctx->blas[h] = std::move(b);
    *out_blas = h;
    return HRT_OK;
}

int hrt_blas_build_spheres(HrtContext *ctx, const HrtFloat3 *d_centers, const float *d_radii, uint32_t n, void *stream, HrtTraversable *out_blas) {
    if (!ctx || !out_blas) return HRT_ERR_INVALID;
    if (n && (!d_centers || !d_radii)) return fail(ctx, HRT_ERR_INVALID, "sphere arrays are NULL");
    (void)hipSetDevice(ctx->device);
    std::shared_ptr<Blas> b(new Blas());
    b->kind = kPrimKindSphere; b->n_prims = n;
    b->centers.resize(3 * (size_t)n); b->radii.resize(n);
    if (n) {
        HIP_TRY(ctx, hipMemcpyAsync(b->centers.data(), d_centers, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, (hipStream_t)stream));
        HIP_TRY(ctx, hipMemcpyAsync(b->radii.data(), d_radii, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, (hipStream_t)stream));
        HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
        for (size_t p = 0; p < n; ++p) {
            const float *c = &b->centers[3 * p]; const float rr = std::fabs(b->radii[p]);
            if (!(std::isfinite(c[0]) && std::isfinite(c[1]) && std::isfinite(c[2]) && std::isfinite(rr))) continue;
            for (int a = 0; a < 3; ++a) { b->lo[a] = std::fmin(b->lo[a], c[a] - rr); b->hi[a] = std::fmax(b->hi[a], c[a] + rr); }
        }
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    const uint64_t h = ctx->next_handle++;
    ctx->blas[h] = std::move(b);
    *out_blas = h;
    return HRT_OK;
}

int hrt_blas_destroy(HrtContext *ctx, HrtTraversable blas) {
    if (!ctx) return HRT_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ctx->blas.erase(blas) ? HRT_OK : fail(ctx, HRT_ERR_INVALID, "unknown BLAS handle");
}

int hrt_tlas_build(HrtContext *ctx, const HrtInstance *d_instances, uint32_t n, void *stream, HrtTraversable *out_tlas) {
    if (!ctx || !out_tlas) return HRT_ERR_INVALID;
    if (n && !d_instances) return fail(ctx, HRT_ERR_INVALID, "d_instances is NULL");
    (void)hipSetDevice(ctx->device);
    std::unique_ptr<Tlas> t(new Tlas());
    std::vector<HrtInstance> inst;
    int rc = download_instances(ctx, d_instances, n, (hipStream_t)stream, inst);
    if (rc == HRT_OK) rc = build_tlas_into(ctx, *t, inst, (hipStream_t)stream, ctx->tlas_instanced > 0);
    if (rc != HRT_OK) { free_tlas_device(*t); free_tlas_host(*t); return rc; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    const uint64_t h = ctx->next_handle++;
    ctx->tlas[h] = std::move(t);
    *out_tlas = h;
    return HRT_OK;
}

// updateIAS (RendererImpl.cu:210-242): the instance transforms changed.  When nothing else did, the tree
// keeps its topology and is refitted on the device (refit.hip), asynchronously on `stream`; a change of
// BLAS handle / visibility, or a refitted tree that has degraded past refit_rebuild_ratio, rebuilds.
int hrt_tlas_update(HrtContext *ctx, HrtTraversable tlas, const HrtInstance *d_instances, uint32_t n, void *stream) {
    if (!ctx) return HRT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    Tlas *t;
    { std::lock_guard<std::mutex> lk(ctx->mu); auto it = ctx->tlas.find(tlas); if (it == ctx->tlas.end()) return fail(ctx, HRT_ERR_INVALID, "unknown TLAS handle"); t = it->second.get(); }
    if (n != t->n_instances) return fail(ctx, HRT_ERR_INVALID, "update must keep the instance count (%u != %u)", n, t->n_instances);
    if (n && !d_instances) return fail(ctx, HRT_ERR_INVALID, "d_instances is NULL");
    std::vector<HrtInstance> inst;
    const int rc = download_instances(ctx, d_instances, n, s, inst);
    if (rc != HRT_OK) return rc;
    bool same = ctx->refit != 0 && !t->bvh.prims.empty();
    for (uint32_t i = 0; i < n && same; ++i)
        same = inst[i].traversableHandle == t->sig_handle[i] && (inst[i].visibilityMask & 1u) == t->sig_visibility[i];
    if (same && t->area_pending) {
        HIP_TRY(ctx, hipEventSynchronize(t->area_ready));
        t->area_pending = false;
        // a refit keeps the topology: once the boxes have grown this much, a fresh build pays for itself
        ctx->tlas_refit_ratio = (double)*t->h_area;
        if (!(ctx->tlas_refit_ratio <= (double)ctx->refit_rebuild_ratio)) same = false;
    }
    if (same) {
        bool sbt_changed = false;
        for (uint32_t i = 0; i < n; ++i) if (inst[i].sbtOffset != t->sbt_offset[i]) { t->sbt_offset[i] = inst[i].sbtOffset; sbt_changed = true; }
        if (sbt_changed) t->generation++;                 // the material tables are re-derived at the next launch
        return refit_tlas(ctx, *t, inst, s);
    }
    HIP_TRY(ctx, hipDeviceSynchronize());                 // launches on other streams may still read the old tree
    // a rebuild in the middle of an animation: the tree over instances costs milliseconds instead of a full SAH build
    return build_tlas_into(ctx, *t, inst, s, ctx->tlas_instanced >= 0 && n >= 2);
}

int hrt_pose_instances(HrtContext *ctx, HrtInstance *d_instances, uint32_t first_instance, uint32_t n_particles,
                       const HrtParticleState *d_current, const HrtParticleState *d_next, const HrtPoseParams *h_params, void *stream) {
    if (!ctx) return HRT_ERR_INVALID;
    if (n_particles == 0) return HRT_OK;
    if (!d_instances || !d_current || !d_next || !h_params) return fail(ctx, HRT_ERR_INVALID, "hrt_pose_instances: NULL argument");
    if (h_params->frame_count == 0) return fail(ctx, HRT_ERR_INVALID, "hrt_pose_instances: frame_count is 0");
    if ((reinterpret_cast<uintptr_t>(d_instances) & 15u) || (reinterpret_cast<uintptr_t>(d_current) & 15u) || (reinterpret_cast<uintptr_t>(d_next) & 15u))
        return fail(ctx, HRT_ERR_INVALID, "hrt_pose_instances: arrays must be 16-byte aligned");
    (void)hipSetDevice(ctx->device);
    PoseArgs a{};
    a.instances = d_instances; a.first_instance = first_instance; a.n = n_particles;
    a.current = reinterpret_cast<const float4 *>(d_current); a.next = reinterpret_cast<const float4 *>(d_next);
    a.duration = h_params->duration; a.frame = h_params->frame; a.frame_count = h_params->frame_count;
    std::memcpy(a.offset, &h_params->particle_offset, 12); std::memcpy(a.scale, &h_params->particle_scale, 12);
    launch_pose_instances(a, (hipStream_t)stream);
    HIP_TRY(ctx, hipGetLastError());
    return HRT_OK;
}

int hrt_tlas_destroy(HrtContext *ctx, HrtTraversable tlas) {
    if (!ctx) return HRT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->tlas.find(tlas);
    if (it == ctx->tlas.end()) return fail(ctx, HRT_ERR_INVALID, "unknown TLAS handle");
    (void)hipDeviceSynchronize();
    free_tlas_device(*it->second); free_tlas_host(*it->second);
    ctx->tlas.erase(it);
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

    // ---- fused path mode: ONE launch, every lane owns a pixel and runs all its samples (generate,
    //      traverse, shade, accumulate in place).  No stage barriers: the choice for small tiles. ----
    const bool use_fused = !count && (ctx->fused > 0 || (ctx->fused < 0 && n <= (uint32_t)ctx->fused_max_pixels));
    if (use_fused) {
        StageCounters *stg = w.set[0].stages;
        TraverseArgs ta{};
        ta.nodes = t->d_nodes; ta.prims = t->d_prims; ta.node_stride = t->node_stride; ta.prim_stride = t->prim_stride;
        ta.fetch_counter = stg[0].fetch;
        ta.inst_inv = t->d_inst_inv; ta.inst_identity = t->d_inst_identity;
        ta.tmin = kFloatZero; ta.tmax = kFloatInfinity;
        ta.refill_threshold = ctx->fused_refill_threshold; ta.postpone_pct = ctx->postpone_pct; ta.tail_split = 0; ta.fetch_chunk = (uint32_t)ctx->fused_fetch_chunk;
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
        // 172 ms on 12 waves per CU, 210 ms on 16).  A full frame has many pixels per lane and keeps the maximum.
        uint32_t blocks_per_cu = (uint32_t)ctx->traverse_blocks_per_cu;
        if (ctx->traverse_blocks_auto) {
            const uint32_t fit = (uint32_t)((10ull * n + 14ull * 64ull * (uint64_t)ctx->n_cu - 1ull) / (14ull * 64ull * (uint64_t)ctx->n_cu));
            blocks_per_cu = std::min(blocks_per_cu, std::max(fit, 4u));
        }
        const uint32_t grid = std::min<uint32_t>((uint32_t)ctx->n_cu * blocks_per_cu, (n + 63u) / 64u);
        // very long renders are cut into launches of at most fused_max_spp samples (a launch should stay in the
        // range of seconds); the RNG states and the running sums carry over, so the result is the same bits
        for (uint32_t done_spp = 0; done_spp < spp;) {
            const uint32_t now = std::min<uint32_t>(spp - done_spp, (uint32_t)ctx->fused_max_spp);
            pa.spp = now; pa.continue_sum = done_spp > 0 ? 1u : 0u;
            HIP_TRY(ctx, hipMemsetAsync(stg, 0, sizeof(StageCounters), s));
            { Timer tm(ctx, s, HRT_K_PATHS); launch_paths(ta, t->has_spheres, grid, s); }
            done_spp += now;
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
    std::vector<Sub> subs(S);
    for (uint32_t k = 0; k < S; ++k) {
        const uint32_t j0 = (uint32_t)((uint64_t)n * k / S), j1 = (uint32_t)((uint64_t)n * (k + 1) / S);
        subs[k].j0 = j0; subs[k].n = j1 - j0; subs[k].st = S > 1 ? ctx->sub_streams[k] : s; subs[k].index = k;
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
        ta.refill_threshold = ctx->refill_threshold; ta.postpone_pct = ctx->postpone_pct; ta.tail_split = ctx->tail_split; ta.fetch_chunk = (uint32_t)ctx->fetch_chunk;
        Timer tm(ctx, sb.st, (da >= kRayTraceDepth && !second) ? HRT_K_TRAVERSE_ANY : HRT_K_TRAVERSE);
        launch_traverse(ta, count, t->has_spheres, ctx->lds_gather != 0, sb.grid_trav, sb.st);
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
    for (uint32_t sample = 0; sample < spp; ++sample) {
        const bool prev = sample > 0, next = sample + 1 < spp;
        for (const Sub &sb : subs) do_after(sb, sample, 1);
        for (const Sub &sb : subs) do_traverse(sb, sample, 2, prev, sample - 1, kRayTraceDepth);
        if (prev) for (const Sub &sb : subs) do_after(sb, sample - 1, kRayTraceDepth);
        for (const Sub &sb : subs) do_after(sb, sample, 2);
        for (const Sub &sb : subs) do_traverse(sb, sample, 3, false, 0, 0);
        for (const Sub &sb : subs) do_after(sb, sample, 3);
        if (next) for (const Sub &sb : subs) { rc = do_generate(sample + 1, sb); if (rc != HRT_OK) return rc; }
        for (const Sub &sb : subs) do_traverse(sb, sample, 4, next, sample + 1, 1);
        for (const Sub &sb : subs) do_after(sb, sample, 4);
    }
    for (const Sub &sb : subs) do_traverse(sb, spp - 1, kRayTraceDepth, false, 0, 0);
    for (const Sub &sb : subs) do_after(sb, spp - 1, kRayTraceDepth);
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
        const Bvh8 &b = it->second->bvh;
        out->bvh_nodes = b.nodes.size(); out->bvh_triangles = b.n_triangles; out->bvh_spheres = b.n_spheres;
        out->bvh_bytes = b.nodes.size() * sizeof(Bvh8Node) + b.prims.size() * sizeof(PrimRecord);
    }
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
    HIP_TRY(ctx, hipMalloc((void **)&rays, sizeof(RayRec) * (size_t)n_rays));
    HIP_TRY(ctx, hipMalloc((void **)&tuvp, sizeof(float4) * (size_t)n_rays));
    HIP_TRY(ctx, hipMalloc((void **)&inst, sizeof(uint32_t) * (size_t)n_rays));
    HIP_TRY(ctx, hipMalloc((void **)&fetch, sizeof(uint32_t) * 8 * 32));
    HIP_TRY(ctx, hipMemsetAsync(fetch, 0, sizeof(uint32_t) * 8 * 32, s));
    launch_pack_rays(reinterpret_cast<const float *>(d_origins), reinterpret_cast<const float *>(d_directions), n_rays, rays, s);
    TraverseArgs ta{};
    ta.nodes = t->d_nodes; ta.prims = t->d_prims; ta.node_stride = t->node_stride; ta.prim_stride = t->prim_stride;
    ta.seg[0].rays = rays; ta.seg[0].n_ptr = nullptr; ta.seg[0].n = n_rays; ta.seg[0].any_hit = any_hit ? 1u : 0u;
    ta.seg[0].hit_tuvp = tuvp; ta.seg[0].hit_inst = inst;
    ta.seg[0].count_nodes = any_hit ? &ctx->d_stats->nodes_any : &ctx->d_stats->nodes_closest;
    ta.seg[0].count_prims = any_hit ? &ctx->d_stats->prims_any : &ctx->d_stats->prims_closest;
    ta.fetch_counter = fetch; ta.inst_inv = t->d_inst_inv; ta.inst_identity = t->d_inst_identity;
    ta.tmin = tmin; ta.tmax = tmax; ta.refill_threshold = ctx->refill_threshold; ta.postpone_pct = ctx->postpone_pct;
    ta.tail_split = ctx->tail_split; ta.fetch_chunk = (uint32_t)ctx->fetch_chunk;
    const uint32_t grid = std::min<uint32_t>((uint32_t)ctx->n_cu * (uint32_t)ctx->traverse_blocks_per_cu, (n_rays + 63u) / 64u);
    { Timer tm(ctx, s, any_hit ? HRT_K_TRAVERSE_ANY : HRT_K_TRAVERSE);
      launch_traverse(ta, (ctx->flags & HRT_CTX_COUNT) != 0, t->has_spheres, ctx->lds_gather != 0, grid, s); }
    launch_unpack_hits(tuvp, inst, n_rays, d_t, d_u, d_v, d_prim, d_inst, s);
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(rays); (void)hipFree(tuvp); (void)hipFree(inst); (void)hipFree(fetch);
    if (e != hipSuccess) return fail(ctx, HRT_ERR_HIP, "hrt_trace_rays: %s", hipGetErrorString(e));
    ctx->last_tlas = tlas;
    return HRT_OK;
}

int hrt_debug_set_linear_output(HrtContext *ctx, HrtFloat4 *d_linear) {
    if (!ctx) return HRT_ERR_INVALID;
    ctx->d_linear = reinterpret_cast<float4 *>(d_linear);
    return HRT_OK;
}

static int fill_blob(const Bvh8 &b, HrtBvhBlob *out) {
    std::memset(out, 0, sizeof *out);
    out->n_nodes = b.nodes.size(); out->n_triangles = b.prims.size();
    out->nodes = std::malloc(std::max<size_t>(1, sizeof(Bvh8Node) * b.nodes.size()));
    out->triangles = std::malloc(std::max<size_t>(1, sizeof(PrimRecord) * b.prims.size()));
    if (!out->nodes || !out->triangles) { std::free(out->nodes); std::free(out->triangles); std::memset(out, 0, sizeof *out); return HRT_ERR_OOM; }
    std::memcpy(out->nodes, b.nodes.data(), sizeof(Bvh8Node) * b.nodes.size());
    std::memcpy(out->triangles, b.prims.data(), sizeof(PrimRecord) * b.prims.size());
    for (int a = 0; a < 3; ++a) { out->bounds[a] = b.lo[a]; out->bounds[3 + a] = b.hi[a]; }
    return HRT_OK;
}

int hrt_host_build_bvh8(const float *h_triangles, uint32_t n_triangles, HrtBvhBlob *out) {
    if (!out || (n_triangles && !h_triangles)) return HRT_ERR_INVALID;
    std::vector<BuildPrim> prims(n_triangles);
    for (uint32_t p = 0; p < n_triangles; ++p) {
        BuildPrim &bp = prims[p]; std::memset(&bp, 0, sizeof bp);
        const float *v = h_triangles + 9 * (size_t)p;
        for (int a = 0; a < 3; ++a) {
            bp.rec.a[a] = v[a]; bp.rec.b[a] = v[3 + a] - v[a]; bp.rec.c[a] = v[6 + a] - v[a];
            bp.lo[a] = std::fmin(v[a], std::fmin(v[3 + a], v[6 + a]));
            bp.hi[a] = std::fmax(v[a], std::fmax(v[3 + a], v[6 + a]));
        }
        bp.rec.prim = p; bp.rec.inst = 0; bp.rec.kind = kPrimKindTriangle;
    }
    Bvh8 b;
    build_bvh8(prims, b, 0);
    const char *err = validate_bvh8(b);
    if (err[0]) { g_create_error = std::string("bvh8 validation: ") + err; return HRT_ERR_STATE; }
    return fill_blob(b, out);
}

int hrt_tlas_download(HrtContext *ctx, HrtTraversable tlas, HrtBvhBlob *out) {
    if (!ctx || !out) return HRT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    Tlas *t;
    { std::lock_guard<std::mutex> lk(ctx->mu); auto it = ctx->tlas.find(tlas); if (it == ctx->tlas.end()) return fail(ctx, HRT_ERR_INVALID, "unknown TLAS handle"); t = it->second.get(); }
    const int rc = fill_blob(t->bvh, out);
    if (rc != HRT_OK) return rc;
    // the device copy is the truth: a refit rewrites it in place
    HIP_TRY(ctx, hipDeviceSynchronize());
    const size_t n_nodes = t->bvh.nodes.size(), n_prims = t->bvh.prims.size();
    HIP_TRY(ctx, hipMemcpy2D(out->nodes, sizeof(Bvh8Node), t->d_nodes, t->node_stride, sizeof(Bvh8Node), n_nodes, hipMemcpyDeviceToHost));
    if (n_prims) HIP_TRY(ctx, hipMemcpy2D(out->triangles, sizeof(PrimRecord), t->d_prims, t->prim_stride, sizeof(PrimRecord), n_prims, hipMemcpyDeviceToHost));
    return HRT_OK;
}

void hrt_host_free(HrtBvhBlob *blob) {
    if (!blob) return;
    std::free(blob->nodes); std::free(blob->triangles);
    std::memset(blob, 0, sizeof *blob);
}

}  // extern "C"
