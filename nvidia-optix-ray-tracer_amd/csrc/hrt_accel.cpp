// hrt_accel.cpp -- acceleration structures behind the C ABI: BLAS objects, the merged world-space build, trees over
// instances, the per-frame device refit (updateIAS), the Time-mode pose kernel's entry point, and the download /
// host-build helpers the tests use.  Entry points and the reference call sites they replace: include/hrt.h.
#include <thread>
#include "hrt_internal.hpp"
#include "build.h"

namespace hrt {

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

// Working memory of the device builds.  A build takes the smallest free arena that is large enough (or allocates one, a quarter
// larger than asked for) and gives it back when it is done; the context keeps up to eight of them, so loader threads building
// side by side each find one.  Without this a 1500-triangle rebuild spent most of its 2 ms in hipMalloc / hipFree.
ScratchArena scratch_acquire(HrtContext *ctx, size_t bytes) {
    {
        std::lock_guard<std::mutex> lk(ctx->scratch_mu);
        int best = -1;
        for (int i = 0; i < (int)ctx->scratch_free.size(); ++i)
            if (ctx->scratch_free[i].bytes >= bytes && (best < 0 || ctx->scratch_free[i].bytes < ctx->scratch_free[best].bytes)) best = i;
        if (best >= 0) { const ScratchArena a = ctx->scratch_free[best]; ctx->scratch_free.erase(ctx->scratch_free.begin() + best); return a; }
    }
    ScratchArena a;
    a.bytes = bytes + bytes / 4 + 4096;
    if (hipMalloc(&a.p, a.bytes) != hipSuccess) {
        // out of memory: what the context keeps for later -- the trees' cached blocks AND the other arenas (up to eight, each too small
        // for this request or it would have been taken above) -- goes back to the runtime first, then exactly what was asked for
        (void)hipGetLastError();
        pool_drain(ctx);
        std::vector<ScratchArena> arenas;
        { std::lock_guard<std::mutex> lk(ctx->scratch_mu); arenas.swap(ctx->scratch_free); }
        for (const ScratchArena &x : arenas) (void)hipFree(x.p);
        a.bytes = bytes;
        if (hipMalloc(&a.p, a.bytes) != hipSuccess) { (void)hipGetLastError(); a = ScratchArena(); }
    }
    return a;
}
void scratch_release(HrtContext *ctx, ScratchArena a) {
    if (!a.p) return;
    if (a.bytes > ((size_t)2 << 30)) { (void)hipFree(a.p); return; }      // the working memory of a very large build is not kept
    ScratchArena drop;
    {
        std::lock_guard<std::mutex> lk(ctx->scratch_mu);
        ctx->scratch_free.push_back(a);
        if (ctx->scratch_free.size() > 8) {                // keep the large ones
            size_t smallest = 0;
            for (size_t i = 1; i < ctx->scratch_free.size(); ++i) if (ctx->scratch_free[i].bytes < ctx->scratch_free[smallest].bytes) smallest = i;
            drop = ctx->scratch_free[smallest];
            ctx->scratch_free.erase(ctx->scratch_free.begin() + (long)smallest);
        }
    }
    if (drop.p) (void)hipFree(drop.p);
}

// Device memory of the trees (nodes, records, a dozen small per-instance tables), kept by the context when a tree is freed and handed
// out again to the next build: an update that rebuilds (every file's first frame in the reference's Time mode) otherwise spends more
// time in ~20 hipMalloc / hipFree pairs -- each hipFree waits for the device -- than in its kernels.  Eight size classes per octave;
// blocks above 64 MiB and anything beyond 1 GiB in all go back to the runtime.
void pool_drain(HrtContext *ctx);
static size_t pool_class(size_t bytes) {
    bytes = std::max<size_t>(bytes, 256);
    size_t p2 = 256;
    while (p2 * 2 <= bytes) p2 *= 2;
    const size_t step = p2 / 8;
    return (bytes + step - 1) / step * step;
}
hipError_t pool_alloc(HrtContext *ctx, void **p, size_t bytes) {
    const size_t cls = pool_class(bytes);
    {
        std::lock_guard<std::mutex> lk(ctx->pool_mu);
        for (size_t i = 0; i < ctx->pool_free.size(); ++i)
            if (ctx->pool_free[i].bytes == cls) {
                *p = ctx->pool_free[i].p;
                ctx->pool_free.erase(ctx->pool_free.begin() + (long)i);
                ctx->pool_bytes -= cls; ctx->pool_live[*p] = cls;
                return hipSuccess;
            }
    }
    hipError_t e = hipMalloc(p, cls);
    if (e != hipSuccess) {          // out of memory: what the pool and the builds' arenas keep goes back to the runtime first
        (void)hipGetLastError();
        pool_drain(ctx);
        std::vector<ScratchArena> arenas;
        { std::lock_guard<std::mutex> lk(ctx->scratch_mu); arenas.swap(ctx->scratch_free); }
        for (const ScratchArena &a : arenas) (void)hipFree(a.p);
        e = hipMalloc(p, cls);
    }
    if (e == hipSuccess) { std::lock_guard<std::mutex> lk(ctx->pool_mu); ctx->pool_live[*p] = cls; }
    return e;
}
// (the caller has made sure the device is done with the block)
void pool_release(HrtContext *ctx, void *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(ctx->pool_mu);
        const auto it = ctx->pool_live.find(p);
        if (it != ctx->pool_live.end()) {
            const size_t bytes = it->second;
            ctx->pool_live.erase(it);
            if (bytes <= ((size_t)64 << 20) && ctx->pool_bytes + bytes <= ((size_t)1 << 30) && ctx->pool_free.size() < 512) {
                ctx->pool_free.push_back({p, bytes}); ctx->pool_bytes += bytes;
                return;
            }
        }
    }
    (void)hipFree(p);
}
void pool_drain(HrtContext *ctx) {
    std::vector<ScratchArena> blocks;
    { std::lock_guard<std::mutex> lk(ctx->pool_mu); blocks.swap(ctx->pool_free); ctx->pool_bytes = 0; }
    for (const ScratchArena &b : blocks) (void)hipFree(b.p);
}

void free_tlas_device(HrtContext *ctx, Tlas &t) {
    if (t.d_nodes || t.d_prims || t.d_inst_inv || t.d_inst_xf) (void)hipDeviceSynchronize();      // (what each hipFree used to do; the blocks go to the context's pool)
    pool_release(ctx, (void *)t.d_nodes);
    pool_release(ctx, (void *)t.d_prims);
    pool_release(ctx, (void *)t.d_inst_inv);
    pool_release(ctx, (void *)t.d_inst_identity);
    pool_release(ctx, (void *)t.d_node_box);
    pool_release(ctx, (void *)t.d_node_ref);
    pool_release(ctx, (void *)t.d_order);
    pool_release(ctx, (void *)t.d_inst_xf);
    pool_release(ctx, (void *)t.d_area);
    pool_release(ctx, (void *)t.d_inst_src);
    pool_release(ctx, (void *)t.d_inst_first);
    pool_release(ctx, (void *)t.d_inst_kind);
    pool_release(ctx, (void *)t.d_inst_root); pool_release(ctx, (void *)t.d_blas_bound); t.d_blas_bound = nullptr;
    t.d_inst_first = t.d_inst_kind = t.d_inst_root = nullptr;
    pool_release(ctx, (void *)t.d_sig_handle);
    pool_release(ctx, (void *)t.d_sig_visibility);
    pool_release(ctx, (void *)t.d_sig_sbt);
    pool_release(ctx, (void *)t.d_blas_box);
    pool_release(ctx, (void *)t.d_update_flags);
    pool_release(ctx, (void *)t.d_rec_box); t.d_rec_box = nullptr;
    t.d_sig_handle = nullptr; t.d_sig_visibility = nullptr; t.d_sig_sbt = nullptr; t.d_blas_box = nullptr; t.d_update_flags = nullptr;
    t.d_nodes = t.d_prims = nullptr; t.d_inst_inv = nullptr; t.d_inst_identity = nullptr;
    t.d_node_box = t.d_node_ref = t.d_inst_xf = t.d_area = nullptr; t.d_inst_src = nullptr; t.d_order = nullptr;
    t.area_pending = false;
}
void free_tlas_host(Tlas &t) {
    if (t.h_area) (void)hipHostFree(t.h_area);
    if (t.h_update_flags) (void)hipHostFree(t.h_update_flags);
    t.h_update_flags = nullptr;
    if (t.area_ready) (void)hipEventDestroy(t.area_ready);
    t.h_area = nullptr; t.area_ready = nullptr;
}

int g_instance_table_threads = 8;      // HRT_TABLE_THREADS

// Per-instance tables of a set of instances: object->world, world->object, identity flags, and the
// largest |coordinate| of the transformed BLAS boxes (what the padding of the tree is derived from).
float instance_tables(const std::vector<HrtInstance> &inst, const std::vector<std::shared_ptr<Blas>> &blas,
                      std::vector<float> &xf, std::vector<float> &inv, std::vector<uint32_t> &ident) {
    const size_t n = inst.size();
    xf.assign(12 * std::max<size_t>(n, 1), 0.0f); inv.assign(12 * std::max<size_t>(n, 1), 0.0f); ident.assign(std::max<size_t>(n, 1), 1u);
    const auto range = [&](size_t i0, size_t i1) {
        float smax = 1.0f;
        for (size_t i = i0; i < i1; ++i) {
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
    };
    // (a DEM time step has 10^5 instances and a synchronous hrt_tlas_update derives these on the host in every frame: a few threads then --
    // every instance writes its own entries, the largest coordinate is a maximum: the result does not depend on the split)
    const unsigned hw = std::min((unsigned)std::max(g_instance_table_threads, 1), std::max(1u, std::thread::hardware_concurrency()));
    if (n < 32768u || hw < 2u) return range(0, n);
    std::vector<float> part(hw, 1.0f);
    std::vector<std::thread> pool;
    for (unsigned k = 1; k < hw; ++k) pool.emplace_back([&, k] { part[k] = range(n * k / hw, n * (k + 1) / hw); });
    part[0] = range(0, n / hw);
    for (std::thread &th : pool) th.join();
    return *std::max_element(part.begin(), part.end());
}

void launch_refit_phases(RefitArgs ra, const std::vector<std::pair<uint32_t, uint32_t>> &phases, hipStream_t s);
static void attach_rec_box(HrtContext *ctx, Tlas &t, RefitArgs &ra, uint32_t n_records);

// The host builder needs the geometry on the host: fetched from the BLAS's device copy the first time it is asked for
// (HRT_BUILD=host only; the device build never brings geometry across the bus).
int ensure_host_geometry(HrtContext *ctx, Blas &b, hipStream_t s) {
    std::lock_guard<std::mutex> lk(b.tmpl_mu);
    if (b.host_geometry || b.n_prims == 0) { b.host_geometry = true; return HRT_OK; }
    if (b.kind == kPrimKindTriangle) {
        b.verts.resize(9 * (size_t)b.n_prims);
        HIP_TRY(ctx, hipMemcpyAsync(b.verts.data(), b.d_verts, sizeof(float) * b.verts.size(), hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
    } else {
        std::vector<float> cr(4 * (size_t)b.n_prims);
        HIP_TRY(ctx, hipMemcpyAsync(cr.data(), b.d_verts, sizeof(float) * cr.size(), hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
        b.centers.resize(3 * (size_t)b.n_prims); b.radii.resize(b.n_prims);
        for (size_t p = 0; p < b.n_prims; ++p) { for (int a = 0; a < 3; ++a) b.centers[3 * p + a] = cr[4 * p + a]; b.radii[p] = cr[4 * p + 3]; }
    }
    b.host_geometry = true;
    return HRT_OK;
}

// Object-space BVH8 of one BLAS (built once): the subtree every instance of it gets in a tree over instances.  Built on the
// device like everything else (one identity instance); only its topology -- nodes' child / primitive bases, masks, the
// primitive ids -- comes back to the host, where assemble_instanced_bvh8 stitches instance subtrees under a top tree.
// keep_device: the topology also stays on the device (Blas::d_tmpl_*), for two-level TLASes to copy from
static int template_to_device(HrtContext *ctx, Blas &b, hipStream_t s) {
    if (b.d_tmpl_nodes || b.tmpl.nodes.empty()) return HRT_OK;
    HIP_TRY(ctx, hipMalloc((void **)&b.d_tmpl_nodes, sizeof(Bvh8Node) * b.tmpl.nodes.size()));
    HIP_TRY(ctx, hipMalloc((void **)&b.d_tmpl_prims, sizeof(PrimRecord) * std::max<size_t>(b.tmpl.prims.size(), 1)));
    HIP_TRY(ctx, hipMemcpyAsync(b.d_tmpl_nodes, b.tmpl.nodes.data(), sizeof(Bvh8Node) * b.tmpl.nodes.size(), hipMemcpyHostToDevice, s));
    if (!b.tmpl.prims.empty()) HIP_TRY(ctx, hipMemcpyAsync(b.d_tmpl_prims, b.tmpl.prims.data(), sizeof(PrimRecord) * b.tmpl.prims.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return HRT_OK;
}
int ensure_template(HrtContext *ctx, Blas &b, hipStream_t s, bool keep_device = false) {
    if (!ctx->build_on_device) {
        const int rc = ensure_host_geometry(ctx, b, s);
        if (rc != HRT_OK) return rc;
    }
    std::lock_guard<std::mutex> lk(b.tmpl_mu);
    if (b.tmpl_built) return keep_device ? template_to_device(ctx, b, s) : HRT_OK;
    if (!ctx->build_on_device) {
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
        return keep_device ? template_to_device(ctx, b, s) : HRT_OK;
    }
    b.tmpl = Bvh8();
    const uint32_t n = b.n_prims;
    if (n == 0) { build_bvh8({}, b.tmpl, 1); b.tmpl_built = true; return HRT_OK; }
    // one block of the context's working memory holds the five small tables, the staged output and the build's own arrays: a template
    // build allocates and frees nothing (eight hipMalloc / hipFree pairs used to cost more than the build)
    const uint32_t h_first[2] = {0u, n}, h_kind = b.kind, h_ident = 1u;
    const float h_xf[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    const void *h_src = b.d_verts;
    auto up = [](size_t x) { return (x + 255u) & ~(size_t)255u; };
    const size_t o_first = 0, o_kind = o_first + 256, o_src = o_kind + 256, o_xf = o_src + 256, o_ident = o_xf + 256;
    const size_t o_nodes = o_ident + 256, o_prims = o_nodes + up(sizeof(Bvh8Node) * (size_t)n), o_ref = o_prims + up(sizeof(PrimRecord) * (size_t)n);
    const size_t o_scratch = o_ref + up(sizeof(float) * 2 * (size_t)n);
    const ScratchArena arena = scratch_acquire(ctx, o_scratch + gpu_build_scratch_bytes(n));
    if (!arena.p) return fail(ctx, HRT_ERR_OOM, "device build of a BLAS template: no working memory");
    // (on an error path kernels may still be writing to the arena: another loader thread must not be handed it before they are done)
    struct Release { HrtContext *c; ScratchArena a; hipStream_t st; ~Release() { (void)hipStreamSynchronize(st); scratch_release(c, a); } } release{ctx, arena, s};
    unsigned char *base = static_cast<unsigned char *>(arena.p);
    unsigned char *d_nodes = base + o_nodes, *d_prims = base + o_prims;
    HIP_TRY(ctx, hipMemcpyAsync(base + o_first, h_first, sizeof h_first, hipMemcpyHostToDevice, s)); HIP_TRY(ctx, hipMemcpyAsync(base + o_kind, &h_kind, 4, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_src, &h_src, sizeof(void *), hipMemcpyHostToDevice, s)); HIP_TRY(ctx, hipMemcpyAsync(base + o_xf, h_xf, sizeof h_xf, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_ident, &h_ident, 4, hipMemcpyHostToDevice, s));
    GpuBuildInput in{};
    in.n_prims = n; in.n_inst = 1;
    in.d_inst_first = reinterpret_cast<const uint32_t *>(base + o_first); in.d_inst_kind = reinterpret_cast<const uint32_t *>(base + o_kind);
    in.d_inst_src = reinterpret_cast<const void *const *>(base + o_src); in.d_inst_xf = reinterpret_cast<const float *>(base + o_xf);
    in.d_inst_identity = reinterpret_cast<const uint32_t *>(base + o_ident);
    in.max_leaf_prims = kMaxLeafPrims; in.width = (uint32_t)ctx->build_width; in.c_node = ctx->build_c_node; in.c_prim = ctx->build_c_prim_bodies; in.ploc_radius = ctx->ploc_radius; in.quant_guard = ctx->quant_guard;
    in.out_nodes = d_nodes; in.node_stride = sizeof(Bvh8Node); in.out_prims = d_prims; in.prim_stride = sizeof(PrimRecord); in.out_node_ref = reinterpret_cast<float *>(base + o_ref);
    in.scratch = base + o_scratch; in.scratch_bytes = arena.bytes - o_scratch;
    const GpuBuildResult r = gpu_build_bvh8(in, s);      // (synchronises the stream before it returns)
    if (r.error != hipSuccess) return fail(ctx, HRT_ERR_HIP, "device build of a BLAS template failed: %s (%s)", hipGetErrorString(r.error), r.where);
    if (r.n_prims == 0) { build_bvh8({}, b.tmpl, 1); b.tmpl_built = true; return HRT_OK; }
    b.tmpl.nodes.resize(r.n_nodes); b.tmpl.prims.resize(r.n_prims);
    HIP_TRY(ctx, hipMemcpyAsync(b.tmpl.nodes.data(), d_nodes, sizeof(Bvh8Node) * (size_t)r.n_nodes, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(b.tmpl.prims.data(), d_prims, sizeof(PrimRecord) * (size_t)r.n_prims, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    b.tmpl.max_depth = r.max_depth; b.tmpl.level_begin = r.level_begin;
    if (b.kind == kPrimKindTriangle) b.tmpl.n_triangles = r.n_prims; else b.tmpl.n_spheres = r.n_prims;
    b.tmpl_built = true;
    return keep_device ? template_to_device(ctx, b, s) : HRT_OK;
}

// Build a TLAS on the host and upload it together with the tables the device refit needs (hrt_tlas_update).
//  * merged (default of hrt_tlas_build): every instance is flattened into world space and ONE tree is built over all
//    primitives -- the best tree, at the price of a full SAH build;
//  * instanced: a top tree over the instances' boxes whose leaves are per-instance copies of object-space template
//    trees; only the topology comes from the host (milliseconds for thousands of instances), the device refit
//    computes every box and world-space record.  The shape the reference's own scenes have (particles instancing a
//    few shapes); used when a refitted tree has degraded and has to be rebuilt while frames are being rendered.
// Either way the result is one world-space BVH8: the traversal kernels do not know the difference.
constexpr int kRetryFlattened = 1;      // build_tlas_fresh: the two-level tree asked for cannot be had (too deep for the path kernel's stack): build the flattened one

// ---- the merged device build (the body of build_tlas_fresh for it): every visible instance's primitives in ONE world-space tree -- topology and
//      primitive ids from build.hip (the top-down SAH phase, with spatial splits under HRT_CTX_FAST_TRACE, then PLOC in the cells, the optimal
//      collapse, the emission); the refit computes every record, box and quantised child, exactly as after an instance update ----
static int build_merged_on_device(HrtContext *ctx, Tlas &t, const std::vector<uint32_t> &first, bool device_split, bool scene_of_bodies, uint32_t n_tri_in,
                                  float scene_scale, size_t pb, RefitArgs ra, hipStream_t s) {
    const uint32_t n = t.n_instances;
    HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_inst_first, sizeof(uint32_t) * first.size()));
    HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_inst_kind, sizeof(uint32_t) * std::max(n, 1u)));
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_first, first.data(), sizeof(uint32_t) * first.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_kind, t.kind.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
    GpuBuildInput in{};
    in.n_prims = first[n]; in.n_inst = n; in.d_inst_first = t.d_inst_first; in.d_inst_kind = t.d_inst_kind; in.d_inst_src = t.d_inst_src;
    in.d_inst_xf = t.d_inst_xf; in.d_inst_identity = t.d_inst_identity;
    in.max_leaf_prims = kMaxLeafPrims; in.width = (uint32_t)ctx->build_width; in.c_node = ctx->build_c_node; in.c_prim = scene_of_bodies ? ctx->build_c_prim_bodies : ctx->build_c_prim; in.ploc_radius = ctx->ploc_radius; in.quant_guard = ctx->quant_guard;
    // the top-down phase always (object splits: what gives the tree its shape above the cells PLOC builds -- on the reference's kind of
    // scene, separate bodies over a huge ground sphere, PLOC alone costs seven times the node visits); spatial splits under HRT_CTX_FAST_TRACE
    in.split.enabled = device_split || ctx->build_topdown != 0; in.split.budget_frac = device_split ? ctx->split_budget : 0.0f; in.split.alpha = ctx->split_alpha; in.split.bias = ctx->split_bias; in.split.cut_bias = ctx->split_cut_bias;
    in.split.cell_refs = (uint32_t)ctx->split_cell_refs; in.split.pad = ra.pad; in.split.verbose = ctx->build_verbose;
    // worst-case node output (one node and two reference floats per leaf: primitive, or reference of a spatial split) at the front of the
    // working memory; a split build's records and their clip boxes too (their number is known afterwards)
    const size_t max_leaves = gpu_build_max_refs(in.n_prims, &in.split);
    const size_t stage_nodes = ((size_t)t.node_stride * max_leaves + 255u) & ~(size_t)255u, stage_ref = (sizeof(float) * 2 * max_leaves + 255u) & ~(size_t)255u;
    const size_t stage_prims = device_split ? ((size_t)t.prim_stride * max_leaves + 255u) & ~(size_t)255u : 0u, stage_clip = device_split ? (sizeof(float) * 6 * max_leaves + 255u) & ~(size_t)255u : 0u;
    const size_t stage_all = stage_nodes + stage_ref + stage_prims + stage_clip, want = gpu_build_scratch_bytes(in.n_prims, &in.split) + stage_all;
    ScratchArena arena = scratch_acquire(ctx, want);
    if (!arena.p && in.split.enabled && !device_split) {      // no room for the top-down phase's buffers: PLOC alone
        in.split.enabled = false;
        arena = scratch_acquire(ctx, gpu_build_scratch_bytes(in.n_prims, nullptr) + stage_all);
    }
    if (!arena.p) return fail(ctx, HRT_ERR_OOM, "device build: no working memory (%zu bytes)", want);
    struct Release { HrtContext *c; ScratchArena a; hipStream_t st; ~Release() { (void)hipStreamSynchronize(st); scratch_release(c, a); } } release{ctx, arena, s};      // (see ensure_template)
    unsigned char *stage = static_cast<unsigned char *>(arena.p);
    in.out_nodes = stage; in.node_stride = t.node_stride; in.out_node_ref = reinterpret_cast<float *>(stage + stage_nodes);
    in.out_prims = device_split ? stage + stage_nodes + stage_ref : ra.prims; in.prim_stride = t.prim_stride;
    in.out_clip = device_split ? reinterpret_cast<float *>(stage + stage_nodes + stage_ref + stage_prims) : nullptr;
    in.scratch = stage + stage_all; in.scratch_bytes = arena.bytes - stage_all;
    GpuBuildResult r = gpu_build_bvh8(in, s);      // (synchronises the stream before it returns)
    // A tree deeper than any kernel's stack (a chain of primitives over many orders of magnitude: nearest-neighbour clustering, like SAH, takes
    // one off the rest at every level): built again by position -- every cluster with its Morton neighbour, log2(n) levels, whatever the areas
    if (r.error == hipSuccess && r.n_prims != 0u && 2 * r.max_depth + 2 > (uint32_t)(8 + 56) && !device_split) {
        if (ctx->build_verbose) std::fprintf(stderr, "[hrt] the tree is %u levels deep: built again by position\n", r.max_depth);
        in.balanced = true; in.split.enabled = false;
        r = gpu_build_bvh8(in, s);
    }
    if (r.error != hipSuccess) return fail(ctx, r.error == hipErrorOutOfMemory ? HRT_ERR_OOM : HRT_ERR_HIP, "device build failed: %s (%s)", hipGetErrorString(r.error), r.where);
    {   // the tree's own buffers, as large as the build turned out to need
        const size_t nn = std::max<size_t>(r.n_prims ? r.n_nodes : 1u, 1u);
        HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_nodes, (size_t)t.node_stride * nn));
        HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_node_box, sizeof(float) * 6 * nn));
        HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_node_ref, sizeof(float) * 2 * nn));
        size_t rec_bytes = pb;
        if (device_split) {
            rec_bytes = (size_t)t.prim_stride * std::max<size_t>(r.n_records, 1);
            HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_prims, rec_bytes));
            ra.prims = reinterpret_cast<unsigned char *>(t.d_prims);
            if (r.n_records) HIP_TRY(ctx, hipMemcpyAsync(t.d_prims, in.out_prims, (size_t)t.prim_stride * r.n_records, hipMemcpyDeviceToDevice, s));
        }
        if (r.n_prims) {
            HIP_TRY(ctx, hipMemcpyAsync(t.d_nodes, stage, (size_t)t.node_stride * nn, hipMemcpyDeviceToDevice, s));
            HIP_TRY(ctx, hipMemcpyAsync(t.d_node_ref, in.out_node_ref, sizeof(float) * 2 * nn, hipMemcpyDeviceToDevice, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));         // the working memory goes back to the context when this scope ends
        }
        ra.nodes = reinterpret_cast<unsigned char *>(t.d_nodes); ra.node_box = t.d_node_box; ra.node_ref = t.d_node_ref;
        t.alloc_bytes = (uint64_t)t.node_stride * nn + sizeof(float) * 8 * nn + rec_bytes;
    }
    t.bvh = Bvh8();
    if (r.n_prims == 0u) {
        // every primitive had non-finite bounds: the empty root (every ray misses)
        build_bvh8({}, t.bvh, 1, scene_scale);
        HIP_TRY(ctx, hipMemcpyAsync(t.d_nodes, t.bvh.nodes.data(), sizeof(Bvh8Node), hipMemcpyHostToDevice, s));
        t.phases.clear();
        t.n_nodes = 1; t.n_prims = 0; t.n_triangles = t.n_spheres = 0; t.max_depth = 0;
    } else {
        if (2 * r.max_depth + 2 > (uint32_t)(8 + 56)) return fail(ctx, HRT_ERR_INVALID, "BVH depth %u exceeds the traversal stack", r.max_depth);
        for (size_t l = r.level_begin.size() - 1; l-- > 0;) t.phases.emplace_back(r.level_begin[l], r.level_begin[l + 1] - r.level_begin[l]);
        // (a split build: every record's box is the one its cell is responsible for, not the primitive's -- this once; a later
        // update would recompute the boxes from whole primitives, so the first update rebuilds instead: has_split_refs)
        if (device_split && r.split_levels) { ra.clip = in.out_clip; t.has_split_refs = true; }
        attach_rec_box(ctx, t, ra, r.n_records);
        launch_refit_phases(ra, t.phases, s);
        HIP_TRY(ctx, hipGetLastError());
        if (device_split && r.split_levels) HIP_TRY(ctx, hipStreamSynchronize(s));      // (the clip boxes live in the working memory)
        ra.clip = nullptr;
        if (ctx->build_verbose)
            std::fprintf(stderr, "[hrt] device build: %u primitives -> %u records, %u nodes, depth %u, %u PLOC rounds (radius %d), %u split levels, %u cells\n",
                         r.n_prims, r.n_records, r.n_nodes, r.max_depth, r.ploc_rounds, ctx->ploc_radius, r.split_levels, r.n_cells);
        const uint32_t dropped = first[n] - r.n_prims;      // non-finite primitives (counted against the triangles unless there are none)
        t.n_nodes = r.n_nodes; t.n_prims = r.n_records; t.max_depth = r.max_depth;
        t.n_triangles = n_tri_in >= dropped ? n_tri_in - dropped : 0u; t.n_spheres = r.n_prims - t.n_triangles;
        for (int a = 0; a < 3; ++a) { t.lo[a] = r.lo[a]; t.hi[a] = r.hi[a]; }
    }
    return HRT_OK;
}

// ---- a two-level tree (the body of build_tlas_fresh for it).  (1) every unique BLAS has its object-space tree (topology),
//      built once per BLAS and kept on the device; (2) the device build over the instances' bounds, whose leaves come out as transform nodes;
//      (3) the BLAS trees are copied in behind the top level and refitted in object space; (4) the top level's refit -- the only part an update
//      repeats.  t's per-instance tables (transforms, inverses, bounds as the top level's "geometry") are in place; `ra` carries them. ----
static int build_two_level(HrtContext *ctx, Tlas &t, const std::vector<HrtInstance> &inst, const std::vector<Blas *> &uniq, const std::vector<uint32_t> &slot_of,
                           const std::vector<uint32_t> &first, float scene_scale, RefitArgs ra, hipStream_t s) {
    const uint32_t n = (uint32_t)inst.size();
    const uint32_t nu = (uint32_t)uniq.size();
    std::vector<uint32_t> node_off(nu + 1, 0u), prim_off(nu + 1, 0u);
    uint32_t blas_depth = 0;
    for (uint32_t j = 0; j < nu; ++j) {
        const int rc = ensure_template(ctx, *uniq[j], s, true);
        if (rc != HRT_OK) return rc;
        const Bvh8 &tp = uniq[j]->tmpl;
        if ((uint64_t)node_off[j] + tp.nodes.size() > 0x7fffffffull || (uint64_t)prim_off[j] + tp.prims.size() > 0x7fffffffull) return fail(ctx, HRT_ERR_INVALID, "two-level tree: more than 2^31 nodes or records");
        node_off[j + 1] = node_off[j] + (uint32_t)tp.nodes.size(); prim_off[j + 1] = prim_off[j] + (uint32_t)tp.prims.size();
        blas_depth = std::max(blas_depth, tp.max_depth);
    }
    // the top level's primitives: one per visible instance of a non-empty BLAS
    std::vector<uint32_t> first2(n + 1, 0u), kind2(std::max(n, 1u), kPrimKindInstance);
    for (uint32_t i = 0; i < n; ++i) {
        const Blas &b = *t.blas_refs[i];
        first2[i + 1] = first2[i] + (((inst[i].visibilityMask & 1u) != 0 && b.n_prims != 0u && b.lo[0] <= b.hi[0]) ? 1u : 0u);
    }
    const uint32_t n2 = first2[n];
    HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_inst_first, sizeof(uint32_t) * first2.size()));
    HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_inst_kind, sizeof(uint32_t) * kind2.size()));
    HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_inst_root, sizeof(uint32_t) * std::max(n, 1u)));
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_first, first2.data(), sizeof(uint32_t) * first2.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_kind, kind2.data(), sizeof(uint32_t) * kind2.size(), hipMemcpyHostToDevice, s));
    GpuBuildInput in{};
    in.n_prims = n2; in.n_inst = n; in.d_inst_first = t.d_inst_first; in.d_inst_kind = t.d_inst_kind; in.d_inst_src = t.d_inst_src;
    in.d_inst_xf = t.d_inst_xf; in.d_inst_identity = t.d_inst_identity;
    in.max_leaf_prims = 1; in.instance_leaves = true; in.c_node = ctx->build_c_node; in.c_prim = ctx->build_c_prim; in.ploc_radius = ctx->ploc_radius; in.quant_guard = ctx->quant_guard;
    in.split.enabled = ctx->build_topdown != 0; in.split.budget_frac = 0.0f; in.split.alpha = ctx->split_alpha; in.split.bias = ctx->split_bias; in.split.cut_bias = ctx->split_cut_bias;
    in.split.cell_refs = (uint32_t)ctx->split_cell_refs; in.split.pad = ra.pad; in.split.verbose = ctx->build_verbose;
    // worst case: a transform node per instance and fewer box nodes than instances
    const size_t max_nodes = 2 * (size_t)n2 + 2;
    const size_t stage_nodes = ((size_t)t.node_stride * max_nodes + 255u) & ~(size_t)255u, stage_ref = (sizeof(float) * 2 * max_nodes + 255u) & ~(size_t)255u;
    // (the pack's refit tables and walk order share the block: nu small tables, one entry per BLAS node)
    const size_t tab_xf = ((sizeof(float) * 12 * nu) + 255u) & ~(size_t)255u, tab_id = ((sizeof(uint32_t) * nu) + 255u) & ~(size_t)255u, tab_src = ((sizeof(void *) * nu) + 255u) & ~(size_t)255u;
    const size_t tab_order = ((sizeof(uint32_t) * (size_t)node_off[nu]) + 255u) & ~(size_t)255u;
    const size_t stage_all = stage_nodes + stage_ref + tab_xf + tab_id + tab_src + tab_order, want = gpu_build_scratch_bytes(n2, &in.split) + stage_all;
    ScratchArena arena = scratch_acquire(ctx, want);
    if (!arena.p) return fail(ctx, HRT_ERR_OOM, "device build: no working memory (%zu bytes)", want);
    struct Release { HrtContext *c; ScratchArena a; hipStream_t st; ~Release() { (void)hipStreamSynchronize(st); scratch_release(c, a); } } release{ctx, arena, s};
    unsigned char *stage = static_cast<unsigned char *>(arena.p);
    in.out_nodes = stage; in.node_stride = t.node_stride; in.out_node_ref = reinterpret_cast<float *>(stage + stage_nodes);
    in.out_prims = stage; in.prim_stride = t.prim_stride;        // (no record is written: the leaves are transform nodes)
    in.scratch = stage + stage_all; in.scratch_bytes = arena.bytes - stage_all;
    const GpuBuildResult r = gpu_build_bvh8(in, s);      // (synchronises the stream before it returns)
    if (r.error != hipSuccess) return fail(ctx, r.error == hipErrorOutOfMemory ? HRT_ERR_OOM : HRT_ERR_HIP, "device build of the top level failed: %s (%s)", hipGetErrorString(r.error), r.where);
    if (r.n_prims == 0u) return kRetryFlattened;                                 // nothing valid to instance: the flattened path emits the empty root
    // the path kernel keeps one sibling group per level of BOTH trees on its node stack
    if (r.max_depth + 1u + blas_depth > (uint32_t)ctx->fused_max_depth) return kRetryFlattened;
    const uint32_t n_top = r.n_nodes, n_all = n_top + node_off[nu], n_rec = prim_off[nu];
    if ((uint64_t)n_all * t.node_stride >= ctx->fused_max_bytes || (uint64_t)std::max(n_rec, 1u) * t.prim_stride >= ctx->fused_max_bytes) return kRetryFlattened;
    HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_nodes, (size_t)t.node_stride * n_all + 16));       // (+16: the path kernel reads 80 bytes of the last node whatever its stride)
    HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_node_box, sizeof(float) * 6 * (size_t)n_all));
    HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_node_ref, sizeof(float) * 2 * (size_t)n_all));
    HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_prims, (size_t)t.prim_stride * std::max(n_rec, 1u)));
    HIP_TRY(ctx, hipMemcpyAsync(t.d_nodes, stage, (size_t)t.node_stride * n_top, hipMemcpyDeviceToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(t.d_node_ref, in.out_node_ref, sizeof(float) * 2 * (size_t)n_top, hipMemcpyDeviceToDevice, s));
    // (3) the BLAS trees behind the top level
    std::vector<uint32_t> roots(std::max(n, 1u), 0u);
    for (uint32_t i = 0; i < n; ++i) roots[i] = n_top + node_off[slot_of[i]];
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_root, roots.data(), sizeof(uint32_t) * roots.size(), hipMemcpyHostToDevice, s));
    for (uint32_t j = 0; j < nu; ++j) {
        PackBlasArgs pa{};
        pa.src_nodes = uniq[j]->d_tmpl_nodes; pa.src_prims = uniq[j]->d_tmpl_prims; pa.n_nodes = node_off[j + 1] - node_off[j]; pa.n_prims = prim_off[j + 1] - prim_off[j];
        pa.dst_nodes = reinterpret_cast<unsigned char *>(t.d_nodes); pa.dst_prims = reinterpret_cast<unsigned char *>(t.d_prims); pa.node_stride = t.node_stride; pa.prim_stride = t.prim_stride;
        pa.node_off = n_top + node_off[j]; pa.prim_off = prim_off[j]; pa.slot = j;
        launch_pack_blas(pa, s);
    }
    // their refit, all BLASes at once: phase k = the k-th level from the bottom of every tree (a template is stored breadth first,
    // children one level below their parents), walked through an order array; per-BLAS tables with the identity transform
    std::vector<float> pxf(12 * (size_t)nu, 0.0f); std::vector<uint32_t> pid(nu, 1u); std::vector<const void *> psrc(nu, nullptr);
    std::vector<uint32_t> order; order.reserve(node_off[nu]);
    std::vector<std::pair<uint32_t, uint32_t>> pack_phases;
    float obj_coord = 1.0f;
    for (uint32_t j = 0; j < nu; ++j) {
        pxf[12 * (size_t)j] = pxf[12 * (size_t)j + 5] = pxf[12 * (size_t)j + 10] = 1.0f; psrc[j] = uniq[j]->d_verts;
        for (int a = 0; a < 3; ++a) obj_coord = std::max(obj_coord, std::max(std::fabs(uniq[j]->lo[a]), std::fabs(uniq[j]->hi[a])));
    }
    for (uint32_t k = 0; k <= blas_depth; ++k) {
        const uint32_t begin = (uint32_t)order.size();
        for (uint32_t j = 0; j < nu; ++j) {
            const std::vector<uint32_t> &lb = uniq[j]->tmpl.level_begin;
            const uint32_t levels = lb.empty() ? 0u : (uint32_t)lb.size() - 1u;
            if (k >= levels) continue;
            const uint32_t l = levels - 1u - k;
            for (uint32_t x = lb[l]; x < lb[l + 1]; ++x) order.push_back(n_top + node_off[j] + x);
        }
        if (order.size() > begin) pack_phases.emplace_back(begin, (uint32_t)order.size() - begin);
    }
    if (order.size() != node_off[nu]) return fail(ctx, HRT_ERR_HIP, "two-level tree: a BLAS template's levels do not cover its nodes");
    unsigned char *tabs = stage + stage_nodes + stage_ref;
    float *d_pxf = reinterpret_cast<float *>(tabs); uint32_t *d_pid = reinterpret_cast<uint32_t *>(tabs + tab_xf);
    const void **d_psrc = reinterpret_cast<const void **>(tabs + tab_xf + tab_id); uint32_t *d_porder = reinterpret_cast<uint32_t *>(tabs + tab_xf + tab_id + tab_src);
    HIP_TRY(ctx, hipMemcpyAsync(d_pxf, pxf.data(), sizeof(float) * pxf.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(d_pid, pid.data(), sizeof(uint32_t) * nu, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync((void *)d_psrc, psrc.data(), sizeof(void *) * nu, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(d_porder, order.data(), sizeof(uint32_t) * order.size(), hipMemcpyHostToDevice, s));
    // The boxes inside a BLAS are padded for the object-space rays that will come: the slab test's error grows with the distance
    // of the ray's origin, which in object space is the world's extent seen through the instance's inverse (a rigid pose: the
    // world's own extent, as for the flattened tree).
    float reach = 1.0f;
    for (uint32_t i = 0; i < n; ++i) {
        if (first2[i + 1] == first2[i]) continue;
        const float *v = &t.h_inv[12 * (size_t)i];
        float rown = 0.0f;
        for (int rr = 0; rr < 3; ++rr) rown = std::max(rown, std::fabs(v[4 * rr]) + std::fabs(v[4 * rr + 1]) + std::fabs(v[4 * rr + 2]));
        if (std::isfinite(rown)) reach = std::max(reach, rown * scene_scale);
    }
    t.built_reach = std::max(reach, obj_coord);
    RefitArgs rp{};
    rp.nodes = reinterpret_cast<unsigned char *>(t.d_nodes); rp.node_stride = t.node_stride; rp.prims = reinterpret_cast<unsigned char *>(t.d_prims); rp.prim_stride = t.prim_stride;
    rp.node_box = t.d_node_box; rp.node_ref = t.d_node_ref; rp.inst_xf = d_pxf; rp.inst_identity = d_pid; rp.inst_src = d_psrc; rp.order = d_porder;
    rp.pad = 4e-6f * t.built_reach; rp.write_reference = 1u;
    launch_refit_phases(rp, pack_phases, s);
    HIP_TRY(ctx, hipGetLastError());
    // (4) the top level
    ra.nodes = reinterpret_cast<unsigned char *>(t.d_nodes); ra.prims = reinterpret_cast<unsigned char *>(t.d_prims); ra.node_box = t.d_node_box; ra.node_ref = t.d_node_ref;
    ra.inst_inv = t.d_inst_inv; ra.inst_root = t.d_inst_root;
    for (size_t l = r.level_begin.size() - 1; l-- > 0;) t.phases.emplace_back(r.level_begin[l], r.level_begin[l + 1] - r.level_begin[l]);
    launch_refit_phases(ra, t.phases, s);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (ctx->build_verbose)
        std::fprintf(stderr, "[hrt] two-level build: %u instances over %u BLASes -> %u top nodes (depth %u, %u split levels), %u BLAS nodes (depth <= %u), %u records (flattened: %u)\n",
                     n2, nu, n_top, r.max_depth, r.split_levels, node_off[nu], blas_depth, n_rec, first[n]);
    t.bvh = Bvh8();
    t.two_level = true; t.n_top_nodes = n_top; t.n_unique_blas = nu;
    t.n_nodes = n_all; t.n_prims = n_rec; t.max_depth = r.max_depth + 1u + blas_depth;
    t.n_triangles = t.n_spheres = 0;
    for (uint32_t j = 0; j < nu; ++j) { t.n_triangles += uniq[j]->tmpl.n_triangles; t.n_spheres += uniq[j]->tmpl.n_spheres; }
    t.alloc_bytes = (uint64_t)t.node_stride * n_all + sizeof(float) * 8 * (uint64_t)n_all + (uint64_t)t.prim_stride * std::max(n_rec, 1u);
    for (int a = 0; a < 3; ++a) { t.lo[a] = r.lo[a]; t.hi[a] = r.hi[a]; }
    return HRT_OK;
}


static int build_tlas_fresh(HrtContext *ctx, Tlas &t, const std::vector<HrtInstance> &inst, hipStream_t s, bool instanced, bool fast_trace, int two_level_mode) {
    const uint32_t n = (uint32_t)inst.size();
    std::vector<std::shared_ptr<Blas>> refs(n);
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        for (uint32_t i = 0; i < n; ++i) {
            auto it = ctx->blas.find(inst[i].traversableHandle);
            if (it == ctx->blas.end()) return fail(ctx, HRT_ERR_INVALID, "instance %u: unknown BLAS handle 0x%llx", i, (unsigned long long)inst[i].traversableHandle);
            refs[i] = it->second;
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
    // HRT_CTX_FAST_TRACE: the static-scene tree with spatial splits -- from the host builder, or (HRT_FAST_TRACE_BUILD=device) from the device's
    bool device_split = fast_trace && ctx->fast_trace_on_device != 0 && !instanced && ctx->build_on_device != 0;
    const bool device_merged = !instanced && ctx->build_on_device != 0 && (!fast_trace || device_split);
    // global primitive numbering of the merged builds: instance after instance, invisible instances contribute nothing
    std::vector<uint32_t> first(n + 1, 0u);
    uint32_t n_tri_in = 0, n_sph_in = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const bool vis = (inst[i].visibilityMask & 1u) != 0;       // the reference traces with mask 1 (Shader.cu:71)
        const uint32_t cnt = vis ? refs[i]->n_prims : 0u;
        if ((uint64_t)first[i] + cnt > 0xfffffff0ull) return fail(ctx, HRT_ERR_INVALID, "more than 2^32 primitives in one TLAS");
        first[i + 1] = first[i] + cnt;
        if (refs[i]->kind == kPrimKindTriangle) n_tri_in += cnt; else n_sph_in += cnt;
        (void)n_sph_in;
    }
    // a scene of bodies (the reference's kind: particles instancing a few shapes) or a soup?  (the collapse's primitive cost: hrt_internal.hpp)
    uint32_t n_vis = 0;
    for (uint32_t i = 0; i < n; ++i) n_vis += first[i + 1] > first[i] ? 1u : 0u;
    const bool scene_of_bodies = n_vis >= 4u && (uint64_t)first[n] < 20000ull * n_vis;
    t.scene_of_bodies = scene_of_bodies;      // (the launch picks the path kernel's leaf-hold by it, hrt_api.cpp)
    // ---- two levels or one?  Flattening costs memory, build and refit time in proportion to instances x primitives; a two-level tree
    //      (transform nodes over one shared tree per BLAS) in proportion to instances + unique primitives, at the price of a ray
    //      transform per instance entered.  Asked for (two_level_mode > 0), or chosen when the flattened tree would leave the caches
    //      while the shared one stays in them.  Only k_fused walks such trees: not under HRT_CTX_COUNT / HRT_FUSED != 1. ----
    bool two_level = false;
    std::vector<Blas *> uniq; std::vector<uint32_t> slot_of(n, 0u);
    if (two_level_mode >= 0 && !instanced && device_merged && !device_split && first[n] != 0u && (ctx->flags & HRT_CTX_COUNT) == 0 && ctx->fused == 1) {
        std::unordered_map<Blas *, uint32_t> seen;
        uint64_t unique_prims = 0;
        for (uint32_t i = 0; i < n; ++i) {
            Blas *b = refs[i].get();
            if ((inst[i].visibilityMask & 1u) == 0 || b->n_prims == 0u || !(b->lo[0] <= b->hi[0])) continue;
            auto it = seen.find(b);
            if (it == seen.end()) { it = seen.emplace(b, (uint32_t)uniq.size()).first; uniq.push_back(b); unique_prims += b->n_prims; }
            slot_of[i] = it->second;
        }
        two_level = !uniq.empty() && (two_level_mode > 0 || ((uint64_t)first[n] >= ctx->two_level_min_prims &&
                                                             (double)first[n] >= (double)ctx->two_level_min_share * (double)unique_prims));
        if (two_level && fast_trace && two_level_mode <= 0) two_level = false;      // (a static scene that asks for the best tree gets the flattened one with spatial splits)
    }
    if (device_split) {
        // the split build's working memory (~1.2 KB per primitive with the staged output) has to be there: otherwise the default build
        SplitParams probe; probe.enabled = true; probe.budget_frac = ctx->split_budget; probe.cell_refs = (uint32_t)ctx->split_cell_refs;
        const size_t leaves = gpu_build_max_refs(first[n], &probe);
        const size_t want = gpu_build_scratch_bytes(first[n], &probe) + leaves * (size_t)(128 + 8 + ctx->prim_stride + 24);
        size_t free_b = 0, total_b = 0, kept = 0;      // (what the context keeps for later counts as free: an allocation that fails gives it back first)
        { std::lock_guard<std::mutex> lk(ctx->scratch_mu); for (const ScratchArena &x : ctx->scratch_free) kept += x.bytes; }
        { std::lock_guard<std::mutex> lk(ctx->pool_mu); kept += ctx->pool_bytes; }
        if (leaves > (1u << 30) || hipMemGetInfo(&free_b, &total_b) != hipSuccess || want > (free_b + kept) / 10 * 9) device_split = false;
    }
    if (instanced) {
        std::vector<const Bvh8 *> tmpl(n, nullptr);
        std::vector<float> box(6 * (size_t)std::max(n, 1u), 0.0f);
        for (uint32_t i = 0; i < n; ++i) {
            Blas &b = *refs[i];
            if ((inst[i].visibilityMask & 1u) == 0 || !(b.lo[0] <= b.hi[0])) continue;
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int c = 0; c < 8; ++c) {
                const float q[3] = {(c & 1) ? b.hi[0] : b.lo[0], (c & 2) ? b.hi[1] : b.lo[1], (c & 4) ? b.hi[2] : b.lo[2]};
                float w[3];
                if (t.h_ident[i]) { w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; } else xf_point(inst[i].transform, q, w);
                for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], w[a]); hi[a] = std::fmax(hi[a], w[a]); }
            }
            if (!finite_box(lo, hi)) continue;                                                 // a NaN transform: nothing to hit
            const int rc = ensure_template(ctx, b, s);
            if (rc != HRT_OK) return rc;
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
    } else if (!device_merged || first[n] == 0u) {
        // the host's binned-SAH build over the flattened scene (HRT_BUILD=host), and the empty scene
        std::vector<BuildPrim> prims;
        prims.reserve(first[n]);
        for (uint32_t i = 0; i < n && first[n]; ++i) {
            Blas *b = refs[i].get();
            const float *m = inst[i].transform;
            const bool id = t.h_ident[i] != 0u;
            if ((inst[i].visibilityMask & 1u) == 0) continue;
            const int rc = ensure_host_geometry(ctx, *b, s);
            if (rc != HRT_OK) return rc;
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
        // HRT_CTX_FAST_TRACE: the static-scene tree, with spatial splits (a later refit recomputes the boxes from whole primitives:
        // valid, conservative, and without the splits' benefit -- the quality guard of hrt_tlas_update then rebuilds on the device)
        build_bvh8(prims, t.bvh, 0, scene_scale, kMaxLeafPrims, fast_trace);
        t.has_split_refs = t.bvh.prims.size() > prims.size();
        for (size_t l = t.bvh.level_begin.size() - 1; l-- > 0;) t.phases.emplace_back(t.bvh.level_begin[l], t.bvh.level_begin[l + 1] - t.bvh.level_begin[l]);
    }
    const bool on_device = device_merged && first[n] != 0u;
    if (!on_device && 2 * t.bvh.max_depth + 2 > (uint32_t)(8 + 56))
        return fail(ctx, HRT_ERR_INVALID, "BVH depth %u exceeds the traversal stack", t.bvh.max_depth);

    free_tlas_device(ctx, t);
    t.n_instances = n;
    t.blas_refs = std::move(refs);
    t.node_stride = (uint32_t)ctx->node_stride; t.prim_stride = (uint32_t)ctx->prim_stride;
    if (ctx->node_stride_auto && first[n] > 3500000u) t.node_stride = 128u;      // a tree that will not fit the Infinity Cache: one 128-byte line per node
    // A device build emits its nodes into the build's working memory, sized for the worst case (one node per primitive;
    // typically a seventh is used); the TLAS gets buffers of the size the build turned out to need (below).
    const size_t n_nodes = on_device ? 0 : t.bvh.nodes.size(), n_prims = on_device ? (size_t)first[n] : t.bvh.prims.size();
    const size_t nb = (size_t)t.node_stride * n_nodes;
    const size_t pb = (size_t)t.prim_stride * std::max<size_t>(n_prims, 1);
    std::vector<const void *> src(std::max(n, 1u), nullptr);
    for (uint32_t i = 0; i < n; ++i) src[i] = t.blas_refs[i]->d_verts;
    if (two_level) {      // the top level's "geometry": the instances' BLAS boxes and bounding spheres
        std::vector<float> bound(10 * (size_t)std::max(n, 1u), 0.0f);
        for (uint32_t i = 0; i < n; ++i) {
            Blas &b = *t.blas_refs[i];
            float *q = &bound[10 * (size_t)i];
            for (int a = 0; a < 3; ++a) { q[a] = b.lo[a]; q[3 + a] = b.hi[a]; }
            q[9] = -1.0f;
            if (b.n_prims != 0u && b.lo[0] <= b.hi[0]) {
                std::lock_guard<std::mutex> lk(b.tmpl_mu);
                if (!b.bsphere_ready) {
                    for (int a = 0; a < 3; ++a) b.bsphere[a] = 0.5f * b.lo[a] + 0.5f * b.hi[a];
                    const ScratchArena arena = scratch_acquire(ctx, kBoundsScratchBytes);
                    const hipError_t e = gpu_blas_radius(b.d_verts, b.n_prims, b.kind, b.bsphere, &b.bsphere[3], arena.p, s);
                    scratch_release(ctx, arena);
                    HIP_TRY(ctx, e);
                    b.bsphere_ready = true;
                }
                for (int a = 0; a < 4; ++a) q[6 + a] = b.bsphere[a];
            }
        }
        HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_blas_bound, sizeof(float) * bound.size()));
        HIP_TRY(ctx, hipMemcpyAsync(t.d_blas_bound, bound.data(), sizeof(float) * bound.size(), hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
        for (uint32_t i = 0; i < n; ++i) src[i] = t.d_blas_bound + 10 * (size_t)i;
    }
    if (!on_device) HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_nodes, nb));
    if (!(on_device && (device_split || two_level))) HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_prims, pb));      // (a split build knows its record count afterwards; a two-level tree holds the unique primitives' records only)
    HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_inst_inv, sizeof(float) * t.h_inv.size()));
    HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_inst_xf, sizeof(float) * t.h_xf.size()));
    HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_inst_identity, sizeof(uint32_t) * t.h_ident.size()));
    HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_inst_src, sizeof(void *) * src.size()));
    if (!on_device) {
        HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_node_box, sizeof(float) * std::max<size_t>(6 * n_nodes, 6)));
        HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_node_ref, sizeof(float) * std::max<size_t>(2 * n_nodes, 2)));
    }
    HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_area, sizeof(float)));
    if (!order.empty()) HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_order, sizeof(uint32_t) * order.size()));
    if (!t.h_area) HIP_TRY(ctx, hipHostMalloc((void **)&t.h_area, sizeof(float), hipHostMallocDefault));
    if (!t.area_ready) HIP_TRY(ctx, hipEventCreateWithFlags(&t.area_ready, hipEventDisableTiming));
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_inv, t.h_inv.data(), sizeof(float) * t.h_inv.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_xf, t.h_xf.data(), sizeof(float) * t.h_xf.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_identity, t.h_ident.data(), sizeof(uint32_t) * t.h_ident.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync((void *)t.d_inst_src, src.data(), sizeof(void *) * src.size(), hipMemcpyHostToDevice, s));
    {   // what asynchronous updates compare against and transform on the device
        std::vector<float> bbox(6 * (size_t)std::max(n, 1u), 0.0f);
        for (uint32_t i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) { bbox[6 * (size_t)i + a] = t.blas_refs[i]->lo[a]; bbox[6 * (size_t)i + 3 + a] = t.blas_refs[i]->hi[a]; }
        std::vector<unsigned long long> sigh(std::max(n, 1u), 0ull);
        for (uint32_t i = 0; i < n; ++i) sigh[i] = t.sig_handle[i];
        HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_sig_handle, sizeof(unsigned long long) * sigh.size()));
        HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_sig_visibility, sizeof(uint32_t) * std::max(n, 1u)));
        HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_sig_sbt, sizeof(uint32_t) * std::max(n, 1u)));
        HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_blas_box, sizeof(float) * bbox.size()));
        HIP_TRY(ctx, pool_alloc(ctx, (void **)&t.d_update_flags, sizeof(uint32_t) * 2));
        if (!t.h_update_flags) HIP_TRY(ctx, hipHostMalloc((void **)&t.h_update_flags, sizeof(uint32_t) * 4, hipHostMallocDefault));
        t.h_update_flags[0] = t.h_update_flags[2] = 0x3f800000u; t.h_update_flags[1] = t.h_update_flags[3] = 0u;
        HIP_TRY(ctx, hipMemcpyAsync(t.d_sig_handle, sigh.data(), sizeof(unsigned long long) * sigh.size(), hipMemcpyHostToDevice, s));
        if (n) HIP_TRY(ctx, hipMemcpyAsync(t.d_sig_visibility, t.sig_visibility.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
        if (n) HIP_TRY(ctx, hipMemcpyAsync(t.d_sig_sbt, t.sbt_offset.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(t.d_blas_box, bbox.data(), sizeof(float) * bbox.size(), hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));            // (the staging vectors go out of scope)
    }
    RefitArgs ra{};
    ra.nodes = reinterpret_cast<unsigned char *>(t.d_nodes); ra.node_stride = t.node_stride;
    ra.prims = reinterpret_cast<unsigned char *>(t.d_prims); ra.prim_stride = t.prim_stride;
    ra.node_box = t.d_node_box; ra.node_ref = t.d_node_ref; ra.inst_xf = t.d_inst_xf; ra.inst_identity = t.d_inst_identity;
    ra.inst_src = t.d_inst_src; ra.pad = 4e-6f * std::max(1.0f, scene_scale); ra.write_reference = 1u;
    if (on_device && two_level) {
        const int rc2 = build_two_level(ctx, t, inst, uniq, slot_of, first, scene_scale, ra, s);
        if (rc2 != HRT_OK) return rc2;
    } else if (on_device) {
        const int rc2 = build_merged_on_device(ctx, t, first, device_split, scene_of_bodies, n_tri_in, scene_scale, pb, ra, s);
        if (rc2 != HRT_OK) return rc2;
    } else {
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
        if (!t.bvh.node_box.empty())
            HIP_TRY(ctx, hipMemcpyAsync(t.d_node_box, t.bvh.node_box.data(), sizeof(float) * t.bvh.node_box.size(), hipMemcpyHostToDevice, s));
        if (!t.bvh.node_ref.empty())
            HIP_TRY(ctx, hipMemcpyAsync(t.d_node_ref, t.bvh.node_ref.data(), sizeof(float) * t.bvh.node_ref.size(), hipMemcpyHostToDevice, s));
        if (!order.empty()) HIP_TRY(ctx, hipMemcpyAsync(t.d_order, order.data(), sizeof(uint32_t) * order.size(), hipMemcpyHostToDevice, s));
        if (instanced && n_prims) {
            // the device computes what the host left blank: world-space records, boxes, origins, exponents, quantised
            // children, and the built areas the quality guard compares later refits with
            ra.order = t.d_order;
            attach_rec_box(ctx, t, ra, (uint32_t)n_prims);
            launch_refit_phases(ra, t.phases, s);
            HIP_TRY(ctx, hipGetLastError());
        }
        t.n_nodes = (uint32_t)t.bvh.nodes.size(); t.n_prims = (uint32_t)t.bvh.prims.size();
        t.alloc_bytes = (uint64_t)nb + sizeof(float) * 8 * std::max<size_t>(n_nodes, 1) + pb;
        t.n_triangles = t.bvh.n_triangles; t.n_spheres = t.bvh.n_spheres; t.max_depth = t.bvh.max_depth;
        for (int a = 0; a < 3; ++a) { t.lo[a] = t.bvh.lo[a]; t.hi[a] = t.bvh.hi[a]; }
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    t.instanced = instanced;
    t.refits_since_build = 0;
    return HRT_OK;
}

// (Re)build the tree of a TLAS.  The new tree is built on the side and takes the place of the old one only when
// everything has succeeded: a failed rebuild (depth limit, out of memory) leaves the registered TLAS as it was --
// valid and traceable -- instead of half overwritten.
int build_tlas_into(HrtContext *ctx, Tlas &t, const std::vector<HrtInstance> &inst, hipStream_t s, bool instanced, bool fast_trace = false, int two_level_mode = -1) {
    Tlas fresh;
    // (the pinned host words and the event move to the new tree: hipHostMalloc / hipHostFree are as slow as their device counterparts)
    if (t.area_ready && t.area_pending) (void)hipEventSynchronize(t.area_ready);
    fresh.h_area = t.h_area; fresh.h_update_flags = t.h_update_flags; fresh.area_ready = t.area_ready;
    t.h_area = nullptr; t.h_update_flags = nullptr; t.area_ready = nullptr;
    int rc = build_tlas_fresh(ctx, fresh, inst, s, instanced, fast_trace, two_level_mode);
    if (rc == kRetryFlattened) {
        free_tlas_device(ctx, fresh);
        fresh.two_level = false; fresh.phases.clear();
        rc = build_tlas_fresh(ctx, fresh, inst, s, instanced, fast_trace, -1);
    }
    if (rc != HRT_OK) {
        t.h_area = fresh.h_area; t.h_update_flags = fresh.h_update_flags; t.area_ready = fresh.area_ready;
        fresh.h_area = nullptr; fresh.h_update_flags = nullptr; fresh.area_ready = nullptr;
        free_tlas_device(ctx, fresh); free_tlas_host(fresh); return rc;
    }
    fresh.generation = t.generation + 1;
    fresh.refits = t.refits; fresh.rebuilds = t.rebuilds + 1;
    std::swap(t, fresh);
    free_tlas_device(ctx, fresh); free_tlas_host(fresh);        // the old tree
    ctx->tlas_rebuilds++;
    return HRT_OK;
}

int download_instances(HrtContext *ctx, const HrtInstance *d_instances, uint32_t n, hipStream_t s, std::vector<HrtInstance> &inst) {
    const size_t bytes = sizeof(HrtInstance) * (size_t)n;
    if (n >= 16384u) {
        // many instances (a DEM time step: 8 MB), every frame of a synchronous update loop: through pinned memory the context keeps
        std::lock_guard<std::mutex> lk(ctx->pin_mu);
        if (ctx->pin_bytes < bytes) {
            if (ctx->pin_stage) (void)hipHostFree(ctx->pin_stage);
            ctx->pin_stage = nullptr; ctx->pin_bytes = 0;
            if (hipHostMalloc(&ctx->pin_stage, bytes, hipHostMallocDefault) == hipSuccess) ctx->pin_bytes = bytes;
            else { (void)hipGetLastError(); ctx->pin_stage = nullptr; }
        }
        if (ctx->pin_stage) {
            HIP_TRY(ctx, hipMemcpyAsync(ctx->pin_stage, d_instances, bytes, hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));
            const HrtInstance *p = static_cast<const HrtInstance *>(ctx->pin_stage);
            inst.assign(p, p + n);
            return HRT_OK;
        }
    }
    inst.resize(n);
    if (n) {
        HIP_TRY(ctx, hipMemcpyAsync(inst.data(), d_instances, bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
    }
    return HRT_OK;
}

// The phases of a refit, children before parents: wide phases get a launch each, the narrow ones at the end (the top
// of the tree, or all of a small tree) run in one single-workgroup launch.
void launch_refit_phases(RefitArgs ra, const std::vector<std::pair<uint32_t, uint32_t>> &phases, hipStream_t s) {
    launch_refit_records(ra, s);                               // (small trees: ra.rec_box)
    size_t tail = phases.size();
    while (tail > 0 && phases.size() - tail < kRefitTopLevels && phases[tail - 1].second <= kRefitTopLevelNodes) --tail;
    for (size_t i = 0; i < tail; ++i) { ra.first_node = phases[i].first; ra.n_nodes = phases[i].second; launch_refit_level(ra, s); }
    RefitLevels top{};
    for (size_t i = tail; i < phases.size(); ++i) { top.first[top.n_levels] = phases[i].first; top.count[top.n_levels] = phases[i].second; ++top.n_levels; }
    launch_refit_top(ra, top, s);
}

// Small trees (the reference's own scenes) are refitted in one workgroup that walks the levels; the records' arithmetic -- most of the
// work, and a chain of dependent loads -- runs before that, a thread per record, into 24 bytes per record (k_refit_records).
constexpr uint32_t kRecBoxMaxRecords = 65536;
static void attach_rec_box(HrtContext *ctx, Tlas &t, RefitArgs &ra, uint32_t n_records) {
    if (n_records == 0u || n_records > kRecBoxMaxRecords) return;
    if (!t.d_rec_box && pool_alloc(ctx, (void **)&t.d_rec_box, sizeof(float) * 6 * (size_t)n_records) != hipSuccess) { (void)hipGetLastError(); t.d_rec_box = nullptr; return; }
    ra.rec_box = t.d_rec_box; ra.n_records = n_records;
}

// Device refit of a built tree under new instance transforms: upload the per-instance tables, then one
// k_refit_level launch per tree level, deepest first.  Asynchronous on s.
int refit_tlas(HrtContext *ctx, Tlas &t, const std::vector<HrtInstance> &inst, hipStream_t s) {
    const float scene_scale = instance_tables(inst, t.blas_refs, t.h_xf, t.h_inv, t.h_ident);
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_inv, t.h_inv.data(), sizeof(float) * t.h_inv.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_xf, t.h_xf.data(), sizeof(float) * t.h_xf.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(t.d_inst_identity, t.h_ident.data(), sizeof(uint32_t) * t.h_ident.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(t.d_area, 0, sizeof(float), s));
    t.async_words_ready = false;
    RefitArgs ra{};
    ra.nodes = reinterpret_cast<unsigned char *>(t.d_nodes); ra.node_stride = t.node_stride;
    ra.prims = reinterpret_cast<unsigned char *>(t.d_prims); ra.prim_stride = t.prim_stride;
    ra.node_box = t.d_node_box; ra.node_ref = t.d_node_ref; ra.inst_xf = t.d_inst_xf; ra.inst_identity = t.d_inst_identity; ra.inst_src = t.d_inst_src;
    ra.pad = 4e-6f * std::max(1.0f, scene_scale);
    ra.area_sum = t.d_area;
    ra.order = t.d_order;
    if (t.two_level) { ra.inst_inv = t.d_inst_inv; ra.inst_root = t.d_inst_root; }      // (the top level only: transform nodes and the boxes above them)
    else attach_rec_box(ctx, t, ra, t.n_prims);
    { Timer tm(ctx, s, HRT_K_REFIT); launch_refit_phases(ra, t.phases, s); }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(t.h_area, t.d_area, sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipEventRecord(t.area_ready, s));
    t.area_pending = true;
    t.refits++; t.refits_since_build++; ctx->tlas_refits++;
    return HRT_OK;
}

}  // namespace hrt

// A rebuild in the middle of an animation.  Large scenes: the merged device build (3.1 ms for 2000 particles / 435 k triangles, 3.3 ms
// for a million triangles: profiles/r03_device_split_build.txt) -- the better tree.  Small scenes and host-build contexts: the tree over
// instances, whose top tree the host assembles in well under a millisecond.
static bool rebuild_over_instances(const HrtContext *ctx, const Tlas &t) {
    uint64_t total = 0;
    for (uint32_t i = 0; i < t.n_instances; ++i) total += t.blas_refs[i] ? t.blas_refs[i]->n_prims : 0u;
    return ctx->tlas_instanced >= 0 && t.n_instances >= 2 && (!ctx->build_on_device || total < 100000ull || ctx->tlas_instanced > 0);
}

// The first update after a build, before anything is refitted: have the instances gone somewhere else altogether?  The reference
// builds every file's IAS with identity transforms and poses it afterwards (RendererTime.cu:111-127): the tree was built over particles
// lying on top of each other, a refit of it is a tree in name only, and the refit, the wait for its verdict and the rebuild that follows
// are the rebuild alone when this says so.  Moved far = an instance's box centre is further from where it was than the box is wide.
static bool instances_moved_far(const Tlas &t, const std::vector<HrtInstance> &inst) {
    uint32_t valid = 0, far = 0;
    for (uint32_t i = 0; i < t.n_instances; ++i) {
        const Blas *b = t.blas_refs[i].get();
        if (!b || (inst[i].visibilityMask & 1u) == 0 || !(b->lo[0] <= b->hi[0])) continue;
        const float c[3] = {0.5f * (b->lo[0] + b->hi[0]), 0.5f * (b->lo[1] + b->hi[1]), 0.5f * (b->lo[2] + b->hi[2])};
        const float e[3] = {b->hi[0] - b->lo[0], b->hi[1] - b->lo[1], b->hi[2] - b->lo[2]};
        float was[3], now[3], ext[3];
        xf_point(&t.h_xf[12 * (size_t)i], c, was); xf_point(inst[i].transform, c, now);
        const float *m = &t.h_xf[12 * (size_t)i];
        for (int k = 0; k < 3; ++k) ext[k] = std::fabs(m[4 * k]) * e[0] + std::fabs(m[4 * k + 1]) * e[1] + std::fabs(m[4 * k + 2]) * e[2];
        const float d2 = (now[0] - was[0]) * (now[0] - was[0]) + (now[1] - was[1]) * (now[1] - was[1]) + (now[2] - was[2]) * (now[2] - was[2]);
        const float w2 = ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2];
        ++valid;
        if (d2 > w2) ++far;
    }
    return valid >= 4 && 2 * far > valid;
}

extern "C" {

// ---- acceleration structures ---------------------------------------------------------
int hrt_blas_build_triangles(HrtContext *ctx, const HrtFloat3 *d_vertices, uint32_t n_vertices, void *stream, HrtTraversable *out_blas) {
    if (!ctx || !out_blas) return HRT_ERR_INVALID;
    if (n_vertices % 3 != 0) return fail(ctx, HRT_ERR_INVALID, "n_vertices (%u) is not a multiple of 3", n_vertices);
    if (n_vertices && !d_vertices) return fail(ctx, HRT_ERR_INVALID, "d_vertices is NULL");
    (void)hipSetDevice(ctx->device);
    std::shared_ptr<Blas> b(new Blas());
    b->kind = kPrimKindTriangle; b->n_prims = n_vertices / 3;
    if (n_vertices) {
        // the geometry stays on the device: a copy of it (the caller may free its buffer, RendererMesh.cu:116) and its
        // object-space bounds (24 bytes come back); trees are built from the copy when a TLAS is built over the BLAS
        const size_t bytes = sizeof(float) * 3 * (size_t)n_vertices;
        HIP_TRY(ctx, hipMalloc((void **)&b->d_verts, bytes));
        HIP_TRY(ctx, hipMemcpyAsync(b->d_verts, d_vertices, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        const ScratchArena arena = scratch_acquire(ctx, kBoundsScratchBytes);
        const hipError_t e = gpu_blas_bounds(b->d_verts, b->n_prims, kPrimKindTriangle, b->lo, b->hi, arena.p, (hipStream_t)stream);
        scratch_release(ctx, arena);
        HIP_TRY(ctx, e);
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
    if (n) {
        HIP_TRY(ctx, hipMalloc((void **)&b->d_verts, sizeof(float) * 4 * (size_t)n));        // {cx, cy, cz, r} per sphere
        launch_pack_spheres(reinterpret_cast<const float *>(d_centers), d_radii, n, b->d_verts, (hipStream_t)stream);
        HIP_TRY(ctx, hipGetLastError());
        const ScratchArena arena = scratch_acquire(ctx, kBoundsScratchBytes);
        const hipError_t e = gpu_blas_bounds(b->d_verts, n, kPrimKindSphere, b->lo, b->hi, arena.p, (hipStream_t)stream);
        scratch_release(ctx, arena);
        HIP_TRY(ctx, e);
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
    const int two_level_mode = (ctx->flags & HRT_CTX_TWO_LEVEL) != 0 ? 1 : ctx->two_level;
    if (rc == HRT_OK) rc = build_tlas_into(ctx, *t, inst, (hipStream_t)stream, ctx->tlas_instanced > 0 && two_level_mode <= 0, (ctx->flags & HRT_CTX_FAST_TRACE) != 0, two_level_mode);
    if (rc != HRT_OK) { free_tlas_device(ctx, *t); free_tlas_host(*t); return rc; }
    // A rebuild in the middle of an animation builds a tree over instances when the scene is small (hrt_tlas_update): the per-BLAS
    // template trees that needs are built now, while the scene is being loaded, not inside the first frame that rebuilds (the
    // reference's shipped sample: 9 templates, 8.7 ms of its first frame)
    if (!t->two_level && rebuild_over_instances(ctx, *t))
        for (uint32_t i = 0; i < n && rc == HRT_OK; ++i) {
            Blas &b = *t->blas_refs[i];
            if ((inst[i].visibilityMask & 1u) != 0 && b.lo[0] <= b.hi[0]) rc = ensure_template(ctx, b, (hipStream_t)stream);
        }
    if (rc != HRT_OK) { free_tlas_device(ctx, *t); free_tlas_host(*t); return rc; }
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
    bool force_rebuild = false, sbt_sync = false;
    // (the first update after a build takes the synchronous path below, which checks that refit on the spot: see there)
    if ((ctx->flags & HRT_CTX_ASYNC_UPDATE) != 0 && ctx->refit != 0 && t->n_prims != 0u && n != 0u && (t->refits_since_build != 0 || t->built_posed)) {
        // ---- asynchronous update: nothing is read back now.  First the verdict of the previous one (long complete). ----
        if (t->area_pending) {
            HIP_TRY(ctx, hipEventSynchronize(t->area_ready));
            t->area_pending = false;
            ctx->tlas_refit_ratio = (double)*t->h_area;
            if (!(ctx->tlas_refit_ratio <= (double)ctx->refit_rebuild_ratio) || (t->h_update_flags[1] & 1u) != 0u) force_rebuild = true;
            else if ((t->h_update_flags[1] & 2u) != 0u) sbt_sync = true;      // an sbtOffset changed: the synchronous path below re-reads them
            t->h_update_flags[1] = 0u;
        }
        if (!force_rebuild && !sbt_sync) {
            // (the device words the tables kernel and the refit accumulate into -- scene scale, verdict bits, area sum -- are left in their
            // initial state by the previous asynchronous update's epilogue; after a build or a synchronous update they are set here)
            if (!t->async_words_ready) {
                HIP_TRY(ctx, hipMemcpyAsync(t->d_update_flags, t->h_update_flags + 2, sizeof(uint32_t) * 2, hipMemcpyHostToDevice, s));
                HIP_TRY(ctx, hipMemsetAsync(t->d_area, 0, sizeof(float), s));
                t->async_words_ready = true;
            }
            InstanceTableArgs ia{};
            ia.instances = d_instances; ia.n = n; ia.sig_handle = t->d_sig_handle; ia.sig_visibility = t->d_sig_visibility; ia.sig_sbt = t->d_sig_sbt; ia.blas_box = t->d_blas_box;
            ia.inst_xf = t->d_inst_xf; ia.inst_inv = t->d_inst_inv; ia.inst_identity = t->d_inst_identity; ia.flags = t->d_update_flags;
            launch_instance_tables(ia, s);
            RefitArgs ra{};
            ra.nodes = reinterpret_cast<unsigned char *>(t->d_nodes); ra.node_stride = t->node_stride;
            ra.prims = reinterpret_cast<unsigned char *>(t->d_prims); ra.prim_stride = t->prim_stride;
            ra.node_box = t->d_node_box; ra.node_ref = t->d_node_ref; ra.inst_xf = t->d_inst_xf; ra.inst_identity = t->d_inst_identity; ra.inst_src = t->d_inst_src;
            ra.scale_bits = t->d_update_flags; ra.area_sum = t->d_area; ra.order = t->d_order;
            if (t->two_level) { ra.inst_inv = t->d_inst_inv; ra.inst_root = t->d_inst_root; }
            else attach_rec_box(ctx, *t, ra, t->n_prims);
            { Timer tm(ctx, s, HRT_K_REFIT); launch_refit_phases(ra, t->phases, s); }
            // one launch where there were two copies to the host, one from it and a fill: results to pinned memory, device words reset
            UpdateEpilogueArgs ue{t->d_area, t->d_update_flags, t->h_area, t->h_update_flags, t->h_update_flags[2], t->h_update_flags[3]};
            launch_update_epilogue(ue, s);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipEventRecord(t->area_ready, s));
            t->area_pending = true;
            t->refits++; t->refits_since_build++; ctx->tlas_refits++;
            return HRT_OK;
        }
    }
    std::vector<HrtInstance> inst;
    const int rc = download_instances(ctx, d_instances, n, s, inst);
    if (rc != HRT_OK) return rc;
    // (a tree with split references is a static-scene tree: refitted, its leaves would fall back to whole-primitive boxes around
    // duplicated records -- worse than no splits -- so the first update replaces it by a device-built tree)
    bool same = ctx->refit != 0 && t->n_prims != 0u && !force_rebuild && !t->has_split_refs;
    for (uint32_t i = 0; i < n && same; ++i)
        same = inst[i].traversableHandle == t->sig_handle[i] && (inst[i].visibilityMask & 1u) == t->sig_visibility[i];
    bool moved_far = false;
    if (same && t->refits_since_build == 0 && ctx->refit_moved_far_check && instances_moved_far(*t, inst)) { same = false; moved_far = true; }
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
        if (sbt_changed) {
            t->generation++;                              // the material tables are re-derived at the next launch
            HIP_TRY(ctx, hipMemcpyAsync(t->d_sig_sbt, t->sbt_offset.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));
        }
        const bool first_after_build = t->refits_since_build == 0;
        const int rc = refit_tlas(ctx, *t, inst, s);
        if (rc != HRT_OK || !first_after_build) return rc;
        // The first refit after a build is checked on the spot (one stream synchronisation per build): the reference builds
        // every file's IAS with identity transforms and poses it afterwards (RendererTime.cu:111-127), so this is the refit
        // that turns a tree built over coinciding particles into the real scene -- and the frame that follows would be
        // traced through boxes that span everything.  Later refits are checked asynchronously, at the next update.
        HIP_TRY(ctx, hipEventSynchronize(t->area_ready));
        t->area_pending = false;
        ctx->tlas_refit_ratio = (double)*t->h_area;
        if (ctx->tlas_refit_ratio <= (double)ctx->refit_rebuild_ratio) return HRT_OK;
    }
    if (ctx->build_verbose)
        std::fprintf(stderr, "[hrt] update %llu of this tree rebuilds: %s (area ratio %.3f, %llu refits since the build)\n", (unsigned long long)(t->refits + t->rebuilds),
                     force_rebuild ? "verdict of the previous asynchronous refit" : moved_far ? "most instances are further from where the tree was built than they are wide" : "handles / visibility changed or the refit just done degraded the tree", ctx->tlas_refit_ratio.load(),
                     (unsigned long long)t->refits_since_build);
    HIP_TRY(ctx, hipDeviceSynchronize());                 // launches on other streams may still read the old tree
    // a two-level tree stays one: its top level is rebuilt, the BLAS trees are copied in again
    const int rb = t->two_level ? build_tlas_into(ctx, *t, inst, s, false, false, 1) : build_tlas_into(ctx, *t, inst, s, rebuild_over_instances(ctx, *t));
    if (rb == HRT_OK) t->built_posed = true;      // (built for the instances as they are now: the next update need not be checked on the spot)
    return rb;
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
    a.mesh_mode = h_params->mesh_mode ? 1u : 0u;
    launch_pose_instances(a, (hipStream_t)stream);
    HIP_TRY(ctx, hipGetLastError());
    return HRT_OK;
}

int hrt_debug_trig(HrtContext *ctx, int function, const float *d_a, const float *d_b, uint32_t first_bits, uint32_t stride_bits, uint64_t n,
                   int force_slow, float *d_out, void *stream) {
    if (!ctx) return HRT_ERR_INVALID;
    if (function < 0 || function > 4) return fail(ctx, HRT_ERR_INVALID, "hrt_debug_trig: function %d (0 sin, 1 cos, 2 acos, 3 asin, 4 atan2)", function);
    if (n && (!d_out || (function == 4 && (!d_a || !d_b)))) return fail(ctx, HRT_ERR_INVALID, "hrt_debug_trig: NULL argument");
    (void)hipSetDevice(ctx->device);
    launch_debug_trig(function, d_a, d_b, first_bits, stride_bits, n, force_slow, d_out, (hipStream_t)stream);
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
    free_tlas_device(ctx, *it->second); free_tlas_host(*it->second);
    ctx->tlas.erase(it);
    return HRT_OK;
}

int alloc_bvh_blob(size_t n_nodes, size_t n_prims, const float *lo, const float *hi, HrtBvhBlob *out);      // bvh8_host_api.cpp

int hrt_tlas_download(HrtContext *ctx, HrtTraversable tlas, HrtBvhBlob *out) {
    if (!ctx || !out) return HRT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    Tlas *t;
    { std::lock_guard<std::mutex> lk(ctx->mu); auto it = ctx->tlas.find(tlas); if (it == ctx->tlas.end()) return fail(ctx, HRT_ERR_INVALID, "unknown TLAS handle"); t = it->second.get(); }
    const int rc = alloc_bvh_blob(t->n_nodes, t->n_prims, t->lo, t->hi, out);
    if (rc != HRT_OK) return rc;
    // the device copy is the truth: device builds exist nowhere else, and a refit rewrites the tree in place
    HIP_TRY(ctx, hipDeviceSynchronize());
    const size_t n_nodes = t->n_nodes, n_prims = t->n_prims;
    HIP_TRY(ctx, hipMemcpy2D(out->nodes, sizeof(Bvh8Node), t->d_nodes, t->node_stride, sizeof(Bvh8Node), n_nodes, hipMemcpyDeviceToHost));
    if (n_prims) HIP_TRY(ctx, hipMemcpy2D(out->triangles, sizeof(PrimRecord), t->d_prims, t->prim_stride, sizeof(PrimRecord), n_prims, hipMemcpyDeviceToHost));
    return HRT_OK;
}

}  // extern "C"
