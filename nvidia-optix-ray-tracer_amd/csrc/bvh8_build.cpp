// bvh8_build.cpp -- deterministic host builder for the packed BVH8 the HIP kernels traverse.
//
// Stands in for optixAccelBuild (reference: src/Global/RendererImpl.cu:30-88 buildASImpl,
// :174-208 buildIAS).  OptiX's builder is closed; this one is:
//   1. binned-SAH BVH2 over padded primitive bounds (16 bins x 3 axes) down to single primitives,
//      big subtrees built on worker threads (topology does not depend on thread timing), optionally with
//      spatial splits (SBVH: references cut by planes where that is cheaper, for static scenes);
//   2. optimal SAH collapse to 8-wide nodes by dynamic programming (Ylitie, Karras, Laine 2017,
//      sec. 4.1): C(n, i) = cheapest way to represent subtree n as a forest of at most i roots,
//      c_node = 1, c_prim = 0.45, leaves <= 3 primitives.  (A first greedy version -- always open
//      the largest inner child -- left 64 % of the nodes with only two children.)
//   3. octant-ordered slot assignment so that (slot ^ (7 - ray_octant)) approximates
//      front-to-back order during traversal;
//   4. outward-rounded 8-bit quantisation of the child boxes against the node origin/exponent.
#include "bvh8.h"
#include "bvh8_geom.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <chrono>
#include <cstdio>
#include <thread>

namespace hrt {
namespace {

struct B2 {
    float lo[3], hi[3];       // padded bounds of the references below
    uint32_t left, right;     // inner
    uint32_t first, count;    // leaf when count > 0: ONE reference, first = its primitive
    uint32_t nprims;          // references in the subtree
};

inline float half_area(const float *lo, const float *hi) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// A reference: a primitive and the (unpadded) box of the part of it this subtree is responsible for.  Without spatial splits
// that is the primitive's own box and every primitive has one reference.
struct Ref { uint32_t prim; float lo[3], hi[3]; };

constexpr float kInfF = std::numeric_limits<float>::infinity();

struct Box3 {
    float lo[3] = {kInfF, kInfF, kInfF}, hi[3] = {-kInfF, -kInfF, -kInfF};
    void grow(const float *l, const float *h) { for (int c = 0; c < 3; ++c) { lo[c] = std::min(lo[c], l[c]); hi[c] = std::max(hi[c], h[c]); } }
    void grow(const Box3 &b) { grow(b.lo, b.hi); }
    bool valid() const { return lo[0] <= hi[0]; }
    float area() const { return valid() ? half_area(lo, hi) : 0.0f; }
};

struct Builder {
    const std::vector<BuildPrim> &in;
    float pad = 0.0f;
    std::vector<B2> nodes;
    std::atomic<uint32_t> n_nodes{0};
    int max_par_depth = 0;
    // spatial splits (Stich, Friedrich, Dietrich 2009, "Spatial splits in bounding volume hierarchies"): a node may cut its
    // references with an axis-aligned plane instead of partitioning them -- a reference that straddles the plane goes to both
    // sides with its box clipped -- when that is cheaper by the SAH and the object split's children overlap enough.  Duplicate
    // references are harmless to the canonical closest hit (min t, then min (instance, primitive)).
    bool spatial = false;
    bool fast_binning = true;            // the spatial BINS take a straddling reference's box cut by the slab, not the clipped triangle: it only decides
                                         // where the plane goes (same tree quality, half the build time; HRT_SBVH_FAST_BINNING=0: clip); the split itself clips
    int n_bins = 16;                     // object-split bins per axis
    uint32_t spatial_min_refs = 48;            // spatial splits are tried in nodes of at least this many references (HRT_SBVH_MIN_REFS): below, they cost build time and references for nothing (profiles/r03_tree_quality_cpu.txt)
    uint32_t spatial_max_refs = 0xffffffffu;   // experiment knob (HRT_SBVH_MAX_REFS): spatial splits only in nodes with at most this many references
    float spatial_bias = 0.95f;          // < 1 favours spatial splits over object splits of equal SAH cost (C4: 3610 -> 3663 Mrays/s at 0.95, no further gain below)
    float alpha = 1e-5f;                 // spatial splits are tried when area(left ∩ right) / area(root) exceeds this
    float root_area = 0.0f;

    explicit Builder(const std::vector<BuildPrim> &p) : in(p) {}

    uint32_t alloc2() { return n_nodes.fetch_add(2); }

    // bounds of the part of reference r between the planes x[a] = p0 and x[a] = p1: the triangle clipped to the reference's box
    // and the slab (bvh8_geom.h clip_triangle_to_box); other primitives: box ∩ slab
    void clip_ref(const Ref &r, int a, float p0, float p1, float *lo, float *hi) const {
        const BuildPrim &bp = in[r.prim];
        float blo[3], bhi[3];
        for (int c = 0; c < 3; ++c) { blo[c] = r.lo[c]; bhi[c] = r.hi[c]; }
        blo[a] = std::max(blo[a], p0); bhi[a] = std::min(bhi[a], p1);
        if (bp.rec.kind != kPrimKindTriangle) { for (int c = 0; c < 3; ++c) { lo[c] = blo[c]; hi[c] = bhi[c]; } return; }
        clip_triangle_to_box(bp.rec.a, bp.rec.b, bp.rec.c, blo, bhi, lo, hi);
    }

    // budget: references the spatial splits of THIS subtree may still add; a split hands what is left to its children in proportion to
    // their sizes, so the tree does not depend on thread timing and the node array (2 x (n + budget)) cannot overflow
    void build(uint32_t node, std::vector<Ref> refs, int depth, int64_t budget) {
        const uint32_t cnt = (uint32_t)refs.size();
        Box3 nb, cb;
        for (const Ref &r : refs) {
            nb.grow(r.lo, r.hi);
            float c3[3]; for (int c = 0; c < 3; ++c) c3[c] = 0.5f * (r.lo[c] + r.hi[c]);
            cb.grow(c3, c3);
        }
        {
            B2 &n = nodes[node];
            for (int c = 0; c < 3; ++c) { n.lo[c] = nb.lo[c] - pad; n.hi[c] = nb.hi[c] + pad; }
            n.left = n.right = 0; n.first = refs[0].prim; n.count = 0; n.nprims = cnt;
            if (cnt == 1) { n.count = 1; return; }
        }
        const float *clo = cb.lo, *chi = cb.hi;

        // ---- object split: binned SAH over the reference centroids (16 bins x 3 axes), boxes padded ----
        constexpr int NBMAX = 64; const int NB = n_bins;
        float best_cost = kInfF;
        int best_axis = -1, best_bin = -1;
        Box3 best_l, best_r;
        for (int a = 0; a < 3; ++a) {
            const float ext = chi[a] - clo[a];
            if (!(ext > 0.0f)) continue;
            const float scale = (float)NB / ext;
            uint32_t bc[NBMAX] = {0};
            Box3 bb[NBMAX];
            for (const Ref &r : refs) {
                int k = (int)((0.5f * (r.lo[a] + r.hi[a]) - clo[a]) * scale);
                k = std::min(std::max(k, 0), NB - 1);
                bc[k]++;
                float l[3], h[3]; for (int c = 0; c < 3; ++c) { l[c] = r.lo[c] - pad; h[c] = r.hi[c] + pad; }
                bb[k].grow(l, h);
            }
            Box3 racc[NBMAX]; float rarea[NBMAX]; uint32_t rcnt[NBMAX];
            Box3 acc; uint32_t rc = 0;
            for (int k = NB - 1; k > 0; --k) {
                rc += bc[k];
                if (bb[k].valid()) acc.grow(bb[k]);
                racc[k] = acc; rcnt[k] = rc; rarea[k] = rc ? half_area(acc.lo, acc.hi) : 0.0f;
            }
            Box3 lacc; uint32_t lc = 0;
            for (int k = 0; k < NB - 1; ++k) {
                lc += bc[k];
                if (bb[k].valid()) lacc.grow(bb[k]);
                if (lc == 0 || rcnt[k + 1] == 0) continue;
                const float cost = half_area(lacc.lo, lacc.hi) * (float)lc + rarea[k + 1] * (float)rcnt[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; best_l = lacc; best_r = racc[k + 1]; }
            }
        }

        // ---- spatial split candidate ----
        int sp_axis = -1; float sp_pos = 0.0f, sp_cost = kInfF;
        Box3 sp_l, sp_r; uint32_t sp_nl = 0, sp_nr = 0;
        if (spatial && cnt <= spatial_max_refs && cnt >= spatial_min_refs && budget > 0) {
            bool try_spatial = best_axis < 0;
            if (!try_spatial) {
                float il[3], ih[3]; bool overlap = true;
                for (int c = 0; c < 3; ++c) { il[c] = std::max(best_l.lo[c], best_r.lo[c]); ih[c] = std::min(best_l.hi[c], best_r.hi[c]); if (!(il[c] < ih[c])) overlap = false; }
                try_spatial = overlap && half_area(il, ih) > alpha * root_area;
            }
            if (try_spatial) {
                constexpr int NS = 32;
                for (int a = 0; a < 3; ++a) {
                    const float lo_a = nb.lo[a], ext = nb.hi[a] - nb.lo[a];
                    if (!(ext > 0.0f)) continue;
                    const float scale = (float)NS / ext, width = ext / (float)NS;
                    Box3 bb[NS]; uint32_t enter[NS] = {0}, leave[NS] = {0};
                    auto bin_of = [&](float x) { int k = (int)((x - lo_a) * scale); return std::min(std::max(k, 0), NS - 1); };
                    for (const Ref &r : refs) {
                        const int k0 = bin_of(r.lo[a]), k1 = bin_of(r.hi[a]);
                        enter[k0]++; leave[k1]++;
                        if (k0 == k1) { bb[k0].grow(r.lo, r.hi); continue; }
                        for (int k = k0; k <= k1; ++k) {
                            const float p0 = k == k0 ? -kInfF : lo_a + width * (float)k, p1 = k == k1 ? kInfF : lo_a + width * (float)(k + 1);
                            float l[3], h[3];
                            if (fast_binning) { for (int c = 0; c < 3; ++c) { l[c] = r.lo[c]; h[c] = r.hi[c]; } l[a] = std::max(l[a], p0); h[a] = std::min(h[a], p1); }
                            else clip_ref(r, a, p0, p1, l, h);
                            bb[k].grow(l, h);
                        }
                    }
                    Box3 racc[NS]; uint32_t rcnt[NS];
                    Box3 acc; uint32_t rc = 0;
                    for (int k = NS - 1; k > 0; --k) { rc += leave[k]; if (bb[k].valid()) acc.grow(bb[k]); racc[k] = acc; rcnt[k] = rc; }
                    Box3 lacc; uint32_t lc = 0;
                    for (int k = 0; k < NS - 1; ++k) {
                        lc += enter[k];
                        if (bb[k].valid()) lacc.grow(bb[k]);
                        if (lc == 0 || rcnt[k + 1] == 0 || !lacc.valid() || !racc[k + 1].valid()) continue;
                        // padded areas, as everywhere
                        float ll[3], lh[3], rl[3], rh[3];
                        for (int c = 0; c < 3; ++c) { ll[c] = lacc.lo[c] - pad; lh[c] = lacc.hi[c] + pad; rl[c] = racc[k + 1].lo[c] - pad; rh[c] = racc[k + 1].hi[c] + pad; }
                        const float cost = half_area(ll, lh) * (float)lc + half_area(rl, rh) * (float)rcnt[k + 1];
                        if (cost < sp_cost && (lc < cnt || rcnt[k + 1] < cnt)) { sp_cost = cost; sp_axis = a; sp_pos = lo_a + width * (float)(k + 1); sp_l = lacc; sp_r = racc[k + 1]; sp_nl = lc; sp_nr = rcnt[k + 1]; }
                    }
                }
            }
        }

        std::vector<Ref> left, right;
        bool done = false;
        if (sp_axis >= 0 && sp_cost * spatial_bias < best_cost) {
            // ---- spatial split: references wholly on one side go there; a straddling one is cut, or -- when that is cheaper by
            //      the SAH of the two children as they stand -- kept whole on one side ("reference unsplitting") ----
            const int a = sp_axis;
            // the two children as the binning pass found them (every straddling reference cut): B1, B2, N1, N2 of the paper's
            // unsplitting test; a reference moved whole to one side grows that side's box and leaves the other's count
            Box3 lb = sp_l, rb = sp_r; uint32_t nl = sp_nl, nr = sp_nr;
            int64_t added = 0, straddlers = 0;
            for (const Ref &r : refs) if (r.lo[a] < sp_pos && r.hi[a] > sp_pos) ++straddlers;
            if (straddlers <= budget)
            for (const Ref &r : refs) {
                if (r.hi[a] <= sp_pos) { left.push_back(r); continue; }
                if (r.lo[a] >= sp_pos) { right.push_back(r); continue; }
                Box3 l_whole = lb, r_whole = rb;
                l_whole.grow(r.lo, r.hi); r_whole.grow(r.lo, r.hi);
                const float c_split = lb.area() * (float)nl + rb.area() * (float)nr;
                const float c_left = l_whole.area() * (float)nl + rb.area() * (float)(nr - 1u);
                const float c_right = lb.area() * (float)(nl - 1u) + r_whole.area() * (float)nr;
                if (c_split <= c_left && c_split <= c_right) {
                    Ref rl = r, rr = r;
                    clip_ref(r, a, -kInfF, sp_pos, rl.lo, rl.hi);
                    clip_ref(r, a, sp_pos, kInfF, rr.lo, rr.hi);
                    left.push_back(rl); right.push_back(rr); ++added;
                } else if (c_left <= c_right) { left.push_back(r); lb = l_whole; --nr; }
                else { right.push_back(r); rb = r_whole; --nl; }
            }
            if (!left.empty() && !right.empty() && left.size() < cnt + (size_t)added && right.size() < cnt + (size_t)added &&
                (left.size() < cnt || right.size() < cnt)) {
                budget -= added;
                done = true;
            } else { left.clear(); right.clear(); }
        }
        if (!done) {
            uint32_t mid;
            if (best_axis >= 0) {
                const float ext = chi[best_axis] - clo[best_axis];
                const float scale = (float)NB / ext;
                const float c0 = clo[best_axis];
                const int a = best_axis, bb = best_bin;
                auto it = std::partition(refs.begin(), refs.end(), [&](const Ref &r) {
                    int k = (int)((0.5f * (r.lo[a] + r.hi[a]) - c0) * scale);
                    k = std::min(std::max(k, 0), NB - 1);
                    return k <= bb;
                });
                mid = (uint32_t)(it - refs.begin());
            } else {
                mid = cnt / 2;      // all centroids coincide: split by index
            }
            if (mid == 0 || mid == cnt) mid = cnt / 2;
            left.assign(refs.begin(), refs.begin() + mid);
            right.assign(refs.begin() + mid, refs.end());
        }
        std::vector<Ref>().swap(refs);

        const uint32_t l = alloc2();
        nodes[node].left = l; nodes[node].right = l + 1;
        const int64_t budget_l = budget > 0 ? (int64_t)((double)budget * (double)left.size() / (double)(left.size() + right.size())) : 0, budget_r = budget - budget_l;
        if (depth < max_par_depth && cnt > 8192) {
            std::thread t([this, l, depth, budget_l, lv = std::move(left)]() mutable { build(l, std::move(lv), depth + 1, budget_l); });
            build(l + 1, std::move(right), depth + 1, budget_r);
            t.join();
        } else {
            build(l, std::move(left), depth + 1, budget_l);
            build(l + 1, std::move(right), depth + 1, budget_r);
        }
    }
};

inline uint8_t unary_count(uint32_t n) { return (uint8_t)((1u << n) - 1u); }

// cost table of a BVH2 node for the optimal collapse: c[i] = cheapest representation of the subtree as a forest of at most i roots, i = 1..7;
// split[j], j = 2..8 = roots given to the left child when j roots are distributed over the two children
struct Cost { float c[8]; uint8_t leaf1; uint8_t use_dist[8]; uint8_t split[9]; };

}  // namespace

void build_bvh8(const std::vector<BuildPrim> &prims, Bvh8 &out, int threads, float scene_scale, uint32_t max_leaf_prims, bool spatial_splits) {
    max_leaf_prims = std::min(std::max(max_leaf_prims, 1u), kMaxLeafPrims);
    out = Bvh8();
    const uint32_t n = (uint32_t)prims.size();
    for (const auto &p : prims) { if (p.rec.kind == kPrimKindSphere) out.n_spheres++; else out.n_triangles++; }
    if (n == 0) {
        // a single empty node: every ray misses
        Bvh8Node root; std::memset(&root, 0, sizeof root);
        root.e[0] = root.e[1] = root.e[2] = 127;
        for (int a = 0; a < 3; ++a) for (int s = 0; s < 8; ++s) { root.qlo[a][s] = 255; root.qhi[a][s] = 0; }
        out.nodes.push_back(root);
        out.level_begin = {0u, 1u};
        out.node_box.assign(6, 0.0f);
        return;
    }

    Builder B(prims);
    float smax = 1.0f;
    for (uint32_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) smax = std::max(smax, std::max(std::fabs(prims[i].lo[a]), std::fabs(prims[i].hi[a])));
    // Padding keeps the quantised slab test conservative w.r.t. the canonical intersector's
    // own rounding (DESIGN.md "conservative boxes"): 4e-6 of the scene scale.
    if (scene_scale > 0.0f) smax = std::max(1.0f, scene_scale);
    const float pad = 4e-6f * smax;
    out.pad = pad;
    B.pad = pad;
    std::vector<Ref> refs(n);
    Box3 scene_box;
    for (uint32_t i = 0; i < n; ++i) {
        refs[i].prim = i;
        for (int a = 0; a < 3; ++a) { refs[i].lo[a] = prims[i].lo[a]; refs[i].hi[a] = prims[i].hi[a]; }
        scene_box.grow(prims[i].lo, prims[i].hi);
    }
    // spatial splits: the caller's choice (hrt_tlas_build under HRT_CTX_FAST_TRACE), HRT_SBVH = 0 / 1 overrides; at most
    // HRT_SBVH_BUDGET x n extra references (default 1.0, the device builder's; C4 uses 0.47), tried where the object split's children overlap by more than
    // HRT_SBVH_ALPHA of the scene's area (default 1e-5, the paper's)
    B.spatial = spatial_splits && max_leaf_prims == kMaxLeafPrims;
    if (const char *e = std::getenv("HRT_SBVH")) B.spatial = std::atoi(e) != 0 && max_leaf_prims == kMaxLeafPrims;
    double budget_frac = 1.0;
    if (const char *e = std::getenv("HRT_SBVH_BUDGET")) budget_frac = std::max(0.0, std::atof(e));
    if (const char *e = std::getenv("HRT_SBVH_ALPHA")) B.alpha = (float)std::atof(e);
    if (const char *e = std::getenv("HRT_SBVH_BIAS")) B.spatial_bias = (float)std::atof(e);
    const int64_t extra = B.spatial ? (int64_t)std::min<double>(budget_frac * (double)n, 3.0e9 - 2.0 * (double)n) : 0;
    B.root_area = scene_box.area();
    B.nodes.resize(2 * ((size_t)n + (size_t)std::max<int64_t>(extra, 0)) + 2);
    B.n_nodes = 1;
    // Worker threads: the hardware's, at most 64 per build (HRT_BUILD_THREADS overrides) -- several processes may build at once (one rank
    // per GPU, each with its own copy of the scene), and a fork per subtree must not turn into thousands of threads on a shared host
    int hw = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (threads <= 0) hw = std::min(hw, 64);
    if (const char *e = std::getenv("HRT_BUILD_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 1024) hw = v; }
    if (hw < 1) hw = 1;
    B.max_par_depth = 0;
    while ((1 << B.max_par_depth) < hw && B.max_par_depth < 10) B.max_par_depth++;      // (a fork per node down to this depth: at most hw subtree tasks alive)
    const auto t_begin = std::chrono::steady_clock::now();
    B.build(0, std::move(refs), 0, extra);
    const auto t_bvh2 = std::chrono::steady_clock::now();

    for (int a = 0; a < 3; ++a) { out.lo[a] = B.nodes[0].lo[a]; out.hi[a] = B.nodes[0].hi[a]; }

    // ---- optimal collapse: cost tables bottom-up (children always have larger indices than parents) ----
    const uint32_t nb2 = B.n_nodes.load();
    // c_prim measured on MI355X (C4): 0.3 -> 1674, 0.45 -> 1691, 0.6 -> 1691, 1.0 -> 1684 Mrays/s
    float kCNode = 1.0f, kCPrim = 0.45f;
    if (const char *e = std::getenv("HRT_BVH_CPRIM")) kCPrim = (float)std::atof(e);     // tuning experiments
    if (const char *e = std::getenv("HRT_BVH_CNODE")) kCNode = (float)std::atof(e);
    std::vector<Cost> cost(nb2);
    {   // post-order over the BVH2, the big subtrees on worker threads like the build itself (a table depends on its children's only)
        struct Pass {
            const Builder &B; std::vector<Cost> &cost; float c_node, c_prim; uint32_t max_leaf; int par_depth;
            void run(uint32_t ni, int depth) {
                const B2 &bn = B.nodes[ni];
                Cost &cn = cost[ni];
                std::memset(&cn, 0, sizeof cn);
                const float kInf = std::numeric_limits<float>::infinity();
                const float area = half_area(bn.lo, bn.hi);
                const float c_leaf = bn.nprims <= max_leaf ? area * c_prim * (float)bn.nprims : kInf;
                if (bn.count > 0) {                      // BVH2 leaf
                    for (int i = 1; i <= 7; ++i) cn.c[i] = c_leaf;
                    cn.leaf1 = 1;
                    return;
                }
                if (depth < par_depth && bn.nprims > 8192) {
                    std::thread t([this, &bn, depth] { run(bn.left, depth + 1); });
                    run(bn.right, depth + 1);
                    t.join();
                } else { run(bn.left, depth + 1); run(bn.right, depth + 1); }
                const Cost &cl = cost[bn.left], &cr = cost[bn.right];
                float dist[9];
                for (int j = 2; j <= 8; ++j) {
                    float best = kInf; int bk = 1;
                    for (int k = 1; k < j; ++k) {
                        const float v = cl.c[std::min(k, 7)] + cr.c[std::min(j - k, 7)];
                        if (v < best) { best = v; bk = k; }
                    }
                    dist[j] = best; cn.split[j] = (uint8_t)bk;
                }
                const float c_internal = dist[8] + area * c_node;
                cn.leaf1 = c_leaf <= c_internal ? 1 : 0;
                cn.c[1] = std::min(c_leaf, c_internal);
                for (int i = 2; i <= 7; ++i) {
                    if (dist[i] < cn.c[i - 1]) { cn.c[i] = dist[i]; cn.use_dist[i] = 1; }
                    else { cn.c[i] = cn.c[i - 1]; cn.use_dist[i] = 0; }
                }
            }
        } pass{B, cost, kCNode, kCPrim, max_leaf_prims, B.max_par_depth};
        pass.run(0, 0);
    }
    // roots of the forest that represents subtree n with a budget of j roots (as chosen by the tables)
    struct Collector {
        const Builder &B; const std::vector<Cost> &cost; uint32_t out[8]; int n = 0;
        void forest(uint32_t x, int i) {            // at most i roots for subtree x
            while (i > 1 && !cost[x].use_dist[i]) --i;
            if (i == 1 || B.nodes[x].count > 0) { out[n++] = x; return; }
            distribute(x, i);
        }
        void distribute(uint32_t x, int j) {        // subtree x opened: j roots over its two children
            const int k = cost[x].split[j];
            forest(B.nodes[x].left, std::min(k, 7));
            forest(B.nodes[x].right, std::min(j - k, 7));
        }
    };

    // ---- emit 8-wide nodes breadth first so that inner children are contiguous: level by level, the nodes of a level in parallel
    //      (children, slots, quantised boxes, leaf records), a serial prefix in between for the child / primitive bases -- the
    //      numbering is that of a breadth-first queue ----
    struct Local {
        Bvh8Node nd; uint32_t inner_b2[8]; uint32_t n_inner = 0, n_leaf = 0;
        uint32_t leaf_prim[8 * kMaxLeafPrims]; const B2 *leaf_box[8 * kMaxLeafPrims];
    };
    auto emit_node = [&](uint32_t b2, Local &L) {
        const B2 &bn = B.nodes[b2];
        uint32_t ch[8]; int nch = 0;
        bool ch_leaf[8];
        if (bn.count > 0 || (b2 == 0 && bn.nprims <= max_leaf_prims)) {
            ch[nch] = b2; ch_leaf[nch] = true; ++nch;         // the whole scene fits one leaf: wrap it
        } else {
            Collector col{B, cost};
            col.distribute(b2, 8);
            for (int k = 0; k < col.n; ++k) {
                ch[nch] = col.out[k];
                ch_leaf[nch] = B.nodes[col.out[k]].count > 0 || cost[col.out[k]].leaf1;
                ++nch;
            }
        }
        // slot assignment: slot s is visited first by rays of octant s (bit2 = -x, bit1 = -y, bit0 = -z)
        float ncx[3];
        for (int a = 0; a < 3; ++a) ncx[a] = 0.5f * (bn.lo[a] + bn.hi[a]);
        float scost[8][8];
        for (int k = 0; k < nch; ++k) {
            const B2 &c = B.nodes[ch[k]];
            const float d[3] = {0.5f * (c.lo[0] + c.hi[0]) - ncx[0], 0.5f * (c.lo[1] + c.hi[1]) - ncx[1], 0.5f * (c.lo[2] + c.hi[2]) - ncx[2]};
            for (int s = 0; s < 8; ++s) {
                const float sx = (s & 4) ? -1.0f : 1.0f, sy = (s & 2) ? -1.0f : 1.0f, sz = (s & 1) ? -1.0f : 1.0f;
                scost[k][s] = d[0] * sx + d[1] * sy + d[2] * sz;
            }
        }
        int slot_child[8]; for (int s = 0; s < 8; ++s) slot_child[s] = -1;
        bool child_done[8] = {false, false, false, false, false, false, false, false};
        for (int round = 0; round < nch; ++round) {
            int bk = -1, bs = -1; float bc = std::numeric_limits<float>::infinity();
            for (int k = 0; k < nch; ++k) {
                if (child_done[k]) continue;
                for (int s = 0; s < 8; ++s) {
                    if (slot_child[s] >= 0) continue;
                    if (scost[k][s] < bc) { bc = scost[k][s]; bk = k; bs = s; }
                }
            }
            if (bk < 0) {   // NaN costs (degenerate boxes): first free pair
                for (int k = 0; k < nch && bk < 0; ++k) if (!child_done[k]) bk = k;
                for (int s = 0; s < 8 && bs < 0; ++s) if (slot_child[s] < 0) bs = s;
            }
            slot_child[bs] = bk; child_done[bk] = true;
        }
        Bvh8Node &nd = L.nd; std::memset(&nd, 0, sizeof nd);
        for (int a = 0; a < 3; ++a) {
            nd.p[a] = bn.lo[a];
            nd.e[a] = node_exponent(bn.hi[a] - bn.lo[a]);
        }
        L.n_inner = L.n_leaf = 0;
        for (int s = 0; s < 8; ++s) {
            const int k = slot_child[s];
            if (k < 0) {
                nd.meta[s] = 0;
                for (int a = 0; a < 3; ++a) { nd.qlo[a][s] = 255; nd.qhi[a][s] = 0; }
                continue;
            }
            const B2 &c = B.nodes[ch[k]];
            for (int a = 0; a < 3; ++a) quantise_axis(nd.p[a], nd.e[a], c.lo[a], c.hi[a], &nd.qlo[a][s], &nd.qhi[a][s]);
            if (ch_leaf[k]) {
                // the <= 3 references of the leaf: walk its little BVH2 subtree, left first; the same primitive twice (both
                // halves of a spatial split ended up here) is stored once
                uint32_t st[8]; int sp = 0; st[sp++] = ch[k];
                const uint32_t first = L.n_leaf; uint32_t nl = 0;
                while (sp > 0) {
                    const B2 &x = B.nodes[st[--sp]];
                    if (x.count > 0) {
                        bool dup = false;
                        for (uint32_t q = 0; q < nl; ++q) if (L.leaf_prim[first + q] == x.first) dup = true;
                        if (!dup && nl < kMaxLeafPrims) { L.leaf_prim[first + nl] = x.first; L.leaf_box[first + nl] = &x; ++nl; }
                    } else { st[sp++] = x.right; st[sp++] = x.left; }
                }
                nd.meta[s] = (uint8_t)((unary_count(nl) << 5) | first);
                L.n_leaf += nl;
            } else {
                nd.meta[s] = (uint8_t)(0x20u | (24u + (uint32_t)s));
                nd.imask |= (uint8_t)(1u << s);
                L.inner_b2[L.n_inner++] = ch[k];
            }
        }
    };
    auto parallel_for = [&](size_t count, const std::function<void(size_t, size_t)> &body) {
        const size_t T = std::min<size_t>((size_t)hw, count / 512 + 1);
        if (T <= 1) { body(0, count); return; }
        std::vector<std::thread> pool;
        for (size_t t = 0; t < T; ++t) pool.emplace_back(body, count * t / T, count * (t + 1) / T);
        for (auto &th : pool) th.join();
    };
    std::vector<uint32_t> level{0u}, next_level, child_base, prim_base;
    std::vector<Local> locals;
    uint32_t level_first = 0, depth = 0;
    while (!level.empty()) {
        const size_t n_level = level.size();
        locals.resize(n_level); child_base.resize(n_level); prim_base.resize(n_level);
        parallel_for(n_level, [&](size_t b, size_t e) { for (size_t i = b; i < e; ++i) emit_node(level[i], locals[i]); });
        uint32_t next_node = level_first + (uint32_t)n_level, next_prim = (uint32_t)out.prims.size();
        for (size_t i = 0; i < n_level; ++i) {
            child_base[i] = next_node; next_node += locals[i].n_inner;
            prim_base[i] = next_prim; next_prim += locals[i].n_leaf;
        }
        out.level_begin.push_back(level_first);
        out.max_depth = depth;
        out.nodes.resize(level_first + n_level);
        out.node_box.resize(6 * (size_t)(level_first + n_level));
        out.prims.resize(next_prim); out.prim_bounds.resize(6 * (size_t)next_prim);
        next_level.assign(next_node - (level_first + n_level), 0u);
        parallel_for(n_level, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {
                Local &L = locals[i];
                const B2 &bn = B.nodes[level[i]];
                L.nd.child_base = child_base[i]; L.nd.prim_base = prim_base[i];
                out.nodes[level_first + i] = L.nd;
                for (int a = 0; a < 3; ++a) { out.node_box[6 * (level_first + i) + a] = bn.lo[a]; out.node_box[6 * (level_first + i) + 3 + a] = bn.hi[a]; }
                for (uint32_t q = 0; q < L.n_leaf; ++q) {
                    out.prims[prim_base[i] + q] = prims[L.leaf_prim[q]].rec;
                    // (host-side checks: the part of the primitive this leaf answers for, unpadded side of the reference's box)
                    for (int a = 0; a < 3; ++a) { out.prim_bounds[6 * (size_t)(prim_base[i] + q) + a] = L.leaf_box[q]->lo[a] + pad; out.prim_bounds[6 * (size_t)(prim_base[i] + q) + 3 + a] = L.leaf_box[q]->hi[a] - pad; }
                }
                for (uint32_t r = 0; r < L.n_inner; ++r) next_level[child_base[i] - (level_first + n_level) + r] = L.inner_b2[r];
            }
        });
        level_first += (uint32_t)n_level;
        level.swap(next_level);
        if (!level.empty()) ++depth;
    }
    out.level_begin.push_back((uint32_t)out.nodes.size());
    if (std::getenv("HRT_BUILD_VERBOSE")) {
        const auto t_end = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[hrt] host build: %u primitives -> %zu references, %zu nodes (%s): BVH2 %.3f s on up to %d threads, collapse + emission %.3f s\n", n, out.prims.size(), out.nodes.size(),
                     B.spatial ? "spatial splits" : "object splits only", std::chrono::duration<double>(t_bvh2 - t_begin).count(), hw, std::chrono::duration<double>(t_end - t_bvh2).count());
    }

    // refit quality reference: per node {weight, 1 / built half area}; weight = primitives below the node,
    // normalised so that the weighted mean of area_now / area_built is 1 for the tree as built (refit.hip)
    const size_t nn = out.nodes.size();
    std::vector<double> below(nn, 0.0);
    double total = 0.0;
    for (size_t i = nn; i-- > 0;) {
        const Bvh8Node &nd = out.nodes[i];
        uint32_t rank = 0;
        for (int s = 0; s < 8; ++s) {
            const uint8_t m = nd.meta[s];
            if (m == 0) continue;
            if ((nd.imask >> s) & 1u) below[i] += below[nd.child_base + rank++];
            else { const uint32_t cb = m >> 5; below[i] += cb == 1 ? 1.0 : cb == 3 ? 2.0 : 3.0; }
        }
    }
    out.node_ref.assign(2 * nn, 0.0f);
    for (size_t i = 0; i < nn; ++i) {
        const float ar = box_half_area(&out.node_box[6 * i], &out.node_box[6 * i + 3]);
        if (ar > 0.0f && std::isfinite(ar)) { total += below[i]; out.node_ref[2 * i + 1] = 1.0f / ar; } else below[i] = 0.0;
    }
    for (size_t i = 0; i < nn; ++i) out.node_ref[2 * i] = total > 0.0 ? (float)(below[i] / total) : 0.0f;
}

// ---- tree over instances -------------------------------------------------------------------------------------
void assemble_instanced_bvh8(const std::vector<const Bvh8 *> &tmpl, const std::vector<float> &box, InstancedTree &out) {
    out = InstancedTree();
    const uint32_t n_inst = (uint32_t)tmpl.size();
    // the top tree: one "primitive" per contributing instance, one instance per leaf slot
    std::vector<BuildPrim> tp;
    std::vector<uint32_t> inst_of;
    for (uint32_t i = 0; i < n_inst; ++i) {
        if (!tmpl[i] || tmpl[i]->prims.empty()) continue;
        BuildPrim bp; std::memset(&bp, 0, sizeof bp);
        for (int a = 0; a < 3; ++a) { bp.lo[a] = box[6 * (size_t)i + a]; bp.hi[a] = box[6 * (size_t)i + 3 + a]; }
        bp.rec.prim = (uint32_t)inst_of.size();
        tp.push_back(bp); inst_of.push_back(i);
    }
    Bvh8 top;
    build_bvh8(tp, top, 0, 0.0f, 1);
    if (tp.empty()) { out.nodes = top.nodes; out.order = {0u}; out.phase_begin = {0u, 1u}; out.weight = {0.0f}; return; }

    // Re-emit the top tree breadth first with the instance roots inside the child blocks: a node's children --
    // top nodes and instance roots alike -- are contiguous in slot order, as the traversal expects.
    struct Pending { uint32_t top_node; uint32_t out_index; uint32_t depth; };
    std::vector<Pending> queue{{0u, 0u, 0u}};
    struct RootRef { uint32_t out_index; uint32_t inst; uint32_t depth; };
    std::vector<RootRef> roots;
    out.nodes.emplace_back();
    for (size_t head = 0; head < queue.size(); ++head) {
        const Pending it = queue[head];
        Bvh8Node nd = top.nodes[it.top_node];
        const uint32_t child_base = (uint32_t)out.nodes.size();
        uint32_t rank_in = 0;
        for (int s = 0; s < 8; ++s) {
            const uint8_t m = nd.meta[s];
            if (m == 0) continue;
            const uint32_t at = (uint32_t)out.nodes.size();
            out.nodes.emplace_back();
            if ((nd.imask >> s) & 1u) queue.push_back({top.nodes[it.top_node].child_base + rank_in++, at, it.depth + 1});
            else {
                const uint32_t k = top.prims[nd.prim_base + (m & 0x1fu)].prim;       // one instance per leaf slot
                roots.push_back({at, inst_of[k], it.depth + 1});
                nd.meta[s] = (uint8_t)(0x20u | (24u + (uint32_t)s));
                nd.imask |= (uint8_t)(1u << s);
            }
        }
        nd.child_base = child_base; nd.prim_base = 0;
        out.nodes[it.out_index] = nd;
    }
    // instance bodies and primitives
    for (const RootRef &r : roots) {
        const Bvh8 &t = *tmpl[r.inst];
        const uint32_t body = (uint32_t)out.nodes.size(), pbase = (uint32_t)out.prims.size();
        for (size_t k = 0; k < t.nodes.size(); ++k) {
            Bvh8Node nd = t.nodes[k];
            nd.child_base = body + (nd.child_base - 1u);            // template node j > 0 lives at body + j - 1
            nd.prim_base = pbase + nd.prim_base;
            if (k == 0) out.nodes[r.out_index] = nd; else out.nodes.push_back(nd);
        }
        for (PrimRecord pr : t.prims) { pr.inst = r.inst; out.prims.push_back(pr); }
        out.n_triangles += t.n_triangles; out.n_spheres += t.n_spheres;
        out.max_depth = std::max(out.max_depth, r.depth + t.max_depth);
    }
    // heights and primitive counts bottom-up: every child has a larger index than its parent
    const size_t nn = out.nodes.size();
    std::vector<uint32_t> height(nn, 0);
    std::vector<double> below(nn, 0.0);
    uint32_t max_h = 0; double total = 0.0;
    for (size_t i = nn; i-- > 0;) {
        const Bvh8Node &nd = out.nodes[i];
        uint32_t rank = 0;
        for (int s = 0; s < 8; ++s) {
            const uint8_t m = nd.meta[s];
            if (m == 0) continue;
            if ((nd.imask >> s) & 1u) { const uint32_t c = nd.child_base + rank++; height[i] = std::max(height[i], height[c] + 1u); below[i] += below[c]; }
            else { const uint32_t cb = m >> 5; below[i] += cb == 1 ? 1.0 : cb == 3 ? 2.0 : 3.0; }
        }
        max_h = std::max(max_h, height[i]); total += below[i];
    }
    out.phase_begin.assign(max_h + 2, 0u);
    for (size_t i = 0; i < nn; ++i) out.phase_begin[height[i] + 1]++;
    for (uint32_t h = 0; h <= max_h; ++h) out.phase_begin[h + 1] += out.phase_begin[h];
    out.order.resize(nn);
    std::vector<uint32_t> cursor(out.phase_begin.begin(), out.phase_begin.end() - 1);
    for (size_t i = 0; i < nn; ++i) out.order[cursor[height[i]]++] = (uint32_t)i;
    out.weight.resize(nn);
    for (size_t i = 0; i < nn; ++i) out.weight[i] = total > 0.0 ? (float)(below[i] / total) : 0.0f;
}

// Walk the packed tree and check containment of every primitive in every ancestor slot box.
const char *validate_bvh8(const Bvh8 &bvh) {
    if (bvh.nodes.empty()) return "no nodes";
    struct E { uint32_t node; };
    std::vector<uint32_t> stack{0u};
    std::vector<uint8_t> seen(bvh.prims.size(), 0);
    // per node: each leaf slot's prims must be inside the slot box; the true bounds of all
    // primitives below an inner slot must be inside that slot's box (nlo/nhi = true bounds).
    std::vector<float> nlo(3 * bvh.nodes.size()), nhi(3 * bvh.nodes.size());
    // bottom-up boxes: nodes are in BFS order, so children have larger indices
    for (size_t i = bvh.nodes.size(); i-- > 0;) {
        const Bvh8Node &nd = bvh.nodes[i];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        uint32_t inner_rank = 0;
        for (int s = 0; s < 8; ++s) {
            const uint8_t m = nd.meta[s];
            if (m == 0) continue;
            float blo[3], bhi[3];
            for (int a = 0; a < 3; ++a) {
                const float sc = std::ldexp(1.0f, (int)nd.e[a] - 127);
                blo[a] = nd.p[a] + (float)nd.qlo[a][s] * sc;
                bhi[a] = nd.p[a] + (float)nd.qhi[a][s] * sc;
            }
            const bool inner = (nd.imask >> s) & 1u;
            if (inner) {
                if ((m & 0xe0u) != 0x20u || (m & 0x1fu) != 24u + (uint32_t)s) return "inner meta mismatch";
                const uint32_t c = nd.child_base + inner_rank++;
                if (c >= bvh.nodes.size() || c <= i) return "child index out of order";
                for (int a = 0; a < 3; ++a) {
                    if (nlo[3 * c + a] < blo[a] || nhi[3 * c + a] > bhi[a]) return "subtree primitives escape slot box";
                    lo[a] = std::min(lo[a], nlo[3 * c + a]); hi[a] = std::max(hi[a], nhi[3 * c + a]);
                }
            } else {
                const uint32_t cntbits = m >> 5, off = m & 0x1fu;
                const uint32_t cnt = cntbits == 1 ? 1 : cntbits == 3 ? 2 : cntbits == 7 ? 3 : 0;
                if (cnt == 0 || off + cnt > 24) return "leaf meta malformed";
                for (uint32_t k = 0; k < cnt; ++k) {
                    const uint32_t p = nd.prim_base + off + k;
                    if (p >= bvh.prims.size()) return "prim index out of range";
                    if (seen[p]) return "primitive referenced twice";
                    seen[p] = 1;
                    for (int a = 0; a < 3; ++a) {
                        if (bvh.prim_bounds[6 * p + a] < blo[a] || bvh.prim_bounds[6 * p + 3 + a] > bhi[a]) return "primitive escapes leaf box";
                        lo[a] = std::min(lo[a], bvh.prim_bounds[6 * p + a]); hi[a] = std::max(hi[a], bvh.prim_bounds[6 * p + 3 + a]);
                    }
                }
            }
        }
        for (int a = 0; a < 3; ++a) { nlo[3 * i + a] = lo[a]; nhi[3 * i + a] = hi[a]; }
    }
    for (size_t p = 0; p < seen.size(); ++p) if (!seen[p]) return "primitive not referenced";
    return "";
}

}  // namespace hrt
