// build.hip -- acceleration-structure build on the device (gfx950): what optixAccelBuild does for the reference
// (src/Global/RendererImpl.cu:30-88 buildASImpl, :89-172 buildGASFor*, :174-208 buildIAS), without a host copy of the geometry.
//
//   1. world-space bounds of every primitive of every visible instance (same transform arithmetic as the refit: bvh8_geom.h),
//      scene centroid bounds by wave reduction + ordered-integer atomics;
//   2. 63-bit Morton codes of the centroids, radix sort (hipCUB);
//   0. for scenes of more than 4096 primitives: the top-down phase (build_split.hip: binned-SAH object splits level by level -- and
//      spatial splits under HRT_CTX_FAST_TRACE --), after which the leaves are references grouped in cells of a few each and PLOC merges
//      within cells only.  PLOC alone shapes a soup well enough, but not separate bodies over a huge ground sphere, the reference's
//      kind of scene: 39 node visits per ray where the top-down phase leaves 5.5 (profiles/r03_particle_scene_trees.txt);
//   3. PLOC (Meister & Bittner 2018): clusters in Morton order repeatedly merge with their nearest neighbour (smallest
//      merged surface area within +-2 positions) when the choice is mutual -- a BVH2 of near-SAH quality in ~40 rounds,
//      every round a handful of O(n) launches; node numbers come from prefix sums, so the BVH2 is deterministic.  The last rounds
//      (4096 clusters or fewer: all rounds of a small build) are one workgroup's, a barrier where the others have a launch;
//   4. the optimal SAH collapse to 8-wide nodes (Ylitie, Karras, Laine 2017, sec. 4.1: the same cost tables as the host
//      builder bvh8_build.cpp): a node's table is computed where the node is made (its children are from earlier rounds);
//   5. emission of the packed BVH8 breadth first, one launch per level (a small build: one workgroup for all levels), eight lanes
//      per node: forest roots from the tables, octant slot assignment, child / primitive blocks from atomic cursors -- topology
//      and primitive ids only;
//   6. the refit kernels (refit.hip) then compute every world-space record, box, origin, exponent and quantised child box
//      bottom-up, exactly as they do after an instance update: one arithmetic for build and refit.
// The result does not depend on the tree: hits are defined by the canonical intersector over conservative boxes.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <cstdio>
#include <vector>
#include "build_dev.h"

#pragma clang fp contract(off)

namespace hrt {

namespace {

constexpr int kPlocMaxRadius = 128;
constexpr uint32_t kMaxSmallLevels = 40u;

__global__ __launch_bounds__(1024) void k_prim_bounds(GpuBuildArgs a) {
    const uint32_t k = blockIdx.x * 1024u + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool valid = false;
    if (k < a.n) {
        valid = prim_bounds(a, k, lo, hi);
        a.pb_lo[k] = make_float4(lo[0], lo[1], lo[2], valid ? 1.0f : 0.0f);
        a.pb_hi[k] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    }
    // centroid bounds and box bounds of the valid primitives: block reduction, then 12 atomics per block
    float mn[6], mx[6];
    for (int q = 0; q < 3; ++q) {
        const float ce = 0.5f * (lo[q] + hi[q]);
        mn[q] = valid ? ce : INFINITY; mx[q] = valid ? ce : -INFINITY;
        mn[3 + q] = valid ? lo[q] : INFINITY; mx[3 + q] = valid ? hi[q] : -INFINITY;
    }
    const uint32_t n_bad = (uint32_t)__popcll(__ballot(k < a.n && !valid));
    if ((threadIdx.x & 63u) == 0u && n_bad) atomicAdd(&a.counters->n_invalid, n_bad);
    block_minmax<6>(mn, mx);
    if (threadIdx.x == 0u && mn[0] <= mx[0])
        for (int q = 0; q < 3; ++q) {
            atomicMin(&a.counters->cmin[q], f2ord(mn[q])); atomicMax(&a.counters->cmax[q], f2ord(mx[q]));
            atomicMin(&a.counters->bmin[q], f2ord(mn[3 + q])); atomicMax(&a.counters->bmax[q], f2ord(mx[3 + q]));
        }
}

__device__ __forceinline__ uint64_t spread21(uint64_t x) {      // bit i -> bit 3i
    x &= 0x1fffffull;
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(256) void k_morton(GpuBuildArgs a) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= a.n) return;
    const float4 lo = a.pb_lo[k], hi = a.pb_hi[k];
    uint64_t key = ~0ull;                                      // invalid primitives sort to the end
    if (lo.w != 0.0f) {
        const float cm[3] = {ord2f(a.counters->cmin[0]), ord2f(a.counters->cmin[1]), ord2f(a.counters->cmin[2])};
        const float cx[3] = {ord2f(a.counters->cmax[0]), ord2f(a.counters->cmax[1]), ord2f(a.counters->cmax[2])};
        const float ce[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
        uint64_t q[3];
        for (int d = 0; d < 3; ++d) {
            const float ext = cx[d] - cm[d];
            float t = ext > 0.0f ? (ce[d] - cm[d]) / ext : 0.0f;
            t = fminf(fmaxf(t, 0.0f), 1.0f);
            const uint32_t v = (uint32_t)(t * 2097151.0f);
            q[d] = v > 2097151u ? 2097151u : v;
        }
        key = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    }
    a.keys[k] = key; a.vals[k] = k;
}

// ---- optimal collapse: cost tables (bvh8_build.cpp, same recurrences) ----
// table of a node: c[1..7] as floats in t[1..7]; t[0] bits: leaf1 | use_dist[i] << i (i = 2..7) | split[j] << (8 + 3 (j - 2)) (j = 2..8)
struct CostRow { float c[8]; };
__device__ __forceinline__ uint32_t row_bits(const CostRow &r) { return __float_as_uint(r.c[0]); }
__device__ __forceinline__ uint32_t row_split(uint32_t bits, int j) { return (bits >> (8 + 3 * (j - 2))) & 7u; }

__device__ __forceinline__ float node_half_area(const float4 lo, const float4 hi) {
    const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    return dx * dy + dy * dz + dz * dx;
}

__device__ void cost_of_node(const GpuBuildArgs &a, uint32_t nd, bool leaf, CostRow &out) {
    const float kInf = INFINITY;
    // (a box without area -- coinciding points, a line of them -- still costs a visit: with zero everywhere the tables tie and the tree comes out
    // nearly binary, 15 levels for 5000 points; any positive area leaves everything else as it was)
    const float area = fmaxf(node_half_area(a.node_lo[nd], a.node_hi[nd]), 1e-30f);
    const uint32_t np = a.node_nprims[nd];
    const float c_leaf = np <= a.max_leaf_prims ? area * a.c_prim * (float)np : kInf;
    if (leaf) {
        for (int i = 1; i <= 7; ++i) out.c[i] = c_leaf;
        out.c[0] = __uint_as_float(1u);
        return;
    }
    const uint32_t l = __float_as_uint(a.node_lo[nd].w), r = __float_as_uint(a.node_hi[nd].w);
    CostRow cl, cr;
    {   // the children's rows: written by an earlier launch
        const float4 *pl = reinterpret_cast<const float4 *>(a.cost + 8 * (size_t)l), *pr = reinterpret_cast<const float4 *>(a.cost + 8 * (size_t)r);
        const float4 l0 = pl[0], l1 = pl[1], r0 = pr[0], r1 = pr[1];
        cl.c[0] = l0.x; cl.c[1] = l0.y; cl.c[2] = l0.z; cl.c[3] = l0.w; cl.c[4] = l1.x; cl.c[5] = l1.y; cl.c[6] = l1.z; cl.c[7] = l1.w;
        cr.c[0] = r0.x; cr.c[1] = r0.y; cr.c[2] = r0.z; cr.c[3] = r0.w; cr.c[4] = r1.x; cr.c[5] = r1.y; cr.c[6] = r1.z; cr.c[7] = r1.w;
    }
    float dist[9]; uint32_t bits = 0u;
    for (int j = 2; j <= 8; ++j) {
        float best = kInf; int bk = 1;
        for (int k = 1; k < j; ++k) {
            const float v = cl.c[k < 7 ? k : 7] + cr.c[(j - k) < 7 ? (j - k) : 7];
            if (v < best) { best = v; bk = k; }
        }
        dist[j] = best; bits |= (uint32_t)bk << (8 + 3 * (j - 2));
    }
    const float c_internal = dist[a.width] + area * a.c_node;      // (a.width children at most: 8, or fewer as an experiment -- HRT_BVH_WIDTH)
    if (c_leaf <= c_internal) bits |= 1u;
    out.c[1] = fminf(c_leaf, c_internal);
    for (int i = 2; i <= 7; ++i) {
        if (dist[i] < out.c[i - 1]) { out.c[i] = dist[i]; bits |= 1u << i; } else out.c[i] = out.c[i - 1];
    }
    out.c[0] = __uint_as_float(bits);
}

__device__ __forceinline__ void store_row(const GpuBuildArgs &a, uint32_t nd, const CostRow &r) {
    float4 *p = reinterpret_cast<float4 *>(a.cost + 8 * (size_t)nd);
    p[0] = make_float4(r.c[0], r.c[1], r.c[2], r.c[3]); p[1] = make_float4(r.c[4], r.c[5], r.c[6], r.c[7]);
}

// builds with spatial splits: the sort key of a reference is its cell, then the Morton code of its centroid within the cell's
// centroid bounds (10 bits an axis: cells are small) -- the references of a cell stay together and PLOC works cell by cell
__global__ __launch_bounds__(256) void k_morton_refs(GpuBuildArgs a, uint32_t n_refs, const SplitSeg *__restrict__ segs, const uint32_t *__restrict__ cell_seg) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_refs) return;
    const float4 lo = a.ref_lo[i], hi = a.ref_hi[i];
    const uint32_t cell = __float_as_uint(hi.w);
    const SplitSeg &sg = segs[cell_seg[cell]];
    const float ce[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
    uint64_t q[3];
    for (int d = 0; d < 3; ++d) {
        const float cm = ord2f(sg.cb[d]), cx = -ord2f(sg.cb[3 + d]);
        const float ext = cx - cm;
        float t = ext > 0.0f ? (ce[d] - cm) / ext : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        const uint32_t v = (uint32_t)(t * 1023.0f);
        q[d] = v > 1023u ? 1023u : v;
    }
    a.keys[i] = ((uint64_t)cell << 30) | (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    a.vals[i] = i;
}

__global__ __launch_bounds__(256) void k_leaves_refs(GpuBuildArgs a, uint32_t n_refs) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_refs) return;
    const uint32_t r = a.vals_sorted[i];
    const float4 lo = a.ref_lo[r], hi = a.ref_hi[r];
    a.node_lo[i] = make_float4(lo.x, lo.y, lo.z, __uint_as_float(kNone));
    a.node_hi[i] = make_float4(hi.x, hi.y, hi.z, lo.w);
    a.node_cell[i] = __float_as_uint(hi.w);
    a.node_nprims[i] = 1u;
    a.cl_a[i] = i;
    CostRow row; cost_of_node(a, i, true, row); store_row(a, i, row);
}

// the segments the top-down phase split are the top of the BVH2: node top_base + t for top node t, children = top nodes or the
// roots PLOC left of the cells (cell_root, in cell order).  (A top node's reference count is the segment's before its split.)
__global__ __launch_bounds__(256) void k_top_link(GpuBuildArgs a, const SplitSeg *__restrict__ segs, const uint32_t *__restrict__ top_seg,
                                                  const uint32_t *__restrict__ cell_root, uint32_t n_top, uint32_t top_base) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_top) return;
    const SplitSeg &sg = segs[top_seg[t]];
    const uint32_t id = top_base + t;
    uint32_t ch[2];
    for (int c = 0; c < 2; ++c) {
        const SplitSeg &cs = segs[sg.child + (uint32_t)c];
        ch[c] = cs.kind >= 2u ? top_base + cs.index : cell_root[cs.index];
    }
    a.node_lo[id] = make_float4(ord2f(sg.nb[0]), ord2f(sg.nb[1]), ord2f(sg.nb[2]), __uint_as_float(ch[0]));
    a.node_hi[id] = make_float4(-ord2f(sg.nb[3]), -ord2f(sg.nb[4]), -ord2f(sg.nb[5]), __uint_as_float(ch[1]));
    a.node_nprims[id] = sg.count;
}

// BVH2 node i: lo.xyz | left, hi.xyz | right.  Leaf: left = kNone, right = global primitive number.
// (nv = kNone: a small build, which does not stop for the host to learn the number of valid primitives -- taken from the counters, and
// the PLOC counters are set here)
__global__ __launch_bounds__(256) void k_leaves(GpuBuildArgs a, uint32_t nv_host) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t nv = nv_host != kNone ? nv_host : a.n - a.counters->n_invalid;
    if (i == 0u && nv_host == kNone) { BuildCounters *c = a.counters; c->m_cur = nv; c->m_next = nv; c->node_base = nv; c->merges = 0u; }
    if (i >= nv) return;
    const uint32_t k = a.vals_sorted[i];
    const float4 lo = a.pb_lo[k], hi = a.pb_hi[k];
    a.node_lo[i] = make_float4(lo.x, lo.y, lo.z, __uint_as_float(kNone));
    a.node_hi[i] = make_float4(hi.x, hi.y, hi.z, __uint_as_float(k));
    a.node_nprims[i] = 1u;
    a.cl_a[i] = i;
    CostRow row; cost_of_node(a, i, true, row); store_row(a, i, row);
}

__device__ __forceinline__ float merged_half_area(const float4 alo, const float4 ahi, const float4 blo, const float4 bhi) {
    const float ex = fmaxf(ahi.x, bhi.x) - fminf(alo.x, blo.x), ey = fmaxf(ahi.y, bhi.y) - fminf(alo.y, blo.y),
                ez = fmaxf(ahi.z, bhi.z) - fminf(alo.z, blo.z);
    return ex * ey + ey * ez + ez * ex;
}

// nearest neighbour of every cluster within +-ploc_radius positions (ties: the lower position)
// (the PLOC kernels take the number of clusters from the device counters: the host launches several rounds over an upper bound of
// it before it reads the counters back)
__global__ __launch_bounds__(256) void k_ploc_nn(GpuBuildArgs a, const uint32_t *__restrict__ cl) {
    __shared__ float4 s_lo[256 + 2 * kPlocMaxRadius], s_hi[256 + 2 * kPlocMaxRadius];
    __shared__ uint32_t s_cell[256 + 2 * kPlocMaxRadius];      // (builds with spatial splits: clusters merge within their cell only)
    const uint32_t m = a.counters->m_cur;
    if (blockIdx.x * 256u >= m) return;
    const int kPlocRadius = a.ploc_radius;
    const int base = (int)(blockIdx.x * 256u) - kPlocRadius;
    for (int t = (int)threadIdx.x; t < 256 + 2 * kPlocRadius; t += 256) {
        const int j = base + t;
        if (j >= 0 && j < (int)m) { const uint32_t nd = cl[j]; s_lo[t] = a.node_lo[nd]; s_hi[t] = a.node_hi[nd]; s_cell[t] = a.node_cell ? a.node_cell[nd] : 0u; }
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const int me = (int)threadIdx.x + kPlocRadius;
    const float4 mlo = s_lo[me], mhi = s_hi[me];
    const uint32_t my_cell = s_cell[me];
    float best = INFINITY; uint32_t bj = kNone;
    for (int d = -kPlocRadius; d <= kPlocRadius; ++d) {
        const int j = (int)i + d;
        if (d == 0 || j < 0 || j >= (int)m || s_cell[me + d] != my_cell) continue;
        const float ar = merged_half_area(mlo, mhi, s_lo[me + d], s_hi[me + d]);
        // strict <: the first (lowest) position wins ties -- except that position i ^ 1 beats any other of the same area: among thousands of
        // coinciding primitives every cluster would otherwise choose the lowest position in reach, one pair a round would be mutual and the
        // tree a chain (6000 copies of a triangle: depth 860); with the pairing rule they halve every round.  NaN areas still pick something
        if (ar < best || bj == kNone || (ar == best && (uint32_t)j == (i ^ 1u))) { best = ar; bj = (uint32_t)j; }
    }
    if (a.balanced) bj = (i ^ 1u) < m ? (i ^ 1u) : kNone;      // by position: the clusters halve every round
    a.nn[i] = bj;
}

// per position: valid << 32 | starts-a-merge
__global__ __launch_bounds__(256) void k_ploc_flags(GpuBuildArgs a, uint32_t m_bound) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t m = a.counters->m_cur;
    if (i >= m) { if (i < m_bound) a.flags[i] = 0ull; return; }      // (the scan runs over the bound)
    const uint32_t j = a.nn[i];
    const bool mutual = j != kNone && a.nn[j] == i;
    uint64_t f = 1ull << 32;
    if (mutual) f = i < j ? ((1ull << 32) | 1ull) : 0ull;
    a.flags[i] = f;
}

__global__ __launch_bounds__(256) void k_ploc_apply(GpuBuildArgs a, const uint32_t *__restrict__ cl_in, uint32_t *__restrict__ cl_out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t m = a.counters->m_cur, node_base = a.counters->node_base;
    if (i >= m) return;
    const uint64_t f = a.flags[i], sc = a.scan[i];
    if (i == m - 1u) { a.counters->m_next = (uint32_t)((sc + f) >> 32); a.counters->merges = (uint32_t)((sc + f) & 0xffffffffu); }
    if ((f >> 32) == 0ull) return;
    const uint32_t pos = (uint32_t)(sc >> 32);
    if ((f & 1ull) == 0ull) { cl_out[pos] = cl_in[i]; return; }
    const uint32_t l = cl_in[i], r = cl_in[a.nn[i]], id = node_base + (uint32_t)(sc & 0xffffffffu);
    const float4 llo = a.node_lo[l], lhi = a.node_hi[l], rlo = a.node_lo[r], rhi = a.node_hi[r];
    a.node_lo[id] = make_float4(fminf(llo.x, rlo.x), fminf(llo.y, rlo.y), fminf(llo.z, rlo.z), __uint_as_float(l));
    a.node_hi[id] = make_float4(fmaxf(lhi.x, rhi.x), fmaxf(lhi.y, rhi.y), fmaxf(lhi.z, rhi.z), __uint_as_float(r));
    a.node_nprims[id] = a.node_nprims[l] + a.node_nprims[r];
    if (a.node_cell) a.node_cell[id] = a.node_cell[l];
    cl_out[pos] = id;
    CostRow row; cost_of_node(a, id, false, row); store_row(a, id, row);       // (the children's tables: earlier rounds')
}

// Few clusters (the whole of a small build, the last dozen rounds of a large one): ONE workgroup runs every remaining round, a barrier
// where the large build has a launch -- a round of six launches costs 30 us of host time whatever its size.  Same arithmetic and the
// same tie rules as k_ploc_nn / _flags / _apply, so a build is the same tree whichever kernels ran its rounds.
constexpr uint32_t kSmallClusters = 16384u;                     // (what k_ploc_small can take; the host hands it 4096 at most)
__global__ __launch_bounds__(1024) void k_ploc_small(GpuBuildArgs a, uint32_t *cl_a, uint32_t *cl_b, uint32_t n_cells) {
    typedef hipcub::BlockScan<uint64_t, 1024> Scan;
    __shared__ typename Scan::TempStorage tmp;
    BuildCounters *c = a.counters;
    uint32_t m = c->m_cur, node_base = c->node_base;
    uint32_t *cin = cl_a, *cout = cl_b;
    const int radius = a.ploc_radius;
    uint32_t rounds = 0u;
    while (m > n_cells) {
        // nearest neighbours: a cluster per thread, tile after tile
        for (uint32_t i = threadIdx.x; i < m; i += 1024u) {
            const uint32_t me = cin[i];
            const float4 mlo = a.node_lo[me], mhi = a.node_hi[me];
            const uint32_t my_cell = a.node_cell ? a.node_cell[me] : 0u;
            float best = INFINITY; uint32_t bj = kNone;
            for (int d = -radius; d <= radius; ++d) {
                const int j = (int)i + d;
                if (d == 0 || j < 0 || j >= (int)m) continue;
                const uint32_t o = cin[j];
                if (a.node_cell && a.node_cell[o] != my_cell) continue;
                const float ar = merged_half_area(mlo, mhi, a.node_lo[o], a.node_hi[o]);
                if (ar < best || bj == kNone || (ar == best && (uint32_t)j == (i ^ 1u))) { best = ar; bj = (uint32_t)j; }      // (k_ploc_nn's rule)
            }
            if (a.balanced) bj = (i ^ 1u) < m ? (i ^ 1u) : kNone;
            a.nn[i] = bj;
        }
        __syncthreads();
        // flags, scan (tiles of 1024 clusters, the running totals carried from tile to tile), apply
        uint64_t run = 0ull;
        for (uint32_t base = 0; base < m; base += 1024u) {
            const uint32_t i = base + threadIdx.x;
            uint64_t f = 0ull, e, total;
            uint32_t j = kNone;
            if (i < m) {
                j = a.nn[i];
                const bool mutual = j != kNone && a.nn[j] == i;
                f = 1ull << 32;
                if (mutual) f = i < j ? ((1ull << 32) | 1ull) : 0ull;
            }
            Scan(tmp).ExclusiveSum(f, e, total);
            __syncthreads();
            e += run; run += total;
            if (i >= m || (f >> 32) == 0ull) continue;
            const uint32_t pos = (uint32_t)(e >> 32);
            if ((f & 1ull) == 0ull) { cout[pos] = cin[i]; continue; }
            const uint32_t l = cin[i], r = cin[j], id = node_base + (uint32_t)(e & 0xffffffffu);
            const float4 llo = a.node_lo[l], lhi = a.node_hi[l], rlo = a.node_lo[r], rhi = a.node_hi[r];
            a.node_lo[id] = make_float4(fminf(llo.x, rlo.x), fminf(llo.y, rlo.y), fminf(llo.z, rlo.z), __uint_as_float(l));
            a.node_hi[id] = make_float4(fmaxf(lhi.x, rhi.x), fmaxf(lhi.y, rhi.y), fmaxf(lhi.z, rhi.z), __uint_as_float(r));
            a.node_nprims[id] = a.node_nprims[l] + a.node_nprims[r];
            if (a.node_cell) a.node_cell[id] = a.node_cell[l];
            cout[pos] = id;
            CostRow row; cost_of_node(a, id, false, row); store_row(a, id, row);
        }
        __threadfence_block();
        __syncthreads();
        const uint32_t merges = (uint32_t)(run & 0xffffffffull);
        ++rounds;
        uint32_t *t2 = cin; cin = cout; cout = t2;
        m = (uint32_t)(run >> 32); node_base += merges;
        if (merges == 0u) break;                                 // (cannot happen while a cell holds two clusters; never loop forever)
    }
    if (threadIdx.x == 0u) { c->m_cur = m; c->m_next = m; c->node_base = node_base; c->merges = 0u; c->root = m ? cin[0] : kNone; c->small_rounds = rounds; c->cl_in_b = cin == cl_b ? 1u : 0u; }
}

// between two rounds: what k_ploc_apply counted becomes the next round's input
__global__ void k_ploc_advance(GpuBuildArgs a) {
    BuildCounters *c = a.counters;
    c->node_base += c->merges; c->m_cur = c->m_next;
}

// a range of BVH2 nodes whose children's tables are complete: a level of the top-down phase (from the deepest up).  The leaves' tables
// are computed by k_leaves, those of PLOC's nodes at the merge (the children are from earlier rounds).  (Round 2 let the second of a
// node's children to arrive compute it, behind agent-scope fences: 5 ms for a million leaves on a part whose L2s are per XCD.)
__global__ __launch_bounds__(256) void k_cost_range(GpuBuildArgs a, uint32_t first, uint32_t count, uint32_t leaf) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    CostRow row;
    cost_of_node(a, first + i, leaf != 0u, row);
    store_row(a, first + i, row);
}

// ---- emission of one BVH8 level ----
struct Forest { uint32_t node[8]; int n; };
__device__ void collect_forest(const GpuBuildArgs &a, uint32_t root, Forest &out) {
    // Collector::distribute(root, 8) of bvh8_build.cpp, iteratively: (node, budget) pairs on a small stack; right pushed first
    uint32_t st_node[16]; int st_budget[16]; bool st_open[16]; int sp = 0;
    out.n = 0;
    st_node[sp] = root; st_budget[sp] = (int)a.width; st_open[sp] = true; ++sp;
    while (sp > 0) {
        --sp;
        const uint32_t x = st_node[sp]; int i = st_budget[sp]; const bool open = st_open[sp];
        const uint32_t bits = __float_as_uint(a.cost[8 * (size_t)x]);
        const bool leaf2 = __float_as_uint(a.node_lo[x].w) == kNone;
        if (!open) {                                            // forest(x, i)
            while (i > 1 && !((bits >> i) & 1u)) --i;
            if (i == 1 || leaf2) { out.node[out.n++] = x; continue; }
        }
        const int k = (int)row_split(bits, i);                   // distribute(x, i)
        const uint32_t l = __float_as_uint(a.node_lo[x].w), r = __float_as_uint(a.node_hi[x].w);
        st_node[sp] = r; st_budget[sp] = (i - k) < 7 ? (i - k) : 7; st_open[sp] = false; ++sp;
        st_node[sp] = l; st_budget[sp] = k < 7 ? k : 7; st_open[sp] = false; ++sp;
    }
}

// the <= 3 references of a leaf child: its little BVH2 subtree, in order
__device__ __forceinline__ int leaf_refs(const GpuBuildArgs &a, uint32_t c, uint32_t *gk, uint32_t *leaf) {
    uint32_t st[4]; int sp = 0, n = 0; st[sp++] = c;
    while (sp > 0) {
        const uint32_t x = st[--sp];
        const uint32_t l = __float_as_uint(a.node_lo[x].w), r = __float_as_uint(a.node_hi[x].w);
        if (l == kNone) { if (n < 3) { gk[n] = r; leaf[n] = x; ++n; } }
        else { st[sp++] = r; st[sp++] = l; }
    }
    return n;
}
// pieces of ONE primitive that a spatial split made and PLOC put into the same leaf again are one record (box: their union)
__device__ __forceinline__ bool first_of_its_prim(const uint32_t *gk, int w) {
    for (int v = 0; v < w; ++v) if (gk[v] == gk[w]) return false;
    return true;
}

// One BVH8 node.  Eight lanes of a wave call this together, one per child slot (lane & 7 = slot): the forest and the slot assignment
// are computed by all of them alike, then every lane walks, counts and writes ITS child -- a thread that did the eight children one
// after the other spent ~100 us in some two hundred dependent loads, and a level cannot start before the one above is done.
// (live: the group has an item; dead groups go through the shuffles without touching memory)
// Two-level trees (a.instance_leaves: every primitive is an instance): a leaf child is not a record but a TRANSFORM NODE (bvh8.h) in
// its parent's child block, so that the traversal meets instances where it meets inner children -- in the node step, nearest first.
// Such a child goes to the next level's items with kXformItem set and is written there by lane 0 of its group: the marker and the
// instance; the refit that completes the build (refit.hip) fills in the matrix, the BLAS's root and the box.
constexpr uint32_t kXformItem = 0x80000000u;
__device__ void emit_item(const GpuBuildArgs &a, uint32_t b2, uint32_t self, uint2 *__restrict__ items_next, uint32_t next_level_begin, uint32_t is_root_level, bool live_item) {
    const int lane = (int)(threadIdx.x & 63u), g0 = lane & ~7, s = lane & 7;
    const bool xform_item = live_item && (b2 & kXformItem) != 0u;
    const bool live = live_item && !xform_item;          // (a transform node has no children: its group goes through the shuffles like a dead one)
    Forest f; f.n = 0;
    bool ch_leaf[8];
    int slot_child[8];
    for (int q = 0; q < 8; ++q) { slot_child[q] = -1; ch_leaf[q] = false; f.node[q] = 0u; }
    if (live) {
        const float4 nlo = a.node_lo[b2], nhi = a.node_hi[b2];
        const bool node_is_leaf2 = __float_as_uint(nlo.w) == kNone;
        if (node_is_leaf2 || (is_root_level && a.node_nprims[b2] <= a.max_leaf_prims)) {
            f.n = 1; f.node[0] = b2; ch_leaf[0] = true;              // the whole scene fits one leaf: wrap it
        } else {
            collect_forest(a, b2, f);
            // The children's boxes are stored on THIS node's 8-bit grid (step ~ extent / 255 per axis).  Where one child dwarfs the others --
            // an outlier far from the rest of the scene, a huge ground sphere beside small particles -- the small children's stored boxes are
            // blown up by the grid's step and every ray that enters the node visits most of them.  Such a node keeps its two BVH2 children
            // instead: the small side becomes ONE child, whose own node has a grid of its own size (one more visit, several fewer).
            // Ordinary nodes store their children 3 % larger than they are; the guard acts above 25 %.
            if (f.n > 2 && a.quant_guard > 0.0f) {
                float step[3];
                { const float ext[3] = {nhi.x - nlo.x, nhi.y - nlo.y, nhi.z - nlo.z}; for (int q = 0; q < 3; ++q) step[q] = exponent_scale(node_exponent(ext[q])); }
                float sum_true = 0.0f, sum_stored = 0.0f;
                for (int k = 0; k < f.n; ++k) {
                    const float4 lo = a.node_lo[f.node[k]], hi = a.node_hi[f.node[k]];
                    const float e[3] = {hi.x - lo.x, hi.y - lo.y, hi.z - lo.z};
                    const float g[3] = {e[0] + step[0], e[1] + step[1], e[2] + step[2]};      // (a stored box is wider by a step per axis on average)
                    sum_true += e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
                    sum_stored += g[0] * g[1] + g[1] * g[2] + g[2] * g[0];
                }
                if (sum_stored > a.quant_guard * sum_true) {
                    f.n = 2; f.node[0] = __float_as_uint(nlo.w); f.node[1] = __float_as_uint(nhi.w);
                }
            }
            for (int k = 0; k < f.n; ++k) {
                const uint32_t c = f.node[k];
                ch_leaf[k] = __float_as_uint(a.node_lo[c].w) == kNone || (__float_as_uint(a.cost[8 * (size_t)c]) & 1u);
            }
        }
        // slot assignment: slot s is visited first by rays of octant s (bit2 = -x, bit1 = -y, bit0 = -z): greedy, as the host builder
        float cost[8][8];
        const float ncx[3] = {0.5f * (nlo.x + nhi.x), 0.5f * (nlo.y + nhi.y), 0.5f * (nlo.z + nhi.z)};
        for (int k = 0; k < f.n; ++k) {
            const float4 lo = a.node_lo[f.node[k]], hi = a.node_hi[f.node[k]];
            const float d[3] = {0.5f * (lo.x + hi.x) - ncx[0], 0.5f * (lo.y + hi.y) - ncx[1], 0.5f * (lo.z + hi.z) - ncx[2]};
            for (int q = 0; q < 8; ++q) {
                const float sx = (q & 4) ? -1.0f : 1.0f, sy = (q & 2) ? -1.0f : 1.0f, sz = (q & 1) ? -1.0f : 1.0f;
                cost[k][q] = d[0] * sx + d[1] * sy + d[2] * sz;
            }
        }
        bool child_done[8];
        for (int q = 0; q < 8; ++q) child_done[q] = false;
        for (int round = 0; round < f.n; ++round) {
            int bk = -1, bs = -1; float bc = INFINITY;
            for (int k = 0; k < f.n; ++k) {
                if (child_done[k]) continue;
                for (int q = 0; q < 8; ++q) {
                    if (slot_child[q] >= 0) continue;
                    if (cost[k][q] < bc) { bc = cost[k][q]; bk = k; bs = q; }
                }
            }
            if (bk < 0) {   // NaN costs (degenerate boxes): first free pair
                for (int k = 0; k < f.n && bk < 0; ++k) if (!child_done[k]) bk = k;
                for (int q = 0; q < 8 && bs < 0; ++q) if (slot_child[q] < 0) bs = q;
            }
            slot_child[bs] = bk; child_done[bk] = true;
        }
    }
    // this lane's child
    int k = -1;
    for (int q = 0; q < 8; ++q) if (q == s) k = slot_child[q];
    uint32_t c = 0u; bool leaf = false;
    for (int q = 0; q < 8; ++q) if (q == k) { c = f.node[q]; leaf = ch_leaf[q]; }
    uint32_t gks[3] = {0u, 0u, 0u}, lf[3] = {0u, 0u, 0u}; int nr = 0; uint32_t np = 0u;
    if (k >= 0 && leaf) {
        nr = leaf_refs(a, c, gks, lf);
        for (int w = 0; w < nr; ++w) if (first_of_its_prim(gks, w)) ++np;
        if (a.instance_leaves) { c = lf[0] | kXformItem; leaf = false; np = 0u; }      // (max_leaf_prims = 1: the child is one BVH2 leaf = one instance)
    }
    const uint32_t inner = (k >= 0 && !leaf) ? 1u : 0u;
    // the group's running counts: where this lane's records and its inner child go
    uint32_t inc_np = np, inc_in = inner;
    for (int off = 1; off < 8; off <<= 1) {
        const uint32_t t1 = __shfl_up(inc_np, off, 8), t2 = __shfl_up(inc_in, off, 8);
        if (s >= off) { inc_np += t1; inc_in += t2; }
    }
    const uint32_t n_leaf_prims = __shfl(inc_np, g0 + 7), n_inner = __shfl(inc_in, g0 + 7);
    const uint32_t prim_off = inc_np - np, rank = inc_in - inner;
    uint32_t child_base = 0u, prim_base = 0u;
    if (live && s == 0) {
        child_base = n_inner ? atomicAdd(&a.counters->next_node, n_inner) : 0u;
        prim_base = n_leaf_prims ? atomicAdd(&a.counters->next_prim, n_leaf_prims) : 0u;
    }
    child_base = __shfl(child_base, g0); prim_base = __shfl(prim_base, g0);
    const uint32_t imask = (uint32_t)((__ballot(inner != 0u) >> g0) & 0xffull);
    if (xform_item && s == 0) {
        const uint32_t leaf2 = b2 & ~kXformItem;
        const uint32_t gk = __float_as_uint(a.node_hi[leaf2].w);                    // global primitive number = which instance
        uint32_t *w32 = reinterpret_cast<uint32_t *>(a.out_nodes + (size_t)self * a.node_stride);
        for (int q = 0; q < 20; ++q) w32[q] = 0u;                                   // word 3 == 0: the marker
        w32[5] = find_instance(a.inst_first, a.n_inst, gk);
        a.out_node_ref[2 * (size_t)self] = 1.0f; a.out_node_ref[2 * (size_t)self + 1] = 0.0f;
        atomicAdd(&a.counters->total_below, 1.0f);
        atomicAdd(&a.counters->next_prim, 1u);                                      // (counted like a record: the host checks that every instance came out)
    }
    if (!live) return;

    unsigned char *nd = a.out_nodes + (size_t)self * a.node_stride;
    uint32_t meta = 0u;
    if (k >= 0 && leaf) {
        uint32_t w_out = 0u;
        uint32_t last_inst = 0u;
        for (int w = 0; w < nr; ++w) {
            if (!first_of_its_prim(gks, w)) continue;
            const uint32_t gk = gks[w];                                     // global primitive number
            // (the primitives of a leaf are mostly of one instance: look there before the binary search)
            const uint32_t inst = (gk >= a.inst_first[last_inst] && gk < a.inst_first[last_inst + 1u]) ? last_inst : find_instance(a.inst_first, a.n_inst, gk);
            last_inst = inst;
            const uint32_t p = gk - a.inst_first[inst];
            uint32_t *rec = reinterpret_cast<uint32_t *>(a.out_prims + (size_t)(prim_base + prim_off + w_out) * a.prim_stride);
            const uint32_t kind = a.inst_kind[inst];
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, b0 = 0.0f;
            if (kind == kPrimKindSphere) {
                const float *src = reinterpret_cast<const float *>(a.inst_src[inst]) + 4 * (size_t)p;
                a0 = src[0]; a1 = src[1]; a2 = src[2]; b0 = src[3];
            }
            rec[0] = __float_as_uint(a0); rec[1] = __float_as_uint(a1); rec[2] = __float_as_uint(a2); rec[3] = p;
            rec[4] = __float_as_uint(b0); rec[5] = 0u; rec[6] = 0u; rec[7] = inst;
            rec[8] = 0u; rec[9] = 0u; rec[10] = 0u; rec[11] = kind;
            if (a.out_clip) {       // the box the refit takes for this record: the reference's (the union of its pieces in this leaf)
                float4 lo = a.node_lo[lf[w]], hi = a.node_hi[lf[w]];
                for (int u = w + 1; u < nr; ++u)
                    if (gks[u] == gk) {
                        const float4 l2 = a.node_lo[lf[u]], h2 = a.node_hi[lf[u]];
                        lo.x = fminf(lo.x, l2.x); lo.y = fminf(lo.y, l2.y); lo.z = fminf(lo.z, l2.z);
                        hi.x = fmaxf(hi.x, h2.x); hi.y = fmaxf(hi.y, h2.y); hi.z = fmaxf(hi.z, h2.z);
                    }
                float *cb = a.out_clip + 6 * (size_t)(prim_base + prim_off + w_out);
                cb[0] = lo.x; cb[1] = lo.y; cb[2] = lo.z; cb[3] = hi.x; cb[4] = hi.y; cb[5] = hi.z;
            }
            ++w_out;
        }
        meta = (((1u << np) - 1u) << 5) | prim_off;
    } else if (k >= 0) {
        meta = 0x20u | (24u + (uint32_t)s);
        const uint32_t out = child_base + rank;
        items_next[out - next_level_begin] = make_uint2(c, out);
    }
    nd[24 + s] = (unsigned char)meta;
    if (s != 0) return;
    uint32_t *w32 = reinterpret_cast<uint32_t *>(nd);
    // origin / exponents / quantised boxes are the refit's to fill in; a valid empty encoding meanwhile
    w32[0] = 0u; w32[1] = 0u; w32[2] = 0u; w32[3] = (imask << 24) | 0x007f7f7fu;
    w32[4] = child_base; w32[5] = prim_base;
    for (int q = 8; q < 14; ++q) w32[q] = 0xffffffffu;           // qlo = 255
    for (int q = 14; q < 20; ++q) w32[q] = 0u;                   // qhi = 0
    // refit quality weight: primitives below this node (normalised afterwards)
    const float below = (float)a.node_nprims[b2];
    a.out_node_ref[2 * (size_t)self] = below; a.out_node_ref[2 * (size_t)self + 1] = 0.0f;
    atomicAdd(&a.counters->total_below, below);
}

__global__ __launch_bounds__(128) void k_emit_level(GpuBuildArgs a, const uint2 *__restrict__ items, uint32_t n_items, uint2 *__restrict__ items_next,
                                                    uint32_t next_level_begin, uint32_t is_root_level) {
    const uint32_t t = blockIdx.x * 16u + (threadIdx.x >> 3);       // eight lanes per item
    const bool live = t < n_items;
    const uint2 it = items[live ? t : 0u];
    emit_item(a, it.x, it.y, items_next, next_level_begin, is_root_level, live);
}

// a small tree: every level by one workgroup, no trip to the host in between (level_begin goes back with the counters)
__global__ __launch_bounds__(1024) void k_emit_small(GpuBuildArgs a, uint2 *items_a, uint2 *items_b) {
    __shared__ uint32_t s_count;
    BuildCounters *c = a.counters;
    if (c->m_cur == 0u) { if (threadIdx.x == 0u) c->n_levels = 0u; return; }      // nothing valid: the caller emits the empty root
    if (threadIdx.x == 0u) { items_a[0] = make_uint2(c->root, 0u); c->level_begin[0] = 0u; }
    __syncthreads();
    uint2 *in = items_a, *out = items_b;
    uint32_t level_begin = 0u, count = 1u, depth = 0u;
    while (count > 0u) {
        const uint32_t next_begin = level_begin + count;
        for (uint32_t base = 0; base < count; base += 128u) {          // eight lanes per item, 128 items at a time
            const uint32_t t = base + (threadIdx.x >> 3);
            const bool live = t < count;
            const uint2 it = in[live ? t : 0u];
            emit_item(a, it.x, it.y, out, next_begin, depth == 0u ? 1u : 0u, live);
        }
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0u) {
            s_count = atomicAdd(&c->next_node, 0u) - next_begin;
            if (depth + 1u < kMaxSmallLevels) c->level_begin[depth + 1u] = next_begin;
        }
        __syncthreads();
        level_begin = next_begin; count = s_count;
        uint2 *tmp = in; in = out; out = tmp;
        if (count) ++depth;
        if (depth + 1u >= kMaxSmallLevels) break;                  // (deeper than the traversal stack allows anyway: the host reports it)
    }
    if (threadIdx.x == 0u) { c->n_levels = depth + 1u; c->max_depth = depth; }
}

__global__ __launch_bounds__(256) void k_normalise_weights(float *node_ref, uint32_t n_nodes_host, const BuildCounters *c) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n_nodes = n_nodes_host != kNone ? n_nodes_host : (c->n_levels ? c->next_node : 0u);
    if (i >= n_nodes) return;
    const float total = c->total_below;
    node_ref[2 * (size_t)i] = total > 0.0f ? node_ref[2 * (size_t)i] / total : 0.0f;
}

// object-space bounds of one BLAS (hrt_blas_build_*): 6 ordered-integer atomics per wave
__global__ __launch_bounds__(1024) void k_blas_bounds(const float *src, uint32_t n_prims, uint32_t kind, BuildCounters *c) {
    const uint32_t p = blockIdx.x * 1024u + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (p < n_prims) {
        if (kind == kPrimKindTriangle) {
            for (int v = 0; v < 3; ++v) {
                const float *q = src + 9 * (size_t)p + 3 * v;
                if (fabsf(q[0]) <= 3.0e38f && fabsf(q[1]) <= 3.0e38f && fabsf(q[2]) <= 3.0e38f)
                    for (int d = 0; d < 3; ++d) { lo[d] = fminf(lo[d], q[d]); hi[d] = fmaxf(hi[d], q[d]); }
            }
        } else {
            const float *q = src + 4 * (size_t)p; const float rr = fabsf(q[3]);
            if (fabsf(q[0]) <= 3.0e38f && fabsf(q[1]) <= 3.0e38f && fabsf(q[2]) <= 3.0e38f && rr <= 3.0e38f)
                for (int d = 0; d < 3; ++d) { lo[d] = q[d] - rr; hi[d] = q[d] + rr; }
        }
    }
    block_minmax<3>(lo, hi);
    if (threadIdx.x == 0u && lo[0] <= hi[0])
        for (int d = 0; d < 3; ++d) { atomicMin(&c->bmin[d], f2ord(lo[d])); atomicMax(&c->bmax[d], f2ord(hi[d])); }
}

// radius of the bounding sphere of one BLAS around a given centre (two-level trees: bvh8_geom.h clamp_to_sphere_bounds): the largest
// distance of a vertex (a sphere: its far side) from it; non-negative floats order like their bits
__global__ __launch_bounds__(1024) void k_blas_radius(const float *src, uint32_t n_prims, uint32_t kind, float cx, float cy, float cz, BuildCounters *c) {
    const uint32_t p = blockIdx.x * 1024u + threadIdx.x;
    float lo[1] = {0.0f}, far[1] = {0.0f};
    if (p < n_prims) {
        if (kind == kPrimKindTriangle) {
            for (int v = 0; v < 3; ++v) {
                const float *q = src + 9 * (size_t)p + 3 * v;
                const float dx = q[0] - cx, dy = q[1] - cy, dz = q[2] - cz;
                const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
                if (d <= 3.0e38f) far[0] = fmaxf(far[0], d);
            }
        } else {
            const float *q = src + 4 * (size_t)p;
            const float dx = q[0] - cx, dy = q[1] - cy, dz = q[2] - cz;
            const float d = sqrtf((dx * dx + dy * dy) + dz * dz) + fabsf(q[3]);
            if (d <= 3.0e38f) far[0] = d;
        }
    }
    block_minmax<1>(lo, far);
    if (threadIdx.x == 0u) atomicMax(&c->cmax[0], __float_as_uint(far[0]));
}

// spheres are kept as {cx, cy, cz, r}: interleave the caller's two arrays
__global__ __launch_bounds__(256) void k_pack_spheres(const float *centers, const float *radii, uint32_t n, float *out) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    out[4 * (size_t)p + 0] = centers[3 * (size_t)p + 0]; out[4 * (size_t)p + 1] = centers[3 * (size_t)p + 1];
    out[4 * (size_t)p + 2] = centers[3 * (size_t)p + 2]; out[4 * (size_t)p + 3] = radii[p];
}

#define B_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { res.error = _e; res.where = #expr; goto done; } } while (0)

}  // namespace

void launch_pack_spheres(const float *centers, const float *radii, uint32_t n, float *out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_pack_spheres, dim3(blocks(n, 256)), dim3(256), 0, s, centers, radii, n, out);
}

// object-space bounds of a BLAS's geometry into lo[3] / hi[3] (synchronises the stream for 24 bytes)
hipError_t gpu_blas_bounds(const float *d_src, uint32_t n_prims, uint32_t kind, float *lo, float *hi, void *d_scratch, hipStream_t s) {
    static_assert(sizeof(BuildCounters) <= kBoundsScratchBytes, "the counters are the only working memory");
    for (int d = 0; d < 3; ++d) { lo[d] = INFINITY; hi[d] = -INFINITY; }
    if (n_prims == 0) return hipSuccess;
    BuildCounters *c = reinterpret_cast<BuildCounters *>(d_scratch);
    hipError_t e = c ? hipSuccess : hipMalloc((void **)&c, sizeof(BuildCounters));
    if (e != hipSuccess) return e;
    BuildCounters h{};
    for (int d = 0; d < 3; ++d) { h.bmin[d] = h.cmin[d] = 0xffffffffu; h.bmax[d] = h.cmax[d] = 0u; }
    e = hipMemcpyAsync(c, &h, sizeof h, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_blas_bounds, dim3(blocks(n_prims, 1024)), dim3(1024), 0, s, d_src, n_prims, kind, c); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(&h, c, sizeof h, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (!d_scratch) (void)hipFree(c);
    if (e != hipSuccess) return e;
    if (h.bmin[0] <= h.bmax[0] && h.bmax[0] != 0u)
        for (int d = 0; d < 3; ++d) { lo[d] = ord2f(h.bmin[d]); hi[d] = ord2f(h.bmax[d]); }
    return hipSuccess;
}

hipError_t gpu_blas_radius(const float *d_src, uint32_t n_prims, uint32_t kind, const float *center, float *radius, void *d_scratch, hipStream_t s) {
    *radius = -1.0f;
    if (n_prims == 0) return hipSuccess;
    BuildCounters *c = reinterpret_cast<BuildCounters *>(d_scratch);
    hipError_t e = c ? hipSuccess : hipMalloc((void **)&c, sizeof(BuildCounters));
    if (e != hipSuccess) return e;
    BuildCounters h{};
    e = hipMemcpyAsync(c, &h, sizeof h, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_blas_radius, dim3(blocks(n_prims, 1024)), dim3(1024), 0, s, d_src, n_prims, kind, center[0], center[1], center[2], c); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(&h, c, sizeof h, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (!d_scratch) (void)hipFree(c);
    if (e != hipSuccess) return e;
    float r; memcpy(&r, &h.cmax[0], 4);
    // (the distances are rounded: a few ULPs of slack, and of the centre's coordinates)
    const float cmax = std::max(std::fabs(center[0]), std::max(std::fabs(center[1]), std::fabs(center[2])));
    if (r <= 3.0e38f) *radius = r * 1.000002f + 4e-7f * cmax;
    return hipSuccess;
}

// every array of gpu_build_bvh8 below (296 bytes per leaf) + the radix sort's and the scan's temporaries + alignment; with spatial
// splits the leaves are references (at most gpu_build_max_refs of them), and the top-down phase's own buffers (build_split.hip: 32 bytes
// per reference that stay, ~120 that are given back before the arrays below are taken) come on top
size_t gpu_build_scratch_bytes(uint32_t n_prims, const SplitParams *split) {
    if (!split || !split->enabled) return (size_t)n_prims * 340u + (1u << 20);
    const size_t cap = gpu_build_max_refs(n_prims, split), tables = gpu_split_table_bytes(n_prims, *split);
    const size_t phase = cap * 64u + std::max<size_t>(cap, n_prims) * 16u + (cap - n_prims + 4096u) * 56u, after = cap * 348u;
    return (size_t)n_prims * 32u + cap * 32u + tables + std::max(phase, after) + (4u << 20);
}

// The build.  in: instance tables + output buffers sized for the worst case (one node and one record per leaf).  Synchronises `s`
// a few dozen times for a counter each (split levels, PLOC rounds, levels); no geometry crosses the bus.
GpuBuildResult gpu_build_bvh8(const GpuBuildInput &in, hipStream_t s) {
    GpuBuildResult res{};
    const uint32_t n = in.n_prims;
    GpuBuildArgs a{};
    void *temp = nullptr; size_t temp_bytes = 0, sort_bytes = 0, scan_bytes = 0;
    uint64_t *keys_out = nullptr; uint32_t *cl_b = nullptr; uint2 *items_a = nullptr, *items_b = nullptr;
    BuildCounters h{};
    uint32_t nv = 0, nl = 0, ns = 0, m = 0, node_base = 0, root = 0, n_cells = 1;
    bool split = false, small_tail = false;
    // one workgroup per round beats six launches per round below a few thousand clusters (measured: 1024 .. 16384 all within 5 %; the
    // reference's 8 k-triangle sample is a little quicker through the launches and the tail: profiles/r03_device_split_build.txt)
    const uint32_t small_limit = 4096u;
    SplitPhaseResult sp{};
    // working memory comes out of the caller's arena while it lasts (a small build otherwise spends more time in ~25 hipMalloc /
    // hipFree pairs, each of which synchronises the device, than in its kernels)
    BuildArena arena; arena.base = in.scratch; arena.bytes = in.scratch_bytes;
    auto alloc = [&](void **p, size_t bytes) -> hipError_t { return arena.alloc(p, bytes); };

    a.n = n; a.n_inst = in.n_inst; a.inst_first = in.d_inst_first; a.inst_kind = in.d_inst_kind; a.inst_src = in.d_inst_src;
    a.inst_xf = in.d_inst_xf; a.inst_identity = in.d_inst_identity;
    a.max_leaf_prims = in.max_leaf_prims; a.c_node = in.c_node; a.c_prim = in.c_prim; a.quant_guard = in.quant_guard;
    a.instance_leaves = in.instance_leaves ? 1u : 0u;
    a.balanced = in.balanced ? 1u : 0u;
    a.width = in.width >= 2u && in.width <= 8u ? in.width : 8u;
    a.ploc_radius = in.ploc_radius < 1 ? 1 : (in.ploc_radius > kPlocMaxRadius ? kPlocMaxRadius : in.ploc_radius);
    a.out_nodes = in.out_nodes; a.node_stride = in.node_stride; a.out_prims = in.out_prims; a.prim_stride = in.prim_stride; a.out_node_ref = in.out_node_ref;

    B_TRY(alloc((void **)&a.counters, sizeof(BuildCounters)));
    B_TRY(alloc((void **)&a.pb_lo, sizeof(float4) * (size_t)n)); B_TRY(alloc((void **)&a.pb_hi, sizeof(float4) * (size_t)n));
    for (int d = 0; d < 3; ++d) { h.bmin[d] = h.cmin[d] = 0xffffffffu; h.bmax[d] = h.cmax[d] = 0u; }
    h.next_node = 1u;                                            // node 0 is the root
    B_TRY(hipMemcpyAsync(a.counters, &h, sizeof h, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_prim_bounds, dim3(blocks(n, 1024)), dim3(1024), 0, s, a);
    B_TRY(hipGetLastError());
    // ---- spatial splits: the top-down phase turns the primitives into references grouped in cells ----
    // (a few thousand primitives: the splits buy nothing a ray could notice, and the small build below is the quicker one)
    split = in.split.enabled && (in.max_leaf_prims == kMaxLeafPrims || in.instance_leaves) && n >= std::max(in.split.cell_refs, 2u) && n > kSmallClusters / 4u;
    if (split) {
        B_TRY(hipMemcpyAsync(&h, a.counters, sizeof h, hipMemcpyDeviceToHost, s));
        B_TRY(hipStreamSynchronize(s));
        nv = n - h.n_invalid;
        if (nv < std::max(in.split.cell_refs, 2u)) split = false;
    }
    if (split) {
        const size_t arena_mark = arena.used;
        sp = gpu_split_phase(a, nv, in.split, arena, s);
        if (sp.error != hipSuccess) {
            // The top-down phase is an improvement, not a necessity: when it gives up (its tables outgrown twice, its ~120 bytes per
            // reference of temporaries not to be had, counts that do not add up) and the stream is healthy, PLOC alone builds the tree --
            // inside an update of a running animation that is a slower frame, not a failed one.
            if (hipStreamSynchronize(s) != hipSuccess) { res.error = sp.error; res.where = sp.where; goto done; }
            (void)hipGetLastError();
            if (in.split.verbose) std::fprintf(stderr, "[hrt] device build: the top-down phase gave up (%s: %s); PLOC alone\n", sp.where, hipGetErrorString(sp.error));
            arena.used = arena_mark;
            split = false; sp = SplitPhaseResult{};
            res.fell_back = true;
        }
    }
    if (split) {
        a.ref_lo = sp.ref_lo; a.ref_hi = sp.ref_hi; a.out_clip = in.out_clip;
        n_cells = sp.n_cells; res.split_levels = sp.levels; res.n_cells = sp.n_cells;
    }
    ns = split ? sp.n_refs : n;                                  // what is sorted: references, or all primitives (the invalid ones last)
    B_TRY(alloc((void **)&a.keys, sizeof(uint64_t) * (size_t)ns)); B_TRY(alloc((void **)&keys_out, sizeof(uint64_t) * (size_t)ns));
    B_TRY(alloc((void **)&a.vals, sizeof(uint32_t) * (size_t)ns)); B_TRY(alloc((void **)&a.vals_sorted, sizeof(uint32_t) * (size_t)ns));
    B_TRY(alloc((void **)&a.node_lo, sizeof(float4) * 2 * (size_t)ns)); B_TRY(alloc((void **)&a.node_hi, sizeof(float4) * 2 * (size_t)ns));
    B_TRY(alloc((void **)&a.node_nprims, sizeof(uint32_t) * 2 * (size_t)ns));
    if (split) B_TRY(alloc((void **)&a.node_cell, sizeof(uint32_t) * 2 * (size_t)ns));
    B_TRY(alloc((void **)&a.cl_a, sizeof(uint32_t) * (size_t)ns)); B_TRY(alloc((void **)&cl_b, sizeof(uint32_t) * (size_t)ns));
    B_TRY(alloc((void **)&a.nn, sizeof(uint32_t) * (size_t)ns));
    B_TRY(alloc((void **)&a.flags, sizeof(uint64_t) * (size_t)ns)); B_TRY(alloc((void **)&a.scan, sizeof(uint64_t) * (size_t)ns));
    B_TRY(alloc((void **)&a.cost, sizeof(float) * 8 * 2 * (size_t)ns));
    B_TRY(alloc((void **)&items_a, sizeof(uint2) * (size_t)ns)); B_TRY(alloc((void **)&items_b, sizeof(uint2) * (size_t)ns));
    B_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, a.keys, keys_out, a.vals, a.vals_sorted, (int)ns, 0, 64, s));
    B_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, a.flags, a.scan, (int)ns, s));
    temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    B_TRY(alloc(&temp, temp_bytes));

    if (split) hipLaunchKernelGGL(k_morton_refs, dim3(blocks(ns, 256)), dim3(256), 0, s, a, ns, sp.segs, sp.cell_seg);
    else hipLaunchKernelGGL(k_morton, dim3(blocks(n, 256)), dim3(256), 0, s, a);
    B_TRY(hipGetLastError());
    B_TRY(hipcub::DeviceRadixSort::SortPairs(temp, sort_bytes, a.keys, keys_out, a.vals, a.vals_sorted, (int)ns, 0, 64, s));
    if (!split && n <= small_limit) {
        // ---- a small build (the reference's own scenes: a few thousand triangles): every PLOC round and every level of the emission in
        //      one workgroup each, and the host looks at the counters once, at the end ----
        hipLaunchKernelGGL(k_leaves, dim3(blocks(n, 256)), dim3(256), 0, s, a, kNone);
        hipLaunchKernelGGL(k_ploc_small, dim3(1), dim3(1024), 0, s, a, a.cl_a, cl_b, 1u);
        hipLaunchKernelGGL(k_emit_small, dim3(1), dim3(1024), 0, s, a, items_a, items_b);
        hipLaunchKernelGGL(k_normalise_weights, dim3(blocks(in.instance_leaves ? 2u * n + 2u : n, 256)), dim3(256), 0, s, in.out_node_ref, kNone, a.counters);      // (a node per thread: at most n of them, with transform nodes 2 n + 1)
        B_TRY(hipGetLastError());
        B_TRY(hipMemcpyAsync(&h, a.counters, sizeof h, hipMemcpyDeviceToHost, s));
        B_TRY(hipStreamSynchronize(s));
        nv = n - h.n_invalid;
        res.n_prims = nv;
        if (nv == 0) goto done;
        for (int d = 0; d < 3; ++d) { res.lo[d] = ord2f(h.bmin[d]); res.hi[d] = ord2f(h.bmax[d]); }
        if (h.m_cur != 1u) { res.error = hipErrorUnknown; res.where = "PLOC made no progress"; goto done; }
        if (h.n_levels == 0u || h.n_levels >= kMaxSmallLevels) { res.error = hipErrorUnknown; res.where = "small build: tree too deep"; goto done; }
        res.ploc_rounds = h.small_rounds; res.n_nodes = h.next_node; res.max_depth = h.max_depth; res.n_records = h.next_prim;
        for (uint32_t l = 0; l < h.n_levels; ++l) res.level_begin.push_back(h.level_begin[l]);
        res.level_begin.push_back(h.next_node);
        if (h.next_prim != nv) { res.error = hipErrorUnknown; res.where = "emitted primitive count differs from the valid count"; goto done; }
        goto done;
    }
    B_TRY(hipMemcpyAsync(&h, a.counters, sizeof h, hipMemcpyDeviceToHost, s));
    B_TRY(hipStreamSynchronize(s));
    nv = n - h.n_invalid;
    res.n_prims = nv;
    if (nv == 0) goto done;                                      // nothing to hit: the caller emits the empty root
    for (int d = 0; d < 3; ++d) { res.lo[d] = ord2f(h.bmin[d]); res.hi[d] = ord2f(h.bmax[d]); }
    nl = split ? ns : nv;                                        // leaves of the BVH2

    // ---- PLOC ----
    if (split) hipLaunchKernelGGL(k_leaves_refs, dim3(blocks(nl, 256)), dim3(256), 0, s, a, nl);
    else hipLaunchKernelGGL(k_leaves, dim3(blocks(nl, 256)), dim3(256), 0, s, a, nl);
    m = nl; node_base = nl;
    {
        uint32_t *cl_in = a.cl_a, *cl_out = cl_b;
        h.m_cur = m; h.node_base = node_base; h.m_next = m; h.merges = 0u;
        B_TRY(hipMemcpyAsync(a.counters, &h, sizeof h, hipMemcpyHostToDevice, s));
        // A round merges at least one pair, typically 40 % of the clusters.  The host launches kRoundsPerBatch rounds over the
        // cluster count it last read (a round past the end of the build copies the clusters left) and only then synchronises:
        // 6 read-backs for a million primitives instead of 44.  With cells the rounds end when every cell is down to one cluster.
        constexpr int kRoundsPerBatch = 8;
        while (m > n_cells) {
            if (!split && m <= small_limit) {                    // the last rounds: one workgroup, one launch
                hipLaunchKernelGGL(k_ploc_small, dim3(1), dim3(1024), 0, s, a, cl_in, cl_out, 1u);
                B_TRY(hipGetLastError());
                B_TRY(hipMemcpyAsync(&h, a.counters, sizeof h, hipMemcpyDeviceToHost, s));
                B_TRY(hipStreamSynchronize(s));
                if (h.m_cur != 1u) { res.error = hipErrorUnknown; res.where = "PLOC made no progress"; goto done; }
                res.ploc_rounds += h.small_rounds; m = 1u; root = h.root; small_tail = true;
                break;
            }
            for (int k = 0; k < kRoundsPerBatch; ++k) {
                hipLaunchKernelGGL(k_ploc_nn, dim3(blocks(m, 256)), dim3(256), 0, s, a, cl_in);
                hipLaunchKernelGGL(k_ploc_flags, dim3(blocks(m, 256)), dim3(256), 0, s, a, m);
                B_TRY(hipcub::DeviceScan::ExclusiveSum(temp, scan_bytes, a.flags, a.scan, (int)m, s));
                hipLaunchKernelGGL(k_ploc_apply, dim3(blocks(m, 256)), dim3(256), 0, s, a, cl_in, cl_out);
                hipLaunchKernelGGL(k_ploc_advance, dim3(1), dim3(1), 0, s, a);
                std::swap(cl_in, cl_out);
                ++res.ploc_rounds;
            }
            B_TRY(hipGetLastError());
            B_TRY(hipMemcpyAsync(&h, a.counters, sizeof h, hipMemcpyDeviceToHost, s));
            B_TRY(hipStreamSynchronize(s));
            if (h.m_cur >= m) { res.error = hipErrorUnknown; res.where = "PLOC made no progress"; goto done; }
            m = h.m_cur;
        }
        if (m != n_cells) { res.error = hipErrorUnknown; res.where = "PLOC left fewer clusters than cells"; goto done; }
        node_base = h.node_base;
        if (split && sp.n_top) {
            // the top of the tree: the split segments, numbered after PLOC's nodes (node_base + t; top node 0 is the root)
            if ((uint64_t)node_base + sp.n_top > 2ull * ns) { res.error = hipErrorUnknown; res.where = "BVH2 node count"; goto done; }
            hipLaunchKernelGGL(k_top_link, dim3(blocks(sp.n_top, 256)), dim3(256), 0, s, a, sp.segs, sp.top_seg, cl_in, sp.n_top, node_base);
            B_TRY(hipGetLastError());
            root = node_base;
        } else if (!small_tail) {
            B_TRY(hipMemcpyAsync(&root, cl_in, sizeof root, hipMemcpyDeviceToHost, s));
            B_TRY(hipStreamSynchronize(s));
        }
    }
    // ---- cost tables: the leaves' and PLOC's nodes' were computed where the nodes were made; the top-down phase's levels from the deepest up ----
    if (split && sp.n_top)
        for (size_t l = sp.top_level_begin.size() - 1; l-- > 0;) {
            const uint32_t b0 = sp.top_level_begin[l], cnt = sp.top_level_begin[l + 1] - b0;
            if (cnt) hipLaunchKernelGGL(k_cost_range, dim3(blocks(cnt, 256)), dim3(256), 0, s, a, node_base + b0, cnt, 0u);
        }
    B_TRY(hipGetLastError());
    // ---- emission, level by level ----
    {
        const uint2 first = make_uint2(root, 0u);
        B_TRY(hipMemcpyAsync(items_a, &first, sizeof first, hipMemcpyHostToDevice, s));
        uint32_t level_begin = 0, level_count = 1, depth = 0;
        uint2 *it_in = items_a, *it_out = items_b;
        res.level_begin.push_back(0u);
        while (level_count > 0u) {
            const uint32_t next_begin = level_begin + level_count;
            hipLaunchKernelGGL(k_emit_level, dim3(blocks(level_count, 16)), dim3(128), 0, s, a, it_in, level_count, it_out, next_begin, depth == 0 ? 1u : 0u);
            B_TRY(hipGetLastError());
            B_TRY(hipMemcpyAsync(&h, a.counters, sizeof h, hipMemcpyDeviceToHost, s));
            B_TRY(hipStreamSynchronize(s));
            res.level_begin.push_back(next_begin);
            level_begin = next_begin; level_count = h.next_node - next_begin;
            std::swap(it_in, it_out);
            if (level_count) ++depth;
        }
        res.n_nodes = h.next_node; res.max_depth = depth; res.n_records = h.next_prim;
        // (with spatial splits, pieces of a primitive that met again in a leaf were emitted once)
        if (split ? (h.next_prim < nv || h.next_prim > nl) : h.next_prim != nv) { res.error = hipErrorUnknown; res.where = "emitted primitive count differs from the valid count"; goto done; }
    }
    hipLaunchKernelGGL(k_normalise_weights, dim3(blocks(res.n_nodes, 256)), dim3(256), 0, s, in.out_node_ref, res.n_nodes, a.counters);
    B_TRY(hipGetLastError());
    B_TRY(hipStreamSynchronize(s));
done:
    return res;
}

}  // namespace hrt
