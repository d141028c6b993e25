// build_split.hip -- the top-down phase of a device build with spatial splits (gfx950): what OPTIX_BUILD_FLAG_PREFER_FAST_TRACE buys
// the reference (src/Global/RendererImpl.cu:30-88 buildASImpl passes the flag to optixAccelBuild), without a host copy of the geometry.
//
// Spatial splits (Stich, Friedrich, Dietrich 2009) pay where a node holds many references: the host builder's experiments
// (profiles/r03_tree_quality_cpu.txt) show no gain from splits in nodes of fewer than ~256 references.  So the device cuts the scene top-down
// only that far -- the same binned-SAH object split / spatial split / reference-unsplitting rules as bvh8_build.cpp, level by level over
// ALL segments of a level at once -- and hands the resulting cells (a few hundred references each, boxes clipped to the cell) to PLOC
// (build.hip), which builds the subtree of every cell bottom-up and never merges across a cell boundary.  The segments that were split
// become the top of the BVH2.
//
// One level = seven launches over the references still in play (retired cells are copied out once):
//   bins      per 2048-reference chunk of a segment: 16 object bins + 32 spatial bins x 3 axes in LDS (ordered-integer min / max,
//             so the result does not depend on the order of arrival), then one flush per chunk;
//   select    8 lanes per segment sweep the bins (a segment of one chunk: the lanes of the workgroup that binned it, the bins never
//             leave the LDS): object split, spatial split or "leave it to PLOC";
//   flags     per reference: left, right or both (a straddler stays whole on one side when that is cheaper: unsplitting);
//   scan      hipCUB exclusive sum of (left << 32 | right): positions come from prefix sums, so the order of the references -- and
//             with it the tree -- is the same in every run;
//   plan      one workgroup: child segments, budgets, the next level's chunk list, cell / top-node numbers, all by block scans;
//   scatter   cut the straddlers (the triangle clipped to the child's box: bvh8_geom.h, the host's arithmetic), write the children,
//             reduce their bounds per workgroup.
// HBM-bound integer / min-max work; nothing here is GEMM-shaped.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include "build_dev.h"

#pragma clang fp contract(off)

namespace hrt {

hipError_t BuildArena::alloc(void **p, size_t n) {
    n = (std::max<size_t>(n, 16) + 255u) & ~(size_t)255u;
    if (base && used + n <= bytes) { *p = static_cast<char *>(base) + used; used += n; return hipSuccess; }
    hipError_t e = hipMalloc(p, n); if (e == hipSuccess) owned.push_back(*p); return e;
}
BuildArena::~BuildArena() { for (void *p : owned) (void)hipFree(p); }

namespace {

constexpr int kObjBins = 16, kSpBins = 32;
constexpr int kObjBox = 0, kSpBox = 3 * kObjBins * 6;                                  // boxes: 6 words per bin (lo as f2ord(x), hi as f2ord(-x))
constexpr int kBinMinMax = kSpBox + 3 * kSpBins * 6;                                   // 864 words that shrink under atomicMin (identity ~0)
constexpr int kObjCnt = kBinMinMax, kSpEnter = kObjCnt + 3 * kObjBins, kSpLeave = kSpEnter + 3 * kSpBins;
constexpr int kBinWords = kSpLeave + 3 * kSpBins;                                      // 1104 words per segment
constexpr uint32_t kChunk = 2048u, kMaxLevels = 64u;

struct SplitCounters { uint32_t n_act, n_chunks, src_total, n_cells, n_top, n_out, n_segs, gave_up, n_big; };

struct SplitArgs {
    GpuBuildArgs b;                                  // instance tables (the scatter clips triangles), primitive bounds, scene counters
    SplitSeg *segs; uint32_t seg_cap;
    uint32_t *act, *act_next;                        // segments of this level / the next
    uint32_t *chunk_act, *chunk_off, *chunk_act_next, *chunk_off_next;       // chunk -> (position in act, offset within the segment)
    float4 *src_lo, *src_hi, *dst_lo, *dst_hi, *out_lo, *out_hi;
    uint32_t *bins; uint64_t *flags, *scan;
    uint32_t *top_seg, *cell_seg;
    SplitCounters *counters;
    uint32_t n_act, n_chunks, cell_refs, level;
    float pad, alpha_area, bias;
};

__device__ __forceinline__ float half_area3(const float *lo, const float *hi) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ float padded_area(const float *b, float pad) {           // b = lo[3], hi[3]
    const float lo[3] = {b[0] - pad, b[1] - pad, b[2] - pad}, hi[3] = {b[3] + pad, b[4] + pad, b[5] + pad};
    return half_area3(lo, hi);
}
__device__ __forceinline__ void box_reset(float *b) { b[0] = b[1] = b[2] = INFINITY; b[3] = b[4] = b[5] = -INFINITY; }
__device__ __forceinline__ void box_grow(float *b, const float *o) {
    for (int c = 0; c < 3; ++c) { b[c] = fminf(b[c], o[c]); b[3 + c] = fmaxf(b[3 + c], o[3 + c]); }
}
__device__ __forceinline__ void seg_bounds(const uint32_t *w, float *lo, float *hi) {
    for (int c = 0; c < 3; ++c) { lo[c] = ord2f(w[c]); hi[c] = -ord2f(w[3 + c]); }
}

// ---- start: references of the valid primitives, compacted ----
__global__ __launch_bounds__(256) void k_split_valid_flags(GpuBuildArgs a, uint64_t *flags) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k < a.n) flags[k] = a.pb_lo[k].w != 0.0f ? 1ull : 0ull;
}
__global__ __launch_bounds__(256) void k_split_init_refs(GpuBuildArgs a, const uint64_t *scan, float4 *lo, float4 *hi) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= a.n) return;
    const float4 l = a.pb_lo[k], h = a.pb_hi[k];
    if (l.w == 0.0f) return;
    const uint32_t pos = (uint32_t)scan[k];
    lo[pos] = make_float4(l.x, l.y, l.z, __uint_as_float(k));
    hi[pos] = make_float4(h.x, h.y, h.z, 0.0f);
}
__global__ __launch_bounds__(256) void k_split_first_level(SplitArgs a, uint32_t n_valid, uint32_t budget) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q < a.n_chunks) { a.chunk_act[q] = 0u; a.chunk_off[q] = q * kChunk; }
    if (q != 0u) return;
    SplitSeg sg{};
    sg.first = 0u; sg.count = n_valid; sg.budget = budget; sg.kind = 0u; sg.level = 0u; sg.bins_slot = 0u;
    const BuildCounters *c = a.b.counters;
    for (int d = 0; d < 3; ++d) {
        sg.nb[d] = c->bmin[d]; sg.nb[3 + d] = f2ord(-ord2f(c->bmax[d]));
        sg.cb[d] = c->cmin[d]; sg.cb[3 + d] = f2ord(-ord2f(c->cmax[d]));
    }
    a.segs[0] = sg; a.act[0] = 0u;
    SplitCounters sc{};
    sc.n_act = 1u; sc.n_chunks = a.n_chunks; sc.src_total = n_valid; sc.n_segs = 1u; sc.n_big = n_valid > kChunk ? 1u : 0u;
    *a.counters = sc;
}

__device__ __forceinline__ void select_split(const SplitArgs &a, SplitSeg &sg, const uint32_t *g, uint32_t role, bool live);

// ---- bins ----
__global__ __launch_bounds__(256) void k_split_init_bins(uint32_t *bins, uint32_t n_words) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n_words) bins[i] = (i % (uint32_t)kBinWords) < (uint32_t)kBinMinMax ? 0xffffffffu : 0u;
}

__global__ __launch_bounds__(256) void k_split_bin(SplitArgs a) {
    __shared__ uint32_t s_bins[kBinWords];
    const uint32_t ai = a.chunk_act[blockIdx.x], off = a.chunk_off[blockIdx.x];
    const SplitSeg &sg = a.segs[a.act[ai]];
    const uint32_t cnt = sg.count, first = sg.first;
    if (cnt < a.cell_refs || sg.level >= kMaxLevels) return;       // a cell: nothing to decide (the same for the whole workgroup)
    for (uint32_t w = threadIdx.x; w < (uint32_t)kBinWords; w += 256u) s_bins[w] = w < (uint32_t)kBinMinMax ? 0xffffffffu : 0u;
    float nlo[3], nhi[3], clo[3], chi[3];
    seg_bounds(sg.nb, nlo, nhi); seg_bounds(sg.cb, clo, chi);
    const bool spatial = sg.budget > 0u;
    __syncthreads();
    const uint32_t end = off + kChunk < cnt ? off + kChunk : cnt;
    for (uint32_t j = off + threadIdx.x; j < end; j += 256u) {
        const float4 l4 = a.src_lo[first + j], h4 = a.src_hi[first + j];
        const float rl[3] = {l4.x, l4.y, l4.z}, rh[3] = {h4.x, h4.y, h4.z};
        for (int ax = 0; ax < 3; ++ax) {
            const float ext = chi[ax] - clo[ax];
            if (ext > 0.0f) {
                const float scale = (float)kObjBins / ext;
                int k = (int)((0.5f * (rl[ax] + rh[ax]) - clo[ax]) * scale);
                k = k < 0 ? 0 : (k > kObjBins - 1 ? kObjBins - 1 : k);
                atomicAdd(&s_bins[kObjCnt + ax * kObjBins + k], 1u);
                uint32_t *bx = &s_bins[kObjBox + (ax * kObjBins + k) * 6];
                for (int c = 0; c < 3; ++c) { atomicMin(&bx[c], f2ord(rl[c])); atomicMin(&bx[3 + c], f2ord(-rh[c])); }
            }
            const float lo_a = nlo[ax], ext2 = nhi[ax] - nlo[ax];
            if (spatial && ext2 > 0.0f) {
                const float scale = (float)kSpBins / ext2, width = ext2 / (float)kSpBins;
                int k0 = (int)((rl[ax] - lo_a) * scale), k1 = (int)((rh[ax] - lo_a) * scale);
                k0 = k0 < 0 ? 0 : (k0 > kSpBins - 1 ? kSpBins - 1 : k0);
                k1 = k1 < 0 ? 0 : (k1 > kSpBins - 1 ? kSpBins - 1 : k1);
                atomicAdd(&s_bins[kSpEnter + ax * kSpBins + k0], 1u);
                atomicAdd(&s_bins[kSpLeave + ax * kSpBins + k1], 1u);
                for (int k = k0; k <= k1; ++k) {         // the reference's box cut by the slab of bin k (the host's "fast binning")
                    float l[3] = {rl[0], rl[1], rl[2]}, h[3] = {rh[0], rh[1], rh[2]};
                    if (k != k0) l[ax] = fmaxf(l[ax], lo_a + width * (float)k);
                    if (k != k1) h[ax] = fminf(h[ax], lo_a + width * (float)(k + 1));
                    uint32_t *bx = &s_bins[kSpBox + (ax * kSpBins + k) * 6];
                    for (int c = 0; c < 3; ++c) { atomicMin(&bx[c], f2ord(l[c])); atomicMin(&bx[3 + c], f2ord(-h[c])); }
                }
            }
        }
    }
    __syncthreads();
    if (cnt <= kChunk) {       // the whole segment was this workgroup's: decide here, the bins never leave the LDS
        if (threadIdx.x < 8u) select_split(a, a.segs[a.act[ai]], s_bins, threadIdx.x, true);
        return;
    }
    uint32_t *g = a.bins + (size_t)sg.bins_slot * kBinWords;
    for (uint32_t w = threadIdx.x; w < (uint32_t)kBinWords; w += 256u) {
        const uint32_t v = s_bins[w];
        if (w < (uint32_t)kBinMinMax) { if (v != 0xffffffffu) atomicMin(&g[w], v); } else if (v) atomicAdd(&g[w], v);
    }
}

// ---- select: eight lanes per segment; lanes 0..2 sweep the object bins of an axis, lanes 3..5 the spatial bins ----
__device__ __forceinline__ bool bin_box(const uint32_t *w, float *b) {
    if (w[0] == 0xffffffffu) return false;
    for (int c = 0; c < 3; ++c) { b[c] = ord2f(w[c]); b[3 + c] = -ord2f(w[3 + c]); }
    return true;
}

// (eight consecutive lanes of a wave call this together: `role` = lane & 7; g = the segment's bins, in LDS or in memory)
__device__ __forceinline__ void select_split(const SplitArgs &a, SplitSeg &sg, const uint32_t *g, uint32_t role, bool live) {
    const uint32_t cnt = sg.count;
    const bool decide = live && cnt >= a.cell_refs && sg.level < kMaxLevels;
    const float pad = a.pad;
    float nlo[3], nhi[3], clo[3], chi[3];
    seg_bounds(sg.nb, nlo, nhi); seg_bounds(sg.cb, clo, chi);

    float cost = INFINITY, bl[6], br[6], pos = 0.0f; int bk = -1; uint32_t n_l = 0u, n_r = 0u;
    box_reset(bl); box_reset(br);
    if (decide && role < 3u) {
        const int ax = (int)role;
        if (chi[ax] - clo[ax] > 0.0f) {
            float rb[kObjBins][6]; uint32_t rc[kObjBins];
            float acc[6]; box_reset(acc); uint32_t n = 0u;
            for (int k = kObjBins - 1; k > 0; --k) {
                n += g[kObjCnt + ax * kObjBins + k];
                float b[6]; if (bin_box(g + kObjBox + (ax * kObjBins + k) * 6, b)) box_grow(acc, b);
                for (int c = 0; c < 6; ++c) rb[k][c] = acc[c];
                rc[k] = n;
            }
            box_reset(acc); n = 0u;
            for (int k = 0; k < kObjBins - 1; ++k) {
                n += g[kObjCnt + ax * kObjBins + k];
                float b[6]; if (bin_box(g + kObjBox + (ax * kObjBins + k) * 6, b)) box_grow(acc, b);
                if (n == 0u || rc[k + 1] == 0u) continue;
                const float v = padded_area(acc, pad) * (float)n + padded_area(rb[k + 1], pad) * (float)rc[k + 1];
                if (v < cost) { cost = v; bk = k; n_l = n; n_r = rc[k + 1]; for (int c = 0; c < 6; ++c) { bl[c] = acc[c]; br[c] = rb[k + 1][c]; } }
            }
        }
    } else if (decide && role < 6u && sg.budget > 0u) {
        const int ax = (int)role - 3;
        const float lo_a = nlo[ax], ext = nhi[ax] - nlo[ax];
        if (ext > 0.0f) {
            const float width = ext / (float)kSpBins;
            float rb[kSpBins][6]; uint32_t rc[kSpBins]; bool rv[kSpBins];
            float acc[6]; box_reset(acc); uint32_t n = 0u; bool any = false;
            for (int k = kSpBins - 1; k > 0; --k) {
                n += g[kSpLeave + ax * kSpBins + k];
                float b[6]; if (bin_box(g + kSpBox + (ax * kSpBins + k) * 6, b)) { box_grow(acc, b); any = true; }
                for (int c = 0; c < 6; ++c) rb[k][c] = acc[c];
                rc[k] = n; rv[k] = any;
            }
            box_reset(acc); n = 0u; any = false;
            for (int k = 0; k < kSpBins - 1; ++k) {
                n += g[kSpEnter + ax * kSpBins + k];
                float b[6]; if (bin_box(g + kSpBox + (ax * kSpBins + k) * 6, b)) { box_grow(acc, b); any = true; }
                if (n == 0u || rc[k + 1] == 0u || !any || !rv[k + 1]) continue;
                const float v = padded_area(acc, pad) * (float)n + padded_area(rb[k + 1], pad) * (float)rc[k + 1];
                if (v < cost && (n < cnt || rc[k + 1] < cnt)) {
                    cost = v; bk = k; pos = lo_a + width * (float)(k + 1); n_l = n; n_r = rc[k + 1];
                    for (int c = 0; c < 6; ++c) { bl[c] = acc[c]; br[c] = rb[k + 1][c]; }
                }
            }
        }
    }
    // the group's best object split and best spatial split (the lowest axis wins ties, as the host's loops do)
    const int lane = (int)(threadIdx.x & 63u), g0 = lane & ~7;
    int ow = -1, sw = -1; float oc = INFINITY, sc = INFINITY;
    for (int r = 0; r < 3; ++r) {
        const float c1 = __shfl(cost, g0 + r), c2 = __shfl(cost, g0 + 3 + r);
        if (c1 < oc) { oc = c1; ow = r; }
        if (c2 < sc) { sc = c2; sw = 3 + r; }
    }
    // spatial splits are worth trying when there is no object split or its children overlap by more than alpha of the scene's area
    // (the object split's boxes as the host takes them: padded)
    int my_try = 0;
    if ((int)role == ow) {
        float il[3], ih[3]; bool overlap = true;
        for (int c = 0; c < 3; ++c) {
            il[c] = fmaxf(bl[c] - pad, br[c] - pad); ih[c] = fminf(bl[3 + c] + pad, br[3 + c] + pad);
            if (!(il[c] < ih[c])) overlap = false;
        }
        my_try = overlap && half_area3(il, ih) > a.alpha_area ? 1 : 0;
    }
    const int try_sp = ow < 0 ? 1 : __shfl(my_try, g0 + (ow < 0 ? 0 : ow));
    const uint32_t s_nl = __shfl(n_l, g0 + (sw < 0 ? 0 : sw)), s_nr = __shfl(n_r, g0 + (sw < 0 ? 0 : sw));
    uint32_t kind = 1u;                                            // a cell unless a split is found
    if (decide) {
        const bool sp_ok = sw >= 0 && try_sp && sc * a.bias < oc && (uint64_t)s_nl + s_nr - cnt <= (uint64_t)sg.budget;
        if (sp_ok) kind = 3u; else if (ow >= 0) kind = 2u;
    }
    if (!live) return;
    if (kind == 3u && (int)role == sw) {
        sg.axis = role - 3u; sg.bin = (uint32_t)bk; sg.c0 = pos; sg.scale = 0.0f; sg.nl = n_l; sg.nr = n_r;
        for (int c = 0; c < 6; ++c) { sg.sl[c] = bl[c]; sg.sr[c] = br[c]; }
    }
    if (kind == 2u && (int)role == ow) {
        const int ax = (int)role;
        sg.axis = role; sg.bin = (uint32_t)bk; sg.c0 = clo[ax]; sg.scale = (float)kObjBins / (chi[ax] - clo[ax]); sg.nl = n_l; sg.nr = n_r;
    }
    if (role == 7u) sg.kind = kind;
}

// the segments of more than one chunk: their bins were flushed to memory (slot sg.bins_slot)
__global__ __launch_bounds__(256) void k_split_select(SplitArgs a) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, ai = t >> 3, role = t & 7u;
    const bool in_range = ai < a.n_act;
    SplitSeg &sg = a.segs[a.act[in_range ? ai : 0u]];
    const bool live = in_range && sg.count > kChunk;
    select_split(a, sg, a.bins + (size_t)(live ? sg.bins_slot : 0u) * kBinWords, role, live);
}

// ---- flags: which side(s) a reference goes to ----
__device__ __forceinline__ uint64_t side_of(const SplitSeg &sg, const float *rl, const float *rh) {
    const uint64_t kLeft = 1ull << 32, kRight = 1ull;
    const int ax = (int)sg.axis;
    if (sg.kind == 2u) {
        int k = (int)((0.5f * (rl[ax] + rh[ax]) - sg.c0) * sg.scale);
        k = k < 0 ? 0 : (k > kObjBins - 1 ? kObjBins - 1 : k);
        return k <= (int)sg.bin ? kLeft : kRight;
    }
    const float pos = sg.c0;
    if (rh[ax] <= pos) return kLeft;
    if (rl[ax] >= pos) return kRight;
    // a straddler: cut it, or keep it whole on one side when the SAH of the two children as binned says that is cheaper
    float lw[6], rw[6];
    for (int c = 0; c < 3; ++c) {
        lw[c] = fminf(sg.sl[c], rl[c]); lw[3 + c] = fmaxf(sg.sl[3 + c], rh[c]);
        rw[c] = fminf(sg.sr[c], rl[c]); rw[3 + c] = fmaxf(sg.sr[3 + c], rh[c]);
    }
    const float al = half_area3(sg.sl, sg.sl + 3), ar = half_area3(sg.sr, sg.sr + 3);
    const float c_split = al * (float)sg.nl + ar * (float)sg.nr;
    const float c_left = half_area3(lw, lw + 3) * (float)sg.nl + ar * (float)(sg.nr - 1u);
    const float c_right = al * (float)(sg.nl - 1u) + half_area3(rw, rw + 3) * (float)sg.nr;
    if (c_split <= c_left && c_split <= c_right) return kLeft | kRight;
    return c_left <= c_right ? kLeft : kRight;
}

__global__ __launch_bounds__(256) void k_split_flags(SplitArgs a) {
    const uint32_t ai = a.chunk_act[blockIdx.x], off = a.chunk_off[blockIdx.x];
    const SplitSeg &sg = a.segs[a.act[ai]];
    const uint32_t cnt = sg.count, first = sg.first, kind = sg.kind;
    const uint32_t end = off + kChunk < cnt ? off + kChunk : cnt;
    for (uint32_t j = off + threadIdx.x; j < end; j += 256u) {
        uint64_t f = 0ull;
        if (kind >= 2u) {
            const float4 l4 = a.src_lo[first + j], h4 = a.src_hi[first + j];
            const float rl[3] = {l4.x, l4.y, l4.z}, rh[3] = {h4.x, h4.y, h4.z};
            f = side_of(sg, rl, rh);
        }
        a.flags[first + j] = f;
    }
}

// ---- plan: one workgroup turns the level's decisions into child segments, budgets, output ranges and the next level's work list ----
__global__ __launch_bounds__(1024) void k_split_plan(SplitArgs a) {
    typedef hipcub::BlockScan<uint32_t, 1024> Scan;
    __shared__ typename Scan::TempStorage tmp;
    __shared__ uint32_t s_split, s_give_up;
    SplitCounters *cn = a.counters;
    if (threadIdx.x == 0u) { s_split = 0u; s_give_up = 0u; }
    __syncthreads();
    // pass 1: the children's sizes from the scan; a split that does not separate anything is left to PLOC
    for (uint32_t ai = threadIdx.x; ai < a.n_act; ai += 1024u) {
        SplitSeg &sg = a.segs[a.act[ai]];
        if (sg.kind >= 2u) {
            const uint32_t last = sg.first + sg.count - 1u;
            const uint64_t tot = a.scan[last] + a.flags[last] - a.scan[sg.first];
            const uint32_t nl = (uint32_t)(tot >> 32), nr = (uint32_t)(tot & 0xffffffffull);
            if (nl == 0u || nr == 0u || (nl >= sg.count && nr >= sg.count)) sg.kind = 1u;
            else { sg.nl = nl; sg.nr = nr; atomicAdd(&s_split, 1u); }
        }
        if (sg.kind < 2u) sg.kind = 1u;
    }
    __syncthreads();
    const uint32_t seg_base = cn->n_segs, cell_base = cn->n_cells, top_base = cn->n_top, out_base = cn->n_out;
    if (threadIdx.x == 0u && (uint64_t)seg_base + 2ull * s_split > (uint64_t)a.seg_cap) s_give_up = 1u;      // the tables are full: everything left is a cell
    __syncthreads();
    const bool give_up = s_give_up != 0u;
    uint32_t run_split = 0u, run_cells = 0u, run_out = 0u, run_dst = 0u, run_chunks = 0u, run_big = 0u;
    for (uint32_t base = 0; base < a.n_act; base += 1024u) {
        const uint32_t ai = base + threadIdx.x;
        const bool live = ai < a.n_act;
        const uint32_t id = live ? a.act[ai] : 0u;
        SplitSeg *sg = live ? &a.segs[id] : nullptr;
        if (live && give_up) sg->kind = 1u;
        const bool split = live && sg->kind >= 2u, cell = live && !split;
        const uint32_t nl = split ? sg->nl : 0u, nr = split ? sg->nr : 0u;
        const uint32_t v_split = split ? 1u : 0u, v_cell = cell ? 1u : 0u, v_out = cell ? sg->count : 0u, v_dst = nl + nr;
        const uint32_t v_chunks = split ? (nl + kChunk - 1u) / kChunk + (nr + kChunk - 1u) / kChunk : 0u;
        const uint32_t v_big = (nl > kChunk ? 1u : 0u) + (nr > kChunk ? 1u : 0u);      // children whose bins go through memory
        uint32_t e_split, e_cell, e_out, e_dst, e_chunks, e_big, t_split, t_cell, t_out, t_dst, t_chunks, t_big;
        Scan(tmp).ExclusiveSum(v_split, e_split, t_split); __syncthreads();
        Scan(tmp).ExclusiveSum(v_cell, e_cell, t_cell); __syncthreads();
        Scan(tmp).ExclusiveSum(v_out, e_out, t_out); __syncthreads();
        Scan(tmp).ExclusiveSum(v_dst, e_dst, t_dst); __syncthreads();
        Scan(tmp).ExclusiveSum(v_chunks, e_chunks, t_chunks); __syncthreads();
        Scan(tmp).ExclusiveSum(v_big, e_big, t_big); __syncthreads();
        if (cell) {
            sg->out_first = out_base + run_out + e_out;
            sg->index = cell_base + run_cells + e_cell;
            a.cell_seg[sg->index] = id;
        }
        if (split) {
            const uint32_t k = run_split + e_split;                  // this level's k-th split
            sg->index = top_base + k; a.top_seg[sg->index] = id;
            sg->child = seg_base + 2u * k;
            sg->out_first = run_dst + e_dst;
            const uint32_t added = nl + nr - sg->count, rem = sg->budget > added ? sg->budget - added : 0u;
            const uint32_t b_l = (uint32_t)((double)rem * (double)nl / (double)(nl + nr)), b_r = rem - b_l;
            SplitSeg c{};
            c.kind = 0u; c.level = sg->level + 1u;
            for (int q = 0; q < 6; ++q) { c.nb[q] = 0xffffffffu; c.cb[q] = 0xffffffffu; }
            c.first = sg->out_first; c.count = nl; c.budget = b_l; c.bins_slot = run_big + e_big; a.segs[sg->child] = c;
            c.first = sg->out_first + nl; c.count = nr; c.budget = b_r; c.bins_slot = run_big + e_big + (nl > kChunk ? 1u : 0u); a.segs[sg->child + 1u] = c;
            a.act_next[2u * k] = sg->child; a.act_next[2u * k + 1u] = sg->child + 1u;
            uint32_t q = run_chunks + e_chunks;
            for (uint32_t o = 0; o < nl; o += kChunk, ++q) { a.chunk_act_next[q] = 2u * k; a.chunk_off_next[q] = o; }
            for (uint32_t o = 0; o < nr; o += kChunk, ++q) { a.chunk_act_next[q] = 2u * k + 1u; a.chunk_off_next[q] = o; }
        }
        run_split += t_split; run_cells += t_cell; run_out += t_out; run_dst += t_dst; run_chunks += t_chunks; run_big += t_big;
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        cn->n_act = 2u * run_split; cn->n_chunks = run_chunks; cn->src_total = run_dst;
        cn->n_cells = cell_base + run_cells; cn->n_top = top_base + run_split; cn->n_out = out_base + run_out;
        cn->n_segs = seg_base + 2u * run_split; cn->gave_up |= give_up ? 1u : 0u; cn->n_big = run_big;
    }
}

// ---- scatter: retire the cells, write the children of the splits ----
__global__ __launch_bounds__(256) void k_split_scatter(SplitArgs a) {
    const uint32_t ai = a.chunk_act[blockIdx.x], off = a.chunk_off[blockIdx.x];
    const SplitSeg &sg = a.segs[a.act[ai]];
    const uint32_t cnt = sg.count, first = sg.first, kind = sg.kind;
    const uint32_t end = off + kChunk < cnt ? off + kChunk : cnt;
    if (kind < 2u) {
        const uint32_t cell = sg.index, out = sg.out_first;
        for (uint32_t j = off + threadIdx.x; j < end; j += 256u) {
            const float4 l4 = a.src_lo[first + j], h4 = a.src_hi[first + j];
            a.out_lo[out + j] = l4;
            a.out_hi[out + j] = make_float4(h4.x, h4.y, h4.z, __uint_as_float(cell));
        }
        return;
    }
    const uint32_t c_l = sg.child, c_r = sg.child + 1u;
    const uint32_t dst_l = a.segs[c_l].first, dst_r = a.segs[c_r].first;
    const uint64_t sc0 = a.scan[first];
    const int ax = (int)sg.axis; const float pos = sg.c0;
    float mn_l[6], mx_l[6], mn_r[6], mx_r[6];        // [0..2] box, [3..5] centroid
    for (int q = 0; q < 6; ++q) { mn_l[q] = mn_r[q] = INFINITY; mx_l[q] = mx_r[q] = -INFINITY; }
    for (uint32_t j = off + threadIdx.x; j < end; j += 256u) {
        const float4 l4 = a.src_lo[first + j], h4 = a.src_hi[first + j];
        const uint64_t f = a.flags[first + j], sc = a.scan[first + j] - sc0;
        const bool to_l = (f >> 32) != 0ull, to_r = (f & 1ull) != 0ull;
        float ll[3] = {l4.x, l4.y, l4.z}, lh[3] = {h4.x, h4.y, h4.z}, rl[3] = {l4.x, l4.y, l4.z}, rh[3] = {h4.x, h4.y, h4.z};
        if (to_l && to_r) {
            float blo[3] = {l4.x, l4.y, l4.z}, bhi[3] = {h4.x, h4.y, h4.z};
            const uint32_t k = __float_as_uint(l4.w);
            const uint32_t inst = find_instance(a.b.inst_first, a.b.n_inst, k), p = k - a.b.inst_first[inst];
            const bool tri = a.b.inst_kind[inst] == kPrimKindTriangle;
            float v0[3] = {0, 0, 0}, e1[3] = {0, 0, 0}, e2[3] = {0, 0, 0};
            if (tri) {
                const float *src = reinterpret_cast<const float *>(a.b.inst_src[inst]) + 9 * (size_t)p;
                float s9[9], tlo[3], thi[3];
                for (int q = 0; q < 9; ++q) s9[q] = src[q];
                triangle_world(s9, a.b.inst_xf + 12 * (size_t)inst, a.b.inst_identity[inst] != 0u, v0, e1, e2, tlo, thi);
            }
            bhi[ax] = fminf(bhi[ax], pos);
            if (tri) clip_triangle_to_box(v0, e1, e2, blo, bhi, ll, lh); else for (int c = 0; c < 3; ++c) { ll[c] = blo[c]; lh[c] = bhi[c]; }
            bhi[ax] = ax == 0 ? h4.x : (ax == 1 ? h4.y : h4.z);
            blo[ax] = fmaxf(blo[ax], pos);
            if (tri) clip_triangle_to_box(v0, e1, e2, blo, bhi, rl, rh); else for (int c = 0; c < 3; ++c) { rl[c] = blo[c]; rh[c] = bhi[c]; }
        }
        if (to_l) {
            const uint32_t o = dst_l + (uint32_t)(sc >> 32);
            a.dst_lo[o] = make_float4(ll[0], ll[1], ll[2], l4.w); a.dst_hi[o] = make_float4(lh[0], lh[1], lh[2], 0.0f);
            for (int c = 0; c < 3; ++c) {
                const float ce = 0.5f * (ll[c] + lh[c]);
                mn_l[c] = fminf(mn_l[c], ll[c]); mx_l[c] = fmaxf(mx_l[c], lh[c]); mn_l[3 + c] = fminf(mn_l[3 + c], ce); mx_l[3 + c] = fmaxf(mx_l[3 + c], ce);
            }
        }
        if (to_r) {
            const uint32_t o = dst_r + (uint32_t)(sc & 0xffffffffull);
            a.dst_lo[o] = make_float4(rl[0], rl[1], rl[2], l4.w); a.dst_hi[o] = make_float4(rh[0], rh[1], rh[2], 0.0f);
            for (int c = 0; c < 3; ++c) {
                const float ce = 0.5f * (rl[c] + rh[c]);
                mn_r[c] = fminf(mn_r[c], rl[c]); mx_r[c] = fmaxf(mx_r[c], rh[c]); mn_r[3 + c] = fminf(mn_r[3 + c], ce); mx_r[3 + c] = fmaxf(mx_r[3 + c], ce);
            }
        }
    }
    // the children's bounds: one set of atomics per workgroup and child
    block_minmax<6>(mn_l, mx_l);
    if (threadIdx.x == 0u && mn_l[0] <= mx_l[0]) {
        SplitSeg &c = a.segs[c_l];
        for (int q = 0; q < 3; ++q) {
            atomicMin(&c.nb[q], f2ord(mn_l[q])); atomicMin(&c.nb[3 + q], f2ord(-mx_l[q]));
            atomicMin(&c.cb[q], f2ord(mn_l[3 + q])); atomicMin(&c.cb[3 + q], f2ord(-mx_l[3 + q]));
        }
    }
    __syncthreads();
    block_minmax<6>(mn_r, mx_r);
    if (threadIdx.x == 0u && mn_r[0] <= mx_r[0]) {
        SplitSeg &c = a.segs[c_r];
        for (int q = 0; q < 3; ++q) {
            atomicMin(&c.nb[q], f2ord(mn_r[q])); atomicMin(&c.nb[3 + q], f2ord(-mx_r[q]));
            atomicMin(&c.cb[q], f2ord(mn_r[3 + q])); atomicMin(&c.cb[3 + q], f2ord(-mx_r[3 + q]));
        }
    }
}

#define S_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { res.error = _e; res.where = #expr; return res; } } while (0)

}  // namespace

uint32_t gpu_build_max_refs(uint32_t n_prims, const SplitParams *split) {
    if (!split || !split->enabled) return n_prims;
    const double cap = (double)n_prims * (1.0 + std::max(0.0, (double)split->budget_frac));
    return (uint32_t)std::min(cap, 4.0e9);
}

namespace {
struct SplitSizes { uint32_t cap, cell_refs, act_cap, seg_cap, chunk_cap, big_cap; };
SplitSizes split_sizes(uint32_t n, const SplitParams &sp) {
    SplitSizes z{};
    z.cap = gpu_build_max_refs(n, &sp);
    z.cell_refs = std::max(sp.cell_refs, 8u);
    z.act_cap = 2u * (z.cap / z.cell_refs) + 2u;      // at most cap / cell_refs segments are split on a level, so a level has at most twice as many segments
    z.seg_cap = (uint32_t)std::min<uint64_t>(4ull * z.act_cap + 64u, 1u << 31);       // (twice what balanced splits make; when it runs out, what is left goes to PLOC as it is)
    z.big_cap = z.cap / kChunk + 2u;                  // segments of more than one chunk on a level: their bins go through memory
    z.chunk_cap = z.cap / kChunk + z.act_cap + 2u;
    return z;
}
}  // namespace

size_t gpu_split_table_bytes(uint32_t n_prims, const SplitParams &sp) {
    const SplitSizes z = split_sizes(n_prims, sp);
    return (size_t)z.seg_cap * (sizeof(SplitSeg) + 8u) + (size_t)z.big_cap * sizeof(uint32_t) * kBinWords + (size_t)z.act_cap * 8u + (size_t)z.chunk_cap * 16u + (1u << 20);
}

SplitPhaseResult gpu_split_phase(const GpuBuildArgs &b, uint32_t n_valid, const SplitParams &sp, BuildArena &arena, hipStream_t s) {
    SplitPhaseResult res{};
    const SplitSizes z = split_sizes(n_valid, sp);
    const uint32_t cap = z.cap, budget = cap - n_valid, cell_refs = z.cell_refs, act_cap = z.act_cap, seg_cap = z.seg_cap, chunk_cap = z.chunk_cap;
    SplitArgs a{};
    a.b = b; a.seg_cap = seg_cap; a.cell_refs = cell_refs; a.pad = sp.pad; a.bias = sp.bias;
    float4 *buf[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t *act[2] = {nullptr, nullptr}, *ch_act[2] = {nullptr, nullptr}, *ch_off[2] = {nullptr, nullptr};
    void *temp = nullptr; size_t scan_bytes = 0;
    // what outlives the phase first: the caller rewinds the arena to res's mark afterwards
    S_TRY(arena.alloc((void **)&a.out_lo, sizeof(float4) * (size_t)cap)); S_TRY(arena.alloc((void **)&a.out_hi, sizeof(float4) * (size_t)cap));
    S_TRY(arena.alloc((void **)&a.segs, sizeof(SplitSeg) * (size_t)seg_cap));
    S_TRY(arena.alloc((void **)&a.top_seg, sizeof(uint32_t) * (size_t)seg_cap)); S_TRY(arena.alloc((void **)&a.cell_seg, sizeof(uint32_t) * (size_t)seg_cap));
    const size_t mark = arena.used;
    for (int q = 0; q < 4; ++q) S_TRY(arena.alloc((void **)&buf[q], sizeof(float4) * (size_t)cap));
    S_TRY(arena.alloc((void **)&a.flags, sizeof(uint64_t) * (size_t)std::max(cap, b.n))); S_TRY(arena.alloc((void **)&a.scan, sizeof(uint64_t) * (size_t)std::max(cap, b.n)));
    S_TRY(arena.alloc((void **)&a.bins, sizeof(uint32_t) * (size_t)kBinWords * z.big_cap));
    for (int q = 0; q < 2; ++q) {
        S_TRY(arena.alloc((void **)&act[q], sizeof(uint32_t) * (size_t)act_cap));
        S_TRY(arena.alloc((void **)&ch_act[q], sizeof(uint32_t) * (size_t)chunk_cap)); S_TRY(arena.alloc((void **)&ch_off[q], sizeof(uint32_t) * (size_t)chunk_cap));
    }
    S_TRY(arena.alloc((void **)&a.counters, sizeof(SplitCounters)));
    S_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, a.flags, a.scan, (int)std::max(cap, b.n), s));
    S_TRY(arena.alloc(&temp, scan_bytes));

    {   // the scene's area for the alpha test: the host takes the unpadded box of all primitives
        BuildCounters h{};
        S_TRY(hipMemcpyAsync(&h, b.counters, sizeof h, hipMemcpyDeviceToHost, s));
        S_TRY(hipStreamSynchronize(s));
        float lo[3], hi[3];
        for (int d = 0; d < 3; ++d) { lo[d] = ord2f(h.bmin[d]); hi[d] = ord2f(h.bmax[d]); }
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        a.alpha_area = sp.alpha * (dx * dy + dy * dz + dz * dx);
    }
    // references of the valid primitives, in primitive order
    hipLaunchKernelGGL(k_split_valid_flags, dim3(blocks(b.n, 256)), dim3(256), 0, s, b, a.flags);
    S_TRY(hipcub::DeviceScan::ExclusiveSum(temp, scan_bytes, a.flags, a.scan, (int)b.n, s));
    hipLaunchKernelGGL(k_split_init_refs, dim3(blocks(b.n, 256)), dim3(256), 0, s, b, a.scan, buf[0], buf[1]);
    int src = 0;
    a.src_lo = buf[0]; a.src_hi = buf[1]; a.dst_lo = buf[2]; a.dst_hi = buf[3];
    a.act = act[0]; a.act_next = act[1]; a.chunk_act = ch_act[0]; a.chunk_off = ch_off[0]; a.chunk_act_next = ch_act[1]; a.chunk_off_next = ch_off[1];
    a.n_act = 1u; a.n_chunks = blocks(n_valid, kChunk); a.level = 0u;
    hipLaunchKernelGGL(k_split_first_level, dim3(blocks(a.n_chunks, 256)), dim3(256), 0, s, a, n_valid, budget);
    S_TRY(hipGetLastError());
    uint32_t src_total = n_valid, n_big = n_valid > kChunk ? 1u : 0u;
    SplitCounters h{};
    for (;;) {
        if (n_big) hipLaunchKernelGGL(k_split_init_bins, dim3(blocks(n_big * (uint32_t)kBinWords, 256)), dim3(256), 0, s, a.bins, n_big * (uint32_t)kBinWords);
        hipLaunchKernelGGL(k_split_bin, dim3(a.n_chunks), dim3(256), 0, s, a);
        if (n_big) hipLaunchKernelGGL(k_split_select, dim3(blocks(a.n_act * 8u, 256)), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_split_flags, dim3(a.n_chunks), dim3(256), 0, s, a);
        S_TRY(hipcub::DeviceScan::ExclusiveSum(temp, scan_bytes, a.flags, a.scan, (int)src_total, s));
        hipLaunchKernelGGL(k_split_plan, dim3(1), dim3(1024), 0, s, a);
        hipLaunchKernelGGL(k_split_scatter, dim3(a.n_chunks), dim3(256), 0, s, a);
        S_TRY(hipGetLastError());
        S_TRY(hipMemcpyAsync(&h, a.counters, sizeof h, hipMemcpyDeviceToHost, s));
        S_TRY(hipStreamSynchronize(s));
        if (sp.verbose)
            std::fprintf(stderr, "[hrt] split level %u: %u segments (%u references) -> %u children (%u references), %u cells so far (%u references)\n",
                         a.level, a.n_act, src_total, h.n_act, h.src_total, h.n_cells, h.n_out);
        ++res.levels;
        if (h.n_act == 0u) break;
        if (h.n_act > act_cap || h.n_chunks > chunk_cap || h.src_total > cap || h.n_big > z.big_cap || a.level > 2u * kMaxLevels) { res.error = hipErrorUnknown; res.where = "split phase: a level outgrew its tables"; return res; }
        src ^= 1;
        a.src_lo = buf[2 * src]; a.src_hi = buf[2 * src + 1]; a.dst_lo = buf[2 * (src ^ 1)]; a.dst_hi = buf[2 * (src ^ 1) + 1];
        a.act = act[src]; a.act_next = act[src ^ 1]; a.chunk_act = ch_act[src]; a.chunk_off = ch_off[src]; a.chunk_act_next = ch_act[src ^ 1]; a.chunk_off_next = ch_off[src ^ 1];
        a.n_act = h.n_act; a.n_chunks = h.n_chunks; src_total = h.src_total; n_big = h.n_big; ++a.level;
    }
    res.n_refs = h.n_out; res.n_cells = h.n_cells; res.n_top = h.n_top;
    res.ref_lo = a.out_lo; res.ref_hi = a.out_hi; res.segs = a.segs; res.top_seg = a.top_seg; res.cell_seg = a.cell_seg;
    if (res.n_refs < n_valid || res.n_refs > cap || res.n_cells == 0u || res.n_top + 1u != res.n_cells) { res.error = hipErrorUnknown; res.where = "split phase: inconsistent counts"; return res; }
    arena.used = mark;          // the temporaries go back (what came from hipMalloc instead is freed with the arena)
    return res;
}

}  // namespace hrt
