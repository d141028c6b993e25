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
// One level = a dozen short launches over the references still in play (retired cells are copied out once); a wave per 512-reference
// chunk of a segment, so the many small segments of the deep levels cost a wave each and no barrier:
//   bins      16 object bins + 32 spatial bins x 3 axes in LDS (ordered-integer min / max, so the result does not depend on the
//             order of arrival), then one flush per chunk -- or none: a segment of one chunk is decided by the wave that binned it;
//   select    the wave's lanes each evaluate one candidate plane (45 object, 93 spatial), argmin by shuffles: object split,
//             spatial split or "leave it to PLOC";
//   flags     per reference: left, right or both (a straddler stays whole on one side when that is cheaper: unsplitting);
//   scan      hipCUB exclusive sum of (left << 32 | right): positions come from prefix sums, so the order of the references -- and
//             with it the tree -- is the same in every run;
//   plan      per segment: the children's sizes; three more exclusive sums give child segments, budgets, the next level's chunk
//             list, cell / top-node numbers;
//   clip      the references that are cut, compacted: the triangle clipped to each child's box (bvh8_geom.h, the host's arithmetic:
//             Sutherland-Hodgman in double -- slow code, so every lane of the waves that run it has work);
//   scatter   write the children, reduce their bounds per wave.
// HBM-bound integer / min-max work; nothing here is GEMM-shaped.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
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

#ifndef HRT_SPLIT_OBJ_BINS
#define HRT_SPLIT_OBJ_BINS 16
#endif
#ifndef HRT_SPLIT_SP_BINS
#define HRT_SPLIT_SP_BINS 32
#endif
constexpr int kObjBins = HRT_SPLIT_OBJ_BINS, kSpBins = HRT_SPLIT_SP_BINS;      // (experiments: make EXTRA_HIPFLAGS=-DHRT_SPLIT_OBJ_BINS=32)
constexpr int kObjBox = 0, kSpBox = 3 * kObjBins * 6;                                  // boxes: 6 words per bin (lo as f2ord(x), hi as f2ord(-x))
constexpr int kBinMinMax = kSpBox + 3 * kSpBins * 6;                                   // 864 words that shrink under atomicMin (identity ~0)
constexpr int kObjCnt = kBinMinMax, kSpEnter = kObjCnt + 3 * kObjBins, kSpLeave = kSpEnter + 3 * kSpBins;
constexpr int kBinWords = kSpLeave + 3 * kSpBins;                                      // 1104 words per segment
constexpr uint32_t kChunk = 512u, kMaxLevels = 64u, kWave = 64u;

struct SplitCounters { uint32_t n_act, n_chunks, src_total, n_cells, n_top, n_out, n_segs, gave_up, n_big, retry, n_cut; float bulk_area; uint32_t bulk_area_next; };      // bulk_area_next: float bits (a positive float orders like its bits), 0 = no candidate this level

struct SplitArgs {
    GpuBuildArgs b;                                  // instance tables (the scatter clips triangles), primitive bounds, scene counters
    SplitSeg *segs; uint32_t seg_cap;
    uint32_t *act, *act_next;                        // segments of this level / the next
    uint32_t *chunk_act, *chunk_off, *chunk_act_next, *chunk_off_next;       // chunk -> (segment, offset within the segment)
    float4 *src_lo, *src_hi, *dst_lo, *dst_hi, *out_lo, *out_hi;
    uint32_t *bins; uint64_t *flags, *scan;
    uint64_t *plan_in[3], *plan_ex[3];               // per segment of the level: (splits | cells), (references out | to the children), (chunks | big children) and their exclusive sums
    uint32_t *top_seg, *cell_seg;
    uint2 *cut_list; float *cut_box; uint32_t cut_cap;      // the level's straddlers that are cut: (reference, segment) and the two clipped boxes, 12 floats
    SplitCounters *counters;
    uint32_t n_act, n_chunks, cell_refs, level;
    uint32_t seg_base, cell_base, top_base, out_base;       // the running totals before this level (the host's copy of the counters)
    float pad, alpha, bias, cut_bias; uint32_t n_total;
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
    {
        float lo[3], hi[3];
        seg_bounds(sg.nb, lo, hi);
        sc.bulk_area = half_area3(lo, hi); sc.bulk_area_next = 0u;
    }
    *a.counters = sc;
}

__device__ __forceinline__ void select_split(const SplitArgs &a, SplitSeg &sg, const uint32_t *g, uint32_t lane);

// ---- bins ----
__global__ __launch_bounds__(256) void k_split_init_bins(uint32_t *bins, uint32_t n_words) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n_words) bins[i] = (i % (uint32_t)kBinWords) < (uint32_t)kBinMinMax ? 0xffffffffu : 0u;
}

__global__ __launch_bounds__(64) void k_split_bin(SplitArgs a) {
    __shared__ uint32_t s_bins[kBinWords];
    const uint32_t ai = a.chunk_act[blockIdx.x], off = a.chunk_off[blockIdx.x];
    const SplitSeg &sg = a.segs[ai];
    const uint32_t cnt = sg.count, first = sg.first;
    if (cnt < a.cell_refs || sg.level >= kMaxLevels) return;       // a cell: nothing to decide (the same for the whole workgroup)
    for (uint32_t w = threadIdx.x; w < (uint32_t)kBinWords; w += kWave) s_bins[w] = w < (uint32_t)kBinMinMax ? 0xffffffffu : 0u;
    float nlo[3], nhi[3], clo[3], chi[3];
    seg_bounds(sg.nb, nlo, nhi); seg_bounds(sg.cb, clo, chi);
    const bool spatial = sg.budget > 0u;
    __syncthreads();
    const uint32_t end = off + kChunk < cnt ? off + kChunk : cnt;
    for (uint32_t j = off + threadIdx.x; j < end; j += kWave) {
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
    if (cnt <= kChunk) { select_split(a, a.segs[ai], s_bins, threadIdx.x); return; }      // the whole segment was this wave's: decide here, the bins never leave the LDS
    uint32_t *g = a.bins + (size_t)sg.bins_slot * kBinWords;
    for (uint32_t w = threadIdx.x; w < (uint32_t)kBinWords; w += kWave) {
        const uint32_t v = s_bins[w];
        if (w < (uint32_t)kBinMinMax) { if (v != 0xffffffffu) atomicMin(&g[w], v); } else if (v) atomicAdd(&g[w], v);
    }
}

// ---- select: a wave per segment, a candidate plane per lane ----
__device__ __forceinline__ bool bin_box(const uint32_t *w, float *b) {
    if (w[0] == 0xffffffffu) return false;
    for (int c = 0; c < 3; ++c) { b[c] = ord2f(w[c]); b[3 + c] = -ord2f(w[3 + c]); }
    return true;
}
// the cheapest candidate of the wave; equal costs: the lowest candidate number (the host's loops keep the first of equals)
__device__ __forceinline__ void wave_argmin(float &cost, uint32_t &idx) {
    for (int off = 32; off > 0; off >>= 1) {
        const float oc = __shfl_xor(cost, off); const uint32_t oi = __shfl_xor(idx, off);
        if (oc < cost || (oc == cost && oi < idx)) { cost = oc; idx = oi; }
    }
}

// (all 64 lanes of a wave call this together; g = the segment's bins, in LDS or in memory)
__device__ __forceinline__ void select_split(const SplitArgs &a, SplitSeg &sg, const uint32_t *g, uint32_t lane) {
    const uint32_t cnt = sg.count;
    if (cnt < a.cell_refs || sg.level >= kMaxLevels) { if (lane == 0u) sg.kind = 1u; return; }
    const float pad = a.pad;
    float nlo[3], nhi[3], clo[3], chi[3];
    seg_bounds(sg.nb, nlo, nhi); seg_bounds(sg.cb, clo, chi);
    // The paper tries spatial splits where the object split's children overlap by more than alpha of the SCENE's area.  A scene with an
    // outlier -- the reference's: a ground sphere of radius 1000 beside particles of size 0.1 -- has a root box that says nothing about
    // the geometry, and no overlap would ever pass.  The yardstick is therefore the box of the BULK: the segment that still holds nine
    // tenths of all references, followed down from the root (it takes effect a level later: plan_apply moves it, so that every segment
    // of a level is judged by the same number whatever the order of the waves).  After heavy spatial splitting two segments of a level
    // can both hold nine tenths of the ORIGINAL count: the larger box wins, whichever wave comes last (atomicMax on the float's bits).
    if (lane == 0u && (uint64_t)cnt * 10u >= (uint64_t)a.n_total * 9u) atomicMax(&a.counters->bulk_area_next, __float_as_uint(fmaxf(half_area3(nlo, nhi), 0.0f)));
    // ---- object split: candidate c = (axis, k): bins 0..k of the axis go left ----
    float o_cost = INFINITY, o_l[6], o_r[6]; uint32_t o_idx = 0xffffffffu, o_nl = 0u, o_nr = 0u;
    for (uint32_t c = lane; c < 3u * (uint32_t)(kObjBins - 1); c += kWave) {
        const int ax = (int)(c / (uint32_t)(kObjBins - 1)), k = (int)(c % (uint32_t)(kObjBins - 1));
        if (!(chi[ax] - clo[ax] > 0.0f)) continue;
        float l[6], r[6]; box_reset(l); box_reset(r); uint32_t nl = 0u, nr = 0u;
        for (int j = 0; j < kObjBins; ++j) {
            const uint32_t cj = g[kObjCnt + ax * kObjBins + j];
            float b[6]; const bool has = bin_box(g + kObjBox + (ax * kObjBins + j) * 6, b);
            if (j <= k) { nl += cj; if (has) box_grow(l, b); } else { nr += cj; if (has) box_grow(r, b); }
        }
        if (nl == 0u || nr == 0u) continue;
        const float v = padded_area(l, pad) * (float)nl + padded_area(r, pad) * (float)nr;
        if (v < o_cost) { o_cost = v; o_idx = c; o_nl = nl; o_nr = nr; for (int q = 0; q < 6; ++q) { o_l[q] = l[q]; o_r[q] = r[q]; } }
    }
    float oc = o_cost; uint32_t ow = o_idx;
    wave_argmin(oc, ow);
    const bool obj_ok = ow != 0xffffffffu, i_am_obj = obj_ok && o_idx == ow;
    // spatial splits are worth trying when there is no object split or its children overlap by more than alpha of the scene's area
    // (the object split's boxes as the host takes them: padded)
    bool my_try = false;
    if (i_am_obj) {
        float il[3], ih[3]; bool overlap = true;
        for (int c = 0; c < 3; ++c) {
            il[c] = fmaxf(o_l[c] - pad, o_r[c] - pad); ih[c] = fminf(o_l[3 + c] + pad, o_r[3 + c] + pad);
            if (!(il[c] < ih[c])) overlap = false;
        }
        my_try = overlap && half_area3(il, ih) > a.alpha * a.counters->bulk_area;
    }
    const bool try_sp = sg.budget > 0u && (!obj_ok || __ballot(my_try) != 0ull);
    // ---- spatial split: candidate c = (axis, k): the plane between bins k and k + 1 ----
    float s_cost = INFINITY, s_l[6], s_r[6], s_pos = 0.0f; uint32_t s_idx = 0xffffffffu, s_nl = 0u, s_nr = 0u;
    if (try_sp)
        for (uint32_t c = lane; c < 3u * (uint32_t)(kSpBins - 1); c += kWave) {
            const int ax = (int)(c / (uint32_t)(kSpBins - 1)), k = (int)(c % (uint32_t)(kSpBins - 1));
            const float lo_a = nlo[ax], ext = nhi[ax] - nlo[ax];
            if (!(ext > 0.0f)) continue;
            float l[6], r[6]; box_reset(l); box_reset(r); uint32_t nl = 0u, nr = 0u; bool hl = false, hr = false;
            for (int j = 0; j < kSpBins; ++j) {
                float b[6]; const bool has = bin_box(g + kSpBox + (ax * kSpBins + j) * 6, b);
                if (j <= k) { nl += g[kSpEnter + ax * kSpBins + j]; if (has) { box_grow(l, b); hl = true; } }
                else { nr += g[kSpLeave + ax * kSpBins + j]; if (has) { box_grow(r, b); hr = true; } }
            }
            if (nl == 0u || nr == 0u || !hl || !hr || !(nl < cnt || nr < cnt)) continue;
            const float v = padded_area(l, pad) * (float)nl + padded_area(r, pad) * (float)nr;
            if (v < s_cost) {
                s_cost = v; s_idx = c; s_nl = nl; s_nr = nr; s_pos = lo_a + (ext / (float)kSpBins) * (float)(k + 1);
                for (int q = 0; q < 6; ++q) { s_l[q] = l[q]; s_r[q] = r[q]; }
            }
        }
    float sc = s_cost; uint32_t sw = s_idx;
    wave_argmin(sc, sw);
    const bool sp_found = sw != 0xffffffffu, i_am_sp = sp_found && s_idx == sw;
    const unsigned long long sp_mask = __ballot(i_am_sp);
    const int sp_lane = sp_mask ? __ffsll((long long)sp_mask) - 1 : 0;
    const uint32_t w_nl = __shfl(s_nl, sp_lane), w_nr = __shfl(s_nr, sp_lane);
    uint32_t kind = 1u;                                            // a cell unless a split is found
    if (sp_found && sc * a.bias < oc && (uint64_t)w_nl + w_nr - cnt <= (uint64_t)sg.budget) kind = 3u;      // (oc is infinite without an object split)
    else if (obj_ok) kind = 2u;
    if (kind == 3u && i_am_sp) {
        sg.axis = s_idx / (uint32_t)(kSpBins - 1); sg.bin = s_idx % (uint32_t)(kSpBins - 1); sg.c0 = s_pos; sg.scale = 0.0f; sg.nl = s_nl; sg.nr = s_nr;
        for (int c = 0; c < 6; ++c) { sg.sl[c] = s_l[c]; sg.sr[c] = s_r[c]; }
    }
    if (kind == 2u && i_am_obj) {
        const int ax = (int)(o_idx / (uint32_t)(kObjBins - 1));
        sg.axis = (uint32_t)ax; sg.bin = o_idx % (uint32_t)(kObjBins - 1); sg.c0 = clo[ax]; sg.scale = (float)kObjBins / (chi[ax] - clo[ax]); sg.nl = o_nl; sg.nr = o_nr;
    }
    if (lane == 0u) sg.kind = kind;
}

// the segments of more than one chunk: their bins were flushed to memory (slot sg.bins_slot); a wave each
__global__ __launch_bounds__(64) void k_split_select(SplitArgs a) {
    SplitSeg &sg = a.segs[a.act[blockIdx.x]];
    if (sg.count <= kChunk) return;
    select_split(a, sg, a.bins + (size_t)sg.bins_slot * kBinWords, threadIdx.x);
}

// ---- flags: which side(s) a reference goes to ----
__device__ __forceinline__ uint64_t side_of(const SplitSeg &sg, const float *rl, const float *rh, float cut_bias) {
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
    if (c_split * cut_bias <= c_left && c_split * cut_bias <= c_right) return kLeft | kRight;
    return c_left <= c_right ? kLeft : kRight;
}

__global__ __launch_bounds__(64) void k_split_flags(SplitArgs a) {
    const uint32_t ai = a.chunk_act[blockIdx.x], off = a.chunk_off[blockIdx.x];
    const SplitSeg &sg = a.segs[ai];
    const uint32_t cnt = sg.count, first = sg.first, kind = sg.kind;
    const uint32_t end = off + kChunk < cnt ? off + kChunk : cnt;
    for (uint32_t j = off + threadIdx.x; j < end; j += kWave) {
        uint64_t f = 0ull;
        if (kind >= 2u) {
            const float4 l4 = a.src_lo[first + j], h4 = a.src_hi[first + j];
            const float rl[3] = {l4.x, l4.y, l4.z}, rh[3] = {h4.x, h4.y, h4.z};
            f = side_of(sg, rl, rh, a.cut_bias);
        }
        // cut: the clipping is a kernel of its own over these (every lane busy); slots from one atomic per wave, the slot rides in the reference
        const bool cut = f == ((1ull << 32) | 1ull);
        const unsigned long long cut_mask = __ballot(cut);
        if (cut_mask) {
            const int leader = __ffsll((long long)cut_mask) - 1;
            uint32_t base = 0u;
            if ((int)threadIdx.x == leader) base = atomicAdd(&a.counters->n_cut, (uint32_t)__popcll(cut_mask));
            base = __shfl(base, leader);
            if (cut) {
                const uint32_t slot = base + (uint32_t)__popcll(cut_mask & ((1ull << threadIdx.x) - 1ull));
                if (slot < a.cut_cap) { a.cut_list[slot] = make_uint2(first + j, ai); a.src_hi[first + j].w = __uint_as_float(slot); }
                else f = 1ull << 32;                   // (no room: the reference stays whole)
            }
        }
        a.flags[first + j] = f;
    }
}

// ---- clip: the two boxes of every reference that is cut (Sutherland-Hodgman in double: slow code, so it runs with all lanes busy) ----
__global__ __launch_bounds__(64) void k_split_clip(SplitArgs a) {
    const uint32_t n = a.counters->n_cut < a.cut_cap ? a.counters->n_cut : a.cut_cap;
    for (uint32_t slot = blockIdx.x * kWave + threadIdx.x; slot < n; slot += gridDim.x * kWave) {
        const uint2 e = a.cut_list[slot];
        const SplitSeg &sg = a.segs[e.y];
        const float4 l4 = a.src_lo[e.x], h4 = a.src_hi[e.x];
        const int ax = (int)sg.axis; const float pos = sg.c0;
        float blo[3] = {l4.x, l4.y, l4.z}, bhi[3] = {h4.x, h4.y, h4.z}, ll[3], lh[3], rl[3], rh[3];
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
        const float whole_hi = bhi[ax];
        bhi[ax] = fminf(bhi[ax], pos);
        if (tri) clip_triangle_to_box(v0, e1, e2, blo, bhi, ll, lh); else for (int c = 0; c < 3; ++c) { ll[c] = blo[c]; lh[c] = bhi[c]; }
        bhi[ax] = whole_hi;
        blo[ax] = fmaxf(blo[ax], pos);
        if (tri) clip_triangle_to_box(v0, e1, e2, blo, bhi, rl, rh); else for (int c = 0; c < 3; ++c) { rl[c] = blo[c]; rh[c] = bhi[c]; }
        float *o = a.cut_box + 12 * (size_t)slot;
        for (int c = 0; c < 3; ++c) { o[c] = ll[c]; o[3 + c] = lh[c]; o[6 + c] = rl[c]; o[9 + c] = rh[c]; }
    }
}

// ---- plan: the level's decisions become child segments, budgets, output ranges and the next level's work list ----
// count: per segment, what it contributes to the six running totals (three packed pairs, scanned by hipCUB)
__global__ __launch_bounds__(256) void k_split_plan_count(SplitArgs a) {
    const uint32_t ai = blockIdx.x * 256u + threadIdx.x;
    if (ai >= a.n_act) return;
    SplitSeg &sg = a.segs[a.act[ai]];
    if (sg.kind >= 2u) {       // the children's sizes from the scan; a split that does not separate anything is left to PLOC
        const uint32_t last = sg.first + sg.count - 1u;
        const uint64_t tot = a.scan[last] + a.flags[last] - a.scan[sg.first];
        const uint32_t nl = (uint32_t)(tot >> 32), nr = (uint32_t)(tot & 0xffffffffull);
        if (nl == 0u || nr == 0u || (nl >= sg.count && nr >= sg.count)) sg.kind = 1u; else { sg.nl = nl; sg.nr = nr; }
    }
    if (sg.kind < 2u) sg.kind = 1u;
    const bool split = sg.kind >= 2u;
    const uint64_t nl = split ? sg.nl : 0u, nr = split ? sg.nr : 0u;
    a.plan_in[0][ai] = split ? 1ull << 32 : 1ull;                                        // splits | cells
    a.plan_in[1][ai] = split ? nl + nr : (uint64_t)sg.count << 32;                       // references retired | references to the children
    a.plan_in[2][ai] = split ? (((nl + kChunk - 1u) / kChunk + (nr + kChunk - 1u) / kChunk) << 32) | ((nl > kChunk ? 1u : 0u) + (nr > kChunk ? 1u : 0u)) : 0ull;   // chunks | big children
}
// (the segment tables are full: what is left of this level goes to PLOC as it is -- the host runs count / scan / apply again)
__global__ __launch_bounds__(256) void k_split_force_cells(SplitArgs a) {
    const uint32_t ai = blockIdx.x * 256u + threadIdx.x;
    if (ai < a.n_act) a.segs[a.act[ai]].kind = 1u;
    if (ai == 0u) { a.counters->retry = 0u; a.counters->gave_up = 1u; }
}
__global__ __launch_bounds__(256) void k_split_plan_apply(SplitArgs a) {
    const uint32_t ai = blockIdx.x * 256u + threadIdx.x;
    if (ai >= a.n_act) return;
    SplitCounters *cn = a.counters;
    const uint32_t last = a.n_act - 1u;
    const uint64_t t0 = a.plan_ex[0][last] + a.plan_in[0][last], t1 = a.plan_ex[1][last] + a.plan_in[1][last], t2 = a.plan_ex[2][last] + a.plan_in[2][last];
    const uint32_t tot_split = (uint32_t)(t0 >> 32), tot_cells = (uint32_t)t0, tot_out = (uint32_t)(t1 >> 32), tot_dst = (uint32_t)t1;
    const uint32_t seg_base = a.seg_base, cell_base = a.cell_base, top_base = a.top_base, out_base = a.out_base;
    if ((uint64_t)seg_base + 2ull * tot_split > (uint64_t)a.seg_cap) { if (ai == 0u) cn->retry = 1u; return; }
    const uint32_t id = a.act[ai];
    SplitSeg *sg = &a.segs[id];
    const uint64_t e0 = a.plan_ex[0][ai], e1 = a.plan_ex[1][ai], e2 = a.plan_ex[2][ai];
    if (sg->kind < 2u) {
        sg->out_first = out_base + (uint32_t)(e1 >> 32);
        sg->index = cell_base + (uint32_t)e0;
        a.cell_seg[sg->index] = id;
    } else {
        const uint32_t k = (uint32_t)(e0 >> 32), nl = sg->nl, nr = sg->nr;       // this level's k-th split
        sg->index = top_base + k; a.top_seg[sg->index] = id;
        sg->child = seg_base + 2u * k;
        sg->out_first = (uint32_t)e1;
        const uint32_t added = nl + nr - sg->count, rem = sg->budget > added ? sg->budget - added : 0u;
        const uint32_t b_l = (uint32_t)((double)rem * (double)nl / (double)(nl + nr)), b_r = rem - b_l;
        const uint32_t big0 = (uint32_t)e2;
        SplitSeg c{};
        c.kind = 0u; c.level = sg->level + 1u;
        for (int q = 0; q < 6; ++q) { c.nb[q] = 0xffffffffu; c.cb[q] = 0xffffffffu; }
        c.first = sg->out_first; c.count = nl; c.budget = b_l; c.bins_slot = big0; a.segs[sg->child] = c;
        c.first = sg->out_first + nl; c.count = nr; c.budget = b_r; c.bins_slot = big0 + (nl > kChunk ? 1u : 0u); a.segs[sg->child + 1u] = c;
        a.act_next[2u * k] = sg->child; a.act_next[2u * k + 1u] = sg->child + 1u;
        uint32_t q = (uint32_t)(e2 >> 32);
        for (uint32_t o = 0; o < nl; o += kChunk, ++q) { a.chunk_act_next[q] = sg->child; a.chunk_off_next[q] = o; }
        for (uint32_t o = 0; o < nr; o += kChunk, ++q) { a.chunk_act_next[q] = sg->child + 1u; a.chunk_off_next[q] = o; }
    }
    if (ai == 0u) {
        cn->n_act = 2u * tot_split; cn->n_chunks = (uint32_t)(t2 >> 32); cn->src_total = tot_dst; cn->n_big = (uint32_t)t2;
        cn->n_cells = cell_base + tot_cells; cn->n_top = top_base + tot_split; cn->n_out = out_base + tot_out;
        cn->n_segs = seg_base + 2u * tot_split; cn->retry = 0u; cn->n_cut = 0u; if (cn->bulk_area_next != 0u) cn->bulk_area = __uint_as_float(cn->bulk_area_next); cn->bulk_area_next = 0u;
    }
}

// ---- scatter: retire the cells, write the children of the splits ----
__global__ __launch_bounds__(64) void k_split_scatter(SplitArgs a) {
    if (a.counters->retry) return;                   // (the plan did not fit the tables: the host redoes it with cells only)
    const uint32_t ai = a.chunk_act[blockIdx.x], off = a.chunk_off[blockIdx.x];
    const SplitSeg &sg = a.segs[ai];
    const uint32_t cnt = sg.count, first = sg.first, kind = sg.kind;
    const uint32_t end = off + kChunk < cnt ? off + kChunk : cnt;
    if (kind < 2u) {
        const uint32_t cell = sg.index, out = sg.out_first;
        for (uint32_t j = off + threadIdx.x; j < end; j += kWave) {
            const float4 l4 = a.src_lo[first + j], h4 = a.src_hi[first + j];
            a.out_lo[out + j] = l4;
            a.out_hi[out + j] = make_float4(h4.x, h4.y, h4.z, __uint_as_float(cell));
        }
        return;
    }
    const uint32_t c_l = sg.child, c_r = sg.child + 1u;
    const uint32_t dst_l = sg.out_first, dst_r = sg.out_first + sg.nl;      // (= the children's `first`)
    const uint64_t sc0 = a.scan[first];
    float mn_l[6], mx_l[6], mn_r[6], mx_r[6];        // [0..2] box, [3..5] centroid
    for (int q = 0; q < 6; ++q) { mn_l[q] = mn_r[q] = INFINITY; mx_l[q] = mx_r[q] = -INFINITY; }
    for (uint32_t j = off + threadIdx.x; j < end; j += kWave) {
        const float4 l4 = a.src_lo[first + j], h4 = a.src_hi[first + j];
        const uint64_t f = a.flags[first + j], sc = a.scan[first + j] - sc0;
        const bool to_l = (f >> 32) != 0ull, to_r = (f & 1ull) != 0ull;
        float ll[3] = {l4.x, l4.y, l4.z}, lh[3] = {h4.x, h4.y, h4.z}, rl[3] = {l4.x, l4.y, l4.z}, rh[3] = {h4.x, h4.y, h4.z};
        if (to_l && to_r) {
            const float *cb = a.cut_box + 12 * (size_t)__float_as_uint(h4.w);      // (k_split_clip)
            for (int c = 0; c < 3; ++c) { ll[c] = cb[c]; lh[c] = cb[3 + c]; rl[c] = cb[6 + c]; rh[c] = cb[9 + c]; }
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
    z.cell_refs = std::max(sp.cell_refs, 2u);
    z.act_cap = 2u * (z.cap / z.cell_refs) + 2u;      // at most cap / cell_refs segments are split on a level, so a level has at most twice as many segments
    z.seg_cap = (uint32_t)std::min<uint64_t>(4ull * z.act_cap + 64u, 1u << 31);       // (twice what balanced splits make; when it runs out, what is left goes to PLOC as it is)
    z.big_cap = z.cap / kChunk + 2u;                  // segments of more than one chunk on a level: their bins go through memory
    z.chunk_cap = z.cap / kChunk + z.act_cap + 2u;
    return z;
}
}  // namespace

size_t gpu_split_table_bytes(uint32_t n_prims, const SplitParams &sp) {
    const SplitSizes z = split_sizes(n_prims, sp);
    return (size_t)z.seg_cap * (sizeof(SplitSeg) + 8u) + (size_t)z.big_cap * sizeof(uint32_t) * kBinWords + (size_t)z.act_cap * (8u + 48u) + (size_t)z.chunk_cap * 16u + (1u << 20);
}

SplitPhaseResult gpu_split_phase(const GpuBuildArgs &b, uint32_t n_valid, const SplitParams &sp, BuildArena &arena, hipStream_t s) {
    SplitPhaseResult res{};
    const SplitSizes z = split_sizes(n_valid, sp);
    const uint32_t cap = z.cap, budget = cap - n_valid, cell_refs = z.cell_refs, act_cap = z.act_cap, seg_cap = z.seg_cap, chunk_cap = z.chunk_cap;
    SplitArgs a{};
    a.b = b; a.seg_cap = seg_cap; a.cell_refs = cell_refs; a.pad = sp.pad; a.bias = sp.bias; a.cut_bias = sp.cut_bias;
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
    for (int q = 0; q < 3; ++q) { S_TRY(arena.alloc((void **)&a.plan_in[q], sizeof(uint64_t) * (size_t)act_cap)); S_TRY(arena.alloc((void **)&a.plan_ex[q], sizeof(uint64_t) * (size_t)act_cap)); }
    a.cut_cap = budget + 4096u;
    S_TRY(arena.alloc((void **)&a.cut_list, sizeof(uint2) * (size_t)a.cut_cap)); S_TRY(arena.alloc((void **)&a.cut_box, sizeof(float) * 12 * (size_t)a.cut_cap));
    S_TRY(arena.alloc((void **)&a.counters, sizeof(SplitCounters)));
    S_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, a.flags, a.scan, (int)(std::max(cap, b.n) + 4096u), s));      // (also the plan's scans over a level's segments: at most cap + 2)
    S_TRY(arena.alloc(&temp, scan_bytes));

    a.alpha = sp.alpha; a.n_total = n_valid;
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
    h.n_segs = 1u;
    uint32_t seg_cap_now = seg_cap;
    if (const char *e = std::getenv("HRT_SBVH_SEG_CAP")) {      // (tests: the tables-are-full path; 0: the phase gives up at once, the build falls back to PLOC alone)
        const unsigned long v = std::strtoul(e, nullptr, 10);
        if (v == 0) { res.error = hipErrorUnknown; res.where = "split phase: gave up on request (HRT_SBVH_SEG_CAP=0)"; return res; }
        if (v < seg_cap) seg_cap_now = (uint32_t)v;
    }
    a.seg_cap = seg_cap_now;
    res.top_level_begin.push_back(0u);
    for (;;) {
        a.seg_base = h.n_segs; a.cell_base = h.n_cells; a.top_base = h.n_top; a.out_base = h.n_out;
        if (n_big) hipLaunchKernelGGL(k_split_init_bins, dim3(blocks(n_big * (uint32_t)kBinWords, 256)), dim3(256), 0, s, a.bins, n_big * (uint32_t)kBinWords);
        hipLaunchKernelGGL(k_split_bin, dim3(a.n_chunks), dim3(kWave), 0, s, a);
        if (n_big) hipLaunchKernelGGL(k_split_select, dim3(a.n_act), dim3(kWave), 0, s, a);
        hipLaunchKernelGGL(k_split_flags, dim3(a.n_chunks), dim3(kWave), 0, s, a);
        S_TRY(hipcub::DeviceScan::ExclusiveSum(temp, scan_bytes, a.flags, a.scan, (int)src_total, s));
        hipLaunchKernelGGL(k_split_clip, dim3(std::min<uint32_t>(blocks(std::min(src_total, a.cut_cap), kWave), 2048u)), dim3(kWave), 0, s, a);
        for (int pass = 0; pass < 2; ++pass) {
            hipLaunchKernelGGL(k_split_plan_count, dim3(blocks(a.n_act, 256)), dim3(256), 0, s, a);
            for (int q = 0; q < 3; ++q) S_TRY(hipcub::DeviceScan::ExclusiveSum(temp, scan_bytes, a.plan_in[q], a.plan_ex[q], (int)a.n_act, s));
            hipLaunchKernelGGL(k_split_plan_apply, dim3(blocks(a.n_act, 256)), dim3(256), 0, s, a);
            hipLaunchKernelGGL(k_split_scatter, dim3(a.n_chunks), dim3(kWave), 0, s, a);
            S_TRY(hipGetLastError());
            S_TRY(hipMemcpyAsync(&h, a.counters, sizeof h, hipMemcpyDeviceToHost, s));
            S_TRY(hipStreamSynchronize(s));
            if (!h.retry) break;
            if (pass == 1) { res.error = hipErrorUnknown; res.where = "split phase: the plan did not fit twice"; return res; }
            hipLaunchKernelGGL(k_split_force_cells, dim3(blocks(a.n_act, 256)), dim3(256), 0, s, a);      // the segment tables are full: the rest is PLOC's
        }
        if (sp.verbose)
            std::fprintf(stderr, "[hrt] split level %u: %u segments (%u references) -> %u children (%u references), %u cells so far (%u references)%s\n",
                         a.level, a.n_act, src_total, h.n_act, h.src_total, h.n_cells, h.n_out, h.gave_up ? " [segment table full]" : "");
        ++res.levels;
        res.top_level_begin.push_back(h.n_top);
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
