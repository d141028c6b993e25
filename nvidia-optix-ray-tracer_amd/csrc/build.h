// build.h -- interface of the device-side acceleration-structure build (build.hip), used by hrt_accel.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace hrt {

// counters and reductions of one build, in device memory (read back a few bytes at a time)
struct BuildCounters {
    uint32_t cmin[3], cmax[3];        // centroid bounds, floats in order-preserving unsigned form
    uint32_t bmin[3], bmax[3];        // bounds of all valid primitives, same form
    uint32_t n_invalid;               // primitives with non-finite bounds (left out of the tree)
    uint32_t m_next, merges;          // PLOC round: clusters after the round, merges made
    uint32_t m_cur, node_base;        // PLOC: clusters before the round, index of the next BVH2 node (advanced on the device between rounds)
    uint32_t root, small_rounds, cl_in_b;   // k_ploc_small: the root cluster, the rounds it ran, which cluster array holds the result
    uint32_t n_levels, max_depth, level_begin[40];      // k_emit_small: the levels of a small tree (nodes of level l: [level_begin[l], level_begin[l + 1]))
    uint32_t next_node, next_prim;    // emission cursors
    float total_below;                // sum over BVH8 nodes of the primitives below them (refit quality weights)
};

struct GpuBuildArgs {
    uint32_t n, n_inst;
    const uint32_t *inst_first;       // n_inst + 1: global number of an instance's first primitive (invisible instances: none)
    const uint32_t *inst_kind;        // kPrimKindTriangle / kPrimKindSphere
    const void *const *inst_src;      // per instance: 9 floats per triangle, or {cx, cy, cz, r} per sphere (object space)
    const float *inst_xf; const uint32_t *inst_identity;
    uint32_t max_leaf_prims; float c_node, c_prim; int ploc_radius;
    float quant_guard;                // emission: a node whose children's stored (8-bit) boxes would have more than this times their true area keeps its two BVH2 children (0: off)
    uint32_t width;                   // children per node at most (8; HRT_BVH_WIDTH: fewer, to measure what a narrower node would cost in visits)
    uint32_t balanced;                // 1: PLOC pairs every cluster with its neighbour in Morton order (position ^ 1) whatever the areas: a tree of log2(n) levels
                                      // for scenes whose nearest-neighbour tree is deeper than the traversal stacks (GpuBuildInput::balanced)
    uint32_t instance_leaves;         // 1: the top level of a two-level tree -- every primitive is an instance (kPrimKindInstance) and is emitted as a
                                      // TRANSFORM NODE in its parent's child block (bvh8.h), not as a record
    BuildCounters *counters;
    float4 *pb_lo, *pb_hi;            // per primitive: world bounds (lo.w = valid)
    uint64_t *keys; uint32_t *vals, *vals_sorted;
    float4 *node_lo, *node_hi;        // BVH2: lo.xyz | left, hi.xyz | right (leaf: left = ~0, right = primitive)
    uint32_t *node_nprims;
    uint32_t *cl_a, *nn; uint64_t *flags, *scan;
    float *cost;                      // 8 floats per BVH2 node
    // builds with spatial splits (build_split.hip): the leaves are REFERENCES -- a primitive and the box of the part of it a cell is responsible for
    const float4 *ref_lo, *ref_hi;    // per reference: lo.xyz | primitive, hi.xyz | cell
    uint32_t *node_cell;              // per BVH2 node: the cell PLOC keeps it in (NULL: one cell)
    float *out_clip;                  // per emitted record: 6 floats, the box the refit takes instead of the primitive's own (NULL: none)
    unsigned char *out_nodes; uint32_t node_stride; unsigned char *out_prims; uint32_t prim_stride; float *out_node_ref;
};

// the top-down phase of a build with spatial splits (build_split.hip)
struct SplitParams {
    bool enabled = false;
    float budget_frac = 1.0f;         // at most this many extra references per primitive
    float alpha = 1e-5f, bias = 0.95f;       // as the host builder's (bvh8_build.cpp)
    float cut_bias = 1.0f;            // a straddler is cut when cut_bias x the SAH cost of cutting it is not above that of keeping it whole on either side
    uint32_t cell_refs = 256;         // segments with fewer references are left to PLOC
    float pad = 0.0f;                 // the padding the SAH areas are computed with (4e-6 of the scene scale)
    bool verbose = false;
};

// one segment of the top-down phase: a range of references of the level's source buffer
struct SplitSeg {
    uint32_t first, count, budget, kind;       // kind: 0 pending, 1 cell, 2 object split, 3 spatial split
    uint32_t nb[6], cb[6];                     // bounds and centroid bounds, lo[3] as f2ord(x), hi[3] as f2ord(-x): both shrink under atomicMin
    uint32_t axis, bin; float c0, scale;       // object split: centroid bins <= bin go left (bin = (int)((c - c0) * scale));  spatial: c0 = the plane
    uint32_t out_first, child, index, level;   // where the references go; first of the two child segments; top-node / cell number
    uint32_t bins_slot, pad_;                  // segments of more than one chunk: where their bins are flushed to
    uint32_t nl, nr; float sl[6], sr[6];       // spatial split: the children as binned (every straddler cut): the unsplitting test's B1, B2, N1, N2
};

struct SplitPhaseResult {
    hipError_t error = hipSuccess; const char *where = "";
    uint32_t n_refs = 0, n_cells = 0, n_top = 0, levels = 0;
    float4 *ref_lo = nullptr, *ref_hi = nullptr;         // n_refs: lo.xyz | primitive, hi.xyz | cell (references of a cell are contiguous)
    SplitSeg *segs = nullptr; uint32_t *top_seg = nullptr, *cell_seg = nullptr;     // top node t is segment top_seg[t]; cell c is segment cell_seg[c]
    std::vector<uint32_t> top_level_begin;                // top nodes of level l: [top_level_begin[l], top_level_begin[l + 1])
};

// working memory of a build: the caller's arena while it lasts, hipMalloc beyond
struct BuildArena {
    void *base = nullptr; size_t bytes = 0, used = 0;
    std::vector<void *> owned;
    hipError_t alloc(void **p, size_t n);
    ~BuildArena();
};

struct GpuBuildInput {
    uint32_t n_prims, n_inst;
    const uint32_t *d_inst_first, *d_inst_kind; const void *const *d_inst_src;
    const float *d_inst_xf; const uint32_t *d_inst_identity;
    uint32_t max_leaf_prims; float c_node, c_prim; int ploc_radius;
    float quant_guard = 1.25f;
    uint32_t width = 8;                                  // children per node at most
    bool instance_leaves = false;                        // the top level of a two-level tree (GpuBuildArgs); max_leaf_prims must be 1
    bool balanced = false;                               // no top-down phase, PLOC by position: the fallback for a tree too deep to traverse (a chain of
                                                         // primitives over many orders of magnitude: SAH peels one off the rest at every level)
    unsigned char *out_nodes; uint32_t node_stride;      // room for n_prims nodes (worst case; instance_leaves: 2 * n_prims + 1)
    unsigned char *out_prims; uint32_t prim_stride;      // room for n_prims records
    float *out_node_ref;                                 // 2 floats per node
    void *scratch = nullptr; size_t scratch_bytes = 0;   // optional working memory (gpu_build_scratch_bytes): what does not fit is hipMalloc'ed
    // spatial splits (HRT_CTX_FAST_TRACE on the device): out_nodes / out_node_ref / out_prims / out_clip then have room for
    // gpu_build_max_refs(n_prims, split) entries
    SplitParams split;
    float *out_clip = nullptr;                           // 6 floats per record (split builds)
};

struct GpuBuildResult {
    hipError_t error = hipSuccess; const char *where = "";
    uint32_t n_nodes = 0, n_prims = 0, max_depth = 0, ploc_rounds = 0;      // n_prims: valid primitives
    uint32_t n_records = 0;                              // primitive records emitted (= n_prims without spatial splits)
    uint32_t split_levels = 0, n_cells = 0;
    bool fell_back = false;                              // the top-down phase gave up and PLOC alone built the tree
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    std::vector<uint32_t> level_begin;                   // nodes of level l: [level_begin[l], level_begin[l + 1])
};

GpuBuildResult gpu_build_bvh8(const GpuBuildInput &in, hipStream_t s);
size_t gpu_build_scratch_bytes(uint32_t n_prims, const SplitParams *split = nullptr);     // working memory of a build of n_prims primitives (an upper estimate)
size_t gpu_split_table_bytes(uint32_t n_prims, const SplitParams &split);                   // segment tables, bins and work lists of the top-down phase
uint32_t gpu_build_max_refs(uint32_t n_prims, const SplitParams *split);                    // most records / nodes a build can emit
// build_split.hip: references of the valid primitives (pb_lo / pb_hi of `a`, scene bounds in a.counters) cut top-down into cells
SplitPhaseResult gpu_split_phase(const GpuBuildArgs &a, uint32_t n_valid, const SplitParams &sp, BuildArena &arena, hipStream_t s);
constexpr size_t kBoundsScratchBytes = 512;           // ... of gpu_blas_bounds
hipError_t gpu_blas_bounds(const float *d_src, uint32_t n_prims, uint32_t kind, float *lo, float *hi, void *d_scratch, hipStream_t s);
// bounding sphere of a BLAS around `center` (radius < 0: none)
hipError_t gpu_blas_radius(const float *d_src, uint32_t n_prims, uint32_t kind, const float *center, float *radius, void *d_scratch, hipStream_t s);
void launch_pack_spheres(const float *centers, const float *radii, uint32_t n, float *out, hipStream_t s);

}  // namespace hrt
