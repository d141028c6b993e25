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
    BuildCounters *counters;
    float4 *pb_lo, *pb_hi;            // per primitive: world bounds (lo.w = valid)
    uint64_t *keys; uint32_t *vals, *vals_sorted;
    float4 *node_lo, *node_hi;        // BVH2: lo.xyz | left, hi.xyz | right (leaf: left = ~0, right = primitive)
    uint32_t *node_parent, *node_nprims, *node_visit;
    uint32_t *cl_a, *nn; uint64_t *flags, *scan;
    float *cost;                      // 8 floats per BVH2 node
    unsigned char *out_nodes; uint32_t node_stride; unsigned char *out_prims; uint32_t prim_stride; float *out_node_ref;
};

struct GpuBuildInput {
    uint32_t n_prims, n_inst;
    const uint32_t *d_inst_first, *d_inst_kind; const void *const *d_inst_src;
    const float *d_inst_xf; const uint32_t *d_inst_identity;
    uint32_t max_leaf_prims; float c_node, c_prim; int ploc_radius;
    unsigned char *out_nodes; uint32_t node_stride;      // room for n_prims nodes (worst case)
    unsigned char *out_prims; uint32_t prim_stride;      // room for n_prims records
    float *out_node_ref;                                 // 2 floats per node
    void *scratch = nullptr; size_t scratch_bytes = 0;   // optional working memory (gpu_build_scratch_bytes): what does not fit is hipMalloc'ed
};

struct GpuBuildResult {
    hipError_t error = hipSuccess; const char *where = "";
    uint32_t n_nodes = 0, n_prims = 0, max_depth = 0, ploc_rounds = 0;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    std::vector<uint32_t> level_begin;                   // nodes of level l: [level_begin[l], level_begin[l + 1])
};

GpuBuildResult gpu_build_bvh8(const GpuBuildInput &in, hipStream_t s);
size_t gpu_build_scratch_bytes(uint32_t n_prims);     // working memory of a build of n_prims primitives (an upper estimate)
constexpr size_t kBoundsScratchBytes = 256;           // ... of gpu_blas_bounds
hipError_t gpu_blas_bounds(const float *d_src, uint32_t n_prims, uint32_t kind, float *lo, float *hi, void *d_scratch, hipStream_t s);
void launch_pack_spheres(const float *centers, const float *radii, uint32_t n, float *out, hipStream_t s);

}  // namespace hrt
