// device_types.h -- records and kernel argument blocks shared by kernels.hip and the host API.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hrt {

constexpr float    kFloatZero     = 1e-6f;    // FLOAT_ZERO_VALUE, include/Global/DeviceFunctions.cuh:18
constexpr float    kFloatInfinity = 1e16f;    // FLOAT_INFINITY_VALUE, :19
constexpr uint32_t kRayTraceDepth = 5u;       // rayTraceDepth, include/Global/Shader.cuh:8
constexpr uint32_t kMissPrim      = 0xffffffffu;

constexpr int kProgramSphereRough = 0, kProgramSphereMetal = 1, kProgramTriangleRough = 2, kProgramTriangleMetal = 3;
constexpr uint32_t kNumPrograms = 4;
constexpr uint32_t kNumBins = 1 + kNumPrograms;      // bin 0: path ends; bin 1+P: closest-hit program P

// curandStateXORWOW_t layout (48 B); only d and v[] are live for curand_uniform.
struct RngState {
    uint32_t d, v[5];
    int32_t  boxmuller_flag, boxmuller_flag_double;
    float    boxmuller_extra, _pad;
    double   boxmuller_extra_double;
};
static_assert(sizeof(RngState) == 48, "curandState is 48 bytes");

// One queued ray (32 B, two 16-byte accesses).  o.w = tile-local pixel index (bits),
// d.w = global pixel index y*W+x (bits) = RNG stream index (shader/Shader.cu:97).
struct alignas(16) RayRec { float4 o, d; };

// HitGroupParams as the shader reads it (include/Global/Shader.cuh:43-70), 32 B.
struct alignas(8) HitGroup {
    const void *ptr0;      // sphere.centers | triangles.vertexNormals
    const void *ptr1;      // sphere.radii
    float albedo[3];
    float fuzz;
};
static_assert(sizeof(HitGroup) == 32, "HitGroupParams is 32 bytes");

struct GenerateArgs {
    RayRec *rays;
    const uint32_t *rows;          // tile rows -> frame rows
    uint32_t first_pixel;          // tile-local index of rays[0] (sub-tiles)
    uint32_t n_tile_pixels, width, height;
    float center[3], U[3], V[3], W[3];
};

// one input queue of a traverse launch and where its hit records go
struct TraverseSeg {
    const RayRec *rays;            // NULL: segment unused
    const uint32_t *n_ptr;         // 4 device counters whose sum is the queue length, or NULL
    uint32_t n;                    // used when n_ptr == NULL
    uint32_t any_hit;              // 1: the rays only need hit / no hit (depth >= rayTraceDepth)
    float4 *hit_tuvp;              // t, u, v, primitive index (bits); t = tmax on miss
    uint32_t *hit_inst;            // instance index, kMissPrim on miss
    uint64_t *count_nodes, *count_prims;   // COUNT builds
};

// fused path modes (k_fused, k_traverse<..., FUSED>): what generate / shade / accumulate need, in one launch
struct PathArgs {
    const uint32_t *rows; uint32_t first_pixel, n_tile_pixels, width, height;
    uint32_t spp;                  // samples taken by THIS launch
    uint32_t continue_sum;         // 1: accum already holds the sum of earlier samples of this render
    uint32_t *slice_cost;          // probe launch: per slice of fetch_chunk pixels, the time its pixels' samples took (clock / 16); else NULL
    const uint32_t *slice_order;   // order in which the slices are handed out (most expensive first); NULL: as they lie
    float center[3], U[3], V[3], W[3], bg[3];
    RngState *states;
    const HitGroup *hitgroups; const uint32_t *inst_program;
    float4 *accum;                 // per tile pixel: sum of the samples' linear radiance
    uint64_t *rays_closest, *rays_any;
    const RayRec *trace_rays;      // fused kernel as hrt_trace_rays: the "pixels" are these rays, traced once; results below (else NULL)
    float4 *trace_tuvp; uint32_t *trace_inst; uint32_t trace_any;
    float4 *primary_cache;         // k_fused<.., REUSE>: two float4 per lane of the grid, the primary hit of the lane's pixel (else NULL)
};

struct TraverseArgs {
    const void *nodes;             // Bvh8Node[]
    const void *prims;             // PrimRecord[]
    uint32_t node_stride, prim_stride;   // bytes between consecutive records (80 / 48 when packed)
    TraverseSeg seg[2];
    uint32_t *fetch_counter;       // 8 slice counters on 128-byte lines, zeroed before the launch
    const float *inst_inv;         // 12 floats per instance: world->object (spheres)
    const uint32_t *inst_identity;
    float tmin, tmax;
    int refill_threshold;          // refill when at least this many lanes are idle
    uint32_t fetch_chunk;          // rays per slice a wave takes from the queue
    int tail_split;                // split long rays across idle lanes once the queue is drained
    int postpone_pct;              // the leaf pass is skipped while fewer than this % of the alive lanes have leaf work and none needs it
    int leaf_quorum;               // k_fused: ... and fewer than this many lanes have nothing but leaf work (they wait; >= 1)
    int leaf_hold;                 // k_fused: a lane with this many leaf groups queued takes no new node until some are tested (2 .. 4 = the leaf stack's depth)
    int tail_regen;                // k_fused, tile used up: regenerate once this many finished rays wait (>= 1)
    PathArgs path;                 // FUSED only
};

struct BinArgs {
    const uint32_t *n_rays_ptr; uint32_t n_rays;
    const uint32_t *hit_inst;
    const uint32_t *inst_program;
    uint32_t depth;
    uint32_t *bin_count;           // kNumBins counters of this stage (zeroed)
    uint32_t *bin_items;           // kNumBins arrays of bin_stride ray indices
    uint32_t bin_stride;
    uint64_t *total_rays;
};

struct ShadeArgs {
    const uint32_t *bin_count; const uint32_t *bin_items; uint32_t bin_stride;
    const RayRec *rays_in; RayRec *rays_out;
    const float4 *hit_tuvp; const uint32_t *hit_inst;
    const HitGroup *hitgroups;
    RngState *states;
    uint32_t *chain;               // 4 instance indices per tile pixel
    uint32_t depth;
};

struct AccumArgs {
    const uint32_t *bin_count; const uint32_t *bin_items;
    const RayRec *rays_in; const uint32_t *hit_inst;
    const HitGroup *hitgroups;
    const uint32_t *chain;
    float4 *result;                // per tile pixel: linear radiance of this sample
    float bg[3];
    uint32_t depth;
};

struct FinalizeArgs {
    const float4 *accum; const uint32_t *rows;
    uint32_t n_tile_pixels, width, spp;
    float4 *color, *albedo, *normal, *linear;
    uint32_t *reset_counters; uint32_t n_reset;      // when set: n_reset words (< the launch's threads) zeroed on the way
};

// one level of the bottom-up refit (refit.hip)
struct RefitArgs {
    unsigned char *nodes; uint32_t node_stride;
    unsigned char *prims; uint32_t prim_stride;
    float *node_box;               // 6 floats per node: padded bounds of everything below it
    uint32_t first_node, n_nodes;  // the phase: nodes first_node .. first_node + n_nodes (positions in `order` when it is set)
    const uint32_t *order;         // NULL: node index = position (trees stored breadth first)
    const float *inst_xf;          // 12 floats per instance: object -> world
    const uint32_t *inst_identity;
    const void *const *inst_src;   // per instance: object-space triangle vertices (9 floats per triangle) of its BLAS; the top level of a
                                   // two-level tree: the BLAS's object-space box (6 floats)
    const float *inst_inv;         // two-level trees: 12 floats per instance, world -> object (written into the transform nodes)
    const uint32_t *inst_root;     // two-level trees: per instance, the node index of its BLAS's root
    float pad;
    const uint32_t *scale_bits;    // when set: pad = 4e-6 * max(1, scene scale) with the scale read from here (float bits; written by
                                   // k_instance_tables on the same stream), instead of the host's `pad`
    float *node_ref;               // 2 floats per node: {weight, 1 / half area as built}
    uint32_t write_reference;      // 1: this refit completes a build -- record the areas instead of comparing with them
    float *area_sum;               // weighted mean of area now / area as built (quality after the refit), may be NULL
    float *rec_box; uint32_t n_records;   // when set (small trees): 6 floats per record, the record's padded box, written by k_refit_records before the levels
                                   // are walked -- the records' arithmetic then runs one thread per record instead of inside the walk up the tree
    const float *clip;             // when set (the refit that completes a device build with spatial splits): 6 floats per record, the box of the part
                                   // of the primitive this record stands for, taken instead of the primitive's own
};
constexpr uint32_t kRefitTopLevels = 16, kRefitTopLevelNodes = 128;      // (one pass of the 1024-thread workgroup per level; wider levels are quicker as launches of their own: 95 -> ~45 us for the reference's sample)
struct RefitLevels { uint32_t n_levels; uint32_t first[kRefitTopLevels], count[kRefitTopLevels]; };   // phases in processing order, each at most kRefitTopLevelNodes wide
// two-level trees: copy of one BLAS's template tree (packed: 80-byte nodes, 48-byte records) into a TLAS's arrays
struct PackBlasArgs {
    const unsigned char *src_nodes, *src_prims; uint32_t n_nodes, n_prims;
    unsigned char *dst_nodes, *dst_prims; uint32_t node_stride, prim_stride;
    uint32_t node_off, prim_off;   // where the tree lands: index of its root in the TLAS's node array, of its first record
    uint32_t slot;                 // the BLAS's number in the tables of the pack's refit (written to the records' instance field)
};
void launch_pack_blas(const PackBlasArgs &a, hipStream_t s);
void launch_refit_level(const RefitArgs &a, hipStream_t s);
void launch_refit_records(const RefitArgs &a, hipStream_t s);
// per-instance tables of an update derived on the device from the caller's instance array (asynchronous updates)
struct InstanceTableArgs {
    const void *instances;         // HrtInstance[] (80 B each): transform[12], instanceId, sbtOffset, visibilityMask, flags, handle (u64), pad
    uint32_t n;
    const unsigned long long *sig_handle; const uint32_t *sig_visibility;   // what the tree was built with
    const uint32_t *sig_sbt;       // the sbtOffsets the material tables were derived from
    const float *blas_box;         // 6 floats per instance: object-space bounds of its BLAS (lo > hi: empty)
    float *inst_xf, *inst_inv; uint32_t *inst_identity;
    uint32_t *flags;               // [0]: scene scale (float bits, starts at 1.0f); [1]: bit 0 set when a handle or visibility bit differs (-> rebuild), bit 1 when an sbtOffset does (-> synchronous update)
};
void launch_instance_tables(const InstanceTableArgs &a, hipStream_t s);
// the end of an asynchronous update, one launch instead of two copies to the host, one from it and a fill: the refit's area sum and the
// tables kernel's verdict go to pinned host memory, the device words are set up for the next update
struct UpdateEpilogueArgs { float *d_area; uint32_t *d_flags; float *h_area; uint32_t *h_flags; uint32_t flags_init0, flags_init1; };
void launch_update_epilogue(const UpdateEpilogueArgs &a, hipStream_t s);
void launch_refit_top(const RefitArgs &a, const RefitLevels &lv, hipStream_t s);

// particle pose update (pose.hip)
struct PoseArgs {
    void *instances;               // HrtInstance[] (80 B each, transform first)
    uint32_t first_instance, n;
    const float4 *current, *next;  // HrtParticleState[] as 3 float4 each
    float duration; uint32_t frame, frame_count;
    float offset[3], scale[3];
    uint32_t mesh_mode;            // RendererMesh's drift-only update (no rotation, position not added)
};
void launch_pose_instances(const PoseArgs &a, hipStream_t s);
void launch_debug_trig(int which, const float *a, const float *b, uint32_t first, uint32_t stride, uint64_t n, int force_slow, float *out, hipStream_t s);

// host-callable launchers (defined in kernels.hip)
void launch_rng_init(RngState *states, uint32_t n, uint64_t salt, const uint32_t *d_jump, hipStream_t s);
void launch_generate(const GenerateArgs &a, hipStream_t s);
void launch_traverse(const TraverseArgs &a, bool count, bool has_spheres, bool dma, uint32_t grid_blocks, hipStream_t s);
void launch_paths_v1(const TraverseArgs &a, bool has_spheres, uint32_t grid_blocks, hipStream_t s);   // round 1's fused kernel k_traverse<.., FUSED> (HRT_FUSED=2)
constexpr int kFusedBlocksPerCu = 16;  // k_fused is compiled for 4 waves per SIMD (125 VGPRs, nothing spilled): more workgroups per CU would only queue
constexpr int kFusedInstancedBlocksPerCu = 12;   // k_fused<.., INSTANCED> is compiled for 3 waves per SIMD (fused.hip)
constexpr uint32_t kFetchShards = 8;         // slice counters (one per XCD-group of blocks)
constexpr uint32_t kFetchShardStride = 32;   // u32s between counters: one 128-byte line each
constexpr int kFusedMaxDepth = 12;     // deepest tree (levels below the root) k_fused takes: its per-lane node stack in LDS (trav_lean.h: kNodeStackLds)
void launch_fused(const TraverseArgs &a, bool has_spheres, uint32_t grid_blocks, hipStream_t s);
void launch_fused_instanced(const TraverseArgs &a, bool has_spheres, uint32_t grid_blocks, hipStream_t s);      // two-level trees (transform nodes, bvh8.h)
void launch_trace_queue(const TraverseArgs &a, bool has_spheres, uint32_t grid_blocks, hipStream_t s);   // k_trace_queue (fused_queue.hip): wavefront mode's traverse kernel, the loop of k_fused over ray queues      // k_fused (fused.hip): the default
void launch_sum(float4 *accum, const float4 *result, uint32_t n, uint32_t first_sample, hipStream_t s);
void launch_bin(const BinArgs &a, uint32_t grid_blocks, hipStream_t s);
void launch_shade(const ShadeArgs &a, int program, uint32_t grid_blocks, hipStream_t s);
void launch_accumulate(const AccumArgs &a, uint32_t grid_blocks, hipStream_t s);
void launch_finalize(const FinalizeArgs &a, hipStream_t s);
void launch_to_rgba8(const float4 *src, uchar4 *dst, uint32_t n, hipStream_t s);
void launch_color_to_float4(const float4 *src, float4 *dst, uint32_t n, hipStream_t s);
void launch_pack_rays(const float *o, const float *d, uint32_t n, RayRec *rays, hipStream_t s);
void launch_unpack_hits(const float4 *tuvp, const uint32_t *inst, uint32_t n, float *t, float *u, float *v,
                        uint32_t *prim, uint32_t *oinst, hipStream_t s);

}  // namespace hrt
