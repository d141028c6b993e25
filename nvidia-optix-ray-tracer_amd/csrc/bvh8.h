// bvh8.h -- packed BVH8 layout shared by the host builder and the HIP traversal kernels.
//
// Replaces what optixAccelBuild produces for the reference (src/Global/RendererImpl.cu:30-88,
// 174-208): an opaque RT-core BVH.  Here the structure is explicit:
//
//   Bvh8Node (80 B, five 16-byte loads per visit), after Ylitie/Karras/Laine 2017
//   ("Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs"):
//     word 0-2  origin p (float3)            child boxes are p + q * 2^e per axis
//     word 3    ex | ey<<8 | ez<<16 | imask<<24     (e biased by 127, imask: bit s = slot s is an inner node)
//     word 4    child_base   (index of the first inner child; inner children are contiguous, slot order)
//     word 5    prim_base    (index of the first primitive record of this node's leaves)
//     word 6-7  meta[8]      inner: 0b001_11000 | slot; leaf: unary count (1,3,7)<<5 | prim offset; empty: 0
//     word 8-9  qlo_x[8]   word 10-11 qlo_y[8]   word 12-13 qlo_z[8]
//     word 14-15 qhi_x[8]  word 16-17 qhi_y[8]   word 18-19 qhi_z[8]
//
//   Transform node of a TWO-LEVEL tree (same 80 B, in the child block of its parent like any inner child): word 3 == 0 marks it
//   (a box node's exponents are never 0).  The instance's BLAS -- an object-space BVH8 shared by every instance of it -- hangs below:
//     word 0-2  centre of the BLAS's bounding sphere, object space     (a ray that misses the sphere does not go in: half of those that
//     word 7    its radius (negative: none)                             cross the instance's box in the top level do, fused.hip)
//     word 4    index of the BLAS's root node (in the same node array)
//     word 5    instance index (= what a hit reports, and the SBT offset's key)
//     word 6    1 when the transform is the identity (the ray is then copied, not multiplied)
//     word 8-19 world -> object, 3x4 row major (inverse by cofactors in double, rounded once: hrt_accel.cpp invert_affine)
//   This is the reference's IAS over shared GASes (src/Global/RendererImpl.cu:174-206, GAS chosen by shapeID at
//   src/Global/RendererTime.cu:116-130): memory and update cost grow with instances + unique primitives.
//
//   PrimRecord (48 B, three 16-byte loads per test), world space (object space inside a BLAS of a two-level tree), leaf order:
//     triangle: {v0.xyz, prim} {e1.xyz, inst} {e2.xyz, 0}        e1 = v1 - v0, e2 = v2 - v0 (float)
//     sphere:   {c.xyz,  prim} {r, 0, 0, inst} {0, 0, 0, 1}      object-space centre/radius
#pragma once
#include <cstdint>
#include <vector>

namespace hrt {

struct alignas(16) Bvh8Node {
    float    p[3];
    uint8_t  e[3];
    uint8_t  imask;
    uint32_t child_base;
    uint32_t prim_base;
    uint8_t  meta[8];
    uint8_t  qlo[3][8];
    uint8_t  qhi[3][8];
};
static_assert(sizeof(Bvh8Node) == 80, "packed BVH8 node is 80 bytes");

struct alignas(16) PrimRecord {
    float    a[3]; uint32_t prim;
    float    b[3]; uint32_t inst;
    float    c[3]; uint32_t kind;      // 0 triangle, 1 sphere
};
static_assert(sizeof(PrimRecord) == 48, "primitive record is 48 bytes");

constexpr uint32_t kMaxLeafPrims = 3;      // unary count in 3 bits
constexpr uint32_t kPrimKindTriangle = 0, kPrimKindSphere = 1;
constexpr uint32_t kPrimKindInstance = 2;      // builder input only: the "primitive" is an instance's BLAS box (the top level of a two-level tree)

// Input primitive for the builder: a record plus its (unpadded) world-space bounds.
struct BuildPrim {
    PrimRecord rec;
    float lo[3], hi[3];
};

struct Bvh8 {
    std::vector<Bvh8Node>   nodes;
    std::vector<PrimRecord> prims;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    uint32_t n_triangles = 0, n_spheres = 0;
    uint32_t max_depth = 0;
    std::vector<float>      prim_bounds;   // 6 floats per primitive, leaf order (host-side checks only)
    // refit support: nodes are emitted breadth first, so level l is nodes [level_begin[l], level_begin[l+1])
    std::vector<uint32_t>   level_begin;
    std::vector<float>      node_box;      // 6 floats per node: union of the padded primitive boxes below it
    float pad = 0.0f;                      // what was added around every primitive box
    std::vector<float>      node_ref;      // 2 floats per node: {primitives-below weight (sums to 1), 1 / half area as built}
};

// Deterministic host build (binned SAH BVH2 -> greedy collapse to 8-wide -> octant slot
// assignment -> outward-rounded 8-bit quantisation).  threads <= 0: hardware concurrency.
// scene_scale > 0 overrides the largest |coordinate| the padding is derived from; max_leaf_prims (1..3) caps the
// primitives per leaf slot (1 for the tree over instances, whose "primitives" are whole subtrees).
// spatial_splits: SBVH -- a primitive may then be referenced from several leaves (one record each), each answering for the
// part of it inside its leaf's box.
void build_bvh8(const std::vector<BuildPrim> &prims, Bvh8 &out, int threads = 0, float scene_scale = 0.0f,
                uint32_t max_leaf_prims = kMaxLeafPrims, bool spatial_splits = false);

// A tree over instances: a top tree built over the instances' world boxes, whose leaf slots point at per-instance
// copies of object-space template trees (one template per BLAS, shared by all its instances).  Only the TOPOLOGY is
// produced here (child / primitive bases, masks, meta bytes, primitive ids): every box, origin, exponent and
// world-space record is computed by the device refit (refit.hip), which walks the nodes bottom-up in the order
// given by `order` / `phase_begin` (phase h = nodes of height h; children always have a smaller height).
struct InstancedTree {
    std::vector<Bvh8Node>   nodes;
    std::vector<PrimRecord> prims;
    std::vector<uint32_t>   order;         // node indices sorted by height
    std::vector<uint32_t>   phase_begin;   // phase h = order[phase_begin[h] .. phase_begin[h+1])
    std::vector<float>      weight;        // per node: primitives below it / sum over nodes (refit quality weights)
    uint32_t max_depth = 0;
    uint32_t n_triangles = 0, n_spheres = 0;
};
// tmpl[i]: template of instance i (NULL: the instance contributes nothing); box: 6 floats per instance (world bounds,
// used for the top tree's topology only).
void assemble_instanced_bvh8(const std::vector<const Bvh8 *> &tmpl, const std::vector<float> &box, InstancedTree &out);

// Structural self-check used by the CPU tests: every primitive's bounds lie inside the
// dequantised box of every ancestor slot.  Returns an empty string when consistent.
const char *validate_bvh8(const Bvh8 &bvh);

}  // namespace hrt
