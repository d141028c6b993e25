// refit.hip -- per-frame update of the flattened world-space BVH8 on the device (gfx950).
//
// Replaces the reference's updateIAS (src/Global/RendererImpl.cu:210-242: optixAccelBuild with
// OPTIX_BUILD_OPERATION_UPDATE, called every frame at src/Global/RendererTime.cu:480 after the host
// loop has rewritten the instance transforms).  The tree keeps its topology; k_refit_level
//   * re-derives every world-space primitive record from the object-space source geometry and the
//     instance's new transform (same arithmetic as the host flatten: bvh8_geom.h), and
//   * recomputes each node's origin, exponents and quantised child boxes bottom-up, one level per
//     launch (nodes are stored breadth first, so a level is a contiguous range).
// Eight lanes cooperate on a node, one per child slot; the union box is a 3-step butterfly.
// HBM-bound: per node 80 B read + 80 B written + 24 B box, per primitive 36 B source + 48 B record.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bvh8_geom.h"
#include "device_types.h"

#pragma clang fp contract(off)

namespace hrt {

__device__ __forceinline__ float shfl_xor_f(float v, int m) { return __shfl_xor(v, m, 64); }

// One primitive record: its world-space form from the object-space source and the instance's transform, and (a.rec_box) its padded box.
// Returns the box in lo / hi (empty when the primitive is not finite).
__device__ __forceinline__ void refit_record(const RefitArgs &a, size_t r, float pad, float *lo, float *hi) {
    unsigned char *rec = a.prims + r * a.prim_stride;
    float *rf = reinterpret_cast<float *>(rec);
    const uint32_t *ru = reinterpret_cast<const uint32_t *>(rec);
    const uint32_t prim = ru[3], inst = ru[7], kind = ru[11];
    const float *m = a.inst_xf + 12 * (size_t)inst;
    const bool ident = a.inst_identity[inst] != 0u;
    float plo[3], phi[3];
    if (kind == 0u) {
        const float *src = reinterpret_cast<const float *>(a.inst_src[inst]) + 9 * (size_t)prim;
        float s9[9];
        for (int q = 0; q < 9; ++q) s9[q] = src[q];
        float v0[3], e1[3], e2[3];
        triangle_world(s9, m, ident, v0, e1, e2, plo, phi);
        rf[0] = v0[0]; rf[1] = v0[1]; rf[2] = v0[2];
        rf[4] = e1[0]; rf[5] = e1[1]; rf[6] = e1[2];
        rf[8] = e2[0]; rf[9] = e2[1]; rf[10] = e2[2];
    } else {
        const float c3[3] = {rf[0], rf[1], rf[2]};
        sphere_world_bounds(c3, rf[4], m, ident, plo, phi);
    }
    if (a.clip) {                                      // a reference of a spatial split: the box its cell is responsible for
        const float *cb = a.clip + 6 * r;
        for (int q = 0; q < 3; ++q) { plo[q] = cb[q]; phi[q] = cb[3 + q]; }
    }
    const bool ok = finite_box(plo, phi);
    for (int q = 0; q < 3; ++q) { lo[q] = ok ? plo[q] - pad : INFINITY; hi[q] = ok ? phi[q] + pad : -INFINITY; }
}

// small trees: every record by its own thread, before the levels are walked
__global__ __launch_bounds__(256) void k_refit_records(RefitArgs a) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= a.n_records) return;
    const float pad = a.scale_bits ? 4e-6f * fmaxf(1.0f, __uint_as_float(a.scale_bits[0])) : a.pad;
    float lo[3], hi[3];
    refit_record(a, r, pad, lo, hi);
    float *b = a.rec_box + 6 * (size_t)r;
    for (int q = 0; q < 3; ++q) { b[q] = lo[q]; b[3 + q] = hi[q]; }
}
void launch_refit_records(const RefitArgs &a, hipStream_t s) {
    if (a.rec_box && a.n_records) hipLaunchKernelGGL(k_refit_records, dim3((a.n_records + 255u) / 256u), dim3(256), 0, s, a);
}

// One child slot of one node; the eight lanes of a node call this together (they exchange boxes by
// shuffles), dead groups (live == false) go through the motions without touching memory.
__device__ __forceinline__ void refit_slot(const RefitArgs &a, uint32_t node, uint32_t slot, bool live) {
    unsigned char *nd = a.nodes + (size_t)node * a.node_stride;
    const uint32_t *ndw = reinterpret_cast<const uint32_t *>(nd);

    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t meta = 0;
    const float pad = a.scale_bits ? 4e-6f * fmaxf(1.0f, __uint_as_float(a.scale_bits[0])) : a.pad;
    // A transform node of a two-level tree (bvh8.h: word 3 == 0): its box is the instance's BLAS box under the instance's transform --
    // lane k of the group transforms corner k, the butterfly below joins them, lane 0 cuts the result down to the bounding sphere's box -- and writes what the traversal reads there:
    // the BLAS's root, the identity flag, world -> object.  Cost per instance: this, whatever the BLAS holds.
    const bool xform = live && ndw[3] == 0u;
    if (xform) {
        const uint32_t inst = ndw[5];
        const float *b6 = reinterpret_cast<const float *>(a.inst_src[inst]);
        const float q[3] = {(slot & 1u) ? b6[3] : b6[0], (slot & 2u) ? b6[4] : b6[1], (slot & 4u) ? b6[5] : b6[2]};
        float w[3];
        if (a.inst_identity[inst] != 0u) { w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; } else xf_point(a.inst_xf + 12 * (size_t)inst, q, w);
        const bool ok = b6[0] <= b6[3];
        for (int k = 0; k < 3; ++k) { lo[k] = ok ? w[k] : INFINITY; hi[k] = ok ? w[k] : -INFINITY; }
    } else if (live) {
        const uint32_t imask = ndw[3] >> 24, child_base = ndw[4], prim_base = ndw[5];
        meta = nd[24 + slot];
        if (meta != 0u) {
            if ((imask >> slot) & 1u) {
                const uint32_t c = child_base + (uint32_t)__popc(imask & ((1u << slot) - 1u));
                const float *b = a.node_box + 6 * (size_t)c;
                for (int k = 0; k < 3; ++k) { lo[k] = b[k]; hi[k] = b[3 + k]; }
            } else {
                const uint32_t cbits = meta >> 5, off = meta & 0x1fu;
                const uint32_t cnt = cbits == 1u ? 1u : cbits == 3u ? 2u : 3u;
                for (uint32_t k = 0; k < cnt; ++k) {
                    float plo[3], phi[3];
                    if (a.rec_box) {                                   // (k_refit_records has been here)
                        const float *b = a.rec_box + 6 * (size_t)(prim_base + off + k);
                        for (int q = 0; q < 3; ++q) { plo[q] = b[q]; phi[q] = b[3 + q]; }
                    } else refit_record(a, (size_t)(prim_base + off + k), pad, plo, phi);
                    for (int q = 0; q < 3; ++q) { lo[q] = fminf(lo[q], plo[q]); hi[q] = fmaxf(hi[q], phi[q]); }
                }
            }
        }
    }
    // union over the node's eight slots
    float nlo[3], nhi[3];
    for (int k = 0; k < 3; ++k) {
        float l = lo[k], h = hi[k];
        l = fminf(l, shfl_xor_f(l, 1)); h = fmaxf(h, shfl_xor_f(h, 1));
        l = fminf(l, shfl_xor_f(l, 2)); h = fmaxf(h, shfl_xor_f(h, 2));
        l = fminf(l, shfl_xor_f(l, 4)); h = fmaxf(h, shfl_xor_f(h, 4));
        nlo[k] = l; nhi[k] = h;
    }
    if (!live) return;
    if (xform) {
        if (slot != 0u) return;
        const uint32_t inst = ndw[5];
        {   // ... within the box of the BLAS's bounding sphere (bvh8_geom.h instance_world_bounds)
            const float *b10 = reinterpret_cast<const float *>(a.inst_src[inst]);
            clamp_to_sphere_bounds(b10 + 6, b10[9], a.inst_xf + 12 * (size_t)inst, a.inst_identity[inst] != 0u, nlo, nhi);
        }
        const bool ok = finite_box(nlo, nhi) && nlo[0] <= nhi[0];
        float *b = a.node_box + 6 * (size_t)node;
        for (int k = 0; k < 3; ++k) { b[k] = ok ? nlo[k] - pad : INFINITY; b[3 + k] = ok ? nhi[k] + pad : -INFINITY; }
        uint32_t *w32 = reinterpret_cast<uint32_t *>(nd);
        w32[4] = a.inst_root[inst]; w32[6] = a.inst_identity[inst] != 0u ? 1u : 0u;
        const float *inv = a.inst_inv + 12 * (size_t)inst;
        float *pf = reinterpret_cast<float *>(nd);
        for (int k = 0; k < 12; ++k) pf[8 + k] = inv[k];
        {   // the BLAS's bounding sphere in object space: what a ray is tested against before it goes in (fused.hip)
            const float *b10 = reinterpret_cast<const float *>(a.inst_src[inst]);
            pf[0] = b10[6]; pf[1] = b10[7]; pf[2] = b10[8]; pf[7] = b10[9];
        }
        if (ok) {
            float2 *ref = reinterpret_cast<float2 *>(a.node_ref) + node;
            const float plo[3] = {nlo[0] - pad, nlo[1] - pad, nlo[2] - pad}, phi[3] = {nhi[0] + pad, nhi[1] + pad, nhi[2] + pad};
            const float area = box_half_area(plo, phi);
            if (a.write_reference) ref->y = (area > 0.0f && area < 3.0e38f) ? 1.0f / area : 0.0f;
            else if (a.area_sum) { const float2 r = *ref; const float g = area * r.y; if (r.x > 0.0f && g < 3.0e38f) atomicAdd(a.area_sum, r.x * g); }
        }
        return;
    }
    const bool empty_node = !(nlo[0] <= nhi[0]);
    if (empty_node) for (int k = 0; k < 3; ++k) { nlo[k] = 0.0f; nhi[k] = 0.0f; }
    uint8_t e[3];
    for (int k = 0; k < 3; ++k) e[k] = node_exponent(nhi[k] - nlo[k]);
    // this slot's quantised box
    for (int k = 0; k < 3; ++k) {
        uint8_t ql = 255, qh = 0;
        if (meta != 0u && lo[k] <= hi[k]) quantise_axis(nlo[k], e[k], lo[k], hi[k], &ql, &qh);
        nd[32 + 8 * k + slot] = ql;
        nd[56 + 8 * k + slot] = qh;
    }
    if (slot == 0u) {
        float *pf = reinterpret_cast<float *>(nd);
        pf[0] = nlo[0]; pf[1] = nlo[1]; pf[2] = nlo[2];
        nd[12] = e[0]; nd[13] = e[1]; nd[14] = e[2];
        float *b = a.node_box + 6 * (size_t)node;
        if (empty_node) { for (int k = 0; k < 3; ++k) { b[k] = INFINITY; b[3 + k] = -INFINITY; } }
        else for (int k = 0; k < 3; ++k) { b[k] = nlo[k]; b[3 + k] = nhi[k]; }
    }
    // tree quality: primitives-below weighted mean of area(node now) / area(node as built); 1.0 for the built tree.
    // The refit that completes a build records the built areas instead (write_reference).
    if (slot == 0u && !empty_node) {
        float2 *ref = reinterpret_cast<float2 *>(a.node_ref) + node;
        const float area = box_half_area(nlo, nhi);
        if (a.write_reference) ref->y = (area > 0.0f && area < 3.0e38f) ? 1.0f / area : 0.0f;
        else if (a.area_sum) {
            const float2 r = *ref;
            const float g = area * r.y;
            if (r.x > 0.0f && g < 3.0e38f) atomicAdd(a.area_sum, r.x * g);
        }
    }
}

// node number k of a phase: the trees built breadth first are walked by index ranges, the instanced ones through `order`
__device__ __forceinline__ uint32_t phase_node(const RefitArgs &a, uint32_t first, uint32_t k) {
    return a.order ? a.order[first + k] : first + k;
}

// a wide phase: one launch, eight lanes per node
__global__ __launch_bounds__(256) void k_refit_level(RefitArgs a) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t local = tid >> 3;
    const bool live = local < a.n_nodes;                       // whole 8-lane groups are live or not
    refit_slot(a, phase_node(a, a.first_node, live ? local : 0u), tid & 7u, live);
}

// the narrow phases at the top of the tree (and all of a small tree): one workgroup walks them in order
// with a barrier in between, instead of one launch per phase
__global__ __launch_bounds__(1024) void k_refit_top(RefitArgs a, RefitLevels lv) {
    const uint32_t group = threadIdx.x >> 3, slot = threadIdx.x & 7u;
    for (uint32_t l = 0; l < lv.n_levels; ++l) {
        const uint32_t first = lv.first[l], n = lv.count[l];
        for (uint32_t base = 0; base < n; base += 128u) {
            const uint32_t local = base + group;
            const bool live = local < n;
            refit_slot(a, phase_node(a, first, live ? local : 0u), slot, live);
        }
        __threadfence_block();
        __syncthreads();                                       // the next level reads this level's node boxes
    }
}

// ---- instance tables on the device (asynchronous hrt_tlas_update): what hrt_accel.cpp's instance_tables / invert_affine
// compute on the host, from the instance array where it lies.  One thread per instance. ----
__global__ __launch_bounds__(256) void k_instance_tables(InstanceTableArgs a) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.n) return;
    const float *m = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(a.instances) + 80 * (size_t)i);
    const uint32_t *u = reinterpret_cast<const uint32_t *>(m);
    const uint32_t visibility = u[14] & 1u;
    const unsigned long long handle = *reinterpret_cast<const unsigned long long *>(u + 16);
    if (handle != a.sig_handle[i] || visibility != a.sig_visibility[i]) atomicOr(&a.flags[1], 1u);
    if (u[13] != a.sig_sbt[i]) atomicOr(&a.flags[1], 2u);         // sbtOffset: the next update refreshes the material tables
    float t[12];
    for (int k = 0; k < 12; ++k) { t[k] = m[k]; a.inst_xf[12 * (size_t)i + k] = t[k]; }
    const bool id = t[0] == 1.0f && t[1] == 0.0f && t[2] == 0.0f && t[3] == 0.0f && t[4] == 0.0f && t[5] == 1.0f && t[6] == 0.0f && t[7] == 0.0f &&
                    t[8] == 0.0f && t[9] == 0.0f && t[10] == 1.0f && t[11] == 0.0f &&
                    !(__float_as_uint(t[1]) | __float_as_uint(t[2]) | __float_as_uint(t[3]) | __float_as_uint(t[4]) | __float_as_uint(t[6]) | __float_as_uint(t[7]) |
                      __float_as_uint(t[8]) | __float_as_uint(t[9]) | __float_as_uint(t[11]));      // bitwise, as the host's memcmp (-0 is not identity)
    a.inst_identity[i] = id ? 1u : 0u;
    {   // world -> object: cofactors in double, rounded once (invert_affine of hrt_accel.cpp)
        const double aa = t[0], b = t[1], c = t[2], d = t[4], e = t[5], f = t[6], g = t[8], h = t[9], ii = t[10];
        const double A = e * ii - f * h, B = -(d * ii - f * g), C = d * h - e * g;
        const double det = aa * A + b * B + c * C;
        const double r = 1.0 / det;
        const double n00 = A * r, n01 = -(b * ii - c * h) * r, n02 = (b * f - c * e) * r;
        const double n10 = B * r, n11 = (aa * ii - c * g) * r, n12 = -(aa * f - c * d) * r;
        const double n20 = C * r, n21 = -(aa * h - b * g) * r, n22 = (aa * e - b * d) * r;
        const double tx = t[3], ty = t[7], tz = t[11];
        float *o = a.inst_inv + 12 * (size_t)i;
        o[0] = (float)n00; o[1] = (float)n01; o[2] = (float)n02; o[3] = (float)(-(n00 * tx + n01 * ty + n02 * tz));
        o[4] = (float)n10; o[5] = (float)n11; o[6] = (float)n12; o[7] = (float)(-(n10 * tx + n11 * ty + n12 * tz));
        o[8] = (float)n20; o[9] = (float)n21; o[10] = (float)n22; o[11] = (float)(-(n20 * tx + n21 * ty + n22 * tz));
    }
    // largest |coordinate| of the transformed BLAS box: the scene scale the padding is derived from
    const float *bb = a.blas_box + 6 * (size_t)i;
    if (visibility && bb[0] <= bb[3]) {
        float smax = 1.0f;
        for (int cidx = 0; cidx < 8; ++cidx) {
            const float q[3] = {(cidx & 1) ? bb[3] : bb[0], (cidx & 2) ? bb[4] : bb[1], (cidx & 4) ? bb[5] : bb[2]};
            float w[3];
            if (id) { w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; } else xf_point(t, q, w);
            for (int k = 0; k < 3; ++k) if (fabsf(w[k]) <= 3.0e38f) smax = fmaxf(smax, fabsf(w[k]));
        }
        atomicMax(&a.flags[0], __float_as_uint(smax));         // positive floats order like their bits
    }
}
__global__ void k_update_epilogue(UpdateEpilogueArgs a) {
    if (threadIdx.x != 0u || blockIdx.x != 0u) return;
    *a.h_area = *a.d_area; a.h_flags[0] = a.d_flags[0]; a.h_flags[1] = a.d_flags[1];
    __threadfence_system();
    *a.d_area = 0.0f; a.d_flags[0] = a.flags_init0; a.d_flags[1] = a.flags_init1;
}
void launch_update_epilogue(const UpdateEpilogueArgs &a, hipStream_t s) { hipLaunchKernelGGL(k_update_epilogue, dim3(1), dim3(64), 0, s, a); }

void launch_instance_tables(const InstanceTableArgs &a, hipStream_t s) {
    if (a.n) hipLaunchKernelGGL(k_instance_tables, dim3((a.n + 255u) / 256u), dim3(256), 0, s, a);
}

// ---- two-level trees: a BLAS's object-space tree (topology: the template, built once per BLAS) goes into the TLAS's arrays behind the
// top level -- its node and record numbers move by where it lands, its records name its slot in the table of BLASes the pack's refit
// reads (geometry source, identity transform); everything else is the refit's to compute, as after any build. ----
__global__ __launch_bounds__(256) void k_pack_blas(PackBlasArgs a) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < a.n_nodes) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.src_nodes + 80u * (size_t)i);
        uint32_t *dst = reinterpret_cast<uint32_t *>(a.dst_nodes + (size_t)(a.node_off + i) * a.node_stride);
        for (int q = 0; q < 20; ++q) dst[q] = src[q];
        dst[4] = src[4] + a.node_off; dst[5] = src[5] + a.prim_off;
    }
    if (i < a.n_prims) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.src_prims + 48u * (size_t)i);
        uint32_t *dst = reinterpret_cast<uint32_t *>(a.dst_prims + (size_t)(a.prim_off + i) * a.prim_stride);
        for (int q = 0; q < 12; ++q) dst[q] = src[q];
        dst[7] = a.slot;
    }
}
void launch_pack_blas(const PackBlasArgs &a, hipStream_t s) {
    const uint32_t n = a.n_nodes > a.n_prims ? a.n_nodes : a.n_prims;
    if (n) hipLaunchKernelGGL(k_pack_blas, dim3((n + 255u) / 256u), dim3(256), 0, s, a);
}

void launch_refit_level(const RefitArgs &a, hipStream_t s) {
    if (a.n_nodes == 0) return;
    const uint32_t threads = a.n_nodes * 8u;
    hipLaunchKernelGGL(k_refit_level, dim3((threads + 255u) / 256u), dim3(256), 0, s, a);
}

void launch_refit_top(const RefitArgs &a, const RefitLevels &lv, hipStream_t s) {
    if (lv.n_levels == 0) return;
    hipLaunchKernelGGL(k_refit_top, dim3(1), dim3(1024), 0, s, a, lv);
}

}  // namespace hrt
