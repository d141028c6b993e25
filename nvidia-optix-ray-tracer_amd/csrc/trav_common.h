// trav_common.h -- device code shared by the kernels of kernels.hip (wavefront pipeline, counting build, round-1 fused
// kernel), fused.hip and fused_queue.hip (the production path kernel and its queue form): vector helpers, XORWOW, the shading arithmetic of shader/Shader.cu,
// the canonical primitive test and the load / wait primitives of the traversal step.  One definition each, hence one
// rounding behaviour everywhere.  Device-only: include from .hip files compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_types.h"

#pragma clang fp contract(off)

namespace hrt {


// ---------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_prefix(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t wave_first_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 add3(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 muls3(V3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 divs3(V3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ V3 neg3(V3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float len2_3(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// normalize(), include/Global/DeviceFunctions.cuh:397-404; rsqrtf pinned as 1/sqrtf
__device__ __forceinline__ V3 normalize3(V3 a) {
    const float len2 = len2_3(a);
    if (len2 <= kFloatZero * kFloatZero) return mk3(0.0f, 0.0f, 1.0f);
    const float invLen = 1.0f / sqrtf(len2);
    return muls3(a, invLen);
}
__device__ __forceinline__ bool finite3(V3 a) { return isfinite(a.x) && isfinite(a.y) && isfinite(a.z); }

// ---------------------------------------------------------------------------------------
// XORWOW (cuRAND curandStateXORWOW_t)
// ---------------------------------------------------------------------------------------
struct Xorwow { uint32_t d, v0, v1, v2, v3, v4; };

__device__ __forceinline__ uint32_t xorwow_next(Xorwow &s) {
    const uint32_t t = s.v0 ^ (s.v0 >> 2);
    s.v0 = s.v1; s.v1 = s.v2; s.v2 = s.v3; s.v3 = s.v4;
    s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v4 + s.d;
}
// curand_uniform: (0, 1]
__device__ __forceinline__ float xorwow_uniform(Xorwow &s) {
    return (float)xorwow_next(s) * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}
__device__ __forceinline__ Xorwow rng_load(const RngState *st) {
    const uint2 *p = reinterpret_cast<const uint2 *>(st);
    const uint2 a = p[0], b = p[1], c = p[2];
    Xorwow s; s.d = a.x; s.v0 = a.y; s.v1 = b.x; s.v2 = b.y; s.v3 = c.x; s.v4 = c.y;
    return s;
}
__device__ __forceinline__ void rng_store(RngState *st, const Xorwow &s) {
    uint2 *p = reinterpret_cast<uint2 *>(st);
    p[0] = make_uint2(s.d, s.v0); p[1] = make_uint2(s.v1, s.v2); p[2] = make_uint2(s.v3, s.v4);
}

// ---------------------------------------------------------------------------------------
// shading arithmetic shared by the wavefront kernels (k_generate / k_shade / k_accumulate) and the
// fused path mode of k_traverse: one definition, hence one rounding behaviour
// ---------------------------------------------------------------------------------------
// pinhole ray through the centre of pixel (ix, iy): shader/Shader.cu:249-261
__device__ __forceinline__ V3 primary_direction(uint32_t ix, uint32_t iy, uint32_t width, uint32_t height,
                                                const float *Uc, const float *Vc, const float *Wc) {
    const float ndcx = (((float)ix + 0.5f) / (float)width) * 2.0f - 1.0f;          // :250
    const float ndcy = (((float)iy + 0.5f) / (float)height) * 2.0f - 1.0f;         // :251
    const V3 U = mk3(Uc[0], Uc[1], Uc[2]), V = mk3(Vc[0], Vc[1], Vc[2]), W = mk3(Wc[0], Wc[1], Wc[2]);
    const float aspect = (float)width / (float)height;                            // :260
    return normalize3(add3(add3(muls3(U, ndcx * aspect), muls3(V, ndcy)), W));    // :261
}

// randomSpaceVector, include/Global/DeviceFunctions.cuh:570-582 (length = 1)
__device__ __forceinline__ V3 random_space_vector(Xorwow &rng) {
    V3 ret; float lengthSquare;
    do {
        ret.x = -1.0f + 2.0f * xorwow_uniform(rng);      // randomDouble(state, -1, 1) :220-222
        ret.y = -1.0f + 2.0f * xorwow_uniform(rng);
        ret.z = -1.0f + 2.0f * xorwow_uniform(rng);
        lengthSquare = len2_3(ret);
    } while (lengthSquare < kFloatZero * kFloatZero);
    ret = normalize3(ret);
    return muls3(ret, 1.0f);
}

// closesthitImpl for one (geometry, material) program, shader/Shader.cu:111-213: hit point and the
// direction of the next ray.  rng is touched only when the program draws (rough, or metal with fuzz > 0).
// hit point of a ray, shader/Shader.cu:111-114
__device__ __forceinline__ V3 hit_point(V3 rayOrigin, V3 rayDirection, float t) { return add3(rayOrigin, muls3(rayDirection, t)); }

template <bool kSphere, bool kRough>
__device__ __forceinline__ void scatter_at(const HitGroup &hg, V3 hitPoint, V3 rayDirection, float u, float v,
                                           uint32_t primitiveIndex, Xorwow &rng, V3 &reflectDirection) {
    V3 normalVector;
    if (kSphere) {                                                        // :122-136
        const float *cp = reinterpret_cast<const float *>(hg.ptr0) + 3 * (size_t)primitiveIndex;
        const V3 sphereCenter = mk3(cp[0], cp[1], cp[2]);
        const float sphereRadius = reinterpret_cast<const float *>(hg.ptr1)[primitiveIndex];
        const V3 outwardNormal = divs3(sub3(hitPoint, sphereCenter), sphereRadius);
        const bool hitFrontFace = dot3(rayDirection, outwardNormal) < 0.0f;
        normalVector = hitFrontFace ? outwardNormal : neg3(outwardNormal);
    } else {                                                              // :137-155
        const float *np = reinterpret_cast<const float *>(hg.ptr0) + 9 * (size_t)primitiveIndex;
        const V3 n1 = mk3(np[0], np[1], np[2]), n2 = mk3(np[3], np[4], np[5]), n3 = mk3(np[6], np[7], np[8]);
        const float w = 1.0f - u - v;
        const V3 _normal = add3(add3(muls3(n1, w), muls3(n2, u)), muls3(n3, v));
        const bool hitFrontFace = dot3(rayDirection, _normal) < 0.0f;
        normalVector = hitFrontFace ? _normal : neg3(_normal);
    }
    if (kRough) {                                                         // :169-179
        reflectDirection = add3(normalVector, random_space_vector(rng));
        if (fabsf(len2_3(reflectDirection) - kFloatZero * kFloatZero) < kFloatZero) reflectDirection = normalVector;
    } else {                                                              // :180-192
        const V3 vv = rayDirection, nn = normalVector;
        reflectDirection = normalize3(sub3(vv, muls3(nn, 2.0f * dot3(vv, nn))));
        if (hg.fuzz > 0.0f) reflectDirection = add3(reflectDirection, muls3(random_space_vector(rng), hg.fuzz));
    }
    // :202-213
    if (!finite3(reflectDirection) || len2_3(reflectDirection) <= kFloatZero * kFloatZero) {
        reflectDirection = normalVector;
        if (len2_3(reflectDirection) <= kFloatZero * kFloatZero || !finite3(reflectDirection))
            reflectDirection = mk3(0.0f, 0.0f, 1.0f);
    }
    // the depth-1 AOV write (:216-227) is overwritten by the terminating program (quirk Q3): nothing to keep
}
template <bool kSphere, bool kRough>
__device__ __forceinline__ void scatter(const HitGroup &hg, V3 rayOrigin, V3 rayDirection, float t, float u, float v,
                                        uint32_t primitiveIndex, Xorwow &rng, V3 &hitPoint, V3 &reflectDirection) {
    hitPoint = hit_point(rayOrigin, rayDirection, t);                     // :114
    scatter_at<kSphere, kRough>(hg, hitPoint, rayDirection, u, v, primitiveIndex, rng, reflectDirection);
}
// All four programs in ONE body, for kernels whose lanes run different programs side by side (fused.hip): the programs differ
// in where the normal comes from (sphere / triangle) and in what becomes of it (rough / metal); the random vector, its
// normalisation, the checks of the result and the hit point are the same instructions for every lane, executed once instead of
// once per program present in the wave.  Operation for operation what scatter<kSphere, kRough> computes.
template <bool HAS_SPHERES>
__device__ __forceinline__ void scatter_programs(uint32_t program, const HitGroup &hg, V3 rayOrigin, V3 rayDirection, float t, float u, float v,
                                                 uint32_t primitiveIndex, Xorwow &rng, V3 &hitPoint, V3 &reflectDirection) {
    hitPoint = hit_point(rayOrigin, rayDirection, t);                     // :114
    const bool rough = (program & 1u) == 0u;                              // kProgramSphereRough = 0, kProgramTriangleRough = 2
    V3 _normal;
    if (HAS_SPHERES && program < (uint32_t)kProgramTriangleRough) {       // :122-136
        const float *cp = reinterpret_cast<const float *>(hg.ptr0) + 3 * (size_t)primitiveIndex;
        const V3 sphereCenter = mk3(cp[0], cp[1], cp[2]);
        const float sphereRadius = reinterpret_cast<const float *>(hg.ptr1)[primitiveIndex];
        _normal = divs3(sub3(hitPoint, sphereCenter), sphereRadius);
    } else {                                                              // :137-155
        const float *np = reinterpret_cast<const float *>(hg.ptr0) + 9 * (size_t)primitiveIndex;
        const V3 n1 = mk3(np[0], np[1], np[2]), n2 = mk3(np[3], np[4], np[5]), n3 = mk3(np[6], np[7], np[8]);
        const float w = 1.0f - u - v;
        _normal = add3(add3(muls3(n1, w), muls3(n2, u)), muls3(n3, v));
    }
    const bool hitFrontFace = dot3(rayDirection, _normal) < 0.0f;
    const V3 normalVector = hitFrontFace ? _normal : neg3(_normal);
    V3 rsv = mk3(0.0f, 0.0f, 0.0f);
    if (rough || hg.fuzz > 0.0f) rsv = random_space_vector(rng);
    if (rough) {                                                          // :169-179
        reflectDirection = add3(normalVector, rsv);
        if (fabsf(len2_3(reflectDirection) - kFloatZero * kFloatZero) < kFloatZero) reflectDirection = normalVector;
    } else {                                                              // :180-192
        const V3 vv = rayDirection, nn = normalVector;
        reflectDirection = normalize3(sub3(vv, muls3(nn, 2.0f * dot3(vv, nn))));
        if (hg.fuzz > 0.0f) reflectDirection = add3(reflectDirection, muls3(rsv, hg.fuzz));
    }
    if (!finite3(reflectDirection) || len2_3(reflectDirection) <= kFloatZero * kFloatZero) {     // :202-213
        reflectDirection = normalVector;
        if (len2_3(reflectDirection) <= kFloatZero * kFloatZero || !finite3(reflectDirection))
            reflectDirection = mk3(0.0f, 0.0f, 1.0f);
    }
}
__device__ __forceinline__ bool program_draws(int program, const HitGroup &hg) {
    return program == kProgramSphereRough || program == kProgramTriangleRough || hg.fuzz > 0.0f;
}

// a path that ends at `depth`: miss colour (Shader.cu:276-287) or black at the depth limit (:102-107),
// then the albedo products of the unwinding recursion (:236-238), innermost bounce first
__device__ __forceinline__ V3 fold_chain(bool miss, const float *bg, const uint32_t *chain, uint32_t depth,
                                         const HitGroup *__restrict__ hitgroups) {
    V3 r = miss ? mk3(bg[0], bg[1], bg[2]) : mk3(0.0f, 0.0f, 0.0f);
    for (int k = (int)depth - 2; k >= 0; --k) {
        const HitGroup hg = hitgroups[chain[k]];
        r.x *= hg.albedo[0]; r.y *= hg.albedo[1]; r.z *= hg.albedo[2];
    }
    return r;
}

// ---------------------------------------------------------------------------------------
// traverse: persistent waves over a ray queue, BVH8 with compressed child boxes
// ---------------------------------------------------------------------------------------
constexpr int kLdsStack = 8;          // entries per lane staged in LDS
constexpr int kSpillStack = 56;       // overflow entries per lane in scratch
constexpr int kTraverseBlock = 64;         // one wave per workgroup

struct TravState {
    float ox, oy, oz, dx, dy, dz;
    float idx, idy, idz;
    float bt, bu, bv;
    uint32_t bprim, binst;
    uint32_t oct_inv4;
    uint2 cur;
    uint2 ptri;                   // leaf group being consumed, one primitive per iteration
    int sp, base;                 // stack = entries [base, sp): the bottom can be given away (tail splitting)
    uint32_t slot;
};

// 1 / d for the slab test.  The counting build divides exactly, so that its node / primitive counts equal the CPU walk of
// the same bytes (tests); the production build takes v_rcp_f32 (1 ULP, one instruction instead of the ~10 of the IEEE
// sequence, three times per ray): the slab test only culls, and its boxes are padded by far more than an ULP.
template <bool EXACT>
__device__ __forceinline__ float safe_rcp_dir(float d) {
    const float lim = 1e-20f;
    // (the sign is the one the slab test's near / far choice and the octant use, `d < 0`: a component of -0.0 -- a ray mirrored by an axis-aligned
    // wall -- counts as positive there, and with copysignf here its reciprocal was negative: near and far swapped, every box culled, the ray left
    // a closed room.  Found in round 4 by tools/stress_modes.py: one pixel of a Cornell box; tests/test_gpu_parity.py has the rays.)
    // (d + 0.0f: -0.0 becomes +0.0, everything else stays -- one instruction, no new constant in a kernel that has no register to spare)
    const float dd = fabsf(d) < lim ? copysignf(lim, d + 0.0f) : d;
    return EXACT ? 1.0f / dd : __builtin_amdgcn_rcpf(dd);
}

// canonical primitive test (DESIGN.md "canonical intersector"); updates the best hit.
// INSTANCED (two-level trees, fused.hip): the record is a shared BLAS's, in object space, and so is the ray in `s` by now (the transform
// node did that for triangles and spheres alike); the record does not know who instances it: `inst_cur` does.
// (test_prim_ray: the ray given explicitly -- a lane of the instanced kernel that has left an instance with leaf tests still to make keeps that
// instance's object-space ray in LDS while its registers hold the world ray again)
template <bool HAS_SPHERES, bool INSTANCED = false>
__device__ __forceinline__ bool test_prim_ray(const float4 A, const float4 B, const float4 C, TravState &s, const V3 o, const V3 d,
                                              float tmin, float tmax_ray,
                                              const float *__restrict__ inst_inv, const uint32_t *__restrict__ inst_identity, uint32_t inst_cur = 0u) {
    float t, u = 0.0f, v = 0.0f;
    uint32_t prim = __float_as_uint(A.w), inst;
    if (HAS_SPHERES && __float_as_uint(C.w) == 1u) {
        inst = INSTANCED ? inst_cur : __float_as_uint(B.w);
        V3 oo = o, dd = d;
        if (!INSTANCED && !inst_identity[inst]) {
            const float *m = inst_inv + 12 * (size_t)inst;
            oo = mk3(((m[0] * o.x + m[1] * o.y) + m[2] * o.z) + m[3], ((m[4] * o.x + m[5] * o.y) + m[6] * o.z) + m[7],
                     ((m[8] * o.x + m[9] * o.y) + m[10] * o.z) + m[11]);
            dd = mk3((m[0] * d.x + m[1] * d.y) + m[2] * d.z, (m[4] * d.x + m[5] * d.y) + m[6] * d.z,
                     (m[8] * d.x + m[9] * d.y) + m[10] * d.z);
        }
        const V3 oc = sub3(oo, mk3(A.x, A.y, A.z));
        const float r = B.x;
        const float a = dot3(dd, dd);
        if (!(a != 0.0f)) return false;
        const float b = dot3(oc, dd);
        const float cc = dot3(oc, oc) - r * r;
        const float disc = b * b - a * cc;
        if (!(disc >= 0.0f)) return false;
        const float sq = sqrtf(disc);
        const float t0 = (-b - sq) / a;
        if (t0 > tmin && t0 < tmax_ray) t = t0;
        else {
            const float t1 = (-b + sq) / a;
            if (t1 > tmin && t1 < tmax_ray) t = t1; else return false;
        }
    } else {
        // Straight-line: every quantity is computed for every lane and ONE condition decides.  The early returns of the
        // textbook form save nothing on a wave (the instructions run as long as one lane is still in) and cost a mask save /
        // branch / restore each; the SIMD issues ~1 instruction of ANY kind per 2.4 cycles, scalar ones included
        // (profiles/r02_valu_issue_patterns_microbench.txt).  Same decisions, same values: a rejected lane's later
        // quantities are never used, det == 0 gives inv = inf and u = NaN or +-inf, which fails the u test like the
        // explicit one does.
        inst = INSTANCED ? inst_cur : __float_as_uint(B.w);
        const V3 e1 = mk3(B.x, B.y, B.z), e2 = mk3(C.x, C.y, C.z);
        const V3 pvec = cross3(d, e2);
        const float det = dot3(e1, pvec);
        const float inv = 1.0f / det;
        const V3 tvec = sub3(o, mk3(A.x, A.y, A.z));
        u = dot3(tvec, pvec) * inv;
        const V3 qvec = cross3(tvec, e1);
        v = dot3(d, qvec) * inv;
        t = dot3(e2, qvec) * inv;
        const bool ok = (det != 0.0f) & (u >= 0.0f) & (u <= 1.0f) & (v >= 0.0f) & (u + v <= 1.0f) & (t > tmin) & (t < tmax_ray);
        if (!HAS_SPHERES) {
            // closest hit: min t, ties -> lowest (instance, primitive); predicated update
            const bool better = ok & ((t < s.bt) | ((t == s.bt) & ((inst < s.binst) | ((inst == s.binst) & (prim < s.bprim)))));
            s.bt = better ? t : s.bt; s.bu = better ? u : s.bu; s.bv = better ? v : s.bv;
            s.bprim = better ? prim : s.bprim; s.binst = better ? inst : s.binst;
            return better;
        }
        if (!ok) return false;
    }
    // closest hit: min t, ties -> lowest (instance, primitive)
    bool better = t < s.bt;
    if (!better && t == s.bt) {
        const uint64_t id = ((uint64_t)inst << 32) | prim, bid = ((uint64_t)s.binst << 32) | s.bprim;
        better = id < bid;
    }
    if (better) { s.bt = t; s.bu = u; s.bv = v; s.bprim = prim; s.binst = inst; }
    return better;
}

template <bool HAS_SPHERES, bool INSTANCED = false>
__device__ __forceinline__ bool test_prim(const float4 A, const float4 B, const float4 C, TravState &s,
                                          float tmin, float tmax_ray,
                                          const float *__restrict__ inst_inv, const uint32_t *__restrict__ inst_identity, uint32_t inst_cur = 0u) {
    return test_prim_ray<HAS_SPHERES, INSTANCED>(A, B, C, s, mk3(s.ox, s.oy, s.oz), mk3(s.dx, s.dy, s.dz), tmin, tmax_ray, inst_inv, inst_identity, inst_cur);
}

#define HRT_BYTE_F(w, k) ((float)(((w) >> (8 * (k))) & 0xffu))

// ---- cooperative gathers of the traversal pipeline ------------------------------------------------
// One ray per lane means 64 unrelated 80-byte nodes (and 48-byte primitives) per wave and step.
// Loaded lane by lane (5 + 3 dwordx4 per lane) that is 8 wave-instructions x 64 separate L1
// look-ups; measured, this address divergence -- not HBM, not L2 -- bounded the kernel (83 G
// steps/s on a 20 KB tree against 232 G with identical rays).  Instead the wave gathers the 64
// nodes with 5 LDS-DMA instructions (global_load_lds_dwordx4): LDS byte x of the 5120-byte
// staging image belongs to node slot x / 80, so lanes 5k..5k+4 of an instruction read the five
// consecutive 16-byte pieces of ONE node -- coalesced into one or two line look-ups -- and the
// data lands in LDS at wave base + 16 * lane, i.e. already as [slot][piece].  Each lane then
// reads its own slot back with ds_read_b128 (stride 80 B / 48 B is bank-conflict-free).
// The asm has no VGPR destination, so nothing can be read before it has landed except through
// LDS, and the two waits below carry a "memory" clobber: register-safe.  vmcnt is counted by
// hand: primitive pieces (3) are issued first, node pieces (5) second, so the primitive needs
// vmcnt(5) and the node vmcnt(0); VMEM operations the compiler adds can only lengthen the waits.
__device__ __forceinline__ void gather_node_pieces(uint32_t lds_base, const void *p0, const void *p1, const void *p2,
                                                   const void *p3, const void *p4) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "s"(lds_base) : "memory", "scc");
}
__device__ __forceinline__ void gather_prim_pieces(uint32_t lds_base, const void *p0, const void *p1, const void *p2) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(p0), "v"(p1), "v"(p2), "s"(lds_base) : "memory", "scc");
}
// Per-lane variant (template parameter DMA = false): every lane loads its own node / primitive into
// registers.  More L1 look-ups per step, but no staging image, so more waves fit per CU.  The
// destinations are named "+v" in the wait statements so that no use is scheduled above the wait.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void issue_prim_loads(const void *p, f32x4 &a, f32x4 &b, f32x4 &c) {
    asm volatile("global_load_dwordx4 %0, %3, off\n\t"
                 "global_load_dwordx4 %1, %3, off offset:16\n\t"
                 "global_load_dwordx4 %2, %3, off offset:32"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(p) : "memory");
}
__device__ __forceinline__ void issue_node_loads(const void *p, u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d, u32x4 &e) {
    asm volatile("global_load_dwordx4 %0, %5, off\n\t"
                 "global_load_dwordx4 %1, %5, off offset:16\n\t"
                 "global_load_dwordx4 %2, %5, off offset:32\n\t"
                 "global_load_dwordx4 %3, %5, off offset:48\n\t"
                 "global_load_dwordx4 %4, %5, off offset:64"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e) : "v"(p) : "memory");
}
// The same loads for a subset of the lanes (mask != 0, a subset of the lanes that are active here; wave-uniform): the vector
// memory pipeline returns 16 bytes per ACTIVE lane and instruction whatever the address -- 64 B per clock and CU in all, the
// resource that bounds the path kernel (profiles/r02_exp_bounds.txt) -- so lanes that have no primitive / node to fetch
// are switched off for the loads instead of fetching record 0.  Registers of the lanes left out keep their old contents
// (the operands are in/out: declare the registers once, outside the loop).
__device__ __forceinline__ void issue_prim_loads_masked(uint64_t mask, const void *p, f32x4 &a, f32x4 &b, f32x4 &c) {
    uint64_t save;
    asm volatile("s_mov_b64 %3, exec\n\t"
                 "s_mov_b64 exec, %5\n\t"
                 "global_load_dwordx4 %0, %4, off\n\t"
                 "global_load_dwordx4 %1, %4, off offset:16\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32\n\t"
                 "s_mov_b64 exec, %3"
                 : "+v"(a), "+v"(b), "+v"(c), "=&s"(save) : "v"(p), "s"(mask) : "memory");
}
__device__ __forceinline__ void issue_node_loads_masked(uint64_t mask, const void *p, u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d, u32x4 &e) {
    uint64_t save;
    asm volatile("s_mov_b64 %5, exec\n\t"
                 "s_mov_b64 exec, %7\n\t"
                 "global_load_dwordx4 %0, %6, off\n\t"
                 "global_load_dwordx4 %1, %6, off offset:16\n\t"
                 "global_load_dwordx4 %2, %6, off offset:32\n\t"
                 "global_load_dwordx4 %3, %6, off offset:48\n\t"
                 "global_load_dwordx4 %4, %6, off offset:64\n\t"
                 "s_mov_b64 exec, %5"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "=&s"(save) : "v"(p), "s"(mask) : "memory");
}
__device__ __forceinline__ void wait_all_prim_loads(f32x4 &a, f32x4 &b, f32x4 &c) {       // no node loads behind them
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c) :: "memory");
}
__device__ __forceinline__ void wait_prim_loads(f32x4 &a, f32x4 &b, f32x4 &c) {
    asm volatile("s_waitcnt vmcnt(5)" : "+v"(a), "+v"(b), "+v"(c) :: "memory");
}
__device__ __forceinline__ void wait_node_loads(u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d, u32x4 &e) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) :: "memory");
}
__device__ __forceinline__ void wait_prim_gather() { asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
__device__ __forceinline__ void wait_node_gather() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

}  // namespace hrt
