// build_dev.h -- device helpers shared by the two halves of the acceleration-structure build (build.hip: bounds, Morton order, PLOC,
// collapse, emission; build_split.hip: the top-down phase with spatial splits).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "build.h"
#include "bvh8.h"
#include "bvh8_geom.h"

namespace hrt {
namespace {

constexpr uint32_t kNone = 0xffffffffu;

// float <-> unsigned with the same order
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__host__ __device__ inline float ord2f(uint32_t u) { const uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; float f; memcpy(&f, &b, 4); return f; }

__device__ __forceinline__ uint32_t find_instance(const uint32_t *__restrict__ first, uint32_t n_inst, uint32_t k) {
    uint32_t lo = 0, hi = n_inst;               // first[lo] <= k < first[hi]
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (first[mid] <= k) lo = mid; else hi = mid; }
    return lo;
}

// world-space bounds of global primitive k (unpadded); false when not finite
__device__ __forceinline__ bool prim_bounds(const GpuBuildArgs &a, uint32_t k, float *lo, float *hi) {
    const uint32_t inst = find_instance(a.inst_first, a.n_inst, k), p = k - a.inst_first[inst];
    const float *m = a.inst_xf + 12 * (size_t)inst;
    const bool ident = a.inst_identity[inst] != 0u;
    if (a.inst_kind[inst] == kPrimKindTriangle) {
        const float *src = reinterpret_cast<const float *>(a.inst_src[inst]) + 9 * (size_t)p;
        float s9[9], v0[3], e1[3], e2[3];
        for (int q = 0; q < 9; ++q) s9[q] = src[q];
        triangle_world(s9, m, ident, v0, e1, e2, lo, hi);
    } else if (a.inst_kind[inst] == kPrimKindInstance) {
        const float *src = reinterpret_cast<const float *>(a.inst_src[inst]);      // the BLAS's object-space box and bounding sphere
        float b10[10];
        for (int q = 0; q < 10; ++q) b10[q] = src[q];
        if (!(b10[0] <= b10[3])) return false;                                     // an empty BLAS
        instance_world_bounds(b10, m, ident, lo, hi);
    } else {
        const float *src = reinterpret_cast<const float *>(a.inst_src[inst]) + 4 * (size_t)p;
        const float c3[3] = {src[0], src[1], src[2]};
        sphere_world_bounds(c3, src[3], m, ident, lo, hi);
    }
    return finite_box(lo, hi);
}

// min / max of up to 12 values over a 1024-thread workgroup: waves by shuffles, then one lane per value through LDS.  Returns
// the block's result in every thread of wave 0 (valid for lane < n_vals there).  (One atomic per value and BLOCK afterwards:
// a single address takes ~88 atomics per microsecond, so per-wave atomics cost milliseconds at a million primitives.)
template <int N>
__device__ __forceinline__ void block_minmax(float (&mn)[N], float (&mx)[N]) {
    __shared__ float s_mn[16][N], s_mx[16][N];
    for (int off = 32; off > 0; off >>= 1)
        for (int q = 0; q < N; ++q) { mn[q] = fminf(mn[q], __shfl_xor(mn[q], off)); mx[q] = fmaxf(mx[q], __shfl_xor(mx[q], off)); }
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, n_waves = (blockDim.x + 63u) >> 6;
    if (lane == 0u) for (int q = 0; q < N; ++q) { s_mn[wave][q] = mn[q]; s_mx[wave][q] = mx[q]; }
    __syncthreads();
    if (wave == 0u)
        for (int q = 0; q < N; ++q) {
            float a = INFINITY, b = -INFINITY;
            for (uint32_t w = 0; w < n_waves; ++w) { a = fminf(a, s_mn[w][q]); b = fmaxf(b, s_mx[w][q]); }
            mn[q] = a; mx[q] = b;
        }
}

inline uint32_t blocks(uint32_t n, uint32_t per) { return (n + per - 1u) / per; }

}  // namespace
}  // namespace hrt
