// bvh8_geom.h -- the arithmetic the host builder and the device refit must agree on, bit for bit:
// instance transform of a point, world-space primitive records and bounds, and the quantisation of
// child boxes against a node's origin / exponents.  Compiled with -ffp-contract=off on both sides, so
// a refit under unchanged transforms reproduces the built tree byte for byte (tests/test_gpu_parity.py).
//
// Replaces (together with the refit kernel) what optixAccelBuild(OPTIX_BUILD_OPERATION_UPDATE) does for
// the reference's per-frame updateIAS (src/Global/RendererImpl.cu:210-242).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HRT_HD __host__ __device__ inline
#else
#define HRT_HD inline
#endif

namespace hrt {

// 3x4 row-major affine transform of a point, fixed operation order (the oracle applies the same one)
HRT_HD void xf_point(const float *m, const float *p, float *o) {
    o[0] = ((m[0] * p[0] + m[1] * p[1]) + m[2] * p[2]) + m[3];
    o[1] = ((m[4] * p[0] + m[5] * p[1]) + m[6] * p[2]) + m[7];
    o[2] = ((m[8] * p[0] + m[9] * p[1]) + m[10] * p[2]) + m[11];
}

// World-space triangle record (v0, e1 = v1 - v0, e2 = v2 - v0) and bounds from 9 object-space floats.
HRT_HD void triangle_world(const float *src9, const float *m, bool identity, float *v0, float *e1, float *e2, float *lo, float *hi) {
    float v[3][3];
    for (int k = 0; k < 3; ++k) {
        const float *s = src9 + 3 * k;
        if (identity) { v[k][0] = s[0]; v[k][1] = s[1]; v[k][2] = s[2]; } else xf_point(m, s, v[k]);
    }
    for (int a = 0; a < 3; ++a) {
        v0[a] = v[0][a]; e1[a] = v[1][a] - v[0][a]; e2[a] = v[2][a] - v[0][a];
        lo[a] = fminf(v[0][a], fminf(v[1][a], v[2][a]));
        hi[a] = fmaxf(v[0][a], fmaxf(v[1][a], v[2][a]));
    }
}

// World-space bounds of an object-space sphere under an affine instance transform: the transformed
// corners of its object-space box (exact for the rotations / scales the reference composes).
HRT_HD void sphere_world_bounds(const float *c, float r, const float *m, bool identity, float *lo, float *hi) {
    const float rr = fabsf(r);
    for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    for (int k = 0; k < 8; ++k) {
        const float q[3] = {c[0] + ((k & 1) ? rr : -rr), c[1] + ((k & 2) ? rr : -rr), c[2] + ((k & 4) ? rr : -rr)};
        float w[3];
        if (identity) { w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; } else xf_point(m, q, w);
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], w[a]); hi[a] = fmaxf(hi[a], w[a]); }
    }
}

// World-space bounds of an object-space box (lo[3], hi[3]) under an instance transform: its transformed corners (the top level of a
// two-level tree: an instance's BLAS box).
HRT_HD void box_world_bounds(const float *b6, const float *m, bool identity, float *lo, float *hi) {
    for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    for (int k = 0; k < 8; ++k) {
        const float q[3] = {(k & 1) ? b6[3] : b6[0], (k & 2) ? b6[4] : b6[1], (k & 4) ? b6[5] : b6[2]};
        float w[3];
        if (identity) { w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; } else xf_point(m, q, w);
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], w[a]); hi[a] = fmaxf(hi[a], w[a]); }
    }
}

// The bounding sphere of a BLAS (centre c, radius r >= 0 around all its geometry) under an affine map is an ellipsoid whose extent along
// world axis k is r x |row k of the linear part|: its box, intersected with the transformed corners' box, bounds the instance.  For the
// compact bodies a DEM run instances that is the tight one of the two: a cube turned by 45 degrees about two axes has a corners' box
// 1.7 times its size, its sphere's box 1.0.  (The box is padded like every other by whoever stores it.)
HRT_HD void clamp_to_sphere_bounds(const float *c3, float r, const float *m, bool identity, float *lo, float *hi) {
    if (!(r >= 0.0f) || !(r <= 3.0e38f)) return;
    float w[3];
    if (identity) { w[0] = c3[0]; w[1] = c3[1]; w[2] = c3[2]; } else xf_point(m, c3, w);
    for (int k = 0; k < 3; ++k) {
        const float n = identity ? 1.0f : sqrtf((m[4 * k] * m[4 * k] + m[4 * k + 1] * m[4 * k + 1]) + m[4 * k + 2] * m[4 * k + 2]);
        const float e = (r * n) * 1.000002f;
        const float slo = w[k] - e, shi = w[k] + e;
        if (slo <= shi) { lo[k] = fmaxf(lo[k], slo); hi[k] = fminf(hi[k], shi); }      // (NaN: the corners' box stands)
    }
}
// an instance of a two-level tree: b10 = its BLAS's object-space box (lo, hi), bounding sphere centre and radius (negative: none)
HRT_HD void instance_world_bounds(const float *b10, const float *m, bool identity, float *lo, float *hi) {
    box_world_bounds(b10, m, identity, lo, hi);
    clamp_to_sphere_bounds(b10 + 6, b10[9], m, identity, lo, hi);
}

HRT_HD float box_half_area(const float *lo, const float *hi) {
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    return ex * ey + ey * ez + ez * ex;
}

HRT_HD bool finite_box(const float *lo, const float *hi) {
    bool ok = true;
    for (int a = 0; a < 3; ++a) ok = ok && (fabsf(lo[a]) <= 3.0e38f) && (fabsf(hi[a]) <= 3.0e38f);
    return ok;
}

// Bounds of the part of a triangle (record form: v0, e1, e2) inside the box [blo, bhi]: Sutherland-Hodgman against the six planes in
// double arithmetic, rounded outwards by one float ULP and never beyond the box.  Used where a spatial split cuts a reference
// (host builder bvh8_build.cpp and device builder build_split.hip: one arithmetic).  (The record keeps the vertices as v0, v0 + e1,
// v0 + e2 in float: the sums in double are the vertices to within half an ULP, which the outward rounding and the builders' padding cover.)
// A numerically empty result leaves the box itself: conservative.
HRT_HD void clip_triangle_to_box(const float *v0, const float *e1, const float *e2, const float *blo, const float *bhi, float *lo, float *hi) {
    double poly[2][16][3]; int np = 3, cur = 0;
    for (int c = 0; c < 3; ++c) {
        poly[0][0][c] = v0[c];
        poly[0][1][c] = (double)v0[c] + (double)e1[c];
        poly[0][2][c] = (double)v0[c] + (double)e2[c];
    }
    for (int c = 0; c < 3 && np > 0; ++c)
        for (int side = 0; side < 2 && np > 0; ++side) {
            const double plane = side == 0 ? (double)blo[c] : (double)bhi[c];
            const double sgn = side == 0 ? 1.0 : -1.0;
            int nq = 0;
            for (int i = 0; i < np; ++i) {
                const double *u = poly[cur][i], *v = poly[cur][(i + 1) % np];
                const double du = sgn * (u[c] - plane), dv = sgn * (v[c] - plane);
                if (du >= 0.0) { for (int k = 0; k < 3; ++k) poly[cur ^ 1][nq][k] = u[k]; ++nq; }
                if ((du > 0.0 && dv < 0.0) || (du < 0.0 && dv > 0.0)) {
                    const double t = du / (du - dv);
                    for (int k = 0; k < 3; ++k) poly[cur ^ 1][nq][k] = k == c ? plane : u[k] + t * (v[k] - u[k]);
                    ++nq;
                }
            }
            np = nq; cur ^= 1;
        }
    if (np == 0) { for (int c = 0; c < 3; ++c) { lo[c] = blo[c]; hi[c] = bhi[c]; } return; }
    for (int c = 0; c < 3; ++c) {
        double l = poly[cur][0][c], h = l;
        for (int i = 1; i < np; ++i) { l = l < poly[cur][i][c] ? l : poly[cur][i][c]; h = h > poly[cur][i][c] ? h : poly[cur][i][c]; }
        float fl = (float)l, fh = (float)h;
        fl = nextafterf(fl, -INFINITY); fh = nextafterf(fh, INFINITY);
        lo[c] = blo[c] > fl ? blo[c] : fl; hi[c] = bhi[c] < fh ? bhi[c] : fh;
        if (lo[c] > hi[c]) { lo[c] = blo[c]; hi[c] = bhi[c]; }
    }
}

// Smallest exponent e (biased by 127, within [1, 254]) with 255 * 2^(e-127) >= ext, from the bits of ext.
HRT_HD uint8_t node_exponent(float ext) {
    if (!(ext > 0.0f)) return 1;
    uint32_t bits; memcpy(&bits, &ext, 4);
    const int E = (int)((bits >> 23) & 0xffu);
    const uint32_t M = bits & 0x7fffffu;
    if (E == 0) return 1;                                  // subnormal extent
    if (E == 255) return 254;                              // inf / nan: callers never build on those
    // ext = (1 + M/2^23) * 2^(E-127); 255 * 2^-8 covers mantissas up to 255/128
    int e = (M <= 0x7f0000u) ? (E - 127) - 7 : (E - 127) - 6;
    if (e < -126) e = -126;
    if (e > 127) e = 127;
    return (uint8_t)(e + 127);
}

HRT_HD float exponent_scale(uint8_t e) {
    const uint32_t bits = (uint32_t)e << 23; float f; memcpy(&f, &bits, 4); return f;
}

// Outward-rounded 8-bit coordinates of [clo, chi] on the grid p + q * 2^(e-127); the float decode
// p + (float)q * scale stays outside the box.
HRT_HD void quantise_axis(float p, uint8_t e, float clo, float chi, uint8_t *qlo, uint8_t *qhi) {
    const float fs = exponent_scale(e);
    const double sc = (double)fs;
    double dl = floor(((double)clo - (double)p) / sc), dh = ceil(((double)chi - (double)p) / sc);
    dl = (dl >= 0.0) ? (dl > 255.0 ? 255.0 : dl) : 0.0;          // NaN -> widest
    dh = (dh <= 255.0) ? (dh < 0.0 ? 0.0 : dh) : 255.0;
    int ql = (int)dl, qh = (int)dh;
    while (ql > 0 && p + (float)ql * fs > clo) --ql;
    while (qh < 255 && p + (float)qh * fs < chi) ++qh;
    *qlo = (uint8_t)ql; *qhi = (uint8_t)qh;
}

}  // namespace hrt
