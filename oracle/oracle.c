/*
 * oracle.c -- CPU restatement of the reference's per-frame ray-tracing launch.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (libhrt.so, the package, the
 * bench's GPU leg) may call into this file; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg do, as the checker / the timed CPU baseline.
 *
 * PARITY STATUS: **parity unpinned**.  The reference has no tests, no golden
 * vectors and no CPU path; it cannot be built here (needs
 * CUDA + OptiX SDK + SDL2 + Vulkan + VTK; writing stand-in headers for them is
 * not allowed), and three of its ingredients are third-party arithmetic that is
 * not in the tree:
 *   - OptiX 9.0 built-in triangle/sphere intersection + RT-core traversal
 *     (src/Global/RendererImpl.cu:295-314, shader/Shader.cu:70-75,111,117,145)
 *       -> replaced by the CANONICAL INTERSECTOR defined below (Moeller-Trumbore,
 *          analytic sphere, closest hit = min t then min (instance, primitive)).
 *   - cuRAND XORWOW curand_init / curand_uniform (CUDA >= 12;
 *     src/Global/HostFunctions.cu:126, include/Global/DeviceFunctions.cuh:216-218)
 *       -> restated from the published algorithm (Marsaglia xorwow + cuRAND's
 *          seed scrambling and 2^67 sub-sequence skip-ahead).  The skip-ahead
 *          algebra is pinned against rocRAND's host-callable xorwow engine in
 *          tests/test_oracle_cpu.py (same recurrence, different scramble constants).
 *   - device rsqrtf / powf and nvcc's FMA contraction
 *       -> pinned as 1.0f/sqrtf(x), the correctly rounded powf (libm double pow + __float128 near ties,
 *          see pow_inv_gamma_cr), and NO contraction (-ffp-contract=off).
 * Everything else follows the reference source line by line; each function cites it.
 * Paths are relative to the reference tree.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------ */
/* small vector helpers: include/Global/DeviceFunctions.cuh:304-416                     */
/* ------------------------------------------------------------------------------------ */
typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;

#define FLOAT_ZERO_VALUE 1e-6f          /* DeviceFunctions.cuh:18 */
#define FLOAT_INFINITY_VALUE 1e16f      /* DeviceFunctions.cuh:19 */
#define RAY_TRACE_DEPTH 5u              /* Shader.cuh:8 */

static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }                              /* :304 */
static inline f3 muls3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }           /* :311-321 */
static inline f3 divs3(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }           /* :340-347 */
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }         /* :371-378 */
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }         /* :379-386 */
static inline float len2_3(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }              /* :389-391 */
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          /* :407-409 */
static inline f3 cross3(f3 a, f3 b) {                                                       /* :410-416 */
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* normalize, DeviceFunctions.cuh:397-404; rsqrtf pinned as 1/sqrtf (see header) */
static inline f3 normalize3(f3 a) {
    const float len2 = len2_3(a);
    if (len2 <= FLOAT_ZERO_VALUE * FLOAT_ZERO_VALUE) return mk3(0.0f, 0.0f, 1.0f);
    const float invLen = 1.0f / sqrtf(len2);
    return muls3(a, invLen);
}
/* MathHelper::floatValueEquals, DeviceFunctions.cuh:38-40 */
static inline int float_value_equals(float v1, float v2) { return fabsf(v1 - v2) < FLOAT_ZERO_VALUE; }

/* ------------------------------------------------------------------------------------ */
/* colour conversion: include/Global/DeviceFunctions.cuh:153-212                        */
/* ------------------------------------------------------------------------------------ */
/* powf(cx, 1.0f / 2.4f) of :161-163 / :196-198.  CUDA's powf is third-party arithmetic that is not in the tree
 * (SURVEY.md 8c); it is pinned here as the CORRECTLY ROUNDED float of cx^y, y = (double)(1.0f / 2.4f) -- the one
 * definition that libm, this file and the HIP kernels can all reproduce bit for bit.  libm's double pow is rounded to
 * float; where that double lies within 2^-45 (relative) of the midpoint of two adjacent floats (733 of the 1 065 353 217
 * floats in [0, 1]) the decision is taken in __float128 (libquadmath).  Over all floats in [0, 1] this differs from
 * (float)pow(double) in exactly ONE input (cx = 0x1.20eb96p-20, a double-rounding case) and from libm's powf in
 * 672 423 (all by 1 ULP); tests/test_oracle_cpu.py pins both counts' bounds. */
static float pow_inv_gamma_cr(float xf) {
    if (!(xf > 0.0f)) return 0.0f;
    const double y = (double)(1.0f / 2.4f);                          /* invGamma :195, widened exactly */
    const double r = pow((double)xf, y);
    const float c = (float)r;
    uint32_t cb; memcpy(&cb, &c, 4);
    uint32_t ub = cb + 1u, db = cb - 1u;
    float cu, cd; memcpy(&cu, &ub, 4); memcpy(&cd, &db, 4);
    const double mu = 0.5 * ((double)c + (double)cu), md = 0.5 * ((double)c + (double)cd), tol = r * 0x1p-45;
    if (fabs(r - mu) < tol || fabs(r - md) < tol) {
        const __float128 q = powq((__float128)xf, (__float128)y);
        float best = c; __float128 bd = fabsq(q - (__float128)c);
        if (fabsq(q - (__float128)cu) < bd) { bd = fabsq(q - (__float128)cu); best = cu; }
        if (fabsq(q - (__float128)cd) < bd) { bd = fabsq(q - (__float128)cd); best = cd; }
        return best;
    }
    return c;
}
float oracle_pow_inv_gamma(float x) { return pow_inv_gamma_cr(x); }
float oracle_pow_inv_gamma_libm_powf(float x) { return powf(x, 1.0f / 2.4f); }      /* tolerance cross-check only */

static inline float srgb_channel(float c) {
    const float cx = fmaxf(0.0f, fminf(c, 1.0f));                    /* :190 */
    const float px = pow_inv_gamma_cr(cx);                           /* :195-196, pinned as above */
    const float sx = cx < 0.0031308f ? 12.92f * cx : 1.055f * px - 0.055f;   /* :199 */
    return fmaxf(0.0f, fminf(sx, 1.0f));                             /* :204 */
}
/* n colours (float4 in, float4 out): colorToFloat4 over an array, for the exhaustive sweeps of the tests */
void oracle_color_to_float4_n(const float *src4, float *dst4, uint64_t n) {
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; ++i) {
        dst4[4 * i + 0] = srgb_channel(src4[4 * i + 0]);
        dst4[4 * i + 1] = srgb_channel(src4[4 * i + 1]);
        dst4[4 * i + 2] = srgb_channel(src4[4 * i + 2]);
        dst4[4 * i + 3] = 1.0f;
    }
}
/* pow_inv_gamma over the floats whose bit patterns are first, first + 1, ... (count of them) */
void oracle_pow_inv_gamma_bits(uint32_t first, uint64_t count, float *out) {
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < count; ++i) {
        const uint32_t b = first + (uint32_t)i; float x; memcpy(&x, &b, 4);
        out[i] = pow_inv_gamma_cr(x);
    }
}
void oracle_color_to_float4(const float *rgb, float *out4) {        /* colorToFloat4 :188-209 */
    out4[0] = srgb_channel(rgb[0]);
    out4[1] = srgb_channel(rgb[1]);
    out4[2] = srgb_channel(rgb[2]);
    out4[3] = 1.0f;
}
void oracle_color_to_uchar4(const float *rgb, unsigned char *out4) { /* colorToUchar4 :153-183 */
    for (int k = 0; k < 3; ++k) {
        const float s = srgb_channel(rgb[k]);
        unsigned int q = (unsigned int)(s * 256.0f);                 /* :178 */
        out4[k] = (unsigned char)(q < 255u ? q : 255u);
    }
    out4[3] = 255u;
}

/* ------------------------------------------------------------------------------------ */
/* camera basis: src/GraphicsAPI/SDL_GraphicsWindow.cu:4-14                              */
/* ------------------------------------------------------------------------------------ */
void oracle_configure_camera(const float *center, const float *target, const float *up,
                             int is_opengl, float *outU, float *outV, float *outW) {
    f3 upDirection = normalize3(mk3(up[0], up[1], up[2]));           /* :5 */
    if (!is_opengl) upDirection = neg3(upDirection);                 /* :7-9 */
    const f3 W = sub3(mk3(target[0], target[1], target[2]), mk3(center[0], center[1], center[2]));  /* :10 */
    const f3 U = normalize3(cross3(W, upDirection));                 /* :11 */
    const f3 V = normalize3(cross3(U, W));                           /* :12 */
    outU[0] = U.x; outU[1] = U.y; outU[2] = U.z;
    outV[0] = V.x; outV[1] = V.y; outV[2] = V.z;
    outW[0] = W.x; outW[1] = W.y; outW[2] = W.z;
}

/* ------------------------------------------------------------------------------------ */
/* XORWOW (cuRAND curandStateXORWOW_t; third-party, restated from the published scheme)  */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t d, v[5];
    int32_t boxmuller_flag, boxmuller_flag_double;
    float boxmuller_extra, _pad;
    double boxmuller_extra_double;
} rng_state;                                                         /* 48 B */

static inline uint32_t xorwow_next(rng_state *s) {
    uint32_t t = s->v[0] ^ (s->v[0] >> 2);
    s->v[0] = s->v[1]; s->v[1] = s->v[2]; s->v[2] = s->v[3]; s->v[3] = s->v[4];
    s->v[4] = (s->v[4] ^ (s->v[4] << 4)) ^ (t ^ (t << 1));
    s->d += 362437u;
    return s->v[4] + s->d;
}
/* curand_uniform: (0,1], x * 2^-32 + 2^-33 */
static inline float xorwow_uniform(rng_state *s) {
    return (float)xorwow_next(s) * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}
float oracle_rng_uniform(void *state) { return xorwow_uniform((rng_state *)state); }
uint32_t oracle_rng_next(void *state) { return xorwow_next((rng_state *)state); }

/* 160x160 GF(2) matrices of the xorshift part, column form: m[i] = image of basis bit i. */
typedef struct { uint32_t col[160][5]; } gf2m;
static gf2m g_seq_jump[32];           /* g_seq_jump[k] = T^(2^(67+k)) */
static int g_jump_ready = 0;

static void gf2_apply(const gf2m *m, const uint32_t in[5], uint32_t out[5]) {
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < 5; ++w)
        for (int b = 0; b < 32; ++b)
            if ((in[w] >> b) & 1u) {
                const uint32_t *c = m->col[w * 32 + b];
                r[0] ^= c[0]; r[1] ^= c[1]; r[2] ^= c[2]; r[3] ^= c[3]; r[4] ^= c[4];
            }
    memcpy(out, r, sizeof r);
}
static void gf2_square(const gf2m *m, gf2m *out) {
    gf2m tmp;
    for (int i = 0; i < 160; ++i) gf2_apply(m, m->col[i], tmp.col[i]);
    *out = tmp;
}
void oracle_init(void) {
    if (g_jump_ready) return;
    #pragma omp critical(oracle_init_lock)
    {
        if (!g_jump_ready) {
            gf2m *t = (gf2m *)malloc(sizeof(gf2m));
            for (int i = 0; i < 160; ++i) {          /* one xorshift step on each basis vector */
                rng_state s; memset(&s, 0, sizeof s);
                s.v[i / 32] = 1u << (i % 32);
                (void)xorwow_next(&s);
                memcpy(t->col[i], s.v, sizeof s.v);
            }
            for (int k = 0; k < 67; ++k) gf2_square(t, t);      /* T^(2^67) */
            g_seq_jump[0] = *t;
            for (int k = 1; k < 32; ++k) gf2_square(&g_seq_jump[k - 1], &g_seq_jump[k]);
            free(t);
            g_jump_ready = 1;
        }
    }
}
/* generic init with explicit scramble constants so the skip-ahead can be pinned against
 * rocRAND's engine (which uses other constants) */
void oracle_rng_init_generic(void *state, uint64_t seed, uint64_t subsequence,
                             uint32_t c_s0, uint32_t c_s1, uint32_t c_m0, uint32_t c_m1) {
    oracle_init();
    rng_state *st = (rng_state *)state;
    const uint32_t s0 = ((uint32_t)seed) ^ c_s0;
    const uint32_t s1 = ((uint32_t)(seed >> 32)) ^ c_s1;
    const uint32_t t0 = c_m0 * s0;
    const uint32_t t1 = c_m1 * s1;
    st->d = 6615241u + t1 + t0;
    st->v[0] = 123456789u + t0;
    st->v[1] = 362436069u ^ t0;
    st->v[2] = 521288629u + t1;
    st->v[3] = 88675123u ^ t1;
    st->v[4] = 5783321u + t0;
    /* skip subsequence * 2^67 draws; d is unchanged because 2^67 * 362437 = 0 mod 2^32 */
    for (int k = 0; k < 32 && (subsequence >> k); ++k)
        if ((subsequence >> k) & 1u) gf2_apply(&g_seq_jump[k], st->v, st->v);
    st->boxmuller_flag = 0; st->boxmuller_flag_double = 0;
    st->boxmuller_extra = 0.0f; st->_pad = 0.0f; st->boxmuller_extra_double = 0.0;
}
/* curand_init(seed, subsequence, 0, state) with cuRAND's constants */
void oracle_rng_init_one(void *state, uint64_t seed, uint64_t subsequence) {
    oracle_rng_init_generic(state, seed, subsequence, 0xaad26b49u, 0xf7dcefddu, 1099087573u, 2591861531u);
}
/* RandomGenerator::initDeviceRandomGenerators, src/Global/HostFunctions.cu:122-136, with the
 * clock64() term pinned to seed_salt (Q8), bounds-checked and indexed by the frame width (Q9). */
void oracle_rng_init(void *states, uint32_t width, uint32_t height, uint64_t seed_salt) {
    oracle_init();
    rng_state *st = (rng_state *)states;
    const long n = (long)width * (long)height;
    #pragma omp parallel for schedule(static)
    for (long tid = 0; tid < n; ++tid)
        oracle_rng_init_one(&st[tid], ((uint64_t)tid) ^ seed_salt, (uint64_t)tid);
}

/* randomDouble(state, min, max), DeviceFunctions.cuh:220-222 */
static inline float random_double_range(rng_state *s, float mn, float mx) {
    return mn + (mx - mn) * xorwow_uniform(s);
}
/* randomSpaceVector, DeviceFunctions.cuh:570-582 */
static f3 random_space_vector(rng_state *s, float length) {
    f3 ret; float lengthSquare;
    do {
        ret.x = random_double_range(s, -1.0f, 1.0f);
        ret.y = random_double_range(s, -1.0f, 1.0f);
        ret.z = random_double_range(s, -1.0f, 1.0f);
        lengthSquare = len2_3(ret);
    } while (lengthSquare < FLOAT_ZERO_VALUE * FLOAT_ZERO_VALUE);
    ret = normalize3(ret);
    return muls3(ret, length);
}

/* ------------------------------------------------------------------------------------ */
/* scene                                                                                 */
/* ------------------------------------------------------------------------------------ */
enum { GEOM_SPHERE = 0, GEOM_TRIANGLE = 1 };                 /* GeometryType, Shader.cuh:11-13 */
enum { MAT_ROUGH = 0, MAT_METAL = 1 };                       /* MaterialType, Shader.cuh:16-18 */

/* One instance = one GAS + one SBT record (src/Global/RendererMesh.cu:131-144). */
typedef struct {
    float transform[12];            /* row-major 3x4 object->world                          */
    int32_t geometry;               /* GEOM_*                                               */
    int32_t material;               /* MAT_*                                                */
    float albedo[3];
    float fuzz;
    uint32_t n_prims;
    const float *vertices;          /* triangles: 9 floats per triangle (object space)      */
    const float *normals;           /* triangles: 9 floats per triangle = HitGroupParams.triangles.vertexNormals */
    const float *centers;           /* spheres: 3 floats each = HitGroupParams.sphere.centers */
    const float *radii;             /* spheres                                              */
} oracle_instance;

typedef struct { f3 v0, e1, e2; uint32_t prim, inst; } wtri;         /* world-space triangle */
typedef struct { f3 c; float r; uint32_t prim, inst; } wsph;         /* object-space sphere   */
typedef struct { float lo[3], hi[3]; uint32_t left, count; } bnode;  /* BVH2: count>0 => leaf [left,left+count) */

typedef struct {
    int n_inst;
    oracle_instance *inst;
    float (*inv)[12];               /* per instance inverse transform (spheres)             */
    int *identity;
    uint32_t n_tri, n_sph, n_prim;
    wtri *tri; wsph *sph;
    /* accel over prim refs: ref < n_tri => triangle, else sphere (ref - n_tri) */
    uint32_t *refs; float *plo, *phi;   /* per-ref padded AABB */
    bnode *nodes; uint32_t n_nodes;
    int brute;
    int object_space;               /* 1: the INSTANCED canonical mode (below): triangles of transformed instances stay in object space */
    /* optional: the PRODUCT's packed BVH8 (hrt_tlas_download), walked instead of the BVH2 above -- the closest hit is canonical, so the
     * image is the same; bench.py's cpu_baseline times the CPU on the very bytes the GPU traverses */
    const void *bvh8_nodes, *bvh8_prims;
} oracle_scene;

/* point / vector transform with a fixed operation order (shared definition with the product) */
static inline f3 xf_point(const float *m, f3 p) {
    return mk3(((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3],
               ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
               ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]);
}
static inline f3 xf_vector(const float *m, f3 p) {
    return mk3((m[0] * p.x + m[1] * p.y) + m[2] * p.z,
               (m[4] * p.x + m[5] * p.y) + m[6] * p.z,
               (m[8] * p.x + m[9] * p.y) + m[10] * p.z);
}
static int is_identity(const float *m) {
    static const float id[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    return memcmp(m, id, sizeof id) == 0;
}
/* inverse of a 3x4 affine map, cofactors in double, fixed order, rounded to float once */
static void invert_affine(const float *m, float *o) {
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C;
    const double r = 1.0 / det;
    const double n00 = A * r, n01 = -(b * i - c * h) * r, n02 = (b * f - c * e) * r;
    const double n10 = B * r, n11 = (a * i - c * g) * r, n12 = -(a * f - c * d) * r;
    const double n20 = C * r, n21 = -(a * h - b * g) * r, n22 = (a * e - b * d) * r;
    const double tx = m[3], ty = m[7], tz = m[11];
    o[0] = (float)n00; o[1] = (float)n01; o[2] = (float)n02; o[3] = (float)(-(n00 * tx + n01 * ty + n02 * tz));
    o[4] = (float)n10; o[5] = (float)n11; o[6] = (float)n12; o[7] = (float)(-(n10 * tx + n11 * ty + n12 * tz));
    o[8] = (float)n20; o[9] = (float)n21; o[10] = (float)n22; o[11] = (float)(-(n20 * tx + n21 * ty + n22 * tz));
}

/* ------------------------------------------------------------------------------------ */
/* CANONICAL INTERSECTOR (defines what OptiX's built-in intersection is replaced by)     */
/* ------------------------------------------------------------------------------------ */
typedef struct { float t, u, v; uint32_t prim, inst; int hit; } hit_rec;

/* Moeller-Trumbore, barycentrics P = (1-u-v) V0 + u V1 + v V2 (shader/Shader.cu:145-151) */
static inline int isect_tri(const wtri *tr, f3 o, f3 d, float tmin, float tmax, float *t, float *u, float *v) {
    const f3 pvec = cross3(d, tr->e2);
    const float det = dot3(tr->e1, pvec);
    if (!(det != 0.0f)) return 0;
    const float inv = 1.0f / det;
    const f3 tvec = sub3(o, tr->v0);
    const float uu = dot3(tvec, pvec) * inv;
    if (!(uu >= 0.0f && uu <= 1.0f)) return 0;
    const f3 qvec = cross3(tvec, tr->e1);
    const float vv = dot3(d, qvec) * inv;
    if (!(vv >= 0.0f && uu + vv <= 1.0f)) return 0;
    const float tt = dot3(tr->e2, qvec) * inv;
    if (!(tt > tmin && tt < tmax)) return 0;
    *t = tt; *u = uu; *v = vv;
    return 1;
}
/* analytic sphere in object space; first root inside (tmin,tmax) wins */
static inline int isect_sph(const wsph *s, f3 o, f3 d, float tmin, float tmax, float *t) {
    const f3 oc = sub3(o, s->c);
    const float a = dot3(d, d);
    if (!(a != 0.0f)) return 0;
    const float b = dot3(oc, d);
    const float cc = dot3(oc, oc) - s->r * s->r;
    const float disc = b * b - a * cc;
    if (!(disc >= 0.0f)) return 0;
    const float sq = sqrtf(disc);
    const float t0 = (-b - sq) / a;
    if (t0 > tmin && t0 < tmax) { *t = t0; return 1; }
    const float t1 = (-b + sq) / a;
    if (t1 > tmin && t1 < tmax) { *t = t1; return 1; }
    return 0;
}
static inline void consider(hit_rec *best, float t, float u, float v, uint32_t prim, uint32_t inst) {
    const uint64_t id = ((uint64_t)inst << 32) | prim;
    const uint64_t bid = ((uint64_t)best->inst << 32) | best->prim;
    if (!best->hit || t < best->t || (t == best->t && id < bid)) {
        best->hit = 1; best->t = t; best->u = u; best->v = v; best->prim = prim; best->inst = inst;
    }
}
static inline void test_ref(const oracle_scene *sc, uint32_t ref, f3 o, f3 d, float tmin, float tmax, hit_rec *best) {
    float t, u = 0.0f, v = 0.0f;
    if (ref < sc->n_tri) {
        const wtri *tr = &sc->tri[ref];
        f3 oo = o, dd = d;
        /* INSTANCED mode: the ray goes into the instance's object space (what OptiX does at an IAS leaf, src/Global/RendererImpl.cu:174-206),
         * the ray parameter t is common to both spaces */
        if (sc->object_space && !sc->identity[tr->inst]) { oo = xf_point(sc->inv[tr->inst], o); dd = xf_vector(sc->inv[tr->inst], d); }
        if (isect_tri(tr, oo, dd, tmin, tmax, &t, &u, &v)) consider(best, t, u, v, tr->prim, tr->inst);
    } else {
        const wsph *s = &sc->sph[ref - sc->n_tri];
        f3 oo = o, dd = d;
        if (!sc->identity[s->inst]) { oo = xf_point(sc->inv[s->inst], o); dd = xf_vector(sc->inv[s->inst], d); }
        if (isect_sph(s, oo, dd, tmin, tmax, &t)) consider(best, t, 0.0f, 0.0f, s->prim, s->inst);
    }
}

typedef struct { uint64_t rays, node_visits, prim_tests; } trace_counters;

static void bvh8_walk(const uint32_t *nodes, const void *prims_blob, const float *inst_inv, const uint32_t *inst_identity, f3 ow, f3 dw,
                      float tmin, float tmax, int any_hit, hit_rec *out, uint64_t *n_nodes, uint64_t *n_prims, uint64_t *n_empty);

static void closest_hit(const oracle_scene *sc, f3 o, f3 d, float tmin, float tmax, int any_hit,
                        hit_rec *best, trace_counters *cnt) {
    best->hit = 0; best->t = tmax; best->u = best->v = 0.0f; best->prim = best->inst = 0xffffffffu;
    if (cnt) cnt->rays++;
    if (sc->bvh8_nodes) {
        uint64_t nn = 0, np = 0, ne = 0;
        bvh8_walk((const uint32_t *)sc->bvh8_nodes, sc->bvh8_prims, &sc->inv[0][0], (const uint32_t *)sc->identity, o, d, tmin, tmax, any_hit, best, &nn, &np, &ne);
        if (cnt) { cnt->node_visits += nn; cnt->prim_tests += np; }
        return;
    }
    if (sc->brute || sc->n_nodes == 0) {
        for (uint32_t r = 0; r < sc->n_prim; ++r) {
            test_ref(sc, r, o, d, tmin, tmax, best);
            if (cnt) cnt->prim_tests++;
            if (any_hit && best->hit) return;
        }
        return;
    }
    /* BVH2 walk; boxes are padded at build time and the slab test is widened, so the result
     * equals the brute-force loop (checked in tests/test_oracle_cpu.py) */
    const float idx = 1.0f / d.x, idy = 1.0f / d.y, idz = 1.0f / d.z;
    uint32_t stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const bnode *n = &sc->nodes[stack[--sp]];
        if (cnt) cnt->node_visits++;
        float t0 = (n->lo[0] - o.x) * idx, t1 = (n->hi[0] - o.x) * idx;
        float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
        t0 = (n->lo[1] - o.y) * idy; t1 = (n->hi[1] - o.y) * idy;
        tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
        t0 = (n->lo[2] - o.z) * idz; t1 = (n->hi[2] - o.z) * idz;
        tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
        tf *= 1.0000005f;
        const float lim = best->hit ? best->t : tmax;
        if (!(tn <= tf && tf >= tmin && tn <= lim)) {
            /* NaN slabs (0 * inf) must not cull: fall through only when all compares were ordered */
            if (tn == tn && tf == tf) continue;
        }
        if (n->count) {
            for (uint32_t k = 0; k < n->count; ++k) {
                test_ref(sc, sc->refs[n->left + k], o, d, tmin, tmax, best);
                if (cnt) cnt->prim_tests++;
                if (any_hit && best->hit) return;
            }
        } else {
            stack[sp++] = n->left; stack[sp++] = n->left + 1;
        }
    }
}

/* ---- BVH2 build (median split on the largest centroid axis; quality is irrelevant to parity) */
static void ref_bounds(const oracle_scene *sc, uint32_t lo, uint32_t hi, float *blo, float *bhi) {
    for (int a = 0; a < 3; ++a) { blo[a] = INFINITY; bhi[a] = -INFINITY; }
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t r = sc->refs[i];
        for (int a = 0; a < 3; ++a) {
            if (sc->plo[3 * r + a] < blo[a]) blo[a] = sc->plo[3 * r + a];
            if (sc->phi[3 * r + a] > bhi[a]) bhi[a] = sc->phi[3 * r + a];
        }
    }
}
static int g_sort_axis; static const oracle_scene *g_sort_scene;
static int cmp_ref(const void *pa, const void *pb) {
    const uint32_t a = *(const uint32_t *)pa, b = *(const uint32_t *)pb;
    const float ca = g_sort_scene->plo[3 * a + g_sort_axis] + g_sort_scene->phi[3 * a + g_sort_axis];
    const float cb = g_sort_scene->plo[3 * b + g_sort_axis] + g_sort_scene->phi[3 * b + g_sort_axis];
    return (ca > cb) - (ca < cb);
}
static void nth_split(oracle_scene *sc, uint32_t lo, uint32_t hi, int axis) {
    /* quickselect around the median centroid on axis (in place on refs[lo,hi)) */
    uint32_t k = lo + (hi - lo) / 2, l = lo, r = hi - 1;
    while (l < r) {
        const uint32_t pr = sc->refs[l + (r - l) / 2];
        const float pv = sc->plo[3 * pr + axis] + sc->phi[3 * pr + axis];
        uint32_t i = l, j = r;
        while (i <= j) {
            while (sc->plo[3 * sc->refs[i] + axis] + sc->phi[3 * sc->refs[i] + axis] < pv) ++i;
            while (sc->plo[3 * sc->refs[j] + axis] + sc->phi[3 * sc->refs[j] + axis] > pv) { if (j == 0) break; --j; }
            if (i <= j) { uint32_t t = sc->refs[i]; sc->refs[i] = sc->refs[j]; sc->refs[j] = t; ++i; if (j == 0) break; --j; }
        }
        if (k <= j) r = j; else if (k >= i) l = i; else break;
    }
}
static void build_rec(oracle_scene *sc, uint32_t node, uint32_t lo, uint32_t hi) {
    bnode *n = &sc->nodes[node];
    ref_bounds(sc, lo, hi, n->lo, n->hi);
    if (hi - lo <= 4) { n->left = lo; n->count = hi - lo; return; }
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t r = sc->refs[i];
        for (int a = 0; a < 3; ++a) {
            const float c = sc->plo[3 * r + a] + sc->phi[3 * r + a];
            if (c < clo[a]) clo[a] = c;
            if (c > chi[a]) chi[a] = c;
        }
    }
    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    if (hi - lo <= 64) { g_sort_axis = axis; g_sort_scene = sc; qsort(sc->refs + lo, hi - lo, sizeof(uint32_t), cmp_ref); }
    else nth_split(sc, lo, hi, axis);
    const uint32_t mid = lo + (hi - lo) / 2;
    const uint32_t l = sc->n_nodes; sc->n_nodes += 2;
    n->left = l; n->count = 0;
    build_rec(sc, l, lo, mid);
    build_rec(sc, l + 1, mid, hi);
}

/*
 * Two canonical modes, one per shape of the product's acceleration structure -- both are "the reference's IAS over GASes" with the
 * closed OptiX intersector replaced by the canonical one; they differ in WHERE the triangle test is evaluated, hence in rounding:
 *   object_space = 0 (FLATTENED): triangle vertices are transformed to world space once (xf_point), the world ray meets world triangles;
 *   object_space = 1 (INSTANCED): the ray is transformed into the instance's object space (inverse by cofactors in double, rounded once:
 *     exactly what both modes already do for spheres) and meets the object-space triangle (v0, v1 - v0, v2 - v0).  Instances with an
 *     identity transform are the same in both modes.  t, u, v of a hit, and ties between instances, are compared as computed.
 */
oracle_scene *oracle_scene_create_mode(const oracle_instance *inst, int n_inst, int force_brute, int object_space) {
    oracle_scene *sc = (oracle_scene *)calloc(1, sizeof *sc);
    sc->n_inst = n_inst;
    sc->object_space = object_space != 0;
    sc->inst = (oracle_instance *)malloc(sizeof(oracle_instance) * (size_t)n_inst);
    memcpy(sc->inst, inst, sizeof(oracle_instance) * (size_t)n_inst);
    sc->inv = malloc(sizeof(float[12]) * (size_t)n_inst);
    sc->identity = (int *)malloc(sizeof(int) * (size_t)n_inst);
    for (int i = 0; i < n_inst; ++i) {
        sc->identity[i] = is_identity(inst[i].transform);
        invert_affine(inst[i].transform, sc->inv[i]);
        if (inst[i].geometry == GEOM_TRIANGLE) sc->n_tri += inst[i].n_prims; else sc->n_sph += inst[i].n_prims;
    }
    sc->n_prim = sc->n_tri + sc->n_sph;
    sc->tri = (wtri *)malloc(sizeof(wtri) * (sc->n_tri ? sc->n_tri : 1));
    sc->sph = (wsph *)malloc(sizeof(wsph) * (sc->n_sph ? sc->n_sph : 1));
    sc->plo = (float *)malloc(sizeof(float) * 3 * (sc->n_prim ? sc->n_prim : 1));
    sc->phi = (float *)malloc(sizeof(float) * 3 * (sc->n_prim ? sc->n_prim : 1));
    sc->refs = (uint32_t *)malloc(sizeof(uint32_t) * (sc->n_prim ? sc->n_prim : 1));
    uint32_t ti = 0, si = 0;
    float smax = 0.0f;
    for (int i = 0; i < n_inst; ++i) {
        const oracle_instance *in = &inst[i];
        for (uint32_t p = 0; p < in->n_prims; ++p) {
            if (in->geometry == GEOM_TRIANGLE) {
                f3 v[3];
                for (int k = 0; k < 3; ++k) {
                    v[k] = mk3(in->vertices[9 * p + 3 * k], in->vertices[9 * p + 3 * k + 1], in->vertices[9 * p + 3 * k + 2]);
                    if (!sc->identity[i]) v[k] = xf_point(in->transform, v[k]);   /* world-space flattening */
                }
                wtri *t = &sc->tri[ti];
                if (sc->object_space) {      /* the record stays in object space; the bounds below are the world-space triangle's */
                    f3 ov[3];
                    for (int k = 0; k < 3; ++k) ov[k] = mk3(in->vertices[9 * p + 3 * k], in->vertices[9 * p + 3 * k + 1], in->vertices[9 * p + 3 * k + 2]);
                    t->v0 = ov[0]; t->e1 = sub3(ov[1], ov[0]); t->e2 = sub3(ov[2], ov[0]);
                } else { t->v0 = v[0]; t->e1 = sub3(v[1], v[0]); t->e2 = sub3(v[2], v[0]); }
                t->prim = p; t->inst = (uint32_t)i;
                float *lo = &sc->plo[3 * ti], *hi = &sc->phi[3 * ti];
                lo[0] = fminf(v[0].x, fminf(v[1].x, v[2].x)); hi[0] = fmaxf(v[0].x, fmaxf(v[1].x, v[2].x));
                lo[1] = fminf(v[0].y, fminf(v[1].y, v[2].y)); hi[1] = fmaxf(v[0].y, fmaxf(v[1].y, v[2].y));
                lo[2] = fminf(v[0].z, fminf(v[1].z, v[2].z)); hi[2] = fmaxf(v[0].z, fmaxf(v[1].z, v[2].z));
                ++ti;
            } else {
                wsph *s = &sc->sph[si];
                s->c = mk3(in->centers[3 * p], in->centers[3 * p + 1], in->centers[3 * p + 2]);
                s->r = in->radii[p]; s->prim = p; s->inst = (uint32_t)i;
                const uint32_t r = sc->n_tri + si;
                float *lo = &sc->plo[3 * r], *hi = &sc->phi[3 * r];
                const float rr = fabsf(s->r);
                for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
                for (int c = 0; c < 8; ++c) {      /* world AABB of the transformed object-space box */
                    f3 q = mk3(s->c.x + ((c & 1) ? rr : -rr), s->c.y + ((c & 2) ? rr : -rr), s->c.z + ((c & 4) ? rr : -rr));
                    if (!sc->identity[i]) q = xf_point(in->transform, q);
                    lo[0] = fminf(lo[0], q.x); hi[0] = fmaxf(hi[0], q.x);
                    lo[1] = fminf(lo[1], q.y); hi[1] = fmaxf(hi[1], q.y);
                    lo[2] = fminf(lo[2], q.z); hi[2] = fmaxf(hi[2], q.z);
                }
                ++si;
            }
        }
    }
    for (uint32_t r = 0; r < sc->n_prim; ++r) {
        sc->refs[r] = r;
        for (int a = 0; a < 3; ++a) { smax = fmaxf(smax, fabsf(sc->plo[3 * r + a])); smax = fmaxf(smax, fabsf(sc->phi[3 * r + a])); }
    }
    const float pad = 1e-5f * (smax > 1.0f ? smax : 1.0f);
    for (uint32_t r = 0; r < sc->n_prim; ++r)
        for (int a = 0; a < 3; ++a) { sc->plo[3 * r + a] -= pad; sc->phi[3 * r + a] += pad; }
    sc->brute = force_brute || sc->n_prim <= 16;
    if (!sc->brute) {
        sc->nodes = (bnode *)malloc(sizeof(bnode) * (2 * (size_t)sc->n_prim + 2));
        sc->n_nodes = 1;
        build_rec(sc, 0, 0, sc->n_prim);
    }
    return sc;
}
oracle_scene *oracle_scene_create(const oracle_instance *inst, int n_inst, int force_brute) {
    return oracle_scene_create_mode(inst, n_inst, force_brute, 0);
}
/* closest_hit then walks this tree (the product's, for the same instances in the same order); NULL detaches.  The blobs stay the caller's. */
void oracle_scene_attach_bvh8(oracle_scene *sc, const void *nodes, const void *prims) {
    sc->bvh8_nodes = nodes; sc->bvh8_prims = nodes ? prims : NULL;
}
void oracle_scene_destroy(oracle_scene *sc) {
    if (!sc) return;
    free(sc->inst); free(sc->inv); free(sc->identity); free(sc->tri); free(sc->sph);
    free(sc->plo); free(sc->phi); free(sc->refs); free(sc->nodes); free(sc);
}

/* batch trace for the traversal parity tests: same outputs as hrt_trace_rays */
void oracle_trace_rays(const oracle_scene *sc, const float *origins, const float *dirs, uint32_t n,
                       float tmin, float tmax, int any_hit, float *t, float *u, float *v,
                       uint32_t *prim, uint32_t *inst) {
    #pragma omp parallel for schedule(dynamic, 256)
    for (long i = 0; i < (long)n; ++i) {
        hit_rec h;
        closest_hit(sc, mk3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]),
                    mk3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), tmin, tmax, any_hit, &h, NULL);
        t[i] = h.hit ? h.t : tmax; u[i] = h.u; v[i] = h.v;
        prim[i] = h.hit ? h.prim : 0xffffffffu; inst[i] = h.hit ? h.inst : 0xffffffffu;
    }
}

/* ------------------------------------------------------------------------------------ */
/* the device programs, shader/Shader.cu                                                 */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    const oracle_scene *sc;
    rng_state *stateArray;          /* params.stateArray, Shader.cuh:23 */
    f3 backgroundColor;             /* MissParams, Shader.cuh:38-40 */
    uint32_t tid;                   /* idx.y * dim.x + idx.x, Shader.cu:97 */
    trace_counters cnt;
} launch_ctx;

static void ray_trace(launch_ctx *lc, f3 origin, f3 direction, float tMin, float tMax,
                      f4 *payload, f4 *albedo, f4 *normal);

/* __miss__missProgram, Shader.cu:276-287 */
static void miss_program(launch_ctx *lc, f4 *payload, f4 *albedo, f4 *normal) {
    const f3 bg = lc->backgroundColor;
    const f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    payload->x = bg.x; payload->y = bg.y; payload->z = bg.z;       /* payload.w (depth) kept, :285 */
    *albedo = z; *normal = z;                                       /* {}, {} :286 */
}

/* closesthitImpl, Shader.cu:94-242 */
static void closesthit_impl(launch_ctx *lc, const hit_rec *hit, f3 rayOrigin, f3 rayDirection,
                            f4 *payload, f4 *albedo, f4 *normal) {
    const f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    const oracle_instance *hp = &lc->sc->inst[hit->inst];           /* SBT record of the instance, :108 */
    if (payload->w >= (float)RAY_TRACE_DEPTH) {                     /* :102-107 */
        payload->x = 0.0f; payload->y = 0.0f; payload->z = 0.0f;
        *albedo = z; *normal = z;
        return;
    }
    const float t = hit->t;                                         /* optixGetRayTmax, :111 */
    const f3 hitPoint = add3(rayOrigin, muls3(rayDirection, t));    /* :114 */
    const uint32_t primitiveIndex = hit->prim;                      /* :117 */

    f3 normalVector = {0.0f, 0.0f, 0.0f};
    if (hp->geometry == GEOM_SPHERE) {                              /* :122-136 */
        const f3 sphereCenter = mk3(hp->centers[3 * primitiveIndex], hp->centers[3 * primitiveIndex + 1], hp->centers[3 * primitiveIndex + 2]);
        const float sphereRadius = hp->radii[primitiveIndex];
        const f3 outwardNormal = divs3(sub3(hitPoint, sphereCenter), sphereRadius);   /* Q1: object-space centre */
        const int hitFrontFace = dot3(rayDirection, outwardNormal) < 0.0f;
        normalVector = hitFrontFace ? outwardNormal : neg3(outwardNormal);
    } else {                                                        /* :137-155 */
        const float *nn = hp->normals + 9 * (size_t)primitiveIndex;
        const f3 n1 = mk3(nn[0], nn[1], nn[2]), n2 = mk3(nn[3], nn[4], nn[5]), n3 = mk3(nn[6], nn[7], nn[8]);
        const float u = hit->u, v = hit->v;
        const float w = 1.0f - u - v;
        const f3 _normal = add3(add3(muls3(n1, w), muls3(n2, u)), muls3(n3, v));      /* Q2: not transformed, not normalised */
        const int hitFrontFace = dot3(rayDirection, _normal) < 0.0f;
        normalVector = hitFrontFace ? _normal : neg3(_normal);
    }

    f3 reflectDirection = {0.0f, 0.0f, 0.0f};
    f3 _albedo = {0.0f, 0.0f, 0.0f};
    rng_state *state = lc->stateArray + lc->tid;
    if (hp->material == MAT_ROUGH) {                                /* :169-179 */
        reflectDirection = add3(normalVector, random_space_vector(state, 1.0f));
        if (float_value_equals(len2_3(reflectDirection), FLOAT_ZERO_VALUE * FLOAT_ZERO_VALUE))
            reflectDirection = normalVector;
        _albedo = mk3(hp->albedo[0], hp->albedo[1], hp->albedo[2]);
    } else {                                                        /* :180-192 */
        const f3 v = rayDirection, n = normalVector;
        reflectDirection = normalize3(sub3(v, muls3(n, 2.0f * dot3(v, n))));
        if (hp->fuzz > 0.0f)
            reflectDirection = add3(reflectDirection, muls3(random_space_vector(state, 1.0f), hp->fuzz));
        _albedo = mk3(hp->albedo[0], hp->albedo[1], hp->albedo[2]);
    }

    /* :202-213 */
    if (!isfinite(reflectDirection.x) || !isfinite(reflectDirection.y) || !isfinite(reflectDirection.z) ||
        len2_3(reflectDirection) <= FLOAT_ZERO_VALUE * FLOAT_ZERO_VALUE) {
        reflectDirection = normalVector;
        if (len2_3(reflectDirection) <= FLOAT_ZERO_VALUE * FLOAT_ZERO_VALUE ||
            !isfinite(reflectDirection.x) || !isfinite(reflectDirection.y) || !isfinite(reflectDirection.z))
            reflectDirection = mk3(0.0f, 0.0f, 1.0f);
    }

    if (float_value_equals(payload->w, 1.0f)) {                     /* :216-227 (overwritten later: Q3) */
        albedo->x = _albedo.x; albedo->y = _albedo.y; albedo->z = _albedo.z; albedo->w = 1.0f;
        const f3 _n = normalize3(normalVector);
        normal->x = _n.x; normal->y = _n.y; normal->z = _n.z; normal->w = 0.0f;
    }

    f4 result = {payload->x, payload->y, payload->z, payload->w + 1.0f};   /* :230 */
    ray_trace(lc, hitPoint, reflectDirection, FLOAT_ZERO_VALUE, FLOAT_INFINITY_VALUE, &result, albedo, normal);

    result.x *= _albedo.x; result.y *= _albedo.y; result.z *= _albedo.z;   /* :236-238 */
    *payload = result;                                              /* setPayload(result, albedo, normal) :241 */
}

/* (debugging aid, oracle_debug_pixel_path: the rays of one path as the oracle traced them -- 10 floats each: origin, direction, any-hit flag, t,
 * primitive and instance as float bits) */
static _Thread_local float *tl_ray_log; static _Thread_local uint32_t tl_ray_log_n, tl_ray_log_cap;

/* rayTrace + optixTrace dispatch, Shader.cu:46-92 */
static void ray_trace(launch_ctx *lc, f3 origin, f3 direction, float tMin, float tMax,
                      f4 *payload, f4 *albedo, f4 *normal) {
    hit_rec hit;
    /* a hit at depth >= rayTraceDepth returns black whatever it is (Shader.cu:102-107), so the
     * oracle may stop at the first accepted intersection there; the result is identical */
    const int any_hit = payload->w >= (float)RAY_TRACE_DEPTH;
    closest_hit(lc->sc, origin, direction, tMin, tMax, any_hit, &hit, &lc->cnt);
    if (tl_ray_log && tl_ray_log_n < tl_ray_log_cap) {
        float *q = tl_ray_log + 10 * (size_t)tl_ray_log_n++;
        q[0] = origin.x; q[1] = origin.y; q[2] = origin.z; q[3] = direction.x; q[4] = direction.y; q[5] = direction.z;
        q[6] = (float)any_hit; q[7] = hit.hit ? hit.t : -1.0f; memcpy(q + 8, &hit.prim, 4); memcpy(q + 9, &hit.inst, 4);
    }
    if (hit.hit) closesthit_impl(lc, &hit, origin, direction, payload, albedo, normal);
    else miss_program(lc, payload, albedo, normal);
}

/* __raygen__raygenProgram, Shader.cu:246-273; returns the linear result before colorToFloat4 */
static f4 raygen_program(launch_ctx *lc, uint32_t ix, uint32_t iy, uint32_t dimx, uint32_t dimy,
                         const float *cam, f4 *albedo_out, f4 *normal_out) {
    const float ndcx = (((float)ix + 0.5f) / (float)dimx) * 2.0f - 1.0f;     /* :250 */
    const float ndcy = (((float)iy + 0.5f) / (float)dimy) * 2.0f - 1.0f;     /* :251 */
    const f3 origin = mk3(cam[0], cam[1], cam[2]);
    const f3 U = mk3(cam[3], cam[4], cam[5]), V = mk3(cam[6], cam[7], cam[8]), W = mk3(cam[9], cam[10], cam[11]);
    const float aspect = (float)dimx / (float)dimy;                          /* :260 (data->width/height == dim) */
    const f3 direction = normalize3(add3(add3(muls3(U, ndcx * aspect), muls3(V, ndcy)), W));   /* :261 */
    f4 result = {0.0f, 0.0f, 0.0f, 1.0f}, albedo = {0, 0, 0, 0}, normal = {0, 0, 0, 0};          /* :264 */
    lc->tid = iy * dimx + ix;
    ray_trace(lc, origin, direction, FLOAT_ZERO_VALUE, FLOAT_INFINITY_VALUE, &result, &albedo, &normal);
    *albedo_out = albedo; *normal_out = normal;
    return result;
}

/*
 * One render call: spp successive launches of the reference's frame on the persistent
 * per-pixel RNG streams (Q8), colour = colorToFloat4(mean of the linear results); spp = 1 is
 * exactly the reference frame (Shader.cu:270).  rows: optional list of row indices to render
 * (NULL = all).  Outputs are W*H float4 arrays; untouched outside the rendered rows.
 */
void oracle_render(const oracle_scene *sc, const float *cam12, uint32_t width, uint32_t height,
                   void *states, const float *bg3, uint32_t spp,
                   const uint32_t *rows, uint32_t n_rows,
                   float *color, float *albedo, float *normal, float *linear,
                   uint64_t *out_counters /* rays, node_visits, prim_tests */) {
    oracle_init();
    const long nr = rows ? (long)n_rows : (long)height;
    uint64_t tot_rays = 0, tot_nodes = 0, tot_prims = 0;
    #pragma omp parallel for schedule(dynamic, 1) reduction(+:tot_rays, tot_nodes, tot_prims)
    for (long ri = 0; ri < nr; ++ri) {
        const uint32_t y = rows ? rows[ri] : (uint32_t)ri;
        launch_ctx lc; memset(&lc, 0, sizeof lc);
        lc.sc = sc; lc.stateArray = (rng_state *)states;
        lc.backgroundColor = mk3(bg3[0], bg3[1], bg3[2]);
        for (uint32_t x = 0; x < width; ++x) {
            f4 a = {0, 0, 0, 0}, n = {0, 0, 0, 0};
            f3 sum = {0.0f, 0.0f, 0.0f};
            for (uint32_t s = 0; s < spp; ++s) {
                const f4 r = raygen_program(&lc, x, y, width, height, cam12, &a, &n);
                if (s == 0) sum = mk3(r.x, r.y, r.z); else sum = add3(sum, mk3(r.x, r.y, r.z));
            }
            if (spp > 1) sum = divs3(sum, (float)spp);
            const size_t p = (size_t)y * width + x;
            if (linear) { linear[4 * p] = sum.x; linear[4 * p + 1] = sum.y; linear[4 * p + 2] = sum.z; linear[4 * p + 3] = 1.0f; }
            if (color) { const float rgb[3] = {sum.x, sum.y, sum.z}; oracle_color_to_float4(rgb, &color[4 * p]); }
            if (albedo) { albedo[4 * p] = a.x; albedo[4 * p + 1] = a.y; albedo[4 * p + 2] = a.z; albedo[4 * p + 3] = a.w; }
            if (normal) { normal[4 * p] = n.x; normal[4 * p + 1] = n.y; normal[4 * p + 2] = n.z; normal[4 * p + 3] = n.w; }
        }
        tot_rays += lc.cnt.rays; tot_nodes += lc.cnt.node_visits; tot_prims += lc.cnt.prim_tests;
    }
    if (out_counters) { out_counters[0] = tot_rays; out_counters[1] = tot_nodes; out_counters[2] = tot_prims; }
}

/* One sample of one pixel, its rays logged (10 floats per ray, see tl_ray_log); the pixel's RNG state advances as in a render.  Returns the
 * number of rays; result3 = the sample's linear radiance. */
uint32_t oracle_debug_pixel_path(const oracle_scene *sc, const float *cam12, uint32_t width, uint32_t height, void *states, const float *bg3,
                                 uint32_t x, uint32_t y, float *log10, uint32_t cap, float *result3) {
    oracle_init();
    launch_ctx lc; memset(&lc, 0, sizeof lc);
    lc.sc = sc; lc.stateArray = (rng_state *)states; lc.backgroundColor = mk3(bg3[0], bg3[1], bg3[2]);
    f4 a, n;
    tl_ray_log = log10; tl_ray_log_n = 0; tl_ray_log_cap = cap;
    const f4 r = raygen_program(&lc, x, y, width, height, cam12, &a, &n);
    tl_ray_log = NULL;
    if (result3) { result3[0] = r.x; result3[1] = r.y; result3[2] = r.z; }
    return tl_ray_log_n;
}

/* convertFloat4ToUchar4Kernel, src/Global/RendererImpl.cu:672-678 */
void oracle_to_rgba8(const float *src4, unsigned char *dst4, uint32_t width, uint32_t height) {
    const long n = (long)width * height;
    for (long i = 0; i < n; ++i) oracle_color_to_uchar4(&src4[4 * i], &dst4[4 * i]);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------ */
/* CPU walk of the PRODUCT's packed BVH8 blob (layout: nvidia-optix-ray-tracer_amd/csrc/bvh8.h). */
/* Test infrastructure: lets the CPU suite check the builder's node encoding and gives   */
/* the per-ray node-visit / primitive-test counts of SURVEY.md 8(d) on the same bytes.   */
/* Same canonical intersector as above; the box test mirrors the kernel's (fmaf)          */
/*, traversal order = octant order, no triangle postponing.                    */
/* ------------------------------------------------------------------------------------ */
typedef struct { float a[3]; uint32_t prim; float b[3]; uint32_t inst; float c[3]; uint32_t kind; } prim48;

static inline float safe_rcp_dir(float d) {
    const float lim = 1e-20f;
    const float dd = fabsf(d) < lim ? (d < 0.0f ? -lim : lim) : d;      /* the sign that `d < 0` sees: -0.0 is positive (csrc/trav_common.h safe_rcp_dir) */
    return 1.0f / dd;
}

/* (a what-if of tools/tree_quality.py --line-mates: node visits whose neighbour in the array, index ^ 1, the same ray had visited
 * before -- what a 64-byte node, two to a 128-byte line, would save in line fetches) */
static _Thread_local uint64_t tl_line_mates;

/* one ray through a packed BVH8 (one- or two-level): the canonical closest hit (or any hit), and what the walk cost */
static void bvh8_walk(const uint32_t *nodes, const void *prims_blob, const float *inst_inv, const uint32_t *inst_identity, f3 ow, f3 dw,
                      float tmin, float tmax, int any_hit, hit_rec *out, uint64_t *n_nodes, uint64_t *n_prims, uint64_t *n_empty) {
    const prim48 *prims = (const prim48 *)prims_blob;
    uint64_t tot_nodes = 0, tot_prims = 0, tot_empty = 0;
    /* the ray in the space being walked: world, or -- below a transform node of a two-level tree -- the instance's object space */
    f3 o = ow, d = dw;
    float idx = safe_rcp_dir(d.x), idy = safe_rcp_dir(d.y), idz = safe_rcp_dir(d.z);
    uint32_t oct = (d.x < 0.0f ? 4u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 1u : 0u);
    uint32_t oct_inv = 7u - oct;
    uint32_t cur_inst = 0xffffffffu;         /* the instance whose BLAS is being walked (two-level trees) */
    hit_rec best; best.hit = 0; best.t = tmax; best.u = best.v = 0.0f; best.prim = best.inst = 0xffffffffu;
    uint32_t stack_x[64], stack_y[64]; int sp = 0;
    uint32_t cur_x = 0, cur_y = 0x80000000u;
    uint32_t seen[96]; int n_seen = 0;
    int done = 0;
    while (!done) {
        uint32_t tri_x = 0, tri_y = 0;
        if (cur_y > 0x00ffffffu) {
            const uint32_t hits_imask = cur_y;
            uint32_t bit = 31; while (!((hits_imask >> bit) & 1u)) --bit;
            cur_y &= ~(1u << bit);
            if (cur_y > 0x00ffffffu) { stack_x[sp] = cur_x; stack_y[sp] = cur_y; ++sp; }
            const uint32_t slot_index = (bit - 24u) ^ oct_inv;
            const uint32_t rel = (uint32_t)__builtin_popcount(hits_imask & ~(0xffffffffu << slot_index));
            const uint32_t *nd = nodes + 20 * (size_t)(cur_x + rel);
            ++tot_nodes;
            for (int k = 0; k < n_seen; ++k) if (seen[k] == ((cur_x + rel) ^ 1u)) { ++tl_line_mates; break; }
            if (n_seen < 96) seen[n_seen++] = cur_x + rel;
            const uint32_t e_imask = nd[3];
            if (e_imask == 0u) {
                /* a transform node (two-level trees, csrc/bvh8.h): word 4 = root of the instance's BLAS, word 5 = instance,
                 * word 6 = identity flag, words 8..19 = world -> object.  The ray goes into object space; a marker on the stack
                 * brings the world ray back when the BLAS has been walked. */
                f3 oo = ow, od = dw;
                if (!nd[6]) { const float *m = (const float *)(nd + 8); oo = xf_point(m, ow); od = xf_vector(m, dw); }
                {   /* words 0-2, 7: the BLAS's bounding sphere in object space (negative radius: none); a ray that misses it does not go in.
                     * Culling only -- the same conservative test as the kernel's (csrc/fused.hip), so that the visit counts agree. */
                    float c[3], R; memcpy(c, nd, 12); memcpy(&R, nd + 7, 4);
                    const float cx = oo.x - c[0], cy = oo.y - c[1], cz = oo.z - c[2];
                    const float cc = fmaf(cx, cx, fmaf(cy, cy, cz * cz)), aa = fmaf(od.x, od.x, fmaf(od.y, od.y, od.z * od.z)), b = fmaf(cx, od.x, fmaf(cy, od.y, cz * od.z));
                    const float R2 = R * R * 1.0001f, ca = cc * aa;
                    /* (not entered: the siblings still to visit were pushed above, the copy in hand goes) */
                    if (R >= 0.0f && (fmaf(-b, b, ca) > fmaf(R2, aa, 4e-6f * ca) || (b > 0.0f && cc > fmaf(4e-6f, cc, R2)))) { cur_y = 0u; continue; }
                }
                stack_x[sp] = 0xffffffffu; stack_y[sp] = 0u; ++sp;
                cur_inst = nd[5];
                o = oo; d = od;
                idx = safe_rcp_dir(d.x); idy = safe_rcp_dir(d.y); idz = safe_rcp_dir(d.z);
                oct = (d.x < 0.0f ? 4u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 1u : 0u); oct_inv = 7u - oct;
                cur_x = nd[4]; cur_y = 0x01000000u;          /* one child: the root (no inner-mask bits: index = base) */
                continue;
            }
            float p[3]; memcpy(p, nd, 12);
            uint32_t eb; float sx, sy, sz;
            eb = (e_imask & 0xffu) << 23; memcpy(&sx, &eb, 4);
            eb = ((e_imask >> 8) & 0xffu) << 23; memcpy(&sy, &eb, 4);
            eb = ((e_imask >> 16) & 0xffu) << 23; memcpy(&sz, &eb, 4);
            const float aix = sx * idx, aiy = sy * idy, aiz = sz * idz;
            const float aox = (p[0] - o.x) * idx, aoy = (p[1] - o.y) * idy, aoz = (p[2] - o.z) * idz;
            const uint8_t *meta = (const uint8_t *)(nd + 6);
            const uint8_t *q = (const uint8_t *)(nd + 8);        /* qlo[3][8], qhi[3][8] */
            uint32_t hitmask = 0;
            const float bt = best.hit ? best.t : tmax;
            for (int s = 0; s < 8; ++s) {
                const uint32_t m = meta[s];
                const int is_inner = ((m & (m << 1)) & 0x10u) != 0;
                const uint32_t bit_index = (m ^ (is_inner ? oct_inv : 0u)) & 0x1fu;
                const uint32_t child_bits = (m >> 5) & 7u;
                const float qlx = q[0 * 8 + s], qly = q[1 * 8 + s], qlz = q[2 * 8 + s];
                const float qhx = q[24 + 0 * 8 + s], qhy = q[24 + 1 * 8 + s], qhz = q[24 + 2 * 8 + s];
                const float tnx = fmaf(d.x < 0.0f ? qhx : qlx, aix, aox), tfx = fmaf(d.x < 0.0f ? qlx : qhx, aix, aox);
                const float tny = fmaf(d.y < 0.0f ? qhy : qly, aiy, aoy), tfy = fmaf(d.y < 0.0f ? qly : qhy, aiy, aoy);
                const float tnz = fmaf(d.z < 0.0f ? qhz : qlz, aiz, aoz), tfz = fmaf(d.z < 0.0f ? qlz : qhz, aiz, aoz);
                const float tlo = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
                const float thi = fminf(fminf(tfx, tfy), fminf(tfz, bt));
                if (tlo <= thi) hitmask |= child_bits << bit_index;
            }
            cur_x = nd[4]; cur_y = (hitmask & 0xff000000u) | (e_imask >> 24);
            tri_x = nd[5]; tri_y = hitmask & 0x00ffffffu;
            if (hitmask == 0u) ++tot_empty;
        }
        while (tri_y) {
            const uint32_t k = (uint32_t)__builtin_ctz(tri_y);
            tri_y &= tri_y - 1u;
            ++tot_prims;
            const prim48 *pr = &prims[tri_x + k];
            const uint32_t pinst = cur_inst != 0xffffffffu ? cur_inst : pr->inst;      /* a shared BLAS does not know who instances it */
            float t, u = 0.0f, v = 0.0f; int h = 0;
            if (pr->kind == 1u) {
                wsph s; s.c = mk3(pr->a[0], pr->a[1], pr->a[2]); s.r = pr->b[0];
                f3 oo = o, dd = d;
                if (cur_inst == 0xffffffffu && inst_identity && !inst_identity[pr->inst]) { oo = xf_point(inst_inv + 12 * (size_t)pr->inst, o); dd = xf_vector(inst_inv + 12 * (size_t)pr->inst, d); }
                h = isect_sph(&s, oo, dd, tmin, tmax, &t);
            } else {
                wtri tr; tr.v0 = mk3(pr->a[0], pr->a[1], pr->a[2]); tr.e1 = mk3(pr->b[0], pr->b[1], pr->b[2]); tr.e2 = mk3(pr->c[0], pr->c[1], pr->c[2]);
                h = isect_tri(&tr, o, d, tmin, tmax, &t, &u, &v);
            }
            if (h) { consider(&best, t, u, v, pr->prim, pinst); if (any_hit) { done = 1; break; } }
        }
        while (!done && cur_y <= 0x00ffffffu) {
            if (sp == 0) { done = 1; break; }
            --sp; cur_x = stack_x[sp]; cur_y = stack_y[sp];
            if (cur_x == 0xffffffffu && cur_y == 0u) {       /* the marker: back to world space */
                o = ow; d = dw; cur_inst = 0xffffffffu;
                idx = safe_rcp_dir(d.x); idy = safe_rcp_dir(d.y); idz = safe_rcp_dir(d.z);
                oct = (d.x < 0.0f ? 4u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 1u : 0u); oct_inv = 7u - oct;
            }
        }
    }
    *out = best; *n_nodes += tot_nodes; *n_prims += tot_prims; *n_empty += tot_empty;
}

void oracle_bvh8_trace(const void *nodes_blob, const void *prims_blob,
                       const float *inst_inv /* 12 per instance or NULL */, const uint32_t *inst_identity,
                       const float *origins, const float *dirs, uint32_t n_rays,
                       float tmin, float tmax, int any_hit,
                       float *t_out, float *u_out, float *v_out, uint32_t *prim_out, uint32_t *inst_out,
                       uint64_t *out_counters /* [4]: node visits, prim tests, node visits that found nothing to enter or test, line-mate visits */, uint32_t *per_ray_nodes /* or NULL */) {
    uint64_t tot_nodes = 0, tot_prims = 0, tot_empty = 0, tot_mates = 0;
    #pragma omp parallel for schedule(dynamic, 256) reduction(+:tot_nodes, tot_prims, tot_empty, tot_mates)
    for (long ri = 0; ri < (long)n_rays; ++ri) {
        hit_rec best; uint64_t nn = 0, np = 0, ne = 0;
        tl_line_mates = 0;
        bvh8_walk((const uint32_t *)nodes_blob, prims_blob, inst_inv, inst_identity, mk3(origins[3 * ri], origins[3 * ri + 1], origins[3 * ri + 2]),
                  mk3(dirs[3 * ri], dirs[3 * ri + 1], dirs[3 * ri + 2]), tmin, tmax, any_hit, &best, &nn, &np, &ne);
        tot_nodes += nn; tot_prims += np; tot_empty += ne; tot_mates += tl_line_mates;
        t_out[ri] = best.hit ? best.t : tmax; u_out[ri] = best.u; v_out[ri] = best.v;
        prim_out[ri] = best.hit ? best.prim : 0xffffffffu; inst_out[ri] = best.hit ? best.inst : 0xffffffffu;
        if (per_ray_nodes) per_ray_nodes[ri] = (uint32_t)nn;
    }
    if (out_counters) { out_counters[0] = tot_nodes; out_counters[1] = tot_prims; out_counters[2] = tot_empty; out_counters[3] = tot_mates; }
}

/* ------------------------------------------------------------------------------------------
 * Time-mode pose pipeline: src/Global/RendererTime.cu:436-472 per particle --
 * slerp :296-340, quatToEuler :343-370, constructTransformMatrix include/Global/DeviceFunctions.cuh:43-148.
 * acosf, sinf, cosf, asinf, atan2f where the reference calls the float overloads: pinned as correctly rounded (below).
 * Reference behaviour kept: the aggregate returns of slerp fill a float4 positionally with the
 * {w, x, y, z} expressions (so they land in .x .y .z .w), and the Z*Y*X Euler angles are composed as Rx*Ry*Rz.
 * ------------------------------------------------------------------------------------------ */
/* sinf / cosf / acosf / asinf / atan2f of slerp, quatToEuler and constructRotateMatrix.  The reference calls its platform's
 * float functions (CUDA's on the device build, the host libm's in the frame loop), third-party arithmetic that is not in the
 * tree and differs between platforms by an ULP -- enough to flip hits in a path tracer.  They are pinned here, like powf
 * above, as the CORRECTLY ROUNDED float of the exact value: libm's double function rounded to float, and where that double
 * lies within 2^-45 (relative) of the midpoint of two adjacent floats the decision is taken in __float128 (libquadmath).
 * The product's csrc/cr_trig.h reaches the same definition with double-double arithmetic; the float libm functions stay
 * available as a tolerance cross-check (oracle_trig_libm). */
static int near_float_midpoint(double r) {
    const float c = (float)r;
    uint32_t cb; memcpy(&cb, &c, 4);
    const uint32_t ub = cb + 1u, db = cb - 1u;
    float cu, cd; memcpy(&cu, &ub, 4); memcpy(&cd, &db, 4);
    const double mu = 0.5 * ((double)c + (double)cu), md = 0.5 * ((double)c + (double)cd), tol = fabs(r) * 0x1p-45;
    return fabs(r - mu) < tol || fabs(r - md) < tol;
}
static float nearest_float_q(__float128 q, double r) {         /* the float nearest q, among (float)r and its two neighbours */
    const float c = (float)r;
    uint32_t cb; memcpy(&cb, &c, 4);
    const uint32_t ub = cb + 1u, db = cb - 1u;
    float cu, cd; memcpy(&cu, &ub, 4); memcpy(&cd, &db, 4);
    float best = c; __float128 bd = fabsq(q - (__float128)c);
    if (cu == cu && fabsq(q - (__float128)cu) < bd) { bd = fabsq(q - (__float128)cu); best = cu; }
    if (cd == cd && fabsq(q - (__float128)cd) < bd) { bd = fabsq(q - (__float128)cd); best = cd; }
    return best;
}
static int trig_settled(double r) { return r == 0.0 || !(fabs(r) <= 1.0e300) || !near_float_midpoint(r); }
static float sinf_cr(float x) { const double r = sin((double)x); return trig_settled(r) ? (float)r : nearest_float_q(sinq((__float128)x), r); }
static float cosf_cr(float x) { const double r = cos((double)x); return trig_settled(r) ? (float)r : nearest_float_q(cosq((__float128)x), r); }
static float acosf_cr(float x) { const double r = acos((double)x); return trig_settled(r) ? (float)r : nearest_float_q(acosq((__float128)x), r); }
static float asinf_cr(float x) { const double r = asin((double)x); return trig_settled(r) ? (float)r : nearest_float_q(asinq((__float128)x), r); }
static float atan2f_cr(float y, float x) {
    const double r = atan2((double)y, (double)x);
    return trig_settled(r) ? (float)r : nearest_float_q(atan2q((__float128)y, (__float128)x), r);
}
/* which: 0 sin, 1 cos, 2 acos, 3 asin, 4 atan2(x = y-argument, y = x-argument ... see below).  force_exact: decide every value
 * in __float128 (what the pin means, without the shortcut) */
static float trig_one(int which, float a, float b, int force_exact) {
    if (!force_exact) {
        switch (which) { case 0: return sinf_cr(a); case 1: return cosf_cr(a); case 2: return acosf_cr(a); case 3: return asinf_cr(a); default: return atan2f_cr(a, b); }
    }
    double r; __float128 q;
    switch (which) {
        case 0: r = sin((double)a); q = sinq((__float128)a); break;
        case 1: r = cos((double)a); q = cosq((__float128)a); break;
        case 2: r = acos((double)a); q = acosq((__float128)a); break;
        case 3: r = asin((double)a); q = asinq((__float128)a); break;
        default: r = atan2((double)a, (double)b); q = atan2q((__float128)a, (__float128)b); break;
    }
    if (r == 0.0 || !(fabs(r) <= 1.0e300)) return (float)r;
    return nearest_float_q(q, r);
}
/* n values: out[i] = f(a[i]) (atan2: f(a[i], b[i]), a = the y argument) */
void oracle_trig(int which, const float *a, const float *b, uint64_t n, int force_exact, float *out) {
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; ++i) out[i] = trig_one(which, a[i], b ? b[i] : 0.0f, force_exact);
}
/* ... over the floats whose bit patterns are first, first + stride, ... (count of them); one-argument functions */
void oracle_trig_bits(int which, uint32_t first, uint32_t stride, uint64_t count, int force_exact, float *out) {
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < count; ++i) {
        const uint32_t bits = first + (uint32_t)(i * stride);
        float x; memcpy(&x, &bits, 4);
        out[i] = trig_one(which, x, 0.0f, force_exact);
    }
}
/* the platform's float libm (tolerance cross-check only) */
void oracle_trig_libm(int which, const float *a, const float *b, uint64_t n, float *out) {
    for (uint64_t i = 0; i < n; ++i)
        switch (which) {
            case 0: out[i] = sinf(a[i]); break; case 1: out[i] = cosf(a[i]); break; case 2: out[i] = acosf(a[i]); break;
            case 3: out[i] = asinf(a[i]); break; default: out[i] = atan2f(a[i], b[i]); break;
        }
}

#define ORACLE_PI 3.1415926f                      /* PI, DeviceFunctions.cuh:19 */
typedef struct { float x, y, z, w; } quat4;
typedef struct { float m[4][4]; } mat4;

static mat4 m4_mul(const mat4 *a, const mat4 *b) {          /* Matrix::operator*, :48-61 */
    mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float sum = 0.0f;
            for (int n = 0; n < 4; ++n) sum += a->m[i][n] * b->m[n][j];
            r.m[i][j] = sum;
        }
    return r;
}
static mat4 m4_identity(void) {
    mat4 r; memset(&r, 0, sizeof r);
    r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0f;
    return r;
}
static mat4 m4_rotation(float degree, int axis) {           /* constructRotateMatrix, :88-123 */
    const float theta = degree * ORACLE_PI / 180.0f;        /* degreeToRadian :27-29 */
    const float c = cosf_cr(theta), s = sinf_cr(theta);
    mat4 r = m4_identity();
    if (axis == 0) { r.m[1][1] = c; r.m[1][2] = -s; r.m[2][1] = s; r.m[2][2] = c; }
    else if (axis == 1) { r.m[0][0] = c; r.m[0][2] = s; r.m[2][0] = -s; r.m[2][2] = c; }
    else { r.m[0][0] = c; r.m[0][1] = -s; r.m[1][0] = s; r.m[1][1] = c; }
    return r;
}
static quat4 pose_slerp(quat4 q1, quat4 q2, float t) {      /* RendererTime.cu:296-340 */
    float dot = q1.w * q2.w + q1.x * q2.x + q1.y * q2.y + q1.z * q2.z;
    if (dot < 0.0f) { q2.w = -q2.w; q2.x = -q2.x; q2.y = -q2.y; q2.z = -q2.z; dot = -dot; }
    if (dot > 0.9995f) {
        quat4 r = {q1.w + t * (q2.w - q1.w), q1.x + t * (q2.x - q1.x), q1.y + t * (q2.y - q1.y), q1.z + t * (q2.z - q1.z)};
        const float mag = sqrtf(r.w * r.w + r.x * r.x + r.y * r.y + r.z * r.z);
        if (mag > 0.0f) { r.w /= mag; r.x /= mag; r.y /= mag; r.z /= mag; }
        return r;
    }
    {
        const float theta_0 = acosf_cr(dot);
        const float theta = theta_0 * t;
        const float sin_theta = sinf_cr(theta);
        const float sin_theta_0 = sinf_cr(theta_0);
        const float s0 = cosf_cr(theta) - dot * sin_theta / sin_theta_0;
        const float s1 = sin_theta / sin_theta_0;
        quat4 r = {(s0 * q1.w) + (s1 * q2.w), (s0 * q1.x) + (s1 * q2.x), (s0 * q1.y) + (s1 * q2.y), (s0 * q1.z) + (s1 * q2.z)};
        return r;
    }
}
static void pose_quat_to_euler(quat4 q, float *deg) {       /* RendererTime.cu:343-370 */
    const float sinr_cosp = 2.0f * (q.w * q.x + q.y * q.z);
    const float cosr_cosp = 1.0f - 2.0f * (q.x * q.x + q.y * q.y);
    const float roll = atan2f_cr(sinr_cosp, cosr_cosp);
    const float sinp = 2.0f * (q.w * q.y - q.z * q.x);
    const float pitch = fabsf(sinp) >= 1.0f ? copysignf(ORACLE_PI / 2.0f, sinp) : asinf_cr(sinp);
    const float siny_cosp = 2.0f * (q.w * q.z + q.x * q.y);
    const float cosy_cosp = 1.0f - 2.0f * (q.y * q.y + q.z * q.z);
    const float yaw = atan2f_cr(siny_cosp, cosy_cosp);
    deg[0] = roll * 180.0f / ORACLE_PI; deg[1] = pitch * 180.0f / ORACLE_PI; deg[2] = yaw * 180.0f / ORACLE_PI;   /* radianToDegree :30-32 */
}

/* constructTransformMatrix(shift, rotate [degrees], scale) -> 12 floats, DeviceFunctions.cuh:133-148 */
void oracle_construct_transform(const float *shift, const float *rotate_deg, const float *scale, float *out12) {
    mat4 s = m4_identity(), sc = m4_identity();
    s.m[0][3] = shift[0]; s.m[1][3] = shift[1]; s.m[2][3] = shift[2];
    sc.m[0][0] = scale[0]; sc.m[1][1] = scale[1]; sc.m[2][2] = scale[2];
    const mat4 rx = m4_rotation(rotate_deg[0], 0), ry = m4_rotation(rotate_deg[1], 1), rz = m4_rotation(rotate_deg[2], 2);
    const mat4 rxy = m4_mul(&rx, &ry), r = m4_mul(&rxy, &rz);
    const mat4 sr = m4_mul(&s, &r), t = m4_mul(&sr, &sc);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) out12[4 * i + j] = t.m[i][j];
}

/* quat4 in / out as 4 floats in float4 field order (x, y, z, w) */
void oracle_slerp(const float *q1, const float *q2, float t, float *out4) {
    const quat4 a = {q1[0], q1[1], q1[2], q1[3]}, b = {q2[0], q2[1], q2[2], q2[3]};
    const quat4 r = pose_slerp(a, b, t);
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}
void oracle_quat_to_euler(const float *q, float *deg3) {
    const quat4 a = {q[0], q[1], q[2], q[3]};
    pose_quat_to_euler(a, deg3);
}

/* The frame loop body for n particles.  states: 12 floats per particle laid out as HrtParticleState
 * {quat.xyzw, position.xyz, velocity.xyz, pad, pad}.  out: 12 floats per particle. */
void oracle_pose_transforms(const float *current, const float *next, uint32_t n, float duration, uint32_t frame,
                            uint32_t frame_count, const float *offset, const float *scale, float *out) {
    const float fcount = (float)frame_count, fframe = (float)frame;
    const float factor = frame_count > 1u ? fframe / (float)(frame_count - 1u) : 1.0f;
    for (uint32_t i = 0; i < n; ++i) {
        const float *c = current + 12 * (size_t)i, *nx = next + 12 * (size_t)i;
        float shift[3], deg[3];
        for (int k = 0; k < 3; ++k) {
            const float total = c[7 + k] * duration;
            const float per_frame = total / fcount;
            shift[k] = offset[k] + (c[4 + k] + per_frame * fframe);
        }
        const quat4 qc = {c[0], c[1], c[2], c[3]}, qn = {nx[0], nx[1], nx[2], nx[3]};
        pose_quat_to_euler(pose_slerp(qc, qn, factor), deg);
        oracle_construct_transform(shift, deg, scale, out + 12 * (size_t)i);
    }
}
