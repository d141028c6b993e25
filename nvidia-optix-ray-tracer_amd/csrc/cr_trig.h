// cr_trig.h -- sinf / cosf / acosf / asinf / atan2f of the Time-mode pose pipeline (slerp, quatToEuler:
// src/Global/RendererTime.cu:296-370; constructRotateMatrix: include/Global/DeviceFunctions.cuh:88-123), pinned as the
// CORRECTLY ROUNDED float of the exact value, for every float argument.
//
// CUDA's / the host libm's float transcendentals are third-party arithmetic that differs between platforms by an ULP; in a
// path tracer one ULP in an instance transform flips hits.  "The float nearest the exact value" is the one definition the
// oracle (libm double + __float128 near rounding boundaries, oracle/oracle.c) and these kernels can both reproduce bit for
// bit -- the same treatment csrc/srgb_pow.h gives the shader's powf.
//   fast path  the platform's DOUBLE function (ocml on the device, libm on a host compile; a few ULP of double at worst),
//              rounded to float -- taken unless that double lies within 2^-44 (relative) of the midpoint of two adjacent
//              floats.  Which double library is used does not matter: away from a midpoint every sufficiently accurate
//              double rounds to the same float.
//   slow path  (one call in ~500 000) the function in double-double arithmetic (+, -, *, /, sqrt, fma on doubles; error
//              < 2^-95), rounded once: Taylor series for sin / cos after an exact Payne-Hanek reduction of the float
//              argument (448 bits of 2/pi), and one Newton step in double-double from the fast path's value for the
//              inverse functions (asin / acos as atan2 of x and sqrt((1 - x)(1 + x))).
// tests/test_oracle_cpu.py sweeps a host compile of this header against the oracle (fast and forced-slow paths);
// tests/test_gpu_parity.py does the same on the GPU through hrt_debug_trig, over all 2^32 floats for sin and cos.
#pragma once
#include <math.h>
#include "srgb_pow.h"

namespace hrt {
namespace crt {

using srgbpow::DD;
using srgbpow::bits_to_float;
using srgbpow::dd_add;
using srgbpow::dd_div_d;
using srgbpow::dd_mul;
using srgbpow::dd_mul_d;
using srgbpow::dd_renorm;
using srgbpow::dd_two_prod;
using srgbpow::dd_two_sum;
using srgbpow::float_to_bits;

HRT_POW_HD DD dd(double hi, double lo = 0.0) { DD r; r.hi = hi; r.lo = lo; return r; }
HRT_POW_HD DD dd_neg(DD a) { return dd(-a.hi, -a.lo); }
HRT_POW_HD DD dd_sub(DD a, DD b) { return dd_add(a, dd_neg(b)); }
HRT_POW_HD DD dd_add_d(DD a, double b) { DD s = dd_two_sum(a.hi, b); s.lo += a.lo; return dd_renorm(s.hi, s.lo); }
HRT_POW_HD DD dd_div(DD a, DD b) {
    const double q1 = a.hi / b.hi;
    DD r = dd_sub(a, dd_mul_d(b, q1));
    const double q2 = r.hi / b.hi;
    r = dd_sub(r, dd_mul_d(b, q2));
    const double q3 = r.hi / b.hi;
    return dd_add_d(dd_renorm(q1, q2), q3);
}
HRT_POW_HD DD dd_sqrt(DD a) {
    if (!(a.hi > 0.0)) return dd(0.0);
    const double x = sqrt(a.hi);
    const DD d = dd_sub(a, dd_two_prod(x, x));
    return dd_renorm(x, d.hi / (2.0 * x));
}

// pi/2 as three doubles (160 bits)
constexpr double kPio2a = 0x1.921fb54442d18p+0, kPio2b = 0x1.1a62633145c07p-54, kPio2c = -0x1.f1976b7ed8fbcp-110;

// sin and cos of a double-double |r| <= ~0.8 by their Taylor series (18 terms each: the first neglected one is < 2^-140)
HRT_POW_HD void sincos_series(DD r, DD *s, DD *c) {
    const DD r2 = dd_mul(r, r);
    DD term = r, sum = r;
    for (int k = 1; k <= 17; ++k) {
        term = dd_neg(dd_div_d(dd_mul(term, r2), (double)((2 * k) * (2 * k + 1))));
        sum = dd_add(sum, term);
    }
    *s = sum;
    term = dd(1.0); sum = dd(1.0);
    for (int k = 1; k <= 17; ++k) {
        term = dd_neg(dd_div_d(dd_mul(term, r2), (double)((2 * k - 1) * (2 * k))));
        sum = dd_add(sum, term);
    }
    *c = sum;
}
// (sin, cos)(r + k pi/2) from (sin, cos)(r)
HRT_POW_HD void rotate_quadrant(int k, DD *s, DD *c) {
    const DD s0 = *s, c0 = *c;
    switch (k & 3) {
        case 1: *s = c0; *c = dd_neg(s0); break;
        case 2: *s = dd_neg(s0); *c = dd_neg(c0); break;
        case 3: *s = dd_neg(c0); *c = s0; break;
        default: break;
    }
}
// sin and cos of a DOUBLE |a| <= 4 (an angle the inverse functions start from), absolute error < 2^-100
HRT_POW_HD void sincos_of_double(double a, DD *s, DD *c) {
    const double kd = a * 0x1.45f306dc9c883p-1;                    // a * 2/pi
    const double kf = kd < 0.0 ? (double)(long long)(kd - 0.5) : (double)(long long)(kd + 0.5);
    DD r = dd(a);
    r = dd_sub(r, dd_two_prod(kf, kPio2a));
    r = dd_sub(r, dd_two_prod(kf, kPio2b));
    r = dd_add_d(r, -kf * kPio2c);
    sincos_series(r, s, c);
    rotate_quadrant((int)(long long)kf, s, c);
}

// 2/pi: the first 448 bits after the binary point, most significant word first
HRT_POW_HD uint32_t two_over_pi_word(int i_from_lsb) {
    constexpr uint32_t T[14] = {0xa2f9836eu, 0x4e441529u, 0xfc2757d1u, 0xf534ddc0u, 0xdb629599u, 0x3c439041u, 0xfe5163abu,
                                0xdebbc561u, 0xb7246e3au, 0x424dd2e0u, 0x06492eeau, 0x09d1921cu, 0xfe1deb1cu, 0xb129a73eu};
    return (i_from_lsb < 0 || i_from_lsb > 13) ? 0u : T[13 - i_from_lsb];
}
// bits [h - 32, h) of floor(2/pi * 2^448) (bit 0 = least significant; bits at and above 448 are zero)
HRT_POW_HD uint32_t two_over_pi_bits(int h) {
    const int lo = h - 32, wi = lo >> 5, sh = lo & 31;
    const uint64_t w = ((uint64_t)two_over_pi_word(wi + 1) << 32) | two_over_pi_word(wi);
    return (uint32_t)(w >> sh);
}
// Payne-Hanek: a finite float ax >= 0.75 -> quadrant k (mod 4) and r in [-pi/4, pi/4] with ax = k pi/2 + r (mod 2 pi), r as a
// double-double with ~106 significant bits however close ax lies to a multiple of pi/2.  ax = M 2^E with a 24-bit integer M:
// ax 2/pi = M 2^E 0.b1b2b3...; the bits of 2/pi that weigh 4 or more after the multiplication only add multiples of 2 pi,
// so a 224-bit window below them times M gives the value mod 4 with 222 fraction bits (the bits dropped below the window
// change the fraction by less than 2^-198).
HRT_POW_HD void payne_hanek(float ax, int *k, DD *r) {
    const uint32_t b = float_to_bits(ax);
    const uint32_t M = (b & 0x7fffffu) | 0x800000u;
    const int E = (int)(b >> 23) - 150;
    const int top = 448 - E + 2;
    uint32_t q[7];
    uint64_t carry = 0;
    for (int i = 6; i >= 0; --i) {
        const uint64_t t = (uint64_t)M * two_over_pi_bits(top - 32 * i) + carry;
        q[i] = (uint32_t)t; carry = t >> 32;
    }
    int quadrant = (int)(q[0] >> 30);
    q[0] &= 0x3fffffffu;
    bool negative = false;
    if (q[0] & 0x20000000u) {                                     // fraction >= 1/2: the next quadrant, fraction - 1
        quadrant += 1; negative = true;
        uint64_t c = 1;
        for (int i = 6; i >= 0; --i) { const uint64_t t = (uint64_t)(~q[i]) + c; q[i] = (uint32_t)t; c = t >> 32; }
        q[0] &= 0x3fffffffu;
    }
    DD f = dd(0.0);
    double w = 0x1p-222;                                           // weight of q[6]'s least significant bit
    for (int i = 6; i >= 0; --i) { f = dd_add_d(f, (double)q[i] * w); w *= 0x1p32; }
    DD v = dd_mul(f, dd(kPio2a, kPio2b));
    *r = negative ? dd_neg(v) : v;
    *k = quadrant & 3;
}

// round-to-nearest of hi + lo to float, either sign
HRT_POW_HD float dd_round_to_float(DD v) {
    if (v.hi < 0.0) return -srgbpow::dd_to_float(dd_neg(v));
    return srgbpow::dd_to_float(v);
}

// true when the double r lies within 2^-44 (relative) of the midpoint of two adjacent floats: its float rounding cannot be
// trusted to be that of the exact value it approximates
HRT_POW_HD bool near_float_midpoint(double r) {
    const float c = (float)r;
    const uint32_t cb = float_to_bits(c);
    const double mu = 0.5 * ((double)c + (double)bits_to_float(cb + 1u)), md = 0.5 * ((double)c + (double)bits_to_float(cb - 1u));
    const double tol = fabs(r) * 0x1p-44;
    return fabs(r - mu) < tol || fabs(r - md) < tol;              // (NaN neighbours compare false)
}
// the fast value settles the result: exact zeros, infinities and NaNs, and everything away from a midpoint
HRT_POW_HD bool settled(double r, bool force_slow) {
    if (r == 0.0 || !(fabs(r) <= 1.0e300)) return true;
    return !force_slow && !near_float_midpoint(r);
}

// sin / cos of a float in double-double
HRT_POW_HD void sincos_dd(float x, DD *s, DD *c) {
    const float ax = fabsf(x);
    DD r = dd((double)ax); int k = 0;
    if (ax >= 0.75f) payne_hanek(ax, &k, &r);
    sincos_series(r, s, c);
    rotate_quadrant(k, s, c);
    if (x < 0.0f) *s = dd_neg(*s);
}

// The angle of the vector (X, Y) given as double-doubles, refined from the double a0 that approximates it to a few ULP:
// rotating the vector by -a0 leaves the small angle delta = atan((Y cos a0 - X sin a0) / (X cos a0 + Y sin a0)).
HRT_POW_HD DD angle_dd(DD Y, DD X, double a0) {
    if (fabs(a0) < 0x1p-20) {                                      // (then X > 0) t - t^3/3 + t^5/5, t = Y / X: relative accuracy for tiny angles
        const DD t = dd_div(Y, X), t2 = dd_mul(t, t);
        DD p = dd_div_d(t2, 5.0);
        p = dd_add_d(p, -1.0 / 3.0); p = dd_add(p, dd(0.0, -0x1.5555555555555p-56));      // -1/3 as a double-double
        p = dd_mul(p, t2);
        p = dd_add_d(p, 1.0);
        return dd_mul(t, p);
    }
    DD s, c;
    sincos_of_double(a0, &s, &c);
    const DD num = dd_sub(dd_mul(Y, c), dd_mul(X, s)), den = dd_add(dd_mul(X, c), dd_mul(Y, s));
    const double delta = num.hi / den.hi;                          // |delta| ~ 2^-52 |a0|: its cube is far below the target
    return dd_add_d(dd(a0), delta);
}

}  // namespace crt

// ---- the five functions, correctly rounded (force_slow: always decide in double-double; the tests sweep it) ----
HRT_POW_HD float sinf_cr(float x, bool force_slow = false) {
    const double r = sin((double)x);
    if (crt::settled(r, force_slow)) return (float)r;
    crt::DD s, c; crt::sincos_dd(x, &s, &c);
    return crt::dd_round_to_float(s);
}
HRT_POW_HD float cosf_cr(float x, bool force_slow = false) {
    const double r = cos((double)x);
    if (crt::settled(r, force_slow)) return (float)r;
    crt::DD s, c; crt::sincos_dd(x, &s, &c);
    return crt::dd_round_to_float(c);
}
HRT_POW_HD float atan2f_cr(float y, float x, bool force_slow = false) {
    const double r = atan2((double)y, (double)x);
    if (crt::settled(r, force_slow) || !(fabsf(x) <= 3.0e38f) || !(fabsf(y) <= 3.0e38f) || (x == 0.0f && y == 0.0f)) return (float)r;
    return crt::dd_round_to_float(crt::angle_dd(crt::dd((double)y), crt::dd((double)x), r));
}
HRT_POW_HD float asinf_cr(float x, bool force_slow = false) {
    const double r = asin((double)x);
    if (crt::settled(r, force_slow)) return (float)r;
    const crt::DD w = crt::dd_sqrt(crt::dd_mul(crt::dd_two_sum(1.0, -(double)x), crt::dd_two_sum(1.0, (double)x)));
    return crt::dd_round_to_float(crt::angle_dd(crt::dd((double)x), w, r));
}
HRT_POW_HD float acosf_cr(float x, bool force_slow = false) {
    const double r = acos((double)x);
    if (crt::settled(r, force_slow)) return (float)r;
    const crt::DD w = crt::dd_sqrt(crt::dd_mul(crt::dd_two_sum(1.0, -(double)x), crt::dd_two_sum(1.0, (double)x)));
    return crt::dd_round_to_float(crt::angle_dd(w, crt::dd((double)x), r));
}

}  // namespace hrt
