// srgb_pow.h -- powf(x, 1.0f / 2.4f) of colorToFloat4 / colorToUchar4 (include/Global/DeviceFunctions.cuh:161-163,
// :196-198), pinned as the CORRECTLY ROUNDED float of x^y, y = (double)(1.0f / 2.4f), for x in [0, 1].
//
// CUDA's powf is closed third-party arithmetic; "correctly rounded" is the one definition every platform
// can reproduce bit for bit.  The oracle reaches it through libm's double pow (+ __float128 near rounding boundaries);
// this file reaches it with +, -, *, / and fma on doubles only -- no libm, no contraction -- so the device and a host
// compile of the same header give the same bits:
//   fast path  x^y in double by series (error < 2^-48 relative), taken unless the result lies within 2^-44 (relative) of
//              the midpoint of two adjacent floats;
//   slow path  (about one input in 500 000) the same evaluation in double-double arithmetic (error < 2^-95), rounded once.
// tests/test_host_cpu.py compares a host compile with the oracle over ALL floats in [0, 1]; tests/test_gpu_parity.py
// does the same sweep on the GPU through hrt_color_to_float4.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HRT_POW_HD __host__ __device__ inline
#else
#define HRT_POW_HD inline
#endif

namespace hrt {
namespace srgbpow {

HRT_POW_HD double bits_to_double(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
HRT_POW_HD uint64_t double_to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }
HRT_POW_HD float bits_to_float(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
HRT_POW_HD uint32_t float_to_bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

#if defined(__HIP_DEVICE_COMPILE__)
HRT_POW_HD double fma_d(double a, double b, double c) { return __builtin_fma(a, b, c); }
#else
HRT_POW_HD double fma_d(double a, double b, double c) { return __builtin_fma(a, b, c); }
#endif

// ---- double-double: value = hi + lo, |lo| <= ulp(hi) / 2 ----
struct DD { double hi, lo; };
HRT_POW_HD DD dd_renorm(double a, double b) { DD r; r.hi = a + b; r.lo = b - (r.hi - a); return r; }     // |a| >= |b|
HRT_POW_HD DD dd_two_sum(double a, double b) { DD r; r.hi = a + b; const double bb = r.hi - a; r.lo = (a - (r.hi - bb)) + (b - bb); return r; }
HRT_POW_HD DD dd_two_prod(double a, double b) { DD r; r.hi = a * b; r.lo = fma_d(a, b, -r.hi); return r; }
HRT_POW_HD DD dd_add(DD a, DD b) { DD s = dd_two_sum(a.hi, b.hi); s.lo += a.lo + b.lo; return dd_renorm(s.hi, s.lo); }
HRT_POW_HD DD dd_mul(DD a, DD b) { DD p = dd_two_prod(a.hi, b.hi); p.lo += a.hi * b.lo + a.lo * b.hi; return dd_renorm(p.hi, p.lo); }
HRT_POW_HD DD dd_mul_d(DD a, double b) { DD p = dd_two_prod(a.hi, b); p.lo += a.lo * b; return dd_renorm(p.hi, p.lo); }
HRT_POW_HD DD dd_div_d(DD a, double b) {         // a / b, b a double
    const double q1 = a.hi / b;
    const double r = fma_d(-q1, b, a.hi) + a.lo;       // exact remainder of the high part + the low part
    return dd_renorm(q1, r / b);
}

// ln 2 as a double-double
HRT_POW_HD DD dd_ln2() { DD r; r.hi = 0x1.62e42fefa39efp-1; r.lo = 0x1.abc9e3b39803fp-56; return r; }

// x^y, x a positive normal float value, in double-double
HRT_POW_HD DD pow_dd(double x, double y) {
    uint64_t bits = double_to_bits(x);
    int e = (int)((bits >> 52) & 0x7ffu) - 1023;
    bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m = bits_to_double(bits);                        // [1, 2), 24 significant bits
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }      // [sqrt(1/2), sqrt(2))
    // ln m = 2 atanh(s), s = (m - 1) / (m + 1): numerator and denominator are exact doubles
    DD num; num.hi = m - 1.0; num.lo = 0.0;
    const DD s = dd_div_d(num, m + 1.0);
    const DD s2 = dd_mul(s, s);
    // sum_{k=0..24} s2^k / (2k+1), |s2| <= 0.0295: the first neglected term is < 2^-128
    DD p; p.hi = 0.0; p.lo = 0.0;
    for (int k = 24; k >= 0; --k) {
        DD one; one.hi = 1.0; one.lo = 0.0;
        p = dd_add(dd_mul(p, s2), dd_div_d(one, (double)(2 * k + 1)));
    }
    DD ln_m = dd_mul(s, p); ln_m.hi *= 2.0; ln_m.lo *= 2.0;
    const DD ln_x = dd_add(dd_mul_d(dd_ln2(), (double)e), ln_m);
    const DD w = dd_mul_d(ln_x, y);                        // ln(x^y)
    const double nf = w.hi / 0.6931471805599453;
    const double n = nf < 0.0 ? (double)(long long)(nf - 0.5) : (double)(long long)(nf + 0.5);
    DD nl = dd_mul_d(dd_ln2(), n); nl.hi = -nl.hi; nl.lo = -nl.lo;
    const DD f = dd_add(w, nl);                            // |f| <= 0.35
    // exp(f) = sum f^k / k!, k <= 32: the first neglected term is < 2^-120
    DD term; term.hi = 1.0; term.lo = 0.0;
    DD sum = term;
    for (int k = 1; k <= 32; ++k) {
        term = dd_div_d(dd_mul(term, f), (double)k);
        sum = dd_add(sum, term);
    }
    const double scale = bits_to_double((uint64_t)((long long)n + 1023) << 52);     // 2^n, exact
    DD r; r.hi = sum.hi * scale; r.lo = sum.lo * scale;
    return r;
}

// round-to-nearest of hi + lo to float (ties cannot be told from near-ties at this precision and keep (float)hi)
HRT_POW_HD float dd_to_float(DD v) {
    const float c = (float)v.hi;
    const double d = (v.hi - (double)c) + v.lo;            // value - c; the first difference is exact
    if (d == 0.0) return c;
    const uint32_t cb = float_to_bits(c);
    const float nb = bits_to_float(d > 0.0 ? cb + 1u : cb - 1u);       // neighbour towards the value (c > 0)
    const double half = 0.5 * ((double)nb - (double)c);
    return (d > 0.0 ? d > half : d < half) ? nb : c;
}

// x^(1/2.4f) in double, +,-,*,/ only; relative error < 2^-48
HRT_POW_HD double pow_fast(double x, double y) {
    uint64_t bits = double_to_bits(x);
    int e = (int)((bits >> 52) & 0x7ffu) - 1023;
    bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m = bits_to_double(bits);
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);                // |s| <= 0.1716
    const double s2 = s * s;
    double p = 1.0 / 27.0;
    p = p * s2 + 1.0 / 25.0; p = p * s2 + 1.0 / 23.0; p = p * s2 + 1.0 / 21.0; p = p * s2 + 1.0 / 19.0;
    p = p * s2 + 1.0 / 17.0; p = p * s2 + 1.0 / 15.0; p = p * s2 + 1.0 / 13.0; p = p * s2 + 1.0 / 11.0;
    p = p * s2 + 1.0 / 9.0;  p = p * s2 + 1.0 / 7.0;  p = p * s2 + 1.0 / 5.0;  p = p * s2 + 1.0 / 3.0;
    p = p * s2 + 1.0;
    const double ln_m = 2.0 * s * p;
    const double log2x = (double)e + ln_m * 1.4426950408889634;
    const double z = y * log2x;                            // <= 0
    const double zr = z < 0.0 ? (double)(long long)(z - 0.5) : (double)(long long)(z + 0.5);
    const double f = (z - zr) * 0.6931471805599453;        // |f| <= 0.347
    double q = 1.0 / 6227020800.0;                         // 1/13!
    q = q * f + 1.0 / 479001600.0; q = q * f + 1.0 / 39916800.0; q = q * f + 1.0 / 3628800.0;
    q = q * f + 1.0 / 362880.0;    q = q * f + 1.0 / 40320.0;    q = q * f + 1.0 / 5040.0;
    q = q * f + 1.0 / 720.0;       q = q * f + 1.0 / 120.0;      q = q * f + 1.0 / 24.0;
    q = q * f + 1.0 / 6.0;         q = q * f + 0.5;              q = q * f + 1.0;
    q = q * f + 1.0;
    return q * bits_to_double((uint64_t)((long long)zr + 1023) << 52);
}

}  // namespace srgbpow

// powf(x, 1.0f / 2.4f), correctly rounded; x in [0, 1] (anything <= 0 or NaN gives 0, as the clamp upstream guarantees)
HRT_POW_HD float pow_inv_gamma(float xf) {
    using namespace srgbpow;
    if (!(xf > 0.0f)) return 0.0f;
    const double y = (double)(1.0f / 2.4f);
    const double x = (double)xf;                                      // exact; a subnormal float is a normal double
    const double r = pow_fast(x, y);
    const float c = (float)r;
    const uint32_t cb = float_to_bits(c);
    const double mid_up = 0.5 * ((double)c + (double)bits_to_float(cb + 1u));
    const double mid_dn = 0.5 * ((double)c + (double)bits_to_float(cb - 1u));
    const double tol = r * 0x1p-44;
    const double du = r - mid_up, dn = r - mid_dn;
    if ((du < tol && du > -tol) || (dn < tol && dn > -tol)) return dd_to_float(pow_dd(x, y));
    return c;
}

}  // namespace hrt
