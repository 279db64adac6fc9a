// rt_trig.h -- acos, atan2 and sin for the texture maps (Sphere.planeMapInverse Sphere.fs:55-61, ParameterisedTexture.Checkered
// Texture.fs:56-62), defined as the CORRECTLY ROUNDED value of the mathematical function.
//
// Why: the reference calls Math.Acos / Math.Atan2 / Math.Sin, i.e. whatever C runtime .NET sits on; those agree with each other
// and with any other libm only to an ulp or so, and one ulp of (u, v) flips a truncating texel index (Texture.fs:65-67).  So the
// path defines the platform-independent value those approximate -- the exact result rounded once -- and reaches it by two
// independent routes that tests hold against each other bit for bit: here in double-double arithmetic (~2^-104), in the oracle
// through binary128 (libquadmath); a third party, mpmath at 200 bits, pins both (tests/test_trig_cr.py).  glibc's and OCML's
// results are within 1 ulp of it (measured in the same test).  A double-double result misrounds only if the exact value lies
// within ~2^-102 (relative) of a rounding boundary: probability ~2^-48 per evaluation.
//
// Method: x = a library approximation of the function (OCML on the device, libm on the host; <= a few ulp off); then ONE Newton
// step in double-double on the inverse relation, which has cubic convergence here:
//   atan2(y, x): t = (x sin a - y cos a) / (x cos a + y sin a) = tan(a - theta), theta = a - t (t^3/3 < 2^-150 is dropped);
//   acos(x) = atan2(sqrt((1-x)(1+x)), x) with the square root carried in double-double;
// sin/cos of a double argument to double-double: exact reduction by pi/2 carried to 212 bits (products by two_prod, the sum as a
// Shewchuk expansion, so nothing is lost however close the argument is to a multiple of pi/2), then sin(j/32 + d), cos(j/32 + d)
// from a 27-entry double-double table and 7-term Taylor series whose three leading coefficients are double-double.
// Plain C++ (explicit fma only; build with -ffp-contract=off) so the same text is compiled for gfx950 and, in the CPU test, for the
// host.  Generated constants: rt_trig_tables.h (scripts/gen_trig_tables.py).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RTT_FN __host__ __device__ __forceinline__
#define RTT_CONST static __device__ __constant__ const
#else
#define RTT_FN static inline
#define RTT_CONST static const
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define RTT_TAB(t, j, k) t[j][k]
#else
#define RTT_TAB(t, j, k) t##_HOST[j][k]
#endif

#include "rt_trig_tables.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace rtt {

#if defined(__HIPCC__) && !defined(__HIP_DEVICE_COMPILE__)
// host pass of a HIP translation unit: the tables above are device symbols; nothing on the host side of the product uses them
static const double RTT_SIN_TAB_HOST[1][2] = {{0.0, 0.0}};
static const double RTT_COS_TAB_HOST[1][2] = {{1.0, 0.0}};
#elif !defined(__HIPCC__)
#define RTT_SIN_TAB_HOST RTT_SIN_TAB
#define RTT_COS_TAB_HOST RTT_COS_TAB
#endif

struct dd { double hi, lo; };

RTT_FN dd mkdd(double hi, double lo) { dd r; r.hi = hi; r.lo = lo; return r; }
RTT_FN dd two_sum(double a, double b) { // exact: a + b = hi + lo
    const double s = a + b, bb = s - a;
    return mkdd(s, (a - (s - bb)) + (b - bb));
}
RTT_FN dd quick_two_sum(double a, double b) { // exact when |a| >= |b|
    const double s = a + b;
    return mkdd(s, b - (s - a));
}
RTT_FN dd two_prod(double a, double b) { // exact: a * b = hi + lo
    const double p = a * b;
    return mkdd(p, __builtin_fma(a, b, -p));
}
RTT_FN dd dd_neg(dd a) { return mkdd(-a.hi, -a.lo); }
RTT_FN dd dd_add(dd a, dd b) { // the accurate sum of the QD library (error <= 2 * 2^-106 relative)
    dd s = two_sum(a.hi, b.hi);
    const dd t = two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return quick_two_sum(s.hi, s.lo);
}
RTT_FN dd dd_sub(dd a, dd b) { return dd_add(a, dd_neg(b)); }
RTT_FN dd dd_add_d(dd a, double b) {
    dd s = two_sum(a.hi, b);
    s.lo += a.lo;
    return quick_two_sum(s.hi, s.lo);
}
RTT_FN dd dd_mul(dd a, dd b) {
    dd p = two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return quick_two_sum(p.hi, p.lo);
}
RTT_FN dd dd_mul_d(dd a, double b) {
    dd p = two_prod(a.hi, b);
    p.lo += a.lo * b;
    return quick_two_sum(p.hi, p.lo);
}
RTT_FN dd dd_div(dd a, dd b) { // three quotient digits (QD's accurate division)
    const double q1 = a.hi / b.hi;
    dd r = dd_sub(a, dd_mul_d(b, q1));
    const double q2 = r.hi / b.hi;
    r = dd_sub(r, dd_mul_d(b, q2));
    const double q3 = r.hi / b.hi;
    return dd_add_d(quick_two_sum(q1, q2), q3);
}
RTT_FN dd dd_sqrt(dd a) { // Karp's trick: one double-double correction of sqrt(a.hi); a > 0
    const double x = 1.0 / sqrt(a.hi), ax = a.hi * x;
    const dd d = dd_sub(a, two_prod(ax, ax));
    return two_sum(ax, d.hi * (x * 0.5));
}

// a - k * pi/2 as a double-double, k = the integer nearest a * 2/pi; returns k mod 4 (0..3).  |a| < 2^20.
// The eight terms a, -k P1 (two doubles), -k P2 (two), -k P3 (two), -k P4 are summed EXACTLY (grow-expansion: the components stay
// non-overlapping), so the result is good to ~2^-104 of itself however much cancels; what is left out is k * 2^-216.
RTT_FN int reduce_pio2(double a, dd &r) {
    const double kf = __builtin_rint(a * RTT_2OPI);
    if (kf == 0.0) { r = mkdd(a, 0.0); return 0; }
    double t[7];
    const dd p1 = two_prod(-kf, RTT_PIO2_1), p2 = two_prod(-kf, RTT_PIO2_2), p3 = two_prod(-kf, RTT_PIO2_3);
    t[0] = p1.hi; t[1] = p1.lo; t[2] = p2.hi; t[3] = p2.lo; t[4] = p3.hi; t[5] = p3.lo; t[6] = -kf * RTT_PIO2_4;
    double e[8];
    e[0] = a;
#pragma unroll
    for (int n = 1; n <= 7; ++n) { // e[0..n) += t[n-1]
        double q = t[n - 1];
#pragma unroll
        for (int i = 0; i < n; ++i) { const dd s = two_sum(q, e[i]); q = s.hi; e[i] = s.lo; }
        e[n] = q;
    }
    dd acc = mkdd(e[7], 0.0);
#pragma unroll
    for (int i = 6; i >= 0; --i) acc = dd_add_d(acc, e[i]);
    r = acc;
    const long long k = (long long) kf;
    return (int) (k & 3);
}

// sin and cos of a double-double r, |r| <= pi/4 + 2^-40: absolute error < 2^-104, and relative error < 2^-102 for the sine.
RTT_FN void sincos_reduced(dd r, dd &s, dd &c) {
    const double jf = __builtin_rint(r.hi * 32.0);
    int j = (int) jf;
    const bool neg = j < 0;
    if (neg) j = -j;
    if (j > 26) j = 26; // cannot happen for |r| <= pi/4 + 2^-40; keeps the table access in range whatever comes in
    const dd d = dd_add_d(r, -(neg ? -(double) j : (double) j) * 0.03125); // r - j/32, |d| <= 1/64
    const dd d2 = dd_mul(d, d);
    const double z = d2.hi;
    // sin d = d + d * d2 * (S1 + d2 * (S2 + d2 * (S3 + z * (S4 + z * (S5 + z * (S6 + z * S7))))))
    const double st = z * (RTT_S4 + z * (RTT_S5 + z * (RTT_S6 + z * RTT_S7)));
    dd ps = dd_add_d(mkdd(RTT_S3_HI, RTT_S3_LO), st);
    ps = dd_add(mkdd(RTT_S2_HI, RTT_S2_LO), dd_mul(d2, ps));
    ps = dd_add(mkdd(RTT_S1_HI, RTT_S1_LO), dd_mul(d2, ps));
    const dd sd = dd_add(d, dd_mul(dd_mul(d, d2), ps));
    // cos d = 1 + d2 * (C1 + d2 * (C2 + d2 * (C3 + z * (C4 + z * (C5 + z * (C6 + z * C7))))))
    const double ct = z * (RTT_C4 + z * (RTT_C5 + z * (RTT_C6 + z * RTT_C7)));
    dd pc = dd_add_d(mkdd(RTT_C3_HI, RTT_C3_LO), ct);
    pc = dd_add(mkdd(RTT_C2_HI, RTT_C2_LO), dd_mul(d2, pc));
    pc = dd_add(mkdd(RTT_C1_HI, RTT_C1_LO), dd_mul(d2, pc));
    const dd cd = dd_add_d(dd_mul(d2, pc), 1.0);
    if (j == 0) { s = sd; c = cd; return; } // keeps the sine's RELATIVE accuracy near 0
    dd sj = mkdd(RTT_TAB(RTT_SIN_TAB, j, 0), RTT_TAB(RTT_SIN_TAB, j, 1));
    const dd cj = mkdd(RTT_TAB(RTT_COS_TAB, j, 0), RTT_TAB(RTT_COS_TAB, j, 1));
    if (neg) sj = dd_neg(sj);
    s = dd_add(dd_mul(sj, cd), dd_mul(cj, sd));
    c = dd_sub(dd_mul(cj, cd), dd_mul(sj, sd));
}

// sin(a) and cos(a) of a double a, |a| < 2^20, as double-doubles
RTT_FN void sincos_dd(double a, dd &s, dd &c) {
    dd r, sr, cr;
    const int q = reduce_pio2(a, r);
    sincos_reduced(r, sr, cr);
    switch (q) {
    case 0: s = sr; c = cr; break;
    case 1: s = cr; c = dd_neg(sr); break;
    case 2: s = dd_neg(sr); c = dd_neg(cr); break;
    default: s = dd_neg(cr); c = sr; break;
    }
}

#define RTT_TRIG_MAX 1048576.0 /* 2^20: beyond this the reduction above would need more of pi */

// Math.Sin (Texture.fs:58).  Beyond 2^20 (rt_scene_create refuses Checkered grid sizes that could get there) the library value.
RTT_FN double cr_sin(double a) {
    if (a == 0.0 || a != a) return a;
    if (!(__builtin_fabs(a) < RTT_TRIG_MAX)) return sin(a);
    dd s, c;
    sincos_dd(a, s, c);
    return s.hi;
}

// theta0 - tan(theta0 - theta) for theta = atan2(y, x) with y, x double-doubles (x^2 + y^2 > 0, both finite), theta0 within ~2^-40 of it
RTT_FN double newton_atan2(dd y, dd x, double theta0) {
    dd s, c;
    sincos_dd(theta0, s, c);
    const dd num = dd_sub(dd_mul(x, s), dd_mul(y, c));
    const dd den = dd_add(dd_mul(x, c), dd_mul(y, s));
    const dd t = dd_div(num, den);
    return dd_add_d(dd_neg(t), theta0).hi;
}

// Math.Atan2 (Sphere.fs:59); the special cases are C's (which .NET follows)
RTT_FN double cr_atan2(double y, double x) {
    if (x != x || y != y) return x + y;
    if (y == 0.0) return (x > 0.0 || (x == 0.0 && !__builtin_signbit(x))) ? y : __builtin_copysign(RTT_PI, y);
    if (x == 0.0) return __builtin_copysign(RTT_PIO2, y);
    if (__builtin_isinf(x) || __builtin_isinf(y)) {
        if (__builtin_isinf(x) && __builtin_isinf(y)) return __builtin_copysign(x > 0.0 ? RTT_PIO4 : RTT_3PIO4, y);
        if (__builtin_isinf(y)) return __builtin_copysign(RTT_PIO2, y);
        return x > 0.0 ? __builtin_copysign(0.0, y) : __builtin_copysign(RTT_PI, y);
    }
    // scale both by a power of two so that products and quotients below stay far from overflow and underflow (exact; the angle does not change)
    int ex, ey;
    (void) __builtin_frexp(x, &ex);
    (void) __builtin_frexp(y, &ey);
    const int e = ex > ey ? ex : ey;
    const double xs = __builtin_ldexp(x, -e), ys = __builtin_ldexp(y, -e);
    const double theta0 = atan2(y, x);
    if (__builtin_fabs(theta0) < 0x1p-500 || xs == 0.0 || ys == 0.0) return theta0; // |y/x| below 2^-500 (or the reverse): the library's result is the rounded ratio
    return newton_atan2(mkdd(ys, 0.0), mkdd(xs, 0.0), theta0);
}

// Math.Acos (Sphere.fs:60): NaN outside [-1, 1]
RTT_FN double cr_acos(double x) {
    if (!(__builtin_fabs(x) <= 1.0)) return x != x ? x : __builtin_nan("");
    if (x == 1.0) return 0.0;
    if (x == -1.0) return RTT_PI;
    const dd om = two_sum(1.0, -x), op = two_sum(1.0, x); // exact 1 - x and 1 + x
    const dd y = dd_sqrt(dd_mul(om, op));
    return newton_atan2(y, mkdd(x, 0.0), acos(x));
}

} // namespace rtt
