// atan2 of two floats (given as doubles) rounded to the nearest double -- the value glibc's atan2 returns on the reference's
// x86-64 (its error stays below an ulp and is, for float arguments, the correctly rounded result in every case looked at).
//
// Why it exists: the reference forms float(-atan2((double)y, (double)x)) (unionFeatureExtract.cpp:1154, :1136-1139).  The
// device math library's atan2 is good to 2 ulp; when the true angle lies within an ulp OF A DOUBLE of the middle between two
// floats, a last-bit difference between the two libraries rounds to different floats.  The randomised campaign found the case
// (seed 836, a 128-ring scan: one point of 1.6e9, the angle 0.93 double-ulps above the middle; profiles/r05x_fuzz_campaign.txt).
// The fast paths of neg_atan2_f already send every angle within 1e-12 of such a middle here; this routine then decides in
// double-double arithmetic (~1e-31) instead of asking the device library.  About three thousand operations, for one point in 1e6
// (k_azimuth_exact, feature.hip: the bucketing kernels only queue those points).
//
// Host and device: tests/test_host.py compiles it with g++ and checks it against mpmath.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MML_CR_HD __host__ __device__
#else
#define MML_CR_HD
#endif

namespace mml_cr {

struct dd {
    double hi, lo;
};
MML_CR_HD inline dd quick_two_sum(double a, double b) {  // |a| >= |b|
    const double s = a + b;
    return dd{s, b - (s - a)};
}
MML_CR_HD inline dd two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}
MML_CR_HD inline dd two_prod(double a, double b) {
    const double p = a * b;
    return dd{p, __builtin_fma(a, b, -p)};
}
MML_CR_HD inline dd add(dd a, dd b) {
    dd s = two_sum(a.hi, b.hi);
    const dd t = two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return quick_two_sum(s.hi, s.lo);
}
MML_CR_HD inline dd neg(dd a) { return dd{-a.hi, -a.lo}; }
MML_CR_HD inline dd mul(dd a, dd b) {
    dd p = two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return quick_two_sum(p.hi, p.lo);
}
MML_CR_HD inline dd mul_d(dd a, double b) {
    dd p = two_prod(a.hi, b);
    p.lo += a.lo * b;
    return quick_two_sum(p.hi, p.lo);
}
MML_CR_HD inline dd div(dd a, dd b) {  // three quotient digits
    const double q1 = a.hi / b.hi;
    dd r = add(a, neg(mul_d(b, q1)));
    const double q2 = r.hi / b.hi;
    r = add(r, neg(mul_d(b, q2)));
    const double q3 = r.hi / b.hi;
    dd q = quick_two_sum(q1, q2);
    return add(q, dd{q3, 0.0});
}

// atan2(y, x), x and y finite, not both zero; nearest double (up to a double-double error of ~1e-31 relative)
MML_CR_HD inline double atan2_cr(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    const double mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    // a zero or non-finite component, or magnitudes whose quotient leaves the double range: the library's special cases are exact
    if (!(mx > 1e-300 && mx < 1e300) || !(mn > 1e-300)) return atan2(y, x);
    dd q = div(dd{mn, 0.0}, dd{mx, 0.0});  // in (0, 1]
    const bool big = q.hi > 0.41421356237309503;
    // atan(q) = pi / 4 + atan((q - 1) / (q + 1)) for q > tan(pi / 8): |t| <= tan(pi / 8) either way
    dd t = big ? div(add(q, dd{-1.0, 0.0}), add(q, dd{1.0, 0.0})) : q;
    // second reduction: atan|t| = atan(c) + atan((|t| - c) / (1 + |t| c)), c = j / 8 the eighth nearest to |t| (j = 0 .. 3): the
    // series then runs on |u| <= 1 / 15 -- 15 terms reach 1e-35 where 46 were needed on |t| <= tan(pi / 8)
    const bool tneg = t.hi < 0.0;
    if (tneg) t = neg(t);
    const int j = (int)(8.0 * t.hi + 0.5);  // 0 .. 3
    dd u = t;
    if (j > 0) {
        const double c = 0.125 * (double)j;
        u = div(add(t, dd{-c, 0.0}), add(dd{1.0, 0.0}, mul_d(t, c)));
    }
    const dd u2 = mul(u, u);
    dd s = u, pw = u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int k = 1; k <= 15; ++k) {
        pw = mul(pw, u2);
        dd term = div(pw, dd{(double)(2 * k + 1), 0.0});
        s = add(s, (k & 1) ? neg(term) : term);
    }
    const dd atan_c = j == 1 ? dd{1.24354994546761438e-01, -3.12532414245393831e-18}
                             : (j == 2 ? dd{2.44978663126864143e-01, 1.06987556187344514e-17}
                                       : (j == 3 ? dd{3.58770670270572245e-01, -2.46238155826386349e-17} : dd{0.0, 0.0}));
    s = add(atan_c, s);
    if (tneg) s = neg(s);
    const dd pi4 = dd{0.78539816339744828, 3.0616169978683830e-17};
    const dd pi2 = dd{1.5707963267948966, 6.1232339957367660e-17};
    const dd pi1 = dd{3.1415926535897931, 1.2246467991473532e-16};
    dd r = big ? add(pi4, s) : s;
    if (ay > ax) r = add(pi2, neg(r));
    if (x < 0.0) r = add(pi1, neg(r));
    if (y < 0.0) r = neg(r);
    return r.hi;  // (normalised: |lo| <= ulp(hi) / 2, hi is the nearest double)
}

}  // namespace mml_cr
