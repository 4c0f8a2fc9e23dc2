// csrc/libm_f32.h -- atanf / atan2f with the bits of the libm the reference binary calls (host and device source).
//
// unionFeatureExtract.cpp:1136-1139,1159,1168 call unqualified atan2 / atan / sqrt on floats in a TU that sees <math.h>
// (:58 -> lidars_extrinsic_cali.h:3 <tf/tf.h> -> tf/LinearMath/Scalar.h), so with libstdc++ >= 6 they resolve to the FLOAT
// overloads = glibc's atan2f / atanf / sqrtf (convention 4).  glibc 2.27 (melodic) to 2.35 (this image)
// compute them with the fdlibm float routines (sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c): argument reduction by one
// division, an 11-term polynomial split into odd and even halves, hi / lo table constants -- float arithmetic only, every
// operation rounded to nearest, no FMA.  This file evaluates the same published algorithm: every rounding of the C routine
// happens here in the same order (the build has -ffp-contract=off; v_div / v_sqrt expansions of hipcc are correctly rounded), the
// four reduction branches become selects in front of ONE division so that the lanes of a wavefront do not diverge.
// Pinned twice: compiled for the host, against this image's atanf on all 2^32 arguments and atan2f on 4e8 pairs
// (tests/test_host.py::test_libm_f32_equals_glibc); on the device, against the host values (tests/test_gpu_parity.py::
// test_device_atan2f_atanf_bits, mml_debug_libm_f32).
#pragma once
#if defined(__HIPCC__)
#define MML_LIBM_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define MML_LIBM_HD inline
#endif

namespace mml_libm {

MML_LIBM_HD int f2i(float f) { return __builtin_bit_cast(int, f); }
MML_LIBM_HD float i2f(int i) { return __builtin_bit_cast(float, i); }

// s_atanf.c behind its two early returns: 2^-29 <= |x| < 2^25 (hx = the bits of x)
MML_LIBM_HD float atanf_core(float x, int hx) {
    const int ix = hx & 0x7fffffff;
    const float ax = i2f(ix);
    const bool small = ix < 0x3ee00000;  // |x| < 7/16: no reduction
    const bool c0 = ix < 0x3f300000;     // < 11/16: (2x - 1) / (2 + x)
    const bool c1 = ix < 0x3f980000;     // < 19/16: (x - 1) / (x + 1)
    const bool c2 = ix < 0x401c0000;     // < 39/16: (x - 1.5) / (1 + 1.5 x); beyond: -1 / x
    const float num = c0 ? 2.0f * ax - 1.0f : c1 ? ax - 1.0f : c2 ? ax - 1.5f : -1.0f;
    const float den = c0 ? 2.0f + ax : c1 ? ax + 1.0f : c2 ? 1.0f + 1.5f * ax : ax;
    const float hi = c0 ? 4.6364760399e-01f : c1 ? 7.8539812565e-01f : c2 ? 9.8279368877e-01f : 1.5707962513e+00f;
    const float lo = c0 ? 5.0121582440e-09f : c1 ? 3.7748947079e-08f : c2 ? 3.4473217170e-08f : 7.5497894159e-08f;
    const float q = num / den;
    const float t = small ? x : q;
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (3.3333334327e-01f + w * (1.4285714924e-01f + w * (9.0908870101e-02f + w * (6.6610731184e-02f + w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 = w * (-2.0000000298e-01f + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f + w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
    const float ts = t * (s1 + s2);
    const float r = hi - ((ts - lo) - t);
    return small ? t - ts : (hx < 0 ? -r : r);
}

MML_LIBM_HD float atanf_fd(float x) {
    const int hx = f2i(x), ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) {  // |x| >= 2^25 (inf, NaN)
        if (ix > 0x7f800000) return x + x;
        const float r = 1.5707962513e+00f + 7.5497894159e-08f;  // atanhi[3] + atanlo[3]
        return hx > 0 ? r : -r;
    }
    if (ix < 0x31000000) return x;  // |x| < 2^-29
    return atanf_core(x, hx);
}

MML_LIBM_HD float atan2f_fd(float y, float x) {
    const float tiny = 1.0e-30f;
    const float pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int hx = f2i(x), ix = hx & 0x7fffffff;
    const int hy = f2i(y), iy = hy & 0x7fffffff;
    // The common case in ONE test (every coordinate pair of a scan but a handful): both arguments normal numbers, x != 1, exponents
    // within 2^24 of each other -- none of the routine's special cases applies and the quotient lies inside atanf's general range
    // [2^-25, 2^25], so what is left is a division, atanf's reduction + polynomial and the quadrant.  Same operations, same order.
    {
        const unsigned dk = (unsigned)(((iy - ix) >> 23) + 24);
        if ((unsigned)(ix - 0x00800000) < 0x7f000000u && (unsigned)(iy - 0x00800000) < 0x7f000000u && hx != 0x3f800000 && dk <= 48u) {
            const int hq = f2i(y / x) & 0x7fffffff;
            const float z = atanf_core(i2f(hq), hq);
            const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
            return m == 0 ? z : m == 1 ? i2f(f2i(z) ^ (int)0x80000000) : m == 2 ? pi - (z - pi_lo) : (z - pi_lo) - pi;
        }
    }
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return atanf_fd(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) return m < 2 ? y : m == 2 ? pi + tiny : -pi - tiny;
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : m == 1 ? -pi_o_4 - tiny : m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
        return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60)
        z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60)
        z = 0.0f;
    else
        z = atanf_fd(i2f(f2i(y / x) & 0x7fffffff));
    return m == 0 ? z : m == 1 ? i2f(f2i(z) ^ (int)0x80000000) : m == 2 ? pi - (z - pi_lo) : (z - pi_lo) - pi;
}

}  // namespace mml_libm
