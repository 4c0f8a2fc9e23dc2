// oracle/libm_f32.h -- atanf / atan2f as the libm the reference links computes them.  TEST INFRASTRUCTURE ONLY.
//
// Why: unionFeatureExtract.cpp:1136-1139,1159,1168 call unqualified atan2 / atan / sqrt on floats.  The TU includes
// lidars_extrinsic_cali.h (:58) -> <tf/tf.h> (lidars_extrinsic_cali.h:3) -> tf/LinearMath/Scalar.h -> <math.h>, and with
// libstdc++ >= 6 (melodic: gcc 7.5) that header is the C++ wrapper that pulls std::atan2(float, float), std::atan(float),
// std::sqrt(float) into the global namespace: overload resolution picks the FLOAT functions
// (tests/test_oracle.py::test_libm_overloads_resolve_to_float compiles the two-line check).  They are glibc's atan2f / atanf /
// sqrtf -- melodic's glibc 2.27: sysdeps/ieee754/flt-32/e_atan2f.c and s_atanf.c, the fdlibm routines (Sun Microsystems 1993,
// float conversion by Ian Lance Taylor), unchanged up to this image's glibc 2.35 (the correctly rounded CORE-MATH versions only
// came with 2.41).  They are pure float arithmetic without FMA on x86-64 (there is no multiarch variant of either), so restating
// the published algorithm reproduces the binary's bits -- and that is PINNED: tests/test_oracle.py compares mmlo_atanf with this
// image's atanf on all 2^32 arguments and mmlo_atan2f with its atan2f on 4e8 pairs (random, octant edges, zeros, denormals,
// infinities, huge ratios).  This is the one place where the oracle is checked against a binary the reference really calls.
// Build with -ffp-contract=off.
#pragma once
#include <cstdint>
#include <cstring>

namespace mmlo_libm {

inline int32_t f2i(float f) {
    int32_t i;
    std::memcpy(&i, &f, 4);
    return i;
}
inline float i2f(int32_t i) {
    float f;
    std::memcpy(&f, &i, 4);
    return f;
}

// s_atanf.c
inline float atanf_fdlibm(float x) {
    static const float atanhi[] = {
        4.6364760399e-01f,  // atan(0.5)hi 0x3eed6338
        7.8539812565e-01f,  // atan(1.0)hi 0x3f490fda
        9.8279368877e-01f,  // atan(1.5)hi 0x3f7b985e
        1.5707962513e+00f,  // atan(inf)hi 0x3fc90fda
    };
    static const float atanlo[] = {
        5.0121582440e-09f,  // atan(0.5)lo 0x31ac3769
        3.7748947079e-08f,  // atan(1.0)lo 0x33222168
        3.4473217170e-08f,  // atan(1.5)lo 0x33140fb4
        7.5497894159e-08f,  // atan(inf)lo 0x33a22168
    };
    static const float aT[] = {
        3.3333334327e-01f,  -2.0000000298e-01f, 1.4285714924e-01f,  -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
        6.6610731184e-02f,  -5.8335702866e-02f, 4.9768779427e-02f,  -3.6531571299e-02f, 1.6285819933e-02f,
    };
    const float one = 1.0f;
    float w, s1, s2, z;
    int32_t ix, hx, id;
    hx = f2i(x);
    ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) {  // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;  // NaN
        if (hx > 0)
            return atanhi[3] + atanlo[3];
        else
            return -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {               // |x| < 0.4375
        if (ix < 0x31000000) return x;  // |x| < 2^-29
        id = -1;
    } else {
        x = i2f(ix);  // fabsf
        if (ix < 0x3f980000) {      // |x| < 1.1875
            if (ix < 0x3f300000) {  // 7/16 <= |x| < 11/16
                id = 0;
                x = (2.0f * x - one) / (2.0f + x);
            } else {  // 11/16 <= |x| < 19/16
                id = 1;
                x = (x - one) / (x + one);
            }
        } else {
            if (ix < 0x401c0000) {  // |x| < 2.4375
                id = 2;
                x = (x - 1.5f) / (one + 1.5f * x);
            } else {  // 2.4375 <= |x| < 2^25
                id = 3;
                x = -1.0f / x;
            }
        }
    }
    z = x * x;
    w = z * z;
    s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx < 0) ? -z : z;
}

// e_atan2f.c
inline float atan2f_fdlibm(float y, float x) {
    const float tiny = 1.0e-30f, zero = 0.0f;
    const float pi_o_4 = 7.8539818525e-01f;  // 0x3f490fdb
    const float pi_o_2 = 1.5707963705e+00f;  // 0x3fc90fdb
    const float pi = 3.1415927410e+00f;      // 0x40490fdb
    const float pi_lo = -8.7422776573e-08f;  // 0xb3bbbd2e
    float z;
    int32_t k, m, hx, hy, ix, iy;
    hx = f2i(x);
    ix = hx & 0x7fffffff;
    hy = f2i(y);
    iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;  // NaN
    if (hx == 0x3f800000) return atanf_fdlibm(y);          // x = 1.0
    m = ((hy >> 31) & 1) | ((hx >> 30) & 2);               // 2 * sign(x) + sign(y)
    if (iy == 0) {
        switch (m) {
            case 0:
            case 1: return y;
            case 2: return pi + tiny;
            case 3: return -pi - tiny;
        }
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                case 3: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
                case 0: return zero;
                case 1: return -zero;
                case 2: return pi + tiny;
                case 3: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    k = (iy - ix) >> 23;
    if (k > 60)
        z = pi_o_2 + 0.5f * pi_lo;  // |y / x| > 2^60
    else if (hx < 0 && k < -60)
        z = 0.0f;  // |y| / x < -2^60
    else
        z = atanf_fdlibm(i2f(f2i(y / x) & 0x7fffffff));
    switch (m) {
        case 0: return z;
        case 1: return i2f(f2i(z) ^ (int32_t)0x80000000);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

}  // namespace mmlo_libm
