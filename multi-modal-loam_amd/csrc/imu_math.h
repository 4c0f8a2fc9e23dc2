// imu_math.h -- the small dense SO(3) / IMU-factor arithmetic of the full-window problem, written once for the host
// solver (window_imu.hip) and the device-resident one (fullwindow_dev.hip):
//   * Sophus::SO3d exp / log (include/sophus/so3.hpp:585-622 and logAndTheta), Eigen quaternion <-> matrix
//   * Cost_NavState_PRV_Bias residual and analytic Jacobian before the sqrt information (ceresfunc.h:321-393)
//   * MarginalizationFactor::Evaluate residual (ceresfunc.h:262-301)
#pragma once
#include <math.h>

#include "../../include/mmloam_hip.h"

#define MML_HD __host__ __device__ inline

namespace {

// ---- small dense helpers (row-major) -------------------------------------------------------------------------------
struct M3 {
    double a[9];
};
MML_HD M3 m3_identity() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
MML_HD M3 m3_mul(const M3& A, const M3& B) {
    M3 C;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C.a[3 * r + c] = (A.a[3 * r] * B.a[c] + A.a[3 * r + 1] * B.a[3 + c]) + A.a[3 * r + 2] * B.a[6 + c];
    return C;
}
MML_HD M3 m3_t(const M3& A) { return M3{{A.a[0], A.a[3], A.a[6], A.a[1], A.a[4], A.a[7], A.a[2], A.a[5], A.a[8]}}; }
MML_HD void m3_vec(const M3& A, const double* v, double* o) {
    for (int r = 0; r < 3; ++r) o[r] = (A.a[3 * r] * v[0] + A.a[3 * r + 1] * v[1]) + A.a[3 * r + 2] * v[2];
}
MML_HD M3 hat(const double* v) { return M3{{0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0}}; }
MML_HD M3 m3_scale(const M3& A, double s) {
    M3 C;
    for (int i = 0; i < 9; ++i) C.a[i] = A.a[i] * s;
    return C;
}
MML_HD M3 m3_add(const M3& A, const M3& B) {
    M3 C;
    for (int i = 0; i < 9; ++i) C.a[i] = A.a[i] + B.a[i];
    return C;
}

// ---- sin / cos / atan written out once for both sides -----------------------------------------------------------------------
// The host iteration and the device-resident one must take the same decisions, and the last bit of a sine decides some of them
// (a step accepted or rejected at a model decrease of 1e-16): libm on the host and the device math library agree to an ulp,
// not to the bit.  These are the classic fdlibm kernels (k_sin.c, k_cos.c, s_atan.c: minimax polynomials, Cody-Waite reduction
// by pi/2 in two pieces -- arguments here are rotation angles, a few pi at most), plain double arithmetic in a fixed order and
// compiled without contraction on both sides: identical results by construction, within an ulp of the libm values.
MML_HD double mml_ksin(double x) {  // |x| <= pi/4
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x, v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x + v * (S1 + z * r);
}
MML_HD double mml_kcos(double x) {  // |x| <= pi/4
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double ax = fabs(x);
    if (ax < 0.3) return 1.0 - (0.5 * z - z * r);
    const double qx = ax > 0.78125 ? 0.28125 : 0.25 * ax;
    const double hz = 0.5 * z - qx, a = 1.0 - qx;
    return a - (hz - z * r);
}
// x = n * pi/2 + y, |y| <= pi/4 (+ rounding); returns n mod 4
MML_HD int mml_rem_pio2(double x, double& y) {
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    if (fabs(x) <= 0.78539816339744830962) {
        y = x;
        return 0;
    }
    const double fn = rint(x * invpio2);
    const double r = x - fn * pio2_1, w = fn * pio2_1t;
    y = r - w;
    return (int)((long long)fn & 3);
}
MML_HD double mml_sin(double x) {
    double y;
    switch (mml_rem_pio2(x, y)) {
        case 0: return mml_ksin(y);
        case 1: return mml_kcos(y);
        case 2: return -mml_ksin(y);
        default: return -mml_kcos(y);
    }
}
MML_HD double mml_cos(double x) {
    double y;
    switch (mml_rem_pio2(x, y)) {
        case 0: return mml_kcos(y);
        case 1: return -mml_ksin(y);
        case 2: return -mml_kcos(y);
        default: return mml_ksin(y);
    }
}
MML_HD double mml_atan(double xin) {
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01, aT2 = 1.42857142725034663711e-01,
                 aT3 = -1.11111104054623557880e-01, aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02, aT8 = 4.97687799461593236017e-02,
                 aT9 = -3.65315727442169155270e-02, aT10 = 1.62858201153657823623e-02;
    if (!(xin == xin)) return xin;
    const bool neg = xin < 0.0;
    double x = fabs(xin);
    if (x >= 73786976294838206464.0) {  // 2^66: atan = +-pi/2
        const double z = 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
        return neg ? -z : z;
    }
    double hi = 0.0, lo = 0.0;
    int id = -1;
    if (x < 0.4375) {
        if (x < 1.862645149230957e-09) return xin;  // 2^-29
    } else if (x < 1.1875) {
        if (x < 0.6875) {
            id = 0;
            hi = 4.63647609000806093515e-01;
            lo = 2.26987774529616870924e-17;
            x = (2.0 * x - 1.0) / (2.0 + x);
        } else {
            id = 1;
            hi = 7.85398163397448278999e-01;
            lo = 3.06161699786838301793e-17;
            x = (x - 1.0) / (x + 1.0);
        }
    } else if (x < 2.4375) {
        id = 2;
        hi = 9.82793723247329054082e-01;
        lo = 1.39033110312309984516e-17;
        x = (x - 1.5) / (1.0 + 1.5 * x);
    } else {
        id = 3;
        hi = 1.57079632679489655800e+00;
        lo = 6.12323399573676603587e-17;
        x = -1.0 / x;
    }
    const double z = x * x, w = z * z;
    const double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) {
        const double r = x - x * (s1 + s2);
        return neg ? -r : r;
    }
    const double r = hi - ((x * (s1 + s2) - lo) - x);
    return neg ? -r : r;
}

// Sophus::SO3d::exp (so3.hpp:585-622, epsilon 1e-10 on theta^2): rotation vector -> unit quaternion (x, y, z, w)
MML_HD void so3_exp_q(const double* w, double* q) {
    const double th2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
    double imag, real;
    if (th2 < 1e-10 * 1e-10) {
        const double th4 = th2 * th2;
        imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        real = 1.0 - 0.125 * th2 + (1.0 / 384.0) * th4;
    } else {
        const double th = sqrt(th2), h = 0.5 * th;
        imag = mml_sin(h) / th;
        real = mml_cos(h);
    }
    q[0] = imag * w[0];
    q[1] = imag * w[1];
    q[2] = imag * w[2];
    q[3] = real;
}
MML_HD M3 quat_to_m3(const double* q) {  // Eigen::Quaterniond::toRotationMatrix
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y,
                 tzz = tz * z;
    return M3{{1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)}};
}
MML_HD M3 so3_exp(const double* w) {
    double q[4];
    so3_exp_q(w, q);
    return quat_to_m3(q);
}
// Eigen quaternionbase_assign_impl<Matrix3d>
MML_HD void m3_to_quat(const M3& M, double* q) {
    const double* m = M.a;
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t;
        q[1] = (m[2] - m[6]) * t;
        q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
        q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
}
// Sophus::SO3d::log of a unit quaternion (so3.hpp logAndTheta)
MML_HD void so3_log_q(const double* q, double* w) {
    const double n2 = (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2];
    const double qw = q[3];
    double two_atan;
    if (n2 < 1e-10 * 1e-10) {
        two_atan = 2.0 / qw - (2.0 / 3.0) * n2 / (qw * qw * qw);
    } else {
        const double n = sqrt(n2);
        if (fabs(qw) < 1e-10)
            two_atan = (qw > 0 ? M_PI : -M_PI) / n;
        else
            two_atan = 2.0 * mml_atan(n / qw) / n;
    }
    w[0] = two_atan * q[0];
    w[1] = two_atan * q[1];
    w[2] = two_atan * q[2];
}
MML_HD void so3_log(const M3& R, double* w) {
    double q[4];
    m3_to_quat(R, q);
    const double n = sqrt((q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3]));
    for (int i = 0; i < 4; ++i) q[i] /= n;
    so3_log_q(q, w);
}
// right Jacobian of SO(3) and its inverse
MML_HD M3 so3_Jr(const double* w) {
    const double th2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
    const M3 K = hat(w), K2 = m3_mul(K, K);
    double a, b;
    if (th2 < 1e-8) {
        a = 0.5 - th2 / 24.0;
        b = 1.0 / 6.0 - th2 / 120.0;
    } else {
        const double th = sqrt(th2);
        a = (1.0 - mml_cos(th)) / th2;
        b = (th - mml_sin(th)) / (th2 * th);
    }
    return m3_add(m3_add(m3_identity(), m3_scale(K, -a)), m3_scale(K2, b));
}
MML_HD M3 so3_Jr_inv(const double* w) {
    const double th2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
    const M3 K = hat(w), K2 = m3_mul(K, K);
    double c;
    if (th2 < 1e-8) {
        c = 1.0 / 12.0 + th2 / 720.0;
    } else {
        const double th = sqrt(th2);
        c = 1.0 / th2 - (1.0 + mml_cos(th)) / (2.0 * th * mml_sin(th));
    }
    return m3_add(m3_add(m3_identity(), m3_scale(K, 0.5)), m3_scale(K2, c));
}

MML_HD void set_block(double* M, int ld, int r0, int c0, const M3& B, double s = 1.0) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M[(r0 + r) * ld + c0 + c] = s * B.a[3 * r + c];
}

MML_HD M3 get_block(const double* M, int ld, int r0, int c0) {
    M3 B;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) B.a[3 * r + c] = M[(r0 + r) * ld + c0 + c];
    return B;
}

// residual (15) and Jacobian (15 x 30, columns [PR_i 6 | VBias_i 9 | PR_j 6 | VBias_j 9]) BEFORE the sqrt information
MML_HD void imu_raw(const mml_imu_preint* pre, const double* g, const double* pri, const double* vbi, const double* prj,
             const double* vbj, double* r, double* J) {
    const double dt = pre->dtime, dt2 = dt * dt;
    const M3 Ri = so3_exp(pri + 3), Rj = so3_exp(prj + 3), RiT = m3_t(Ri);
    const double dbg[3] = {vbi[3] - pre->bg[0], vbi[4] - pre->bg[1], vbi[5] - pre->bg[2]};
    const double dba[3] = {vbi[6] - pre->ba[0], vbi[7] - pre->ba[1], vbi[8] - pre->ba[2]};
    const double* PJ = pre->jacobian;
    const M3 Jpbg = get_block(PJ, 15, 0, 9), Jpba = get_block(PJ, 15, 0, 12), Jrbg = get_block(PJ, 15, 3, 9),
             Jvbg = get_block(PJ, 15, 6, 9), Jvba = get_block(PJ, 15, 6, 12);
    double a[3], b[3];
    for (int k = 0; k < 3; ++k) {
        a[k] = prj[k] - pri[k] - vbi[k] * dt - 0.5 * g[k] * dt2;
        b[k] = vbj[k] - vbi[k] - g[k] * dt;
    }
    double Ra[3], Rb[3], t1[3], t2[3];
    m3_vec(RiT, a, Ra);
    m3_vec(RiT, b, Rb);
    m3_vec(Jpbg, dbg, t1);
    m3_vec(Jpba, dba, t2);
    for (int k = 0; k < 3; ++k) r[k] = Ra[k] - (pre->dp[k] + t1[k] + t2[k]);
    m3_vec(Jvbg, dbg, t1);
    m3_vec(Jvba, dba, t2);
    for (int k = 0; k < 3; ++k) r[6 + k] = Rb[k] - (pre->dv[k] + t1[k] + t2[k]);
    double jd[3];
    m3_vec(Jrbg, dbg, jd);
    const M3 dR = quat_to_m3(pre->dq);
    const M3 C = m3_mul(dR, so3_exp(jd));
    const M3 E = m3_mul(m3_t(C), m3_mul(RiT, Rj));
    so3_log(E, r + 3);
    for (int k = 0; k < 6; ++k) r[9 + k] = vbj[3 + k] - vbi[3 + k];
    if (!J) return;
    for (int i = 0; i < 15 * 30; ++i) J[i] = 0.0;
    const M3 Jri = so3_Jr(pri + 3), Jrj = so3_Jr(prj + 3), JrInv = so3_Jr_inv(r + 3);
    // position rows
    set_block(J, 30, 0, 0, RiT, -1.0);
    set_block(J, 30, 0, 3, m3_mul(hat(Ra), Jri));
    set_block(J, 30, 0, 6, RiT, -dt);
    set_block(J, 30, 0, 9, Jpbg, -1.0);
    set_block(J, 30, 0, 12, Jpba, -1.0);
    set_block(J, 30, 0, 15, RiT);
    // rotation rows
    set_block(J, 30, 3, 3, m3_mul(JrInv, m3_mul(m3_mul(m3_t(Rj), Ri), Jri)), -1.0);
    set_block(J, 30, 3, 9, m3_mul(JrInv, m3_mul(m3_t(E), m3_mul(so3_Jr(jd), Jrbg))), -1.0);
    set_block(J, 30, 3, 18, m3_mul(JrInv, Jrj));
    // velocity rows
    set_block(J, 30, 6, 3, m3_mul(hat(Rb), Jri));
    set_block(J, 30, 6, 6, RiT, -1.0);
    set_block(J, 30, 6, 9, Jvbg, -1.0);
    set_block(J, 30, 6, 12, Jvba, -1.0);
    set_block(J, 30, 6, 21, RiT);
    // bias rows
    for (int k = 0; k < 6; ++k) {
        J[(9 + k) * 30 + 9 + k] = -1.0;
        J[(9 + k) * 30 + 24 + k] = 1.0;
    }
}


// MarginalizationFactor::Evaluate (ceresfunc.h:262-301): residual at x, the (constant) Jacobian is P.J.  P: anything
// with J (15 x 15 row-major), r0, x0 -- mml_prior or the solver's copy of it.
template <class PriorT>
MML_HD void prior_residual(const PriorT& P, const double* x15, double* r) {
    double dx[15];
    for (int k = 0; k < 3; ++k) dx[k] = x15[k] - P.x0[k];
    const M3 E = m3_mul(m3_t(so3_exp(x15 + 3)), so3_exp(P.x0 + 3));  // exp(x)^-1 * exp(x0)  (:279)
    so3_log(E, dx + 3);
    for (int k = 6; k < 15; ++k) dx[k] = x15[k] - P.x0[k];
    for (int i = 0; i < 15; ++i) {
        double s = P.r0[i];
        for (int k = 0; k < 15; ++k) s += P.J[i * 15 + k] * dx[k];
        r[i] = s;
    }
}

}  // namespace
