// undistort_dev.h -- the per-point arithmetic of RemoveLidarDistortion (mm-loam/src/unionPoseEstimation.cpp:402-421),
// shared by k_undistort (undistort_voxel.hip) and by the bucketing pass of the fused step (feature.hip), which writes
// the fused cloud already undistorted when the sweep motion is known up front.  Compiled with -ffp-contract=off.
#ifndef MML_UNDISTORT_DEV_H
#define MML_UNDISTORT_DEV_H
#include <hip/hip_runtime.h>
#include <math.h>

#include "mml_internal.h"

namespace mml_und {

struct Q4 {
    double x, y, z, w;
};
struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 v3(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 vcross(const V3& a, const V3& b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// Eigen Vector4d reduction with 2-wide packets: (c0 + c2) + (c1 + c3), coefficients stored x,y,z,w
__device__ __forceinline__ double qdot(const Q4& a, const Q4& b) {
    return (a.x * b.x + a.z * b.z) + (a.y * b.y + a.w * b.w);
}
__device__ __forceinline__ Q4 qnormalized(const Q4& q) {
    double n = sqrt(qdot(q, q));
    return Q4{q.x / n, q.y / n, q.z / n, q.w / n};
}
// Eigen quaternionbase_assign_impl<Matrix3d>: rotation matrix (row-major) -> quaternion
__device__ __forceinline__ Q4 quat_from_matrix(const double* m) {
    Q4 q;
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m[7] - m[5]) * t;
        q.y = (m[2] - m[6]) * t;
        q.z = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        int j = (i + 1) % 3;
        int k = (j + 1) % 3;
        double qv[3];
        t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
        qv[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (m[3 * k + j] - m[3 * j + k]) * t;
        qv[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        qv[k] = (m[3 * k + i] + m[3 * i + k]) * t;
        q.x = qv[0];
        q.y = qv[1];
        q.z = qv[2];
    }
    return q;
}

__device__ __forceinline__ double rsqrt_nr(double z) {
    double y = __builtin_amdgcn_rsq(z);
    y = y * (1.5 - (0.5 * z) * (y * y));
    y = y * (1.5 - (0.5 * z) * (y * y));
    return y;
}
// Is the double v farther than tol from every float rounding boundary (the midpoints between adjacent floats)?  A float
// keeps the top 23 of the 52 mantissa bits; the midpoint of v's float cell is the low 29 bits == 2^28, so the distance
// is |low29 - 2^28| units of 2^(e-52), read straight from the bit pattern.  Values below the normal float range are
// never declared safe.
__device__ __forceinline__ bool float_round_safe(double v, double tol) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const int ef = (int)((hi >> 20) & 0x7ffu);
    const int d = abs((int)(lo & 0x1fffffffu) - 0x10000000);
    const double unit = __hiloint2double((ef - 52) << 20, 0);
    return (ef >= 1023 - 126) && ((double)d * unit > tol);
}

// The per-point arithmetic of RemoveLidarDistortion in double, written exactly as the reference (slerp with two
// divisions, normalized() with sqrt + 4 divisions).
__device__ __forceinline__ void undistort_exact(const double* dR, const double* dt, const double* dv, float s, float4& p) {
    const Q4 qlc{dv[0], dv[1], dv[2], dv[3]};
    const double dd = (0.0 * qlc.x + 0.0 * qlc.z) + (0.0 * qlc.y + 1.0 * qlc.w);
    double scale0, scale1;
    const double t = s;
    if (dv[7] != 0.0) {
        scale0 = 1.0 - t;
        scale1 = t;
    } else {
        const double theta = dv[4], sinTheta = dv[5];
        scale0 = sin((1.0 - t) * theta) / sinTheta;
        scale1 = sin((t * theta)) / sinTheta;
    }
    if (dd < 0.0) scale1 = -scale1;
    Q4 q{scale0 * 0.0 + scale1 * qlc.x, scale0 * 0.0 + scale1 * qlc.y, scale0 * 0.0 + scale1 * qlc.z,
         scale0 * 1.0 + scale1 * qlc.w};
    const Q4 dq = qnormalized(q);
    V3 qv = v3(dq.x, dq.y, dq.z);
    V3 v = v3(p.x, p.y, p.z);
    V3 uv = vcross(qv, v);
    uv = v3(uv.x + uv.x, uv.y + uv.y, uv.z + uv.z);
    V3 c2 = vcross(qv, uv);
    V3 startP = v3((v.x + dq.w * uv.x) + c2.x, (v.y + dq.w * uv.y) + c2.y, (v.z + dq.w * uv.z) + c2.z);
    startP = v3(startP.x + s * dt[0], startP.y + s * dt[1], startP.z + s * dt[2]);
    V3 w = v3(startP.x - dt[0], startP.y - dt[1], startP.z - dt[2]);
    p.x = (dR[0] * w.x + dR[3] * w.y) + dR[6] * w.z;
    p.y = (dR[1] * w.x + dR[4] * w.y) + dR[7] * w.z;
    p.z = (dR[2] * w.x + dR[5] * w.y) + dR[8] * w.z;
}

// One point.  dR / dt: the 12 per-scan doubles; dv: the 8 derived doubles of k_undistort_prep (qlc x,y,z,w | theta |
// sinTheta | 1/sinTheta | linear-branch flag); s: the point's in-sweep time.  Fast form first (closed-form slerp of a
// unit quaternion, small-angle sin / cos polynomials), the reference expression when a coordinate of the double
// result lies within 1e-13 (relative to the input scale) of a float rounding boundary.
__device__ __forceinline__ void undistort_point(const double* dR, const double* dt, const double* dv, float s, float4& p) {
    const double t = s;
    const double qx = dv[0], qy = dv[1], qz = dv[2], qw = dv[3];
    double ax, ay, az, aw;
    const double theta = dv[4];
    if (dv[7] == 0.0 && theta < 0.5) {
        // slerp(Identity, q, t) of a unit quaternion q = (sin(theta) n, +-cos(theta)) is (+-sin(t theta) n, cos(t theta)),
        // already of unit length: the two sines, the divisions and the normalisation of the reference expression
        // collapse to one sine / cosine of a small angle (theta = half the rotation over one sweep), evaluated here by
        // their Taylor polynomials (truncation < 1e-18 for theta < 0.5).  Like the other fast forms the result is
        // only trusted away from float rounding boundaries (below).
        const double x = t * theta, z = x * x;
        double ps = sconst(-1.0 / 1307674368000.0);
        ps = __builtin_fma(ps, z, sconst(1.0 / 6227020800.0));
        ps = __builtin_fma(ps, z, sconst(-1.0 / 39916800.0));
        ps = __builtin_fma(ps, z, sconst(1.0 / 362880.0));
        ps = __builtin_fma(ps, z, sconst(-1.0 / 5040.0));
        ps = __builtin_fma(ps, z, sconst(1.0 / 120.0));
        ps = __builtin_fma(ps, z, sconst(-1.0 / 6.0));
        const double sn = __builtin_fma(x * z, ps, x);
        double pc = sconst(-1.0 / 87178291200.0);
        pc = __builtin_fma(pc, z, sconst(1.0 / 479001600.0));
        pc = __builtin_fma(pc, z, sconst(-1.0 / 3628800.0));
        pc = __builtin_fma(pc, z, sconst(1.0 / 40320.0));
        pc = __builtin_fma(pc, z, sconst(-1.0 / 720.0));
        pc = __builtin_fma(pc, z, sconst(1.0 / 24.0));
        pc = __builtin_fma(pc, z, -0.5);
        aw = __builtin_fma(z, pc, 1.0);
        const double k = (qw < 0.0 ? -dv[6] : dv[6]) * sn;
        ax = k * qx;
        ay = k * qy;
        az = k * qz;
    } else {
        double scale0, scale1;
        if (dv[7] != 0.0) {
            scale0 = 1.0 - t;
            scale1 = t;
        } else {
            const double inv = dv[6];
            scale0 = sin((1.0 - t) * theta) * inv;
            scale1 = sin((t * theta)) * inv;
        }
        if (qw < 0.0) scale1 = -scale1;
        ax = scale1 * qx;
        ay = scale1 * qy;
        az = scale1 * qz;
        aw = scale0 + scale1 * qw;
        const double inv_n = rsqrt_nr((ax * ax + az * az) + (ay * ay + aw * aw));
        ax *= inv_n;
        ay *= inv_n;
        az *= inv_n;
        aw *= inv_n;
    }
    // (this is the guarded form: its double result only has to be within `tol` of the reference expression, so the products
    //  are fused -- two thirds of the instructions of the separate multiplies and adds the reference order needs)
    const double vx = p.x, vy = p.y, vz = p.z;
    double ux = __builtin_fma(ay, vz, -(az * vy)), uy = __builtin_fma(az, vx, -(ax * vz)), uz = __builtin_fma(ax, vy, -(ay * vx));
    ux += ux;
    uy += uy;
    uz += uz;
    const double cx = __builtin_fma(ay, uz, -(az * uy)), cy = __builtin_fma(az, ux, -(ax * uz)), cz = __builtin_fma(ax, uy, -(ay * ux));
    const double sm1 = (double)s - 1.0;  // exact
    const double wx = __builtin_fma(sm1, dt[0], __builtin_fma(aw, ux, vx) + cx);
    const double wy = __builtin_fma(sm1, dt[1], __builtin_fma(aw, uy, vy) + cy);
    const double wz = __builtin_fma(sm1, dt[2], __builtin_fma(aw, uz, vz) + cz);
    const double ox = __builtin_fma(dR[6], wz, __builtin_fma(dR[3], wy, dR[0] * wx));
    const double oy = __builtin_fma(dR[7], wz, __builtin_fma(dR[4], wy, dR[1] * wx));
    const double oz = __builtin_fma(dR[8], wz, __builtin_fma(dR[5], wy, dR[2] * wx));
    const double tol = 1e-13 * (((fabs(vx) + fabs(vy)) + fabs(vz)) + ((fabs(dt[0]) + fabs(dt[1])) + fabs(dt[2])) + 1e-30);
    if (float_round_safe(ox, tol) & float_round_safe(oy, tol) & float_round_safe(oz, tol)) {
        p.x = ox;
        p.y = oy;
        p.z = oz;
    } else {
        undistort_exact(dR, dt, dv, s, p);
    }
}

}  // namespace mml_und
#endif
