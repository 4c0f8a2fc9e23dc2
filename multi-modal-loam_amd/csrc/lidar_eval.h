// lidar_eval.h -- the per-frame lidar evaluation shared by the device solvers (solve.hip: k_solve, k_linearize,
// k_window_round; fullwindow_dev.hip: the 15 W window with IMU factors): pose construction, the two residual types
// with their analytic Jacobians and Huber correction, and the 28-value workgroup reduction.
//   a17/a18 Cost_NavState_IMU_Line / Cost_NavState_IMU_Plan_Vec (mm-loam/include/utils/ceresfunc.h:397-458, 517-570)
//   a19     analytic Jacobians replacing ceres::AutoDiffCostFunction<...,6> through Sophus::SO3<Jet>::exp
//           (include/sophus/so3.hpp:585-622): dP/dt = I, dP/dphi = -[R p_b]x J_l(phi)
//   a20     J^T J / J^T r with Ceres' Huber correction (corrector.cc: rho'' <= 0 => scale by sqrt(rho'))
#pragma once
#include <math.h>
#include <stddef.h>

#include "mml_internal.h"

namespace {

// Threads per one-frame evaluation (eval_frame + block_reduce28): every kernel that evaluates lidar factors -- k_solve, k_linearize,
// k_window_round, k_fw_eval -- uses this one number, so that all of them accumulate a frame's 28 sums in the same order (they are held
// bit-identical, tests/test_gpu_multi.py).  128 since round 5: with 256 three of a problem's four wavefronts idle through the
// trust-region proposal (40 % of k_solve's time) while holding their 256 registers; two-wavefront workgroups put four problems on a CU
// instead of two: k_solve 0.325 -> 0.226 ms per 1024 problems (64 threads: 0.321).  The price is the single-problem latency: the
// factor pass of ONE problem takes twice as long (configs[2]: 0.82 -> 0.85 ms per scan).
#ifndef MML_SOLVE_THREADS
#define MML_SOLVE_THREADS 128
#endif
constexpr int SOLVE_THREADS = MML_SOLVE_THREADS;
constexpr int SOLVE_WAVES = SOLVE_THREADS / 64;
constexpr int MAXW = 8;  // frames per window problem
constexpr double kLidarM = 1.5e-3;  // IMUIntegrator.h:83

struct Pose {
    double R[9];   // R_wl
    double t[3];   // t_wl
    double tb[3];  // t_wb (= x[0:3])
    double Jl[9];  // left Jacobian of SO(3) at phi
};

__device__ void quat_to_R(double qx, double qy, double qz, double qw, double* R) {
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R[0] = 1 - (tyy + tzz);
    R[1] = txy - twz;
    R[2] = txz + twy;
    R[3] = txy + twz;
    R[4] = 1 - (txx + tzz);
    R[5] = tyz - twx;
    R[6] = txz - twy;
    R[7] = tyz + twx;
    R[8] = 1 - (txx + tyy);
}

// x = [t, phi]; T_bl row-major 4x4.  sophus/so3.hpp:585-622 exp with the theta^2 < 1e-20 Taylor branch.
__device__ void make_pose(const double* x, const double* T_bl, Pose& P) {
    const double px = x[3], py = x[4], pz = x[5];
    const double th2 = (px * px + py * py) + pz * pz;
    double imag, real, a, b;
    if (th2 < 1e-10 * 1e-10) {
        double th4 = th2 * th2;
        imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        real = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
        a = 0.5;
        b = 1.0 / 6.0;
    } else {
        double th = sqrt(th2);
        double half = 0.5 * th;
        // (sincos: one argument reduction per angle instead of two -- every thread of every solver kernel forms the pose once per pass:
        //  batch solve 0.228 -> 0.220 ms per 1024 problems, B = 1 solve 0.123 -> 0.118 ms; $MML_POSE_SINCOS was the A/B switch)
        double sh, ch, st, ct;
        sincos(half, &sh, &ch);
        sincos(th, &st, &ct);
        imag = sh / th;
        real = ch;
        a = (1.0 - ct) / th2;
        b = (th - st) / (th2 * th);
    }
    double Rwb[9];
    quat_to_R(imag * px, imag * py, imag * pz, real, Rwb);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
            P.R[3 * r + c] = (Rwb[3 * r] * T_bl[c] + Rwb[3 * r + 1] * T_bl[4 + c]) + Rwb[3 * r + 2] * T_bl[8 + c];
        P.t[r] = ((Rwb[3 * r] * T_bl[3] + Rwb[3 * r + 1] * T_bl[7]) + Rwb[3 * r + 2] * T_bl[11]) + x[r];
        P.tb[r] = x[r];
    }
    const double K[9] = {0, -pz, py, pz, 0, -px, -py, px, 0};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double k2 = K[3 * r] * K[c] + K[3 * r + 1] * K[3 + c] + K[3 * r + 2] * K[6 + c];
            P.Jl[3 * r + c] = a * K[3 * r + c] + b * k2 + (r == c ? 1.0 : 0.0);
        }
}

// sqrt(x) and 1 / sqrt(x) together: v_rsq_f64 and two coupled Goldschmidt steps (nine instructions, ~1e-16 relative; the IEEE
// sqrt() and operator/ of the compiler are 28 and 14 -- a plane factor had three of each, 126 of its ~330 instructions).  The
// solvers are held to the oracle by tolerances (1e-9 on the iteration trace, 1e-11 on H and g), and to each other bit for bit:
// every device solver evaluates its factors through this header.
// x = 0 (a point exactly on its line / plane: ceres::sqrt gives the zero residual there, ceresfunc.h:426-437): the argument of the
// v_rsq_f64 is clamped at 1e-300, so that root = 0 * 1e150 = 0 and inv_root is a large FINITE number -- every caller multiplies it
// with a numerator that is exactly zero with x (the cross product, the offset d), and the factor then contributes a zero residual and
// a zero Jacobian row instead of NaN (the reference's Jet has a NaN derivative at that point; oracle/estimate.cpp takes the same
// zero-row convention).  One v_max_f64; below 1e-300 the root is only approximate, which no squared length in metres reaches.
__device__ __forceinline__ void sqrt_pair(double x, double& root, double& inv_root) {
    const double y = __builtin_amdgcn_rsq(fmax(x, 1e-300));
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    root = __builtin_fma(g, r, g);
    inv_root = 2.0 * __builtin_fma(h, r, h);
}

// ceres/loss_function.cc HuberLoss::Evaluate
__device__ __forceinline__ void huber(double s, double a, double& rho0, double& rho1) {
    rho0 = s;
    rho1 = 1.0;
    if (a > 0) {
        double bb = a * a;
        if (s > bb) {
            double r, ir;
            sqrt_pair(s, r, ir);
            rho0 = 2.0 * a * r - bb;
            rho1 = fmax(2.2250738585072014e-308, a * ir);
        }
    }
}

// accumulate rho1 * J J^T (upper triangle, 21) and rho1 * J r (6) for one scalar residual row with dr/dP = gr
__device__ __forceinline__ void row_jacobian(const Pose& P, const double* Pw, const double* gr, double* J) {
    // dP/dx = [I, -[Rpb]x Jl],  Rpb = P - t_wb
    const double rx = Pw[0] - P.tb[0], ry = Pw[1] - P.tb[1], rz = Pw[2] - P.tb[2];
    // gr^T * (-[Rpb]x) = (Rpb x gr)^T ... (-[a]x)^T g = a x g  => row = (gr x Rpb)?  use explicit form:
    // (-[r]x) = [[0, rz, -ry], [-rz, 0, rx], [ry, -rx, 0]] ; v^T = gr^T (-[r]x)
    // (fused multiply-adds throughout the factor evaluation: the solvers are bound to the oracle by tolerances and to each other
    //  by sharing this header; a factor is 40 % of k_solve's life and issue-bound on its SIMD)
    const double v0 = __builtin_fma(gr[2], ry, -(gr[1] * rz));
    const double v1 = __builtin_fma(gr[0], rz, -(gr[2] * rx));
    const double v2 = __builtin_fma(gr[1], rx, -(gr[0] * ry));
    J[0] = gr[0];
    J[1] = gr[1];
    J[2] = gr[2];
    J[3] = __builtin_fma(v2, P.Jl[6], __builtin_fma(v1, P.Jl[3], v0 * P.Jl[0]));
    J[4] = __builtin_fma(v2, P.Jl[7], __builtin_fma(v1, P.Jl[4], v0 * P.Jl[1]));
    J[5] = __builtin_fma(v2, P.Jl[8], __builtin_fma(v1, P.Jl[5], v0 * P.Jl[2]));
}

__device__ __forceinline__ void accum(double* acc, const double* J, double r, double w) {
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const double wj = w * J[a];
#pragma unroll
        for (int b = a; b < 6; ++b) {
            acc[k] = __builtin_fma(wj, J[b], acc[k]);
            ++k;
        }
        acc[21 + a] = __builtin_fma(wj, r, acc[21 + a]);
    }
}

// Evaluate one frame at pose P: thread-strided over the factors; acc[28] per thread.
// (forced inline: with a third caller in this header -- eval_frame_pairs -- the compiler stopped inlining it into k_solve, and the
//  batch solve paid 10 % for two real calls per pass)
__device__ __forceinline__ void eval_frame(const MmlLineFactor* lf, int nlf, const MmlPlaneFactor* pf, int npf, const Pose& P,
                           double w_tan, double huber_delta, double* acc) {
    for (int k = 0; k < 28; ++k) acc[k] = 0;
    const double ka = 1.0 / kLidarM;
    for (int i = threadIdx.x; i < nlf; i += SOLVE_THREADS) {
        const MmlLineFactor f = lf[i];
        if (f.src < 0 || !(fabs(f.error) > 1e-5)) continue;  // Estimator.cpp:1385
        const double cx = f.ori[0], cy = f.ori[1], cz = f.ori[2];
        const double ax = f.p1[0], ay = f.p1[1], az = f.p1[2], bx = f.p2[0], by = f.p2[1], bz = f.p2[2];
        double Pw[3];
        Pw[0] = __builtin_fma(P.R[2], cz, __builtin_fma(P.R[1], cy, __builtin_fma(P.R[0], cx, P.t[0])));
        Pw[1] = __builtin_fma(P.R[5], cz, __builtin_fma(P.R[4], cy, __builtin_fma(P.R[3], cx, P.t[1])));
        Pw[2] = __builtin_fma(P.R[8], cz, __builtin_fma(P.R[7], cy, __builtin_fma(P.R[6], cx, P.t[2])));
        double l12, il12, a012, ia012, s12, is12, rs, sm14;
        const double dx = ax - bx, dy = ay - by, dz = az - bz;
        sqrt_pair(__builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx)), l12, il12);
        const double pax = Pw[0] - ax, pay = Pw[1] - ay, paz = Pw[2] - az, pbx = Pw[0] - bx, pby = Pw[1] - by, pbz = Pw[2] - bz;
        const double c0 = __builtin_fma(pax, pby, -(pbx * pay));
        const double c1 = __builtin_fma(pax, pbz, -(pbx * paz));
        const double c2 = __builtin_fma(pay, pbz, -(pby * paz));
        sqrt_pair(__builtin_fma(c2, c2, __builtin_fma(c1, c1, c0 * c0)), a012, ia012);
        const double ld2 = a012 * il12;
        const double s = __builtin_fma(Pw[2], Pw[2], __builtin_fma(Pw[1], Pw[1], Pw[0] * Pw[0]));
        sqrt_pair(s, s12, is12);        // s^(1/2), s^(-1/2)
        sqrt_pair(s12, rs, sm14);       // s^(1/4), s^(-1/4)
        const double weight = 1.0 - 0.9 * fabs(ld2) * sm14;
        const double r = ka * weight * ld2;
        // gradient of ld wrt P: ((a-b) x u_hat) / l12, u = (c2, -c1, c0)
        const double ux = c2 * ia012, uy = -c1 * ia012, uz = c0 * ia012;
        double gl[3] = {__builtin_fma(dy, uz, -(dz * uy)) * il12, __builtin_fma(dz, ux, -(dx * uz)) * il12, __builtin_fma(dx, uy, -(dy * ux)) * il12};
        const double sm54 = sm14 * (is12 * is12);
        double gr[3];
        for (int c = 0; c < 3; ++c) {
            double gw = (-0.9) * __builtin_fma(fabs(ld2) * (-0.5) * sm54, Pw[c], sm14 * gl[c]);
            gr[c] = ka * __builtin_fma(ld2, gw, weight * gl[c]);
        }
        double J[6];
        row_jacobian(P, Pw, gr, J);
        double rho0, rho1;
        huber(r * r, huber_delta, rho0, rho1);
        acc[27] += 0.5 * rho0;
        accum(acc, J, r, rho1);
    }
    const double kb = w_tan / kLidarM;
    // (the next record is requested before the current one is worked on: one exposed memory round trip per pass
    //  instead of one per factor)
    MmlPlaneFactor nxt;
    if ((int)threadIdx.x < npf) nxt = pf[threadIdx.x];
    for (int i = threadIdx.x; i < npf; i += SOLVE_THREADS) {
        const MmlPlaneFactor f = nxt;
        if (i + SOLVE_THREADS < npf) nxt = pf[i + SOLVE_THREADS];
        if (f.src < 0 || !(fabs(f.error) > 1e-5)) continue;  // Estimator.cpp:1396
        const double cx = f.ori[0], cy = f.ori[1], cz = f.ori[2];
        double Pw[3];
        Pw[0] = __builtin_fma(P.R[2], cz, __builtin_fma(P.R[1], cy, __builtin_fma(P.R[0], cx, P.t[0])));
        Pw[1] = __builtin_fma(P.R[5], cz, __builtin_fma(P.R[4], cy, __builtin_fma(P.R[3], cx, P.t[1])));
        Pw[2] = __builtin_fma(P.R[8], cz, __builtin_fma(P.R[7], cy, __builtin_fma(P.R[6], cx, P.t[2])));
        const double d[3] = {Pw[0] - f.proj[0], Pw[1] - f.proj[1], Pw[2] - f.proj[2]};
        double nd, ind, s12, is12, rs, sm14;
        sqrt_pair(__builtin_fma(d[2], d[2], __builtin_fma(d[1], d[1], d[0] * d[0])), nd, ind);
        const double s = __builtin_fma(Pw[2], Pw[2], __builtin_fma(Pw[1], Pw[1], Pw[0] * Pw[0]));
        sqrt_pair(s, s12, is12);   // s^(1/2), s^(-1/2)
        sqrt_pair(s12, rs, sm14);  // s^(1/4), s^(-1/4)
        const double weight = 1.0 - 0.9 * nd * sm14;
        const double sm54 = sm14 * (is12 * is12);
        double gw[3];
        const double gwa = sm14 * ind, gwb = nd * (-0.5) * sm54;
        for (int c = 0; c < 3; ++c) gw[c] = (-0.9) * __builtin_fma(gwb, Pw[c], gwa * d[c]);
        // e = weight * d ;  de/dP = weight I + d gw^T ;  row^T de/dP = weight row + (row . d) gw
        const double w[3] = {f.omega[0], f.omega[1], f.omega[2]};
        double rows[3][3];
        int nrows = 1;
        rows[0][0] = ka * w[0];
        rows[0][1] = ka * w[1];
        rows[0][2] = ka * w[2];
        if (kb != 0.0) {
            // deterministic tangent basis (any orthonormal completion gives the same H, g, cost)
            double h[3] = {0, 0, 0};
            if (fabs(w[0]) <= fabs(w[1]) && fabs(w[0]) <= fabs(w[2]))
                h[0] = 1;
            else if (fabs(w[1]) <= fabs(w[2]))
                h[1] = 1;
            else
                h[2] = 1;
            double t0 = w[1] * h[2] - w[2] * h[1], t1 = w[2] * h[0] - w[0] * h[2], t2 = w[0] * h[1] - w[1] * h[0];
            double n, in;
            sqrt_pair((t0 * t0 + t1 * t1) + t2 * t2, n, in);
            t0 *= in;
            t1 *= in;
            t2 *= in;
            rows[1][0] = kb * t0;
            rows[1][1] = kb * t1;
            rows[1][2] = kb * t2;
            rows[2][0] = kb * (w[1] * t2 - w[2] * t1);
            rows[2][1] = kb * (w[2] * t0 - w[0] * t2);
            rows[2][2] = kb * (w[0] * t1 - w[1] * t0);
            nrows = 3;
        }
        double rr[3], sq = 0;
        for (int q = 0; q < nrows; ++q) {
            rr[q] = weight * __builtin_fma(rows[q][2], d[2], __builtin_fma(rows[q][1], d[1], rows[q][0] * d[0]));
            sq = __builtin_fma(rr[q], rr[q], sq);
        }
        double rho0, rho1;
        huber(sq, huber_delta, rho0, rho1);
        acc[27] += 0.5 * rho0;
        for (int q = 0; q < nrows; ++q) {
            const double rd = __builtin_fma(rows[q][2], d[2], __builtin_fma(rows[q][1], d[1], rows[q][0] * d[0]));
            double gr[3] = {__builtin_fma(rd, gw[0], weight * rows[q][0]), __builtin_fma(rd, gw[1], weight * rows[q][1]),
                            __builtin_fma(rd, gw[2], weight * rows[q][2])};
            double J[6];
            row_jacobian(P, Pw, gr, J);
            accum(acc, J, rr[q], rho1);
        }
    }
}

// ---- the same evaluation with TWO plane factors of a thread in flight (the live path's launches: k_solve<true>) -----------------
// One problem per CU leaves one wavefront per SIMD, and a factor is a chain of ~350 dependent double-precision instructions: the
// SIMD waits out every one of them.  Here a thread forms the rows of its factors i and i + SOLVE_THREADS side by side -- two
// independent chains the scheduler interleaves -- and then adds them to its 28 sums in the order eval_frame adds them (i first).
// Every factor goes through the same operations in the same order as in eval_frame (plane_row restates its loop body; the two
// are held bit-identical by tests/test_gpu_shapes.py: a slot solved alone against the same slot in a 4096-slot launch).
// (for plan_weight_tan = 0 -- the one-frame mode's only setting, Estimator.cpp:1206 --: a plane factor is ONE residual row, and
//  everything below is straight-line code on registers; with the tangential rows the callers keep eval_frame)
struct PlaneRow {
    double J[6], rr, rho0, rho1;
    bool valid;  // false: the factor is skipped (no record, or |error| <= 1e-5)
};
__device__ __forceinline__ void plane_row(const MmlPlaneFactor& f, const Pose& P, double ka, double huber_delta, PlaneRow& o) {
    o.valid = !(f.src < 0 || !(fabs(f.error) > 1e-5));  // Estimator.cpp:1396
    const double cx = f.ori[0], cy = f.ori[1], cz = f.ori[2];
    double Pw[3];
    Pw[0] = __builtin_fma(P.R[2], cz, __builtin_fma(P.R[1], cy, __builtin_fma(P.R[0], cx, P.t[0])));
    Pw[1] = __builtin_fma(P.R[5], cz, __builtin_fma(P.R[4], cy, __builtin_fma(P.R[3], cx, P.t[1])));
    Pw[2] = __builtin_fma(P.R[8], cz, __builtin_fma(P.R[7], cy, __builtin_fma(P.R[6], cx, P.t[2])));
    const double d[3] = {Pw[0] - f.proj[0], Pw[1] - f.proj[1], Pw[2] - f.proj[2]};
    double nd, ind, s12, is12, rs, sm14;
    sqrt_pair(__builtin_fma(d[2], d[2], __builtin_fma(d[1], d[1], d[0] * d[0])), nd, ind);
    const double s = __builtin_fma(Pw[2], Pw[2], __builtin_fma(Pw[1], Pw[1], Pw[0] * Pw[0]));
    sqrt_pair(s, s12, is12);
    sqrt_pair(s12, rs, sm14);
    const double weight = 1.0 - 0.9 * nd * sm14;
    const double sm54 = sm14 * (is12 * is12);
    double gw[3];
    const double gwa = sm14 * ind, gwb = nd * (-0.5) * sm54;
#pragma unroll
    for (int c = 0; c < 3; ++c) gw[c] = (-0.9) * __builtin_fma(gwb, Pw[c], gwa * d[c]);
    const double row[3] = {ka * (double)f.omega[0], ka * (double)f.omega[1], ka * (double)f.omega[2]};
    o.rr = weight * __builtin_fma(row[2], d[2], __builtin_fma(row[1], d[1], row[0] * d[0]));
    double sq = 0;
    sq = __builtin_fma(o.rr, o.rr, sq);
    huber(sq, huber_delta, o.rho0, o.rho1);
    const double rd = __builtin_fma(row[2], d[2], __builtin_fma(row[1], d[1], row[0] * d[0]));
    double gr[3] = {__builtin_fma(rd, gw[0], weight * row[0]), __builtin_fma(rd, gw[1], weight * row[1]), __builtin_fma(rd, gw[2], weight * row[2])};
    row_jacobian(P, Pw, gr, o.J);
}
// plan_weight_tan must be 0 (one residual row per plane factor)
__device__ __forceinline__ void eval_frame_pairs(const MmlLineFactor* lf, int nlf, const MmlPlaneFactor* pf, int npf, const Pose& P,
                                 double huber_delta, double* acc) {
    // the line factors (a fifth of the records) as eval_frame takes them: its loop runs with zero plane factors
    eval_frame(lf, nlf, pf, 0, P, 0.0, huber_delta, acc);
    const double ka = 1.0 / kLidarM;
    for (int i = threadIdx.x; i < npf; i += 2 * SOLVE_THREADS) {
        const bool hb = i + SOLVE_THREADS < npf;
        const MmlPlaneFactor fa = pf[i];
        const MmlPlaneFactor fb = pf[hb ? i + SOLVE_THREADS : i];
        PlaneRow a, b;
        plane_row(fa, P, ka, huber_delta, a);
        plane_row(fb, P, ka, huber_delta, b);
        if (a.valid) {
            acc[27] += 0.5 * a.rho0;
            accum(acc, a.J, a.rr, a.rho1);
        }
        if (hb && b.valid) {
            acc[27] += 0.5 * b.rho0;
            accum(acc, b.J, b.rr, b.rho1);
        }
    }
}

// block reduction of acc[28] into out[28] (LDS or global); result valid for thread 0 after the trailing barrier.
// Inside a wavefront the 28 sums are reduced as a butterfly that halves the number of values a lane carries at every
// step (at offset o the lanes with bit o set keep the upper half of their values and hand over the lower half): 16 + 8 +
// 4 + 2 + 1 + 1 = 32 exchanged doubles instead of 28 x 6.  Every value still meets its partners in the order xor 32, 16,
// 8, 4, 2, 1, i.e. it is the same summation tree as one xor-butterfly per value, bit for bit.
__device__ __forceinline__ double shfl_xor_f64(double v, int o) {
    const unsigned lo = __shfl_xor((unsigned)__double2loint(v), o), hi = __shfl_xor((unsigned)__double2hiint(v), o);
    return __hiloint2double((int)hi, (int)lo);
}
// value of lane (l ^ O) without the LDS crossbar: gfx950's v_permlane32_swap / v_permlane16_swap (halves of the wavefront, rows of
// 16 lanes) and DPP moves inside a row (xor 8 = half-mirror o mirror, xor 4 = quad-reverse o half-mirror, xor 2 / 1 = quad_perm).
// A ds_bpermute round trip is ~300 cycles and block_reduce28's tree has six of them one after the other; these are a few VALU
// instructions each.
template <int O>
__device__ __forceinline__ unsigned lane_xor_u32(unsigned x, int lane) {
    if constexpr (O == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);  // r[0]: both halves = the lower, r[1]: = the upper
        return lane < 32 ? r[1] : r[0];
    } else if constexpr (O == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);  // r[0]: odd rows = the even row below, r[1]: even rows = the odd row above
        return (lane & 16) ? r[0] : r[1];
    } else if constexpr (O == 8) {
        const int m = __builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, false);        // row_mirror: i -> 15 - i
        return (unsigned)__builtin_amdgcn_update_dpp(0, m, 0x141, 0xf, 0xf, false);          // row_half_mirror: -> 8 (i / 8) + 7 - i % 8
    } else if constexpr (O == 4) {
        const int m = __builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false);
        return (unsigned)__builtin_amdgcn_update_dpp(0, m, 0x1B, 0xf, 0xf, false);           // quad_perm [3, 2, 1, 0]
    } else if constexpr (O == 2) {
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false);      // quad_perm [2, 3, 0, 1]
    } else {
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false);      // quad_perm [1, 0, 3, 2]
    }
}
template <int O>
__device__ __forceinline__ double lane_xor_f64(double v, int lane) {
    const unsigned lo = lane_xor_u32<O>((unsigned)__double2loint(v), lane), hi = lane_xor_u32<O>((unsigned)__double2hiint(v), lane);
    return __hiloint2double((int)hi, (int)lo);
}

#ifdef MML_REDUCE_DPP
template <int C, int O>
__device__ __forceinline__ void halve_step(double* v, int lane, int& idx) {
    const bool up = (lane & O) != 0;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const double keep = up ? v[j + C] : v[j], send = up ? v[j] : v[j + C];
        v[j] = keep + lane_xor_f64<O>(send, lane);
    }
    idx += up ? C : 0;
}
#endif
__device__ void block_reduce28(double* acc, double* s_part /*SOLVE_WAVES*28*/, double* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = k < 28 ? acc[k] : 0.0;
    int idx = 0;  // which of the 32 sums this lane ends up holding
#ifdef MML_REDUCE_DPP
    halve_step<16, 32>(v, lane, idx);
    halve_step<8, 16>(v, lane, idx);
    halve_step<4, 8>(v, lane, idx);
    halve_step<2, 4>(v, lane, idx);
    halve_step<1, 2>(v, lane, idx);
    const double tot = v[0] + lane_xor_f64<1>(v[0], lane);
#else
#pragma unroll
    for (int c = 16, o = 32; c >= 1; c >>= 1, o >>= 1) {
        const bool up = (lane & o) != 0;
#pragma unroll
        for (int j = 0; j < c; ++j) {
            const double keep = up ? v[j + c] : v[j], send = up ? v[j] : v[j + c];
            v[j] = keep + shfl_xor_f64(send, o);
        }
        idx += up ? c : 0;
    }
    const double tot = v[0] + shfl_xor_f64(v[0], 1);
#endif
    if (!(lane & 1) && idx < 28) s_part[wave * 28 + idx] = tot;
    __syncthreads();
    if (threadIdx.x < 28) {
        double r = s_part[threadIdx.x];
        for (int w = 1; w < SOLVE_WAVES; ++w) r += s_part[w * 28 + threadIdx.x];
        out[threadIdx.x] = r;
    }
    __syncthreads();
}

// ---- the live path's factor pass on all FOUR SIMDs of its CU (k_solve_wide: W = 1, plan_weight_tan = 0) ---------------------------
// A launch of at most one problem per CU runs the 128-thread pass above on two of the CU's four SIMDs, and the pass is bound by the
// double-precision issue rate there (1013 factors of configs[1] x ~350 instructions = 2 800 wave-instructions per SIMD at 4.1 cycles).
// Here the workgroup has 4 x 128 threads: thread (vt, sub) stands for a quarter of the work of thread vt of the 128-thread pass.
//   rows:  the factors of "virtual thread" vt, in the order eval_frame takes them (lines vt, vt + 128, ..., then planes vt, vt + 128,
//          ...), are its SLOTS 0, 1, 2, ...; thread (vt, sub) forms the rows (J, r, rho) of slots sub, sub + 4, sub + 8 of a round of
//          twelve -- line_row / plane_row: the operations of eval_frame's loop bodies in their order -- two plane rows side by side
//          where it can, and leaves them in LDS (9 doubles a row).
//   sums:  every one of the 28 sums of virtual thread vt is its own chain of fma(w J_a, J_b, acc) over the slots in order; the
//          chains are dealt to the four sub-threads by Jacobian row (wide_sums: a = 0 + cost | a = 1, 5 | a = 2, 4 | a = 3), each
//          walks the round's rows in slot order: every sum sees the same operands in the same order as acc[k] of thread vt in
//          eval_frame.
//   tree:  the 128 partial sums of a value meet in block_reduce28's order (xor 32, 16, 8, 4, 2, 1 inside the wavefront, then
//          wavefront 0 + wavefront 1), eight values a lane instead of 32, the partner's value by v_permlane*_swap / DPP.
// The results are bit-identical to eval_frame + block_reduce28 (tests/test_gpu_shapes.py (3): slots solved alone and in launches of
// <= 256 against the 4096-slot launches).
constexpr int WIDE_SUBS = 4;
constexpr int WIDE_THREADS = WIDE_SUBS * SOLVE_THREADS;
constexpr int WIDE_ROUND = 3 * WIDE_SUBS;  // slots (rows per virtual thread) per round
constexpr int ROW_DOUBLES = 9;             // J[6], r, rho0, rho1 (-1: no row)
constexpr int WIDE_ROW_LDS = WIDE_ROUND * ROW_DOUBLES * SOLVE_THREADS;  // doubles
constexpr int WIDE_SAVE_LDS = 8 * WIDE_THREADS;                          // doubles: the sums between the rounds of a long scan
static_assert(SOLVE_WAVES == 2, "eval_frame_wide restates block_reduce28 for two wavefronts");

__device__ __forceinline__ void line_row(const MmlLineFactor& f, const Pose& P, double ka, double huber_delta, PlaneRow& o) {
    o.valid = !(f.src < 0 || !(fabs(f.error) > 1e-5));  // Estimator.cpp:1385
    const double cx = f.ori[0], cy = f.ori[1], cz = f.ori[2];
    const double ax = f.p1[0], ay = f.p1[1], az = f.p1[2], bx = f.p2[0], by = f.p2[1], bz = f.p2[2];
    double Pw[3];
    Pw[0] = __builtin_fma(P.R[2], cz, __builtin_fma(P.R[1], cy, __builtin_fma(P.R[0], cx, P.t[0])));
    Pw[1] = __builtin_fma(P.R[5], cz, __builtin_fma(P.R[4], cy, __builtin_fma(P.R[3], cx, P.t[1])));
    Pw[2] = __builtin_fma(P.R[8], cz, __builtin_fma(P.R[7], cy, __builtin_fma(P.R[6], cx, P.t[2])));
    double l12, il12, a012, ia012, s12, is12, rs, sm14;
    const double dx = ax - bx, dy = ay - by, dz = az - bz;
    sqrt_pair(__builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx)), l12, il12);
    const double pax = Pw[0] - ax, pay = Pw[1] - ay, paz = Pw[2] - az, pbx = Pw[0] - bx, pby = Pw[1] - by, pbz = Pw[2] - bz;
    const double c0 = __builtin_fma(pax, pby, -(pbx * pay));
    const double c1 = __builtin_fma(pax, pbz, -(pbx * paz));
    const double c2 = __builtin_fma(pay, pbz, -(pby * paz));
    sqrt_pair(__builtin_fma(c2, c2, __builtin_fma(c1, c1, c0 * c0)), a012, ia012);
    const double ld2 = a012 * il12;
    const double s = __builtin_fma(Pw[2], Pw[2], __builtin_fma(Pw[1], Pw[1], Pw[0] * Pw[0]));
    sqrt_pair(s, s12, is12);
    sqrt_pair(s12, rs, sm14);
    const double weight = 1.0 - 0.9 * fabs(ld2) * sm14;
    const double r = ka * weight * ld2;
    const double ux = c2 * ia012, uy = -c1 * ia012, uz = c0 * ia012;
    double gl[3] = {__builtin_fma(dy, uz, -(dz * uy)) * il12, __builtin_fma(dz, ux, -(dx * uz)) * il12, __builtin_fma(dx, uy, -(dy * ux)) * il12};
    const double sm54 = sm14 * (is12 * is12);
    double gr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double gw = (-0.9) * __builtin_fma(fabs(ld2) * (-0.5) * sm54, Pw[c], sm14 * gl[c]);
        gr[c] = ka * __builtin_fma(ld2, gw, weight * gl[c]);
    }
    row_jacobian(P, Pw, gr, o.J);
    huber(r * r, huber_delta, o.rho0, o.rho1);
    o.rr = r;
}

struct WideRec {
    uint4 q[4];
};
static_assert(sizeof(MmlLineFactor) == 48 && sizeof(MmlPlaneFactor) == 64, "WideRec holds either record");
__device__ __forceinline__ void wide_load(WideRec& r, bool is_line, const MmlLineFactor* lf, int il, const MmlPlaneFactor* pf, int ip) {
    if (is_line) {
        const uint4* p = reinterpret_cast<const uint4*>(lf + il);
        r.q[0] = p[0];
        r.q[1] = p[1];
        r.q[2] = p[2];
    } else {
        const uint4* p = reinterpret_cast<const uint4*>(pf + ip);
        r.q[0] = p[0];
        r.q[1] = p[1];
        r.q[2] = p[2];
        r.q[3] = p[3];
    }
}
// (field by field: a memcpy of the record makes a private copy that the compiler then keeps in LDS)
__device__ __forceinline__ MmlLineFactor wide_line(const WideRec& r) {
    MmlLineFactor f;
    f.ori[0] = __uint_as_float(r.q[0].x);
    f.ori[1] = __uint_as_float(r.q[0].y);
    f.ori[2] = __uint_as_float(r.q[0].z);
    f.p1[0] = __uint_as_float(r.q[0].w);
    f.p1[1] = __uint_as_float(r.q[1].x);
    f.p1[2] = __uint_as_float(r.q[1].y);
    f.p2[0] = __uint_as_float(r.q[1].z);
    f.p2[1] = __uint_as_float(r.q[1].w);
    f.p2[2] = __uint_as_float(r.q[2].x);
    f.src = (int)r.q[2].y;
    f.error = __hiloint2double((int)r.q[2].w, (int)r.q[2].z);
    return f;
}
__device__ __forceinline__ MmlPlaneFactor wide_plane(const WideRec& r) {
    MmlPlaneFactor f;
    f.ori[0] = __uint_as_float(r.q[0].x);
    f.ori[1] = __uint_as_float(r.q[0].y);
    f.ori[2] = __uint_as_float(r.q[0].z);
    f.omega[0] = __uint_as_float(r.q[0].w);
    f.omega[1] = __uint_as_float(r.q[1].x);
    f.omega[2] = __uint_as_float(r.q[1].y);
    f.proj[0] = __hiloint2double((int)r.q[1].w, (int)r.q[1].z);
    f.proj[1] = __hiloint2double((int)r.q[2].y, (int)r.q[2].x);
    f.proj[2] = __hiloint2double((int)r.q[2].w, (int)r.q[2].z);
    f.error = __hiloint2double((int)r.q[3].y, (int)r.q[3].x);
    f.src = (int)r.q[3].z;
    f._pad = 0;
    return f;
}
static_assert(offsetof(MmlLineFactor, src) == 36 && offsetof(MmlLineFactor, error) == 40 && offsetof(MmlPlaneFactor, proj) == 24 &&
                  offsetof(MmlPlaneFactor, error) == 48 && offsetof(MmlPlaneFactor, src) == 56,
              "wide_line / wide_plane unpack the records by offset");

__device__ __forceinline__ void row_store(double* s_rows, int j, int vt, const PlaneRow& o, bool in_range) {
    double* p = s_rows + (size_t)j * ROW_DOUBLES * SOLVE_THREADS + vt;
#pragma unroll
    for (int k = 0; k < 6; ++k) p[k * SOLVE_THREADS] = o.J[k];
    p[6 * SOLVE_THREADS] = o.rr;
    p[7 * SOLVE_THREADS] = o.rho0;
    p[8 * SOLVE_THREADS] = (o.valid && in_range) ? o.rho1 : -1.0;  // rho' > 0 for every row (huber: >= DBL_MIN); -1: the factor is skipped
}

// the chains of sub-thread SUB over the ns rows of a round, in slot order: T slots a turn, their values requested together (an LDS
// round trip is a few hundred cycles with every wavefront of the workgroup reading: one per turn instead of two per slot), straight-
// line code.  The 28 chains are dealt to the four sub-threads by Jacobian row:
//   SUB 0: a = 0 (sums 0 .. 5, 21) + cost (27) | SUB 1: a = 1 (6 .. 10, 22), a = 5 (20, 26) | SUB 2: a = 2 (11 .. 14, 23), a = 4 (18, 19, 25)
//   | SUB 3: a = 3 (15 .. 17, 24)
// (measured, cycles per pass of configs[1]'s 187 + 826 factors: four summing sub-threads with a select per sum and slot 2 980; two
//  sub-threads of 14 chains 4 120 -- fewer LDS reads, but 18 dependent-free instructions per slot and thread on two SIMDs instead of
//  10 on four, and two or three slots a turn is all the registers allow there)
template <int SUB>
__device__ __forceinline__ void wide_sums(const double* s_rows, int vt, int ns, double* v) {
    constexpr int T = 5;
    for (int j0 = 0; j0 < ns; j0 += T) {
        double J0[T], J1[T], J2[T], J3[T], J4[T], J5[T], rr[T], rho0[T], w[T];
#pragma unroll
        for (int u = 0; u < T; ++u) {
            const double* p = s_rows + (size_t)min(j0 + u, ns - 1) * ROW_DOUBLES * SOLVE_THREADS + vt;
            if constexpr (SUB == 0) J0[u] = p[0];
            if constexpr (SUB <= 1) J1[u] = p[SOLVE_THREADS];
            if constexpr (SUB <= 2) J2[u] = p[2 * SOLVE_THREADS];
            J3[u] = p[3 * SOLVE_THREADS];
            J4[u] = p[4 * SOLVE_THREADS];
            J5[u] = p[5 * SOLVE_THREADS];
            rr[u] = p[6 * SOLVE_THREADS];
            if constexpr (SUB == 0) rho0[u] = p[7 * SOLVE_THREADS];
            w[u] = p[8 * SOLVE_THREADS];
        }
#pragma unroll
        for (int u = 0; u < T; ++u) {
            const bool ok = j0 + u < ns && w[u] > 0.0;  // no row: the factor is skipped (rho' of a row is > 0)
            double n[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) n[k] = v[k];
            if constexpr (SUB == 0) {
                n[7] = v[7] + 0.5 * rho0[u];
                const double wj = w[u] * J0[u];
                n[0] = __builtin_fma(wj, J0[u], v[0]);
                n[1] = __builtin_fma(wj, J1[u], v[1]);
                n[2] = __builtin_fma(wj, J2[u], v[2]);
                n[3] = __builtin_fma(wj, J3[u], v[3]);
                n[4] = __builtin_fma(wj, J4[u], v[4]);
                n[5] = __builtin_fma(wj, J5[u], v[5]);
                n[6] = __builtin_fma(wj, rr[u], v[6]);
            } else if constexpr (SUB == 1) {
                const double wj = w[u] * J1[u];
                n[0] = __builtin_fma(wj, J1[u], v[0]);
                n[1] = __builtin_fma(wj, J2[u], v[1]);
                n[2] = __builtin_fma(wj, J3[u], v[2]);
                n[3] = __builtin_fma(wj, J4[u], v[3]);
                n[4] = __builtin_fma(wj, J5[u], v[4]);
                n[5] = __builtin_fma(wj, rr[u], v[5]);
                const double wj5 = w[u] * J5[u];
                n[6] = __builtin_fma(wj5, J5[u], v[6]);
                n[7] = __builtin_fma(wj5, rr[u], v[7]);
            } else if constexpr (SUB == 2) {
                const double wj = w[u] * J2[u];
                n[0] = __builtin_fma(wj, J2[u], v[0]);
                n[1] = __builtin_fma(wj, J3[u], v[1]);
                n[2] = __builtin_fma(wj, J4[u], v[2]);
                n[3] = __builtin_fma(wj, J5[u], v[3]);
                n[4] = __builtin_fma(wj, rr[u], v[4]);
                const double wj4 = w[u] * J4[u];
                n[5] = __builtin_fma(wj4, J4[u], v[5]);
                n[6] = __builtin_fma(wj4, J5[u], v[6]);
                n[7] = __builtin_fma(wj4, rr[u], v[7]);
            } else {
                const double wj = w[u] * J3[u];
                n[0] = __builtin_fma(wj, J3[u], v[0]);
                n[1] = __builtin_fma(wj, J4[u], v[1]);
                n[2] = __builtin_fma(wj, J5[u], v[2]);
                n[3] = __builtin_fma(wj, rr[u], v[3]);
            }
            // (a slot whose 64 factors of this wavefront all have a row -- nearly every one -- takes the sums as they come: the
            //  selects of the general case were half of this phase's instructions)
            if (__all(ok)) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = n[k];
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = ok ? n[k] : v[k];
            }
        }
    }
}

#ifdef MML_SV_TIMING
// phase clocks of eval_frame_wide, thread 0 of problem MML_SV_TIMING: rows, wait at the barrier, sums, tree; [4] passes
__device__ unsigned long long g_svw_dbg[8];
#define SVW_MARK(id)                                   \
    do {                                               \
        if (svw_dbg) {                                 \
            const unsigned long long now_ = clock64(); \
            g_svw_dbg[id] += now_ - svw_prev;          \
            svw_prev = now_;                           \
        }                                              \
    } while (0)
#else
#define SVW_MARK(id)
#endif
// workgroup of WIDE_THREADS; s_rows: WIDE_ROW_LDS doubles, s_vsave: WIDE_SAVE_LDS, s_part: SOLVE_WAVES * 28; out[28] is valid for the FIRST WAVEFRONT on return
__device__ __forceinline__ void eval_frame_wide(const MmlLineFactor* lf, int nlf, const MmlPlaneFactor* pf, int npf, const Pose& P,
                                                double huber_delta, double* s_rows, double* s_vsave, double* s_part, double* out) {
    // (the thread number is made opaque to the compiler at every call: as a known function of threadIdx.x every address below is formed
    //  once in front of the solve's loop and held across the trust-region code and the rows phase -- the registers for that are not
    //  there, and a spilled address is a memory round trip when it is needed; see k_solve_wide)
    int tx = threadIdx.x;
    asm volatile("" : "+v"(tx));
    const int vt = tx & (SOLVE_THREADS - 1);
    const int sub = __builtin_amdgcn_readfirstlane(tx / SOLVE_THREADS);  // (uniform in a wavefront)
    const int lane = tx & 63, wv = vt >> 6;
    const int SL = (nlf + SOLVE_THREADS - 1) / SOLVE_THREADS, SP = (npf + SOLVE_THREADS - 1) / SOLVE_THREADS, NS = SL + SP;
    const double ka = 1.0 / kLidarM;
#ifdef MML_SV_TIMING
    const bool svw_dbg = threadIdx.x == 0 && blockIdx.x == MML_SV_TIMING;
    unsigned long long svw_prev = clock64();
    if (svw_dbg) g_svw_dbg[4] += 1;
#endif
    for (int s0 = 0; s0 < NS; s0 += WIDE_ROUND) {
        // ---- rows: slots s0 + sub, + 4, + 8 of this round ----
        const int sA = s0 + sub, sB = sA + WIDE_SUBS, sC = sB + WIDE_SUBS;
        const bool hA = sA < NS, hB = sB < NS, hC = sC < NS;            // the slot exists
        const bool lA = sA < SL, lB = sB < SL, lC = sC < SL;            // ... and is a line slot
        const int iA = vt + SOLVE_THREADS * (lA ? sA : sA - SL), iB = vt + SOLVE_THREADS * (lB ? sB : sB - SL),
                  iC = vt + SOLVE_THREADS * (lC ? sC : sC - SL);
        // (all records of the round requested before the first row is formed: one exposed round trip per pass; a slot's record,
        //  48 bytes of a line factor or 64 of a plane factor, lands in the same registers)
        WideRec qa, qb, qc;
        if (hA) wide_load(qa, lA, lf, min(iA, nlf - 1), pf, min(iA, npf - 1));
        if (hB) wide_load(qb, lB, lf, min(iB, nlf - 1), pf, min(iB, npf - 1));
        if (hC) wide_load(qc, lC, lf, min(iC, nlf - 1), pf, min(iC, npf - 1));
        const bool inA = iA < (lA ? nlf : npf), inB = iB < (lB ? nlf : npf), inC = iC < (lC ? nlf : npf);
        PlaneRow ra, rb, rc;
        if (hA && hB && !lA && !lB) {  // two plane rows side by side, then the third slot
            plane_row(wide_plane(qa), P, ka, huber_delta, ra);
            plane_row(wide_plane(qb), P, ka, huber_delta, rb);
            row_store(s_rows, sub, vt, ra, inA);
            row_store(s_rows, sub + WIDE_SUBS, vt, rb, inB);
            if (hC) {
                plane_row(wide_plane(qc), P, ka, huber_delta, rc);  // (slots ascend: behind a plane slot there are only plane slots)
                row_store(s_rows, sub + 2 * WIDE_SUBS, vt, rc, inC);
            }
        } else {
            if (hA) {
                if (lA)
                    line_row(wide_line(qa), P, ka, huber_delta, ra);
                else
                    plane_row(wide_plane(qa), P, ka, huber_delta, ra);
                row_store(s_rows, sub, vt, ra, inA);
            }
            if (hB && hC && !lB && !lC) {
                plane_row(wide_plane(qb), P, ka, huber_delta, rb);
                plane_row(wide_plane(qc), P, ka, huber_delta, rc);
                row_store(s_rows, sub + WIDE_SUBS, vt, rb, inB);
                row_store(s_rows, sub + 2 * WIDE_SUBS, vt, rc, inC);
            } else {
                if (hB) {
                    if (lB)
                        line_row(wide_line(qb), P, ka, huber_delta, rb);
                    else
                        plane_row(wide_plane(qb), P, ka, huber_delta, rb);
                    row_store(s_rows, sub + WIDE_SUBS, vt, rb, inB);
                }
                if (hC) {
                    if (lC)
                        line_row(wide_line(qc), P, ka, huber_delta, rc);
                    else
                        plane_row(wide_plane(qc), P, ka, huber_delta, rc);
                    row_store(s_rows, sub + 2 * WIDE_SUBS, vt, rc, inC);
                }
            }
        }
        SVW_MARK(0);
        __syncthreads();
        SVW_MARK(1);
        // ---- sums: the summing sub-threads' chains over the round's rows, in slot order; behind the last round the tree ----
        // (the sums live in registers only from here to the end of the round: a scan with more than WIDE_ROUND slots per thread parks
        //  them in LDS between its rounds -- held across the rows phase they cost it the registers it needs)
        const int ns = min(WIDE_ROUND, NS - s0);
        const bool last = s0 + WIDE_ROUND >= NS;
        {
            double v[8];
            double* sv = s_vsave + (size_t)sub * SOLVE_THREADS + vt;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0;
            if (s0 > 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = sv[(size_t)k * WIDE_THREADS];
            }
            if (sub == 0)
                wide_sums<0>(s_rows, vt, ns, v);
            else if (sub == 1)
                wide_sums<1>(s_rows, vt, ns, v);
            else if (sub == 2)
                wide_sums<2>(s_rows, vt, ns, v);
            else
                wide_sums<3>(s_rows, vt, ns, v);
            SVW_MARK(2);
            if (!last) {
#pragma unroll
                for (int k = 0; k < 8; ++k) sv[(size_t)k * WIDE_THREADS] = v[k];
            } else {
                // ---- tree: block_reduce28's, eight values a lane (the halving butterfly for xor 32, 16, 8; then the one value
                //      left); the partner's value by lane_xor_f64, not through the LDS crossbar ----
                const bool u32 = (lane & 32) != 0, u16 = (lane & 16) != 0, u8 = (lane & 8) != 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double keep = u32 ? v[j + 4] : v[j], send = u32 ? v[j] : v[j + 4];
                    v[j] = keep + lane_xor_f64<32>(send, lane);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const double keep = u16 ? v[j + 2] : v[j], send = u16 ? v[j] : v[j + 2];
                    v[j] = keep + lane_xor_f64<16>(send, lane);
                }
                {
                    const double keep = u8 ? v[1] : v[0], send = u8 ? v[0] : v[1];
                    v[0] = keep + lane_xor_f64<8>(send, lane);
                }
                const int idx = (u32 ? 4 : 0) + (u16 ? 2 : 0) + (u8 ? 1 : 0);
                double tot = v[0] + lane_xor_f64<4>(v[0], lane);
                tot = tot + lane_xor_f64<2>(tot, lane);
                tot = tot + lane_xor_f64<1>(tot, lane);
                // which of the 28 sums value idx of this sub-thread is (one byte each, wide_sums' order; 0xff: none)
                const unsigned long long km = sub == 0 ? 0x1B15050403020100ull
                                                       : (sub == 1 ? 0x1A14160A09080706ull : (sub == 2 ? 0x191312170E0D0C0Bull : 0xFFFFFFFF1811100Full));
                const int k = (int)((km >> (8 * idx)) & 0xffull);
                if (!(lane & 7) && k < 28) s_part[wv * 28 + k] = tot;
            }
        }
        if (!last) __syncthreads();  // (the next round overwrites the rows)
    }
    if (NS == 0 && tx < SOLVE_WAVES * 28) s_part[tx] = 0.0;  // (no factor at all: the sums of nothing)
    __syncthreads();
    if (tx < 28) {
        double r = s_part[tx];
        r += s_part[28 + tx];
        out[tx] = r;
    }
    // (no barrier behind this: out[] is read by the first wavefront, which has just written it -- k_solve_wide's trust-region code --
    //  and everybody else meets that wavefront at the caller's next barrier before s_rows / s_part are written again)
    SVW_MARK(3);
}

__host__ __device__ __forceinline__ int tri(int a, int b) {  // index into 21-entry upper triangle, a <= b
    return a * 6 - (a * (a - 1)) / 2 + (b - a);
}
__host__ __device__ __forceinline__ double Hget(const double* rec, int a, int b) { return a <= b ? rec[tri(a, b)] : rec[tri(b, a)]; }

}  // namespace
