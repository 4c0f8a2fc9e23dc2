// fullwindow_dev.hip -- SURVEY.md section 8(f) rank 1 with the trust-region loop resident on the device.
// Estimator::Estimate in full-window mode (mm-loam/src/lio/Estimator.cpp:1226-1254 problem set-up, :1425-1432
// ceres::Solve): W <= 8 frames x [PR 6 | VBias 9], lidar factors of every frame, IMU factors between consecutive frames
// (Cost_NavState_PRV_Bias, ceresfunc.h:321-393), the marginalization prior on frame 0 (ceresfunc.h:244-303).
// window_imu.hip runs this minimisation on the host and comes back to the device for every evaluation of the lidar
// factors; here the Ceres 2.1 TRADITIONAL_DOGLEG iteration is a device-resident state machine advanced by two kernels
// per evaluation, enqueued max_iterations + 1 times without reading anything back (a finished state machine turns the
// remaining launches into no-ops):
//   k_fw_eval   2 W workgroups.  Workgroup f < W: the lidar factors of frame f at the candidate (lidar_eval.h) -> its
//               28-value record.  Workgroup W + f: IMU factor f -- residual and Jacobian on one lane (imu_math.h, the
//               code the host solver runs), then the sqrt-information products and the 30 x 30 J^T J / J^T r of the
//               factor element-parallel; workgroup W: the prior residual and its J^T r.
//   k_fw_step   1 workgroup.  Adds the pieces into the normal equations in the order the host assembly adds them
//               (lidar, IMU f, IMU f + 1, prior).  The 15 W system is block tridiagonal (a factor couples consecutive
//               frames only), so H lives as a band of three 15-column blocks per row, in LDS.  Accept / reject the
//               previous candidate, then Jacobi scaling, the regularised Gauss-Newton step by a right-looking band
//               Cholesky run barrier-free by ONE wavefront (every element receives its subtractions in the order of
//               the dense left-looking host routine; the skipped out-of-band terms are exact zeros there),
//               substitutions with the right-hand side in registers (one lane per row, the pivot handed over by a
//               lane read), dogleg interpolation, the next candidate.  Element-wise work runs on all lanes; sums that
//               Ceres takes sequentially are taken sequentially by one lane each, on different wavefronts at once.
// Differences to the host loop are limited to libm vs device sin / cos / atan / sqrt and the summation order of the
// back substitution.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "fullwindow_internal.h"
#include "imu_math.h"
#include "lidar_eval.h"
#include "mml_internal.h"

// k_fw_step -- the dense 15 W-dimensional trust-region iteration -- keeps 256 threads; the lidar evaluation (k_fw_eval: eval_frame +
// block_reduce28, shared with k_solve / k_linearize / k_window_round so that all of them sum in the same order) runs SOLVE_THREADS.
constexpr int FW_THREADS = 256;
namespace {

constexpr int FW_N = 15 * MAXW;  // 120 parameters at most
constexpr int FW_BW = 45;        // band of H: the blocks blk - 1, blk, blk + 1 of a row
constexpr int FW_LW = 31;        // LDS doubles per row reserved for the Cholesky band (30 used, see lbi())

struct FwDevParams {  // uploaded once per solve
    int W, max_iters, fixed, first, has_prior, pad_[3];
    int have_imu[MAXW];
    double huber, w_tan;
    double gravity[3];
    double pad2_;
    double Tbl[16];
    double x[FW_N];
    mml_prior prior;
    double PP[225];            // prior.J^T prior.J (constant over the solve)
    mml_imu_preint imu[MAXW];  // imu[f]: between frames f - 1 and f
    double U[MAXW][225];       // sqrt information of imu[f], upper triangular
};

#ifdef MML_FW_TIMING
#define FW_T(k) do { __syncthreads(); if (threadIdx.x == 0) { const long long t_ = wall_clock64(); sh.s.ticks[k] += (double)(t_ - sh.s.t_last); sh.s.t_last = t_; } } while (0)
#define FW_TW(k) do { if (threadIdx.x == 0) { const long long t_ = wall_clock64(); sh.s.ticks[k] += (double)(t_ - sh.s.t_last); sh.s.t_last = t_; } } while (0)
#else
#define FW_T(k) do { } while (0)
#define FW_TW(k) do { } while (0)
#endif

struct FwScalars {
    double radius, mu, alpha, dogleg_norm, x_norm, model_change, step_norm, sgd, q, gg, gradient_norm, gn_norm, gdot, sn;
    double cost[2];
    double initial_cost;
    int reuse, num_invalid, iter, successful, termination, cur, flag, evals, go, steps;
#ifdef MML_FW_TIMING
    long long t_last;
    double ticks[16];
#endif
};

struct FwVectors {
    double x[FW_N], xc[FW_N], x_init[FW_N], scale[FW_N], diag[FW_N], grad[FW_N], gn[FW_N], step[FW_N];
    double g[2][FW_N];
};

struct FwGlobal {  // the state machine between the launches
    FwScalars s;
    FwVectors v;
    double Hb[2][FW_N * FW_BW];  // band of the normal equations at x and at the candidate
    // what k_fw_eval leaves for k_fw_step
    double rec[MAXW][28];
    double JJ[MAXW][900], Jr[MAXW][30], rr[MAXW];  // per IMU factor: J^T J, J^T r, r^T r (after the sqrt information)
    double gp[15], rp2;                             // prior: J^T r, r^T r
};

struct FwDevOut {
    double x[FW_N];
    double initial_cost, final_cost;
    int iterations, successful, termination, evaluations, steps, pad_;
#ifdef MML_FW_TIMING
    double ticks[16];
#endif
};

struct FwKernelArgs {
    const FwDevParams* P;
    FwGlobal* G;
    FwDevOut* out;
    const int* ft_n;
    const MmlLineFactor* lf;
    const MmlPlaneFactor* pf;
    int B, MF, round, last;
};

__device__ __forceinline__ int band0(int a) { return 15 * (a / 15 - 1); }  // first column of row a's band (may be -15)

// sum_i a[i] * b[i] for i = 0 .. n - 1 in that order, n a multiple of 15; one lane, loads issued 15 at a time
__device__ double ordered_dot(const double* a, const double* b, int n) {
    double s = 0;
    for (int f = 0; f < n; f += 15) {
        double u[15], w[15];
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            u[i] = a[f + i];
            w[i] = b[f + i];
        }
#pragma unroll
        for (int i = 0; i < 15; ++i) s += u[i] * w[i];
    }
    return s;
}

// ---- evaluation ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SOLVE_THREADS) void k_fw_eval(FwKernelArgs A) {
    __shared__ double s_part[SOLVE_WAVES * 28];
    __shared__ double s_x[30];
    __shared__ double s_r[15], s_rs[15];
    __shared__ double s_J[450], s_Js[450];
    const FwDevParams* P = A.P;
    FwGlobal* G = A.G;
    if (!G->s.go) return;
    const int tid = threadIdx.x, W = P->W, blk = blockIdx.x;
    if (blk < W) {  // lidar factors of frame blk
        const int b = P->first + blk;
        if (tid < 6) s_x[tid] = G->v.xc[15 * blk + tid];
        __syncthreads();
        Pose pose;
        make_pose(s_x, P->Tbl, pose);
        double acc[28];
        eval_frame(A.lf + (size_t)b * A.MF, A.ft_n[b], A.pf + (size_t)b * A.MF, A.ft_n[A.B + b], pose, P->w_tan, P->huber, acc);
        block_reduce28(acc, s_part, G->rec[blk]);
        return;
    }
    const int f = blk - W;
    if (f == 0) {  // MarginalizationFactor on frame 0
        if (!P->has_prior) return;
        if (tid < 15) s_x[tid] = G->v.xc[tid];
        __syncthreads();
        if (tid == 0) prior_residual(P->prior, s_x, s_r);
        __syncthreads();
        if (tid < 15) {
            double p = 0;
            for (int i = 0; i < 15; ++i) p += P->prior.J[i * 15 + tid] * s_r[i];
            G->gp[tid] = p;
        } else if (tid == 64) {
            double c = 0;
            for (int i = 0; i < 15; ++i) c += s_r[i] * s_r[i];
            G->rp2 = c;
        }
        return;
    }
    if (!P->have_imu[f]) return;
    if (tid < 30) s_x[tid] = G->v.xc[15 * (f - 1) + tid];
    __syncthreads();
    if (tid == 0) imu_raw(&P->imu[f], P->gravity, s_x, s_x + 6, s_x + 15, s_x + 21, s_r, s_J);
    __syncthreads();
    // eResiduals.applyOnTheLeft(sqrt_information), the same for the Jacobian (ceresfunc.h:352,388-391)
    const double* U = P->U[f];
    for (int o = tid; o < 465; o += SOLVE_THREADS) {
        double s = 0;
        if (o < 450) {
            const int i = o / 30, c = o - 30 * i;
            for (int k = i; k < 15; ++k) s += U[i * 15 + k] * s_J[k * 30 + c];
            s_Js[o] = s;
        } else {
            const int i = o - 450;
            for (int k = i; k < 15; ++k) s += U[i * 15 + k] * s_r[k];
            s_rs[i] = s;
        }
    }
    __syncthreads();
    for (int o = tid; o < 931; o += SOLVE_THREADS) {
        if (o < 900) {
            const int a = o / 30, b = o - 30 * a;
            double h = 0;
            for (int i = 0; i < 15; ++i) h += s_Js[i * 30 + a] * s_Js[i * 30 + b];
            G->JJ[f][o] = h;
        } else if (o < 930) {
            const int a = o - 900;
            double h = 0;
            for (int i = 0; i < 15; ++i) h += s_Js[i * 30 + a] * s_rs[i];
            G->Jr[f][a] = h;
        } else {
            double c = 0;
            for (int i = 0; i < 15; ++i) c += s_rs[i] * s_rs[i];
            G->rr[f] = c;
        }
    }
}

// ---- trust-region step -----------------------------------------------------------------------------------------------
struct FwShared {
    double H[FW_N * FW_BW];    // band of the normal equations at the current point
    double big[FW_N * FW_LW];  // the Cholesky band
    FwVectors v;
    double tmp[FW_N], rows[FW_N], bvec[FW_N];
    FwScalars s;
};

// v^T (S H S) v over the band (the out-of-band terms of the dense host loop are exact zeros)
__device__ double fw_quad(FwShared& sh, int n, const double* v) {
    const int tid = threadIdx.x;
    if (tid < n) {
        const int c0 = band0(tid);
        const int lo = max(0, -c0), hi = min(FW_BW, n - c0);
        double row = 0;
        for (int cb = lo; cb < hi; ++cb) row += sh.H[tid * FW_BW + cb] * sh.v.scale[c0 + cb] * v[c0 + cb];
        sh.rows[tid] = v[tid] * sh.v.scale[tid];
        sh.tmp[tid] = row;
    }
    __syncthreads();
    if (tid == 0) sh.s.q = ordered_dot(sh.rows, sh.tmp, n);
    __syncthreads();
    return sh.s.q;
}

// The Cholesky band in LDS: element (i, k), i - 29 <= k <= i, at 29 i + k + 29 -- linear in both indices, so a column
// step addresses everything as (a base that advances by 30 per column) + (a per-lane constant).
__device__ __forceinline__ int lbi(int i, int k) { return 29 * i + k + 29; }

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// the triangle behind a column, enumerated (0,0), (1,0), (1,1), (2,0), ...: entry t is (row, column) relative to j + 1
struct TriTab {
    unsigned char i[448], c[448];
};
constexpr TriTab make_tri_tab() {
    TriTab t{};
    int k = 0;
    for (int i = 0; i < 30; ++i)
        for (int c = 0; c <= i; ++c)
            if (k < 448) {
                t.i[k] = (unsigned char)i;
                t.c[k] = (unsigned char)c;
                ++k;
            }
    return t;
}
__device__ const TriTab kTriTab = make_tri_tab();

// In-place band Cholesky of sh.big (lower) by wavefront 0, then L L^T z = bvec, z -> bvec.  Returns 0 when the matrix is
// not positive definite or the solution is not finite.  Right-looking: column j is scaled, then the (at most 29 x 29)
// triangle behind it is updated, 64 elements at a time; no workgroup barrier inside.
__device__ int fw_factor_solve(FwShared& sh, int n) {
    const int lane = threadIdx.x;  // called by threads 0 .. 63
    double* L = sh.big;
    int pi[7], oa[7], ob[7], oe[7];  // the triangle entries this lane owns and their offsets from the column base
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int t = lane + 64 * k;
        const int ii = kTriTab.i[t], cc = kTriTab.c[t];
        pi[k] = ii;
        oa[k] = 58 + 29 * ii;        // L[j + 1 + ii][j]
        ob[k] = 58 + 29 * cc;        // L[j + 1 + cc][j]
        oe[k] = 59 + 29 * ii + cc;   // A[j + 1 + ii][j + 1 + cc]
    }
    int ok = 1;
    FW_TW(7);
#ifdef MML_FW_TIMING
    const long long c0_ = clock64();
#endif
    double d0 = L[29];
    for (int j = 0; j < n; ++j) {
        double* Lj = L + 30 * j;  // Lj[29] = (j, j), Lj[58 + 29 l] = (j + 1 + l, j)
        const int jend = min(n, 15 * (j / 15 + 2)), m = jend - j - 1;
        const double col = Lj[lane < m ? 58 + 29 * lane : 29];  // requested before the square root is taken
        if (!(d0 > 0.0) || !isfinite(d0)) {  // the same value in every lane
            ok = 0;
            break;
        }
        const double d = sqrt(d0);
        if (lane < m) Lj[58 + 29 * lane] = col / d;
        if (lane == 0) Lj[29] = d;
        wave_sync();
        // all loads of the triangle first, then the arithmetic, then the stores: one LDS round trip per column
        double ua[7], ub[7], ue[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const bool act = pi[k] < m;
            ua[k] = Lj[act ? oa[k] : 29];
            ub[k] = Lj[act ? ob[k] : 29];
            ue[k] = Lj[act ? oe[k] : 29];
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) ue[k] -= ua[k] * ub[k];
        d0 = readlane_f64(ue[0], 0);  // element (j + 1, j + 1): the next pivot, without waiting for the store below
#pragma unroll
        for (int k = 0; k < 7; ++k)
            if (pi[k] < m) Lj[oe[k]] = ue[k];
        wave_sync();
    }
    FW_TW(8);
#ifdef MML_FW_TIMING
    if (threadIdx.x == 0) sh.s.ticks[11] += (double)(clock64() - c0_);
#endif
    if (!ok) return 0;
    // forward substitution: lane l carries the right-hand sides of rows l and l + 64 (never inside one band at once) and
    // the reciprocals of their pivots; the element of L a step needs is requested one step ahead
    double s0 = lane < n ? sh.bvec[lane] : 0.0, s1 = lane + 64 < n ? sh.bvec[lane + 64] : 0.0;
    const double r0 = lane < n ? 1.0 / L[lbi(lane, lane)] : 0.0, r1 = lane + 64 < n ? 1.0 / L[lbi(lane + 64, lane + 64)] : 0.0;
    auto fwd_row = [&](int k) { return (k + 1) + ((lane - (k + 1)) & 63); };  // the row >= k + 1 this lane carries
    double lnext = 0.0;
    {
        const int i = fwd_row(0);
        lnext = L[i < min(n, 30) ? lbi(i, 0) : 29];
    }
    for (int k = 0; k < n; ++k) {
        const double lcur = lnext;
        const int i = fwd_row(k), kend = min(n, 15 * (k / 15 + 2));
        if (k + 1 < n) {
            const int i2 = fwd_row(k + 1), kend2 = min(n, 15 * ((k + 1) / 15 + 2));
            lnext = L[i2 < kend2 ? lbi(i2, k + 1) : 29];
        }
        const double sk = readlane_f64((k & 64) ? s1 : s0, k & 63);
        const double yk = sk * readlane_f64((k & 64) ? r1 : r0, k & 63);
        const double u = lcur * yk;
        const bool act = i < kend, hi = (i & 64) != 0, own = lane == (k & 63), khi = (k & 64) != 0;
        s0 = (own && !khi) ? yk : ((act && !hi) ? s0 - u : s0);
        s1 = (own && khi) ? yk : ((act && hi) ? s1 - u : s1);
    }
    FW_TW(9);
    // backward substitution (the columns become known from the last one down)
    auto bwd_row = [&](int k) { return (k - 1) - (((k - 1) - lane) & 63); };  // the row <= k - 1 this lane carries
    {
        const int k = n - 1, i = bwd_row(k);
        lnext = L[(i >= max(0, band0(k)) && k >= 1) ? lbi(k, i) : 29];
    }
    for (int k = n - 1; k >= 0; --k) {
        const double lcur = lnext;
        const int i = bwd_row(k), k0 = max(0, band0(k));
        if (k >= 1) {
            const int k2 = k - 1, i2 = bwd_row(k2);
            lnext = L[(i2 >= max(0, band0(k2)) && k2 >= 1) ? lbi(k2, i2) : 29];
        }
        const double sk = readlane_f64((k & 64) ? s1 : s0, k & 63);
        const double zk = sk * readlane_f64((k & 64) ? r1 : r0, k & 63);
        const double u = lcur * zk;
        const bool act = i >= k0 && k >= 1, hi = (i & 64) != 0, own = lane == (k & 63), khi = (k & 64) != 0;
        s0 = (own && !khi) ? zk : ((act && !hi) ? s0 - u : s0);
        s1 = (own && khi) ? zk : ((act && hi) ? s1 - u : s1);
    }
    FW_TW(10);
    int bad = 0;
    if (lane < n) {
        sh.bvec[lane] = s0;
        bad |= !isfinite(s0);
    }
    if (lane + 64 < n) {
        sh.bvec[lane + 64] = s1;
        bad |= !isfinite(s1);
    }
    return __any(bad) ? 0 : 1;
}

// One proposal (mml_fullwindow_step's propose): 1 = sh.v.xc holds a candidate, 0 = invalid step (propose again),
// -1 = the minimiser has stopped.  The return value is the same in every thread.
__device__ int fw_propose(const FwDevParams* P, FwShared& sh, int n) {
    const int tid = threadIdx.x;
    FwScalars& S = sh.s;
    FwVectors& V = sh.v;
    __syncthreads();
    if (S.iter >= P->max_iters || S.radius < 1e-32) return -1;
    const int reuse = S.reuse;
    const double* g = V.g[S.cur];
    __syncthreads();
    if (tid == 0) S.iter++;
    bool solve_ok = true;
    if (!reuse) {
        if (tid == 0) S.reuse = 1;
        if (tid < n) {
            double d = sh.H[tid * FW_BW + (tid - band0(tid))] * V.scale[tid] * V.scale[tid];
            d = fmin(fmax(d, 1e-6), 1e32);
            const double dg = sqrt(d);
            V.diag[tid] = dg;
            const double gr = g[tid] * V.scale[tid] / dg;
            V.grad[tid] = gr;
            sh.bvec[tid] = gr / dg;  // the Cauchy direction in the scaled space
        }
        __syncthreads();
        if (tid == 64) S.gg = ordered_dot(V.grad, V.grad, n);
        const double q = fw_quad(sh, n, sh.bvec);
        if (tid == 0) {
            S.alpha = S.gg / q;
            S.gradient_norm = sqrt(S.gg);
        }
        FW_T(2);
        solve_ok = false;
        for (;;) {
            __syncthreads();
            const double mu = S.mu;
            if (!(mu < 1.0)) break;
            for (int o = tid; o < n * 30; o += FW_THREADS) {
                const int i = o / 30, k = band0(i) + (o - 30 * i);
                if (k < 0 || k > i) continue;
                double a = sh.H[i * FW_BW + (k - band0(i))] * V.scale[i] * V.scale[k];
                if (k == i) a += mu * V.diag[i] * V.diag[i];
                sh.big[lbi(i, k)] = a;
            }
            if (tid < n) sh.bvec[tid] = g[tid] * V.scale[tid];
            __syncthreads();
            FW_T(3);
            if (tid < 64) {
                const int ok = fw_factor_solve(sh, n);
                if (tid == 0) S.flag = ok;
            }
            __syncthreads();
            FW_T(4);
            if (!S.flag) {
                __syncthreads();
                if (tid == 0) S.mu = mu * 10.0;
                continue;
            }
            if (tid < n) V.gn[tid] = -V.diag[tid] * sh.bvec[tid];
            solve_ok = true;
            break;
        }
        __syncthreads();
    }
    bool step_valid = solve_ok;
    if (solve_ok) {
        // the three sums of the dogleg on three wavefronts
        if (tid == 0) S.gn_norm = sqrt(ordered_dot(V.gn, V.gn, n));
        if (tid == 64) S.gdot = ordered_dot(V.grad, V.gn, n);
        __syncthreads();
        const double gradient_norm = S.gradient_norm, gn_norm = S.gn_norm, radius = S.radius, alpha = S.alpha;
        int branch;
        double beta = 0;
        if (gn_norm <= radius) {
            branch = 0;
            if (tid < n) V.step[tid] = V.gn[tid];
        } else if (gradient_norm * alpha >= radius) {
            branch = 1;
            if (tid < n) V.step[tid] = -(radius / gradient_norm) * V.grad[tid];
        } else {
            branch = 2;
            const double b_dot_a = -alpha * S.gdot;
            const double a_sq = (alpha * gradient_norm) * (alpha * gradient_norm);
            const double bma_sq = a_sq - 2 * b_dot_a + gn_norm * gn_norm;
            const double c = b_dot_a - a_sq;
            const double d = sqrt(c * c + bma_sq * (radius * radius - a_sq));
            beta = (c <= 0) ? (d - c) / bma_sq : (radius * radius - a_sq) / (d + c);
            if (tid < n) V.step[tid] = (-alpha * (1.0 - beta)) * V.grad[tid] + beta * V.gn[tid];
        }
        __syncthreads();
        if (tid == 128 && branch == 2) S.sn = ordered_dot(V.step, V.step, n);
        __syncthreads();
        if (tid < n) {
            const double st = V.step[tid] / V.diag[tid];
            V.step[tid] = st;
            sh.bvec[tid] = st * g[tid];
        }
        __syncthreads();
        if (tid == 64) S.sgd = ordered_dot(sh.bvec, V.scale, n);
        if (tid == 0) S.dogleg_norm = branch == 0 ? gn_norm : (branch == 1 ? radius : sqrt(S.sn));
        const double q = fw_quad(sh, n, V.step);
        const double model_change = -(S.sgd + 0.5 * q);
        if (!(model_change > 0.0)) step_valid = false;
        __syncthreads();
        if (tid == 0) S.model_change = model_change;
        FW_T(5);
    }
    if (!step_valid) {
        __syncthreads();
        if (tid == 0) {
            if (++S.num_invalid >= 5) {  // HandleInvalidStep: FAILURE, the parameters go back as they came in
                S.termination = 4;
                S.flag = -1;
            } else {
                S.mu *= 10.0;
                S.reuse = 0;
                S.flag = 0;
            }
        }
        __syncthreads();
        const int flag = S.flag;
        if (flag < 0 && tid < n) V.x[tid] = V.x_init[tid];
        return flag;
    }
    if (tid < n) {
        const double delta = V.step[tid] * V.scale[tid];
        V.xc[tid] = V.x[tid] + delta;
        sh.bvec[tid] = delta;
    }
    __syncthreads();
    if (tid == 0) {
        S.num_invalid = 0;
        S.step_norm = sqrt(ordered_dot(sh.bvec, sh.bvec, n));
    }
    __syncthreads();
    FW_T(6);
    return 1;
}

// accept / reject after the candidate was evaluated into slot 1 - cur; true when the minimiser stops.  On acceptance
// sh.H (holding the candidate's band) becomes the current band; on rejection the current band is read back.
__device__ bool fw_decide(const FwKernelArgs& A, FwShared& sh, int n) {
    FwScalars& S = sh.s;
    FwVectors& V = sh.v;
    const int tid = threadIdx.x;
    const bool fixed = A.P->fixed != 0;
    __syncthreads();
    const double cur = S.cost[S.cur], cand = S.cost[1 - S.cur];
    if (!fixed) {
        int stop = 0, term = 0;
        if (S.step_norm <= 1e-8 * (S.x_norm + 1e-8)) {
            term = 2;
            stop = 1;
        } else if (fabs(cur - cand) <= 1e-6 * cur) {
            term = 3;
            stop = 1;
        }
        if (stop) {
            __syncthreads();
            if (tid == 0) S.termination = term;
            return true;
        }
    }
    const double rel = (cur - cand) / S.model_change;
    const int cur_slot = S.cur;
    __syncthreads();
    if (rel > 1e-3) {
        if (tid < n) V.x[tid] = V.xc[tid];
        __syncthreads();
        if (tid == 0) {
            S.x_norm = sqrt(ordered_dot(V.x, V.x, n));
            S.cur = 1 - cur_slot;
            S.successful++;
            int stop = 0;
            if (!fixed) {
                double gm = 0;
                for (int i = 0; i < n; ++i) gm = fmax(gm, fabs(V.g[1 - cur_slot][i]));
                if (gm <= 1e-10) {
                    S.termination = 1;
                    stop = 1;
                }
            }
            if (!stop) {
                if (rel < 0.25) S.radius *= 0.5;
                if (rel > 0.75) S.radius = fmax(S.radius, 3.0 * S.dogleg_norm);
                S.mu = fmax(1e-8, 2.0 * S.mu / 10.0);
                S.reuse = 0;
            }
            S.flag = stop;
        }
    } else {
        for (int o = tid; o < n * FW_BW; o += FW_THREADS) sh.H[o] = A.G->Hb[cur_slot][o];
        if (tid == 0) {
            S.radius *= 0.5;
            S.reuse = 1;
            S.flag = 0;
        }
    }
    __syncthreads();
    return S.flag != 0;
}

__global__ __launch_bounds__(FW_THREADS) void k_fw_step(FwKernelArgs A) {
    __shared__ FwShared sh;
    const FwDevParams* P = A.P;
    FwGlobal* G = A.G;
    const int tid = threadIdx.x, W = P->W, n = 15 * W;
    if (!G->s.go) {
        if (A.last && tid == 0) A.out->steps = G->s.steps;
        return;
    }
    FwScalars& S = sh.s;
    FwVectors& V = sh.v;
    {
        const double* src = reinterpret_cast<const double*>(&G->v);
        double* dst = reinterpret_cast<double*>(&V);
        for (int i = tid; i < (int)(sizeof(FwVectors) / sizeof(double)); i += FW_THREADS) dst[i] = src[i];
        if (tid == 0) S = G->s;
    }
    __syncthreads();
#ifdef MML_FW_TIMING
    if (tid == 0) S.t_last = wall_clock64();
#endif
    // the normal equations at the candidate, every element summed in the order of the host assembly
    const int e = A.round == 0 ? 0 : 1 - S.cur;
    // (every load is issued whether its term exists or not -- from a clamped address, the value then replaced by +0.0,
    //  which leaves the sum unchanged -- so the loads of several elements are in flight together)
    const int has_prior = P->has_prior;
    unsigned have = 0;  // bit f: IMU factor f present
    for (int f = 1; f < W; ++f) have |= P->have_imu[f] ? 1u << f : 0u;
#pragma unroll 4
    for (int o = tid; o < n * FW_BW; o += FW_THREADS) {
        const int a = o / FW_BW, F = a / 15, la = a - 15 * F;
        const int b = 15 * (F - 1) + (o - FW_BW * a);
        const bool inb = b >= 0 && b < n;
        const int bb = inb ? b : a, Gb = bb / 15, lb = bb - 15 * Gb;
        const bool dg = Gb == F, up = Gb == F + 1, lo = Gb == F - 1;
        const int F1 = min(F + 1, MAXW - 1);
        const bool use_l = inb && dg && la < 6 && lb < 6;
        const bool use_1 = inb && (dg || lo) && ((have >> F) & 1u);        // IMU factor F (frames F - 1, F)
        const bool use_2 = inb && (dg || up) && ((have >> (F + 1)) & 1u);  // IMU factor F + 1 (frames F, F + 1)
        const bool use_p = inb && dg && F == 0 && has_prior;
        const int la6 = min(la, 5), lb6 = min(lb, 5);
        const double v_l = G->rec[F][la6 <= lb6 ? tri(la6, lb6) : tri(lb6, la6)];
        const double v_1 = G->JJ[F][(15 + la) * 30 + (dg ? 15 + lb : lb)];
        const double v_2 = G->JJ[F1][la * 30 + (dg ? lb : 15 + lb)];
        const double v_p = P->PP[la * 15 + lb];
        double h = 0.0;
        h += use_l ? v_l : 0.0;
        h += use_1 ? v_1 : 0.0;
        h += use_2 ? v_2 : 0.0;
        h += use_p ? v_p : 0.0;
        sh.H[o] = h;
        G->Hb[e][o] = h;
    }
    if (tid < n) {
        const int F = tid / 15, la = tid - 15 * F;
        double g = 0.0;
        if (la < 6) g += G->rec[F][21 + la];
        if (F >= 1 && P->have_imu[F]) g += G->Jr[F][15 + la];
        if (F + 1 < W && P->have_imu[F + 1]) g += G->Jr[F + 1][la];
        if (F == 0 && P->has_prior) g += G->gp[la];
        V.g[e][tid] = g;
    }
    if (tid == FW_THREADS - 1) {
        double cost = 0;
        for (int f = 0; f < W; ++f) cost += G->rec[f][27];
        for (int f = 1; f < W; ++f)
            if (P->have_imu[f]) cost += 0.5 * G->rr[f];
        if (P->has_prior) cost += 0.5 * G->rp2;
        S.cost[e] = cost;
        S.evals++;
        S.steps++;
    }
    __syncthreads();
    FW_T(0);
    bool done;
    if (A.round == 0) {
        if (tid < n) V.scale[tid] = 1.0 / (1.0 + sqrt(sh.H[tid * FW_BW + (tid - band0(tid))]));  // Jacobi scaling
        if (tid == 0) {
            S.initial_cost = S.cost[0];
            S.x_norm = sqrt(ordered_dot(V.x, V.x, n));
            S.flag = 0;
            if (!P->fixed) {
                double gm = 0;
                for (int i = 0; i < n; ++i) gm = fmax(gm, fabs(V.g[0][i]));
                if (gm <= 1e-10) {
                    S.termination = 1;
                    S.flag = 1;
                }
            }
        }
        __syncthreads();
        done = S.flag != 0;
    } else {
        done = fw_decide(A, sh, n);
    }
    FW_T(1);
    if (!done) {
        int p;
        do p = fw_propose(P, sh, n);
        while (p == 0);
        if (p < 0) done = true;
    }
    __syncthreads();
    if (tid == 0) {
        if (done) S.go = 0;
        G->s = S;
    }
    {
        double* dst = reinterpret_cast<double*>(&G->v);
        const double* src = reinterpret_cast<const double*>(&V);
        for (int i = tid; i < (int)(sizeof(FwVectors) / sizeof(double)); i += FW_THREADS) dst[i] = src[i];
    }
    if (done || A.last) {
        if (tid < n) A.out->x[tid] = V.x[tid];
        if (tid == 0) {
            A.out->initial_cost = S.initial_cost;
            A.out->final_cost = S.cost[S.cur];
            A.out->iterations = S.iter;
            A.out->successful = S.successful;
            A.out->termination = S.termination;
            A.out->evaluations = S.evals;
            A.out->steps = S.steps;
#ifdef MML_FW_TIMING
            for (int k = 0; k < 16; ++k) A.out->ticks[k] = S.ticks[k];
#endif
        }
    }
}

__global__ void k_fw_init(const FwDevParams* P, FwGlobal* G) {
    const int tid = threadIdx.x, n = 15 * P->W;
    if (tid < n) G->v.x[tid] = G->v.xc[tid] = G->v.x_init[tid] = P->x[tid];
    if (tid == 0) {
        FwScalars s;
        memset(&s, 0, sizeof(s));
        s.radius = 1e4;
        s.mu = 1e-8;
        s.go = 1;
        G->s = s;
    }
}

}  // namespace

struct MmlFwDev {
    FwDevParams* d_par = nullptr;
    FwGlobal* d_state = nullptr;
    FwDevOut* d_out = nullptr;
    FwDevParams* h_par = nullptr;  // pinned
    FwDevOut* h_out = nullptr;     // pinned
};

void mml_fullwindow_dev_release(mml_ctx* ctx) {
    MmlFwDev* d = ctx->fwdev;
    if (!d) return;
    if (d->d_par) hipFree(d->d_par);
    if (d->d_state) hipFree(d->d_state);
    if (d->d_out) hipFree(d->d_out);
    if (d->h_par) hipHostFree(d->h_par);
    if (d->h_out) hipHostFree(d->h_out);
    delete d;
    ctx->fwdev = nullptr;
}

extern "C" int mml_fullwindow_solve(mml_ctx* ctx, mml_fullwindow* fw, int first_slot, const double* T_bl, double* x,
                                    mml_solve_summary* summary, int* evaluations) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE(fw && T_bl && x, MML_ERR_INVALID, "mml_fullwindow_solve: null argument");
    const int W = fw->W;
    MML_REQUIRE(W >= 1 && W <= MAXW && first_slot >= 0 && first_slot + W <= ctx->B, MML_ERR_INVALID,
                "mml_fullwindow_solve: window does not fit the scan slots");
    MML_REQUIRE(fw->opts.max_num_iterations >= 0 && fw->opts.max_num_iterations <= 1000, MML_ERR_INVALID,
                "mml_fullwindow_solve: max_num_iterations out of range");
    MML_HIP(hipSetDevice(ctx->device));
    if (!ctx->fwdev) {
        MmlFwDev* d = new MmlFwDev();
        ctx->fwdev = d;
        MML_HIP(hipMalloc(reinterpret_cast<void**>(&d->d_par), sizeof(FwDevParams)));
        MML_HIP(hipMalloc(reinterpret_cast<void**>(&d->d_state), sizeof(FwGlobal)));
        MML_HIP(hipMalloc(reinterpret_cast<void**>(&d->d_out), sizeof(FwDevOut)));
        MML_HIP(hipHostMalloc(reinterpret_cast<void**>(&d->h_par), sizeof(FwDevParams), hipHostMallocDefault));
        MML_HIP(hipHostMalloc(reinterpret_cast<void**>(&d->h_out), sizeof(FwDevOut), hipHostMallocDefault));
    }
    MmlFwDev* d = ctx->fwdev;
    FwDevParams& p = *d->h_par;
    memset(&p, 0, sizeof(p));
    p.W = W;
    p.max_iters = fw->opts.max_num_iterations;
    p.fixed = fw->opts.fixed_iterations;
    p.first = first_slot;
    p.huber = fw->opts.huber_delta;
    p.w_tan = fw->opts.plan_weight_tan;
    memcpy(p.gravity, fw->gravity, sizeof(p.gravity));
    memcpy(p.Tbl, T_bl, sizeof(p.Tbl));
    memcpy(p.x, x, sizeof(double) * 15 * W);
    p.has_prior = fw->prior.valid ? 1 : 0;
    if (fw->prior.valid) {
        memcpy(p.prior.J, fw->prior.J, sizeof(p.prior.J));
        memcpy(p.prior.r0, fw->prior.r0, sizeof(p.prior.r0));
        memcpy(p.prior.x0, fw->prior.x0, sizeof(p.prior.x0));
        for (int a = 0; a < 15; ++a)
            for (int b = 0; b < 15; ++b) {
                double h = 0;
                for (int i = 0; i < 15; ++i) h += fw->prior.J[i * 15 + a] * fw->prior.J[i * 15 + b];
                p.PP[a * 15 + b] = h;
            }
    }
    for (int f = 1; f < W; ++f) {
        if (!fw->have_imu[f]) continue;
        p.have_imu[f] = 1;
        p.imu[f] = fw->imu[f];
        MML_REQUIRE(fw->U_ok[f], MML_ERR_STATE, "mml_fullwindow_solve: pre-integration covariance is not positive definite");
        memcpy(p.U[f], &fw->U[225 * (size_t)f], sizeof(p.U[f]));
    }
    hipStream_t s = MML_STREAM(ctx);
    MmlStageScope t(ctx, "fullwindow");
    MML_HIP(hipMemcpyAsync(d->d_par, d->h_par, sizeof(FwDevParams), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_fw_init, dim3(1), dim3(128), 0, s, d->d_par, d->d_state);
    FwKernelArgs a;
    a.P = d->d_par;
    a.G = d->d_state;
    a.out = d->d_out;
    a.ft_n = ctx->ft_n;
    a.lf = ctx->lf;
    a.pf = ctx->pf;
    a.B = ctx->B;
    a.MF = ctx->MF;
    // at most max_iterations + 1 evaluations: the first point and one candidate per iteration (an invalid step consumes
    // an iteration without an evaluation).  Enqueued in chunks of four; between chunks the `go` flag comes back (one
    // 4-byte copy), so that a solve that converged early does not pay for the launches of the remaining no-op rounds.
    const int rounds = p.max_iters + 1;
    for (int r = 0; r < rounds; ++r) {
        a.round = r;
        a.last = r + 1 == rounds;
        hipLaunchKernelGGL(k_fw_eval, dim3(2 * W), dim3(SOLVE_THREADS), 0, s, a);
        hipLaunchKernelGGL(k_fw_step, dim3(1), dim3(FW_THREADS), 0, s, a);
        if ((r & 3) == 3 && r + 1 < rounds) {
            MML_HIP(hipGetLastError());
            MML_HIP(hipMemcpyAsync(&d->h_out->pad_, &d->d_state->s.go, sizeof(int), hipMemcpyDeviceToHost, s));
            MML_HIP(hipStreamSynchronize(s));
            if (!d->h_out->pad_) break;
        }
    }
    MML_HIP(hipGetLastError());
    MML_HIP(hipMemcpyAsync(d->h_out, d->d_out, sizeof(FwDevOut), hipMemcpyDeviceToHost, s));
    MML_HIP(hipStreamSynchronize(s));
    const FwDevOut& o = *d->h_out;
#ifdef MML_FW_TIMING
    {
        static const char* names[12] = {"assemble", "decide", "prep", "build", "factor_solve_rest", "dogleg", "xc", "fs_setup", "fs_chol", "fs_fwd", "fs_bwd", "chol_shader_clocks/100"};
        fprintf(stderr, "[fw step timing W=%d evals=%d iters=%d] us:", W, o.evaluations, o.iterations);
        for (int k = 0; k < 12; ++k) fprintf(stderr, " %s=%.1f", names[k], o.ticks[k] * 0.01);
        fprintf(stderr, "\n");
    }
#endif
    memcpy(x, o.x, sizeof(double) * 15 * W);
    // the handle reports this solve through mml_fullwindow_summary, and marginalizes at the returned x
    fw->iter = o.iterations;
    fw->successful = o.successful;
    fw->termination = o.termination;
    fw->initial_cost = o.initial_cost;
    fw->cur.cost = o.final_cost;
    fw->started = fw->done = 1;
    fw->x.assign(x, x + 15 * W);
    if (summary) {
        summary->iterations = o.iterations;
        summary->successful = o.successful;
        summary->initial_cost = o.initial_cost;
        summary->final_cost = o.final_cost;
        summary->termination = o.termination;
    }
    if (evaluations) *evaluations = o.evaluations;
    return MML_OK;
}
