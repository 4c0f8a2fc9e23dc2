// solve.hip -- rows a17..a21 of SURVEY.md section 8.
//   a17..a20 residuals, analytic Jacobians, Huber correction and the J^T J / J^T r reduction: lidar_eval.h
//   a20/a21 ceres::Solve replacement (mm-loam/src/lio/Estimator.cpp:1425-1432): Ceres 2.1.0 trust-region loop with
//           TRADITIONAL_DOGLEG and Jacobi scaling, one workgroup per window problem, no host round trip between
//           the iterations.  The 6x6 per-frame Cholesky and the dogleg bookkeeping run on lane 0.
// Each iteration makes ONE fused pass over the factors at the candidate point (cost + H + g), instead of Ceres'
// cost-only pass followed by a Jacobian pass after acceptance.
#include <math.h>
#include <stdlib.h>

#include "lidar_eval.h"
#include "mml_internal.h"

namespace {

// 6x6 Cholesky solve on one lane.  A (36, row-major) is destroyed; b -> x.  false if not positive definite.
__host__ __device__ bool chol6(double* A, double* b) {
    for (int j = 0; j < 6; ++j) {
        double d = A[j * 6 + j];
        for (int k = 0; k < j; ++k) d -= A[j * 6 + k] * A[j * 6 + k];
        if (!(d > 0.0)) return false;
        d = sqrt(d);
        A[j * 6 + j] = d;
        for (int i = j + 1; i < 6; ++i) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s / d;
        }
    }
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= A[i * 6 + k] * b[k];
        b[i] = s / A[i * 6 + i];
    }
    for (int i = 5; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < 6; ++k) s -= A[k * 6 + i] * b[k];
        b[i] = s / A[i * 6 + i];
    }
    return true;
}

struct SolveParams {
    int first, B, MF, window, max_iters, fixed;
    double huber, w_tan;
    const int* ft_n;
    const MmlLineFactor* lf;
    const MmlPlaneFactor* pf;
    const double* Tbl;  // 16
    double* x;          // B * 6
    double* summ;       // per problem 8 doubles
    double* trace;      // per problem max_iters * 6 * window, or nullptr
    // a handful of slots (the live path): every small transfer is a launch of its own in the chain, so the start poses come in one
    // copy with the call's other parameter blocks and everything the host reads back afterwards leaves in ONE record per problem
    const double* x_in;     // start poses of THIS launch's problems (6 * window each), nullptr: x holds them
    double* result;         // per problem MML_SOLVE_RESULT doubles, or nullptr: x (6 * window <= 48 ... window 1 only), the two stack
                            // sizes of slot b0, the 16 association statistics of slot b0
    const double* stats;    // assoc_stats (16 doubles per slot) or nullptr
    int pairs;              // k_solve<true>: two plane factors of a thread side by side (measurement switch $MML_SOLVE_PAIRS=0: off)
};

// Trust-region state kept in LDS, manipulated by lane 0 (restates ceres 2.1.0 trust_region_minimizer.cc +
// dogleg_strategy.cc; constants are Ceres defaults, see oracle/estimate.cpp for the line-by-line commentary).
struct TRState {
    double x[6 * MAXW], xc[6 * MAXW], x_init[6 * MAXW];
    double rec[28 * MAXW], recc[28 * MAXW];  // per frame: H upper (21), g (6), cost
    double scale[6 * MAXW], diag[6 * MAXW], grad[6 * MAXW], gn[6 * MAXW], step[6 * MAXW];
    double work[42];  // the 6 x 6 system of one frame while tr_propose factors it (kept out of private memory)
    double cost, radius, mu, alpha, dogleg_norm, x_norm, model_change, step_norm;
    int reuse, num_invalid, iter, successful, termination, go, evaluate;
};

__host__ __device__ double quad_form(const TRState& S, int W, const double* v) {  // v^T (S H S) v
    double q = 0;
    for (int f = 0; f < W; ++f)
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b)
                q += v[6 * f + a] * (Hget(S.rec + 28 * f, a, b) * S.scale[6 * f + a] * S.scale[6 * f + b]) * v[6 * f + b];
    return q;
}

__host__ __device__ inline __attribute__((always_inline)) void tr_propose(TRState& S, int W, int max_iters) {
    const int n = 6 * W;
    S.evaluate = 0;
    if (S.iter >= max_iters || S.radius < 1e-32) {
        S.go = 0;
        return;
    }
    S.iter++;
    bool solve_ok = true;
    if (!S.reuse) {
        S.reuse = 1;
        for (int f = 0; f < W; ++f)
            for (int i = 0; i < 6; ++i) {
                double d = Hget(S.rec + 28 * f, i, i) * S.scale[6 * f + i] * S.scale[6 * f + i];
                d = fmin(fmax(d, 1e-6), 1e32);
                S.diag[6 * f + i] = sqrt(d);
            }
        double gg = 0;
        double* sg = S.step;  // scratch: the step itself is only formed further down
        for (int f = 0; f < W; ++f)
            for (int i = 0; i < 6; ++i) {
                int k = 6 * f + i;
                S.grad[k] = S.rec[28 * f + 21 + i] * S.scale[k] / S.diag[k];
                sg[k] = S.grad[k] / S.diag[k];
                gg += S.grad[k] * S.grad[k];
            }
        S.alpha = gg / quad_form(S, W, sg);
        solve_ok = false;
        while (S.mu < 1.0) {
            bool ok = true;
            for (int f = 0; f < W && ok; ++f) {
                double* A = S.work;
                double* bvec = S.work + 36;
                for (int a = 0; a < 6; ++a) {
                    for (int b = 0; b < 6; ++b)
                        A[6 * a + b] = Hget(S.rec + 28 * f, a, b) * S.scale[6 * f + a] * S.scale[6 * f + b];
                    A[7 * a] += S.mu * S.diag[6 * f + a] * S.diag[6 * f + a];
                    bvec[a] = S.rec[28 * f + 21 + a] * S.scale[6 * f + a];
                }
                ok = chol6(A, bvec);
                for (int a = 0; a < 6 && ok; ++a) {
                    if (!isfinite(bvec[a])) ok = false;
                    S.gn[6 * f + a] = bvec[a];
                }
            }
            if (!ok) {
                S.mu *= 10.0;
                continue;
            }
            solve_ok = true;
            break;
        }
        if (solve_ok)
            for (int i = 0; i < n; ++i) S.gn[i] *= -S.diag[i];
    }
    bool step_valid = solve_ok;
    if (solve_ok) {
        double gradient_norm = 0, gn_norm = 0;
        for (int i = 0; i < n; ++i) {
            gradient_norm += S.grad[i] * S.grad[i];
            gn_norm += S.gn[i] * S.gn[i];
        }
        gradient_norm = sqrt(gradient_norm);
        gn_norm = sqrt(gn_norm);
        if (gn_norm <= S.radius) {
            for (int i = 0; i < n; ++i) S.step[i] = S.gn[i];
            S.dogleg_norm = gn_norm;
        } else if (gradient_norm * S.alpha >= S.radius) {
            for (int i = 0; i < n; ++i) S.step[i] = -(S.radius / gradient_norm) * S.grad[i];
            S.dogleg_norm = S.radius;
        } else {
            double gdot = 0;
            for (int i = 0; i < n; ++i) gdot += S.grad[i] * S.gn[i];
            double b_dot_a = -S.alpha * gdot;
            double a_sq = (S.alpha * gradient_norm) * (S.alpha * gradient_norm);
            double bma_sq = a_sq - 2 * b_dot_a + gn_norm * gn_norm;
            double c = b_dot_a - a_sq;
            double d = sqrt(c * c + bma_sq * (S.radius * S.radius - a_sq));
            double beta = (c <= 0) ? (d - c) / bma_sq : (S.radius * S.radius - a_sq) / (d + c);
            double sn = 0;
            for (int i = 0; i < n; ++i) {
                S.step[i] = (-S.alpha * (1.0 - beta)) * S.grad[i] + beta * S.gn[i];
                sn += S.step[i] * S.step[i];
            }
            S.dogleg_norm = sqrt(sn);
        }
        for (int i = 0; i < n; ++i) S.step[i] /= S.diag[i];
        double sgd = 0;
        for (int f = 0; f < W; ++f)
            for (int i = 0; i < 6; ++i) sgd += S.step[6 * f + i] * S.rec[28 * f + 21 + i] * S.scale[6 * f + i];
        S.model_change = -(sgd + 0.5 * quad_form(S, W, S.step));
        if (!(S.model_change > 0.0)) step_valid = false;
    }
    if (!step_valid) {
        // TrustRegionMinimizer::HandleInvalidStep: the max_num_consecutive_invalid_steps-th (5th) invalid step in a row
        // ends the solve with FAILURE before the strategy is told; Solver::Solve then hands the parameters back as they
        // were on entry (Summary::IsSolutionUsable() is false)
        if (++S.num_invalid >= 5) {
            for (int i = 0; i < n; ++i) S.x[i] = S.x_init[i];
            S.termination = 4;
            S.go = 0;
            return;
        }
        S.mu *= 10.0;
        S.reuse = 0;
        return;  // go stays 1, evaluate 0: next round proposes again
    }
    S.num_invalid = 0;
    double sn = 0;
    for (int i = 0; i < n; ++i) {
        double delta = S.step[i] * S.scale[i];
        S.xc[i] = S.x[i] + delta;
        sn += delta * delta;
    }
    S.step_norm = sqrt(sn);
    S.evaluate = 1;
}

// tr_propose for the one-frame problem of the live mode (W = 1), run by the 64 lanes of the first wavefront.  The
// generic form is a single lane walking ~50 double-precision square roots and divisions and two 36-term quadratic forms
// one after the other while the rest of the workgroup waits; here everything that is independent runs on its own lane
// (the 36 matrix entries, the 6 diagonal / gradient entries, the rows of a Cholesky column) and only the sums whose
// order of addition defines the result stay serial.  Every value is computed by the same expression, and every sum
// accumulated in the same order, as in tr_propose: the iterates are identical.
// w: 92 doubles of LDS.  Lanes exchange through LDS; inside one wavefront LDS operations execute in program order, so a
// compiler barrier is all the synchronisation needed.
#define WSYNC()                          \
    do {                                 \
        __builtin_amdgcn_wave_barrier(); \
        asm volatile("" ::: "memory");   \
    } while (0)
#ifdef MML_SV_TIMING
// phase clocks of one k_solve workgroup (problem MML_SV_TIMING of every launch): cycles per phase, summed over the iterations and
// launches; [7] counts the launches
__device__ unsigned long long g_sv_dbg[16];
#define SV_MARK(id)                                    \
    do {                                               \
        if (sv_dbg) {                                  \
            const unsigned long long now_ = clock64(); \
            g_sv_dbg[id] += now_ - sv_prev;            \
            sv_prev = now_;                            \
        }                                              \
    } while (0)
extern "C" int mml_debug_sv_timing(unsigned long long* out, int reset) {
    if (reset) {
        unsigned long long z[16] = {};
        return hipMemcpyToSymbol(HIP_SYMBOL(g_sv_dbg), z, sizeof(z)) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sv_dbg), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
extern "C" int mml_debug_svw_timing(unsigned long long* out, int reset) {
    if (reset) {
        unsigned long long z[8] = {};
        return hipMemcpyToSymbol(HIP_SYMBOL(g_svw_dbg), z, sizeof(z)) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_svw_dbg), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#else
#define SV_MARK(id)
#endif
#ifdef MML_SV_TIMING
#define PV_MARK(id)                                                         \
    do {                                                                    \
        if (threadIdx.x == 0 && blockIdx.x == MML_SV_TIMING) {              \
            const unsigned long long now_ = clock64();                      \
            g_sv_dbg[id] += now_ - pv_prev;                                 \
            pv_prev = now_;                                                 \
        }                                                                   \
    } while (0)
#else
#define PV_MARK(id)
#endif
// (forced inline, as tr_propose and tr_decide_wave: with two instantiations of k_solve calling them the compiler made them real
//  functions -- a call inside the iteration loop of a kernel with 256 live registers: the batch solve 0.251 -> 0.28 ms per 1024 problems)
// (lane: threadIdx.x of the first wavefront; k_solve_wide hands over an opaque copy per iteration, see there)
__device__ __forceinline__ void tr_propose_w1_wave(TRState& S, double* w, int max_iters, const int lane = threadIdx.x) {
    double* T = w;         // 36 terms of a quadratic form
    double* A = w + 36;    // 36: the damped matrix, then its Cholesky factor (lower triangle)
    double* bv = w + 72;   // 6
    double* sg = w + 78;   // 6
    double* st = w + 84;   // 6: the step
    double* fl = w + 90;   // 2 flags
    const int iter = S.iter, num_invalid = S.num_invalid, reuse = S.reuse;
    const double radius = S.radius;
#ifdef MML_SV_TIMING
    unsigned long long pv_prev = clock64();
#endif
    WSYNC();
    if (lane == 0) S.evaluate = 0;
    if (iter >= max_iters || radius < 1e-32) {
        if (lane == 0) S.go = 0;
        return;
    }
    if (lane == 0) S.iter = iter + 1;
    const int a = lane / 6, b = lane - 6 * a;
    const bool in36 = lane < 36, in6 = lane < 6;
    // entry (a, b) of S H S as quad_form and the matrix build compute it
    const double mab = in36 ? Hget(S.rec, a, b) * S.scale[a] * S.scale[b] : 0.0;
    bool solve_ok = true;
    if (!reuse) {
        if (lane == 0) S.reuse = 1;
        if (in6) {
            double d = Hget(S.rec, lane, lane) * S.scale[lane] * S.scale[lane];
            d = fmin(fmax(d, 1e-6), 1e32);
            const double dg = sqrt(d);
            S.diag[lane] = dg;
            const double gr = S.rec[21 + lane] * S.scale[lane] / dg;
            S.grad[lane] = gr;
            sg[lane] = gr / dg;
        }
        WSYNC();
        if (in36) T[lane] = sg[a] * mab * sg[b];
        WSYNC();
        if (lane == 0) {
            double gg = 0, q = 0;
            for (int k = 0; k < 6; ++k) gg += S.grad[k] * S.grad[k];
            for (int k = 0; k < 36; ++k) q += T[k];
            S.alpha = gg / q;
        }
        PV_MARK(8);  // diag / gradient / alpha
        double mu = S.mu;
        solve_ok = false;
        while (mu < 1.0) {
            if (in36) A[lane] = (a == b) ? mab + mu * S.diag[a] * S.diag[a] : mab;
            if (in6) bv[lane] = S.rec[21 + lane] * S.scale[lane];
            WSYNC();
            bool ok = true;
            for (int j = 0; j < 6; ++j) {
                if (lane == 0) {
                    double d = A[7 * j];
                    for (int k = 0; k < j; ++k) d -= A[j * 6 + k] * A[j * 6 + k];
                    const bool pos = d > 0.0;
                    fl[0] = pos ? 1.0 : 0.0;
                    if (pos) A[7 * j] = sqrt(d);
                }
                WSYNC();
                if (fl[0] == 0.0) {
                    ok = false;
                    break;
                }
                if (lane > j && lane < 6) {
                    double t = A[lane * 6 + j];
                    for (int k = 0; k < j; ++k) t -= A[lane * 6 + k] * A[j * 6 + k];
                    A[lane * 6 + j] = t / A[7 * j];
                }
                WSYNC();
            }
            PV_MARK(9);  // Cholesky
            if (ok) {
                if (lane == 0) {
                    for (int i = 0; i < 6; ++i) {
                        double t = bv[i];
                        for (int k = 0; k < i; ++k) t -= A[i * 6 + k] * bv[k];
                        bv[i] = t / A[i * 6 + i];
                    }
                    bool fin = true;
                    for (int i = 5; i >= 0; --i) {
                        double t = bv[i];
                        for (int k = i + 1; k < 6; ++k) t -= A[k * 6 + i] * bv[k];
                        bv[i] = t / A[i * 6 + i];
                    }
                    for (int i = 0; i < 6; ++i) fin = fin && isfinite(bv[i]);
                    fl[1] = fin ? 1.0 : 0.0;
                }
                WSYNC();
                ok = fl[1] != 0.0;
            }
            WSYNC();
            if (!ok) {
                mu *= 10.0;
                continue;
            }
            solve_ok = true;
            break;
        }
        if (lane == 0) S.mu = mu;
        if (solve_ok && in6) S.gn[lane] = bv[lane] * -S.diag[lane];
        WSYNC();
        PV_MARK(10);  // substitutions
    }
    bool step_valid = solve_ok;
    if (solve_ok) {
        if (lane == 0) {
            const double alpha = S.alpha;
            double gradient_norm = 0, gn_norm = 0;
            for (int i = 0; i < 6; ++i) {
                gradient_norm += S.grad[i] * S.grad[i];
                gn_norm += S.gn[i] * S.gn[i];
            }
            gradient_norm = sqrt(gradient_norm);
            gn_norm = sqrt(gn_norm);
            if (gn_norm <= radius) {
                for (int i = 0; i < 6; ++i) st[i] = S.gn[i];
                S.dogleg_norm = gn_norm;
            } else if (gradient_norm * alpha >= radius) {
                for (int i = 0; i < 6; ++i) st[i] = -(radius / gradient_norm) * S.grad[i];
                S.dogleg_norm = radius;
            } else {
                double gdot = 0;
                for (int i = 0; i < 6; ++i) gdot += S.grad[i] * S.gn[i];
                double b_dot_a = -alpha * gdot;
                double a_sq = (alpha * gradient_norm) * (alpha * gradient_norm);
                double bma_sq = a_sq - 2 * b_dot_a + gn_norm * gn_norm;
                double c = b_dot_a - a_sq;
                double d = sqrt(c * c + bma_sq * (radius * radius - a_sq));
                double beta = (c <= 0) ? (d - c) / bma_sq : (radius * radius - a_sq) / (d + c);
                double sn = 0;
                for (int i = 0; i < 6; ++i) {
                    st[i] = (-alpha * (1.0 - beta)) * S.grad[i] + beta * S.gn[i];
                    sn += st[i] * st[i];
                }
                S.dogleg_norm = sqrt(sn);
            }
            for (int i = 0; i < 6; ++i) {
                st[i] /= S.diag[i];
                S.step[i] = st[i];
            }
        }
        WSYNC();
        PV_MARK(11);  // dogleg
        if (in36) T[lane] = st[a] * mab * st[b];
        WSYNC();
        if (lane == 0) {
            double sgd = 0, q = 0;
            for (int i = 0; i < 6; ++i) sgd += st[i] * S.rec[21 + i] * S.scale[i];
            for (int k = 0; k < 36; ++k) q += T[k];
            const double mc = -(sgd + 0.5 * q);
            S.model_change = mc;
            fl[0] = (mc > 0.0) ? 1.0 : 0.0;
        }
        WSYNC();
        step_valid = fl[0] != 0.0;
        PV_MARK(12);  // model change
    }
    if (!step_valid) {
        if (lane == 0) {
            if (num_invalid + 1 >= 5) {  // HandleInvalidStep: FAILURE, parameters as on entry (see tr_propose)
                for (int i = 0; i < 6; ++i) S.x[i] = S.x_init[i];
                S.termination = 4;
                S.go = 0;
            } else {
                S.mu *= 10.0;
                S.reuse = 0;
            }
            S.num_invalid = num_invalid + 1;
        }
        return;  // go stays 1 unless failed, evaluate 0: next round proposes again
    }
    if (lane == 0) {
        S.num_invalid = 0;
        double sn = 0;
        for (int i = 0; i < 6; ++i) {
            double delta = st[i] * S.scale[i];
            S.xc[i] = S.x[i] + delta;
            sn += delta * delta;
        }
        S.step_norm = sqrt(sn);
        S.evaluate = 1;
    }
}
#undef WSYNC

// tr_propose for W = 1 once more, for k_solve_wide: the scalar algorithm of tr_propose on REGISTER copies of the state, every lane of
// the first wavefront computing the same values (lane 0 stores them).  tr_propose_w1_wave spreads the independent entries over lanes
// and exchanges them through LDS: ~120 dependent LDS round trips of ~130 cycles per call -- 15.7 k cycles, 45 % of the live path's
// solve.  With no exchange at all the ~1 700 double-precision instructions of the scalar form run back to back out of registers.
// Every value comes from the expression tr_propose (and tr_propose_w1_wave) forms it with, every sum is added in their order.
// (w: 36 doubles of LDS: the matrix S H S is formed once, parked there by lane 0 and read back -- one broadcast round trip -- where the
//  damped system and the second quadratic form need it, instead of being held in 72 registers across the Cholesky factorisation)
#define RSYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
__device__ __forceinline__ void tr_propose_w1_regs(TRState& S, double* w, int max_iters, const int lane) {
    const int iter = S.iter, num_invalid = S.num_invalid, reuse = S.reuse;
    const double radius = S.radius;
    if (lane == 0) S.evaluate = 0;
    if (iter >= max_iters || radius < 1e-32) {
        if (lane == 0) S.go = 0;
        return;
    }
    if (lane == 0) S.iter = iter + 1;
    double rec[28], scale[6];
#pragma unroll
    for (int k = 0; k < 28; ++k) rec[k] = S.rec[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) scale[k] = S.scale[k];
    double diag[6], grad[6], gn[6], alpha = S.alpha, mu = S.mu;
#pragma unroll
    for (int k = 0; k < 6; ++k) {  // (what a call with reuse = 1 finds from the call before)
        diag[k] = S.diag[k];
        grad[k] = S.grad[k];
        gn[k] = S.gn[k];
    }
    // entry (a, b) of S H S as quad_form and the matrix build compute it
    double M[36];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) M[6 * a + b] = Hget(rec, a, b) * scale[a] * scale[b];
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 36; ++k) w[k] = M[k];
    }
    bool solve_ok = true;
    int reuse_out = reuse;
    if (!reuse) {
        reuse_out = 1;
        double sg[6], gg = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double d = M[7 * i];  // (= Hget(rec, i, i) * scale[i] * scale[i])
            d = fmin(fmax(d, 1e-6), 1e32);
            diag[i] = sqrt(d);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            grad[i] = rec[21 + i] * scale[i] / diag[i];
            sg[i] = grad[i] / diag[i];
            gg += grad[i] * grad[i];
        }
        double q = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) q += sg[a] * M[6 * a + b] * sg[b];
        alpha = gg / q;
        solve_ok = false;
        while (mu < 1.0) {
            double A[36], bv[6];
            RSYNC();
#pragma unroll
            for (int a = 0; a < 6; ++a) {
#pragma unroll
                for (int b = 0; b <= a; ++b) A[6 * a + b] = w[6 * a + b];  // (chol6 reads the lower triangle only)
                A[7 * a] += mu * diag[a] * diag[a];
                bv[a] = rec[21 + a] * scale[a];
            }
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 6; ++j) {  // chol6, the lower triangle
                double d = A[7 * j];
#pragma unroll
                for (int k = 0; k < j; ++k) d -= A[j * 6 + k] * A[j * 6 + k];
                ok = ok && d > 0.0;
                d = sqrt(d);
                A[7 * j] = d;
#pragma unroll
                for (int i = j + 1; i < 6; ++i) {
                    double t = A[i * 6 + j];
#pragma unroll
                    for (int k = 0; k < j; ++k) t -= A[i * 6 + k] * A[j * 6 + k];
                    A[i * 6 + j] = t / d;
                }
            }
            if (ok) {  // (a pivot that is not positive ends chol6 there: nothing behind it is used)
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    double t = bv[i];
#pragma unroll
                    for (int k = 0; k < i; ++k) t -= A[i * 6 + k] * bv[k];
                    bv[i] = t / A[i * 6 + i];
                }
#pragma unroll
                for (int i = 5; i >= 0; --i) {
                    double t = bv[i];
#pragma unroll
                    for (int k = i + 1; k < 6; ++k) t -= A[k * 6 + i] * bv[k];
                    bv[i] = t / A[i * 6 + i];
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) ok = ok && isfinite(bv[i]);
            }
            if (!ok) {
                mu *= 10.0;
                continue;
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) gn[i] = bv[i] * -diag[i];
            solve_ok = true;
            break;
        }
    }
    double mu_out = mu;
    bool step_valid = solve_ok;
    double st[6] = {0, 0, 0, 0, 0, 0}, dogleg_norm = 0, model_change = 0;
    if (solve_ok) {
        double gradient_norm = 0, gn_norm = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            gradient_norm += grad[i] * grad[i];
            gn_norm += gn[i] * gn[i];
        }
        gradient_norm = sqrt(gradient_norm);
        gn_norm = sqrt(gn_norm);
        if (gn_norm <= radius) {
#pragma unroll
            for (int i = 0; i < 6; ++i) st[i] = gn[i];
            dogleg_norm = gn_norm;
        } else if (gradient_norm * alpha >= radius) {
#pragma unroll
            for (int i = 0; i < 6; ++i) st[i] = -(radius / gradient_norm) * grad[i];
            dogleg_norm = radius;
        } else {
            double gdot = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i) gdot += grad[i] * gn[i];
            const double b_dot_a = -alpha * gdot;
            const double a_sq = (alpha * gradient_norm) * (alpha * gradient_norm);
            const double bma_sq = a_sq - 2 * b_dot_a + gn_norm * gn_norm;
            const double c = b_dot_a - a_sq;
            const double d = sqrt(c * c + bma_sq * (radius * radius - a_sq));
            const double beta = (c <= 0) ? (d - c) / bma_sq : (radius * radius - a_sq) / (d + c);
            double sn = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                st[i] = (-alpha * (1.0 - beta)) * grad[i] + beta * gn[i];
                sn += st[i] * st[i];
            }
            dogleg_norm = sqrt(sn);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) st[i] /= diag[i];
        double sgd = 0, q = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) sgd += st[i] * rec[21 + i] * scale[i];
        RSYNC();
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) q += st[a] * w[6 * a + b] * st[b];
        model_change = -(sgd + 0.5 * q);
        step_valid = model_change > 0.0;
    }
    if (lane == 0) {
        if (!reuse) {
            S.alpha = alpha;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                S.diag[i] = diag[i];
                S.grad[i] = grad[i];
                if (solve_ok) S.gn[i] = gn[i];
            }
        }
        if (solve_ok) {
            S.dogleg_norm = dogleg_norm;
            S.model_change = model_change;
#pragma unroll
            for (int i = 0; i < 6; ++i) S.step[i] = st[i];
        }
    }
    if (!step_valid) {
        if (lane == 0) {
            if (num_invalid + 1 >= 5) {  // HandleInvalidStep: FAILURE, parameters as on entry (see tr_propose)
#pragma unroll
                for (int i = 0; i < 6; ++i) S.x[i] = S.x_init[i];
                S.termination = 4;
                S.go = 0;
                S.mu = mu_out;
                S.reuse = reuse_out;
            } else {
                S.mu = mu_out * 10.0;
                S.reuse = 0;
            }
            S.num_invalid = num_invalid + 1;
        }
        return;  // go stays 1 unless failed, evaluate 0: next round proposes again
    }
    if (lane == 0) {
        // (the small integers below formed HERE: left to itself the compiler forms them in front of the solve's loop, runs out of
        //  registers for them across the factor pass and reloads them from scratch memory -- a memory round trip at this point)
        int zero = 0, one = 1;
        asm volatile("" : "+v"(zero), "+v"(one));
        S.mu = mu_out;
        S.reuse = reuse_out;
        S.num_invalid = zero;
        double sn = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double delta = st[i] * scale[i];
            S.xc[i] = S.x[i] + delta;
            sn += delta * delta;
        }
        S.step_norm = sqrt(sn);
        S.evaluate = one;
    }
}

__host__ __device__ void tr_decide(TRState& S, int W, int fixed) {
    const int n = 6 * W;
    double cand = 0;
    for (int f = 0; f < W; ++f) cand += S.recc[28 * f + 27];
    if (!fixed) {
        if (S.step_norm <= 1e-8 * (S.x_norm + 1e-8)) {
            S.termination = 2;
            S.go = 0;
            return;
        }
        if (fabs(S.cost - cand) <= 1e-6 * S.cost) {
            S.termination = 3;
            S.go = 0;
            return;
        }
    }
    double rel = (S.cost - cand) / S.model_change;
    if (rel > 1e-3) {
        double xn = 0;
        for (int i = 0; i < n; ++i) {
            S.x[i] = S.xc[i];
            xn += S.x[i] * S.x[i];
        }
        S.x_norm = sqrt(xn);
        for (int i = 0; i < 28 * W; ++i) S.rec[i] = S.recc[i];
        S.cost = cand;
        S.successful++;
        if (!fixed) {
            double gm = 0;
            for (int f = 0; f < W; ++f)
                for (int i = 0; i < 6; ++i) gm = fmax(gm, fabs(S.rec[28 * f + 21 + i]));
            if (gm <= 1e-10) {
                S.termination = 1;
                S.go = 0;
                return;
            }
        }
        if (rel < 0.25) S.radius *= 0.5;
        if (rel > 0.75) S.radius = fmax(S.radius, 3.0 * S.dogleg_norm);
        S.mu = fmax(1e-8, 2.0 * S.mu / 10.0);
        S.reuse = 0;
    } else {
        S.radius *= 0.5;
        S.reuse = 1;
    }
}

// ---- tr_propose / tr_decide of a W-frame window on the 64 lanes of one wavefront --------------------------------------
// The scalar forms above walk 6 W parameters, W six-by-six Cholesky solves and two 36 W-term quadratic forms on ONE lane
// (about 75 us for W = 8 -- ten times the frame evaluation it sits between).  Here lane k owns parameter k (6 W <= 48), lane
// f the six-by-six system of frame f, and the terms of every sum are formed in parallel; the sums themselves are then taken
// by one lane in the order the scalar code adds them (independent sums on neighbouring lanes at the same time).  Every
// value comes from the same expression and every sum from the same sequence of additions as in tr_propose / tr_decide:
// the iterates are identical, which tests/test_gpu_multi.py holds bit for bit against k_solve.
// w: TRW_DOUBLES doubles of LDS scratch.
constexpr int TRW_DOUBLES = 36 * MAXW + 3 * 64 + 8;
#define WSYNC()                          \
    do {                                 \
        __builtin_amdgcn_wave_barrier(); \
        asm volatile("" ::: "memory");   \
    } while (0)
// lane l < nsum returns t[64 l + 0] + t[64 l + 1] + ... + t[64 l + m - 1], added in that order
__device__ __forceinline__ double ordered_sums(const double* t, int m, int nsum, const int lane = threadIdx.x) {
    const double* p = t + 64 * (lane < nsum ? lane : 0);
    double s = 0;
    int k = 0;
    for (; k + 6 <= m; k += 6) {
        const double v0 = p[k], v1 = p[k + 1], v2 = p[k + 2], v3 = p[k + 3], v4 = p[k + 4], v5 = p[k + 5];
        s += v0;
        s += v1;
        s += v2;
        s += v3;
        s += v4;
        s += v5;
    }
    for (; k < m; ++k) s += p[k];
    return s;
}
// sum over the 36 W terms of a quadratic form, in (frame, row, column) order; the same value in every lane
__device__ __forceinline__ double quad_sum(const double* T, int W) {
    double s = 0;
    for (int k = 0; k < 36 * W; k += 6) {
        const double v0 = T[k], v1 = T[k + 1], v2 = T[k + 2], v3 = T[k + 3], v4 = T[k + 4], v5 = T[k + 5];
        s += v0;
        s += v1;
        s += v2;
        s += v3;
        s += v4;
        s += v5;
    }
    return s;
}
// the 36 W terms v_a (H_ab s_a s_b) v_b of quad_form
__device__ __forceinline__ void quad_terms(const TRState& S, int W, const double* v, double* T) {
    for (int t = threadIdx.x; t < 36 * W; t += 64) {
        const int f = t / 36, r = t - 36 * f, a = r / 6, b = r - 6 * a;
        T[t] = v[6 * f + a] * (Hget(S.rec + 28 * f, a, b) * S.scale[6 * f + a] * S.scale[6 * f + b]) * v[6 * f + b];
    }
}

__device__ void tr_propose_wave(TRState& S, double* w, int W, int max_iters) {
    const int lane = threadIdx.x;  // 0..63
    const int n = 6 * W;
    double* T = w;                // 36 W terms of a quadratic form
    double* U = w + 36 * MAXW;    // 3 x 64 terms of the shorter sums
    double* fl = U + 3 * 64;      // flags / broadcast values
    const int iter = S.iter, num_invalid = S.num_invalid, reuse = S.reuse;
    const double radius = S.radius;
    WSYNC();
    if (lane == 0) S.evaluate = 0;
    if (iter >= max_iters || radius < 1e-32) {
        if (lane == 0) S.go = 0;
        return;
    }
    if (lane == 0) S.iter = iter + 1;
    const bool in_n = lane < n;
    const int kf = lane / 6, ki = lane - 6 * kf;  // this lane's parameter: frame, component
    const double gk = in_n ? S.rec[28 * kf + 21 + ki] : 0.0, sk = in_n ? S.scale[lane] : 0.0;
    bool solve_ok = true;
    if (!reuse) {
        if (lane == 0) S.reuse = 1;
        if (in_n) {
            double d = Hget(S.rec + 28 * kf, ki, ki) * sk * sk;
            d = fmin(fmax(d, 1e-6), 1e32);
            const double dg = sqrt(d);
            S.diag[lane] = dg;
            const double gr = gk * sk / dg;
            S.grad[lane] = gr;
            S.step[lane] = gr / dg;  // scratch, as in tr_propose
            U[lane] = gr * gr;
        }
        WSYNC();
        quad_terms(S, W, S.step, T);
        WSYNC();
        {
            const double gg = ordered_sums(U, n, 1);
            const double q = quad_sum(T, W);
            if (lane == 0) S.alpha = gg / q;
        }
        double mu = S.mu;
        solve_ok = false;
        while (mu < 1.0) {
            bool ok = true;
            double bvec[6] = {0, 0, 0, 0, 0, 0};
            if (lane < W) {  // the damped system of frame `lane`, factored and solved in registers
                double A[36];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
#pragma unroll
                    for (int b = 0; b < 6; ++b)
                        A[6 * a + b] = Hget(S.rec + 28 * lane, a, b) * S.scale[6 * lane + a] * S.scale[6 * lane + b];
                    A[7 * a] += mu * S.diag[6 * lane + a] * S.diag[6 * lane + a];
                    bvec[a] = S.rec[28 * lane + 21 + a] * S.scale[6 * lane + a];
                }
                ok = chol6(A, bvec);
#pragma unroll
                for (int a = 0; a < 6; ++a) ok = ok && isfinite(bvec[a]);
            }
            if (!__all(ok)) {
                mu *= 10.0;
                continue;
            }
            if (lane < W) {
#pragma unroll
                for (int a = 0; a < 6; ++a) S.gn[6 * lane + a] = bvec[a];
            }
            solve_ok = true;
            break;
        }
        if (lane == 0) S.mu = mu;
        WSYNC();
        if (solve_ok && in_n) S.gn[lane] *= -S.diag[lane];
        WSYNC();
    }
    bool step_valid = solve_ok;
    if (solve_ok) {
        const double alpha = S.alpha;
        double gr = 0, gnk = 0;
        if (in_n) {
            gr = S.grad[lane];
            gnk = S.gn[lane];
            U[lane] = gr * gr;
            U[64 + lane] = gnk * gnk;
            U[128 + lane] = gr * gnk;
        }
        WSYNC();
        {
            const double r = ordered_sums(U, n, 3);  // lane 0: |grad|^2, lane 1: |gn|^2, lane 2: grad . gn
            if (lane < 3) fl[lane] = r;
        }
        WSYNC();
        const double gradient_norm = sqrt(fl[0]), gn_norm = sqrt(fl[1]), gdot = fl[2];
        WSYNC();
        double st = 0;
        if (gn_norm <= radius) {
            st = gnk;
            if (lane == 0) S.dogleg_norm = gn_norm;
        } else if (gradient_norm * alpha >= radius) {
            st = -(radius / gradient_norm) * gr;
            if (lane == 0) S.dogleg_norm = radius;
        } else {
            const double b_dot_a = -alpha * gdot;
            const double a_sq = (alpha * gradient_norm) * (alpha * gradient_norm);
            const double bma_sq = a_sq - 2 * b_dot_a + gn_norm * gn_norm;
            const double c = b_dot_a - a_sq;
            const double d = sqrt(c * c + bma_sq * (radius * radius - a_sq));
            const double beta = (c <= 0) ? (d - c) / bma_sq : (radius * radius - a_sq) / (d + c);
            st = (-alpha * (1.0 - beta)) * gr + beta * gnk;
            if (in_n) U[lane] = st * st;
            WSYNC();
            const double sn = ordered_sums(U, n, 1);
            if (lane == 0) S.dogleg_norm = sqrt(sn);
            WSYNC();
        }
        if (in_n) {
            st /= S.diag[lane];
            S.step[lane] = st;
            U[lane] = st * gk * sk;
        }
        WSYNC();
        quad_terms(S, W, S.step, T);
        WSYNC();
        {
            const double sgd = ordered_sums(U, n, 1);
            const double q = quad_sum(T, W);
            const double mc = -(sgd + 0.5 * q);
            if (lane == 0) {
                S.model_change = mc;
                fl[3] = (mc > 0.0) ? 1.0 : 0.0;
            }
        }
        WSYNC();
        step_valid = fl[3] != 0.0;
        WSYNC();
    }
    if (!step_valid) {
        if (num_invalid + 1 >= 5) {  // HandleInvalidStep: FAILURE, parameters as on entry (see tr_propose)
            if (in_n) S.x[lane] = S.x_init[lane];
            if (lane == 0) {
                S.termination = 4;
                S.go = 0;
            }
        } else if (lane == 0) {
            S.mu *= 10.0;
            S.reuse = 0;
        }
        if (lane == 0) S.num_invalid = num_invalid + 1;
        return;  // go stays 1 unless failed, evaluate 0: the caller proposes again
    }
    if (in_n) {
        const double delta = S.step[lane] * sk;
        S.xc[lane] = S.x[lane] + delta;
        U[lane] = delta * delta;
    }
    WSYNC();
    {
        const double sn = ordered_sums(U, n, 1);
        if (lane == 0) {
            S.num_invalid = 0;
            S.step_norm = sqrt(sn);
            S.evaluate = 1;
        }
    }
}

__device__ __forceinline__ void tr_decide_wave(TRState& S, double* w, int W, int fixed, const int lane = threadIdx.x) {
    const int n = 6 * W;
    double* U = w + 36 * MAXW;
    double* fl = U + 3 * 64;
    WSYNC();
    if (lane < W) U[lane] = S.recc[28 * lane + 27];
    WSYNC();
    const double cand_l = ordered_sums(U, W, 1, lane);
    if (lane == 0) fl[4] = cand_l;
    WSYNC();
    const double cand = fl[4], cost = S.cost;
    if (!fixed) {
        if (S.step_norm <= 1e-8 * (S.x_norm + 1e-8)) {
            if (lane == 0) {
                S.termination = 2;
                S.go = 0;
            }
            return;
        }
        if (fabs(cost - cand) <= 1e-6 * cost) {
            if (lane == 0) {
                S.termination = 3;
                S.go = 0;
            }
            return;
        }
    }
    const double rel = (cost - cand) / S.model_change;
    WSYNC();
    if (rel > 1e-3) {
        double gabs = 0;
        if (lane < n) {
            const double xv = S.xc[lane];
            S.x[lane] = xv;
            U[lane] = xv * xv;
            gabs = fabs(S.recc[28 * (lane / 6) + 21 + (lane - 6 * (lane / 6))]);
        }
        for (int i = lane; i < 28 * W; i += 64) S.rec[i] = S.recc[i];
        WSYNC();
        const double xn = ordered_sums(U, n, 1, lane);
        double gm = gabs;  // a maximum does not depend on the order
        for (int o = 32; o > 0; o >>= 1) gm = fmax(gm, __shfl_xor(gm, o));
        if (lane == 0) {
            S.x_norm = sqrt(xn);
            S.cost = cand;
            S.successful++;
            if (!fixed && gm <= 1e-10) {
                S.termination = 1;
                S.go = 0;
            } else {
                if (rel < 0.25) S.radius *= 0.5;
                if (rel > 0.75) S.radius = fmax(S.radius, 3.0 * S.dogleg_norm);
                S.mu = fmax(1e-8, 2.0 * S.mu / 10.0);
                S.reuse = 0;
            }
        }
    } else if (lane == 0) {
        S.radius *= 0.5;
        S.reuse = 1;
    }
}
#undef WSYNC

// tr_decide for W = 1 on register copies (k_solve_wide; see tr_propose_w1_regs): every lane of the first wavefront forms the same
// values -- tr_decide's expressions in its order --, lane 0 stores them, the 28 values of the accepted record are copied by 28 lanes.
__device__ __forceinline__ void tr_decide_w1_regs(TRState& S, int fixed, const int lane) {
    double cand = 0;
    cand += S.recc[27];
    const double cost = S.cost;
    int zero = 0, one = 1;  // (formed here: see tr_propose_w1_regs)
    asm volatile("" : "+v"(zero), "+v"(one));
    if (!fixed) {
        if (S.step_norm <= 1e-8 * (S.x_norm + 1e-8)) {
            if (lane == 0) {
                S.termination = one + one;
                S.go = zero;
            }
            return;
        }
        if (fabs(cost - cand) <= 1e-6 * cost) {
            if (lane == 0) {
                S.termination = one + one + one;
                S.go = zero;
            }
            return;
        }
    }
    const double rel = (cost - cand) / S.model_change;
    if (rel > 1e-3) {
        double xc[6], xn = 0, gm = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            xc[i] = S.xc[i];
            xn += xc[i] * xc[i];
            gm = fmax(gm, fabs(S.recc[21 + i]));  // (a maximum does not depend on the order)
        }
        if (lane < 28) S.rec[lane] = S.recc[lane];
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) S.x[i] = xc[i];
            S.x_norm = sqrt(xn);
            S.cost = cand;
            S.successful++;
            if (!fixed && gm <= 1e-10) {
                S.termination = one;
                S.go = zero;
            } else {
                if (rel < 0.25) S.radius *= 0.5;
                if (rel > 0.75) S.radius = fmax(S.radius, 3.0 * S.dogleg_norm);
                S.mu = fmax(1e-8, 2.0 * S.mu / 10.0);
                S.reuse = zero;
            }
        }
    } else if (lane == 0) {
        S.radius *= 0.5;
        S.reuse = one;
    }
}

// (256 registers, two wavefronts per SIMD.  Held to 168 for three it spills 400 bytes per lane: 0.32 -> 0.93 ms per 1024 problems.)
// SMALL: the variant of launches of at most one problem per CU (the live path).  It evaluates two plane factors of a thread side by
// side (lidar_eval.h eval_frame_pairs: same sums in the same order, half the exposed latency of the factor passes; one-row plane
// factors only, i.e. plan_weight_tan = 0), and it is the one that takes its start poses from a packed parameter block (x_in) and
// leaves a result record per problem (result).  The batch variant carries none of that: a pointer test and 24 conditional stores
// in a kernel that sits on its 256-register limit cost it 10 % (0.253 -> 0.28 ms per 1024 problems) while they were in both.
template <bool SMALL>
__global__ __launch_bounds__(SOLVE_THREADS) void k_solve(SolveParams P) {
    __shared__ TRState S;
    __shared__ double s_part[SOLVE_WAVES * 28];
    __shared__ double s_work[92];
    __shared__ double s_wd[TRW_DOUBLES];
    const int prob = blockIdx.x;
    const int W = P.window;
    const int b0 = P.first + prob * W;
    const int tid = threadIdx.x;
#ifdef MML_SV_TIMING
    const bool sv_dbg = tid == 0 && prob == MML_SV_TIMING;
    unsigned long long sv_prev = clock64();
    if (sv_dbg) g_sv_dbg[7] += 1;
#endif
    if constexpr (SMALL) {
        if (tid < 6 * W) S.x[tid] = S.x_init[tid] = P.x_in ? P.x_in[(size_t)prob * 6 * W + tid] : P.x[(size_t)b0 * 6 + tid];
    } else {
        if (tid < 6 * W) S.x[tid] = S.x_init[tid] = P.x[(size_t)b0 * 6 + tid];
    }
    const bool pairs = SMALL && P.pairs && P.w_tan == 0.0;  // (uniform)
    __syncthreads();

    double acc[28];
    // initial evaluation
    for (int f = 0; f < W; ++f) {
        Pose pose;
        make_pose(S.x + 6 * f, P.Tbl, pose);
        const int b = b0 + f;
        if (SMALL && pairs)
            eval_frame_pairs(P.lf + (size_t)b * P.MF, P.ft_n[b], P.pf + (size_t)b * P.MF, P.ft_n[P.B + b], pose, P.huber, acc);
        else
            eval_frame(P.lf + (size_t)b * P.MF, P.ft_n[b], P.pf + (size_t)b * P.MF, P.ft_n[P.B + b], pose, P.w_tan, P.huber, acc);
        block_reduce28(acc, s_part, S.rec + 28 * f);
    }
    if (tid == 0) {
        S.cost = 0;
        double xn = 0;
        for (int f = 0; f < W; ++f) {
            S.cost += S.rec[28 * f + 27];
            for (int i = 0; i < 6; ++i) {
                S.scale[6 * f + i] = 1.0 / (1.0 + sqrt(Hget(S.rec + 28 * f, i, i)));
                xn += S.x[6 * f + i] * S.x[6 * f + i];
            }
        }
        S.x_norm = sqrt(xn);
        S.radius = 1e4;
        S.mu = 1e-8;
        S.reuse = 0;
        S.num_invalid = 0;
        S.iter = 0;
        S.successful = 0;
        S.termination = 0;
        S.go = 1;
        S.alpha = 0;
        S.dogleg_norm = 0;
        P.summ[8 * prob + 2] = S.cost;  // initial cost
        if (!P.fixed) {
            double gm = 0;
            for (int f = 0; f < W; ++f)
                for (int i = 0; i < 6; ++i) gm = fmax(gm, fabs(S.rec[28 * f + 21 + i]));
            if (gm <= 1e-10) {
                S.termination = 1;
                S.go = 0;
            }
        }
    }
    __syncthreads();

    int go = S.go;
    __syncthreads();
    SV_MARK(0);  // set-up + initial evaluation
    while (go) {
        // lane 0 owns the trust-region state between the barriers; every other lane only reads it after one
        // lane 0 (first wavefront for W = 1) owns the trust-region state between the barriers
        if (W == 1) {
#ifdef MML_BATCH_PROPOSE_REGS
            if (tid < 64) tr_propose_w1_regs(S, s_work, P.max_iters, tid);
#else
            if (tid < 64) tr_propose_w1_wave(S, s_work, P.max_iters);
#endif
        } else if (tid == 0) {
            tr_propose(S, W, P.max_iters);
        }
        __syncthreads();
        SV_MARK(1);  // trust-region proposal (first wavefront) + barrier
        go = S.go;
        const int ev = S.evaluate;
        if (!go) break;
        if (ev) {
            for (int f = 0; f < W; ++f) {
                Pose pose;
                make_pose(S.xc + 6 * f, P.Tbl, pose);
                const int b = b0 + f;
                if (SMALL && pairs)
                    eval_frame_pairs(P.lf + (size_t)b * P.MF, P.ft_n[b], P.pf + (size_t)b * P.MF, P.ft_n[P.B + b], pose, P.huber, acc);
                else
                    eval_frame(P.lf + (size_t)b * P.MF, P.ft_n[b], P.pf + (size_t)b * P.MF, P.ft_n[P.B + b], pose, P.w_tan, P.huber, acc);
                SV_MARK(2);  // factor pass of this thread
                block_reduce28(acc, s_part, S.recc + 28 * f);
                SV_MARK(3);  // block reduction (waits for the slowest wavefront's factor pass)
            }
            if (tid < 64) tr_decide_wave(S, s_wd, W, P.fixed);
        }
        __syncthreads();
        SV_MARK(4);  // accept / reject (first wavefront) + barrier
        go = S.go;
        if (P.trace && tid < 6 * W) P.trace[((size_t)prob * P.max_iters + (S.iter - 1)) * 6 * W + tid] = S.x[tid];
        __syncthreads();
    }
    if (tid < 6 * W) P.x[(size_t)b0 * 6 + tid] = S.x[tid];
    if (SMALL && P.result && W == 1) {  // (W = 1: the callers that ask for it)
        double* r = P.result + (size_t)prob * MML_SOLVE_RESULT;
        if (tid < 6) r[tid] = S.x[tid];
        if (tid == 6) r[6] = (double)P.ft_n[b0];
        if (tid == 7) r[7] = (double)P.ft_n[P.B + b0];
        if (tid >= 8 && tid < 24) r[tid] = P.stats ? P.stats[16 * (size_t)b0 + (tid - 8)] : 0.0;
    }
    // rows of the trace beyond the last iteration that ran repeat the final point
    if (P.trace && tid < 6 * W)
        for (int it = S.iter; it < P.max_iters; ++it) P.trace[((size_t)prob * P.max_iters + it) * 6 * W + tid] = S.x[tid];
    if (tid == 0) {
        double* o = P.summ + 8 * prob;
        o[0] = S.iter;
        o[1] = S.successful;
        o[3] = S.cost;
        o[4] = S.termination;
        o[5] = 0;  // (finished: see k_window_export)
    }
}

// k_solve<true> for one-frame problems with one-row plane factors (the live path's only setting), on all four SIMDs of the problem's
// CU: 4 x 128 threads, the factor passes through eval_frame_wide (lidar_eval.h: rows formed by 512 threads, sums and reduction tree of
// the 128-thread pass -- bit-identical), the trust-region code between the passes on the first wavefront as in k_solve.  123 KB of
// LDS for the rows of a round: one workgroup per CU, which is what these launches (at most one problem per CU) have anyway.
__global__ __launch_bounds__(WIDE_THREADS) void k_solve_wide(SolveParams P) {
    __shared__ TRState S;
    __shared__ double s_part[SOLVE_WAVES * 28];
    __shared__ double s_work[92];
#ifdef MML_WIDE_PROPOSE_WAVE
    __shared__ double s_wd[TRW_DOUBLES];
#endif
    __shared__ double s_rows[WIDE_ROW_LDS];
    __shared__ double s_vsave[WIDE_SAVE_LDS];
    const int prob = blockIdx.x;
    const int b0 = P.first + prob;
    const int tid = threadIdx.x;
#ifdef MML_SV_TIMING
    const unsigned long long svk_t0 = clock64();
    const bool sv_dbg = tid == 0 && prob == MML_SV_TIMING;
    unsigned long long sv_prev = svk_t0;
#endif
    if (tid < 6) S.x[tid] = S.x_init[tid] = P.x_in ? P.x_in[(size_t)prob * 6 + tid] : P.x[(size_t)b0 * 6 + tid];
    __syncthreads();
    const MmlLineFactor* lf = P.lf + (size_t)b0 * P.MF;
    const MmlPlaneFactor* pf = P.pf + (size_t)b0 * P.MF;
    const int nlf = P.ft_n[b0], npf = P.ft_n[P.B + b0];
    {
        Pose pose;
        make_pose(S.x, P.Tbl, pose);
        eval_frame_wide(lf, nlf, pf, npf, pose, P.huber, s_rows, s_vsave, s_part, S.rec);
    }
    if (tid == 0) {
        S.cost = 0;
        double xn = 0;
        S.cost += S.rec[27];
        for (int i = 0; i < 6; ++i) {
            S.scale[i] = 1.0 / (1.0 + sqrt(Hget(S.rec, i, i)));
            xn += S.x[i] * S.x[i];
        }
        S.x_norm = sqrt(xn);
        S.radius = 1e4;
        S.mu = 1e-8;
        S.reuse = 0;
        S.num_invalid = 0;
        S.iter = 0;
        S.successful = 0;
        S.termination = 0;
        S.go = 1;
        S.alpha = 0;
        S.dogleg_norm = 0;
        P.summ[8 * prob + 2] = S.cost;  // initial cost
        if (!P.fixed) {
            double gm = 0;
            for (int i = 0; i < 6; ++i) gm = fmax(gm, fabs(S.rec[21 + i]));
            if (gm <= 1e-10) {
                S.termination = 1;
                S.go = 0;
            }
        }
    }
    __syncthreads();
    int go = S.go;
    __syncthreads();
    while (go) {
        // (the lane number is made opaque to the compiler in every iteration: with it a known function of threadIdx.x the dozens of
        //  LDS addresses of the trust-region code are formed once in front of the loop and held -- across the factor pass, which needs
        //  the 256 registers two wavefronts per SIMD leave: they were spilled, and every reload is a memory round trip on the ONE
        //  wavefront everybody waits for: +35 k cycles per solve)
        int lane_op = tid;
        asm volatile("" : "+v"(lane_op));
        SV_MARK(0);  // (everything outside the two trust-region phases: set-up, factor passes, the loop's barriers)
#ifdef MML_WIDE_PROPOSE_WAVE
        if (tid < 64) tr_propose_w1_wave(S, s_work, P.max_iters, lane_op);
#else
        if (tid < 64) tr_propose_w1_regs(S, s_work, P.max_iters, lane_op);
#endif
        __syncthreads();
        SV_MARK(1);  // trust-region proposal (first wavefront) + barrier
        go = S.go;
        const int ev = S.evaluate;
        if (!go) break;
        if (ev) {
            Pose pose;
            make_pose(S.xc, P.Tbl, pose);
            eval_frame_wide(lf, nlf, pf, npf, pose, P.huber, s_rows, s_vsave, s_part, S.recc);
            asm volatile("" : "+v"(lane_op));
            SV_MARK(0);
#ifdef MML_WIDE_PROPOSE_WAVE
            if (tid < 64) tr_decide_wave(S, s_wd, 1, P.fixed, lane_op);
#else
            if (tid < 64) tr_decide_w1_regs(S, P.fixed, lane_op);
#endif
        }
        __syncthreads();
        SV_MARK(4);  // accept / reject (first wavefront) + barrier
        go = S.go;
        if (P.trace && tid < 6) P.trace[((size_t)prob * P.max_iters + (S.iter - 1)) * 6 + tid] = S.x[tid];
        __syncthreads();
    }
    if (tid < 6) P.x[(size_t)b0 * 6 + tid] = S.x[tid];
    if (P.result) {
        double* r = P.result + (size_t)prob * MML_SOLVE_RESULT;
        if (tid < 6) r[tid] = S.x[tid];
        if (tid == 6) r[6] = (double)nlf;
        if (tid == 7) r[7] = (double)npf;
        if (tid >= 8 && tid < 24) r[tid] = P.stats ? P.stats[16 * (size_t)b0 + (tid - 8)] : 0.0;
    }
    if (P.trace && tid < 6)
        for (int it = S.iter; it < P.max_iters; ++it) P.trace[((size_t)prob * P.max_iters + it) * 6 + tid] = S.x[tid];
    if (tid == 0) {
        double* o = P.summ + 8 * prob;
        o[0] = S.iter;
        o[1] = S.successful;
        o[3] = S.cost;
        o[4] = S.termination;
        o[5] = 0;  // (finished: see k_window_export)
#ifdef MML_SV_TIMING
        if (prob == MML_SV_TIMING) {
            g_svw_dbg[5] += clock64() - svk_t0;
            g_svw_dbg[6] += 1;
        }
#endif
    }
}

// linearisation of frame blockIdx.x of a window to its 32-double record (SURVEY 8(e) all-gather payload): slot b0 + f at
// x + 6 f -> record + 32 f
__global__ __launch_bounds__(SOLVE_THREADS) void k_linearize(int b0, int B, int MF, const int* ft_n,
                                                            const MmlLineFactor* lf, const MmlPlaneFactor* pf,
                                                            const double* x, const double* Tbl, double w_tan,
                                                            double huber_delta, const double* stats, double* record) {
    __shared__ double s_part[SOLVE_WAVES * 28];
    __shared__ double s_out[28];
    const int b = b0 + blockIdx.x;
    x += 6 * blockIdx.x;
    record += 32 * blockIdx.x;
    Pose pose;
    make_pose(x, Tbl, pose);
    double acc[28];
    eval_frame(lf + (size_t)b * MF, ft_n[b], pf + (size_t)b * MF, ft_n[B + b], pose, w_tan, huber_delta, acc);
    block_reduce28(acc, s_part, s_out);
    if (threadIdx.x < 28) record[threadIdx.x] = s_out[threadIdx.x];
    if (threadIdx.x == 0) {
        record[28] = stats[16 * b + 2];
        record[29] = stats[16 * b + 3];
        record[30] = 0;
        record[31] = 0;
    }
}

// ---- multi-GPU joint window solve: one ROUND of the device-resident state machine -----------------
// The window of W = n_ranks * n_local frames is solved redundantly by every rank; a rank evaluates only its own
// n_local frames (slots first .. first + n_local) and the 32-double records travel by ncclAllGather (comm.hip).
// One kernel per exchange: ADVANCE the trust-region state with the records gathered by the previous exchange
// (initialisation / tr_decide, then tr_propose until a candidate needs evaluating), then EVALUATE the own frames at that
// candidate straight into this rank's section of the gather buffer.  The state lives in global memory between the
// rounds and in LDS inside one; the host enqueues max_iterations + 1 rounds without ever reading a flag back (a
// finished state machine turns the remaining rounds into no-ops), so all ranks issue the same collectives.
struct WindowRoundParams {
    int first, n_local, rank, W, B, MF, max_iters, fixed, round, do_eval;
    double huber, w_tan;
    const int* ft_n;
    const MmlLineFactor* lf;
    const MmlPlaneFactor* pf;
    const double* Tbl;
    const double* x_all;   // W x 6: the gathered initial poses (round 0)
    const double* rec_in;  // W x 32: the records of the previous round's evaluation
    double* rec_out;       // W x 32: own records out (the same buffer as rec_in across GPUs, where every rank owns its buffer
                           // and the all-gather runs between the launches; the other half of a double buffer on one GPU)
    TRState* state;        // global copy of the state machine (one per workgroup)
    double* aux;           // per workgroup 4 doubles: [0] initial cost, [1] evaluations done, [2] rounds that did work
};

// On ONE GPU the same kernel solves windows frame-parallel (mml_solve with window > 1): grid = (W, problems), workgroup
// (f, p) carries its own copy of problem p's state machine -- the copies see identical records and therefore stay identical
// -- and evaluates frame f; the "exchange" is the kernel boundary, the records go through the two halves of a double buffer.
__global__ __launch_bounds__(SOLVE_THREADS) void k_window_round(WindowRoundParams P) {
    __shared__ TRState S;
    __shared__ double s_part[SOLVE_WAVES * 28];
    __shared__ double s_out[28];
    __shared__ double s_w[TRW_DOUBLES];
    const int tid = threadIdx.x, W = P.W;
    {
        const int prob = blockIdx.y, wg = prob * gridDim.x + blockIdx.x;
        P.rank += blockIdx.x;
        P.first += prob * W + blockIdx.x * P.n_local;
        P.x_all += 6 * W * prob;
        P.rec_in += 32 * W * prob;
        P.rec_out += 32 * W * prob;
        P.state += wg;
        P.aux += 4 * wg;
    }
    {
        const double* src = reinterpret_cast<const double*>(P.state);
        double* dst = reinterpret_cast<double*>(&S);
        for (int i = tid; i < (int)(sizeof(TRState) / sizeof(double)); i += SOLVE_THREADS) dst[i] = src[i];
    }
    __syncthreads();
    if (P.round >= 1 && S.go) {  // the records of the previous evaluation, fetched by all lanes at once
        double* dst = P.round == 1 ? S.rec : S.recc;
        for (int i = tid; i < 28 * W; i += SOLVE_THREADS) {
            const int f = i / 28, k = i - 28 * f;
            dst[i] = P.rec_in[32 * f + k];
        }
    }
    __syncthreads();
    if (tid < 64) {  // the first wavefront advances the state machine (tr_decide_wave / tr_propose_wave)
        const int lane = tid;
        const int was_go = S.go;
        __builtin_amdgcn_wave_barrier();
        if (P.round == 0) {
            if (lane < 6 * W) S.x[lane] = S.xc[lane] = S.x_init[lane] = P.x_all[lane];
            if (lane == 0) {
                S.go = 1;
                S.evaluate = 1;
                S.iter = 0;
                S.successful = 0;
                S.termination = 0;
                S.cost = 0;
            }
        } else if (was_go) {
            if (P.round == 1) {  // records at x0: the initialisation of k_solve / mml_window_solver_step
                if (lane < 6 * W) S.scale[lane] = 1.0 / (1.0 + sqrt(Hget(S.rec + 28 * (lane / 6), lane % 6, lane % 6)));
                if (lane == 0) {
                    double xn = 0, cost = 0;
                    for (int f = 0; f < W; ++f) {
                        cost += S.rec[28 * f + 27];
                        for (int i = 0; i < 6; ++i) xn += S.x[6 * f + i] * S.x[6 * f + i];
                    }
                    S.cost = cost;
                    S.x_norm = sqrt(xn);
                    S.radius = 1e4;
                    S.mu = 1e-8;
                    S.reuse = 0;
                    S.num_invalid = 0;
                    S.alpha = 0;
                    S.dogleg_norm = 0;
                    P.aux[0] = cost;
                    if (!P.fixed) {
                        double gm = 0;
                        for (int f = 0; f < W; ++f)
                            for (int i = 0; i < 6; ++i) gm = fmax(gm, fabs(S.rec[28 * f + 21 + i]));
                        if (gm <= 1e-10) {
                            S.termination = 1;
                            S.go = 0;
                        }
                    }
                }
            } else {
                tr_decide_wave(S, s_w, W, P.fixed);
            }
            for (;;) {
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
                if (!S.go) break;
                tr_propose_wave(S, s_w, W, P.max_iters);
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
                if (!S.go || S.evaluate) break;
            }
            if (lane == 0) P.aux[2] += 1.0;
        }
    }
    __syncthreads();
    const int go = S.go;
    if (P.do_eval && go) {
        double acc[28];
        for (int j = 0; j < P.n_local; ++j) {
            const int f = P.rank * P.n_local + j, b = P.first + j;
            Pose pose;
            make_pose(S.xc + 6 * f, P.Tbl, pose);
            eval_frame(P.lf + (size_t)b * P.MF, P.ft_n[b], P.pf + (size_t)b * P.MF, P.ft_n[P.B + b], pose, P.w_tan, P.huber, acc);
            block_reduce28(acc, s_part, s_out);
            if (tid < 32) P.rec_out[32 * f + tid] = tid < 28 ? s_out[tid] : 0.0;
            __syncthreads();
        }
        if (tid == 0) P.aux[1] += 1.0;
    }
    __syncthreads();
    {
        double* dst = reinterpret_cast<double*>(P.state);
        const double* src = reinterpret_cast<const double*>(&S);
        for (int i = tid; i < (int)(sizeof(TRState) / sizeof(double)); i += SOLVE_THREADS) dst[i] = src[i];
    }
}
static_assert(sizeof(TRState) % sizeof(double) == 0, "TRState is copied as doubles");

}  // namespace

size_t mml_window_state_bytes() { return sizeof(TRState); }

int mml_launch_window_round(mml_ctx* ctx, int first, int n_local, int rank, int W, const double* d_Tbl, mml_solve_opts opts,
                            int round, bool do_eval, const double* d_x_all, double* d_rec_all, void* d_state, double* d_aux) {
    WindowRoundParams P;
    P.first = first;
    P.n_local = n_local;
    P.rank = rank;
    P.W = W;
    P.B = ctx->B;
    P.MF = ctx->MF;
    P.max_iters = opts.max_num_iterations;
    P.fixed = opts.fixed_iterations;
    P.round = round;
    P.do_eval = do_eval ? 1 : 0;
    P.huber = opts.huber_delta;
    P.w_tan = opts.plan_weight_tan;
    P.ft_n = ctx->ft_n;
    P.lf = ctx->lf;
    P.pf = ctx->pf;
    P.Tbl = d_Tbl;
    P.x_all = d_x_all;
    P.rec_in = d_rec_all;
    P.rec_out = d_rec_all;
    P.state = reinterpret_cast<TRState*>(d_state);
    P.aux = d_aux;
    MmlStageScope t(ctx, "window_round");
    hipLaunchKernelGGL(k_window_round, dim3(1), dim3(SOLVE_THREADS), 0, MML_STREAM(ctx), P);
    MML_HIP(hipGetLastError());
    return MML_OK;
}

namespace {
// results of the frame-parallel window solve in the layout k_solve leaves them: poses and the 8-double summaries
__global__ void k_window_export(int first, int W, const TRState* state, const double* aux, double* x, double* summ) {
    const int prob = blockIdx.x, tid = threadIdx.x;
    const TRState& S = state[(size_t)prob * W];  // the copy of frame 0's workgroup (all copies are equal)
    if (tid < 6 * W) x[((size_t)first + (size_t)prob * W) * 6 + tid] = S.x[tid];
    if (tid == 0) {
        double* o = summ + 8 * prob;
        o[0] = S.iter;
        o[1] = S.successful;
        o[2] = aux[4 * (size_t)prob * W];
        o[3] = S.cost;
        o[4] = S.termination;
        o[5] = S.go;  // still running: the caller enqueues the remaining rounds (mml_window_solve_continue)
    }
}
}  // namespace

// mml_solve with window > 1 on one GPU: the W frames of every problem are evaluated by W workgroups at once instead of one
// after the other by a single workgroup (k_solve) -- the serial chain of an 8-frame window drops from 11 x 8 frame passes
// to 12 rounds of one.  Same functions, same records, same decisions: the poses equal k_solve's bit for bit.
// The rounds are enqueued in two chunks: [0, kFirstChunk) with the clean state in front -- enough for a solve that stops
// within five iterations -- and, only when the exported `go` flag of some problem is still up, the rest.
constexpr int kFirstChunk = 7;
static int launch_solve_frame_parallel(mml_ctx* ctx, int first, int count, int W, const double* d_Tbl, mml_solve_opts opts,
                                       bool second_chunk) {
    if (!ctx->wstate) {
        MML_HIP(hipMalloc(&ctx->wstate, sizeof(TRState) * (size_t)ctx->B));
        MML_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->wrec), sizeof(double) * 2 * 32 * (size_t)ctx->B));
        MML_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->waux), sizeof(double) * 4 * (size_t)ctx->B));
    }
    hipStream_t s = MML_STREAM(ctx);
    const int nprob = count / W;
    TRState* st = reinterpret_cast<TRState*>(ctx->wstate) + first;
    double* aux = ctx->waux + 4 * (size_t)first;
    double* rec[2] = {ctx->wrec + 32 * (size_t)first, ctx->wrec + 32 * ((size_t)ctx->B + first)};
    WindowRoundParams P;
    P.first = first;
    P.n_local = 1;
    P.rank = 0;
    P.W = W;
    P.B = ctx->B;
    P.MF = ctx->MF;
    P.max_iters = opts.max_num_iterations;
    P.fixed = opts.fixed_iterations;
    P.huber = opts.huber_delta;
    P.w_tan = opts.plan_weight_tan;
    P.ft_n = ctx->ft_n;
    P.lf = ctx->lf;
    P.pf = ctx->pf;
    P.Tbl = d_Tbl;
    P.x_all = ctx->d_x + 6 * (size_t)first;
    P.state = st;
    P.aux = aux;
    const int rounds = opts.max_num_iterations + 2;  // see mml_window_solve_allgather
    const int r_begin = second_chunk ? kFirstChunk : 0, r_end = second_chunk ? rounds : (rounds < kFirstChunk ? rounds : kFirstChunk);
    auto enqueue = [&]() -> hipError_t {
        if (!second_chunk) {
            hipError_t e = hipMemsetAsync(st, 0, sizeof(TRState) * (size_t)count, s);
            if (e != hipSuccess) return e;
            e = hipMemsetAsync(aux, 0, sizeof(double) * 4 * (size_t)count, s);
            if (e != hipSuccess) return e;
        }
        for (int r = r_begin; r < r_end; ++r) {
            P.round = r;
            P.do_eval = r + 1 < rounds ? 1 : 0;
            P.rec_in = rec[(r + 1) & 1];
            P.rec_out = rec[r & 1];
            hipLaunchKernelGGL(k_window_round, dim3(W, nprob), dim3(SOLVE_THREADS), 0, s, P);
        }
        hipLaunchKernelGGL(k_window_export, dim3(nprob), dim3(64), 0, s, first, W, st, aux, ctx->d_x, ctx->d_summ + 8 * (size_t)first);
        return hipGetLastError();
    };
    MmlStageScope t(ctx, "solve");
    // The chain is launch-bound (a dozen kernels of ~10 us that depend on each other): it is captured once per
    // (slot range, window, options, stream) into a HIP graph and replayed, which removes the per-launch gaps.  Every
    // pointer in it is a fixed function of the key.
    static const bool use_graph = getenv("MML_NO_GRAPH") == nullptr;
    if (use_graph) {
        mml_ctx::WinGraph* g = nullptr;
        for (auto& c : ctx->win_graphs)
            if (c.first == first && c.count == count && c.W == W && c.max_iters == opts.max_num_iterations &&
                c.fixed == opts.fixed_iterations && c.huber == opts.huber_delta && c.w_tan == opts.plan_weight_tan && c.stream == s &&
                c.Tbl == d_Tbl && c.second == second_chunk)
                g = &c;
        if (!g && hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            hipError_t e = enqueue();
            hipGraph_t graph = nullptr;
            hipError_t e2 = hipStreamEndCapture(s, &graph);
            hipGraphExec_t exec = nullptr;
            if (e == hipSuccess && e2 == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                if (ctx->win_graphs.size() >= 16) {  // small cache: drop the oldest
                    hipGraphExecDestroy(ctx->win_graphs.front().exec);
                    ctx->win_graphs.erase(ctx->win_graphs.begin());
                }
                mml_ctx::WinGraph c;
                c.first = first;
                c.count = count;
                c.W = W;
                c.max_iters = opts.max_num_iterations;
                c.fixed = opts.fixed_iterations;
                c.huber = opts.huber_delta;
                c.w_tan = opts.plan_weight_tan;
                c.stream = s;
                c.Tbl = d_Tbl;
                c.second = second_chunk;
                c.exec = exec;
                ctx->win_graphs.push_back(c);
                g = &ctx->win_graphs.back();
            }
            if (graph) hipGraphDestroy(graph);
            (void)hipGetLastError();
        }
        if (g) {
            MML_HIP(hipGraphLaunch(g->exec, s));
            return MML_OK;
        }
    }
    MML_HIP(enqueue());
    return MML_OK;
}

// the rounds after the first chunk, for the problems whose state machine has not stopped yet (a no-op for the others)
int mml_window_solve_continue(mml_ctx* ctx, int first, int count, int window, const double* d_Tbl, mml_solve_opts opts) {
    if (opts.max_num_iterations + 2 <= kFirstChunk) return MML_OK;
    return launch_solve_frame_parallel(ctx, first, count, window, d_Tbl, opts, true);
}

// final pose / summary of the window state machine (device TRState -> host)
int mml_window_state_read(mml_ctx* ctx, const void* d_state, int W, double* x_window, mml_solve_summary* summ, double initial_cost) {
    TRState* h = new TRState();
    hipError_t e = hipMemcpyAsync(h, d_state, sizeof(TRState), hipMemcpyDeviceToHost, MML_STREAM(ctx));
    if (e == hipSuccess) e = hipStreamSynchronize(MML_STREAM(ctx));
    if (e != hipSuccess) {
        delete h;
        ctx->err = std::string("window state read-back: ") + hipGetErrorString(e);
        return MML_ERR_HIP;
    }
    if (x_window)
        for (int i = 0; i < 6 * W; ++i) x_window[i] = h->x[i];
    if (summ) {
        summ->iterations = h->iter;
        summ->successful = h->successful;
        summ->initial_cost = initial_cost;
        summ->final_cost = h->cost;
        summ->termination = h->termination;
    }
    delete h;
    return MML_OK;
}

int mml_launch_solve(mml_ctx* ctx, int first, int count, int window, const double* d_Tbl, mml_solve_opts opts,
                     bool want_trace, const double* d_x_in, double* d_result) {
    MML_REQUIRE(window >= 1 && window <= MAXW && count % window == 0, MML_ERR_INVALID,
                "window must be in [1,8] and divide count");
    if (window > 1 && !want_trace && ctx->window_frame_parallel) return launch_solve_frame_parallel(ctx, first, count, window, d_Tbl, opts, false);
    SolveParams P;
    P.first = first;
    P.B = ctx->B;
    P.MF = ctx->MF;
    P.window = window;
    P.max_iters = opts.max_num_iterations;
    P.fixed = opts.fixed_iterations;
    P.huber = opts.huber_delta;
    P.w_tan = opts.plan_weight_tan;
    P.ft_n = ctx->ft_n;
    P.lf = ctx->lf;
    P.pf = ctx->pf;
    P.Tbl = d_Tbl;
    P.x = ctx->d_x;
    P.summ = ctx->d_summ + 8 * (size_t)first;
    P.trace = want_trace ? ctx->d_trace + (size_t)first * 6 * 64 : nullptr;
    P.x_in = d_x_in;
    P.result = d_result;
    P.stats = ctx->assoc_stats;
    MML_REQUIRE((!d_x_in && !d_result) || window == 1, MML_ERR_INVALID, "packed start poses / result records: one-frame problems only");
    MmlStageScope t(ctx, "solve");
    static int n_cus = -1, pairs_on = 1;
    if (n_cus < 0) {
        const char* e = getenv("MML_SOLVE_PAIRS");
        pairs_on = (e && atoi(e) == 0) ? 0 : 1;
        hipDeviceProp_t prop;
        n_cus = hipGetDeviceProperties(&prop, ctx->device) == hipSuccess ? prop.multiProcessorCount : 64;
    }
    const bool small = count / window <= n_cus;  // at most one problem per CU
    MML_REQUIRE((!d_x_in && !d_result) || small, MML_ERR_INVALID, "packed start poses / result records: small launches only");
    P.pairs = pairs_on;
    // one-frame problems with one-row plane factors (the live path): the 512-thread form ($MML_SOLVE_WIDE=0: measurement switch)
    static const bool wide_on = !(getenv("MML_SOLVE_WIDE") && atoi(getenv("MML_SOLVE_WIDE")) == 0);
    if (small && wide_on && window == 1 && opts.plan_weight_tan == 0.0)
        hipLaunchKernelGGL(k_solve_wide, dim3(count), dim3(WIDE_THREADS), 0, MML_STREAM(ctx), P);
    else if (small)
        hipLaunchKernelGGL(k_solve<true>, dim3(count / window), dim3(SOLVE_THREADS), 0, MML_STREAM(ctx), P);
    else
        hipLaunchKernelGGL(k_solve<false>, dim3(count / window), dim3(SOLVE_THREADS), 0, MML_STREAM(ctx), P);
    MML_HIP(hipGetLastError());
    return MML_OK;
}

int mml_launch_linearize(mml_ctx* ctx, int slot, const double* d_x, const double* d_Tbl, double w_tan, double huber,
                         double* d_record, int frames) {
    {  // (the records carry the used-factor counts of the association's statistics)
        const int rs = mml_ensure_assoc_stats(ctx, slot, frames);
        if (rs != MML_OK) return rs;
    }
    MmlStageScope t(ctx, "linearize");
    hipLaunchKernelGGL(k_linearize, dim3(frames), dim3(SOLVE_THREADS), 0, MML_STREAM(ctx), slot, ctx->B, ctx->MF, ctx->ft_n,
                       ctx->lf, ctx->pf, d_x, d_Tbl, w_tan, huber, ctx->assoc_stats, d_record);
    MML_HIP(hipGetLastError());
    return MML_OK;
}

// ---- host-side joint window solver: the same trust-region code, fed by all-gathered records ----
struct mml_window_solver {
    TRState S;
    int W;
    mml_solve_opts opts;
    int phase;  // 0: waiting for the records at x0, 1: waiting for the records at xc, 2: finished
    double initial_cost;
};

extern "C" mml_window_solver* mml_window_solver_create(int W, const mml_solve_opts* opts) {
    if (W < 1 || W > MAXW || !opts) return nullptr;
    mml_window_solver* s = new mml_window_solver();
    s->W = W;
    s->opts = *opts;
    s->phase = 0;
    s->initial_cost = 0;
    return s;
}
extern "C" void mml_window_solver_destroy(mml_window_solver* s) { delete s; }

extern "C" int mml_window_solver_step(mml_window_solver* s, const double* records, double* x_eval) {
    if (!s || !records || !x_eval) return MML_ERR_INVALID;
    TRState& S = s->S;
    const int W = s->W, n = 6 * W;
    if (s->phase == 2) {
        for (int i = 0; i < n; ++i) x_eval[i] = S.x[i];
        return 1;
    }
    if (s->phase == 0) {
        double xn = 0;
        S.cost = 0;
        for (int f = 0; f < W; ++f) {
            for (int k = 0; k < 28; ++k) S.rec[28 * f + k] = records[MML_NEQ_RECORD_DOUBLES * f + k];
            S.cost += S.rec[28 * f + 27];
            for (int i = 0; i < 6; ++i) {
                S.x[6 * f + i] = S.x_init[6 * f + i] = x_eval[6 * f + i];
                S.scale[6 * f + i] = 1.0 / (1.0 + sqrt(Hget(S.rec + 28 * f, i, i)));
                xn += S.x[6 * f + i] * S.x[6 * f + i];
            }
        }
        S.x_norm = sqrt(xn);
        S.radius = 1e4;
        S.mu = 1e-8;
        S.reuse = 0;
        S.num_invalid = 0;
        S.iter = 0;
        S.successful = 0;
        S.termination = 0;
        S.go = 1;
        S.alpha = 0;
        S.dogleg_norm = 0;
        s->initial_cost = S.cost;
        if (!s->opts.fixed_iterations) {
            double gm = 0;
            for (int f = 0; f < W; ++f)
                for (int i = 0; i < 6; ++i) gm = fmax(gm, fabs(S.rec[28 * f + 21 + i]));
            if (gm <= 1e-10) {
                S.termination = 1;
                S.go = 0;
            }
        }
        s->phase = 1;
    } else {
        for (int f = 0; f < W; ++f)
            for (int k = 0; k < 28; ++k) S.recc[28 * f + k] = records[MML_NEQ_RECORD_DOUBLES * f + k];
        tr_decide(S, W, s->opts.fixed_iterations);
    }
    while (S.go) {
        tr_propose(S, W, s->opts.max_num_iterations);
        if (!S.go) break;
        if (S.evaluate) {
            for (int i = 0; i < n; ++i) x_eval[i] = S.xc[i];
            return 0;
        }
    }
    s->phase = 2;
    for (int i = 0; i < n; ++i) x_eval[i] = S.x[i];
    return 1;
}

extern "C" int mml_window_solver_summary(const mml_window_solver* s, mml_solve_summary* out) {
    if (!s || !out) return MML_ERR_INVALID;
    out->iterations = s->S.iter;
    out->successful = s->S.successful;
    out->initial_cost = s->initial_cost;
    out->final_cost = s->S.cost;
    out->termination = s->S.termination;
    return MML_OK;
}
