// window_imu.hip -- SURVEY.md section 8(f) rank 1: the IMU side of the joint window solve.  Host code (W <= 8 frames,
// 15 parameters each: O(W) tiny dense work next to the per-frame lidar normal equations that come from the device),
// compiled into the same library so that it sits behind the same C-ABI; the same trust-region loop resident on the
// device is fullwindow_dev.hip, the shared arithmetic imu_math.h.
//   * IMUIntegrator::PreIntegration            mm-loam/src/lio/IMUIntegrator.cpp:108-166  -> mml_imu_preintegrate
//   * Cost_NavState_PRV_Bias (15 residuals)    mm-loam/include/utils/ceresfunc.h:321-393   -> mml_imu_factor
//     Jacobians: the reference lets Ceres autodiff the functor; here they are analytic (checked against central
//     differences of the residual in tests/test_imu.py)
//   * MarginalizationInfo / MarginalizationFactor  ceresfunc.h:92-317, hookup Estimator.cpp:1448-1567
//                                                                                          -> mml_fullwindow_marginalize
//   * the full-window problem of Estimator::Estimate (Estimator.cpp:1226-1254,1425-1432): lidar blocks on para_PR[f],
//     IMU factors between consecutive frames on (para_PR, para_VBias), the marginalization prior on frame 0, solved
//     with the same trust-region dogleg iteration as mml_solve but on the dense 15 W system     -> mml_fullwindow_*
// Rotation blocks are rotation VECTORS in a plain Euclidean parameter block (Estimator.cpp:1228), R = Sophus exp.
#include <math.h>
#include <string.h>

#include <vector>

#include "fullwindow_internal.h"
#include "imu_math.h"

namespace {

// in-place Cholesky A = L L^T (lower, row-major n x n); false when not positive definite
bool cholesky(double* A, int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0) || !isfinite(d)) return false;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return true;
}
void chol_solve(const double* L, int n, double* b) {
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * b[k];
        b[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * b[k];
        b[i] = s / L[i * n + i];
    }
}
// The same with the reciprocals of the diagonal taken once and multiplied in: the form the device-resident solver
// (fullwindow_dev.hip) uses, where a division would sit on the dependency chain of every substitution step.  The terms of a
// row are subtracted in the order a column-by-column (right-looking) substitution meets them -- ascending k forwards,
// DESCENDING k backwards, "the columns become known from the last one down" -- which is the device's order: the two
// iterations then hold bit-identical steps.
void chol_solve_rcp(const double* L, int n, double* b) {
    std::vector<double> rd(n);
    for (int i = 0; i < n; ++i) rd[i] = 1.0 / L[i * n + i];
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * b[k];
        b[i] = s * rd[i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = n - 1; k > i; --k) s -= L[k * n + i] * b[k];
        b[i] = s * rd[i];
    }
}
// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix: A = V diag(ev) V^T, eigenvalues ascending
void sym_eig(const double* Ain, int n, double* ev, double* V) {
    std::vector<double> A(Ain, Ain + n * n);
    for (int i = 0; i < n * n; ++i) V[i] = 0;
    for (int i = 0; i < n; ++i) V[i * n + i] = 1;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) (i == j ? diag : off) += A[i * n + j] * A[i * n + j];
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (A[order[j] * n + order[j]] < A[order[i] * n + order[i]]) {
                int t = order[i];
                order[i] = order[j];
                order[j] = t;
            }
    std::vector<double> Vs(n * n);
    for (int c = 0; c < n; ++c) {
        ev[c] = A[order[c] * n + order[c]];
        for (int r = 0; r < n; ++r) Vs[r * n + c] = V[r * n + order[c]];
    }
    memcpy(V, Vs.data(), sizeof(double) * n * n);
}

constexpr double kGnorm = 9.805;                                        // IMUIntegrator.h:84
constexpr double kAccN = 0.08, kGyrN = 0.004, kAccW = 2.0e-4, kGyrW = 2.0e-5;  // IMUIntegrator.h:79-82

// sqrt information of one pre-integration: LLT(covariance^-1).matrixL().transpose() (Estimator.cpp:1240-1242)
bool imu_sqrt_info(const mml_imu_preint* pre, double* U /*15x15 upper*/) {
    double L[225];
    memcpy(L, pre->covariance, sizeof(L));
    if (!cholesky(L, 15)) return false;
    double inv[225];
    for (int c = 0; c < 15; ++c) {
        double e[15] = {0};
        e[c] = 1.0;
        chol_solve(L, 15, e);
        for (int r = 0; r < 15; ++r) inv[r * 15 + c] = e[r];
    }
    for (int r = 0; r < 15; ++r)
        for (int c = r + 1; c < 15; ++c) inv[r * 15 + c] = inv[c * 15 + r] = 0.5 * (inv[r * 15 + c] + inv[c * 15 + r]);
    if (!cholesky(inv, 15)) return false;
    for (int r = 0; r < 15; ++r)
        for (int c = 0; c < 15; ++c) U[r * 15 + c] = (c >= r) ? inv[c * 15 + r] : 0.0;  // U = L^T
    return true;
}

}  // namespace

bool mml_imu_sqrt_info(const mml_imu_preint* pre, double* U) { return imu_sqrt_info(pre, U); }

extern "C" {

int mml_imu_preintegrate(const double* samples, int n, const double* bg, const double* ba, mml_imu_preint* out) {
    if (!out || n < 0 || (n > 0 && !samples) || !bg || !ba) return MML_ERR_INVALID;
    // Reset() (:49-58)
    memset(out, 0, sizeof(*out));
    out->dq[3] = 1.0;
    for (int i = 0; i < 15; ++i) out->jacobian[i * 15 + i] = 1.0;
    for (int k = 0; k < 3; ++k) {
        out->bg[k] = bg[k];
        out->ba[k] = ba[k];
    }
    double noise[12] = {kGyrN * kGyrN, kGyrN* kGyrN, kGyrN* kGyrN, kAccN* kAccN, kAccN* kAccN, kAccN* kAccN,
                        kGyrW* kGyrW, kGyrW* kGyrW, kGyrW* kGyrW, kAccW* kAccW, kAccW* kAccW, kAccW* kAccW};  // :33-37 (diagonal)
    std::vector<double> A(225), B(15 * 12), T(225), T2(225);
    for (int s = 0; s < n; ++s) {
        const double* m = samples + 7 * s;
        double gyr[3] = {m[0] - bg[0], m[1] - bg[1], m[2] - bg[2]};
        double acc[3] = {m[3] * kGnorm - ba[0], m[4] * kGnorm - ba[1], m[5] * kGnorm - ba[2]};
        const double dt = m[6], dt2 = dt * dt;
        double gdt[3] = {gyr[0] * dt, gyr[1] * dt, gyr[2] * dt};
        const M3 dR = so3_exp(gdt);
        M3 Jr = m3_identity();
        const double nrm = sqrt((gdt[0] * gdt[0] + gdt[1] * gdt[1]) + gdt[2] * gdt[2]);
        if (nrm > 0.00001) {  // :129-135
            double k[3] = {gdt[0] / nrm, gdt[1] / nrm, gdt[2] / nrm};
            const M3 K = hat(k);
            Jr = m3_add(m3_add(m3_identity(), m3_scale(K, -(1 - cos(nrm)) / nrm)), m3_scale(m3_mul(K, K), 1 - sin(nrm) / nrm));
        }
        const M3 Rq = quat_to_m3(out->dq);
        const M3 RqA = m3_mul(Rq, hat(acc));
        for (int i = 0; i < 225; ++i) A[i] = 0;
        for (int i = 0; i < 15; ++i) A[i * 15 + i] = 1.0;
        set_block(A.data(), 15, 0, 3, RqA, -0.5 * dt2);
        set_block(A.data(), 15, 0, 6, m3_identity(), dt);
        set_block(A.data(), 15, 0, 12, Rq, -0.5 * dt2);
        set_block(A.data(), 15, 3, 3, m3_t(dR));
        set_block(A.data(), 15, 3, 9, Jr, -dt);
        set_block(A.data(), 15, 6, 3, RqA, -dt);
        set_block(A.data(), 15, 6, 12, Rq, -dt);
        for (int i = 0; i < 15 * 12; ++i) B[i] = 0;
        set_block(B.data(), 12, 0, 3, Rq, 0.5 * dt2);
        set_block(B.data(), 12, 3, 0, Jr, dt);
        set_block(B.data(), 12, 6, 3, Rq, dt);
        set_block(B.data(), 12, 9, 6, m3_identity(), dt);
        set_block(B.data(), 12, 12, 9, m3_identity(), dt);
        // jacobian = A * jacobian
        for (int r = 0; r < 15; ++r)
            for (int c = 0; c < 15; ++c) {
                double acc_ = 0;
                for (int k = 0; k < 15; ++k) acc_ += A[r * 15 + k] * out->jacobian[k * 15 + c];
                T[r * 15 + c] = acc_;
            }
        memcpy(out->jacobian, T.data(), sizeof(double) * 225);
        // covariance = A * covariance * A^T + B * noise * B^T
        for (int r = 0; r < 15; ++r)
            for (int c = 0; c < 15; ++c) {
                double acc_ = 0;
                for (int k = 0; k < 15; ++k) acc_ += A[r * 15 + k] * out->covariance[k * 15 + c];
                T[r * 15 + c] = acc_;
            }
        for (int r = 0; r < 15; ++r)
            for (int c = 0; c < 15; ++c) {
                double acc_ = 0;
                for (int k = 0; k < 15; ++k) acc_ += T[r * 15 + k] * A[c * 15 + k];
                double bn = 0;
                for (int k = 0; k < 12; ++k) bn += B[r * 12 + k] * noise[k] * B[c * 12 + k];
                T2[r * 15 + c] = acc_ + bn;
            }
        memcpy(out->covariance, T2.data(), sizeof(double) * 225);
        // dp += dv*dt + 0.5*dq*acc*dt2 ; dv += dq*acc*dt   (:159-160)
        double Ra[3];
        m3_vec(Rq, acc, Ra);
        for (int k = 0; k < 3; ++k) out->dp[k] += out->dv[k] * dt + 0.5 * Ra[k] * dt2;
        for (int k = 0; k < 3; ++k) out->dv[k] += Ra[k] * dt;
        // dq = normalized(Quaterniond(dq.matrix() * dR)), w >= 0  (:161-165)
        double q[4];
        m3_to_quat(m3_mul(Rq, dR), q);
        if (q[3] < 0)
            for (int k = 0; k < 4; ++k) q[k] = -q[k];
        const double nq = sqrt((q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3]));
        for (int k = 0; k < 4; ++k) out->dq[k] = q[k] / nq;
        out->dtime += dt;
    }
    return MML_OK;
}

static void imu_factor_with_U(const mml_imu_preint* pre, const double* U, const double* gravity, const double* pri, const double* vbi,
                              const double* prj, const double* vbj, double* residual, double* jacobian);

int mml_imu_factor(const mml_imu_preint* pre, const double* gravity, const double* pri, const double* vbi, const double* prj,
                   const double* vbj, double* residual, double* jacobian) {
    if (!pre || !gravity || !pri || !vbi || !prj || !vbj || !residual) return MML_ERR_INVALID;
    double U[225];
    if (!imu_sqrt_info(pre, U)) return MML_ERR_STATE;
    imu_factor_with_U(pre, U, gravity, pri, vbi, prj, vbj, residual, jacobian);
    return MML_OK;
}

// the factor for a sqrt information that is already known (the full-window solver factors each covariance once, when the
// pre-integration is handed over, instead of once per evaluation)
static void imu_factor_with_U(const mml_imu_preint* pre, const double* U, const double* gravity, const double* pri, const double* vbi,
                              const double* prj, const double* vbj, double* residual, double* jacobian) {
    double r[15], J[15 * 30];
    imu_raw(pre, gravity, pri, vbi, prj, vbj, r, jacobian ? J : nullptr);
    for (int i = 0; i < 15; ++i) {  // eResiduals.applyOnTheLeft(sqrt_information)
        double s = 0;
        for (int k = i; k < 15; ++k) s += U[i * 15 + k] * r[k];
        residual[i] = s;
    }
    if (jacobian)
        for (int i = 0; i < 15; ++i)
            for (int c = 0; c < 30; ++c) {
                double s = 0;
                for (int k = i; k < 15; ++k) s += U[i * 15 + k] * J[k * 30 + c];
                jacobian[i * 30 + c] = s;
            }
}

}  // extern "C"

// ---- the full-window problem -------------------------------------------------------------------------------------------
namespace {

// adds the IMU factors and the prior, evaluated at x (W x 15: PR 6 | VBias 9), to the lidar records (W x 32)
int assemble(const mml_fullwindow* s, const double* records, const double* x, MmlFwEval& e) {
    const int W = s->W, n = s->n;
    e.H.assign((size_t)n * n, 0.0);
    e.g.assign(n, 0.0);
    e.cost = 0;
    for (int f = 0; f < W; ++f) {
        const double* rec = records + 32 * f;
        int k = 0;
        for (int a = 0; a < 6; ++a)
            for (int b = a; b < 6; ++b) {
                const double v = rec[k++];
                e.H[(size_t)(15 * f + a) * n + 15 * f + b] += v;
                if (b != a) e.H[(size_t)(15 * f + b) * n + 15 * f + a] += v;
            }
        for (int a = 0; a < 6; ++a) e.g[15 * f + a] += rec[21 + a];
        e.cost += rec[27];
    }
    for (int f = 1; f < W; ++f) {
        if (!s->have_imu[f]) continue;
        double r[15], J[15 * 30];
        const double* xi = x + 15 * (f - 1);
        const double* xj = x + 15 * f;
        if (!s->U_ok[f]) return MML_ERR_STATE;
        imu_factor_with_U(&s->imu[f], &s->U[225 * (size_t)f], s->gravity, xi, xi + 6, xj, xj + 6, r, J);
        const int base = 15 * (f - 1);  // the 30 columns are exactly the two consecutive frames
        for (int a = 0; a < 30; ++a) {
            double ga = 0;
            for (int i = 0; i < 15; ++i) ga += J[i * 30 + a] * r[i];
            e.g[base + a] += ga;
            for (int b = 0; b < 30; ++b) {
                double h = 0;
                for (int i = 0; i < 15; ++i) h += J[i * 30 + a] * J[i * 30 + b];
                e.H[(size_t)(base + a) * n + base + b] += h;
            }
        }
        double c = 0;
        for (int i = 0; i < 15; ++i) c += r[i] * r[i];
        e.cost += 0.5 * c;
    }
    if (s->prior.valid) {
        double r[15];
        prior_residual(s->prior, x, r);
        for (int a = 0; a < 15; ++a) {
            double ga = 0;
            for (int i = 0; i < 15; ++i) ga += s->prior.J[i * 15 + a] * r[i];
            e.g[a] += ga;
            for (int b = 0; b < 15; ++b) {
                double h = 0;
                for (int i = 0; i < 15; ++i) h += s->prior.J[i * 15 + a] * s->prior.J[i * 15 + b];
                e.H[(size_t)a * n + b] += h;
            }
        }
        double c = 0;
        for (int i = 0; i < 15; ++i) c += r[i] * r[i];
        e.cost += 0.5 * c;
    }
    return MML_OK;
}

double quad(const mml_fullwindow* s, const std::vector<double>& v) {  // v^T (S H S) v
    const int n = s->n;
    double q = 0;
    for (int a = 0; a < n; ++a) {
        double row = 0;
        for (int b = 0; b < n; ++b) row += s->cur.H[(size_t)a * n + b] * s->scale[b] * v[b];
        q += v[a] * s->scale[a] * row;
    }
    return q;
}

// one proposal: returns 1 when s->xc holds a candidate to evaluate, 0 when the step was invalid (propose again),
// -1 when the minimiser has stopped
int propose(mml_fullwindow* s) {
    const int n = s->n;
    if (s->iter >= s->opts.max_num_iterations || s->radius < 1e-32) return -1;
    s->iter++;
    bool solve_ok = true;
    if (!s->reuse) {
        s->reuse = 1;
        for (int i = 0; i < n; ++i) {
            double d = s->cur.H[(size_t)i * n + i] * s->scale[i] * s->scale[i];
            d = fmin(fmax(d, 1e-6), 1e32);
            s->diag[i] = sqrt(d);
        }
        double gg = 0;
        std::vector<double> sg(n);
        for (int i = 0; i < n; ++i) {
            s->grad[i] = s->cur.g[i] * s->scale[i] / s->diag[i];
            sg[i] = s->grad[i] / s->diag[i];
            gg += s->grad[i] * s->grad[i];
        }
        s->alpha = gg / quad(s, sg);
        solve_ok = false;
        std::vector<double> A((size_t)n * n), b(n);
        while (s->mu < 1.0) {
            for (int a = 0; a < n; ++a) {
                for (int c = 0; c < n; ++c) A[(size_t)a * n + c] = s->cur.H[(size_t)a * n + c] * s->scale[a] * s->scale[c];
                A[(size_t)a * n + a] += s->mu * s->diag[a] * s->diag[a];
                b[a] = s->cur.g[a] * s->scale[a];
            }
            bool ok = cholesky(A.data(), n);
            if (ok) {
                chol_solve_rcp(A.data(), n, b.data());
                for (int a = 0; a < n; ++a)
                    if (!isfinite(b[a])) ok = false;
            }
            if (!ok) {
                s->mu *= 10.0;
                continue;
            }
            for (int a = 0; a < n; ++a) s->gn[a] = -s->diag[a] * b[a];
            solve_ok = true;
            break;
        }
    }
    bool step_valid = solve_ok;
    if (solve_ok) {
        double gradient_norm = 0, gn_norm = 0;
        for (int i = 0; i < n; ++i) {
            gradient_norm += s->grad[i] * s->grad[i];
            gn_norm += s->gn[i] * s->gn[i];
        }
        gradient_norm = sqrt(gradient_norm);
        gn_norm = sqrt(gn_norm);
        if (gn_norm <= s->radius) {
            for (int i = 0; i < n; ++i) s->step[i] = s->gn[i];
            s->dogleg_norm = gn_norm;
        } else if (gradient_norm * s->alpha >= s->radius) {
            for (int i = 0; i < n; ++i) s->step[i] = -(s->radius / gradient_norm) * s->grad[i];
            s->dogleg_norm = s->radius;
        } else {
            double gdot = 0;
            for (int i = 0; i < n; ++i) gdot += s->grad[i] * s->gn[i];
            const double b_dot_a = -s->alpha * gdot;
            const double a_sq = (s->alpha * gradient_norm) * (s->alpha * gradient_norm);
            const double bma_sq = a_sq - 2 * b_dot_a + gn_norm * gn_norm;
            const double c = b_dot_a - a_sq;
            const double d = sqrt(c * c + bma_sq * (s->radius * s->radius - a_sq));
            const double beta = (c <= 0) ? (d - c) / bma_sq : (s->radius * s->radius - a_sq) / (d + c);
            double sn = 0;
            for (int i = 0; i < n; ++i) {
                s->step[i] = (-s->alpha * (1.0 - beta)) * s->grad[i] + beta * s->gn[i];
                sn += s->step[i] * s->step[i];
            }
            s->dogleg_norm = sqrt(sn);
        }
        for (int i = 0; i < n; ++i) s->step[i] /= s->diag[i];
        double sgd = 0;
        for (int i = 0; i < n; ++i) sgd += s->step[i] * s->cur.g[i] * s->scale[i];
        s->model_change = -(sgd + 0.5 * quad(s, s->step));
        if (!(s->model_change > 0.0)) step_valid = false;
    }
    if (!step_valid) {
        if (++s->num_invalid >= 5) {  // HandleInvalidStep: FAILURE, Solver::Solve restores the parameters it was given
            s->x = s->x_init;
            s->termination = 4;
            return -1;
        }
        s->mu *= 10.0;
        s->reuse = 0;
        return 0;
    }
    s->num_invalid = 0;
    double sn = 0;
    for (int i = 0; i < n; ++i) {
        const double delta = s->step[i] * s->scale[i];
        s->xc[i] = s->x[i] + delta;
        sn += delta * delta;
    }
    s->step_norm = sqrt(sn);
    return 1;
}

// after the candidate has been evaluated into s->cand: accept / reject; returns true when the minimiser stops
bool decide(mml_fullwindow* s) {
    const int n = s->n;
    const bool fixed = s->opts.fixed_iterations != 0;
    const double cand = s->cand.cost;
    if (!fixed) {
        if (s->step_norm <= 1e-8 * (s->x_norm + 1e-8)) {
            s->termination = 2;
            return true;
        }
        if (fabs(s->cur.cost - cand) <= 1e-6 * s->cur.cost) {
            s->termination = 3;
            return true;
        }
    }
    const double rel = (s->cur.cost - cand) / s->model_change;
    if (rel > 1e-3) {
        double xn = 0;
        for (int i = 0; i < n; ++i) {
            s->x[i] = s->xc[i];
            xn += s->x[i] * s->x[i];
        }
        s->x_norm = sqrt(xn);
        s->cur = s->cand;
        s->successful++;
        if (!fixed) {
            double gm = 0;
            for (int i = 0; i < n; ++i) gm = fmax(gm, fabs(s->cur.g[i]));
            if (gm <= 1e-10) {
                s->termination = 1;
                return true;
            }
        }
        if (rel < 0.25) s->radius *= 0.5;
        if (rel > 0.75) s->radius = fmax(s->radius, 3.0 * s->dogleg_norm);
        s->mu = fmax(1e-8, 2.0 * s->mu / 10.0);
        s->reuse = 0;
    } else {
        s->radius *= 0.5;
        s->reuse = 1;
    }
    return false;
}

}  // namespace

extern "C" {

mml_fullwindow* mml_fullwindow_create(int W, const mml_solve_opts* opts) {
    if (W < 1 || W > 8 || !opts) return nullptr;
    mml_fullwindow* s = new mml_fullwindow();
    s->W = W;
    s->n = 15 * W;
    s->opts = *opts;
    s->imu.resize(W);
    s->have_imu.assign(W, 0);
    s->U.assign(225 * (size_t)W, 0.0);
    s->U_ok.assign(W, 0);
    const int n = s->n;
    s->x.assign(n, 0);
    s->xc.assign(n, 0);
    s->scale.assign(n, 1);
    s->diag.assign(n, 1);
    s->grad.assign(n, 0);
    s->gn.assign(n, 0);
    s->step.assign(n, 0);
    return s;
}

void mml_fullwindow_destroy(mml_fullwindow* s) { delete s; }

int mml_fullwindow_set_imu(mml_fullwindow* s, int f, const mml_imu_preint* pre, const double* gravity) {
    if (!s || !pre || !gravity || f < 1 || f >= s->W) return MML_ERR_INVALID;
    s->imu[f] = *pre;
    s->have_imu[f] = 1;
    s->U_ok[f] = imu_sqrt_info(pre, &s->U[225 * (size_t)f]) ? 1 : 0;  // (a covariance that is not positive definite is reported by the solve)
    for (int k = 0; k < 3; ++k) s->gravity[k] = gravity[k];
    return MML_OK;
}

int mml_fullwindow_set_prior(mml_fullwindow* s, const mml_prior* p) {
    if (!s) return MML_ERR_INVALID;
    if (!p) {
        s->prior.valid = false;
        return MML_OK;
    }
    s->prior.valid = true;
    memcpy(s->prior.J, p->J, sizeof(s->prior.J));
    memcpy(s->prior.r0, p->r0, sizeof(s->prior.r0));
    memcpy(s->prior.x0, p->x0, sizeof(s->prior.x0));
    return MML_OK;
}

// Protocol of mml_window_solver_step: `records` are the lidar normal equations (W x 32, from mml_linearize_record /
// the all-gather) evaluated at x_eval (W x 15).  Returns 1 when finished (x_eval holds the solution), 0 when x_eval
// holds the next point to evaluate, < 0 on error.
int mml_fullwindow_step(mml_fullwindow* s, const double* records, double* x_eval) {
    if (!s || !records || !x_eval) return MML_ERR_INVALID;
    const int n = s->n;
    if (s->done) {
        memcpy(x_eval, s->x.data(), sizeof(double) * n);
        return 1;
    }
    if (!s->started) {
        s->started = 1;
        memcpy(s->x.data(), x_eval, sizeof(double) * n);
        s->x_init = s->x;
        int rc = assemble(s, records, s->x.data(), s->cur);
        if (rc != MML_OK) return rc;
        s->initial_cost = s->cur.cost;
        double xn = 0;
        for (int i = 0; i < n; ++i) xn += s->x[i] * s->x[i];
        s->x_norm = sqrt(xn);
        for (int i = 0; i < n; ++i) s->scale[i] = 1.0 / (1.0 + sqrt(s->cur.H[(size_t)i * n + i]));  // Jacobi scaling
        if (!s->opts.fixed_iterations) {
            double gm = 0;
            for (int i = 0; i < n; ++i) gm = fmax(gm, fabs(s->cur.g[i]));
            if (gm <= 1e-10) {
                s->termination = 1;
                s->done = 1;
                return 1;
            }
        }
    } else {
        int rc = assemble(s, records, s->xc.data(), s->cand);
        if (rc != MML_OK) return rc;
        if (decide(s)) {
            s->done = 1;
            memcpy(x_eval, s->x.data(), sizeof(double) * n);
            return 1;
        }
    }
    for (;;) {
        const int p = propose(s);
        if (p < 0) {
            s->done = 1;
            memcpy(x_eval, s->x.data(), sizeof(double) * n);
            return 1;
        }
        if (p == 1) break;
    }
    memcpy(x_eval, s->xc.data(), sizeof(double) * n);
    return 0;
}

// The dense normal equations Ceres would assemble at x: H = sum J^T J (n x n row-major, n = 15 W), g = sum J^T r,
// cost = 1/2 sum r^2 over the lidar blocks (from `records`), the IMU factors and the prior.
int mml_fullwindow_normal_equations(const mml_fullwindow* s, const double* records, const double* x, double* H, double* g,
                                    double* cost) {
    if (!s || !records || !x || !H || !g || !cost) return MML_ERR_INVALID;
    MmlFwEval e;
    int rc = assemble(s, records, x, e);
    if (rc != MML_OK) return rc;
    memcpy(H, e.H.data(), sizeof(double) * e.H.size());
    memcpy(g, e.g.data(), sizeof(double) * e.g.size());
    *cost = e.cost;
    return MML_OK;
}

int mml_fullwindow_summary(const mml_fullwindow* s, mml_solve_summary* out) {
    if (!s || !out) return MML_ERR_INVALID;
    out->iterations = s->iter;
    out->successful = s->successful;
    out->initial_cost = s->initial_cost;
    out->final_cost = s->cur.cost;
    out->termination = s->termination;
    return MML_OK;
}

// MarginalizationInfo::preMarginalize + marginalize (ceresfunc.h:132-228) for the factor set of Estimator.cpp:1453-1546:
// the previous prior (all of its blocks are dropped: it lives on frame 0), the IMU factor between frames 0 and 1, and
// the lidar factors of frame 0 given as their normal equations `lidar_record0` (32 doubles, evaluated WITHOUT a loss
// function at x[0..6)).  x: W x 15 current values.  Result: the prior on (PR, VBias) of frame 1, i.e. of frame 0 once
// the window has slid (:1552-1562).
int mml_fullwindow_marginalize(const mml_fullwindow* s, const double* lidar_record0, const double* x, mml_prior* out) {
    if (!s || !lidar_record0 || !x || !out || s->W < 2 || !s->have_imu[1]) return MML_ERR_INVALID;
    const int m = 15, n = 15, N = 30;
    std::vector<double> A((size_t)N * N, 0.0), b(N, 0.0);
    if (s->prior.valid) {
        double r[15];
        prior_residual(s->prior, x, r);
        for (int a = 0; a < 15; ++a) {
            for (int i = 0; i < 15; ++i) b[a] += s->prior.J[i * 15 + a] * r[i];
            for (int c = 0; c < 15; ++c)
                for (int i = 0; i < 15; ++i) A[(size_t)a * N + c] += s->prior.J[i * 15 + a] * s->prior.J[i * 15 + c];
        }
    }
    {
        double r[15], J[15 * 30];
        int rc = s->U_ok[1] ? MML_OK : MML_ERR_STATE;
        if (rc == MML_OK) imu_factor_with_U(&s->imu[1], &s->U[225], s->gravity, x, x + 6, x + 15, x + 21, r, J);
        if (rc != MML_OK) return rc;
        for (int a = 0; a < 30; ++a) {
            for (int i = 0; i < 15; ++i) b[a] += J[i * 30 + a] * r[i];
            for (int c = 0; c < 30; ++c)
                for (int i = 0; i < 15; ++i) A[(size_t)a * N + c] += J[i * 30 + a] * J[i * 30 + c];
        }
    }
    {
        int k = 0;
        for (int a = 0; a < 6; ++a)
            for (int c = a; c < 6; ++c) {
                const double v = lidar_record0[k++];
                A[(size_t)a * N + c] += v;
                if (c != a) A[(size_t)c * N + a] += v;
            }
        for (int a = 0; a < 6; ++a) b[a] += lidar_record0[21 + a];
    }
    const double eps = 1e-8;
    // Amm^-1 through the eigen-decomposition of its symmetric part, eigenvalues <= eps dropped (:203-206)
    double Amm[225], ev[15], V[225], Ainv[225];
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < m; ++c) Amm[r * m + c] = 0.5 * (A[(size_t)r * N + c] + A[(size_t)c * N + r]);
    sym_eig(Amm, m, ev, V);
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < m; ++c) {
            double sum = 0;
            for (int k = 0; k < m; ++k)
                if (ev[k] > eps) sum += V[r * m + k] * (1.0 / ev[k]) * V[c * m + k];
            Ainv[r * m + c] = sum;
        }
    // Schur complement (:208-214)
    double T[225];  // Arm * Amm_inv
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < m; ++c) {
            double sum = 0;
            for (int k = 0; k < m; ++k) sum += A[(size_t)(m + r) * N + k] * Ainv[k * m + c];
            T[r * m + c] = sum;
        }
    double Ar[225], br[15];
    for (int r = 0; r < n; ++r) {
        for (int c = 0; c < n; ++c) {
            double sum = 0;
            for (int k = 0; k < m; ++k) sum += T[r * m + k] * A[(size_t)k * N + m + c];
            Ar[r * n + c] = A[(size_t)(m + r) * N + m + c] - sum;
        }
        double sb = 0;
        for (int k = 0; k < m; ++k) sb += T[r * m + k] * b[k];
        br[r] = b[m + r] - sb;
    }
    for (int r = 0; r < n; ++r)
        for (int c = r + 1; c < n; ++c) Ar[r * n + c] = Ar[c * n + r] = 0.5 * (Ar[r * n + c] + Ar[c * n + r]);
    // linearized_jacobians = sqrt(S) V^T, linearized_residuals = sqrt(S^-1) V^T b  (:216-227)
    double ev2[15], V2[225];
    sym_eig(Ar, n, ev2, V2);
    for (int i = 0; i < n; ++i) {
        const double sv = ev2[i] > eps ? sqrt(ev2[i]) : 0.0;
        const double si = ev2[i] > eps ? sqrt(1.0 / ev2[i]) : 0.0;
        double vb = 0;
        for (int k = 0; k < n; ++k) {
            out->J[i * 15 + k] = sv * V2[k * n + i];
            vb += V2[k * n + i] * br[k];
        }
        out->r0[i] = si * vb;
    }
    memcpy(out->x0, x + 15, sizeof(double) * 15);  // parameter_block_data of the kept blocks = their current values
    return MML_OK;
}

}  // extern "C"
