// oracle/gicp.cpp -- CPU restatement of the per-frame GICP extrinsic refresh of the feature node.
// TEST INFRASTRUCTURE ONLY (see mml_oracle.h).  PARITY UNPINNED: the arithmetic lives in PCL, which is not under
// /root/reference; this restates the published algorithm of pcl::GeneralizedIterativeClosestPoint (PCL 1.8.1, ROS melodic:
// registration/include/pcl/registration/impl/gicp.hpp, bfgs.h -- the latter a port of GSL's vector_bfgs2.c /
// linear_minimize.c) as the reference configures it in icp_ext_matching (mm-loam/src/unionFeatureExtract.cpp:74-123:
// setMaximumIterations(10), setTransformationEpsilon(1e-6), everything else PCL's defaults: 20 correspondences for the
// covariances, gicp_epsilon 1e-3, rotation_epsilon 2e-3, 20 inner BFGS iterations, no correspondence distance limit,
// identity guess).
//   * computeCovariances: 20-NN of every point in its own cloud (exact; ties by index), mean / second moments accumulated
//     in neighbour order (float products, double sums), SVD-regularised covariance U diag(1, 1, eps) U^T
//   * outer loop: 1-NN of every transformed source point in the target, M = (R C1 R^T + C2)^-1, BFGS on
//     x = (t, roll, pitch, yaw) of f = 1/m sum res^T M res with the FLOAT transformation PCL builds in applyState,
//     convergence on the change of the 4 x 4 matrix (rotation entries / 2e-3, all others / 1e-6) or 10 iterations
// FLANN's tie order is unspecified (index order here); Eigen's JacobiSVD is replaced by the symmetric eigen-solver of
// linalg.h (same subspaces; when the two small eigenvalues coincide the split is arbitrary in both).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "linalg.h"
#include "mml_oracle.h"

using namespace mmlo;

namespace {

constexpr int K_CORR = 20;
constexpr double GICP_EPS = 1e-3, ROT_EPS = 2e-3;

struct M3 {
    double m[9];
};

// exact k-NN by brute force, ascending (d2, index); d2 as FLANN's L2_Simple<float>: ((dx*dx + dy*dy) + dz*dz) in float
void knn_brute(const float* pts, int n, const float* q, int k, int* idx) {
    std::vector<std::pair<float, int>> best;
    best.reserve(k + 1);
    for (int i = 0; i < n; ++i) {
        const float dx = pts[3 * i] - q[0], dy = pts[3 * i + 1] - q[1], dz = pts[3 * i + 2] - q[2];
        const float d = (dx * dx + dy * dy) + dz * dz;
        if ((int)best.size() == k && !(d < best.back().first)) continue;
        auto it = std::upper_bound(best.begin(), best.end(), std::make_pair(d, i));
        best.insert(it, std::make_pair(d, i));
        if ((int)best.size() > k) best.pop_back();
    }
    for (int j = 0; j < k; ++j) idx[j] = j < (int)best.size() ? best[j].second : -1;
}

void covariances(const float* pts, int n, std::vector<M3>& out) {
    out.assign(n, M3{{0, 0, 0, 0, 0, 0, 0, 0, 0}});
    std::vector<int> nn(K_CORR);
    for (int i = 0; i < n; ++i) {
        knn_brute(pts, n, pts + 3 * i, K_CORR, nn.data());
        double mean[3] = {0, 0, 0}, c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < K_CORR; ++j) {
            const float* p = pts + 3 * nn[j];
            mean[0] += p[0];
            mean[1] += p[1];
            mean[2] += p[2];
            c[0] += p[0] * p[0];
            c[3] += p[1] * p[0];
            c[4] += p[1] * p[1];
            c[6] += p[2] * p[0];
            c[7] += p[2] * p[1];
            c[8] += p[2] * p[2];
        }
        for (int k = 0; k < 3; ++k) mean[k] /= static_cast<double>(K_CORR);
        for (int k = 0; k < 3; ++k)
            for (int l = 0; l <= k; ++l) {
                c[3 * k + l] /= static_cast<double>(K_CORR);
                c[3 * k + l] -= mean[k] * mean[l];
                c[3 * l + k] = c[3 * k + l];
            }
        double ev[3], U[9];
        eig3_sym(c, ev, U);  // ascending eigenvalues, eigenvectors in the columns of U (row-major)
        // PCL (gicp.hpp computeCovariances): cov += v * col * col^T for k = 0, 1, 2 over the columns of the SVD's U, singular
        // values descending: the two largest -> 1, the smallest -> gicp_epsilon.  Column 2 of U here = largest eigenvalue.
        M3& o = out[i];
        for (int k = 0; k < 3; ++k) {
            const int col = 2 - k;
            const double v = k == 2 ? GICP_EPS : 1.0;
            for (int r = 0; r < 3; ++r)
                for (int s = 0; s < 3; ++s) o.m[3 * r + s] += (v * U[3 * r + col]) * U[3 * s + col];
        }
    }
}

bool inv3(const double* a, double* o) {  // Eigen's fixed-size 3 x 3 inverse: cofactors / determinant
    const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const double det = (a[0] * c00 + a[1] * c01) + a[2] * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id;
    o[1] = (a[2] * a[7] - a[1] * a[8]) * id;
    o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    o[3] = c01 * id;
    o[4] = (a[0] * a[8] - a[2] * a[6]) * id;
    o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    o[6] = c02 * id;
    o[7] = (a[1] * a[6] - a[0] * a[7]) * id;
    o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    return std::isfinite(id);
}

// GeneralizedIterativeClosestPoint::applyState on the identity base: R = Rz(x5) Ry(x4) Rx(x3) in float, t = x[0..2]
void apply_state(const double* x, float* T /*16 row-major*/) {
    // Eigen::AngleAxisf -> cos / sin of a float angle; evaluated through the double functions and rounded (equal to cosf /
    // sinf except in rare rounding ties, and the same on every platform)
    const float cz = (float)std::cos((double)(float)x[5]), sz = (float)std::sin((double)(float)x[5]);
    const float cy = (float)std::cos((double)(float)x[4]), sy = (float)std::sin((double)(float)x[4]);
    const float cx = (float)std::cos((double)(float)x[3]), sx = (float)std::sin((double)(float)x[3]);
    const float Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
    float A[9], R[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[3 * r + c] = (Rz[3 * r] * Ry[c] + Rz[3 * r + 1] * Ry[3 + c]) + Rz[3 * r + 2] * Ry[6 + c];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[3 * r + c] = (A[3 * r] * Rx[c] + A[3 * r + 1] * Rx[3 + c]) + A[3 * r + 2] * Rx[6 + c];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T[4 * r + c] = R[3 * r + c];
        T[4 * r + 3] = (float)x[r];
    }
    T[12] = T[13] = T[14] = 0.f;
    T[15] = 1.f;
}

inline void tf_point(const float* T, const float* p, float* o) {
    for (int r = 0; r < 3; ++r) o[r] = ((T[4 * r] * p[0] + T[4 * r + 1] * p[1]) + T[4 * r + 2] * p[2]) + T[4 * r + 3];
}

struct Problem {
    const float* src;
    const float* tgt;
    const std::vector<int>* is;
    const std::vector<int>* it;
    const std::vector<M3>* maha;
    // f (and optionally the gradient) at x: OptimizationFunctorWithIndices::operator() / df / fdf
    double eval(const double* x, double* g) const {
        float T[16];
        apply_state(x, T);
        const int m = (int)is->size();
        double f = 0, gt[3] = {0, 0, 0}, Rm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < m; ++i) {
            const float* ps = src + 3 * (*is)[i];
            const float* pt = tgt + 3 * (*it)[i];
            float pp[3];
            tf_point(T, ps, pp);
            const double res[3] = {(double)(pp[0] - pt[0]), (double)(pp[1] - pt[1]), (double)(pp[2] - pt[2])};
            const double* M = (*maha)[(*is)[i]].m;
            double tmp[3];
            for (int r = 0; r < 3; ++r) tmp[r] = (M[3 * r] * res[0] + M[3 * r + 1] * res[1]) + M[3 * r + 2] * res[2];
            f += (res[0] * tmp[0] + res[1] * tmp[1]) + res[2] * tmp[2];
            if (g) {
                for (int r = 0; r < 3; ++r) {
                    gt[r] += tmp[r];
                    for (int c = 0; c < 3; ++c) Rm[3 * r + c] += (double)ps[r] * tmp[c];  // base transformation = identity
                }
            }
        }
        if (g) {
            for (int r = 0; r < 3; ++r) g[r] = gt[r] * (2.0 / m);
            for (int k = 0; k < 9; ++k) Rm[k] *= 2.0 / m;
            // computeRDerivative: d/d(roll, pitch, yaw) of Rz Ry Rx, contracted with sum p temp^T
            const double phi = x[3], theta = x[4], psi = x[5];
            const double cphi = std::cos(phi), sphi = std::sin(phi), cth = std::cos(theta), sth = std::sin(theta), cpsi = std::cos(psi),
                         spsi = std::sin(psi);
            const double dPhi[9] = {0, sphi * spsi + cphi * cpsi * sth, cphi * spsi - cpsi * sphi * sth,
                                    0, -cpsi * sphi + cphi * spsi * sth, -cphi * cpsi - sphi * spsi * sth,
                                    0, cphi * cth, -cth * sphi};
            const double dTh[9] = {-cpsi * sth, cpsi * cth * sphi, cphi * cpsi * cth,
                                   -spsi * sth, cth * sphi * spsi, cphi * cth * spsi,
                                   -cth, -sphi * sth, -cphi * sth};
            const double dPsi[9] = {-cth * spsi, -cphi * cpsi - sphi * spsi * sth, cpsi * sphi - cphi * spsi * sth,
                                    cpsi * cth, -cphi * spsi + cpsi * sphi * sth, sphi * spsi + cphi * cpsi * sth,
                                    0, 0, 0};
            // matricesInnerProd(dR, R) = sum_ij dR(j, i) R(i, j)
            auto inner = [&](const double* d) {
                double r = 0;
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) r += d[3 * j + i] * Rm[3 * i + j];
                return r;
            };
            g[3] = inner(dPhi);
            g[4] = inner(dTh);
            g[5] = inner(dPsi);
        }
        return f / m;
    }
};

// ---- BFGS of PCL's bfgs.h == GSL vector_bfgs2 + linear_minimize (Fletcher's line search, cubic interpolation) ----------
double cubic(double c0, double c1, double c2, double c3, double z) { return c0 + z * (c1 + z * (c2 + z * c3)); }
void check_extremum(double c0, double c1, double c2, double c3, double z, double* zmin, double* fmin) {
    const double y = cubic(c0, c1, c2, c3, z);
    if (y < *fmin) {
        *zmin = z;
        *fmin = y;
    }
}
int solve_quadratic(double a, double b, double c, double* x0, double* x1) {
    if (a == 0) {
        if (b == 0) return 0;
        *x0 = -c / b;
        return 1;
    }
    const double disc = b * b - 4 * a * c;
    if (disc > 0) {
        if (b == 0) {
            const double r = std::sqrt(-c / a);
            *x0 = -r;
            *x1 = r;
        } else {
            const double sgnb = b > 0 ? 1 : -1;
            const double temp = -0.5 * (b + sgnb * std::sqrt(disc));
            const double r1 = temp / a, r2 = c / temp;
            if (r1 < r2) {
                *x0 = r1;
                *x1 = r2;
            } else {
                *x0 = r2;
                *x1 = r1;
            }
        }
        return 2;
    }
    if (disc == 0) {
        *x0 = -0.5 * b / a;
        *x1 = -0.5 * b / a;
        return 2;
    }
    return 0;
}
double interp_quad(double f0, double fp0, double f1, double zl, double zh) {
    const double fl = f0 + zl * (fp0 + zl * (f1 - f0 - fp0));
    const double fh = f0 + zh * (fp0 + zh * (f1 - f0 - fp0));
    const double c = 2 * (f1 - f0 - fp0);
    double zmin = zl, fmin = fl;
    if (fh < fmin) {
        zmin = zh;
        fmin = fh;
    }
    if (c > 0) {
        const double z = -fp0 / c;
        if (z > zl && z < zh) {
            const double f = f0 + z * (fp0 + z * (f1 - f0 - fp0));
            if (f < fmin) {
                zmin = z;
                fmin = f;
            }
        }
    }
    return zmin;
}
double interp_cubic(double f0, double fp0, double f1, double fp1, double zl, double zh) {
    const double eta = 3 * (f1 - f0) - 2 * fp0 - fp1;
    const double xi = fp0 + fp1 - 2 * (f1 - f0);
    const double c0 = f0, c1 = fp0, c2 = eta, c3 = xi;
    double zmin = zl, fmin = cubic(c0, c1, c2, c3, zl), z0 = 0, z1 = 0;
    check_extremum(c0, c1, c2, c3, zh, &zmin, &fmin);
    const int n = solve_quadratic(3 * c3, 2 * c2, c1, &z0, &z1);
    if (n == 2) {
        if (z0 > zl && z0 < zh) check_extremum(c0, c1, c2, c3, z0, &zmin, &fmin);
        if (z1 > zl && z1 < zh) check_extremum(c0, c1, c2, c3, z1, &zmin, &fmin);
    } else if (n == 1) {
        if (z0 > zl && z0 < zh) check_extremum(c0, c1, c2, c3, z0, &zmin, &fmin);
    }
    return zmin;
}
double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax) {
    double ymin = (xmin - a) / (b - a), ymax = (xmax - a) / (b - a);
    if (ymin > ymax) std::swap(ymin, ymax);
    double y;
    if (std::isfinite(fpb))
        y = interp_cubic(fa, fpa * (b - a), fb, fpb * (b - a), ymin, ymax);
    else
        y = interp_quad(fa, fpa * (b - a), fb, ymin, ymax);
    return a + y * (b - a);
}

struct Bfgs {
    const Problem* P;
    // state of vector_bfgs2
    double x0[6], g0[6], p[6], dx0[6], dg0[6];
    double step, g0norm, pnorm, delta_f, fp0;
    // line wrapper with its caches
    double f_alpha, df_alpha, x_alpha[6], g_alpha[6], f_key, df_key, x_key, g_key;
    double x[6], f, g[6];
    int evals = 0;

    static double nrm(const double* v) {
        double s = 0;
        for (int i = 0; i < 6; ++i) s += v[i] * v[i];
        return std::sqrt(s);
    }
    static double dotp(const double* a, const double* b) {
        double s = 0;
        for (int i = 0; i < 6; ++i) s += a[i] * b[i];
        return s;
    }
    void moveto(double alpha) {
        if (alpha == x_key) return;
        for (int i = 0; i < 6; ++i) x_alpha[i] = x0[i] + alpha * p[i];
        x_key = alpha;
    }
    double wf(double alpha) {
        if (alpha == f_key) return f_alpha;
        moveto(alpha);
        f_alpha = P->eval(x_alpha, nullptr);
        ++evals;
        f_key = alpha;
        return f_alpha;
    }
    double wdf(double alpha) {
        if (alpha == df_key) return df_alpha;
        moveto(alpha);
        if (alpha != g_key) {
            P->eval(x_alpha, g_alpha);
            ++evals;
            g_key = alpha;
        }
        df_alpha = dotp(g_alpha, p);
        df_key = alpha;
        return df_alpha;
    }
    void wfdf(double alpha, double* fo, double* dfo) {
        if (alpha == f_key && alpha == df_key) {
            *fo = f_alpha;
            *dfo = df_alpha;
            return;
        }
        if (alpha == f_key || alpha == df_key) {
            *fo = wf(alpha);
            *dfo = wdf(alpha);
            return;
        }
        moveto(alpha);
        f_alpha = P->eval(x_alpha, g_alpha);
        ++evals;
        f_key = alpha;
        g_key = alpha;
        df_alpha = dotp(g_alpha, p);
        df_key = alpha;
        *fo = f_alpha;
        *dfo = df_alpha;
    }
    void prepare_wrapper() {  // caches describe alpha = 0: the current point
        for (int i = 0; i < 6; ++i) {
            x_alpha[i] = x0[i];
            g_alpha[i] = g0[i];
        }
        x_key = 0.0;
        f_alpha = f;
        f_key = 0.0;
        g_key = 0.0;
        df_alpha = dotp(g_alpha, p);
        df_key = 0.0;
    }
    void init(const double* xin) {
        step = 1.0;
        delta_f = 0;
        for (int i = 0; i < 6; ++i) x[i] = xin[i];
        f = P->eval(x, g);
        ++evals;
        for (int i = 0; i < 6; ++i) {
            x0[i] = x[i];
            g0[i] = g[i];
        }
        g0norm = nrm(g0);
        for (int i = 0; i < 6; ++i) p[i] = g[i] * (-1.0 / g0norm);
        pnorm = nrm(p);
        fp0 = -g0norm;
        prepare_wrapper();
    }
    // 0 success, 1 no progress
    int line_search(double alpha1, double* alpha_new) {
        const double rho = 0.01, sigma = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5;
        double f0, fp0l, falpha, falpha_prev, fpalpha = 0, fpalpha_prev, delta, alpha_next;
        double alpha = alpha1, alpha_prev = 0.0;
        double a = 0.0, b = alpha, fa, fb = 0.0, fpa, fpb = 0.0;
        int i = 0;
        wfdf(0.0, &f0, &fp0l);
        falpha_prev = f0;
        fpalpha_prev = fp0l;
        fa = f0;
        fpa = fp0l;
        while (i++ < 100) {  // bracketing
            falpha = wf(alpha);
            if (falpha > f0 + alpha * rho * fp0l || falpha >= falpha_prev) {
                a = alpha_prev;
                fa = falpha_prev;
                fpa = fpalpha_prev;
                b = alpha;
                fb = falpha;
                fpb = std::numeric_limits<double>::quiet_NaN();
                break;
            }
            fpalpha = wdf(alpha);
            if (std::fabs(fpalpha) <= -sigma * fp0l) {
                *alpha_new = alpha;
                return 0;
            }
            if (fpalpha >= 0) {
                a = alpha;
                fa = falpha;
                fpa = fpalpha;
                b = alpha_prev;
                fb = falpha_prev;
                fpb = fpalpha_prev;
                break;
            }
            delta = alpha - alpha_prev;
            alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta, alpha + tau1 * delta);
            alpha_prev = alpha;
            falpha_prev = falpha;
            fpalpha_prev = fpalpha;
            alpha = alpha_next;
        }
        while (i++ < 100) {  // sectioning
            delta = b - a;
            alpha = interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta);
            falpha = wf(alpha);
            if ((a - alpha) * fpa <= std::numeric_limits<double>::epsilon()) return 1;  // roundoff prevents progress
            if (falpha > f0 + rho * alpha * fp0l || falpha >= fa) {
                b = alpha;
                fb = falpha;
                fpb = std::numeric_limits<double>::quiet_NaN();
            } else {
                fpalpha = wdf(alpha);
                if (std::fabs(fpalpha) <= -sigma * fp0l) {
                    *alpha_new = alpha;
                    return 0;
                }
                if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
                    b = a;
                    fb = fa;
                    fpb = fpa;
                    a = alpha;
                    fa = falpha;
                    fpa = fpalpha;
                } else {
                    a = alpha;
                    fa = falpha;
                    fpa = fpalpha;
                }
            }
        }
        *alpha_new = alpha;
        return 0;
    }
    int iterate() {  // minimizeOneStep
        double alpha = 0.0, alpha1;
        const double f0 = f;
        if (pnorm == 0.0 || g0norm == 0.0 || fp0 == 0) return 1;
        if (delta_f < 0) {
            const double del = std::max(-delta_f, 10 * std::numeric_limits<double>::epsilon() * std::fabs(f0));
            alpha1 = std::min(1.0, 2.0 * del / (-fp0));
        } else {
            alpha1 = std::fabs(step);
        }
        const int status = line_search(alpha1, &alpha);
        if (status != 0) return status;
        // update_position: x = x0 + alpha p with f and g there
        double fn, dfn;
        wfdf(alpha, &fn, &dfn);
        for (int i = 0; i < 6; ++i) {
            x[i] = x_alpha[i];
            g[i] = g_alpha[i];
        }
        f = fn;
        delta_f = f - f0;
        for (int i = 0; i < 6; ++i) {
            dx0[i] = x[i] - x0[i];
            dg0[i] = g[i] - g0[i];
        }
        const double dxg = dotp(dx0, g), dgg = dotp(dg0, g), dxdg = dotp(dx0, dg0), dgnorm = nrm(dg0);
        double A = 0, B = 0;
        if (dxdg != 0) {
            B = dxg / dxdg;
            A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg;
        }
        for (int i = 0; i < 6; ++i) p[i] = (g[i] - A * dx0[i]) - B * dg0[i];
        for (int i = 0; i < 6; ++i) {
            g0[i] = g[i];
            x0[i] = x[i];
        }
        g0norm = nrm(g0);
        pnorm = nrm(p);
        const double pg = dotp(p, g);
        const double dir = (pg >= 0.0) ? -1.0 : +1.0;
        for (int i = 0; i < 6; ++i) p[i] *= dir / pnorm;
        pnorm = nrm(p);
        fp0 = dotp(p, g0);
        prepare_wrapper();
        return 0;
    }
};

}  // namespace

// icp_ext_matching (unionFeatureExtract.cpp:74-123) on plain buffers.  T_inout: row-major 4 x 4 float, written only on
// convergence (the reference assigns icp_mtx only when hasConverged()).  Returns 1 converged, 0 failed.
extern "C" int mmlo_gicp_align(const float* src, int n_src, const float* tgt, int n_tgt, float* T_inout, int* outer_iterations,
                               int* bfgs_evaluations, double* last_objective) {
    if (outer_iterations) *outer_iterations = 0;
    if (bfgs_evaluations) *bfgs_evaluations = 0;
    if (n_src < K_CORR || n_tgt < K_CORR) return 0;  // "Number of points in cloud smaller than k_correspondences_"
    std::vector<M3> c_src, c_tgt, maha(n_src, M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}});
    covariances(tgt, n_tgt, c_tgt);
    covariances(src, n_src, c_src);
    float T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // transformation_ (guess = identity)
    int nr = 0, evals = 0;
    bool converged = false;
    double fobj = 0;
    while (!converged) {
        std::vector<int> is, it;
        double R[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R[3 * r + c] = (double)T[4 * r + c];
        for (int i = 0; i < n_src; ++i) {
            float q[3];
            tf_point(T, src + 3 * i, q);
            int j;
            knn_brute(tgt, n_tgt, q, 1, &j);
            // corr_dist_threshold_ = sqrt(DBL_MAX): every finite distance passes
            const double* C1 = c_src[i].m;
            const double* C2 = c_tgt[j].m;
            double RC[9], tmp[9];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) RC[3 * r + c] = (R[3 * r] * C1[c] + R[3 * r + 1] * C1[3 + c]) + R[3 * r + 2] * C1[6 + c];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c)
                    tmp[3 * r + c] = ((RC[3 * r] * R[3 * c] + RC[3 * r + 1] * R[3 * c + 1]) + RC[3 * r + 2] * R[3 * c + 2]) + C2[3 * r + c];
            inv3(tmp, maha[i].m);
            is.push_back(i);
            it.push_back(j);
        }
        if (is.size() < 4) break;  // NotEnoughPointsException: the loop ends unconverged
        float prev[16];
        memcpy(prev, T, sizeof(T));
        double x[6] = {T[3], T[7], T[11], std::atan2((double)T[9], (double)T[10]), std::asin(-(double)T[8]), std::atan2((double)T[4], (double)T[0])};
        Problem P{src, tgt, &is, &it, &maha};
        Bfgs b;
        b.P = &P;
        b.init(x);
        int inner = 0, result = 0;
        do {
            ++inner;
            result = b.iterate();
            if (result) break;                             // NoProgress
            result = b.nrm(b.g) < 1e-2 ? 2 : 0;            // testGradient(gradient_tol): Success / Running
        } while (result == 0 && inner < 20);
        evals += b.evals;
        fobj = b.f;
        apply_state(b.x, T);
        double delta = 0;
        for (int k = 0; k < 4; ++k)
            for (int l = 0; l < 4; ++l) {
                const double ratio = (k < 3 && l < 3) ? 1.0 / ROT_EPS : 1.0 / 1e-6;
                const double c_delta = ratio * std::fabs((double)prev[4 * k + l] - (double)T[4 * k + l]);
                if (c_delta > delta) delta = c_delta;
            }
        ++nr;
        if (nr >= 10 || delta < 1) converged = true;
    }
    if (outer_iterations) *outer_iterations = nr;
    if (bfgs_evaluations) *bfgs_evaluations = evals;
    if (last_objective) *last_objective = fobj;
    if (!converged) return 0;
    memcpy(T_inout, T, sizeof(T));
    return 1;
}

// exposed for the numpy cross-checks: the objective and its gradient for given correspondences
extern "C" double mmlo_gicp_objective(const float* src, const float* tgt, const int* idx_src, const int* idx_tgt, int m, const double* maha9,
                                      int n_src, const double* x, double* grad) {
    std::vector<int> is(idx_src, idx_src + m), it(idx_tgt, idx_tgt + m);
    std::vector<M3> maha(n_src);
    for (int i = 0; i < n_src; ++i) memcpy(maha[i].m, maha9 + 9 * (size_t)i, sizeof(double) * 9);
    Problem P{src, tgt, &is, &it, &maha};
    return P.eval(x, grad);
}
extern "C" void mmlo_gicp_covariances(const float* pts, int n, double* out9) {
    std::vector<M3> c;
    covariances(pts, n, c);
    for (int i = 0; i < n; ++i) memcpy(out9 + 9 * (size_t)i, c[i].m, sizeof(double) * 9);
}
