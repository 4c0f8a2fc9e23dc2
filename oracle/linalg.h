// oracle/linalg.h -- small dense linear algebra for the CPU oracle (TEST INFRASTRUCTURE ONLY).
// Restates the published algorithms of Eigen 3.3.4 (the version ROS melodic ships; Eigen is not in
// /root/reference) that the reference calls at Estimator.cpp:251,335 (SelfAdjointEigenSolver<Matrix3d>),
// :640,714 (colPivHouseholderQr on 5x3) and :1240 (LLT).  Summation order inside Eigen's dynamic
// blocks depends on run-time alignment and cannot be pinned; these are held to numpy within 1e-10.
#ifndef MML_ORACLE_LINALG_H
#define MML_ORACLE_LINALG_H

#include <cmath>
#include <limits>
#include <utility>

namespace mmlo {

struct Vec3 {
    double x, y, z;
};
inline Vec3 mk(double x, double y, double z) { return Vec3{x, y, z}; }
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator*(double s, const Vec3& a) { return mk(s * a.x, s * a.y, s * a.z); }
inline double dot(const Vec3& a, const Vec3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline double norm(const Vec3& a) { return std::sqrt(dot(a, a)); }
inline Vec3 cross(const Vec3& a, const Vec3& b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

struct Quat {  // Eigen coefficient order x,y,z,w
    double x, y, z, w;
};
// Eigen Vector4d redux with Packet2d: (c0+c2)+(c1+c3)
inline double qdot(const Quat& a, const Quat& b) { return (a.x * b.x + a.z * b.z) + (a.y * b.y + a.w * b.w); }
inline Quat qnormalized(const Quat& q) {
    double n = std::sqrt(qdot(q, q));
    return Quat{q.x / n, q.y / n, q.z / n, q.w / n};
}
inline Quat qmul(const Quat& a, const Quat& b) {  // Eigen quat_product (generic form)
    return Quat{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
                a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline Quat qconj(const Quat& q) { return Quat{-q.x, -q.y, -q.z, q.w}; }
// QuaternionBase::_transformVector
inline Vec3 qrot(const Quat& q, const Vec3& v) {
    Vec3 qv = mk(q.x, q.y, q.z);
    Vec3 uv = cross(qv, v);
    uv = uv + uv;
    return (v + q.w * uv) + cross(qv, uv);
}
// quaternionbase_assign_impl<Other,3,3>: rotation matrix (row-major m[9]) -> quaternion
inline Quat quat_from_matrix(const double* m) {
    Quat q;
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
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
        t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
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
// Quaternion::toRotationMatrix, row-major out
inline void quat_to_matrix(const Quat& q, double* R) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
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

// sophus/so3.hpp:585-622 SO3::expAndTheta (epsilon = 1e-10, common.hpp:117)
inline Quat so3_exp(const Vec3& omega) {
    double theta_sq = dot(omega, omega);
    double imag_factor, real_factor;
    if (theta_sq < 1e-10 * 1e-10) {
        double theta_po4 = theta_sq * theta_sq;
        imag_factor = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
        real_factor = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
    } else {
        double theta = std::sqrt(theta_sq);
        double half_theta = 0.5 * theta;
        double sin_half_theta = std::sin(half_theta);
        imag_factor = sin_half_theta / theta;
        real_factor = std::cos(half_theta);
    }
    return Quat{imag_factor * omega.x, imag_factor * omega.y, imag_factor * omega.z, real_factor};
}
// sophus/so3.hpp:247-287 SO3::logAndTheta.  The SO3(Quaternion) constructor normalises first (so3.hpp:185-189).
inline Vec3 so3_log(const Quat& q_in) {
    Quat q = qnormalized(q_in);
    double squared_n = (q.x * q.x + q.y * q.y) + q.z * q.z;
    double w = q.w;
    double two_atan_nbyw_by_n;
    if (squared_n < 1e-10 * 1e-10) {
        double squared_w = w * w;
        two_atan_nbyw_by_n = 2.0 / w - (2.0 / 3.0) * (squared_n) / (w * squared_w);
    } else {
        double n = std::sqrt(squared_n);
        if (std::fabs(w) < 1e-10) {
            if (w > 0.0)
                two_atan_nbyw_by_n = M_PI / n;
            else
                two_atan_nbyw_by_n = -M_PI / n;
        } else {
            two_atan_nbyw_by_n = 2.0 * std::atan(n / w) / n;
        }
    }
    return mk(two_atan_nbyw_by_n * q.x, two_atan_nbyw_by_n * q.y, two_atan_nbyw_by_n * q.z);
}

// Eigen/src/Jacobi/Jacobi.h JacobiRotation::makeGivens (real)
inline void make_givens(double p, double q, double& c, double& s) {
    if (q == 0.0) {
        c = p < 0.0 ? -1.0 : 1.0;
        s = 0.0;
    } else if (p == 0.0) {
        c = 0.0;
        s = q < 0.0 ? 1.0 : -1.0;
    } else if (std::fabs(p) > std::fabs(q)) {
        double t = q / p;
        double u = std::sqrt(1.0 + t * t);
        if (p < 0.0) u = -u;
        c = 1.0 / u;
        s = -t * c;
    } else {
        double t = p / q;
        double u = std::sqrt(1.0 + t * t);
        if (q < 0.0) u = -u;
        s = -1.0 / u;
        c = -t * s;
    }
}
inline double eig_hypot(double x, double y) {
    double ax = std::fabs(x), ay = std::fabs(y);
    double p, qp;
    if (ax > ay) {
        p = ax;
        qp = ay / p;
    } else {
        p = ay;
        qp = ax / p;
    }
    if (p == 0.0) return 0.0;
    return p * std::sqrt(1.0 + qp * qp);
}

// Eigen 3.3.4 SelfAdjointEigenSolver<Matrix3d>::compute(A, ComputeEigenvectors):
// scale to [-1,1], tridiagonalization_inplace (3x3 real special case), implicit symmetric QR steps
// with Wilkinson shift, ascending selection sort.  A: row-major symmetric (lower triangle is read).
// evecs: row-major, column k = eigenvector of evals[k].
inline void eig3_sym(const double* A, double* evals, double* evecs) {
    double m00 = A[0], m10 = A[3], m11 = A[4], m20 = A[6], m21 = A[7], m22 = A[8];
    double scale = std::fabs(m00);
    const double cand[5] = {m10, m11, m20, m21, m22};
    for (double cnd : cand)
        if (std::fabs(cnd) > scale) scale = std::fabs(cnd);
    if (scale == 0.0) scale = 1.0;
    m00 /= scale;
    m10 /= scale;
    m11 /= scale;
    m20 /= scale;
    m21 /= scale;
    m22 /= scale;

    double diag[3], sub[2];
    double Q[9];  // row-major
    const double tol = std::numeric_limits<double>::min();
    diag[0] = m00;
    double v1norm2 = m20 * m20;
    if (v1norm2 <= tol) {
        diag[1] = m11;
        diag[2] = m22;
        sub[0] = m10;
        sub[1] = m21;
        Q[0] = 1; Q[1] = 0; Q[2] = 0; Q[3] = 0; Q[4] = 1; Q[5] = 0; Q[6] = 0; Q[7] = 0; Q[8] = 1;
    } else {
        double beta = std::sqrt(m10 * m10 + v1norm2);
        double invBeta = 1.0 / beta;
        double m01 = m10 * invBeta;
        double m02 = m20 * invBeta;
        double q = 2.0 * m01 * m21 + m02 * (m22 - m11);
        diag[1] = m11 + m02 * q;
        diag[2] = m22 - m02 * q;
        sub[0] = beta;
        sub[1] = m21 - m01 * q;
        Q[0] = 1; Q[1] = 0; Q[2] = 0; Q[3] = 0; Q[4] = m01; Q[5] = m02; Q[6] = 0; Q[7] = m02; Q[8] = -m01;
    }

    const int n = 3;
    int end = n - 1, start = 0, iter = 0;
    const int maxIterations = 30;
    const double considerAsZero = std::numeric_limits<double>::min();
    const double precision = 2.0 * std::numeric_limits<double>::epsilon();
    while (end > 0) {
        for (int i = start; i < end; ++i)
            if (std::fabs(sub[i]) <= (std::fabs(diag[i]) + std::fabs(diag[i + 1])) * precision ||
                std::fabs(sub[i]) <= considerAsZero)
                sub[i] = 0;
        while (end > 0 && sub[end - 1] == 0.0) end--;
        if (end <= 0) break;
        iter++;
        if (iter > maxIterations * n) break;
        start = end - 1;
        while (start > 0 && sub[start - 1] != 0.0) start--;

        // tridiagonal_qr_step
        double td = (diag[end - 1] - diag[end]) * 0.5;
        double e = sub[end - 1];
        double mu = diag[end];
        if (td == 0.0) {
            mu -= std::fabs(e);
        } else {
            double e2 = e * e;
            double h = eig_hypot(td, e);
            if (e2 == 0.0)
                mu -= (e / (td + (td > 0.0 ? 1.0 : -1.0))) * (e / h);
            else
                mu -= e2 / (td + (td > 0.0 ? h : -h));
        }
        double x = diag[start] - mu;
        double z = sub[start];
        for (int k = start; k < end; ++k) {
            double c, s;
            make_givens(x, z, c, s);
            double sdk = s * diag[k] + c * sub[k];
            double dkp1 = s * sub[k] + c * diag[k + 1];
            diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
            diag[k + 1] = s * sdk + c * dkp1;
            sub[k] = c * sdk - s * dkp1;
            if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
            x = sub[k];
            if (k < end - 1) {
                z = -s * sub[k + 1];
                sub[k + 1] = c * sub[k + 1];
            }
            // Q = Q * G : columns k, k+1
            for (int r = 0; r < 3; ++r) {
                double xi = Q[3 * r + k], yi = Q[3 * r + k + 1];
                Q[3 * r + k] = c * xi - s * yi;
                Q[3 * r + k + 1] = s * xi + c * yi;
            }
        }
    }
    // ascending selection sort
    for (int i = 0; i < n - 1; ++i) {
        int k = 0;
        double mn = diag[i];
        for (int j = 1; j < n - i; ++j)
            if (diag[i + j] < mn) {
                mn = diag[i + j];
                k = j;
            }
        if (k > 0) {
            std::swap(diag[i], diag[k + i]);
            for (int r = 0; r < 3; ++r) std::swap(Q[3 * r + i], Q[3 * r + k + i]);
        }
    }
    for (int i = 0; i < 3; ++i) evals[i] = diag[i] * scale;
    for (int i = 0; i < 9; ++i) evecs[i] = Q[i];
}

// Eigen 3.3.4 ColPivHouseholderQR<Matrix<double,5,3>>::compute + solve(b = -1):  A row-major 5x3.
inline void plane_fit5(const double* A_in, double* xout) {
    const int rows = 5, cols = 3, size = 3;
    double qr[5][3];
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) qr[r][c] = A_in[3 * r + c];
    double hCoeffs[3];
    int transp[3];
    double normsUpdated[3], normsDirect[3];
    for (int k = 0; k < cols; ++k) {
        double s = 0;
        for (int r = 0; r < rows; ++r) s += qr[r][k] * qr[r][k];
        normsDirect[k] = std::sqrt(s);
        normsUpdated[k] = normsDirect[k];
    }
    const double eps = std::numeric_limits<double>::epsilon();
    double maxn = normsUpdated[0];
    for (int k = 1; k < cols; ++k)
        if (normsUpdated[k] > maxn) maxn = normsUpdated[k];
    double threshold_helper = (maxn * eps) * (maxn * eps) / double(rows);
    double norm_downdate_threshold = std::sqrt(eps);
    int nonzero_pivots = size;

    for (int k = 0; k < size; ++k) {
        int big = k;
        double bigv = normsUpdated[k];
        for (int j = k + 1; j < cols; ++j)
            if (normsUpdated[j] > bigv) {
                bigv = normsUpdated[j];
                big = j;
            }
        double biggest_col_sq_norm = bigv * bigv;
        if (nonzero_pivots == size && biggest_col_sq_norm < threshold_helper * double(rows - k)) nonzero_pivots = k;
        transp[k] = big;
        if (k != big) {
            for (int r = 0; r < rows; ++r) std::swap(qr[r][k], qr[r][big]);
            std::swap(normsUpdated[k], normsUpdated[big]);
            std::swap(normsDirect[k], normsDirect[big]);
        }
        // makeHouseholderInPlace on qr[k..rows-1][k]
        double tailSqNorm = 0;
        for (int r = k + 1; r < rows; ++r) tailSqNorm += qr[r][k] * qr[r][k];
        double c0 = qr[k][k];
        double tau, beta;
        const double tol = std::numeric_limits<double>::min();
        if (tailSqNorm <= tol) {
            tau = 0;
            beta = c0;
            for (int r = k + 1; r < rows; ++r) qr[r][k] = 0;
        } else {
            beta = std::sqrt(c0 * c0 + tailSqNorm);
            if (c0 >= 0.0) beta = -beta;
            for (int r = k + 1; r < rows; ++r) qr[r][k] = qr[r][k] / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        hCoeffs[k] = tau;
        qr[k][k] = beta;
        // applyHouseholderOnTheLeft to bottomRightCorner(rows-k, cols-k-1)
        if (rows - k == 1) {
            for (int j = k + 1; j < cols; ++j) qr[k][j] *= (1.0 - tau);
        } else if (tau != 0.0) {
            for (int j = k + 1; j < cols; ++j) {
                double tmp = 0;
                for (int r = k + 1; r < rows; ++r) tmp += qr[r][k] * qr[r][j];
                tmp += qr[k][j];
                qr[k][j] -= tau * tmp;
                for (int r = k + 1; r < rows; ++r) qr[r][j] -= tau * qr[r][k] * tmp;
            }
        }
        // norm downdate (LAPACK xGEQPF)
        for (int j = k + 1; j < cols; ++j) {
            if (normsUpdated[j] != 0.0) {
                double temp = std::fabs(qr[k][j]) / normsUpdated[j];
                temp = (1.0 + temp) * (1.0 - temp);
                temp = temp < 0.0 ? 0.0 : temp;
                double ratio = normsUpdated[j] / normsDirect[j];
                double temp2 = temp * (ratio * ratio);
                if (temp2 <= norm_downdate_threshold) {
                    double s = 0;
                    for (int r = k + 1; r < rows; ++r) s += qr[r][j] * qr[r][j];
                    normsDirect[j] = std::sqrt(s);
                    normsUpdated[j] = normsDirect[j];
                } else {
                    normsUpdated[j] *= std::sqrt(temp);
                }
            }
        }
    }
    // column permutation: P = T0 * T1 * T2  (indices after applying transpositions in order)
    int perm[3] = {0, 1, 2};
    for (int k = 0; k < size; ++k) std::swap(perm[k], perm[transp[k]]);

    xout[0] = xout[1] = xout[2] = 0;
    if (nonzero_pivots == 0) return;
    double c[5] = {-1, -1, -1, -1, -1};
    // c = Q^T b : apply H_0, H_1, ... in order
    for (int k = 0; k < nonzero_pivots; ++k) {
        double tau = hCoeffs[k];
        if (rows - k == 1) {
            c[k] *= (1.0 - tau);
        } else if (tau != 0.0) {
            double tmp = 0;
            for (int r = k + 1; r < rows; ++r) tmp += qr[r][k] * c[r];
            tmp += c[k];
            c[k] -= tau * tmp;
            for (int r = k + 1; r < rows; ++r) c[r] -= tau * qr[r][k] * tmp;
        }
    }
    // back substitution on the leading nonzero_pivots block
    for (int i = nonzero_pivots - 1; i >= 0; --i) {
        double s = c[i];
        for (int j = i + 1; j < nonzero_pivots; ++j) s -= qr[i][j] * c[j];
        c[i] = s / qr[i][i];
    }
    for (int i = 0; i < nonzero_pivots; ++i) xout[perm[i]] = c[i];
}

// Dense Cholesky solve (n <= 64): A row-major SPD (overwritten), b overwritten with x.  Returns false
// when a pivot is not positive (Ceres DENSE_SCHUR -> Eigen LLT failure -> LINEAR_SOLVER_FAILURE).
inline bool chol_solve(double* A, double* b, int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= A[i * n + k] * b[k];
        b[i] = s / A[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= A[k * n + i] * b[k];
        b[i] = s / A[i * n + i];
    }
    return true;
}

}  // namespace mmlo
#endif
