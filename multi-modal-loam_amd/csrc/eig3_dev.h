// eig3_dev.h -- Eigen 3.3.4 SelfAdjointEigenSolver<Matrix3d>::compute restated for one lane; shared by map_assoc.hip (line
// fit) and gicp.hip (covariance regularisation).  Include inside an anonymous namespace of the translation unit.
#ifndef MML_EIG3_DEV_H
#define MML_EIG3_DEV_H
// ---------------------------------------------------------------------------------------------------
// Eigen 3.3.4 SelfAdjointEigenSolver<Matrix3d>::compute restated for one lane (see oracle/linalg.h for the
// line-by-line citation of the algorithm: scaling, 3x3 tridiagonalisation, implicit QR with Wilkinson shift).
__device__ __forceinline__ void make_givens(double p, double q, double& c, double& s) {
    if (q == 0.0) {
        c = p < 0.0 ? -1.0 : 1.0;
        s = 0.0;
    } else if (p == 0.0) {
        c = 0.0;
        s = q < 0.0 ? 1.0 : -1.0;
    } else if (fabs(p) > fabs(q)) {
        double t = q / p;
        double u = sqrt(1.0 + t * t);
        if (p < 0.0) u = -u;
        c = 1.0 / u;
        s = -t * c;
    } else {
        double t = p / q;
        double u = sqrt(1.0 + t * t);
        if (q < 0.0) u = -u;
        s = -1.0 / u;
        c = -t * s;
    }
}
__device__ __forceinline__ double eig_hypot(double x, double y) {
    double ax = fabs(x), ay = fabs(y), p, qp;
    if (ax > ay) {
        p = ax;
        qp = ay / p;
    } else {
        p = ay;
        qp = ax / p;
    }
    if (p == 0.0) return 0.0;
    return p * sqrt(1.0 + qp * qp);
}
// A: lower triangle m00,m10,m11,m20,m21,m22.  Returns eigenvalues ascending in ev[], eigenvector of ev[2] in v2[].
// Register-only formulation: every array index is a compile-time constant after unrolling (dynamic positions are
// resolved with selects), so nothing spills to scratch memory -- the per-lane scratch traffic of an indexed
// implementation dominated the association kernel.
struct Tri3 {
    double d0, d1, d2, e0, e1;
};
__device__ __forceinline__ double tri_d(const Tri3& t, int i) { return i == 0 ? t.d0 : (i == 1 ? t.d1 : t.d2); }
__device__ __forceinline__ double tri_e(const Tri3& t, int i) { return i == 0 ? t.e0 : t.e1; }
__device__ __forceinline__ void tri_set_d(Tri3& t, int i, double v) {
    t.d0 = i == 0 ? v : t.d0;
    t.d1 = i == 1 ? v : t.d1;
    t.d2 = i == 2 ? v : t.d2;
}
__device__ __forceinline__ void tri_set_e(Tri3& t, int i, double v) {
    t.e0 = i == 0 ? v : t.e0;
    t.e1 = i == 1 ? v : t.e1;
}

__device__ void eig3_sym(double m00, double m10, double m11, double m20, double m21, double m22, double* ev,
                         double* v2, double* v0 = nullptr, double* v1 = nullptr) {
    double scale = fabs(m00);
    scale = fmax(scale, fabs(m10));
    scale = fmax(scale, fabs(m11));
    scale = fmax(scale, fabs(m20));
    scale = fmax(scale, fabs(m21));
    scale = fmax(scale, fabs(m22));
    if (scale == 0.0) scale = 1.0;
    m00 /= scale;
    m10 /= scale;
    m11 /= scale;
    m20 /= scale;
    m21 /= scale;
    m22 /= scale;
    Tri3 t;
    // Q as three columns q0,q1,q2 (each 3 rows)
    double q00, q10, q20, q01, q11, q21, q02, q12, q22;
    const double tol = 2.2250738585072014e-308;
    t.d0 = m00;
    const double v1norm2 = m20 * m20;
    if (v1norm2 <= tol) {
        t.d1 = m11;
        t.d2 = m22;
        t.e0 = m10;
        t.e1 = m21;
        q00 = 1; q10 = 0; q20 = 0;
        q01 = 0; q11 = 1; q21 = 0;
        q02 = 0; q12 = 0; q22 = 1;
    } else {
        const double beta = sqrt(m10 * m10 + v1norm2);
        const double invBeta = 1.0 / beta;
        const double m01 = m10 * invBeta;
        const double m02 = m20 * invBeta;
        const double q = 2.0 * m01 * m21 + m02 * (m22 - m11);
        t.d1 = m11 + m02 * q;
        t.d2 = m22 - m02 * q;
        t.e0 = beta;
        t.e1 = m21 - m01 * q;
        q00 = 1; q10 = 0; q20 = 0;
        q01 = 0; q11 = m01; q21 = m02;
        q02 = 0; q12 = m02; q22 = -m01;
    }
    int end = 2, start = 0, iter = 0;
    const double precision = 2.0 * 2.220446049250313e-16;
    while (end > 0) {
        for (int i = start; i < end; ++i) {
            const double si = tri_e(t, i);
            if (fabs(si) <= (fabs(tri_d(t, i)) + fabs(tri_d(t, i + 1))) * precision || fabs(si) <= tol) tri_set_e(t, i, 0.0);
        }
        while (end > 0 && tri_e(t, end - 1) == 0.0) end--;
        if (end <= 0) break;
        iter++;
        if (iter > 90) break;
        start = end - 1;
        while (start > 0 && tri_e(t, start - 1) != 0.0) start--;
        const double td = (tri_d(t, end - 1) - tri_d(t, end)) * 0.5;
        const double e = tri_e(t, end - 1);
        double mu = tri_d(t, end);
        if (td == 0.0) {
            mu -= fabs(e);
        } else {
            const double e2 = e * e;
            const double h = eig_hypot(td, e);
            if (e2 == 0.0)
                mu -= (e / (td + (td > 0.0 ? 1.0 : -1.0))) * (e / h);
            else
                mu -= e2 / (td + (td > 0.0 ? h : -h));
        }
        double x = tri_d(t, start) - mu;
        double z = tri_e(t, start);
        for (int k = start; k < end; ++k) {
            double c, s;
            make_givens(x, z, c, s);
            const double dk = tri_d(t, k), dk1 = tri_d(t, k + 1), sk = tri_e(t, k);
            const double sdk = s * dk + c * sk;
            const double dkp1 = s * sk + c * dk1;
            tri_set_d(t, k, c * (c * dk - s * sk) - s * (c * sk - s * dk1));
            tri_set_d(t, k + 1, s * sdk + c * dkp1);
            tri_set_e(t, k, c * sdk - s * dkp1);
            if (k > start) tri_set_e(t, k - 1, c * tri_e(t, k - 1) - s * z);
            x = tri_e(t, k);
            if (k < end - 1) {
                const double sk1 = tri_e(t, k + 1);
                z = -s * sk1;
                tri_set_e(t, k + 1, c * sk1);
            }
            // Q = Q * G on columns (k, k+1)
            if (k == 0) {
                double a, b;
                a = q00; b = q01; q00 = c * a - s * b; q01 = s * a + c * b;
                a = q10; b = q11; q10 = c * a - s * b; q11 = s * a + c * b;
                a = q20; b = q21; q20 = c * a - s * b; q21 = s * a + c * b;
            } else {
                double a, b;
                a = q01; b = q02; q01 = c * a - s * b; q02 = s * a + c * b;
                a = q11; b = q12; q11 = c * a - s * b; q12 = s * a + c * b;
                a = q21; b = q22; q21 = c * a - s * b; q22 = s * a + c * b;
            }
        }
    }
    // ascending selection sort (two passes), columns follow
    double d0 = t.d0, d1 = t.d1, d2 = t.d2;
    {
        // i = 0: k = argmin(d0,d1,d2) (first minimum)
        int k = 0;
        double mn = d0;
        if (d1 < mn) { mn = d1; k = 1; }
        if (d2 < mn) { mn = d2; k = 2; }
        if (k == 1) {
            double u;
            u = d0; d0 = d1; d1 = u;
            u = q00; q00 = q01; q01 = u;
            u = q10; q10 = q11; q11 = u;
            u = q20; q20 = q21; q21 = u;
        } else if (k == 2) {
            double u;
            u = d0; d0 = d2; d2 = u;
            u = q00; q00 = q02; q02 = u;
            u = q10; q10 = q12; q12 = u;
            u = q20; q20 = q22; q22 = u;
        }
        // i = 1
        if (d2 < d1) {
            double u;
            u = d1; d1 = d2; d2 = u;
            u = q01; q01 = q02; q02 = u;
            u = q11; q11 = q12; q12 = u;
            u = q21; q21 = q22; q22 = u;
        }
    }
    ev[0] = d0 * scale;
    ev[1] = d1 * scale;
    ev[2] = d2 * scale;
    v2[0] = q02;
    v2[1] = q12;
    v2[2] = q22;
    if (v0) {  // eigenvector of the smallest eigenvalue
        v0[0] = q00;
        v0[1] = q10;
        v0[2] = q20;
    }
    if (v1) {  // ... and of the middle one
        v1[0] = q01;
        v1[1] = q11;
        v1[2] = q21;
    }
}

#endif
