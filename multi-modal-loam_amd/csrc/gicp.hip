// gicp.hip -- SURVEY.md 8(f) rank 4: the per-frame GICP extrinsic refresh of the feature node on the device.
// Replaces icp_ext_matching (mm-loam/src/unionFeatureExtract.cpp:74-123, called at :302-312 with the Livox surf cloud as source
// and the Velodyne surf cloud as target): pcl::GeneralizedIterativeClosestPoint with setMaximumIterations(10),
// setTransformationEpsilon(1e-6) and PCL's defaults otherwise (20 correspondences per covariance, gicp_epsilon 1e-3,
// rotation_epsilon 2e-3, 20 inner BFGS iterations, no correspondence distance limit, identity guess).  PCL is not in the
// reference tree; this follows the published algorithm of PCL 1.8.1 (gicp.hpp; bfgs.h = GSL's vector_bfgs2 with
// Fletcher's line search).
//
// What runs where (everything on the ctx stream, one 96-byte read-back at the end):
//   k_gicp_cov   one lane per point: exact 20-NN in its own cloud by brute force over LDS tiles (the clouds are a few
//                hundred to a few thousand surf points -- the search grid of the association would cost more to build
//                than the N^2 distance tests), top-20 list as sorted 64-bit (distance, index) keys in registers, second
//                moments in neighbour order, symmetric eigen-solver, covariance = sum of v_k u_k u_k^T as PCL forms it (v = 1, 1, 1e-3)
//   k_gicp_corr  one lane per source point: transform with the current estimate, exact 1-NN in the target, Mahalanobis
//                matrix (R C1 R^T + C2)^-1
//   k_gicp_bfgs  ONE workgroup: the whole inner BFGS minimisation of f(x) = 1/m sum res^T M res over x = (t, roll, pitch, yaw).
//                Every thread runs the (scalar) BFGS / line-search control flow on identical values; an objective
//                evaluation is a strided pass over the correspondences + a block reduction of 13 doubles, broadcast
//                back to all threads.  The kernel then forms the new 4 x 4 float transformation, the convergence measure
//                of the outer loop and the iteration count, so the host enqueues 10 (corr, bfgs) rounds blindly: a
//                converged or failed state turns the remaining rounds into no-ops.
#include <math.h>
#include <string.h>

#include "mml_internal.h"

namespace {

#include "eig3_dev.h"

constexpr int GK = 20;            // k_correspondences_
constexpr int TILE = 1024;        // points per LDS tile of the brute-force searches
constexpr int BF_THREADS = 256;
constexpr double GICP_EPS = 1e-3, ROT_EPS = 2e-3, TRANS_EPS = 1e-6;

struct GicpState {  // device-resident state of the outer loop
    float T[16];    // transformation_ (row-major)
    int converged, failed, nr, evals;
    double fobj, delta;
};

__device__ __forceinline__ unsigned long long dkey(float d, int id) {
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)id;
}

__global__ __launch_bounds__(256) void k_gicp_cov(const float4* __restrict__ pts, int n, double* __restrict__ cov) {
    __shared__ float4 tile[TILE];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float4 q = pts[i < n ? i : n - 1];
    unsigned long long k[GK];
#pragma unroll
    for (int j = 0; j < GK; ++j) k[j] = ~0ull;
    for (int base = 0; base < n; base += TILE) {
        const int cnt = min(TILE, n - base);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += blockDim.x) tile[t] = pts[base + t];
        __syncthreads();
        for (int t = 0; t < cnt; ++t) {
            const float4 p = tile[t];
            const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
            const float d = (dx * dx + dy * dy) + dz * dz;
            const unsigned long long key = dkey(d, base + t);
            if (key < k[GK - 1]) {
                k[GK - 1] = key;
#pragma unroll
                for (int j = GK - 1; j > 0; --j) {
                    const unsigned long long lo = k[j - 1] < k[j] ? k[j - 1] : k[j], hi = k[j - 1] < k[j] ? k[j] : k[j - 1];
                    k[j - 1] = lo;
                    k[j] = hi;
                }
            }
        }
    }
    if (i >= n) return;
    // computeCovariances: sums in neighbour order, float products accumulated in double
    double mean[3] = {0, 0, 0}, c00 = 0, c10 = 0, c11 = 0, c20 = 0, c21 = 0, c22 = 0;
#pragma unroll
    for (int j = 0; j < GK; ++j) {
        const float4 p = pts[(unsigned)k[j]];
        mean[0] += p.x;
        mean[1] += p.y;
        mean[2] += p.z;
        c00 += p.x * p.x;
        c10 += p.y * p.x;
        c11 += p.y * p.y;
        c20 += p.z * p.x;
        c21 += p.z * p.y;
        c22 += p.z * p.z;
    }
    const double kk = static_cast<double>(GK);
    mean[0] /= kk;
    mean[1] /= kk;
    mean[2] /= kk;
    c00 = c00 / kk - mean[0] * mean[0];
    c10 = c10 / kk - mean[1] * mean[0];
    c11 = c11 / kk - mean[1] * mean[1];
    c20 = c20 / kk - mean[2] * mean[0];
    c21 = c21 / kk - mean[2] * mean[1];
    c22 = c22 / kk - mean[2] * mean[2];
    double ev[3], u2[3], u0[3], u1[3];
    eig3_sym(c00, c10, c11, c20, c21, c22, ev, u2, u0, u1);
    // PCL (gicp.hpp computeCovariances): cov = sum_k v_k col_k col_k^T over the columns of U in the SVD's order -- singular values
    // descending --, v = 1, 1, gicp_epsilon; each term is (v col[r]) col[s], added to the matrix one k after the other
    double* o = cov + 9 * (size_t)i;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            double a = 0.0;
            a += (1.0 * u2[r]) * u2[s];
            a += (1.0 * u1[r]) * u1[s];
            a += (GICP_EPS * u0[r]) * u0[s];
            o[3 * r + s] = a;
        }
}

__device__ __forceinline__ void tf_pt(const float* T, float x, float y, float z, float* o) {
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] = ((T[4 * r] * x + T[4 * r + 1] * y) + T[4 * r + 2] * z) + T[4 * r + 3];
}

__global__ __launch_bounds__(256) void k_gicp_corr(const float4* __restrict__ src, int ns, const float4* __restrict__ tgt, int nt,
                                                   const double* __restrict__ csrc, const double* __restrict__ ctgt,
                                                   const GicpState* st, int* __restrict__ corr, double* __restrict__ maha) {
    __shared__ float4 tile[TILE];
    if (st->converged || st->failed) return;
    float T[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) T[c] = st->T[c];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float4 p = src[i < ns ? i : ns - 1];
    float q[3];
    tf_pt(T, p.x, p.y, p.z, q);
    unsigned long long best = ~0ull;
    for (int base = 0; base < nt; base += TILE) {
        const int cnt = min(TILE, nt - base);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += blockDim.x) tile[t] = tgt[base + t];
        __syncthreads();
        for (int t = 0; t < cnt; ++t) {
            const float4 g = tile[t];
            const float dx = g.x - q[0], dy = g.y - q[1], dz = g.z - q[2];
            const float d = (dx * dx + dy * dy) + dz * dz;
            const unsigned long long key = dkey(d, base + t);
            best = key < best ? key : best;
        }
    }
    if (i >= ns) return;
    const int j = (int)(unsigned)best;
    corr[i] = j;
    double R[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) R[3 * r + c] = (double)T[4 * r + c];
    const double* C1 = csrc + 9 * (size_t)i;
    const double* C2 = ctgt + 9 * (size_t)j;
    double RC[9], a[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) RC[3 * r + c] = (R[3 * r] * C1[c] + R[3 * r + 1] * C1[3 + c]) + R[3 * r + 2] * C1[6 + c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            a[3 * r + c] = ((RC[3 * r] * R[3 * c] + RC[3 * r + 1] * R[3 * c + 1]) + RC[3 * r + 2] * R[3 * c + 2]) + C2[3 * r + c];
    // 3 x 3 inverse by cofactors (Eigen's fixed-size inverse)
    const double k00 = a[4] * a[8] - a[5] * a[7], k01 = a[5] * a[6] - a[3] * a[8], k02 = a[3] * a[7] - a[4] * a[6];
    const double det = (a[0] * k00 + a[1] * k01) + a[2] * k02;
    const double id = 1.0 / det;
    double* M = maha + 9 * (size_t)i;
    M[0] = k00 * id;
    M[1] = (a[2] * a[7] - a[1] * a[8]) * id;
    M[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    M[3] = k01 * id;
    M[4] = (a[0] * a[8] - a[2] * a[6]) * id;
    M[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    M[6] = k02 * id;
    M[7] = (a[1] * a[6] - a[0] * a[7]) * id;
    M[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

// ---- the inner BFGS, executed by every thread of one workgroup on identical values --------------------------------------
__device__ void apply_state(const double* x, float* T) {  // GeneralizedIterativeClosestPoint::applyState on the identity
    // Eigen::AngleAxisf: cos / sin of the float angle, through the double functions (platform-independent rounding)
    const float cz = (float)cos((double)(float)x[5]), sz = (float)sin((double)(float)x[5]);
    const float cy = (float)cos((double)(float)x[4]), sy = (float)sin((double)(float)x[4]);
    const float cx = (float)cos((double)(float)x[3]), sx = (float)sin((double)(float)x[3]);
    const float Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
    float A[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) A[3 * r + c] = (Rz[3 * r] * Ry[c] + Rz[3 * r + 1] * Ry[3 + c]) + Rz[3 * r + 2] * Ry[6 + c];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) T[4 * r + c] = (A[3 * r] * Rx[c] + A[3 * r + 1] * Rx[3 + c]) + A[3 * r + 2] * Rx[6 + c];
        T[4 * r + 3] = (float)x[r];
    }
    T[12] = T[13] = T[14] = 0.f;
    T[15] = 1.f;
}

constexpr int EV_CH = 512;            // correspondences per LDS chunk of an objective evaluation
constexpr int EV_ROW = EV_CH + 1;     // row stride of the term table (odd number of 8-byte words: the 13 summing lanes hit 13 banks)
struct EvalCtx {
    const float4* src;
    const float4* tgt;
    const int* corr;
    const double* maha;
    int m;
    double* s_terms;  // LDS: 13 rows x EV_ROW terms of the current chunk
    double* s_tot;    // LDS: the 13 totals
};

// OptimizationFunctorWithIndices::operator() / fdf: every thread returns the same f and (optionally) gradient.
// PCL adds the m terms of every sum one after the other; so does this: the terms of a chunk of correspondences are formed in
// parallel into an LDS table and ONE lane per sum adds them in correspondence order (the 13 sums on 13 lanes at once), so
// that f and the gradient equal the sequential loop's bit for bit -- the line search compares objective values that differ
// in their last bits near the flat minimum of a cross-sensor alignment, and a strided pass with a tree reduction decides
// some of those comparisons the other way (round 2: matrices equal to 5e-3 only).
__device__ double gicp_eval(const EvalCtx& E, const double* x, double* g) {
    float T[16];
    apply_state(x, T);
    const int nsum = g ? 13 : 1;
    double run = 0;
    for (int c0 = 0; c0 < E.m; c0 += EV_CH) {
        const int cnt = min(EV_CH, E.m - c0);
        for (int j = threadIdx.x; j < cnt; j += BF_THREADS) {
            const int i = c0 + j;
            const float4 ps = E.src[i], pt = E.tgt[E.corr[i]];
            float pp[3];
            tf_pt(T, ps.x, ps.y, ps.z, pp);
            const double res[3] = {(double)(pp[0] - pt.x), (double)(pp[1] - pt.y), (double)(pp[2] - pt.z)};
            const double* M = E.maha + 9 * (size_t)i;
            double tmp[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) tmp[r] = (M[3 * r] * res[0] + M[3 * r + 1] * res[1]) + M[3 * r + 2] * res[2];
            E.s_terms[j] = (res[0] * tmp[0] + res[1] * tmp[1]) + res[2] * tmp[2];
            if (g) {
                const double p3[3] = {(double)ps.x, (double)ps.y, (double)ps.z};
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    E.s_terms[(1 + r) * EV_ROW + j] = tmp[r];
#pragma unroll
                    for (int c = 0; c < 3; ++c) E.s_terms[(4 + 3 * r + c) * EV_ROW + j] = p3[r] * tmp[c];
                }
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < nsum) {
            const double* row = E.s_terms + threadIdx.x * EV_ROW;
            for (int j = 0; j < cnt; ++j) run += row[j];
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < nsum) E.s_tot[threadIdx.x] = run;
    __syncthreads();
    double tot[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) tot[k] = k < nsum ? E.s_tot[k] : 0.0;
    __syncthreads();
    const double m = (double)E.m;
    if (g) {
        double Rm[9];
#pragma unroll
        for (int r = 0; r < 3; ++r) g[r] = tot[1 + r] * (2.0 / m);
#pragma unroll
        for (int k = 0; k < 9; ++k) Rm[k] = tot[4 + k] * (2.0 / m);
        // computeRDerivative
        const double phi = x[3], theta = x[4], psi = x[5];
        const double cphi = cos(phi), sphi = sin(phi), cth = cos(theta), sth = sin(theta), cpsi = cos(psi), spsi = sin(psi);
        const double dPhi[9] = {0, sphi * spsi + cphi * cpsi * sth, cphi * spsi - cpsi * sphi * sth,
                                0, -cpsi * sphi + cphi * spsi * sth, -cphi * cpsi - sphi * spsi * sth,
                                0, cphi * cth, -cth * sphi};
        const double dTh[9] = {-cpsi * sth, cpsi * cth * sphi, cphi * cpsi * cth,
                               -spsi * sth, cth * sphi * spsi, cphi * cth * spsi,
                               -cth, -sphi * sth, -cphi * sth};
        const double dPsi[9] = {-cth * spsi, -cphi * cpsi - sphi * spsi * sth, cpsi * sphi - cphi * spsi * sth,
                                cpsi * cth, -cphi * spsi + cpsi * sphi * sth, sphi * spsi + cphi * cpsi * sth,
                                0, 0, 0};
        double g3 = 0, g4 = 0, g5 = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                g3 += dPhi[3 * j + i] * Rm[3 * i + j];
                g4 += dTh[3 * j + i] * Rm[3 * i + j];
                g5 += dPsi[3 * j + i] * Rm[3 * i + j];
            }
        g[3] = g3;
        g[4] = g4;
        g[5] = g5;
    }
    return tot[0] / m;
}

__device__ __forceinline__ double cubic(double c0, double c1, double c2, double c3, double z) { return c0 + z * (c1 + z * (c2 + z * c3)); }
__device__ __forceinline__ void check_extremum(double c0, double c1, double c2, double c3, double z, double& zmin, double& fmin) {
    const double y = cubic(c0, c1, c2, c3, z);
    if (y < fmin) {
        zmin = z;
        fmin = y;
    }
}
__device__ int solve_quadratic(double a, double b, double c, double& x0, double& x1) {
    if (a == 0) {
        if (b == 0) return 0;
        x0 = -c / b;
        return 1;
    }
    const double disc = b * b - 4 * a * c;
    if (disc > 0) {
        if (b == 0) {
            const double r = sqrt(-c / a);
            x0 = -r;
            x1 = r;
        } else {
            const double sgnb = b > 0 ? 1 : -1;
            const double temp = -0.5 * (b + sgnb * sqrt(disc));
            const double r1 = temp / a, r2 = c / temp;
            x0 = r1 < r2 ? r1 : r2;
            x1 = r1 < r2 ? r2 : r1;
        }
        return 2;
    }
    if (disc == 0) {
        x0 = -0.5 * b / a;
        x1 = x0;
        return 2;
    }
    return 0;
}
__device__ double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax) {
    double ymin = (xmin - a) / (b - a), ymax = (xmax - a) / (b - a);
    if (ymin > ymax) {
        const double t = ymin;
        ymin = ymax;
        ymax = t;
    }
    const double f0 = fa, fp0 = fpa * (b - a), f1 = fb;
    double zmin, fmin;
    if (isfinite(fpb)) {  // cubic through both slopes
        const double fp1 = fpb * (b - a);
        const double eta = 3 * (f1 - f0) - 2 * fp0 - fp1, xi = fp0 + fp1 - 2 * (f1 - f0);
        zmin = ymin;
        fmin = cubic(f0, fp0, eta, xi, ymin);
        check_extremum(f0, fp0, eta, xi, ymax, zmin, fmin);
        double z0 = 0, z1 = 0;
        const int n = solve_quadratic(3 * xi, 2 * eta, fp0, z0, z1);
        if (n >= 1 && z0 > ymin && z0 < ymax) check_extremum(f0, fp0, eta, xi, z0, zmin, fmin);
        if (n == 2 && z1 > ymin && z1 < ymax) check_extremum(f0, fp0, eta, xi, z1, zmin, fmin);
    } else {  // quadratic
        const double fl = f0 + ymin * (fp0 + ymin * (f1 - f0 - fp0)), fh = f0 + ymax * (fp0 + ymax * (f1 - f0 - fp0));
        const double c = 2 * (f1 - f0 - fp0);
        zmin = ymin;
        fmin = fl;
        if (fh < fmin) {
            zmin = ymax;
            fmin = fh;
        }
        if (c > 0) {
            const double z = -fp0 / c;
            if (z > ymin && z < ymax) {
                const double f = f0 + z * (fp0 + z * (f1 - f0 - fp0));
                if (f < fmin) {
                    zmin = z;
                    fmin = f;
                }
            }
        }
    }
    return a + zmin * (b - a);
}

struct Bfgs {  // vector_bfgs2 state + the line wrapper with its caches (bfgs.h)
    double x0[6], g0[6], p[6];
    double step, g0norm, pnorm, delta_f, fp0;
    double f_alpha, df_alpha, x_alpha[6], g_alpha[6], f_key, df_key, x_key, g_key;
    double x[6], f, g[6];
    int evals;
};
__device__ __forceinline__ double nrm6(const double* v) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += v[i] * v[i];
    return sqrt(s);
}
__device__ __forceinline__ double dot6(const double* a, const double* b) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += a[i] * b[i];
    return s;
}
__device__ __forceinline__ void moveto(Bfgs& B, double alpha) {
    if (alpha == B.x_key) return;
#pragma unroll
    for (int i = 0; i < 6; ++i) B.x_alpha[i] = B.x0[i] + alpha * B.p[i];
    B.x_key = alpha;
}
__device__ double wf(Bfgs& B, const EvalCtx& E, double alpha) {
    if (alpha == B.f_key) return B.f_alpha;
    moveto(B, alpha);
    B.f_alpha = gicp_eval(E, B.x_alpha, nullptr);
    ++B.evals;
    B.f_key = alpha;
    return B.f_alpha;
}
__device__ double wdf(Bfgs& B, const EvalCtx& E, double alpha) {
    if (alpha == B.df_key) return B.df_alpha;
    moveto(B, alpha);
    if (alpha != B.g_key) {
        gicp_eval(E, B.x_alpha, B.g_alpha);
        ++B.evals;
        B.g_key = alpha;
    }
    B.df_alpha = dot6(B.g_alpha, B.p);
    B.df_key = alpha;
    return B.df_alpha;
}
__device__ void wfdf(Bfgs& B, const EvalCtx& E, double alpha, double& fo, double& dfo) {
    if (alpha == B.f_key && alpha == B.df_key) {
        fo = B.f_alpha;
        dfo = B.df_alpha;
        return;
    }
    if (alpha == B.f_key || alpha == B.df_key) {
        fo = wf(B, E, alpha);
        dfo = wdf(B, E, alpha);
        return;
    }
    moveto(B, alpha);
    B.f_alpha = gicp_eval(E, B.x_alpha, B.g_alpha);
    ++B.evals;
    B.f_key = alpha;
    B.g_key = alpha;
    B.df_alpha = dot6(B.g_alpha, B.p);
    B.df_key = alpha;
    fo = B.f_alpha;
    dfo = B.df_alpha;
}
__device__ void prepare_wrapper(Bfgs& B) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        B.x_alpha[i] = B.x0[i];
        B.g_alpha[i] = B.g0[i];
    }
    B.x_key = 0.0;
    B.f_alpha = B.f;
    B.f_key = 0.0;
    B.g_key = 0.0;
    B.df_alpha = dot6(B.g_alpha, B.p);
    B.df_key = 0.0;
}
// Fletcher's line search (linear_minimize.c): 0 success, 1 no progress
__device__ int line_search(Bfgs& B, const EvalCtx& E, double alpha1, double& alpha_new) {
    const double rho = 0.01, sigma = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5;
    double f0, fp0, falpha, falpha_prev, fpalpha = 0, fpalpha_prev, delta;
    double alpha = alpha1, alpha_prev = 0.0;
    double a = 0.0, b = alpha, fa, fb = 0.0, fpa, fpb = 0.0;
    int i = 0;
    wfdf(B, E, 0.0, f0, fp0);
    falpha_prev = f0;
    fpalpha_prev = fp0;
    fa = f0;
    fpa = fp0;
    while (i++ < 100) {
        falpha = wf(B, E, alpha);
        if (falpha > f0 + alpha * rho * fp0 || falpha >= falpha_prev) {
            a = alpha_prev;
            fa = falpha_prev;
            fpa = fpalpha_prev;
            b = alpha;
            fb = falpha;
            fpb = NAN;
            break;
        }
        fpalpha = wdf(B, E, alpha);
        if (fabs(fpalpha) <= -sigma * fp0) {
            alpha_new = alpha;
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
        const double alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta, alpha + tau1 * delta);
        alpha_prev = alpha;
        falpha_prev = falpha;
        fpalpha_prev = fpalpha;
        alpha = alpha_next;
    }
    while (i++ < 100) {
        delta = b - a;
        alpha = interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta);
        falpha = wf(B, E, alpha);
        if ((a - alpha) * fpa <= 2.220446049250313e-16) return 1;
        if (falpha > f0 + rho * alpha * fp0 || falpha >= fa) {
            b = alpha;
            fb = falpha;
            fpb = NAN;
        } else {
            fpalpha = wdf(B, E, alpha);
            if (fabs(fpalpha) <= -sigma * fp0) {
                alpha_new = alpha;
                return 0;
            }
            if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
                b = a;
                fb = fa;
                fpb = fpa;
            }
            a = alpha;
            fa = falpha;
            fpa = fpalpha;
        }
    }
    alpha_new = alpha;
    return 0;
}
__device__ int bfgs_iterate(Bfgs& B, const EvalCtx& E) {
    double alpha = 0.0, alpha1;
    const double f0 = B.f;
    if (B.pnorm == 0.0 || B.g0norm == 0.0 || B.fp0 == 0) return 1;
    if (B.delta_f < 0) {
        const double del = fmax(-B.delta_f, 10 * 2.220446049250313e-16 * fabs(f0));
        alpha1 = fmin(1.0, 2.0 * del / (-B.fp0));
    } else {
        alpha1 = fabs(B.step);
    }
    const int status = line_search(B, E, alpha1, alpha);
    if (status != 0) return status;
    double fn, dfn;
    wfdf(B, E, alpha, fn, dfn);
    double dx0[6], dg0[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        B.x[i] = B.x_alpha[i];
        B.g[i] = B.g_alpha[i];
        dx0[i] = B.x[i] - B.x0[i];
        dg0[i] = B.g[i] - B.g0[i];
    }
    B.f = fn;
    B.delta_f = B.f - f0;
    const double dxg = dot6(dx0, B.g), dgg = dot6(dg0, B.g), dxdg = dot6(dx0, dg0), dgnorm = nrm6(dg0);
    double A = 0, Bc = 0;
    if (dxdg != 0) {
        Bc = dxg / dxdg;
        A = -(1.0 + dgnorm * dgnorm / dxdg) * Bc + dgg / dxdg;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        B.p[i] = (B.g[i] - A * dx0[i]) - Bc * dg0[i];
        B.g0[i] = B.g[i];
        B.x0[i] = B.x[i];
    }
    B.g0norm = nrm6(B.g0);
    B.pnorm = nrm6(B.p);
    const double pg = dot6(B.p, B.g);
    const double dir = (pg >= 0.0) ? -1.0 : +1.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) B.p[i] *= dir / B.pnorm;
    B.pnorm = nrm6(B.p);
    B.fp0 = dot6(B.p, B.g0);
    prepare_wrapper(B);
    return 0;
}

__global__ __launch_bounds__(BF_THREADS) void k_gicp_bfgs(const float4* src, const float4* tgt, const int* corr, const double* maha, int m,
                                                         GicpState* st) {
    __shared__ double s_terms[13 * EV_ROW];
    __shared__ double s_tot[13];
    if (st->converged || st->failed) return;
    if (m < 4) {  // NotEnoughPointsException: the outer loop ends unconverged
        if (threadIdx.x == 0) st->failed = 1;
        return;
    }
    EvalCtx E{src, tgt, corr, maha, m, s_terms, s_tot};
    float T[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) T[c] = st->T[c];
    Bfgs B;
    double xin[6] = {(double)T[3], (double)T[7], (double)T[11], atan2((double)T[9], (double)T[10]), asin(-(double)T[8]),
                     atan2((double)T[4], (double)T[0])};
    B.step = 1.0;
    B.delta_f = 0;
    B.evals = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) B.x[i] = xin[i];
    B.f = gicp_eval(E, B.x, B.g);
    ++B.evals;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        B.x0[i] = B.x[i];
        B.g0[i] = B.g[i];
    }
    B.g0norm = nrm6(B.g0);
#pragma unroll
    for (int i = 0; i < 6; ++i) B.p[i] = B.g[i] * (-1.0 / B.g0norm);
    B.pnorm = nrm6(B.p);
    B.fp0 = -B.g0norm;
    prepare_wrapper(B);
    int inner = 0, result = 0;
    do {
        ++inner;
        result = bfgs_iterate(B, E);
        if (result) break;
        result = nrm6(B.g) < 1e-2 ? 2 : 0;  // testGradient(gradient_tol)
    } while (result == 0 && inner < 20);
    float Tn[16];
    apply_state(B.x, Tn);
    double delta = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const double ratio = (k < 3 && l < 3) ? 1.0 / ROT_EPS : 1.0 / TRANS_EPS;
            const double cd = ratio * fabs((double)T[4 * k + l] - (double)Tn[4 * k + l]);
            delta = cd > delta ? cd : delta;
        }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < 16; ++c) st->T[c] = Tn[c];
        st->nr += 1;
        st->evals += B.evals;
        st->fobj = B.f;
        st->delta = delta;
        if (st->nr >= 10 || delta < 1) st->converged = 1;
    }
}

__global__ void k_gicp_init(GicpState* st) {
    if (threadIdx.x < 16) st->T[threadIdx.x] = (threadIdx.x % 5 == 0) ? 1.f : 0.f;
    if (threadIdx.x == 0) {
        st->converged = 0;
        st->failed = 0;
        st->nr = 0;
        st->evals = 0;
        st->fobj = 0;
        st->delta = 0;
    }
}

// The slot's surf clouds as the feature node builds them (unionFeatureExtract.cpp:1024-1031, :1242-1252): the points labelled 2 in
// the order of the sensor's RAW cloud, the Velodyne one cropped near + far (:1287-1293), the Livox one near only (:925) --
// its labelled points beyond far_th carry bit 7 in their label byte.  One wavefront per sensor walks the raw line ids in
// order and rebuilds every point's bucketed position (line start + rank among the earlier points of its line), which is
// where its label and its coordinates live.  Not a hot path: one slot, once per refresh.
__global__ __launch_bounds__(64) void k_gicp_gather_raw(const uint8_t* raw_line, const int* n_in, const int* seg_flat, const int* seg_flat_n, int blk_points,
                                                       const uint8_t* label, const float4* ln_pts, int NV, float4* velo, float4* livox, int* counts) {
    __shared__ int s_run[256];
    const int sensor = blockIdx.x, lane = threadIdx.x;
    const int n = n_in[sensor], region = sensor == 0 ? 0 : NV;
    for (int k = lane; k < 256; k += 64) s_run[k] = 0;
    __syncthreads();
    float4* out = sensor == 0 ? velo : livox;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    int n_out = 0;
    // storage: block-major (blocks of blk_points raw points; one block = the whole scan after the three-pass bucketing), line-bucketed
    // inside a block, raw order inside a (block, line) segment -- segment (blk, key) starts at flat[blk * nkeys + key]
    const int* flat = seg_flat + sensor * MML_SEG_FLAT;
    const int nkeys = seg_flat_n[2 * sensor + 1];
    for (int i0 = 0; i0 < n; i0 += 64) {
        if (i0 % blk_points == 0 && i0 > 0) {  // (blk_points is a multiple of 64: a wavefront round never straddles blocks)
            __syncthreads();
            for (int k = lane; k < 256; k += 64) s_run[k] = 0;
            __syncthreads();
        }
        const int blk = i0 / blk_points;
        const int i = i0 + lane;
        const int key = i < n ? (int)raw_line[region + i] : 255;
        const bool valid = key < 254;
        unsigned long long eq = __ballot(valid);  // lanes holding the same line id as mine
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool set = (key >> bit) & 1;
            const unsigned long long bm = __ballot(valid && set);
            eq &= set ? bm : ~bm;
        }
        int pos = 0;
        if (valid) pos = s_run[key] + __popcll(eq & lt);
        __builtin_amdgcn_wave_barrier();
        if (valid && (eq & lt) == 0) s_run[key] += __popcll(eq);
        __builtin_amdgcn_wave_barrier();
        bool take = false;
        int p = 0;
        if (valid) {
            p = flat[blk * nkeys + key] + pos;
            const unsigned l = label[p];
            take = (l & 3u) == 2u && l < (sensor == 0 ? 0x80u : 0x100u);
        }
        const unsigned long long wm = __ballot(take);
        if (take) out[n_out + __popcll(wm & lt)] = ln_pts[p];
        n_out += __popcll(wm);
    }
    if (lane == 0) counts[sensor] = n_out;
}

__global__ void k_gicp_apply(float4* pts, int n, const float* T) {  // pcl::transformPointCloud, float (PCL 1.8.1)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = pts[i];
    const float x = T[0] * p.x + T[1] * p.y + T[2] * p.z + T[3];
    const float y = T[4] * p.x + T[5] * p.y + T[6] * p.z + T[7];
    const float z = T[8] * p.x + T[9] * p.y + T[10] * p.z + T[11];
    p.x = x;
    p.y = y;
    p.z = z;
    pts[i] = p;
}

struct Scratch {
    float4 *src = nullptr, *tgt = nullptr;
    double *csrc = nullptr, *ctgt = nullptr, *maha = nullptr;
    int* corr = nullptr;
    GicpState* st = nullptr;
    int* counts = nullptr;
    float* dT = nullptr;
    std::vector<void*> owned;
    bool take(void** p, size_t bytes) {
        if (hipMalloc(p, bytes ? bytes : 16) != hipSuccess) return false;
        owned.push_back(*p);
        return true;
    }
    ~Scratch() {
        for (void* p : owned) hipFree(p);
    }
};

// align on device clouds already in S.src / S.tgt; T_inout written only on convergence
int gicp_run(mml_ctx* ctx, Scratch& S, int n_src, int n_tgt, float* T_inout, int* converged, mml_gicp_info* info) {
    hipStream_t s = MML_STREAM(ctx);
    *converged = 0;
    if (info) memset(info, 0, sizeof(*info));
    if (n_src < GK || n_tgt < GK) return MML_OK;  // PCL: "number of points smaller than k_correspondences_" -> no alignment
    bool ok = S.take((void**)&S.csrc, sizeof(double) * 9 * (size_t)n_src) && S.take((void**)&S.ctgt, sizeof(double) * 9 * (size_t)n_tgt) &&
              S.take((void**)&S.maha, sizeof(double) * 9 * (size_t)n_src) && S.take((void**)&S.corr, sizeof(int) * (size_t)n_src) &&
              S.take((void**)&S.st, sizeof(GicpState));
    MML_REQUIRE(ok, MML_ERR_HIP, "mml_gicp: device allocation failed");
    MmlStageScope t(ctx, "gicp");
    hipLaunchKernelGGL(k_gicp_init, dim3(1), dim3(64), 0, s, S.st);
    hipLaunchKernelGGL(k_gicp_cov, dim3((n_tgt + 255) / 256), dim3(256), 0, s, S.tgt, n_tgt, S.ctgt);
    hipLaunchKernelGGL(k_gicp_cov, dim3((n_src + 255) / 256), dim3(256), 0, s, S.src, n_src, S.csrc);
    for (int it = 0; it < 10; ++it) {  // setMaximumIterations(10); finished states skip their rounds
        hipLaunchKernelGGL(k_gicp_corr, dim3((n_src + 255) / 256), dim3(256), 0, s, S.src, n_src, S.tgt, n_tgt, S.csrc, S.ctgt, S.st, S.corr,
                           S.maha);
        hipLaunchKernelGGL(k_gicp_bfgs, dim3(1), dim3(BF_THREADS), 0, s, S.src, S.tgt, S.corr, S.maha, n_src, S.st);
    }
    MML_HIP(hipGetLastError());
    GicpState h;
    MML_HIP(hipMemcpyAsync(&h, S.st, sizeof(h), hipMemcpyDeviceToHost, s));
    MML_HIP(hipStreamSynchronize(s));
    *converged = (h.converged && !h.failed) ? 1 : 0;
    if (*converged) memcpy(T_inout, h.T, sizeof(float) * 16);
    if (info) {
        info->outer_iterations = h.nr;
        info->objective_evaluations = h.evals;
        info->objective = h.fobj;
        info->n_source = n_src;
        info->n_target = n_tgt;
    }
    return MML_OK;
}

}  // namespace

extern "C" int mml_gicp_align(mml_ctx* ctx, const float* src_xyz, int n_src, const float* tgt_xyz, int n_tgt, float* T_inout, int* converged,
                              mml_gicp_info* info) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE(n_src >= 0 && n_tgt >= 0 && (n_src == 0 || src_xyz) && (n_tgt == 0 || tgt_xyz) && T_inout && converged, MML_ERR_INVALID,
                "bad arguments");
    MML_HIP(hipSetDevice(ctx->device));
    int rc = mml_sync_all(ctx);
    if (rc != MML_OK) return rc;
    ctx->cur = 0;
    hipStream_t s = MML_STREAM(ctx);
    Scratch S;
    bool ok = S.take((void**)&S.src, sizeof(float4) * (size_t)(n_src > 0 ? n_src : 1)) && S.take((void**)&S.tgt, sizeof(float4) * (size_t)(n_tgt > 0 ? n_tgt : 1));
    MML_REQUIRE(ok, MML_ERR_HIP, "mml_gicp_align: device allocation failed");
    std::vector<float4> h((size_t)(n_src > n_tgt ? n_src : n_tgt) + 1);
    for (int i = 0; i < n_src; ++i) h[i] = make_float4(src_xyz[3 * i], src_xyz[3 * i + 1], src_xyz[3 * i + 2], 0.f);
    if (n_src) MML_HIP(hipMemcpyAsync(S.src, h.data(), sizeof(float4) * n_src, hipMemcpyHostToDevice, s));
    MML_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < n_tgt; ++i) h[i] = make_float4(tgt_xyz[3 * i], tgt_xyz[3 * i + 1], tgt_xyz[3 * i + 2], 0.f);
    if (n_tgt) MML_HIP(hipMemcpyAsync(S.tgt, h.data(), sizeof(float4) * n_tgt, hipMemcpyHostToDevice, s));
    MML_HIP(hipStreamSynchronize(s));
    return gicp_run(ctx, S, n_src, n_tgt, T_inout, converged, info);
}

extern "C" int mml_gicp_refresh(mml_ctx* ctx, int slot, float* extrinsic_inout, int apply, int* refreshed, mml_gicp_info* info) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE(slot >= 0 && slot < ctx->B && extrinsic_inout, MML_ERR_INVALID, "bad arguments");
    MML_HIP(hipSetDevice(ctx->device));
    int rc = mml_sync_all(ctx);
    if (rc != MML_OK) return rc;
    ctx->cur = 0;
    hipStream_t s = MML_STREAM(ctx);
    if (refreshed) *refreshed = 0;
    if (info) memset(info, 0, sizeof(*info));
    int fi[8], cb[2], fl[2];  // counters; valid points per sensor region of the slot's storage; the slot's state flags
    MML_HIP(hipMemcpyAsync(fi, ctx->fu_info + 8 * (size_t)slot, sizeof(fi), hipMemcpyDeviceToHost, s));
    MML_HIP(hipMemcpyAsync(cb, ctx->cb_n + 2 * (size_t)slot, sizeof(cb), hipMemcpyDeviceToHost, s));
    MML_HIP(hipMemcpyAsync(fl, ctx->slot_flags + 2 * (size_t)slot, sizeof(fl), hipMemcpyDeviceToHost, s));
    MML_HIP(hipStreamSynchronize(s));
    // the refresh belongs to the feature node: it aligns the surf clouds of the EXTRACTED scan (raw coordinates, raw order)
    MML_REQUIRE((fl[0] & 3) == 0, MML_ERR_STATE,
                "mml_gicp_refresh: the slot holds an uploaded or undistorted cloud (call it after mml_extract, before mml_undistort)");
    // (the one-pass bucketing keeps no per-point line ids; they are re-derived below from the slot's RAW buffers, which must
    //  therefore still be the ones the extraction read -- not a scan staged early for the next call)
    MML_REQUIRE(ctx->raw_extracted[slot], MML_ERR_STATE,
                "mml_gicp_refresh: the slot's raw scan was re-uploaded after mml_extract (or never extracted)");
    if (!(fi[4] > 100)) return MML_OK;  // union_msg.livox_corner_num > 100 (unionFeatureExtract.cpp:302)
    Scratch S;
    bool ok = S.take((void**)&S.src, sizeof(float4) * (size_t)(cb[1] + 1)) && S.take((void**)&S.tgt, sizeof(float4) * (size_t)(cb[0] + 1)) &&
              S.take((void**)&S.counts, sizeof(int) * 2) && S.take((void**)&S.dT, sizeof(float) * 16);
    MML_REQUIRE(ok, MML_ERR_HIP, "mml_gicp_refresh: device allocation failed");
    int cnt[2] = {0, 0};
    {
        const size_t o = (size_t)slot * ctx->NT;
        // (the one-pass bucketing keeps no per-point line ids: they are recomputed for this one slot)
        if (ctx->onepass) {
            rc = mml_launch_raw_lines(ctx, slot);
            if (rc != MML_OK) return rc;
        }
        hipLaunchKernelGGL(k_gicp_gather_raw, dim3(2), dim3(64), 0, s, ctx->raw_line + o, ctx->d_n_in + 2 * (size_t)slot,
                           ctx->seg_flat + (size_t)slot * 2 * MML_SEG_FLAT, ctx->seg_flat_n + (size_t)slot * 4,
                           ctx->onepass ? MML_OP_BLK : (1 << 30), ctx->ln_label + o, ctx->ln_pts + o, ctx->NV, S.tgt, S.src, S.counts);
        MML_HIP(hipMemcpyAsync(cnt, S.counts, sizeof(cnt), hipMemcpyDeviceToHost, s));
        MML_HIP(hipStreamSynchronize(s));
    }
    int conv = 0;
    rc = gicp_run(ctx, S, cnt[1], cnt[0], extrinsic_inout, &conv, info);  // source: Livox surf, target: Velodyne surf (:307)
    if (rc != MML_OK) return rc;
    if (refreshed) *refreshed = conv;
    if (apply && cb[1] > 0) {  // pcl::transformPointCloud(*livoCombinePtr, *livoCombinePtr, extri_mtx) (:312): the Livox region
        MML_HIP(hipMemcpyAsync(S.dT, extrinsic_inout, sizeof(float) * 16, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_gicp_apply, dim3((cb[1] + 255) / 256), dim3(256), 0, s, ctx->ln_pts + (size_t)slot * ctx->NT + ctx->NV,
                           cb[1], S.dT);
        MML_HIP(hipGetLastError());
        MML_HIP(hipStreamSynchronize(s));
    }
    return MML_OK;
}
