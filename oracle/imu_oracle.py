"""Test infrastructure only (never imported by the product): numpy restatement of the IMU side of the reference's
full-window solve, written independently of multi-modal-loam_amd/csrc/window_imu.hip so that the two can check each
other.  PARITY UNPINNED: the reference has no tests or golden vectors and cannot be built here (Eigen, Sophus, Ceres
absent).
  * preintegrate   IMUIntegrator::PreIntegration                 mm-loam/src/lio/IMUIntegrator.cpp:108-166
  * imu_residual   Cost_NavState_PRV_Bias::operator()            mm-loam/include/utils/ceresfunc.h:330-379
  * sqrt_info      LLT(covariance^-1).matrixL().transpose()      mm-loam/src/lio/Estimator.cpp:1240-1242
  * prior_residual MarginalizationFactor::Evaluate               ceresfunc.h:262-301
  * marginalize    MarginalizationInfo::marginalize              ceresfunc.h:149-228
Jacobians here are central differences of the residuals (the reference uses Ceres autodiff)."""
import numpy as np
from scipy.spatial.transform import Rotation as Rsc

GNORM = 9.805
ACC_N, GYR_N, ACC_W, GYR_W = 0.08, 0.004, 2.0e-4, 2.0e-5


def hat(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


def exp_so3(w):
    return Rsc.from_rotvec(np.asarray(w, dtype=np.float64)).as_matrix()


def log_so3(R):
    return Rsc.from_matrix(R).as_rotvec()


def preintegrate(samples, bg, ba):
    samples = np.asarray(samples, dtype=np.float64).reshape(-1, 7)
    dq = np.eye(3)
    dp = np.zeros(3)
    dv = np.zeros(3)
    jac = np.eye(15)
    cov = np.zeros((15, 15))
    noise = np.diag([GYR_N ** 2] * 3 + [ACC_N ** 2] * 3 + [GYR_W ** 2] * 3 + [ACC_W ** 2] * 3)
    dtime = 0.0
    for m in samples:
        gyr = m[0:3] - bg
        acc = m[3:6] * GNORM - ba
        dt = m[6]
        dt2 = dt * dt
        gdt = gyr * dt
        dR = exp_so3(gdt)
        Jr = np.eye(3)
        nrm = np.linalg.norm(gdt)
        if nrm > 0.00001:
            K = hat(gdt / nrm)
            Jr = np.eye(3) - (1 - np.cos(nrm)) / nrm * K + (1 - np.sin(nrm) / nrm) * K @ K
        A = np.eye(15)
        A[0:3, 3:6] = -0.5 * dq @ hat(acc) * dt2
        A[0:3, 6:9] = np.eye(3) * dt
        A[0:3, 12:15] = -0.5 * dq * dt2
        A[3:6, 3:6] = dR.T
        A[3:6, 9:12] = -Jr * dt
        A[6:9, 3:6] = -dq @ hat(acc) * dt
        A[6:9, 12:15] = -dq * dt
        B = np.zeros((15, 12))
        B[0:3, 3:6] = 0.5 * dq * dt2
        B[3:6, 0:3] = Jr * dt
        B[6:9, 3:6] = dq * dt
        B[9:12, 6:9] = np.eye(3) * dt
        B[12:15, 9:12] = np.eye(3) * dt
        jac = A @ jac
        cov = A @ cov @ A.T + B @ noise @ B.T
        dp = dp + dv * dt + 0.5 * dq @ acc * dt2
        dv = dv + dq @ acc * dt
        dq = dq @ dR
        dtime += dt
    return dict(dp=dp, dv=dv, dR=dq, dtime=dtime, bg=np.asarray(bg, float), ba=np.asarray(ba, float), jacobian=jac,
                covariance=cov)


def sqrt_info(cov):
    L = np.linalg.cholesky(np.linalg.inv(cov))
    return L.T


def imu_residual_raw(pre, g, pr_i, vb_i, pr_j, vb_j):
    Pi, Ri = pr_i[:3], exp_so3(pr_i[3:])
    Pj, Rj = pr_j[:3], exp_so3(pr_j[3:])
    Vi, Vj = vb_i[:3], vb_j[:3]
    dbg = vb_i[3:6] - pre["bg"]
    dba = vb_i[6:9] - pre["ba"]
    dt = pre["dtime"]
    J = pre["jacobian"]
    rP = Ri.T @ (Pj - Pi - Vi * dt - 0.5 * g * dt * dt) - (pre["dp"] + J[0:3, 9:12] @ dbg + J[0:3, 12:15] @ dba)
    dR_dbg = exp_so3(J[3:6, 9:12] @ dbg)
    rPhi = log_so3((pre["dR"] @ dR_dbg).T @ Ri.T @ Rj)
    rV = Ri.T @ (Vj - Vi - g * dt) - (pre["dv"] + J[6:9, 9:12] @ dbg + J[6:9, 12:15] @ dba)
    return np.concatenate([rP, rPhi, rV, vb_j[3:9] - vb_i[3:9]])


def imu_residual(pre, g, pr_i, vb_i, pr_j, vb_j):
    return sqrt_info(pre["covariance"]) @ imu_residual_raw(pre, g, pr_i, vb_i, pr_j, vb_j)


def numeric_jacobian(fun, x, h=1e-6):
    x = np.asarray(x, dtype=np.float64)
    f0 = fun(x)
    J = np.zeros((len(f0), len(x)))
    for k in range(len(x)):
        e = np.zeros_like(x)
        e[k] = h
        J[:, k] = (fun(x + e) - fun(x - e)) / (2 * h)
    return J


def prior_residual(prior, x15):
    dx = np.zeros(15)
    dx[0:3] = x15[0:3] - prior["x0"][0:3]
    dx[3:6] = log_so3(exp_so3(x15[3:6]).T @ exp_so3(prior["x0"][3:6]))
    dx[6:] = x15[6:] - prior["x0"][6:]
    return prior["r0"] + prior["J"] @ dx


def marginalize(A, b, m, eps=1e-8):
    """Schur complement of the first m parameters with eigenvalue thresholding, then the square-root form."""
    Amm = 0.5 * (A[:m, :m] + A[:m, :m].T)
    ev, V = np.linalg.eigh(Amm)
    inv = V @ np.diag(np.where(ev > eps, 1.0 / np.where(ev > eps, ev, 1.0), 0.0)) @ V.T
    Ar = A[m:, m:] - A[m:, :m] @ inv @ A[:m, m:]
    br = b[m:] - A[m:, :m] @ inv @ b[:m]
    ev2, V2 = np.linalg.eigh(0.5 * (Ar + Ar.T))
    S = np.where(ev2 > eps, ev2, 0.0)
    Sinv = np.where(ev2 > eps, 1.0 / np.where(ev2 > eps, ev2, 1.0), 0.0)
    J = np.diag(np.sqrt(S)) @ V2.T
    r = np.diag(np.sqrt(Sinv)) @ V2.T @ br
    return J, r, Ar, br


def dense_trust_region(evaluate, x0, max_iters=10, fixed=False):
    """Ceres 2.1.0 trust-region loop with TRADITIONAL_DOGLEG on dense normal equations (trust_region_minimizer.cc,
    dogleg_strategy.cc; defaults: initial radius 1e4, eta 1e-3, Jacobi scaling, mu 1e-8..1 x10, diagonal clamp
    1e-6 / 1e32, function / gradient / parameter tolerances 1e-6 / 1e-10 / 1e-8).  evaluate(x) -> (H, g, cost).
    Returns (x, trace of accepted costs, iterations, termination)."""
    x = np.array(x0, dtype=np.float64).reshape(-1)
    H, g, cost = evaluate(x)
    scale = 1.0 / (1.0 + np.sqrt(np.diag(H)))
    radius, mu, reuse, invalid, it, term = 1e4, 1e-8, False, 0, 0, 0
    trace = [cost]
    if not fixed and np.abs(g).max() <= 1e-10:
        return x, trace, 0, 1
    x_norm = np.linalg.norm(x)
    gn = grad = diag = None
    alpha = 0.0
    while True:
        if it >= max_iters or radius < 1e-32:
            break
        it += 1
        ok = True
        Hs = H * np.outer(scale, scale)
        if not reuse:
            reuse = True
            diag = np.sqrt(np.clip(np.diag(Hs), 1e-6, 1e32))
            grad = g * scale / diag
            sg = grad / diag
            alpha = (grad @ grad) / (sg @ Hs @ sg)
            ok = False
            while mu < 1.0:
                try:
                    L = np.linalg.cholesky(Hs + mu * np.diag(diag * diag))
                    sol = np.linalg.solve(L.T, np.linalg.solve(L, g * scale))
                    if not np.all(np.isfinite(sol)):
                        raise np.linalg.LinAlgError
                except np.linalg.LinAlgError:
                    mu *= 10.0
                    continue
                gn = -diag * sol
                ok = True
                break
        valid = ok
        if ok:
            gnorm, nnorm = np.linalg.norm(grad), np.linalg.norm(gn)
            if nnorm <= radius:
                step, dnorm = gn.copy(), nnorm
            elif gnorm * alpha >= radius:
                step, dnorm = -(radius / gnorm) * grad, radius
            else:
                b_dot_a = -alpha * (grad @ gn)
                a_sq = (alpha * gnorm) ** 2
                bma_sq = a_sq - 2 * b_dot_a + nnorm ** 2
                c = b_dot_a - a_sq
                d = np.sqrt(c * c + bma_sq * (radius ** 2 - a_sq))
                beta = (d - c) / bma_sq if c <= 0 else (radius ** 2 - a_sq) / (d + c)
                step = (-alpha * (1 - beta)) * grad + beta * gn
                dnorm = np.linalg.norm(step)
            step = step / diag
            model_change = -(step @ (g * scale) + 0.5 * step @ Hs @ step)
            valid = model_change > 0
        if not valid:
            invalid += 1
            if invalid >= 5:       # HandleInvalidStep: FAILURE; Solver::Solve hands back the parameters it was given
                x, term = np.array(x0, dtype=np.float64).reshape(-1), 4
                break
            mu *= 10.0
            reuse = False
            continue
        invalid = 0
        delta = step * scale
        xc = x + delta
        step_norm = np.linalg.norm(delta)
        Hc, gc, cc = evaluate(xc)
        if not fixed:
            if step_norm <= 1e-8 * (x_norm + 1e-8):
                term = 2
                break
            if abs(cost - cc) <= 1e-6 * cost:
                term = 3
                break
        rel = (cost - cc) / model_change
        if rel > 1e-3:
            x, H, g, cost = xc, Hc, gc, cc
            x_norm = np.linalg.norm(x)
            trace.append(cost)
            if not fixed and np.abs(g).max() <= 1e-10:
                term = 1
                break
            if rel < 0.25:
                radius *= 0.5
            if rel > 0.75:
                radius = max(radius, 3.0 * dnorm)
            mu = max(1e-8, 2.0 * mu / 10.0)
            reuse = False
        else:
            radius *= 0.5
            reuse = True
    return x, trace, it, term
