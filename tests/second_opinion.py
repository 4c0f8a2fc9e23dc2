"""A second, independently structured restatement of the two riskiest pieces of the oracle, used ONLY to cross-check the
oracle on the CPU (tests/test_oracle.py).  The reference ships no tests and cannot be built here, so parity stays
"unpinned"; what these give is protection against a transcription slip in oracle/feature.cpp / oracle/estimate.cpp:

  detect_feature_points_py   feature_extraction::detectFeaturePoints (mm-loam/src/unionFeatureExtract.cpp:341-844), written
                             from the reference text as array passes + explicit state-machine loops in numpy float32 /
                             Python float arithmetic (the oracle is a line-by-line C++ restatement)
  ceres_trust_region_py      the control flow of Ceres 2.1.0's TrustRegionMinimizer + DoglegStrategy(TRADITIONAL_DOGLEG)
                             in Ceres' own terms -- stacked residuals r, dense Jacobian J, loss corrector applied to the
                             rows -- where the oracle and the device work from (H, g, cost) per frame

Conventions shared with the oracle because they describe third-party behaviour, not this code: uninitialised reference
arrays read as zero; Eigen's 3-vector dot is (x0 y0 + x1 y1) + x2 y2 and normalize() divides by the norm unless it is
zero; sqrt on a float argument rounds like the float sqrt.
"""
import math

import numpy as np

F = np.float32


def _norm3(v):
    return math.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])


def _dot3(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def _unit(v):
    z = _dot3(v, v)
    if z > 0.0:
        n = math.sqrt(z)
        return (v[0] / n, v[1] / n, v[2] / n)
    return v


def _cosine(a, b):
    """a.dot(b) / (a.norm() * b.norm()) in double; nan / inf propagate exactly as IEEE division does."""
    den = _norm3(a) * _norm3(b)
    num = _dot3(a, b)
    if den == 0.0:
        return math.nan if num == 0.0 or math.isnan(num) else math.copysign(math.inf, num)
    return num / den


def detect_feature_points_py(pts):
    """pts: (n, 4) float32 finite x, y, z, intensity of ONE scan line.  Returns (sharp idx, flat idx, flags)."""
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 4)
    n = len(pts)
    flags = np.zeros(max(n, 1), np.int64)
    if n < 11:   # neither the stencil nor any later loop has an index to visit; the partition loop then walks indices the
        return np.zeros(0, np.int32), np.zeros(0, np.int32), flags[:n].astype(np.int32)   # reference never initialised
    x, y, z, inten = pts[:, 0], pts[:, 1], pts[:, 2], pts[:, 3]
    xd, yd, zd = x.astype(np.float64), y.astype(np.float64), z.astype(np.float64)
    TH_FAR = F(50.0)
    TH_FLAT = F(0.02)

    # ---- stencil (:407-451): depth, grazing test against both neighbours, window 2 or 3, curvature, reflectivity ----
    sq = (x * x + y * y) + z * z                     # float, left to right
    depth = np.sqrt(sq).astype(np.float32)
    curv = np.zeros(n, np.float32)
    refl = np.zeros(n, np.float32)
    angle_flag = np.zeros(n, bool)
    win = np.full(n, 2, np.int64)
    for i in range(5, n - 5):
        cur = (xd[i], yd[i], zd[i])
        dl = (xd[i - 1] - cur[0], yd[i - 1] - cur[1], zd[i - 1] - cur[2])
        dn = (xd[i + 1] - cur[0], yd[i + 1] - cur[1], zd[i + 1] - cur[2])
        a_last, a_next = _cosine(dl, cur), _cosine(dn, cur)
        grazing = abs(a_last) > 0.966 and abs(a_next) > 0.966
        w = 2 if (depth[i] > TH_FAR or grazing) else 3
        win[i] = w
        angle_flag[i] = grazing
        dx = dy = dz = F(0)
        dr = F(-2 * w) * inten[i]
        for j in range(1, w + 1):
            dx = dx + (x[i - j] + x[i + j])
            dy = dy + (y[i - j] + y[i + j])
            dz = dz + (z[i - j] + z[i + j])
            dr = dr + (inten[i - j] + inten[i + j])
        dx = dx - F(2 * w) * x[i]
        dy = dy - F(2 * w) * y[i]
        dz = dz - F(2 * w) * z[i]
        curv[i] = (dx * dx + dy * dy) + dz * dz
        refl[i] = dr
    last_w = int(win[n - 6])                         # the value thNumCurvSize keeps after the loop (:492,505 read it)

    # gap between consecutive points, squared, float (:494-500, 507-513)
    def gap2(a, b):
        ex, ey, ez = x[a] - x[b], y[a] - y[b], z[a] - z[b]
        return (ex * ex + ey * ey) + ez * ez

    # ---- 50 partitions: two stable ascending orders, greedy flat picks, promotion (:453-539) ----
    s0, s1 = 5, n - 6
    for j in range(50):
        sp = s0 + ((s1 - s0) * j) // 50              # n >= 11 keeps the products non-negative: floor == C truncation
        ep = s0 + ((s1 - s0) * (j + 1)) // 50 - 1
        idx = list(range(sp, ep + 1))
        # insertion sort with a strict "<" swap == stable ascending order (ties keep index order); NaN never occurs in
        # these two arrays for finite input, so a keyed stable sort is the same permutation
        by_curv = sorted(idx, key=lambda k: float(curv[k]))
        by_refl = sorted(idx, key=lambda k: float(refl[k]))
        for ind in by_curv:
            if flags[ind] != 0:
                continue
            lim = ((TH_FLAT * depth[ind]) * TH_FLAT) * depth[ind]
            if curv[ind] < lim:
                flags[ind] = 3
                far = depth[ind] > TH_FAR
                for l in range(1, last_w + 1):
                    if float(gap2(ind + l, ind + l - 1)) > 0.02 or far:
                        break
                    flags[ind + l] = 1
                for l in range(-1, -last_w - 1, -1):
                    if float(gap2(ind + l, ind + l + 1)) > 0.02 or far:
                        break
                    flags[ind + l] = 1
        n_flat = 1
        n_refl = 1
        for pos in range(len(idx)):
            ind = by_curv[pos]
            if (flags[ind] == 3 and n_flat <= 1) or (flags[ind] == 3 and depth[ind] > TH_FAR) or angle_flag[ind]:
                n_flat += 1
                flags[ind] = 2
            k = by_refl[pos]
            lim = 0.7 * float(TH_FLAT) * float(depth[k]) * float(TH_FLAT) * float(depth[k])    # 0.7 promotes the product to double
            if float(curv[k]) < lim and n_refl <= 3 and float(refl[k]) > 20.0:
                n_refl += 1
                flags[k] = 300

    # ---- plane-intersection corners, stride 4 after a flat right half (:543-650) ----
    i = 5
    while i < n - 5:
        d = depth[i]

        def half(sgn):
            ax = ((x[i + 4 * sgn] + x[i + 3 * sgn]) - F(4) * x[i + 2 * sgn]) + x[i + sgn] + x[i]
            ay = ((y[i + 4 * sgn] + y[i + 3 * sgn]) - F(4) * y[i + 2 * sgn]) + y[i + sgn] + y[i]
            az = ((z[i + 4 * sgn] + z[i + 3 * sgn]) - F(4) * z[i + 2 * sgn]) + z[i + sgn] + z[i]
            return (ax * ax + ay * ay) + az * az
        left_flat = half(-1) < TH_FLAT * d
        right_flat = half(+1) < TH_FLAT * d
        if left_flat and right_flat:
            nl = [0.0, 0.0, 0.0]
            nr = [0.0, 0.0, 0.0]
            for k in range(1, 5):
                t = _unit((float(x[i - k] - x[i]), float(y[i - k] - y[i]), float(z[i - k] - z[i])))
                nl = [nl[c] + (k / 10.0) * t[c] for c in range(3)]
            for k in range(1, 5):
                t = _unit((float(x[i + k] - x[i]), float(y[i + k] - y[i]), float(z[i + k] - z[i])))
                nr = [nr[c] + (k / 10.0) * t[c] for c in range(3)]
            cc = abs(_cosine(nl, nr))
            span_l = _norm3((float(x[i - 4] - x[i]), float(y[i - 4] - y[i]), float(z[i - 4] - z[i])))
            span_r = _norm3((float(x[i + 4] - x[i]), float(y[i + 4] - y[i]), float(z[i + 4] - z[i])))
            if cc < 0.5 and span_l > 0.05 and span_r > 0.05:
                flags[i] = 150
        i += 4 if right_flat else 1

    # ---- break points (:651-806) ----
    def fdist(a, b):
        ex, ey, ez = x[a] - x[b], y[a] - y[b], z[a] - z[b]
        return np.sqrt((ex * ex + ey * ey) + ez * ez).astype(np.float32)
    for i in range(5, n - 5):
        d_right, d_left = fdist(i + 1, i), fdist(i - 1, i)
        if abs(d_right - d_left) > F(1):
            side = -1 if d_right > d_left else 1      # the surface the point belongs to lies on the nearer side
            sv = (float(x[i + side] - x[i]), float(y[i + side] - y[i]), float(z[i + side] - z[i]))
            cc = abs(_cosine(sv, (xd[i], yd[i], zd[i])))
            if cc < 0.95:
                dr, dl = depth[i + 1], depth[i - 1]
                if side == -1:
                    if dr > dl or dr == 0:
                        flags[i] = 100
                else:
                    if dr < dl or dl == 0:
                        flags[i] = 100
        if flags[i] == 100:
            nf = [0.0, 0.0, 0.0]
            nb = [0.0, 0.0, 0.0]
            for k in range(1, 4):
                if depth[i - k] < F(1):
                    continue
                t = _unit((float(x[i - k] - x[i]), float(y[i - k] - y[i]), float(z[i - k] - z[i])))
                nf = [nf[c] + (k / 6.0) * t[c] for c in range(3)]
            for k in range(1, 4):
                if depth[i - k] < F(1):               # the reference tests the point at i - k here as well (:782-784)
                    continue
                t = _unit((float(x[i + k] - x[i]), float(y[i + k] - y[i]), float(z[i + k] - z[i])))
                nb = [nb[c] + (k / 6.0) * t[c] for c in range(3)]
            cc = abs(_cosine(nf, nb))
            if not (cc < 0.95):
                flags[i] = 101

    # ---- emit (:818-842) ----
    sharp, flat = [], []
    for i in range(5, n - 5):
        if sq[i] < F(1):
            continue
        if flags[i] == 2:
            flat.append(i)
        elif flags[i] == 100 or flags[i] == 150:
            sharp.append(i)
    return np.array(sharp, np.int32), np.array(flat, np.int32), flags[:n].astype(np.int32)


# ---------------------------------------------------------------------------------------------------------------------
def _huber(s, a):
    """ceres::HuberLoss::Evaluate: rho, rho', rho''."""
    b = a * a
    if s > b:
        r = math.sqrt(s)
        return 2.0 * a * r - b, max(np.finfo(np.float64).tiny, a / r), -a / (2.0 * r * s)   # rho'' = -rho' / (2 s)
    return s, 1.0, 0.0


def _corrected_block(r, J, huber_delta):
    """One residual block through ceres::Corrector (corrector.cc).  Returns cost, corrected r, corrected J."""
    r = np.atleast_1d(np.asarray(r, dtype=np.float64))
    J = np.atleast_2d(np.asarray(J, dtype=np.float64))
    sq = float(r @ r)
    if huber_delta <= 0:
        return 0.5 * sq, r, J
    rho0, rho1, rho2 = _huber(sq, huber_delta)
    sqrt_rho1 = math.sqrt(rho1)
    if sq == 0.0 or rho2 <= 0.0:
        scaling, alpha_sq = sqrt_rho1, 0.0
    else:
        D = 1.0 + 2.0 * sq * rho2 / rho1
        alpha = 1.0 - math.sqrt(D)
        scaling, alpha_sq = sqrt_rho1 / (1.0 - alpha), alpha / sq
    Jc = sqrt_rho1 * (J - alpha_sq * np.outer(r, r @ J)) if alpha_sq != 0.0 else sqrt_rho1 * J
    return 0.5 * rho0, scaling * r, Jc


def ceres_trust_region_py(blocks_at, x0, max_iters=10, huber_delta=0.0, fixed=False):
    """TrustRegionMinimizer::Minimize with DoglegStrategy(TRADITIONAL_DOGLEG), Jacobi scaling and a dense normal-equation
    solve (what DENSE_SCHUR amounts to when every parameter block is eliminated jointly), Ceres 2.1.0 defaults.
    blocks_at(x) -> list of (r, J) per residual block, J against the FULL parameter vector.  Returns
    (x, iterates after every iteration, iterations, termination) with terminations 0 max iterations, 1 gradient,
    2 parameter, 3 function tolerance, 4 failure."""
    x = np.array(x0, dtype=np.float64).reshape(-1)
    x_entry = x.copy()

    def evaluate(xx):
        cost, rs, Js = 0.0, [], []
        for r, J in blocks_at(xx):
            c, rc, Jc = _corrected_block(r, J, huber_delta)
            cost += c
            rs.append(rc)
            Js.append(Jc)
        return cost, np.concatenate(rs), np.vstack(Js)

    cost, r, J = evaluate(x)
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))              # jacobi_scaling, computed once
    J = J * scale
    x_norm = np.linalg.norm(x)
    trace = []
    # gradient tolerance is tested on the UNSCALED gradient
    def grad_max():
        return np.abs((J / scale).T @ r).max()
    if not fixed and grad_max() <= 1e-10:
        return x, trace, 0, 1
    radius, mu, reuse = 1e4, 1e-8, False
    min_diag, max_diag = 1e-6, 1e32
    invalid = 0
    it = 0
    term = 0
    diag = gn = grad = None
    alpha = dogleg_norm = 0.0
    while True:
        # FinalizeIterationAndCheckIfMinimizerCanContinue: iterations, then gradient, then radius
        if it >= max_iters:
            break
        if not fixed and it > 0 and grad_max() <= 1e-10:
            term = 1
            break
        if radius < 1e-32:
            break
        it += 1
        # ---- DoglegStrategy::ComputeStep ----
        ok = True
        if not reuse:
            reuse = True
            diag = np.sqrt(np.clip((J * J).sum(0), min_diag, max_diag))
            grad = (J.T @ r) / diag                                 # ComputeGradient: gradient of the rescaled problem
            Jg = J @ (grad / diag)
            alpha = float(grad @ grad) / float(Jg @ Jg)             # ComputeCauchyPoint
            ok = False
            while mu < 1.0:                                         # ComputeGaussNewtonStep
                A = J.T @ J + mu * np.diag(diag * diag)
                try:
                    L = np.linalg.cholesky(A)
                    sol = np.linalg.solve(L.T, np.linalg.solve(L, J.T @ r))
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
            if nnorm <= radius:                                     # ComputeTraditionalDoglegStep, case 1
                step, dogleg_norm = gn.copy(), nnorm
            elif gnorm * alpha >= radius:                           # case 2
                step, dogleg_norm = -(radius / gnorm) * grad, radius
            else:                                                   # case 3
                b_dot_a = -alpha * float(grad @ gn)
                a_sq = (alpha * gnorm) ** 2
                bma_sq = a_sq - 2.0 * b_dot_a + nnorm ** 2
                c = b_dot_a - a_sq
                d = math.sqrt(c * c + bma_sq * (radius ** 2 - a_sq))
                beta = (d - c) / bma_sq if c <= 0 else (radius ** 2 - a_sq) / (d + c)
                step = (-alpha * (1.0 - beta)) * grad + beta * gn
                dogleg_norm = np.linalg.norm(step)
            step = step / diag
            model = J @ step                                        # model_residuals
            model_cost_change = -float(model @ (r + model / 2.0))
            valid = model_cost_change > 0.0
        if not valid:                                               # HandleInvalidStep
            invalid += 1
            if invalid >= 5:
                x, term = x_entry.copy(), 4
                trace.append(x.copy())
                break
            mu *= 10.0                                              # StepIsInvalid
            reuse = False
            trace.append(x.copy())
            continue
        invalid = 0
        delta = step * scale
        xc = x + delta
        step_norm = np.linalg.norm(delta)
        if not fixed and step_norm <= 1e-8 * (x_norm + 1e-8):       # ParameterToleranceReached
            term = 2
            trace.append(x.copy())
            break
        cost_c, r_c, J_c = evaluate(xc)
        if not fixed and abs(cost - cost_c) <= 1e-6 * cost:         # FunctionToleranceReached
            term = 3
            trace.append(x.copy())
            break
        rel = (cost - cost_c) / model_cost_change                   # TrustRegionStepEvaluator, monotonic steps
        if rel > 1e-3:                                              # HandleSuccessfulStep
            x, cost, r, J = xc, cost_c, r_c, J_c * scale
            x_norm = np.linalg.norm(x)
            trace.append(x.copy())
            if rel < 0.25:                                          # DoglegStrategy::StepAccepted
                radius *= 0.5
            if rel > 0.75:
                radius = max(radius, 3.0 * dogleg_norm)
            mu = max(1e-8, 2.0 * mu / 10.0)
            reuse = False
        else:                                                       # HandleUnsuccessfulStep / StepRejected
            radius *= 0.5
            reuse = True
            trace.append(x.copy())
    return x, trace, it, term
