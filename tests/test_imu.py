"""CPU suite for SURVEY section 8(f) rank 1 (host-side IMU factor, marginalization prior, full-window solve) against
the independent numpy restatement in oracle/imu_oracle.py, finite differences and scipy's least-squares solver."""
import os
import sys

import numpy as np
import pytest
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation as Rsc

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import imu_oracle as IO  # noqa: E402

G = np.array([0.0, 0.0, -9.805])


def _trajectory(n_frames, dt_frame=0.1, rate=2000, seed=0):
    """Ground-truth states at the frame times of a smooth motion + the IMU samples between consecutive frames."""
    rng = np.random.default_rng(seed)
    w_body = np.array([0.3, -0.2, 0.5])           # constant body rate
    a_world = np.array([0.8, -0.5, 0.3])          # constant world acceleration
    P0, V0 = np.array([1.0, 2.0, 0.5]), np.array([0.5, 0.1, -0.05])
    R0 = Rsc.from_rotvec([0.1, -0.05, 0.3]).as_matrix()
    states, imus = [], []
    per = int(round(dt_frame * rate))
    h = 1.0 / rate
    for f in range(n_frames):
        t = f * dt_frame
        R = R0 @ Rsc.from_rotvec(w_body * t).as_matrix()
        states.append(dict(P=P0 + V0 * t + 0.5 * a_world * t * t, V=V0 + a_world * t, R=R))
        if f > 0:
            smp = []
            for k in range(per):
                tk = (f - 1) * dt_frame + k * h   # forward Euler in the reference: the sample holds over the next dt
                Rk = R0 @ Rsc.from_rotvec(w_body * tk).as_matrix()
                acc = Rk.T @ (a_world - G) / IO.GNORM
                smp.append(np.concatenate([w_body, acc, [h]]))
            imus.append(np.array(smp))
    return states, imus


def _x15(st, bg=np.zeros(3), ba=np.zeros(3)):
    return np.concatenate([st["P"], Rsc.from_matrix(st["R"]).as_rotvec(), st["V"], bg, ba])


def test_preintegration_matches_numpy_restatement_and_closed_form(M):
    rng = np.random.default_rng(1)
    smp = np.concatenate([rng.normal(0, 0.3, (40, 3)), rng.normal(0, 0.2, (40, 3)) + [0, 0, 1.0],
                          rng.uniform(0.004, 0.006, (40, 1))], axis=1)
    bg, ba = np.array([0.01, -0.02, 0.005]), np.array([0.05, 0.02, -0.03])
    pre = M.imu_preintegrate(smp, bg, ba)
    ref = IO.preintegrate(smp, bg, ba)
    assert np.allclose(np.array(pre.dp), ref["dp"], rtol=0, atol=1e-12)
    assert np.allclose(np.array(pre.dv), ref["dv"], rtol=0, atol=1e-12)
    Rq = Rsc.from_quat(np.array(pre.dq)).as_matrix()
    assert np.allclose(Rq, ref["dR"], atol=1e-12) and abs(pre.dtime - ref["dtime"]) < 1e-15
    assert np.allclose(np.array(pre.jacobian).reshape(15, 15), ref["jacobian"], rtol=1e-10, atol=1e-13)
    assert np.allclose(np.array(pre.covariance).reshape(15, 15), ref["covariance"], rtol=1e-10, atol=1e-20)
    # no rotation, constant specific force: dv = a t, dp = a t^2 / 2 up to the forward-Euler term
    n, h = 100, 0.001
    a = np.array([0.2, -0.1, 1.0])
    smp = np.concatenate([np.zeros((n, 3)), np.tile(a, (n, 1)), np.full((n, 1), h)], axis=1)
    pre = M.imu_preintegrate(smp, np.zeros(3), np.zeros(3))
    T = n * h
    assert np.allclose(np.array(pre.dv), a * IO.GNORM * T, rtol=1e-12)
    assert np.allclose(np.array(pre.dp), 0.5 * a * IO.GNORM * T * T, rtol=1e-12)
    assert np.allclose(np.array(pre.dq), [0, 0, 0, 1])


def test_imu_factor_residual_and_analytic_jacobian(M):
    states, imus = _trajectory(2)
    bg, ba = np.array([0.002, -0.001, 0.003]), np.array([0.02, -0.01, 0.015])
    pre = M.imu_preintegrate(imus[0], bg, ba)
    ref = IO.preintegrate(imus[0], bg, ba)
    rng = np.random.default_rng(3)
    xi = _x15(states[0], bg + rng.normal(0, 1e-3, 3), ba + rng.normal(0, 1e-2, 3)) + np.concatenate([rng.normal(0, 0.02, 9), np.zeros(6)])
    xj = _x15(states[1], bg + rng.normal(0, 1e-3, 3), ba + rng.normal(0, 1e-2, 3)) + np.concatenate([rng.normal(0, 0.02, 9), np.zeros(6)])
    r, J = M.imu_factor(pre, G, xi[:6], xi[6:], xj[:6], xj[6:])
    r_ref = IO.imu_residual(ref, G, xi[:6], xi[6:], xj[:6], xj[6:])
    assert np.allclose(r, r_ref, rtol=1e-7, atol=1e-7 * np.abs(r_ref).max())

    def f(z):
        return M.imu_factor(pre, G, z[:6], z[6:15], z[15:21], z[21:30], jac=False)[0]

    Jn = IO.numeric_jacobian(f, np.concatenate([xi, xj]), h=1e-6)
    assert np.abs(J - Jn).max() < 2e-6 * max(1.0, np.abs(Jn).max())
    # the numpy restatement differentiated numerically agrees as well (independent residual code)
    Jo = IO.numeric_jacobian(lambda z: IO.imu_residual(ref, G, z[:6], z[6:15], z[15:21], z[21:30]), np.concatenate([xi, xj]), h=1e-6)
    assert np.abs(J - Jo).max() < 1e-5 * max(1.0, np.abs(Jo).max())
    # at the generating states the unweighted residual is the forward-Euler discretisation error only
    x0, x1 = _x15(states[0]), _x15(states[1])
    pre0 = IO.preintegrate(imus[0], np.zeros(3), np.zeros(3))
    raw = IO.imu_residual_raw(pre0, G, x0[:6], x0[6:], x1[:6], x1[6:])
    assert np.abs(raw).max() < 2e-3


def _lidar(W, seed):
    """Per frame a positive definite 6x6 information and a measured pose: a linear-Gaussian stand-in for the device."""
    rng = np.random.default_rng(seed)
    out = []
    for f in range(W):
        A = rng.normal(0, 1.0, (40, 6)) * np.array([30, 30, 30, 60, 60, 60])
        out.append(A)
    return out


def _records(M, lid, meas, x):
    recs = []
    for f, A in enumerate(lid):
        d = x[f][:6] - meas[f]
        H = A.T @ A
        recs.append(M.pack_record(H, H @ d, 0.5 * d @ H @ d))
    return np.stack(recs)


def _solve(M, W, lid, meas, pres, x0, prior=None, max_iters=50):
    fw = M.FullWindowSolver(W, max_iters=max_iters, fixed=False, huber=0.0, w_tan=3e-4)
    for f in range(1, W):
        fw.set_imu(f, pres[f - 1], G)
    if prior is not None:
        fw.set_prior(prior)
    x = x0.copy()
    for _ in range(1000):
        done, x = fw.step(_records(M, lid, meas, x), x)
        if done:
            return x, fw
    raise AssertionError("full-window solver did not terminate")


def _stack_residuals(lid, meas, pres_np, z, W, prior=None):
    x = z.reshape(W, 15)
    res = [lid[f] @ (x[f][:6] - meas[f]) for f in range(W)]
    for f in range(1, W):
        res.append(IO.imu_residual(pres_np[f - 1], G, x[f - 1][:6], x[f - 1][6:], x[f][:6], x[f][6:]))
    if prior is not None:
        res.append(IO.prior_residual(prior, x[0]))
    return np.concatenate(res)


def _problem(M, W, seed):
    states, imus = _trajectory(W, seed=seed)
    bg, ba = np.array([0.001, -0.002, 0.0015]), np.array([0.01, 0.02, -0.01])
    pres = [M.imu_preintegrate(s, bg, ba) for s in imus]
    pres_np = [IO.preintegrate(s, bg, ba) for s in imus]
    rng = np.random.default_rng(seed + 100)
    truth = np.stack([_x15(s, bg, ba) for s in states])
    meas = [truth[f][:6] + rng.normal(0, 0.01, 6) for f in range(W)]
    lid = _lidar(W, seed + 200)
    x0 = truth + np.concatenate([rng.normal(0, 0.03, (W, 9)), rng.normal(0, 1e-3, (W, 6))], axis=1)
    return pres, pres_np, meas, lid, x0


def test_full_window_normal_equations_and_gauss_newton(M):
    """H, g, cost assembled by the library = J^T J, J^T r, r^T r / 2 of the independently written residual stack;
    Gauss-Newton on the library's equations reaches the optimum scipy finds."""
    W = 4
    pres, pres_np, meas, lid, x0 = _problem(M, W, 2)
    fw = M.FullWindowSolver(W)
    for f in range(1, W):
        fw.set_imu(f, pres[f - 1], G)
    fun = lambda z: _stack_residuals(lid, meas, pres_np, z, W)
    H, g, c = fw.normal_equations(_records(M, lid, meas, x0), x0)
    r = fun(x0.reshape(-1))
    J = IO.numeric_jacobian(fun, x0.reshape(-1), h=1e-6)
    assert abs(c - 0.5 * r @ r) < 1e-9 * c
    assert np.allclose(g, J.T @ r, rtol=1e-6, atol=1e-6 * np.abs(g).max())
    assert np.allclose(H, J.T @ J, rtol=1e-5, atol=1e-6 * np.abs(H).max()) and np.allclose(H, H.T)
    x = x0.copy()
    for _ in range(6):
        H, g, c = fw.normal_equations(_records(M, lid, meas, x), x)
        d = np.sqrt(np.diag(H))
        x = x - (np.linalg.solve(H / np.outer(d, d), g / d) / d).reshape(W, 15)
    ref = least_squares(fun, x0.reshape(-1), method="lm", xtol=1e-14, ftol=1e-14, gtol=1e-14)
    c_final = fw.normal_equations(_records(M, lid, meas, x), x)[2]
    assert abs(c_final - ref.cost) < 1e-6 * ref.cost and np.abs(x.reshape(-1) - ref.x).max() < 1e-5


@pytest.mark.parametrize("fixed", [False, True])
def test_full_window_trust_region_matches_numpy_restatement(M, fixed):
    """The library's trust-region loop against oracle/imu_oracle.dense_trust_region fed with the oracle's own normal
    equations (numerically differentiated residuals): same accepted iterates, same termination."""
    W = 3
    pres, pres_np, meas, lid, x0 = _problem(M, W, 6)
    fun = lambda z: _stack_residuals(lid, meas, pres_np, z, W)

    def evaluate(z):
        r = fun(z)
        J = IO.numeric_jacobian(fun, z, h=1e-6)
        return J.T @ J, J.T @ r, 0.5 * r @ r

    xo, trace, it, term = IO.dense_trust_region(evaluate, x0, max_iters=10, fixed=fixed)
    fw = M.FullWindowSolver(W, max_iters=10, fixed=fixed, huber=0.0, w_tan=3e-4)
    for f in range(1, W):
        fw.set_imu(f, pres[f - 1], G)
    x = x0.copy()
    for _ in range(100):
        done, x = fw.step(_records(M, lid, meas, x), x)
        if done:
            break
    sm = fw.summary()
    assert done and sm.iterations == it and sm.termination == term and sm.successful == len(trace) - 1
    assert abs(sm.final_cost - trace[-1]) < 1e-6 * trace[-1] and sm.final_cost < 1e-6 * sm.initial_cost
    assert np.abs(x.reshape(-1) - xo).max() < 1e-6
    # the IMU factors matter: without them the velocity / bias blocks would not move at all
    assert np.abs(x[:, 6:9] - x0[:, 6:9]).max() > 1e-3


def test_marginalization_prior_matches_numpy_and_preserves_the_estimate(M):
    W = 3
    states, imus = _trajectory(W, seed=4)
    bg, ba = np.zeros(3), np.zeros(3)
    pres = [M.imu_preintegrate(s, bg, ba) for s in imus]
    pres_np = [IO.preintegrate(s, bg, ba) for s in imus]
    rng = np.random.default_rng(9)
    truth = np.stack([_x15(s) for s in states])
    meas = [truth[f][:6] + rng.normal(0, 0.01, 6) for f in range(W)]
    lid = _lidar(W, 11)
    x0 = truth + np.concatenate([rng.normal(0, 0.02, (W, 9)), np.zeros((W, 6))], axis=1)
    x_full, fw = _solve(M, W, lid, meas, pres, x0)
    rec0 = _records(M, lid, meas, x_full)[0]
    prior = fw.marginalize(rec0, x_full)
    Jp, r0, xk = np.array(prior.J).reshape(15, 15), np.array(prior.r0), np.array(prior.x0)
    assert np.array_equal(xk, x_full[1])
    # numpy: assemble A, b of (frame 0 | frame 1) from numerically differentiated oracle residuals
    def imu01(z):
        return IO.imu_residual(pres_np[0], G, z[:6], z[6:15], z[15:21], z[21:30])
    z = np.concatenate([x_full[0], x_full[1]])
    Ji = IO.numeric_jacobian(imu01, z, h=1e-6)
    A = Ji.T @ Ji
    b = Ji.T @ imu01(z)
    H0 = lid[0].T @ lid[0]
    A[:6, :6] += H0
    b[:6] += H0 @ (x_full[0][:6] - meas[0])
    Jn, rn, Ar, br = IO.marginalize(A, b, 15)
    assert np.allclose(Jp.T @ Jp, Jn.T @ Jn, rtol=1e-5, atol=1e-5 * np.abs(Ar).max())      # J^T J = reduced information
    assert np.allclose(Jp.T @ r0, Jn.T @ rn, rtol=1e-5, atol=1e-5 * np.abs(br).max())      # J^T r0 = reduced gradient
    # sliding the window: at the linearisation point the normal equations of (frames 1, 2 + prior) are the Schur
    # complement of the joint (0, 1, 2) equations with frame 0 eliminated
    H3, g3, _ = fw.normal_equations(_records(M, lid, meas, x_full), x_full)
    inv = np.linalg.inv(H3[:15, :15])
    Hs = H3[15:, 15:] - H3[15:, :15] @ inv @ H3[:15, 15:]
    gs = g3[15:] - H3[15:, :15] @ inv @ g3[:15]
    fw2 = M.FullWindowSolver(2, max_iters=10)
    fw2.set_imu(1, pres[1], G)
    fw2.set_prior(prior)
    H2, g2, c2 = fw2.normal_equations(_records(M, lid[1:], meas[1:], x_full[1:]), x_full[1:])
    assert np.allclose(H2, Hs, rtol=1e-6, atol=1e-7 * np.abs(Hs).max())
    assert np.allclose(g2, gs, rtol=1e-5, atol=1e-6 * np.abs(gs).max())
    x_slid = x_full[1:] + rng.normal(0, 0.005, (2, 15)) * np.r_[np.ones(9), np.zeros(6)]
    for _ in range(100):
        done, x_slid = fw2.step(_records(M, lid[1:], meas[1:], x_slid), x_slid)
        if done:
            break
    assert done and fw2.summary().final_cost < fw2.summary().initial_cost
    # and a second marginalization that consumes the prior runs and stays finite / symmetric positive semi-definite
    prior2 = fw2.marginalize(_records(M, lid[1:], meas[1:], x_slid)[0], x_slid)
    J2 = np.array(prior2.J).reshape(15, 15)
    assert np.all(np.isfinite(J2)) and np.linalg.eigvalsh(J2.T @ J2).min() > -1e-9
    # the prior residual function is the reference's (rotation part log(exp(x)^-1 exp(x0)))
    xt = x_full[1] + rng.normal(0, 0.01, 15)
    fwp = M.FullWindowSolver(1, max_iters=1)
    pr = dict(J=Jp, r0=r0, x0=xk)
    assert np.allclose(IO.prior_residual(pr, xt)[:3], r0[:3] + (Jp @ np.r_[xt[:3] - xk[:3], IO.log_so3(IO.exp_so3(xt[3:6]).T @ IO.exp_so3(xk[3:6])), xt[6:] - xk[6:]])[:3])


def test_golden_imu_fixture(M):
    """tests/golden/imu_map_small.npz pins the numpy oracle; the product is held to the same numbers."""
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "imu_map_small.npz"))
    ref = IO.preintegrate(g["imu_samples"], g["bg"], g["ba"])
    for k in ("dp", "dv", "dR", "jacobian", "covariance"):
        assert np.allclose(ref[k], g[k], rtol=1e-12, atol=1e-15), k
    pre = M.imu_preintegrate(g["imu_samples"], g["bg"], g["ba"])
    assert np.allclose(np.array(pre.dp), g["dp"], atol=1e-12) and np.allclose(np.array(pre.dv), g["dv"], atol=1e-12)
    assert np.allclose(Rsc.from_quat(np.array(pre.dq)).as_matrix(), g["dR"], atol=1e-12)
    assert np.allclose(np.array(pre.covariance).reshape(15, 15), g["covariance"], rtol=1e-9, atol=1e-20)
    xi, xj = g["xi"], g["xj"]
    assert np.allclose(IO.imu_residual(ref, g["gravity"], xi[:6], xi[6:], xj[:6], xj[6:]), g["residual"], rtol=1e-9, atol=1e-9)
    r, J = M.imu_factor(pre, g["gravity"], xi[:6], xi[6:], xj[:6], xj[6:])
    assert np.allclose(r, g["residual"], rtol=1e-6, atol=1e-6 * np.abs(g["residual"]).max())
    assert np.abs(J - g["residual_jacobian"]).max() < 1e-5 * np.abs(g["residual_jacobian"]).max()
    Jm, rm, _, _ = IO.marginalize(g["marg_A"], g["marg_b"], 15)
    assert np.allclose(Jm.T @ Jm, g["marg_JtJ"], rtol=1e-9, atol=1e-9 * np.abs(g["marg_JtJ"]).max())
    assert np.allclose(Jm.T @ rm, g["marg_Jtr"], rtol=1e-8, atol=1e-9 * np.abs(g["marg_Jtr"]).max())


_GLOO_FULLWINDOW = r"""
import importlib, os, sys
import numpy as np
import torch
import torch.distributed as dist
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import test_imu as T
M = importlib.import_module("multi-modal-loam_amd")
dist.init_process_group("gloo")
rank, W = dist.get_rank(), dist.get_world_size()
Wf = 2 * W                      # two frames of the window per rank
pres, pres_np, meas, lid, x0 = T._problem(M, Wf, 21)
mine = [2 * rank, 2 * rank + 1]
fw = M.FullWindowSolver(Wf, max_iters=10, fixed=False, huber=0.0, w_tan=3e-4)
for f in range(1, Wf):
    fw.set_imu(f, pres[f - 1], T.G)          # O(W) host data, replicated on every rank
x = x0.copy()
for _ in range(100):
    full = T._records(M, lid, meas, x)                                     # stand-in for the device linearisation ...
    local = torch.from_numpy(np.ascontiguousarray(full[mine]).reshape(-1))  # ... of THIS rank's frames only
    out = [torch.zeros(2 * 32, dtype=torch.float64) for _ in range(W)]
    dist.all_gather(out, local)                                            # the 32-double-per-frame exchange
    rec = np.concatenate([t.numpy().reshape(2, 32) for t in out])
    done, x = fw.step(rec, x)
    if done:
        break
chk = [torch.zeros(15 * Wf, dtype=torch.float64) for _ in range(W)]
dist.all_gather(chk, torch.from_numpy(x.reshape(-1).copy()))
assert all(torch.equal(chk[0], c) for c in chk)
if rank == 0:
    xs, fws = T._solve(M, Wf, lid, meas, pres, x0, max_iters=10)           # single-process reference
    assert np.array_equal(xs, x) and fw.summary().iterations == fws.summary().iterations
    print("GLOO_FULLWINDOW_OK", fw.summary().iterations)
dist.destroy_process_group()
"""


def test_full_window_two_ranks_gloo(tmp_path):
    """world_size 2, gloo: every rank contributes the lidar records of its own frames, all ranks run the 15 W host
    solve (IMU factors included) redundantly and end with bit-identical states."""
    import subprocess
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_FULLWINDOW)
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "GLOO_FULLWINDOW_OK" in out.stdout
