"""Host-side mirror of Estimator::EstimateLidarPose (Estimator.cpp:967-1140) for the live 1-frame mode: down-sample,
Estimate against the local map, hand the pose back, apply the key-scan rule and grow the local map.  Every stage
that touches points runs on the device through the C-ABI (mml_downsample, mml_estimate, mml_map_increment_local);
this file only carries the control flow and the 4x4 / quaternion bookkeeping the reference does in Eigen.
"""
import numpy as np


def _quat_to_matrix(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class LidarOdometry:
    """One Estimator instance: `ctx` owns the scan slots and the maps.  lidar_mode 1 = Horizon, 2 = Velodyne
    (the fused cloud is processed as mode 2, unionPoseEstimation.cpp:872)."""

    def __init__(self, ctx, exTlb=None, lidar_mode=2, max_outer=5, inner_iters=10):
        self.ctx = ctx
        self.exTlb = np.eye(4) if exTlb is None else np.asarray(exTlb, dtype=np.float64)
        self.exRbl = self.exTlb[:3, :3].T.copy()                      # :973
        self.exPbl = -1.0 * self.exRbl @ self.exTlb[:3, 3]            # :974
        self.lidar_mode = lidar_mode
        self.max_outer, self.inner_iters = max_outer, inner_iters
        self.last_velo_update_pose = np.array([-1.0, -1.0, -1.0])     # Estimator.h:339-340
        self.last_hori_update_pose = np.array([-1.0, -1.0, -1.0])
        self.n_corner_local = 0
        self.n_surf_local = 0
        self.fail_detected = False
        self.key_scans = 0
        ctx.map_local_reset()

    def transform_to_be_mapped(self, P, Q):
        T = np.eye(4)
        R = _quat_to_matrix(Q)
        T[:3, :3] = R @ self.exRbl
        T[:3, 3] = R @ self.exPbl + P
        return T

    def estimate_lidar_pose(self, slot, P, Q):
        """P (3), Q (x, y, z, w): predicted body pose of the scan in `slot` (already extracted and undistorted).
        Returns the estimated (P, Q) and whether the local map grew."""
        ctx = self.ctx
        P = np.asarray(P, dtype=np.float64).copy()
        Q = np.asarray(Q, dtype=np.float64).copy()
        T = self.transform_to_be_mapped(P, Q)                          # :975-977
        corner_cnt = ctx.scan_info(slot).fused_corner_num              # :990-996
        ctx.downsample(slot, 1)                                        # :1013-1024
        is_degenerate = False
        if self.n_corner_local > 0 and self.n_surf_local > 100:        # :1032-1035 (local-map half of the gate)
            Pn, Qn, info = ctx.estimate(slot, 1, self.exTlb, P[None], Q[None], self.max_outer, self.inner_iters)
            P, Q = Pn[0], Qn[0]
            is_degenerate = bool(info[0].is_degenerate)
        if (self.lidar_mode == 1 and not is_degenerate and corner_cnt > 100) or (self.lidar_mode == 2 and corner_cnt > 50):
            T = self.transform_to_be_mapped(P, Q)                      # :1041-1049
        else:                                                          # :1050-1066: keep the predicted x / y, old z
            T[:3, 3] = [P[0], P[1], T[2, 3]]
        grew = False
        if not is_degenerate:                                          # :1070-1136
            cur = T[:3, 3].copy()
            # both modes compare with last_velo_update_pose (:1082, :1119); mode 1 writes last_hori_update_pose (:1115), which
            # nothing reads -- the reference's bookkeeping as it is
            d = self.last_velo_update_pose - cur
            dis = float(np.float32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])) if self.lidar_mode == 2 else float(d @ d)
            if dis >= 0.5 and self.lidar_mode in (1, 2):
                self.n_corner_local, self.n_surf_local = ctx.map_increment_local(slot, T)   # :1112, :1125-1130
                if self.lidar_mode == 2:
                    self.last_velo_update_pose = cur
                else:
                    self.last_hori_update_pose = cur
                self.key_scans += 1
                grew = True
        self.fail_detected = is_degenerate                              # :1139
        return P, Q, grew


def _rotvec_from_quat(q):
    """Sophus::SO3d(Q).log() (vector2double, Estimator.cpp:942)."""
    x, y, z, w = q
    n = np.sqrt(x * x + y * y + z * z)
    if n < 1e-10:
        k = 2.0 / w - (2.0 / 3.0) * n * n / (w * w * w)
    elif abs(w) < 1e-10:
        k = (np.pi if w > 0 else -np.pi) / n
    else:
        k = 2.0 * np.arctan(n / w) / n
    return np.array([k * x, k * y, k * z])


def _quat_from_rotvec(phi):
    """Sophus::SO3d::exp(phi).unit_quaternion() (double2vector, Estimator.cpp:958)."""
    th = np.linalg.norm(phi)
    if th * th < 1e-20:
        im, re = 0.5 - th * th / 48.0, 1.0 - th * th / 8.0
    else:
        im, re = np.sin(0.5 * th) / th, np.cos(0.5 * th)
    return np.array([im * phi[0], im * phi[1], im * phi[2], re])


class WindowEstimator:
    """Estimator::Estimate in full-window mode (windowSize == SLIDEWINDOWSIZE, Estimator.cpp:1143-1581): lidar factors
    of every frame (associated once, thres_dist 1, plan_weight_tan 3e-4, no loss), IMU factors between consecutive
    frames, the marginalization prior of the previous call.  Association runs on the device.  solver="device" (the
    default when the window sits in consecutive slots): the 15 W trust-region iteration with its IMU factors and
    prior is one kernel launch (mml_fullwindow_solve); solver="host": the iteration is host code behind the same
    C-ABI (mml_fullwindow_step) and every evaluation fetches the per-frame lidar normal equations from the device.
    The marginalization is host code either way."""

    def __init__(self, ctx, exTlb=None, gravity=(0.0, 0.0, -9.805), max_outer=5, inner_iters=10, solver="device"):
        if solver not in ("device", "host"):
            raise ValueError("solver must be 'device' or 'host'")
        self.solver = solver
        import importlib
        self.M = importlib.import_module(__package__)
        self.ctx = ctx
        self.exTlb = np.eye(4) if exTlb is None else np.asarray(exTlb, dtype=np.float64)
        self.T_bl = np.linalg.inv(self.exTlb)
        self.exRbl = self.exTlb[:3, :3].T.copy()
        self.exPbl = -1.0 * self.exRbl @ self.exTlb[:3, 3]
        self.gravity = np.asarray(gravity, dtype=np.float64)
        self.max_outer, self.inner_iters = max_outer, inner_iters
        self.prior = None                 # last_marginalization_info
        self.plan_weight_tan = 0.0003     # :1203
        self.thres_dist = 1.0             # :1204

    def _T_wl(self, x15):
        R = _quat_to_matrix(_quat_from_rotvec(x15[3:6]))
        T = np.eye(4)
        T[:3, :3] = R @ self.exRbl
        T[:3, 3] = R @ self.exPbl + x15[:3]
        return T

    def _records(self, slots, x):
        # one launch + one read-back per trust-region evaluation when the window sits in consecutive slots
        if all(s == slots[0] + f for f, s in enumerate(slots)):
            return self.ctx.linearize_window(slots[0], len(slots), x, self.T_bl, self.plan_weight_tan, 0.0)
        M = self.M
        return np.stack([M.pack_record(*self.ctx.linearize(s, x[f][:6], self.T_bl, self.plan_weight_tan, 0.0))
                         for f, s in enumerate(slots)])

    def estimate(self, slots, frames, preints):
        """slots: scan slot of every frame (already down-sampled); frames: list of dicts with P, Q (x,y,z,w), V, bg, ba
        (updated in place); preints[f] (f >= 1): mml_imu_preint between frames f-1 and f.  Returns the new prior."""
        M, ctx, W = self.M, self.ctx, len(slots)
        info = dict(outer=0, summaries=[])
        for it in range(self.max_outer):
            x = np.stack([np.concatenate([fr["P"], _rotvec_from_quat(fr["Q"]), fr["V"], fr["bg"], fr["ba"]]) for fr in frames])
            if it == 0:                                        # vLineFeatures / vPlanFeatures are empty only here
                if all(s == slots[0] + f for f, s in enumerate(slots)):   # one enqueue for the whole window, no read-back
                    ctx.associate(slots[0], W, np.stack([self._T_wl(x[f]) for f in range(W)]), self.thres_dist, stats=False)
                else:
                    for f, s in enumerate(slots):
                        ctx.associate(s, 1, self._T_wl(x[f])[None], self.thres_dist, stats=False)
            q_before, t_before = frames[-1]["Q"].copy(), frames[-1]["P"].copy()
            fw = M.FullWindowSolver(W, max_iters=self.inner_iters, fixed=False, huber=0.0, w_tan=self.plan_weight_tan)
            for f in range(1, W):
                fw.set_imu(f, preints[f], self.gravity)
            if self.prior is not None:
                fw.set_prior(self.prior)
            if self.solver == "device" and all(s == slots[0] + f for f, s in enumerate(slots)):
                x, _, evals = fw.solve_device(ctx, slots[0], self.T_bl, x)
                info["evaluations"] = info.get("evaluations", 0) + evals
            else:
                for _ in range(20 * self.inner_iters):
                    done, x = fw.step(self._records(slots, x), x)
                    info["evaluations"] = info.get("evaluations", 0) + 1
                    if done:
                        break
            info["summaries"].append(fw.summary())
            for f, fr in enumerate(frames):                    # double2vector
                fr["P"], fr["Q"] = x[f][0:3].copy(), _quat_from_rotvec(x[f][3:6])
                fr["V"], fr["bg"], fr["ba"] = x[f][6:9].copy(), x[f][9:12].copy(), x[f][12:15].copy()
            info["outer"] = it + 1
            d = abs(float(np.dot(q_before, frames[-1]["Q"])))
            deltaR = 2.0 * np.arccos(min(1.0, d)) * 180.0 / np.pi      # angularDistance, :1443
            deltaT = float(np.linalg.norm(t_before - frames[-1]["P"]))
            if (deltaR < 0.05 and deltaT < 0.05) or it + 1 == self.max_outer:
                # marginalize frame 0 (:1453-1546): previous prior, IMU factor 0-1, the stored lidar factors of frame 0
                rec0 = M.pack_record(*ctx.linearize(slots[0], x[0][:6], self.T_bl, self.plan_weight_tan, 0.0))
                self.prior = fw.marginalize(rec0, x) if W >= 2 else None
                break
        return info
