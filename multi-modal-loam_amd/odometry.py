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
        self.last_update_pose = np.array([-1.0, -1.0, -1.0])          # Estimator.h:339-340
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
            d = self.last_update_pose - cur
            dis = float(np.float32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])) if self.lidar_mode == 2 else float(d @ d)
            if dis >= 0.5:
                self.n_corner_local, self.n_surf_local = ctx.map_increment_local(slot, T)   # :1125-1130
                self.last_update_pose = cur
                self.key_scans += 1
                grew = True
        self.fail_detected = is_degenerate                              # :1139
        return P, Q, grew
