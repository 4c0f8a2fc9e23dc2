"""GPU parity at the large BASELINE configurations (configs[3] and configs[4] shapes): 128-ring x 2048 scans against a
2 M-point map, and 240 k-point fused scans against a 10 M-point map with 20 GN iterations.  Where the oracle finishes in
seconds it is the checker; at 10 M map points exactness is checked against brute force on a sample of queries and
through size-independent properties (determinism, slot independence, pose against the generating ground truth)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rsc

from conftest import perturbed, pose_to_x

pytestmark = pytest.mark.gpu

PITCH0, STEP = -25.0, 40.0 / 127.0


def _dense_scan(synth, k, n_az):
    return synth.velo_scan(k, n_rings=128, n_az=n_az, pitch0=PITCH0, pitch_step=STEP)


def _oracle_fused(O, v, l):
    ev = O.extract_velo(v, n_rings=128, pitch0=PITCH0, pitch_step=STEP)
    xyz, lab = ev["xyzi"][:, :3], ev["label"]
    if l is not None:
        el = O.extract_livox(l)
        xyz, lab = np.concatenate([xyz, el["xyzi"][:, :3]]), np.concatenate([lab, el["label"]])
    return xyz, lab


def _map_from(M, O, synth, c, ks, n_az, livox):
    cm, sm = [], []
    for k in ks:
        v = _dense_scan(synth, k, n_az)
        xyz, lab = _oracle_fused(O, v, synth.livox_scan(k) if livox else None)
        T = synth.pose_matrix(k)
        cm.append(synth.transform(T, O.voxel_downsample(xyz[lab == 1], 0.4).astype(np.float64)).astype(np.float32))
        sm.append(synth.transform(T, O.voxel_downsample(xyz[lab == 2], 0.2).astype(np.float64)).astype(np.float32))
    return O.voxel_downsample(np.concatenate(cm), 0.4), O.voxel_downsample(np.concatenate(sm), 0.2)


def test_configs3_128x2048_scan_2m_map(M, O, synth):
    n_az = 2048
    c = M.Context(max_scans=2, max_velo_points=128 * n_az, max_livox_points=64, n_rings=128, pitch0_deg=PITCH0,
                  pitch_step_deg=STEP, max_features=1 << 16, max_map_points=(1 << 21) + (1 << 18))
    try:
        cm, sm = _map_from(M, O, synth, c, (0, 2, 4), n_az, False)
        corner_map = synth.grow_map(cm, 200000, seed=7)
        surf_map = synth.grow_map(sm, 1800000, seed=8)
        c.map_set_local(0, corner_map)
        c.map_set_local(1, surf_map)
        empty = np.zeros(0, synth.LIVOX_DTYPE)
        scans = [_dense_scan(synth, k, n_az) for k in (10, 11)]
        for s, v in enumerate(scans):
            c.scan_upload(s, v, empty)
        c.extract(0, 2)
        c.undistort(0, 2, np.tile(np.eye(3).reshape(1, 9), (2, 1)), np.zeros((2, 3)))
        c.downsample(0, 2)
        feats = []
        for s, v in enumerate(scans):
            xyz, lab = _oracle_fused(O, v, None)
            d = c.scan_download(s)
            assert len(d["label"]) == len(lab)
            assert np.array_equal(d["label"], lab) and np.array_equal(d["xyzi"][:, :3], xyz)   # bit-exact labels, 262k points
            cf, sf = O.voxel_downsample(xyz[lab == 1], 0.4), O.voxel_downsample(xyz[lab == 2], 0.2)
            assert c.features_download(s, 0).tobytes() == cf.tobytes() and c.features_download(s, 1).tobytes() == sf.tobytes()
            feats.append((cf, sf))
        # association + 10 fixed GN iterations against the 2 M-point map
        T = np.stack([perturbed(synth.pose_matrix(k)) for k in (10, 11)])
        st = c.associate(0, 2, T, 25.0)
        tc, ts = O.KdTree(corner_map), O.KdTree(surf_map)
        for s in range(2):
            lf, lsrc = O.associate_lines(feats[s][0], tc, T[s], 25.0)
            pf, psrc = O.associate_planes(feats[s][1], ts, T[s], 25.0)
            gl, glsrc = c.factors_download(s, 0)
            gp, gpsrc = c.factors_download(s, 1)
            assert len(pf) > 3000 and np.array_equal(glsrc, lsrc) and np.array_equal(gpsrc, psrc)
            ol = np.concatenate([lf["point_ori"], lf["p1"], lf["p2"], lf["error"][:, None]], axis=1)
            op = np.concatenate([pf["point_ori"], pf["point_proj"], pf["omega"], pf["error"][:, None]], axis=1)
            assert np.allclose(gl, ol, rtol=0, atol=1e-9) and np.allclose(gp, op, rtol=0, atol=1e-9)
            assert st[s].n_line == len(lf) and st[s].n_plane == len(pf)
            x0 = pose_to_x(T[s])[None]
            xs, _, _ = c.solve(s, 1, x0, np.eye(4), max_iters=10, fixed=True, huber=0.1 / 1.5e-3, w_tan=0.0)
            xo, _, _ = O.solve_window([lf], [pf], x0, np.eye(4), 10, fixed=True)
            assert np.abs(xs[0] - xo[0]).max() < 1e-6
            assert np.abs(xs[0][:3] - synth.pose_matrix((10, 11)[s])[:3, 3]).max() < 0.02
    finally:
        c.close()


def test_configs4_240k_scans_10m_map_properties(M, O, synth):
    n_az = 1687
    B = 4
    c = M.Context(max_scans=B, max_velo_points=128 * n_az, max_livox_points=24000, n_rings=128, pitch0_deg=PITCH0,
                  pitch_step_deg=STEP, max_features=1 << 16, max_map_points=10_500_000)
    try:
        cm, sm = _map_from(M, O, synth, c, (0, 3), n_az, True)
        corner_map = synth.grow_map(cm, 1_000_000, seed=7)
        surf_map = synth.grow_map(sm, 9_000_000, seed=8)
        c.map_set_local(0, corner_map)
        c.map_set_local(1, surf_map)
        # exact 5-NN against 9 M points: brute force on a sample of queries
        rng = np.random.default_rng(0)
        q = surf_map[rng.integers(0, len(surf_map), 48)] + rng.normal(0, 0.05, (48, 3)).astype(np.float32)
        gi, gd = c.knn5(1, q)
        for j in range(len(q)):
            d = surf_map - q[j]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            order = np.lexsort((np.arange(len(d2)), d2))[:5]
            assert np.array_equal(gi[j], order) and np.array_equal(gd[j], d2[order])
        ks = [10, 11, 12, 10]
        for s, k in enumerate(ks):
            c.scan_upload(s, _dense_scan(synth, k, n_az), synth.livox_scan(k))
        dR = np.tile(np.eye(3).reshape(1, 9), (B, 1))
        x0 = np.stack([pose_to_x(perturbed(synth.pose_matrix(k))) for k in ks])
        x1 = c.step(0, B, dR, np.zeros((B, 3)), np.eye(4), 25.0, 20, x0)
        x2 = c.step(0, B, dR, np.zeros((B, 3)), np.eye(4), 25.0, 20, x0)
        assert np.array_equal(x1, x2) and np.array_equal(x1[0], x1[3])      # deterministic, slot-independent
        for s, k in enumerate(ks):
            assert np.abs(x1[s][:3] - synth.pose_matrix(k)[:3, 3]).max() < 0.02
        # labels and down-sampled stacks of one 240 k-point fused scan against the oracle
        xyz, lab = _oracle_fused(O, _dense_scan(synth, 11, n_az), synth.livox_scan(11))
        d = c.scan_download(1)
        assert np.array_equal(d["label"], lab)
        assert c.features_download(1, 1).tobytes() == O.voxel_downsample(xyz[lab == 2], 0.2).tobytes()
    finally:
        c.close()
