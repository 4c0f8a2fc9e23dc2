"""GPU suite: the C++ adapter (multi-modal-loam_amd/host/mmloam_adapter.hpp -- the reference-language mirror of
feature_extraction / RemoveLidarDistortion / Estimator) driven on a real device by tests/cpp/adapter_gpu_probe.cpp the
way the two reference nodes would be: the feature node extracts in one context, the labelled cloud crosses the
"topic" as 48-byte PointXYZINormal records, the pose node takes the CALLER's cloud (mml_cloud_upload underneath) through
RemoveLidarDistortion(cloud, dR, dt) and EstimateLidarPose(list<LidarFrame>) / EstimateFullWindow.  Compared with the
oracle loop (1-frame mode) and with the Python control flow that test_gpu_parity.py pins to the oracle (full window)."""
import importlib
import os
import struct
import subprocess

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rsc

from conftest import ROOT, perturbed

pytestmark = pytest.mark.gpu


def _build_probe(tmp_path):
    exe = tmp_path / "adapter_gpu_probe"
    libdir = os.path.join(ROOT, "multi-modal-loam_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(libdir, "host"),
           os.path.join(ROOT, "tests", "cpp", "adapter_gpu_probe.cpp"), "-o", str(exe), "-L", libdir, "-lmmloam_hip",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    return str(exe)


def _write_scene(path, mode, scans, corner_map, surf_map, windows=None):
    with open(path, "wb") as f:
        f.write(struct.pack("<iii", 0x4d4d4c31, mode, len(scans)))
        for s in scans:
            v = np.ascontiguousarray(s["velo"], np.float32).reshape(-1, 4)
            f.write(struct.pack("<i", len(v)))
            f.write(v.tobytes())
            f.write(struct.pack("<i", len(s["livox"])))
            f.write(np.ascontiguousarray(s["livox"]).tobytes())
            for key, n in (("dR", 9), ("dt", 3), ("P", 3), ("Q", 4), ("V", 3)):
                a = np.ascontiguousarray(s[key], np.float64).reshape(-1)
                assert len(a) == n
                f.write(a.tobytes())
            imu = np.ascontiguousarray(s.get("imu", np.zeros((0, 7))), np.float64).reshape(-1, 7)
            f.write(struct.pack("<i", len(imu)))
            f.write(imu.tobytes())
        for m in (corner_map, surf_map):
            m = np.ascontiguousarray(m, np.float32).reshape(-1, 3)
            f.write(struct.pack("<i", len(m)))
            f.write(m.tobytes())
        if windows is not None:
            f.write(struct.pack("<ii", len(windows), len(windows[0])))
            for w in windows:
                f.write(np.asarray(w, np.int32).tobytes())


def _run(exe, scene_file):
    run = subprocess.run([exe, scene_file], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert "ADAPTER_PROBE_DONE" in run.stdout and "ctor_rejects_mismatch 1" in run.stdout, run.stdout[-2000:]
    return [ln.split() for ln in run.stdout.splitlines()]


def test_cpp_adapter_odometry_with_callers_cloud_matches_oracle(M, O, synth, tmp_path):
    """unionCloud (context A) -> cloud over the 'topic' -> RemoveLidarDistortion(cloud, dR, dt) -> EstimateLidarPose with
    LidarFrame::laserCloud (context B), eight scans: map grown on the device by the key-scan rule."""
    exe = _build_probe(tmp_path)
    ks = list(range(20, 20 + 4 * 8, 4))
    scans = []
    for k in ks:
        v, l = synth.velo_scan(k, motion=True), synth.livox_scan(k, motion=True)
        dR, dt = synth.sweep_motion(k)
        Tp = perturbed(synth.pose_matrix(k), dt=(0.02, -0.015, 0.01), rotvec=(0.002, -0.001, 0.003)) if k != ks[0] else synth.pose_matrix(k)
        scans.append(dict(velo=v, livox=l, dR=dR, dt=dt, P=Tp[:3, 3], Q=Rsc.from_matrix(Tp[:3, :3]).as_quat(), V=np.zeros(3)))
    scene_file = str(tmp_path / "scene0.bin")
    _write_scene(scene_file, 0, scans, np.zeros((0, 3)), np.zeros((0, 3)))
    out_rows = _run(exe, scene_file)
    rows = [r for r in out_rows if r and r[0] == "scan"]
    assert len(rows) == len(ks)
    # oracle loop: extract -> undistort -> down-sample -> Estimate (5 x 10) -> key-scan rule -> MapIncrementLocal
    lm = O.LocalMap(window=50, leaf_corner=0.4, leaf_surf=0.2)
    last_update = np.array([-1.0, -1.0, -1.0])
    n_est = 0
    for s, r in zip(scans, rows):
        ev, el = O.extract_velo(s["velo"]), O.extract_livox(s["livox"])
        xyz = np.concatenate([ev["xyzi"][:, :3], el["xyzi"][:, :3]])
        rel = np.concatenate([ev["reltime"], el["reltime"]])
        lab = np.concatenate([ev["label"], el["label"]])
        und = O.undistort(xyz, rel, s["dR"], s["dt"])
        cf, sf = O.voxel_downsample(und[lab == 1], 0.4), O.voxel_downsample(und[lab == 2], 0.2)
        Po, Qo = np.array(s["P"], dtype=np.float64), np.array(s["Q"], dtype=np.float64)
        cm, sm = lm.get(0), lm.get(1)
        deg = False
        if len(cm) > 0 and len(sm) > 100:
            Po, Qo, _, deg, _ = O.estimate_single(cf, sf, cm, sm, np.eye(4), Po, Qo, 5, 10)
            n_est += 1
        if not deg:
            d = last_update - Po
            if float(np.float32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])) >= 0.5:
                T = np.eye(4)
                T[:3, :3] = Rsc.from_quat(Qo).as_matrix()
                T[:3, 3] = Po
                lm.increment(cf, sf, T)
                last_update = Po.copy()
        # what the C++ side printed
        assert int(r[3]) == len(xyz) and int(r[5]) == len(ev["xyzi"])
        assert int(r[9]) == int(deg)
        Pg = np.array([float(x) for x in r[11:14]])
        Qg = np.array([float(x) for x in r[15:19]])
        assert np.abs(Pg - Po).max() < 1e-6 and min(np.abs(Qg - Qo).max(), np.abs(Qg + Qo).max()) < 1e-6
        # the caller's cloud was undistorted in place (x, y, z rewritten, normal_x = 1)
        mid = len(xyz) // 2
        assert np.allclose(np.array([float(x) for x in r[20:23]]), und[mid], rtol=1e-6, atol=1e-6)   # (bit parity of a9 is test_gpu_parity's job)
        assert not np.allclose(und[mid], xyz[mid], atol=1e-4) or np.abs(xyz[mid]).max() < 0.5
        assert float(r[24]) == 1.0
    assert n_est >= 5
    # the last scan's features through the adapter's processPointToLine / processPointToPlanVec: same counts and error sums as
    # the oracle's association at the printed pose against the oracle's local map; the cube store holds that scan's features
    ar = [r for r in out_rows if r and r[0] == "assoc"][0]
    T = np.eye(4)
    T[:3, :3] = Rsc.from_quat(Qg).as_matrix()
    T[:3, 3] = Pg
    lf, _ = O.associate_lines(cf, O.KdTree(lm.get(0)), T, 1.0)
    pf, _ = O.associate_planes(sf, O.KdTree(lm.get(1)), T, 1.0)
    assert int(ar[2]) == len(lf) and int(ar[8]) == len(pf)
    assert int(ar[4]) == int((np.abs(lf["error"]) > 1e-5).sum()) and int(ar[10]) == int((np.abs(pf["error"]) > 1e-5).sum())
    assert abs(float(ar[6]) - lf["error"].sum()) < 1e-6 * max(1.0, abs(lf["error"].sum()))
    assert abs(float(ar[12]) - pf["error"].sum()) < 1e-6 * max(1.0, abs(pf["error"].sum()))
    # (cubes beyond 300 points are voxel-filtered at 0.4 m, Map_Manager.cpp:225-233: the surf cube shrinks)
    assert int(ar[16]) == len(cf) <= 300 and 0 < int(ar[17]) <= len(sf)


def test_cpp_adapter_full_window_with_callers_clouds(M, O, synth, scene, tmp_path):
    """EstimateFullWindow on frames that carry their clouds (LidarFrame::laserCloud), two consecutive 5-frame windows
    with IMU factors and the carried prior, against odometry.WindowEstimator (itself pinned to the oracle loop by
    test_full_window_estimate_with_imu_matches_oracle_loop) on the same inputs."""
    exe = _build_probe(tmp_path)
    odometry = importlib.import_module("multi-modal-loam_amd.odometry")
    W = 5
    rng = np.random.default_rng(23)
    ks = list(range(10, 10 + W + 1))
    scans = []
    for k in ks:
        T = perturbed(synth.pose_matrix(k), dt=rng.normal(0, 0.02, 3), rotvec=rng.normal(0, 0.003, 3))
        Q = Rsc.from_matrix(T[:3, :3]).as_quat()
        if Q[3] < 0:
            Q = -Q
        scans.append(dict(velo=synth.velo_scan(k), livox=synth.livox_scan(k), dR=np.eye(3), dt=np.zeros(3), P=T[:3, 3].copy(), Q=Q,
                          V=synth.velocity_at(k) + rng.normal(0, 0.02, 3), imu=synth.imu_samples(k - 1, k)))
    windows = [list(range(0, W)), list(range(1, W + 1))]
    scene_file = str(tmp_path / "scene1.bin")
    _write_scene(scene_file, 1, scans, scene["corner_map"], scene["surf_map"], windows)
    rows = [r for r in _run(exe, scene_file) if r and r[0] == "window"]
    assert len(rows) == 2 * W
    c = M.Context(max_scans=W)
    try:
        c.map_set_local(0, scene["corner_map"])
        c.map_set_local(1, scene["surf_map"])
        west = odometry.WindowEstimator(c, gravity=synth.GRAVITY)
        for w, idx in enumerate(windows):
            frames, pres = [], [None]
            for f, i in enumerate(idx):
                s = scans[i]
                c.scan_upload(f, s["velo"], s["livox"])
                c.extract(f, 1)
                c.downsample(f, 1)
                frames.append(dict(P=s["P"].copy(), Q=s["Q"].copy(), V=s["V"].copy(), bg=np.zeros(3), ba=np.zeros(3)))
                if f > 0:
                    pres.append(M.imu_preintegrate(s["imu"], np.zeros(3), np.zeros(3)))
            west.estimate(list(range(W)), frames, pres)
            for f in range(W):
                r = rows[w * W + f]
                vals = np.array([float(x) for x in r[5:8] + r[9:13] + r[14:17] + r[18:21] + r[22:25]])
                ref = np.concatenate([frames[f]["P"], frames[f]["Q"], frames[f]["V"], frames[f]["bg"], frames[f]["ba"]])
                assert np.abs(vals - ref).max() < 1e-9, (w, f, np.abs(vals - ref).max())
                assert np.abs(vals[:3] - synth.pose_matrix(ks[idx[f]])[:3, 3]).max() < 0.03
    finally:
        c.close()


def test_cloud_upload_is_the_inverse_of_download(M, synth):
    """mml_cloud_upload(mml_scan_download_pointxyzinormal(slot)) reproduces the slot: cloud, labels, counts, and the
    down-sampled stacks that follow from them."""
    a, b = M.Context(max_scans=2), M.Context(max_scans=2)
    try:
        v, l = synth.velo_scan(5), synth.livox_scan(5)
        a.scan_upload(1, v, l)
        a.extract(1, 1)
        rec = a.scan_download_pointxyzinormal(1)
        ia = a.scan_info(1)
        b.cloud_upload(0, rec, ia.n_velo)
        ib = b.scan_info(0)
        assert (ib.n_points, ib.n_velo, ib.fused_corner_num, ib.fused_surf_num, ib.velo_corner_num, ib.velo_surf_num) == \
               (ia.n_points, ia.n_velo, ia.fused_corner_num, ia.fused_surf_num, ia.velo_corner_num, ia.velo_surf_num)
        da, db = a.scan_download(1), b.scan_download(0)
        for k in ("xyzi", "reltime", "ring", "label"):
            assert np.array_equal(da[k], db[k]), k
        assert np.array_equal(b.scan_download_pointxyzinormal(0), rec)
        dR = Rsc.from_rotvec([0.001, -0.002, 0.02]).as_matrix().reshape(1, 9)
        dt = np.array([[0.05, -0.01, 0.002]])
        for c, s in ((a, 1), (b, 0)):
            c.undistort(s, 1, dR, dt)
            c.downsample(s, 1)
        for kind in (0, 1):
            assert np.array_equal(a.features_download(1, kind), b.features_download(0, kind))
        # empty cloud and argument checks
        b.cloud_upload(1, np.zeros((0, 12), np.float32), 0)
        assert b.scan_info(1).n_points == 0
        with pytest.raises(M.MmlError):
            b.cloud_upload(0, rec, len(rec) + 1)
    finally:
        a.close()
        b.close()
