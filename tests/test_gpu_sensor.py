"""GPU parity on sensor-faithful inputs: what a real VLP-16 / Livox Horizon bag differs in from the ideal synthetic grid
(synth.velo_scan_vlp16, synth.livox_scan_horizon) -- firing-order point sequence, encoder-quantised azimuths and a scan cut
past one revolution (unionFeatureExtract.cpp:1133-1195 start / end orientation, halfPassed), 2 mm ranges and integer
intensities (mass ties in both partition sorts, :453-479), NaN and (0,0,0) no-returns (:369-388, :1133, :1163-1166),
millimetre Livox coordinates, `tag` bits, stray `line` ids and (0,0,0) records (:985-998)."""
import numpy as np
import pytest

from conftest import GOLDEN, oracle_pipeline, perturbed, pose_to_x

pytestmark = pytest.mark.gpu

NV = 1824 * 16


def sensor_case(synth, k, mode, motion=False):
    v = synth.velo_scan_vlp16(k, dropout=mode, motion=motion)
    l = synth.livox_scan_horizon(k, motion=motion)
    dR, dt = synth.sweep_motion(k) if motion else (np.eye(3), np.zeros(3))
    T0 = perturbed(synth.pose_matrix(k))
    return dict(velo=v, livox=l, dR=dR, dt=dt, T0=T0, x0=pose_to_x(T0), k=k)


def test_sensor_golden(M):
    g = np.load(GOLDEN + "/extract_sensor.npz", allow_pickle=False)
    c = M.Context(max_scans=2, max_velo_points=len(g["velo"]), max_livox_points=len(g["livox"]))
    try:
        vz = np.where(np.isnan(g["velo"]), np.float32(0), g["velo"])
        for slot, v in enumerate((g["velo"], vz)):
            c.scan_upload(slot, v, g["livox"])
        c.extract(0, 2)
        for slot in range(2):
            d = c.scan_download(slot)
            assert np.array_equal(d["xyzi"], np.concatenate([g["velo_xyzi"], g["livox_xyzi"]]))
            assert np.array_equal(d["label"], np.concatenate([g["velo_label"], g["livox_label"]]))
            assert np.array_equal(d["ring"], np.concatenate([g["velo_ring"], g["livox_ring"]]))
            assert np.array_equal(d["reltime"], np.concatenate([g["velo_rel"], g["livox_rel"]]))
            i = d["info"]
            assert [i.velo_corner_num, i.velo_surf_num] == list(g["velo_counts"])
            assert [i.livox_corner_num, i.livox_surf_num] == list(g["livox_counts"])
        s, f, fl = c.detect_line(g["ring"])
        assert np.array_equal(s, g["ring_sharp"]) and np.array_equal(f, g["ring_flat"]) and np.array_equal(fl, g["ring_flags"])
    finally:
        c.close()


def test_batch72_sensor_faithful_scans_match_oracle(M, O, synth):
    """72 slots = 24 distinct sensor-faithful fused scans x 3 (1.2 M distinct points): no-returns absent / NaN / (0,0,0) by
    turns, a third of them with intra-sweep motion.  Reaches the batch kernels (k_stencil<0>, batch k_select_part,
    k_select_list, k_voxel<256>/<512>, lane search, the default 2 stream lanes of 36 slots).  Extraction bit for bit on every slot; after
    mml_step the undistorted clouds, labels, stacks, factor records and poses against the oracle pipeline."""
    B, ND = 72, 24
    modes = ("skip", "nan", "zero")
    cases = [sensor_case(synth, 40 + j, modes[j % 3], motion=(j % 3 == 1)) for j in range(ND)]
    # map: sensor-faithful scans 0..7 through the oracle
    maps = [[], []]
    for k in range(8):
        o = oracle_pipeline(O, sensor_case(synth, k, "skip"), None, None)
        T = synth.pose_matrix(k)
        maps[0].append(synth.transform(T, o["corner"].astype(np.float64)).astype(np.float32))
        maps[1].append(synth.transform(T, o["surf"].astype(np.float64)).astype(np.float32))
    cm, sm = O.voxel_downsample(np.concatenate(maps[0]), 0.4), O.voxel_downsample(np.concatenate(maps[1]), 0.2)
    tc, ts = O.KdTree(cm), O.KdTree(sm)
    ora = [oracle_pipeline(O, cs, tc, ts) for cs in cases]
    assert sum(len(o["xyzi"]) for o in ora) > 1_000_000
    c = M.Context(max_scans=B, max_velo_points=NV, max_livox_points=24000)
    try:
        c.map_set_local(0, cm)       # (no set_lanes: the library's default two lanes, 36 slots each)
        c.map_set_local(1, sm)
        for s in range(B):
            c.scan_upload(s, cases[s % ND]["velo"], cases[s % ND]["livox"])
        c.extract(0, B)
        redo = brk = pts = 0
        for s in range(B):
            d, o = c.scan_download(s), ora[s % ND]
            assert d["info"].n_points == len(o["xyzi"])
            for key in ("xyzi", "label", "ring", "reltime"):
                assert np.array_equal(d[key], o[key]), (s, key)
            i = d["info"]
            assert (i.velo_corner_num, i.velo_surf_num, i.livox_corner_num, i.livox_surf_num) == o["counts"]
            r, b = c.extract_queue_counts(s)
            redo, brk, pts = redo + r, brk + b, pts + len(o["xyzi"])
        print("sensor-faithful batch: %d points, guard-band fall-back (redo queue) %.4f %%, break-point queue %.4f %%"
              % (pts, 100.0 * redo / pts, 100.0 * brk / pts))
        dR = np.stack([cases[s % ND]["dR"].reshape(9) for s in range(B)])
        dt = np.stack([cases[s % ND]["dt"] for s in range(B)])
        x0 = np.stack([cases[s % ND]["x0"] for s in range(B)])
        x = c.step(0, B, dR, dt, np.eye(4), 25.0, 10, x0)
        for s in range(B):
            d, o = c.scan_download(s), ora[s % ND]
            assert np.array_equal(d["label"], o["label"]) and np.array_equal(d["xyzi"][:, :3], o["und"])
            assert c.features_download(s, 0).tobytes() == o["corner"].tobytes()
            assert c.features_download(s, 1).tobytes() == o["surf"].tobytes()
            gl, glsrc = c.factors_download(s, 0)
            gp, gpsrc = c.factors_download(s, 1)
            assert np.array_equal(glsrc, o["lsrc"]) and np.array_equal(gpsrc, o["psrc"])
            assert np.allclose(gl, o["lf_arr"], rtol=0, atol=1e-9) and np.allclose(gp, o["pf_arr"], rtol=0, atol=1e-9)
            assert np.abs(x[s] - o["x"]).max() < 1e-6
            assert np.abs(x[s][:3] - synth.pose_matrix(cases[s % ND]["k"])[:3, 3]).max() < 0.05   # and it registers
    finally:
        c.close()
