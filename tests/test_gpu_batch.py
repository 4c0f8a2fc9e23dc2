"""GPU parity of the LARGE-BATCH kernel variants -- the ones bench.py and production batches run -- against the oracle.

Batch size routes the path to different kernels in five places:
  * count > 16  -> k_stencil<0> (one wavefront per scan line)            instead of k_stencil<1>/<2> (tile per wavefront)
  * count > 16  -> batch k_select_part<u64/u128> grid + k_select_list + list-mode k_select   instead of direct k_select
  * count > 16  -> k_voxel<256> for the corner lists + k_voxel<512> for the surf lists up to 4096 labelled points + the list-walking
                   k_voxel<1024> launch for the slots with more                         instead of one k_voxel<1024>
  * count > 8   -> lane-per-feature k_associate search                   instead of the <= 8-slot group search
  * count >= 64 -> mml_step pipelines sub-batches over stream lanes (set to 4 in some tests; test_full_size_step_properties and
    test_gpu_shapes.py run the library's default of 2) instead of one stream
Every test below is sized so that EACH lane's sub-batch is still > 16 slots, and compares what those kernels leave
behind with the oracle's restatement of unionFeatureExtract.cpp:341-844,952-1035,1113-1317, unionPoseEstimation.cpp:402-421,
Estimator.cpp:148-365,573-777,992-1026 and the solver restatement: labels / rings / times / coordinates / stacks bit for
bit, factor records and poses to 1e-9."""
import numpy as np
import pytest

from conftest import batch_cases, oracle_pipeline, perturbed, pose_to_x

pytestmark = pytest.mark.gpu


def _check_extraction(d, o):
    assert d["info"].n_points == len(o["xyzi"])
    for key in ("xyzi", "label", "ring"):
        assert np.array_equal(d[key], o[key]), key
    # (a Livox part of ONE record has timeSpan 0: its time is 0 / 0 in the reference as well, :985-995)
    assert np.array_equal(d["reltime"].view(np.uint32), o["reltime"].view(np.uint32)) or np.array_equal(d["reltime"], o["reltime"], equal_nan=True)
    i = d["info"]
    got = (i.velo_corner_num, i.velo_surf_num, i.livox_corner_num, i.livox_surf_num)
    assert got == o["counts"], (got, o["counts"])


def _check_after_step(c, s, o, x_s, factors=True, tol=1e-9, pose_tol=1e-6):
    """Slot s after mml_step against the oracle pipeline record o of the scan it holds.  mml_step runs a FIXED number of
    trust-region iterations: past convergence they divide rounding noise by a vanishing model decrease, so the pose bar is
    1e-6 there (as in test_linearize_and_solve_trace; two orders below north_star's 1e-4) -- the solve that stops at
    convergence is held to 1e-9 with equal iteration counts in test_batch96 part (3)."""
    d = c.scan_download(s)
    assert np.array_equal(d["label"], o["label"]) and np.array_equal(d["ring"], o["ring"])
    assert np.array_equal(d["xyzi"][:, :3], o["und"])              # undistorted in place, 0 ulp
    assert np.array_equal(d["xyzi"][:, 3], o["xyzi"][:, 3])
    assert np.all(d["reltime"] == 1.0)                              # normal_x reads 1 afterwards (unionPoseEstimation.cpp:419)
    assert c.features_download(s, 0).tobytes() == o["corner"].tobytes()   # k_voxel<256> (or its overflow redo)
    assert c.features_download(s, 1).tobytes() == o["surf"].tobytes()     # k_voxel<1024>
    if factors:
        gl, glsrc = c.factors_download(s, 0)
        gp, gpsrc = c.factors_download(s, 1)
        assert np.array_equal(glsrc, o["lsrc"]) and np.array_equal(gpsrc, o["psrc"])
        assert np.allclose(gl, o["lf_arr"], rtol=0, atol=tol) and np.allclose(gp, o["pf_arr"], rtol=0, atol=tol)
    assert np.abs(x_s - o["x"]).max() < pose_tol, np.abs(x_s - o["x"]).max()


def test_batch96_default_layout_matches_oracle(M, O, synth, scene):
    """96 slots = 12 distinct fused scans x 8 (slot s holds scan s % 12), 16 rings + 6 Livox lines.
    Reaches: k_stencil<0>, batch k_select_part<u64m>/<u128m>, k_select_list + list-mode k_select (the 100-point Livox
    lines, the 10 k-point ragged line, the 3-point scans), k_voxel<256> (incl. a 1 570-corner scan and a
    3 350-corner scan that overflows into the global-sort redo), k_voxel<1024>, the lane-per-feature k_associate search
    (count > 8) and mml_step on 4 stream lanes (24 slots each) -- asserted equal to the same step on 1 lane."""
    B = 96
    cases = batch_cases(synth)
    assert len(cases) == 12
    tc, ts = O.KdTree(scene["corner_map"]), O.KdTree(scene["surf_map"])
    ora = [oracle_pipeline(O, cs, tc, ts) for cs in cases]
    assert 1024 < (ora[10]["label"] == 1).sum() < 2048 < (ora[11]["label"] == 1).sum()
    # (more than 4096 surf labels: the slot k_voxel<512> lists for the list-walking launch of the large form, k_voxel<1024, true>)
    assert (ora[10]["label"] == 2).sum() < 4096 < (ora[11]["label"] == 2).sum()
    c = M.Context(max_scans=B)
    try:
        c.map_set_local(0, scene["corner_map"])
        c.map_set_local(1, scene["surf_map"])
        for s in range(B):
            cs = cases[s % 12]
            c.scan_upload(s, cs["velo"], cs["livox"])
        # (1) the staged extraction of the whole batch: every field of every slot
        c.extract(0, B)
        for s in range(B):
            _check_extraction(c.scan_download(s), ora[s % 12])
        # (2) the fused step, 4 lanes and 1 lane
        c.set_lanes(4)
        dR = np.stack([cases[s % 12]["dR"].reshape(9) for s in range(B)])
        dt = np.stack([cases[s % 12]["dt"] for s in range(B)])
        x0 = np.stack([cases[s % 12]["x0"] for s in range(B)])
        x4 = c.step(0, B, dR, dt, np.eye(4), 25.0, 10, x0)
        for s in range(B):
            _check_after_step(c, s, ora[s % 12], x4[s])
        c.set_lanes(1)
        x1 = c.step(0, B, dR, dt, np.eye(4), 25.0, 10, x0)
        assert np.array_equal(x1, x4)
        for s in (0, 17, 41, 66, 95, 34, 35, 46, 47):
            _check_after_step(c, s, ora[s % 12], x1[s])
        # (3) staged association of the batch (count > 8: one lane per feature) with the statistics
        c.set_lanes(4)
        c.extract(0, B)
        c.undistort(0, B, dR, dt)
        c.downsample(0, B)
        Tw = np.stack([cases[s % 12]["T0"] for s in range(B)])
        st = c.associate(0, B, Tw, 25.0)
        for s in range(B):
            o = ora[s % 12]
            gl, glsrc = c.factors_download(s, 0)
            gp, gpsrc = c.factors_download(s, 1)
            assert np.array_equal(glsrc, o["lsrc"]) and np.array_equal(gpsrc, o["psrc"])
            assert np.allclose(gl, o["lf_arr"], rtol=0, atol=1e-9) and np.allclose(gp, o["pf_arr"], rtol=0, atol=1e-9)
            assert st[s].n_line == len(o["lsrc"]) and st[s].n_plane == len(o["psrc"])
            if len(o["psrc"]) > 10:
                assert abs(st[s].min_singular - o["min_singular"]) < 1e-9 * max(1.0, abs(o["min_singular"]))
        # ... and the solve that terminates on its own (k_solve, 96 problems in one launch): every iterate to 1e-9
        xg, sg, tg = c.solve(0, B, x0, np.eye(4), window=1, max_iters=10, trace=True)
        for s in range(B):
            o = ora[s % 12]
            xo, so, to = O.solve_window([o["lf"]], [o["pf"]], x0[s][None], np.eye(4), 10)
            assert (sg[s].iterations, sg[s].successful, sg[s].termination) == (so["iterations"], so["successful"], so["termination"])
            assert np.abs(tg[s][:so["iterations"]] - to.reshape(-1, 6)).max() < 1e-9
            assert np.abs(xg[s] - xo[0]).max() < 1e-9
    finally:
        c.close()


PITCH0, STEP = -25.0, 40.0 / 127.0


def test_batch24_dense_128_ring_layout_matches_oracle(M, O, synth):
    """24 slots of 128-ring x 1024 scans (+ Livox on some): the dense layout's batch path -- k_assign_c_staged, k_stencil<0>,
    batch k_select_part, one k_voxel<1024> launch per kind with 8192-key lists, lane-per-feature association -- against the
    oracle, every slot."""
    B, n_az = 24, 1024
    kw = dict(n_rings=128, pitch0=PITCH0, pitch_step=STEP)
    c = M.Context(max_scans=B, max_velo_points=128 * n_az, max_livox_points=24000, n_rings=128, pitch0_deg=PITCH0,
                  pitch_step_deg=STEP, max_features=1 << 15)
    try:
        def dense(k, **a):
            return synth.velo_scan(k, n_az=n_az, **kw, **a)
        dirty = dense(12).copy()
        dirty[1000:1300, 0] = np.nan
        dirty[5::211, 2] = 80.0
        dirty[40000:40400, :3] *= 0.04
        cases = [dict(velo=dense(10), livox=None), dict(velo=dense(11), livox=synth.livox_scan(11)),
                 dict(velo=dirty, livox=synth.livox_scan(12)[:9000]), dict(velo=dense(13)[:77777], livox=None)]
        maps = [[], []]
        for k in (0, 2, 4):
            cs = dict(velo=dense(k), livox=None, dR=np.eye(3), dt=np.zeros(3), T0=np.eye(4), x0=np.zeros(6))
            o = oracle_pipeline(O, cs, None, None, **kw)
            T = synth.pose_matrix(k)
            maps[0].append(synth.transform(T, o["corner"].astype(np.float64)).astype(np.float32))
            maps[1].append(synth.transform(T, o["surf"].astype(np.float64)).astype(np.float32))
        cm, sm = O.voxel_downsample(np.concatenate(maps[0]), 0.4), O.voxel_downsample(np.concatenate(maps[1]), 0.2)
        c.map_set_local(0, cm)
        c.map_set_local(1, sm)
        tc, ts = O.KdTree(cm), O.KdTree(sm)
        for j, cs in enumerate(cases):
            T0 = perturbed(synth.pose_matrix(10 + j))
            cs.update(dR=np.eye(3), dt=np.zeros(3), T0=T0, x0=pose_to_x(T0))
        ora = [oracle_pipeline(O, cs, tc, ts, **kw) for cs in cases]
        for s in range(B):
            cs = cases[s % 4]
            c.scan_upload(s, cs["velo"], cs["livox"])
        c.extract(0, B)
        for s in range(B):
            _check_extraction(c.scan_download(s), ora[s % 4])
        dR = np.tile(np.eye(3).reshape(1, 9), (B, 1))
        x0 = np.stack([cases[s % 4]["x0"] for s in range(B)])
        x = c.step(0, B, dR, np.zeros((B, 3)), np.eye(4), 25.0, 10, x0)
        for s in range(B):
            _check_after_step(c, s, ora[s % 4], x[s])
    finally:
        c.close()


def test_full_size_step_properties(M, O, scene, synth):
    """BASELINE configs[1] shape (fused 52.8 k-point scans, 200 k-point map, 10 GN iterations) at batch size 80 on the
    library's DEFAULT stream lanes (no set_lanes call: two lanes of 40 slots), i.e. the kernels of the bench line
    (k_stencil<0>, batch k_select_part, k_voxel<256>/<1024>, lane search): deterministic, slot-independent, every slot's
    labels, stacks, factor records and pose equal to the oracle pipeline, on the 200 k-point tiled map bench.py registers
    against."""
    B = 80
    frames = scene["frames"]
    cm = synth.grow_map(scene["corner_map"], 40000, seed=7)
    sm = synth.grow_map(scene["surf_map"], 160000, seed=8)
    tc, ts = O.KdTree(cm), O.KdTree(sm)
    cases = []
    for fr in frames:
        T0 = perturbed(fr["T_gt"])
        cases.append(dict(velo=fr["velo"], livox=fr["livox"], dR=np.eye(3), dt=np.zeros(3), T0=T0, x0=pose_to_x(T0)))
    ora = [oracle_pipeline(O, cs, tc, ts) for cs in cases]
    c = M.Context(max_scans=B, max_map_points=200000)
    try:
        c.map_set_local(0, cm)
        c.map_set_local(1, sm)
        for s in range(B):
            c.scan_upload(s, frames[s % 4]["velo"], frames[s % 4]["livox"])
        dR = np.tile(np.eye(3).reshape(1, 9), (B, 1))
        dt = np.zeros((B, 3))
        x0 = np.stack([cases[s % 4]["x0"] for s in range(B)])
        x1 = c.step(0, B, dR, dt, np.eye(4), 25.0, 10, x0)
        x2 = c.step(0, B, dR, dt, np.eye(4), 25.0, 10, x0)
        assert np.array_equal(x1, x2)                       # deterministic
        for s in range(4, B):
            assert np.array_equal(x1[s], x1[s % 4])         # identical inputs in different slots give identical poses
        for s in range(B):
            _check_after_step(c, s, ora[s % 4], x1[s])
        for s in range(4):
            assert np.abs(x1[s][:3] - frames[s]["T_gt"][:3, 3]).max() < 0.02
        # mml_step does not compute the association's statistics (nothing in it reads them): a record linearised straight after
        # the step gets them on demand and must equal the record after an association that computes them eagerly
        rec_lazy = c.linearize_window(0, 4, x1[:4], np.eye(4))
        st = c.associate(0, 4, np.stack([cases[s]["T0"] for s in range(4)]), 25.0)
        rec_eager = c.linearize_window(0, 4, x1[:4], np.eye(4))
        # (the eager association starts from T0, the step from the rotation vector of T0: the same factors up to the last bit of the pose)
        assert np.allclose(rec_lazy, rec_eager, rtol=1e-9, atol=1e-9) and np.array_equal(rec_lazy[:, 28:30], rec_eager[:, 28:30])
        assert [int(r[28]) for r in rec_lazy] == [q.n_line_used for q in st] and [int(r[29]) for r in rec_lazy] == [q.n_plane_used for q in st]
        assert all(q.n_plane_used > 100 for q in st)
    finally:
        c.close()


def test_sparse_lines_take_the_general_translation_path(M, O, synth):
    """The one-pass bucketing stores a line as one segment per 4096-point block of the raw scan; k_stencil, k_select_part,
    k_stencil_redo / _break translate line indices through per-64-index records that allow ONE segment boundary per run.  A ring
    that keeps only a few points per block -- sky above the upper rings of an outdoor scan -- has segments shorter than that:
    those runs take the general path through the segment table (stencil tile load, label pass, point windows).  Here: rings
    thinned to 1 point in 8 / 1 in 40 over parts of the sweep (NaN, (0,0,0) and absent records), a ring that survives in two
    blocks only, Livox lines thinned the same way; 24 slots (batch kernels) and 2 slots (segment-mode kernels), every field
    against the oracle."""
    rng = np.random.default_rng(77)
    cases = []
    for k, mode in ((50, "nan"), (51, "zero"), (52, "skip")):
        v = synth.velo_scan(k).reshape(1800, 16, 4).copy()
        thin = rng.random((1800, 16)) < 0.875
        thin[:, :11] = False                                   # rings 11..15: 1 point in 8 ...
        thin[600:1300, 13] = rng.random(700) < 0.975           # ... ring 13: 1 in 40 over a third of the sweep
        thin[:, 15] = True
        thin[100:140, 15] = False                              # ring 15: two short stretches only
        thin[1500:1530, 15] = False
        if mode == "nan":
            v[thin, 0] = np.nan
        elif mode == "zero":
            v[thin, :3] = 0.0
        v = v.reshape(-1, 4)
        if mode == "skip":
            v = v[~thin.reshape(-1)]
        l = synth.livox_scan(k).copy()
        drop = (rng.random(len(l)) < 0.93) & np.isin(l["line"], (1, 4))
        l["x"][drop] = 0.0                                      # (x < 0.01: the record is skipped, :990)
        cases.append(dict(velo=v, livox=l, dR=np.eye(3), dt=np.zeros(3)))
    ora = [oracle_pipeline(O, cs, None, None) for cs in cases]
    for B in (24, 2):
        c = M.Context(max_scans=B)
        try:
            for s in range(B):
                c.scan_upload(s, cases[s % 3]["velo"], cases[s % 3]["livox"])
            c.extract(0, B)
            for s in range(B):
                _check_extraction(c.scan_download(s), ora[s % 3])
            c.undistort(0, B, np.tile(np.eye(3).reshape(1, 9), (B, 1)), np.zeros((B, 3)))
            c.downsample(0, B)
            for s in range(B):
                assert c.features_download(s, 0).tobytes() == ora[s % 3]["corner"].tobytes()
                assert c.features_download(s, 1).tobytes() == ora[s % 3]["surf"].tobytes()
        finally:
            c.close()
    assert sum(int((o["label"] > 0).sum()) for o in ora) > 2000


def test_bucketing_block_boundaries_and_both_forms(M, O, synth):
    """The one-pass bucketing works in 4096-point blocks with a look-back over the blocks in front: scans whose sensors end exactly
    on, one short of and one past a block boundary, a scan of one block, one-sensor scans, and a long Livox part (16 blocks, the
    table limit) -- 20 slots (batch kernels; the second half of the slots holds the same scans in another order, so a block's
    predecessors differ) -- against the oracle.  Then the same scans through the three-pass form (MML_ASSIGN_ONEPASS=0, read when
    the context is created): byte-equal downloads."""
    import os
    v0, l0 = synth.velo_scan(60), synth.livox_scan(60, n=65536)
    shapes = [(4096, 4096), (4095, 4097), (4097, 4095), (8192, 12288), (28800, 65536), (1, 65535), (28799, 1), (0, 8193),
              (12289, 0), (4096 * 7, 4096 * 5)]
    cases = [dict(velo=v0[:nv] if nv else None, livox=l0[:nl] if nl else None, dR=np.eye(3), dt=np.zeros(3)) for nv, nl in shapes]
    ora = [oracle_pipeline(O, cs, None, None) for cs in cases]
    B = 2 * len(cases)
    order = list(range(len(cases))) + list(reversed(range(len(cases))))
    downloads = {}
    for form in ("1", "0"):
        os.environ["MML_ASSIGN_ONEPASS"] = form
        try:
            c = M.Context(max_scans=B, max_velo_points=28800, max_livox_points=65536)
        finally:
            del os.environ["MML_ASSIGN_ONEPASS"]
        try:
            for s in range(B):
                c.scan_upload(s, cases[order[s]]["velo"], cases[order[s]]["livox"])
            c.extract(0, B)
            got = [c.scan_download(s) for s in range(B)]
            for s in range(B):
                _check_extraction(got[s], ora[order[s]])
            c.undistort(0, B, np.tile(np.eye(3).reshape(1, 9), (B, 1)), np.zeros((B, 3)))
            c.downsample(0, B)
            downloads[form] = [(g["xyzi"].tobytes(), g["label"].tobytes(), c.features_download(s, 1).tobytes()) for s, g in enumerate(got)]
            for s in range(B):
                assert c.features_download(s, 1).tobytes() == ora[order[s]]["surf"].tobytes()
        finally:
            c.close()
    assert downloads["1"] == downloads["0"]


def test_voxel_grid_index_overflow_returns_the_labelled_cloud_unfiltered(M, O, synth):
    """pcl::VoxelGrid with more than INT_MAX voxels in its bounding box returns its input (Estimator.cpp:1015-1024 then hands the
    labelled points on unfiltered): a scene of 1.2 km extent at far_th = 100 km.  Slot 0 (25 k surf-labelled points) takes the
    global-sort filter, slot 1 (a shorter scan: 3.3 k surf points, corner box NOT overflowing) the LDS sort; both against the oracle."""
    cfg = M.default_config(2, far_th=100000.0, max_livox_points=64, max_features=32768)
    c = M.Context(cfg)
    try:
        scans = []
        for nkeep in (28800, 6000):
            v = synth.velo_scan(3, noise=0.002).copy()
            v[:, :3] *= 60.0
            scans.append(v[:nkeep])
        ev = [O.extract_velo(v, far=100000.0) for v in scans]
        for s, v in enumerate(scans):
            c.scan_upload(s, v, None)
        c.extract(0, 2)
        c.undistort(0, 2, np.tile(np.eye(3).reshape(1, 9), (2, 1)), np.zeros((2, 3)))
        c.downsample(0, 2)
        unfiltered = 0
        for s in range(2):
            xyz, lab = ev[s]["xyzi"][:, :3], ev[s]["label"]
            assert np.array_equal(c.scan_download(s)["label"], lab)
            for kind, leaf in ((0, 0.4), (1, 0.2)):
                want = O.voxel_downsample(xyz[lab == kind + 1], leaf)
                assert c.features_download(s, kind).tobytes() == want.tobytes()
                unfiltered += int(np.array_equal(want, xyz[lab == kind + 1]))
        assert unfiltered >= 3
    finally:
        c.close()


def test_azimuths_where_the_float_and_the_double_arctangent_differ(M, O, synth):
    """unionFeatureExtract.cpp:1136-1139,1168 call glibc's atan2f (the float overload, DESIGN.md section 2 convention 4): a scan
    salted with points on which atan2f and float(atan2(double)) -- what rounds 1-5 computed -- differ by an ulp, with ratios on
    atanf's reduction thresholds and octant edges (pairs of tests/golden/libm_f32_kat.npz, known answers of this image's libm).
    In-sweep times bit for bit as :1154-1186 form them, through the one-pass bucketing of a single scan (sweep ends found inline),
    of a batch (k_assign_ends), and through the three-pass bucketing of a 64-ring layout."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "libm_f32_kat.npz"))
    y, x, want = g["atan2_y"], g["atan2_x"], g["atan2_out"]
    r = np.hypot(x.astype(np.float64), y.astype(np.float64))
    d64 = np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(np.float32)
    pick = np.isfinite(want) & (r > 3) & (r < 40) & (d64 != want)
    edge = np.stack([x[pick], y[pick]], 1)[:400]
    assert len(edge) >= 300

    def salted(v, z_of):
        v = v.copy()
        idx = np.linspace(50, len(v) - 50, len(edge)).astype(int)
        v[idx, 0], v[idx, 1] = edge[:, 0], edge[:, 1]
        v[idx, 2] = np.hypot(edge[:, 0].astype(np.float64), edge[:, 1].astype(np.float64)) * np.tan(np.deg2rad(z_of))
        return v

    # default layout: 16 rings at -15 .. 15 degrees; the salted points sit on the +1 degree ring
    v16 = salted(synth.velo_scan(21), 1.0)
    l16 = synth.livox_scan(21)
    o16 = oracle_pipeline(O, dict(velo=v16, livox=l16, dR=np.eye(3), dt=np.zeros(3)), None, None)
    for B in (1, 20):
        c = M.Context(max_scans=B)
        try:
            for s in range(B):
                c.scan_upload(s, v16, l16)
            c.extract(0, B)
            for s in range(B):
                _check_extraction(c.scan_download(s), o16)
            c.extract(0, B)  # (a second extraction finds the same)
            _check_extraction(c.scan_download(B - 1), o16)
        finally:
            c.close()
    # 64 rings x 512 azimuths: the three-pass bucketing
    p0, st = -16.0, np.float32(32.0 / 63.0)
    kw = dict(n_rings=64, pitch0=p0, pitch_step=st)
    v64 = salted(synth.velo_scan(22, n_az=512, **kw), float(p0 + 30 * st))
    o64 = oracle_pipeline(O, dict(velo=v64, livox=None, dR=np.eye(3), dt=np.zeros(3)), None, None, **kw)
    for B in (1, 18):
        c = M.Context(max_scans=B, max_velo_points=64 * 512, max_livox_points=24000, n_rings=64, pitch0_deg=p0, pitch_step_deg=st)
        try:
            for s in range(B):
                c.scan_upload(s, v64, None)
            c.extract(0, B)
            for s in range(B):
                _check_extraction(c.scan_download(s), o64)
        finally:
            c.close()


def test_ring_ids_on_rounding_boundaries(M, O, synth):
    """`int((angle + 15) / 2 + 0.5)` (unionFeatureExtract.cpp:1159-1162) for pitches within a few float ulps of a ring boundary,
    where the ring id depends on every rounding of `atan(z / sqrt(x * x + y * y)) * 180 / M_PI` as the reference's float overloads
    evaluate it (sqrtf, a float division, glibc's atanf, a float product, a double division).  A third of a scan's points are
    moved onto boundaries (z chosen so that the oracle's own float pitch lands within 3 ulps of an even degree, both sides); ring
    ids, times and labels equal the oracle's in the one-pass and the three-pass bucketing.  The device decides such points with the
    reference expression (csrc/libm_f32.h), everything else from a fast estimate."""
    rng = np.random.default_rng(5)

    def on_boundaries(v, p0, step, n_rings):
        v = v.copy()
        idx = rng.choice(len(v), len(v) // 3, replace=False)
        x, y = v[idx, 0].astype(np.float64), v[idx, 1].astype(np.float64)
        k = rng.integers(0, n_rings + 1, len(idx))
        ang = np.deg2rad(float(p0) + (k - 0.5) * float(step))          # the boundary below ring k
        z = (np.hypot(x, y) * np.tan(ang)).astype(np.float32)
        z = (z.view(np.int32) + rng.integers(-3, 4, len(idx)).astype(np.int32)).view(np.float32)
        v[idx, 2] = z
        return v

    v16 = on_boundaries(synth.velo_scan(23), -15.0, 2.0, 16)
    l16 = synth.livox_scan(23)
    o16 = oracle_pipeline(O, dict(velo=v16, livox=l16, dR=np.eye(3), dt=np.zeros(3)), None, None)
    # (the salted points really sit on both sides of boundaries: drops at the table's two ends occur, not all of them)
    e16 = O.extract_velo(v16, near=0.0, far=1e9)
    assert len(v16) - len(v16) // 3 // 8 < len(e16["xyzi"]) < len(v16) - len(v16) // 3 // 64
    for B in (1, 20):
        c = M.Context(max_scans=B)
        try:
            for s in range(B):
                c.scan_upload(s, v16, l16)
            c.extract(0, B)
            for s in range(B):
                _check_extraction(c.scan_download(s), o16)
        finally:
            c.close()
    p0, st = -16.0, np.float32(32.0 / 63.0)
    kw = dict(n_rings=64, pitch0=p0, pitch_step=st)
    v64 = on_boundaries(synth.velo_scan(24, n_az=512, **kw), p0, st, 64)
    o64 = oracle_pipeline(O, dict(velo=v64, livox=None, dR=np.eye(3), dt=np.zeros(3)), None, None, **kw)
    for B in (1, 18):
        c = M.Context(max_scans=B, max_velo_points=64 * 512, max_livox_points=24000, n_rings=64, pitch0_deg=p0, pitch_step_deg=st)
        try:
            for s in range(B):
                c.scan_upload(s, v64, None)
            c.extract(0, B)
            for s in range(B):
                _check_extraction(c.scan_download(s), o64)
        finally:
            c.close()
