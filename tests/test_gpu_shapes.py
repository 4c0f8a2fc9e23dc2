"""GPU parity at the LAUNCH SHAPES of the benchmark: thousands of slots per mml_step call on the library's default stream lanes.

The other batch tests (test_gpu_batch.py, test_gpu_sensor.py) put the large-batch kernel variants under the oracle at 72-96 slots;
bench.py runs 4096 slots per lane.  Code that first executes more than once -- or at all -- beyond a few hundred slots per call:
  * k_assoc_prefix: the loop over tiles of 1024 (kind, slot) items (map_assoc.hip, 2 * count > 1024)
  * the fixed-grid work lists of k_associate / k_associate_fit_all (4096 x 128 threads walking > 1 M features), the far-query
    list k_associate_hard / k_associate_fit walk at that fill
  * list-mode k_select beyond its 256 / 128 row caps (feature.hip MML_SEL_ROWS_V / _L): thousands of listed lines per launch
  * gridDim.y = 2048 (two lanes) and 4096 (one lane) in every per-slot grid
Here: 4096 resident slots of the configs[1] layout (fused 52.8 k-point scans, 200 k-point map), 20 DISTINCT scans -- the twelve of
batch_cases (clean, motion, dirty, one-sensor, tiny, ragged lines, rough scenes) and eight sensor-faithful ones -- dealt to the
slots at random.  The oracle runs 20 times, not 4096: every distinct scan's first slot is compared with oracle_pipeline as
_check_after_step does (labels, undistorted cloud, stacks, factor records, pose), a sample of far slots likewise, and EVERY
replica must be bit-identical to that first slot in all ten pieces of mml_slot_digest (counts, labels, rings, cloud, times, both
stacks, both factor lists, pose) -- the digest itself being recomputed on the host from the download entry points for the
compared slots.  Reference functions at stake: unionFeatureExtract.cpp:453-541 (selection), Estimator.cpp:148-365,573-777
(association + model fit), :992-1026 (stacks)."""
import time

import numpy as np
import pytest

from conftest import batch_cases, host_digest, oracle_pipeline
from test_gpu_batch import _check_after_step
from test_gpu_sensor import NV, sensor_case

pytestmark = pytest.mark.gpu

PIECES = ("counts", "label", "line", "xyzi", "reltime", "corner stack", "surf stack", "line factors", "plane factors", "pose")


def test_bench_launch_shapes_replicas_match_oracle(M, O, synth, scene):
    B = 4096
    t0 = time.time()
    cases = batch_cases(synth)
    modes = ("skip", "nan", "zero")
    cases += [sensor_case(synth, 60 + j, modes[j % 3], motion=(j % 4 == 1)) for j in range(8)]
    ND = len(cases)
    assert ND == 20
    cm = synth.grow_map(scene["corner_map"], 40000, seed=7)
    sm = synth.grow_map(scene["surf_map"], 160000, seed=8)
    tc, ts = O.KdTree(cm), O.KdTree(sm)
    ora = [oracle_pipeline(O, cs, tc, ts) for cs in cases]
    rng = np.random.default_rng(2024)
    assign = rng.integers(0, ND, B)
    assign[:ND] = np.arange(ND)              # every scan occurs; its first slot is in [0, ND)
    assign[B - ND:] = np.arange(ND)[::-1]    # ... and every scan also sits in the last wavefronts' worth of slots
    c = M.Context(max_scans=B, max_velo_points=NV, max_livox_points=24000, max_map_points=200000)
    try:
        c.map_set_local(0, cm)
        c.map_set_local(1, sm)
        for s in range(B):
            cs = cases[assign[s]]
            c.scan_upload(s, cs["velo"], cs["livox"])
        dR = np.stack([cases[k]["dR"].reshape(9) for k in assign])
        dt = np.stack([cases[k]["dt"] for k in assign])
        x0 = np.stack([cases[k]["x0"] for k in assign])
        # (1) the library's DEFAULT lanes (no set_lanes call): two lanes of 2048 slots
        x = c.step(0, B, dR, dt, np.eye(4), 25.0, 10, x0)
        dg = c.slot_digest(0, B)
        first = np.array([int(np.argmax(assign == k)) for k in range(ND)])
        mism = []
        for s in range(B):
            f = first[assign[s]]
            if not np.array_equal(dg[s], dg[f]) or not np.array_equal(x[s], x[f]):
                mism.append((s, int(assign[s]), [PIECES[w] for w in range(10) if dg[s][w] != dg[f][w]]))
        assert not mism, "replicas differ from the first slot of their scan: %d slots, e.g. %s" % (len(mism), mism[:5])
        # the first slot of every distinct scan, and 12 replicas spread over the call (incl. the last slot), against the oracle;
        # the digest of each of them recomputed on the host from the downloads
        far = sorted(set([B - 1, B // 2 - 1, B // 2, 1023, 1024, 2049] + list(rng.integers(ND, B, 6))))
        for s in list(first) + far:
            _check_after_step(c, s, ora[assign[s]], x[s])
            hd = host_digest(c, s, x[s])
            assert np.array_equal(hd, dg[s]), (s, [PIECES[w] for w in range(10) if hd[w] != dg[s][w]])
        # distinct scans have distinct digests (the digest is not blind)
        for w in (1, 3, 5, 6, 9):
            assert len(set(int(v) for v in dg[first, w])) >= ND - 2, PIECES[w]
        # (2) one lane: every per-slot grid at gridDim.y = 4096; same results, slot for slot
        c.set_lanes(1)
        x1 = c.step(0, B, dR, dt, np.eye(4), 25.0, 10, x0)
        dg1 = c.slot_digest(0, B)
        assert np.array_equal(x1, x) and np.array_equal(dg1, dg)
        # (3) ONE slot at a time -- the live path's kernel variants (workgroup-per-line k_select, tile-per-wavefront k_stencil, the
        #     combined bucketing launch with the sweep's ends found inline, the 16-lane group search of <= 8 slots, the single
        #     1024-thread k_voxel launch): every distinct scan's first slot and three far slots stepped alone must come out
        #     bit-identical, digest word for digest word and pose for pose, to what the 4096-slot launches left there
        for s in list(first) + [B - 1, B // 2, 1023]:
            xs = c.step(int(s), 1, dR[s:s + 1], dt[s:s + 1], np.eye(4), 25.0, 10, x0[s:s + 1])
            d1 = c.slot_digest(int(s), 1)[0]
            assert np.array_equal(d1, dg[s]), (int(s), int(assign[s]), [PIECES[w] for w in range(10) if d1[w] != dg[s][w]])
            assert np.array_equal(xs[0], x[s]), (int(s), xs[0] - x[s])
        # (4) launches of at most one problem per CU (k_solve_wide: the factor pass on 4 x 128 threads, the sums and the reduction
        #     tree of the 128-thread pass): 200 and 17 slots at once, same bits as the 4096-slot launches
        for s0, n in ((1000, 200), (B - 17, 17)):
            xs = c.step(s0, n, dR[s0:s0 + n], dt[s0:s0 + n], np.eye(4), 25.0, 10, x0[s0:s0 + n])
            dn = c.slot_digest(s0, n)
            assert np.array_equal(dn, dg[s0:s0 + n]), (s0, n, np.argwhere(dn != dg[s0:s0 + n])[:5])
            assert np.array_equal(xs, x[s0:s0 + n]), (s0, n)
        # (5) the small launches repeated: 150 times 200 slots and 150 times one slot -- every repetition the bits of the first (the
        #     wide solve keeps its rows, sums and trust-region state in LDS across barriers that were cut to the minimum: a race
        #     there would show as a result that moves)
        s0, n = 1000, 200
        for rep in range(150):
            xs = c.step(s0, n, dR[s0:s0 + n], dt[s0:s0 + n], np.eye(4), 25.0, 10, x0[s0:s0 + n])
            assert np.array_equal(xs, x[s0:s0 + n]), (rep, np.argwhere(xs != x[s0:s0 + n])[:4])
            s1 = int(first[rep % ND])
            x1s = c.step(s1, 1, dR[s1:s1 + 1], dt[s1:s1 + 1], np.eye(4), 25.0, 10, x0[s1:s1 + 1])
            assert np.array_equal(x1s[0], x[s1]), (rep, s1)
        assert np.array_equal(c.slot_digest(s0, n), dg[s0:s0 + n])
        print("launch shapes: %d slots, %d distinct scans, 0 mismatching replicas, %.1f s" % (B, ND, time.time() - t0))
    finally:
        c.close()


def test_replica_stress(M, O, synth):
    """The 4096-slot step repeated: 30 rounds on the library's default lanes, then 10 alternating between one lane and two --
    every replica equal to the first slot of its scan in all ten digest words and in its pose, every round equal to the first
    (tests/replica_stress.py; round 5's suite ran the step twice and one run of it met a slot whose surf stack did not match,
    with the since-removed undistortion fork on -- HISTORY.md A.000)."""
    import replica_stress
    bad, per = replica_stress.run(M, O, synth, 40, lanes=[None] * 30 + [1, 2] * 5)
    assert bad == 0
    print("replica stress: %.3f s per round" % per)


def test_wide_solve_long_factor_lists_equal_batch_kernel(M, O, scene):
    """k_solve_wide (launches of at most one one-frame problem per CU) against the batch kernel k_solve<false> (more problems than
    CUs) on factor lists LONGER than one round of its row exchange (12 x 128 factors per round: 2 300 line + 3 700 plane records
    are four rounds, the sums parked in LDS between them), with skipped records (|error| <= 1e-5, Estimator.cpp:1385,1396) spread
    through both lists, ragged ends, an empty line list, an empty problem, and with / without the Huber loss, fixed and free
    iteration counts: poses, iteration counts and terminations bit for bit; one problem of each setting against the oracle."""
    from conftest import perturbed, pose_to_x
    B = 320
    c = M.Context(max_scans=B, max_features=4096)
    try:
        tc, ts = O.KdTree(scene["corner_map"]), O.KdTree(scene["surf_map"])
        fr = scene["frames"][1]
        T = perturbed(fr["T_gt"])
        lf, _ = O.associate_lines(fr["corner"], tc, T, 25.0)
        pf, _ = O.associate_planes(fr["surf"], ts, T, 25.0)
        rng = np.random.default_rng(77)

        def long_list(a, n):
            out = np.concatenate([a] * (n // len(a) + 1))[:n].copy()
            out["point_ori"] = (out["point_ori"] + rng.normal(0, 0.02, out["point_ori"].shape)).astype(np.float32)  # distinct rows (floats, as the feature clouds)
            out["error"] = np.where(np.abs(out["error"]) > 1e-5, out["error"], 0.25)
            out["error"][rng.integers(0, n, n // 9)] = 0.0                                                  # skipped records
            return out

        arr_l = lambda a: np.concatenate([a["point_ori"], a["p1"], a["p2"], a["error"][:, None]], axis=1)
        arr_p = lambda a: np.concatenate([a["point_ori"], a["point_proj"], a["omega"], a["error"][:, None]], axis=1)
        x0 = pose_to_x(T)
        cases = [(long_list(lf, 2300), long_list(pf, 3700)),      # 18 + 29 slots a thread: four rounds
                 (lf[:0], long_list(pf, 1537)),                   # no line factor; 13 slots: two rounds, the second nearly empty
                 (long_list(lf, 129), long_list(pf, 1)),          # ragged: a second line slot of one factor, one plane factor
                 (lf, pf),                                        # the association's own lists (one round)
                 (lf[:0], pf[:0])]                                # nothing at all: the pose stays where it is
        for huber, fixed in ((0.1 / 1.5e-3, True), (0.0, False), (0.1 / 1.5e-3, False)):
            for ci, (l2, p2) in enumerate(cases):
                for s in range(B):
                    if s < 3 or ci == 0 and s < 40 or s == B - 1:   # (the slots compared below; the rest keep what they hold)
                        c.factors_upload(s, 0, arr_l(l2))
                        c.factors_upload(s, 1, arr_p(p2))
                xs = np.tile(x0, (B, 1))
                xb, sb, _ = c.solve(0, B, xs, np.eye(4), max_iters=10, fixed=fixed, huber=huber)          # k_solve<false>
                x1, s1, _ = c.solve(0, 1, xs[:1], np.eye(4), max_iters=10, fixed=fixed, huber=huber)      # k_solve_wide, one problem
                x3, s3, _ = c.solve(0, 3, xs[:3], np.eye(4), max_iters=10, fixed=fixed, huber=huber)      # ... three
                xl, sl, _ = c.solve(B - 1, 1, xs[:1], np.eye(4), max_iters=10, fixed=fixed, huber=huber)
                key = lambda q: (q.iterations, q.successful, q.termination, q.initial_cost, q.final_cost)
                assert np.array_equal(x1[0], xb[0]) and key(s1[0]) == key(sb[0]), (ci, huber, fixed, x1[0] - xb[0])
                assert np.array_equal(x3, xb[:3]) and [key(q) for q in s3] == [key(q) for q in sb[:3]], (ci, huber, fixed)
                assert np.array_equal(xl[0], xb[B - 1]) and key(sl[0]) == key(sb[B - 1]), (ci, huber, fixed)
                if ci == 0:
                    x40, s40, _ = c.solve(0, 40, xs[:40], np.eye(4), max_iters=10, fixed=fixed, huber=huber)
                    assert np.array_equal(x40, xb[:40]) and [key(q) for q in s40] == [key(q) for q in sb[:40]], (huber, fixed)
                if len(l2) + len(p2) == 0:
                    assert np.array_equal(x1[0], x0)
                elif ci in (0, 3):
                    xo, so, _ = O.solve_window([l2], [p2], x0[None], np.eye(4), 10, fixed=fixed, huber=huber, w_tan=0.0)
                    assert np.abs(x1 - xo).max() < 1e-9 and s1[0].iterations == so["iterations"], (ci, huber, fixed, np.abs(x1 - xo).max())
    finally:
        c.close()
