"""GPU suite, SURVEY.md 8(e) on ONE device: the 8-frame joint window (BASELINE configs[2]'s window size) against the
oracle, and the RCCL path of the C-ABI (mml_comm_* / mml_window_solve_allgather / the two broadcasts) at world size 1 --
ncclCommInitRank, ncclAllGather and ncclBroadcast really run, the state machine is the one N ranks execute, and with one
rank owning all W frames the result must equal mml_solve(window = W) bit for bit."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rsc

from conftest import perturbed, pose_to_x

pytestmark = pytest.mark.gpu

W8 = 8


def _window8(M, O, scene):
    """8 frames: the four scene frames twice, the second time from differently perturbed poses."""
    c = M.Context(max_scans=W8)
    c.map_set_local(0, scene["corner_map"])
    c.map_set_local(1, scene["surf_map"])
    tc, ts = O.KdTree(scene["corner_map"]), O.KdTree(scene["surf_map"])
    rng = np.random.default_rng(8)
    T, lfs, pfs = [], [], []
    for f in range(W8):
        fr = scene["frames"][f % 4]
        c.features_upload(f, 0, fr["corner"])
        c.features_upload(f, 1, fr["surf"])
        Tf = perturbed(fr["T_gt"], dt=rng.normal(0, 0.02, 3), rotvec=rng.normal(0, 0.003, 3))
        T.append(Tf)
        lfs.append(O.associate_lines(fr["corner"], tc, Tf, 1.0)[0])
        pfs.append(O.associate_planes(fr["surf"], ts, Tf, 1.0)[0])
    T = np.stack(T)
    assert c.associate(0, W8, T, 1.0, stats=False) is None     # enqueue only: the solves below follow on the same stream
    T_bl = np.eye(4)
    T_bl[:3, :3] = Rsc.from_rotvec([0.01, -0.02, 0.015]).as_matrix()
    T_bl[:3, 3] = [0.03, 0.01, -0.02]
    x0 = np.stack([pose_to_x(T[f]) for f in range(W8)])
    return c, lfs, pfs, x0, T_bl


@pytest.mark.parametrize("w_tan,huber", [(3e-4, 0.0), (0.0, 0.1 / 1.5e-3)])
def test_window8_joint_solve_matches_oracle(M, O, scene, w_tan, huber):
    c, lfs, pfs, x0, T_bl = _window8(M, O, scene)
    try:
        xg, sg, tg = c.solve(0, W8, x0, T_bl, window=W8, max_iters=10, huber=huber, w_tan=w_tan, trace=True)
        xo, so, to = O.solve_window(lfs, pfs, x0, T_bl, 10, huber=huber, w_tan=w_tan)
        assert (sg[0].iterations, sg[0].successful, sg[0].termination) == (so["iterations"], so["successful"], so["termination"])
        n = so["iterations"]
        assert n >= 2
        assert np.abs(tg[0][:n].reshape(n, W8, 6) - np.asarray(to).reshape(-1, W8, 6)[:n]).max() < 1e-9   # bar: 1e-4 per iteration
        assert np.abs(xg - xo).max() < 1e-9
        assert np.isclose(sg[0].final_cost, so["final_cost"], rtol=1e-10)
        # two 4-frame windows in one launch are independent problems
        xh, sh, _ = c.solve(0, W8, x0, T_bl, window=4, max_iters=10, huber=huber, w_tan=w_tan)
        for h in range(2):
            xo4, so4, _ = O.solve_window(lfs[4 * h:4 * h + 4], pfs[4 * h:4 * h + 4], x0[4 * h:4 * h + 4], T_bl, 10, huber=huber, w_tan=w_tan)
            assert sh[h].iterations == so4["iterations"] and np.abs(xh[4 * h:4 * h + 4] - xo4).max() < 1e-9
    finally:
        c.close()


def test_frame_parallel_window_solve_equals_the_sequential_kernel(M, O, scene):
    """mml_solve with window > 1 runs the frames of a window on W workgroups at once (rounds of k_window_round, the records
    crossing through a double buffer); with a trace requested it runs k_solve, one workgroup taking the frames one after
    the other.  Same functions, same records, same decisions: poses and summaries are equal bit for bit -- for every
    window size, batched problems, tolerances on and off, and starts far enough for rejected steps."""
    c, lfs, pfs, x0, T_bl = _window8(M, O, scene)
    try:
        rng = np.random.default_rng(4)
        seen_reject = False
        for trial in range(12):
            W = (2, 4, 8, 3)[trial % 4]
            count = (W8 // W) * W
            x = x0[:count].copy()
            if trial >= 4:
                x[:, :3] += rng.normal(0, 0.25, (count, 3))
                x[:, 3:] += rng.normal(0, 0.03, (count, 3))
            kw = dict(window=W, max_iters=(10, 4, 25)[trial % 3], fixed=bool(trial % 5 == 2), huber=(0.0, 0.1 / 1.5e-3)[trial % 2],
                      w_tan=(3e-4, 0.0)[trial % 2])
            xa, sa, _ = c.solve(0, count, x, T_bl, **kw)                    # frame-parallel
            xb, sb, _ = c.solve(0, count, x, T_bl, trace=True, **kw)        # sequential k_solve
            assert np.array_equal(xa, xb), (trial, np.abs(xa - xb).max())
            for a, b in zip(sa, sb):
                assert (a.iterations, a.successful, a.termination, a.initial_cost, a.final_cost) == \
                       (b.iterations, b.successful, b.termination, b.initial_cost, b.final_cost)
                seen_reject = seen_reject or a.successful < a.iterations
        assert seen_reject                                                  # rejected / invalid steps were part of it
    finally:
        c.close()


def test_window_solve_allgather_world_size_one(M, O, scene):
    c, lfs, pfs, x0, T_bl = _window8(M, O, scene)
    try:
        c.comm_init(1, 0, M.comm_unique_id())
        assert c.comm_info() == (1, 0)
        for fixed, huber, w_tan in ((False, 0.0, 3e-4), (True, 0.1 / 1.5e-3, 0.0)):
            xs, ss, _ = c.solve(0, W8, x0, T_bl, window=W8, max_iters=10, fixed=fixed, huber=huber, w_tan=w_tan)
            xl, xw, sm, tim = c.window_solve_allgather(0, W8, x0, T_bl, max_iters=10, fixed=fixed, huber=huber, w_tan=w_tan)
            assert np.array_equal(xw, xs) and np.array_equal(xl, xs)          # the same iteration, bit for bit
            assert (sm.iterations, sm.successful, sm.termination) == (ss[0].iterations, ss[0].successful, ss[0].termination)
            assert sm.initial_cost == ss[0].initial_cost and sm.final_cost == ss[0].final_cost
            assert tim.rounds == 12 and 1 <= tim.evaluations <= 11 and tim.device_ms > 0
            if fixed:
                assert sm.iterations == 10 and tim.evaluations == 11
        # one frame per rank is the 8-GPU layout; with one rank that is the W = 1 problem of slot 3
        xs, ss, _ = c.solve(3, 1, x0[3:4], T_bl, window=1, max_iters=10, huber=0.0, w_tan=3e-4)
        xl, xw, sm, tim = c.window_solve_allgather(3, 1, x0[3:4], T_bl, max_iters=10, huber=0.0, w_tan=3e-4)
        assert np.array_equal(xl, xs) and sm.iterations == ss[0].iterations
        xo, so, _ = O.solve_window(lfs[3:4], pfs[3:4], x0[3:4], T_bl, 10, huber=0.0, w_tan=3e-4)
        assert np.abs(xl - xo).max() < 1e-9
        # broadcasts with a single rank leave everything as it was (and must not hang or fault)
        f0 = [c.features_download(2, k).copy() for k in (0, 1)]
        c.comm_broadcast_features(2, 0)
        c.synchronize()
        for k in (0, 1):
            assert np.array_equal(c.features_download(2, k), f0[k])
        q = scene["frames"][0]["surf"][:64]
        i0, d0 = c.knn5(1, q)
        c.comm_broadcast_local_map(0)
        i1, d1 = c.knn5(1, q)
        assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
        with pytest.raises(M.MmlError):
            c.window_solve_allgather(0, 9, np.zeros((9, 6)), T_bl)
        c.comm_destroy()
        with pytest.raises(M.MmlError):
            c.comm_info()
    finally:
        c.close()


def _rank_contexts(M, scene, n_ranks, n_local, x0, first_slots):
    """The 8-frame window of _window8 split over n_ranks contexts on this device: rank r holds window frames
    [r n_local, (r + 1) n_local) in ITS slots first_slots[r] ..., associated at the window's starting poses."""
    ctxs = []
    rng = np.random.default_rng(8)
    T = [perturbed(scene["frames"][f % 4]["T_gt"], dt=rng.normal(0, 0.02, 3), rotvec=rng.normal(0, 0.003, 3)) for f in range(W8)]
    assert np.allclose(np.stack([pose_to_x(t) for t in T]), x0)          # the same frames and poses as _window8
    for r in range(n_ranks):
        c = M.Context(max_scans=first_slots[r] + n_local)
        c.map_set_local(0, scene["corner_map"])
        c.map_set_local(1, scene["surf_map"])
        for k in range(n_local):
            f = r * n_local + k
            fr = scene["frames"][f % 4]
            c.features_upload(first_slots[r] + k, 0, fr["corner"])
            c.features_upload(first_slots[r] + k, 1, fr["surf"])
        c.associate(first_slots[r], n_local, np.stack(T[r * n_local:(r + 1) * n_local]), 1.0, stats=False)
        ctxs.append(c)
    return ctxs


@pytest.mark.parametrize("n_ranks,n_local", [(2, 4), (8, 1), (4, 2)])
def test_window_solve_ranks_on_one_device(M, O, scene, n_ranks, n_local):
    """The rank > 0 side of mml_window_solve_allgather on a single GPU: N contexts as the N ranks of a loopback group (the
    all-gather = device copies between the ranks' gather buffers; kernels, buffers, state machine as in the RCCL path).
    Every rank linearises only ITS frames into ITS section of the record buffer, all advance the same dogleg -- and every
    rank must end with the poses of mml_solve(window = 8) on one context, bit for bit (Estimator.cpp:1265-1299,1425-1432)."""
    ref, lfs, pfs, x0, T_bl = _window8(M, O, scene)
    first_slots = [(r * 3) % 2 for r in range(n_ranks)]                  # ranks keep their frames in different slots
    ctxs = _rank_contexts(M, scene, n_ranks, n_local, x0, first_slots)
    try:
        grp = M.LoopbackGroup(ctxs)
        assert [c.comm_info() for c in ctxs] == [(n_ranks, r) for r in range(n_ranks)]
        for fixed, huber, w_tan in ((False, 0.0, 3e-4), (True, 0.1 / 1.5e-3, 0.0)):
            xs, ss, _ = ref.solve(0, W8, x0, T_bl, window=W8, max_iters=10, fixed=fixed, huber=huber, w_tan=w_tan)
            xr, sr = grp.window_solve(first_slots, n_local, x0, T_bl, max_iters=10, fixed=fixed, huber=huber, w_tan=w_tan)
            for r in range(n_ranks):
                assert np.array_equal(xr[r], xs), (r, np.abs(xr[r] - xs).max())
                assert (sr[r].iterations, sr[r].successful, sr[r].termination) == (ss[0].iterations, ss[0].successful, ss[0].termination)
                assert sr[r].initial_cost == ss[0].initial_cost and sr[r].final_cost == ss[0].final_cost
        # a start far enough away for rejected steps
        xf = x0.copy()
        xf[:, :3] += np.random.default_rng(3).normal(0, 0.25, (W8, 3))
        xs, ss, _ = ref.solve(0, W8, xf, T_bl, window=W8, max_iters=25, huber=0.0, w_tan=3e-4)
        xr, sr = grp.window_solve(first_slots, n_local, xf, T_bl, max_iters=25, huber=0.0, w_tan=3e-4)
        assert all(np.array_equal(xr[r], xs) for r in range(n_ranks)) and sr[-1].iterations == ss[0].iterations
        # the RCCL entry point refuses a loopback group instead of touching a null communicator
        with pytest.raises(M.MmlError):
            ctxs[0].window_solve_allgather(first_slots[0], n_local, x0[:n_local], T_bl)
    finally:
        for c in ctxs:
            c.close()
        ref.close()


def test_broadcasts_from_another_rank_on_one_device(M, O, scene):
    """mml_comm_broadcast_features / _local_map with root != rank (Estimator.cpp:1083-1085, 1125-1130: the key scan's features
    and the local map reach every replica): three ranks, root 1; ranks 0 and 2 start without a map."""
    ctxs = [M.Context(max_scans=3) for _ in range(3)]
    try:
        root = ctxs[1]
        root.map_set_local(0, scene["corner_map"])
        root.map_set_local(1, scene["surf_map"])
        fr = scene["frames"][1]
        root.features_upload(2, 0, fr["corner"])
        root.features_upload(2, 1, fr["surf"])
        other = scene["frames"][2]
        for c in (ctxs[0], ctxs[2]):                                      # something else in the slot: it must be replaced
            c.features_upload(2, 0, other["corner"][:50])
            c.features_upload(2, 1, other["surf"][:70])
        grp = M.LoopbackGroup(ctxs)
        grp.broadcast_features(2, 1)
        for c in ctxs:
            assert np.array_equal(c.features_download(2, 0), fr["corner"]) and np.array_equal(c.features_download(2, 1), fr["surf"])
        with pytest.raises(M.MmlError):
            grp.broadcast_local_map(0)                                    # rank 0 has no map to give
        grp.broadcast_local_map(1)
        q = fr["surf"][:200]
        i1, d1 = root.knn5(1, q)
        ic, dc = root.knn5(0, fr["corner"][:100])
        for c in (ctxs[0], ctxs[2]):
            i, d = c.knn5(1, q)
            assert np.array_equal(i, i1) and np.array_equal(d, d1)
            i, d = c.knn5(0, fr["corner"][:100])
            assert np.array_equal(i, ic) and np.array_equal(d, dc)
        # and the replicas register like the root does
        T = np.stack([perturbed(fr["T_gt"], dt=[0.02, -0.01, 0.01], rotvec=[0.002, 0.001, -0.003])])
        st = [c.associate(2, 1, T, 1.0) for c in ctxs]
        assert st[0][0].n_line == st[1][0].n_line == st[2][0].n_line and st[0][0].n_plane == st[1][0].n_plane == st[2][0].n_plane
    finally:
        for c in ctxs:
            c.close()
