"""Developer diagnostic: run every stage on the GPU and diff against the oracle, verbosely.
(Not a collected test: the test_*.py files hold the assertions.  It lives under tests/ because it calls the oracle.
Usage on the GPU box: python tests/dev_gpu_check.py)"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mml_oracle as O  # noqa: E402

M = importlib.import_module("multi-modal-loam_amd")
synth = importlib.import_module("multi-modal-loam_amd.synth")


def main():
    B = 4
    ctx = M.Context(max_scans=B)
    print("device:", ctx.device_info())
    # ---- detect_line ----
    v = synth.velo_scan(3)
    for ring in (0, 3, 8, 15):
        line = v.reshape(1800, 16, 4)[:, ring, :]
        s0, f0, fl0 = O.detect_feature_points(line)
        s1, f1, fl1 = ctx.detect_line(line)
        print("ring", ring, "sharp eq", np.array_equal(s0, s1), "flat eq", np.array_equal(f0, f1), "flags eq",
              np.array_equal(fl0, fl1), len(s0), len(f0))
        if not np.array_equal(fl0, fl1):
            bad = np.nonzero(fl0 != fl1)[0]
            print("   first diffs", bad[:10], fl0[bad[:10]], fl1[bad[:10]])
    # ---- extract ----
    scans = []
    for k in range(B):
        vk, lk = synth.velo_scan(10 + k), synth.livox_scan(10 + k)
        scans.append((vk, lk))
        ctx.scan_upload(k, vk, lk)
    t = time.time()
    ctx.extract(0, B)
    ctx.synchronize()
    print("extract wall ms", (time.time() - t) * 1e3)
    fused = []
    for k in range(B):
        g = ctx.scan_download(k)
        ev = O.extract_velo(scans[k][0])
        el = O.extract_livox(scans[k][1])
        oxyz = np.concatenate([ev["xyzi"], el["xyzi"]])
        olab = np.concatenate([ev["label"], el["label"]])
        orel = np.concatenate([ev["reltime"], el["reltime"]])
        oring = np.concatenate([ev["ring"], el["ring"]])
        info = g["info"]
        print("scan", k, "n", info.n_points, len(oxyz), "counts", (info.velo_corner_num, info.velo_surf_num,
              info.livox_corner_num, info.livox_surf_num), (ev["n_corner"], ev["n_surf"], el["n_corner"], el["n_surf"]))
        if info.n_points == len(oxyz):
            print("   xyzi eq", np.array_equal(g["xyzi"], oxyz), "label eq", np.array_equal(g["label"], olab), "ring eq",
                  np.array_equal(g["ring"], oring), "rel eq", np.array_equal(g["reltime"], orel),
                  "rel maxdiff", np.abs(g["reltime"] - orel).max())
            if not np.array_equal(g["label"], olab):
                bad = np.nonzero(g["label"] != olab)[0]
                print("   label diffs", len(bad), bad[:10], g["label"][bad[:10]], olab[bad[:10]])
        fused.append((oxyz, olab, orel))
    # ---- undistort ----
    from scipy.spatial.transform import Rotation as Rsc
    dR = np.stack([Rsc.from_rotvec([0.001 * (k + 1), -0.002, 0.02]).as_matrix() for k in range(B)])
    dt = np.stack([[0.05, 0.002 * k, -0.001] for k in range(B)])
    ctx.undistort(0, B, dR, dt)
    und = []
    for k in range(B):
        g = ctx.scan_download(k)
        o = O.undistort(fused[k][0][:, :3], fused[k][2], dR[k], dt[k])
        d = np.abs(g["xyzi"][:, :3] - o)
        print("undistort", k, "exact frac", np.mean(g["xyzi"][:, :3] == o), "max abs", d.max())
        und.append(o)
    # ---- downsample ----
    ctx.downsample(0, B)
    feats = []
    for k in range(B):
        fc = ctx.features_download(k, 0)
        fs = ctx.features_download(k, 1)
        oc = O.voxel_downsample(und[k][fused[k][1] == 1], 0.4)
        os_ = O.voxel_downsample(und[k][fused[k][1] == 2], 0.2)
        print("voxel", k, fc.shape, oc.shape, fs.shape, os_.shape, "eq", np.array_equal(fc, oc), np.array_equal(fs, os_))
        feats.append((oc, os_))
    # ---- map + knn ----
    cm, sm = [], []
    for k in range(0, 8):
        ev = O.extract_velo(synth.velo_scan(k))
        el = O.extract_livox(synth.livox_scan(k))
        xyz = np.concatenate([ev["xyzi"][:, :3], el["xyzi"][:, :3]])
        lab = np.concatenate([ev["label"], el["label"]])
        T = synth.pose_matrix(k)
        cm.append(synth.transform(T, O.voxel_downsample(xyz[lab == 1], 0.4).astype(np.float64)).astype(np.float32))
        sm.append(synth.transform(T, O.voxel_downsample(xyz[lab == 2], 0.2).astype(np.float64)).astype(np.float32))
    cm = O.voxel_downsample(np.concatenate(cm), 0.4)
    sm = O.voxel_downsample(np.concatenate(sm), 0.2)
    print("maps", cm.shape, sm.shape)
    ctx.map_set_local(0, cm)
    ctx.map_set_local(1, sm)
    rng = np.random.default_rng(0)
    q = (sm[rng.integers(0, len(sm), 500)] + rng.normal(0, 0.3, (500, 3))).astype(np.float32)
    gi, gd = ctx.knn5(1, q)
    oi, od = O.bruteforce_knn5(sm, q)
    print("knn idx eq", np.array_equal(gi, oi), "d2 eq", np.array_equal(gd, od))
    if not np.array_equal(gi, oi):
        bad = np.nonzero((gi != oi).any(axis=1))[0]
        print("  bad", len(bad), gi[bad[0]], oi[bad[0]], gd[bad[0]], od[bad[0]])
    # ---- associate ----
    T_wl = np.stack([synth.pose_matrix(10 + k) for k in range(B)])
    for k in range(B):  # perturb
        T_wl[k][:3, 3] += [0.03, -0.02, 0.01]
    st = ctx.associate(0, B, T_wl, 25.0)
    tc, ts = O.KdTree(cm), O.KdTree(sm)
    for k in range(B):
        lf, lsrc = O.associate_lines(feats[k][0], tc, T_wl[k], 25.0)
        pf, psrc = O.associate_planes(feats[k][1], ts, T_wl[k], 25.0)
        gl, glsrc = ctx.factors_download(k, 0)
        gp, gpsrc = ctx.factors_download(k, 1)
        print("assoc", k, "lines", len(lf), len(gl), "planes", len(pf), len(gp), "stats", st[k].n_line, st[k].n_plane,
              st[k].min_singular, O.check_localizability(pf))
        if len(lf) == len(gl) and len(lf):
            ol = np.concatenate([lf["point_ori"], lf["p1"], lf["p2"], lf["error"][:, None]], axis=1)
            print("   line src eq", np.array_equal(lsrc, glsrc), "max diff", np.abs(ol - gl).max(), "exact", np.array_equal(ol, gl))
        if len(pf) == len(gp) and len(pf):
            op = np.concatenate([pf["point_ori"], pf["point_proj"], pf["omega"], pf["error"][:, None]], axis=1)
            print("   plane src eq", np.array_equal(psrc, gpsrc), "max diff", np.abs(op - gp).max(), "exact", np.array_equal(op, gp))
    # ---- linearize / solve ----
    T_bl = np.eye(4)
    for k in range(B):
        lf, _ = O.associate_lines(feats[k][0], tc, T_wl[k], 25.0)
        pf, _ = O.associate_planes(feats[k][1], ts, T_wl[k], 25.0)
        R = T_wl[k][:3, :3]
        x0 = np.concatenate([T_wl[k][:3, 3], Rsc.from_matrix(R).as_rotvec()])
        Ho, go, co = O.linearize(lf, pf, x0, T_bl, 0.0, 0.1 / 1.5e-3)
        Hg, gg, cg = ctx.linearize(k, x0, T_bl)
        print("linearize", k, "cost", co, cg, "H rel", np.abs(Ho - Hg).max() / np.abs(Ho).max(), "g rel",
              np.abs(go - gg).max() / np.abs(go).max())
        xo, so, tro = O.solve_window([lf], [pf], x0[None], T_bl, 10)
        xg, sg, trg = ctx.solve(k, 1, x0[None], T_bl, trace=True)
        print("   solve oracle", so, "gpu", (sg[0].iterations, sg[0].successful, sg[0].initial_cost, sg[0].final_cost,
              sg[0].termination), "x diff", np.abs(xo - xg).max())
    # ---- estimate ----
    P0 = np.stack([T_wl[k][:3, 3] for k in range(B)])
    Q0 = np.stack([Rsc.from_matrix(T_wl[k][:3, :3]).as_quat() for k in range(B)])
    Pg, Qg, info = ctx.estimate(0, B, np.eye(4), P0, Q0)
    for k in range(B):
        Po, Qo, it, deg, _ = O.estimate_single(feats[k][0], feats[k][1], cm, sm, np.eye(4), P0[k], Q0[k])
        print("estimate", k, "outer", it, info[k].outer_iterations, "dP", np.abs(Po - Pg[k]).max(), "dQ", np.abs(Qo - Qg[k]).max(),
              "gt err", np.abs(Pg[k] - synth.pose_matrix(10 + k)[:3, 3]).max())
    # ---- step timing ----
    ctx.profile_enable(True)
    x0 = np.stack([np.concatenate([T_wl[k][:3, 3], Rsc.from_matrix(T_wl[k][:3, :3]).as_rotvec()]) for k in range(B)])
    for rep in range(3):
        t = time.time()
        x = ctx.step(0, B, dR, dt, np.eye(4), 25.0, 10, x0)
        print("step wall ms", (time.time() - t) * 1e3)
    pr = ctx.profile_get()
    for k, (ms, n) in pr.items():
        print("  %-18s %8.3f ms / %d launches" % (k, ms, n))
    print("copy bw GB/s", ctx.copy_bandwidth(1 << 28, 5))


if __name__ == "__main__":
    main()
