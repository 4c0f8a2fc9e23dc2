"""Randomised parity campaign on a real device: the same comparisons as tests/test_gpu_parity.py (device through the C-ABI
against the CPU oracle), over many fresh seeds instead of the few the suite pins.  Not collected by pytest (run by hand).

    python tests/gpu_fuzz.py --seed 1 --lines 2000 --scans 60 --poses 40

Sections: `lines` = detectFeaturePoints on randomised scan lines (flags and both index lists bit for bit);
`scans` = whole fused scans with dirt (NaN, rings out of range, near / far crops, truncation) through mml_extract,
then undistort with a random sweep motion and the voxel down-sample; `poses` = association + Estimate from random
pose perturbations; `cubes` = random trajectories through the global cube store; `dense` = other ring layouts / scans beyond 64 k points / labelled clouds beyond the LDS sort; `solves` = factor records of random associations and the lidar-only window solve; `maps` = random walks of key scans through the local-map upkeep; `batch` (--batch N) = N rounds of 96 slots of fresh random scans -- ideal grid with dirt and sensor-faithful streams -- through the large-batch kernels and mml_step; `dense-batch` (--dense-batch N) = N rounds of 24 slots of random 64 / 128-ring scans through the dense layout's batch kernels; `windows` = the full-window (IMU factors, prior) trust-region loop on the device against the host
loop on random window sizes, missing factors, iteration limits.  Prints one line per section and exits non-zero at the first mismatch (the offending seed / trial
is printed so that it can be replayed)."""
import argparse
import importlib
import os
import sys
import time

import numpy as np
from scipy.spatial.transform import Rotation as Rsc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def random_dirt(rng, v, l):
    """The dirt of the `scans` section: NaN runs, rings out of range, near / far crops, repeated points, bad Livox records."""
    for _ in range(int(rng.integers(0, 6))):
        kind = int(rng.integers(0, 8))
        a = int(rng.integers(0, max(1, len(v) - 400)))
        n = int(rng.integers(1, 400))
        if kind == 0:
            v[a:a + n, int(rng.integers(0, 3))] = np.nan
        elif kind == 1:
            v[a::int(rng.integers(50, 200)), 2] = rng.uniform(20, 80)
        elif kind == 2:
            v[a:a + n, :3] *= rng.uniform(0.01, 0.2)
        elif kind == 3:
            v[a:a + n, :3] *= rng.uniform(5, 40)
        elif kind == 4:
            l["line"][int(rng.integers(0, 50))::int(rng.integers(20, 90))] = int(rng.integers(6, 9))
        elif kind == 5:
            l["x"][int(rng.integers(0, 50))::int(rng.integers(20, 90))] = rng.uniform(-1, 0.01)
        elif kind == 6:
            v[a:a + n, :3] = 0.0                        # (0,0,0) no-return records
        else:
            v[a:a + n] = v[a]
    return v, l


def plane_records_close(g, o):
    """Plane factor records (point_ori, point_proj, omega, error) against the oracle's: <= 1e-9 in every field -- or, for a record whose
    omega is one FLOAT ulp off in some component, <= 1e-6 in the fields computed from it.  omega is `float pa = X[0]` of the plane fit's
    double solution (Estimator.cpp:722-738) normalised in float: the oracle's and the device's QR agree on X to ~1e-16 relative, so the
    cast rounds the other way for about one value in 1e8 -- campaign seed 930 met the first one in 1.3e8 plane records (round 6,
    identical on the libraries before and after that round's changes: pose difference 8.5e-14).  Returns (ok, flipped records)."""
    if len(g) != len(o):
        return False, 0
    if len(g) == 0:
        return True, 0
    d = np.abs(g - o)
    bad = np.flatnonzero(d.max(axis=1) > 1e-9)
    if len(bad) == 0:
        return True, 0
    og, oo = g[bad, 6:9].astype(np.float32), o[bad, 6:9].astype(np.float32)
    one_ulp = np.abs(og - oo) <= np.spacing(np.maximum(np.abs(og), np.abs(oo)))
    return bool(np.all(one_ulp) and np.all(d[bad] <= 1e-6)), len(bad)


def batch_section(args, M, O, synth, rng):
    """Large batches -- the kernels bench.py runs: k_stencil<0>, batch k_select_part + k_select_list, k_voxel<256> / <1024>, the
    lane-per-feature search, mml_step on 4 / 1 / 2 stream lanes by turns -- on 96 slots of 24 fresh random scans per round: ideal-grid scans with
    dirt and truncations, sensor-faithful scans (firing order, quantised ranges, NaN / (0,0,0) / absent no-returns, rosette,
    tag bits, stray lines), with and without intra-sweep motion, lines made ragged by hand.  Everything mml_step leaves behind
    against the oracle pipeline of every slot."""
    from conftest import oracle_pipeline, perturbed, pose_to_x
    B, ND = 96, 24
    NVM = 1824 * 16
    t0 = time.time()
    # the map: sensor-faithful and ideal scans of the room, merged
    maps = [[], []]
    for k in range(8):
        mk = (synth.velo_scan_vlp16(k), synth.livox_scan_horizon(k)) if k % 2 else (synth.velo_scan(k), synth.livox_scan(k))
        o = oracle_pipeline(O, dict(velo=mk[0], livox=mk[1], dR=np.eye(3), dt=np.zeros(3)), None, None)
        T = synth.pose_matrix(k)
        maps[0].append(synth.transform(T, o["corner"].astype(np.float64)).astype(np.float32))
        maps[1].append(synth.transform(T, o["surf"].astype(np.float64)).astype(np.float32))
    cm, sm = O.voxel_downsample(np.concatenate(maps[0]), 0.4), O.voxel_downsample(np.concatenate(maps[1]), 0.2)
    tc, ts = O.KdTree(cm), O.KdTree(sm)
    c = M.Context(max_scans=B, max_velo_points=NVM, max_livox_points=24000)
    c.map_set_local(0, cm)
    c.map_set_local(1, sm)
    n_pts = n_fac = redo = 0
    worst_pose = 0.0
    flips = 0  # plane records (slots) whose omega is one float ulp off the oracle's (plane_records_close)
    for rnd in range(args.batch):
        cases = []
        for j in range(ND):
            k = int(rng.integers(0, 5000))
            motion = bool(rng.integers(0, 2))
            if rng.integers(0, 2):
                v = synth.velo_scan_vlp16(k, dropout=("skip", "nan", "zero")[int(rng.integers(0, 3))], motion=motion,
                                          drop_rate=float(rng.choice([0.0, 0.015, 0.1])), seed=int(rng.integers(0, 1 << 30)))
                l = synth.livox_scan_horizon(k, motion=motion, drop_rate=float(rng.choice([0.0, 0.04, 0.2])), seed=int(rng.integers(0, 1 << 30)))
            else:
                v, l = synth.velo_scan(k, motion=motion, noise=float(rng.choice([0.01, 0.003, 0.03]))).copy(), synth.livox_scan(k, motion=motion).copy()
            v, l = random_dirt(rng, v.copy(), l.copy())
            r = int(rng.integers(0, 8))
            if r == 0:
                v = v[:int(rng.integers(16, len(v)))]
            elif r == 1:
                l = l[:int(rng.integers(3, len(l)))]
            elif r == 2:                                    # ragged lines: one very long Livox line, the others short
                a = int(rng.integers(0, len(l) // 2))
                l["line"][a:a + int(rng.integers(2000, 12000))] = int(rng.integers(0, 6))
            elif r == 3:
                v, l = (None, l) if rng.integers(0, 2) else (v, None)
            dR, dt = synth.sweep_motion(k) if motion else (np.eye(3), np.zeros(3))
            T0 = perturbed(synth.pose_matrix(k), dt=rng.normal(0, 0.03, 3), rotvec=rng.normal(0, 0.004, 3))
            cases.append(dict(velo=v, livox=l, dR=dR, dt=dt, T0=T0, x0=pose_to_x(T0), k=k))
        ora = [oracle_pipeline(O, cs, tc, ts) for cs in cases]
        perm = rng.permutation(B) % ND                      # which scan a slot holds
        for s in range(B):
            c.scan_upload(s, cases[perm[s]]["velo"], cases[perm[s]]["livox"])
        dR = np.stack([cases[perm[s]]["dR"].reshape(9) for s in range(B)])
        dt = np.stack([cases[perm[s]]["dt"] for s in range(B)])
        x0 = np.stack([cases[perm[s]]["x0"] for s in range(B)])
        c.set_lanes((4, 1, 2)[rnd % 3])                     # mml_step on four, one, two (the default) stream lanes by turns
        c.extract(0, B)
        for s in range(B):
            d, o = c.scan_download(s), ora[perm[s]]
            i = d["info"]
            ok = i.n_points == len(o["xyzi"]) and all(np.array_equal(d[key], o[key]) for key in ("xyzi", "label", "ring", "reltime"))
            ok = ok and (i.velo_corner_num, i.velo_surf_num, i.livox_corner_num, i.livox_surf_num) == o["counts"]
            if not ok:
                print("BATCH EXTRACT MISMATCH seed %d round %d slot %d (scan %d of the round)" % (args.seed, rnd, s, perm[s]))
                np.save("gpurun_out/fuzz_batch_seed%d_round%d_v.npy" % (args.seed, rnd), cases[perm[s]]["velo"] if cases[perm[s]]["velo"] is not None else np.zeros(0))
                np.save("gpurun_out/fuzz_batch_seed%d_round%d_l.npy" % (args.seed, rnd), cases[perm[s]]["livox"] if cases[perm[s]]["livox"] is not None else np.zeros(0))
                return 1
            redo += c.extract_queue_counts(s)[0]
        x = c.step(0, B, dR, dt, np.eye(4), 25.0, 10, x0)
        for s in range(B):
            d, o = c.scan_download(s), ora[perm[s]]
            gl, glsrc = c.factors_download(s, 0)
            gp, gpsrc = c.factors_download(s, 1)
            ok = (np.array_equal(d["label"], o["label"]) and np.array_equal(d["xyzi"][:, :3], o["und"]) and np.all(d["reltime"] == 1.0)
                  and c.features_download(s, 0).tobytes() == o["corner"].tobytes() and c.features_download(s, 1).tobytes() == o["surf"].tobytes()
                  and np.array_equal(glsrc, o["lsrc"]) and np.array_equal(gpsrc, o["psrc"])
                  and np.allclose(gl, o["lf_arr"], rtol=0, atol=1e-9))
            pok, nflip = plane_records_close(gp, o["pf_arr"])
            ok = ok and pok
            flips += nflip
            dd = float(np.abs(x[s] - o["x"]).max())
            worst_pose = max(worst_pose, dd)
            if not ok or not dd < 1e-6:
                pieces = dict(label=np.array_equal(d["label"], o["label"]), und=np.array_equal(d["xyzi"][:, :3], o["und"]), reltime=bool(np.all(d["reltime"] == 1.0)),
                              corner=c.features_download(s, 0).tobytes() == o["corner"].tobytes(), surf=c.features_download(s, 1).tobytes() == o["surf"].tobytes(),
                              lsrc=np.array_equal(glsrc, o["lsrc"]), psrc=np.array_equal(gpsrc, o["psrc"]))
                if pieces["lsrc"]: pieces["lf |d|"] = float(np.abs(gl - o["lf_arr"]).max()) if len(gl) else 0.0
                else: pieces["lsrc diff"] = sorted(set(glsrc.tolist()) ^ set(o["lsrc"].tolist()))[:8]
                if pieces["psrc"]: pieces["pf |d|"] = float(np.abs(gp - o["pf_arr"]).max()) if len(gp) else 0.0
                else: pieces["psrc diff"] = sorted(set(gpsrc.tolist()) ^ set(o["psrc"].tolist()))[:8]
                print("BATCH STEP MISMATCH seed %d round %d slot %d (scan %d): records ok %s, pose diff %.3g; %s" % (args.seed, rnd, s, perm[s], ok, dd, pieces))
                return 1
            n_pts += len(o["xyzi"])
            n_fac += len(glsrc) + len(gpsrc)
    if flips > 16:
        print("BATCH: %d plane records with a float-ulp omega -- more than the cast's rounding explains" % flips)
        return 1
    print("batch: %d rounds x %d slots ok (%d points, %d factor records%s, redo-queue share %.3f %%, worst pose difference %.2e) %.0f s"
          % (args.batch, B, n_pts, n_fac, ", %d of them with omega one float ulp off" % flips if flips else "", 100.0 * redo / max(n_pts, 1), worst_pose,
             time.time() - t0), flush=True)
    c.close()
    return 0


def dense_batch_section(args, M, O, synth, rng):
    """Dense ring layouts in batches of 24 slots -- the dense layout's batch kernels: pass A on 1024-point blocks, k_assign_b,
    k_assign_c_staged (2048-point tiles dealt to the XCDs slot by slot), k_stencil<0> / batch k_select_part over 70 / 134 lines,
    the chunked label pass, the batch undistortion and both down-sampler forms -- on 6 fresh random scans per round (64 or 128
    rings, 512 .. 2048 azimuths, noise levels, NaN / (0,0,0) runs, near / far crops, truncations, with and without a Livox part),
    every slot against the oracle: extraction fields bit for bit, undistorted points within 1 ulp, both stacks byte for byte."""
    from scipy.spatial.transform import Rotation as Rsc
    B, ND = 24, 6
    t0 = time.time()
    npts = 0
    for rnd in range(args.dense_batch):
        n_rings = int(rng.choice([64, 128]))
        n_az = int(rng.choice([512, 1024, 1536, 2048] if n_rings == 64 else [512, 1024, 2048]))
        pitch0 = float(rng.choice([-15.5, -25.0, -16.0]))
        step = np.float32((abs(pitch0) * 2 + rng.uniform(-1, 3)) / (n_rings - 1))
        far = float(rng.choice([50.0, 1000.0]))
        kw = dict(n_rings=n_rings, pitch0=pitch0, pitch_step=step)
        cases = []
        for j in range(ND):
            v = synth.velo_scan(int(rng.integers(0, 3000)), n_rings=n_rings, n_az=n_az, pitch0=pitch0, pitch_step=step,
                                noise=float(rng.choice([0.0, 0.002, 0.01]))).copy()
            l = synth.livox_scan(int(rng.integers(0, 3000))).copy() if rng.integers(0, 2) else None
            if l is None:
                v, _ = random_dirt(rng, v, synth.livox_scan(0).copy())
            else:
                v, l = random_dirt(rng, v, l)
            r = int(rng.integers(0, 6))
            if r == 0:
                v = v[:int(rng.integers(n_rings, len(v)))]          # a sweep cut short (the last blocks / tiles are partial)
            elif r == 1:
                v = v[::int(rng.integers(2, 5))]                    # every 2nd .. 4th record: firings with missing rings
            elif r == 2:
                v[:, :3] *= float(rng.choice([0.3, 6.0]))           # mostly inside the near crop / beyond 50 m
            cases.append((v, l))
        cfgd = M.default_config(B, n_rings=n_rings, pitch0_deg=pitch0, pitch_step_deg=step, far_th=far, max_velo_points=n_rings * n_az,
                                max_livox_points=24000, max_features=1 << 18)
        cd = M.Context(cfgd)
        ora = []
        for v, l in cases:
            parts = [O.extract_velo(v, far=far, **kw)] + ([O.extract_livox(l, far=far)] if l is not None else [])
            ora.append({key: np.concatenate([e[key] for e in parts]) for key in ("xyzi", "label", "reltime", "ring")})
        perm = rng.permutation(B) % ND
        for s in range(B):
            cd.scan_upload(s, cases[perm[s]][0], cases[perm[s]][1])
        cd.extract(0, B)
        for s in range(B):
            d, o = cd.scan_download(s), ora[perm[s]]
            if not (d["info"].n_points == len(o["xyzi"]) and all(np.array_equal(d[key], o[key]) for key in o)):
                print("DENSE BATCH EXTRACT MISMATCH seed %d round %d slot %d (scan %d) rings %d az %d pitch0 %g step %g far %g"
                      % (args.seed, rnd, s, perm[s], n_rings, n_az, pitch0, step, far))
                np.save("gpurun_out/fuzz_dense_batch_seed%d_round%d_v.npy" % (args.seed, rnd), cases[perm[s]][0])
                if cases[perm[s]][1] is not None:
                    np.save("gpurun_out/fuzz_dense_batch_seed%d_round%d_l.npy" % (args.seed, rnd), cases[perm[s]][1])
                # which fields differ, where, and whether the other slots that hold the same scan agree with this one
                print("  n_points device %d oracle %d" % (d["info"].n_points, len(o["xyzi"])))
                for key in o:
                    if d[key].shape != o[key].shape:
                        print("  %s: shapes %s / %s" % (key, d[key].shape, o[key].shape))
                        continue
                    neq = d[key] != o[key]
                    bad = np.flatnonzero(neq.reshape(len(neq), -1).any(axis=1)) if neq.ndim > 1 else np.flatnonzero(neq)
                    if len(bad):
                        print("  %s: %d differing entries, first at %s: device %s oracle %s (ring %s)"
                              % (key, len(bad), bad[:8], d[key][bad[:3]].tolist(), o[key][bad[:3]].tolist(), o["ring"][bad[:8]].tolist()))
                same = [t for t in range(B) if perm[t] == perm[s]]
                agree = [t for t in same if all(np.array_equal(cd.scan_download(t)[key], d[key]) for key in o)]
                print("  slots holding the same scan: %s, of which equal to slot %d: %s" % (same, s, agree))
                cd.extract(0, B)
                d2 = cd.scan_download(s)
                print("  a second extract of the batch: slot equal to the oracle now: %s, equal to the first result: %s"
                      % (all(np.array_equal(d2[key], o[key]) for key in o), all(np.array_equal(d2[key], d[key]) for key in o)))
                return 1
        dR = [Rsc.from_rotvec(rng.normal(0, 0.02, 3)).as_matrix() for _ in range(ND)]
        dt = [rng.normal(0, 0.05, 3) for _ in range(ND)]
        cd.undistort(0, B, np.stack([dR[perm[s]].reshape(9) for s in range(B)]), np.stack([dt[perm[s]] for s in range(B)]))
        cd.downsample(0, B)
        exp = []
        for j in range(ND):
            ou = O.undistort(ora[j]["xyzi"][:, :3], ora[j]["reltime"], dR[j], dt[j])
            exp.append(ou)
        for s in range(B):
            j = perm[s]
            und = cd.scan_download(s)["xyzi"][:, :3]
            ulp = np.abs(und.view(np.int32).astype(np.int64) - exp[j].view(np.int32).astype(np.int64))
            ok = (ulp.max(initial=0) <= 1 and np.array_equal(cd.features_download(s, 0), O.voxel_downsample(und[ora[j]["label"] == 1], 0.4))
                  and np.array_equal(cd.features_download(s, 1), O.voxel_downsample(und[ora[j]["label"] == 2], 0.2)))
            if not ok:
                print("DENSE BATCH UNDISTORT / VOXEL MISMATCH seed %d round %d slot %d rings %d az %d far %g ulp %d"
                      % (args.seed, rnd, s, n_rings, n_az, far, ulp.max(initial=0)))
                return 1
            npts += len(und)
        cd.close()
    print("dense-batch: %d rounds x %d slots ok (%d points) %.0f s" % (args.dense_batch, B, npts, time.time() - t0), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--lines", type=int, default=500)
    ap.add_argument("--scans", type=int, default=20)
    ap.add_argument("--poses", type=int, default=12)
    ap.add_argument("--windows", type=int, default=10)
    ap.add_argument("--solves", type=int, default=10)
    ap.add_argument("--maps", type=int, default=2)
    ap.add_argument("--dense", type=int, default=2)
    ap.add_argument("--cubes", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="rounds of the large-batch section (96 slots each)")
    ap.add_argument("--only-batch", action="store_true", help="run the batch section(s) alone")
    ap.add_argument("--dense-batch", type=int, default=0, help="rounds of the dense-layout batch section (24 slots of 64 / 128-ring scans each)")
    args = ap.parse_args()
    M = importlib.import_module("multi-modal-loam_amd")
    synth = importlib.import_module("multi-modal-loam_amd.synth")
    import mml_oracle as O
    O.build()
    O.lib()
    from conftest import fuzz_line, perturbed

    rng = np.random.default_rng(args.seed)
    if args.batch > 0:
        rc = batch_section(args, M, O, synth, rng)
        if rc:
            return rc
    if args.dense_batch > 0:
        rc = dense_batch_section(args, M, O, synth, rng)
        if rc:
            return rc
    if args.only_batch:
        return 0
    ctx = M.Context(max_scans=4)
    t0 = time.time()

    # ---- lines ----------------------------------------------------------------------------------------------------------
    counts = dict(n150=0, n100=0, n2=0, n300=0)
    for trial in range(args.lines):
        pts = fuzz_line(rng)
        so, fo, flo = O.detect_feature_points(pts)
        s, f, fl = ctx.detect_line(pts)
        if not (np.array_equal(fl, flo) and np.array_equal(s, so) and np.array_equal(f, fo)):
            bad = np.flatnonzero(fl != flo)[:8]
            print("LINE MISMATCH seed %d trial %d n %d at %s device %s oracle %s" % (args.seed, trial, len(pts), bad, fl[bad], flo[bad]))
            np.save("gpurun_out/fuzz_line_seed%d_trial%d.npy" % (args.seed, trial), pts)
            return 1
        counts["n150"] += int((flo == 150).sum())
        counts["n100"] += int(((flo == 100) | (flo == 101)).sum())
        counts["n2"] += int((flo == 2).sum())
        counts["n300"] += int((flo == 300).sum())
    print("lines: %d ok (%s) %.0f s" % (args.lines, counts, time.time() - t0), flush=True)

    # ---- whole scans -----------------------------------------------------------------------------------------------------
    t0 = time.time()
    worst_ulp, n_pts = 0, 0
    for trial in range(args.scans):
        k = int(rng.integers(0, 5000))
        v = synth.velo_scan(k).copy()
        l = synth.livox_scan(k, motion=bool(rng.integers(0, 2))).copy()
        v, l = random_dirt(rng, v, l)
        if rng.integers(0, 4) == 0:
            v = v[:int(rng.integers(16, len(v)))]
        if rng.integers(0, 4) == 0:
            l = l[:int(rng.integers(3, len(l)))]
        ctx.scan_upload(0, v, l)
        ctx.extract(0, 1)
        d = ctx.scan_download(0)
        ev, el = O.extract_velo(v), O.extract_livox(l)
        o = {key: np.concatenate([ev[key], el[key]]) for key in ("xyzi", "label", "reltime", "ring")}
        ok = d["info"].n_points == len(o["xyzi"]) and all(np.array_equal(d[key], o[key]) for key in o)
        if not ok:
            print("SCAN MISMATCH seed %d trial %d k %d" % (args.seed, trial, k))
            for key in o:
                if len(d[key]) != len(o[key]):
                    print("  ", key, "length", len(d[key]), len(o[key]))
                elif not np.array_equal(d[key], o[key]):
                    bad = np.flatnonzero(np.any(np.atleast_2d(d[key].T != o[key].T), axis=0))[:8]
                    print("  ", key, "differs at", bad)
            np.save("gpurun_out/fuzz_scan_seed%d_trial%d_v.npy" % (args.seed, trial), v)
            np.save("gpurun_out/fuzz_scan_seed%d_trial%d_l.npy" % (args.seed, trial), l)
            return 1
        # undistortion with a random sweep motion, then the voxel filter of the labelled points
        dR = Rsc.from_rotvec(rng.normal(0, 0.02, 3) * (0 if rng.integers(0, 5) == 0 else 1)).as_matrix()
        dt = rng.normal(0, 0.05, 3)
        ctx.undistort(0, 1, dR[None], dt[None])
        ctx.downsample(0, 1)
        d2 = ctx.scan_download(0)
        ou = O.undistort(o["xyzi"][:, :3], o["reltime"], dR, dt)
        ulp = np.abs(d2["xyzi"][:, :3].view(np.int32).astype(np.int64) - ou.view(np.int32).astype(np.int64))
        worst_ulp = max(worst_ulp, int(ulp.max()) if len(ulp) else 0)
        n_pts += len(ulp)
        und = d2["xyzi"][:, :3]
        vox_ok = (np.array_equal(ctx.features_download(0, 0), O.voxel_downsample(und[o["label"] == 1], 0.4)) and
                  np.array_equal(ctx.features_download(0, 1), O.voxel_downsample(und[o["label"] == 2], 0.2)))
        if worst_ulp > 1 or not vox_ok or not np.all(d2["reltime"] == 1.0):
            print("UNDISTORT / VOXEL MISMATCH seed %d trial %d k %d ulp %d voxel_ok %s" % (args.seed, trial, k, worst_ulp, vox_ok))
            return 1
    print("scans: %d ok (%d points, worst undistort difference %d ulp) %.0f s" % (args.scans, n_pts, worst_ulp, time.time() - t0), flush=True)

    # ---- association + Estimate from random perturbations --------------------------------------------------------------------
    t0 = time.time()
    from conftest import build_scene
    scene = build_scene(O, synth)
    ctx.map_set_local(0, scene["corner_map"])
    ctx.map_set_local(1, scene["surf_map"])
    for k, fr in enumerate(scene["frames"]):
        ctx.scan_upload(k, fr["velo"], fr["livox"])
    ctx.extract(0, 4)
    ctx.undistort(0, 4, np.tile(np.eye(3).reshape(1, 9), (4, 1)), np.zeros((4, 3)))
    ctx.downsample(0, 4)
    worst = 0.0
    for trial in range(args.poses):
        T = np.stack([perturbed(fr["T_gt"], dt=rng.normal(0, 0.05, 3), rotvec=rng.normal(0, 0.01, 3)) for fr in scene["frames"]])
        P0 = T[:, :3, 3]
        Q0 = np.stack([Rsc.from_matrix(T[k][:3, :3]).as_quat() for k in range(4)])
        ex = np.eye(4)
        if trial % 2:
            ex[:3, :3] = Rsc.from_rotvec(rng.normal(0, 0.01, 3)).as_matrix()
            ex[:3, 3] = rng.normal(0, 0.03, 3)
        Pg, Qg, info = ctx.estimate(0, 4, ex, P0, Q0)
        for k, fr in enumerate(scene["frames"]):
            Po, Qo, it, deg, _ = O.estimate_single(fr["corner"], fr["surf"], scene["corner_map"], scene["surf_map"], ex, P0[k], Q0[k])
            dd = max(np.abs(Pg[k] - Po).max(), np.abs(Qg[k] - Qo).max())
            worst = max(worst, dd)
            if info[k].outer_iterations != it or info[k].is_degenerate != int(deg) or dd > 1e-7:
                print("ESTIMATE MISMATCH seed %d trial %d frame %d: outer %d vs %d, degenerate %d vs %d, pose diff %.3g"
                      % (args.seed, trial, k, info[k].outer_iterations, it, info[k].is_degenerate, int(deg), dd))
                return 1
    print("poses: %d x 4 ok (worst pose difference %.2e) %.0f s" % (args.poses, worst, time.time() - t0), flush=True)
    # ---- global cube store: random trajectories through MapIncrement / MapMove, then the two-level association ---------------
    t0 = time.time()
    states = 0
    for trial in range(args.cubes):
        c2 = M.Context(max_scans=2)
        cs = O.CubeStore()
        pos = np.array([rng.uniform(0, 60), rng.uniform(0, 60), rng.uniform(-2, 8)])
        yaw = 0.0
        nsteps = int(rng.choice([4, 8, 14]))
        for step in range(nsteps):
            jump = rng.integers(0, 6) == 0
            pos = pos + (rng.normal(0, 60, 3) * [1, 1, 0.1] if jump else rng.normal(0, 2.0, 3) * [1, 1, 0.1])
            yaw += rng.normal(0, 0.1)
            reps = int(rng.integers(0, 3))
            cw, sw = [np.zeros((0, 3), np.float32)], [np.zeros((0, 3), np.float32)]
            T = np.eye(4)
            T[:3, :3] = Rsc.from_rotvec([0, 0, yaw]).as_matrix()
            T[:3, 3] = pos
            for r in range(reps):
                fr = scene["frames"][int(rng.integers(0, 4))]
                Tr = T.copy()
                Tr[:3, 3] += rng.normal(0, 0.1, 3)
                thin = rng.uniform(0.1, 1.0)
                cf = fr["corner"][rng.random(len(fr["corner"])) < thin]
                sf = fr["surf"][rng.random(len(fr["surf"])) < thin]
                c2.features_upload(0, 0, cf)
                c2.features_upload(0, 1, sf)
                c2.map_global_append(0, Tr)
                for feats, acc in ((cf, cw), (sf, sw)):
                    x, y, z = (feats[:, i].astype(np.float64) for i in range(3))
                    acc.append(np.stack([(((Tr[q, 0] * x + Tr[q, 1] * y) + Tr[q, 2] * z) + Tr[q, 3]).astype(np.float32) for q in range(3)], 1))
                T = Tr
            nc, ns = c2.map_global_increment(T)
            cs.increment(np.concatenate(cw), np.concatenate(sw), T)
            for kind, n in ((0, nc), (1, ns)):
                gx, gc, gcen = c2.map_global_download(kind)
                ox, oc, ocen = cs.get(kind)
                order = np.argsort(gc, kind="stable")
                if not (n == len(ox) and np.array_equal(gcen, ocen) and np.array_equal(gc[order], oc) and gx[order].tobytes() == ox.tobytes()):
                    print("CUBE STORE MISMATCH seed %d trial %d step %d kind %d (%d vs %d points)" % (args.seed, trial, step, kind, n, len(ox)))
                    return 1
                states += 1
        c2.close()
    print("cubes: %d trajectories ok (%d store states equal) %.0f s" % (args.cubes, states, time.time() - t0), flush=True)

    # ---- other ring layouts, scans beyond 64 k points and labelled clouds beyond the 8192-key LDS sort (global-sort path) ------
    t0 = time.time()
    npts = 0
    for trial in range(args.dense):
        n_rings = int(rng.choice([32, 64, 128]))
        n_az = int(rng.choice([512, 900, 1800]))
        pitch0 = float(rng.choice([-15.5, -25.0, -16.0]))
        step = np.float32((abs(pitch0) * 2 + rng.uniform(-1, 3)) / (n_rings - 1))
        far = float(rng.choice([50.0, 1000.0]))
        scale = float(rng.choice([1.0, 1.0, 6.0]))       # x 6: most points beyond 50 m, all promoted to surf (:524)
        v = synth.velo_scan(int(rng.integers(0, 3000)), n_rings=n_rings, n_az=n_az, pitch0=pitch0, pitch_step=step,
                            noise=float(rng.choice([0.0, 0.002, 0.01]))).copy()
        v[:, :3] *= scale
        l = synth.livox_scan(int(rng.integers(0, 3000))) if rng.integers(0, 2) else None
        kw = dict(n_rings=n_rings, pitch0=pitch0, pitch_step=step)
        cfgd = M.default_config(1, n_rings=n_rings, pitch0_deg=pitch0, pitch_step_deg=step, far_th=far, max_velo_points=len(v),
                                max_livox_points=24000 if l is not None else 0, max_features=1 << 18)
        cd = M.Context(cfgd)
        cd.scan_upload(0, v, l)
        cd.extract(0, 1)
        d = cd.scan_download(0)
        ev = O.extract_velo(v, far=far, **kw)
        parts = [ev] + ([O.extract_livox(l, far=far)] if l is not None else [])
        o = {key: np.concatenate([e[key] for e in parts]) for key in ("xyzi", "label", "reltime", "ring")}
        if not (d["info"].n_points == len(o["xyzi"]) and all(np.array_equal(d[key], o[key]) for key in o)):
            print("DENSE SCAN MISMATCH seed %d trial %d rings %d az %d pitch0 %g step %g far %g scale %g" % (args.seed, trial, n_rings, n_az, pitch0, step, far, scale))
            return 1
        dR = Rsc.from_rotvec(rng.normal(0, 0.02, 3)).as_matrix()
        dt = rng.normal(0, 0.05, 3)
        cd.undistort(0, 1, dR[None], dt[None])
        cd.downsample(0, 1)
        d2 = cd.scan_download(0)
        ou = O.undistort(o["xyzi"][:, :3], o["reltime"], dR, dt)
        ulp = np.abs(d2["xyzi"][:, :3].view(np.int32).astype(np.int64) - ou.view(np.int32).astype(np.int64))
        und = d2["xyzi"][:, :3]
        ok = (ulp.max() <= 1 and np.array_equal(cd.features_download(0, 0), O.voxel_downsample(und[o["label"] == 1], 0.4))
              and np.array_equal(cd.features_download(0, 1), O.voxel_downsample(und[o["label"] == 2], 0.2)))
        if not ok:
            print("DENSE UNDISTORT / VOXEL MISMATCH seed %d trial %d rings %d az %d far %g scale %g ulp %d" % (args.seed, trial, n_rings, n_az, far, scale, ulp.max()))
            return 1
        npts += len(ulp)
        cd.close()
    print("dense: %d ok (%d points)  %.0f s" % (args.dense, npts, time.time() - t0), flush=True)

    # ---- association (factor records) + the lidar-only window solve, random thresholds / weights / window sizes -----------
    t0 = time.time()
    tc, ts = O.KdTree(scene["corner_map"]), O.KdTree(scene["surf_map"])
    for k, fr in enumerate(scene["frames"]):
        ctx.features_upload(k, 0, fr["corner"])
        ctx.features_upload(k, 1, fr["surf"])
    worst, nfac, terms = 0.0, 0, {}
    for trial in range(args.solves):
        big = rng.integers(0, 3) == 0
        T = np.stack([perturbed(fr["T_gt"], dt=rng.normal(0, 0.4 if big else 0.03, 3), rotvec=rng.normal(0, 0.06 if big else 0.004, 3))
                      for fr in scene["frames"]])
        thres = float(rng.choice([25.0, 10.0, 1.0]))
        st = ctx.associate(0, 4, T, thres)
        lfs, pfs = [], []
        for k, fr in enumerate(scene["frames"]):
            lf, lsrc = O.associate_lines(fr["corner"], tc, T[k], thres)
            pf, psrc = O.associate_planes(fr["surf"], ts, T[k], thres)
            gl, glsrc = ctx.factors_download(k, 0)
            gp, gpsrc = ctx.factors_download(k, 1)
            ol = np.concatenate([lf["point_ori"], lf["p1"], lf["p2"], lf["error"][:, None]], axis=1)
            op = np.concatenate([pf["point_ori"], pf["point_proj"], pf["omega"], pf["error"][:, None]], axis=1)
            if not (np.array_equal(glsrc, lsrc) and np.array_equal(gpsrc, psrc) and np.allclose(gl, ol, rtol=0, atol=1e-9)
                    and np.allclose(gp, op, rtol=0, atol=1e-9) and st[k].n_line == len(lf) and st[k].n_plane == len(pf)):
                print("ASSOCIATION MISMATCH seed %d trial %d frame %d thres %g" % (args.seed, trial, k, thres))
                return 1
            nfac += len(lf) + len(pf)
            lfs.append(lf)
            pfs.append(pf)
        T_bl = np.eye(4)
        if rng.integers(0, 2):
            T_bl[:3, :3] = Rsc.from_rotvec(rng.normal(0, 0.02, 3)).as_matrix()
            T_bl[:3, 3] = rng.normal(0, 0.03, 3)
        x0 = np.stack([np.concatenate([T[k][:3, 3], Rsc.from_matrix(T[k][:3, :3]).as_rotvec()]) for k in range(4)])
        W = int(rng.choice([1, 2, 4]))
        huber = float(rng.choice([0.0, 0.1 / 1.5e-3]))
        w_tan = float(rng.choice([0.0, 3e-4]))
        iters = int(rng.choice([1, 4, 10, 10, 30]))
        fixed = bool(rng.integers(0, 4) == 0)
        xg, sg, _ = ctx.solve(0, 4, x0, T_bl, window=W, max_iters=iters, fixed=fixed, huber=huber, w_tan=w_tan)
        for p in range(4 // W):
            sl = slice(p * W, (p + 1) * W)
            xo, so, _ = O.solve_window(lfs[sl], pfs[sl], x0[sl], T_bl, iters, fixed=fixed, huber=huber, w_tan=w_tan)
            dd = float(np.abs(xg[sl] - xo).max())
            same = (sg[p].iterations, sg[p].termination) == (so["iterations"], so["termination"])
            if not fixed:     # (iterations forced past convergence accept or reject steps of size 1e-12 on rounding noise)
                same = same and sg[p].successful == so["successful"]
            terms[so["termination"]] = terms.get(so["termination"], 0) + 1
            worst = max(worst, dd)
            if not same or dd > (1e-6 if fixed else 1e-8) or abs(sg[p].final_cost - so["final_cost"]) > 1e-9 * so["final_cost"] + 1e-18:
                print("SOLVE MISMATCH seed %d trial %d problem %d W %d iters %d fixed %s huber %g w_tan %g: oracle %s device (%d %d %d) |dx| %.3g"
                      % (args.seed, trial, p, W, iters, fixed, huber, w_tan, so, sg[p].iterations, sg[p].successful, sg[p].termination, dd))
                return 1
    print("solves: %d ok (%d factor records equal, worst |dx| %.2e, terminations %s) %.0f s" % (args.solves, nfac, worst, terms, time.time() - t0), flush=True)
    # ---- local map upkeep: random walks of key scans through MapIncrementLocal (ring of 50, VoxelGrid of the merged map) --------
    t0 = time.time()
    checked = 0
    for trial in range(args.maps):
        c2 = M.Context(max_scans=2)
        lm = O.LocalMap(window=50, leaf_corner=c2.cfg.leaf_corner, leaf_surf=c2.cfg.leaf_surf)
        pos, yaw = rng.normal(0, 1, 3), 0.0
        steps = int(rng.choice([8, 30, 70]))
        for step in range(steps):
            fr = scene["frames"][int(rng.integers(0, 4))]
            thin_c, thin_s = (1.0, 1.0) if step < 4 else (rng.uniform(0.05, 0.5), rng.uniform(0.03, 0.3))
            keep_c = fr["corner"][rng.random(len(fr["corner"])) < thin_c]
            keep_s = fr["surf"][rng.random(len(fr["surf"])) < thin_s]
            pos = pos + rng.normal(0, 0.3, 3) * [1, 1, 0.05]
            yaw += rng.normal(0, 0.05)
            T = perturbed(fr["T_gt"], dt=pos, rotvec=(rng.normal(0, 0.01), rng.normal(0, 0.01), yaw))
            c2.features_upload(1, 0, keep_c)
            c2.features_upload(1, 1, keep_s)
            nc, ns = c2.map_increment_local(1, T)
            lm.increment(keep_c, keep_s, T)
            if step in (0, steps // 2, steps - 1) or rng.integers(0, 10) == 0:
                for kind, n in ((0, nc), (1, ns)):
                    want = lm.get(kind)
                    got = c2.map_local_download(kind)
                    if n != len(want) or got.tobytes() != want.tobytes():
                        print("LOCAL MAP MISMATCH seed %d trial %d step %d kind %d: %d vs %d points" % (args.seed, trial, step, kind, n, len(want)))
                        return 1
                    checked += 1
        c2.close()
    print("maps: %d walks ok (%d map states equal byte for byte) %.0f s" % (args.maps, checked, time.time() - t0), flush=True)
    ctx.close()

    # ---- full-window problems (IMU factors, prior): trust-region loop on the device against the host loop -----------------
    t0 = time.time()
    odometry = importlib.import_module("multi-modal-loam_amd.odometry")
    G = synth.GRAVITY
    WMAX = 8
    c = M.Context(max_scans=WMAX)
    c.map_set_local(0, scene["corner_map"])
    c.map_set_local(1, scene["surf_map"])
    west = odometry.WindowEstimator(c, gravity=G, solver="host")
    soft, loose, worst, terms = 0, 0, 0.0, {}
    prior = None
    for trial in range(args.windows):
        k0 = int(rng.integers(5, 200))
        W = int(rng.integers(1, WMAX + 1))
        big = rng.integers(0, 4) == 0            # now and then a start far enough for rejected / invalid steps
        lost = rng.integers(0, 12) == 0          # ... or a window 500 m away from the map: no lidar factor at all
        x0, pres = [], [None]
        for f in range(W):
            k = k0 + f
            c.scan_upload(f, synth.velo_scan(k), synth.livox_scan(k))
            c.extract(f, 1)
            c.undistort(f, 1, np.eye(3).reshape(1, 9), np.zeros((1, 3)))
            c.downsample(f, 1)
            T = perturbed(synth.pose_matrix(k), dt=rng.normal(0, 0.3 if big else 0.02, 3) + (500.0 if lost else 0.0),
                          rotvec=rng.normal(0, 0.05 if big else 0.003, 3))
            x0.append(np.concatenate([T[:3, 3], Rsc.from_matrix(T[:3, :3]).as_rotvec(), synth.velocity_at(k) + rng.normal(0, 0.5 if big else 0.02, 3),
                                      rng.normal(0, 1e-3, 3), rng.normal(0, 1e-2, 3)]))
            if f > 0:
                pres.append(M.imu_preintegrate(synth.imu_samples(k - 1, k), np.zeros(3), np.zeros(3)))
            c.associate(f, 1, west._T_wl(x0[f])[None], 1.0)
        x0 = np.stack(x0)
        skip = set(int(v) for v in rng.integers(1, max(2, W), int(rng.integers(0, 3)))) if W > 2 else set()
        use_prior = prior is not None and rng.integers(0, 2) == 0
        if lost and rng.integers(0, 2) == 0:     # nothing constrains anything: gradient stop, or five invalid steps when fixed
            skip, use_prior = set(range(1, W)), False
        restart = (not lost) and rng.integers(0, 6) == 0   # a second solve from the converged state (parameter / function stop)
        max_iters = int(rng.choice([1, 3, 10, 10, 10, 25]))
        fixed = bool(rng.integers(0, 5) == 0)

        def make():
            fw = M.FullWindowSolver(W, max_iters=max_iters, fixed=fixed, huber=0.0, w_tan=3e-4)
            for f in range(1, W):
                if f not in skip:
                    fw.set_imu(f, pres[f], G)
            if use_prior:
                fw.set_prior(prior)
            return fw
        if restart:
            fw0 = make()
            xs = x0.copy()
            for _ in range(400):
                done, xs = fw0.step(c.linearize_window(0, W, xs, west.T_bl, 3e-4, 0.0), xs)
                if done:
                    break
            x0 = xs
        fh = make()
        xh = x0.copy()
        evals_h = 0
        for _ in range(400):
            done, xh = fh.step(c.linearize_window(0, W, xh, west.T_bl, 3e-4, 0.0), xh)
            evals_h += 1
            if done:
                break
        sh = fh.summary()
        fd = make()
        xd, sd, evals_d = fd.solve_device(c, 0, west.T_bl, x0)
        dd = float(np.abs(xd - xh).max())
        same = (sd.iterations, sd.successful, sd.termination, evals_d) == (sh.iterations, sh.successful, sh.termination, evals_h)
        terms[sh.termination] = terms.get(sh.termination, 0) + 1
        if not same:
            soft += 1
        worst = max(worst, dd)
        loose += dd > 1e-8
        # Both loops run the same functions in the same order of operations -- imu_math.h (its own sin / cos / atan included),
        # the band Cholesky against the dense one element by element, the substitutions in the device's order -- so iterates,
        # costs, counts and termination codes must be EQUAL, also for windows far from the map, restarts from a converged state
        # and problems no factor constrains, where the last bit of a sine used to flip a decision (round 2: 11 of 3000).
        if not same or dd != 0.0 or sd.final_cost != sh.final_cost:
            print("WINDOW MISMATCH seed %d trial %d W %d skip %s prior %s max_iters %d fixed %s: host (it %d ok %d term %d ev %d cost %.17g) "
                  "device (it %d ok %d term %d ev %d cost %.17g) max |dx| %.3g"
                  % (args.seed, trial, W, sorted(skip), use_prior, max_iters, fixed, sh.iterations, sh.successful, sh.termination, evals_h,
                     sh.final_cost, sd.iterations, sd.successful, sd.termination, evals_d, sd.final_cost, dd))
            return 1
        if W >= 2 and 1 not in skip and rng.integers(0, 3) == 0:
            rec0 = M.pack_record(*c.linearize(0, xh[0][:6], west.T_bl, 3e-4, 0.0))
            prior = fh.marginalize(rec0, xh)
    print("windows: %d ok, device loop EQUAL to the host loop (worst |dx| %.2e, %d above 1e-8, %d with a flipped decision, terminations %s) %.0f s"
          % (args.windows, worst, loose, soft, terms, time.time() - t0), flush=True)
    c.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
