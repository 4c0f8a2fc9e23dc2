#!/usr/bin/env python3
"""bench.py -- scans/s of the mm-loam scan-registration hot path on MI355X.

Workload (BASELINE.json configs[1]): fused VLP-16 (16 x 1800) + Livox Horizon (24 000) scans = 52 800 points,
local map of 200 000 points, one step = feature extraction + undistortion + down-sampling + one 5-NN association
pass + 10 trust-region (GN/dogleg) iterations for a batch of B scans whose raw points are already resident in
HBM.  value = whole-job scans/s.  One process per GPU (torch.distributed / RCCL only for the barrier and the
max-over-ranks clock; the path shards by scan, no data-path collective: "scaling": "weak").

Adds to the JSON line:
  roofline     -- dominant kernel, algorithmic bytes per launch / its mean launch time (HIP events on the library's
                  own stream, recorded around every launch inside the timed region) against 8 TB/s HBM
  cpu_baseline -- the CPU oracle (oracle/, a line-by-line port of the reference arithmetic; the reference binary
                  cannot be built here) timed on this box's host cores over a bounded sample of the same scans
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s achievable

# Algorithmic bytes per unit for every stage (DESIGN.md section "Kernels"): N fused points, F features per scan.
# The SURVEY 8(d) figures: extraction 20 B/pt in total, undistort 28 B/pt, association 112 B/feature,
# linearisation 72 B/factor/iteration.  The extraction chain is split over its kernels by what each must move.
STAGE_BYTES = {
    "assign_count":     lambda n_v, n_l, nf, it: 16 * n_v + 20 * n_l, # k_assign_a: read the raw records, ring id / crop test
    "assign_scan":      lambda n_v, n_l, nf, it: 0,                   # k_assign_b: block records only
    "assign_scatter":   lambda n_v, n_l, nf, it: 16 * n_v + 20 * n_l + 20 * (n_v + n_l),  # k_assign_c: read again, write xyzi + label slot
    "stencil":          lambda n_v, n_l, nf, it: 16 * (n_v + n_l),    # read xyzi of every bucketed point
    "select":           lambda n_v, n_l, nf, it: 11 * (n_v + n_l),    # attr 2 B + 2 order keys 8 B + label 1 B
    "crop_compact":     lambda n_v, n_l, nf, it: 4 * (n_v + n_l),     # write the 4 B label/line/time record
    "undistort":        lambda n_v, n_l, nf, it: 28 * (n_v + n_l),
    "voxel_downsample": lambda n_v, n_l, nf, it: 17 * (n_v + n_l),    # label scan + xyz of labelled points
    "associate":        lambda n_v, n_l, nf, it: 112 * nf,
    "associate_far":    lambda n_v, n_l, nf, it: 0,                   # queue of the few far queries (bytes counted in associate)
    "associate_fit":    lambda n_v, n_l, nf, it: 0,                   # model fit of the searched features (bytes counted in associate)
    "assoc_stats":      lambda n_v, n_l, nf, it: 0,
    "solve":            lambda n_v, n_l, nf, it: 72 * nf * (it + 1),  # it iterations + the initial linearisation
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2048, help="scans per step per GPU")
    ap.add_argument("--map-points", type=int, default=200000)
    ap.add_argument("--gn-iters", type=int, default=10)
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even with one rank (exercises the N > 1 code path)")
    ap.add_argument("--window-demo", action="store_true", help="run the joint window solve section on one GPU as well")
    ap.add_argument("--kernel-steps", type=int, default=4, help="single-stream steps after the timed region (per-kernel timing)")
    ap.add_argument("--cpu-scans", type=int, default=-1, help="CPU baseline sample size (-1: auto, 0: skip)")
    ap.add_argument("--cpu-threads", type=int, default=-1, help="threads of the scan-parallel CPU leg (-1: all cores up to 64, 0: skip)")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic scans cycled through the batch")
    ap.add_argument("--cell-corner", type=float, default=0.0, help="kNN grid cell edge for the corner map (0: library default)")
    ap.add_argument("--cell-surf", type=float, default=0.0, help="kNN grid cell edge for the surf map (0: library default)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    M = importlib.import_module("multi-modal-loam_amd")
    synth = importlib.import_module("multi-modal-loam_amd.synth")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    B = args.batch
    ctx = M.Context(max_scans=B, device=local_rank, max_map_points=max(args.map_points, 1 << 16),
                    cell_corner=args.cell_corner, cell_surf=args.cell_surf)
    dev_name, cus, hbm = ctx.device_info()

    # ---- synthetic inputs (same on every rank except the seed offset) ---------------------------------------
    base = 100 + 1000 * rank
    nd = max(1, min(args.distinct, B))
    scans = [(synth.velo_scan(base + k, motion=True), synth.livox_scan(base + k, motion=True)) for k in range(nd)]
    # map: features of the 8 scans preceding the batch (the role of the 50-keyframe local map, Estimator.cpp:1585-1643),
    # extracted with the product path itself, moved to the world frame with the generating poses, then grown to
    # --map-points by tiled replication (synth.grow_map)
    cm, sm = [], []
    for k in range(base - 8, base):
        ctx.scan_upload(0, synth.velo_scan(k), synth.livox_scan(k))
        ctx.extract(0, 1)
        ctx.undistort(0, 1, np.eye(3).reshape(1, 9), np.zeros((1, 3)))
        ctx.downsample(0, 1)
        T = synth.pose_matrix(k)
        cm.append(synth.transform(T, ctx.features_download(0, 0).astype(np.float64)).astype(np.float32))
        sm.append(synth.transform(T, ctx.features_download(0, 1).astype(np.float64)).astype(np.float32))
    # the reference voxel-filters the merged local map on every update (Estimator.cpp:1630-1637)
    cm = synth.voxel_filter(np.concatenate(cm), ctx.cfg.leaf_corner)
    sm = synth.voxel_filter(np.concatenate(sm), ctx.cfg.leaf_surf)
    n_corner_map = max(64, args.map_points // 10)
    corner_map = synth.grow_map(cm, n_corner_map, seed=7)
    surf_map = synth.grow_map(sm, args.map_points - n_corner_map, seed=8)
    ctx.map_set_local(0, corner_map)
    ctx.map_set_local(1, surf_map)

    from scipy.spatial.transform import Rotation as Rsc
    dR = np.zeros((B, 9))
    dt = np.zeros((B, 3))
    x0 = np.zeros((B, 6))
    for s in range(B):
        k = s % nd
        ctx.scan_upload(s, scans[k][0], scans[k][1])
        # true motion over the sweep (the scans are simulated with it) and a perturbed initial pose
        mR, mt = synth.sweep_motion(base + k)
        dR[s], dt[s] = mR.reshape(9), mt
        Tp = synth.pose_matrix(base + k).copy()
        Tp[:3, 3] += [0.03, -0.02, 0.01]
        Tp[:3, :3] = Tp[:3, :3] @ Rsc.from_rotvec([0.002, -0.001, 0.004]).as_matrix()
        x0[s] = np.concatenate([Tp[:3, 3], Rsc.from_matrix(Tp[:3, :3]).as_rotvec()])
    ctx.synchronize()
    exTlb = np.eye(4)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.step(0, B, dR, dt, exTlb, 25.0, args.gn_iters, x0)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = ctx.step(0, B, dR, dt, exTlb, 25.0, args.gn_iters, x0)
    barrier()
    elapsed = time.perf_counter() - t0
    # Per-kernel durations for the roofline object: the timed region overlaps sub-batches on 4 streams, so kernel
    # time there is shared between concurrent kernels.  The same step is therefore repeated on ONE stream (every
    # kernel covers the whole batch and owns the device) with HIP events around each stage on that stream; these
    # are the launches of grid size `KB scans` in the rocprofv3 summary under profiles/.
    # The kernel pass runs over KB = min(B, 1024) scans: launches of that size are distinct, in the rocprofv3 summary,
    # from the timed region's per-lane launches of B / 4 scans.
    KB = min(B, 1024)
    ctx.set_lanes(1)
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(args.kernel_steps):
        ctx.step(0, KB, dR[:KB], dt[:KB], exTlb, 25.0, args.gn_iters, x0[:KB])
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # pose sanity: the step must actually have registered the scans
    gt_err = max(np.abs(x[s][:3] - synth.pose_matrix(base + (s % nd))[:3, 3]).max() for s in range(B))

    # ---- roofline for the dominant kernel ---------------------------------------------------------------------
    info = [ctx.scan_info(s) for s in range(min(B, nd))]
    # (per kernel-pass launch of KB scans)
    n_v = float(np.mean([i.n_velo for i in info])) * KB
    n_l = float(np.mean([i.n_points - i.n_velo for i in info])) * KB
    nf = float(np.mean([len(ctx.features_download(s, 0)) + len(ctx.features_download(s, 1)) for s in range(min(B, nd))])) * KB
    stage_ms = {k: v[0] / max(v[1], 1) for k, v in prof.items() if v[1] > 0}
    dom = max(stage_ms, key=stage_ms.get)
    alg_bytes = STAGE_BYTES.get(dom, lambda *a: 0)(n_v, n_l, nf, args.gn_iters)
    achieved = alg_bytes / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
    traffic = None
    tr_file = os.path.join(ROOT, "profiles", "traffic_r01.json")
    if os.path.exists(tr_file):
        try:
            tr = json.load(open(tr_file))
            traffic = tr.get(dom)
            if traffic is not None:
                traffic = traffic * KB / float(tr.get("scans_per_launch", 256))
        except Exception:
            traffic = None
    bytes_per_scan = 48 * (n_v + n_l) / KB + 112 * nf / KB + 72 * nf / KB * args.gn_iters
    total_scans = world * B * args.steps
    value = total_scans / elapsed

    # ---- joint window solve across ranks: one frame per GPU, RCCL all-gather of the 32-double
    # normal-equation record per iteration, every rank advancing the same host-side dogleg state machine.  Outside the
    # timed region; reported next to the sharded throughput because the live path never exchanges data.
    window = None
    if dist is not None or args.window_demo:
        try:
            if dist is None:
                import torch.distributed as dist_w
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29577")
                dist_w.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
            else:
                dist_w = dist
            W = dist_w.get_world_size()
            dev = torch.device("cuda", local_rank)
            xw = torch.from_numpy(x0[0].copy()).to(dev)
            xs_all = torch.zeros(W * 6, dtype=torch.float64, device=dev)
            dist_w.all_gather_into_tensor(xs_all, xw)
            x_eval = xs_all.cpu().numpy().reshape(W, 6)
            ws = M.WindowSolver(W, max_iters=args.gn_iters, fixed=False, huber=0.1 / 1.5e-3, w_tan=0.0)
            rec = torch.zeros(32, dtype=torch.float64, device=dev)
            recs = torch.zeros(W * 32, dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            its = 0
            while True:
                ctx.linearize_record(0, x_eval[rank], np.eye(4), rec.data_ptr())   # slot 0 keeps the factors of the last step
                ctx.synchronize()
                dist_w.all_gather_into_tensor(recs, rec)
                done, x_eval = ws.step(recs.cpu().numpy().reshape(W, 32), x_eval)
                its += 1
                if done or its > 4 * args.gn_iters:
                    break
            torch.cuda.synchronize()
            t_win = time.perf_counter() - t1
            chk = torch.from_numpy(x_eval.reshape(-1).copy()).to(dev)
            ref = chk.clone()
            dist_w.broadcast(ref, src=0)
            agree = torch.tensor([1.0 if torch.equal(ref, chk) else 0.0], device=dev)
            dist_w.all_reduce(agree, op=dist_w.ReduceOp.MIN)
            sm = ws.summary()
            window = {"frames": W, "evaluations": its, "iterations": sm.iterations, "termination": sm.termination,
                      "ms_per_evaluation": t_win / its * 1e3, "ranks_agree_bitwise": bool(agree.item() == 1.0),
                      "own_frame_err_vs_gt_m": float(np.abs(x_eval[rank][:3] - synth.pose_matrix(base)[:3, 3]).max())}
            if dist is None:
                dist_w.destroy_process_group()
        except Exception as e:  # never lose the bench line to the demo
            window = {"error": repr(e)[:300]}

    # ---- CPU baseline: the oracle on this box's host cores (rank 0, N = 1 only) ----------------------------------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_scans != 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import mml_oracle as O
        O.build()
        tc, ts = O.KdTree(corner_map), O.KdTree(surf_map)  # map build excluded on both sides
        T_bl = np.eye(4)

        def cpu_scan(k):
            v, l = scans[k % nd]
            ev, el = O.extract_velo(v), O.extract_livox(l)
            xyz = np.concatenate([ev["xyzi"][:, :3], el["xyzi"][:, :3]])
            rel = np.concatenate([ev["reltime"], el["reltime"]])
            lab = np.concatenate([ev["label"], el["label"]])
            und = O.undistort(xyz, rel, dR[k % B].reshape(3, 3), dt[k % B])
            cf = O.voxel_downsample(und[lab == 1], 0.4)
            sf = O.voxel_downsample(und[lab == 2], 0.2)
            xx = x0[k % B]
            Tw = np.eye(4)
            Tw[:3, :3] = Rsc.from_rotvec(xx[3:]).as_matrix()
            Tw[:3, 3] = xx[:3]
            lf, _ = O.associate_lines(cf, tc, Tw, 25.0)
            pf, _ = O.associate_planes(sf, ts, Tw, 25.0)
            xs, _, _ = O.solve_window([lf], [pf], xx[None], T_bl, args.gn_iters, fixed=True)
            return xs

        t1 = time.perf_counter()
        xs = cpu_scan(0)
        one = time.perf_counter() - t1
        n_cpu = args.cpu_scans if args.cpu_scans > 0 else int(max(8, min(2048, 15.0 / max(one, 1e-3))))
        t1 = time.perf_counter()
        for k in range(n_cpu):
            xs = cpu_scan(k)
        cpu_t = time.perf_counter() - t1
        cpu = {"value": n_cpu / cpu_t, "unit": "scans/s", "cores": 1, "kind": "port",
               "sample": "%d fused 52.8k-pt scans (cycled over %d distinct), same map / poses / 10 fixed iterations, "
                         "single thread of %d host cores, kd-tree build excluded" % (n_cpu, nd, os.cpu_count()),
               "pose_diff_vs_gpu": float(np.abs(xs[0] - x[(n_cpu - 1) % B]).max())}
        # the same port with one scan per host thread (the C++ calls release the GIL): an upper bound for what the
        # reference's 6-thread layout (unionFeatureExtract.cpp:1008-1015, Estimator.cpp:1271-1297,1430) could reach here
        if args.cpu_threads != 0:
            from concurrent.futures import ThreadPoolExecutor
            nthr = args.cpu_threads if args.cpu_threads > 0 else min(os.cpu_count() or 1, 64)
            n_mt = int(max(2 * nthr, min(4096, 10.0 * nthr / max(one, 1e-3))))
            t1 = time.perf_counter()
            with ThreadPoolExecutor(nthr) as ex:
                list(ex.map(cpu_scan, range(n_mt)))
            mt_t = time.perf_counter() - t1
            cpu["all_threads"] = {"value": n_mt / mt_t, "unit": "scans/s", "cores": nthr, "sample": "%d scans, one scan per thread" % n_mt}

    if rank == 0:
        out = {
            "metric": "scans/s (feature-extract+kNN+10 GN iters) on 16-ring x1800 + Livox 24k fused cloud",
            "value": value, "unit": "scans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32/f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: fused VLP-16 16x1800 + Livox Horizon 24000 scan (52800 pts), "
                                   "local map %d pts, 1 association pass (thres_dist 25), %d GN iterations, W=1"
                                   % (args.map_points, args.gn_iters),
                       "scans_per_step_per_gpu": B, "distinct_scans": nd, "parallelism": "scan-sharded x%d" % world,
                       "device": dev_name, "cus": cus, "features_per_scan": nf / KB,
                       "algorithmic_bytes_per_scan": bytes_per_scan, "max_pose_err_vs_gt_m": float(gt_err)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "avg_launch_ms": stage_ms[dom], "algorithmic_bytes_per_launch": alg_bytes,
                         "scans_per_launch": KB, "timing": "HIP events, %d single-stream steps after the timed region" % args.kernel_steps,
                         "whole_path_frac": bytes_per_scan * value / world / 1e9 / HBM_PEAK_GBPS,
                         "stage_ms_per_launch": stage_ms},
            "cpu_baseline": cpu,
            "window_solve": window,
        }
        line = json.dumps(out)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(line, flush=True)  # the one JSON line, after anything the collectives library may have printed


if __name__ == "__main__":
    main()
