"""Developer probe: N contexts x B/N scans driven from N host threads on one GPU (stream-level overlap)."""
import importlib, os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = importlib.import_module("multi-modal-loam_amd")
synth = importlib.import_module("multi-modal-loam_amd.synth")
from scipy.spatial.transform import Rotation as Rsc

def setup(B, base=100):
    ctx = M.Context(max_scans=B, max_map_points=1 << 18)
    cm, sm = [], []
    for k in range(base - 8, base):
        ctx.scan_upload(0, synth.velo_scan(k), synth.livox_scan(k)); ctx.extract(0, 1)
        ctx.undistort(0, 1, np.eye(3).reshape(1, 9), np.zeros((1, 3))); ctx.downsample(0, 1)
        T = synth.pose_matrix(k)
        cm.append(synth.transform(T, ctx.features_download(0, 0).astype(np.float64)).astype(np.float32))
        sm.append(synth.transform(T, ctx.features_download(0, 1).astype(np.float64)).astype(np.float32))
    cm = synth.voxel_filter(np.concatenate(cm), 0.4); sm = synth.voxel_filter(np.concatenate(sm), 0.2)
    ctx.map_set_local(0, synth.grow_map(cm, 20000, seed=7)); ctx.map_set_local(1, synth.grow_map(sm, 180000, seed=8))
    nd = 16
    scans = [(synth.velo_scan(base + k, motion=True), synth.livox_scan(base + k, motion=True)) for k in range(nd)]
    dR = np.zeros((B, 9)); dt = np.zeros((B, 3)); x0 = np.zeros((B, 6))
    for s in range(B):
        k = s % nd
        ctx.scan_upload(s, scans[k][0], scans[k][1])
        mR, mt = synth.sweep_motion(base + k); dR[s], dt[s] = mR.reshape(9), mt
        Tp = synth.pose_matrix(base + k).copy(); Tp[:3, 3] += [0.03, -0.02, 0.01]
        x0[s] = np.concatenate([Tp[:3, 3], Rsc.from_matrix(Tp[:3, :3]).as_rotvec()])
    ctx.synchronize()
    return ctx, dR, dt, x0

def main():
    total = int(sys.argv[1]); nctx = int(sys.argv[2]); steps = 10
    B = total // nctx
    ctxs = [setup(B) for _ in range(nctx)]
    def run(c, n):
        ctx, dR, dt, x0 = c
        for _ in range(n):
            ctx.step(0, B, dR, dt, np.eye(4), 25.0, 10, x0)
    for c in ctxs: run(c, 2)
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(c, steps)) for c in ctxs]
    for t in th: t.start()
    for t in th: t.join()
    el = time.perf_counter() - t0
    print("total %d nctx %d: %.0f scans/s (%.2f ms per %d scans)" % (total, nctx, total * steps / el, el / steps * 1e3, total))

if __name__ == "__main__":
    main()
