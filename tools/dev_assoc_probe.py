"""Developer probe: where does k_associate spend its time? (GPU box)"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = importlib.import_module("multi-modal-loam_amd")
synth = importlib.import_module("multi-modal-loam_amd.synth")
from scipy.spatial.transform import Rotation as Rsc

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    ctx = M.Context(max_scans=B, max_map_points=1 << 18, max_features=int(os.environ.get("MF", "8192")))
    base = 100
    cm, sm = [], []
    for k in range(base - 8, base):
        ctx.scan_upload(0, synth.velo_scan(k), synth.livox_scan(k)); ctx.extract(0, 1)
        ctx.undistort(0, 1, np.eye(3).reshape(1, 9), np.zeros((1, 3))); ctx.downsample(0, 1)
        T = synth.pose_matrix(k)
        cm.append(synth.transform(T, ctx.features_download(0, 0).astype(np.float64)).astype(np.float32))
        sm.append(synth.transform(T, ctx.features_download(0, 1).astype(np.float64)).astype(np.float32))
    cm = synth.voxel_filter(np.concatenate(cm), 0.4); sm = synth.voxel_filter(np.concatenate(sm), 0.2)
    cmap = synth.grow_map(cm, 20000, seed=7); smap = synth.grow_map(sm, 180000, seed=8)
    ctx.map_set_local(0, cmap); ctx.map_set_local(1, smap)
    nd = 16
    feats = []
    for k in range(nd):
        ctx.scan_upload(0, synth.velo_scan(base + k, motion=True), synth.livox_scan(base + k, motion=True)); ctx.extract(0, 1)
        dR, dt = synth.sweep_motion(base + k)
        ctx.undistort(0, 1, dR.reshape(1, 9), dt.reshape(1, 3)); ctx.downsample(0, 1)
        feats.append((ctx.features_download(0, 0).copy(), ctx.features_download(0, 1).copy()))
    T = np.zeros((B, 16))
    for s in range(B):
        Tp = synth.pose_matrix(base + s % nd).copy(); Tp[:3, 3] += [0.03, -0.02, 0.01]
        Tp[:3, :3] = Tp[:3, :3] @ Rsc.from_rotvec([0.002, -0.001, 0.004]).as_matrix()
        T[s] = Tp.reshape(16)
    empty = np.zeros((0, 3), np.float32)
    ctx.profile_enable(True)
    for name, uc, us in (("both", True, True), ("corner-only", True, False), ("surf-only", False, True)):
        for s in range(B):
            ctx.features_upload(s, 0, feats[s % nd][0] if uc else empty)
            ctx.features_upload(s, 1, feats[s % nd][1] if us else empty)
        for thres in (25.0, 1.0):
            ctx.associate(0, B, T, thres); ctx.profile_reset()
            for _ in range(5):
                ctx.associate(0, B, T, thres)
            p = ctx.profile_get()
            print("%-12s thres %5.1f associate %.3f ms" % (name, thres, p["associate"][0] / p["associate"][1]))
    # pure kNN on world-frame surf queries
    q = np.concatenate([synth.transform(T[s].reshape(4, 4), feats[s % nd][1].astype(np.float64)).astype(np.float32) for s in range(B)])
    for md in (25.0, 1.0):
        ctx.knn5(1, q, max_d2=md); ctx.profile_reset()
        for _ in range(3):
            gi, gd = ctx.knn5(1, q, max_d2=md)
        p = ctx.profile_get()
        print("pure knn surf nq=%d max_d2 %.0f: %.3f ms; d5 quantiles" % (len(q), md, p["knn5"][0] / p["knn5"][1]),
              np.quantile(np.sqrt(gd[:, 4][np.isfinite(gd[:, 4])]), [0.5, 0.9, 0.99, 1.0]))

if __name__ == "__main__":
    main()
