"""Stage times (HIP events, one stream) of a 1024-scan configs[1] step without bench.py's checks: for A/B builds whose results need
not be right ($MML_LIB_PATH selects the library).  python tools/stage_probe.py [slots] [reps] [extract only: 0 | 1] [config: 1 | 3 | 4]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main(B=1024, reps=4, extract_only=0, config=1):
    M = importlib.import_module("multi-modal-loam_amd")
    synth = importlib.import_module("multi-modal-loam_amd.synth")
    cfg = dict(bench.CONFIGS[config])
    args = type("A", (), dict(cell_corner=0.0, cell_surf=0.0))()
    ctx = bench.make_context(M, cfg, B, 0, args, cfg["map_points"])
    scans = [bench.make_scan(synth, cfg, 100 + k) for k in range(16)]
    if extract_only:   # (builds whose label lists are wrong on purpose: the extraction stages only)
        for s in range(B):
            ctx.scan_upload(s, scans[s % 16][0], scans[s % 16][1])
        ctx.synchronize()
        ctx.set_lanes(1)
        for _ in range(2):
            ctx.extract(0, B)
        ctx.profile_enable(True)
        ctx.profile_reset()
        for _ in range(reps):
            ctx.extract(0, B)
        prof = ctx.profile_get()
        st = {k: v[0] / reps for k, v in prof.items() if v[1] > 0}
        print(os.environ.get("MML_LIB_PATH", "default"), "extract sum %.3f" % sum(st.values()), " ".join("%s %.3f" % (k, v) for k, v in st.items()))
        return
    cm, sm, tiles_n = bench.build_maps(ctx, synth, cfg, 100, cfg["map_points"])
    ctx.map_set_local(0, cm)
    ctx.map_set_local(1, sm)
    dR, dt, x0 = np.zeros((B, 9)), np.zeros((B, 3)), np.zeros((B, 6))
    from scipy.spatial.transform import Rotation as Rsc
    for s in range(B):
        k = s % 16
        ctx.scan_upload(s, scans[k][0], scans[k][1])
        mR, mt = synth.sweep_motion(100 + k)
        dR[s], dt[s] = mR.reshape(9), mt
        Tp = synth.pose_matrix(100 + k).copy()
        Tp[:3, 3] += [0.03, -0.02, 0.01]
        x0[s] = np.concatenate([Tp[:3, 3], Rsc.from_matrix(Tp[:3, :3]).as_rotvec()])
    ctx.synchronize()
    ctx.set_lanes(1)
    for _ in range(2):
        ctx.step(0, B, dR, dt, np.eye(4), 25.0, cfg.get("gn_iters", 10), x0)
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(reps):
        ctx.step(0, B, dR, dt, np.eye(4), 25.0, cfg.get("gn_iters", 10), x0)
    prof = ctx.profile_get()
    st = {k: v[0] / reps for k, v in prof.items() if v[1] > 0}
    print(os.environ.get("MML_LIB_PATH", "default"), "sum %.3f" % sum(st.values()), " ".join("%s %.3f" % (k, v) for k, v in st.items()))


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:5]])
