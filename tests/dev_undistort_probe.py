import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mml_oracle as O
M = importlib.import_module("multi-modal-loam_amd")
synth = importlib.import_module("multi-modal-loam_amd.synth")
ctx = M.Context(max_scans=4)
tot = 0; bad = 0
for k in range(100, 108):
    v, l = synth.velo_scan(k, motion=True), synth.livox_scan(k, motion=True)
    ctx.scan_upload(0, v, l); ctx.extract(0, 1)
    d0 = ctx.scan_download(0)
    dR, dt = synth.sweep_motion(k)
    ctx.undistort(0, 1, dR.reshape(1, 9), dt.reshape(1, 3))
    d1 = ctx.scan_download(0)
    o = O.undistort(d0["xyzi"][:, :3], d0["reltime"], dR, dt)
    neq = (d1["xyzi"][:, :3] != o)
    tot += o.size; bad += int(neq.sum())
    if neq.any():
        idx = np.argwhere(neq)[:3]
        for i, c in idx:
            print("scan", k, "pt", i, c, d1["xyzi"][i, c], o[i, c], "in", d0["xyzi"][i, :3], d0["reltime"][i])
print("mismatching coordinates:", bad, "of", tot)
# larger sweep rotations: theta = angle / 2 on both sides of the 0.5 switch between the polynomial and the libm form
from scipy.spatial.transform import Rotation as Rsc
tot = bad = 0
for ang in (0.05, 0.3, 0.9, 0.99, 1.01, 1.5, 3.0):
    v, l = synth.velo_scan(100, motion=True), synth.livox_scan(100, motion=True)
    ctx.scan_upload(0, v, l); ctx.extract(0, 1)
    d0 = ctx.scan_download(0)
    dR = Rsc.from_rotvec(np.array([0.3, -0.5, 0.81]) / np.linalg.norm([0.3, -0.5, 0.81]) * ang).as_matrix()
    dt = np.array([0.4, -0.2, 0.05])
    ctx.undistort(0, 1, dR.reshape(1, 9), dt.reshape(1, 3))
    d1 = ctx.scan_download(0)
    o = O.undistort(d0["xyzi"][:, :3], d0["reltime"], dR, dt)
    neq = (d1["xyzi"][:, :3] != o)
    tot += o.size; bad += int(neq.sum())
    print("angle", ang, "mismatches", int(neq.sum()))
print("large rotations: mismatching coordinates:", bad, "of", tot)
