"""Dev aid: per-stage times of mml_extract / mml_undistort / mml_downsample alone (HIP events, one stream), for kernel experiments that may break the
downstream stages.  Usage: python tools/dev_stage_time.py [batch]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M = importlib.import_module("multi-modal-loam_amd")
synth = importlib.import_module("multi-modal-loam_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = M.Context(max_scans=B)
scans = [(synth.velo_scan(100 + k, motion=True), synth.livox_scan(100 + k, motion=True)) for k in range(8)]
for s in range(B):
    ctx.scan_upload(s, *scans[s % 8])
ctx.synchronize()
ctx.set_lanes(1)
for _ in range(2):
    ctx.extract(0, B)
    ctx.undistort(0, B, np.tile(np.eye(3).reshape(1, 9), (B, 1)), np.zeros((B, 3)))
    ctx.downsample(0, B)
ctx.profile_enable(True)
ctx.profile_reset()
for _ in range(4):
    ctx.extract(0, B)
    ctx.undistort(0, B, np.tile(np.eye(3).reshape(1, 9), (B, 1)), np.zeros((B, 3)))
    ctx.downsample(0, B)
prof = ctx.profile_get()
for k, v in prof.items():
    if v[1]:
        print("%-18s %8.3f ms" % (k, v[0] / v[1]))
