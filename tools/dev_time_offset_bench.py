"""Dev aid: wall time of mml_time_offset_search at the reference's size (8 Livox messages x 24 k points vs one 28.8 k-point
Velodyne scan, resolution 30, 12 000-point windows)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M = importlib.import_module("multi-modal-loam_amd")
synth = importlib.import_module("multi-modal-loam_amd.synth")
velo = synth.velo_scan(31)[:, :3]
parts = [synth.livox_scan(31 + k, motion=True) for k in range(8)]
livox = np.concatenate([np.stack([p["x"], p["y"], p["z"]], 1) for p in parts]).astype(np.float32)
c = M.Context(max_scans=1)
c.time_offset_search(velo, livox)
t0 = time.perf_counter()
for _ in range(5):
    g = c.time_offset_search(velo, livox)
t_gpu = (time.perf_counter() - t0) / 5
print("points %d vs %d, windows %d, best %d" % (len(livox), len(velo), len(g["window_error"]), g["best_window"]))
print("device (incl. upload, grid build, download): %.2f ms" % (1e3 * t_gpu))
