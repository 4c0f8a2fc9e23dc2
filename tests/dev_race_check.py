"""Dev aid: the same scan in every slot, many extractions: every slot must produce the same labels every time (a data race
inside a workgroup would show up as a slot that differs)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = importlib.import_module("multi-modal-loam_amd")
synth = importlib.import_module("multi-modal-loam_amd.synth")
B = 256
ctx = M.Context(max_scans=B)
bad = 0
for scene in range(3):
    v, l = synth.velo_scan(200 + scene, motion=True), synth.livox_scan(200 + scene, motion=True)
    for s in range(B):
        ctx.scan_upload(s, v, l)
    for rep in range(4):
        ctx.extract(0, B)
        ref = ctx.scan_download(0)["label"]
        for s in range(1, B):
            if not np.array_equal(ctx.scan_download(s)["label"], ref):
                bad += 1
print("slots checked", 3 * 4 * (B - 1), "mismatching", bad)
