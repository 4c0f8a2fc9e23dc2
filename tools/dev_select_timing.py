"""Dev aid: per-phase shader-clock stamps of one k_select workgroup.  Build the library with
`make -C multi-modal-loam_amd/csrc EXTRA=-DMML_SEL_TIMING=<line>` first (line 0..15 = ring, 16..21 = Livox line)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M = importlib.import_module("multi-modal-loam_amd")
synth = importlib.import_module("multi-modal-loam_amd.synth")
B = 64
ctx = M.Context(max_scans=B)
for s in range(B):
    ctx.scan_upload(s, synth.velo_scan(100 + s % 4, motion=True), synth.livox_scan(100 + s % 4, motion=True))
for _ in range(3):
    ctx.extract(0, B)
ctx.synchronize()
out = (C.c_ulonglong * 64)()
assert M.lib().mml_debug_sel_timing(out) == 0
t = np.array(out[:12], dtype=np.int64)
names = ["loads issued+init", "phase0 record", "phase1 masks", "dep rounds", "phase2", "refl rounds", "a2 ranks", "b bits",
         "p4 masks+vis", "p4 scan", "phase5"]
for nme, d in zip(names, np.diff(t)):
    print("%-20s %8d clk" % (nme, d))
print("total", t[11] - t[0], " dependency rounds (accumulated over launches):", out[20])
