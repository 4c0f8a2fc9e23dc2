"""Phase clocks of the live path's solve (one slot per mml_step call; library built with -DMML_SV_TIMING=0):
MML_SOLVE_WIDE=1 -> k_solve_wide's factor pass (rows / barrier wait / sums / tree), MML_SOLVE_WIDE=0 -> k_solve<true>'s phases.
  MML_LIB_PATH=multi-modal-loam_amd/libmmloam_hip_svt.so python tools/solve_wide_phases.py [reps]"""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import stage_probe  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
M = importlib.import_module("multi-modal-loam_amd")
lib = M.lib()
a = (C.c_ulonglong * 16)()
w = (C.c_ulonglong * 8)()
lib.mml_debug_sv_timing(a, 1)
lib.mml_debug_svw_timing(w, 1)
stage_probe.main(1, reps)
lib.mml_debug_sv_timing(a, 0)
lib.mml_debug_svw_timing(w, 0)
if w[6]:
    n, p = int(w[6]), max(int(w[4]), 1)
    print("k_solve_wide: %d launches, %d cycles per launch, %d passes per launch" % (n, w[5] // n, p // n))
    for i, nme in enumerate(["rows (thread 0)", "wait at the barrier", "sums", "tree + final barriers"]):
        print("  %-24s %7d cycles per pass" % (nme, w[i] // p))
    print("  per launch: proposal + barrier %d, accept / reject + barrier %d, everything else %d" % (a[1] // n, a[4] // n, a[0] // n))
if a[7]:
    n = int(a[7])
    tot = sum(a[i] for i in range(5))
    print("k_solve<true>: %d launches, %d cycles per launch" % (n, tot // n))
    for i, nme in enumerate(["set-up + first evaluation", "propose + barrier", "factor pass (thread 0)", "block reduction", "decide + barrier"]):
        print("  %-28s %8d per launch" % (nme, a[i] // n))
