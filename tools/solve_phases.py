"""Cycles per phase of one k_solve workgroup over the default bench workload (library built with -DMML_SV_TIMING=<problem>):
python tools/solve_phases.py"""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--kernel-steps", "2"]
import bench  # noqa: E402
import torch  # noqa: E402

torch.cuda.init()  # (torch's HIP runtime first, as in bench.py: the library is loaded next to it)

M = importlib.import_module("multi-modal-loam_amd")
lib = M.lib()
out = (C.c_ulonglong * 16)()
lib.mml_debug_sv_timing(out, 1)
bench.main()
lib.mml_debug_sv_timing(out, 0)
names = ["set-up + first evaluation", "propose (wave 0) + barrier", "factor pass (thread 0)", "block reduction", "decide (wave 0) + barrier"]
n = max(int(out[7]), 1)
tot = sum(out[i] for i in range(5))
print("k_solve phases, cycles per launch of one workgroup (%d launches), total %d" % (n, tot // n))
for i, nme in enumerate(names):
    print("  %-28s %8d  %5.1f%%" % (nme, out[i] // n, 100.0 * out[i] / max(tot, 1)))
print("inside the proposal (first wavefront):")
for i, nme in zip(range(8, 13), ["diag / gradient / alpha", "Cholesky", "substitutions", "dogleg", "model change"]):
    print("  %-28s %8d  %5.1f%%" % (nme, out[i] // n, 100.0 * out[i] / max(tot, 1)))
