"""Cycles per phase of one k_assign_onepass_s<0> workgroup (library built with -DMML_OP_TIMING=<block>): python tools/assign_phases.py"""
import ctypes as C, importlib, sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
M = importlib.import_module("multi-modal-loam_amd"); synth = importlib.import_module("multi-modal-loam_amd.synth")
B = 1024
ctx = M.Context(max_scans=B, device=0)
scans = [(synth.velo_scan(k), synth.livox_scan(k)) for k in range(8)]
for s in range(B):
    v, l = scans[s % 8]; ctx.scan_upload(s, v, l)
ctx.synchronize()
for _ in range(2): ctx.extract(0, B)
ctx.synchronize()
lib = M.lib()
out = (C.c_ulonglong * 16)()
lib.mml_debug_op_timing(out, 1)
R = 5
for _ in range(R): ctx.extract(0, B)
ctx.synchronize()
lib.mml_debug_op_timing(out, 0)
names = ["records requested, point count read, counters zeroed", "barrier 1", "evaluate 8 points (waits for its records)", "barrier 2", "scan over groups", "barrier 3",
         "barrier 4 (wavefront 0: publish, look back)", "16-byte stores + LDS rows", "barrier 5", "row stores + drain"]
tot = sum(out[i] for i in range(10))
print("k_assign_onepass_s<0>, second wavefront of one block: clock64 ticks per launch, total %d" % (tot // R))
for i, nme in enumerate(names): print("  %-48s %8d  %5.1f%%" % (nme, out[i] // R, 100.0 * out[i] / max(tot, 1)))
print("first wavefront: from the kernel's start to its serial section %d, publish + look-back poll %d, reduce + block record %d" % (out[10] // R, out[11] // R, out[12] // R))
