"""Cycles per phase of one k_stencil wavefront (library built with -DMML_ST_TIMING=<line>): python tools/stencil_phases.py"""
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
lib.mml_debug_st_timing(out, 1)
R = 5
for _ in range(R): ctx.extract(0, B)
ctx.synchronize()
lib.mml_debug_st_timing(out, 0)
names = ["wait-prev", "load+sq", "rounds", "walk", "list+c150", "write"]
tot = sum(out[i] for i in range(6))
print("k_stencil phases, cycles per launch (one workgroup), total %d" % (tot // R))
for i, nme in enumerate(names): print("  %-10s %8d  %5.1f%%" % (nme, out[i] // R, 100.0 * out[i] / max(tot, 1)))
