"""Cycles per phase of one k_voxel<512> workgroup (library built with -DMML_VX_TIMING=<kind: 0 corner | 1 surf>): python tools/voxel_phases.py"""
import ctypes as C, importlib, sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
M = importlib.import_module("multi-modal-loam_amd"); synth = importlib.import_module("multi-modal-loam_amd.synth")
B = 1024
ctx = M.Context(max_scans=B, device=0)
scans = [(synth.velo_scan(k), synth.livox_scan(k)) for k in range(8)]
for s in range(B):
    v, l = scans[s % 8]; ctx.scan_upload(s, v, l)
ctx.synchronize()
dR, dt = np.tile(np.eye(3).reshape(1, 9), (B, 1)), np.zeros((B, 3))
ctx.extract(0, B); ctx.undistort(0, B, dR, dt); ctx.downsample(0, B)
ctx.synchronize()
lib = M.lib(); out = (C.c_ulonglong * 16)()
lib.mml_debug_vx_timing(out, 1)
R = 5
for _ in range(R): ctx.downsample(0, B)
ctx.synchronize(); lib.mml_debug_vx_timing(out, 0)
names = [(4, "list + point gathers, min / max"), (5, "min / max reduction (1 barrier)"), (6, "voxel keys"), (7, "radix sort"), (0, "(return)"),
         (1, "centroid: gathers to LDS, heads (per chunk)"), (2, "centroid: per-voxel sums"), (3, "centroid: barriers")]
tot = sum(out[i] for i, _ in names)
print("k_voxel<512> phases, clock64 ticks per launch (second wavefront of one workgroup), total %d" % (tot // R))
for i, nme in names: print("  %-46s %8d  %5.1f%%" % (nme, out[i] // R, 100.0 * out[i] / max(tot, 1)))
