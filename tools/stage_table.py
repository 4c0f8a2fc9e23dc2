"""Markdown table of DESIGN.md section 7 from the committed measurement files of one round:
python tools/stage_table.py profiles/r03f_bench.json profiles/sq_r03.json profiles/traffic_r03.json"""
import json
import sys

PEAK = 256 * 4 * 2.4e9 / 4  # VALU wave-instructions / s (fall-back: the bench line's measured v_fma_f32 rate is used when it has one)

GROUPS = [("assign", ["assign_count", "assign_scan", "assign_scatter", "assign_ends", "assign_onepass", "assign_tables"]),
          ("stencil (3 kernels)", ["stencil"]),
          ("select", ["select"]), ("undistort", ["undistort"]), ("solve", ["solve"]),
          ("associate + fit + far", ["associate", "associate_fit", "associate_far"]), ("voxel", ["voxel_downsample"]),
          ("crop, stats", ["crop_compact", "assoc_stats", "livox_extrinsic"])]


def main(bench, sq, traffic):
    b = json.loads(open(bench).read().strip().splitlines()[-1])
    st = b["roofline"]["stage_ms_per_launch"]
    global PEAK
    iss = b["roofline"].get("issue") or {}
    if iss.get("peak_source") == "measured":
        PEAK = iss["peak"]
        print("(issue peak: measured v_fma_f32 rate of the box, %.3g wave-instructions/s; v_add_u32: %.3g)\n" % (iss["peak"], iss.get("peak_int") or 0))
    sq = json.load(open(sq))
    tr = json.load(open(traffic))
    print("| stage | ms | VALU (M wave-instructions, % of issue peak in the stage's own time) | HBM (GB, TB/s) |")
    print("|---|---|---|---|")
    tv = tb = tm = 0.0
    for name, keys in GROUPS:
        keys = [k for k in keys if k in st]   # (three-pass or one-pass bucketing: whichever stages the run had)
        if not keys:
            continue
        ms = sum(st[k] for k in keys)
        v = sum(sq[k]["valu_wave_instr"] for k in keys if k in sq)
        by = sum(tr[k] for k in keys if k in tr)
        tv += v
        tb += by
        tm += ms
        print("| %s | %.2f | %.0f, %.0f %% | %.2f, %.1f |" % (name, ms, v / 1e6, 100 * v / (ms * 1e-3) / PEAK, by / 1e9, by / 1e9 / ms))
    print("| all | %.2f | %.0f | %.2f |" % (tm, tv / 1e6, tb / 1e9))
    n = b["roofline"]["scans_per_launch"]
    step_ms = n / b["value"] * 1e3
    print("\nstep: %.2f ms per %d scans at %.1f k scans/s; VALU at the issue peak %.2f ms (%.0f %% of the step); traffic at the measured copy roof "
          "(%.2f TB/s) %.2f ms (%.0f %%)" % (step_ms, n, b["value"] / 1e3, tv / PEAK * 1e3, 100 * tv / PEAK * 1e3 / step_ms,
                                             b["roofline"]["measured_copy_GBps"] / 1e3, tb / (b["roofline"]["measured_copy_GBps"] * 1e9) * 1e3,
                                             100 * tb / (b["roofline"]["measured_copy_GBps"] * 1e9) * 1e3 / step_ms))


if __name__ == "__main__":
    main(*sys.argv[1:4])
