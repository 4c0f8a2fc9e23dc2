"""Per-kernel SQ counter summary (waves, VALU / SALU instructions per wave, wait / active shares of the wave cycles) from the
counter_collection.csv of a rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU pass (split over as many passes as the counter slots need; concatenate the CSVs).
With --json <file> <scans per launch> it also writes, per stage (tools/stages.py), the VALU / SALU wave-instructions of ONE
launch over that many scans (the largest-grid launches of every kernel variant, i.e. bench.py's single-stream pass): bench.py
divides them by the stage's launch time to price issue-bound kernels against the VALU issue peak.
Usage: python tools/pmc_sq.py counter_collection.csv [--json profiles/sq_rNN.json 1024] > profiles/rNN_sq_counters.md"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    k = re.sub(r"\(.*", "", k)
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        cnt[k] += 1
print("%-34s %4s %9s %9s %9s %6s %6s %6s %6s" % ("kernel", "n", "waves", "valu/wv", "salu/wv", "wait%", "stall%", "act%", "valu%"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:24]:
    w = v["SQ_WAVES"] or 1
    wc = v["SQ_WAVE_CYCLES"] or 1
    print("%-34s %4d %9.3g %9.0f %9.0f %6.1f %6.1f %6.1f %6.1f" % (
        k[:34], cnt[k], v["SQ_WAVES"], v["SQ_INSTS_VALU"] / w, v["SQ_INSTS_SALU"] / w, 100 * v["SQ_WAIT_ANY"] / wc,
        100 * v["SQ_WAIT_INST_ANY"] / wc, 100 * v["SQ_ACTIVE_INST_ANY"] / wc, 100 * v["SQ_ACTIVE_INST_VALU"] / wc))

if "--json" in sys.argv:
    import json
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from stages import short_name, stage_of
    out_path, scans = sys.argv[sys.argv.index("--json") + 1], int(sys.argv[sys.argv.index("--json") + 2])
    per = collections.defaultdict(lambda: collections.defaultdict(float))     # (variant, dispatch) -> counter sums
    grid = {}
    for r in rows:
        kf = short_name(r["Kernel_Name"], keep_template=True)
        per[(kf, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        grid[(kf, r["Dispatch_Id"])] = int(r["Grid_Size"])
    gmax = collections.defaultdict(int)
    for (kf, _), g in grid.items():
        gmax[kf] = max(gmax[kf], g)
    acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for (kf, d), v in per.items():
        if grid[(kf, d)] != gmax[kf]:
            continue
        a = acc[kf]
        a[0] += v["SQ_INSTS_VALU"]
        a[1] += v["SQ_INSTS_SALU"]
        a[2] += 1
    stage = collections.defaultdict(lambda: {"valu_wave_instr": 0.0, "salu_wave_instr": 0.0})
    for kf, (va, sa, n) in acc.items():
        st = stage_of(kf)
        if st is None or n == 0:
            continue
        stage[st]["valu_wave_instr"] += va / n
        stage[st]["salu_wave_instr"] += sa / n
    res = {k: {kk: round(vv) for kk, vv in v.items()} for k, v in stage.items()}
    res["scans_per_launch"] = scans
    res["lib_sha16"] = os.environ.get("MML_LIB_SHA16")  # the library these counts were taken from (bench.py marks a mismatch as stale)
    json.dump(res, open(out_path, "w"), indent=1)
