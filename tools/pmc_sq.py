"""Per-kernel SQ counter summary (waves, VALU / SALU instructions per wave, wait / active shares of the wave cycles) from the
counter_collection.csv of a rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU pass (split over as many passes as the counter slots need; concatenate the CSVs).
Usage: python tools/pmc_sq.py counter_collection.csv > profiles/rNN_sq_counters.md"""
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
