"""Per-launch durations of the full-window kernels from a rocprofv3 --kernel-trace CSV (debug helper)."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
by = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    if "k_fw_" in name:
        by[re.search(r"k_fw_\w+", name).group(0)].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in by.items():
    d = sorted(e - s for s, e in v)
    print(k, "launches", len(d), "min/median/p90/max us: %.1f %.1f %.1f %.1f" % (d[0] / 1e3, d[len(d) // 2] / 1e3, d[int(len(d) * 0.9)] / 1e3, d[-1] / 1e3))
    print("   all (us):", " ".join("%.0f" % (x / 1e3) for x in d))
allk = sorted((s, e, k) for k, v in by.items() for s, e in v)
gaps = [allk[i + 1][0] - allk[i][1] for i in range(len(allk) - 1)]
gaps = sorted(g for g in gaps if g < 100000)
if gaps:
    print("gap between consecutive fw kernels us: median %.1f p90 %.1f" % (gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3))
