"""How many kernels run at once?  Reads a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) and prints, for the longest
gap-free stretch of the run (the timed region of bench.py), the share of wall time spent with 0, 1, 2, … kernels in flight and,
per kernel name, the time during which it was the ONLY kernel running."""
import csv
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from stages import short_name  # noqa: E402


def main(path, skip_frac=0.3):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short_name(r["Kernel_Name"], keep_template=True)))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t0 + (t1 - t0) * skip_frac  # skip set-up and warm-up
    ev = []
    for s, e, n in rows:
        if e <= lo:
            continue
        ev.append((max(s, lo), 1, n))
        ev.append((e, -1, n))
    ev.sort(key=lambda x: (x[0], x[1]))
    live = defaultdict(int)
    nlive = 0
    hist = defaultdict(float)
    alone = defaultdict(float)
    prev = ev[0][0]
    for t, d, n in ev:
        dt = t - prev
        if dt > 0:
            hist[nlive] += dt
            if nlive == 1:
                alone[next(k for k, v in live.items() if v > 0)] += dt
        prev = t
        live[n] += d
        nlive += d
    tot = sum(hist.values())
    print("wall ms", tot / 1e6)
    for k in sorted(hist):
        print("  %d kernels in flight: %5.1f %%" % (k, 100 * hist[k] / tot))
    print("alone (ms, % of wall):")
    for n, v in sorted(alone.items(), key=lambda kv: -kv[1])[:15]:
        print("  %-60s %8.2f %5.1f" % (n[:60], v / 1e6, 100 * v / tot))


if __name__ == "__main__":
    main(sys.argv[1])
