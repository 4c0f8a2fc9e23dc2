"""Turn a rocprofv3 rocpd database (…_results.db) into the per-kernel summary committed under profiles/.
Usage: python tools/rocpd_summary.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.md

Rows are (kernel, grid size) groups.  The FIRST dispatch of a group is its warm-up call (code object load, cold instruction and
data caches, the first touch of freshly allocated buffers: 1.3-2x the steady duration in these traces) and is left out of the
statistics when the group has at least three dispatches, so that `avg us` of a row is the steady launch time bench.py's HIP events
report as `stage_ms_per_launch`; the column `warm-up us` shows what was dropped."""
import sqlite3
import sys
from collections import OrderedDict


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    order = "start" if "start" in cols else ("start_timestamp" if "start_timestamp" in cols else "rowid")
    rows = c.execute("select name, grid_x*grid_y*grid_z as g, duration, vgpr_count, sgpr_count, lds_size, scratch_size "
                     "from kernels order by %s" % order).fetchall()
    groups = OrderedDict()
    for n, g, d, vg, sg, lds, scr in rows:
        groups.setdefault((n, g), []).append((d, vg, sg, lds, scr))
    out = []
    for (n, g), calls in groups.items():
        warm = None
        if len(calls) >= 3:
            warm, calls = calls[0][0], calls[1:]
        ds = [x[0] for x in calls]
        out.append((n, g, len(ds), sum(ds), sum(ds) / len(ds), min(ds), max(ds), warm, max(x[1] for x in calls), max(x[2] for x in calls),
                    max(x[3] for x in calls), max(x[4] for x in calls)))
    out.sort(key=lambda r: -r[3])
    total = sum(r[3] for r in out) or 1
    print("rocprofv3 --kernel-trace --stats, grouped by (kernel, grid size); durations from the dispatch timestamps; the first dispatch "
          "of every group of >= 3 is its warm-up call and is not in the statistics (column `warm-up us`).")
    print()
    print("| kernel | grid threads | calls | total ms | avg us | min us | max us | warm-up us | % | vgpr | sgpr | lds B | scratch B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for n, g, cnt, tot, avg, mn, mx, warm, vg, sg, lds, scr in out:
        short = n.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short.split("(")[0].split("<")[0][:60]
        print("| %s | %d | %d | %.3f | %.1f | %.1f | %.1f | %s | %.1f | %s | %s | %s | %s |" % (
            short, g, cnt, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, "-" if warm is None else "%.1f" % (warm / 1e3), 100.0 * tot / total,
            vg, sg, lds, scr))


if __name__ == "__main__":
    main(sys.argv[1])
