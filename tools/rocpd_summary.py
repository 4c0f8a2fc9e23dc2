"""Turn a rocprofv3 rocpd database (…_results.db) into the per-kernel summary committed under profiles/.
Usage: python tools/rocpd_summary.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.md"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), grid_x*grid_y*grid_z as g "
                     "from kernels group by name, g order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print("rocprofv3 --kernel-trace --stats, grouped by (kernel, grid size); durations from the dispatch timestamps.")
    print()
    print("| kernel | grid threads | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | scratch B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for n, cnt, tot, avg, mn, mx, vg, sg, lds, scr, g in rows:
        short = n.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short.split("(")[0].split("<")[0][:60]
        print("| %s | %d | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s |" % (
            short, g, cnt, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, vg, sg, lds, scr))


if __name__ == "__main__":
    main(sys.argv[1])
