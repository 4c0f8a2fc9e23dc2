"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, as
MI355X_MICROARCH.md prescribes: TCC slots do not fit both).  On gfx950 FETCH_SIZE counts 64 B per 128-B request for
wide coalesced streams, i.e. half the bytes: it is doubled here (calibrated on k_undistort: 13.5 M points x 20 B
read = 270 MB, FETCH_SIZE says 134 MB; WRITE_SIZE matches the 270 MB written).  Counters are in KiB.  Per kernel variant the
launches of the LARGEST grid are taken -- the single-stream pass bench.py runs after its timed region, every kernel over the
whole `scans per launch` batch -- and summed per stage (tools/stages.py).
Usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <scans per launch> > profiles/traffic_rNN.json"""
import json
import os
import sys

import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stages import short_name, stage_of  # noqa: E402


def per_stage(path, counter):
    d = pd.read_csv(path)
    d = d[d.Counter_Name == counter].copy()
    d["kfull"] = d["Kernel_Name"].map(lambda n: short_name(n, keep_template=True))  # k_select<4> and <8> are separate launches
    d["stage"] = d["Kernel_Name"].map(stage_of)
    d = d[d.stage.notna()]
    mx = d.groupby("kfull")["Grid_Size"].transform("max")
    per_variant = d[d.Grid_Size == mx].groupby(["stage", "kfull"])["Counter_Value"].mean()
    by_kernel = {k: float(v) for (st, k), v in per_variant.items()}
    return per_variant.groupby(level=0).sum(), sorted(set(d[d.Grid_Size == mx]["kfull"])), by_kernel


def main(fp, wp, scans):
    f, kf, fk = per_stage(fp, "FETCH_SIZE")
    w, _, wk = per_stage(wp, "WRITE_SIZE")
    out = {}
    for st in sorted(set(f.index) | set(w.index)):
        out[st] = round(2.0 * float(f.get(st, 0.0)) * 1024.0 + float(w.get(st, 0.0)) * 1024.0)
    out["scans_per_launch"] = int(scans)
    out["kernels"] = kf
    # per kernel variant: [bytes fetched (doubled as above), bytes written] per launch
    out["by_kernel"] = {k: [round(2.0 * fk.get(k, 0.0) * 1024.0), round(wk.get(k, 0.0) * 1024.0)] for k in sorted(set(fk) | set(wk))}
    out["lib_sha16"] = os.environ.get("MML_LIB_SHA16")  # the library these bytes were counted on (bench.py marks a mismatch as stale)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else 256)
