"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, as
MI355X_MICROARCH.md prescribes: TCC slots do not fit both).  On gfx950 FETCH_SIZE counts 64 B per 128-B request for
wide coalesced streams, i.e. half the bytes: it is doubled here (calibrated on k_undistort: 13.5 M points x 20 B
read = 270 MB, FETCH_SIZE says 134 MB; WRITE_SIZE matches the 270 MB written).  Counters are in KiB.
Usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <scans per launch> > profiles/traffic_r01.json"""
import json
import sys

import pandas as pd

STAGE = {"k_assign_init": "assign_count", "k_assign_a": "assign_count", "k_assign_b": "assign_scan", "k_assign_c": "assign_scatter",
         "k_stencil": "stencil", "k_stencil_break": "stencil", "k_stencil_redo": "stencil", "k_select": "select", "k_crop_a": "crop_compact", "k_crop_b": "crop_compact",
         "k_crop_c": "crop_compact", "k_crop": "crop_compact", "k_undistort_prep": "undistort", "k_undistort": "undistort",
         "k_voxel": "voxel_downsample", "k_assoc_prefix": "associate", "k_associate": "associate",
         "k_associate_hard": "associate_far", "k_associate_fit": "associate_far", "k_associate_fit_all": "associate_fit", "k_assoc_stats": "assoc_stats", "k_solve": "solve"}


def per_kernel(path, counter):
    d = pd.read_csv(path)
    full = d["Kernel_Name"].str.replace("(anonymous namespace)::", "", regex=False).str.replace("void ", "", regex=False).str.split("(").str[0]
    d["kfull"] = full                     # template arguments kept: k_select<4> and k_select<8> are separate launches
    d["k"] = full.str.split("<").str[0]
    d = d[d.Counter_Name == counter]
    mx = d.groupby("kfull")["Grid_Size"].transform("max")
    per_variant = d[d.Grid_Size == mx].groupby(["k", "kfull"])["Counter_Value"].mean()
    return per_variant.groupby(level=0).sum()


def main(fp, wp, scans):
    f, w = per_kernel(fp, "FETCH_SIZE"), per_kernel(wp, "WRITE_SIZE")
    out = {}
    for k, st in STAGE.items():
        b = 2.0 * float(f.get(k, 0.0)) * 1024.0 + float(w.get(k, 0.0)) * 1024.0
        out[st] = out.get(st, 0.0) + b
    out = {k: round(v) for k, v in out.items()}
    out["scans_per_launch"] = int(scans)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else 256)
