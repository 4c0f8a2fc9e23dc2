"""Per-kernel LDS and L2 counter summary from two rocprofv3 --pmc passes (their counter_collection.csv files concatenated or given
one after the other): SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS, and TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum.
bank-conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra cycles over all LDS-array cycles, MI355X_MICROARCH.md);
L2 hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum).
Usage: python tools/pmc_lds_l2.py lds.csv tcc.csv > profiles/rNN_lds_l2_counters.md"""
import collections
import csv
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.sub(r"\(.*", "", k)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add((path, r["Dispatch_Id"]))
print("| kernel | dispatches | LDS instr (M) | LDS-array cycles (M) | bank-conflict cycles (M) | conflict share | L2 requests (M) | L2 hit rate |")
print("|---|---|---|---|---|---|---|---|")
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1]["SQ_LDS_IDX_ACTIVE"] + kv[1]["TCC_REQ_sum"]))[:28]:
    idx = v["SQ_LDS_IDX_ACTIVE"]
    hm = v["TCC_HIT_sum"] + v["TCC_MISS_sum"]
    print("| %s | %d | %.1f | %.1f | %.1f | %s | %.1f | %s |" % (
        k[:40], len(cnt[k]) // max(1, len(sys.argv) - 1), v["SQ_INSTS_LDS"] / 1e6, idx / 1e6, v["SQ_LDS_BANK_CONFLICT"] / 1e6,
        ("%.1f %%" % (100 * v["SQ_LDS_BANK_CONFLICT"] / idx)) if idx else "-", v["TCC_REQ_sum"] / 1e6,
        ("%.1f %%" % (100 * v["TCC_HIT_sum"] / hm)) if hm else "-"))
