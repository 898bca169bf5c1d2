"""Per-kernel averages of rocprofv3 --pmc counters (csv output).  usage: python scripts/pmc_summary.py <dir-or-csv> [min_launches]"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*\)$", "", name)


def main():
    path = sys.argv[1]
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    tot = collections.defaultdict(lambda: collections.Counter())
    n = collections.defaultdict(lambda: collections.Counter())
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]] += 1
    for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0)):
        c = {name: tot[k][name] / n[k][name] for name in tot[k]}
        extra = ""
        if c.get("SQ_LDS_IDX_ACTIVE"):
            extra += f" lds_conflict_frac={c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.3f}"
        if c.get("SQ_BUSY_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            extra += f" mfma_busy/sq_busy={c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']:.3f}"
        print(f"{k[:90]:90s} n={max(n[k].values()):5d} " + " ".join(f"{a}={v:.3g}" for a, v in sorted(c.items())) + extra)


if __name__ == "__main__":
    main()
