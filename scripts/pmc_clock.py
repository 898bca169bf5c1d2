"""Effective shader clock per kernel from a rocprofv3 --pmc GRBM_GUI_ACTIVE pass (csv): the MI355X clocks to its power
budget (MI355X_MICROARCH.md, "DVFS give-back"), so the cycle-level MFMA bound of a kernel is the nominal peak x clock / 2.4 GHz.
clock = sum(GRBM_GUI_ACTIVE) / 8 XCDs / sum(dispatch duration).   usage: pmc_clock.py <dir-or-csv> [out.json]"""
import collections
import csv
import glob
import json
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
    cyc, ns, n = collections.Counter(), collections.Counter(), collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                continue
            k = short(r["Kernel_Name"])
            cyc[k] += float(r["Counter_Value"]) / 8.0
            ns[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            n[k] += 1
    out = {}
    print("# effective shader clock per kernel = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (counter pass: kernels run serialised;")
    print("# short kernels over-read because the counter window is wider than the dispatch timestamps -> rows under 50 us are indicative only)")
    print(f"{'kernel':90s} {'calls':>6s} {'avg_us':>9s} {'clock_GHz':>9s} {'mfma_peak_at_clock_TFLOPs':>26s}")
    for k in sorted(ns, key=lambda k: -ns[k]):
        ghz = cyc[k] / ns[k]
        out[k] = {"calls": n[k], "avg_us": ns[k] / n[k] / 1e3, "clock_ghz": ghz}
        print(f"{k[:90]:90s} {n[k]:6d} {ns[k] / n[k] / 1e3:9.1f} {ghz:9.3f} {2500.0 * ghz / 2.4:26.0f}")
    tot = sum(cyc.values()) / sum(ns.values())
    out["__all__"] = {"clock_ghz": tot}
    print(f"{'all kernels, time-weighted':90s} {sum(n.values()):6d} {'':9s} {tot:9.3f} {2500.0 * tot / 2.4:26.0f}")
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
