"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) into per-kernel HBM bytes per launch.

MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB-like units of the L2's fabric-side request counters
(bytes = value * 1024), and on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read
-> the read side is doubled.  WRITE_SIZE is used as reported (uncalibrated per the guide).
usage: python scripts/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*\)$", "", name)


def load(path, counter):
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
    return tot, n


def main():
    f, nf = load(sys.argv[1], "FETCH_SIZE")
    w, nw = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w), key=lambda k: -(f[k] + w[k])):
        launches = max(nf[k], nw[k], 1)
        rd = 2.0 * f[k] * 1024 / max(nf[k], 1)
        wr = w[k] * 1024 / max(nw[k], 1)
        out[k] = {"launches": launches, "fetch_bytes_per_launch_x2_corrected": round(rd), "write_bytes_per_launch": round(wr),
                  "hbm_bytes_per_launch": round(rd + wr)}
    import datetime
    out["_meta"] = {"collected": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%d %H:%M UTC"),
                    "what": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate) of `python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile "
                            "--ab-steps 0`; reads x 2 (gfx950 correction, MI355X_MICROARCH.md), bytes = counter x 1024; side-stream and main-stream launches of a symbol are averaged together"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in [kv for kv in out.items() if kv[0] != "_meta"][:12]:
        print(f"{k[:80]:80s} {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch (rd {v['fetch_bytes_per_launch_x2_corrected'] / 1e6:.2f} wr {v['write_bytes_per_launch'] / 1e6:.2f})")


if __name__ == "__main__":
    main()
