"""Per-kernel summary (calls, total / average / min / max duration, share) from a rocprofv3 --kernel-trace run,
either the rocpd SQLite database (default output of rocprofv3 in ROCm 7.2) or *_kernel_trace.csv."""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name[:90]


def from_db(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, (end - start) from kernels").fetchall()
    return [(short(n), d) for n, d in rows]


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((short(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main():
    path = sys.argv[1]
    rows = from_db(path) if path.endswith(".db") else from_csv(path)
    agg = {}
    for n, d in rows:
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print(f"# {path}: {len(rows)} dispatches, {total / 1e6:.2f} ms of kernel time")
    print(f"{'kernel':92s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'pct':>6s}")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:92s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:10.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:10.2f} {100 * a[1] / total:6.2f}")


if __name__ == "__main__":
    main()
