"""Per-kernel summary (calls, total / average / min / max duration, share) from a rocprofv3 --kernel-trace run,
either the rocpd SQLite database (default output of rocprofv3 in ROCm 7.2) or *_kernel_trace.csv.

  python scripts/rocprof_summary.py trace_kernel_trace.csv [--by-queue]

--by-queue (csv only): one row per (kernel, queue) -- with the span step's side streams (csrc/engine.hip Runner::side_mode) launches
of one symbol run on different HIP streams = hardware queues, and a lowest-priority launch's begin-to-end span includes waiting for
compute units: its average is a queueing time, not a kernel time.  The header also prints the UNION of all dispatch intervals (the
time at least one kernel was running) beside their sum: sum - union = overlapped kernel time."""
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
    rows = db.execute("select name, start, end from kernels").fetchall()
    return [(short(n), None, int(s), int(e)) for n, s, e in rows]


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            q = r.get("Queue_Id", r.get("Stream_Id"))
            out.append((short(r["Kernel_Name"]), q, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    return out


def union_ns(intervals):
    """total length of the union of [start, end) intervals"""
    total, cur_s, cur_e = 0, None, None
    for s, e in sorted(intervals):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                total += cur_e - cur_s
            cur_s, cur_e = s, e
        elif e > cur_e:
            cur_e = e
    if cur_e is not None:
        total += cur_e - cur_s
    return total


def summarise(rows, by_queue=False):
    """rows: (name, queue, start, end) -> (header dict, [(label, calls, total_ns, min_ns, max_ns)] sorted by total)"""
    agg = {}
    for n, q, s, e in rows:
        key = (n, q) if by_queue else (n, None)
        d = e - s
        a = agg.setdefault(key, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    head = {"dispatches": len(rows), "sum_ns": total, "union_ns": union_ns([(s, e) for _, _, s, e in rows])}
    table = [((n if q is None else f"{n} @queue {q}"), a[0], a[1], a[2], a[3]) for (n, q), a in sorted(agg.items(), key=lambda kv: -kv[1][1])]
    return head, table


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    by_queue = "--by-queue" in sys.argv
    path = args[0]
    rows = from_db(path) if path.endswith(".db") else from_csv(path)
    head, table = summarise(rows, by_queue)
    total = head["sum_ns"]
    print(f"# {path}: {head['dispatches']} dispatches, {total / 1e6:.2f} ms of kernel time"
          + (f" (union of the dispatch intervals {head['union_ns'] / 1e6:.2f} ms: {(total - head['union_ns']) / 1e6:.2f} ms overlapped)" if head["union_ns"] < total else ""))
    w = 106 if by_queue else 92
    print(f"{'kernel':{w}s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'pct':>6s}")
    for n, calls, tot, mn, mx in table:
        print(f"{n:{w}s} {calls:7d} {tot / 1e6:10.3f} {tot / calls / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:10.2f} {100 * tot / total:6.2f}")


if __name__ == "__main__":
    main()
