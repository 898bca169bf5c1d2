"""scripts/rocprof_summary.py (the per-kernel table committed under profiles/): aggregation, the per-queue split and the union of the
dispatch intervals on a synthetic kernel trace -- two streams, one symbol on both, overlapping launches."""
import csv
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("rocprof_summary", os.path.join(ROOT, "scripts", "rocprof_summary.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


ROWS = [  # name, queue, start, end (ns)
    ("void (anonymous namespace)::gemm<true>(Args)", "1", 0, 100),
    ("void (anonymous namespace)::gemm<true>(Args)", "1", 100, 250),
    ("void (anonymous namespace)::gemm<true>(Args)", "2", 50, 450),   # low-priority filler: overlaps both launches of queue 1
    ("ln_kernel(float*)", "1", 250, 300),
    ("ln_kernel(float*)", "1", 500, 520),                             # after a gap
]


def test_union_of_intervals():
    m = _load()
    assert m.union_ns([]) == 0
    assert m.union_ns([(0, 10), (10, 20)]) == 20
    assert m.union_ns([(5, 7), (0, 10), (8, 12), (20, 21)]) == 13


def test_summary_by_symbol_and_by_queue(tmp_path):
    m = _load()
    rows = [(m.short(n), q, s, e) for n, q, s, e in ROWS]
    head, table = m.summarise(rows)
    assert head == {"dispatches": 5, "sum_ns": 100 + 150 + 400 + 50 + 20, "union_ns": 450 + 20}
    assert table[0] == ("void gemm<true>", 3, 650, 100, 400) and table[1] == ("ln_kernel", 2, 70, 20, 50)
    head_q, table_q = m.summarise(rows, by_queue=True)
    assert head_q == head
    assert table_q[0] == ("void gemm<true> @queue 2", 1, 400, 400, 400)
    assert table_q[1] == ("void gemm<true> @queue 1", 2, 250, 100, 150)
    # the command line on a csv in rocprofv3's column naming
    path = tmp_path / "t_kernel_trace.csv"
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Agent_Id", "Queue_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for n, q, s, e in ROWS:
            w.writerow(["KERNEL_DISPATCH", "0", q, n, s, e])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rocprof_summary.py"), str(path), "--by-queue"], capture_output=True, text=True, check=True).stdout
    assert "5 dispatches" in out  # (720 ns of kernel time, 470 ns with at least one kernel running)
    assert "union of the dispatch intervals" in out
    assert "void gemm<true> @queue 2" in out and "ln_kernel @queue 1" in out
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rocprof_summary.py"), str(path)], capture_output=True, text=True, check=True).stdout
    assert "@queue" not in plain and "void gemm<true>" in plain
