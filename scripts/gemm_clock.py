"""Effective shader clock of the ping-pong GEMM as a function of how many CUs run it (DVFS: the chip clocks to its power budget).
Run under  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace  (scripts/gemm_clock.sh): clock = GRBM_GUI_ACTIVE / kernel duration."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import _native as N  # noqa: E402
N.enable_testing_hooks()  # noqa: E402 -- this script steers kernel selection (include/oasr_testing.h)
from olmoasr_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def main():
    K = Nn = 4096
    w = (torch.randn(Nn, K, device=DEV) * 0.02).to(BF)
    N.lib().oasr_gemm_force_general(4)
    N.lib().oasr_gemm_set_variant(24)
    try:
        for rows, reps in ((512, 40), (2048, 40), (4096, 40), (192000, 6)):
            x = torch.randn(rows, K, device=DEV).to(BF)
            out = torch.empty(rows, Nn, device=DEV, dtype=BF)
            for _ in range(reps):
                ops.gemm(x, w, rows, Nn, K, out=out)
            torch.cuda.synchronize()
    finally:
        N.lib().oasr_gemm_force_general(0)
        N.lib().oasr_gemm_set_variant(-1)


def summarise(d):
    import collections
    import csv
    import glob
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    dur = {}
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X") or r.get("Grid_Size"))
    rows = collections.defaultdict(list)
    for f in cc:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "pp_kernel" not in r["Kernel_Name"]:
                continue
            if "Start_Timestamp" in r and r["Start_Timestamp"]:
                ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            else:
                ns = dur[r["Dispatch_Id"]][0]
            rows[r["Grid_Size"]].append((float(r["Counter_Value"]), ns))
    print("# oasr_gemm_pp_kernel, N = K = 4096, bf16, plain launches; GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3 -> / 8")
    print(f"{'grid (threads)':>15s} {'workgroups':>10s} {'launches':>8s} {'avg us':>9s} {'GRBM_GUI_ACTIVE':>16s} {'clock GHz':>10s} {'clock GHz (/8)':>14s}")
    for g in sorted(rows, key=lambda g: int(g)):
        v = rows[g][len(rows[g]) // 2:]  # second half: warmed up
        c = sum(a for a, _ in v) / len(v)
        ns = sum(b for _, b in v) / len(v)
        print(f"{g:>15s} {int(g) // 512:10d} {len(rows[g]):8d} {ns / 1e3:9.1f} {c:16.4g} {c / ns:10.3f} {c / ns / 8:14.3f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "summarise":
        summarise(sys.argv[2])
    else:
        main()
