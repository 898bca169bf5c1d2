"""MFMA utilisation per kernel from the round's three profile summaries.
usage: mfma_util.py <kernel_stats.txt> <sq_counters.txt> <effective_clock.json> > profiles/rNN_mfma_utilisation.txt"""
import json
import re
import sys


def main():
    stats, sq, clk = sys.argv[1:4]
    dur, pct = {}, {}
    for line in open(stats):
        m = re.match(r"^(?:void )?(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m:
            dur[m.group(1).strip()] = float(m.group(4))
            pct[m.group(1).strip()] = float(m.group(7))
    clocks = json.load(open(clk))
    rows = []
    for line in open(sq):
        name = line[:90].strip()
        vals = dict((k, float(v)) for k, v in re.findall(r"(\w+)=([\d.e+]+)", line[90:]))
        if name in dur and vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
            ghz = clocks.get(name, {}).get("clock_ghz")
            ghz = ghz if ghz and ghz > 0.5 else None  # (first-dispatch counter set-up inflates a 4-launch kernel's window)
            rows.append((name, dur[name], pct[name], vals["SQ_VALU_MFMA_BUSY_CYCLES"], ghz, vals.get("lds_conflict_frac", 0.0)))
    print("# MFMA utilisation per kernel (OLMoASR-medium bench step, one MI355X).")
    print("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES per launch (rocprofv3 --pmc pass): cycles the matrix pipe of a SIMD is busy, summed over the")
    print("# chip's 1024 SIMDs.  util_2.4GHz = mfma_busy / (1024 x avg duration [--kernel-trace pass] x 2.4 GHz) -- for the GEMMs this equals their")
    print("# algorithmic-FLOP fraction of the 2.5 PFLOP/s peak in the bench line.  clock = effective shader clock of the kernel in the step")
    print("# (GRBM_GUI_ACTIVE pass, scripts/pmc_clock.py): the chip clocks to its power budget, so util_at_clock = the fraction of the cycles")
    print("# that actually elapsed in which the matrix pipe was busy.  lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.")
    print(f"{'kernel':64s} {'avg_us':>8s} {'step_pct':>8s} {'mfma_busy/launch':>17s} {'util_2.4GHz':>11s} {'clock_GHz':>9s} {'util_at_clock':>13s} {'lds_conflict':>12s}")
    for name, us, p, busy, ghz, ldsc in sorted(rows, key=lambda r: -r[2]):
        u = busy / (1024 * us * 1e-6 * 2.4e9)
        ua = busy / (1024 * us * 1e-6 * ghz * 1e9) if ghz else float("nan")
        print(f"{name[:64]:64s} {us:8.1f} {p:8.2f} {busy:17.3e} {u:11.3f} {ghz or float('nan'):9.3f} {ua:13.3f} {ldsc:12.3f}")


if __name__ == "__main__":
    main()
