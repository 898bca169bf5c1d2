#!/bin/bash
# round 4, GPU call 22: is the span step's cross-attention backward bound by memory latency?  keys / values per sample vs shared (L2-resident)
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 300 python scripts/cross_bwd_probe.py 2>&1 | tee gpurun_out/r04/call22_cross_probe.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/cross_trace -o t -- python scripts/cross_bwd_probe.py > /dev/null 2>&1
f=$(find gpurun_out/r04/cross_trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee -a gpurun_out/r04/call22_cross_probe.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# the probe alternates per-sample / shared in four blocks of 11 launches (1 warm + 10): split each kernel's launches in four
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "attn_bwd" in n or "attn_fwd" in n:
        by[n.split("(")[0]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for n, v in by.items():
    v.sort()
    q = len(v) // 4
    print(n[:60], " | ".join(f"{sum(x[1] for x in v[i*q:(i+1)*q]) / max(q,1):8.1f} us" for i in range(4)), "(per sample | shared | per sample | shared)")
PY
rm -rf gpurun_out/r04/cross_trace
