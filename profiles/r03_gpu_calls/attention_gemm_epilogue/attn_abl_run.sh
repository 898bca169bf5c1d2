#!/bin/bash
# runs every scratch/abl/liboasr_*.so given as args through rocprofv3 kernel stats on the encoder problem
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/attn_abl; mkdir -p $O
WHICH=${WHICH:-enc}; BB=${BB:-32}
for tag in "$@"; do
  rm -rf /tmp/prof_$tag
  OASR_LIB=/root/repo/scratch/abl/liboasr_$tag.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o t -- python scripts/attn_kernel_times.py $WHICH $BB 5 > $O/$tag.log 2>&1
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag" | tee -a $O/summary.txt
  python - "$f" <<'PY' | tee -a $O/summary.txt
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "attn_" in n: print(f'  {n[:40]:40s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}')
PY
done
