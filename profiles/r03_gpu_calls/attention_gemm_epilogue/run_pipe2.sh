cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/pipe2; mkdir -p $O; rm -f $O/summary.txt
for p in 0 1 2; do
  rm -rf /tmp/prof_$p
  OASR_ATTN_PIPE=$p rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$p -o t -- python scripts/attn_kernel_times.py ${WHICH:-enc} ${BB:-32} 8 > $O/$p.log 2>&1
  f=$(find /tmp/prof_$p -name "*kernel_stats.csv" | head -1)
  echo "== PIPE=$p" | tee -a $O/summary.txt
  python - "$f" <<'PY' | tee -a $O/summary.txt
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "attn_" in n: print(f'  {n[:60]:60s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}')
PY
done
