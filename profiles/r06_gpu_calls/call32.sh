#!/bin/bash
# round 6, call 32: rocprofv3 of the KV-cached decoder step at the library default (chip-wide engine): kernel trace + stats, HBM traffic counters (separate passes)
O=gpurun_out/r06dp
mkdir -p $O
export TMPDIR=/tmp OASR_TESTING_HOOKS=1
CMD="python scripts/decode_xcd_probe.py medium 1 32 -1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- $CMD > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/rocprof_summary.py "$f" > $O/r06_decode_step_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- $CMD > $O/pmc_$c.log 2>&1
done
ff=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
fw=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python scripts/pmc_traffic.py "$ff" "$fw" $O/r06_decode_step_hbm_traffic.json > $O/r06_decode_step_hbm_traffic.txt
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
grep -v amdgpu $O/trace.log | tail -2; head -12 $O/r06_decode_step_kernel_stats.txt | cut -c1-200; head -8 $O/r06_decode_step_hbm_traffic.txt
