#!/bin/bash
# round 6, call 1: side-stream mode 7 as the default with per-lane GEMM statistics: span tests, A/B 5 vs 7 on one box, the default bench line, by-queue rocprof
mkdir -p gpurun_out/r06a
O=gpurun_out/r06a
python -m pytest tests/test_gpu_span.py tests/test_gpu_bench_shapes.py -m gpu -q --timeout 1500 2>&1 | tail -5 > $O/span_tests.log
for m in 5 7 5 7; do
  OASR_TESTING_HOOKS=1 OASR_SIDE_STREAMS=$m python bench.py --steps 10 --warmup 2 --ab-steps 0 --no-cpu-baseline > $O/ab_$m.json 2>> $O/err.log
  python - <<PY >> $O/ab.txt
import json
j=json.loads(open("$O/ab_$m.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("side_mode=$m ms/step", j["ms_per_step"], j["per_step_ms"], "dominant", r["kernel"], "frac", r["frac"], "launches", r["launches_per_step"], "avg_us", r["avg_launch_us"], "side rows", {k:v for k,v in r["by_symbol"].items() if k.endswith("[side]")})
PY
done
python bench.py > $O/r06_bench_default_call1.json 2>> $O/err.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --ab-steps 0 > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/rocprof_summary.py "$f" --by-queue > $O/kernel_stats_by_queue.txt && python scripts/rocprof_summary.py "$f" > $O/kernel_stats.txt
rm -rf $O/trace
cat $O/span_tests.log; cat $O/ab.txt | cut -c1-400; head -14 $O/kernel_stats_by_queue.txt | cut -c1-200
