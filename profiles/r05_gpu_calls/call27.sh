#!/bin/bash
# round 5, call 27: evidence of the FINAL binary (side streams of the span step on by default, mode 5): the GPU suite, smoke(), the default bench
# line, the driver-length line, and the rocprofv3 kernel statistics of the default command
mkdir -p gpurun_out/r05y
python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -8 > gpurun_out/r05y/suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05y/smoke.log 2>&1
python bench.py > gpurun_out/r05y/r05_bench_default.json 2> gpurun_out/r05y/err.log
python bench.py --steps 20 --warmup 3 > gpurun_out/r05y/r05_bench_steps20.json 2>> gpurun_out/r05y/err.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05y/trace -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --ab-steps 0 > gpurun_out/r05y/trace.log 2>&1
f=$(find gpurun_out/r05y/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/rocprof_summary.py "$f" > gpurun_out/r05y/r05_bench_default_kernel_stats.txt
rm -rf gpurun_out/r05y/trace
tail -3 gpurun_out/r05y/suite.log; tail -2 gpurun_out/r05y/smoke.log; head -12 gpurun_out/r05y/r05_bench_default_kernel_stats.txt | cut -c1-200
python - <<PY
import json
for f in ("r05_bench_default", "r05_bench_steps20"):
    j=json.loads(open(f"gpurun_out/r05y/{f}.json").read().strip().splitlines()[-1])
    print(f, j["ms_per_step"], j["value"], j["per_step_ms"], j["plain_step_ms"], j["span_bwd_ms"], j["step_frac_algorithmic"], j["step_frac_executed"], j["roofline"]["frac"], j["roofline"]["traffic"], j["config"]["side_streams"])
PY
