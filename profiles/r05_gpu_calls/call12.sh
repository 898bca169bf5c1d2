#!/bin/bash
# round 5, call 12: the default bench line of the evidence binary with profiles/r05_hbm_traffic.json in place (roofline.traffic), + the other variants
mkdir -p gpurun_out/r05b
python bench.py > gpurun_out/r05b/r05_bench_default.json 2> gpurun_out/r05b/err.log
for v in base small large; do
  pgb=256; [ $v = large ] && pgb=512
  python bench.py --variant $v --per-gpu-batch $pgb --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 > gpurun_out/r05b/variant_$v.json
done
python - <<PY
import json
j=json.loads(open("gpurun_out/r05b/r05_bench_default.json").read().strip().splitlines()[-1])
print("default", j["ms_per_step"], j["value"], j["per_step_ms"], j["plain_step_ms"], j["span_bwd_ms"], j["step_frac_algorithmic"], j["step_frac_executed"], j["roofline"]["frac"], j["roofline"]["traffic"])
for v in ("base","small","large"):
    j=json.loads(open(f"gpurun_out/r05b/variant_{v}.json").read())
    print(v, j["ms_per_step"], j["value"], j["plain_step_ms"], j["span_bwd_ms"], j["step_frac_algorithmic"], j["config"]["micro_batch"], j["config"]["workload"])
PY
