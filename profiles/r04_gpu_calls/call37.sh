#!/bin/bash
# round 4, GPU call 37: the other BASELINE configurations on the final code
mkdir -p gpurun_out/r04
for cfg in "base 256 128" "small 256 128" "large 512 64"; do
  set -- $cfg
  timeout 400 python bench.py --variant $1 --per-gpu-batch $2 --micro-batch $3 --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print(json.dumps({k:(j.get(k) if k!='workload' else j['config']['workload']) for k in ('value','ms_per_step','step_model_tflops_per_gpu','step_frac_of_mfma_peak','executed_over_algorithmic_flops','workload')}))"
done | tee gpurun_out/r04/call37_variants.txt
