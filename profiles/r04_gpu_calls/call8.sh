#!/bin/bash
# round 4, GPU call 8: the whole GPU suite on the final code, smoke(), the other BASELINE configurations
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -15 > gpurun_out/r04/call8_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/call8_smoke.log 2>&1
python scripts/hostinfo.py > gpurun_out/r04/r04_hostinfo.txt 2>&1
for cfg in "base 256 0" "small 256 0" "large 512 64"; do
  set -- $cfg
  python bench.py --variant $1 --per-gpu-batch $2 --micro-batch $3 --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps({k: j[k] for k in ('value','ms_per_step','step_model_tflops_per_gpu','step_frac_of_mfma_peak','executed_over_algorithmic_flops')} | {'workload': j['config']['workload']}))"
done > gpurun_out/r04/call8_variants.txt
tail -6 gpurun_out/r04/call8_suite.log; tail -3 gpurun_out/r04/call8_smoke.log; cat gpurun_out/r04/call8_variants.txt
