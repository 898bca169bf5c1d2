#!/bin/bash
# call 24: decoder-backward weight gradients on a low-priority side stream (OASR_SIDE_WGRAD=1): span parity tests, then same-box A/B of the step
set -x
mkdir -p gpurun_out/r05s
export OASR_TESTING_HOOKS=1
OASR_SIDE_WGRAD=1 timeout 600 python -m pytest tests/test_gpu_span.py tests/test_gpu_bench_shapes.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05s/tests_side.log
cat gpurun_out/r05s/tests_side.log
for v in 0 1 0 1; do
  OASR_SIDE_WGRAD=$v timeout 600 python bench.py --steps 10 --warmup 2 --ab-steps 0 --no-cpu-baseline 2>gpurun_out/r05s/bench_$v.err | tail -1 > gpurun_out/r05s/bench_$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r05s/bench_$v.json"))
r=d["roofline"]
print("side_wgrad=$v ms/step", d["ms_per_step"], d.get("per_step_ms"), "gemm_ms", r.get("gemm_ms_per_step"), "kernel_sum_ms", r.get("kernel_ms_per_step"), "span parity", d.get("parity",{}).get("span_step_vs_plain_step",{}).get("grad_rel_l2"), "loss", d.get("parity",{}).get("span_step_vs_plain_step",{}).get("loss_span"))
PY
done 2>&1 | grep side_wgrad | tee gpurun_out/r05s/ab.txt
