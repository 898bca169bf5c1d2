#!/bin/bash
# call 25: side streams of the span step (OASR_SIDE_WGRAD bit 0: R-row weight gradients, bit 1: forward key|value projections, bit 2: backward
# key|value gradients, bit 3: lazy join without events): span parity tests under 7 and 15, then same-box A/B of the step
set -x
mkdir -p gpurun_out/r05s2
export OASR_TESTING_HOOKS=1
for m in 7 15; do
OASR_SIDE_WGRAD=$m timeout 600 python -m pytest tests/test_gpu_span.py tests/test_gpu_bench_shapes.py -x -q -m gpu 2>&1 | tail -2 > gpurun_out/r05s2/tests_side_$m.log
cat gpurun_out/r05s2/tests_side_$m.log
done
for v in 0 1 3 7 15 0 7; do
  OASR_SIDE_WGRAD=$v timeout 600 python bench.py --steps 10 --warmup 2 --ab-steps 0 --no-cpu-baseline 2>gpurun_out/r05s2/bench_$v.err | tail -1 > gpurun_out/r05s2/bench_$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r05s2/bench_$v.json"))
r=d["roofline"]
pc=d.get("parity",{}).get("span_step_vs_plain_step",{})
print("side_mode=$v ms/step", d["ms_per_step"], d.get("per_step_ms"), "dominant frac", r["frac"], "span parity", pc.get("grad_rel_l2"), "loss", pc.get("loss_span"), pc.get("loss_full"))
PY
done 2>&1 | grep side_mode | tee gpurun_out/r05s2/ab.txt
