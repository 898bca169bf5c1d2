#!/bin/bash
# call 26: side streams, backward only (5 = R-row weight gradients + key|value gradients) against 7 and 0 on one box; span tests under 5
set -x
mkdir -p gpurun_out/r05s3
export OASR_TESTING_HOOKS=1
OASR_SIDE_WGRAD=5 timeout 600 python -m pytest tests/test_gpu_span.py -x -q -m gpu 2>&1 | tail -2 > gpurun_out/r05s3/tests_side_5.log
cat gpurun_out/r05s3/tests_side_5.log
for v in 5 0 7 5; do
  OASR_SIDE_WGRAD=$v timeout 600 python bench.py --steps 10 --warmup 2 --ab-steps 0 --no-cpu-baseline 2>gpurun_out/r05s3/bench_$v.err | tail -1 > gpurun_out/r05s3/bench_$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r05s3/bench_$v.json"))
r=d["roofline"]
pc=d.get("parity",{}).get("span_step_vs_plain_step",{})
print("side_mode=$v ms/step", d["ms_per_step"], d.get("per_step_ms"), "dominant frac", r["frac"], "span parity", pc.get("grad_rel_l2"), "loss", pc.get("loss_span"), pc.get("loss_full"))
PY
done 2>&1 | grep side_mode | tee gpurun_out/r05s3/ab.txt
