#!/bin/bash
# round 5, call 1: the new parity bounds (bf16 ulp bound at tiny / medium B=4 direct vs oracle, tightened autograd bound), span tests,
# then the new default bench line (span-forward headline + plain / span-backward in the same run)
set -x
mkdir -p gpurun_out/r05c1
python -m pytest tests/test_gpu_model.py tests/test_gpu_autograd.py tests/test_gpu_span.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r05c1/tests_a.log
python -m pytest tests/test_gpu_parity_sizes.py -x -q -m gpu -s -k "step_vs_oracle" 2>&1 | grep -v "^$" | tail -80 > gpurun_out/r05c1/tests_b.log
python bench.py > gpurun_out/r05c1/bench_default.json 2> gpurun_out/r05c1/bench_default.err
tail -5 gpurun_out/r05c1/tests_a.log gpurun_out/r05c1/tests_b.log
tail -c 3000 gpurun_out/r05c1/bench_default.json
