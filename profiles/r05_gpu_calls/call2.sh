#!/bin/bash
# round 5, call 2: the one-launch decoder step engine (csrc/decode_xcd.hip): bit-identity tests against the multi-launch engines, then timing
set -x
mkdir -p gpurun_out/r05c2
timeout 600 python -m pytest tests/test_gpu_decode_step.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r05c2/tests_step.log
tail -5 gpurun_out/r05c2/tests_step.log
for v in small medium; do timeout 300 python scripts/decode_xcd_probe.py $v 1 32 1,2,3,4 2>&1 | grep -v "^$" | tail -6 >> gpurun_out/r05c2/probe.log; done
timeout 200 python scripts/decode_xcd_probe.py medium 1 200 1,2 2>&1 | tail -2 >> gpurun_out/r05c2/probe.log
timeout 200 python scripts/decode_xcd_probe.py small 4 32 1,2,3 2>&1 | tail -3 >> gpurun_out/r05c2/probe.log
cat gpurun_out/r05c2/probe.log
timeout 900 python -m pytest tests/test_gpu_decode_parity.py tests/test_gpu_modules.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05c2/tests_parity.log
tail -4 gpurun_out/r05c2/tests_parity.log
