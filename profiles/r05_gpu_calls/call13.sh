#!/bin/bash
# round 5, call 13: one-launch decoder step, version 6 (table-driven producer: units listed once per launch, unit boundaries cost one LDS read):
# bit-identity tests (plain + the checked instantiation), stamps, timing of every engine
set -x
mkdir -p gpurun_out/r05c13
timeout 600 python -m pytest tests/test_gpu_decode_step.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05c13/tests_step.log
OASR_XCD_FLAGS=8 timeout 300 python -m pytest tests/test_gpu_decode_step.py -x -q -m gpu -k "long_window or bit_identical" 2>&1 | tail -5 >> gpurun_out/r05c13/tests_step.log
cat gpurun_out/r05c13/tests_step.log | tail -12
OASR_XCD_FLAGS=256 timeout 200 python scripts/decode_xcd_probe.py medium 1 32 2 2>&1 | grep -v "^$\|amdgpu.ids" >> gpurun_out/r05c13/stamps.log
OASR_XCD_FLAGS=256 timeout 200 python scripts/decode_xcd_probe.py small 1 32 2 2>&1 | grep -v "^$\|amdgpu.ids" >> gpurun_out/r05c13/stamps.log
for v in small medium; do timeout 300 python scripts/decode_xcd_probe.py $v 1 32 1,2,3,4 2>&1 | grep -v "^$\|amdgpu.ids" | tail -4 >> gpurun_out/r05c13/probe.log; done
timeout 200 python scripts/decode_xcd_probe.py medium 1 200 1,2 2>&1 | tail -2 >> gpurun_out/r05c13/probe.log
timeout 200 python scripts/decode_xcd_probe.py small 4 32 1,2 2>&1 | tail -2 >> gpurun_out/r05c13/probe.log
cat gpurun_out/r05c13/stamps.log gpurun_out/r05c13/probe.log
