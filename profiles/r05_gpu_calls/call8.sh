#!/bin/bash
# round 5, call 8: one-launch decoder step, version 5 (producer units span the tiles of a segment, a phase streams its tiles back to back with one hand-over per group, merge in head groups):
# bit-identity tests (plain + the checked instantiation), stamps, timing of every engine
set -x
mkdir -p gpurun_out/r05c8
timeout 600 python -m pytest tests/test_gpu_decode_step.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05c8/tests_step.log
OASR_XCD_FLAGS=8 timeout 300 python -m pytest tests/test_gpu_decode_step.py -x -q -m gpu -k "long_window or bit_identical" 2>&1 | tail -5 >> gpurun_out/r05c8/tests_step.log
cat gpurun_out/r05c8/tests_step.log | tail -12
OASR_XCD_FLAGS=256 timeout 200 python scripts/decode_xcd_probe.py medium 1 32 2 2>&1 | grep -v "^$\|amdgpu.ids" >> gpurun_out/r05c8/stamps.log
OASR_XCD_FLAGS=256 timeout 200 python scripts/decode_xcd_probe.py small 1 32 2 2>&1 | grep -v "^$\|amdgpu.ids" >> gpurun_out/r05c8/stamps.log
for v in small medium; do timeout 300 python scripts/decode_xcd_probe.py $v 1 32 1,2,3,4 2>&1 | grep -v "^$\|amdgpu.ids" | tail -4 >> gpurun_out/r05c8/probe.log; done
timeout 200 python scripts/decode_xcd_probe.py medium 1 200 1,2 2>&1 | tail -2 >> gpurun_out/r05c8/probe.log
timeout 200 python scripts/decode_xcd_probe.py small 4 32 1,2 2>&1 | tail -2 >> gpurun_out/r05c8/probe.log
cat gpurun_out/r05c8/stamps.log gpurun_out/r05c8/probe.log
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_decode_parity.py -x -q -m gpu -k "attention or decode or pick or c5" 2>&1 | tail -4 | tee -a gpurun_out/r05c8/tests_step.log
