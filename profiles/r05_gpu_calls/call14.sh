#!/bin/bash
# round 5, call 14: one-launch decoder step, version 7 (v6 + merge broadcasts by v_readlane / DPP, self-attention V rows fetched with their K rows)
set -x
mkdir -p gpurun_out/r05c14
timeout 600 python -m pytest tests/test_gpu_decode_step.py tests/test_gpu_decode_parity.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05c14/tests_step.log
OASR_XCD_FLAGS=8 timeout 300 python -m pytest tests/test_gpu_decode_step.py -x -q -m gpu -k "long_window or bit_identical" 2>&1 | tail -3 >> gpurun_out/r05c14/tests_step.log
cat gpurun_out/r05c14/tests_step.log
OASR_XCD_FLAGS=256 timeout 200 python scripts/decode_xcd_probe.py medium 1 32 2 2>&1 | grep -v "^$\|amdgpu.ids" >> gpurun_out/r05c14/stamps.log
for v in small medium; do timeout 300 python scripts/decode_xcd_probe.py $v 1 32 1,2 2>&1 | grep -v "^$\|amdgpu.ids" | tail -2 >> gpurun_out/r05c14/probe.log; done
timeout 200 python scripts/decode_xcd_probe.py medium 1 300 1,2 2>&1 | tail -2 >> gpurun_out/r05c14/probe.log
timeout 200 python scripts/decode_xcd_probe.py medium 2 32 1,2 2>&1 | tail -2 >> gpurun_out/r05c14/probe.log
cat gpurun_out/r05c14/stamps.log gpurun_out/r05c14/probe.log
