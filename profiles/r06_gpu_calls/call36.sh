#!/bin/bash
# round 6, call 36: packets, the kernel instantiated per rows-per-packet (2, 4, 6) with per-phase unit-loop bounds (code 38-59 KB)
O=gpurun_out/r06w17
mkdir -p $O
export OASR_TESTING_HOOKS=1
for v in medium small large base tiny; do timeout 300 python scripts/decode_xcd_probe.py $v 1 32 1,-1 2>&1 | grep -v amdgpu | cut -c1-250 >> $O/probe.txt; done
timeout 300 python scripts/decode_xcd_probe.py medium 1 300 1,-1 2>&1 | grep -v amdgpu | cut -c1-250 >> $O/probe.txt
OASR_XCD_FLAGS=256 timeout 300 python scripts/decode_xcd_probe.py medium 1 32 5 2>&1 | grep -v amdgpu | cut -c1-250 >> $O/probe.txt
cat $O/probe.txt
timeout 900 python -m pytest tests/test_gpu_decode_step.py tests/test_gpu_decode_parity.py -m gpu -x -q --timeout 800 2>&1 | tail -6
