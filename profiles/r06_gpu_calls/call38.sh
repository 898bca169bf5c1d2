#!/bin/bash
# round 6, call 38: packets also for the new position q | k | v row (projection -> self-attention) and the self-attention output (-> output projection)
O=gpurun_out/r06w18
mkdir -p $O
export OASR_TESTING_HOOKS=1
for v in medium small large base tiny; do timeout 300 python scripts/decode_xcd_probe.py $v 1 32 1,-1 2>&1 | grep -v amdgpu | cut -c1-250 >> $O/probe.txt; done
timeout 300 python scripts/decode_xcd_probe.py medium 1 300 1,-1 2>&1 | grep -v amdgpu | cut -c1-250 >> $O/probe.txt
OASR_XCD_FLAGS=256 timeout 300 python scripts/decode_xcd_probe.py medium 1 32 5 2>&1 | grep -v amdgpu | cut -c1-250 >> $O/probe.txt
cat $O/probe.txt
timeout 900 python -m pytest tests/test_gpu_decode_step.py tests/test_gpu_decode_parity.py -m gpu -x -q --timeout 800 2>&1 | tail -6
