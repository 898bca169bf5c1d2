#!/bin/bash
# round 5, call 21: one-launch decoder step vs the multi-launch step at 2-4 sequences (where should the default switch?)
mkdir -p gpurun_out/r05j
export OASR_TESTING_HOOKS=1
for v in small medium; do for b in 2 3 4; do timeout 200 python scripts/decode_xcd_probe.py $v $b 32 1,2 2>&1 | grep -v "^$\|amdgpu.ids" | tail -2 | cut -c1-110 >> gpurun_out/r05j/probe.log; done; done
cat gpurun_out/r05j/probe.log
