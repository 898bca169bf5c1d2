#!/bin/bash
# round 6, call 21: chip-wide decoder step, finer stamps (self-attention stages, end of phase)
O=gpurun_out/r06w5
mkdir -p $O
export OASR_TESTING_HOOKS=1
OASR_XCD_FLAGS=256 timeout 300 python scripts/decode_xcd_probe.py medium 1 32 5 > $O/probe.txt 2>&1
OASR_XCD_FLAGS=256 timeout 300 python scripts/decode_xcd_probe.py medium 1 300 5 >> $O/probe.txt 2>&1
grep -v amdgpu.ids $O/probe.txt
