#!/bin/bash
# round 5, call 3: where the one-launch decoder step's ~16 us per phase go: in-kernel stamps, and the experiment flags
# (1 = plain activation stores, 4 = no s_sleep in the barrier poll)
set -x
mkdir -p gpurun_out/r05c3
for f in 256 257 260 261; do
  echo "=== OASR_XCD_FLAGS=$f" >> gpurun_out/r05c3/stamps.log
  OASR_XCD_FLAGS=$f timeout 200 python scripts/decode_xcd_probe.py medium 1 32 2 2>&1 | grep -v "^$\|amdgpu.ids" >> gpurun_out/r05c3/stamps.log
done
OASR_XCD_FLAGS=256 timeout 200 python scripts/decode_xcd_probe.py small 1 32 2 2>&1 | grep -v "^$\|amdgpu.ids" >> gpurun_out/r05c3/stamps.log
cat gpurun_out/r05c3/stamps.log
