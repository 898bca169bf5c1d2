#!/bin/bash
# round 5, call 9: one-launch decoder step v5 -- is a ring block's ~1.8 k cycles memory wait or instruction issue?
# OASR_XCD_FLAGS: 256 stamps; +16 consume without waiting for the DMA; +32 issue no DMA at all (timing only)
set -x
mkdir -p gpurun_out/r05c9
for f in 256 272 304; do
  echo "=== OASR_XCD_FLAGS=$f" >> gpurun_out/r05c9/stamps.log
  OASR_XCD_FLAGS=$f timeout 200 python scripts/decode_xcd_probe.py medium 1 32 2 2>&1 | grep -v "^$\|amdgpu.ids" >> gpurun_out/r05c9/stamps.log
done
cat gpurun_out/r05c9/stamps.log
