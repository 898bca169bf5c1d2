#!/bin/bash
# round 4, GPU call 14: attention forward at 4 waves per SIMD (launch_bounds(256, 4): 128 VGPRs, 20 spilled) vs the shipped 3
mkdir -p gpurun_out/r04
for i in 1 2 3; do
  for lib in liboasr.so liboasr_occ4.so; do
    OASR_LIB=$PWD/olmoasr_amd/$lib python scripts/attn_bench.py 20 2>&1 | grep -E "encoder self|cross" | sed "s/^/$lib /"
  done
done | tee gpurun_out/r04/call14_attn_occ4.txt
