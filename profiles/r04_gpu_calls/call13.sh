#!/bin/bash
# round 4, GPU call 13: what bounds the attention forward?  Timing-only ablations of the exponentials (results are garbage)
mkdir -p gpurun_out/r04
for i in 1 2; do
  for lib in liboasr.so liboasr_qscale.so liboasr_abl1.so liboasr_abl2.so; do
    OASR_LIB=$PWD/olmoasr_amd/$lib python scripts/attn_bench.py 20 2>&1 | grep "encoder self" | sed "s/^/$lib /"
  done
done | tee gpurun_out/r04/call13_attn_exp_ablation.txt
