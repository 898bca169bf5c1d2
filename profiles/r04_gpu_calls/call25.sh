#!/bin/bash
# round 4, GPU call 25: attention forward staging variants: register-staged (nodma), DMA ring (liboasr.so), DMA ring + mid-iteration barrier + K-fragment prefetch (dma2)
mkdir -p gpurun_out/r04
for lib in liboasr_nodma.so liboasr.so liboasr_dma2.so; do OASR_LIB=$PWD/olmoasr_amd/$lib timeout 120 python scripts/attn_fwd_crc.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$lib /"; done | tee gpurun_out/r04/call25_crc.txt
for i in 1 2 3; do
  for lib in liboasr_nodma.so liboasr.so liboasr_dma2.so; do
    OASR_LIB=$PWD/olmoasr_amd/$lib python scripts/attn_bench.py 20 2>&1 | grep -E "encoder self|cross|decoder" | sed "s/^/$lib /"
  done
done | tee gpurun_out/r04/call25_attn_dma.txt
