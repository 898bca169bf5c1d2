#!/bin/bash
# round 4, GPU call 24: attention forward with DMA staging (3-stage ring, prefetch distance 2) vs register staging: bit identity, tests, A/B
mkdir -p gpurun_out/r04
for lib in liboasr_nodma.so liboasr.so; do OASR_LIB=$PWD/olmoasr_amd/$lib timeout 120 python scripts/attn_fwd_crc.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$lib /"; done | tee gpurun_out/r04/call24_crc.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_span.py tests/test_gpu_bench_shapes.py tests/test_gpu_modules.py -q -x -k "attn or attention or span" 2>&1 | tail -4 | tee gpurun_out/r04/call24_tests.txt
for i in 1 2 3; do
  for lib in liboasr_nodma.so liboasr.so; do
    OASR_LIB=$PWD/olmoasr_amd/$lib python scripts/attn_bench.py 20 2>&1 | grep -E "encoder self|cross|decoder" | sed "s/^/$lib /"
  done
done | tee gpurun_out/r04/call24_attn_dma.txt
