#!/bin/bash
# round 4, GPU call 18: attention forward with the reference maximum subtracted inside the matrix core (Q pre-scaled, fifth K-step) vs the fma form
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_span.py tests/test_gpu_bench_shapes.py tests/test_gpu_modules.py -q -x -k "attn or attention or span" 2>&1 | tail -8 | tee gpurun_out/r04/call18_tests.txt
for i in 1 2 3; do
  for lib in liboasr_noseed.so liboasr.so; do
    OASR_LIB=$PWD/olmoasr_amd/$lib python scripts/attn_bench.py 20 2>&1 | grep -E "encoder self|cross|decoder" | sed "s/^/$lib /"
  done
done | tee gpurun_out/r04/call18_attn_seed.txt
