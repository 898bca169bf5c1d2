#!/bin/bash
# round 4, GPU call 15: attention forward with 64 queries per wave (OASR_ATTN_FWD64=1) vs the shipped 32: correctness, then A/B
mkdir -p gpurun_out/r04
OASR_ATTN_FWD64=1 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_span.py tests/test_gpu_bench_shapes.py -q -x -k "attn or attention" 2>&1 | tail -8 | tee gpurun_out/r04/call15_tests.txt
for i in 1 2 3; do
  for v in 0 1; do
    OASR_ATTN_FWD64=$v python scripts/attn_bench.py 20 2>&1 | grep -E "encoder self|cross" | sed "s/^/fwd64=$v /"
  done
done | tee gpurun_out/r04/call15_attn_fwd64.txt
