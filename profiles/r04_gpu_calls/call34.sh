#!/bin/bash
# round 4, GPU call 34: dQ ping-pong kernel of the span step: waves past the span skip their compute segments; tests, cross probe, whole-step A/B (prev = HEAD's library)
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_span.py tests/test_gpu_bench_shapes.py -q -x 2>&1 | tail -3 | tee gpurun_out/r04/call34_tests.txt
for lib in liboasr_prev.so liboasr.so liboasr_prev.so liboasr.so; do
  OASR_LIB=$PWD/olmoasr_amd/$lib timeout 120 python scripts/cross_bwd_probe.py 2>&1 | grep "per sample" | head -1 | sed "s/^/$lib /"
done | tee gpurun_out/r04/call34_cross.txt
for lib in liboasr_prev.so liboasr.so liboasr_prev.so liboasr.so; do
  OASR_LIB=$PWD/olmoasr_amd/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$lib', j['ms_per_step'])"
done | tee gpurun_out/r04/call34_step.txt
