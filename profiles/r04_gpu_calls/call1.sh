#!/bin/bash
# round 4, GPU call 1: the supervised-span step -- new tests, bench A/B (span vs full backward)
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_span.py tests/test_gpu_bench_shapes.py -q --timeout 900 -s 2>&1 | tail -80 > gpurun_out/r04/call1_tests.log
python -m pytest tests/test_gpu_ops.py -q --timeout 900 -k "attention" 2>&1 | tail -15 > gpurun_out/r04/call1_attn_tests.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04/call1_bench_span.json 2> gpurun_out/r04/call1_bench_span.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --full-backward > gpurun_out/r04/call1_bench_full.json 2> gpurun_out/r04/call1_bench_full.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/r04/call1_bench_span2.json 2> gpurun_out/r04/call1_bench_span2.err
tail -5 gpurun_out/r04/call1_tests.log
cat gpurun_out/r04/call1_bench_span2.json gpurun_out/r04/call1_bench_full.json | cut -c1-400
