#!/bin/bash
# round 4, GPU call 2: full GPU suite on the span-step code + bench A/B incl. --span-forward
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_span.py tests/test_gpu_bench_shapes.py -q --timeout 900 -s > gpurun_out/r04/call2_new_tests_full.log 2>&1
grep -E "rel-L2|passed|failed|FAILED|Error" gpurun_out/r04/call2_new_tests_full.log > gpurun_out/r04/call2_new_tests.log
python -m pytest tests -m gpu -q --timeout 1200 --deselect tests/test_gpu_span.py --deselect tests/test_gpu_bench_shapes.py 2>&1 | tail -25 > gpurun_out/r04/call2_suite.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/r04/call2_bench_span.json 2> gpurun_out/r04/call2_bench_span.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --span-forward > gpurun_out/r04/call2_bench_spanfwd.json 2> gpurun_out/r04/call2_bench_spanfwd.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --full-backward > gpurun_out/r04/call2_bench_full.json 2> gpurun_out/r04/call2_bench_full.err
cat gpurun_out/r04/call2_new_tests.log
tail -6 gpurun_out/r04/call2_suite.log
for f in span spanfwd full; do python - <<PY
import json
try:
    j=json.load(open("gpurun_out/r04/call2_bench_$f.json")); print("$f", j["ms_per_step"], j["value"], j["step_frac_of_mfma_peak"], j["final_loss"])
except Exception as e:
    print("$f", "ERR", e)
PY
done
