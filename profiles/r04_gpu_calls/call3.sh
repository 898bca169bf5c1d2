#!/bin/bash
# round 4, GPU call 3: native beam / sampling selection, raw log-mel through the span step, trained-like C5 line, bench
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_decode_parity.py tests/test_gpu_span.py tests/test_gpu_model.py tests/test_gpu_data.py -q --timeout 1200 2>&1 | tail -30 > gpurun_out/r04/call3_tests.log
python scripts/transcribe_trained_bench.py 20 small > gpurun_out/r04/call3_c5_trained.json 2> gpurun_out/r04/call3_c5_trained.err
python scripts/transcribe_bench.py small 600 20 > gpurun_out/r04/call3_c5_random_notimestamps.json 2>/dev/null
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04/call3_bench.json 2> gpurun_out/r04/call3_bench.err
tail -12 gpurun_out/r04/call3_tests.log
cat gpurun_out/r04/call3_c5_trained.json; tail -3 gpurun_out/r04/call3_c5_trained.err
cat gpurun_out/r04/call3_c5_random_notimestamps.json
python - <<PY
import json
j=json.load(open("gpurun_out/r04/call3_bench.json")); print(j["ms_per_step"], j["value"], j["step_frac_of_mfma_peak"], j["final_loss"])
for k,v in j["roofline"]["hbm_kernels"].items(): print(k, v["achieved_GBps"], v["frac_of_8TBps"], v["us"])
PY
