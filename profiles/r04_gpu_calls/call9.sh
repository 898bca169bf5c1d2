#!/bin/bash
# round 4, GPU call 9: final evidence -- the GPU suite, rocprof / PMC passes and the default bench line of the SAME binary
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -8 > gpurun_out/r04/call9_suite.log
bash scripts/profile_round.sh r04 > gpurun_out/r04_profile.log 2>&1
python bench.py > gpurun_out/r04/r04_bench_default.json 2> gpurun_out/r04/r04_bench_default.err
tail -4 gpurun_out/r04/call9_suite.log
python - <<PY
import json
j=json.loads(open("gpurun_out/r04/r04_bench_default.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"], j["step_frac_of_mfma_peak"], j["roofline"]["frac"], j["roofline"]["avg_launch_us"], j["cpu_baseline"]["value"], j["parity"])
PY
