#!/bin/bash
# round 5, call 11: evidence of the round's binary -- the GPU suite, smoke(), rocprof / PMC passes of the default bench command, the default bench line,
# the C5 long-form line on a trained-like model, the decode-step probe
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -8 > gpurun_out/r05/call11_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/smoke.log 2>&1
bash scripts/profile_round.sh r05 > gpurun_out/r05_profile.log 2>&1
python bench.py > gpurun_out/r05/r05_bench_default.json 2> gpurun_out/r05/r05_bench_default.err
python scripts/transcribe_trained_bench.py 20 small > gpurun_out/r05/c5_trained.log 2>&1
OASR_TESTING_HOOKS=1 python scripts/decode_xcd_probe.py medium 1 32 1,2 2>&1 | grep -v "^$\|amdgpu.ids" > gpurun_out/r05/decode_probe.log
OASR_TESTING_HOOKS=1 python scripts/decode_xcd_probe.py small 1 32 1,2 2>&1 | grep -v "^$\|amdgpu.ids" >> gpurun_out/r05/decode_probe.log
python scripts/hostinfo.py > gpurun_out/r05/r05_hostinfo.txt 2>&1
tail -4 gpurun_out/r05/call11_suite.log; tail -3 gpurun_out/r05/smoke.log; tail -2 gpurun_out/r05/c5_trained.log | cut -c1-600; cut -c1-140 gpurun_out/r05/decode_probe.log
python - <<PY
import json
j=json.loads(open("gpurun_out/r05/r05_bench_default.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"], j["per_step_ms"], j["plain_step_ms"], j["span_bwd_ms"], j["step_frac_algorithmic"], j["step_frac_executed"], j["roofline"]["frac"], j["roofline"]["avg_launch_us"], j["roofline"]["traffic"], j["cpu_baseline"]["value"], j["parity"], j["roofline"]["hbm_kernels"]["decode_step(B=1)"]["us"])
PY
