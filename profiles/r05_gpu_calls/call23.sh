#!/bin/bash
# round 5, call 23: evidence of the FINAL binary (after the stream-priority and log-mel changes): the GPU suite, smoke(), the default bench line, the driver-length line, C5, decode probe
mkdir -p gpurun_out/r05z
python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -8 > gpurun_out/r05z/suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05z/smoke.log 2>&1
python bench.py > gpurun_out/r05z/r05_bench_default.json 2> gpurun_out/r05z/err.log
python bench.py --steps 20 --warmup 3 > gpurun_out/r05z/r05_bench_steps20.json 2>> gpurun_out/r05z/err.log
python scripts/transcribe_trained_bench.py 20 small > gpurun_out/r05z/c5_trained.log 2>&1
OASR_TESTING_HOOKS=1 python scripts/decode_xcd_probe.py medium 1 32 1,2 2>&1 | grep -v "^$\|amdgpu.ids" > gpurun_out/r05z/decode_probe.log
OASR_TESTING_HOOKS=1 python scripts/decode_xcd_probe.py small 1 32 1,2 2>&1 | grep -v "^$\|amdgpu.ids" >> gpurun_out/r05z/decode_probe.log
tail -3 gpurun_out/r05z/suite.log; tail -2 gpurun_out/r05z/smoke.log; tail -1 gpurun_out/r05z/c5_trained.log | cut -c1-500; cut -c1-140 gpurun_out/r05z/decode_probe.log
python - <<PY
import json
for f in ("r05_bench_default", "r05_bench_steps20"):
    j=json.loads(open(f"gpurun_out/r05z/{f}.json").read().strip().splitlines()[-1])
    print(f, j["ms_per_step"], j["value"], j["per_step_ms"], j["plain_step_ms"], j["span_bwd_ms"], j["step_frac_algorithmic"], j["step_frac_executed"], j["roofline"]["frac"], j["roofline"]["traffic"], j["roofline"]["hbm_kernels"]["decode_step(B=1)"]["us"], j["roofline"]["hbm_kernels"]["decode_step(B=1)"]["frac_of_8TBps"])
PY
