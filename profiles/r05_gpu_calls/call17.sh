#!/bin/bash
# round 5, call 17: the N > 1 code path of bench.py rehearsed with a world of one over RCCL (OASR_BENCH_FORCE_DDP=1): the `ddp` block with its HIP-event
# timings and --reducer both
mkdir -p gpurun_out/r05g
OASR_BENCH_FORCE_DDP=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --reducer both > gpurun_out/r05g/force_ddp.json 2> gpurun_out/r05g/err.log
tail -3 gpurun_out/r05g/err.log
python - <<PY
import json
j=json.loads(open("gpurun_out/r05g/force_ddp.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["per_step_ms"], j["plain_step_ms"], j["span_bwd_ms"], json.dumps(j["ddp"]), j["config"]["micro_batch"], j["config"]["free_hbm_gib"])
PY
