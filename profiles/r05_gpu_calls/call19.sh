#!/bin/bash
# round 5, call 19: world-1 RCCL rehearsal after moving the communication stream (and RCCL's own stream) to high priority: does the first bucket now start
# while the backward is still running (comm_lead_ms ~ the length of the last micro-batch's backward instead of ~0)?
mkdir -p gpurun_out/r05i
for r in allreduce direct both; do
  OASR_BENCH_FORCE_DDP=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --ab-steps 0 --reducer $r 2>/dev/null | tail -1 > gpurun_out/r05i/$r.json
done
python -m pytest tests/test_gpu_model.py tests/test_gpu_autograd.py -x -q -m gpu -k "multi_gpu or ddp or DistributedDataParallel or rccl or world" 2>&1 | tail -3
python - <<PY
import json
for r in ("allreduce","direct","both"):
    j=json.loads(open(f"gpurun_out/r05i/{r}.json").read())
    print(r, j["ms_per_step"], json.dumps(j["ddp"]))
PY
