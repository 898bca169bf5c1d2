#!/bin/bash
# round 5, call 18: world-1 RCCL rehearsal, each reducer alone: is the late first bucket of call 17's "allreduce" an artefact of --reducer both?
mkdir -p gpurun_out/r05h
for r in allreduce direct; do
  OASR_BENCH_FORCE_DDP=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --ab-steps 0 --reducer $r 2>/dev/null | tail -1 > gpurun_out/r05h/$r.json
done
python - <<PY
import json
for r in ("allreduce","direct"):
    j=json.loads(open(f"gpurun_out/r05h/{r}.json").read())
    print(r, j["ms_per_step"], json.dumps(j["ddp"]))
PY
