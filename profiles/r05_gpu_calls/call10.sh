#!/bin/bash
# round 5, call 10: the two GELU-epilogue GEMM shapes (mlp.0 forward epi=5, dgrad through GELU' epi=24) launched persistent (next tile's prologue issued
# ahead of the epilogue), whole-step same-box A/B with per-shape HIP-event times: plain launches / GELU shapes persistent / every pp launch persistent
set -x
mkdir -p gpurun_out/r05c10
run() { env "$@" OASR_PROF_SHAPES=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --ab-steps 0 2>/dev/null | tail -1 > gpurun_out/r05c10/$TAG.json; }
TAG=plain run X=1
TAG=persist_gelu run OASR_PP_PERSIST_GELU=1
TAG=persist_all run OASR_PP_PERSISTENT=1
TAG=plain2 run X=1
python - <<'PY' | tee gpurun_out/r05c10/summary.txt
import json
for tag in ("plain", "persist_gelu", "persist_all", "plain2"):
    j = json.loads(open(f"gpurun_out/r05c10/{tag}.json").read())
    sym = j["roofline"]["by_symbol"]
    rows = [(k, v) for k, v in sym.items() if ("N=4096 K=1024" in k and ("epi=5" in k or "epi=24" in k or "epi=8" in k)) or "M=192000 N=3072 K=1024 epi=1" in k or "M=192000 N=1024 K=4096 epi=3" in k]
    print(f"{tag:14s} ms_per_step {j['ms_per_step']:8.2f}  per_step {j['per_step_ms']}  gemm_ms {j['roofline']['gemm_ms_per_step']}")
    for k, v in sorted(rows, key=lambda kv: -kv[1]['launches'] * kv[1]['avg_us'])[:6]:
        print(f"      {v['tflops']:7.1f} TF/s  {v['avg_us']:8.1f} us x {v['launches']:3d}  {k}")
PY
