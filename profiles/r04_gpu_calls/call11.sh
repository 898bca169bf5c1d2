#!/bin/bash
# round 4, GPU call 11: per-SHAPE GEMM times inside the step (HIP events per launch, OASR_PROF_SHAPES=1) next to the same shapes standalone
mkdir -p gpurun_out/r04
OASR_PROF_SHAPES=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04/call11_bench_shapes.json
python - <<PY | tee gpurun_out/r04/call11_gemm_by_shape.txt
import json
j=json.load(open("gpurun_out/r04/call11_bench_shapes.json"))
print("# ms_per_step", j["ms_per_step"])
rows=[]
for k,v in j["roofline"]["by_symbol"].items():
    rows.append((v["launches"]*v["avg_us"], k, v))
tot=sum(r[0] for r in rows)
for t,k,v in sorted(rows, reverse=True)[:60]:
    print(f"{t/1000:8.2f} ms {100*t/tot:5.1f}%  n={v['launches']:4d} avg {v['avg_us']:8.1f} us {v['tflops']:7.1f} TF/s  {k}")
PY
MS=192000 python scripts/gemm_ab.py 3 4 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/call11_gemm_by_shape.txt
