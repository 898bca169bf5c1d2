#!/bin/bash
# round 6, call 10: tail-split launches (OASR_GEMM_TAILSPLIT=1: q full rounds on the ping-pong kernel + the remainder rows on the 256x128 kernel):
# span / bench-shape tests under the switch, then the whole step with and without it, interleaved on one box
O=gpurun_out/r06j
mkdir -p $O
export OASR_TESTING_HOOKS=1
OASR_GEMM_TAILSPLIT=1 python -m pytest tests/test_gpu_span.py tests/test_gpu_bench_shapes.py tests/test_gpu_parity_sizes.py -m gpu -q --timeout 1500 2>&1 | tail -4
for rep in 1 2 3; do
  for ts in 0 1; do
    OASR_GEMM_TAILSPLIT=$ts python bench.py --steps 10 --warmup 2 --ab-steps 0 --no-cpu-baseline 2>>$O/err.log | tail -1 > $O/bench_$ts.json
    python - <<PY | tee -a $O/step_ab.txt
import json
j=json.loads(open("$O/bench_$ts.json").read())
r=j["roofline"]
sh=sum(v["launches"]*v["avg_us"] for k,v in r["by_symbol"].items() if k.endswith("[shared]"))/1e3
print("tailsplit=$ts", "ms/step", j["ms_per_step"], j["per_step_ms"], "dominant frac", r["frac"], "main-stream GEMM ms", r["gemm_ms_per_step"], "[shared] launches ms", round(sh,2), "span parity", j["parity"]["span_step_vs_plain_step"]["grad_rel_l2"])
PY
  done
done
