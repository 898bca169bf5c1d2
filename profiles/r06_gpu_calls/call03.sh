#!/bin/bash
# round 6, call 3: OLMoASR-base (K = 512 layers): plain launches vs persistent ping-pong launches vs the 256x128 / 2-workgroup geometry, whole step, one box
O=gpurun_out/r06c
mkdir -p $O
export OASR_TESTING_HOOKS=1
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --variant base --steps 10 --warmup 2 --ab-steps 0 --no-cpu-baseline 2>>$O/err.log | tail -1 > $O/$name.json
  python - <<PY >> $O/ab.txt
import json
j=json.loads(open("$O/$name.json").read())
r=j["roofline"]
top=sorted(r["by_symbol"].items(), key=lambda kv:-kv[1]["launches"]*kv[1]["avg_us"])[:6]
print("$name", "ms/step", j["ms_per_step"], j["per_step_ms"], "executed frac", j["step_frac_executed"], "gemm_ms", r["gemm_ms_per_step"], " | ".join("%s %dx%.0fus" % (k.split("kernel")[1][:40], v["launches"], v["avg_us"]) for k, v in top))
PY
}
for rep in 1 2; do
run default_$rep X=1
run persistent_$rep OASR_PP_PERSISTENT=1
run geom128_$rep OASR_GEMM_GEOM=1
done
cat $O/ab.txt | cut -c1-420
