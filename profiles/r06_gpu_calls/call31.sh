#!/bin/bash
# round 6, call 31: the FINAL binary (chip-wide decoder step as the B = 1 default, ABI 211): GPU suite, smoke(), the default bench line, the driver-length
# line, every GEMM launch of one step by shape
O=gpurun_out/r06z2
mkdir -p $O
python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -8 > $O/suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py > $O/r06_bench_default.json 2> $O/err.log
python bench.py --steps 20 --warmup 3 > $O/r06_bench_steps20.json 2>> $O/err.log
OASR_TESTING_HOOKS=1 OASR_PROF_SHAPES=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --ab-steps 0 2>>$O/err.log | tail -1 > $O/shapes.json
tail -3 $O/suite.log; tail -2 $O/smoke.log | cut -c1-200
python - <<PY
import json
for f in ("r06_bench_default", "r06_bench_steps20"):
    j=json.loads(open(f"$O/{f}.json").read().strip().splitlines()[-1])
    r=j["roofline"]
    print(f, j["ms_per_step"], j["value"], j["per_step_ms"], j["plain_step_ms"], j["span_bwd_ms"], j["step_frac_algorithmic"], j["step_frac_executed"], r["frac"], r["main_stream_all"]["frac"], r["traffic"], j["config"]["side_streams"], {k[:12]: v.get("frac_of_8TBps") for k, v in r["hbm_kernels"].items()})
j=json.loads(open("$O/shapes.json").read())
rows=sorted(j["roofline"]["by_symbol"].items(), key=lambda kv:-kv[1]["launches"]*kv[1]["avg_us"])
with open("$O/r06_gemm_by_shape_in_step.txt","w") as f:
    f.write("# Round 6, call 31: every GEMM launch of ONE training step (default bench: medium, 2 x 128 clips, span-forward step, side-stream mode 7) timed with HIP events on its stream and\n# summed per (kernel symbol, lane, shape, epilogue) -- OASR_PROF_SHAPES=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --ab-steps 0.  [side] = lowest-priority side stream (queueing spans),\n# [shared] = main stream while side-stream filler is in flight.  epi bits: 1 bias, 2 residual, 4 GELU, 8 GELU' side input, 16 fused column sums, 32 atomic split-K output.\n# ms_per_step %s\n" % j["ms_per_step"])
    for k,v in rows:
        f.write("%9.2f ms  n=%4d avg %9.1f us %8.1f TF/s  %s\n" % (v["launches"]*v["avg_us"]/1e3, v["launches"], v["avg_us"], v["tflops"], k))
print(open("$O/r06_gemm_by_shape_in_step.txt").read()[:3000])
PY
