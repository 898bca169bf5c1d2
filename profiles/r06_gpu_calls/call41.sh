#!/bin/bash
# round 6, call 41: the FINAL binary (chip-wide decoder step with packets and the logits phase as the B = 1 default, ABI 211): GPU suite, smoke(), the default bench line,
# the driver-length line, every GEMM launch of one step by shape, the decoder-step probe at the default, its rocprofv3 stats and the C5 lines
O=gpurun_out/r06z6
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -8 > $O/suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py > $O/r06_bench_default.json 2> $O/err.log
python bench.py --steps 20 --warmup 3 > $O/r06_bench_steps20.json 2>> $O/err.log
OASR_TESTING_HOOKS=1 OASR_PROF_SHAPES=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --ab-steps 0 2>>$O/err.log | tail -1 > $O/shapes.json
for v in tiny base small medium large; do OASR_TESTING_HOOKS=1 python scripts/decode_xcd_probe.py $v 1 32 -1,1,2 2>&1 | grep -v amdgpu | cut -c1-230 >> $O/decode_probe.txt; done
OASR_TESTING_HOOKS=1 python scripts/decode_xcd_probe.py medium 1 300 -1,2 2>&1 | grep -v amdgpu | cut -c1-230 >> $O/decode_probe.txt
OASR_TESTING_HOOKS=1 OASR_XCD_FLAGS=256 python scripts/decode_xcd_probe.py medium 1 32 5 2>&1 | grep -v amdgpu | cut -c1-230 > $O/decode_stamps.txt
OASR_TESTING_HOOKS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python scripts/decode_xcd_probe.py medium 1 32 -1 > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/rocprof_summary.py "$f" > $O/r06_decode_step_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  OASR_TESTING_HOOKS=1 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python scripts/decode_xcd_probe.py medium 1 32 -1 > $O/pmc_$c.log 2>&1
done
ff=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
fw=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python scripts/pmc_traffic.py "$ff" "$fw" $O/r06_decode_step_hbm_traffic.json > $O/r06_decode_step_hbm_traffic.txt
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
python scripts/transcribe_bench.py 2>&1 | tail -1 > $O/c5_random.log
python scripts/transcribe_trained_bench.py 20 small 2>&1 | tail -1 > $O/c5_trained.log
python scripts/transcribe_trained_bench.py 20 small 2>&1 | tail -1 >> $O/c5_trained.log
tail -3 $O/suite.log; tail -2 $O/smoke.log | cut -c1-200; cat $O/decode_probe.txt | grep "mode -1"; head -6 $O/r06_decode_step_kernel_stats.txt | cut -c1-180; head -3 $O/r06_decode_step_hbm_traffic.txt; cat $O/c5_trained.log | cut -c1-60,300-520
python - <<PY
import json
for f in ("r06_bench_default", "r06_bench_steps20"):
    j=json.loads(open(f"$O/{f}.json").read().strip().splitlines()[-1])
    r=j["roofline"]
    print(f, j["ms_per_step"], j["value"], j["per_step_ms"], j["plain_step_ms"], j["span_bwd_ms"], j["step_frac_algorithmic"], j["step_frac_executed"], r["frac"], r["main_stream_all"]["frac"], r["traffic"], j["config"]["side_streams"], {k[:12]: v.get("frac_of_8TBps") for k, v in r["hbm_kernels"].items()})
j=json.loads(open("$O/shapes.json").read())
rows=sorted(j["roofline"]["by_symbol"].items(), key=lambda kv:-kv[1]["launches"]*kv[1]["avg_us"])
with open("$O/r06_gemm_by_shape_in_step.txt","w") as f:
    f.write("# Round 6, call 41: every GEMM launch of ONE training step (default bench: medium, 2 x 128 clips, span-forward step, side-stream mode 7) timed with HIP events on its stream and\n# summed per (kernel symbol, lane, shape, epilogue) -- OASR_PROF_SHAPES=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --ab-steps 0.  [side] = lowest-priority side stream (queueing spans),\n# [shared] = main stream while side-stream filler is in flight.  epi bits: 1 bias, 2 residual, 4 GELU, 8 GELU' side input, 16 fused column sums, 32 atomic split-K output.\n# ms_per_step %s\n" % j["ms_per_step"])
    for k,v in rows:
        f.write("%9.2f ms  n=%4d avg %9.1f us %8.1f TF/s  %s\n" % (v["launches"]*v["avg_us"]/1e3, v["launches"], v["avg_us"], v["tflops"], k))
PY
