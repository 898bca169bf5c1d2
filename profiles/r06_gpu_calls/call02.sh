#!/bin/bash
# round 6, call 2: (a) new GPU tests of this round's host-side fixes (decode fallback, re-indexed cache), (b) C2 = OLMoASR-base profile (verdict item 7):
# bench line with by-shape GEMM statistics, rocprofv3 kernel stats, SQ counters (MFMA busy) and effective clock per kernel
O=gpurun_out/r06b
mkdir -p $O
python -m pytest tests/test_gpu_decode_step.py -m gpu -q --timeout 1500 2>&1 | tail -5 > $O/decode_tests.log
export TMPDIR=/tmp
OASR_TESTING_HOOKS=1 OASR_PROF_SHAPES=1 python bench.py --variant base --steps 5 --warmup 2 --no-cpu-baseline > $O/base_bench_shapes.json 2> $O/err.log
python bench.py --variant base --steps 10 --warmup 2 --no-cpu-baseline > $O/base_bench.json 2>> $O/err.log
BENCH="python bench.py --variant base --steps 2 --warmup 1 --no-cpu-baseline --no-profile --ab-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- $BENCH > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/rocprof_summary.py "$f" > $O/r06_base_kernel_stats.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $O/pmc_sq -o pmc -- $BENCH > $O/pmc_sq.log 2>&1
python scripts/pmc_summary.py $O/pmc_sq > $O/r06_base_sq_counters.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_clk -o pmc -- $BENCH > $O/pmc_clk.log 2>&1
python scripts/pmc_clock.py $O/pmc_clk $O/r06_base_effective_clock.json > $O/r06_base_effective_clock.txt 2>&1
python scripts/mfma_util.py $O/r06_base_kernel_stats.txt $O/r06_base_sq_counters.txt $O/r06_base_effective_clock.json > $O/r06_base_mfma_utilisation.txt 2>&1
rm -rf $O/trace $O/pmc_sq $O/pmc_clk
cat $O/decode_tests.log
python - <<PY
import json
for f in ("base_bench",):
    j=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
    print(f, j["ms_per_step"], j["value"], j["plain_step_ms"], j["step_frac_algorithmic"], j["step_frac_executed"], j["roofline"]["kernel"], j["roofline"]["frac"])
j=json.loads(open("$O/base_bench_shapes.json").read().strip().splitlines()[-1])
rows=sorted(j["roofline"]["by_symbol"].items(), key=lambda kv:-kv[1]["launches"]*kv[1]["avg_us"])
for k,v in rows[:40]:
    print("%9.2f ms %5d x %8.1f us %7.1f TF/s  %s" % (v["launches"]*v["avg_us"]/1e3, v["launches"], v["avg_us"], v["tflops"], k))
PY
head -30 $O/r06_base_kernel_stats.txt | cut -c1-180
head -24 $O/r06_base_mfma_utilisation.txt | cut -c1-200
