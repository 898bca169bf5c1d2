#!/bin/bash
# Profiles of the default bench command for profiles/ (run on the GPU box through gpurun; everything lands in gpurun_out/).
#   1. rocprofv3 --kernel-trace --stats          -> per-kernel calls / avg duration (scripts/rocprof_summary.py)
#   2. rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes: the two do not fit the TCC slots together)
#                                                -> HBM bytes per launch per kernel (scripts/pmc_traffic.py)
#   3. rocprofv3 --pmc SQ_* (scripts/pmc_sq.txt) -> MFMA-busy / SQ-busy per kernel (scripts/pmc_summary.py)
#   4. rocprofv3 --pmc GRBM_GUI_ACTIVE         -> effective shader clock per kernel under the step's load (scripts/pmc_clock.py)
# Counter passes never carry --kernel-trace/--stats-unrelated trace domains (gpurun refuses pmc + sys/hip/hsa tracing).
set -u
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --ab-steps 0"
BENCH1="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --ab-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/rocprof_summary.py "$f" > $OUT/${TAG}_bench_default_kernel_stats.txt
[ -n "$f" ] && python scripts/rocprof_summary.py "$f" --by-queue > $OUT/${TAG}_bench_default_kernel_stats_by_queue.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- $BENCH1 > $OUT/pmc_$c.log 2>&1
done
ff=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
fw=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python scripts/pmc_traffic.py "$ff" "$fw" $OUT/${TAG}_hbm_traffic.json > $OUT/${TAG}_hbm_traffic.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $OUT/pmc_sq -o pmc -- $BENCH1 > $OUT/pmc_sq.log 2>&1
python scripts/pmc_summary.py $OUT/pmc_sq > $OUT/${TAG}_sq_counters.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_clk -o pmc -- $BENCH1 > $OUT/pmc_clk.log 2>&1
python scripts/pmc_clock.py $OUT/pmc_clk $OUT/${TAG}_effective_clock.json > $OUT/${TAG}_effective_clock.txt 2>&1
# raw csv files are large: keep the summaries only
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq $OUT/pmc_clk
ls -la $OUT
