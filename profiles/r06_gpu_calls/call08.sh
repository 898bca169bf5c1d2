#!/bin/bash
# round 6, call 8: SQ counters of the quad log-mel kernel (what fills the cycles that are not VALU issue?) + its effective clock
O=gpurun_out/r06h
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' > $O/sq_counter_names.txt
CMD="python scripts/mel_bench.py"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/pmc1 -o pmc -- $CMD > $O/pmc1.log 2>&1
python scripts/pmc_summary.py $O/pmc1 > $O/sq1.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc2 -o pmc -- $CMD > $O/pmc2.log 2>&1
python scripts/pmc_summary.py $O/pmc2 > $O/sq2.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc3 -o pmc -- $CMD > $O/pmc3.log 2>&1
python scripts/pmc_summary.py $O/pmc3 > $O/clk.txt 2>&1
f=$(find $O/pmc3 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/rocprof_summary.py "$f" > $O/kstats.txt
rm -rf $O/pmc1 $O/pmc2 $O/pmc3
grep -i "logmel" $O/sq1.txt $O/sq2.txt $O/clk.txt $O/kstats.txt | cut -c1-400
tail -3 $O/pmc1.log
