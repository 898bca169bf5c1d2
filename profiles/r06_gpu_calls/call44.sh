#!/bin/bash
# round 6, call 44: SQ counters of the decoder step at the library default (rocprofv3 --pmc only: no trace domains)
O=gpurun_out/r06dsq
mkdir -p $O
export TMPDIR=/tmp OASR_TESTING_HOOKS=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc_sq -o pmc -- python scripts/decode_xcd_probe.py medium 1 32 -1 > $O/pmc_sq.log 2>&1
python scripts/pmc_summary.py $O/pmc_sq > $O/r06_decode_step_sq_counters.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o pmc -- python scripts/decode_xcd_probe.py medium 1 32 -1 > $O/pmc_sq2.log 2>&1
python scripts/pmc_summary.py $O/pmc_sq2 > $O/r06_decode_step_sq_counters2.txt 2>&1
rm -rf $O/pmc_sq $O/pmc_sq2
head -20 $O/r06_decode_step_sq_counters.txt | cut -c1-220; head -12 $O/r06_decode_step_sq_counters2.txt | cut -c1-220
