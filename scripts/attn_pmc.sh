export TMPDIR=/tmp
mkdir -p gpurun_out/attn_pmc
rocprofv3 -L > gpurun_out/attn_pmc/counters_list.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS_F32" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_IFETCH" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_VALU_FMA_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/attn_pmc/p$i -o pmc -- python scripts/attn_bench.py 2 > gpurun_out/attn_pmc/p$i.log 2>&1
  python scripts/pmc_summary.py gpurun_out/attn_pmc/p$i > gpurun_out/attn_pmc/summary$i.txt 2>&1
  rm -rf gpurun_out/attn_pmc/p$i
done
cat gpurun_out/attn_pmc/summary*.txt | grep attn_ | cut -c1-400
