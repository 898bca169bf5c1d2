export TMPDIR=/tmp
mkdir -p gpurun_out/attn_lds
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/attn_lds/p -o pmc -- python scripts/attn_bench.py 2 > gpurun_out/attn_lds/p.log 2>&1
python scripts/pmc_summary.py gpurun_out/attn_lds/p | grep attn_ | cut -c1-60,95-300
rm -rf gpurun_out/attn_lds/p
