cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/clk; mkdir -p $O; rm -f $O/summary.txt
for p in 0 1 2; do
  rm -rf /tmp/clk$p
  OASR_LIB=/root/repo/scratch/abl/liboasr_noslp.so OASR_ATTN_PIPE=$p rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/clk$p -o clk -- python scripts/attn_kernel_times.py enc ${BB:-32} ${IT:-30} > $O/$p.log 2>&1
  echo "== PIPE=$p" | tee -a $O/summary.txt
  python scripts/pmc_clock.py /tmp/clk$p | grep attn | tee -a $O/summary.txt
done
