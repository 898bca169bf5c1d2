export TMPDIR=/tmp
mkdir -p gpurun_out/r02d
rm -rf /tmp/clk
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/clk -o clk -- python scripts/gemm_clock.py > gpurun_out/r02d/gemm_clock.log 2>&1
ls -R /tmp/clk | head -20 >> gpurun_out/r02d/gemm_clock.log
python scripts/gemm_clock.py summarise /tmp/clk 2>&1 | tee gpurun_out/r02d/gemm_clock.txt
head -3 $(find /tmp/clk -name "*counter_collection.csv" | head -1) >> gpurun_out/r02d/gemm_clock.log
