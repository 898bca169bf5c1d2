cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/pipe1; mkdir -p $O
rm -f /tmp/eq.pt
OASR_ATTN_PIPE=0 python scratch/attn_eq.py /tmp/eq.pt 2>&1 | tail -2 | tee $O/eq.txt
timeout 120 python scratch/attn_eq.py /tmp/eq.pt 2>&1 | grep -v "cs\|cv" | tail -30 | tee -a $O/eq.txt
timeout 300 python -m pytest tests/test_gpu_ops.py -q -k "attention" 2>&1 | tail -5 | tee $O/pytest.txt
for p in 0 2; do echo "PIPE=$p" | tee -a $O/bench.txt; OASR_ATTN_PIPE=$p timeout 120 python scripts/attn_bench.py 10 2>&1 | grep -v amdgpu.ids | tee -a $O/bench.txt; done
WHICH=enc bash scratch/run_pipe2.sh
