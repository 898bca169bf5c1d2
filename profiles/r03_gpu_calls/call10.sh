cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c10; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -k "cross_entropy" 2>&1 | tail -2 | tee -a $O/summary.txt
python scripts/ce_bench.py 2>&1 | tail -2 | tee -a $O/summary.txt
