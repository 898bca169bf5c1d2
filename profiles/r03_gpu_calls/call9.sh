set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c9; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -6 $O/pytest_gpu.log | tee -a $O/summary.txt
grep -E "memorised after|C5 sample" $O/pytest_gpu.log | tee -a $O/summary.txt
