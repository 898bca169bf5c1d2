set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c8; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -4 $O/pytest_gpu.log | tee -a $O/summary.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json | tee -a $O/summary.txt
