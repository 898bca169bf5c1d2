set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c7; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_data.py -x -q -k "mel or shard or wave" > $O/pytest_mel.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -2 $O/pytest_mel.log | tee -a $O/summary.txt
for a in 0 8 32; do OASR_LOGMEL_ABL=$a python scripts/mel_bench.py 2>&1 | tail -1 | sed "s/^/ABL $a: /" | tee -a $O/summary.txt; done
