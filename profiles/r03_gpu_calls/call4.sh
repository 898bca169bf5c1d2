# round 3, GPU call 4: FFT log-mel with per-stage LDS tables, quad GEMM with the deeper operand pipeline
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c4; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -k "mel or quad" > $O/pytest_ops.log 2>&1; echo "pytest ops rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_ops.log | tee -a $O/summary.txt
python scripts/mel_bench.py 2>&1 | tail -1 | tee -a $O/summary.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/mel_prof -- python $GRAFT_REPO_ROOT/scripts/mel_bench.py > /dev/null 2>&1 )
find $O/mel_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -8 {}' | cut -c1-200 | tee -a $O/summary.txt
python scripts/gemm_ab.py 5 4 7 > $O/gemm_ab_pp_quad.log 2>&1; cat $O/gemm_ab_pp_quad.log | tee -a $O/summary.txt
