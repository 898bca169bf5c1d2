# round 3, GPU call 1: the full -m gpu suite on the round's first code + the duo GEMM kernel A/B (layer shapes, whole step)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c1; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -5 $O/pytest_gpu.log | tee -a $O/summary.txt
python scripts/gemm_ab.py 5 4 6 2 > $O/gemm_ab_pp_duo_fast.log 2>&1; cat $O/gemm_ab_pp_duo_fast.log | tee -a $O/summary.txt
for k in 0 1024 8192 0 1024; do
  if [ "$k" = "0" ]; then unset OASR_GEMM_DUO_K; else export OASR_GEMM_DUO_K=$k; fi
  python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_duo_k$k.json
  python - "$k" $O/bench_duo_k$k.json <<'PY' | tee -a $O/summary.txt
import json,sys
d=json.load(open(sys.argv[2])); r=d['roofline']
print('DUO_K', sys.argv[1], 'ms/step', d['ms_per_step'], 'gemm_ms', r.get('gemm_ms_per_step'), ' | '.join(f"{k.split('<')[0][-8:]}<{k.split('<')[1][:26]}:{v['avg_us']:.0f}us x{v['launches']}" for k,v in r['by_symbol'].items() if 'gemm' in k))
PY
done
