# round 3, GPU call 3: FFT log-mel (tests + us/clip vs the MFMA DFT), quad GEMM kernel (tests + layer-shape A/B + whole step), selective duo rule
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c3; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -k "mel or gemm" > $O/pytest_ops.log 2>&1; echo "pytest ops rc=$?" | tee -a $O/summary.txt; tail -4 $O/pytest_ops.log | tee -a $O/summary.txt
python -m pytest tests/test_gpu_data.py tests/test_gpu_properties.py -x -q > $O/pytest_data.log 2>&1; echo "pytest data rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_data.log | tee -a $O/summary.txt
python scripts/mel_bench.py 2>&1 | tail -1 | tee -a $O/summary.txt
OASR_LOGMEL=mfma python scripts/mel_bench.py 2>&1 | tail -1 | tee -a $O/summary.txt
python scripts/gemm_ab.py 5 4 7 > $O/gemm_ab_pp_quad.log 2>&1; cat $O/gemm_ab_pp_quad.log | tee -a $O/summary.txt
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$name.json
  python - "$name" $O/bench_$name.json <<'PY' | tee -a $O/summary.txt
import json,sys
try:
    d=json.load(open(sys.argv[2])); r=d['roofline']
    print('RUN', sys.argv[1], 'ms/step', d['ms_per_step'], 'gemm_ms', r.get('gemm_ms_per_step'), ' | '.join(f"{k.split('_kernel')[0][-4:]}<{k.split('<')[1][:22]}:{v['avg_us']:.0f}us x{v['launches']}" for k,v in r['by_symbol'].items() if 'gemm' in k and v['launches']>20))
    print('   hbm_kernels', json.dumps(r.get('hbm_kernels'))[:600])
except Exception as e:
    print('RUN', sys.argv[1], 'FAILED', e)
PY
}
run base OASR_LANE=0
run quad OASR_GEMM_QUAD=1
run duoNK OASR_GEMM_DUO_K=1024 OASR_GEMM_DUO_N=1024
run base2 OASR_LANE=0
run quad2 OASR_GEMM_QUAD=1
