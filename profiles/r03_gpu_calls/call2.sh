# round 3, GPU call 2: side-lane correctness + whole-step A/B of lane modes and of the selective duo rule; new decode/parity tests
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c2; mkdir -p $O
python -m pytest tests/test_gpu_model.py -x -q -k "side_lane or multi_gpu_paths or reducer_events" > $O/pytest_lane.log 2>&1; echo "pytest lane rc=$?" | tee -a $O/summary.txt; tail -4 $O/pytest_lane.log | tee -a $O/summary.txt
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$name.json
  python - "$name" $O/bench_$name.json <<'PY' | tee -a $O/summary.txt
import json,sys
try:
    d=json.load(open(sys.argv[2])); r=d['roofline']
    print('RUN', sys.argv[1], 'ms/step', d['ms_per_step'], 'gemm_ms', r.get('gemm_ms_per_step'), 'power', r.get('power_limited',{}).get('package_power_w_avg'), 'sclk', r.get('power_limited',{}).get('sclk_ghz_avg'))
except Exception as e:
    print('RUN', sys.argv[1], 'FAILED', e)
PY
}
run base0 OASR_LANE=0
run lane1 OASR_LANE=1
run lane3 OASR_LANE=3
run lane7 OASR_LANE=7
run lane7lo OASR_LANE=7 OASR_LANE_PRIO=1
run base0b OASR_LANE=0
run lane7b OASR_LANE=7
python -m pytest tests/test_gpu_decode_parity.py tests/test_gpu_parity_sizes.py tests/test_gpu_decode_step.py -x -q > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/summary.txt; tail -6 $O/pytest_new.log | tee -a $O/summary.txt
python scripts/transcribe_bench.py small 600 20 > $O/transcribe_greedy.json 2>&1; tail -1 $O/transcribe_greedy.json | tee -a $O/summary.txt
python scripts/transcribe_bench.py small 600 1 ts > $O/transcribe_ts.json 2>&1; tail -1 $O/transcribe_ts.json | tee -a $O/summary.txt
