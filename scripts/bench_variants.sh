# same-box A/B of the whole step under forced ping-pong GEMM variants (OASR_PP_VARIANT)
export OASR_TESTING_HOOKS=1  # the OASR_PP_* switches are inert without it (csrc/common.h::oasr_experiment_env)
for v in 0 2 7 -1 0 -1; do
  if [ "$v" = "-1" ]; then unset OASR_PP_VARIANT; else export OASR_PP_VARIANT=$v; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
r=d['roofline']
print('variant', os.environ.get('OASR_PP_VARIANT','default'), 'ms/step', d['ms_per_step'], 'gemm_ms', r['gemm_ms_per_step'], ' | '.join(f\"{k.split('<')[1][:28]}:{v['avg_us']:.0f}us\" for k,v in r['by_symbol'].items() if 'pp_kernel' in k or 'fast' in k))"
done
