#!/bin/bash
# round 6, call 40: LayerNorm forward / backward with the wave sums on the DPP path (common.h wave_sum_dpp): parity tests, kernel times, the decode engines' bit-identity, a bench line
O=gpurun_out/r06ln
mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_decode_step.py tests/test_gpu_modules.py -m gpu -q --timeout 900 2>&1 | tail -3 | tee $O/tests.txt
python scripts/ln_bench.py 2>&1 | tail -12 | tee $O/ln_bench.txt
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print(j['ms_per_step'], j['value'], r['frac'], {k[:14]: (v.get('frac_of_8TBps'), v.get('us')) for k,v in r['hbm_kernels'].items()})" | tee $O/bench.txt
